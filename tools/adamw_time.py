"""Time the fused AdamW kernel stand-alone on arena ranges of the sizes the overlapped optimizer step uses (one bucket per layer ... the tied matrix).
python tools/adamw_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
eng = model._engine
for i, (off, cnt) in enumerate(eng.buckets()):
    for _ in range(3):
        eng.adamw_step(1e-5, offset=off, count=cnt, bump=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.adamw_step(1e-5, offset=off, count=cnt, bump=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    print(f"bucket {i:2d}: {cnt / 1e6:7.2f} M parameters  {us:7.1f} us  {cnt * 30 / us * 1e-6:.2f} TB/s (30 B per parameter)")
