"""Generation throughput (BASELINE.json config 5): vcg_base, beam search with num_beams=5, KV-cached decoder steps,
cross-attention K/V computed once per batch item.  Prints one JSON line.

    python tools/gen_bench.py [--batch 64] [--beams 5] [--max-length 20] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--beams", type=int, default=5)
ap.add_argument("--max-length", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).eval()
b = make_batch(args.batch, seed=4321)
ids, am = b["input_ids"].to(dev), b["attention_mask"].to(dev)
feats = [f.to(dev) for f in b["image_features"]]
kw = dict(num_beams=args.beams, num_return_sequences=1, max_length=args.max_length, early_stopping=True)
out = model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    out = model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
steps = out.shape[1] - 1
print(json.dumps({"metric": "generate_sequences_per_sec", "value": round(args.batch / dt, 1), "unit": "sequences/s",
                  "decoder_steps_per_sec": round(steps / dt, 1), "ms_per_generate": round(dt * 1e3, 2),
                  "config": {"workload": "vcg_base generate, beam search", "batch": args.batch, "num_beams": args.beams,
                             "max_length": args.max_length, "decoder_steps": steps}, "dtype": "bf16",
                  "data": "synthetic", "n_gpus": 1}))
