"""Host-side cost of one training step: wall time to ENQUEUE forward + backward + AdamW (no synchronisation inside the
timed region) against the step's GPU time, per batch size.  Small batches are bound by the former."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-5)
opt.allow_overlap(True)
for B in (int(a) for a in (sys.argv[1:] or ["32", "64", "128", "256", "512"])):
    b = make_batch(B, seed=1)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["image_features"] = [f.to(dev) for f in b["image_features"]]
    for _ in range(4):
        model.train_step_fwd_bwd(batch)
        opt.step()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_step_fwd_bwd(batch)
        opt.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
    e, t = sorted(enq)[5] * 1e3, sorted(tot)[5] * 1e3
    print(f"batch {B:4d}: enqueue {e:6.2f} ms, step (enqueue + drain) {t:6.2f} ms -> {B * 96 / t * 1e3:9.0f} tokens/s", flush=True)
