"""How much of the AdamW update a small-batch step hides: step time (20 steps, no host synchronisation inside) with the
optimizer's per-bucket updates overlapped with backward (product), issued after backward (allow_overlap(False)) and left out
(timing only), and the host's enqueue time of the two halves of a step.  python tools/adamw_overlap_time.py [64 256 ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart.data import PackedFeatures  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-5)


def timed(fn, n=20, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (int(a) for a in (sys.argv[1:] or ["64", "256"])):
    b = make_batch(B, seed=1)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["image_features"] = PackedFeatures.from_list(b["image_features"], 2052).to(dev)

    def both():
        model.train_step_fwd_bwd(batch)
        opt.step()

    def no_opt():
        model.train_step_fwd_bwd(batch)

    res = {}
    for rnd in range(2):
        opt.allow_overlap(True)
        res.setdefault("overlapped", []).append(timed(both))
        opt.allow_overlap(False)
        res.setdefault("after backward", []).append(timed(both))
        res.setdefault("no optimizer", []).append(timed(no_opt))
    opt.allow_overlap(True)
    for _ in range(4):
        both()
    fb, st = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_step_fwd_bwd(batch)
        t1 = time.perf_counter()
        opt.step()
        t2 = time.perf_counter()
        fb.append(t1 - t0)
        st.append(t2 - t1)
    print("batch %4d: step ms %s | host enqueue: forward + backward %.2f ms, optimizer.step %.2f ms" % (
        B, ", ".join("%s %s" % (k, "/".join("%.3f" % x for x in v)) for k, v in res.items()),
        sorted(fb)[5] * 1e3, sorted(st)[5] * 1e3), flush=True)
