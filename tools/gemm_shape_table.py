"""Per-shape GEMM timing of one training step (HIP events around every launch): python tools/gemm_shape_table.py [batch]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
if os.environ.get("KMB_USE_DIAG") == "1":   # A/B runs (tools/gemm_ab_*.sh): the knobs live in the diagnostic build only (csrc/diag.h)
    _diag.use_diag_lib()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-5)
opt.allow_overlap(True)
b = make_batch(B, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
for _ in range(int(os.environ.get("KMB_TABLE_WARM_STEPS", "12"))):   # enough steps for the in-step refinement of every shape (12 launches of the once-per-step shapes)
    model.train_step_fwd_bwd(batch)
    opt.step()
torch.cuda.synchronize()
lib = _lib.load()
lib.kmb_set_side_stream(model._engine.h, 0)  # time the launches one at a time, not overlapped with each other
lib.kmb_profile_gemm(1)
model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()
path = os.path.join(ROOT, "gpurun_out", "gemm_shapes_b%d.txt" % B)
_lib.check(lib.kmb_profile_dump(path.encode()))
lib.kmb_profile_gemm(0)
agg = collections.OrderedDict()
for line in open(path):
    v, M, N, K, sp, act, us, *_ = line.split()
    key = (int(v), int(M), int(N), int(K), int(sp), int(act))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(us)
tot = sum(a[1] for a in agg.values())
print("variant(3=fwd,2=dgrad,0=wgrad) M N K split act | launches avg_us TFLOP/s share")
for (v, M, N, K, sp, act), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = 2.0 * M * N * K * n / (us * 1e-6) / 1e12
    print(f"{v} {M:6d} {N:6d} {K:6d} s{sp:<2d} a{act} | {n:3d} {us / n:9.1f} {tf:8.1f} {100 * us / tot:5.1f}%")
print("total GEMM ms", tot / 1e3)
