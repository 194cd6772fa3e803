"""How much of the in-step GEMM time is epilogue?  Three real training steps fill every buffer with real values, then the
same step is timed per GEMM launch (HIP events, weight gradients on the caller's stream) twice: as it is, and with EVERY
GEMM epilogue skipped (KMB_GEMM_ABLATE: accumulators kept alive, nothing staged, computed or stored; the other kernels then
work on the previous step's buffers -- realistic values, timing only).  The difference is the upper bound of what hiding
epilogues under the next tile's K loop could buy inside a step.

    KMB_GEMM_ABLATE_DYNAMIC=1 python tools/gemm_epilogue_bound.py [batch]
"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402
_diag.use_diag_lib()   # the A/B knobs live in the diagnostic build only (csrc/diag.h)

os.environ["KMB_GEMM_ABLATE_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart import _lib  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
b = make_batch(B, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
lib = _lib.load()
lib.kmb_set_side_stream(model._engine.h, 0)
for _ in range(3):
    model.train_step_fwd_bwd(batch)      # no optimizer step: weights stay as they are
torch.cuda.synchronize()


def table(tag):
    path = os.path.join(ROOT, "gpurun_out", "epi_bound_%s_b%d.txt" % (tag, B))
    tot = collections.OrderedDict()
    for rep in range(2):
        lib.kmb_profile_gemm(1)
        model.train_step_fwd_bwd(batch)
        torch.cuda.synchronize()
        _lib.check(lib.kmb_profile_dump(path.encode()))
        lib.kmb_profile_gemm(0)
        for line in open(path):
            v, M, N, K, sp, act, us, *_ = line.split()
            a = tot.setdefault((int(v), int(M), int(N), int(K), int(sp), int(act)), [0, 0.0])
            a[0] += 1
            a[1] += float(us)
    return tot


full = table("full")
os.environ["KMB_GEMM_ABLATE"] = "1"
model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()
bare = table("noepi")
os.environ["KMB_GEMM_ABLATE"] = "0"
print("layout M N K split act | n/step | with epilogue us | without us | epilogue share")
tf = tb = 0.0
for k, (n, us) in sorted(full.items(), key=lambda kv: -kv[1][1]):
    ub = bare[k][1]
    tf += us
    tb += ub
    print("%d %6d %6d %6d s%-2d a%d | %3d | %9.1f | %9.1f | %5.1f %%" % (*k, n // 2, us / n, ub / n, 100 * (1 - ub / us)))
print("total GEMM ms per step: with epilogues %.3f, without %.3f  -> epilogues are %.1f %% of the in-step GEMM time"
      % (tf / 2e3, tb / 2e3, 100 * (1 - tb / tf)))
