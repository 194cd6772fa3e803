"""One serial training step whose GEMM launch list is dumped in launch order (for joining with rocprofv3 per-dispatch
counters): python tools/one_step_gemm_trace.py <batch> <out.txt>.  Run it under `rocprofv3 --pmc ... --kernel-trace` with
KMB_GEMM_TUNE_FILE preloaded so that the trace contains no tuning launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart import _lib  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

B = int(sys.argv[1])
out = sys.argv[2]
dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-5)
b = make_batch(B, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
lib = _lib.load()
lib.kmb_set_side_stream(model._engine.h, 0)
for _ in range(2):
    model.train_step_fwd_bwd(batch)
    opt.step()
torch.cuda.synchronize()
lib.kmb_profile_gemm(1)
model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()
_lib.check(lib.kmb_profile_dump(out.encode()))
lib.kmb_profile_gemm(0)
print("dumped", out)
