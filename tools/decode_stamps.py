"""In-kernel timeline of the fused decode blocks (diagnostic build, never the product library).

`python tools/decode_stamps.py --build` (no GPU needed) builds km-bart_amd/lib/libkmbart_hip_dstamp.so with
-DKMB_DECODE_STAMP; on the GPU box `python tools/decode_stamps.py` runs one beam-5 generate of vcg_base (batch 64) with
that library and prints, for the LAST launch of each kernel type, the median time between the per-workgroup
s_memrealtime stamps (10 ns ticks) and the span from the first workgroup's entry to the last one's exit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "km-bart_amd")
LIB = os.path.join(PKG, "lib", "libkmbart_hip_dstamp.so")
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)

if "--build" in sys.argv:
    import build as b
    print(b.build_variant("dstamp", ["KMB_DECODE_STAMP"], sources=("decode.hip",)))
    sys.exit(0)

os.environ["KMB_LIB_PATH"] = LIB
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart import _lib  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).eval()
b = make_batch(64, seed=4321)
ids, am = b["input_ids"].to(dev), b["attention_mask"].to(dev)
feats = [f.to(dev) for f in b["image_features"]]
kw = dict(num_beams=5, num_return_sequences=1, max_length=20, early_stopping=True)
model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
lib = _lib.load()
lib.kmb_debug_set_decode_stamps.restype = C.c_int
lib.kmb_debug_set_decode_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros((4, 4096, 8), dtype=torch.int64, device=dev)
assert lib.kmb_debug_set_decode_stamps(C.c_void_p(stamps.data_ptr())) == 0
model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
lib.kmb_debug_set_decode_stamps(None)
s = stamps.cpu().numpy()
names = {0: ("projection K=768 (last launch: fc1)", ["weight loads issued", "rows staged (+LN), barrier", "MFMAs", "epilogue + store"]),
         1: ("projection K=3072 (fc2)", ["weight loads issued", "rows staged, barrier", "MFMAs", "epilogue + store"]),
         2: ("self-attention block", ["weights issued, rows staged, barrier", "MFMAs + q|k|v to LDS / cache", "barrier", "scores", "softmax + values", "store"]),
         3: ("cross-attention block", ["weights issued, rows staged, barrier", "MFMAs + q to LDS", "barrier", "scores", "softmax + values", "store"])}
for t, (name, phases) in names.items():
    v = s[t]
    v = v[v[:, 0] != 0]
    if not len(v):
        continue
    last = len(phases)
    print(f"{name}: {len(v)} workgroups, span {(v[:, last].max() - v[:, 0].min()) * 0.01:.2f} us, "
          f"entries spread over {(v[:, 0].max() - v[:, 0].min()) * 0.01:.2f} us")
    for i, ph in enumerate(phases):
        d = (v[:, i + 1] - v[:, i]) * 0.01
        print(f"   {ph:42s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    if t >= 2:
        d = (v[:, 7] - v[:, 0]) * 0.01
        print(f"   {'(of the first phase: until all loads issued)':42s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    d = (v[:, last] - v[:, 0]) * 0.01
    print(f"   {'whole workgroup':42s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
