"""In-kernel timeline of the resident decoder-layers kernel (tools/experiments/decode_resident.hip; diagnostic build -DKMB_DECODE_STAMP, never the product
library): per-workgroup s_memrealtime stamps (10 ns ticks) at the phase boundaries of layer KMB_DL_STAMP_LAYER (default 1: a layer
whose first weights were prefetched) of the last decode step of one beam-5 generate (batch 64).

    python tools/experiments/decode_resident_stamps.py --build      (no GPU needed)
    python tools/experiments/decode_resident_stamps.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "km-bart_amd")
LIB = os.path.join(PKG, "lib", "libkmbart_hip_dstampres.so")
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    import build as b
    print(b.build_variant("dstampres", ["KMB_DECODE_STAMP", "KMB_WITH_RESIDENT_DECODE"], sources=("decode.hip", "engine.cpp"), extra_sources=("decode_resident.hip",)))
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
stamps = torch.zeros((4096, 32), dtype=torch.int64, device=dev)
assert lib.kmb_debug_set_decode_stamps(C.c_void_p(stamps.data_ptr())) == 0
model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
lib.kmb_debug_set_decode_stamps(None)
v = stamps.cpu().numpy()
v = v[v[:, 0] != 0]
names = ["wait: previous layer's last barrier", "P1 rows (sc1) + LayerNorm", "P1 q|k|v MFMAs + cache append", "P1 attention", "P1 publish (store, drain, signal)",
         "P2 weights issued + wait barrier", "P2 o rows staged", "P2 MFMA + epilogue", "P2 publish",
         "P3 weights + cross K/V issued + wait", "P3 z rows + staging + LayerNorm", "P3 q MFMA + attention (matrix cores)", "P3 publish",
         "P4 weights + wait", "P4 rows + MFMA + epilogue", "P4 publish",
         "P5 weights (half) + wait", "P5 rows + LayerNorm", "P5 rest of weights + MFMAs + GeLU", "P5 publish",
         "P6 weights + wait", "P6 hidden rows staged", "P6 MFMAs + epilogue", "P6 publish"]
print("%d workgroups; layer span %.2f us (first entry to last exit), per-workgroup median %.2f us" %
      (len(v), (v[:, 24].max() - v[:, 0].min()) * 0.01, np.median(v[:, 24] - v[:, 0]) * 0.01))
for i, n in enumerate(names):
    d = (v[:, i + 1] - v[:, i]) * 0.01
    print("  %-48s median %6.2f  p10 %6.2f  p90 %6.2f us" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
for a_, b_, n in ((1, 25, "P1: rows issued"), (25, 26, "P1: drain (prefetched weights + rows landed)"), (26, 2, "P1: LayerNorm + LDS")):
    d = (v[:, b_] - v[:, a_]) * 0.01
    print("  %-48s median %6.2f  p10 %6.2f  p90 %6.2f us" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
