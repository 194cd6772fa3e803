"""Find the first intermediate buffer of backward that differs between passes (diagnostic for DESIGN.md section 5).
kmb_debug_trace makes kmb_backward checksum its intermediate buffers with tiny kernels on the same stream (no
synchronisation, so the timing that provokes the difference is kept)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/km-bart_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch, bench
from src.data.synthetic import make_batch
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = make_batch(B, seed=77)
d = {k: v.to(DEV) for k, v in b.items() if torch.is_tensor(v)}
d["image_features"] = [f.to(DEV) for f in b["image_features"]]
torch.manual_seed(3)
m = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(dict(bench.VCG_BASE, dropout=0.0))).to(DEV)
m.train()
eng = m._need_engine()
lib = eng.lib
traces = []
grads = []
names = [(n, p._kmb_range) for n, p in m.named_parameters()]
for i in range(6):
    out = m(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"],
            decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"], labels=d["labels"])
    lib.kmb_debug_trace(1)
    out[0].backward()
    torch.cuda.synchronize()
    lib.kmb_debug_trace_dump(b"/tmp/kmb_trace.txt")
    lib.kmb_debug_trace(0)
    traces.append([ln.split() for ln in open("/tmp/kmb_trace.txt")])
    grads.append(eng.grads.clone())
ref = traces[1]
for i in range(2, 6):
    diffs = [(a[0], a[1], a[2]) for a, c in zip(ref, traces[i]) if a[3] != c[3]]
    bad = [n for n, (o, k) in names if bool((grads[1][o:o + k] != grads[i][o:o + k]).any())]
    print(f"pass 1 vs pass {i}: {len(diffs)} of {len(ref)} traced buffers differ; first:", diffs[:6], f"| {len(bad)} parameters' gradients differ:", bad[:8])
