"""Run-to-run determinism of one backward pass: the same model and batch twice, which gradients differ bit-wise."""
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
gs = []
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("KMB_PROBE_USER_STREAM") else contextlib.nullcontext()
for _ in range(4):
  with ctx:
    out = m(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"],
            decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"], labels=d["labels"])
    out[0].backward()
    torch.cuda.synchronize()
    gs.append((float(out[0]), eng.grads.clone()))
    torch.cuda.synchronize()
names = [(n, p._kmb_range) for n, p in m.named_parameters()]
if os.environ.get("KMB_PRINT_OFFSETS"):
    for n, (o, k) in names:
        if "decoder.layers.5" in n and "weight" in n and "norm" not in n: print("OFF", n, o / 1e6, (o + k) / 1e6)
for i in (2, 3):
    a, c = gs[1][1], gs[i][1]
    bad = [n for n, (o, k) in names if bool((a[o:o + k] != c[o:o + k]).any())]
    worst = sorted(((float((a[o:o + k] - c[o:o + k]).norm() / (a[o:o + k].norm() + 1e-30)), n) for n, (o, k) in names if 'k_proj.bias' not in n), reverse=True)
    print('   median relative difference', f'{worst[len(worst)//2][0]:.1e}')
    worst = worst[:6]
    print("   largest relative differences:", [(f"{e:.1e}", n) for e, n in worst])
    print("   differing:", bad if len(bad) <= 40 else bad[:40] + ["..."])
    print(f"batch {B}: pass 1 vs pass {i}: loss {gs[1][0]!r} vs {gs[i][0]!r}; parameters with differing gradients: {len(bad)} of {len(names)}",
          bad[:6])
