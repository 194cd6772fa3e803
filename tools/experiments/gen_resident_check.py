"""Resident decoder-layers kernel (tools/experiments/decode_resident.hip, round 5; experiment library `python km-bart_amd/build.py --variant resident`) against the six-launches-per-layer blocks (KMB_GEN_FUSED=1), whose
arithmetic it repeats: teacher-forced logits of every decode step must be bit-identical; then the time of a beam-5 generate
(batch 64, 20 tokens) on each path, KMB_GEN_LAYERS = layers per launch.

    python tools/experiments/gen_resident_check.py [batch=64] [beams=5]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("KMB_LIB_PATH", os.path.join(ROOT, "km-bart_amd", "lib", "libkmbart_hip_resident.so"))   # the experiment library
import torch  # noqa: E402

from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

sys.argv += ["64", "5"]
B, NB = int(sys.argv[1]), int(sys.argv[2])
DEV = torch.device("cuda", 0)
from bench import VCG_BASE  # noqa: E402

torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(VCG_BASE)).to(DEV).eval()
with torch.no_grad():
    model._engine.view(model._engine.params, "model.shared.weight").mul_(8.0)
model._engine.sync_params()
b = make_batch(B, seed=4321)
kw = dict(input_ids=b["input_ids"].to(DEV), image_features=[f.to(DEV) for f in b["image_features"]], attention_mask=b["attention_mask"].to(DEV))
eng = model._engine


def logits(mode, layers=None, T=8):
    os.environ["KMB_GEN_FUSED"] = mode
    if layers:
        os.environ["KMB_GEN_LAYERS"] = str(layers)
    else:
        os.environ.pop("KMB_GEN_LAYERS", None)
    out = []
    g = torch.Generator().manual_seed(7)
    toks = torch.randint(3, 50000, (T, B * NB), generator=g).to(DEV)
    eng.gen_begin(kw["input_ids"], kw["image_features"], kw["attention_mask"], NB, 20)
    for t in range(T):
        out.append(eng.gen_step(toks[t], t)[:, :50320].clone())
        eng.gen_reorder(torch.arange(B * NB, dtype=torch.int32, device=DEV), t)
    torch.cuda.synchronize()
    st = eng.read_status() if hasattr(eng, "read_status") else None
    return torch.stack(out), st


ref, _ = logits("1")
ok = True
for layers in (1, 2, 3, 6):
    got, st = logits("2", layers)
    same = torch.equal(got, ref)
    d = float((got - ref).abs().max())
    nan = bool(torch.isnan(got).any())
    print("resident, %d layer(s) per launch: bit-identical to the six-launch blocks: %s (max |diff| %.3e, nan %s, status %s)" % (layers, same, d, nan, st))
    ok &= same
plain, _ = logits("0")
print("launch-per-operation path vs blocks: max |diff| %.3e" % float((plain - ref).abs().max()))


def timed(mode, layers=None, n=5):
    os.environ["KMB_GEN_FUSED"] = mode
    if layers:
        os.environ["KMB_GEN_LAYERS"] = str(layers)
    else:
        os.environ.pop("KMB_GEN_LAYERS", None)
    g = dict(kw, num_beams=NB, num_return_sequences=1, max_length=20, early_stopping=True)
    o = model.generate(**g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = model.generate(**g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, o


t1, o1 = timed("1")
print("generate, six-launch blocks: %.2f ms" % t1)
for layers in (1, 2, 3, 6):
    t2, o2 = timed("2", layers)
    print("generate, resident %d layer(s)/launch: %.2f ms, ids equal: %s" % (layers, t2, torch.equal(o1, o2)))
sys.exit(0 if ok else 1)
