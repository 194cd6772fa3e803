"""EXPERIMENT test (not collected by the product suite): the resident decoder-layers kernel (tools/experiments/decode_resident.hip) against the
six-launch blocks, bit for bit.  Build the experiment library on the build container, then run on a GPU box:
    python km-bart_amd/build.py --variant resident
    gpurun -- 'KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_resident.so python -m pytest tools/experiments/test_decode_resident.py -q'
(Moved out of tests/test_decode_fused_gpu.py in round 6 together with the kernel: measured 17 % slower than the blocks)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("KMB_LIB_PATH", os.path.join(ROOT, "km-bart_amd", "lib", "libkmbart_hip_resident.so"))
from test_decode_fused_gpu import *  # noqa: E402,F401,F403  (BASE, DEV, make_batch, the model classes)


def test_resident_decoder_layers_kernel_is_bit_identical_to_the_blocks():
    """The resident decoder-layers kernel (csrc/decode.hip, KMB_GEN_FUSED=2: all layers of a decode step in one launch, twelve
    co-resident workgroups per row tile behind counter barriers, sc1 hand-offs) repeats the arithmetic of the six-launch blocks:
    teacher-forced logits of every step are BIT-identical for every layers-per-launch grouping, the status word stays clean, and
    a beam-5 generate returns the same ids.  (It is opt-in: measured slower than the blocks)"""
    torch.manual_seed(0)
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE)).to(DEV).eval()
    with torch.no_grad():
        model._engine.view(model._engine.params, "model.shared.weight").mul_(8.0)
    model._engine.sync_params()
    eng = model._engine
    B, nb, T = 7, 5, 6          # 35 rows: three row tiles, the last one ragged (3 rows), items straddling tiles
    b = make_batch(B, seed=77, regions=(36, 20, 36, 7, 12, 36, 30), event_lens=(23, 7, 15, 23, 9, 20, 4), label_lens=(32,) * B)
    ids, am = b["input_ids"].to(DEV), b["attention_mask"].to(DEV)
    feats = [f.to(DEV) for f in b["image_features"]]
    g = torch.Generator().manual_seed(7)
    toks = torch.randint(3, 50000, (T, B * nb), generator=g).to(DEV)

    def run(mode, layers=None):
        os.environ["KMB_GEN_FUSED"] = mode
        if layers:
            os.environ["KMB_GEN_LAYERS"] = str(layers)
        try:
            out = []
            eng.gen_begin(ids, feats, am, nb, 12)
            for t in range(T):
                out.append(eng.gen_step(toks[t], t)[:, : model.config.vocab_size].clone())
                eng.gen_reorder(torch.randperm(B * nb, generator=torch.Generator().manual_seed(t)).to(DEV, torch.int32)
                                if t == 2 else torch.arange(B * nb, dtype=torch.int32, device=DEV), t)
            torch.cuda.synchronize()
            return torch.stack(out), eng.read_status()
        finally:
            os.environ.pop("KMB_GEN_FUSED", None)
            os.environ.pop("KMB_GEN_LAYERS", None)

    ref, st = run("1")
    assert st == 0 and bool(torch.isfinite(ref).all())
    for layers in (6, 2, 1):
        got, st = run("2", layers)
        assert st == 0, "a group barrier gave up (status %d)" % st
        assert torch.equal(got, ref), "resident kernel, %d layers per launch: logits differ from the six-launch blocks" % layers
    kw = dict(input_ids=ids, image_features=feats, attention_mask=am, num_beams=nb, max_length=10, early_stopping=True)
    want = model.generate(**kw)
    os.environ["KMB_GEN_FUSED"] = "2"
    try:
        got = model.generate(**kw)
    finally:
        os.environ.pop("KMB_GEN_FUSED", None)
    assert torch.equal(got, want)


