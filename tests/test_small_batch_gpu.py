"""GPU: the small-batch regime (the reference's default per-GPU batch is 64, vcg_train.py:330).  With M = 1024-4096 rows
the forward / data-gradient GEMMs with N = 768 split their K loop over workgroups and one pass sums the slabs and applies
the linear layer's epilogue (bias, q-scale, dropout, residual: engine.cpp run_gemm, norm.hip reduce_slabs_epi_kernel).

  * vcg_base b = 16 (1024 encoder / 512 decoder rows) against oracle autograd: loss and every gradient;
  * with dropout ON the split path must draw the SAME masks as the un-split GEMM epilogue (and as the LayerNorm backward
    that recomputes them): same seed, KMB_SMALL_SPLIT=0 vs 1 in two processes, loss and gradients must agree."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import goldenlib as G  # noqa: E402
from oracle import kmbart_oracle as O  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402
from test_fullsize_parity_gpu import BASE, DEV, check_grads, to_dev  # noqa: E402


def test_vcg_base_b16_every_gradient():
    ocfg = O.OracleConfig.from_dict(BASE)
    sd = G.golden_state_dict(ocfg, seed=5)
    b = make_batch(16, seed=4321)
    osd = {k: v.clone().requires_grad_(k != "final_logits_bias") for k, v in sd.items()}
    ref_loss, _, _ = O.forward(osd, ocfg, b["input_ids"], b["image_features"], b["attention_mask"],
                               b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"])
    ref_loss.backward()
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(BASE))
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    d = to_dev(b)
    loss = model(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"],
                 decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"],
                 labels=d["labels"])[0]
    assert abs(float(loss) - float(ref_loss)) / float(ref_loss) < 1e-3
    loss.backward()
    torch.cuda.synchronize()
    check_grads(model, {k: v.grad for k, v in osd.items() if v.grad is not None}, "vcg_base b=16 (split-K small-batch path)")


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "km-bart_amd"))
import torch
from src.data.synthetic import make_batch
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
from tests.test_fullsize_parity_gpu import BASE
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(dict(BASE, dropout=0.1))).to("cuda:0").train()
model._engine.set_seed(77)
b = make_batch(16, seed=99)
batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to("cuda:0") for f in b["image_features"]]
loss = model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()
g = model._engine.grads
names = ["model.encoder.layers.0.fc2.weight", "model.decoder.layers.5.self_attn.out_proj.weight", "model.encoder.layers.3.self_attn.q_proj.bias"]
out = {"loss": float(loss)}
for n in names:
    o, r, c = model._engine.index[n]
    out[n] = g[o: o + r * c].double().cpu().numpy().tolist()[:4096]
print("JSON" + json.dumps(out))
"""


def test_split_path_draws_the_same_dropout_masks():
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, KMB_SMALL_SPLIT=flag)
        r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600,
                           cwd=ROOT)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        assert r.returncode == 0 and line, r.stderr[-3000:]
        res[flag] = json.loads(line[0][4:])
    a, b = res["0"], res["1"]
    assert abs(a["loss"] - b["loss"]) <= 2e-3 * abs(a["loss"]), (a["loss"], b["loss"])
    for k in a:
        if k == "loss":
            continue
        x, y = torch.tensor(a[k]), torch.tensor(b[k])
        # a different mask at p = 0.1 would move these gradients by tens of percent
        assert float((x - y).norm() / x.norm()) < 3e-2, k
