"""Shared helpers for the committed golden vectors.  TEST INFRASTRUCTURE ONLY (see kmbart_oracle.py).

Weights and region features of the golden cases are defined by a closed-form integer
generator (exact on every machine), so the fixtures under tests/golden/ only need to hold
the small inputs and the expected outputs.
"""
import numpy as np
import torch

from .kmbart_oracle import OracleConfig, head_param_names, param_names, param_shape

# special ids of the tiny vocabulary: same ordering as src/data/tokenization.py:36-57,
# re-based from 50265 to 480 so they fit V=512
TINY_SPECIAL_BASE = 480
TINY = dict(
    vocab_size=512, d_model=128, encoder_layers=2, decoder_layers=2,
    encoder_attention_heads=2, decoder_attention_heads=2,
    encoder_ffn_dim=256, decoder_ffn_dim=256, max_position_embeddings=64,
    image_feature_size=2052, img_feat_id=TINY_SPECIAL_BASE + 8, cls_token_id=TINY_SPECIAL_BASE + 11,
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, init_std=0.02,
)


def tiny_config(**over):
    d = dict(TINY)
    d.update(over)
    return OracleConfig.from_dict(d)


def lcg_uniform(n, seed):
    """n floats in [-1, 1): splitmix64 of (index, seed); pure uint64 arithmetic -> bit-exact anywhere."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    top = (x >> np.uint64(40)).astype(np.float64)  # 24 bits
    return (top / float(1 << 23) - 1.0).astype(np.float32)


def golden_state_dict(cfg, seed=7):
    """Every matrix ~ U(-a, a) with std init_std; biases and LayerNorm parameters are perturbed
    too (non-trivial values exercise the bias / gamma / beta paths); pad rows zero."""
    sd = {}
    a = cfg.init_std * (3.0 ** 0.5)
    for k, n in enumerate(param_names(cfg) + head_param_names(cfg)):
        shp = param_shape(cfg, n)
        u = torch.from_numpy(lcg_uniform(int(np.prod(shp)), seed * 1000 + k)).view(shp)
        if "layer_norm" in n or "layernorm" in n:
            t = 1.0 + 0.1 * u if n.endswith("weight") else 0.05 * u
        elif n.endswith("bias"):
            t = 0.02 * u
        else:
            t = a * u
            if n == "model.shared.weight" or n.endswith("embed_positions.weight"):
                t[cfg.pad_token_id].zero_()
        sd[n] = t.contiguous()
    sd["final_logits_bias"] = 0.01 * torch.from_numpy(lcg_uniform(cfg.vocab_size, seed * 1000 + 999)).view(1, -1)
    return sd


def golden_features(regions, seed=11, feat_dim=2048):
    out = []
    for i, r in enumerate(regions):
        if r == 0:
            out.append(torch.empty(0))
            continue
        f = torch.from_numpy(lcg_uniform(r * feat_dim, seed + 17 * i)).view(r, feat_dim).abs() * 2.0
        u = torch.from_numpy(lcg_uniform(r * 4, seed + 17 * i + 5)).view(r, 4) * 0.5 + 0.5  # [0,1)
        xy = u[:, :2] * 500.0
        wh = u[:, 2:] * 484.0 + 16.0
        out.append(torch.cat([f, xy, xy + wh], dim=1).contiguous())
    return out


def trained_state_dict():
    """The tiny model trained on the reverse-copy task (oracle/make_golden.py step 3), stored as fp16: generation
    fixtures that are not degenerate.  fp16 -> fp32 is exact, so the oracle and the engine load identical weights."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "tiny_trained_fp16.npz")
    z = np.load(path)
    return {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}


class IdTokenizer:
    """Stand-in for BartTokenizer.decode in the generate_text fixtures (the BART vocabulary files are absent offline): the
    text of a sequence is its ids in decimal, special ids (< 3: <s>, <pad>, </s>) dropped when skip_special_tokens."""

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(t)) for t in ids if not (skip_special_tokens and int(t) < 3))
