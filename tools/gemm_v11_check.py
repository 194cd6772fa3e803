"""The four-stage 128x128 variant 5 (launches of at most 256 workgroups) and the persistent GEMMs (variant 11: 256x256 tiles, 12: 256x128, 13: 256x192, 14 / 15: 256x256 / 256x192 with eight waves) against the 128x128 variant 7 on every epilogue class and operand layout:
outputs must be bit-identical (same per-element accumulation order), column sums equal after folding their partial
rows.  Also prints the time of each variant per case.  Re-runs itself once per variant (the variant is a per-process
environment choice).

    python tools/gemm_v11_check.py            # compare 7 and 11 (and time 8)
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    # name, M, N, K, a_kc, b_kc, options
    ("fwd_bias", 8192, 2304, 768, True, True, dict(bias=True, qscale=True)),
    ("fwd_gelu_preact", 8192, 3072, 768, True, True, dict(bias=True, act=1, preact=True)),
    ("fwd_res_drop", 32768, 768, 3072, True, True, dict(bias=True, residual=True, drop=0.1)),
    ("dgrad_gelugrad_cs", 8192, 3072, 768, True, False, dict(act=2, aux=True, colsum=True)),
    ("dgrad_gelugrad", 8192, 3072, 768, True, False, dict(act=2, aux=True)),
    ("dgrad_cs", 16384, 1536, 768, True, False, dict(colsum=True)),
    ("dgrad_res_cs", 16384, 1536, 768, True, False, dict(colsum=True, residual=True)),
    ("wgrad_layout_edges", 20000, 1000, 256, False, False, dict(f32=True)),
    ("lm_head_f32_edges", 8192, 50320, 768, True, True, dict(bias=True, f32=True, ld_f32=50432)),
    ("fwd_edges_rows", 16300, 1024, 512, True, True, dict(bias=True, residual=True)),
    ("tanh_generic", 8192, 2048, 768, True, True, dict(bias=True, act=3)),
    ("k128", 8192, 2048, 128, True, True, dict(bias=True)),
    ("tiles192", 16384, 768, 768, True, True, dict(bias=True, residual=True, drop=0.1)),   # fewer tiles than CUs
    ("tiles200_edges", 12800, 1000, 256, True, False, dict(residual=True)),               # 50 x 4 = 200 tiles, edge columns
    ("tiles147", 12500, 768, 320, True, True, dict(bias=True)),                           # 49 x 3 = 147 tiles -> grid 144
    # the N = 768 shapes the 256x192 tile exists for (4 column tiles: 256 / 512 tiles at M = 16384 / 32768)
    ("n768_fc2_fwd", 16384, 768, 3072, True, True, dict(bias=True, residual=True, drop=0.1)),
    ("n768_fc1_dgrad", 16384, 768, 3072, True, False, dict(residual=True)),
    ("n768_qkv_dgrad", 32768, 768, 2304, True, False, dict(residual=True)),
    ("n768_out_fwd", 32768, 768, 768, True, True, dict(bias=True, residual=True, drop=0.1)),
    ("n768_out_dgrad", 16384, 768, 768, True, False, dict()),
    ("n2304_qkv_fwd", 16384, 2304, 768, True, True, dict(bias=True, qscale=True)),
    ("n192_edges", 16300, 1000, 320, True, False, dict(colsum=True)),                     # 64 x 6 tiles of 192, edge rows + columns
    # split-K (raw fp32 slabs): the (tile, slice) enumeration (slice-minor / per-XCD ranges / slice-major) must not matter
    ("wgrad_split7", 3072, 768, 8192, False, False, dict(split=7)),
    ("wgrad_split14_edges", 700, 760, 4096, False, False, dict(split=14)),
    ("fwd_split3", 2048, 768, 3072, True, True, dict(split=3)),
    # at most 256 workgroups (the reference's default batch of 64): the four-stage variant 5 takes these, every other variant number
    # falls back to what it is eligible for
    ("small_fc2_fwd", 4096, 768, 3072, True, True, dict(bias=True, residual=True, drop=0.1)),
    ("small_out_dgrad", 4096, 768, 768, True, False, dict()),
    ("small_qkv_fwd", 1024, 2304, 768, True, True, dict(bias=True, qscale=True)),
    ("small_fc1_gelu", 1024, 3072, 768, True, True, dict(bias=True, act=1, preact=True)),
    ("small_fc1_dgrad_cs", 1024, 3072, 768, True, False, dict(act=2, aux=True, colsum=True)),
    ("small_wgrad", 768, 768, 4096, False, False, dict(f32=True)),
    ("small_wgrad_split7", 768, 768, 4096, False, False, dict(split=7)),
    ("small_edges_k320", 1000, 520, 320, True, False, dict(residual=True)),
    ("small_k256", 2048, 768, 256, True, True, dict(bias=True)),
    ("small_k192", 2048, 768, 192, True, True, dict(bias=True)),
    # benchmark-batch shapes of the classes whose epilogue is long (timing of the eight-wave variant 14 against 11 / 13 / 8)
    ("b1024_fc1_gelu", 32768, 3072, 768, True, True, dict(bias=True, act=1, preact=True)),
    ("b1024_fc1_dgrad", 32768, 3072, 768, True, False, dict(act=2, aux=True, colsum=True)),
    ("b1024_qkv_fwd", 32768, 2304, 768, True, True, dict(bias=True, qscale=True)),
    ("b1024_fc2_fwd", 32768, 768, 3072, True, True, dict(bias=True, residual=True, drop=0.1)),
    ("b1024_fc2_dgrad", 32768, 768, 3072, True, False, dict(residual=True)),
]


def run_variant():
    import torch
    from gpu_util import DEV, bf, gemm
    from kmbart import _lib
    _lib.load().kmb_gemm_shared_device(int(os.environ.get("KMB_V11_SHARED", "0")))
    out = {}
    for ci, (name, M, N, K, akc, bkc, o) in enumerate(CASES):
        g = torch.Generator(device=DEV).manual_seed(100 + ci)
        rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
        A = bf(rn(M, K) if akc else rn(K, ((M + 7) // 8) * 8)) * 0.5
        B = bf(rn(N, K) if bkc else rn(K, ((N + 7) // 8) * 8)) * 0.5
        kw = dict(a_kc=akc, b_kc=bkc, M=M, N=N, K=K)
        if o.get("bias"):
            kw["bias"] = rn(N)
        if o.get("qscale"):
            kw["col_scale"], kw["col_scale_n"] = 0.125, 768
        kw["act"] = o.get("act", 0)
        Np = ((N + 7) // 8) * 8
        if o.get("preact"):
            kw["preact"] = torch.zeros(M, Np, dtype=torch.bfloat16, device=DEV)
        if o.get("aux"):
            kw["aux"] = bf(rn(M, Np))
        if o.get("residual"):
            kw["residual"] = bf(rn(M, Np))
        if o.get("drop"):
            kw["drop_p"], kw["drop_seed"] = o["drop"], 1234
        nparts = (M + 63) // 64
        if o.get("colsum"):
            kw["colsum"] = torch.full((nparts, N), 7.0, dtype=torch.float32, device=DEV)
        if o.get("split"):
            kw["split_k"] = o["split"]
            kw["slab"] = torch.zeros(o["split"] * M * Np, dtype=torch.float32, device=DEV)
        elif o.get("f32"):
            ld = o.get("ld_f32", ((N + 3) // 4) * 4)
            kw["out_f32"] = torch.zeros(M, ld, dtype=torch.float32, device=DEV)
        else:
            kw["out_bf16"] = torch.zeros(M, Np, dtype=torch.bfloat16, device=DEV)
        gemm(A, B, **kw)
        torch.cuda.synchronize()
        rec = {}
        for k in ("out_bf16", "out_f32", "preact", "slab"):
            if kw.get(k) is not None:
                rec[k] = hashlib.md5(kw[k].cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
        if kw.get("colsum") is not None:
            rec["colsum"] = kw["colsum"].sum(0).cpu().tolist()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gemm(A, B, **kw)
        e1.record()
        torch.cuda.synchronize()
        rec["us"] = e0.elapsed_time(e1) * 100.0
        # a relaunch into cleared outputs must reproduce the first launch (the persistent variants' tile counters
        # have to come back to zero after every launch)
        for k in ("out_bf16", "out_f32", "preact", "slab"):
            if kw.get(k) is not None:
                kw[k].zero_()
        gemm(A, B, **kw)
        torch.cuda.synchronize()
        for k in ("out_bf16", "out_f32", "preact", "slab"):
            if kw.get(k) is not None:
                again = hashlib.md5(kw[k].cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
                if again != rec[k]:
                    rec[k] = "relaunch-differs-" + again
        rec["tflops"] = 2.0 * M * N * K / rec["us"] * 1e-6
        out[name] = rec
        del A, B, kw
    print("JSON" + json.dumps(out))


VARIANTS = ("7", "8", "11", "12", "13", "14", "15", "11o1", "12o1", "13o1", "14o1", "15o1", "11s", "12o1s", "13s", "14s", "15o1s", "11o1p",
            "13p", "14o1p", "15p", "7o5", "8o5", "5", "5o1", "5o5")


def main():
    global VARIANTS
    if os.environ.get("KMB_V11_CHILD"):
        return run_variant()
    if os.environ.get("KMB_V11_CHECK_VARIANTS"):   # a subset, e.g. "7,8,11,14,14o1" ("7" and "8" are the references: keep them)
        VARIANTS = tuple(os.environ["KMB_V11_CHECK_VARIANTS"].split(","))
    res = {}
    # o1: tile_order bit 0 = per-XCD contiguous tile ranges; o5: that plus slice-major split-K enumeration (bit 2); s: shared-device mode (every tile from the atomic counter);
    # p: the activation-panel L2 prefetch on for every shape (the others: off, so both paths of the kernel are compared)
    for v in VARIANTS:
        env = dict(os.environ, KMB_GEMM_VARIANT=v.rstrip("sp").split("o")[0], KMB_V11_CHILD="1",
                   KMB_TILE_ORDER=v.rstrip("sp").split("o")[1] if "o" in v else "0", KMB_V11_SHARED="1" if v.endswith("s") else "0",
                   KMB_GEMM_PREFETCH="1" if v.endswith("p") else "0")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        if r.returncode != 0 or not line:
            print("variant", v, "failed:\n", r.stdout[-2000:], r.stderr[-3000:])
            sys.exit(1)
        res[v] = json.loads(line[0][4:])
    bad = 0
    for name, *_ in CASES:
        a, c = res["7"][name], res["8"][name]
        line = f"{name:20s} v7 {a['us']:7.1f} us {a['tflops']:5.0f} TF | v8 {c['us']:7.1f} us {c['tflops']:5.0f} TF"
        for v in VARIANTS[1:]:
            b = res[v][name]
            ok = all(a[k] == b[k] for k in a if k not in ("us", "tflops", "colsum"))
            d = 0.0
            if "colsum" in a:
                d = max(abs(x - y) / (abs(x) + 1.0) for x, y in zip(a["colsum"], b["colsum"]))
                ok = ok and d < 1e-3
            bad += not ok
            line += f" | v{v} {b['us']:7.1f} us {'same bits' if ok else 'MISMATCH'}"
        print(line, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
