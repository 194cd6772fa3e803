"""(Round 5: the role-split kernel left the product library; build it with `python km-bart_amd/build.py --variant rolesplit` and run
this tool with KMB_LIB_PATH=km-bart_amd/lib/libkmbart_hip_rolesplit.so -- in the product library KMB_GEMM_VARIANT=10 means 11.)
Times the role-split GEMM (variant 10) against the tuner-free persistent variants on the benchmark-batch shapes, one
process per variant (KMB_GEMM_VARIANT is read once); KMB_LIB_PATH selects an experiment build (-DKMB_RS_NODMA / -DKMB_RS_NOEPI:
timing only, results are garbage).

    python tools/rs_time.py [variants=10,11,13]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [  # name, M, N, K, b_kc, options
    ("qkv_fwd", 65536, 2304, 768, True, dict(bias=True, qscale=True)),
    ("out_fwd", 65536, 768, 768, True, dict(bias=True, residual=True, drop=0.1)),
    ("fc1_gelu", 65536, 3072, 768, True, dict(bias=True, act=1, preact=True)),
    ("fc2_fwd", 65536, 768, 3072, True, dict(bias=True, residual=True, drop=0.1)),
    ("fc2_dgrad_a2", 65536, 3072, 768, False, dict(act=2, aux=True, colsum=True)),
    ("fc1_dgrad", 65536, 768, 3072, False, dict(residual=True)),
    ("qkv_dgrad", 65536, 768, 2304, False, dict(residual=True)),
    ("out_dgrad", 65536, 768, 768, False, dict()),
    ("dec_out_fwd", 32768, 768, 768, True, dict(bias=True, residual=True, drop=0.1)),
]


def child():
    import torch
    from gpu_util import DEV, bf, gemm
    out = {}
    for ci, (name, M, N, K, bkc, o) in enumerate(SHAPES):
        g = torch.Generator(device=DEV).manual_seed(100 + ci)
        rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
        A = bf(rn(M, K)) * 0.5
        B = bf(rn(N, K) if bkc else rn(K, N)) * 0.5
        kw = dict(a_kc=True, b_kc=bkc, M=M, N=N, K=K)
        if o.get("bias"):
            kw["bias"] = rn(N)
        if o.get("qscale"):
            kw["col_scale"], kw["col_scale_n"] = 0.125, 768
        kw["act"] = o.get("act", 0)
        if o.get("preact"):
            kw["preact"] = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        if o.get("aux"):
            kw["aux"] = bf(rn(M, N))
        if o.get("residual"):
            kw["residual"] = bf(rn(M, N))
        if o.get("drop"):
            kw["drop_p"], kw["drop_seed"] = o["drop"], 1234
        if o.get("colsum"):
            kw["colsum"] = torch.zeros((M + 63) // 64, N, dtype=torch.float32, device=DEV)
        kw["out_bf16"] = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        for _ in range(3):
            gemm(A, B, **kw)
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gemm(A, B, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 200.0)
        out[name] = min(ts)
        del A, B, kw
    print("JSON" + json.dumps(out))


def main():
    if os.environ.get("KMB_RS_CHILD"):
        return child()
    variants = (sys.argv[1] if len(sys.argv) > 1 else "10,11,13").split(",")
    res = {}
    for v in variants:
        env = dict(os.environ, KMB_GEMM_VARIANT=v, KMB_RS_CHILD="1", KMB_TILE_ORDER="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        if r.returncode != 0 or not line:
            print("variant", v, "failed:\n", r.stdout[-2000:], r.stderr[-3000:])
            sys.exit(1)
        res[v] = json.loads(line[0][4:])
    print("lib:", os.environ.get("KMB_LIB_PATH", "product"))
    for name, M, N, K, *_ in SHAPES:
        line = f"{name:14s} {M:6d}x{N:5d}x{K:5d}"
        for v in variants:
            us = res[v][name]
            line += f" | v{v} {us:7.1f} us {2.0 * M * N * K / us * 1e-6:5.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
