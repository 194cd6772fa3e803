"""Weight-gradient GEMM (dY^T X, both operands token-major) with split-K: time per (variant, split) -- the GEMM launch
alone plus an estimate of the slab reduction ((S+1) x M x N x 4 B at 4 TB/s) -- to pick the engine's split rule.

    python tools/wgrad_split_sweep.py          # re-runs itself per variant
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
SHAPES = [(3072, 768, 32768), (768, 3072, 32768), (2304, 768, 32768), (1536, 768, 32768), (768, 768, 32768),
          (3072, 768, 16384), (768, 768, 16384), (2304, 768, 16384)]
if os.environ.get("KMB_WG_TOKENS"):   # e.g. KMB_WG_TOKENS=65536: the b = 1024 encoder shapes
    _t = int(os.environ["KMB_WG_TOKENS"])
    SHAPES = [(3072, 768, _t), (768, 3072, _t), (2304, 768, _t), (1536, 768, _t), (768, 768, _t)]


def child():
    import torch
    from gpu_util import DEV, bf, gemm
    out = {}
    slab = torch.empty(48 << 20, dtype=torch.float32, device=DEV)
    for M, N, K in SHAPES:
        A = bf(torch.randn(K, M, device=DEV))
        B = bf(torch.randn(K, N, device=DEV))
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        cands = sorted({max(1, min(16, (384 + t128 - 1) // t128)), max(1, 256 // t256), max(1, 512 // t256), max(1, 512 // t128),
                        max(1, 768 // t128), max(1, 384 // t256)})
        for S in cands:
            if S < 2 or S * M * N > slab.numel() or S > K // 128:
                continue
            for o in (1, 5):   # per-XCD ranges; 5: plus slice-major enumeration
                kw = dict(a_kc=False, b_kc=False, M=M, N=N, K=K, split_k=S, slab=slab, tile_order=o)
                for _ in range(2):
                    gemm(A, B, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    gemm(A, B, **kw)
                e1.record()
                torch.cuda.synchronize()
                out[f"{M}x{N}x{K} S{S} o{o}"] = e0.elapsed_time(e1) * 200.0
        del A, B
    print("JSON" + json.dumps(out))


def main():
    if os.environ.get("KMB_WG_CHILD"):
        return child()
    res = {}
    for v in ("7", "8"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, KMB_GEMM_VARIANT=v, KMB_WG_CHILD="1"),
                           capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)
        res[v] = json.loads(line[0][4:])
    for M, N, K in SHAPES:
        print(f"{M}x{N}x{K}:")
        rows = []
        for v in res:
            for key, us in res[v].items():
                if key.startswith(f"{M}x{N}x{K} "):
                    S = int(key.split()[1][1:])
                    red = (S + 1) * M * N * 4 / 4e12 * 1e6
                    rows.append((us + red, f"  v{v} {key.split()[1]:>4s} {key.split()[2]}  gemm {us:7.1f} us + reduce ~{red:5.1f} us = {us + red:7.1f} us  "
                                           f"({2.0 * M * N * K / (us + red) * 1e-6:6.0f} TF)"))
        for _, line in sorted(rows)[:5]:
            print(line)


if __name__ == "__main__":
    main()
