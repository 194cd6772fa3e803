"""Library yardstick for the engine's GEMM shape table: times torch.mm (hipBLASLt / rocBLAS behind it) in bf16 on every
(variant, M, N, K) line of profiles/r01_gemm_shape_table_b512.txt and prints it next to the engine's own time.

Measurement only -- nothing in the product path calls a library GEMM.  The library runs the bare product (no bias,
activation, residual, dropout, column sums or split-K reduction), so its time is a lower bound for what a library-based
path would pay before its separate elementwise kernels.

Two library columns: WARM = back-to-back launches on the same operands (they stay in the 256 MB Infinity Cache) and COLD
= a 600 MB fill between launches (operands come from HBM).  The engine's column is measured INSIDE a training step,
where the activation operand was just streamed out by the previous kernel: the cold column is the like-for-like one
(tools/gemm_cold_warm.py shows the same 25-35 % warm/cold gap on the engine's own kernels at K = 3072).

    python tools/gemm_yardstick.py [profiles/r01_gemm_shape_table_b512.txt] > gpurun_out/yardstick.txt
"""
import sys

import torch


def time_mm(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def time_cold(fn, junk, iters=8):
    ts = []
    for _ in range(iters):
        junk.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01_gemm_shape_table_b512.txt"
    dev = torch.device("cuda", 0)
    rows = []
    for line in open(path):
        parts = line.replace("|", " ").split()
        if len(parts) < 9 or not parts[0].isdigit():
            continue
        var, M, N, K = (int(parts[i]) for i in range(4))
        launches, mine_us = int(parts[6]), float(parts[7])
        rows.append((var, M, N, K, parts[4], parts[5], launches, mine_us))
    print("variant M N K split act | launches engine_us engine_TF | library warm_us TF | library cold_us TF | engine speed vs warm / vs cold")
    junk = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    tot_mine = tot_lib = tot_cold = tot_fl = 0.0
    for var, M, N, K, sp, act, launches, mine_us in rows:
        g = torch.Generator(device=dev).manual_seed(1)
        if var == 3:     # forward: C[M,N] = A[M,K] . B[N,K]^T
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
            B = torch.randn(N, K, device=dev, dtype=torch.bfloat16, generator=g)
            fn = lambda: torch.mm(A, B.t())
        elif var == 2:   # data gradient: C[M,N] = A[M,K] . B[K,N]
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
            B = torch.randn(K, N, device=dev, dtype=torch.bfloat16, generator=g)
            fn = lambda: torch.mm(A, B)
        else:            # weight gradient: C[M,N] = A[K,M]^T . B[K,N]
            A = torch.randn(K, M, device=dev, dtype=torch.bfloat16, generator=g)
            B = torch.randn(K, N, device=dev, dtype=torch.bfloat16, generator=g)
            fn = lambda: torch.mm(A.t(), B)
        lib_us = time_mm(fn)
        cold_us = time_cold(fn, junk)
        fl = 2.0 * M * N * K
        tot_mine += mine_us * launches
        tot_lib += lib_us * launches
        tot_cold += cold_us * launches
        tot_fl += fl * launches
        print(f"{var} {M:6d} {N:6d} {K:6d} {sp:>3s} {act} | {launches:3d} {mine_us:8.1f} {fl / mine_us * 1e-6:7.1f} | "
              f"{lib_us:8.1f} {fl / lib_us * 1e-6:7.1f} | {cold_us:8.1f} {fl / cold_us * 1e-6:7.1f} | "
              f"{lib_us / mine_us:5.2f}x {cold_us / mine_us:5.2f}x", flush=True)
        del A, B
    print(f"whole table: engine {tot_mine * 1e-3:.2f} ms = {tot_fl / tot_mine * 1e-6:.1f} TFLOP/s (in-step, fused epilogues, "
          f"split-K reductions excluded); library warm {tot_lib * 1e-3:.2f} ms = {tot_fl / tot_lib * 1e-6:.1f} TFLOP/s, "
          f"cold {tot_cold * 1e-3:.2f} ms = {tot_fl / tot_cold * 1e-6:.1f} TFLOP/s (bare products; single cold launches "
          f"carry ~5 us of event overhead each)")


if __name__ == "__main__":
    main()
