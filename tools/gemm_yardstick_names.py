"""Runs a few library GEMMs once each so that a rocprofv3 --kernel-trace shows which library kernels they map to."""
import torch
dev = torch.device("cuda", 0)
shapes = [(3, 32768, 3072, 768), (3, 16384, 3072, 768), (3, 32768, 768, 3072), (2, 32768, 768, 3072),
          (3, 16384, 768, 768), (2, 16384, 768, 768), (3, 8192, 50320, 768), (3, 32768, 2304, 768), (3, 4096, 4096, 4096)]
for var, M, N, K in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    B = torch.randn((N, K) if var == 3 else (K, N), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        C = torch.mm(A, B.t() if var == 3 else B)
    torch.cuda.synchronize()
