"""Debug: every GEMM variant (KMB_GEMM_VARIANT=1, 7, 8) must agree bit for bit; run once per variant
and diff the printed checksums, or repeat launches to screen for races."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

torch.manual_seed(0)
shapes = [(256, 128, 128, True, True), (512, 384, 256, True, True), (300, 768, 768, True, True),
          (256, 256, 256, True, False), (4096, 768, 3072, True, False), (1024, 768, 768, False, False),
          (2048, 3072, 768, False, False)]
for (M, N, K, akc, bkc) in shapes:
    A = bf(torch.randn((M, K) if akc else (K, M), device=DEV))
    B = bf(torch.randn((N, K) if bkc else (K, N), device=DEV))
    ref = None
    bad = 0
    for rep in range(20):
        out = torch.empty((M, N), dtype=torch.float32, device=DEV)
        gemm(A, B, a_kc=akc, b_kc=bkc, out_f32=out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(ref, out):
            bad += 1
    h = hashlib.md5(ref.cpu().numpy().tobytes()).hexdigest()[:12]
    print(M, N, K, akc, bkc, "md5", h, "nondeterministic_reps", bad, flush=True)
