"""Kernels must give the same bits whether or not another kernel shares the GPU (data-parallel training and the
weight-gradient side stream both run kernels concurrently).  tools/contention_determinism.py runs attention backward,
data-gradient / forward GEMMs and LayerNorm backward alone and beside a long split-K GEMM on a second stream."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_kernels_are_bit_reproducible_beside_another_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "contention_determinism.py")], capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if "differ from the solo result" in ln]
    print("\n".join(lines))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(lines) >= 9 and all(ln.split(":")[1].strip().startswith("0 of") for ln in lines), lines
