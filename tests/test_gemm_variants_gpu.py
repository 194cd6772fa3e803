"""GPU: every GEMM variant the autotuner may pick (v7 128x128, v8 256x256, persistent v11 256x256 / v12 256x128 / v13
256x192 with four waves, v14 256x256 / v15 256x192 with eight waves, with and without per-XCD tile ranges, the
shared-device tile counter and the activation-panel L2 prefetch) must produce the SAME BITS as the
128x128 kernel on every epilogue class, operand layout and edge shape, and reproduce them on a relaunch
(tools/gemm_v11_check.py, one subprocess per variant: the variant is a per-process choice)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_all_variants_bit_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_v11_check.py")], capture_output=True, text=True,
                       timeout=2400)
    print(r.stdout[-6000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MISMATCH" not in r.stdout
