"""The persistent GEMM variants (11: 256x256 tiles, 12: 256x128; one workgroup per CU walking its tiles as one linear
K-step sequence) must reproduce the 128x128 variant bit for bit on every epilogue class, operand layout and edge shape.
The variant is a per-process choice (KMB_GEMM_VARIANT), so the comparison runs tools/gemm_v11_check.py, which re-runs
itself once per variant and diffs checksums of the outputs (column sums: after folding their partial rows)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_persistent_variants_match_the_128x128_variant_bitwise():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_v11_check.py")], capture_output=True, text=True,
                       timeout=1800)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MISMATCH" not in r.stdout and r.stdout.count("same bits") >= 105 + 4 * 30
