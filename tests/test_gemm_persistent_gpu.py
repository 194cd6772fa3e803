"""The column-block tile enumeration of the persistent GEMM variants (tile_order bit 3: blocks of eight column tiles, all
row panels of a block before the next one -- what the LM head runs with) on every persistent variant, four- and
eight-wave: results must not depend on the enumeration, so every case must reproduce the 128x128 variant bit for bit
(tools/gemm_v11_check.py re-runs itself once per variant and diffs checksums; cases with 9 .. 197 column tiles exercise
full blocks, the partial last block and the fall-back for narrow outputs).  Variants 6 (eight waves around the bare K loop, gemm_lean.hip), 9 (two workgroups per CU, gemm_pair.hip) -- both also
with column blocks: 6o8, 9o8, and 6 with its tiles handed out dynamically: 6o8s -- and 10 (role split) ride along: forced, they run on every case their launch rule admits.  The plain orders of every variant are covered by
tests/test_gemm_variants_gpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_column_block_order_is_bitwise_neutral():
    env = dict(os.environ, KMB_V11_CHECK_VARIANTS="7,8,6,6o8,6o8s,9,9o8,10,11o9,12o9,13o9,14o9,15o9")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_v11_check.py")], capture_output=True, text=True,
                       timeout=1800, env=env)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MISMATCH" not in r.stdout and r.stdout.count("same bits") >= 11 * 30
