"""The three drivers (vcg_train.py, pretrain.py, vcg_generate.py) end to end on files in the reference's dataset
format: tools/cli_smoke.py writes a tiny corpus + vocabulary + config, fine-tunes (with validation loss), pre-trains
on COCO + Visual Genome + VCG records and generates with beam search from the fine-tuned checkpoint."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_drivers_run_on_files():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_smoke.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "CLI smoke OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
