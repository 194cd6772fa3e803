import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "km-bart_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    try:  # os.cpu_count() reports the whole host; oversubscribed OpenMP pools make the CPU oracle crawl
        import torch
        torch.set_num_threads(_usable_cores())
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD
