"""bench.py's N > 1 control flow on a one-GPU box: two ranks share cuda:0 and gloo carries the collectives (RCCL
refuses two ranks on one device; test hooks KMB_BENCH_ONE_DEVICE / KMB_BENCH_BACKEND).  What this pins: every rank
issues the same sequence of collectives -- parameter broadcast, bucketed gradient all-reduce in every step, barrier,
MAX over ranks, final barrier -- and the rank-0-only legs (roofline) issue none, so the driver's 2/4/8-GPU launch
cannot hang on a mismatched collective."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_bench_completes_and_reports():
    env = dict(os.environ, KMB_BENCH_BACKEND="gloo", KMB_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 2
    assert "roofline" in out and out["roofline"]["achieved"] > 0
    assert "cpu_baseline" not in out                   # rank-0, N = 1 only


def test_plain_command_spawns_its_ranks():
    """The EXACT form the driver issues for N > 1 -- `python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE -- goes through
    bench.py::spawn_ranks (fresh child ranks before any GPU call, a free loopback port; reference: vcg_train.py:350-355 spawns its ranks
    from one command).  Same one-GPU hooks as above; the JSON line must come out of THIS process's stdout and carry the exchange report."""
    env = dict(os.environ, KMB_BENCH_BACKEND="gloo", KMB_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr      # spawn_ranks ran, not an in-process fallback
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak" and out["value"] > 0
    assert out["comm"]["rccl_ranks"] == 2, out.get("comm")
