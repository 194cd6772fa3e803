"""The data-parallel gradient path on the real RCCL backend.  The box has ONE GPU, so the group
has one rank -- the arithmetic is the identity -- but everything else is the production path: parameter broadcast,
per-bucket HIP events recorded inside kmb_backward, the communication stream waiting on them, `nccl` all-reduce (AVG)
of every bucket of the flat gradient arena, the compute stream waiting for the collectives before AdamW.  The
world_size-2 arithmetic is covered on gloo (tests/test_dp_gloo_cpu.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["native", "native_rsag", "torch"])
def test_single_rank_rccl_reducer_matches_plain_step(mode):
    """mode native: the library's own communicator (kmb_comm_init / kmb_allreduce_grads: every bucket's ncclAllReduce and
    the fused AdamW pieces enqueued from C++ on the library's communication stream); native_rsag: ncclReduceScatter ->
    AdamW on the shard -> ncclAllGather; torch: the torch.distributed path (BucketedAllReducer)."""
    from oracle import goldenlib as G
    from oracle.make_golden import tiny_batch
    from kmbart.optim import AdamW
    from kmbart.parallel import DistributedDataParallel
    from test_model_gpu import build
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str({"native": 29533, "native_rsag": 29534, "torch": 29535}[mode])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ocfg = G.tiny_config(dropout=0.0)
        sd = G.golden_state_dict(ocfg, seed=7)
        b = tiny_batch()
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
        batch["image_features"] = [f.to(dev) for f in b["image_features"]]

        def run(wrap):
            model = build(ocfg, sd).train()
            ddp = DistributedDataParallel(model, device_ids=[0], reduce_single_rank=True, native=(mode != "torch"),
                                          algo="rsag" if mode == "native_rsag" else "allreduce") if wrap else model
            opt = AdamW(model.parameters(), lr=1e-3)
            opt.allow_overlap(True)
            if wrap:   # the production tail: AdamW of every piece behind its all-reduce on the communication stream
                assert ddp.attach_optimizer(opt)
            losses, grads = [], None
            for i in range(3):
                losses.append(float(ddp.train_step_fwd_bwd(batch)))
                if i == 0:
                    torch.cuda.synchronize()
                    grads = model._engine.grads.clone()
                opt.step()
            torch.cuda.synchronize()
            if wrap and mode == "torch":
                assert ddp.reducer is not None and len(ddp.reducer.pieces) >= len(model._engine.buckets())
            if wrap:
                assert ddp.module is model and ddp.native == (mode != "torch")
                rep = ddp.comm_report()
                assert rep["rccl_ranks"] == 1 and rep["pieces"] >= len(model._engine.buckets()) and rep["fused_optimizer"]
                if mode != "torch":
                    assert rep["backend"] == "rccl-native" and rep["algo"] == ("rsag" if mode == "native_rsag" else "allreduce")
                    ddp.gather_optimizer_state()      # a no-op all-gather on one rank; must not disturb anything
                    torch.cuda.synchronize()
            assert model._engine.step_count == 3
            return losses, grads, model._engine, model._engine.params.clone()

        plain_losses, plain_grads, eng, plain_params = run(False)
        rccl_losses, rccl_grads, _, rccl_params = run(True)
        # AVG over one rank is the identity: every gradient slice comes back bit for bit -- except the tied embedding
        # matrix, whose token-gradient scatter uses fp32 atomics and differs run to run in the last bit either way
        off, rows, cols = eng.index["model.shared.weight"]
        same = plain_grads == rccl_grads
        same[off: off + rows * cols] = True
        assert bool(same.all())
        assert torch.allclose(plain_grads[off: off + rows * cols], rccl_grads[off: off + rows * cols], rtol=0, atol=1e-6)
        for a, b_ in zip(plain_losses, rccl_losses):
            assert abs(a - b_) <= 1e-4 * abs(a)
        # three fused (all-reduce -> AdamW per piece) steps land on the same parameters as three plain steps
        same = plain_params == rccl_params
        same[off: off + rows * cols] = True
        assert bool(same.all())
        assert torch.allclose(plain_params[off: off + rows * cols], rccl_params[off: off + rows * cols], rtol=0, atol=1e-5)
    finally:
        dist.destroy_process_group()


def test_native_bootstrap_failure_takes_the_torch_path(monkeypatch, capfd):
    """If creating the library's communicator raises, every rank agrees (all-reduce of the outcome over the bootstrap group) to use
    the torch.distributed exchange instead -- loudly, and still on RCCL: the step runs, the report names the reason."""
    from oracle import goldenlib as G
    from oracle.make_golden import tiny_batch
    from kmbart import engine as E
    from kmbart.parallel import DistributedDataParallel
    from test_model_gpu import build
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29536"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def boom(self, process_group=None):
            raise RuntimeError("simulated ncclCommInitRank failure")
        monkeypatch.setattr(E.Engine, "comm_init", boom)
        ocfg = G.tiny_config(dropout=0.0)
        model = build(ocfg, G.golden_state_dict(ocfg, seed=7)).train()
        ddp = DistributedDataParallel(model, device_ids=[0], reduce_single_rank=True, native=True)
        assert not ddp.native and ddp.reducer is not None and "simulated" in ddp.native_error
        assert "native RCCL exchange unavailable" in capfd.readouterr().err
        b = tiny_batch()
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
        batch["image_features"] = [f.to(dev) for f in b["image_features"]]
        loss = float(ddp.train_step_fwd_bwd(batch))
        torch.cuda.synchronize()
        assert loss == loss
        rep = ddp.comm_report()
        assert rep["backend"] == "nccl" and "simulated" in rep["native_fallback"]
    finally:
        dist.destroy_process_group()


def _n_gpus():
    return torch.cuda.device_count()   # counting devices does not initialise the GPU


@pytest.mark.parametrize("algo", ["allreduce", "rsag"])
@pytest.mark.parametrize("world", sorted({1, min(max(_n_gpus(), 1), 8)}))
def test_multi_rank_rccl(world, algo):
    """N ranks over RCCL on a multi-GPU node (world = min(device_count, 8)); on the one-GPU test box only the world = 1
    case exists, which still runs the worker end to end (process group, broadcast, bucketed all-reduce on the
    communication stream, fused optimizer tail, replica comparison)."""
    import subprocess
    port = 29540 + world + (20 if algo == "rsag" else 0)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KMB_DP_ALGO=algo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_rccl_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    line = [l for l in r.stdout.splitlines() if l.startswith("DP_RCCL_OK ")]
    assert line, tail
    import json
    rep = json.loads(line[0][len("DP_RCCL_OK "):])
    print("[dp rccl world=%d]" % world, rep)
    assert rep["rccl_ranks"] == world and rep["backend"] == "rccl-native" and rep["bytes_reduced_per_step"] > 0
    assert rep["grad_rel_err"] < 1e-5 and rep["algo"] == (algo if 8 % world == 0 else "allreduce")
