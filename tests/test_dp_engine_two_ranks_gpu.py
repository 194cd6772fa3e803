"""GPU, engine-level data parallelism (reference vcg_train.py:98 DDP semantics): TWO ranks,
each with its own model replica and its own minibatch, run forward + backward through the production
`kmbart.parallel.DistributedDataParallel` (parameter broadcast, per-bucket HIP events recorded inside kmb_backward, the
communication stream waiting on them, bucketed all-reduce).  Afterwards every rank's gradient arena must equal the
MEAN of the two ranks' serially computed gradients.  The box has one GPU, so both ranks sit on cuda:0 and gloo carries
the collectives (RCCL refuses two ranks on one device); the RCCL transport itself is covered by
tests/test_dp_rccl_gpu.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import goldenlib as G
    from oracle.make_golden import tiny_batch
    from kmbart.optim import AdamW
    from kmbart.parallel import DistributedDataParallel
    from test_model_gpu import build
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ocfg = G.tiny_config(dropout=0.0)
        # rank 1 starts from DIFFERENT weights: the wrapper's broadcast must overwrite them with rank 0's
        sd = G.golden_state_dict(ocfg, seed=7 if rank == 0 else 8)
        shapes = [dict(regions=(6, 3), event_lens=(8, 4), label_lens=(12, 7)),
                  dict(regions=(4, 0), event_lens=(10, 9), label_lens=(9, 12))]
        batches = []
        for r in range(world):
            b = tiny_batch(seed=100 + r, **shapes[r])
            bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
            bd["image_features"] = [f.to(dev) for f in b["image_features"]]
            batches.append(bd)
        # serial reference: rank-0 weights, each rank's batch, plain engine
        ref_model = build(ocfg, G.golden_state_dict(ocfg, seed=7)).train()
        serial = []
        for r in range(world):
            ref_model.train_step_fwd_bwd(batches[r])
            torch.cuda.synchronize()
            serial.append(ref_model._engine.grads.clone())
        mean = (serial[0] + serial[1]) / world
        model = build(ocfg, sd).train()
        ddp = DistributedDataParallel(model, device_ids=[0])
        assert ddp.reducer is not None
        ok = torch.equal(model._engine.params, ref_model._engine.params)      # broadcast from rank 0
        results = []
        for it in range(2):   # the second pass reuses events / reducer state
            ddp.train_step_fwd_bwd(batches[rank])
            torch.cuda.synchronize()
            got = model._engine.grads.clone()
            off, rows, cols = model._engine.index["model.shared.weight"]
            same = got == mean
            same[off: off + rows * cols] = True      # tied matrix: fp32 atomics of the embedding scatter-add (last bit)
            results.append(bool(same.all()) and torch.allclose(got[off: off + rows * cols], mean[off: off + rows * cols],
                                                               rtol=1e-5, atol=1e-7))
        # the fused tail: each piece's AdamW enqueued on the communication stream behind its all-reduce
        # (DistributedDataParallel.attach_optimizer); the result must be ONE optimizer step on the mean gradients
        ref_opt = AdamW(ref_model.parameters(), lr=1e-3)
        ref_model._engine.grads.copy_(mean)
        ref_opt.step()
        opt = AdamW(model.parameters(), lr=1e-3)
        opt.allow_overlap(True)
        fused = ddp.attach_optimizer(opt)
        ddp.train_step_fwd_bwd(batches[rank])
        stepped_inside = model._engine.step_count == 1
        opt.step()
        torch.cuda.synchronize()
        off, rows, cols = model._engine.index["model.shared.weight"]
        same = model._engine.params == ref_model._engine.params
        same[off: off + rows * cols] = True
        matches_ref = bool(same.all()) and torch.allclose(model._engine.params[off: off + rows * cols],
                                                          ref_model._engine.params[off: off + rows * cols], rtol=0, atol=2e-6)
        mine = model._engine.params.detach().cpu()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        # tied rows touched by the atomics may differ in the last bit between ranks -> allclose
        in_sync = fused and stepped_inside and model._engine.step_count == 1 and matches_ref and \
            torch.allclose(gathered[0], gathered[1], rtol=0, atol=2e-6)
        # bf16 gradient buckets (282 MB on the wire instead of 564): same mean within bf16 rounding
        model2 = build(ocfg, G.golden_state_dict(ocfg, seed=7)).train()
        ddp2 = DistributedDataParallel(model2, device_ids=[0], grad_dtype="bf16")
        ddp2.train_step_fwd_bwd(batches[rank])
        torch.cuda.synchronize()
        g2 = model2._engine.grads
        in_sync = in_sync and float((g2 - mean).norm() / mean.norm()) < 4e-3
        out[rank] = (ok, results, in_sync)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_the_mean_of_serial_gradients():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: (True, [True, True], True), 1: (True, [True, True], True)}, dict(out)
