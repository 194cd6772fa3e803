"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce leaves every rank with the MEAN of the
per-rank gradients (DDP semantics of the reference, vcg_train.py:98), for any bucket order / split."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "km-bart_amd"))
    from kmbart.parallel import BucketedAllReducer
    from src.utils import cleanup_process, setup_process
    setup_process(rank, world, master_port=str(port), backend="gloo")
    n = 1000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    # backward completion order: tail first, tied matrix last; one bucket bigger than the split threshold
    buckets = [(600, 300), (300, 300), (0, 300), (900, 100)]
    waited = []
    red = BucketedAllReducer(flat, buckets, wait_ready=lambda i, s: waited.append(i), max_bucket_elems=128)
    assert sum(c for _, _, c in red.pieces) == n and max(c for _, _, c in red.pieces) <= 128
    red.launch()
    red.finish()
    others = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(others, mine)
    ref = torch.stack(others).mean(0)
    ok = torch.allclose(flat, ref, atol=1e-6) and waited == [0, 1, 2, 3]
    # a second step reuses the reducer
    flat.copy_(mine * 2)
    red.launch()
    red.finish()
    ok = ok and torch.allclose(flat, ref * 2, atol=1e-6)
    out[rank] = bool(ok)
    cleanup_process()


def test_bucketed_allreduce_is_a_mean_over_ranks():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_single_process_is_a_no_op():
    import sys
    from kmbart.parallel import BucketedAllReducer
    flat = torch.arange(10.0)
    red = BucketedAllReducer(flat, [(0, 10)])
    red.launch()
    red.finish()
    assert torch.equal(flat, torch.arange(10.0))
