"""Worker of tests/test_dp_rccl_gpu.py::test_multi_rank_rccl (one process per GPU, launched by torch.distributed.run):
the production data-parallel path over RCCL with WORLD_SIZE ranks.

Every rank holds the same weights and its own minibatch (seed 100 + rank), runs ONE DistributedDataParallel step with
the fused optimizer tail, and checks on its own device:
  * gradients after the bucketed all-reduce == the arithmetic mean over ranks of the per-rank gradients, which the rank
    recomputes serially on a plain replica (reference semantics: vcg_train.py:98 DDP mean of per-rank mean-token losses);
  * after the fused AdamW step every rank holds the same parameters (all_gather of a checksum and of a sample);
  * rank 0 prints the wrapper's comm report (ranks, buckets, bytes reduced, exposed tail).
Exit code != 0 on any mismatch (the launcher propagates it)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from oracle import goldenlib as G
    from oracle.make_golden import tiny_batch
    from kmbart.optim import AdamW
    from kmbart.parallel import DistributedDataParallel
    from test_model_gpu import build
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    if os.environ.get("KMB_TEST_ONE_DEVICE") == "1":   # probe: several ranks on ONE device (RCCL normally refuses duplicate devices)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        ocfg = G.tiny_config(dropout=0.0)
        sd = G.golden_state_dict(ocfg, seed=7)

        def batch_of(r):
            b = tiny_batch(seed=100 + r)
            out = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
            out["image_features"] = [f.to(dev) for f in b["image_features"]]
            return out

        model = build(ocfg, sd, device=dev).train()
        ddp = DistributedDataParallel(model, device_ids=[local], reduce_single_rank=True)
        opt = AdamW(model.parameters(), lr=1e-3)
        opt.allow_overlap(True)
        assert ddp.attach_optimizer(opt)
        eng = model._engine
        # step 0 without the optimizer fused in, so the reduced gradients can be read before they are consumed
        ddp.detach_optimizer()
        ddp.train_step_fwd_bwd(batch_of(rank))
        torch.cuda.synchronize()
        got = eng.grads.clone()
        plain = build(ocfg, sd, device=dev).train()
        ref = torch.zeros_like(got, dtype=torch.float64)
        for r in range(world):
            plain.train_step_fwd_bwd(batch_of(r))
            torch.cuda.synchronize()
            ref += plain._engine.grads.double()
        ref /= world
        err = float((got.double() - ref).norm() / (ref.norm() + 1e-30))
        worst = float((got.double() - ref).abs().max())
        assert err < 1e-5, ("rank %d: all-reduced gradients differ from the mean of the per-rank gradients" % rank, err, worst)
        # fused tail: one step, replicas must stay identical
        assert ddp.attach_optimizer(opt)
        ddp.train_step_fwd_bwd(batch_of(rank))
        opt.step()
        torch.cuda.synchronize()
        probe = torch.cat([eng.params.double().sum().reshape(1), eng.params[:: max(1, eng.n // 1000)][:1000].double()])
        gathered = [torch.empty_like(probe) for _ in range(world)]
        dist.all_gather(gathered, probe)
        for r, g in enumerate(gathered):
            assert torch.equal(g, gathered[0]), "replica %d diverged from replica 0 after a fused step" % r
        if rank == 0:
            print("DP_RCCL_OK " + json.dumps(dict(ddp.comm_report(), grad_rel_err=err)), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
