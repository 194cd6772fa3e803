"""Debug aid: which gradient slices change when the one-rank RCCL reducer runs (they must not)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.distributed as dist
from oracle import goldenlib as G
from oracle.make_golden import tiny_batch
from kmbart.parallel import DistributedDataParallel
from test_model_gpu import build
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29534"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ocfg = G.tiny_config(dropout=0.0); sd = G.golden_state_dict(ocfg, seed=7); b = tiny_batch()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
res = []
for wrap in (False, True, True):
    model = build(ocfg, sd).train()
    ddp = DistributedDataParallel(model, device_ids=[0], reduce_single_rank=True) if wrap else model
    ddp.train_step_fwd_bwd(batch); torch.cuda.synchronize()
    res.append((model._engine.grads.clone(), model))
g0, m = res[0]
for k in (1, 2):
    g1 = res[k][0]
    bad = (g0 != g1).nonzero().flatten()
    print("run", k, "differing elements:", bad.numel())
    if bad.numel():
        eng = m._engine
        for name, (off, rows, cols) in list(eng.index.items()):
            cnt = rows * cols
            sel = ((bad >= off) & (bad < off + cnt)).sum().item()
            if sel:
                d = (g0[off:off+cnt] - g1[off:off+cnt]).abs().max().item()
                print("  %-60s %7d of %7d differ, max abs %.3e, ref norm %.3e" % (name, sel, cnt, d, g0[off:off+cnt].norm().item()))
print("buckets", res[1][1]._engine.buckets())
dist.destroy_process_group()
