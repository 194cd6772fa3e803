"""A GEMM launch beside a REAL RCCL gradient exchange (`kmb_gemm_shared_device` had only been measured beside
parked dummy workgroups).  One-rank communicator on the one-GPU box: kmb_allreduce_grads enqueues the 564 MB exchange of a
vcg_base gradient arena (one ncclAllReduce per <= 64 MB piece on the library's communication stream); while it runs, a
16384 x 3072 x 768 forward GEMM is timed on the main stream -- persistent kernel with a fixed first tile
(shared_device = 0), with every tile handed out dynamically (shared_device = 1); KMB_GEMM_VARIANT=8 in the environment times the one-workgroup-per-tile kernel instead.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/gemm_beside_rccl.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402
from kmbart import _lib  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)), device_id=dev)
lib = _lib.load()
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
eng = model._engine
eng.comm_init(None)
b = make_batch(64, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
for _ in range(2):   # bucket events exist after a backward
    model.train_step_fwd_bwd(batch)
torch.cuda.synchronize()

M, N, K = 16384, 3072, 768
A = bf(torch.randn(M, K, device=DEV))
B = bf(torch.randn(N, K, device=DEV) * 0.05)
out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)


def exchange_ms():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    eng.allreduce_grads(algo=0, max_piece_elems=16 << 20, after_compute=True)
    eng.comm_wait()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def gemm_us(beside, reps=6):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        if beside:
            eng.allreduce_grads(algo=0, max_piece_elems=16 << 20, after_compute=True)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record()
            gemm(A, B, out_bf16=out)
            e1.record()
        if beside:
            eng.comm_wait()
        torch.cuda.synchronize()
        ts += [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]
    ts.sort()
    return ts[len(ts) // 2], ts[-1]


print("one-rank exchange of %d MB alone: %.2f ms" % (eng.n * 4 >> 20, sorted(exchange_ms() for _ in range(5))[2]))
variant = os.environ.get("KMB_GEMM_VARIANT", "tuner's pick")   # read once per process by the launcher: one variant per run
for shared in (0, 1):
    lib.kmb_gemm_shared_device(shared)
    for _ in range(3):
        gemm(A, B, out_bf16=out)
    alone = gemm_us(False)
    beside = gemm_us(True)
    print("gemm %dx%dx%d shared_device=%d (variant env %s): alone median %.1f us (max %.1f); beside the exchange median %.1f us (max %.1f)"
          % (M, N, K, shared, variant, alone[0], alone[1], beside[0], beside[1]))
dist.destroy_process_group()
