"""Stability soak: vcg_base on ONE fixed synthetic batch (b=256, or argv[2]; dropout 0.1) for N steps of fused AdamW; the loss must
fall monotonically-ish from ln(V) as the batch is memorised and stay finite (bf16 path, no loss scaling).
    python tools/train_soak.py [steps=300] [batch=256]      (batch <= 48 takes the grouped weight-gradient launches)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from kmbart.optim import AdamW
from src.data.synthetic import make_batch
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-4)
opt.allow_overlap(True)
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
b = make_batch(BATCH, seed=1)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batch["image_features"] = [f.to(dev) for f in b["image_features"]]
trace = []
for i in range(steps):
    loss = model.train_step_fwd_bwd(batch)
    opt.step()
    if i % 25 == 0 or i == steps - 1:
        trace.append((i, round(float(loss), 4)))
print(json.dumps({"steps": steps, "lr": 1e-4, "batch": BATCH, "loss_trace": trace,
                  "finite": all(l == l and abs(l) < 1e4 for _, l in trace)}))
