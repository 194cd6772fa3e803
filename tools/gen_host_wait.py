"""Is the beam-search loop bound by the host or by the GPU?  Runs tools/gen_bench.py's generate with
torch.cuda.Event.synchronize timed: the loop waits there for decode step t - 1 while step t is queued, so the time spent
waiting is the host's slack per generate (about zero = the host cannot keep the queue full)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = make_batch(B, seed=4321)
ids, am = b["input_ids"].to(dev), b["attention_mask"].to(dev)
feats = [f.to(dev) for f in b["image_features"]]
kw = dict(num_beams=5, num_return_sequences=1, max_length=20, early_stopping=True)
for _ in range(3):
    model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
waits = []
orig = torch.cuda.Event.synchronize


def timed(self):
    t = time.perf_counter()
    orig(self)
    waits.append(time.perf_counter() - t)


torch.cuda.Event.synchronize = timed
reps = 8
t0 = time.perf_counter()
for _ in range(reps):
    model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"batch {B}: {dt * 1e3:.2f} ms per generate; {len(waits) / reps:.0f} event waits per generate, "
      f"{sum(waits) / reps * 1e3:.2f} ms spent in them (median {sorted(waits)[len(waits) // 2] * 1e6:.0f} us, "
      f"{sum(1 for w in waits if w < 20e-6) / len(waits) * 100:.0f} % under 20 us = the step had already finished)")
