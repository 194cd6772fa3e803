"""cProfile of one beam-search generate (host-side cost of the beam bookkeeping)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).eval()
b = make_batch(64, seed=4321)
ids, am = b["input_ids"].to(dev), b["attention_mask"].to(dev)
feats = [f.to(dev) for f in b["image_features"]]
kw = dict(num_beams=5, num_return_sequences=1, max_length=20, early_stopping=True)
model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
model.generate(input_ids=ids, image_features=feats, attention_mask=am, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
