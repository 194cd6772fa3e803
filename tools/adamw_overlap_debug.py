"""Debug: parameters / gradients after one step with the optimizer issued per bucket beside backward vs in one launch,
and two plain runs against each other (which arena ranges differ)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/km-bart_amd", ROOT+"/tests"): sys.path.insert(0,p)
import torch, bench
from kmbart.optim import AdamW
from src.data.synthetic import make_batch
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
DEV="cuda:0"
B=int(sys.argv[1])
b=make_batch(B, seed=77)
d={k:v.to(DEV) for k,v in b.items() if torch.is_tensor(v)}; d["image_features"]=[f.to(DEV) for f in b["image_features"]]
def run(overlap):
    torch.manual_seed(3)
    m=MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(dict(bench.VCG_BASE, dropout=0.0))).to(DEV); m.train()
    opt=AdamW(m.parameters(), lr=1e-3); opt.overlap=overlap
    out=m(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"], decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"], labels=d["labels"])
    out[0].backward(); 
    eng=m._need_engine()
    torch.cuda.synchronize(); g=eng.grads.clone()
    # redo to include step timing overlap: second iteration does step right after backward enqueue
    out=m(input_ids=d["input_ids"], image_features=d["image_features"], attention_mask=d["attention_mask"], decoder_input_ids=d["decoder_input_ids"], decoder_attention_mask=d["decoder_attention_mask"], labels=d["labels"])
    out[0].backward(); opt.step(); torch.cuda.synchronize()
    return g, eng.grads.clone(), eng.params.clone(), eng.buckets(), [ (n,p._kmb_range) for n,p in m.named_parameters()]
r=[run(False), run(False), run(True)]
names=r[0][4]
def where(a,b):
    idx=(a!=b).nonzero().flatten()
    if idx.numel()==0: return "identical"
    lo,hi=int(idx.min()),int(idx.max())
    hit=[n for n,(o,c) in names if o<=hi and o+c>lo and bool((a[o:o+c]!=b[o:o+c]).any())]
    return f"{idx.numel()} differ in [{lo},{hi}] params: {hit[:8]}{'...' if len(hit)>8 else ''}"
print("grads step1 noovl vs noovl:", where(r[0][0], r[1][0]))
print("grads step2 noovl vs noovl:", where(r[0][1], r[1][1]))
print("params noovl vs noovl:", where(r[0][2], r[1][2]))
print("grads step2 noovl vs ovl:", where(r[0][1], r[2][1]))
print("params noovl vs ovl:", where(r[0][2], r[2][2]))
