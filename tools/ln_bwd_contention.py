"""LayerNorm backward alone and beside a long GEMM on a second stream: which outputs differ from the solo result, by how
much and where (the SLP-vectoriser finding of DESIGN.md section 5; 0 differing runs with the library as built now)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT+"/km-bart_amd"); sys.path.insert(0, ROOT+"/tests")
import torch
from gpu_util import DEV, bf, check, gemm, ptr, stream
from kmbart import _lib
lib=_lib.load(); torch.manual_seed(0)
side=torch.cuda.Stream()
hA=bf(torch.randn(32768,3072,device=DEV)); hB=bf(torch.randn(32768,768,device=DEV)); hslab=torch.empty(3*3072*768,dtype=torch.float32,device=DEV)
def hog():
    with torch.cuda.stream(side):
        for _ in range(3): gemm(hA,hB,a_kc=False,b_kc=False,M=3072,N=768,K=32768,split_k=3,slab=hslab)
M,D=2048,768
z=bf(torch.randn(M,D,device=DEV)); gamma=torch.randn(D,device=DEV)
mean,rstd=z.float().mean(-1),(z.float().var(-1,unbiased=False)+1e-5).rsqrt()
dy=bf(torch.randn(M,D,device=DEV)); dz=torch.zeros_like(z); dg,db=torch.zeros(D,device=DEV),torch.zeros(D,device=DEV)
scratch=torch.empty(int(lib.kmb_op_ln_bwd_scratch(M,D)),device=DEV)
def f(): check(lib.kmb_op_ln_bwd(ptr(dy),ptr(z),ptr(mean),ptr(rstd),ptr(gamma),ptr(dz),None,None,None,ptr(dg),ptr(db),ptr(scratch),M,D,stream()))
f(); torch.cuda.synchronize(); ref=[t.clone() for t in (dz,dg,db)]
for mode in ("solo","hog"):
    cnt=[0,0,0]; mx=[0,0,0]
    for _ in range(20):
        for t in (dz,dg,db): t.zero_()
        torch.cuda.synchronize()
        if mode=="hog": hog()
        f(); torch.cuda.synchronize()
        for i,(a,b) in enumerate(zip(ref,(dz,dg,db))):
            if not torch.equal(a,b):
                cnt[i]+=1; mx[i]=max(mx[i], float((a.float()-b.float()).abs().max()/ (a.float().abs().max()+1e-30)))
    print(mode,"differing runs dz/dg/db:",cnt,"max rel diff:",mx)
# is the other kernel writing where it should not?  zero the outputs, run ONLY the side GEMM, look again
for t in (dz, dg, db, scratch): t.zero_()
guard = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
torch.cuda.synchronize()
hog(); torch.cuda.synchronize()
print("after the side GEMM alone: dz nonzero", int((dz != 0).sum()), "scratch nonzero", int((scratch != 0).sum()), "guard nonzero", int((guard != 0).sum()))
# where do the differences sit, and do two contended runs agree with each other?
outs = []
for _ in range(3):
    dz.zero_(); torch.cuda.synchronize(); hog(); f(); torch.cuda.synchronize(); outs.append(dz.clone())
d01 = (outs[0] != outs[1]); dr = (outs[0] != ref[0])
print("contended run 0 vs 1: differing elements", int(d01.sum()), "| vs solo:", int(dr.sum()), "of", dz.numel())
rows = dr.any(1).nonzero().flatten(); cols = dr.any(0).nonzero().flatten()
print("rows with differences:", rows.numel(), rows[:12].tolist(), "... cols:", cols.numel(), cols[:12].tolist())
r0 = int(rows[0]); bad = dr[r0].nonzero().flatten()
print("row", r0, "bad cols", bad[:16].tolist(), "solo", ref[0][r0, bad[:4]].tolist(), "contended", outs[0][r0, bad[:4]].tolist())
