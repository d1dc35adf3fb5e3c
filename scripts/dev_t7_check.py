import sys, os, math
sys.path.insert(0, "/root/repo")
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(0)
for (M,N,K) in [(65536,320,320),(16384,640,640),(65536,960,320)]:
    a = torch.randn(M,K,generator=g).to(dev,dt); w=(torch.randn(N,K,generator=g)/math.sqrt(K)).to(dev,dt); b=torch.randn(N,generator=g).to(dev,dt); r=torch.randn(M,N,generator=g).to(dev,dt)
    o0 = ops.gemm(a,w,M,N,K,bias=b,res=r)
    o7 = ops.gemm(a,w,M,N,K,bias=b,res=r,force_tile=7)
    ref = (a.float()@w.float().t()+b.float()+r.float())
    print(M,N,K, "equal" if torch.equal(o0,o7) else "diff", ((o7.float()-ref).norm()/ref.norm()).item())
