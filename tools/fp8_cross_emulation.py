#!/usr/bin/env python3
"""CPU emulation for the round-2 precision plan: hi*hi on fp16 plus the two cross terms on block-scaled fp8
(e4m3, one power-of-two scale per 32 elements) against fp64, next to single-pass fp16 and the current f16x3."""
import torch, math
torch.manual_seed(0)
M,N,K=512,512,1024
A=torch.randn(M,K)*torch.exp(torch.randn(M,1)*0.5)   # rows of different scale
W=torch.randn(N,K)/math.sqrt(K)
ref=(A.double()@W.double().t())
def rtz16(x):
    h=x.half(); over=h.float().abs()>x.abs()
    h=torch.where(over, torch.nextafter(h, torch.zeros_like(h)), h); return h
def rel(x): return ((x.double()-ref).norm()/ref.norm()).item()
Ah=rtz16(A); Al=(A-Ah.float()).half(); Wh=rtz16(W); Wl=(W-Wh.float()).half()
single=(A.half().float()@W.half().float().t())
x3=(Ah.float()@Wh.float().t())+(Ah.float()@Wl.float().t())+(Al.float()@Wh.float().t())
def mxq(x, bits="e4m3"):
    # per-32-block power-of-two scale (e8m0) then fp8
    xb=x.reshape(x.shape[0],-1,32)
    amax=xb.abs().amax(dim=-1,keepdim=True).clamp_min(1e-30)
    e=torch.floor(torch.log2(amax))
    scale=torch.exp2(e-7)          # map block max into [128,256) < 448 (e4m3 max)
    q=(xb/scale).to(torch.float8_e4m3fn).float()*scale
    return q.reshape(x.shape)
cross=(mxq(Ah.float())@mxq(Wl.float()).t())+(mxq(Al.float())@mxq(Wh.float()).t())
mix=(Ah.float()@Wh.float().t())+cross
print("single-pass fp16 rel err %.2e"%rel(single))
print("f16x3           rel err %.2e"%rel(x3))
print("hi*hi + MX-fp8 cross terms rel err %.2e"%rel(mix))
print("hi*hi only      rel err %.2e"%rel(Ah.float()@Wh.float().t()))
