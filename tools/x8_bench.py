#!/usr/bin/env python3
"""Time the encoder linears on the 8-bit cross-term kernel against the 3-product kernels (GPU box)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime
from vdn import pack, _abi
rt = Runtime(torch.device("cuda:0"), torch.float16, split=True)
_abi.set_tuning(sk=0)
torch.manual_seed(0)
M = int(os.environ.get('M', 8 * 1370))
for name, N, K, kw in (("qkv", 3072, 1024, {}), ("proj", 1024, 1024, {"res": True}), ("fc1", 4096, 1024, {"gelu": True}), ("fc2", 1024, 4096, {"res": True})):
    a = rt.to_half(torch.randn(M, K, device="cuda"))
    w = pack.linear(torch.randn(N, K, device="cuda") / math.sqrt(K), rt.prec)
    a8, w8 = pack.planes8(a), pack.planes8(w)
    bias = torch.randn(N, device="cuda")
    for mode in ("x3", "x8"):
        ts = []
        for i in range(12):
            if kw.get("res"):
                out = torch.zeros(M, N, device="cuda"); args = dict(out=out, bias=bias, res1=out)
            elif kw.get("gelu"):
                out = rt.hbuf(f"xb_{N}", (M, N)); args = dict(out=out, bias=bias, act=_abi.ACT_GELU)
            else:
                out = rt.hbuf(f"xb_{N}", (M, N)); args = dict(out=out, bias=bias)
            if mode == "x8":
                args.update(a8=a8, w8=w8)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rt.gemm(a, w, M, N, K, **args); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ts = sorted(ts[2:]); med = ts[len(ts) // 2]
        print(f"{name:5s} M={M} N={N:5d} K={K:5d} {mode}: {med*1e3:7.1f} us  {2.0*M*N*K/med/1e9:7.1f} TF/s algorithmic", flush=True)
