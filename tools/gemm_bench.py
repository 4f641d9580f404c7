#!/usr/bin/env python3
"""GPU micro-benchmark of vdn_gemm on the path's shapes (HIP events, interleaved rounds)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime
from vdn import pack, _abi

split = "--single" not in sys.argv
rt = Runtime(torch.device("cuda:0"), torch.float16, split=split)
B = 8
M = B * 1370
if "--kscan" in sys.argv:
    shapes = [(f"N1024_K{k}", M, 1024, k, {}) for k in (32, 64, 128, 256, 512, 1024, 2048, 4096)] + \
             [(f"N4096_K{k}", M, 4096, k, dict(act=_abi.ACT_GELU)) for k in (32, 256, 1024)] + \
             [(f"N3072h_K{k}", M, 3072, k, {}) for k in (32, 1024)]
else:
  shapes = [("qkv", M, 3072, 1024, dict(heads=16)), ("proj", M, 1024, 1024, dict(res=True)), ("fc1", M, 4096, 1024, dict(act=_abi.ACT_GELU)), ("fc2", M, 1024, 4096, dict(res=True)),
            ("mem_kv", B * 1369, 2048, 1024, {}), ("conv256@148", B * 148 * 148, 256, 2304, dict(conv=(148, 256))),
            ("conv256@296->128", B * 296 * 296, 128, 2304, dict(conv=(296, 256)))]
torch.manual_seed(0)
res = []
for name, m, n, k, opt in shapes:
    if "conv" in opt:
        side, cin = opt["conv"]
        a = rt.to_half(torch.randn(B * side * side, cin, device="cuda"))
        kw = dict(conv=dict(B=B, H=side, W=side, C=cin, OH=side, OW=side, stride=1))
    else:
        a = rt.to_half(torch.randn(m, k, device="cuda"))
        kw = {}
    w = pack._pad_k(torch.randn(n, k, device="cuda") / math.sqrt(k), rt.prec)
    out = rt.hbuf("o_" + name, (m, n))
    bias = torch.randn(n, device="cuda")
    if opt.get("res"):      # the encoder's residual update: x = x + gamma * (A W^T + b), f32 in place
        xres = torch.zeros(m, n, device="cuda")
        kw.update(gamma=torch.full((n,), 1e-3, device="cuda"), res1=xres)
        out = xres
    if opt.get("heads"):    # the encoder's QKV projection: per-head Q / K rows and V^T columns
        Hh, tok, npad = opt["heads"], m // B, ((m // B + 63) // 64) * 64
        q, kk = (rt.hbuf("hq_" + name + str(i), (B * Hh, npad, 64), zero=True) for i in range(2))
        vt = rt.hbuf("hv_" + name, (B * Hh, 64, npad), zero=True)
        kw.update(store=_abi.ST_HEADS, heads=dict(dst=[q, kk, vt], transposed=[0, 0, 1], heads=Hh, tokens=tok, tpad=npad))
        out = None
    def run():
        rt.gemm(a, w, m, n, k, out=out, bias=bias, act=opt.get("act", 0), **kw)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    med = ts[len(ts) // 2]
    fl = 2.0 * m * n * k
    print(f"{name:18s} M={m:7d} N={n:5d} K={k:5d}  {med*1e3:8.1f} us  alg {fl/med/1e9:7.1f} TF/s  executed {(3 if split else 1)*fl/med/1e9:7.1f} TF/s  (BM={os.environ.get('VDN_GEMM_BM','auto')})", flush=True)
