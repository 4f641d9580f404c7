#!/usr/bin/env python3
"""Race screen (GPU box): the counted-vmcnt ping-pong GEMM and the pipelined attention, many repetitions at
several sizes, every run compared bit for bit with the first (an ordering mistake shows up as rare differing tiles)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime, ceil_to
from vdn import pack, _abi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rt = Runtime(torch.device("cuda:0"), torch.float16, split=True)
torch.manual_seed(0)
bad = 0
for bm in ("256", "192", "128"):
    _abi.set_tuning(force_bm=int(bm))
    for (M, N, K) in ((10960, 4096, 1024), (5480, 1024, 4096), (1370, 3072, 1024), (2050, 512, 96)):
        a = rt.to_half(torch.randn(M, K, device="cuda"))
        w = pack._pad_k(torch.randn(N, K, device="cuda") / math.sqrt(K), rt.prec)
        bias = torch.randn(N, device="cuda")
        out = rt.hbuf(f"rs_{M}_{N}", (M, N))
        rt.gemm(a, w, M, N, K, out=out, bias=bias, act=_abi.ACT_GELU)
        h0, l0 = out.hi.clone(), out.lo.clone()
        diff = 0
        for _ in range(reps):
            rt.gemm(a, w, M, N, K, out=out, bias=bias, act=_abi.ACT_GELU)
            diff += int(not (torch.equal(out.hi, h0) and torch.equal(out.lo, l0)))
        bad += diff
        print(f"gemm BM={bm} M={M} N={N} K={K}: {diff} of {reps} runs differ", flush=True)
_abi.set_tuning(force_bm=0)
for (B, H, nq, nk) in ((8, 16, 1370, 1370), (4, 16, 1369, 8214), (2, 6, 150, 200), (1, 16, 361, 1369)):
    qp, kp = ceil_to(nq, 64), ceil_to(nk, 64)
    q = rt.to_half(torch.randn(B * H, qp, 64, device="cuda"))
    k = rt.to_half(torch.randn(B * H, kp, 64, device="cuda"))
    v = rt.to_half(torch.randn(B * H, 64, kp, device="cuda"))
    o = rt.hbuf(f"rs_o{nq}_{nk}", (B * nq, H * 64))
    rt.flash_attn(q, k, v, o, B, H, nq, qp, nk, kp, 0.125)
    h0, l0 = o.hi.clone(), o.lo.clone()
    diff = 0
    for _ in range(reps):
        rt.flash_attn(q, k, v, o, B, H, nq, qp, nk, kp, 0.125)
        diff += int(not (torch.equal(o.hi, h0) and torch.equal(o.lo, l0)))
    bad += diff
    print(f"attn B={B} H={H} nq={nq} nk={nk}: {diff} of {reps} runs differ", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
