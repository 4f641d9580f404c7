#!/usr/bin/env python3
"""The encoder linears on the 8-bit cross-term kernel (K-tile-major planes, as the engine feeds it) against the 3-product
kernels, interleaved rounds in one process (GPU box). M=... B=... override the row count / batch."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime, HL
from vdn import pack, _abi
rt = Runtime(torch.device("cuda:0"), torch.float16, split=True)
if os.environ.get("BM"):
    _abi.set_tuning(force_bm=int(os.environ["BM"]))
torch.manual_seed(0)
B = int(os.environ.get("B", "8"))
M = int(os.environ.get("M", B * 1370))
tok = M // B


def kt16(t, rows, K):
    return t.reshape(rows, K // 32, 32).permute(1, 0, 2).contiguous()


def kt8(t, rows, K):
    return t.reshape(rows, K // 64, 64).permute(1, 0, 2).contiguous()


cases = []
for name, N, K, kind in (("qkv", 3072, 1024, "heads"), ("proj", 1024, 1024, "res"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "res")):
    a = rt.to_half(torch.randn(M, K, device="cuda"))
    w = pack.linear(torch.randn(N, K, device="cuda") / math.sqrt(K), rt.prec)
    ak, a8k = HL(kt16(a.hi, M, K)), pack.planes8(a, kt=True)
    x8 = pack.X8(w)
    bias = torch.randn(N, device="cuda")
    kw3, kw8 = {}, dict(a8=a8k, w8=x8.p8, a_kt=True, w_kt=True)
    if kind == "res":
        out = torch.zeros(M, N, device="cuda")
        kw3 = dict(out=out, bias=bias, res1=out, gamma=torch.full((N,), 1e-3, device="cuda"))
        kw8.update(kw3)
    elif kind == "gelu":
        kw3 = dict(out=rt.hbuf("xb3_" + name, (M, N)), bias=bias, act=_abi.ACT_GELU)
        kw8.update(out=HL(rt.buf("xb8_" + name, (M, N), torch.float16)), out8=rt.buf("xb88_" + name, (2, M, N), torch.uint8),
                   out_kt=True, bias=bias, act=_abi.ACT_GELU)
    else:
        Hh, npad = 16, (tok + 63) // 64 * 64
        q, k = (rt.hbuf(f"xb_q{i}", (B * Hh, npad, 64), zero=True) for i in range(2))
        vt = rt.hbuf("xb_vt", (B * Hh, 64, npad), zero=True)
        q8, k8 = rt.qk8("xb_q8", B * Hh, npad), rt.qk8("xb_k8", B * Hh, npad)
        kw3 = dict(bias=bias, store=_abi.ST_HEADS, heads=dict(dst=[rt.qk_dst(q, q8), rt.qk_dst(k, k8), rt.v_dst(vt)], dst8=[q8, k8, None],
                                                              transposed=[0, 0, 1], heads=Hh, tokens=tok, tpad=npad))
        kw8.update(kw3)
    cases.append((name, N, K, (a, w, kw3), (ak, HL(x8.hi), kw8)))
times = {}
for rnd in range(9):
    for name, N, K, c3, c8 in cases:
        for mode, (a, w, kw) in (("x3", c3), ("x8", c8)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rt.gemm(a, w, M, N, K, **kw); e.record(); torch.cuda.synchronize()
            if rnd >= 2:
                times.setdefault((name, mode), []).append(s.elapsed_time(e))
tot = {"x3": 0.0, "x8": 0.0}
for name, N, K, _, _ in cases:
    line = f"{name:5s} M={M} N={N:5d} K={K:5d}"
    for mode in ("x3", "x8"):
        ts = sorted(times[(name, mode)]); med = ts[len(ts) // 2]; tot[mode] += med
        line += f" | {mode}: {med*1e3:7.1f} us (min {ts[0]*1e3:6.1f}) {2.0*M*N*K/med/1e9:6.1f} TF/s alg"
    print(line, flush=True)
if os.environ.get("X8_CLOCK"):   # diagnostic library (-DVDN_X8_ABL=8): shader cycles / 100 MHz ticks of the last x8 launch's main loop
    ws = rt.buf("splitk_ws", (32 * 1024 * 1024,), torch.float32).view(torch.int64)[:12].cpu().tolist()
    pairs = [(ws[i], ws[i + 1]) for i in range(0, 12, 2) if ws[i + 1] > 0]
    print("in-kernel clock (GHz) of the last launch's main loop, per reporting workgroup:", [round(c / r * 0.1, 3) for c, r in pairs], "loop us:", [round(r / 100.0, 1) for c, r in pairs])
print("block total: " + ", ".join(f"{k} {v*1e3:.1f} us" for k, v in tot.items()), flush=True)
