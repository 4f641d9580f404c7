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
# 8-bit cross-term kernel (5-slot ring, counted waits in units, two tile heights): K-tile-major planes, the three hot flavours
from vdn.runtime import HL


def kt16(t, rows, K):
    return t.reshape(rows, K // 32, 32).permute(1, 0, 2).contiguous()


def kt8(t, rows, K):
    return t.reshape(rows, K // 64, 64).permute(1, 0, 2).contiguous()


for bm in (256, 192):
    _abi.set_tuning(force_bm=bm)
    for (M, N, K) in ((10960, 4096, 1024), (5480, 1024, 4096), (10960, 1024, 64), (4100, 3072, 192)):
        a = rt.to_half(torch.randn(M, K, device="cuda"))
        w = pack.linear(torch.randn(N, K, device="cuda") / math.sqrt(K), rt.prec)
        ak, a8k = HL(kt16(a.hi, M, K)), pack.planes8(a, kt=True)
        x8 = pack.X8(w)
        bias = torch.randn(N, device="cuda")
        kt = dict(a8=a8k, w8=x8.p8, a_kt=True, w_kt=True)
        oh, o8 = HL(torch.zeros(M, N, dtype=torch.float16, device="cuda")), torch.zeros(2, M, N, dtype=torch.uint8, device="cuda")
        res0 = torch.randn(M, N, device="cuda")
        res = res0.clone()

        def run():
            rt.gemm(ak, HL(x8.hi), M, N, K, out=oh, out8=o8, out_kt=True, bias=bias, act=_abi.ACT_GELU, **kt)
            res.copy_(res0)
            rt.gemm(ak, HL(x8.hi), M, N, K, out=res, bias=bias, gamma=bias, res1=res, **kt)

        run()
        h0, p0, r0 = oh.hi.clone(), o8.clone(), res.clone()
        diff = 0
        for _ in range(reps // 2):
            run()
            diff += int(not (torch.equal(oh.hi, h0) and torch.equal(o8, p0) and torch.equal(res, r0)))
        bad += diff
        print(f"gemm_x8 BM={bm} M={M} N={N} K={K} (GELU planes + residual): {diff} of {reps // 2} runs differ", flush=True)
    # head-split epilogues of the cross-term kernel: lane-half swap of the 8-bit planes, and the RoPE'd flavour
    side, Bh, Hh, K = 37, 4, 4, 256
    P, C = side * side, Hh * 64
    tp, M = ceil_to(P, 64), Bh * P
    a = rt.to_half(torch.randn(M, K, device="cuda"))
    ws, bs = [torch.randn(C, K, device="cuda") / math.sqrt(K) for _ in range(3)], [torch.randn(C, device="cuda") for _ in range(3)]
    cs = pack.rope_table(side, side, 64, device="cuda")
    a8 = pack.planes8(a)
    for rope in (0, 1):
        wp, bp = pack.cat_proj(ws, bs, [rope, rope, 0], rt.prec)
        w8 = pack.planes8(wp)
        q, k = rt.hbuf(f"rs_q{rope}", (Bh * Hh, tp, 64), zero=True), rt.hbuf(f"rs_k{rope}", (Bh * Hh, tp, 64), zero=True)
        vt = rt.hbuf(f"rs_vt{rope}", (Bh * Hh, 64, tp), zero=True)
        q8, k8 = rt.qk8(f"rs_q8{rope}", Bh * Hh, tp), rt.qk8(f"rs_k8{rope}", Bh * Hh, tp)
        hd = dict(dst=[q, k, vt], dst8=[q8, k8, None], transposed=[0, 0, 1], heads=Hh, tokens=P, tpad=tp)
        if rope:
            hd.update(rope=[1, 1, 0], rope_cs=cs, rope_mod=P)

        def run_h():
            rt.gemm(a, wp, M, 3 * C, K, bias=bp, store=_abi.ST_HEADS, heads=hd, a8=a8, w8=w8)

        run_h()
        ref = [t.clone() for t in (q.hi, k.hi, vt.hi, q8, k8)]
        diff = 0
        for _ in range(reps // 2):
            run_h()
            diff += int(not all(torch.equal(x, y) for x, y in zip((q.hi, k.hi, vt.hi, q8, k8), ref)))
        bad += diff
        print(f"gemm_x8 BM={bm} head split{' + RoPE' if rope else ''} M={M} N={3 * C} K={K}: {diff} of {reps // 2} runs differ", flush=True)
_abi.set_tuning(force_bm=0)
for (B, H, nq, nk) in ((8, 16, 1370, 1370), (4, 16, 1369, 8214), (2, 6, 150, 200), (1, 16, 361, 1369)):
    qp, kp = ceil_to(nq, 64), ceil_to(nk, 64)
    q = rt.to_half(torch.randn(B * H, qp, 64, device="cuda"))
    k = rt.to_half(torch.randn(B * H, kp, 64, device="cuda"))
    v = rt.to_half(torch.randn(B * H, 64, kp, device="cuda"))
    o = rt.hbuf(f"rs_o{nq}_{nk}", (B * nq, H * 64))

    def planes8(t):  # the 8-bit planes the projection epilogue would have written (score cross terms on the e5m2 MFMA)
        return torch.cat([t.float().to(torch.float8_e5m2).view(torch.uint8),
                          (t.lo.float() * 1024.0).to(torch.float8_e5m2).view(torch.uint8)], dim=-1).contiguous()

    for tag, kw in (("3-product scores", {}), ("8-bit cross terms", dict(q8=planes8(q), k8=planes8(k)))):
        rt.flash_attn(q, k, v, o, B, H, nq, qp, nk, kp, 0.125, **kw)
        h0, l0 = o.hi.clone(), o.lo.clone()
        diff = 0
        for _ in range(reps):
            rt.flash_attn(q, k, v, o, B, H, nq, qp, nk, kp, 0.125, **kw)
            diff += int(not (torch.equal(o.hi, h0) and torch.equal(o.lo, l0)))
        bad += diff
        print(f"attn ({tag}) B={B} H={H} nq={nq} nk={nk}: {diff} of {reps} runs differ", flush=True)
# fused depth tail (LDS-DMA source patch + register-resident weights) and the persistent LayerNorm
x = torch.randn(8 * 296 * 296, 128, device="cuda")
w = pack.conv3x3_taps(torch.randn(32, 128, 3, 3, device="cuda") / math.sqrt(9 * 128), rt.prec)
b2, w1 = torch.randn(32, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.3
d = torch.empty(8, 518, 518, device="cuda")
rt.depth_tail(x, w, b2, w1, 0.2, d, 8, 296, 296, 128, 518, 518, True)
d0 = d.clone()
diff = 0
for _ in range(reps // 4):
    rt.depth_tail(x, w, b2, w1, 0.2, d, 8, 296, 296, 128, 518, 518, True)
    diff += int(not torch.equal(d, d0))
bad += diff
print(f"depth_tail B=8 296->518: {diff} of {reps // 4} runs differ", flush=True)
t = torch.randn(10960, 1024, device="cuda")
g, be = torch.randn(1024, device="cuda"), torch.randn(1024, device="cuda")
oh = rt.hbuf("rs_ln", (10960, 1024))
rt.layernorm(t, 10960, 1024, g, be, 1e-6, out_h=oh)
h0, l0 = oh.hi.clone(), oh.lo.clone()
diff = 0
for _ in range(reps):
    rt.layernorm(t, 10960, 1024, g, be, 1e-6, out_h=oh)
    diff += int(not (torch.equal(oh.hi, h0) and torch.equal(oh.lo, l0)))
bad += diff
print(f"layernorm 10960 x 1024: {diff} of {reps} runs differ", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
