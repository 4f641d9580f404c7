#!/usr/bin/env python3
"""GPU micro-benchmark of vdn_flash_attn on the path's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime, ceil_to
split = "--single" not in sys.argv
from vdn import _abi
rt = Runtime(torch.device("cuda:0"), torch.float16, split=split)
B, H = 8, 16
for name, nq, nk in (("encoder", 1370, 1370), ("mem_self", 1369, 1369), ("mem_cross_S6", 1369, 6 * 1369)):
    qp, kp = ceil_to(nq, 64), ceil_to(nk, 64)
    q = rt.to_half(torch.randn(B * H, qp, 64, device="cuda"))
    k = rt.to_half(torch.randn(B * H, kp, 64, device="cuda"))
    v = rt.to_half(torch.randn(B * H, 64, kp, device="cuda"))
    o = rt.hbuf("o" + name, (B * nq, H * 64))

    def planes8(t):  # the 8-bit planes the projection epilogue writes (score cross terms on the e5m2 MFMA: the shipped path)
        return torch.cat([t.float().to(torch.float8_e5m2).view(torch.uint8),
                          (t.lo.float() * 1024.0).to(torch.float8_e5m2).view(torch.uint8)], dim=-1).contiguous()

    kw = dict(q8=planes8(q), k8=planes8(k)) if split and "--no-qk8" not in sys.argv else {}
    run = lambda: rt.flash_attn(q, k, v, o, B, H, nq, qp, nk, kp, 0.125, **kw)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); med = ts[len(ts) // 2]
    fl = 4.0 * B * H * nq * nk * 64
    if os.environ.get("VDN_ATTN_STAMPS"):  # timing build (VDN_ATTN_ABL & 64): cycle sums of one wave, 4 stream quarters + tail
        st = o.hi.reshape(-1)[:20].view(torch.int64).cpu().tolist()
        print("   stamps (cycles per tile: 4 quarters of the stream, then bump/barrier tail):", [round(v / ((nk + 63) // 64), 1) for v in st], flush=True)
    print(f"{name:14s} nq={nq} nk={nk}  {med*1e3:8.1f} us  alg {fl/med/1e9:7.1f} TF/s  executed {(3 if split else 1)*fl/med/1e9:7.1f} TF/s", flush=True)
