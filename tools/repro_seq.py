#!/usr/bin/env python3
"""Debug (GPU): run the A/vits 518 stream first (like the first e2e test), drop the model, then the 266 batch-2 diag."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vdn
from common import inputs, rel_l2, synth_sd
from oracle import ref_cpu as O
enc = "vits"
sd = synth_sd("A", enc)
def make():
    m = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS[enc]); m.load_state_dict(sd, strict=True); return m.to("cuda").eval()
m = make()
x = inputs(8, 518, 518)
for t in range(8):
    m.forward(x[t:t+1].cuda(), _pre_relu=True)
del m; gc.collect()
m = make()
B, H = 2, 266
P, C, F, ph = 19 * 19, 384, 64, 19
x = inputs(2 * B, H, H)
mem = O.MemoryState(6)
for t in range(2):
    tr = {}
    with torch.no_grad():
        ref = O.depth_anything_v2_forward(sd, x[t*B:(t+1)*B], mem, enc, pre_relu=True, trace=tr)
    got = m.forward(x[t*B:(t+1)*B].cuda(), _pre_relu=True).cpu()
    rt = m._eng["rt"]
    print(f"--- frame {t}  nan in out: {int(torch.isnan(got).sum())}")
    print(f"tokens0 {rel_l2(rt.fbuf('tokens', (B*(P+1), C)).cpu()*0+0, tr['tokens0'].reshape(-1, C)*0):.1e} (n/a)")
    for i in range(3):
        print(f"tap{i}    {rel_l2(rt.hbuf(f'tap{i}', (B*P, C)).float().cpu(), tr['taps'][i][0].reshape(B*P, C)):.2e}")
    print(f"tap3    {rel_l2(rt.fbuf('tap_last_f32', (B*P, C)).cpu(), tr['taps'][3][0].reshape(B*P, C)):.2e}")
    print(f"mem_out {rel_l2(rt.hbuf('mem_out', (B*P, C)).float().cpu(), tr['mem_out'].reshape(B*P, C)):.2e}")
    for i, s in ((4, ph), (3, 2*ph), (2, 4*ph), (1, 8*ph)):
        g = rt.hbuf(f"path{i}", (B*s*s, F)).float().cpu().reshape(B, s, s, F).permute(0, 3, 1, 2)
        print(f"path_{i}  {rel_l2(g, tr[f'path_{i}']):.2e}")
    print(f"pre     {rel_l2(got, ref):.2e}")
rt = m._eng["rt"]
print("buffers with non-finite values (after frame 1):")
for (name, shape, dt), t in sorted(rt._bufs.items(), key=lambda kv: kv[0][0]):
    if t.is_floating_point():
        n = int((~torch.isfinite(t)).sum())
        if n:
            idx = (~torch.isfinite(t)).reshape(-1).nonzero()[:3].reshape(-1).tolist()
            print(f"  {name:16s} {str(shape):26s} bad {n} ({100.0*n/t.numel():.3f}%) first flat idx {idx}")
l2 = rt.hbuf("l2", (2888, 96))
print("l2.hi rows 2846..2853, cols 0..5:\n", l2.hi[2846:2854, :6].float().cpu())
print("l2.lo rows 2846..2853, cols 0..5:\n", l2.lo[2846:2854, :6].float().cpu())
p1 = rt.hbuf("proj1", (722, 96))
print("proj1 finite:", bool(torch.isfinite(p1.float()).all()), "rows 703/721 absmax", float(p1.float()[703].abs().max()), float(p1.float()[721].abs().max()))
for (name, shape, dt), t in sorted(rt._bufs.items(), key=lambda kv: kv[1].data_ptr()):
    print(f"  {t.data_ptr():#x} .. {t.data_ptr() + t.numel()*t.element_size():#x}  {name} {shape}")
