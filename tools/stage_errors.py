#!/usr/bin/env python3
"""Diagnostic (GPU box): per-stage relative error of the HIP path against the oracle trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import vdn
from common import inputs, rel_l2, synth_sd
from oracle import ref_cpu as O

enc = sys.argv[1] if len(sys.argv) > 1 else "vits"
which = sys.argv[2] if len(sys.argv) > 2 else "A"
H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 518
BB = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sd = synth_sd(which, enc)
C = O.ENCODERS[enc]["dim"]
F = O.MODEL_CONFIGS[enc]["features"]
ph = pw = H // 14
P = ph * pw
if which == "A":
    m = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS[enc]); m.load_state_dict(sd); m = m.cuda()
    x = inputs(2 * BB, H, W)
    mem = O.MemoryState(6)
    for t in range(2):
        tr = {}
        with torch.no_grad():
            ref = O.depth_anything_v2_forward(sd, x[t*BB:(t+1)*BB], mem, enc, pre_relu=True, trace=tr)
        got = m.forward(x[t*BB:(t+1)*BB].cuda(), _pre_relu=True).cpu()
        rt = m._eng["rt"]
        print(f"--- {enc} A frame {t}")
        for i in range(3):
            print(f"tap{i}    {rel_l2(rt.hbuf(f'tap{i}', (BB * P, C)).float().cpu(), tr['taps'][i][0].reshape(BB * P, C)):.2e}")
        print(f"tap3    {rel_l2(rt.fbuf('tap_last_f32', (BB * P, C)).cpu(), tr['taps'][3][0].reshape(BB * P, C)):.2e}")
        print(f"mem_out {rel_l2(rt.hbuf('mem_out', (BB * P, C)).float().cpu(), tr['mem_out'].reshape(BB * P, C)):.2e}")
        for i, s in ((4, ph), (3, 2 * ph), (2, 4 * ph), (1, 8 * ph)):
            g = rt.hbuf(f"path{i}", (BB * s * s, F)).float().cpu().reshape(BB, s, s, F).permute(0, 3, 1, 2)
            print(f"path_{i}  {rel_l2(g, tr[f'path_{i}']):.2e}")
        print(f"pre     {rel_l2(got, ref):.2e}   post {rel_l2(torch.relu(got), torch.relu(ref)):.2e}")
else:
    T = 4
    m = vdn.VideoDepthAnything(**vdn.MODEL_CONFIGS[enc]); m.load_state_dict(sd); m = m.cuda()
    x = inputs(T, H, W).reshape(1, T, 3, H, W)
    tr = {}
    with torch.no_grad():
        ref = O.video_depth_anything_forward(sd, x, enc, pre_relu=True, trace=tr)
    got = m.forward(x.cuda(), _pre_relu=True).cpu()
    rt = m._eng["rt"]
    for i in range(4):
        print(f"tap{i}    {rel_l2(rt.hbuf(f'tap{i}', (T * P, C)).float().cpu(), tr['taps'][i][0].reshape(T * P, C)):.2e}")
    for i, s in ((4, ph), (3, 2 * ph), (2, 4 * ph), (1, 8 * ph)):
        g = rt.hbuf(f"path{i}", (T * s * s, F)).float().cpu().reshape(T, s, s, F).permute(0, 3, 1, 2)
        print(f"path_{i}  {rel_l2(g, tr[f'path_{i}']):.2e}   (note path4/3 buffers hold pre-temporal values)")
    print(f"pre     {rel_l2(got, ref):.2e}   post {rel_l2(torch.relu(got), torch.relu(ref)):.2e}")
