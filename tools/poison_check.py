#!/usr/bin/env python3
"""Debug (GPU): run one forward with NaN-poisoned workspace; report NaNs in the output and in buffers."""
import os, sys
os.environ["VDN_POISON"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vdn
from common import inputs, synth_sd
which, enc, H, B = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sd = synth_sd(which, enc)
if which == "A":
    m = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS[enc]); m.load_state_dict(sd); m = m.cuda()
    x = inputs(2 * B, H, H)
    for t in range(2):
        out = m.forward(x[t*B:(t+1)*B].cuda())
        print(f"frame {t}: output NaNs {int(torch.isnan(out).sum())} / {out.numel()}")
else:
    T = 4
    m = vdn.VideoDepthAnything(**vdn.MODEL_CONFIGS[enc]); m.load_state_dict(sd); m = m.cuda()
    out = m.forward(inputs(T * B, H, H).reshape(B, T, 3, H, H).cuda())
    print(f"output NaNs {int(torch.isnan(out).sum())} / {out.numel()}")
rt = m._eng["rt"]
for (name, shape, dt), t in sorted(rt._bufs.items(), key=lambda kv: kv[0][0]):
    if t.is_floating_point():
        n = int(torch.isnan(t).sum())
        if n:
            print(f"  buffer {name:16s} {str(shape):28s} NaNs {n} ({100.0*n/t.numel():.2f}%)")
