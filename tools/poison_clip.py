#!/usr/bin/env python3
"""Debug (GPU): one 32-frame ViT-L window with a NaN-poisoned workspace (VDN_POISON=1): a kernel that consumes a buffer nobody
wrote shows up as NaNs in the depth maps. Companion of tools/poison_check.py for the window path (temporal modules, big launches)."""
import os, sys
os.environ["VDN_POISON"] = "1"
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch, vdn
from vdn import synth
m = vdn.VideoDepthAnything(**vdn.MODEL_CONFIGS["vitl"])
shapes = [(k, tuple(v.shape)) for k, v in m.named_parameters()]
sd = m.state_dict(); sd.update(synth.fast_state_dict(shapes, 1234)); m.load_state_dict(sd, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(synth.normalize_frames(synth.frames_u8(1234, 8, 518, 518))).cuda().repeat(4, 1, 1, 1)[None].contiguous()
for t in range(2):
    out = m.forward(x)
    print(f"pass {t}: output NaNs {int(torch.isnan(out).sum())} / {out.numel()}, finite {bool(torch.isfinite(out).all())}")
