#!/usr/bin/env python3
"""Soak (GPU box): 300 consecutive batch-8 forwards with changing inputs; output must stay finite, the memory-bank
bookkeeping consistent and the allocator's footprint flat (nothing is allocated per call after warm-up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
import vdn
from vdn import synth
dev = torch.device("cuda:0")
model = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS["vitl"])
sd = model.state_dict()
sd.update(synth.fast_state_dict([(k, tuple(v.shape)) for k, v in model.named_parameters()], 1234))
model.load_state_dict(sd)
model = model.to(dev).eval()
fr = torch.from_numpy(synth.normalize_frames(synth.frames_u8(7, 8, 518, 518))).to(dev)
base = None
for i in range(300):
    x = torch.roll(fr, shifts=i % 8, dims=0) * (1.0 + 0.01 * (i % 5))
    d = model.forward(x)
    if i % 50 == 49:
        torch.cuda.synchronize()
        assert torch.isfinite(d).all(), i
        mem = torch.cuda.memory_allocated(dev)
        if base is None:
            base = mem
        print(f"step {i + 1}: depth mean {d.mean().item():.4f} max {d.max().item():.3f} allocated {mem / 2**30:.2f} GiB", flush=True)
        assert mem <= base * 1.01, (mem, base)
print("SOAK OK")
