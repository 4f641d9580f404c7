#!/usr/bin/env python3
"""Experiment (GPU): one batch-8 forward vs two batch-4 forwards on two HIP streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch, vdn
from vdn import synth
cfg = vdn.MODEL_CONFIGS["vitl"]
def make():
    m = vdn.DepthAnythingV2(**cfg)
    sd = m.state_dict(); sd.update(synth.fast_state_dict([(k, tuple(v.shape)) for k, v in m.named_parameters()], 1234))
    m.load_state_dict(sd); return m.cuda().eval()
x = torch.from_numpy(synth.normalize_frames(synth.frames_u8(1, 8, 518, 518))).cuda()
m8 = make()
for _ in range(7): m8.forward(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m8.forward(x)
torch.cuda.synchronize(); t8 = (time.perf_counter() - t0) / 5
print(f"one stream  B=8: {t8*1e3:.2f} ms/step  {8/t8:.1f} fps", flush=True)
del m8; torch.cuda.empty_cache()
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ms = [make() for _ in range(nS)]
ss = [torch.cuda.Stream() for _ in range(nS)]
xs = list(x.chunk(nS))
def step():
    for m, s, xx in zip(ms, ss, xs):
        with torch.cuda.stream(s):
            m.forward(xx)
for _ in range(7): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 5
print(f"{nS} streams B={8//nS} each: {t2*1e3:.2f} ms/step  {8/t2:.1f} fps", flush=True)
