#!/usr/bin/env python3
"""How long the host takes to ISSUE one step of the default bench workload (DepthAnythingV2 vitl, batch 8, two lanes) against
how long the GPU takes to run it: if the host needs a large part of the step, the second lane's kernels are queued late and
the lanes overlap less than they could. Prints per-step host issue time (no sync inside), GPU step time, launches per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
import vdn
from vdn import synth

dev = torch.device("cuda:0")
model = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS["vitl"])
shapes = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
sd = model.state_dict()
sd.update(synth.fast_state_dict(shapes, 1234))
model.load_state_dict(sd, strict=True)
model = model.to(dev).eval()
x = torch.from_numpy(synth.normalize_frames(synth.frames_u8(1234, 8, 518, 518))).to(dev)
for _ in range(8):
    model.forward(x)
torch.cuda.synchronize()
n = 0
rts = [model._engines()["rt"]] + [ln["rt"] for ln in (getattr(model, "_lanes", None) or [])[1:]]
orig = [rt._launch for rt in rts]


def counted(f):
    def g(*a, **k):
        global n
        n += 1
        return f(*a, **k)
    return g


for rt, f in zip(rts, orig):
    rt._launch = counted(f)
host, gpu = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.forward(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    gpu.append((t2 - t0) * 1e3)
host.sort(); gpu.sort()
print(f"launches per step {n // 10}; host issue time median {host[5]:.2f} ms (min {host[0]:.2f}); issue + drain median {gpu[5]:.2f} ms")
# back-to-back steps (the bench loop): the host may run ahead
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    model.forward(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"10 steps back to back: host done after {(t1 - t0) * 1e3:.1f} ms, GPU after {(t2 - t0) * 1e3:.1f} ms")
