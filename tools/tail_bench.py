#!/usr/bin/env python3
"""Time vdn_depth_tail at the ViT-L batch-8 size (GPU box). VDN_LIB selects a variant library."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
import torch
from vdn.runtime import Runtime
from vdn import pack
rt = Runtime(torch.device("cuda:0"), torch.float16, split=True)
B, IH, C, OH = 8, 296, 128, 518
x = torch.randn(B * IH * IH, C, device="cuda")
w = pack.conv3x3_taps(torch.randn(32, C, 3, 3, device="cuda") / math.sqrt(9 * C), rt.prec)
b2, w1 = torch.randn(32, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.3
d = torch.empty(B, OH, OH, device="cuda")
ts = []
for i in range(12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); rt.depth_tail(x, w, b2, w1, 0.2, d, B, IH, IH, C, OH, OH, True); e.record()
    torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
ts = sorted(ts[2:])
fl = 2.0 * B * OH * OH * 32 * 9 * C
print(f"depth_tail B={B} {IH}->{OH} C={C}: median {ts[len(ts)//2]*1e3:.1f} us  ({fl/ts[len(ts)//2]/1e9:.1f} TF/s algorithmic)  lib={os.environ.get('VDN_LIB','default')}")
