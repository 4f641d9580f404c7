"""Host <-> device copy rates on the GPU box (pageable vs pinned): the numbers behind vdn.util.to_host (profiles/r02_d2h_probe.log)."""
import time, torch
x = torch.randn(256, 518, 518, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); y = x.cpu(); t1 = time.perf_counter()
    print(f"pageable .cpu(): {1e3*(t1-t0):.1f} ms  ({x.numel()*4/(t1-t0)/1e9:.1f} GB/s)")
t0 = time.perf_counter(); p = torch.empty(x.shape, dtype=x.dtype, pin_memory=True); t1 = time.perf_counter()
print(f"pinned alloc: {1e3*(t1-t0):.1f} ms")
for _ in range(2):
    t0 = time.perf_counter(); p.copy_(x); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"pinned copy_: {1e3*(t1-t0):.1f} ms  ({x.numel()*4/(t1-t0)/1e9:.1f} GB/s)")
import numpy as np
u = (torch.rand(256, 518, 518, 3) * 255).to(torch.uint8).numpy()
for _ in range(2):
    t0 = time.perf_counter(); d = torch.from_numpy(u).to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"pageable H2D of the u8 clip: {1e3*(t1-t0):.1f} ms ({u.nbytes/(t1-t0)/1e9:.1f} GB/s)")
