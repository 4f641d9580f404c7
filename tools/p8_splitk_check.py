import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-depth-normal-v2_amd"))
from vdn.runtime import Runtime
from vdn import pack, _abi
rt = Runtime(torch.device("cuda:0"), torch.float16, split=True)
torch.manual_seed(0)
for (M, N, K) in ((5480, 1024, 4096), (10960, 1024, 4096), (1370, 1024, 2048)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K)
    b = torch.randn(N, device="cuda"); g = torch.randn(N, device="cuda"); x = torch.randn(M, N, device="cuda")
    ref = (x.double() + (a.double() @ w.double().t() + b.double()) * g.double()).float()
    A, W = rt.to_half(a), pack.linear(w, rt.prec)
    for ks in ("0", "2", "3"):
        _abi.set_tuning(splitk_p8=int(ks))
        xd = x.clone()
        rt.gemm(A, W, M, N, K, out=xd, bias=b, gamma=g, res1=xd)
        err = ((xd.double() - ref.double()).norm() / ref.double().norm()).item()
        ts = []
        for _ in range(8):
            xd = x.clone(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rt.gemm(A, W, M, N, K, out=xd, bias=b, gamma=g, res1=xd); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ts.sort()
        print(f"M={M} N={N} K={K} ks={ks}: rel err {err:.2e}  {ts[len(ts)//2]*1e3:.1f} us", flush=True)
