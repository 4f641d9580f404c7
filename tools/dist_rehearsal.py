#!/usr/bin/env python3
"""The N > 1 path with the REAL kernels on a one-GPU box.

R ranks share cuda:0 under the gloo backend (vdn/dist.py stages device tensors through the host for gloo groups), so
everything the 8-GPU run executes except RCCL itself runs here: the schedule, the subgroups, the encoder tap exchange,
`head_from_planes` with `TemporalEngine.run_sharded` (frames<->pixels all-to-all around every temporal module on the
HIP engine), the gather and the device stitcher. Rank 0 then runs the same clip alone through `infer_video_depth`
and the two depth videos must agree to fp32 summation noise (the sharded modules see other GEMM shapes).

    python tools/dist_rehearsal.py --ranks 2 --frames 50      # 2 whole windows + 1 window frame-sharded over 2 ranks
    python tools/dist_rehearsal.py --ranks 4 --frames 20      # one window frame-sharded over 4 ranks

The parent starts the ranks as a child `torch.distributed.run` BEFORE touching the GPU. Never a performance number.
"""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
sys.path.insert(0, ROOT)
TOL = 2e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--size", type=int, default=140)
    ap.add_argument("--encoder", default="vits")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        assert 2 <= a.ranks <= 4, "at most 6 processes may use the card of a GPU box"
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    import vdn
    from vdn import synth, util
    from vdn.dist import infer_video_depth_sharded, plan_schedule

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import datetime
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("VDN_DIST_TIMEOUT_S", "300"))))
    P, r = dist.get_world_size(), dist.get_rank()
    assert P == a.ranks
    torch.cuda.set_device(0)
    model = vdn.VideoDepthAnything(**vdn.MODEL_CONFIGS[a.encoder])
    sd = model.state_dict()
    sd.update(synth.fast_state_dict([(k, tuple(v.shape)) for k, v in model.named_parameters()], 1234))
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    frames = synth.frames_u8(7, a.frames, a.size, a.size + 28)
    jobs = plan_schedule(len(util.window_table(a.frames)), P)

    d, _ = infer_video_depth_sharded(model, frames, 24, input_size=a.size, all_ranks=True)            # staged path
    d1, _ = infer_video_depth_sharded(model, frames, 24, input_size=a.size, forward=model.forward,     # per-job path
                                      forward_sharded=model.forward_sharded, all_ranks=False)
    ok = np.isfinite(d).all() and d.shape == (a.frames, a.size, a.size + 28)
    err = [0.0, 0.0, 0.0]
    if r == 0:
        ref, _ = model.infer_video_depth(frames, 24, input_size=a.size)
        nrm = float(np.linalg.norm(ref))
        err = [float(np.linalg.norm(d - ref)) / nrm, float(np.linalg.norm(d1 - ref)) / nrm, 0.0]
        print(f"[rehearsal] ranks {P} frames {a.frames} schedule {jobs}")
        print(f"[rehearsal] staged sharded vs single process rel-L2 {err[0]:.2e}; per-job sharded {err[1]:.2e}; depth mean {ref.mean():.4f}")
        ok = ok and err[0] < TOL and err[1] < TOL
    t = torch.from_numpy(d).double().sum().reshape(1)   # every rank got rank 0's stitched clip
    ts = [torch.zeros_like(t) for _ in range(P)]
    dist.all_gather(ts, t)
    same = all(float(x) == float(ts[0]) for x in ts)
    flag = torch.tensor([1 if (ok and same) else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if r == 0:
        print("REHEARSAL OK" if int(flag) else f"REHEARSAL FAILED (ok={ok} same={same} err={err})")
    dist.destroy_process_group()
    sys.exit(0 if int(flag) else 1)


if __name__ == "__main__":
    main()
