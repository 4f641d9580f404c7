#!/usr/bin/env python3
"""Headline benchmark: depth frames/sec at 518x518, ViT-L, on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM.
Default workload (BASELINE.json configs[1]): DepthAnythingV2('vitl'), batch of 8 independent
518x518 streams per GPU, memory bank full (6 stored frames, reached during warm-up). Path A is
recurrent per stream, so N GPUs run N independent replicas of that batch ("replicas only",
weak scaling, no data-path collective — SURVEY.md §8e); ranks meet only for the timing barrier.
`--workload clip` runs VideoDepthAnything('vitl') on one 32-frame window per step instead (configs[2]).

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, one JSON line from rank 0.
Extra objects: `roofline` (dominant kernel = the encoder-linear GEMM, timed live with HIP events on the
launch stream inside the timed region) and `cpu_baseline` (the oracle on the host cores, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS_F16 = 2500.0  # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def _pmc_traffic(prec_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(prec_name, {}).get("enc_linear_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=7)
    ap.add_argument("--workload", choices=["stream", "clip", "video", "vstream", "refine5"], default="stream")
    ap.add_argument("--video-frames", type=int, default=256, help="video workload: clip length (BASELINE configs[3]: 256)")
    ap.add_argument("--batch", type=int, default=8, help="streams per GPU (stream) / frames per window (clip: 32)")
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--precision", default=None, help="f16x3 (default, parity-green) | f16 | bf16x3 | bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--single-pass-too", action="store_true", help="also time the f16 single-product mode")
    ap.add_argument("--lanes", type=int, default=None,
                    help="stream workload: HIP-stream lanes the batch is dealt to (default: VDN_STREAMS or 2)")
    a = ap.parse_args()

    if a.lanes is not None:
        os.environ["VDN_STREAMS"] = str(a.lanes)
    lanes = int(os.environ.get("VDN_STREAMS", "2")) if a.workload == "stream" else 1
    if lanes < 2 or a.batch < int(os.environ.get("VDN_LANE_MIN_BATCH", "4")) or a.batch % lanes:
        lanes = 1
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = max(world, 1)
    assert n_gpus == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import vdn
    from vdn import synth
    from vdn.depth_anything_v2 import _precision

    enc = a.encoder
    cfg = vdn.MODEL_CONFIGS[enc]
    H = W = 518
    if a.workload == "stream":
        model = vdn.DepthAnythingV2(**cfg)
        frames_per_step = a.batch
    elif a.workload == "clip":
        model = vdn.VideoDepthAnything(**cfg)
        frames_per_step = 32
    elif a.workload == "refine5":  # v5 depth refiner (BASELINE configs[4]): num_frames 64, [1, 64, 1024, 1024] raw depth clip
        from vdn.video_depth_model_v5 import VideoDepthAnything as RefinerV5
        model = RefinerV5(num_frames=64, **cfg)
        frames_per_step = 64
    elif a.workload == "vstream":  # latency mode: one frame per step against the 31-frame projected K/V cache
        model = vdn.VideoDepthAnything(**cfg)
        frames_per_step = 1
    else:  # whole driver: u8 frames on the host -> windows -> device stitcher -> f32 depth on the host
        model = vdn.VideoDepthAnything(**cfg)
        frames_per_step = a.video_frames
    shapes = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    sd = model.state_dict()
    sd.update(synth.fast_state_dict(shapes, 1234))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    prec_name, _ = _precision(a.precision)
    model.set_precision(prec_name)

    fr = synth.frames_u8(1234 + (rank if a.workload != "video" else 0), min(frames_per_step, 8), H, W)
    if a.workload == "video":
        import numpy as np
        from vdn import util
        from vdn.dist import infer_video_depth_sharded
        video = np.ascontiguousarray(np.tile(fr, ((frames_per_step + 7) // 8, 1, 1, 1))[:frames_per_step])
    x = torch.from_numpy(synth.normalize_frames(fr)).to(dev)
    if a.workload == "refine5":
        xd = torch.from_numpy(synth.depth_clip(1234 + rank, 4, 1024, 1024)).to(dev).repeat(16, 1, 1)[None].contiguous()
    if x.shape[0] < frames_per_step:
        x = x.repeat((frames_per_step + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:frames_per_step]
    if a.workload == "clip":
        x = x[None]
    x = x.contiguous()

    def step():
        if a.workload == "video":  # windows are sharded over the ranks (strong scaling), every rank gets the result
            if dist is not None:
                return infer_video_depth_sharded(model, video, 24, input_size=518)[0]
            return model.infer_video_depth(video, 24, input_size=518)[0]
        if a.workload == "vstream":
            return model.stream_step(x[:1][None])
        if a.workload == "refine5":
            return model.forward(xd)
        return model.forward(x)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def runtimes():
        e = model._engines()
        return [e["rt"]] + [ln["rt"] for ln in (getattr(model, "_lanes", None) or [])[1:]]

    def timed(nsteps, events):
        for rt in runtimes():
            rt.timing = [] if events else None
        sync_all()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        sync_all()
        dt = time.perf_counter() - t0
        ev = []
        for rt in runtimes():
            ev += rt.timing or []
            rt.timing = None
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ev

    if a.workload == "stream":
        # the metric is quoted at steady state (memory bank full, S = 6): whatever W is, fill the bank first
        for _ in range(max(0, 7 - a.warmup)):
            step()
    for _ in range(a.warmup):
        step()
    # With several lanes per-launch durations are not meaningful (the HIP events bracket kernels that share CUs
    # with the other lane's) and recording ~240 event pairs per step costs the timed region ~3 %: the timed
    # region then runs without events, and the roofline figures come from K more steps of the same workload
    # issued on ONE lane right after it (state carried over, memory bank still full).
    dt, ev = timed(a.steps, (not a.no_kernel_events) and lanes == 1)
    fps = frames_per_step * a.steps * (1 if a.workload == "video" else n_gpus) / dt
    dt1 = None
    if lanes > 1 and not a.no_kernel_events:
        os.environ["VDN_STREAMS"] = "1"
        step()
        dt1, ev = timed(a.steps, True)
        os.environ["VDN_STREAMS"] = str(lanes)

    out = {
        "metric": ("refined depth frames/sec at 1024x1024 (v5 refiner), " + enc) if a.workload == "refine5" else
                  ("depth frames/sec at 518x518, ViT-L" if enc == "vitl" else f"depth frames/sec at 518x518, {enc}"),
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
        "scaling": "strong" if a.workload == "video" else "weak", "vs_baseline": None,
        "dtype": prec_name, "data": "synthetic",
        "config": {"workload": ("DepthAnythingV2(%s) batch=%d 518x518 streams/GPU, memory bank full (S=6); replicas per GPU"
                                % (enc, a.batch)) if a.workload == "stream" else
                   ("VideoDepthAnything(%s) one 32-frame 518x518 window per step; one window per GPU" % enc) if a.workload == "clip" else
                   ("video_depth_model_v5.VideoDepthAnything(%s, num_frames=64).forward on a [1,64,1024,1024] raw depth clip (median scale, "
                    "224x224 network, shift + residual); the metric counts refined 1024x1024 frames" % enc) if a.workload == "refine5" else
                   ("VideoDepthAnything(%s).stream_step: one 518x518 frame per step against 31 cached frames (projected K/V cache)" % enc)
                   if a.workload == "vstream" else
                   ("VideoDepthAnything(%s).infer_video_depth on a %d-frame 518x518 u8 clip = %d windows of 32 (host frames in, "
                    "host depth out: H2D, pre-processing, device stitcher and D2H inside the timed region); windows sharded over the GPUs"
                    % (enc, frames_per_step, len(util.window_table(frames_per_step)))),
                   "frames_per_step_per_gpu": frames_per_step, "lanes": lanes, "precision": prec_name,
                   "precision_note": "f16x3 = fp16 hi/lo planes, 3 MFMA products per term (fp32-faithful, parity <=1e-3)"},
    }

    # ---------------- roofline of the dominant kernel (encoder linear GEMMs)
    if ev:
        C = vdn.modules.ENCODERS[enc]["dim"]
        per_launch = a.batch if a.workload == "stream" else (1 if a.workload == "vstream" else (64 if a.workload == "refine5" else 32))  # frames per encoder launch
        M = per_launch * ((16 * 16 + 1) if a.workload == "refine5" else (37 * 37 + 1))
        ms = [s.elapsed_time(e) for (tag, s, e) in ev if tag == "enc_linear"]
        if ms:
            avg_ms = sum(ms) / len(ms)
            flop_per_launch = 6.0 * M * C * C  # mean over qkv/proj/fc1/fc2 = 2*M*12*C^2 / 4
            ach = flop_per_launch / (avg_ms * 1e-3) / 1e12
            nprod = 3 if prec_name.endswith("x3") else 1
            out["roofline"] = {
                "bound": "mfma", "kernel": "gemm_x3_p8_kernel<256x256x32> (fc1) / gemm_x3_big_kernel<192x256x32> (qkv, proj, fc2): the 4 encoder linears" if nprod == 3 else "gemm_kernel<128x128x64> on the 4 encoder linears (qkv, proj, fc1, fc2)",
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_F16, 4),
                "traffic": _pmc_traffic(prec_name), "launches_timed": len(ms), "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_gflop_per_launch": round(flop_per_launch / 1e9, 2),
                "mfma_products_per_term": nprod, "executed_frac": round(nprod * ach / PEAK_TFLOPS_F16, 4)}
            if dt1 is not None:
                out["roofline"]["measured_in"] = (
                    "single-lane pass of %d steps run right after the timed region (%.1f frames/s); the timed region "
                    "deals the batch to %d HIP-stream lanes whose kernels co-run" % (a.steps, frames_per_step * a.steps * n_gpus / dt1, lanes))
        att = [s.elapsed_time(e) for (tag, s, e) in ev if tag == "enc_attn"]
        if att:
            avg = sum(att) / len(att)
            fl = 4.0 * per_launch * (C // 64) * 1370 * 1370 * 64
            nprod = 3 if prec_name.endswith("x3") else 1
            out["attention_kernel"] = {"avg_launch_ms": round(avg, 4), "achieved_tflops": round(fl / (avg * 1e-3) / 1e12, 2),
                                       "frac_of_mfma_peak": round(fl / (avg * 1e-3) / 1e12 / PEAK_TFLOPS_F16, 4),
                                       "executed_frac": round(nprod * fl / (avg * 1e-3) / 1e12 / PEAK_TFLOPS_F16, 4)}

    if a.single_pass_too and prec_name != "f16":
        model.set_precision("f16")
        model = model.to(dev)
        for _ in range(a.warmup):
            step()
        dt2, _ = timed(a.steps, False)
        out["single_pass_f16"] = {"value": round(frames_per_step * a.steps * n_gpus / dt2, 3), "unit": "frames/s",
                                  "note": "1 MFMA product per term; 0.7e-3..2.5e-3 from the fp32 reference (tests/test_gpu_e2e.py)"}
        model.set_precision(prec_name)

    # ---------------- CPU baseline: the oracle on the host cores (rank 0, N == 1)
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        from oracle import ref_cpu as O
        sd_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        xc = x.reshape(-1, 3, H, W)[:1].cpu()
        # the GPU box gives one GPU a 16-core CPU share: oversubscribing all visible cores is slower
        threads = min(torch.get_num_threads(), int(os.environ.get("VDN_CPU_THREADS", "16")))
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            if a.workload == "stream":
                mem = O.MemoryState(6)
                n = 4  # ~10 s on the GPU box's 16-core share
                for _ in range(n):
                    O.depth_anything_v2_forward(sd_cpu, xc, mem, enc)
                sample = f"{n} consecutive DepthAnythingV2({enc}) frames, batch 1, fp32, memory depth 0->{n - 1}"
            else:
                n = 2
                O.video_depth_anything_forward(sd_cpu, xc.repeat(n, 1, 1, 1)[None], enc)
                sample = f"one {n}-frame VideoDepthAnything({enc}) clip, fp32"
            tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / tc, 4), "unit": "frames/s", "cores": threads, "kind": "port",
                               "sample": sample, "seconds": round(tc, 2), "torch": torch.__version__}
        out["speedup_vs_cpu_baseline"] = round(fps / (n / tc), 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
