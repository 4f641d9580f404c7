#!/usr/bin/env python3
"""Headline benchmark: depth frames/sec at 518x518, ViT-L, on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input.

* N = 1 (default workload `stream`, BASELINE.json configs[1]): DepthAnythingV2('vitl'), batch of 8 independent
  518x518 streams already resident in HBM, memory bank full (6 stored frames, reached during warm-up).
* N > 1 (default workload `video`, configs[3]): VideoDepthAnything('vitl').infer_video_depth on ONE 256-frame
  518x518 clip = 12 windows of 32 frames, sharded over the N GPUs (STRONG scaling: the clip is fixed). Full rounds
  run one window per GPU; the windows of the last partial round are frame-sharded over groups of GPUs with an
  all-to-all over RCCL around each temporal module (vdn/dist.py). `value` = 256 * K / time.
  Path A is recurrent per stream, so `--workload stream --gpus N` runs N independent replicas ("replicas only").

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment STARTS the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py`, before this process touches the GPU) and exits
with the child's status; under torchrun it is one of the ranks. `--gpus` must equal WORLD_SIZE.

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize() on
both sides, MAX over ranks, one JSON line from rank 0. Extra objects: `roofline` (dominant kernel = the
encoder-linear GEMMs, timed live with HIP events on the launch stream), `cpu_baseline` (the oracle on the host
cores, rank 0, N = 1) and `pcie_inclusive` (the same step with H2D of the frames and D2H of the maps inside).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_F16 = 2500.0  # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
# what a register-only loop of v_mfma_f32_32x32x16_f16 sustains on this part under its power-managed clock (SURVEY.md §8d asks
# for both): tools/micro/mfma_scale_probe.hip, profiles/r02_mfma_scale_probe.log
PEAK_MEASURED_TFLOPS_F16 = 1808.0


def _pmc_traffic(prec_name):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).
    Not measured by this process: the JSON names the file in `traffic_source`."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                v = json.load(f).get(prec_name, {}).get("enc_linear_bytes_per_launch")
            if v is not None:
                return v, "profiles/" + name
        except (OSError, ValueError):
            continue
    return None, None


def _attn_mfma_busy():
    """Matrix-pipe busy share of the attention kernel from the newest committed PMC passes (tools/pmc_attn.sh over
    tools/attn_bench.py): mean of SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 (north star: >= 0.40).
    Not measured by this process."""
    import re
    for name in ("r03_pmc_attention.txt", "r02_pmc_attention_final.txt"):
        try:
            txt = open(os.path.join(ROOT, "profiles", name)).read()
        except OSError:
            continue
        vals = {}
        for m in re.finditer(r"grid\s+(\d+)\s+(SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES)\s+n=\s*\d+\s+mean\s+([0-9.]+)", txt):
            vals.setdefault(int(m.group(1)), {})[m.group(2)] = float(m.group(3))
        fr = [(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["SQ_BUSY_CYCLES"] / 32.0) for v in vals.values() if len(v) == 2 and v["SQ_BUSY_CYCLES"] > 0]
        if fr:
            return {"value": round(sum(fr) / len(fr), 3), "source": "profiles/" + name,
                    "definition": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 over SQ_BUSY_CYCLES / 32, rocprofv3 --pmc (separate passes), mean over the bench shapes"}
    return None


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10; 3 for the video workload)")
    ap.add_argument("--warmup", type=int, default=None, help="warm-up steps (default 7; 1 for the video workload)")
    ap.add_argument("--workload", choices=["stream", "clip", "video", "vstream", "refine5"], default=None,
                    help="default: stream for --gpus 1, video (256-frame clip, strong scaling) for --gpus > 1")
    ap.add_argument("--video-frames", type=int, default=256, help="video workload: clip length (BASELINE configs[3]: 256)")
    ap.add_argument("--batch", type=int, default=8, help="streams per GPU (stream workload)")
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--precision", default=None, help="f16x3 (default, parity-green) | f16 | bf16x3 | bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--single-pass-too", action="store_true", help="also time the f16 single-product mode")
    ap.add_argument("--lanes", type=int, default=None,
                    help="stream workload: HIP-stream lanes the batch is dealt to (default: VDN_STREAMS or 2)")
    ap.add_argument("--stub", action="store_true",
                    help="plumbing rehearsal without a GPU: gloo backend, CPU tensors, a stand-in network "
                         "(tests/test_dist.py); never a performance number")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="rehearsal of the N > 1 path with the real kernels on a one-GPU box: every rank on cuda:0, gloo "
                         "backend with host-staged collectives (vdn/dist.py); never a performance number")
    a = ap.parse_args(argv)
    if a.workload is None:
        a.workload = "stream" if a.gpus == 1 else "video"
    if a.steps is None:
        a.steps = 3 if a.workload == "video" else 10
    if a.warmup is None:
        a.warmup = 1 if a.workload == "video" else 7
    return a


def spawn_ranks(a) -> int:
    """Start `a.gpus` ranks of this script under torch.distributed.run. Runs before anything in this process has
    touched the GPU (importing torch does not), as a CHILD process — never an exec of a GPU-initialised one."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class _StubVideoModel:
    """Stand-in for VideoDepthAnything in `--stub` runs: a per-pixel temporal mix, so that the frame-sharded
    path really needs its all-to-all; CPU tensors only."""

    def preprocess_frames(self, frames, input_size):
        import torch
        return torch.from_numpy(frames.astype("float32") / 255.0).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def _mix(x):  # [T, HW, c] -> [T, HW, c], mixes frames per pixel
        import torch
        T = x.shape[0]
        w = torch.softmax(torch.arange(T * T, dtype=torch.float32).reshape(T, T).sin(), dim=-1)
        return torch.einsum("ft,tpc->fpc", w, x)

    def forward(self, x):  # [1,T,3,H,W] -> [1,T,H,W]
        _, T, c, H, W = x.shape
        y = self._mix(x[0].permute(0, 2, 3, 1).reshape(T, H * W, c))
        return y.mean(-1).abs().reshape(1, T, H, W) + 0.25

    def forward_sharded(self, x_local, group=None):
        from vdn.dist import FrameShardExchange, shard_core, world
        _, Tl, c, H, W = x_local.shape
        ex = FrameShardExchange(Tl * world(group), group)
        planes = [x_local[0].permute(0, 2, 3, 1).reshape(Tl * H * W, c).contiguous()]
        y = shard_core(ex, planes, H * W, lambda pl, D: [self._mix(pl[0].reshape(ex.T, D, c)).reshape(ex.T * D, c)])[0]
        return y.reshape(Tl, H * W, c).mean(-1).abs().reshape(1, Tl, H, W) + 0.25

    # staged interface (vdn/dist.py: encode distinct frames once, exchange taps, run heads): same numbers as forward()
    def encode_frames(self, x):
        k, c, H, W = x.shape
        return [x.permute(0, 2, 3, 1).reshape(k * H * W, c).contiguous()], H * W, (H, W)

    # chunked form of encode_frames (vdn/dist.py TapExchange: a chunk's taps travel while the next chunk is encoded)
    def tap_planes(self, x):
        import torch
        k, c, H, W = x.shape
        self._planes = [torch.empty((k * H * W, c), dtype=torch.float32)]
        return self._planes, H * W, (H, W)

    def encode_into(self, x, planes, first):
        k, c, H, W = x.shape
        planes[0][first * H * W:(first + k) * H * W].copy_(x.permute(0, 2, 3, 1).reshape(k * H * W, c))

    def head_from_planes(self, planes, Tl, T, hw, group=None):
        from vdn.dist import FrameShardExchange, shard_core
        H, W = hw
        c = planes[0].shape[-1]
        if group is None:
            y = self._mix(planes[0].reshape(T, H * W, c))
        else:
            ex = FrameShardExchange(T, group)
            y = shard_core(ex, [planes[0].contiguous()], H * W,
                           lambda pl, D: [self._mix(pl[0].reshape(ex.T, D, c)).reshape(ex.T * D, c)])[0].reshape(Tl, H * W, c)
        return y.mean(-1).abs().reshape(Tl, H, W) + 0.25

    def infer_video_depth(self, frames, fps, input_size=518):
        from vdn.dist import infer_video_depth_sharded
        return infer_video_depth_sharded(self, frames, fps, input_size=input_size)


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if a.shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU "
                 f"(python bench.py --gpus N starts them itself)")
    if a.lanes is not None:
        os.environ["VDN_STREAMS"] = str(a.lanes)
    lanes = int(os.environ.get("VDN_STREAMS", "2")) if a.workload == "stream" else 1
    if lanes < 2 or a.batch < int(os.environ.get("VDN_LANE_MIN_BATCH", "4")) or a.batch % lanes:
        lanes = 1
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # fail fast: a collective that does not complete within VDN_DIST_TIMEOUT_S (default 300 s) raises instead of hanging
        # the node; main() then exits non-zero and the launcher tears the other ranks down
        tmo = datetime.timedelta(seconds=int(os.environ.get("VDN_DIST_TIMEOUT_S", "300")))
        if a.stub or a.shared_gpu:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            torch.cuda.set_device(local)
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    n_gpus = world
    dev = torch.device("cpu") if a.stub else torch.device("cuda", local)
    if not a.stub:
        torch.cuda.set_device(dev)

    import numpy as np
    from vdn import synth, util

    enc = a.encoder
    H = W = 518
    if a.stub:
        assert a.workload == "video", "--stub rehearses the multi-rank video driver only"
        H = W = 28
        model, prec_name = _StubVideoModel(), "stub"
        frames_per_step = a.video_frames
    else:
        import vdn
        from vdn.depth_anything_v2 import _precision
        cfg = vdn.MODEL_CONFIGS[enc]
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
        from vdn.dist import infer_video_depth_sharded, plan_schedule, schedule_rounds
        video = np.ascontiguousarray(np.tile(fr, ((frames_per_step + 7) // 8, 1, 1, 1))[:frames_per_step])
        n_windows = len(util.window_table(frames_per_step))
    else:
        x = torch.from_numpy(synth.normalize_frames(fr)).to(dev)
        if a.workload == "refine5":
            xd = torch.from_numpy(synth.depth_clip(1234 + rank, 4, 1024, 1024)).to(dev).repeat(16, 1, 1)[None].contiguous()
        if x.shape[0] < frames_per_step:
            x = x.repeat((frames_per_step + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:frames_per_step]
        if a.workload == "clip":
            x = x[None]
        x = x.contiguous()

    def step():
        if a.workload == "video":  # the clip is sharded over the ranks (strong scaling); rank 0 holds the result
            if dist is not None:
                return infer_video_depth_sharded(model, video, 24, input_size=H, all_ranks=False)[0]
            return model.infer_video_depth(video, 24, input_size=H)[0]
        if a.workload == "vstream":
            return model.stream_step(x[:1][None])
        if a.workload == "refine5":
            return model.forward(xd)
        return model.forward(x)

    def sync_all():
        if dist is not None:
            dist.barrier()
        if not a.stub:
            torch.cuda.synchronize(dev)

    def runtimes():
        if a.stub:
            return []
        e = model._engines()
        return [e["rt"]] + [ln["rt"] for ln in (getattr(model, "_lanes", None) or [])[1:]]

    step_ms = []   # per-step HIP-event durations of the last timed() call

    def timed(nsteps, events, fn=None):
        fn = fn or step
        for rt in runtimes():
            rt.timing = [] if events else None
        sync_all()
        marks = []
        t0 = time.perf_counter()
        for _ in range(nsteps):
            if not a.stub:   # HIP events on the stream the step is issued on (lanes fork from and join into it)
                s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ev.record()
            fn()
            if not a.stub:
                e_ev.record()
                marks.append((s_ev, e_ev))
        sync_all()
        dt = time.perf_counter() - t0
        step_ms[:] = [s_.elapsed_time(e_) for s_, e_ in marks]
        ev = []
        for rt in runtimes():
            ev += rt.timing or []
            rt.timing = None
        if dist is not None:
            t = torch.tensor([dt], device=("cpu" if a.shared_gpu else dev), dtype=torch.float64)
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
    # issued on ONE lane right after it (same memory bank: the lanes share one ring, still full).
    want_events = (not a.no_kernel_events) and not a.stub
    dt, ev = timed(a.steps, want_events and lanes == 1)
    step_ms_main = list(step_ms)
    strong = a.workload == "video"
    fps = frames_per_step * a.steps * (1 if strong else n_gpus) / dt
    dt1 = None
    if lanes > 1 and want_events:
        os.environ["VDN_STREAMS"] = "1"
        step()
        dt1, ev = timed(a.steps, True)
        os.environ["VDN_STREAMS"] = str(lanes)

    if a.workload == "stream":
        workload = ("DepthAnythingV2(%s) batch=%d 518x518 streams/GPU, memory bank full (S=6); replicas per GPU" % (enc, a.batch))
    elif a.workload == "clip":
        workload = "VideoDepthAnything(%s) one 32-frame 518x518 window per step; one window per GPU" % enc
    elif a.workload == "refine5":
        workload = ("video_depth_model_v5.VideoDepthAnything(%s, num_frames=64).forward on a [1,64,1024,1024] raw depth clip (median "
                    "scale, 224x224 network, shift + residual); the metric counts refined 1024x1024 frames" % enc)
    elif a.workload == "vstream":
        workload = "VideoDepthAnything(%s).stream_step: one 518x518 frame per step against 31 cached frames (projected K/V cache)" % enc
    else:
        workload = ("VideoDepthAnything(%s).infer_video_depth on ONE %d-frame %dx%d u8 clip = %d windows of 32 (host frames in, host "
                    "depth out on rank 0: H2D, pre-processing, device stitcher and D2H inside the timed region); windows sharded "
                    "over the GPUs, last partial round frame-sharded (all-to-all over RCCL per temporal module)"
                    % (enc, frames_per_step, H, W, n_windows))
    out = {
        "metric": ("refined depth frames/sec at 1024x1024 (v5 refiner), " + enc) if a.workload == "refine5" else
                  ("depth frames/sec at 518x518, ViT-L" if enc == "vitl" else f"depth frames/sec at 518x518, {enc}"),
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
        # `value` / ms_per_step: wall clock over the K steps between two barrier + synchronize pairs (the contract; includes the
        # host's launch work). ms_per_step_event_median: HIP-event pair around every step on its stream, median
        "ms_per_step_event_median": round(statistics.median(step_ms_main), 3) if step_ms_main else None,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": prec_name, "data": "synthetic",
        "config": {"workload": workload, "frames_per_step_per_gpu": frames_per_step if not strong else frames_per_step / n_gpus,
                   "lanes": lanes, "precision": prec_name, "ranks": n_gpus,
                   "backend": ("gloo (stub rehearsal on CPU)" if a.stub else "gloo (ranks share one GPU: rehearsal)" if a.shared_gpu and dist is not None
                               else ("nccl (RCCL)" if dist is not None else "none")),
                   "precision_note": "f16x3 = fp16 hi/lo planes, 3 MFMA products per term (fp32-faithful, parity <=1e-3); attention: "
                                     "8-bit score cross terms, P V products per term = attention_pv_products (1: V as one fp16 plane "
                                     "rounded to nearest; measured 2e-5..9e-5 end to end against the 1e-3 tolerance)"},
    }
    if not a.stub:
        from vdn import _abi
        out["config"]["attention_pv_products"] = int(model._engines()["rt"].pv_products)
    if strong:
        jobs = plan_schedule(n_windows, n_gpus)
        out["config"]["schedule"] = {"windows": n_windows, "whole_window_jobs": sum(1 for j in jobs if j[2] == 1),
                                     "frame_sharded_jobs": [[j[0], j[1], j[2]] for j in jobs if j[2] > 1],
                                     "window_times_on_slowest_rank": schedule_rounds(n_windows, n_gpus)}

    # ---------------- roofline of the dominant kernel (encoder linear GEMMs): algorithmic flops / HIP-event time
    if ev:
        C = vdn.modules.ENCODERS[enc]["dim"]
        lin = [(s.elapsed_time(e), fl) for (tag, s, e, fl) in ev if tag == "enc_linear"]
        nprod = 3 if prec_name.endswith("x3") else 1
        enc_eng = model._engines()["enc"]
        rows = frames_per_step * ((H // 14) * (W // 14) + 1)   # the events come from a single-lane pass: the whole batch per launch
        x8 = bool(getattr(enc_eng, "x8", False)) and rows >= 4096 and prec_name == "f16x3"
        if lin:
            tot_ms, tot_fl = sum(t for t, _ in lin), sum(f for _, f in lin)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            # the PMC passes (tools/pmc_traffic.sh) ran the batch-8 stream workload: their per-launch bytes describe that launch size only
            traffic, traffic_src = _pmc_traffic(prec_name) if (a.workload == "stream" and a.batch == 8 and enc == "vitl") else (None, None)
            out["roofline"] = {
                "bound": "mfma",
                "kernel": ("gemm_x8_kernel<256|192 x 256 x 64> on the 4 encoder linears (qkv, proj, fc1, fc2): A_hi W_hi^T on fp16 MFMAs + the two "
                           "split-precision cross terms on the block-scaled MFMA from 6-bit (e3m2 + E8M0 per 32) K-tile-major planes" if x8 else
                           "gemm_x3_p8_kernel<256x256x32> (fc1) / gemm_x3_big_kernel<192x256x32> (qkv, proj, fc2): the 4 encoder linears" if nprod == 3
                           else "gemm_kernel<128x128x64> on the 4 encoder linears (qkv, proj, fc1, fc2)"),
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_F16, 4),
                "peak_measured": PEAK_MEASURED_TFLOPS_F16, "frac_of_measured": round(ach / PEAK_MEASURED_TFLOPS_F16, 4),
                "peak_measured_source": "profiles/r02_mfma_scale_probe.log (register-only fp16 32x32x16 loop on the box: what the power-managed clock sustains)",
                "traffic": traffic, "traffic_source": traffic_src, "launches_timed": len(lin),
                "avg_launch_ms": round(tot_ms / len(lin), 4), "algorithmic_gflop_per_launch": round(tot_fl / len(lin) / 1e9, 2)}
            # matrix-pipe work actually issued per algorithmic product: 3 fp16 products (x3), or 1 fp16 + 2 e3m2 products that the
            # scaled MFMA runs at 4.19x the fp16 rate (measured, same probe) = 1.48 fp16-product equivalents
            eq = (1.0 + 2.0 / 4.19) if x8 else float(nprod)
            out["roofline"].update({"mfma_products_per_term": ("1 fp16 + 2 e3m2 (block-scaled 6-bit, 4.19x rate)" if x8 else nprod),
                                    "fp16_product_equivalents": round(eq, 2), "executed_frac": round(eq * ach / PEAK_TFLOPS_F16, 4),
                                    "executed_frac_of_measured": round(eq * ach / PEAK_MEASURED_TFLOPS_F16, 4)})
            if dt1 is not None:
                out["roofline"]["measured_in"] = (
                    "single-lane pass of %d steps run right after the timed region (%.1f frames/s, same full memory bank); the "
                    "timed region deals the batch to %d HIP-stream lanes whose kernels co-run"
                    % (a.steps, frames_per_step * a.steps * n_gpus / dt1, lanes))
        att = [(s.elapsed_time(e), fl) for (tag, s, e, fl) in ev if tag == "enc_attn"]
        if att:
            tot_ms, tot_fl = sum(t for t, _ in att), sum(f for _, f in att)
            tf = tot_fl / (tot_ms * 1e-3) / 1e12
            out["attention_kernel"] = {"avg_launch_ms": round(tot_ms / len(att), 4), "achieved_tflops": round(tf, 2),
                                       "frac_of_mfma_peak": round(tf / PEAK_TFLOPS_F16, 4),
                                       "mfma_busy_frac": _attn_mfma_busy(),
                                       "note": "algorithmic flops (4 nq nk 64 per head) / HIP-event time; the kernel issues 1 fp16 + 2 "
                                               "block-scaled 8-bit products for QK^T and attention_pv_products for P V"}

    # ---------------- where an N-GPU step spends its time: one more DIAGNOSTIC step with the device synchronised at every phase
    # boundary (not the timed region), seconds per phase as MAX over ranks, bytes as SUM over ranks
    if strong and dist is not None:
        st = {}
        infer_video_depth_sharded(model, video, 24, input_size=H, all_ranks=False, stats=st)
        keys = ["preprocess", "encode", "tap_exchange_wait", "heads", "gather", "stitch", "to_host"]
        bkeys = ["bytes_taps_sent", "bytes_temporal_a2a_sent", "bytes_gather_sent"]
        cdev = "cpu" if (a.shared_gpu or a.stub) else dev
        tv = torch.tensor([st.get(k, 0.0) for k in keys], dtype=torch.float64, device=cdev)
        bv = torch.tensor([float(st.get(k, 0)) for k in bkeys], dtype=torch.float64, device=cdev)
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        dist.all_reduce(bv, op=dist.ReduceOp.SUM)
        out["phases"] = {"seconds_max_over_ranks": {k: round(float(v), 4) for k, v in zip(keys, tv.tolist())},
                         "bytes_sum_over_ranks": {k: int(v) for k, v in zip(bkeys, bv.tolist())},
                         "encoder_chunk_frames": int(os.environ.get("VDN_ENC_CHUNK", "8")),
                         "note": "one diagnostic step after the timed region, device synchronised at every phase boundary (the timed "
                                 "steps overlap the tap exchange with the encoder and do not synchronise); tap_exchange_wait = what "
                                 "was left of the exchange after the last encoder chunk"}

    # ---------------- the same workload on ONE GPU (rank 0 alone) next to the N-GPU number of the strong-scaling run
    if strong and dist is not None and not a.stub:
        if rank == 0:
            model.infer_video_depth(video, 24, input_size=H)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2):
                model.infer_video_depth(video, 24, input_size=H)
            torch.cuda.synchronize(dev)
            t1 = (time.perf_counter() - t0) / 2
            out["single_gpu_same_workload"] = {"value": round(frames_per_step / t1, 3), "unit": "frames/s", "steps": 2,
                                               "note": "rank 0 alone, infer_video_depth on the same clip, right after the timed region"}
        dist.barrier()

    # ---------------- PCIe-inclusive rate: host frames in (H2D) and host maps out (D2H) inside the step
    if a.workload == "stream" and not a.no_pcie and not a.stub:
        xh = x.cpu().pin_memory()
        oh = torch.empty((frames_per_step, H, W), dtype=torch.float32).pin_memory()

        def step_pcie():
            oh.copy_(model.forward(xh.to(dev, non_blocking=True)), non_blocking=True)

        step_pcie()
        dtp, _ = timed(a.steps, False, step_pcie)
        out["pcie_inclusive"] = {"value": round(frames_per_step * a.steps * n_gpus / dtp, 3), "unit": "frames/s",
                                 "ms_per_step": round(1e3 * dtp / a.steps, 3),
                                 "note": "f32 frames from pinned host memory (H2D) and f32 depth maps back (D2H) inside every timed step"}

    if a.single_pass_too and prec_name != "f16" and not a.stub:
        model.set_precision("f16")
        model = model.to(dev)
        for _ in range(a.warmup):
            step()
        dt2, _ = timed(a.steps, False)
        out["single_pass_f16"] = {"value": round(frames_per_step * a.steps * (1 if strong else n_gpus) / dt2, 3), "unit": "frames/s",
                                  "note": "1 MFMA product per term; 0.7e-3..2.5e-3 from the fp32 reference (tests/test_gpu_e2e.py)"}
        model.set_precision(prec_name)

    # ---------------- CPU baseline: the oracle on the host cores, rank 0, N == 1, steady state
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline and not a.stub:
        from oracle import ref_cpu as O
        sd_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        # the GPU box gives one GPU a 16-core CPU share: oversubscribing all visible cores is slower
        threads = min(torch.get_num_threads(), int(os.environ.get("VDN_CPU_THREADS", "16")))
        torch.set_num_threads(threads)
        runs = []
        with torch.no_grad():
            if a.workload == "stream":
                xc = x.reshape(-1, 3, H, W)[:1].cpu()
                mem = O.MemoryState(6)
                for _ in range(6):  # fill the bank to S = 6 (doubles as the warm-up)
                    O.depth_anything_v2_forward(sd_cpu, xc, mem, enc)
                for _ in range(3):
                    t0 = time.perf_counter()
                    O.depth_anything_v2_forward(sd_cpu, xc, mem, enc)
                    runs.append(time.perf_counter() - t0)
                n = 1
                sample = (f"DepthAnythingV2({enc}) batch 1, fp32, memory bank full (S=6): 6 fill frames (warm-up), then 3 timed "
                          f"frames, median")
            elif a.workload in ("clip", "video", "vstream"):
                n = 2
                xc = torch.from_numpy(synth.normalize_frames(fr[:n]))[None]
                for i in range(3):  # 1 warm-up + 2 runs
                    t0 = time.perf_counter()
                    O.video_depth_anything_forward(sd_cpu, xc, enc)
                    if i:
                        runs.append(time.perf_counter() - t0)
                sample = f"one {n}-frame VideoDepthAnything({enc}) clip, fp32: 1 warm-up + 2 runs, median"
        if runs:
            tc = statistics.median(runs)
            out["cpu_baseline"] = {"value": round(n / tc, 4), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": sample, "seconds_per_run": round(tc, 2), "runs": len(runs), "cpu": _cpu_model(),
                                   "torch": torch.__version__}
            out["speedup_vs_cpu_baseline"] = round(fps / (n / tc), 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:   # a rank that fails (collective timeout, HIP error) must take the job down, not leave peers waiting
        import traceback
        traceback.print_exc()
        print(f"bench.py: rank {os.environ.get('RANK', '0')} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        os._exit(1)
