"""N>1 paths on CPU: gloo process groups of 2 and 3 ranks. The GPU kernels cannot
run here, so the compute is a per-pixel stand-in; everything else — schedule, subgroups, the frame<->pixel staging
of vdn.dist.shard_core that TemporalEngine.run_sharded uses, the gather to rank 0, bench.py's rank spawning — is
the product code."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _per_pixel_temporal_op(x):
    """stand-in for the temporal core: mixes frames, independent per pixel and channel-mixing"""
    T = x.shape[0]
    w = torch.softmax(torch.arange(T * T, dtype=torch.float32).reshape(T, T).sin(), dim=-1)
    y = torch.einsum("ft,tpc->fpc", w, x)
    return y + 0.1 * y.roll(1, dims=-1)


def _worker_exchange(rank, world, port, out):
    _init(rank, world, port)
    from vdn.dist import FrameShardExchange, shard_core
    T, HW, c = 8, 37, 6          # 37 pixels: not divisible by the world size -> padded shards
    full = torch.arange(T * HW * c, dtype=torch.float32).reshape(T, HW, c).cos()
    ex = FrameShardExchange(T)
    Tl = T // world
    mine = full[rank * Tl:(rank + 1) * Tl].contiguous()
    px = ex.frames_to_pixels(mine)
    assert px.shape == (T, ex.pix_per_rank(HW), c)
    lo = rank * ex.pix_per_rank(HW)
    hi = min(lo + ex.pix_per_rank(HW), HW)
    assert torch.equal(px[:, : hi - lo], full[:, lo:hi])          # all frames of my pixel shard
    assert (px[:, hi - lo:] == 0).all()                            # pad rows are zeros
    rt = torch.equal(ex.pixels_to_frames(px, HW), mine)             # round trip is the identity
    # the staging TemporalEngine.run_sharded uses, with two planes (hi, lo) and a stand-in core
    planes = [mine.reshape(Tl * HW, c), (0.5 * mine).reshape(Tl * HW, c)]

    def core(pl, D):
        return [_per_pixel_temporal_op(p.reshape(T, D, c)).reshape(T * D, c) for p in pl]

    back = shard_core(ex, planes, HW, core)
    ref = _per_pixel_temporal_op(full)[rank * Tl:(rank + 1) * Tl].reshape(Tl * HW, c)
    ok = torch.allclose(back[0], ref, atol=1e-6) and torch.allclose(back[1], 0.5 * ref, atol=1e-6)
    out[rank] = bool(ok and rt)
    dist.destroy_process_group()


def _reference_clip(frames):
    """single-process run of the same driver on the same stand-in network"""
    import bench
    from vdn import util
    m = bench._StubVideoModel()
    net = m.preprocess_frames(frames, 28)
    dl = []
    for idxs in util.window_table(len(frames)):
        dn = m.forward(net[idxs][None])[0].numpy()
        dl += [dn[i] for i in range(32)]
    return util.stitch(dl, len(frames))


def _worker_hybrid(rank, world, port, out, n, staged=False):
    """world 2, 50 frames = 3 windows: one full round of whole windows + the third window frame-sharded over both
    ranks. world 3, 20 frames = 1 window: sharded over ranks 0-1, rank 2 owns nothing (the idle-rank path)."""
    _init(rank, world, port)
    import bench
    from vdn import synth
    import vdn.dist
    from vdn.dist import infer_video_depth_sharded, plan_schedule
    vdn.dist._FORCE_STAGING = staged   # the byte staging a gloo group applies to device tensors (shared-GPU rehearsal)
    frames = synth.frames_u8(7, n, 28, 42)
    m = bench._StubVideoModel()
    os.environ["VDN_ENC_CHUNK"] = "3"   # several encoder chunks per rank, the last one ragged: every chunk's taps go out on their own
    st = {}
    d, fps = infer_video_depth_sharded(m, frames, 30, input_size=28, all_ranks=True, stats=st)
    assert all(k in st for k in ("preprocess", "encode", "tap_exchange_wait", "heads", "gather", "stitch", "bytes_taps_sent",
                                 "bytes_temporal_a2a_sent", "bytes_gather_sent")), st
    os.environ["VDN_ENC_CHUNK"] = "64"  # one chunk
    d0, _ = infer_video_depth_sharded(m, frames, 30, input_size=28, all_ranks=False)
    # the same clip through the per-job path (whole forward / forward_sharded per window, no tap exchange)
    d1, _ = infer_video_depth_sharded(m, frames, 30, input_size=28, forward=m.forward, forward_sharded=m.forward_sharded)
    ref = _reference_clip(frames)
    ok = d.shape == (n, 28, 42) and np.allclose(d, ref, rtol=1e-5, atol=1e-6) and fps == 30
    ok = ok and ((d0 is None) if rank else np.array_equal(d0, d)) and np.allclose(d1, d, rtol=1e-6, atol=1e-7)
    jobs = plan_schedule(len(__import__("vdn.util", fromlist=["x"]).window_table(n)), world)
    out[rank] = (bool(ok), jobs)
    dist.destroy_process_group()


@pytest.mark.parametrize("port", [29611])
def test_two_rank_exchange_and_plane_staging(port):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_exchange, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


@pytest.mark.parametrize("world,n,port,expect,staged", [
    (2, 50, 29612, [(0, 0, 1), (1, 1, 1), (2, 0, 2)], False),
    (3, 20, 29613, [(0, 0, 2)], False),
    (2, 50, 29614, [(0, 0, 1), (1, 1, 1), (2, 0, 2)], True),
    (3, 20, 29615, [(0, 0, 2)], True),
    # BASELINE configs[3] itself: the 256-frame clip on 8 ranks = 8 whole windows + 4 windows frame-sharded over rank pairs
    (8, 256, 29616, [(w, w, 1) for w in range(8)] + [(8, 0, 2), (9, 2, 2), (10, 4, 2), (11, 6, 2)], False),
])
def test_hybrid_schedule_driver_gloo(world, n, port, expect, staged):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_hybrid, args=(world, port, out, n, staged), nprocs=world, join=True)
    for r in range(world):
        assert out[r][0] is True, r
        assert [tuple(j) for j in out[r][1]] == expect


def test_schedule_and_payload():
    sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
    from vdn.dist import FrameShardExchange, plan_schedule, schedule_rounds, window_owner
    assert window_owner(12, 8) == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]
    # BASELINE configs[3]: 12 windows on 8 GPUs = 8 whole windows, then 4 windows on 2 GPUs each: 1.5 window-times
    jobs = plan_schedule(12, 8)
    assert jobs[:8] == [(w, w, 1) for w in range(8)] and jobs[8:] == [(8, 0, 2), (9, 2, 2), (10, 4, 2), (11, 6, 2)]
    assert schedule_rounds(12, 8) == 1.5 and 12 / (8 * schedule_rounds(12, 8)) == 1.0
    assert schedule_rounds(12, 4) == 3.0 and schedule_rounds(12, 2) == 6.0 and schedule_rounds(12, 1) == 12.0
    assert plan_schedule(1, 8) == [(0, 0, 8)]                        # one window: all 8 ranks share its frames
    assert [j[2] for j in plan_schedule(5, 8)] == [1] * 5            # 5 left-over windows cannot pair up on 8 ranks
    assert plan_schedule(3, 2) == [(0, 0, 1), (1, 1, 1), (2, 0, 2)]
    for nw in range(1, 30):                                          # every window exactly once, groups inside the world
        for P in (1, 2, 3, 4, 8):
            jobs = plan_schedule(nw, P)
            assert sorted(j[0] for j in jobs) == list(range(nw))
            assert all(0 <= j[1] and j[1] + j[2] <= P and 32 % j[2] == 0 for j in jobs)
    # BASELINE configs[3]: the 256 distinct frames of the clip are encoded once, 32 per rank (not 48 = 1.5 windows);
    # every rank's heads read 48 frames' taps, of which at most 32 are its own
    from vdn import util
    from vdn.dist import tap_exchange_plan
    table = util.window_table(256)
    frames, per, local, need, send = tap_exchange_plan(table, plan_schedule(len(table), 8), 8)
    assert len(frames) == 256 and per == 32 and all(len(l) == 32 for l in local)
    assert sorted(len(x) for x in need) == [47, 47, 47, 47, 48, 48, 48, 48] or max(len(x) for x in need) <= 48
    for dst in range(8):
        got = sorted(f for src in range(8) for f in send[src][dst])
        assert got == need[dst] and all(f in local[src] for src in range(8) for f in send[src][dst])
    # frames are encoded where they are read when possible: a rank's heads read 45..48 frames, 32 of them its own
    recv = [sum(len(send[src][dst]) for src in range(8) if src != dst) for dst in range(8)]
    assert max(recv) <= 20 and all(recv[d] == len(need[d]) - len(set(need[d]) & set(local[d])) for d in range(8))
    for P, bound in ((2, 16), (4, 16)):                              # whole windows only: contiguous runs of windows per rank
        t2 = util.window_table(256)
        j2 = plan_schedule(len(t2), P)
        assert [j[1] for j in j2] == sorted(j[1] for j in j2)        # rank r owns windows r*k .. r*k+k-1
        _, per2, loc2, need2, send2 = tap_exchange_plan(t2, j2, P)
        assert all(len(l) == per2 for l in loc2)
        assert max(sum(len(send2[s][d]) for s in range(P) if s != d) for d in range(P)) <= bound   # was 92 / 77 with round-robin windows
    ex = FrameShardExchange(32)      # world size 1: identity exchange
    x = torch.randn(32, 10, 4)
    assert ex.frames_to_pixels(x) is x and ex.pixels_to_frames(x, 10) is x


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (gloo + stand-in network via --stub)
    and reports them; a --gpus / WORLD_SIZE mismatch is an error, not a silent single-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "1", "--warmup", "0",
           "--video-frames", "50"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and j["scaling"] == "strong"
    assert j["config"]["schedule"]["frame_sharded_jobs"] == [[2, 0, 2]] and j["value"] > 0
    ph = j["phases"]   # the first 8-GPU run must be attributable: per-phase seconds (max over ranks) and bytes exchanged
    assert set(ph["seconds_max_over_ranks"]) >= {"encode", "tap_exchange_wait", "heads", "gather", "stitch"}
    assert ph["bytes_sum_over_ranks"]["bytes_taps_sent"] > 0 and ph["bytes_sum_over_ranks"]["bytes_gather_sent"] > 0
    env["WORLD_SIZE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
