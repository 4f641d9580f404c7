"""Multi-GPU sharding of hot path B. One process per GPU, torch.distributed with
backend "nccl" (= RCCL over xGMI); every function also runs on CPU tensors under "gloo", which is how
tests/test_dist.py covers it without GPUs.

Both levels are new design (the reference has no multi-GPU inference):

* windows — `plan_schedule`, `infer_video_depth_sharded`: the 32-frame windows of a clip are independent
  given the input frames (vdn.util.window_table). Full rounds deal one window to every rank; the windows
  left over for the last, partial round are each FRAME-SHARDED over a group of ranks, so no rank idles:
  12 windows on 8 GPUs = 8 whole windows + 4 windows on 2 GPUs each = 1.5 window-times instead of 2
  (ideal strong-scaling efficiency 1.0 instead of 0.75). The ENCODER does not follow the windows at all: it is
  per-frame, and 10 of every window's 32 slots repeat earlier input frames, so the clip's distinct frames are split
  evenly over the ranks, each is encoded once, and the ranks exchange the taps their heads read
  (`tap_exchange_plan`, `exchange_taps`: the all-gather of per-frame feature memory of the north star, sent only to
  the ranks that need each frame). The per-window depth maps are gathered to rank 0 once per clip for the (cheap,
  sequential) stitcher.

* frames inside a window — `FrameShardExchange`, `shard_core`: the encoder and every convolution are
  per-frame, only the 4 temporal modules mix frames, and they do so independently per pixel. Each rank keeps
  T/P frames; around each temporal module the activations are re-sharded frames<->pixels with an
  all-to-all (each rank then holds all T frames of 1/P of the pixels). An all-gather of the module input
  (the north star's first suggestion) would move P times the bytes and make every rank project K/V for all
  T frames; the all-to-all moves 1/P of that and lets the module's GEMMs scale with P as well. xGMI is
  point-to-point (7 links per GPU), and an all-to-all uses all of them at once.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import util


def world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


# --------------------------------------------------------------------------------------------- collectives
def _staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves host memory only. Device tensors under a gloo group (the shared-GPU rehearsal of the N > 1 path on
    a one-GPU box: tools/dist_rehearsal.py, `bench.py --shared-gpu`) are staged through the host as bytes; under
    nccl (= RCCL, the production backend) and for CPU tensors the collective gets the tensor itself."""
    return (t.is_cuda or _FORCE_STAGING) and dist.get_backend(group) == "gloo"


_FORCE_STAGING = False   # tests/test_dist.py: run the byte staging on CPU tensors too
# tests/test_gpu_dist.py: issue every collective even in a world of one rank, so that a one-GPU box runs the real RCCL calls
# (variable-split all_to_all_single with async handles, gather, broadcast) on a real communicator
_FORCE_COLLECTIVES = False


def _collective(P: int) -> bool:
    return P > 1 or (_FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def _row_bytes(t: torch.Tensor) -> int:
    return int(np.prod(t.shape[1:], dtype=np.int64)) * t.element_size()


def _bytes(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().reshape(t.shape[0], _row_bytes(t) // t.element_size()).view(torch.uint8).cpu()


def _unbytes(dst: torch.Tensor, host: torch.Tensor):
    dst.copy_(host.to(dst.device).view(dst.dtype).reshape(dst.shape))


COUNTERS = {"a2a_bytes_sent": 0, "a2a_calls": 0}   # payload this rank handed to all_to_all (diagnostics: bench.py's phase report)


def all_to_all(recv: torch.Tensor, send: torch.Tensor, out_split=None, in_split=None, group=None, async_op: bool = False):
    """dist.all_to_all_single over dim 0 (row counts in the split lists). async_op (RCCL only): the collective is enqueued
    on the communicator's own stream behind the work already on the current stream and a handle is returned; the caller's
    later kernels run beside it and `handle.wait()` orders the current stream behind the exchange. The host-staged gloo
    path is synchronous and returns None."""
    COUNTERS["a2a_bytes_sent"] += send.numel() * send.element_size()
    COUNTERS["a2a_calls"] += 1
    if not _staged(send, group):
        if async_op and dist.get_backend(group) != "gloo":
            return dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group, async_op=True)
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        return None
    hs = _bytes(send)
    hr = torch.empty((recv.shape[0], _row_bytes(recv)), dtype=torch.uint8)
    dist.all_to_all_single(hr, hs, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    _unbytes(recv, hr)
    return None


def gather_to(t: torch.Tensor, parts, dst: int = 0, group=None):
    if not _staged(t, group):
        dist.gather(t, parts, dst=dst, group=group)
        return
    h = _bytes(t)
    hp = [torch.empty_like(h) for _ in parts] if parts is not None else None
    dist.gather(h, hp, dst=dst, group=group)
    if parts is not None:
        for p_, h_ in zip(parts, hp):
            _unbytes(p_, h_)


def broadcast_from(t: torch.Tensor, src: int = 0, group=None):
    if not _staged(t, group):
        dist.broadcast(t, src=src, group=group)
        return
    h = _bytes(t)
    dist.broadcast(h, src=src, group=group)
    _unbytes(t, h)


# --------------------------------------------------------------------------------------------- schedule
def window_owner(n_windows: int, nranks: int) -> List[int]:
    """Plain round-robin (window w on rank w % nranks); kept for callers that want whole windows only."""
    return [w % nranks for w in range(n_windows)]


def plan_schedule(n_windows: int, nranks: int, frames_per_window: int = util.INFER_LEN) -> List[Tuple[int, int, int]]:
    """Jobs (window, first_rank, group_size) in execution order. Whole windows (group_size 1) fill the full
    rounds: rank r owns the CONTIGUOUS run of `n_windows // nranks` windows r*k .. r*k + k - 1 (neighbouring windows
    share 10 of their 32 frames and a rank's windows then read one stretch of the clip, so most of the encoder taps its
    heads need are the ones it encoded itself); the `n_windows % nranks` windows of the last round are each sharded by
    frame over a group of `group_size` consecutive ranks (a power of two that divides the frames of a window)."""
    jobs = []
    k = n_windows // nranks
    full = k * nranks
    for w in range(full):
        jobs.append((w, w // k, 1))
    rem = n_windows - full
    if rem:
        g = 1
        while g * 2 * rem <= nranks and frames_per_window % (g * 2) == 0:
            g *= 2
        for j in range(rem):
            jobs.append((full + j, j * g, g))
    return jobs


def schedule_rounds(n_windows: int, nranks: int) -> float:
    """Window-times the slowest rank spends (ideal, no exchange cost): DESIGN.md's expected efficiency is
    n_windows / (nranks * schedule_rounds)."""
    load = [0.0] * nranks
    for _, r0, g in plan_schedule(n_windows, nranks):
        for r in range(r0, r0 + g):
            load[r] += 1.0 / g
    return max(load)


_GROUPS: Dict[Tuple[int, int], object] = {}
_GROUPS_OF = [None]   # the default process group the cached subgroups belong to


def _subgroups(nranks: int, g: int):
    """Process groups of `g` consecutive ranks (aligned blocks). Collective: every rank creates every group,
    in the same order (torch.distributed.new_group contract). Cached per default process group: after
    destroy_process_group / init_process_group the cache is dropped (its groups died with the old world)."""
    if g == 1 or nranks == 1:
        return None
    cur = dist.distributed_c10d._get_default_group()
    if _GROUPS_OF[0] is not cur:
        _GROUPS.clear()
        _GROUPS_OF[0] = cur
    key = (nranks, g)
    if key not in _GROUPS:
        _GROUPS[key] = [dist.new_group(list(range(b, b + g))) for b in range(0, nranks - g + 1, g)]
    return _GROUPS[key]


# --------------------------------------------------------------------------------------------- driver
def tap_exchange_plan(table, jobs, nranks: int, T: int = util.INFER_LEN):
    """Who encodes which frame and who needs which frame's encoder taps (a pure function of the window table and
    the schedule, identical on every rank). Every distinct frame is encoded ONCE, at most ceil(frames / ranks) per rank
    (the encoder is per-frame, so this is perfectly balanced: 256 frames / 8 ranks = 32 each, not 48 = 1.5 windows), and
    preferably by a rank whose own head jobs read it: frames are dealt in ascending order to the least loaded rank that
    needs them, else to the least loaded rank at all. On 8 ranks 32 of the 45..48 frames a rank's heads read are then
    its own. Returns (frames, per, local[r], need[r], send[src][dst])."""
    frames = sorted({f for row in table for f in row})
    per = (len(frames) + nranks - 1) // nranks
    need = [[] for _ in range(nranks)]
    for (w, r0, g) in jobs:
        Tl = T // g
        for k in range(g):
            need[r0 + k] += table[w][k * Tl:(k + 1) * Tl]
    need = [sorted(set(n)) for n in need]
    wants = [set(n) for n in need]
    count = [0] * nranks
    owner = {}
    for f in frames:
        cands = [r for r in range(nranks) if f in wants[r] and count[r] < per] or [r for r in range(nranks) if count[r] < per]
        r = min(cands, key=lambda q: (count[q], q))
        owner[f] = r
        count[r] += 1
    local = [[f for f in frames if owner[f] == r] for r in range(nranks)]
    send = [[[f for f in need[dst] if owner[f] == src] for dst in range(nranks)] for src in range(nranks)]
    return frames, per, local, need, send


def exchange_taps(planes: List[torch.Tensor], rows_per_frame: int, local: List[int], send, r: int, group=None):
    """One all-to-all (variable splits) per plane: every rank hands each peer the tap rows of the frames that peer's
    head jobs read (`send[r][dst]`) — the RCCL exchange of per-frame feature memory over xGMI. Returns
    (received planes, {frame: first row}). planes: [len(local) * rows_per_frame, C] tensors of this rank's frames."""
    ex = TapExchange(planes, rows_per_frame, local, send, r, chunk=max(1, len(local)), group=group)
    ex.send_chunk(0)
    return ex.finish()


class TapExchange:
    """The encoder-tap exchange in CHUNKS of `chunk` local frames, so that it runs behind the encoder: the driver encodes
    chunk c, calls `send_chunk(c)` — one variable-split all-to-all per plane, asynchronous under RCCL: it waits for the
    encoder kernels already on the stream and runs beside the next chunk's — and `finish()` waits for all of them. Every
    rank issues the same number of chunks (`nchunks`: from the largest local share; a rank without frames in a chunk
    sends nothing but still takes part). Send buffers are built with ONE index_select per plane and chunk; the rows
    received from every (chunk, source) land in one buffer per plane, `where[frame]` = first row."""

    def __init__(self, planes, rows_per_frame: int, local: List[int], send, r: int, chunk: int, group=None):
        self.planes, self.n, self.local, self.r, self.group = planes, rows_per_frame, local, r, group
        P = self.P = len(send)
        self.chunk = chunk
        per = max(len(f) for f in _owned(send, P))
        self.nchunks = max(1, (per + chunk - 1) // chunk)
        owned = _owned(send, P)                       # frames each rank encodes, in its local (ascending) order
        cpos = [{f: i // chunk for i, f in enumerate(owned[q])} for q in range(P)]
        # rows this rank receives: chunk-major, then source rank, then the order of send[src][r]
        self.where, self.out_split, self.in_split, self.idx = {}, [], [], []
        row = 0
        lpos = {f: i for i, f in enumerate(local)}
        dev = planes[0].device
        self.recv_off = []
        for c in range(self.nchunks):
            self.recv_off.append(row)
            osp = []
            for src in range(P):
                fr = [f for f in send[src][r] if cpos[src][f] == c]
                for f in fr:
                    self.where[f] = row
                    row += rows_per_frame
                osp.append(len(fr) * rows_per_frame)
            self.out_split.append(osp)
            mine = [[f for f in send[r][dst] if cpos[r][f] == c] for dst in range(P)]
            self.in_split.append([len(m) * rows_per_frame for m in mine])
            rows = [lpos[f] * rows_per_frame + k for m in mine for f in m for k in range(rows_per_frame)]
            self.idx.append(torch.tensor(rows, dtype=torch.long, device=dev))
        self.recv = [pl.new_empty((row, pl.shape[-1])) for pl in planes]
        self.handles = []
        self.bytes_sent = 0

    def send_chunk(self, c: int):
        for pl, rb in zip(self.planes, self.recv):
            sbuf = pl.index_select(0, self.idx[c])
            end = self.recv_off[c] + sum(self.out_split[c])
            dst = rb[self.recv_off[c]:end]
            self.bytes_sent += (sum(self.in_split[c]) - self.in_split[c][self.r]) * _row_bytes(pl)
            if _collective(self.P):
                h = all_to_all(dst, sbuf, self.out_split[c], self.in_split[c], self.group, async_op=True)
                if h is not None:
                    self.handles.append((h, sbuf))   # keep the send buffer alive until the collective has run
            else:
                dst.copy_(sbuf)

    def finish(self):
        for h, _ in self.handles:
            h.wait()
        self.handles = []
        return self.recv, self.where


def _owned(send, P: int) -> List[List[int]]:
    """Frames each rank encodes (ascending): the union of what it sends to anyone — every encoded frame is needed by at
    least one head job — as tap_exchange_plan's `local`."""
    return [sorted({f for dst in range(P) for f in send[src][dst]}) for src in range(P)]


def _slot_rows(planes: List[torch.Tensor], where, slots: Sequence[int], rows_per_frame: int) -> List[torch.Tensor]:
    """Rows of `slots` (frame ids, repeats allowed) as [len(slots) * rows_per_frame, C] tensors: in place when the
    slots are one ascending run of the received rows, else copied run by run."""
    starts = [where[f] for f in slots]
    n = rows_per_frame
    if all(starts[i + 1] == starts[i] + n for i in range(len(starts) - 1)):
        return [pl[starts[0]:starts[0] + len(slots) * n] for pl in planes]
    out = [pl.new_empty((len(slots) * n, pl.shape[-1])) for pl in planes]
    i = 0
    while i < len(slots):
        k = i
        while k + 1 < len(slots) and starts[k + 1] == starts[k] + n:
            k += 1
        for src, dst in zip(planes, out):
            dst[i * n:(k + 1) * n].copy_(src[starts[i]:starts[k] + n])
        i = k + 1
    return out


def infer_video_depth_sharded(model, frames: np.ndarray, target_fps, input_size: int = 518, group=None,
                              forward: Optional[Callable] = None, forward_sharded: Optional[Callable] = None,
                              all_ranks: bool = True, stats: Optional[dict] = None):
    """Multi-GPU twin of VideoDepthAnything.infer_video_depth (video_depth.py:67-156) over the default process
    group. Returns (f32 [N,h,w], target_fps) on rank 0 — and on every rank when `all_ranks` (one broadcast of
    the stitched clip) — else (None, target_fps).

    A model with `encode_frames` / `head_from_planes` (the product, and bench.py's stand-in) runs in three phases:
    every rank ENCODES its contiguous share of the clip's distinct frames (no frame is encoded twice, on any rank),
    the ranks EXCHANGE the encoder taps their head jobs read (`exchange_taps`: one variable-split all-to-all per
    plane), then the HEADS run per the schedule: whole windows, and the last partial round's windows frame-sharded
    over groups of ranks. Otherwise `forward(window [1,32,3,H,W]) -> [1,32,H,W]` / `forward_sharded(local frames
    [1,32/g,3,H,W], group) -> [1,32/g,H,W]` are called per job (tests inject stubs).

    The encoder runs in chunks of VDN_ENC_CHUNK (8) frames and the taps of chunk c travel while chunk c + 1 is encoded
    (`TapExchange`). `stats` (a dict, optional): filled with this rank's seconds per phase — preprocess, encode,
    tap_exchange_wait, heads, gather, stitch — and bytes it sent (taps, temporal all-to-alls, gather); the device is then
    synchronised at every phase boundary, so pass it for a DIAGNOSTIC run, not for the timed one."""
    import os
    import time

    def mark(name, t0):
        if stats is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            stats[name] = stats.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    assert group is None, "the schedule builds its own subgroups of the default group"
    P, r = world(), rank()
    fh, fw = frames[0].shape[:2]
    ratio = max(fh, fw) / min(fh, fw)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14
    n = frames.shape[0]
    table = util.window_table(n)
    T = util.INFER_LEN
    jobs = plan_schedule(len(table), P, T)
    fwd = forward if forward is not None else model.forward
    fwd_sh = forward_sharded if forward_sharded is not None else getattr(model, "forward_sharded", None)
    prep = getattr(model, "preprocess_frames", None)
    resize = getattr(model, "resize_depth", None)
    if hasattr(model, "_engines"):
        dev = model._engines()["rt"].device
    else:
        dev = torch.device("cpu")
    groups = {g: _subgroups(P, g) for g in sorted({j[2] for j in jobs})}  # collective: before any rank-local branch

    def net_input(idxs: Sequence[int]) -> torch.Tensor:
        need = sorted(set(idxs))
        pos = {f: i for i, f in enumerate(need)}
        net = prep(frames[need], input_size) if prep else torch.from_numpy(frames[need]).float()
        return net[[pos[f] for f in idxs]]

    pieces: List[torch.Tensor] = []   # this rank's depth frames in job order
    staged = forward is None and forward_sharded is None and hasattr(model, "encode_frames")
    a2a0 = COUNTERS["a2a_bytes_sent"]
    t0 = time.perf_counter()
    if staged:
        _, _, local, _, send = tap_exchange_plan(table, jobs, P, T)
        mine = local[r]
        x_mine = net_input(mine) if mine else net_input(table[0][:1])[:0]  # a rank may own no frame (tiny clips)
        t0 = mark("preprocess", t0)
        chunk = max(1, int(os.environ.get("VDN_ENC_CHUNK", "8")))
        if hasattr(model, "tap_planes"):       # the product: encode chunk by chunk, each chunk's taps leave while the next is encoded
            planes, rpf, hw = model.tap_planes(x_mine)
            ex = TapExchange(planes, rpf, mine, send, r, chunk)
            for c in range(ex.nchunks):
                c0, c1 = c * chunk, min(len(mine), (c + 1) * chunk)
                if c1 > c0:
                    model.encode_into(x_mine[c0:c1], planes, c0)
                ex.send_chunk(c)
        else:                                  # stand-in models: one pass, one exchange
            planes, rpf, hw = model.encode_frames(x_mine)   # rows per frame, (H, W) of the network input
            ex = TapExchange(planes, rpf, mine, send, r, max(1, max(len(l) for l in local)))
            ex.send_chunk(0)
        t0 = mark("encode", t0)
        got, where = ex.finish()
        t0 = mark("tap_exchange_wait", t0)
        if stats is not None:
            stats["bytes_taps_sent"] = ex.bytes_sent
        for (w, r0, g) in jobs:
            if not (r0 <= r < r0 + g):
                continue
            Tl, k = T // g, r - r0
            slots = table[w][k * Tl:(k + 1) * Tl]
            d = model.head_from_planes(_slot_rows(got, where, slots, rpf), Tl, T, hw, None if g == 1 else groups[g][r0 // g])
            pieces.append((resize(d, fh, fw) if resize else d).float().clone())
    else:
        for (w, r0, g) in jobs:
            if not (r0 <= r < r0 + g):
                continue
            if g == 1:
                d = fwd(net_input(table[w])[None])[0]
            else:
                Tl = T // g
                k = r - r0
                d = fwd_sh(net_input(table[w][k * Tl:(k + 1) * Tl])[None], groups[g][r0 // g])[0]
            pieces.append((resize(d, fh, fw) if resize else d).float())
    t0 = mark("heads", t0)

    # ---- gather every rank's frames to rank 0 (equal slabs; the schedule tells who holds what)
    counts = [0] * P
    for (_, r0, g) in jobs:
        for q in range(r0, r0 + g):
            counts[q] += T // g
    slab = torch.zeros((max(counts), fh, fw), dtype=torch.float32, device=dev)
    if pieces:
        cat = torch.cat(pieces)
        slab[: cat.shape[0]].copy_(cat)
    if _collective(P):
        parts = [torch.empty_like(slab) for _ in range(P)] if r == 0 else None
        gather_to(slab, parts, dst=0)
    else:
        parts = [slab]
    t0 = mark("gather", t0)
    if stats is not None:
        stats["bytes_gather_sent"] = 0 if r == 0 else slab.numel() * 4
        stats["bytes_temporal_a2a_sent"] = COUNTERS["a2a_bytes_sent"] - a2a0 - (stats.get("bytes_taps_sent", 0) if staged else 0)
    out = None
    if r == 0:
        allw = torch.empty((len(table), T, fh, fw), dtype=torch.float32, device=dev)
        cursor = [0] * P
        for (w, r0, g) in jobs:
            Tl = T // g
            for k in range(g):
                q = r0 + k
                allw[w, k * Tl:(k + 1) * Tl].copy_(parts[q][cursor[q]:cursor[q] + Tl])
                cursor[q] += Tl
        if allw.is_cuda and hasattr(model, "_engines"):  # stitch on the device, one D2H per clip
            from .video_depth import DeviceStitcher
            st = DeviceStitcher(model._engines()["rt"], len(table), fh, fw)
            for w in range(len(table)):
                st.push(allw[w])
            out = st.result(n)
        else:
            dn = allw.cpu().numpy()
            out = torch.from_numpy(util.stitch([dn[w, i] for w in range(len(table)) for i in range(T)], n)).to(dev)
    t0 = mark("stitch", t0)
    if all_ranks and _collective(P):
        if r != 0:
            out = torch.empty((n, fh, fw), dtype=torch.float32, device=dev)
        out = out.contiguous()
        broadcast_from(out, src=0)
    res = None if out is None else util.to_host(out.contiguous())
    mark("to_host", t0)
    return res, target_fps


# --------------------------------------------------------------------------------------------- frames
class FrameShardExchange:
    """Re-shard [frames, pixels, channels] activations between 'my frames, all pixels' and
    'all frames, my pixels' with one all-to-all each way. Pixels are padded to a multiple of the
    world size (37*37 = 1369 is not divisible by 8); pad rows are zeros and are dropped on the way back.
    Each direction costs ONE staging copy (the pad + rank-major permute of the send buffer, or its inverse on
    the receive side); the other side of each all-to-all is used in place."""

    def __init__(self, T: int, group=None):
        self.group = group
        self.P, self.r = world(group), rank(group)
        assert T % self.P == 0, f"frames per window ({T}) must divide by the number of ranks ({self.P})"
        self.T, self.Tl = T, T // self.P

    def pix_per_rank(self, HW: int) -> int:
        return (HW + self.P - 1) // self.P

    def frames_to_pixels(self, x: torch.Tensor) -> torch.Tensor:
        """x [Tl, HW, c] (this rank's frames) -> [T, HWp, c] (all frames, this rank's pixel shard)."""
        Tl, HW, c = x.shape
        assert Tl == self.Tl
        P, HWp = self.P, self.pix_per_rank(HW)
        if P == 1:
            return x
        send = x.new_zeros((P, Tl, HWp, c)) if P * HWp != HW else x.new_empty((P, Tl, HWp, c))
        full = HW // HWp                       # shards completely covered by real pixels
        if full:
            send[:full].copy_(x[:, :full * HWp].unflatten(1, (full, HWp)).permute(1, 0, 2, 3))
        if full < P and HW > full * HWp:
            send[full, :, :HW - full * HWp].copy_(x[:, full * HWp:])
        recv = torch.empty_like(send)
        all_to_all(recv, send, group=self.group)
        return recv.reshape(P * Tl, HWp, c)  # rank-major == frame order (rank q owns frames q*Tl..)

    def pixels_to_frames(self, y: torch.Tensor, HW: int) -> torch.Tensor:
        """y [T, HWp, c] -> [Tl, HW, c]."""
        P = self.P
        if P == 1:
            return y
        T, HWp, c = y.shape
        send = y.reshape(P, self.Tl, HWp, c)
        if not send.is_contiguous():
            send = send.contiguous()
        recv = torch.empty_like(send)
        all_to_all(recv, send, group=self.group)
        # recv[q] = my frames' pixel shard q
        out = y.new_empty((self.Tl, HW, c))
        full = HW // HWp
        if full:
            out[:, :full * HWp].unflatten(1, (full, HWp)).copy_(recv[:full].permute(1, 0, 2, 3))
        if full < P and HW > full * HWp:
            out[:, full * HWp:].copy_(recv[full, :, :HW - full * HWp])
        return out

    def bytes_per_module(self, HW: int, c: int, planes: int = 2, elem: int = 2) -> int:
        """payload one rank sends per direction for one temporal module"""
        return self.Tl * self.pix_per_rank(HW) * (self.P - 1) * c * planes * elem


def shard_core(exch: FrameShardExchange, planes: List[torch.Tensor], HW: int, core: Callable) -> List[torch.Tensor]:
    """The staging around one temporal module of a frame-sharded window, as a pure function of tensors:
    every plane [Tl*HW, c] of this rank's frames (hi and lo planes travel as separate all-to-alls, no
    concatenation) -> all frames of this rank's pixel shard -> `core(list of [T*D, c] planes, D)` (the per-pixel
    temporal block, any callable) -> back to this rank's frames [Tl*HW, c']. Used by TemporalEngine.run_sharded
    with the HIP core and by tests/test_dist.py with a CPU stand-in under gloo."""
    Tl = exch.Tl
    px = [exch.frames_to_pixels(p.reshape(Tl, HW, p.shape[-1])) for p in planes]
    D = px[0].shape[1]
    out = core([p.reshape(exch.T * D, p.shape[-1]) for p in px], D)
    back = [exch.pixels_to_frames(o.reshape(exch.T, D, o.shape[-1]), HW) for o in out]
    return [b.reshape(Tl * HW, b.shape[-1]) for b in back]
