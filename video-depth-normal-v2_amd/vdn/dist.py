"""Multi-GPU sharding of hot path B. One process per GPU, torch.distributed with
backend "nccl" (= RCCL over xGMI); every function also runs on CPU tensors under "gloo", which is how
tests/test_dist.py covers it without GPUs.

Two levels, both new design (the reference has no multi-GPU inference):

* windows — `window_owner`, `infer_video_depth_sharded`: the 32-frame windows of a clip are
  independent given the input frames (vdn.util.window_table), so they are dealt round-robin to the
  ranks; the only communication is one all-gather of the per-window depth maps for the (cheap,
  sequential) host stitcher. No data-path collective.

* frames inside a window — `FrameShardExchange`: the encoder and every convolution are per-frame,
  only the 4 temporal modules mix frames, and they do so independently per pixel. Each rank keeps
  T/P frames; around each temporal module the activations are re-sharded frames<->pixels with an
  all-to-all (each rank then holds all T frames of 1/P of the pixels), which moves 1/P of what an
  all-gather of the module input would and lets the module's GEMMs scale with P as well.
  xGMI is point-to-point (7 links per GPU), and an all-to-all uses all of them at once.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import util


def world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


# --------------------------------------------------------------------------------------------- windows
def window_owner(n_windows: int, nranks: int) -> List[int]:
    """Round-robin: window w runs on rank w % nranks (12 windows of a 256-frame clip on 8 GPUs: 2 rounds)."""
    return [w % nranks for w in range(n_windows)]


def gather_windows(local: torch.Tensor, n_windows: int, group=None) -> Optional[torch.Tensor]:
    """local: this rank's windows stacked in ascending window order [n_local, 32, h, w] (may be empty).
    Returns [n_windows, 32, h, w] in window order on every rank (all-gather of equal-sized slabs)."""
    P, r = world(group), rank(group)
    if P == 1:
        return local
    per = (n_windows + P - 1) // P
    slab = local.new_zeros((per,) + tuple(local.shape[1:]))
    slab[: local.shape[0]] = local
    parts = [torch.empty_like(slab) for _ in range(P)]
    dist.all_gather(parts, slab, group=group)
    out = []
    for w in range(n_windows):
        out.append(parts[w % P][w // P])
    return torch.stack(out)


def infer_video_depth_sharded(model, frames: np.ndarray, target_fps, input_size: int = 518, group=None,
                              forward=None):
    """Window-parallel twin of VideoDepthAnything.infer_video_depth: identical output on every rank.
    `forward(window_input [1,32,3,H,W]) -> [1,32,H,W]` defaults to model.forward (tests inject a stub)."""
    P, r = world(group), rank(group)
    fh, fw = frames[0].shape[:2]
    ratio = max(fh, fw) / min(fh, fw)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14
    n = frames.shape[0]
    table = util.window_table(n)
    owner = window_owner(len(table), P)
    fwd = forward if forward is not None else model.forward
    prep = model.preprocess_frames if hasattr(model, "preprocess_frames") else None
    mine = []
    for w, idxs in enumerate(table):
        if owner[w] != r:
            continue
        need = sorted(set(idxs))
        pos = {f: i for i, f in enumerate(need)}
        net = prep(frames[need], input_size) if prep else torch.from_numpy(frames[need]).float()
        cur = net[[pos[f] for f in idxs]][None]
        d = fwd(cur)[0]
        mine.append(model.resize_depth(d, fh, fw) if hasattr(model, "resize_depth") else d)
    dev = mine[0].device if mine else (net.device if prep else torch.device("cpu"))
    local = torch.stack(mine) if mine else torch.zeros((0, util.INFER_LEN, fh, fw), device=dev)
    allw = gather_windows(local.float().contiguous(), len(table), group)
    if allw.is_cuda and hasattr(model, "_engines"):  # stitch on the device, one D2H per clip
        from .video_depth import DeviceStitcher
        st = DeviceStitcher(model._engines()["rt"], len(table), fh, fw)
        for w in range(len(table)):
            st.push(allw[w])
        return st.result(n).cpu().numpy(), target_fps
    dn = allw.cpu().numpy()
    depth_list = [dn[w, i] for w in range(len(table)) for i in range(util.INFER_LEN)]
    return util.stitch(depth_list, n), target_fps


# --------------------------------------------------------------------------------------------- frames
class FrameShardExchange:
    """Re-shard [frames, pixels, channels] activations between 'my frames, all pixels' and
    'all frames, my pixels' with one all-to-all each way. Pixels are padded to a multiple of the
    world size (37*37 = 1369 is not divisible by 8); pad rows are zeros and are dropped on the way back."""

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
        send = x.new_zeros((P, Tl, HWp, c))
        xp = x.new_zeros((Tl, P * HWp, c))
        xp[:, :HW] = x
        send.copy_(xp.reshape(Tl, P, HWp, c).permute(1, 0, 2, 3))
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return recv.reshape(P * Tl, HWp, c)  # rank-major == frame order (rank q owns frames q*Tl..)

    def pixels_to_frames(self, y: torch.Tensor, HW: int) -> torch.Tensor:
        """y [T, HWp, c] -> [Tl, HW, c]."""
        P = self.P
        if P == 1:
            return y
        T, HWp, c = y.shape
        send = y.reshape(P, self.Tl, HWp, c).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        # recv[q] = my frames' pixel shard q
        return recv.permute(1, 0, 2, 3).reshape(self.Tl, P * HWp, c)[:, :HW].contiguous()

    def bytes_per_module(self, HW: int, c: int, planes: int = 2, elem: int = 2) -> int:
        """payload one rank sends per direction for one temporal module"""
        return self.Tl * self.pix_per_rank(HW) * (self.P - 1) * c * planes * elem
