"""Host-side pieces of the drivers (numpy): resize sizing, window table, scale/shift alignment,
overlap blending. Mirrors utils/util.py and the windowing in video_depth_anything/video_depth.py."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [0, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8


def get_size(width: int, height: int, target: int = 518, multiple: int = 14) -> Tuple[int, int]:
    """Resize.get_size (util/transform.py:62-107) for keep_aspect_ratio=True, resize_method='lower_bound'.
    Returns (new_width, new_height)."""
    scale_h, scale_w = target / height, target / width
    if scale_w > scale_h:
        scale_h = scale_w
    else:
        scale_w = scale_h

    def constrain(x: float, min_val: int) -> int:
        y = int(np.round(x / multiple) * multiple)
        if y < min_val:
            y = int(np.ceil(x / multiple) * multiple)
        return y

    return constrain(scale_w * width, target), constrain(scale_h * height, target)


def compute_scale_and_shift(prediction: np.ndarray, target: np.ndarray, mask: np.ndarray):
    """Closed-form least squares of utils/util.py:40-62 (compute_scale_and_shift_full)."""
    p = prediction.astype(np.float32)
    t = target.astype(np.float32)
    m = mask.astype(np.float32)
    a00, a01, a11 = np.sum(m * p * p), np.sum(m * p), np.sum(m)
    b0, b1 = np.sum(m * p * t), np.sum(m * t)
    det = a00 * a11 - a01 * a01
    if det == 0:
        return 1, 0
    return (a11 * b0 - a01 * b1) / det, (-a01 * b0 + a00 * b1) / det


def get_interpolate_frames(pre: List[np.ndarray], post: List[np.ndarray]) -> List[np.ndarray]:
    """Linear cross-fade of utils/util.py:65-73 (weights 0, 1/(n-1), ..., 1 on the new window)."""
    assert len(pre) == len(post)
    n = len(pre)
    step = 1.0 / (n - 1)
    w = [0.0] + [i * step for i in range(1, n - 1)] + [1.0]
    return [pre[i] * (1 - w[i]) + post[i] * w[i] for i in range(n)]


def window_table(n_frames: int) -> List[List[int]]:
    """Frame index feeding every slot of every 32-frame window (video_depth.py:88-102). The
    overlap copy acts on INPUT tensors only, so the table is known up front and windows are
    independent work units — this is what the multi-GPU sharding uses."""
    step = INFER_LEN - OVERLAP
    pad = (step - (n_frames % step)) % step + (INFER_LEN - step)
    padded = list(range(n_frames)) + [n_frames - 1] * pad
    table, prev = [], None
    for start in range(0, n_frames, step):
        cur = [padded[start + i] for i in range(INFER_LEN)]
        if prev is not None:
            for i, kf in enumerate(KEYFRAMES):
                cur[i] = prev[kf]
        table.append(cur)
        prev = cur
    return table


def stitch(depth_list: List[np.ndarray], org_len: int) -> np.ndarray:
    """Affine-align each window to the running result and blend the 8 overlap frames
    (video_depth.py:118-156)."""
    out: List[np.ndarray] = []
    ref: List[np.ndarray] = []
    align_len = OVERLAP - INTERP_LEN
    kfs = KEYFRAMES[:align_len]
    for base in range(0, len(depth_list), INFER_LEN):
        if not out:
            out.extend(depth_list[:INFER_LEN])
            ref = [depth_list[base + k] for k in kfs]
            continue
        cur = [depth_list[base + i] for i in range(len(kfs))]
        scale, shift = compute_scale_and_shift(np.concatenate(cur), np.concatenate(ref),
                                               np.concatenate(np.ones_like(ref) == 1))

        def fit(d):
            return np.maximum(d * scale + shift, 0)

        post = [fit(d) for d in depth_list[base + align_len: base + OVERLAP]]
        out[-INTERP_LEN:] = get_interpolate_frames(out[-INTERP_LEN:], post)
        out.extend(fit(depth_list[base + i]) for i in range(OVERLAP, INFER_LEN))
        ref = ref[:1] + [fit(depth_list[base + k]) for k in kfs[1:]]
    return np.stack(out[:org_len], axis=0)


# --------------------------------------------------------------------------------------------- device -> host
_HOST_OUT = {}   # (shape, dtype) -> (pinned tensor, weakref to the ndarray handed out last time)


def release_host_buffers():
    """Drop the pinned result buffers `to_host` keeps for reuse (two 275 MB buffers after 256-frame clips)."""
    _HOST_OUT.clear()


def check_finite(t: torch.Tensor, what: str):
    """Loud range check of a driver's result before it leaves the device (one reduction next to the copy that synchronises
    anyway). 16-bit operand planes are fp16: an activation beyond +-1.3e5 (hi saturates at 65 504, lo carries the next
    65 504) turns into inf / NaN and reaches the
    depth map; no checkpoint the reference ships comes near (INTEGRATION.md 'Range'). VDN_PRECISION=bf16x3 has fp32's range."""
    if not bool(torch.isfinite(t).all()):
        raise FloatingPointError(
            f"{what}: non-finite depth values — an activation left the fp16 range of the operand planes (|x| >= 1.3e5, or 5.7e4 on "
            f"the 8-bit cross-term path). Re-run with VDN_PRECISION=bf16x3 (fp32 range); see INTEGRATION.md 'Range'.")


def to_host(t: torch.Tensor) -> np.ndarray:
    """The clip drivers' single device-to-host copy, through PINNED memory: a pageable `.cpu()` of the 275 MB result of a
    256-frame clip runs at 6-8 GB/s (35-45 ms, measured on the MI355X box), the same copy into a pinned buffer at 55 GB/s
    (5 ms) — 3 % of a one-GPU clip and a fifth of an 8-GPU one. Page-locking itself costs as much as the slow copy, so the
    buffer is kept and REUSED for the next result of the same shape — but only once the caller has dropped the array it got
    (a weak reference tells): a result that is still alive is never overwritten, a fresh buffer is pinned instead.
    NOTE for callers of infer_video_depth / infer_video_depth_sharded: the returned array is a VIEW of that pinned buffer
    (`owndata` is False, `resize` fails); NumPy views and torch.from_numpy keep it alive, a raw pointer / ctypes /
    memoryview export does not — copy (`arr.copy()`) before dropping the array if you hold such an export. Up to two
    result shapes stay pinned; `release_host_buffers()` frees them."""
    if not t.is_cuda:
        return t.numpy()
    import weakref
    key = (tuple(t.shape), t.dtype)
    ent = _HOST_OUT.get(key)
    if ent is not None and ent[1]() is None:
        buf = ent[0]
    else:
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t)                       # blocking device-to-host copy (the stream is synchronised before it returns)
    arr = buf.numpy()
    if len(_HOST_OUT) > 1 and key not in _HOST_OUT:
        _HOST_OUT.clear()              # at most two shapes stay pinned
    _HOST_OUT[key] = (buf, weakref.ref(arr))
    return arr
