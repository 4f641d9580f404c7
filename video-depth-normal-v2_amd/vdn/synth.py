"""Deterministic synthetic weights and frames (no checkpoints or datasets exist offline).

Counter-based generator: value(seed, key, i) = BoxMuller(splitmix64(fnv1a(key) ^ mix(seed) + i)).
The same numbers come out on any machine / numpy version, so the golden fixtures generated in the
build container (tools/make_golden.py, which loads these weights into the *imported reference*)
stay valid for the GPU box where only this generator travels.

Per-key scale rules keep activations O(1) through 24 blocks and keep the last conv off the
ReLU floor (default init gives an all-zero depth map).
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, key: str, n: int) -> np.ndarray:
    base = (_fnv1a(key) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(base)
    return _splitmix64(idx)


def uniform01(seed: int, key: str, n: int) -> np.ndarray:
    """n float64 in (0,1)."""
    bits = _stream(seed, key, n) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, key: str, shape: Tuple[int, ...]) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = uniform01(seed, key + "#a", m)
    u2 = uniform01(seed, key + "#b", m)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * math.pi * u2), r * np.sin(2 * math.pi * u2)])[:n]
    return z.reshape(shape).astype(np.float32)


def frames_u8(seed: int, n: int, h: int = 518, w: int = 518, smooth: int = 8) -> np.ndarray:
    """Synthetic RGB frames u8[n,h,w,3]: a smooth drifting pattern plus hash noise, so neighbouring
    frames are correlated the way video is (exercises the temporal / memory paths with signal)."""
    out = np.empty((n, h, w, 3), np.uint8)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for t in range(n):
        noise = uniform01(seed, f"frame{t}", h * w * 3).reshape(h, w, 3).astype(np.float32)
        for c in range(3):
            wave = 0.5 + 0.5 * np.sin((xx * (0.013 + 0.004 * c) + yy * (0.009 + 0.003 * c)) + 0.21 * t + c)
            out[t, :, :, c] = np.clip((0.7 * wave + 0.3 * noise[:, :, c]) * 255.0, 0, 255).astype(np.uint8)
    return out


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def depth_clip(seed: int, n: int, h: int, w: int, max_depth: float = 65535.0) -> np.ndarray:
    """Synthetic raw depth clip f32 [n,h,w] in (0, max_depth): a smooth tilted surface with a bump that moves
    from frame to frame plus counter-hash noise (inputs of the v4 / v5 depth refiners)."""
    yy, xx = np.meshgrid(np.linspace(0.0, 1.0, h, dtype=np.float32), np.linspace(0.0, 1.0, w, dtype=np.float32), indexing="ij")
    out = np.empty((n, h, w), dtype=np.float32)
    for t in range(n):
        cx, cy = 0.3 + 0.1 * t, 0.6 - 0.05 * t
        base = 0.25 + 0.3 * xx + 0.15 * yy + 0.2 * np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / 0.02)).astype(np.float32)
        noise = normal(seed, f"depth_clip.{t}", (h, w)).astype(np.float32)
        out[t] = np.clip(base * (1.0 + 0.05 * t) + 0.01 * noise, 0.01, 0.99) * max_depth
    return out


def normalize_frames(frames: np.ndarray) -> np.ndarray:
    """u8[n,h,w,3] RGB -> f32[n,3,h,w], (x/255 - mean)/std  (depth_anything_v2.py:78, transform.py:133-148)."""
    x = frames.astype(np.float32) / 255.0
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------

def _fan_in(shape: Tuple[int, ...], key: str) -> int:
    if len(shape) == 2:
        return shape[1]
    if len(shape) == 4:
        if "resize_layers.0" in key or "resize_layers.1" in key:
            # ConvTranspose2d weight is [Cin, Cout, k, k]; k == stride so each output sees Cin taps
            return shape[0]
        return shape[1] * shape[2] * shape[3]
    return max(1, shape[-1])


def _scale_rules(key: str, shape: Tuple[int, ...], z):
    """Per-key scaling of a unit-normal tensor `z` (numpy array or torch tensor)."""
    leaf = key.split(".")[-1]
    if leaf == "cls_token":
        return 0.5 * z
    if leaf == "pos_embed":
        return 0.3 * z
    if leaf in ("mask_token",):
        return 0.02 * z
    if leaf in ("curr_pos_enc", "maskmem_tpos_enc", "no_mem_embed"):
        return 0.5 * z
    if leaf == "gamma":
        if ".fuser." in key:  # CXBlock layer scale (reference init 1e-6 would hide the block)
            return 0.5 + 0.1 * z
        return 1.0 + 0.1 * z  # ls1/ls2 (reference init_values=1.0)
    is_norm = bool(re.search(r"(^|\.)(norm\d*|norms\.\d+|ff_norm|encoder\.1|bn\d)\.(weight|bias)$", key))
    if is_norm:
        return 1.0 + 0.1 * z if leaf == "weight" else 0.05 * z
    if leaf == "bias":
        if key.endswith("output_conv2.2.bias"):
            return (1.5 if key.startswith(("head.", "temporal_head.")) else 0.5) + 0.0 * z
        if key.endswith("output_conv2.0.bias"):
            return 0.1 + 0.05 * z
        return 0.05 * z
    if leaf == "weight":
        g = 1.0
        if ".attn.qkv." in key or key.endswith(("q_proj.weight", "k_proj.weight", "to_q.weight", "to_k.weight")):
            g = 1.2  # logits with std ~1: attention is neither uniform nor one-hot
        if ".resConfUnit" in key:
            g = 0.7  # residual units: keep the fusion ladder from doubling its scale per stage
        if ".temporal_transformer.proj_out." in key:
            g = 0.5  # reference zero-inits this (motion_module.py:57-58); non-zero so the module is visible
        if key.endswith("output_conv2.2.weight"):
            # zero-sum weights over the (non-negative, post-ReLU) inputs + a positive bias:
            # pre-ReLU depth mostly positive with both signs present
            z = z - z.mean()
            g = 0.6
        return z * (g / math.sqrt(_fan_in(shape, key)))
    return 0.1 * z


def synth_param(seed: int, key: str, shape: Tuple[int, ...]) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    return np.asarray(_scale_rules(key, shape, normal(seed, key, shape)), dtype=np.float32)


def synth_buffer(seed: int, key: str, shape: Tuple[int, ...]):
    """Deterministic values for the BatchNorm buffers of the use_bn head (every other buffer keeps its module-computed
    value): non-trivial running statistics, so that folding them into the convolution is really exercised."""
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    if leaf == "running_mean":
        return np.asarray(0.2 * normal(seed, key, shape), dtype=np.float32)
    if leaf == "running_var":
        return np.asarray(0.6 + 0.5 * np.abs(normal(seed, key, shape)), dtype=np.float32)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    return None


def fast_state_dict(named_shapes, seed: int = 1234):
    """Same scale rules, torch's generator instead of the portable counter hash: seconds instead of
    a minute for ViT-L. Used where only timing matters (bench.py); NOT what the fixtures pin."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in named_shapes:
        shp = tuple(int(v) for v in shp)
        out[k] = _scale_rules(k, shp, torch.randn(shp, generator=g)).float().contiguous()
    return out


def synth_state_dict(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 1234) -> Dict[str, np.ndarray]:
    """named_shapes: (key, shape) for every *parameter* (buffers keep their module-computed values)."""
    return {k: synth_param(seed, k, tuple(s)) for k, s in named_shapes}


# ---------------------------------------------------------------------------------------------
# "checkpoint-like" overlay (fixtures *_heavy): what a trained DINOv2 / DPT checkpoint has and the O(1) synthetic
# weights above do not — a few residual-stream OUTLIER channels two orders of magnitude above the rest at a handful of
# tokens (DINOv2's massive activations), LayerScale factors spread over 1e-2 .. 1, attention heads whose logits spread
# over +-40 (near one-hot softmax rows), and MLP hidden units whose pre-activations reach 1e3 .. 1e4.
# Pure index arithmetic on a finished state dict (numpy arrays or torch tensors, modified in place): the GPU box
# rebuilds exactly the weights the fixtures were made with.
def heavy_overlay(sd, prefix: str = "pretrained."):
    def get(k):
        return sd[prefix + k]

    C = int(get("pos_embed").shape[-1])
    P = int(get("pos_embed").shape[1]) - 1
    depth = 1 + max(int(k.split(".")[-3]) for k in sd if k.startswith(prefix + "blocks.") and k.endswith(".ls1.gamma"))
    # (1) outlier channels of the residual stream at the cls token and three patch tokens
    chans = [7, C // 3 + 5, C - 11]
    toks = [0, 18, 1 + P // 2, P]
    pe = get("pos_embed")
    for i, c in enumerate(chans):
        for j, t in enumerate(toks):
            pe[0, t, c] = (150.0 + 25.0 * i) * (1.0 if (i + j) % 2 == 0 else -1.0)
    # (2) LayerScale spread over two decades (golden-ratio sequence: deterministic, equidistributed)
    for L in range(depth):
        for n, ph in (("ls1", 0.0), ("ls2", 0.37)):
            g = get(f"blocks.{L}.{n}.gamma")
            for c in range(C):
                u = (c * 0.6180339887498949 + ph + 0.11 * L) % 1.0
                g[c] = 10.0 ** (-2.0 + 2.0 * u)
    # (3) peaked attention: head 0 of two blocks gets q and k rows x5 (logits x25: spread >= 40)
    for L in (2 % depth, depth - 2):
        w, b = get(f"blocks.{L}.attn.qkv.weight"), get(f"blocks.{L}.attn.qkv.bias")
        w[0:64] *= 5.0; b[0:64] *= 5.0
        w[C:C + 64] *= 5.0; b[C:C + 64] *= 5.0
    # (4) MLP hidden units with pre-activations of 1e3 .. 1e4 (their fc2 columns then write massive values into the stream)
    for L, units in ((1, ((3, 1000.0), (77, 3000.0))), (depth // 2, ((500, 1000.0),))):
        w1, b1 = get(f"blocks.{L}.mlp.fc1.weight"), get(f"blocks.{L}.mlp.fc1.bias")
        for j, gain in units:
            w1[j] *= gain; b1[j] *= gain
    return sd
