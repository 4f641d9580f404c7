"""One-time weight packing into the layouts the kernels read (done on the device, at model build).

All GEMM weights become half [N, ldb] with K contiguous and zero padded to a multiple of 64:
  linear    [N,K]                 -> as is
  conv1x1   [Co,Ci,1,1]           -> [Co, Ci]
  conv3x3   [Co,Ci,3,3]           -> [Co, (ky,kx,ci)]      (matches the NHWC gather order)
  convT k=s [Ci,Co,k,k]           -> [(ky,kx,co), ci]      (pixel-shuffle epilogue order)
  patch     [C,3,14,14]           -> [C, (c,ky,kx)] padded 588 -> 640
  GEGLU     [8c,c] = [h ; gate]   -> 16-row blocks alternating h / gate
  RoPE q/k  rows of each head (2i, 2i+1) -> [re 0-15 | im 0-15 | re 16-31 | im 16-31]
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


class Prec:
    """Operand precision: 16-bit dtype + whether tensors carry a second (lo) plane."""

    def __init__(self, dtype: torch.dtype, split: bool):
        self.dtype, self.split = dtype, split


def _pad_k(w2: torch.Tensor, half) -> "HL":
    """f32 [N,K] -> HL of half [N, ceil64(K)] (zero tail). `half` is a Prec or a bare dtype."""
    from .runtime import HL
    prec = half if isinstance(half, Prec) else Prec(half, False)
    n, k = w2.shape
    kp = (k + 63) // 64 * 64
    full = torch.zeros((n, kp), dtype=torch.float32, device=w2.device)
    full[:, :k] = w2
    return HL.from_float(full, prec.dtype, prec.split)


def linear(w: torch.Tensor, half) -> torch.Tensor:
    return _pad_k(w.detach().float(), half)


def conv1x1(w: torch.Tensor, half) -> torch.Tensor:
    return _pad_k(w.detach().float().reshape(w.shape[0], w.shape[1]), half)


def conv_korder(ci: int) -> int:
    """K order of the implicit-GEMM 3x3 conv: 1 = (ci/64, tap, ci%64) when Cin is a multiple of 64 (consecutive
    K steps then re-read the same 128-byte lines: L2 instead of HBM), else 0 = (tap, ci)."""
    return 1 if ci % 64 == 0 else 0


def conv3x3(w: torch.Tensor, half) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and ci % 8 == 0, w.shape
    wf = w.detach().float()
    if conv_korder(ci):
        # [co, c64, 64, ky, kx] -> [co, c64, ky, kx, 64]
        wk = wf.reshape(co, ci // 64, 64, 3, 3).permute(0, 1, 3, 4, 2).reshape(co, 9 * ci)
    else:
        wk = wf.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    return _pad_k(wk, half)


def conv_transpose(w: torch.Tensor, b: torch.Tensor, half) -> Tuple[torch.Tensor, torch.Tensor]:
    ci, co, k, k2 = w.shape
    assert k == k2
    wg = w.detach().float().permute(2, 3, 1, 0).reshape(k * k * co, ci)  # n = (ky*k+kx)*co + c
    bias = b.detach().float().repeat(k * k).contiguous()
    return _pad_k(wg, half), bias


def patch_embed(w: torch.Tensor, half) -> torch.Tensor:
    c = w.shape[0]
    return _pad_k(w.detach().float().reshape(c, -1), half)  # 588 -> 640


def _geglu_perm(n_half: int, device) -> torch.Tensor:
    assert n_half % 16 == 0
    t = torch.arange(n_half // 16, device=device)
    r = torch.arange(16, device=device)
    h_rows = (t[:, None] * 16 + r[None, :])              # [blocks,16]
    g_rows = h_rows + n_half
    return torch.stack([h_rows, g_rows], dim=1).reshape(-1)  # block t: 16 h rows then 16 gate rows


def geglu(w: torch.Tensor, b: torch.Tensor, half) -> Tuple[torch.Tensor, torch.Tensor]:
    n = w.shape[0]
    perm = _geglu_perm(n // 2, w.device)
    return _pad_k(w.detach().float()[perm], half), b.detach().float()[perm].contiguous()


def rope_perm(c: int, device) -> torch.Tensor:
    """Row permutation for a [C, K] projection whose output is RoPE-rotated per 64-wide head:
    packed position p of head h holds original row h*64 + src(p)."""
    p = torch.arange(64, device=device)
    blk, r = p // 16, p % 16
    pair = (blk // 2) * 16 + r
    src = 2 * pair + (blk % 2)
    heads = torch.arange(c // 64, device=device)
    return (heads[:, None] * 64 + src[None, :]).reshape(-1)


def cat_proj(ws, bs, ropes, half) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Concatenate projections along N; splits flagged in `ropes` get the pair-split row order."""
    wl, bl = [], []
    for w, b, rp in zip(ws, bs, ropes):
        w = w.detach().float()
        b = None if b is None else b.detach().float()
        if rp:
            perm = rope_perm(w.shape[0], w.device)
            w = w[perm]
            b = None if b is None else b[perm]
        wl.append(w)
        bl.append(b)
    wcat = torch.cat(wl, dim=0)
    bias = None if bl[0] is None else torch.cat(bl, dim=0).contiguous()
    return _pad_k(wcat, half), bias


def rope_table(side_y: int, side_x: int, dim: int = 64, theta: float = 10000.0, device=None) -> torch.Tensor:
    """(cos, sin) of sam2 compute_axial_cis (position_encoding.py:192-201): f32 [side_y*side_x, dim/2, 2].
    Pairs 0..dim/4-1 rotate with the x coordinate, dim/4..dim/2-1 with y."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(side_y * side_x, dtype=torch.float32)
    tx = (t % side_x).float()
    ty = torch.div(t, side_x, rounding_mode="floor").float()
    ang = torch.cat([torch.outer(tx, freqs), torch.outer(ty, freqs)], dim=-1)  # [P, dim/2]
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(device)


def temporal_pe(d_model: int, max_len: int) -> torch.Tensor:
    """motion_module.py:195-209 PositionalEncoding buffer [1, max_len, d_model]."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()
