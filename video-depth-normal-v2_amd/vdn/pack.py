"""One-time weight packing into the layouts the kernels read: thin callers of the C-ABI packers
(include/vdn.h: vdn_pack_weight / vdn_pack_bias, csrc/pack.hip), run on the device at model build.

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


def _prec(half) -> Prec:
    return half if isinstance(half, Prec) else Prec(half, False)


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _pack_into(kind: int, w: torch.Tensor, d0: int, d1: int, d2: int, out, row0: int = 0):
    """vdn_pack_weight (include/vdn.h) of one fp32 parameter into rows [row0, row0 + rows) of the planes `out`."""
    from . import _abi as abi
    wf = w.detach().float().contiguous()
    if not wf.is_cuda:
        raise abi.VdnError("weights are packed on the device by libvdn_hip.so: move the module to the GPU first")
    ldb = out.hi.shape[1]
    off = row0 * ldb * out.hi.element_size()
    dt = abi.F16 if out.hi.dtype == torch.float16 else abi.BF16
    abi.check(abi.lib.vdn_pack_weight(dt, kind, wf.data_ptr(), d0, d1, d2, out.hi.data_ptr() + off,
                                      None if out.lo is None else out.lo.data_ptr() + off, ldb, _stream(wf)), "vdn_pack_weight")
    return out


def _planes(kind: int, d0: int, d1: int, d2: int, half, device, rows=None):
    from . import _abi as abi
    from .runtime import HL
    prec = _prec(half)
    r, ldb = abi.lib.vdn_pack_rows(kind, d0, d1, d2), abi.lib.vdn_pack_ldb(kind, d0, d1, d2)
    if r < 0 or ldb < 0:
        raise ValueError(f"vdn_pack: unsupported shape for layout {kind}: {(d0, d1, d2)}")
    hi = torch.empty((rows or r, ldb), dtype=prec.dtype, device=device)
    return HL(hi, torch.empty_like(hi) if prec.split else None)


def _pack(kind: int, w: torch.Tensor, d0: int, d1: int, d2: int, half):
    return _pack_into(kind, w, d0, d1, d2, _planes(kind, d0, d1, d2, half, w.device))


def _pack_bias(kind: int, b: torch.Tensor, d0: int, d1: int, d2: int) -> torch.Tensor:
    from . import _abi as abi
    bf = b.detach().float().contiguous()
    out = torch.empty((abi.lib.vdn_pack_rows(kind, d0, d1, d2),), dtype=torch.float32, device=bf.device)
    abi.check(abi.lib.vdn_pack_bias(kind, bf.data_ptr(), d0, d1, d2, out.data_ptr(), _stream(bf)), "vdn_pack_bias")
    return out


def _pad_k(w2: torch.Tensor, half) -> "HL":
    """f32 [N,K] -> HL of half [N, ceil64(K)] (zero tail). `half` is a Prec or a bare dtype."""
    from . import _abi as abi
    return _pack(abi.PACK_LINEAR, w2, w2.shape[0], w2.shape[1], 0, half)


def linear(w: torch.Tensor, half) -> torch.Tensor:
    return _pad_k(w, half)


def conv1x1(w: torch.Tensor, half) -> torch.Tensor:
    return _pad_k(w.reshape(w.shape[0], w.shape[1]), half)


def conv_korder(ci: int) -> int:
    """K order of the implicit-GEMM 3x3 conv: 1 = (ci/64, tap, ci%64) when Cin is a multiple of 64 (consecutive
    K steps then re-read the same 128-byte lines: L2 instead of HBM), else 0 = (tap, ci)."""
    return 1 if ci % 64 == 0 else 0


def conv3x3(w: torch.Tensor, half) -> torch.Tensor:
    from . import _abi as abi
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and ci % 8 == 0, w.shape
    return _pack(abi.PACK_CONV3X3, w, co, ci, 0, half)


def conv3x3_taps(w: torch.Tensor, half) -> torch.Tensor:
    """[Co,Ci,3,3] -> [Co, (ky,kx,ci)] always tap-major (the fused depth tail, csrc/tail.hip, walks K tap by tap)."""
    from . import _abi as abi
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and ci % 32 == 0, w.shape
    return _pack(abi.PACK_CONV3X3_TAPS, w, co, ci, 0, half)


def conv_transpose(w: torch.Tensor, b: torch.Tensor, half) -> Tuple[torch.Tensor, torch.Tensor]:
    from . import _abi as abi
    ci, co, k, k2 = w.shape
    assert k == k2
    return _pack(abi.PACK_CONVT, w, ci, co, k, half), _pack_bias(abi.PACK_CONVT, b, ci, co, k)  # n = (ky*k+kx)*co + c


def patch_embed(w: torch.Tensor, half) -> torch.Tensor:
    return _pad_k(w.reshape(w.shape[0], -1), half)  # 588 -> 640


def geglu(w: torch.Tensor, b: torch.Tensor, half) -> Tuple[torch.Tensor, torch.Tensor]:
    from . import _abi as abi
    n, k = w.shape
    return _pack(abi.PACK_GEGLU, w, n, k, 0, half), _pack_bias(abi.PACK_GEGLU, b, n, k, 0)


def cat_proj(ws, bs, ropes, half) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Concatenate projections along N; splits flagged in `ropes` get the pair-split row order."""
    from . import _abi as abi
    k = ws[0].shape[1]
    total = sum(w.shape[0] for w in ws)
    out = _planes(abi.PACK_LINEAR, total, k, 0, half, ws[0].device)
    bl, row0 = [], 0
    for w, b, rp in zip(ws, bs, ropes):
        kind = abi.PACK_ROPE if rp else abi.PACK_LINEAR
        _pack_into(kind, w, w.shape[0], k, 0, out, row0)
        bl.append(None if b is None else _pack_bias(kind, b, w.shape[0], k, 0))
        row0 += w.shape[0]
    bias = None if bl[0] is None else torch.cat(bl, dim=0).contiguous()
    return out, bias


ORDER_NATURAL, ORDER_GEMM, ORDER_ATTN = 0, 1, 2   # include/vdn.h vdn_pack_x8: who writes the A planes of the GEMM


def planes8(t, order: int = ORDER_NATURAL, kt: bool = False) -> torch.Tensor:
    """HL [rows, ld] (ld % 64 == 0) -> u8 [2, rows, ld]: the x6 rows (e3m2 codes + E8M0 scale per 32 values) of the hi plane
    and of the remainder plane, the operand planes of the cross-term GEMM (include/vdn.h A8 / W8), made on the device by
    vdn_pack_x8; kt: K-tile-major ([ld/64][rows][64] per plane)."""
    from . import _abi as abi
    rows, ld = t.hi.shape
    assert ld % 64 == 0 and t.lo is not None and t.hi.dtype == torch.float16
    p8 = torch.empty((2, rows, ld), dtype=torch.uint8, device=t.hi.device)
    abi.check(abi.lib.vdn_pack_x8(t.hi.data_ptr(), t.lo.data_ptr(), rows, ld, None, p8.data_ptr(), int(kt), order, _stream(t.hi)),
              "vdn_pack_x8")
    return p8


def x6_columns(order: int) -> torch.Tensor:
    """[2, 32]: the column of a 64-wide slab at stream position p of half h (include/vdn.h vdn_pack_x8 `order`)."""
    h, p = torch.arange(2)[:, None], torch.arange(32)[None, :]
    if order == ORDER_GEMM:
        return 32 * (p >> 4) + 16 * ((p >> 3) & 1) + 8 * h + (p & 7)
    if order == ORDER_ATTN:
        return 32 * (p >> 4) + 8 * ((p >> 2) & 3) + 4 * h + (p & 3)
    return 32 * h + p


def decode6(plane: torch.Tensor, rows: int, K: int, order: int = ORDER_NATURAL, kt: bool = False) -> torch.Tensor:
    """The values one plane of x6 rows stands for, f32 [rows, K] in natural column order (tests and tools: what the MFMA sees)."""
    b = plane.reshape(K // 64, rows, 2, 32) if kt else plane.reshape(rows, K // 64, 2, 32).permute(1, 0, 2, 3)
    b = b.to(torch.int64)                                              # [slab, row, half, 32 bytes]
    words = [sum(b[..., 8 * w + i] << (8 * i) for i in range(8)) for w in range(3)]   # the 24 code bytes as three 64-bit words
    codes = []
    for p in range(32):
        bit = 6 * p
        w, o = bit // 64, bit % 64
        v = (words[w] >> o) & 63 if o <= 58 else ((words[w] >> o) & ((1 << (64 - o)) - 1)) | ((words[w + 1] & ((1 << (o - 58)) - 1)) << (64 - o))
        codes.append(v & 63)
    c = torch.stack(codes, dim=-1)                                     # [slab, row, half, 32]
    sign, e, m = (c >> 5) & 1, (c >> 2) & 7, c & 3
    val = torch.where(e > 0, (1.0 + 0.25 * m.double()) * torch.pow(2.0, (e - 3).double()), 0.25 * m.double() * 0.25)
    val = torch.where(sign > 0, -val, val) * torch.pow(2.0, (b[..., 24] - 127).double())[..., None]
    out = torch.empty(rows, K, dtype=torch.float64, device=plane.device)
    cols = x6_columns(order).to(plane.device)                          # [2, 32]
    for s in range(K // 64):
        for h in range(2):
            out[:, 64 * s + cols[h]] = val[s, :, h, :]
    return out.float()


class X8:
    """Weight planes of the cross-term GEMM (include/vdn.h W8 / w_kt): the fp16 hi plane K-tile-major ([ld/32][N][32]) and
    the two planes of x6 rows u8 [2][ld/64][N][64], made on the device by vdn_pack_x8 in the stream ORDER of the kernel that
    writes the GEMM's activation planes."""
    __slots__ = ("hi", "p8", "rows", "ld", "order")

    def __init__(self, w, order: int = ORDER_NATURAL):
        from . import _abi as abi
        assert w.lo is not None and w.hi.dtype == torch.float16 and w.hi.shape[1] % 64 == 0
        self.rows, self.ld = w.hi.shape
        self.order = order
        self.hi = torch.empty_like(w.hi)
        self.p8 = torch.empty((2, self.rows, self.ld), dtype=torch.uint8, device=w.hi.device)
        abi.check(abi.lib.vdn_pack_x8(w.hi.data_ptr(), w.lo.data_ptr(), self.rows, self.ld, self.hi.data_ptr(),
                                      self.p8.data_ptr(), 1, order, _stream(w.hi)), "vdn_pack_x8")


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
