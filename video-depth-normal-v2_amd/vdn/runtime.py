"""Device runtime: workspace arena and tensor-level wrappers over the C-ABI.

PyTorch is plumbing here (device memory, streams); every arithmetic op on the path is a launch of
libvdn_hip.so. Launches go to torch's current stream, so the caller's stream semantics (and
torch.cuda.CUDAGraph capture) apply unchanged. Workspace pointers are stable: buffers come from the
arena by (name, shape, dtype) and nothing is allocated per call after warm-up.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _abi as abi
from ._abi import F16, BF16, F32

_TDT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}
_POISON = bool(int(__import__("os").environ.get("VDN_POISON", "0")))


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class HL:
    """A 16-bit tensor as (hi, lo) planes; lo is None in single-precision-pass mode. In split ("x3")
    mode value = float(hi) + float(lo) carries ~21 mantissa bits and every MFMA product is
    accumulated as hi*hi + hi*lo + lo*hi (include/vdn.h)."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi: torch.Tensor, lo: Optional[torch.Tensor] = None):
        self.hi, self.lo = hi, lo

    @staticmethod
    def from_float(x: torch.Tensor, half: torch.dtype, split: bool) -> "HL":
        hi = x.to(half)
        if not split:
            return HL(hi.contiguous())
        lo = (x.float() - hi.float()).to(half)
        return HL(hi.contiguous(), lo.contiguous())

    def float(self) -> torch.Tensor:
        return self.hi.float() if self.lo is None else self.hi.float() + self.lo.float()

    @property
    def shape(self):
        return self.hi.shape

    def data_ptr(self):
        return self.hi.data_ptr()

    def narrow0(self, start: int, length: int) -> "HL":
        """Rows [start, start+length) of the leading dimension, both planes (a contiguous view)."""
        return HL(self.hi.narrow(0, start, length), None if self.lo is None else self.lo.narrow(0, start, length))

    def zero_(self):
        self.hi.zero_()
        if self.lo is not None:
            self.lo.zero_()
        return self


def _hl(t):
    """(hi tensor, lo pointer or None) of an HL or a plain tensor."""
    if isinstance(t, HL):
        return t.hi, (None if t.lo is None else t.lo.data_ptr())
    return t, None


class Runtime:
    def __init__(self, device: torch.device, half: torch.dtype = torch.float16, split: bool = False):
        if device.type != "cuda":
            raise abi.VdnError("vdn kernels run on an MI355X ('cuda' device under ROCm); there is no CPU path")
        self.device = device
        self.half = half
        self.split = split
        from .pack import Prec
        self.prec = Prec(half, split)
        self.dt = _TDT[half]
        self.zeros = torch.zeros(256, dtype=torch.uint8, device=device)
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self.cu_hint = 0  # vdn_gemm_desc.cu_hint: 0 = whole chip; lanes that co-run set their share
        # MFMA products per P V term of vdn_flash_attn (include/vdn.h), per call: 1 (default) | 2 | 3. Decides which planes the
        # projections write (v_dst / qk_dst) and which kernel every attention launch of this runtime takes — one value per model.
        self.pv_products = int(os.environ.get("VDN_ATTN_PV", "1"))
        assert self.pv_products in (1, 2, 3), self.pv_products
        self.timing: Optional[list] = None  # bench.py: [(tag, start_event, end_event, flop)] for tagged launches

    # ------------------------------------------------------------------ memory
    def buf(self, name: str, shape: Sequence[int], dtype: torch.dtype, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(int(s) for s in shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(key[1], dtype=dtype, device=self.device)
            if not zero and _POISON and t.is_floating_point():
                t.fill_(float("nan"))  # debug: any consumed-before-written element poisons the output
            self._bufs[key] = t
        return t

    def hbuf(self, name, shape, zero=False) -> HL:
        hi = self.buf(name, shape, self.half, zero)
        return HL(hi, self.buf(name + "#lo", shape, self.half, zero) if self.split else None)

    def to_half(self, x: torch.Tensor) -> HL:
        return HL.from_float(x, self.half, self.split)

    def qk8(self, name: str, bh: int, tpad: int) -> Optional[torch.Tensor]:
        """u8 [bh, tpad, 128] zero-initialised plane pair for the attention's 8-bit cross terms, or None when the
        mode has none (single-product or bf16 planes, or VDN_ATTN_QK8=0)."""
        if not (self.split and self.half == torch.float16) or os.environ.get("VDN_ATTN_QK8", "1") == "0":
            return None
        return self.buf(name, (bh, tpad, 128), torch.uint8, zero=True)

    def v_dst(self, vt: HL) -> HL:
        """Destination planes of a V^T head split: when the attention in use is the one-product P V kernel (fp16 split planes
        with the 8-bit Q / K planes, pv_products 1) nothing ever reads V^T's lo plane,
        so the projection does not write it — 2-byte scattered stores: 218 -> 210 us on the batch-8 QKV GEMM."""
        if (vt.lo is not None and self.half == torch.float16 and os.environ.get("VDN_ATTN_QK8", "1") != "0"
                and self.pv_products == 1):
            return HL(vt.hi, None)
        return vt

    def qk_dst(self, t: HL, t8) -> HL:
        """Destination planes of a Q / K head split: with the 8-bit planes (t8) the attention reads e5m2(lo 2^10) and never
        the fp16 lo plane, so the projection does not write it."""
        if self.pv_products == 3:
            return t   # the 3-product P V kernel (flash_attn_kernel<.., false, false>) takes its score cross terms from the fp16 lo planes
        return HL(t.hi, None) if (t8 is not None and t.lo is not None) else t

    def fbuf(self, name, shape, zero=False):
        return self.buf(name, shape, torch.float32, zero)

    def workspace_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    # ------------------------------------------------------------------ launch
    def _launch(self, fn, *args, tag: Optional[str] = None, flop: float = 0.0):
        if tag is not None and self.timing is not None:
            # HIP events on the launch stream (torch's current stream is the stream the kernel runs on)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream)
            e.record()
            self.timing.append((tag, s, e, flop))
        else:
            rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream)
        abi.check(rc, fn.__name__)

    # ------------------------------------------------------------------ ops
    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    def gemm(self, A: torch.Tensor, W: torch.Tensor, M: int, N: int, K: int, *, lda: Optional[int] = None,
             out: Optional[torch.Tensor] = None, ldc: Optional[int] = None, bias=None, act: int = 0, gamma=None,
             rowadd=None, tab=None, tab_mod: int = 0, tab_off: int = 0, res1=None, ldr1=None, res2=None, ldr2=None,
             conv: Optional[dict] = None, relu_a: bool = False, store: int = abi.ST_PLAIN, row_group: int = 0,
             row_skip: int = 0, heads: Optional[dict] = None, convt: Optional[dict] = None, tag: Optional[str] = None,
             a8: Optional[torch.Tensor] = None, w8: Optional[torch.Tensor] = None, out8: Optional[torch.Tensor] = None,
             a_kt: bool = False, w_kt: bool = False, out_kt: bool = False, x8_terms: int = 0):
        d = abi.GemmDesc()
        d.dt = self.dt
        d.M, d.N, d.K = M, N, K
        A, d.A_lo = _hl(A)
        W, d.W_lo = _hl(W)
        d.A = A.data_ptr()
        d.relu_a = 1 if relu_a else 0
        if conv is not None:
            d.a_mode = abi.A_CONV3X3
            d.cB, d.cH, d.cW, d.cC = conv["B"], conv["H"], conv["W"], conv["C"]
            d.cOH, d.cOW, d.cstride = conv["OH"], conv["OW"], conv["stride"]
            d.lda = conv["C"]
            d.conv_korder = conv.get("korder", 1 if conv["C"] % 64 == 0 else 0)  # must match pack.conv3x3
        else:
            d.a_mode = abi.A_PLAIN
            d.lda = lda if lda is not None else K
        d.W = W.data_ptr()
        d.ldb = W.shape[1]
        assert W.shape[0] == N and W.dtype == self.half and A.dtype == self.half, (W.shape, N, W.dtype, A.dtype)
        assert W.is_contiguous()
        d.bias = self._p(bias)
        d.rowadd = self._p(rowadd)
        d.act = act
        d.gamma = self._p(gamma)
        d.tab = self._p(tab)
        d.tab_mod, d.tab_off = tab_mod, tab_off
        if res1 is not None:
            r1, d.res1_lo = _hl(res1)
            d.res1, d.res1_dt, d.ldr1 = r1.data_ptr(), _TDT[r1.dtype], (ldr1 if ldr1 is not None else N)
        if res2 is not None:
            r2, d.res2_lo = _hl(res2)
            d.res2, d.res2_dt, d.ldr2 = r2.data_ptr(), _TDT[r2.dtype], (ldr2 if ldr2 is not None else N)
        d.store = store
        if out is not None:
            oh, d.out_lo = _hl(out)
            d.out = oh.data_ptr()
            d.out_dt = _TDT[oh.dtype]
            d.ldc = ldc if ldc is not None else (N // 2 if store == abi.ST_GEGLU else N)
        d.row_group, d.row_skip = row_group, row_skip
        if heads is not None:
            dst = heads["dst"]
            d.nsplit = len(dst)
            for i, t in enumerate(dst):
                th, d.dst_lo[i] = _hl(t)
                d.dst[i] = th.data_ptr()
                d.transposed[i] = int(heads["transposed"][i])
                d.rope[i] = int(heads.get("rope", (0, 0, 0))[i])
                t8 = (heads.get("dst8") or (None, None, None))[i]
                if t8 is not None:
                    d.dst8[i] = t8.data_ptr()
            d.heads, d.tokens, d.tok_off, d.tpad = heads["heads"], heads["tokens"], heads.get("tok_off", 0), heads["tpad"]
            if heads.get("rope_cs") is not None:
                d.rope_cs, d.rope_mod = heads["rope_cs"].data_ptr(), heads["rope_mod"]
        if convt is not None:
            d.ck, d.cout = convt["k"], convt["cout"]
            d.cB, d.cH, d.cW = convt["B"], convt["H"], convt["W"]
        d.zeros = self.zeros.data_ptr()
        if a8 is not None and w8 is not None:  # 8-bit cross-term planes of both operands (include/vdn.h A8 / W8)
            d.A8, d.W8 = a8.data_ptr(), w8.data_ptr()
        if out8 is not None:
            d.out8 = out8.data_ptr()
        d.a_kt, d.w_kt, d.out_kt = int(a_kt), int(w_kt), int(out_kt)   # K-tile-major planes (include/vdn.h)
        d.x8_terms = int(x8_terms)
        d.cu_hint = self.cu_hint
        if abi.OVERRIDE is not None:   # per-launch kernel-selection knobs (tests / tools); the library itself is stateless
            d.tuning = C.addressof(abi.OVERRIDE)
        if self.split:  # split-K scratch for launches whose tile grid covers a fraction of the chip (include/vdn.h)
            ws = self.buf("splitk_ws", (32 * 1024 * 1024,), torch.float32)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
        self._launch(abi.lib.vdn_gemm, C.byref(d), tag=tag, flop=2.0 * M * N * K)  # the library copies the descriptor before it returns
        return out

    def layernorm(self, x: torch.Tensor, rows: int, Cn: int, w, b, eps: float, *, out_h=None, out_f=None, addvec=None,
                  alpha: float = 1.0, addtab=None, tab_div: int = 1, tab_mod: int = 1, out_group: int = 0, out8=None,
                  kt: bool = False):
        """out8: u8 [2, rows, C] planes of 6-bit rows of the output for the cross-term GEMM (include/vdn.h A8); kt: K-tile-major planes."""
        oh, ol = _hl(out_h) if out_h is not None else (None, None)
        self._launch(abi.lib.vdn_layernorm, x.data_ptr(), _TDT[x.dtype], rows, Cn, w.data_ptr(), b.data_ptr(), eps,
                     self._p(addvec), alpha, self._p(addtab), tab_div, tab_mod, out_group, self._p(oh), ol, self.dt,
                     self._p(out_f), self._p(out8), int(kt))

    def flash_attn(self, Q, K, Vt, out, B: int, H: int, nq: int, nq_pad: int, nk: int, nk_pad: int, scale: float,
                   tag: Optional[str] = None, q8: Optional[torch.Tensor] = None, k8: Optional[torch.Tensor] = None,
                   out8: Optional[torch.Tensor] = None, out_kt: bool = False):
        """q8 / k8: the u8 [B*H, n_pad, 128] planes the projection wrote through heads['dst8'] (8-bit cross terms).
        out8 / out_kt: planes of 6-bit rows of the output and the K-tile-major layout for the cross-term GEMM that follows."""
        (Q, ql), (K, kl), (Vt, vl), (out, ol) = _hl(Q), _hl(K), _hl(Vt), _hl(out)
        self._launch(abi.lib.vdn_flash_attn, self.dt, Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), out.data_ptr(), ql, kl,
                     vl, ol, self._p(q8), self._p(k8), self._p(out8), int(out_kt), B, H, nq, nq_pad, nk, nk_pad, scale, self.pv_products, tag=tag,
                     flop=4.0 * B * H * nq * nk * 64)

    def temporal_attn(self, qkv, out, Bv: int, T: int, D: int, c: int, heads: int, scale: float, rope_cs=None):
        (qkv, ql), (out, ol) = _hl(qkv), _hl(out)
        self._launch(abi.lib.vdn_temporal_attn, self.dt, qkv.data_ptr(), out.data_ptr(), ql, ol, Bv, T, D, c, heads, scale,
                     self._p(rope_cs))

    def temporal_attn_last(self, pool: torch.Tensor, slots, pe_q, pe_k, pe_v, out, HW: int, c: int, scale: float):
        """Newest frame attends over the projected cache: `pool` f32 [ring slots, HW, 3c], `slots` the window's ring-slot
        indices, oldest first (host ints: they travel in the launch arguments)."""
        T = len(slots)
        tab = (C.c_int32 * T)(*slots)
        (out, ol) = _hl(out)
        self._launch(abi.lib.vdn_temporal_attn_last, self.dt, pool.data_ptr(), pool.stride(0), tab, T, HW, c, pe_q.data_ptr(),
                     pe_k.data_ptr(), pe_v.data_ptr(), scale, out.data_ptr(), ol)

    def groupnorm(self, x, y, F: int, HW: int, Cn: int, groups: int, w, b, eps: float):
        nsplit = 16 if HW >= 1024 else 4
        part = self.fbuf("gn_partial", (F, nsplit, groups, 2))
        (x, xl), (y, yl) = _hl(x), _hl(y)
        self._launch(abi.lib.vdn_groupnorm, self.dt, x.data_ptr(), xl, y.data_ptr(), yl, F, HW, Cn, groups, w.data_ptr(),
                     b.data_ptr(), eps, part.data_ptr(), nsplit)

    def upsample(self, x, y, B: int, IH: int, IW: int, OH: int, OW: int, Cn: int):
        (x, xl), (y, yl) = _hl(x), _hl(y)
        self._launch(abi.lib.vdn_upsample_bilinear, self.dt, x.data_ptr(), xl, y.data_ptr(), yl, B, IH, IW, OH, OW, Cn)

    def upsample_f32(self, x, y, B: int, IH: int, IW: int, OH: int, OW: int, relu: bool = False):
        self._launch(abi.lib.vdn_upsample_bilinear_f32, x.data_ptr(), y.data_ptr(), B, IH, IW, OH, OW, int(relu))

    def stitch_fit(self, pred: torch.Tensor, target: torch.Tensor, coef: torch.Tensor):
        """coef[0:2] <- least-squares (scale, shift) of pred onto target (utils/util.py:40-62), on the device."""
        ws = self.buf("stitch_ws", (abi.lib.vdn_stitch_workspace_bytes() // 8,), torch.float64)
        assert pred.is_contiguous() and target.is_contiguous() and pred.numel() == target.numel()
        self._launch(abi.lib.vdn_stitch_fit, pred.data_ptr(), target.data_ptr(), pred.numel(), ws.data_ptr(), coef.data_ptr())

    def stitch_apply(self, window: torch.Tensor, coef: torch.Tensor, out_tail: torch.Tensor, out_new: torch.Tensor,
                     ref1: torch.Tensor, align_len: int, overlap: int, ref_frame: int):
        T, hw = window.shape[0], window[0].numel()
        assert window.is_contiguous() and out_tail.is_contiguous() and out_new.is_contiguous() and ref1.is_contiguous()
        assert out_tail.shape[0] == overlap - align_len and out_new.shape[0] == T - overlap
        self._launch(abi.lib.vdn_stitch_apply, window.data_ptr(), coef.data_ptr(), out_tail.data_ptr(), out_new.data_ptr(),
                     ref1.data_ptr(), hw, T, align_len, overlap, ref_frame)

    def frame_median(self, x: torch.Tensor, median: torch.Tensor):
        """median[f] = torch.quantile(x[f], 0.5) for f32 x [F, ...] (exact radix select on the device)."""
        F = x.shape[0]
        ws = self.buf("median_ws", ((abi.lib.vdn_frame_median_workspace_bytes(F) + 7) // 8,), torch.int64)
        assert x.is_contiguous() and x.dtype == torch.float32 and median.numel() == F
        self._launch(abi.lib.vdn_frame_median, x.data_ptr(), F, x[0].numel(), median.data_ptr(), ws.data_ptr())

    def refine_scale(self, x, median, w: float, b: float, max_log_scale: float, max_depth: float, out, scale_out=None):
        self._launch(abi.lib.vdn_refine_scale, x.data_ptr(), median.data_ptr(), x.shape[0], x[0].numel(), w, b, max_log_scale,
                     max_depth, out.data_ptr(), self._p(scale_out))

    def refine_pack(self, d, out, normals: bool = True):
        F, H, W = d.shape
        self._launch(abi.lib.vdn_refine_pack, d.data_ptr(), out.data_ptr(), F, H, W, int(normals))

    def refine_finish(self, scaled, depth, w: float, b: float, max_depth: float, residual: bool, out):
        self._launch(abi.lib.vdn_refine_finish, self._p(scaled), depth.data_ptr(), w, b, max_depth, int(residual), out.data_ptr(),
                     depth.numel())

    def patchify(self, img, rows, B: int, H: int, W: int, ldk: int):
        rows, rl = _hl(rows)
        self._launch(abi.lib.vdn_patchify, self.dt, img.data_ptr(), rows.data_ptr(), rl, B, H, W, ldk)

    def fill_row(self, x, vec, B: int, rows_per_b: int, row: int, Cn: int):
        self._launch(abi.lib.vdn_fill_row, x.data_ptr(), vec.data_ptr(), B, rows_per_b, row, Cn)

    def bicubic(self, src, dst, ih: int, iw: int, oh: int, ow: int, Cn: int, scale_rows: float, scale_cols: float):
        self._launch(abi.lib.vdn_bicubic, src.data_ptr(), dst.data_ptr(), ih, iw, oh, ow, Cn, scale_rows, scale_cols)

    def preprocess_u8(self, frames_u8: torch.Tensor, H: int, W: int, mean, std, swap_rb: bool = False) -> torch.Tensor:
        """u8 [n,h,w,3] on the device -> normalised f32 [n,3,H,W] (cubic resize + /255 + mean / std), one launch."""
        n, h, w, _ = frames_u8.shape
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
        out = torch.empty((n, 3, H, W), dtype=torch.float32, device=self.device)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        self._launch(abi.lib.vdn_preprocess, frames_u8.data_ptr(), n, h, w, int(swap_rb), out.data_ptr(), H, W, m3, s3)
        return out

    def add_vec(self, x, vec, alpha: float, y, rows: int, Cn: int):
        self._launch(abi.lib.vdn_add_vec, x.data_ptr(), vec.data_ptr(), alpha, y.data_ptr(), rows, Cn)

    def depth_tail(self, x: torch.Tensor, w: HL, bias2, w1, b1: float, depth, B: int, IH: int, IW: int, Cn: int, OH: int,
                   OW: int, relu: bool):
        """resize(align_corners) -> conv3x3 + ReLU -> conv1x1 [+ ReLU] in one launch; x f32 NHWC (split-plane modes only)."""
        assert x.dtype == torch.float32 and x.is_contiguous() and w.lo is not None and w.hi.shape[0] == 32
        self._launch(abi.lib.vdn_depth_tail, self.dt, x.data_ptr(), B, IH, IW, Cn, w.hi.data_ptr(), w.lo.data_ptr(),
                     w.hi.shape[1], bias2.data_ptr(), w1.data_ptr(), b1, depth.data_ptr(), OH, OW, int(relu))

    def head_out(self, feat, w, bias: float, depth, M: int, Cn: int, relu: bool):
        feat, fl = _hl(feat)
        self._launch(abi.lib.vdn_head_out, self.dt, feat.data_ptr(), fl, w.data_ptr(), bias, depth.data_ptr(), M, Cn,
                     int(relu))

    def mask_down1(self, depth, out, B, H, W, OH, OW, w):
        self._launch(abi.lib.vdn_mask_down1, depth.data_ptr(), out.data_ptr(), B, H, W, OH, OW, w.data_ptr())

    def mask_down2(self, x, out, B, H, W, OH, OW, w):
        self._launch(abi.lib.vdn_mask_down2, x.data_ptr(), out.data_ptr(), B, H, W, OH, OW, w.data_ptr())

    def dwconv7(self, x, y, B, H, W, Cn, w, bias):
        self._launch(abi.lib.vdn_dwconv7, x.data_ptr(), y.data_ptr(), B, H, W, Cn, w.data_ptr(), bias.data_ptr())

    def addtab_cast(self, x, tab, tab_div: int, tab_mod: int, y, rows: int, Cn: int):
        yh, yl = _hl(y)
        self._launch(abi.lib.vdn_addtab_cast, self.dt, x.data_ptr(), self._p(tab), tab_div, tab_mod, yh.data_ptr(), yl, rows, Cn)

    def pack_x8(self, t: HL, hi_kt: torch.Tensor, planes8: torch.Tensor, order: int = 0):
        """Split planes [rows, ld] -> the K-tile-major hi plane and the two planes of 6-bit rows of the cross-term GEMM's A operand
        (include/vdn.h vdn_pack_x8), for activations whose producer writes plain split planes."""
        rows, ld = t.hi.shape
        self._launch(abi.lib.vdn_pack_x8, t.hi.data_ptr(), t.lo.data_ptr(), rows, ld, hi_kt.data_ptr(), planes8.data_ptr(), 1, order)

    def pack_x8_f32(self, x: torch.Tensor, hi_kt: torch.Tensor, planes8: torch.Tensor):
        """fp32 [rows, ld] -> the cross-term GEMM's A operand (include/vdn.h vdn_pack_x8_f32)."""
        rows, ld = x.shape
        self._launch(abi.lib.vdn_pack_x8_f32, x.data_ptr(), rows, ld, hi_kt.data_ptr(), planes8.data_ptr())

    def cast(self, x, y):
        self._launch(abi.lib.vdn_cast, x.data_ptr(), _TDT[x.dtype], y.data_ptr(), _TDT[y.dtype], x.numel())
