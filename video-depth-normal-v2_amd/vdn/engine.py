"""Executors: the reference's forward passes as sequences of libvdn_hip.so launches.

Data layout in HBM (everything channels-last, resident for the whole forward):
  residual streams (ViT tokens, memory-attention state, temporal hidden state)  f32 [rows, C]
  GEMM operands / activations between convs                                      half [rows, C]
  attention operands   Q,K  half [B*heads, tpad, 64];  V^T half [B*heads, 64, tpad]
  memory bank          per layer ring of projected+rotated K / V^T (6 slots), written once per frame
Every function cites the reference code it stands for (paths relative to the reference root).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import _abi as abi
from . import pack
from .runtime import Runtime, ceil_to

GELU, RELU = abi.ACT_GELU, abi.ACT_RELU
PATCH = 14


# =============================================================================================
class EncoderEngine:
    """DINOv2 ViT: depth_anything_v2/dinov2.py:212-231,271-321 + dinov2_layers/{block,attention,mlp}.py."""

    def __init__(self, rt: Runtime, mod, cfg: dict):
        self.rt, self.cfg = rt, cfg
        self.C, self.heads, self.depth, self.taps = cfg["dim"], cfg["heads"], cfg["depth"], cfg["taps"]
        h = rt.prec
        self.w_patch = pack.patch_embed(mod.patch_embed.proj.weight, h)
        self.b_patch = pack.f32(mod.patch_embed.proj.bias)
        self.cls = pack.f32(mod.cls_token).reshape(-1)
        self.pos = pack.f32(mod.pos_embed).reshape(-1, self.C)  # [1+37*37, C]
        self.blocks = []
        self.hidden = cfg.get("swiglu") or 4 * self.C   # FFN width: Mlp 4C, or the SwiGLU hidden size of ViT-g
        for b in mod.blocks:
            blk = dict(
                n1w=pack.f32(b.norm1.weight), n1b=pack.f32(b.norm1.bias),
                wqkv=pack.linear(b.attn.qkv.weight, h), bqkv=pack.f32(b.attn.qkv.bias),
                wproj=pack.linear(b.attn.proj.weight, h), bproj=pack.f32(b.attn.proj.bias),
                ls1=pack.f32(b.ls1.gamma),
                n2w=pack.f32(b.norm2.weight), n2b=pack.f32(b.norm2.bias),
                ls2=pack.f32(b.ls2.gamma))
            if cfg.get("swiglu"):
                # SwiGLU (swiglu_ffn.py:29-33): silu(x1) * x2 with [x1 ; x2] = w12 x. The gated epilogue computes
                # h * act(gate) from rows packed [h ; gate], so the halves are swapped: h = x2, gate = x1, act = SiLU.
                Hd = cfg["swiglu"]
                w12, b12 = b.mlp.w12.weight, b.mlp.w12.bias
                blk["wfc1"], blk["bfc1"] = pack.geglu(torch.cat([w12[Hd:], w12[:Hd]]), torch.cat([b12[Hd:], b12[:Hd]]), h)
                blk["wfc2"], blk["bfc2"] = pack.linear(b.mlp.w3.weight, h), pack.f32(b.mlp.w3.bias)
            else:
                blk["wfc1"], blk["bfc1"] = pack.linear(b.mlp.fc1.weight, h), pack.f32(b.mlp.fc1.bias)
                blk["wfc2"], blk["bfc2"] = pack.linear(b.mlp.fc2.weight, h), pack.f32(b.mlp.fc2.bias)
            self.blocks.append(blk)
        self.nw, self.nb = pack.f32(mod.norm.weight), pack.f32(mod.norm.bias)
        self._pos_cache = {}
        # 8-bit cross terms for the four linears of a block (csrc/gemm_x8.hip): fp16 split planes only, K-tile-major
        # weight planes made once here. VDN_X8=0 keeps the three-fp16-product kernels.
        self.x8 = (rt.split and rt.half == torch.float16 and not cfg.get("swiglu") and self.C % 64 == 0
                   and os.environ.get("VDN_X8", "1") != "0")
        if self.x8:
            for blk in self.blocks:
                blk["x8"] = {k: pack.X8(blk[k], o) for k, o in (("wqkv", pack.ORDER_NATURAL), ("wproj", pack.ORDER_ATTN),
                                                                   ("wfc1", pack.ORDER_NATURAL), ("wfc2", pack.ORDER_GEMM))}
        # per-layer precision budget (include/vdn.h x8_terms; profiles/r03_precision_budget.md): VDN_X8_TERMS = "fc2=1,proj=1@12-23"
        # drops a cross term of the named linears (1: A_lo W_hi^T, 2: A_hi W_lo^T), optionally for blocks a..b only
        self.x8_terms = [dict(qkv=0, proj=0, fc1=0, fc2=0) for _ in self.blocks]
        for item in filter(None, os.environ.get("VDN_X8_TERMS", "").split(",")):
            name, val = item.split("=")
            val, _, rng = val.partition("@")
            a, b = (int(v) for v in rng.split("-")) if rng else (0, len(self.blocks) - 1)
            for L in range(a, b + 1):
                self.x8_terms[L][name] = int(val)

    def _pos_for(self, ph: int, pw: int):
        """interpolate_pos_encoding (dinov2.py:179-210): identity for the square 37x37 grid, else bicubic
        resample with scale ((ph+0.1)/37, (pw+0.1)/37). Returns (f32 [1+P, C] table, f32 [C] cls row)."""
        key = (ph, pw)
        if key not in self._pos_cache:
            rt, C = self.rt, self.C
            gs = int(math.sqrt(self.pos.shape[0] - 1))
            if ph * pw == gs * gs and ph == pw:
                table = self.pos
            else:
                table = torch.empty((1 + ph * pw, C), dtype=torch.float32, device=rt.device)
                table[0].copy_(self.pos[0])
                sx, sy = float(ph + 0.1) / gs, float(pw + 0.1) / gs
                rt.bicubic(self.pos[1:], table[1:], gs, gs, ph, pw, C, sx, sy)
            self._pos_cache[key] = (table, (self.cls + table[0]).contiguous())
        return self._pos_cache[key]

    def run(self, x: torch.Tensor, want_f32_last: bool = False, tap_out=None):
        """x f32 [Bf,3,H,W] -> 4 final-normed patch-token maps, half [Bf*P, C] each (cls dropped,
        dinov2.py:309-312); optionally also the last one in f32 (input of the memory block).
        `tap_out`: 4 caller-owned HL destinations [Bf*P, C] (the clip-level tap cache of the video driver)."""
        rt, C, Hh = self.rt, self.C, self.heads
        Bf, _, H, W = x.shape
        assert H % PATCH == 0 and W % PATCH == 0, "input sides must be multiples of 14 (patch_embed.py:73-74)"
        ph, pw = H // PATCH, W // PATCH
        P, N = ph * pw, ph * pw + 1
        M = Bf * N
        table, cls_row = self._pos_for(ph, pw)
        rows = rt.hbuf("patch_rows", (Bf * P, 640))
        rt.patchify(x, rows, Bf, H, W, 640)
        tok = rt.fbuf("tokens", (M, C))
        rt.fill_row(tok, cls_row, Bf, N, 0, C)
        rt.gemm(rows, self.w_patch, Bf * P, C, 640, out=tok, bias=self.b_patch, tab=table, tab_mod=P, tab_off=1,
                row_group=P, row_skip=1)
        npad = ceil_to(N, 64)
        q = rt.hbuf("enc_q", (Bf * Hh, npad, 64), zero=True)
        k = rt.hbuf("enc_k", (Bf * Hh, npad, 64), zero=True)
        vt = rt.hbuf("enc_vt", (Bf * Hh, 64, npad), zero=True)
        q8, k8 = rt.qk8("enc_q8", Bf * Hh, npad), rt.qk8("enc_k8", Bf * Hh, npad)  # e5m2 planes for the score cross terms
        Hd = self.hidden
        heads = dict(dst=[rt.qk_dst(q, q8), rt.qk_dst(k, k8), rt.v_dst(vt)], dst8=[q8, k8, None], transposed=[0, 0, 1], heads=Hh, tokens=N, tpad=npad)
        outs, last_f32 = [], None
        readout = getattr(self, "readout", None)   # ReadoutEngine when the head was built with use_clstoken
        probe = getattr(self, "probe", None)   # tests only: callable(block index, fp32 token stream [M, C]); -1 = input of block 0
        if probe is not None:
            probe(-1, tok)
        # 8-bit cross-term path: large batches only (its kernel has 256 x 256 tiles: M >= 4096 keeps every launch near a
        # round of the chip or more), with the default attention (the only producer of the 8-bit output planes)
        use8 = self.x8 and M >= int(os.environ.get("VDN_X8_MIN_ROWS", "4096")) and q8 is not None and rt.pv_products != 3
        if use8:
            from .runtime import HL
            # activations between the linears as K-tile-major planes: fp16 hi + the 6-bit rows of hi and remainder — no fp16 lo plane
            hn_k, hn8 = HL(rt.buf("enc_ln_kt", (M, C), rt.half)), rt.buf("enc_ln8", (2, M, C), torch.uint8)
            att_k, att8 = HL(rt.buf("enc_att_kt", (M, C), rt.half)), rt.buf("enc_att8", (2, M, C), torch.uint8)
            f1_k, f18 = HL(rt.buf("enc_fc1_kt", (M, Hd), rt.half)), rt.buf("enc_fc18", (2, M, Hd), torch.uint8)
            kt = dict(a_kt=True, w_kt=True)
        else:
            hn, att, f1 = rt.hbuf("enc_ln", (M, C)), rt.hbuf("enc_att", (M, C)), rt.hbuf("enc_fc1", (M, Hd))
        for i, b in enumerate(self.blocks):
            if use8:
                x8, xt = b["x8"], self.x8_terms[i]
                rt.layernorm(tok, M, C, b["n1w"], b["n1b"], 1e-6, out_h=hn_k, out8=hn8, kt=True)
                rt.gemm(hn_k, HL(x8["wqkv"].hi), M, 3 * C, C, bias=b["bqkv"], store=abi.ST_HEADS, heads=heads, tag="enc_linear",
                        a8=hn8, w8=x8["wqkv"].p8, x8_terms=xt["qkv"], **kt)
                rt.flash_attn(q, k, vt, att_k, Bf, Hh, N, npad, N, npad, 64 ** -0.5, tag="enc_attn", q8=q8, k8=k8, out8=att8, out_kt=True)
                rt.gemm(att_k, HL(x8["wproj"].hi), M, C, C, bias=b["bproj"], gamma=b["ls1"], res1=tok, out=tok, tag="enc_linear",
                        a8=att8, w8=x8["wproj"].p8, x8_terms=xt["proj"], **kt)
                rt.layernorm(tok, M, C, b["n2w"], b["n2b"], 1e-6, out_h=hn_k, out8=hn8, kt=True)
                rt.gemm(hn_k, HL(x8["wfc1"].hi), M, Hd, C, bias=b["bfc1"], act=GELU, out=f1_k, out8=f18, out_kt=True, tag="enc_linear",
                        a8=hn8, w8=x8["wfc1"].p8, x8_terms=xt["fc1"], **kt)
                rt.gemm(f1_k, HL(x8["wfc2"].hi), M, C, Hd, bias=b["bfc2"], gamma=b["ls2"], res1=tok, out=tok, tag="enc_linear",
                        a8=f18, w8=x8["wfc2"].p8, x8_terms=xt["fc2"], **kt)
            else:
                rt.layernorm(tok, M, C, b["n1w"], b["n1b"], 1e-6, out_h=hn)
                rt.gemm(hn, b["wqkv"], M, 3 * C, C, bias=b["bqkv"], store=abi.ST_HEADS, heads=heads, tag="enc_linear")
                rt.flash_attn(q, k, vt, att, Bf, Hh, N, npad, N, npad, 64 ** -0.5, tag="enc_attn", q8=q8, k8=k8)
                rt.gemm(att, b["wproj"], M, C, C, bias=b["bproj"], gamma=b["ls1"], res1=tok, out=tok, tag="enc_linear")
                rt.layernorm(tok, M, C, b["n2w"], b["n2b"], 1e-6, out_h=hn)
                if self.cfg.get("swiglu"):
                    rt.gemm(hn, b["wfc1"], M, 2 * Hd, C, bias=b["bfc1"], store=abi.ST_GEGLU, act=abi.ACT_SILU, out=f1, tag="enc_linear")
                else:
                    rt.gemm(hn, b["wfc1"], M, Hd, C, bias=b["bfc1"], act=GELU, out=f1, tag="enc_linear")
                rt.gemm(f1, b["wfc2"], M, C, Hd, bias=b["bfc2"], gamma=b["ls2"], res1=tok, out=tok, tag="enc_linear")
            if probe is not None:
                probe(i, tok)
            if i in self.taps:
                j = self.taps.index(i)
                t = tap_out[j] if tap_out is not None else rt.hbuf(f"tap{j}", (Bf * P, C))
                f = None
                if want_f32_last and j == len(self.taps) - 1:
                    f = last_f32 = rt.fbuf("tap_last_f32", (Bf * P, C))
                if readout is None:
                    rt.layernorm(tok, M, C, self.nw, self.nb, 1e-6, out_h=t, out_f=f, out_group=N)
                else:
                    # the final-normed cls row of every frame, then the readout projection on the normed patch tokens;
                    # with want_f32_last (path A) the last tap goes through the memory block first: its readout is the caller's
                    last = want_f32_last and j == len(self.taps) - 1
                    raw = t if last else rt.hbuf("tap_raw", (Bf * P, C))
                    rt.layernorm(tok, M, C, self.nw, self.nb, 1e-6, out_h=raw, out_f=f, out_group=N)
                    cls = rt.fbuf("cls_norm_last" if last else "cls_norm", (Bf, C))
                    rt.layernorm(tok.view(Bf, N, C)[:, 0].contiguous(), Bf, C, self.nw, self.nb, 1e-6, out_f=cls)
                    if last:
                        self.cls_last = cls
                    else:
                        readout.apply(j, raw, cls, Bf, P, t)
                outs.append(t)
        return outs, last_f32, (ph, pw)


# =============================================================================================
class ReadoutEngine:
    """use_clstoken (dpt.py:81-88,119-123): tap' = GELU(Linear_2C->C(cat(tap, cls expanded))). The cls half of the weight
    acts on ONE row per frame, so it becomes a per-frame bias: c_b = W[:, C:] cls_b + bias, tap'_b = GELU(tap_b W[:, :C]^T + c_b)
    — one small GEMM for the biases and one GEMM per frame. No shipped configuration enables the flag; this path is
    about the constructor contract, not speed."""

    def __init__(self, rt: Runtime, mods, C: int):
        self.rt, self.C = rt, C
        h = rt.prec
        self.w1 = [pack.linear(m[0].weight[:, :C].contiguous(), h) for m in mods]
        self.w2 = [pack.linear(m[0].weight[:, C:].contiguous(), h) for m in mods]
        self.b = [pack.f32(m[0].bias) for m in mods]

    def apply(self, j: int, x, cls_f32: torch.Tensor, Bf: int, P: int, out):
        rt, C = self.rt, self.C
        cb = rt.fbuf("ro_cb", (Bf, C))
        rt.gemm(rt.to_half(cls_f32), self.w2[j], Bf, C, C, bias=self.b[j], out=cb)
        for b in range(Bf):
            rt.gemm(x.narrow0(b * P, P), self.w1[j], P, C, C, bias=cb[b], act=GELU, out=out.narrow0(b * P, P))
        return out


# =============================================================================================
class TemporalEngine:
    """TemporalModule: video_depth_anything/motion_module/motion_module.py:102-136,174-192,245-326,
    attention.py:182-211 (softmax attention over frames), :296-384 (GEGLU feed-forward)."""

    def __init__(self, rt: Runtime, mod, c: int, idx: int):
        self.rt, self.c, self.idx = rt, c, idx
        h = rt.prec
        tt = mod.temporal_transformer
        blk = tt.transformer_blocks[0]
        self.gnw, self.gnb = pack.f32(tt.norm.weight), pack.f32(tt.norm.bias)
        self.w_in, self.b_in = pack.linear(tt.proj_in.weight, h), pack.f32(tt.proj_in.bias)
        self.att = []
        for i in range(2):
            a = blk.attention_blocks[i]
            wqkv, _ = pack.cat_proj([a.to_q.weight, a.to_k.weight, a.to_v.weight], [None, None, None], [0, 0, 0], h)
            rope = not hasattr(a, "pos_encoder")   # pe = 'rope' (motion_module.py:236-240): q / k rotated by frame index, no additive term
            if rope:
                Tm = a.max_len
                pe64 = torch.zeros((Tm, c), dtype=torch.float64, device=a.to_q.weight.device)
                fr = 1.0 / (10000.0 ** (torch.arange(0, c, 2, dtype=torch.float32)[: c // 2] / c))     # attention.py:403-408
                ang = torch.outer(torch.arange(Tm, dtype=torch.float32), fr)
                rope_cs = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(a.to_q.weight.device)   # [T, c/2, 2]
            else:
                pe64 = a.pos_encoder.pe.detach().double().reshape(-1, c)
                rope_cs = None
            self.att.append(dict(
                nw=pack.f32(blk.norms[i].weight), nb=pack.f32(blk.norms[i].bias), wqkv=wqkv,
                pe=None if rope else pack.f32(a.pos_encoder.pe).reshape(-1, c), rope_cs=rope_cs, max_len=pe64.shape[0],
                # (streaming with 'rope': the reference rotates by freqs_cis[:1] = position 0 = the identity
                # (motion_module.py:279-282 with a one-frame query), i.e. no position enters: the zero tables below)
                # streaming mode: W (x + pe[t]) = W x + W pe[t] -> the position term of each projection as a [T, c] table
                pe_q=(pe64 @ a.to_q.weight.detach().double().t()).float().contiguous(),
                pe_k=(pe64 @ a.to_k.weight.detach().double().t()).float().contiguous(),
                pe_v=(pe64 @ a.to_v.weight.detach().double().t()).float().contiguous(),
                wo=pack.linear(a.to_out[0].weight, h), bo=pack.f32(a.to_out[0].bias)))
        self.fnw, self.fnb = pack.f32(blk.ff_norm.weight), pack.f32(blk.ff_norm.bias)
        self.wg, self.bg = pack.geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, h)
        # the LayerNorm-fed linears (q|k|v of both attention blocks, the gated feed-forward's first layer: 14 of the module's
        # 22 c^2 products per row) on the cross-term kernel when a window brings >= 4096 rows; the others take their input from
        # kernels that write no 6-bit rows (GroupNorm, the attention over frames, the gated epilogue)
        self.x8 = rt.split and rt.half == torch.float16 and c % 64 == 0 and os.environ.get("VDN_X8", "1") != "0"
        if self.x8:
            for at in self.att:
                at["x8"] = pack.X8(at["wqkv"], pack.ORDER_NATURAL)
            self.xg = pack.X8(self.wg, pack.ORDER_NATURAL)
        self.wf2, self.bf2 = pack.linear(blk.ff.net[2].weight, h), pack.f32(blk.ff.net[2].bias)
        self.w_out, self.b_out = pack.linear(tt.proj_out.weight, h), pack.f32(tt.proj_out.bias)

    def gn(self, x, F: int, HW: int):
        """GroupNorm(32, eps 1e-6) per frame (motion_module.py:112): needs all pixels of a frame."""
        g = self.rt.hbuf("tm_gn", (F * HW, self.c))
        self.rt.groupnorm(x, g, F, HW, self.c, 32, self.gnw, self.gnb, 1e-6)
        return g

    def core(self, g, B: int, T: int, D: int):
        """proj_in -> 2 x (LN + PE, q/k/v, attention over frames, out) -> GEGLU FF (motion_module.py:116-192).
        Everything here is per pixel, so `D` may be any subset of a frame's pixels (vdn/dist.py)."""
        rt, c = self.rt, self.c
        M = B * T * D
        assert T <= self.att[0]["max_len"], "clip longer than temporal_max_len (motion_module.py:200-213)"
        hs = rt.fbuf("tm_h", (M, c))
        rt.gemm(g, self.w_in, M, c, c, bias=self.b_in, out=hs)
        # (the kernel takes launches of >= 2^20 outputs: narrow modules of small windows stay on the three-product kernels)
        use8 = self.x8 and M >= int(os.environ.get("VDN_X8_MIN_ROWS", "4096")) and M * 3 * c >= (1 << 20)
        if use8:
            from .runtime import HL
            n_k, n8 = HL(rt.buf("tm_n_kt", (M, c), rt.half)), rt.buf("tm_n8", (2, M, c), torch.uint8)
            kt = dict(a8=n8, a_kt=True, w_kt=True)
        else:
            n = rt.hbuf("tm_n", (M, c))
        qkv = rt.hbuf("tm_qkv", (M, 3 * c))
        a = rt.hbuf("tm_a", (M, c))
        for at in self.att:
            if use8:
                rt.layernorm(hs, M, c, at["nw"], at["nb"], 1e-5, out_h=n_k, out8=n8, kt=True, addtab=at["pe"], tab_div=D, tab_mod=T)
                rt.gemm(n_k, HL(at["x8"].hi), M, 3 * c, c, out=qkv, w8=at["x8"].p8, **kt)
            else:
                rt.layernorm(hs, M, c, at["nw"], at["nb"], 1e-5, out_h=n, addtab=at["pe"], tab_div=D, tab_mod=T)
                rt.gemm(n, at["wqkv"], M, 3 * c, c, out=qkv)
            rt.temporal_attn(qkv, a, B, T, D, c, 8, (c // 8) ** -0.5, rope_cs=at["rope_cs"])
            rt.gemm(a, at["wo"], M, c, c, bias=at["bo"], res1=hs, out=hs)
        gg = rt.hbuf("tm_gg", (M, 4 * c))
        if use8:
            rt.layernorm(hs, M, c, self.fnw, self.fnb, 1e-5, out_h=n_k, out8=n8, kt=True)
            rt.gemm(n_k, HL(self.xg.hi), M, 8 * c, c, bias=self.bg, store=abi.ST_GEGLU, out=gg, w8=self.xg.p8, **kt)
        else:
            rt.layernorm(hs, M, c, self.fnw, self.fnb, 1e-5, out_h=n)
            rt.gemm(n, self.wg, M, 8 * c, c, bias=self.bg, store=abi.ST_GEGLU, out=gg)
        hh = rt.hbuf("tm_hh", (M, c))
        rt.gemm(gg, self.wf2, M, c, 4 * c, bias=self.bf2, res1=hs, out=hh)
        return hh

    def out(self, hh, x, M: int):
        """proj_out + residual with the module input (motion_module.py:131-135)."""
        y = self.rt.hbuf(f"tm_out{self.idx}", (M, self.c))
        self.rt.gemm(hh, self.w_out, M, self.c, self.c, bias=self.b_out, res1=x, out=y)
        return y

    def run(self, x, B: int, T: int, HW: int):
        """x half [(b f) * HW, c] (NHWC frames) -> same shape."""
        return self.out(self.core(self.gn(x, B * T, HW), B, T, HW), x, B * T * HW)

    STREAM_SLOTS = 44  # the reference keeps at most 42 cached frames (video_depth_stream.py:155-158) + the new one

    def run_stream(self, x, HW: int, window_slots, new_slot: int):
        """Streaming step (video_depth_stream.py:76-160, motion_module.py:255-277): x is ONE frame [HW, c].
        `window_slots`: ring slots of the (up to 31) cached frames this step attends over, oldest first; the new frame's
        projections go to ring slot `new_slot` and it attends last.
        A slot holds the frame's q|k|v PROJECTION without the position term (f32 [HW, 3c]): the
        reference caches the LayerNorm output and re-projects all 32 frames with the sliding window's positions
        every step; here each frame is projected once and the positions enter as [T, c] tables inside
        `vdn_temporal_attn_last` (W(x + pe) = Wx + W pe), which also computes the newest frame's query only.
        The two attention blocks own one fixed ring each (nothing is allocated per step, no pointer table)."""
        rt, c = self.rt, self.c
        g = self.gn(x, 1, HW)
        hs = rt.fbuf("ts_h", (HW, c))
        rt.gemm(g, self.w_in, HW, c, c, bias=self.b_in, out=hs)
        nh = rt.hbuf("ts_nh", (HW, c))
        a = rt.hbuf("ts_a", (HW, c))
        slots = list(window_slots) + [new_slot]
        for j, at in enumerate(self.att):
            pool = rt.fbuf(f"ts_pool{self.idx}_{j}", (self.STREAM_SLOTS, HW, 3 * c))
            rt.layernorm(hs, HW, c, at["nw"], at["nb"], 1e-5, out_h=nh)
            rt.gemm(nh, at["wqkv"], HW, 3 * c, c, out=pool[new_slot])
            rt.temporal_attn_last(pool, slots, at["pe_q"], at["pe_k"], at["pe_v"], a, HW, c, (c // 8) ** -0.5)
            rt.gemm(a, at["wo"], HW, c, c, bias=at["bo"], res1=hs, out=hs)
        n = rt.hbuf("ts_n", (HW, c))
        rt.layernorm(hs, HW, c, self.fnw, self.fnb, 1e-5, out_h=n)
        gg = rt.hbuf("ts_gg", (HW, 4 * c))
        rt.gemm(n, self.wg, HW, 8 * c, c, bias=self.bg, store=abi.ST_GEGLU, out=gg)
        hh = rt.hbuf("ts_hh", (HW, c))
        rt.gemm(gg, self.wf2, HW, c, 4 * c, bias=self.bf2, res1=hs, out=hh)
        return self.out(hh, x, HW)

    def run_sharded(self, x, exch, HW: int):
        """Frame-sharded window (vdn/dist.py): x holds this rank's Tl frames. GroupNorm and proj_out run
        on the frame shard; the per-pixel core runs on all T frames of this rank's pixel shard, with one
        all-to-all before and one after per plane (staging: vdn.dist.shard_core, a pure function of tensors
        that tests/test_dist.py drives under gloo). Buffers are consumed in stream order, so plain local
        references keep them alive long enough."""
        from .dist import shard_core
        from .runtime import HL
        Tl = exch.Tl
        g = self.gn(x, Tl, HW)

        def core(planes, D):
            hh = self.core(HL(planes[0], planes[1] if len(planes) > 1 else None), 1, exch.T, D)
            return [hh.hi] + ([hh.lo] if hh.lo is not None else [])

        back = shard_core(exch, [g.hi] + ([g.lo] if g.lo is not None else []), HW, core)
        return self.out(HL(back[0], back[1] if len(back) > 1 else None), x, Tl * HW)


# =============================================================================================
class DPTEngine:
    """DPTHead / DPTHeadTemporal: depth_anything_v2/dpt.py:116-151, util/blocks.py:57-148,
    video_depth_anything/dpt_temporal.py:53-127."""

    def __init__(self, rt: Runtime, mod, in_ch: int, features: int, out_channels, temporal: bool):
        self.rt, self.C, self.F, self.oc = rt, in_ch, features, list(out_channels)
        h = rt.prec
        self.proj = [(pack.conv1x1(p.weight, h), pack.f32(p.bias)) for p in mod.projects]
        self.rt0 = pack.conv_transpose(mod.resize_layers[0].weight, mod.resize_layers[0].bias, h)
        self.rt1 = pack.conv_transpose(mod.resize_layers[1].weight, mod.resize_layers[1].bias, h)
        self.rs3 = (pack.conv3x3(mod.resize_layers[3].weight, h), pack.f32(mod.resize_layers[3].bias))
        s = mod.scratch
        self.rn = [pack.conv3x3(getattr(s, f"layer{i + 1}_rn").weight, h) for i in range(4)]

        def fold(conv, bn):
            """use_bn (util/blocks.py:49-51,71-77): eval-mode BatchNorm is a per-channel affine map, folded into the conv"""
            w, b = conv.weight.detach().float(), conv.bias.detach().float()
            if bn is None:
                return w, b
            sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + 1e-5)
            return w * sc[:, None, None, None], (b - bn.running_mean.detach().float()) * sc + bn.bias.detach().float()

        def rcu(r):
            w1, b1 = fold(r.conv1, getattr(r, "bn1", None))
            w2, b2 = fold(r.conv2, getattr(r, "bn2", None))
            return dict(w1=pack.conv3x3(w1, h), b1=pack.f32(b1), w2=pack.conv3x3(w2, h), b2=pack.f32(b2))

        self.ref = {}
        for i in range(1, 5):
            f = getattr(s, f"refinenet{i}")
            self.ref[i] = dict(r1=rcu(f.resConfUnit1), r2=rcu(f.resConfUnit2),
                               wo=pack.conv1x1(f.out_conv.weight, h), bo=pack.f32(f.out_conv.bias))
        self.oc1 = (pack.conv3x3(s.output_conv1.weight, h), pack.f32(s.output_conv1.bias))
        self.oc2 = (pack.conv3x3(s.output_conv2[0].weight, h), pack.f32(s.output_conv2[0].bias))
        # fused tail (csrc/tail.hip): needs split planes, 32 output channels and 32-channel input blocks
        w2 = s.output_conv2[0].weight
        fused = h.split and w2.shape[0] == 32 and w2.shape[1] % 32 == 0 and not os.environ.get("VDN_TAIL_UNFUSED")  # A/B switch
        self.oc2_taps = pack.conv3x3_taps(w2, h) if fused else None
        self.w_last = pack.f32(s.output_conv2[2].weight).reshape(-1)
        self.b_last = float(s.output_conv2[2].bias.detach().float().item())
        self.temporal = None
        if temporal:
            chans = [self.oc[2], self.oc[3], features, features]
            self.temporal = [TemporalEngine(rt, mod.motion_modules[i], chans[i], i) for i in range(4)]

    # -- helpers
    def _conv3(self, x, w, Bf, H, W, Cin, Cout, name, *, stride=1, bias=None, relu_a=False, act=0, res1=None, res2=None,
               f32_out=False):
        OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = (self.rt.fbuf if f32_out else self.rt.hbuf)(name, (Bf * OH * OW, Cout))
        self.rt.gemm(x, w, Bf * OH * OW, Cout, 9 * Cin, out=out, bias=bias, act=act, relu_a=relu_a, res1=res1, res2=res2,
                     conv=dict(B=Bf, H=H, W=W, C=Cin, OH=OH, OW=OW, stride=stride))
        return out

    def _rcu(self, r, x, Bf, H, W, name, extra=None):
        """ResidualConvUnit (blocks.py:67-80): conv2(relu(conv1(relu(x)))) + x [+ extra]."""
        F = self.F
        t = self._conv3(x, r["w1"], Bf, H, W, F, F, "rcu_t", bias=r["b1"], relu_a=True, act=RELU)
        return self._conv3(t, r["w2"], Bf, H, W, F, F, name, bias=r["b2"], res1=x, res2=extra)

    def _fusion(self, i, Bf, size_in, size_out, prev, skip=None):
        """FeatureFusionBlock (blocks.py:123-148). The 1x1 out_conv commutes with the bilinear resize
        (both linear, interpolation weights sum to 1), so it runs at the LOW resolution: 4x fewer
        flops, same result up to rounding."""
        rt, F = self.rt, self.F
        H, W = size_in
        f = self.ref[i]
        x = prev
        if skip is not None:
            x = self._rcu(f["r1"], skip, Bf, H, W, f"ff{i}_s", extra=prev)
        u = self._rcu(f["r2"], x, Bf, H, W, f"ff{i}_u")
        v = rt.hbuf(f"ff{i}_v", (Bf * H * W, F))
        rt.gemm(u, f["wo"], Bf * H * W, F, F, bias=f["bo"], out=v)
        OH, OW = size_out
        p = rt.hbuf(f"path{i}", (Bf * OH * OW, F))
        rt.upsample(v, p, Bf, H, W, OH, OW, F)
        return p

    def run(self, taps: List[torch.Tensor], Bf: int, ph: int, pw: int, T: Optional[int] = None, relu: bool = True,
            exch=None, stream=None, out: Optional[torch.Tensor] = None):
        """`out` (f32 [Bf, 14 ph, 14 pw], contiguous): the caller's result tensor, written by the last kernel directly;
        without it the depth lands in this engine's arena (valid until the next run)."""
        rt, C, F, oc = self.rt, self.C, self.F, self.oc
        P = ph * pw
        pr = []
        for i in range(4):
            o = rt.hbuf(f"proj{i}", (Bf * P, oc[i]))
            rt.gemm(taps[i], self.proj[i][0], Bf * P, oc[i], C, bias=self.proj[i][1], out=o)
            pr.append(o)
        s1, s2, s3 = (4 * ph, 4 * pw), (2 * ph, 2 * pw), (ph, pw)
        s4 = ((ph + 2 - 3) // 2 + 1, (pw + 2 - 3) // 2 + 1)
        l1 = rt.hbuf("l1", (Bf * s1[0] * s1[1], oc[0]))
        rt.gemm(pr[0], self.rt0[0], Bf * P, 16 * oc[0], oc[0], bias=self.rt0[1], store=abi.ST_CONVT, out=l1,
                convt=dict(k=4, cout=oc[0], B=Bf, H=ph, W=pw))
        l2 = rt.hbuf("l2", (Bf * s2[0] * s2[1], oc[1]))
        rt.gemm(pr[1], self.rt1[0], Bf * P, 4 * oc[1], oc[1], bias=self.rt1[1], store=abi.ST_CONVT, out=l2,
                convt=dict(k=2, cout=oc[1], B=Bf, H=ph, W=pw))
        l3 = pr[2]
        l4 = self._conv3(pr[3], self.rs3[0], Bf, ph, pw, oc[3], oc[3], "l4", stride=2, bias=self.rs3[1])
        def tm(i, x, hw):
            if stream is not None:  # streaming: one new frame against the cached projections of up to 31 earlier ones
                return self.temporal[i].run_stream(x, hw, stream["window"], stream["new"])
            if exch is not None:  # frame-sharded window: Bf == this rank's frames of ONE clip
                return self.temporal[i].run_sharded(x, exch, hw)
            return self.temporal[i].run(x, Bf // T, T, hw)

        if self.temporal is not None:
            l3 = tm(0, l3, s3[0] * s3[1])
            l4 = tm(1, l4, s4[0] * s4[1])
        r1 = self._conv3(l1, self.rn[0], Bf, s1[0], s1[1], oc[0], F, "l1_rn")
        r2 = self._conv3(l2, self.rn[1], Bf, s2[0], s2[1], oc[1], F, "l2_rn")
        r3 = self._conv3(l3, self.rn[2], Bf, s3[0], s3[1], oc[2], F, "l3_rn")
        r4 = self._conv3(l4, self.rn[3], Bf, s4[0], s4[1], oc[3], F, "l4_rn")
        p4 = self._fusion(4, Bf, s4, s3, r4)
        if self.temporal is not None:
            p4 = tm(2, p4, s3[0] * s3[1])
        p3 = self._fusion(3, Bf, s3, s2, p4, r3)
        if self.temporal is not None:
            p3 = tm(3, p3, s2[0] * s2[1])
        p2 = self._fusion(2, Bf, s2, s1, p3, r2)
        s0 = (2 * s1[0], 2 * s1[1])
        p1 = self._fusion(1, Bf, s1, s0, p2, r1)
        H, W = ph * PATCH, pw * PATCH
        if out is not None:
            assert out.shape == (Bf, H, W) and out.dtype == torch.float32 and out.is_contiguous(), (out.shape, out.dtype)
        depth = out if out is not None else rt.fbuf("depth", (Bf, H, W))
        sy, sx = (s0[0] - 1) / max(H - 1, 1), (s0[1] - 1) / max(W - 1, 1)
        if self.oc2_taps is not None and int(17 * sy) + 3 <= 13 and int(17 * sx) + 3 <= 13:
            # resize -> conv3x3 + ReLU -> conv1x1 (+ ReLU) without leaving the chip (the scale is 8/14 for every DPT head);
            # output_conv1 then writes ONE fp32 plane, which is what the fused kernel interpolates from
            o1 = self._conv3(p1, self.oc1[0], Bf, s0[0], s0[1], F, F // 2, "out1_f32", bias=self.oc1[1], f32_out=True)
            rt.depth_tail(o1, self.oc2_taps, self.oc2[1], self.w_last, self.b_last, depth, Bf, s0[0], s0[1], F // 2, H, W, relu)
            return depth
        o1 = self._conv3(p1, self.oc1[0], Bf, s0[0], s0[1], F, F // 2, "out1", bias=self.oc1[1])
        up = rt.hbuf("out_up", (Bf * H * W, F // 2))
        rt.upsample(o1, up, Bf, s0[0], s0[1], H, W, F // 2)
        o2 = self._conv3(up, self.oc2[0], Bf, H, W, F // 2, 32, "out2", bias=self.oc2[1], act=RELU)
        rt.head_out(o2, self.w_last, self.b_last, depth, Bf * H * W, 32, relu)
        return depth


# =============================================================================================
class MemoryEngine:
    """MemoryBlock: depth_anything_v2/memory_block.py:83-125 over sam2/modeling/{memory_attention.py:58-169,
    sam/transformer.py:275-311, position_encoding.py:192-239, memory_encoder.py:158-181}.

    MI355X-first differences that leave the result unchanged:
      * the reference re-projects all S stored frames through k_proj/v_proj of every layer on every
        call; here each pushed frame's rotated K and V^T are written once into a per-layer 6-slot
        ring in HBM (keys are order-invariant under softmax, RoPE depends on t % P only), so the
        cross-attention reads them in place;
      * memory_pos_enc / maskmem_tpos_enc never reach the output under the flags MemoryBlock sets
        (pos_enc_at_cross_attn_keys=False, memory_block.py:39), so they are not computed.
    """

    def __init__(self, rt: Runtime, mod, C: int, max_len: int):
        self.rt, self.C, self.heads, self.max_len = rt, C, C // 64, max_len
        h = rt.prec
        ma = mod.memory_attention
        self.layers = []
        for l in ma.layers:
            sa, ca = l.self_attn, l.cross_attn_image
            wqkv, bqkv = pack.cat_proj([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight],
                                       [sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias], [1, 1, 0], h)
            wq, bq = pack.cat_proj([ca.q_proj.weight], [ca.q_proj.bias], [1], h)
            wkv, bkv = pack.cat_proj([ca.k_proj.weight, ca.v_proj.weight], [ca.k_proj.bias, ca.v_proj.bias], [1, 0], h)
            self.layers.append(dict(
                n1w=pack.f32(l.norm1.weight), n1b=pack.f32(l.norm1.bias),
                n2w=pack.f32(l.norm2.weight), n2b=pack.f32(l.norm2.bias),
                n3w=pack.f32(l.norm3.weight), n3b=pack.f32(l.norm3.bias),
                wqkv=wqkv, bqkv=bqkv, wso=pack.linear(sa.out_proj.weight, h), bso=pack.f32(sa.out_proj.bias),
                wq=wq, bq=bq, wkv=wkv, bkv=bkv, wco=pack.linear(ca.out_proj.weight, h), bco=pack.f32(ca.out_proj.bias),
                w1=pack.linear(l.linear1.weight, h), b1=pack.f32(l.linear1.bias),
                w2=pack.linear(l.linear2.weight, h), b2=pack.f32(l.linear2.bias)))
        self.nw, self.nb = pack.f32(ma.norm.weight), pack.f32(ma.norm.bias)
        self.curr_pos = pack.f32(mod.curr_pos_enc).reshape(-1)
        self.no_mem = mod.no_mem_embed.detach().float().reshape(1, -1)
        me = mod.memory_encoder

        def flat(*ts):
            return torch.cat([t.detach().float().reshape(-1) for t in ts]).contiguous()

        e0, e1 = me.mask_downsampler[0].encoder, me.mask_downsampler[1].encoder
        self.md1 = flat(e0[0].weight, e0[0].bias, e0[1].weight, e0[1].bias, e0[3].weight, e0[3].bias)
        self.md2 = flat(e1[0].weight, e1[0].bias, e1[1].weight, e1[1].bias, e1[3].weight, e1[3].bias)
        assert self.md1.numel() == 53 and self.md2.numel() == 2598
        self.wpix, self.bpix = pack.conv1x1(me.pix_feat_proj.weight, h), pack.f32(me.pix_feat_proj.bias)
        self.cx = []
        for b in me.fuser.layers:
            self.cx.append(dict(
                wdw=b.dwconv.weight.detach().float().reshape(C, 49).t().contiguous(), bdw=pack.f32(b.dwconv.bias),
                nw=pack.f32(b.norm.weight), nb=pack.f32(b.norm.bias),
                w1=pack.linear(b.pwconv1.weight, h), b1=pack.f32(b.pwconv1.bias),
                w2=pack.linear(b.pwconv2.weight, h), b2=pack.f32(b.pwconv2.bias), g=pack.f32(b.gamma)))
        # 8-bit cross terms for the plain linears of the memory attention and the memory encoder (as in EncoderEngine)
        self.x8 = rt.split and rt.half == torch.float16 and C % 64 == 0 and os.environ.get("VDN_X8", "1") != "0"
        if self.x8:
            for L in self.layers:
                L["x8"] = {k: pack.X8(L[k], o) for k, o in (("wqkv", pack.ORDER_NATURAL), ("wso", pack.ORDER_ATTN), ("wq", pack.ORDER_NATURAL),
                                                            ("wco", pack.ORDER_ATTN), ("w1", pack.ORDER_NATURAL), ("w2", pack.ORDER_GEMM),
                                                            ("wkv", pack.ORDER_NATURAL))}
            for cx in self.cx:
                cx["x8"] = {"w1": pack.X8(cx["w1"], pack.ORDER_NATURAL), "w2": pack.X8(cx["w2"], pack.ORDER_GEMM)}
        # Bank state shared by every lane copy of this engine (DepthAnythingV2._stream_lanes): ONE ring for the whole
        # batch, lane i of n works on batch rows [i B/n, (i+1) B/n) of it, so laned and single-lane calls see the
        # same memory and `count` advances once per forward (commit()).
        self.state = {"count": 0, "shape": None}
        self.lane = (0, 1)
        self.bank_rt = rt
        self._rope = {}
        self._nomem = {}

    def clear(self):
        self.state["count"] = 0
        self.state["shape"] = None

    @property
    def S(self):
        return min(self.state["count"], self.max_len)

    def _rope_for(self, side):
        if side not in self._rope:
            self._rope[side] = pack.rope_table(side, side, 64, device=self.rt.device)
        return self._rope[side]

    def prepare(self, B: int, P: int):
        """Called once per forward on the caller's stream BEFORE any lane forks: checks the bank against the batch
        (the reference asserts on a batch mismatch between the frame and its memories, memory_attention.py:135-137)
        and makes sure the ring and the RoPE table exist."""
        st = self.state
        if st["shape"] != (B, P):
            if st["count"] > 0:
                raise RuntimeError(f"memory bank holds {self.S} frame(s) of batch/grid {st['shape']} but the new frame is "
                                   f"{(B, P)}: call clear_memory() before changing the batch size or resolution")
            st["shape"] = (B, P)
        side = int(math.sqrt(P))
        assert side * side == P, "MemoryBlock assumes square inputs (memory_block.py:85)"
        self._rope_for(side)
        self._bank_full(B, P)

    def commit(self):
        self.state["count"] += 1

    def _bank_full(self, B, P):
        rt, Hh = self.bank_rt, self.heads
        tp = ceil_to(self.max_len * P, 64)
        ks = [rt.hbuf(f"mem_k{l}", (B * Hh, tp, 64), zero=True) for l in range(len(self.layers))]
        vs = [rt.hbuf(f"mem_vt{l}", (B * Hh, 64, tp), zero=True) for l in range(len(self.layers))]
        k8 = [rt.qk8(f"mem_k8{l}", B * Hh, tp) for l in range(len(self.layers))]  # e5m2 planes of the stored keys
        return ks, vs, k8, tp

    def _bank(self, B, P):
        """This lane's rows of the ring: B is the lane's batch."""
        i, n = self.lane
        ks, vs, k8, tp = self._bank_full(B * n, P)
        r0, nr = i * B * self.heads, B * self.heads
        return ([k.narrow0(r0, nr) for k in ks], [v.narrow0(r0, nr) for v in vs],
                [None if t is None else t.narrow(0, r0, nr) for t in k8], tp)

    def forward(self, feat_f32: torch.Tensor, B: int, P: int) -> torch.Tensor:
        """feat_f32 [B*P, C] (final-normed tap 4) -> half [B*P, C] (memory_block.py:92-125)."""
        rt, C, Hh = self.rt, self.C, self.heads
        assert self.state["shape"] == (B * self.lane[1], P), "MemoryEngine.prepare() not called for this batch"
        side = int(math.sqrt(P))
        M = B * P
        cs = self._rope_for(side)
        pp = ceil_to(P, 64)
        x = rt.fbuf("ma_x", (M, C))
        rt.add_vec(feat_f32, self.curr_pos, 0.1, x, M, C)  # memory_attention.py:140-141
        q = rt.hbuf("ma_q", (B * Hh, pp, 64), zero=True)
        k = rt.hbuf("ma_k", (B * Hh, pp, 64), zero=True)
        vt = rt.hbuf("ma_vt", (B * Hh, 64, pp), zero=True)
        q8, k8 = rt.qk8("ma_q8", B * Hh, pp), rt.qk8("ma_k8", B * Hh, pp)
        ks, vs, k8s, tp = self._bank(B, P)
        S = self.S
        if S == 0:
            # empty bank: keys/values come from no_mem_embed broadcast to P tokens (memory_block.py:115-123)
            nk, nk_pad = P, pp
            if M not in self._nomem:  # built once per (lane, batch): the broadcast no_mem_embed rows as half planes
                self._nomem[M] = rt.to_half(self.no_mem.expand(M, C).contiguous())
            a_nm = self._nomem[M]
            ks = [rt.hbuf(f"nomem_k{l}", (B * Hh, pp, 64), zero=True) for l in range(len(self.layers))]
            vs = [rt.hbuf(f"nomem_vt{l}", (B * Hh, 64, pp), zero=True) for l in range(len(self.layers))]
            k8s = [rt.qk8(f"nomem_k8{l}", B * Hh, pp) for l in range(len(self.layers))]
            for l, L in enumerate(self.layers):
                rt.gemm(a_nm, L["wkv"], M, 2 * C, C, bias=L["bkv"], store=abi.ST_HEADS,
                        heads=dict(dst=[rt.qk_dst(ks[l], k8s[l]), rt.v_dst(vs[l])], dst8=[k8s[l], None], transposed=[0, 1], rope=[1, 0], rope_cs=cs,
                                   rope_mod=P, heads=Hh, tokens=P, tpad=pp))
        else:
            nk, nk_pad = S * P, tp
        sh = dict(dst=[rt.qk_dst(q, q8), rt.qk_dst(k, k8), rt.v_dst(vt)], dst8=[q8, k8, None], transposed=[0, 0, 1], rope=[1, 1, 0], rope_cs=cs, rope_mod=P, heads=Hh,
                  tokens=P, tpad=pp)
        qh = dict(dst=[rt.qk_dst(q, q8)], dst8=[q8], transposed=[0], rope=[1], rope_cs=cs, rope_mod=P, heads=Hh, tokens=P, tpad=pp)
        use8 = self.x8 and M >= int(os.environ.get("VDN_X8_MIN_ROWS", "4096")) and q8 is not None and rt.pv_products != 3
        if use8:   # K-tile-major fp16 hi + planes of 6-bit rows between the linears (no fp16 lo plane), as in EncoderEngine.run
            from .runtime import HL
            n_k, n8 = HL(rt.buf("ma_n_kt", (M, C), rt.half)), rt.buf("ma_n8", (2, M, C), torch.uint8)
            att_k, att8 = HL(rt.buf("ma_att_kt", (M, C), rt.half)), rt.buf("ma_att8", (2, M, C), torch.uint8)
            h2_k, h28 = HL(rt.buf("ma_h2_kt", (M, 2 * C), rt.half)), rt.buf("ma_h28", (2, M, 2 * C), torch.uint8)
            kt = dict(a_kt=True, w_kt=True)
            for l, L in enumerate(self.layers):
                x8 = L["x8"]
                rt.layernorm(x, M, C, L["n1w"], L["n1b"], 1e-5, out_h=n_k, out8=n8, kt=True)
                rt.gemm(n_k, HL(x8["wqkv"].hi), M, 3 * C, C, bias=L["bqkv"], store=abi.ST_HEADS, heads=sh, a8=n8, w8=x8["wqkv"].p8, **kt)
                rt.flash_attn(q, k, vt, att_k, B, Hh, P, pp, P, pp, 0.125, q8=q8, k8=k8, out8=att8, out_kt=True)
                rt.gemm(att_k, HL(x8["wso"].hi), M, C, C, bias=L["bso"], res1=x, out=x, a8=att8, w8=x8["wso"].p8, **kt)
                rt.layernorm(x, M, C, L["n2w"], L["n2b"], 1e-5, out_h=n_k, out8=n8, kt=True, addvec=self.curr_pos, alpha=1.0)
                rt.gemm(n_k, HL(x8["wq"].hi), M, C, C, bias=L["bq"], store=abi.ST_HEADS, heads=qh, a8=n8, w8=x8["wq"].p8, **kt)
                rt.flash_attn(q, ks[l], vs[l], att_k, B, Hh, P, pp, nk, nk_pad, 0.125, q8=q8, k8=k8s[l], out8=att8, out_kt=True)
                rt.gemm(att_k, HL(x8["wco"].hi), M, C, C, bias=L["bco"], res1=x, out=x, a8=att8, w8=x8["wco"].p8, **kt)
                rt.layernorm(x, M, C, L["n3w"], L["n3b"], 1e-5, out_h=n_k, out8=n8, kt=True)
                rt.gemm(n_k, HL(x8["w1"].hi), M, 2 * C, C, bias=L["b1"], act=GELU, out=h2_k, out8=h28, out_kt=True, a8=n8, w8=x8["w1"].p8, **kt)
                rt.gemm(h2_k, HL(x8["w2"].hi), M, C, 2 * C, bias=L["b2"], res1=x, out=x, a8=h28, w8=x8["w2"].p8, **kt)
            out = rt.hbuf("mem_out", (M, C))
            rt.layernorm(x, M, C, self.nw, self.nb, 1e-5, out_h=out)
            return out
        n, att, h2 = rt.hbuf("ma_n", (M, C)), rt.hbuf("ma_att", (M, C)), rt.hbuf("ma_h2", (M, 2 * C))   # split planes of the 3-product path
        for l, L in enumerate(self.layers):
            rt.layernorm(x, M, C, L["n1w"], L["n1b"], 1e-5, out_h=n)
            rt.gemm(n, L["wqkv"], M, 3 * C, C, bias=L["bqkv"], store=abi.ST_HEADS, heads=sh)
            rt.flash_attn(q, k, vt, att, B, Hh, P, pp, P, pp, 0.125, q8=q8, k8=k8)
            rt.gemm(att, L["wso"], M, C, C, bias=L["bso"], res1=x, out=x)
            rt.layernorm(x, M, C, L["n2w"], L["n2b"], 1e-5, out_h=n, addvec=self.curr_pos, alpha=1.0)
            rt.gemm(n, L["wq"], M, C, C, bias=L["bq"], store=abi.ST_HEADS, heads=qh)
            rt.flash_attn(q, ks[l], vs[l], att, B, Hh, P, pp, nk, nk_pad, 0.125, q8=q8, k8=k8s[l])
            rt.gemm(att, L["wco"], M, C, C, bias=L["bco"], res1=x, out=x)
            rt.layernorm(x, M, C, L["n3w"], L["n3b"], 1e-5, out_h=n)
            rt.gemm(n, L["w1"], M, 2 * C, C, bias=L["b1"], act=GELU, out=h2)
            rt.gemm(h2, L["w2"], M, C, 2 * C, bias=L["b2"], res1=x, out=x)
        out = rt.hbuf("mem_out", (M, C))
        rt.layernorm(x, M, C, self.nw, self.nb, 1e-5, out_h=out)
        return out

    def update(self, mem_out: torch.Tensor, depth: torch.Tensor, B: int, ph: int, pw: int):
        """update_memory (memory_block.py:83-90): MemoryEncoder then push; here the push writes the
        frame's per-layer rotated K / V^T into ring slot count % max_len."""
        rt, C, Hh = self.rt, self.C, self.heads
        P, M = ph * pw, B * ph * pw
        H, W = depth.shape[-2], depth.shape[-1]
        h1, w1 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        h2, w2 = (h1 - 7) // 7 + 1, (w1 - 7) // 7 + 1
        assert (h2, w2) == (ph, pw), "mask downsampler output must match the feature grid (memory_encoder.py:173)"
        m1 = rt.fbuf("me_m1", (B, h1, w1))
        rt.mask_down1(depth, m1, B, H, W, h1, w1, self.md1)
        m2 = rt.fbuf("me_m2", (M,))
        rt.mask_down2(m1, m2, B, h1, w1, h2, w2, self.md2)
        x = rt.fbuf("me_x", (M, C))
        rt.gemm(mem_out, self.wpix, M, C, C, bias=self.bpix, rowadd=m2, out=x)
        d = rt.fbuf("me_dw", (M, C))
        use8 = self.x8 and M >= int(os.environ.get("VDN_X8_MIN_ROWS", "4096"))
        if use8:
            from .runtime import HL
            n_k, n8 = HL(rt.buf("me_n_kt", (M, C), rt.half)), rt.buf("me_n8", (2, M, C), torch.uint8)
            h4_k, h48 = HL(rt.buf("me_h4_kt", (M, 4 * C), rt.half)), rt.buf("me_h48", (2, M, 4 * C), torch.uint8)
            feat = None   # the memory feature stays in the fp32 stream x and goes straight to the projections' operand planes
        else:
            n, h4, feat = rt.hbuf("me_n", (M, C)), rt.hbuf("me_h4", (M, 4 * C)), rt.hbuf("mem_feat", (M, C))
        for j, cx in enumerate(self.cx):
            rt.dwconv7(x, d, B, ph, pw, C, cx["wdw"], cx["bdw"])
            if use8:
                x8 = cx["x8"]
                rt.layernorm(d, M, C, cx["nw"], cx["nb"], 1e-6, out_h=n_k, out8=n8, kt=True)
                rt.gemm(n_k, HL(x8["w1"].hi), M, 4 * C, C, bias=cx["b1"], act=GELU, out=h4_k, out8=h48, out_kt=True, a8=n8, w8=x8["w1"].p8,
                        a_kt=True, w_kt=True)
                rt.gemm(h4_k, HL(x8["w2"].hi), M, C, 4 * C, bias=cx["b2"], gamma=cx["g"], res1=x, out=x, a8=h48, w8=x8["w2"].p8, a_kt=True, w_kt=True)
                continue
            rt.layernorm(d, M, C, cx["nw"], cx["nb"], 1e-6, out_h=n)
            rt.gemm(n, cx["w1"], M, 4 * C, C, bias=cx["b1"], act=GELU, out=h4)
            rt.gemm(h4, cx["w2"], M, C, 4 * C, bias=cx["b2"], gamma=cx["g"], res1=x, out=(x if j == 0 else feat))
        ks, vs, k8s, tp = self._bank(B, P)
        slot = self.state["count"] % self.max_len  # commit() advances the count once every lane has pushed
        cs = self._rope_for(int(math.sqrt(P)))
        if use8:
            # the four key / value projections of the pushed frame on the cross-term kernel: one pass turns the fp32 memory feature
            # into their A operand (K-tile-major hi plane + 6-bit rows); its two ConvNeXt blocks both ran in place on x
            feat_k, feat8 = HL(rt.buf("mem_feat_kt", (M, C), rt.half)), rt.buf("mem_feat8", (2, M, C), torch.uint8)
            rt.pack_x8_f32(x, feat_k.hi, feat8)
        for l, L in enumerate(self.layers):
            hd = dict(dst=[rt.qk_dst(ks[l], k8s[l]), rt.v_dst(vs[l])], dst8=[k8s[l], None], transposed=[0, 1], rope=[1, 0], rope_cs=cs, rope_mod=P,
                      heads=Hh, tokens=P, tok_off=slot * P, tpad=tp)
            if use8:
                xw = L["x8"]["wkv"]
                rt.gemm(feat_k, HL(xw.hi), M, 2 * C, C, bias=L["bkv"], store=abi.ST_HEADS, heads=hd, a8=feat8, w8=xw.p8, a_kt=True, w_kt=True)
            else:
                rt.gemm(feat, L["wkv"], M, 2 * C, C, bias=L["bkv"], store=abi.ST_HEADS, heads=hd)
        return feat
