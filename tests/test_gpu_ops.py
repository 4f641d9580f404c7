"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 CPU statement of the same op
on the same (half-rounded) inputs. Shapes are deliberately ragged (tails in M, N, K, keys, frames).
Tolerances: half outputs carry one rounding (2^-11 fp16 / 2^-8 bf16 relative) on top of fp32
accumulation-order noise."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def rt():
    from vdn.runtime import Runtime
    from vdn import _abi
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    assert _abi.lib.vdn_arch_ok() == 1, "not a gfx950 device"
    return Runtime(torch.device("cuda:0"), torch.float16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


def h(t):  # half-rounded copy kept in f32 (what the kernel actually sees)
    return t.half().float()


PV_DEFAULT = 1   # default pv_products of vdn_flash_attn (include/vdn.h); tests that change Runtime.pv_products put it back


def close(got, ref, tol):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    err = (got - ref).norm() / (ref.norm() + 1e-30)
    mx = (got - ref).abs().max() / (ref.abs().max() + 1e-30)
    assert err < tol and mx < 8 * tol, (float(err), float(mx))


def _w(n, k, seed):
    from vdn import pack
    w = rnd(n, k, seed=seed, scale=1 / math.sqrt(k))
    return w, pack.linear(w.to(DEV), torch.float16)


# --------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 192, 200), (1370, 384, 384), (77, 32, 64), (130, 48, 1152), (257, 1024, 640)])
def test_gemm_plain_bias_gelu(rt, M, N, K):
    from vdn import _abi
    a = h(rnd(M, K, seed=1))
    w, wp = _w(N, K, 2)
    b = rnd(N, seed=3)
    ref = F.gelu(a @ h(w).t() + b)
    out = torch.empty(M, N, device=DEV)
    rt.gemm(a.half().to(DEV), wp, M, N, K, out=out, bias=b.to(DEV), act=_abi.ACT_GELU)
    close(out, ref, 2e-5 * math.sqrt(K) / 8 + 1e-5)


def test_gemm_layerscale_residual_inplace_and_half_out(rt):
    M, N, K = 500, 256, 512
    a = h(rnd(M, K, seed=4))
    w, wp = _w(N, K, 5)
    b, g = rnd(N, seed=6), rnd(N, seed=7)
    x = rnd(M, N, seed=8)
    ref = x + (a @ h(w).t() + b) * g
    xd = x.clone().to(DEV)
    rt.gemm(a.half().to(DEV), wp, M, N, K, out=xd, bias=b.to(DEV), gamma=g.to(DEV), res1=xd)
    close(xd, ref, 1e-4)
    r2 = h(rnd(M, N, seed=9))
    oh = torch.empty(M, N, device=DEV, dtype=torch.float16)
    rt.gemm(a.half().to(DEV), wp, M, N, K, out=oh, bias=b.to(DEV), res1=x.to(DEV), res2=r2.half().to(DEV), rowadd=None)
    close(oh, x + r2 + a @ h(w).t() + b, 1e-3)


def test_gemm_patch_embed_row_remap_and_table(rt):
    # tokens land after each image's cls row; pos table row = m % P + 1 (dinov2.py:219-220)
    B, P, C, K = 3, 10, 64, 640
    a = h(rnd(B * P, K, seed=10))
    w, wp = _w(C, K, 11)
    b = rnd(C, seed=12)
    tab = rnd(P + 1, C, seed=13)
    tok = torch.full((B * (P + 1), C), -7.0, device=DEV)
    rt.gemm(a.half().to(DEV), wp, B * P, C, K, out=tok, bias=b.to(DEV), tab=tab.to(DEV), tab_mod=P, tab_off=1,
            row_group=P, row_skip=1)
    ref = torch.full((B, P + 1, C), -7.0)
    ref[:, 1:] = (a @ h(w).t() + b).reshape(B, P, C) + tab[1:]
    close(tok.reshape(B, P + 1, C), ref, 1e-4)


def test_gemm_rowadd(rt):
    M, N, K = 200, 128, 128
    a = h(rnd(M, K, seed=14))
    w, wp = _w(N, K, 15)
    ra = rnd(M, seed=16)
    out = torch.empty(M, N, device=DEV)
    rt.gemm(a.half().to(DEV), wp, M, N, K, out=out, rowadd=ra.to(DEV))
    close(out, a @ h(w).t() + ra[:, None], 1e-4)


@pytest.mark.parametrize("B,H,W,Ci,Co,stride,relu_a", [(2, 13, 11, 48, 64, 1, True), (1, 37, 37, 64, 256, 2, False),
                                                      (2, 9, 20, 256, 32, 1, False), (1, 6, 5, 96, 48, 1, True)])
def test_gemm_conv3x3(rt, B, H, W, Ci, Co, stride, relu_a):
    from vdn import pack, _abi
    x = h(rnd(B, H, W, Ci, seed=20))
    w = rnd(Co, Ci, 3, 3, seed=21, scale=1 / math.sqrt(9 * Ci))
    b = rnd(Co, seed=22)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    r1 = h(rnd(B * OH * OW, Co, seed=23))
    xin = F.relu(x) if relu_a else x
    ref = F.conv2d(xin.permute(0, 3, 1, 2), h(w), b, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    ref = F.relu(ref) + r1
    out = torch.empty(B * OH * OW, Co, device=DEV, dtype=torch.float16)
    rt.gemm(x.half().to(DEV), pack.conv3x3(w.to(DEV), torch.float16), B * OH * OW, Co, 9 * Ci, out=out, bias=b.to(DEV),
            act=_abi.ACT_RELU, relu_a=relu_a, res1=r1.half().to(DEV),
            conv=dict(B=B, H=H, W=W, C=Ci, OH=OH, OW=OW, stride=stride))
    close(out, ref, 1.5e-3)


@pytest.mark.parametrize("k,Ci,Co", [(4, 48, 48), (2, 96, 96)])
def test_gemm_conv_transpose(rt, k, Ci, Co):
    from vdn import pack, _abi
    B, H, W = 2, 5, 7
    x = h(rnd(B, H, W, Ci, seed=30))
    w = rnd(Ci, Co, k, k, seed=31, scale=1 / math.sqrt(Ci))
    b = rnd(Co, seed=32)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2), h(w), b, stride=k).permute(0, 2, 3, 1)
    wp, bp = pack.conv_transpose(w.to(DEV), b.to(DEV), torch.float16)
    out = torch.empty(B, H * k, W * k, Co, device=DEV, dtype=torch.float16)
    rt.gemm(x.half().to(DEV), wp, B * H * W, k * k * Co, Ci, out=out, bias=bp, store=_abi.ST_CONVT,
            convt=dict(k=k, cout=Co, B=B, H=H, W=W))
    close(out, ref, 1.5e-3)


def test_gemm_geglu(rt):
    from vdn import pack, _abi
    M, c = 150, 64
    a = h(rnd(M, c, seed=40))
    w = rnd(8 * c, c, seed=41, scale=1 / math.sqrt(c))
    b = rnd(8 * c, seed=42)
    y = a @ h(w).t() + b
    hh, gate = y.chunk(2, dim=-1)
    ref = hh * F.gelu(gate)
    wp, bp = pack.geglu(w.to(DEV), b.to(DEV), torch.float16)
    out = torch.empty(M, 4 * c, device=DEV, dtype=torch.float16)
    rt.gemm(a.half().to(DEV), wp, M, 8 * c, c, out=out, bias=bp, store=_abi.ST_GEGLU)
    close(out, ref, 1.5e-3)


def test_gemm_heads_qkv_split(rt):
    from vdn import _abi
    B, T, Hh = 2, 50, 6
    C = Hh * 64
    a = h(rnd(B * T, C, seed=50))
    w, wp = _w(3 * C, C, 51)
    b = rnd(3 * C, seed=52)
    y = (a @ h(w).t() + b).reshape(B, T, 3, Hh, 64)
    tp = 64
    q = torch.zeros(B * Hh, tp, 64, device=DEV, dtype=torch.float16)
    k = torch.zeros_like(q)
    vt = torch.zeros(B * Hh, 64, tp, device=DEV, dtype=torch.float16)
    rt.gemm(a.half().to(DEV), wp, B * T, 3 * C, C, bias=b.to(DEV), store=_abi.ST_HEADS,
            heads=dict(dst=[q, k, vt], transposed=[0, 0, 1], heads=Hh, tokens=T, tpad=tp))
    close(q.reshape(B, Hh, tp, 64)[:, :, :T], y[:, :, 0].permute(0, 2, 1, 3), 1.5e-3)
    close(k.reshape(B, Hh, tp, 64)[:, :, :T], y[:, :, 1].permute(0, 2, 1, 3), 1.5e-3)
    close(vt.reshape(B, Hh, 64, tp)[:, :, :, :T], y[:, :, 2].permute(0, 2, 3, 1), 1.5e-3)
    assert float(vt.reshape(B, Hh, 64, tp)[:, :, :, T:].abs().max()) == 0.0  # pad columns untouched


def test_gemm_heads_rope_with_slot_offset(rt):
    """k (rotated) + v^T into ring slot 1 of a 2-slot bank: sam2 position_encoding.py:212-239 semantics."""
    from vdn import pack, _abi
    from oracle import ref_cpu as O
    B, side, Hh = 2, 5, 2
    P, C = side * side, Hh * 64
    a = h(rnd(B * P, C, seed=60))
    wk, wv = rnd(C, C, seed=61, scale=1 / math.sqrt(C)), rnd(C, C, seed=62, scale=1 / math.sqrt(C))
    bk, bv = rnd(C, seed=63), rnd(C, seed=64)
    kk = (a @ h(wk).t() + bk).reshape(B, P, Hh, 64).transpose(1, 2)
    vv = (a @ h(wv).t() + bv).reshape(B, P, Hh, 64).transpose(1, 2)
    fc = O.compute_axial_cis(64, side, side)
    _, kr = O.apply_rotary_enc(kk, kk, fc, False)
    wp, bp = pack.cat_proj([wk.to(DEV), wv.to(DEV)], [bk.to(DEV), bv.to(DEV)], [1, 0], torch.float16)
    cs = pack.rope_table(side, side, 64, device=DEV)
    tp = 64
    kd = torch.zeros(B * Hh, tp, 64, device=DEV, dtype=torch.float16)
    vd = torch.zeros(B * Hh, 64, tp, device=DEV, dtype=torch.float16)
    rt.gemm(a.half().to(DEV), wp, B * P, 2 * C, C, bias=bp, store=_abi.ST_HEADS,
            heads=dict(dst=[kd, vd], transposed=[0, 1], rope=[1, 0], rope_cs=cs, rope_mod=P, heads=Hh, tokens=P,
                       tok_off=P, tpad=tp))
    close(kd.reshape(B, Hh, tp, 64)[:, :, P:2 * P], kr, 1.5e-3)
    close(vd.reshape(B, Hh, 64, tp)[:, :, :, P:2 * P], vv.transpose(2, 3), 1.5e-3)
    assert float(kd.reshape(B, Hh, tp, 64)[:, :, :P].abs().max()) == 0.0


# --------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("C", [64, 384, 1024])
def test_layernorm_variants(rt, C):
    rows, grp = 30, 10
    x = rnd(rows, C, seed=70) * 3 + 1
    w, b = rnd(C, seed=71), rnd(C, seed=72)
    vec, tab = rnd(C, seed=73), rnd(4, C, seed=74)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    oh = torch.empty(rows, C, device=DEV, dtype=torch.float16)
    of = torch.empty(rows, C, device=DEV)
    rt.layernorm(x.to(DEV), rows, C, w.to(DEV), b.to(DEV), 1e-6, out_h=oh, out_f=of)
    close(of, ref, 1e-5)
    close(oh, ref, 1e-3)
    # + alpha*vec + table[(row / 5) % 4]
    rt.layernorm(x.to(DEV), rows, C, w.to(DEV), b.to(DEV), 1e-5, out_f=of, addvec=vec.to(DEV), alpha=0.5,
                 addtab=tab.to(DEV), tab_div=5, tab_mod=4)
    idx = (torch.arange(rows) // 5) % 4
    close(of, F.layer_norm(x, (C,), w, b, 1e-5) + 0.5 * vec + tab[idx], 1e-5)
    # drop the first row of each group of 10 (cls token) and compact
    oc = torch.empty(rows - rows // grp, C, device=DEV)
    rt.layernorm(x.to(DEV), rows, C, w.to(DEV), b.to(DEV), 1e-6, out_f=oc, out_group=grp)
    keep = [r for r in range(rows) if r % grp != 0]
    close(oc, ref[keep], 1e-5)
    # half input
    xh = h(x)
    rt.layernorm(xh.half().to(DEV), rows, C, w.to(DEV), b.to(DEV), 1e-6, out_f=of)
    close(of, F.layer_norm(xh, (C,), w, b, 1e-6), 1e-5)


@pytest.mark.parametrize("F_,HW,C", [(3, 50, 192), (2, 1369, 256), (2, 361, 1024), (1, 100, 64)])
def test_groupnorm(rt, F_, HW, C):
    x = h(rnd(F_, HW, C, seed=80) * 2 + 0.5)
    w, b = rnd(C, seed=81), rnd(C, seed=82)
    ref = F.group_norm(x.permute(0, 2, 1).reshape(F_, C, HW, 1), 32, w, b, 1e-6).reshape(F_, C, HW).permute(0, 2, 1)
    y = torch.empty(F_, HW, C, device=DEV, dtype=torch.float16)
    rt.groupnorm(x.half().to(DEV), y, F_, HW, C, 32, w.to(DEV), b.to(DEV), 1e-6)
    close(y, ref, 1.5e-3)


# --------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,nq,nk", [(2, 3, 150, 200), (1, 2, 1370, 1370), (1, 1, 37, 64), (2, 2, 129, 2738)])
def test_flash_attention(rt, B, H, nq, nk):
    from vdn.runtime import ceil_to
    q, k, v = h(rnd(B, H, nq, 64, seed=90)), h(rnd(B, H, nk, 64, seed=91)), h(rnd(B, H, nk, 64, seed=92))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, nq, H * 64)
    qp, kp = ceil_to(nq, 64), ceil_to(nk, 64)
    qd = torch.zeros(B * H, qp, 64, device=DEV, dtype=torch.float16)
    kd = torch.full((B * H, kp, 64), float("nan"), device=DEV, dtype=torch.float16)  # pad rows must not matter
    vd = torch.zeros(B * H, 64, kp, device=DEV, dtype=torch.float16)
    qd[:, :nq] = q.reshape(B * H, nq, 64).half().to(DEV)
    kd[:, :nk] = k.reshape(B * H, nk, 64).half().to(DEV)
    vd[:, :, :nk] = v.reshape(B * H, nk, 64).transpose(1, 2).half().to(DEV)
    out = torch.empty(B, nq, H * 64, device=DEV, dtype=torch.float16)
    rt.flash_attn(qd, kd, vd, out, B, H, nq, qp, nk, kp, 0.125)
    close(out, ref, 2e-3)


def test_flash_attention_online_max_rescale(rt):
    """A key whose score jumps far above the running max in a LATE tile forces the rescale branch."""
    nq, nk = 64, 320
    q, k, v = h(rnd(1, 1, nq, 64, seed=93)), h(rnd(1, 1, nk, 64, seed=94)), h(rnd(1, 1, nk, 64, seed=95))
    k[0, 0, 290] = q[0, 0, 5] * 6.0   # spike for query 5 in the last tile
    k[0, 0, 10] = q[0, 0, 9] * 6.0    # and an early spike for query 9
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(1, nq, 64)
    vd = v.reshape(1, nk, 64).transpose(1, 2).contiguous().half().to(DEV)
    out = torch.empty(1, nq, 64, device=DEV, dtype=torch.float16)
    rt.flash_attn(q.reshape(1, nq, 64).half().to(DEV), k.reshape(1, nk, 64).half().to(DEV), vd, out, 1, 1, nq, nq, nk, nk, 0.125)
    close(out, ref, 2e-3)


@pytest.mark.parametrize("Bv,T,D,c", [(2, 32, 9, 64), (1, 7, 5, 192), (1, 32, 3, 1024), (2, 4, 6, 256), (1, 1, 4, 384),
                                       (1, 64, 5, 256), (2, 45, 3, 1024), (1, 33, 4, 64)])
def test_temporal_attention(rt, Bv, T, D, c):
    heads, dh = 8, c // 8
    qkv = h(rnd(Bv * T, D, 3 * c, seed=100))
    x = qkv.reshape(Bv, T, D, 3, heads, dh).permute(3, 0, 2, 4, 1, 5)  # [3, b, d, head, f, e]
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2])              # softmax over frames
    ref = ref.permute(0, 3, 1, 2, 4).reshape(Bv * T, D, c)
    out = torch.empty(Bv * T, D, c, device=DEV, dtype=torch.float16)
    rt.temporal_attn(qkv.half().to(DEV), out, Bv, T, D, c, heads, dh ** -0.5)
    close(out, ref, 2e-3)


# --------------------------------------------------------------------------------------------- spatial
def test_upsample_bilinear_align_corners(rt):
    B, C = 2, 64
    for (ih, iw, oh, ow) in [(19, 19, 37, 37), (37, 28, 74, 56), (296, 296, 518, 518), (5, 7, 5, 7)]:
        b = 1 if oh > 200 else B
        x = h(rnd(b, ih, iw, C, seed=110))
        ref = F.interpolate(x.permute(0, 3, 1, 2), (oh, ow), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        y = torch.empty(b, oh, ow, C, device=DEV, dtype=torch.float16)
        rt.upsample(x.half().to(DEV), y, b, ih, iw, oh, ow, C)
        close(y, ref, 1e-3)
    x = rnd(2, 30, 41, seed=111)
    ref = F.relu(F.interpolate(x[:, None], (77, 50), mode="bilinear", align_corners=True)[:, 0])
    y = torch.empty(2, 77, 50, device=DEV)
    rt.upsample_f32(x.to(DEV), y, 2, 30, 41, 77, 50, relu=True)
    close(y, ref, 1e-5)


def test_patchify_fill_row_addvec_cast(rt):
    B, H, W = 2, 28, 42
    img = rnd(B, 3, H, W, seed=120)
    rows = torch.empty(B * 2 * 3, 640, device=DEV, dtype=torch.float16)
    rt.patchify(img.to(DEV), rows, B, H, W, 640)
    ref = F.unfold(img, 14, stride=14).transpose(1, 2).reshape(B * 6, 588)  # (c,ky,kx) order == conv weight order
    close(rows[:, :588], h(ref), 1e-6)
    assert float(rows[:, 588:].abs().max()) == 0.0
    x = torch.zeros(B * 5, 16, device=DEV)
    vec = rnd(16, seed=121)
    rt.fill_row(x, vec.to(DEV), B, 5, 0, 16)
    close(x.reshape(B, 5, 16)[:, 0], vec.expand(B, 16), 1e-7)
    assert float(x.reshape(B, 5, 16)[:, 1:].abs().max()) == 0.0
    a = rnd(12, 16, seed=122)
    y = torch.empty(12, 16, device=DEV)
    rt.add_vec(a.to(DEV), vec.to(DEV), 0.1, y, 12, 16)
    close(y, a + 0.1 * vec, 1e-6)
    z = torch.empty(12, 16, device=DEV, dtype=torch.float16)
    rt.cast(a.to(DEV), z)
    close(z, a.half(), 1e-6)


@pytest.mark.parametrize("oh,ow", [(16, 16), (28, 37), (19, 19), (40, 31)])
def test_bicubic_matches_torch_scale_factor_semantics(rt, oh, ow):
    """interpolate_pos_encoding (dinov2.py:193-203): scale_factor=((oh+0.1)/37, (ow+0.1)/37), bicubic."""
    C, gs = 24, 37
    g = rnd(gs, gs, C, seed=130)
    sx, sy = (oh + 0.1) / gs, (ow + 0.1) / gs
    ref = F.interpolate(g.permute(2, 0, 1)[None], scale_factor=(sx, sy), mode="bicubic")[0].permute(1, 2, 0)
    assert ref.shape[:2] == (oh, ow)
    y = torch.empty(oh, ow, C, device=DEV)
    rt.bicubic(g.to(DEV), y, gs, gs, oh, ow, C, sx, sy)
    close(y, ref, 1e-5)


def test_head_out(rt):
    M, C = 1000, 32
    f = F.relu(h(rnd(M, C, seed=140)))
    w = rnd(C, seed=141)
    ref = f @ w + 0.3
    d = torch.empty(M, device=DEV)
    rt.head_out(f.half().to(DEV), w.to(DEV), 0.3, d, M, C, relu=False)
    close(d, ref, 1e-5)
    rt.head_out(f.half().to(DEV), w.to(DEV), 0.3, d, M, C, relu=True)
    close(d, F.relu(ref), 1e-5)


def test_mask_downsampler_stages(rt):
    from oracle import ref_cpu as O
    B, H, W = 2, 266, 266
    depth = F.relu(rnd(B, 1, H, W, seed=150))
    p = {}
    e0 = dict(w0=rnd(4, 1, 3, 3, seed=151), b0=rnd(4, seed=152), lw=rnd(4, seed=153) + 1, lb=rnd(4, seed=154), w3=rnd(1, 4, 1, 1, seed=155), b3=rnd(1, seed=156))
    e1 = dict(w0=rnd(49, 1, 7, 7, seed=157, scale=0.2), b0=rnd(49, seed=158), lw=rnd(49, seed=159) + 1, lb=rnd(49, seed=160), w3=rnd(1, 49, 1, 1, seed=161), b3=rnd(1, seed=162))
    m = torch.sigmoid(depth)
    m = F.conv2d(m, e0["w0"], e0["b0"], stride=2, padding=1)
    m = F.conv2d(F.gelu(O.layer_norm_2d(m, e0["lw"], e0["lb"])), e0["w3"], e0["b3"])
    ref1 = m
    m = F.conv2d(m, e1["w0"], e1["b0"], stride=7)
    ref2 = F.conv2d(F.gelu(O.layer_norm_2d(m, e1["lw"], e1["lb"])), e1["w3"], e1["b3"])
    flat = lambda e: torch.cat([e[k].reshape(-1) for k in ("w0", "b0", "lw", "lb", "w3", "b3")]).to(DEV)
    h1, w1 = ref1.shape[-2:]
    h2, w2 = ref2.shape[-2:]
    o1 = torch.empty(B, h1, w1, device=DEV)
    rt.mask_down1(depth[:, 0].contiguous().to(DEV), o1, B, H, W, h1, w1, flat(e0))
    close(o1, ref1[:, 0], 1e-5)
    o2 = torch.empty(B, h2, w2, device=DEV)
    rt.mask_down2(o1, o2, B, h1, w1, h2, w2, flat(e1))
    close(o2, ref2[:, 0], 1e-4)


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 11, 64), (2, 37, 37, 1024), (1, 19, 19, 384), (1, 10, 150, 64), (1, 80, 74, 32)])   # W > 73: 64-column tiles
def test_dwconv7(rt, B, H, W, C):
    """row tiles with a tail (37 = 4 x 8 + 5), column runs with a tail (37 = 9 x 4 + 1), several channel blocks"""
    x = rnd(B, H, W, C, seed=170)
    w, b = rnd(C, 1, 7, 7, seed=171, scale=0.15), rnd(C, seed=172)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    y = torch.empty(B, H, W, C, device=DEV)
    rt.dwconv7(x.to(DEV), y, B, H, W, C, w.reshape(C, 49).t().contiguous().to(DEV), b.to(DEV))
    close(y, ref, 1e-5)


def test_bf16_operand_mode(rt):
    """The same GEMM / attention kernels instantiated for bf16 operands (VDN_HALF=bf16 mode)."""
    from vdn.runtime import Runtime
    from vdn import pack
    rb = Runtime(torch.device("cuda:0"), torch.bfloat16)
    M, N, K = 200, 128, 256
    a = rnd(M, K, seed=180).bfloat16().float()
    w = (rnd(N, K, seed=181) / 16).bfloat16().float()
    out = torch.empty(M, N, device=DEV)
    rb.gemm(a.bfloat16().to(DEV), pack.linear(w.to(DEV), torch.bfloat16), M, N, K, out=out)
    close(out, a @ w.t(), 1e-4)
    nq = nk = 100
    q, k, v = [rnd(1, 2, nq, 64, seed=s).bfloat16().float() for s in (182, 183, 184)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(1, nq, 128)
    kd = torch.zeros(2, 128, 64, device=DEV, dtype=torch.bfloat16)
    vd = torch.zeros(2, 64, 128, device=DEV, dtype=torch.bfloat16)
    kd[:, :nk] = k[0].bfloat16().to(DEV)
    vd[:, :, :nk] = v[0].transpose(1, 2).bfloat16().to(DEV)
    o = torch.empty(1, nq, 128, device=DEV, dtype=torch.bfloat16)
    rb.flash_attn(q[0].bfloat16().contiguous().to(DEV), kd, vd, o, 1, 2, nq, nq, nk, 128, 0.125)
    close(o, ref, 1.5e-2)


# --------------------------------------------------------------------------------------------- split precision
@pytest.fixture(scope="module")
def rt3():
    from vdn.runtime import Runtime
    return Runtime(torch.device("cuda:0"), torch.float16, split=True)


def test_x3_gemm_is_fp32_faithful(rt3):
    """hi/lo planes on both operands: error vs the UNROUNDED fp32 product drops from ~4e-4 to ~1e-6."""
    from vdn import pack, _abi
    M, N, K = 300, 192, 520
    a = rnd(M, K, seed=200)
    w = rnd(N, K, seed=201, scale=1 / math.sqrt(K))
    b = rnd(N, seed=202)
    ref = F.gelu(a.double() @ w.double().t() + b.double()).float()
    out = torch.empty(M, N, device=DEV)
    rt3.gemm(rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec), M, N, K, out=out, bias=b.to(DEV), act=_abi.ACT_GELU)
    close(out, ref, 3e-6)
    # half output written as planes
    oh = rt3.hbuf("t_x3_out", (M, N))
    rt3.gemm(rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec), M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU)
    close(oh.float(), ref, 3e-6)
    assert bool(((oh.hi.float() * oh.lo.float()) >= 0).all())  # planes share their sign


def test_x3_conv_relu_residual(rt3):
    from vdn import pack, _abi
    B, H, W, Ci, Co = 2, 13, 11, 48, 64
    x = rnd(B, H, W, Ci, seed=210)
    w = rnd(Co, Ci, 3, 3, seed=211, scale=1 / math.sqrt(9 * Ci))
    b = rnd(Co, seed=212)
    r1 = rnd(B * H * W, Co, seed=213)
    ref = F.conv2d(F.relu(x).permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    ref = (F.relu(ref) + r1.double()).float()
    # the A planes must come from the device split (round-toward-zero hi) for relu_a to be plane-wise
    xa = rt3.hbuf("t_x3_a", (B * H * W, Ci))
    xf = x.reshape(-1, Ci).to(DEV)
    # RTZ split on the host: hi = trunc-toward-zero to fp16
    h_rtn = xf.half()
    over = h_rtn.float().abs() > xf.abs()
    step = torch.nextafter(h_rtn, torch.zeros_like(h_rtn))
    hi = torch.where(over, step, h_rtn)
    xa.hi.copy_(hi)
    xa.lo.copy_((xf - hi.float()).half())
    out = rt3.hbuf("t_x3_o", (B * H * W, Co))
    rt3.gemm(xa, pack.conv3x3(w.to(DEV), rt3.prec), B * H * W, Co, 9 * Ci, out=out, bias=b.to(DEV), act=_abi.ACT_RELU,
             relu_a=True, res1=rt3.to_half(r1.to(DEV)), conv=dict(B=B, H=H, W=W, C=Ci, OH=H, OW=W, stride=1))
    close(out.float(), ref, 5e-6)


def test_x3_heads_rope_and_flash(rt3):
    from vdn import pack, _abi
    from vdn.runtime import ceil_to
    from oracle import ref_cpu as O
    B, side, Hh = 1, 7, 2
    P, C = side * side, Hh * 64
    a = rnd(B * P, C, seed=220)
    ws = [rnd(C, C, seed=221 + i, scale=1.5 / math.sqrt(C)) for i in range(3)]
    bs = [rnd(C, seed=225 + i) for i in range(3)]
    q, k, v = [(a.double() @ w.double().t() + b.double()).float().reshape(B, P, Hh, 64).transpose(1, 2) for w, b in zip(ws, bs)]
    fc = O.compute_axial_cis(64, side, side)
    q, k = O.apply_rotary_enc(q, k, fc, False)
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).float().transpose(1, 2).reshape(B, P, C)
    wp, bp = pack.cat_proj([w.to(DEV) for w in ws], [b.to(DEV) for b in bs], [1, 1, 0], rt3.prec)
    tp = ceil_to(P, 64)
    qd, kd = rt3.hbuf("t_q", (B * Hh, tp, 64), zero=True), rt3.hbuf("t_k", (B * Hh, tp, 64), zero=True)
    vd = rt3.hbuf("t_v", (B * Hh, 64, tp), zero=True)
    cs = pack.rope_table(side, side, 64, device=DEV)
    q8, k8 = rt3.qk8("t_q8", B * Hh, tp), rt3.qk8("t_k8", B * Hh, tp)
    rt3.gemm(rt3.to_half(a.to(DEV)), wp, B * P, 3 * C, C, bias=bp, store=_abi.ST_HEADS,
             heads=dict(dst=[qd, kd, vd], dst8=[q8, k8, None], transposed=[0, 0, 1], rope=[1, 1, 0], rope_cs=cs, rope_mod=P,
                        heads=Hh, tokens=P, tpad=tp))
    # the 8-bit planes the epilogue wrote: e5m2(value) | e5m2(remainder * 2^10) per token
    for plane8, planes in ((q8, qd), (k8, kd)):
        want_hi = planes.float()[:, :P].to(torch.float8_e5m2).view(torch.uint8)
        want_lo = (planes.lo.float()[:, :P] * 1024.0).to(torch.float8_e5m2).view(torch.uint8)
        got = plane8[:, :P]
        # e5m2(value) is rounded from the fp32 accumulator, the check rounds hi + lo (21 bits): ties may differ by one code
        assert (got[..., :64].int() - want_hi.int()).abs().max() <= 1 and (got[..., :64] != want_hi).float().mean() < 1e-3
        assert torch.equal(got[..., 64:], want_lo)
    close(qd.float().reshape(B, Hh, tp, 64)[:, :, :P], q, 5e-6)
    close(vd.float().reshape(B, Hh, 64, tp)[:, :, :, :P], v.transpose(2, 3), 5e-6)
    out = rt3.hbuf("t_o", (B * P, C))
    for pv, tol in ((3, 1e-5), (2, 3e-4)):  # P split into planes / P rounded once to 16 bits (include/vdn.h)
        rt3.pv_products = pv
        try:
            rt3.flash_attn(qd, kd, vd, out, B, Hh, P, tp, P, tp, 0.125)
        finally:
            rt3.pv_products = PV_DEFAULT
        close(out.float().reshape(B, P, C), ref, tol)
    rt3.pv_products = 2
    try:
        rt3.flash_attn(qd, kd, vd, out, B, Hh, P, tp, P, tp, 0.125, q8=q8, k8=k8)  # score cross terms on the 8-bit MFMA
    finally:
        rt3.pv_products = PV_DEFAULT
    close(out.float().reshape(B, P, C), ref, 3e-4)
    # one-product P V: the projection rounds V^T's hi plane to NEAREST (lo = remainder); the kernel reads hi alone
    vt_ref = v.transpose(2, 3).reshape(B * Hh, 64, P).to(DEV)
    got_hi = vd.hi[:, :, :P]
    # round-to-nearest: half an fp16 ulp at most, and the same code as RN(reference) except where the fp32 accumulator
    # and the fp64 reference straddle a rounding boundary (the toward-zero split differs in about half of the elements)
    assert ((got_hi.float() - vt_ref).abs() <= vt_ref.abs() * 2.0 ** -11 * 1.01 + 1e-6).all()
    assert (got_hi == vt_ref.half()).float().mean() > 0.99
    close(vd.float().reshape(B, Hh, 64, tp)[:, :, :, :P], v.transpose(2, 3), 5e-6)
    rt3.pv_products = 1
    try:
        rt3.flash_attn(qd, kd, vd, out, B, Hh, P, tp, P, tp, 0.125, q8=q8, k8=k8)
    finally:
        rt3.pv_products = PV_DEFAULT
    close(out.float().reshape(B, P, C), ref, 4e-4)


@pytest.mark.parametrize("pv,tol,qk8", [(3, 1e-5, False), (2, 3e-4, False), (2, 3e-4, True), (1, 4e-4, True)])
@pytest.mark.parametrize("nq,nk,gain", [(150, 200, 1.0), (1370, 1370, 1.0), (37, 64, 1.0), (70, 128, 1.0), (100, 130, 1.0),
                                        (129, 777, 6.0), (40, 8214, 1.0), (300, 321, 3.0)])
def test_x3_flash_attention(rt3, nq, nk, gain, pv, tol, qk8):
    """1 / 2 / 3 / many key tiles (the software pipeline's prologue, peeled first and last iterations), ragged
    last tile; gain 6 makes row maxima jump by far more than the lazy-rescale threshold between tiles.
    pv = 3: P carried as hi/lo planes, fp32-faithful (1e-5 against fp64). pv = 2 (the default): every softmax weight
    rounded once to fp16 and normalised by the sum of the rounded weights: <= 2^-11 relative per weight, which on these
    independent random V rows (the worst case: nothing in common to cancel) gives ~1e-4 (pv = 1 adds V's own fp16
    rounding, 2^-12 per element, to that); end to end it is invisible
    (tests/test_gpu_e2e.py prints 4e-6..1e-5 either way). qk8: the score cross terms K_hi Q_lo^T + K_lo Q_hi^T on the
    block-scaled e5m2 MFMA from the 8-bit planes (built here as the projection epilogue builds them); its own error
    (~1e-5 of a logit) disappears under the pv = 2 rounding. With the 8-bit planes the kernel is flash_attn2_kernel (generated
    instruction stream, S and P double-buffered by tile parity — 1, 2, 3, 4, 6, 13, 22 and 129 tiles walk every
    first / even / odd / last instantiation). pv is the per-call pv_products argument of vdn_flash_attn (Runtime.pv_products)."""
    from vdn import _abi
    from vdn.runtime import ceil_to
    rt3.pv_products = pv
    B, H = 1, 2
    q, k, v = rnd(B, H, nq, 64, seed=230, scale=gain), rnd(B, H, nk, 64, seed=231), rnd(B, H, nk, 64, seed=232)
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).float().transpose(1, 2).reshape(B, nq, H * 64)
    qp, kp = ceil_to(nq, 64), ceil_to(nk, 64)
    qd, kd, vd = rt3.hbuf("t2_q", (B * H, qp, 64), zero=True), rt3.hbuf("t2_k", (B * H, kp, 64), zero=True), rt3.hbuf("t2_v", (B * H, 64, kp), zero=True)
    for dst, src in ((qd, q.reshape(B * H, nq, 64)), (kd, k.reshape(B * H, nk, 64))):
        s = rt3.to_half(src.to(DEV))
        dst.hi[:, :src.shape[1]] = s.hi
        dst.lo[:, :src.shape[1]] = s.lo
    vt = v.reshape(B * H, nk, 64).transpose(1, 2).contiguous().to(DEV)
    s = rt3.to_half(vt)
    vd.hi[:, :, :nk] = s.hi
    vd.lo[:, :, :nk] = s.lo
    if pv == 1:  # one-product P V reads the hi plane alone: rounded to nearest, as the projection writes it in that mode
        vd.hi[:, :, :nk] = vt.half()
        vd.lo[:, :, :nk] = (vt - vt.half().float()).half()
    out = rt3.hbuf("t2_o", (B * nq, H * 64))
    q8 = k8 = None
    if qk8:
        def planes8(t):  # HL [BH, pad, 64] -> u8 [BH, pad, 128]
            return torch.cat([t.float().to(torch.float8_e5m2).view(torch.uint8),
                              (t.lo.float() * 1024.0).to(torch.float8_e5m2).view(torch.uint8)], dim=-1).contiguous()
        q8, k8 = planes8(qd), planes8(kd)
    try:
        rt3.flash_attn(qd, kd, vd, out, B, H, nq, qp, nk, kp, 0.125, q8=q8, k8=k8)
        close(out.float().reshape(B, nq, H * 64), ref, tol)
        first = (out.hi.clone(), out.lo.clone())
        for _ in range(3):  # bitwise repeatable (race screen for the LDS ring / counted waits)
            rt3.flash_attn(qd, kd, vd, out, B, H, nq, qp, nk, kp, 0.125, q8=q8, k8=k8)
            assert torch.equal(out.hi, first[0]) and torch.equal(out.lo, first[1])
    finally:
        rt3.pv_products = PV_DEFAULT


@pytest.mark.parametrize("T", [32, 64, 50])
def test_x3_temporal_attention_up_to_64_frames(rt3, T):
    Bv, D, c = 1, 5, 192
    qkv = rnd(Bv * T, D, 3 * c, seed=245)
    x = qkv.reshape(Bv, T, D, 3, 8, c // 8).permute(3, 0, 2, 4, 1, 5).double()
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).permute(0, 3, 1, 2, 4).reshape(Bv * T, D, c).float()
    out = rt3.hbuf(f"t3_o{T}", (Bv * T, D, c))
    rt3.temporal_attn(rt3.to_half(qkv.to(DEV)), out, Bv, T, D, c, 8, (c // 8) ** -0.5)
    close(out.float(), ref, 1e-5)


def test_x3_temporal_norms_upsample_headout(rt3):
    Bv, T, D, c = 1, 32, 5, 192
    qkv = rnd(Bv * T, D, 3 * c, seed=240)
    x = qkv.reshape(Bv, T, D, 3, 8, c // 8).permute(3, 0, 2, 4, 1, 5).double()
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2]).permute(0, 3, 1, 2, 4).reshape(Bv * T, D, c).float()
    out = rt3.hbuf("t3_o", (Bv * T, D, c))
    rt3.temporal_attn(rt3.to_half(qkv.to(DEV)), out, Bv, T, D, c, 8, (c // 8) ** -0.5)
    close(out.float(), ref, 1e-5)
    # LayerNorm -> planes
    xs = rnd(40, 384, seed=241)
    w, b = rnd(384, seed=242), rnd(384, seed=243)
    o = rt3.hbuf("t3_ln", (40, 384))
    rt3.layernorm(xs.to(DEV), 40, 384, w.to(DEV), b.to(DEV), 1e-6, out_h=o)
    close(o.float(), F.layer_norm(xs, (384,), w, b, 1e-6), 2e-6)
    # GroupNorm + upsample on planes
    xg = rnd(2, 100, 64, seed=244)
    refg = F.group_norm(xg.permute(0, 2, 1).reshape(2, 64, 100, 1), 32, w[:64], b[:64], 1e-6).reshape(2, 64, 100).permute(0, 2, 1)
    y = rt3.hbuf("t3_gn", (2, 100, 64))
    rt3.groupnorm(rt3.to_half(xg.to(DEV)), y, 2, 100, 64, 32, w[:64].contiguous().to(DEV), b[:64].contiguous().to(DEV), 1e-6)
    close(y.float(), refg, 5e-6)
    xu = rnd(1, 10, 10, 64, seed=245)
    refu = F.interpolate(xu.permute(0, 3, 1, 2), (23, 17), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    yu = rt3.hbuf("t3_up", (1, 23, 17, 64))
    rt3.upsample(rt3.to_half(xu.to(DEV)), yu, 1, 10, 10, 23, 17, 64)
    close(yu.float(), refu, 3e-6)
    f = F.relu(rnd(500, 32, seed=246))
    wv = rnd(32, seed=247)
    d = torch.empty(500, device=DEV)
    rt3.head_out(rt3.to_half(f.to(DEV)), wv.to(DEV), 0.1, d, 500, 32, relu=False)
    close(d, f @ wv + 0.1, 3e-6)
    img = rnd(1, 3, 28, 28, seed=248)
    rows = rt3.hbuf("t3_rows", (4, 640))
    rt3.patchify(img.to(DEV), rows, 1, 28, 28, 640)
    close(rows.float()[:, :588], F.unfold(img, 14, stride=14).transpose(1, 2).reshape(4, 588), 1e-6)


@pytest.mark.parametrize("bm", ["128", "192", "256"])
@pytest.mark.parametrize("M,N,K", [(724, 1152, 384), (724, 384, 1536), (1370, 1152, 384), (2050, 256, 2304), (700, 1024, 64), (513, 200, 96)])
def test_x3_big_tile_gemm_ragged(rt3, bm, M, N, K, tune):
    """The 8-wave BM x 256 kernels on ragged M / N (tails in both), every BM variant forced."""
    from vdn import pack, _abi
    tune(force_bm=int(bm))
    a = rnd(M, K, seed=300)
    w = rnd(N, K, seed=301, scale=1 / math.sqrt(K))
    b, g = rnd(N, seed=302), rnd(N, seed=303)
    x = rnd(M, N, seed=304)
    ref = (x.double() + (a.double() @ w.double().t() + b.double()) * g.double()).float()
    xd = x.clone().to(DEV)
    rt3.gemm(rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec), M, N, K, out=xd, bias=b.to(DEV), gamma=g.to(DEV), res1=xd)
    close(xd, ref, 3e-6)
    oh = rt3.hbuf(f"t_big_{M}_{N}", (M, N))
    rt3.gemm(rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec), M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU)
    close(oh.float(), F.gelu(a.double() @ w.double().t() + b.double()).float(), 3e-6)


@pytest.mark.parametrize("bm", ["128", "256"])
def test_x3_big_tile_heads_and_conv(rt3, bm, tune):
    from vdn import pack, _abi
    from vdn.runtime import ceil_to
    tune(force_bm=int(bm))
    B, T, Hh = 2, 362, 6
    C = Hh * 64
    a = rnd(B * T, C, seed=310)
    w = rnd(3 * C, C, seed=311, scale=1 / math.sqrt(C))
    b = rnd(3 * C, seed=312)
    y = (a.double() @ w.double().t() + b.double()).float().reshape(B, T, 3, Hh, 64)
    tp = ceil_to(T, 64)
    q, k = rt3.hbuf("t_bq", (B * Hh, tp, 64), zero=True), rt3.hbuf("t_bk", (B * Hh, tp, 64), zero=True)
    vt = rt3.hbuf("t_bv", (B * Hh, 64, tp), zero=True)
    rt3.gemm(rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec), B * T, 3 * C, C, bias=b.to(DEV), store=_abi.ST_HEADS,
             heads=dict(dst=[q, k, vt], transposed=[0, 0, 1], heads=Hh, tokens=T, tpad=tp))
    close(q.float().reshape(B, Hh, tp, 64)[:, :, :T], y[:, :, 0].permute(0, 2, 1, 3), 3e-6)
    close(k.float().reshape(B, Hh, tp, 64)[:, :, :T], y[:, :, 1].permute(0, 2, 1, 3), 3e-6)
    close(vt.float().reshape(B, Hh, 64, tp)[:, :, :, :T], y[:, :, 2].permute(0, 2, 3, 1), 3e-6)
    # conv 3x3 with relu-on-load through the big kernel (M*N >= 256K)
    Bc, Hc, Wc, Ci, Co = 2, 40, 38, 64, 256
    x = rnd(Bc, Hc, Wc, Ci, seed=313)
    wc = rnd(Co, Ci, 3, 3, seed=314, scale=1 / math.sqrt(9 * Ci))
    bc = rnd(Co, seed=315)
    ref = F.conv2d(F.relu(x).permute(0, 3, 1, 2).double(), wc.double(), bc.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co).float()
    xa = rt3.hbuf("t_bca", (Bc * Hc * Wc, Ci))
    xf = x.reshape(-1, Ci).to(DEV)
    h_rtn = xf.half()
    hi = torch.where(h_rtn.float().abs() > xf.abs(), torch.nextafter(h_rtn, torch.zeros_like(h_rtn)), h_rtn)
    xa.hi.copy_(hi)
    xa.lo.copy_((xf - hi.float()).half())
    out = rt3.hbuf("t_bco", (Bc * Hc * Wc, Co))
    rt3.gemm(xa, pack.conv3x3(wc.to(DEV), rt3.prec), Bc * Hc * Wc, Co, 9 * Ci, out=out, bias=bc.to(DEV), relu_a=True,
             conv=dict(B=Bc, H=Hc, W=Wc, C=Ci, OH=Hc, OW=Wc, stride=1))
    close(out.float(), ref, 5e-6)


def test_x3_big_tile_k_smaller_than_padded_stride_with_poisoned_neighbour(rt3, tune):
    """K = 96 (weights padded to 128): the A rows must not be read past K — the memory right after the
    A buffer is NaN here, which a read of columns 96..127 of the last row would pull in (0 x NaN)."""
    from vdn import pack, _abi
    tune(force_bm=128)
    M, N, K = 722, 384, 96
    a = rnd(M, K, seed=320)
    w = rnd(N, K, seed=321, scale=1 / math.sqrt(K))
    slab = torch.full((2, M * K + 4096), float("nan"), device=DEV, dtype=torch.float16)
    ah = rt3.to_half(a.to(DEV))
    from vdn.runtime import HL
    slab[0, : M * K] = ah.hi.reshape(-1)
    slab[1, : M * K] = ah.lo.reshape(-1)
    A = HL(slab[0, : M * K].reshape(M, K), slab[1, : M * K].reshape(M, K))
    out = torch.empty(M, N, device=DEV)
    rt3.gemm(A, pack.linear(w.to(DEV), rt3.prec), M, N, K, out=out)
    close(out, (a.double() @ w.double().t()).float(), 3e-6)


@pytest.mark.parametrize("p8,bm", [("1", "256"), ("2", "192")])
@pytest.mark.parametrize("K", [32, 64, 160])
def test_x3_pingpong_gemm_short_k_and_repeatability(rt3, p8, bm, K, tune):
    """The 8-phase ping-pong kernel with 1, 2 and 5 K tiles (prologue only / drain counts 2,0 / steady state),
    ragged M and N, run 8 times: every run must be bitwise identical (a counted-vmcnt or barrier-parity
    mistake shows up as rare differing tiles long before it shows up as a wrong mean)."""
    from vdn import pack, _abi
    tune(force_bm=int(bm))
    tune(p8=int(p8))
    M, N = 1370 + 77, 1024 + 200
    a = rnd(M, K, seed=330)
    w = rnd(N, K, seed=331, scale=1 / math.sqrt(K))
    b = rnd(N, seed=332)
    ref = F.gelu(a.double() @ w.double().t() + b.double()).float()
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    oh = rt3.hbuf(f"t_p8_{K}", (M, N))
    rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU)
    close(oh.float(), ref, 3e-6)
    first = (oh.hi.clone(), oh.lo.clone())
    for _ in range(7):
        rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU)
        assert torch.equal(oh.hi, first[0]) and torch.equal(oh.lo, first[1])


@pytest.mark.parametrize("bm", ["128", "192", "256"])
@pytest.mark.parametrize("two", [False, True])
def test_x3_conv_residual_plane_flavours(rt3, bm, two, tune):
    """3x3 conv + bias + one / two split-half residuals -> split-half output (the ResidualConvUnit's second conv:
    the straight-line epilogue flavours) on every 8-wave tile height, against fp64."""
    from vdn import pack
    tune(force_bm=int(bm))
    B, H, W, Ci, Co = 2, 41, 37, 64, 256
    x = rnd(B, H, W, Ci, seed=340)
    w = rnd(Co, Ci, 3, 3, seed=341, scale=1 / math.sqrt(9 * Ci))
    b = rnd(Co, seed=342)
    r1, r2 = rnd(B * H * W, Co, seed=343), rnd(B * H * W, Co, seed=344)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    ref = (ref + r1.double() + (r2.double() if two else 0)).float()
    out = rt3.hbuf(f"t_crf_{bm}", (B * H * W, Co))
    rt3.gemm(rt3.to_half(x.reshape(-1, Ci).to(DEV)), pack.conv3x3(w.to(DEV), rt3.prec), B * H * W, Co, 9 * Ci, out=out,
             bias=b.to(DEV), res1=rt3.to_half(r1.to(DEV)), res2=rt3.to_half(r2.to(DEV)) if two else None,
             conv=dict(B=B, H=H, W=W, C=Ci, OH=H, OW=W, stride=1))
    close(out.float(), ref, 5e-6)


@pytest.mark.parametrize("nw,fh,fw", [(1, 20, 24), (2, 37, 41), (4, 30, 33)])
def test_device_stitcher_matches_host_restatement(rt, nw, fh, fw):
    """vdn_stitch_fit / vdn_stitch_apply (the on-device window stitcher) against vdn.util.stitch, the host
    restatement of video_depth.py:118-156 that tests/test_host.py pins to the reference's own run."""
    from vdn import util
    from vdn.video_depth import DeviceStitcher
    wins = [(rnd(32, fh, fw, seed=400 + w).abs() * (1.0 + 0.3 * w) + 0.1 * w) for w in range(nw)]
    n = 32 + 22 * (nw - 1) - 5
    ref = util.stitch([w_[i].numpy() for w_ in wins for i in range(32)], n)
    st = DeviceStitcher(rt, nw, fh, fw)
    for w_ in wins:
        st.push(w_.to(DEV))
    got = st.result(n).cpu()
    assert got.shape == ref.shape
    close(got, torch.from_numpy(ref), 2e-6)


def test_device_stitcher_singular_fit_is_identity(rt):
    """all-equal alignment frames make the 2x2 normal matrix singular: scale 1, shift 0 (utils/util.py:55-56)."""
    pred = torch.full((2, 16, 16), 3.0, device=DEV)
    coef = torch.empty(2, device=DEV)
    rt.stitch_fit(pred, pred.clone(), coef)
    # a00*a11 - a01^2 = (9n)(n) - (3n)^2 = 0
    assert coef.cpu().tolist() == [1.0, 0.0]


@pytest.mark.parametrize("T,HW,c", [(1, 37, 256), (7, 50, 1024), (32, 133, 256), (32, 20, 64), (9, 31, 384), (32, 40, 192)])
def test_temporal_attention_last_frame_over_projected_cache(rt3, T, HW, c):
    """vdn_temporal_attn_last: W(x + pe[t]) = Wx + W pe[t] — cached projections without the position term plus
    [T, c] position tables must equal attention of the newest frame over re-projected (state + pe) inputs
    (motion_module.py:255-277), 8 heads."""
    g = torch.Generator().manual_seed(500 + T + c)
    states = torch.randn(T, HW, c, generator=g)
    pe = torch.randn(T, c, generator=g) * 0.5
    wq, wk, wv = (torch.randn(c, c, generator=g) / math.sqrt(c) for _ in range(3))
    xin = (states + pe[:, None, :]).double()
    q = (xin[-1] @ wq.double().t()).reshape(HW, 8, c // 8)
    k = (xin @ wk.double().t()).reshape(T, HW, 8, c // 8)
    v = (xin @ wv.double().t()).reshape(T, HW, 8, c // 8)
    att = torch.softmax(torch.einsum("phd,tphd->pht", q, k) * (c // 8) ** -0.5, dim=-1)
    ref = torch.einsum("pht,tphd->phd", att, v).reshape(HW, c).float()
    entries = [torch.cat([st.double() @ w.double().t() for w in (wq, wk, wv)], dim=1).float().to(DEV).contiguous() for st in states]
    tabs = [(pe.double() @ w.double().t()).float().to(DEV).contiguous() for w in (wq, wk, wv)]
    out = rt3.hbuf(f"t_tal_{T}_{c}", (HW, c))
    # the frames sit in scattered slots of a ring (as the streaming driver leaves them); the window lists them oldest first
    nslots = T + 5
    slots = torch.randperm(nslots, generator=g)[:T].tolist()
    pool = torch.full((nslots, HW, 3 * c), float("nan"), device=DEV)
    for t in range(T):
        pool[slots[t]] = entries[t]
    rt3.temporal_attn_last(pool, slots, tabs[0], tabs[1], tabs[2], out, HW, c, (c // 8) ** -0.5)
    close(out.float(), ref, 5e-6)


@pytest.mark.parametrize("F_,n", [(1, 1), (3, 2), (2, 1001), (5, 4096), (2, 90 * 121), (1, 300001)])
def test_frame_median_matches_torch_quantile(rt, F_, n):
    """vdn_frame_median = torch.quantile(x, 0.5) (linear interpolation): odd / even counts, negatives, duplicates."""
    g = torch.Generator().manual_seed(600 + n)
    x = torch.randn(F_, n, generator=g) * 3.0
    x[0, : n // 3] = x[0, 0]                       # a run of duplicates straddling the median region
    if F_ > 1:
        x[1] = torch.round(x[1])                   # heavy ties
    med = torch.empty(F_, device=DEV)
    rt.frame_median(x.to(DEV).contiguous(), med)
    ref = torch.quantile(x, 0.5, dim=-1)
    assert torch.equal(med.cpu(), ref), (med.cpu() - ref).abs().max()


def test_refiner_scale_pack_finish_kernels(rt):
    """vdn_refine_scale / _pack / _finish against the oracle's torch statement (utils/normal_utils.py:4-51)."""
    from oracle import ref_cpu as O
    F_, H, W = 3, 19, 23
    g = torch.Generator().manual_seed(610)
    x = torch.rand(F_, H, W, generator=g) * 60000.0 + 100.0
    med = torch.quantile(x.reshape(F_, -1), 0.5, dim=-1)
    w, b = 0.7, -0.2
    s_ref = torch.exp(torch.tanh(med / 65535.0 * w + b))
    scaled, sc = torch.empty(F_, H, W, device=DEV), torch.empty(F_, device=DEV)
    rt.refine_scale(x.to(DEV), med.to(DEV), w, b, 1.0, 65535.0, scaled, sc)
    close(sc.cpu(), s_ref, 1e-6)
    close(scaled.cpu(), x / 65535.0 * s_ref.reshape(F_, 1, 1), 1e-6)
    d = scaled.cpu()
    packed = torch.empty(F_, 3, H, W, device=DEV)
    rt.refine_pack(scaled, packed, normals=True)
    ref = torch.cat([d[:, None], O.sobel_normals(d[:, None])[:, :2]], dim=1)
    close(packed.cpu(), ref, 2e-6)
    rt.refine_pack(scaled, packed, normals=False)
    assert torch.equal(packed.cpu(), d[:, None].expand(-1, 3, -1, -1))
    depth = torch.rand(F_, H, W, generator=g)
    out = torch.empty(F_, H, W, device=DEV)
    rt.refine_finish(scaled, depth.to(DEV), 1.3, 0.05, 65535.0, True, out)
    close(out.cpu(), (d + (depth * 1.3 + 0.05)) * 65535.0, 1e-6)
    rt.refine_finish(None, depth.to(DEV), 1.3, 0.05, 65535.0, False, out)
    close(out.cpu(), depth * 65535.0, 1e-6)


def test_cu_hint_only_changes_the_tiling(rt3):
    """vdn_gemm_desc.cu_hint (the share of the chip a co-running lane can count on) picks another M tile and must
    not change the result beyond fp32 summation order."""
    from vdn import pack
    M, N, K = 5480, 1024, 256
    a = rnd(M, K, seed=700)
    w = rnd(N, K, seed=701, scale=1 / math.sqrt(K))
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    outs = []
    for hint in (0, 128, 64):
        rt3.cu_hint = hint
        o = torch.empty(M, N, device=DEV)
        rt3.gemm(A, W, M, N, K, out=o)
        outs.append(o.cpu())
    rt3.cu_hint = 0
    ref = (a.double() @ w.double().t()).float()
    for o in outs:
        close(o, ref, 3e-6)


@pytest.mark.parametrize("B,H,W,Ci,Co,two,relu_a", [(1, 19, 19, 256, 256, False, False), (2, 19, 19, 1024, 256, False, False),
                                                    (1, 37, 37, 256, 256, True, True), (4, 19, 19, 512, 1024, False, False)])
def test_x3_conv_split_k_matches_single_pass(rt3, B, H, W, Ci, Co, two, relu_a, tune):
    """Low-resolution convolutions with deep reductions run as K slices + an ordered reduce (vdn.h: splitk_ws):
    against fp64 and against the unsplit kernel (VDN_GEMM_NOSPLITK), bitwise repeatable."""
    from vdn import pack, _abi
    x = rnd(B, H, W, Ci, seed=800)
    w = rnd(Co, Ci, 3, 3, seed=801, scale=1 / math.sqrt(9 * Ci))
    b = rnd(Co, seed=802)
    r1, r2 = rnd(B * H * W, Co, seed=803), rnd(B * H * W, Co, seed=804)
    xin = F.relu(x) if relu_a else x
    ref = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    if two:
        ref = ref + r1.double() + r2.double()
    ref = ref.float()
    xa = rt3.hbuf(f"t_sk_a{Ci}", (B * H * W, Ci))
    xf = x.reshape(-1, Ci).to(DEV)
    h_rtn = xf.half()
    hi = torch.where(h_rtn.float().abs() > xf.abs(), torch.nextafter(h_rtn, torch.zeros_like(h_rtn)), h_rtn)
    xa.hi.copy_(hi)
    xa.lo.copy_((xf - hi.float()).half())
    wp = pack.conv3x3(w.to(DEV), rt3.prec)
    kw = dict(bias=b.to(DEV), relu_a=relu_a, conv=dict(B=B, H=H, W=W, C=Ci, OH=H, OW=W, stride=1))
    if two:
        kw.update(res1=rt3.to_half(r1.to(DEV)), res2=rt3.to_half(r2.to(DEV)))
    out = rt3.hbuf(f"t_sk_o{Ci}_{Co}", (B * H * W, Co))
    rt3.gemm(xa, wp, B * H * W, Co, 9 * Ci, out=out, **kw)
    close(out.float(), ref, 5e-6)
    first = (out.hi.clone(), out.lo.clone())
    rt3.gemm(xa, wp, B * H * W, Co, 9 * Ci, out=out, **kw)
    assert torch.equal(out.hi, first[0]) and torch.equal(out.lo, first[1])
    tune(no_splitk=1)
    rt3.gemm(xa, wp, B * H * W, Co, 9 * Ci, out=out, **kw)
    close(out.float(), first[0].float() + first[1].float(), 3e-6)


@pytest.mark.parametrize("M,N,K", [(1370, 1024, 4096), (1369, 1024, 1024), (361, 256, 2048)])
def test_x3_plain_split_k_residual_and_planes(rt3, M, N, K, tune):
    """Small-M linears (batch-1 encoder: 44 tiles of 128x256) run as K slices + ordered reduce: the in-place
    LayerScale residual update and a plane-output projection, against fp64 and the unsplit kernel."""
    from vdn import pack
    a = rnd(M, K, seed=810)
    w = rnd(N, K, seed=811, scale=1 / math.sqrt(K))
    b, g = rnd(N, seed=812), rnd(N, seed=813)
    x = rnd(M, N, seed=814)
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    ref = (x.double() + (a.double() @ w.double().t() + b.double()) * g.double()).float()
    xd = x.clone().to(DEV)
    rt3.gemm(A, W, M, N, K, out=xd, bias=b.to(DEV), gamma=g.to(DEV), res1=xd)
    close(xd, ref, 3e-6)
    oh = rt3.hbuf(f"t_psk_{M}_{N}", (M, N))
    rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV))
    close(oh.float(), (a.double() @ w.double().t() + b.double()).float(), 3e-6)
    first = (oh.hi.clone(), oh.lo.clone())
    rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV))
    assert torch.equal(oh.hi, first[0]) and torch.equal(oh.lo, first[1])
    tune(no_splitk=1)
    xd2 = x.clone().to(DEV)
    rt3.gemm(A, W, M, N, K, out=xd2, bias=b.to(DEV), gamma=g.to(DEV), res1=xd2)
    close(xd2, xd, 3e-6)


@pytest.mark.parametrize("B,IH,IW,C,OH,OW", [(2, 76, 76, 32, 133, 133), (1, 40, 24, 64, 70, 42), (1, 296, 296, 128, 518, 518)])
def test_depth_tail_fused_against_fp64_and_unfused(rt3, B, IH, IW, C, OH, OW):
    """vdn_depth_tail (resize align_corners -> conv3x3 + ReLU -> conv1x1 [+ ReLU], dpt.py:146-151) against the same
    ops in fp64 torch, with tile tails in both directions (133 = 8 x 16 + 5), a non-square map, every channel-block
    count (1, 2, 4 passes) and the full ViT-L size; and against the three-launch path it replaces."""
    from vdn import pack
    x = rnd(B, IH, IW, C, seed=900)
    w2 = rnd(32, C, 3, 3, seed=901, scale=1 / math.sqrt(9 * C))
    b2 = rnd(32, seed=902, scale=0.1)
    w1 = rnd(32, seed=903, scale=0.3)
    w1 = w1 - w1.mean()
    b1 = 0.2
    xd = x.double().permute(0, 3, 1, 2)
    up = F.interpolate(xd, (OH, OW), mode="bilinear", align_corners=True)
    mid = F.relu(F.conv2d(up, w2.double(), b2.double(), padding=1))
    ref = (mid * w1.double().reshape(1, 32, 1, 1)).sum(1) + b1
    xa = rt3.to_half(x.reshape(-1, C).to(DEV))
    xf = x.reshape(-1, C).to(DEV).contiguous()
    wt = pack.conv3x3_taps(w2.to(DEV), rt3.prec)
    d = torch.empty(B, OH, OW, device=DEV)
    rt3.depth_tail(xf, wt, b2.to(DEV), w1.to(DEV), b1, d, B, IH, IW, C, OH, OW, relu=False)
    close(d, ref.float(), 2e-5)  # the zero-mean 1x1 weights cancel ~4x: 2e-5 of the output is ~5e-6 of the conv sums
    rt3.depth_tail(xf, wt, b2.to(DEV), w1.to(DEV), b1, d, B, IH, IW, C, OH, OW, relu=True)
    close(d, F.relu(ref).float(), 2e-5)
    # the unfused path: upsample kernel -> implicit-GEMM conv -> head_out
    upb = rt3.hbuf(f"t_tail_up{C}_{OH}", (B * OH * OW, C))
    rt3.upsample(xa, upb, B, IH, IW, OH, OW, C)
    o2 = rt3.hbuf(f"t_tail_o2{C}_{OH}", (B * OH * OW, 32))
    from vdn import _abi
    rt3.gemm(upb, pack.conv3x3(w2.to(DEV), rt3.prec), B * OH * OW, 32, 9 * C, out=o2, bias=b2.to(DEV), act=_abi.ACT_RELU,
             conv=dict(B=B, H=OH, W=OW, C=C, OH=OH, OW=OW, stride=1))
    d2 = torch.empty(B, OH, OW, device=DEV)
    rt3.head_out(o2, w1.to(DEV), b1, d2, B * OH * OW, 32, relu=True)
    close(d, d2, 3e-5)  # two fp32 summation orders, each ~5e-6 from fp64, through the cancelling 1x1 weights


def test_pack_weight_abi_against_torch_layouts(rt3):
    """vdn_pack_weight / vdn_pack_bias (csrc/pack.hip) against the layouts written out in torch: every kind, K tails
    (zero padding), hi = nearest fp16, hi + lo within 2^-21 of the fp32 weight."""
    from vdn import pack

    def check(got, ref):  # got: HL [rows, ldb]; ref: f32 [rows, K]
        rows, K = ref.shape
        assert got.hi.shape[0] == rows and got.hi.shape[1] == (K + 63) // 64 * 64
        assert torch.equal(got.hi[:, :K].cpu(), ref.half())
        assert float((got.float()[:, :K].cpu() - ref).abs().max()) <= 2.0 ** -20 * float(ref.abs().max())
        assert float(got.hi[:, K:].abs().max() if got.hi.shape[1] > K else 0) == 0 and (got.lo is None or float(got.lo[:, K:].abs().sum()) == 0)

    w = rnd(200, 588, seed=950)
    check(pack.linear(w.to(DEV), rt3.prec), w)
    for ci in (48, 64, 128):
        w = rnd(40, ci, 3, 3, seed=951 + ci)
        if ci % 64 == 0:
            ref = w.reshape(40, ci // 64, 64, 3, 3).permute(0, 1, 3, 4, 2).reshape(40, 9 * ci)
        else:
            ref = w.permute(0, 2, 3, 1).reshape(40, 9 * ci)
        check(pack.conv3x3(w.to(DEV), rt3.prec), ref)
    w = rnd(32, 96, 3, 3, seed=955)
    check(pack.conv3x3_taps(w.to(DEV), rt3.prec), w.permute(0, 2, 3, 1).reshape(32, 9 * 96))
    w, b = rnd(24, 16, 4, 4, seed=956), rnd(16, seed=957)
    wp, bp = pack.conv_transpose(w.to(DEV), b.to(DEV), rt3.prec)
    check(wp, w.permute(2, 3, 1, 0).reshape(16 * 16, 24))
    assert torch.equal(bp.cpu(), b.repeat(16))
    w, b = rnd(128, 40, seed=958), rnd(128, seed=959)
    wp, bp = pack.geglu(w.to(DEV), b.to(DEV), rt3.prec)
    t, r = torch.arange(4), torch.arange(16)
    perm = torch.stack([t[:, None] * 16 + r[None, :], t[:, None] * 16 + r[None, :] + 64], dim=1).reshape(-1)
    check(wp, w[perm])
    assert torch.equal(bp.cpu(), b[perm])
    ws, bs = [rnd(128, 72, seed=960 + i) for i in range(3)], [rnd(128, seed=965 + i) for i in range(3)]
    wp, bp = pack.cat_proj([x.to(DEV) for x in ws], [x.to(DEV) for x in bs], [1, 1, 0], rt3.prec)
    p = torch.arange(64)
    blk, r = p // 16, p % 16
    src = 2 * ((blk // 2) * 16 + r) + (blk % 2)
    rp = (torch.arange(2)[:, None] * 64 + src[None, :]).reshape(-1)
    check(wp, torch.cat([ws[0][rp], ws[1][rp], ws[2]]))
    assert torch.equal(bp.cpu(), torch.cat([bs[0][rp], bs[1][rp], bs[2]]))
    one = pack.linear(rnd(64, 64, seed=970).to(DEV), torch.float16)  # 1-product modes: no lo plane
    assert one.lo is None


@pytest.mark.parametrize("M,N,K", [(2740, 1024, 1024), (1370 * 2 + 77, 3072, 384), (2048, 512, 4096)])
def test_x8_gemm_cross_terms_on_the_8bit_mfma(rt3, M, N, K, tune):
    """gemm_x8_kernel: A_hi W_hi^T on fp16 MFMAs + the two cross terms on the block-scaled e3m2 MFMA, from the planes of
    6-bit rows; ragged M (tile tail), 1 / 6 / 64 slabs; against fp64, against the 3-product kernel, bitwise repeatable."""
    from vdn import pack, _abi
    a = rnd(M, K, seed=980)
    w = rnd(N, K, seed=981, scale=1 / math.sqrt(K))
    b, g = rnd(N, seed=982), rnd(N, seed=983)
    x = rnd(M, N, seed=984)
    ref = (x.double() + (a.double() @ w.double().t() + b.double()) * g.double()).float()
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    A8, W8 = pack.planes8(A), pack.planes8(W)
    xd = x.clone().to(DEV)
    rt3.gemm(A, W, M, N, K, out=xd, bias=b.to(DEV), gamma=g.to(DEV), res1=xd, a8=A8, w8=W8)
    close(xd, ref, 2e-5)
    x3 = x.clone().to(DEV)
    rt3.gemm(A, W, M, N, K, out=x3, bias=b.to(DEV), gamma=g.to(DEV), res1=x3)  # three fp16 products
    close(xd, x3, 2e-5)
    # bias + GELU -> split planes + the 8-bit planes of the output (what fc1 hands to fc2)
    oh = rt3.hbuf(f"t_x8_{M}_{N}", (M, N))
    o8 = torch.zeros(2, M, N, dtype=torch.uint8, device=DEV)
    rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU, a8=A8, w8=W8, out8=o8)
    close(oh.float(), F.gelu(a.double() @ w.double().t() + b.double()).float(), 2e-5)
    _same_rows(o8, pack.planes8(oh, pack.ORDER_GEMM))    # the epilogue's rows == the packer's on the planes it wrote
    _check_x6(o8, oh, M, N, pack.ORDER_GEMM)
    first = (oh.hi.clone(), oh.lo.clone())
    for _ in range(5):
        rt3.gemm(A, W, M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU, a8=A8, w8=W8, out8=o8)
        assert torch.equal(oh.hi, first[0]) and torch.equal(oh.lo, first[1])


def test_nan_in_a_conv_input_reaches_the_output(rt3):
    """The optional ReLU of the plane-output conv epilogues must not swallow NaNs (fmaxf(NaN, 0) = 0 would hide a
    consumed-before-written element from the VDN_POISON screen and differs from torch.relu)."""
    from vdn import pack, _abi
    B, H, W, Ci, Co = 1, 20, 20, 64, 256
    x = rnd(B * H * W, Ci, seed=990)
    x[137, 5] = float("nan")
    w = rnd(Co, Ci, 3, 3, seed=991, scale=1 / math.sqrt(9 * Ci))
    xa = rt3.to_half(x.to(DEV))
    for act in (_abi.ACT_RELU, _abi.ACT_NONE):
        out = rt3.hbuf("t_nan_o", (B * H * W, Co))
        rt3.gemm(xa, pack.conv3x3(w.to(DEV), rt3.prec), B * H * W, Co, 9 * Ci, out=out, bias=rnd(Co, seed=992).to(DEV), act=act,
                 conv=dict(B=B, H=H, W=W, C=Ci, OH=H, OW=W, stride=1))
        bad = torch.isnan(out.float()).any(dim=1).reshape(H, W).cpu()
        assert int(bad.sum()) == 9 and bool(bad[5:8, 16:19].all())  # the 3x3 neighbourhood of pixel 137 = (6, 17)


# --------------------------------------------------------------------------------------------- 8-bit cross terms, K-tile-major planes
def _kt16(t, rows, K):   # row-major [rows, K] -> K-tile-major [K/32][rows][32] (include/vdn.h a_kt)
    return t.reshape(rows, K // 32, 32).permute(1, 0, 2).contiguous()


def _kt8(t, rows, K):    # byte plane [rows, K] -> [K/64][rows][64]
    return t.reshape(rows, K // 64, 64).permute(1, 0, 2).contiguous()


def _unkt16(t, rows, K):
    return t.reshape(K // 32, rows, 32).permute(1, 0, 2).reshape(rows, K)


def _unkt8(t, rows, K):
    return t.reshape(K // 64, rows, 64).permute(1, 0, 2).reshape(rows, K)


def _same_rows(a, b):
    """Two plane pairs of x6 rows agree on every byte that means something (24 code bytes + the scale byte per 32-byte half)."""
    a, b = a.reshape(-1, 32), b.reshape(-1, 32)
    assert torch.equal(a[:, :25], b[:, :25])


def _check_x6(p8, t, rows, K, order, kt=False):
    """What the planes stand for: plane 0 ~ hi within the e3m2 grid of its half (2 mantissa bits below the half's largest
    value: |err| <= max / 16 once saturation of (30, 32) is counted), plane 1 ~ lo on a grid 2^-10 finer."""
    from vdn import pack
    hi, lo = t.hi.float(), t.lo.float()
    d0, d1 = pack.decode6(p8[0], rows, K, order, kt), pack.decode6(p8[1], rows, K, order, kt)
    cols = pack.x6_columns(order).to(hi.device)
    for h in range(2):
        idx = (torch.arange(K // 64, device=hi.device)[:, None] * 64 + cols[h][None, :])          # [slabs, 32]
        mx = hi[:, idx].abs().amax(dim=-1, keepdim=True)                                           # [rows, slabs, 1]
        assert ((d0[:, idx] - hi[:, idx]).abs() <= mx / 8 + 1e-30).all()
        assert ((d0[:, idx] - hi[:, idx]).abs() <= 0.13 * hi[:, idx].abs() + mx / 256).all()       # 2 mantissa bits, subnormal step max / 256
        assert ((d1[:, idx] - lo[:, idx]).abs() <= 0.13 * lo[:, idx].abs() + mx / 1024 / 256 + 1e-30).all()


@pytest.mark.parametrize("bm", [0, 192, 256])
@pytest.mark.parametrize("M,N,K", [(2740, 1024, 1024), (1370 * 2 + 77, 3072, 384), (2048, 512, 4096), (4100, 1024, 64)])
def test_x8_gemm_k_tile_major_planes(rt3, M, N, K, bm, tune):
    """The cross-term kernel on K-tile-major operand planes ([K/32][rows][32] halves, [K/64][rows][64] bytes of 6-bit rows):
    weights through vdn_pack_x8, activations permuted on the host; bitwise equal to the same kernel on row-major planes (the
    arithmetic does not depend on the layout), fp64-close; the same product from planes in the other two stream orders
    (A and W packed alike); GELU output written K-tile-major with its planes of 6-bit rows and no fp16 lo plane, as fc1
    hands it to fc2, and fc2 consuming them."""
    from vdn import pack, _abi
    from vdn.runtime import HL
    tune(force_bm=bm)   # 0: the launch picks its M tile (192 or 256 rows); else pinned — the results must not depend on it
    a = rnd(M, K, seed=990)
    w = rnd(N, K, seed=991, scale=1 / math.sqrt(K))
    b, g = rnd(N, seed=992), rnd(N, seed=993)
    x = rnd(M, N, seed=994)
    ref = (x.double() + (a.double() @ w.double().t() + b.double()) * g.double()).float()
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    A8, W8 = pack.planes8(A), pack.planes8(W)
    xr = x.clone().to(DEV)
    rt3.gemm(A, W, M, N, K, out=xr, bias=b.to(DEV), gamma=g.to(DEV), res1=xr, a8=A8, w8=W8)      # row-major planes
    X = pack.X8(W)
    assert torch.equal(_unkt16(X.hi, N, K), W.hi) and torch.equal(_unkt8(X.p8[0], N, K), W8[0]) and torch.equal(_unkt8(X.p8[1], N, K), W8[1])
    Ak = HL(_kt16(A.hi, M, K))
    A8k = torch.stack([_kt8(A8[0], M, K), _kt8(A8[1], M, K)]).contiguous()
    xk = x.clone().to(DEV)
    rt3.gemm(Ak, HL(X.hi), M, N, K, out=xk, bias=b.to(DEV), gamma=g.to(DEV), res1=xk, a8=A8k, w8=X.p8, a_kt=True, w_kt=True)
    assert torch.equal(xk, xr)
    close(xk, ref, 2e-5)
    _check_x6(A8, A, M, K, pack.ORDER_NATURAL)
    for order in (pack.ORDER_GEMM, pack.ORDER_ATTN):   # any order, as long as both operands use it
        xo = x.clone().to(DEV)
        rt3.gemm(Ak, HL(X.hi), M, N, K, out=xo, bias=b.to(DEV), gamma=g.to(DEV), res1=xo, a8=pack.planes8(A, order, kt=True),
                 w8=pack.X8(W, order).p8, a_kt=True, w_kt=True)
        close(xo, ref, 2e-5)
    if N % 64 == 0:
        oh = HL(torch.zeros(M, N, dtype=torch.float16, device=DEV))
        o8 = torch.zeros(2, M, N, dtype=torch.uint8, device=DEV)
        rt3.gemm(Ak, HL(X.hi), M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU, a8=A8k, w8=X.p8, out8=o8, a_kt=True, w_kt=True, out_kt=True)
        refg = F.gelu(a.double() @ w.double().t() + b.double()).float()
        hi = _unkt16(oh.hi, M, N).float()
        close(hi + pack.decode6(o8[1], M, N, pack.ORDER_GEMM, kt=True), refg, 3e-5)   # hi + the 6-bit remainder: 2^-14 relative
        close(pack.decode6(o8[0], M, N, pack.ORDER_GEMM, kt=True), refg, 0.08)        # 2 mantissa bits
        # ... and the next GEMM consuming them (weights packed in the epilogue's order): fc1 -> fc2
        w2 = rnd(512, N, seed=995, scale=1 / math.sqrt(N))
        X2 = pack.X8(pack.linear(w2.to(DEV), rt3.prec), pack.ORDER_GEMM)
        y = torch.zeros(M, 512, device=DEV)
        rt3.gemm(oh, HL(X2.hi), M, 512, N, out=y, a8=o8, w8=X2.p8, a_kt=True, w_kt=True)
        close(y, (refg.double() @ w2.double().t()).float(), 6e-5)   # the input planes (2^-14) and the product's own cross terms


@pytest.mark.parametrize("bm", [192, 256])
def test_x8_heads_split_with_attention_planes(rt3, bm, tune):
    """The QKV head split on the cross-term kernel (paired epilogue): Q / K hi planes, V^T, and the attention's 8-bit planes
    (64 B e5m2(v) | 64 B e5m2(remainder 2^10) per token and head, natural channel order: the two lanes of a token swap halves and
    store 32-byte runs) — against the same launch on the three-product kernel and against fp64."""
    from vdn import pack, _abi
    from vdn.runtime import ceil_to
    tune(force_bm=bm)
    B, T, Hh, K = 3, 1370, 2, 128
    C, tp, M = Hh * 64, ceil_to(1370, 64), 3 * 1370
    a = rnd(M, K, seed=1230)
    w, b = rnd(3 * C, K, seed=1231, scale=1 / math.sqrt(K)), rnd(3 * C, seed=1232)
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    outs = []
    for x8 in (False, True):
        q, k = rt3.hbuf(f"t_hs_q{x8}", (B * Hh, tp, 64), zero=True), rt3.hbuf(f"t_hs_k{x8}", (B * Hh, tp, 64), zero=True)
        vt = rt3.hbuf(f"t_hs_v{x8}", (B * Hh, 64, tp), zero=True)
        q8, k8 = rt3.qk8(f"t_hs_q8{x8}", B * Hh, tp), rt3.qk8(f"t_hs_k8{x8}", B * Hh, tp)
        q8.zero_(); k8.zero_()
        kw = dict(a8=pack.planes8(A), w8=pack.planes8(W)) if x8 else {}
        rt3.gemm(A, W, M, 3 * C, K, bias=b.to(DEV), store=_abi.ST_HEADS,
                 heads=dict(dst=[q, k, vt], dst8=[q8, k8, None], transposed=[0, 0, 1], heads=Hh, tokens=T, tpad=tp), **kw)
        outs.append((q, k, vt, q8, k8))
    (q3, k3, v3, _, _), (q, k, vt, q8, k8) = outs
    close(q.float(), q3.float(), 2e-5)
    close(vt.float(), v3.float(), 2e-5)
    ref = (a.double() @ w.double().t() + b.double()).float().reshape(B, T, 3, Hh, 64)
    for got, idx in ((q, 0), (k, 1)):
        close(got.float().reshape(B, Hh, tp, 64)[:, :, :T], ref[:, :, idx].transpose(1, 2), 2e-5)
    close(vt.float().reshape(B, Hh, 64, tp)[..., :T], ref[:, :, 2].permute(0, 2, 3, 1), 2e-5)
    for p8, planes in ((q8, q), (k8, k)):
        want_hi = planes.float()[:, :T].to(torch.float8_e5m2).view(torch.uint8)
        want_lo = (planes.lo.float()[:, :T] * 1024.0).to(torch.float8_e5m2).view(torch.uint8) if planes.lo is not None else None
        got = p8[:, :T]
        assert (got[..., :64].int() - want_hi.int()).abs().max() <= 1 and (got[..., :64] != want_hi).float().mean() < 2e-3
        if want_lo is not None:
            assert torch.equal(got[..., 64:], want_lo)


@pytest.mark.parametrize("bm", [192, 256])
def test_x8_gated_and_plain_half_plane_stores(rt3, bm, tune):
    """The temporal module's LayerNorm-fed linears on the cross-term kernel: the gated (GEGLU) store and the plain split-plane
    store without bias, A from vdn_layernorm's K-tile-major planes with the position table added — against the three-product
    kernel on the same LayerNorm's split planes, and the gated one against fp64."""
    from vdn import pack, _abi
    from vdn.runtime import HL
    tune(force_bm=bm)
    M, c, T, D = 4160, 256, 4, 1040
    x = rnd(M, c, seed=1260).to(DEV)
    w, b = rnd(c, seed=1261).to(DEV), rnd(c, seed=1262).to(DEV)
    tab = rnd(T, c, seed=1263).to(DEV)
    n = rt3.hbuf("t_gp_n", (M, c))
    rt3.layernorm(x, M, c, w, b, 1e-5, out_h=n, addtab=tab, tab_div=D, tab_mod=T)
    n_k, n8 = HL(torch.zeros(M, c, dtype=torch.float16, device=DEV)), torch.zeros(2, M, c, dtype=torch.uint8, device=DEV)
    rt3.layernorm(x, M, c, w, b, 1e-5, out_h=n_k, out8=n8, kt=True, addtab=tab, tab_div=D, tab_mod=T)
    kt = dict(a8=n8, a_kt=True, w_kt=True)
    wq = pack.linear(rnd(3 * c, c, seed=1264, scale=1 / math.sqrt(c)).to(DEV), rt3.prec)
    q3, q8 = rt3.hbuf("t_gp_q3", (M, 3 * c)), rt3.hbuf("t_gp_q8", (M, 3 * c))
    rt3.gemm(n, wq, M, 3 * c, c, out=q3)
    Xq = pack.X8(wq)
    rt3.gemm(n_k, HL(Xq.hi), M, 3 * c, c, out=q8, w8=Xq.p8, **kt)
    close(q8.float(), q3.float(), 5e-5)   # K = 256: the 2^-14 cross terms average over fewer products than at K = 1024
    wgf, bgf = rnd(8 * c, c, seed=1265, scale=1 / math.sqrt(c)), rnd(8 * c, seed=1266)
    wg, bg = pack.geglu(wgf.to(DEV), bgf.to(DEV), rt3.prec)
    g3, g8 = rt3.hbuf("t_gp_g3", (M, 4 * c)), rt3.hbuf("t_gp_g8", (M, 4 * c))
    rt3.gemm(n, wg, M, 8 * c, c, bias=bg, store=_abi.ST_GEGLU, out=g3)
    Xg = pack.X8(wg)
    rt3.gemm(n_k, HL(Xg.hi), M, 8 * c, c, bias=bg, store=_abi.ST_GEGLU, out=g8, w8=Xg.p8, **kt)
    close(g8.float(), g3.float(), 6e-5)
    y = n.float().double().cpu() @ wgf.double().t() + bgf.double()
    close(g8.float(), (y[:, : 4 * c] * F.gelu(y[:, 4 * c:])).float(), 6e-5)


@pytest.mark.parametrize("bm", [192, 256])
def test_x8_heads_split_with_rope(rt3, bm, tune):
    """The RoPE'd QKV head split of the memory attention on the cross-term kernel (paired epilogue: a pair's real and imaginary
    columns sit in one lane, the 16 rotated channels are consecutive): Q / K planes, their 8-bit planes and V^T against the same
    launch on the three-product kernel (generic epilogue, itself fp64-checked in test_x3_heads_rope_and_flash)."""
    from vdn import pack, _abi
    from vdn.runtime import ceil_to
    tune(force_bm=bm)
    side, B, Hh, K = 37, 3, 2, 128
    P, C = side * side, Hh * 64
    tp, M = ceil_to(P, 64), B * P
    a = rnd(M, K, seed=1240)
    ws, bs = [rnd(C, K, seed=1241 + i, scale=1 / math.sqrt(K)) for i in range(3)], [rnd(C, seed=1245 + i) for i in range(3)]
    wp, bp = pack.cat_proj([w.to(DEV) for w in ws], [b.to(DEV) for b in bs], [1, 1, 0], rt3.prec)
    cs = pack.rope_table(side, side, 64, device=DEV)
    A = rt3.to_half(a.to(DEV))
    outs = []
    for x8 in (False, True):
        q, k = rt3.hbuf(f"t_hr_q{x8}", (B * Hh, tp, 64), zero=True), rt3.hbuf(f"t_hr_k{x8}", (B * Hh, tp, 64), zero=True)
        vt = rt3.hbuf(f"t_hr_v{x8}", (B * Hh, 64, tp), zero=True)
        q8, k8 = rt3.qk8(f"t_hr_q8{x8}", B * Hh, tp), rt3.qk8(f"t_hr_k8{x8}", B * Hh, tp)
        q8.zero_(); k8.zero_()
        kw = dict(a8=pack.planes8(A), w8=pack.planes8(wp)) if x8 else {}
        rt3.gemm(A, wp, M, 3 * C, K, bias=bp, store=_abi.ST_HEADS,
                 heads=dict(dst=[q, k, vt], dst8=[q8, k8, None], transposed=[0, 0, 1], rope=[1, 1, 0], rope_cs=cs, rope_mod=P, heads=Hh,
                            tokens=P, tpad=tp), **kw)
        outs.append((q, k, vt, q8, k8))
    (q3, k3, v3, _, _), (q, k, vt, q8, k8) = outs
    for got, want in ((q, q3), (k, k3), (vt, v3)):
        close(got.float(), want.float(), 2e-5)
    for p8, planes in ((q8, q), (k8, k)):
        want_hi = planes.float()[:, :P].to(torch.float8_e5m2).view(torch.uint8)
        got = p8[:, :P]
        assert (got[..., :64].int() - want_hi.int()).abs().max() <= 1 and (got[..., :64] != want_hi).float().mean() < 2e-3
        if planes.lo is not None:
            assert torch.equal(got[..., 64:], (planes.lo.float()[:, :P] * 1024.0).to(torch.float8_e5m2).view(torch.uint8))


@pytest.mark.parametrize("M,N,K", [(5500, 192, 128), (4200, 320, 1024), (4099, 448, 64)])
def test_x8_gemm_column_tail_tiles(rt3, M, N, K, tune):
    """Column counts that leave the last 256-wide tile partly empty (W rows clamped on load, columns masked on store), with
    ragged M: residual flavour and the bias + GELU flavour with its 6-bit output rows; both tile heights."""
    from vdn import pack, _abi
    from vdn.runtime import HL
    a, w = rnd(M, K, seed=1220), rnd(N, K, seed=1221, scale=1 / math.sqrt(K))
    b, x = rnd(N, seed=1222), rnd(M, N, seed=1223)
    ref = (x.double() + a.double() @ w.double().t() + b.double()).float()
    refg = F.gelu(a.double() @ w.double().t() + b.double()).float()
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    Ak, A8k, X = HL(_kt16(A.hi, M, K)), pack.planes8(A, kt=True), pack.X8(W)
    for bm in (192, 256):
        tune(force_bm=bm)
        xo = x.clone().to(DEV)
        rt3.gemm(Ak, HL(X.hi), M, N, K, out=xo, bias=b.to(DEV), res1=xo, a8=A8k, w8=X.p8, a_kt=True, w_kt=True)
        close(xo, ref, 2e-5)
        oh = HL(torch.zeros(M, N, dtype=torch.float16, device=DEV))
        o8 = torch.full((2, M, N), 0xA5, dtype=torch.uint8, device=DEV)
        rt3.gemm(Ak, HL(X.hi), M, N, K, out=oh, bias=b.to(DEV), act=_abi.ACT_GELU, a8=A8k, w8=X.p8, out8=o8, a_kt=True, w_kt=True, out_kt=True)
        close(_unkt16(oh.hi, M, N).float() + pack.decode6(o8[1], M, N, pack.ORDER_GEMM, kt=True), refg, 3e-5)


@pytest.mark.parametrize("terms", [1, 2])
def test_x8_gemm_dropped_cross_term(rt3, terms):
    """vdn_gemm_desc.x8_terms: 1 leaves out A_lo W_hi^T, 2 leaves out A_hi W_lo^T (the scale byte of that plane is zeroed in
    the kernel, no branch): the result is the full product minus exactly that term."""
    from vdn import pack
    M, N, K = 4100, 512, 1024
    a, w = rnd(M, K, seed=1210), rnd(N, K, seed=1211, scale=1 / math.sqrt(K))
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    ah, al, wh, wl = A.hi.double().cpu(), A.lo.double().cpu(), W.hi.double().cpu()[:, :K], W.lo.double().cpu()[:, :K]
    full = (ah + al) @ (wh + wl).t()
    dropped = al @ wh.t() if terms == 1 else ah @ wl.t()
    out = torch.zeros(M, N, device=DEV)
    rt3.gemm(A, W, M, N, K, out=out, a8=pack.planes8(A), w8=pack.planes8(W), x8_terms=terms)
    close(out, (full - dropped).float(), 3e-5)
    err_full = ((out.cpu().double() - full).norm() / full.norm()).item()
    assert 1e-4 < err_full < 1e-3, err_full          # the missing term is 2^-11-sized: visible, as the budget experiment found


def test_x8_gemm_outlier_channels_zero_rows_and_large_values(rt3):
    """The block-scaled 6-bit rows where their scale matters: activation channels 200x above their neighbours (the 31 other
    values of such a half fall into the subnormal codes: only their CROSS terms lose precision), all-zero rows, and values
    near the top of fp16 (6e4: above what the earlier e5m2 planes could carry): against fp64 and against the
    three-fp16-product kernel."""
    from vdn import pack
    M, N, K = 4100, 1024, 1024
    a = rnd(M, K, seed=1200)
    a[:, [5, 77, 300, 301, 1000]] *= 200.0
    a[17] = 0.0
    a[100:164] = 0.0
    a[200, 64:128] = 6.0e4 * torch.sign(a[200, 64:128])
    w = rnd(N, K, seed=1201, scale=1 / math.sqrt(K))
    w[:, 5] *= 30.0
    b = rnd(N, seed=1202)
    ref = (a.double() @ w.double().t() + b.double()).float()
    A, W = rt3.to_half(a.to(DEV)), pack.linear(w.to(DEV), rt3.prec)
    out8, out3 = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    rt3.gemm(A, W, M, N, K, out=out8, bias=b.to(DEV), a8=pack.planes8(A), w8=pack.planes8(W))
    rt3.gemm(A, W, M, N, K, out=out3, bias=b.to(DEV))
    close(out3, ref, 2e-5)
    close(out8, ref, 3e-5)
    assert torch.equal(out8[17].cpu(), b) and torch.equal(out8[100:164].cpu(), b.expand(64, N))   # zero rows: exactly the bias
    _check_x6(pack.planes8(A), A, M, K, pack.ORDER_NATURAL)


@pytest.mark.parametrize("rows,C", [(77, 384), (4100, 1024)])
def test_pack_x8_from_fp32(rt3, rows, C):
    """vdn_pack_x8_f32: an fp32 activation straight to the cross-term GEMM's A operand: the hi plane is the value toward zero,
    and the 6-bit rows are those vdn_pack_x8 makes of that split."""
    from vdn import pack
    from vdn.runtime import HL
    x = (rnd(rows, C, seed=1250) * 3.0).to(DEV)
    x[5] = 0.0
    x[:, 7] *= 120.0
    hi_kt = torch.zeros(rows, C, dtype=torch.float16, device=DEV)
    p8 = torch.zeros(2, rows, C, dtype=torch.uint8, device=DEV)
    rt3.pack_x8_f32(x, hi_kt, p8)
    hi = _unkt16(hi_kt, rows, C)
    assert (hi.float().abs() <= x.abs()).all() and ((x - hi.float()).abs() <= x.abs() * 2.0 ** -10 + 2.0 ** -24).all()   # toward zero, within an ulp (fp16 subnormals: 2^-24)
    split = HL(hi.contiguous(), (x - hi.float()).half())
    _same_rows(p8, pack.planes8(split, pack.ORDER_NATURAL, kt=True))
    _check_x6(p8, split, rows, C, pack.ORDER_NATURAL, kt=True)


@pytest.mark.parametrize("rows,C", [(77, 384), (1370, 1024), (300, 64)])
def test_layernorm_8bit_planes_and_k_tile_major(rt3, rows, C):
    """vdn_layernorm(out8, kt): the hi plane + the planes of 6-bit rows of the cross-term GEMM's A operand, row-major and
    K-tile-major, against the plain split-plane output of the same launch pushed through vdn_pack_x8."""
    from vdn import pack
    from vdn.runtime import HL
    x = rnd(rows, C, seed=995).to(DEV)
    w, b = rnd(C, seed=996).to(DEV), rnd(C, seed=997).to(DEV)
    ref = rt3.hbuf(f"t_ln8_ref_{rows}_{C}", (rows, C))
    rt3.layernorm(x, rows, C, w, b, 1e-6, out_h=ref)
    for kt in (False, True):
        oh = HL(torch.zeros(rows, C, dtype=torch.float16, device=DEV))
        o8 = torch.zeros(2, rows, C, dtype=torch.uint8, device=DEV)
        rt3.layernorm(x, rows, C, w, b, 1e-6, out_h=oh, out8=o8, kt=kt)
        hi = _unkt16(oh.hi, rows, C) if kt else oh.hi
        assert torch.equal(hi, ref.hi)
        _same_rows(o8, pack.planes8(ref, pack.ORDER_NATURAL, kt=kt))
        _check_x6(o8, ref, rows, C, pack.ORDER_NATURAL, kt=kt)


def test_flash_attention_8bit_output_planes_k_tile_major(rt3):
    """vdn_flash_attn(out8, out_kt): the attention output as the proj GEMM's K-tile-major A planes == the row-major planes
    of the same launch, permuted; the planes of 6-bit rows == vdn_pack_x8 (attention order) on the split output."""
    from vdn import pack, _abi
    from vdn.runtime import HL
    B, Hh, T = 2, 3, 150
    C, tp = Hh * 64, 192
    qkv = rnd(B * T, 3 * C, seed=998)
    q, k = rt3.hbuf("t_a8_q", (B * Hh, tp, 64), zero=True), rt3.hbuf("t_a8_k", (B * Hh, tp, 64), zero=True)
    vt = rt3.hbuf("t_a8_v", (B * Hh, 64, tp), zero=True)
    q8, k8 = rt3.qk8("t_a8_q8", B * Hh, tp), rt3.qk8("t_a8_k8", B * Hh, tp)
    w = torch.eye(3 * C)
    rt3.gemm(rt3.to_half(qkv.to(DEV)), pack.linear(w.to(DEV), rt3.prec), B * T, 3 * C, 3 * C, store=_abi.ST_HEADS,
             heads=dict(dst=[q, k, vt], dst8=[q8, k8, None], transposed=[0, 0, 1], heads=Hh, tokens=T, tpad=tp))
    ref = rt3.hbuf("t_a8_ref", (B * T, C))
    rt3.flash_attn(q, k, vt, ref, B, Hh, T, tp, T, tp, 0.125, q8=q8, k8=k8)
    oh = HL(torch.zeros(B * T, C, dtype=torch.float16, device=DEV))
    o8 = torch.zeros(2, B * T, C, dtype=torch.uint8, device=DEV)
    rt3.flash_attn(q, k, vt, oh, B, Hh, T, tp, T, tp, 0.125, q8=q8, k8=k8, out8=o8, out_kt=True)
    assert torch.equal(_unkt16(oh.hi, B * T, C), ref.hi)
    _same_rows(o8, pack.planes8(ref, pack.ORDER_ATTN, kt=True))
    _check_x6(o8, ref, B * T, C, pack.ORDER_ATTN, kt=True)
    o8r = torch.zeros(2, B * T, C, dtype=torch.uint8, device=DEV)    # row-major planes
    ohr = HL(torch.zeros(B * T, C, dtype=torch.float16, device=DEV))
    rt3.flash_attn(q, k, vt, ohr, B, Hh, T, tp, T, tp, 0.125, q8=q8, k8=k8, out8=o8r)
    _same_rows(o8r, pack.planes8(ref, pack.ORDER_ATTN))
