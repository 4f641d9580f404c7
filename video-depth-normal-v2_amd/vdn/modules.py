"""Parameter containers with the reference's module tree and state-dict keys.

These nn.Modules only HOLD weights (so `load_state_dict(torch.load(ckpt))`, `.to(device)`,
`.state_dict()` behave as for the reference); they never run torch compute. The forward pass lives
in vdn/engine.py and runs on libvdn_hip.so. tests/test_host.py checks the key/shape schema
against the one dumped from the imported reference (tests/golden/schema_*.json).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import pack

ENCODERS = {  # depth_anything_v2/dinov2.py:339-378; taps depth_anything_v2.py:24-29
    "vits": dict(dim=384, depth=12, heads=6, taps=[2, 5, 8, 11]),
    "vitb": dict(dim=768, depth=12, heads=12, taps=[2, 5, 8, 11]),
    "vitl": dict(dim=1024, depth=24, heads=16, taps=[4, 11, 17, 23]),
    "vitg": dict(dim=1536, depth=40, heads=24, taps=[9, 19, 29, 39], swiglu=4096),  # dinov2.py:381-395; FFN = SwiGLU, hidden (int(4C 2/3)+7)//8*8
}
MODEL_CONFIGS = {  # run_video.py:28-33
    "vits": {"encoder": "vits", "features": 64, "out_channels": [48, 96, 192, 384]},
    "vitb": {"encoder": "vitb", "features": 128, "out_channels": [96, 192, 384, 768]},
    "vitl": {"encoder": "vitl", "features": 256, "out_channels": [256, 512, 1024, 1024]},
    "vitg": {"encoder": "vitg", "features": 384, "out_channels": [1536, 1536, 1536, 1536]},
}


class Holder(nn.Module):
    """Empty container (child modules / parameters are attached by name)."""


def _param(*shape):
    return nn.Parameter(torch.zeros(*shape), requires_grad=False)


class Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = _param(o, i)
        if bias:
            self.bias = _param(o)
        else:
            self.register_parameter("bias", None)


class Conv(nn.Module):
    def __init__(self, i, o, k, bias=True, groups=1, transpose=False):
        super().__init__()
        self.weight = _param(i, o, k, k) if transpose else _param(o, i // groups, k, k)
        if bias:
            self.bias = _param(o)
        else:
            self.register_parameter("bias", None)


class Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = _param(c)
        self.bias = _param(c)


class BatchNorm(nn.Module):
    """nn.BatchNorm2d's parameters and buffers (the use_bn head): folded into the preceding convolution at engine build."""

    def __init__(self, c):
        super().__init__()
        self.weight = _param(c)
        self.bias = _param(c)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Gamma(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gamma = _param(c)


def dinov2(encoder: str) -> nn.Module:
    cfg = ENCODERS[encoder]
    C, depth = cfg["dim"], cfg["depth"]
    m = Holder()
    m.embed_dim = C
    m.cls_token = _param(1, 1, C)
    m.pos_embed = _param(1, 37 * 37 + 1, C)
    m.mask_token = _param(1, C)
    m.patch_embed = Holder()
    m.patch_embed.proj = Conv(3, C, 14)
    blocks = []
    for _ in range(depth):
        b = Holder()
        b.norm1 = Norm(C)
        b.attn = Holder()
        b.attn.qkv = Lin(C, 3 * C)
        b.attn.proj = Lin(C, C)
        b.ls1 = Gamma(C)
        b.norm2 = Norm(C)
        b.mlp = Holder()
        if cfg.get("swiglu"):  # dinov2_layers/swiglu_ffn.py:13-33
            b.mlp.w12 = Lin(C, 2 * cfg["swiglu"])
            b.mlp.w3 = Lin(cfg["swiglu"], C)
        else:
            b.mlp.fc1 = Lin(C, 4 * C)
            b.mlp.fc2 = Lin(4 * C, C)
        b.ls2 = Gamma(C)
        blocks.append(b)
    m.blocks = nn.ModuleList(blocks)
    m.norm = Norm(C)
    return m


def _rcu(f, bn=False):
    r = Holder()
    r.conv1 = Conv(f, f, 3)
    r.conv2 = Conv(f, f, 3)
    if bn:  # util/blocks.py:49-51
        r.bn1 = BatchNorm(f)
        r.bn2 = BatchNorm(f)
    return r


def _fusion(f, bn=False):
    b = Holder()
    b.out_conv = Conv(f, f, 1)
    b.resConfUnit1 = _rcu(f, bn)
    b.resConfUnit2 = _rcu(f, bn)
    return b


def dpt_head(in_channels: int, features: int, out_channels, use_bn: bool = False, use_clstoken: bool = False) -> nn.Module:
    h = Holder()
    h.projects = nn.ModuleList([Conv(in_channels, oc, 1) for oc in out_channels])
    if use_clstoken:  # dpt.py:81-88: Linear(2C -> C) + GELU per tap
        h.readout_projects = nn.ModuleList([nn.Sequential(Lin(2 * in_channels, in_channels), nn.Identity()) for _ in out_channels])
    h.resize_layers = nn.ModuleList([
        Conv(out_channels[0], out_channels[0], 4, transpose=True),
        Conv(out_channels[1], out_channels[1], 2, transpose=True),
        nn.Identity(),
        Conv(out_channels[3], out_channels[3], 3),
    ])
    s = Holder()
    for i in range(4):
        setattr(s, f"layer{i + 1}_rn", Conv(out_channels[i], features, 3, bias=False))
    for i in range(1, 5):
        setattr(s, f"refinenet{i}", _fusion(features, use_bn))
    s.output_conv1 = Conv(features, features // 2, 3)
    s.output_conv2 = nn.Sequential(Conv(features // 2, 32, 3), nn.Identity(), Conv(32, 1, 1), nn.Identity(), nn.Identity())
    h.scratch = s
    return h


def _temporal_attention(c, max_len, pe_type="ape"):
    a = Holder()
    a.max_len = max_len
    if pe_type == "ape":    # 'rope' (motion_module.py:236-240) has no buffer in the state dict: freqs_cis is a plain attribute there
        pe = Holder()
        pe.register_buffer("pe", pack.temporal_pe(c, max_len))
        a.pos_encoder = pe
    a.to_q = Lin(c, c, bias=False)
    a.to_k = Lin(c, c, bias=False)
    a.to_v = Lin(c, c, bias=False)
    a.to_out = nn.ModuleList([Lin(c, c), nn.Identity()])
    return a


def temporal_module(c: int, max_len: int, pe_type: str = "ape") -> nn.Module:
    t = Holder()
    tt = Holder()
    tt.norm = Norm(c)
    tt.proj_in = Lin(c, c)
    blk = Holder()
    blk.attention_blocks = nn.ModuleList([_temporal_attention(c, max_len, pe_type) for _ in range(2)])
    blk.norms = nn.ModuleList([Norm(c) for _ in range(2)])
    ff = Holder()
    g = Holder()
    g.proj = Lin(c, 8 * c)
    ff.net = nn.ModuleList([g, nn.Identity(), Lin(4 * c, c)])
    blk.ff = ff
    blk.ff_norm = Norm(c)
    tt.transformer_blocks = nn.ModuleList([blk])
    tt.proj_out = Lin(c, c)
    t.temporal_transformer = tt
    return t


def dpt_head_temporal(in_channels, features, out_channels, num_frames, use_bn: bool = False, use_clstoken: bool = False,
                      pe: str = "ape") -> nn.Module:
    h = dpt_head(in_channels, features, out_channels, use_bn, use_clstoken)
    h.motion_modules = nn.ModuleList([
        temporal_module(out_channels[2], num_frames, pe), temporal_module(out_channels[3], num_frames, pe),
        temporal_module(features, num_frames, pe), temporal_module(features, num_frames, pe)])
    return h


def _rope_attention(c):
    a = Holder()
    a.q_proj = Lin(c, c)
    a.k_proj = Lin(c, c)
    a.v_proj = Lin(c, c)
    a.out_proj = Lin(c, c)
    return a


def _mask_downsampler(cmid, k):
    d = Holder()
    d.encoder = nn.Sequential(Conv(1, cmid, k), Norm(cmid), nn.Identity(), Conv(cmid, 1, 1))
    return d


def memory_block(c: int, max_len: int, layers: int) -> nn.Module:
    m = Holder()
    ma = Holder()
    ls = []
    for _ in range(layers):
        l = Holder()
        l.self_attn = _rope_attention(c)
        l.cross_attn_image = _rope_attention(c)
        l.linear1 = Lin(c, 2 * c)
        l.linear2 = Lin(2 * c, c)
        l.norm1, l.norm2, l.norm3 = Norm(c), Norm(c), Norm(c)
        ls.append(l)
    ma.layers = nn.ModuleList(ls)
    ma.norm = Norm(c)
    m.memory_attention = ma
    m.curr_pos_enc = _param(1, 1, c)
    m.maskmem_tpos_enc = _param(1, max_len, c)
    m.no_mem_embed = _param(1, 1, c)
    me = Holder()
    me.mask_downsampler = nn.Sequential(_mask_downsampler(4, 3), _mask_downsampler(49, 7))
    me.pix_feat_proj = Conv(c, c, 1)
    fu = Holder()
    cx = []
    for _ in range(2):
        b = Holder()
        b.dwconv = Conv(c, c, 7, groups=c)
        b.norm = Norm(c)
        b.pwconv1 = Lin(c, 4 * c)
        b.pwconv2 = Lin(4 * c, c)
        b.gamma = _param(c)
        cx.append(b)
    fu.layers = nn.ModuleList(cx)
    me.fuser = fu
    m.memory_encoder = me
    return m
