"""Depth-refiner wrappers: drop-ins for models/video_depth_model_v4.py:83-148 and
models/video_depth_model_v5.py:124-192 — same constructor, `forward(input_depth)` and state-dict keys
(`pretrained.*`, `scale_head.feat.1.*`, `temporal_head.*`, `shift_head.0.*`). The network itself is the same
DINOv2 encoder + temporal DPT head as `vdn.VideoDepthAnything`; what the wrappers add runs in
csrc/refine.hip: per-frame median (exact radix select), tanh/exp scale, Sobel normals, scalar shift + residual."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import modules
from .depth_anything_v2 import _EngineOwner
from .engine import DPTEngine, EncoderEngine, ReadoutEngine


class _DepthRefiner(_EngineOwner):
    VERSION = 5  # 5: the network sees a 224x224 bilinear resize of the clip; 4: the clip itself (H, W multiples of 14)

    def __init__(self, encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], use_bn=False, use_clstoken=False,
                 num_frames=32, max_depth=65535, pe="ape", use_residual=True, input_normal=True):
        super().__init__()
        if pe != "ape":
            raise NotImplementedError("pe='rope' is not enabled by any configuration the reference ships")
        if encoder not in ("vits", "vitl"):
            raise KeyError(encoder)
        self.intermediate_layer_idx = {"vits": [2, 5, 8, 11], "vitl": [4, 11, 17, 23]}
        self.max_depth, self.use_residual, self.input_normal = max_depth, use_residual, input_normal
        self.encoder = encoder
        cfg = modules.ENCODERS[encoder]
        self.pretrained = modules.dinov2(encoder)
        self.scale_head = modules.Holder()
        self.scale_head.feat = nn.Sequential(nn.Identity(), modules.Conv(1, 1, 1))  # quantile pool has no weights
        self.temporal_head = modules.dpt_head_temporal(cfg["dim"], features, out_channels, num_frames, use_bn, use_clstoken)
        self.shift_head = nn.Sequential(modules.Conv(1, 1, 1))
        self._features, self._out_channels = features, list(out_channels)

    def _engines(self):
        if self._eng is None:
            rt = self._runtime()
            cfg = modules.ENCODERS[self.encoder]
            sc, sh = self.scale_head.feat[1], self.shift_head[0]
            self._eng = dict(rt=rt, enc=EncoderEngine(rt, self.pretrained, cfg),
                             head=DPTEngine(rt, self.temporal_head, cfg["dim"], self._features, self._out_channels, temporal=True),
                             scale_wb=(float(sc.weight.reshape(()).item()), float(sc.bias.reshape(()).item())),
                             shift_wb=(float(sh.weight.reshape(()).item()), float(sh.bias.reshape(()).item())))
            if hasattr(self.temporal_head, "readout_projects"):   # use_clstoken
                self._eng["enc"].readout = ReadoutEngine(rt, self.temporal_head.readout_projects, cfg["dim"])
        return self._eng

    @torch.no_grad()
    def forward(self, input_depth: torch.Tensor) -> torch.Tensor:
        """input_depth f32 [B,S,H,W] in [0, max_depth] -> refined depth [B,S,H,W] (v5:160-192 / v4:117-148)."""
        e = self._engines()
        rt, enc, head = e["rt"], e["enc"], e["head"]
        B, S, H0, W0 = input_depth.shape
        F = B * S
        x = input_depth.to(device=rt.device, dtype=torch.float32).reshape(F, H0, W0).contiguous()
        med = torch.empty(F, dtype=torch.float32, device=rt.device)
        rt.frame_median(x, med)
        scaled = torch.empty_like(x)
        rt.refine_scale(x, med, e["scale_wb"][0], e["scale_wb"][1], 1.0, float(self.max_depth), scaled)
        if self.VERSION == 5:
            H = W = 224
            r = torch.empty((F, H, W), dtype=torch.float32, device=rt.device)
            rt.upsample_f32(scaled, r, F, H0, W0, H, W)
        else:
            if H0 % 14 or W0 % 14:
                raise AssertionError(f"input resolution {H0}x{W0} must be a multiple of the patch size 14")  # patch_embed.py:73-74
            H, W, r = H0, W0, scaled
        net_in = torch.empty((F, 3, H, W), dtype=torch.float32, device=rt.device)
        rt.refine_pack(r, net_in, normals=self.input_normal)
        taps, _, (ph, pw) = enc.run(net_in)
        depth = head.run(taps, F, ph, pw, T=S, relu=True).reshape(F, H, W)  # rectified before the resize, as the head does
        if (H, W) != (H0, W0):
            d0 = torch.empty((F, H0, W0), dtype=torch.float32, device=rt.device)
            rt.upsample_f32(depth.contiguous(), d0, F, H, W, H0, W0, relu=True)
        else:
            d0 = depth.contiguous()
        out = torch.empty((F, H0, W0), dtype=torch.float32, device=rt.device)
        rt.refine_finish(scaled, d0, e["shift_wb"][0], e["shift_wb"][1], float(self.max_depth), self.use_residual, out)
        return out.reshape(B, S, H0, W0)
