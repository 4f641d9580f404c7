"""Drop-in for depth_anything_v2/depth_anything_v2.py:12-92 — same class name, constructor, methods and
state-dict keys; the forward pass runs on libvdn_hip.so (MI355X), never on torch ops."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import modules, util
from .engine import DPTEngine, EncoderEngine, MemoryEngine, ReadoutEngine
from .runtime import Runtime

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


PRECISIONS = {"f16x3": (torch.float16, True), "f16": (torch.float16, False),
              "bf16x3": (torch.bfloat16, True), "bf16": (torch.bfloat16, False)}
DEFAULT_PRECISION = "f16x3"


def _precision(name=None):
    """MFMA operand precision:
      f16x3 (default)  every 16-bit operand carried as hi+lo planes, 3 MFMA products per term —
                       fp32-faithful, meets the 1e-3 parity bar on every fixture;
      f16 / bf16       single product, full MFMA rate; ~1e-3..2.5e-3 from the fp32 reference."""
    name = (name or os.environ.get("VDN_PRECISION", DEFAULT_PRECISION)).lower()
    if name not in PRECISIONS:
        raise ValueError(f"VDN_PRECISION must be one of {sorted(PRECISIONS)}, got {name!r}")
    return name, PRECISIONS[name]


class _EngineOwner(nn.Module):
    """Shared plumbing: engines are (re)built lazily from the current parameters and device."""

    def __init__(self):
        super().__init__()
        self._eng = None
        self.precision = None  # None -> $VDN_PRECISION -> DEFAULT_PRECISION

    def set_precision(self, name: str):
        _precision(name)
        self.precision, self._eng = name, None
        return self

    def set_attention_pv(self, n: int):
        """MFMA products per P V term of the attention (include/vdn.h vdn_flash_attn pv_products): 1 (default; $VDN_ATTN_PV),
        2 or 3. A per-model setting handed to every attention launch; the engines are rebuilt (a memory bank keeps the V
        planes its producer wrote)."""
        assert n in (1, 2, 3), n
        self.attention_pv, self._eng = n, None
        return self

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._eng = None
        return super().load_state_dict(*a, **k)

    def _runtime(self) -> Runtime:
        dev = next(self.parameters()).device
        _, (dtype, split) = _precision(self.precision)
        rt = Runtime(dev, dtype, split)
        if getattr(self, "attention_pv", None):
            rt.pv_products = self.attention_pv
        return rt

    @staticmethod
    def preprocess(rt: Runtime, frames_u8: torch.Tensor, input_size: int, swap_rb: bool = False) -> torch.Tensor:
        """u8 [n,h,w,3] on the device (RGB, or BGR with swap_rb) -> f32 [n,3,H,W]: /255, Resize(lower_bound, multiple of 14,
        cubic) + NormalizeImage + PrepareForNet (util/transform.py:109-148) in ONE launch (vdn_preprocess). The cubic kernel
        is the A=-0.75 half-pixel one cv2.INTER_CUBIC uses; cv2 itself is absent offline so this step is parity-unpinned
        (SURVEY.md §8c)."""
        n, h, w, _ = frames_u8.shape
        nw, nh = util.get_size(w, h, input_size)
        return rt.preprocess_u8(frames_u8.contiguous(), nh, nw, _MEAN, _STD, swap_rb)


class DepthAnythingV2(_EngineOwner):
    def __init__(self, encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], use_bn=False,
                 use_clstoken=False, max_memory_length=6):
        super().__init__()
        self.intermediate_layer_idx = {k: v["taps"] for k, v in modules.ENCODERS.items()}
        self.encoder = encoder
        cfg = modules.ENCODERS[encoder]
        self.max_memory_length = max_memory_length
        self.pretrained = modules.dinov2(encoder)
        self.memory_block = modules.memory_block(cfg["dim"], max_memory_length, 4)
        self.depth_head = modules.dpt_head(cfg["dim"], features, out_channels, use_bn, use_clstoken)
        self._features, self._out_channels = features, list(out_channels)

    def _engines(self):
        if self._eng is None:
            rt = self._runtime()
            cfg = modules.ENCODERS[self.encoder]
            self._eng = dict(
                rt=rt, enc=EncoderEngine(rt, self.pretrained, cfg),
                mem=MemoryEngine(rt, self.memory_block, cfg["dim"], self.max_memory_length),
                head=DPTEngine(rt, self.depth_head, cfg["dim"], self._features, self._out_channels, temporal=False))
            if hasattr(self.depth_head, "readout_projects"):   # use_clstoken
                self._eng["enc"].readout = ReadoutEngine(rt, self.depth_head.readout_projects, cfg["dim"])
            self._lanes = None
        return self._eng

    def _stream_lanes(self, n: int):
        """`n` independent execution lanes = HIP stream + own workspace arena, all sharing the packed weights and
        ONE memory-bank ring (lane i owns batch rows [i B/n, (i+1) B/n) of it, MemoryEngine.lane). The batch elements
        of path A are independent video streams, so the batch is dealt to the lanes and their kernels interleave on
        the GPU: one lane's epilogues and tile-quantisation tails overlap the other's main loops (measured +6 % at
        batch 8, 2 lanes)."""
        import copy
        e = self._engines()
        if self._lanes is None or len(self._lanes) != n:
            lanes = []
            for i in range(n):
                rt = e["rt"] if i == 0 else Runtime(e["rt"].device, e["rt"].half, e["rt"].split)
                rt.pv_products = e["rt"].pv_products
                enc, mem, head = (copy.copy(e[k]) for k in ("enc", "mem", "head"))
                enc.rt = mem.rt = head.rt = rt
                if getattr(enc, "readout", None) is not None:
                    enc.readout = copy.copy(enc.readout)
                    enc.readout.rt = rt
                mem.lane, mem._nomem = (i, n), {}   # bank state, ring and RoPE table stay shared (built in prepare())
                if i > 0:  # lazily built per-lane tables are built on the lane's own stream
                    enc._pos_cache = {}
                lanes.append(dict(rt=rt, enc=enc, mem=mem, head=head, stream=torch.cuda.Stream(device=rt.device)))
            self._lanes = lanes
        return self._lanes

    def clear_memory(self):
        if self._eng is not None:
            self._eng["mem"].clear()

    def _forward_lane(self, ln, x, _pre_relu, out):
        rt, enc, mem, head = ln["rt"], ln["enc"], ln["mem"], ln["head"]
        B = x.shape[0]
        taps, last_f32, (ph, pw) = enc.run(x, want_f32_last=True)
        fm = mem.forward(last_f32, B, ph * pw)
        t3 = fm   # the memory bank is updated with fm itself (depth_anything_v2.py:52-54); the readout happens inside the head
        if getattr(enc, "readout", None) is not None:   # use_clstoken: the memory output takes the last tap's readout (dpt.py:119-123)
            t3 = enc.readout.apply(3, fm, enc.cls_last, B, ph * pw, rt.hbuf("ro_fm", (B * ph * pw, enc.C)))
        head.run([taps[0], taps[1], taps[2], t3], B, ph, pw, relu=not _pre_relu, out=out)   # the depth tail writes this lane's rows of the result
        mem.update(fm, out.clamp(min=0) if _pre_relu else out, B, ph, pw)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, _pre_relu: bool = False) -> torch.Tensor:
        """x f32 [B,3,H,W] (H,W multiples of 14, square) -> f32 [B,H,W]; mutates the memory bank
        (depth_anything_v2.py:45-55). `_pre_relu` (tests only) returns the signed map before the
        final ReLUs; the memory update always sees the ReLU'd depth as in the reference."""
        e = self._engines()
        rt = e["rt"]
        x = x.to(device=rt.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        e["mem"].prepare(B, (H // 14) * (W // 14))
        out = torch.empty((B, H, W), dtype=torch.float32, device=rt.device)   # the one allocation of a forward: its result
        nl = int(os.environ.get("VDN_STREAMS", "2"))
        if nl < 2 or B < int(os.environ.get("VDN_LANE_MIN_BATCH", "4")) or B % nl:
            rt.cu_hint = 0
            self._forward_lane(dict(rt=rt, enc=e["enc"], mem=e["mem"], head=e["head"]), x, _pre_relu, out)
            e["mem"].commit()
            return out
        lanes = self._stream_lanes(nl)
        for ln in lanes:
            ln["rt"].cu_hint = 256 // nl  # each lane's GEMM tiles are sized for its share of the CUs
        cur = torch.cuda.current_stream(rt.device)
        for ln, xs, os_ in zip(lanes, x.chunk(nl), out.chunk(nl)):
            ln["stream"].wait_stream(cur)
            with torch.cuda.stream(ln["stream"]):
                self._forward_lane(ln, xs.contiguous(), _pre_relu, os_)
        for ln in lanes:
            cur.wait_stream(ln["stream"])
        e["mem"].commit()
        return out

    @torch.no_grad()
    def infer_image(self, raw_image: np.ndarray, input_size: int = 518) -> np.ndarray:
        """BGR u8 [h,w,3] -> f32 [h,w] (depth_anything_v2.py:57-65)."""
        image, (h, w) = self.image2tensor(raw_image, input_size)
        depth = self.forward(image)
        rt = self._engines()["rt"]
        if tuple(depth.shape[-2:]) != (h, w):
            out = torch.empty((1, h, w), dtype=torch.float32, device=rt.device)
            rt.upsample_f32(depth.contiguous(), out, 1, depth.shape[-2], depth.shape[-1], h, w)
            depth = out
        from .util import check_finite
        check_finite(depth, "infer_image")
        return depth[0].cpu().numpy()

    def image2tensor(self, raw_image: np.ndarray, input_size: int = 518):
        """depth_anything_v2.py:67-92."""
        rt = self._engines()["rt"]
        h, w = raw_image.shape[:2]
        img = torch.from_numpy(np.ascontiguousarray(raw_image)).to(rt.device)   # BGR u8: the kernel swaps the channels
        return self.preprocess(rt, img[None], input_size, swap_rb=True), (h, w)
