"""Drop-in for video_depth_anything/video_depth.py:35-156 (VideoDepthAnything) on libvdn_hip.so."""
from __future__ import annotations

import numpy as np
import torch

from . import modules, util
from .depth_anything_v2 import _EngineOwner
from .engine import DPTEngine, EncoderEngine, ReadoutEngine

INFER_LEN, OVERLAP, KEYFRAMES, INTERP_LEN = util.INFER_LEN, util.OVERLAP, util.KEYFRAMES, util.INTERP_LEN


class VideoDepthAnything(_EngineOwner):
    def __init__(self, encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], use_bn=False,
                 use_clstoken=False, num_frames=32, pe="ape"):
        super().__init__()
        if pe not in ("ape", "rope"):
            raise NotImplementedError(pe)   # motion_module.py:242
        if encoder not in ("vits", "vitl"):
            raise KeyError(encoder)  # video_depth.py:48-51
        self.intermediate_layer_idx = {"vits": [2, 5, 8, 11], "vitl": [4, 11, 17, 23]}
        self.encoder = encoder
        cfg = modules.ENCODERS[encoder]
        self.pretrained = modules.dinov2(encoder)
        self.head = modules.dpt_head_temporal(cfg["dim"], features, out_channels, num_frames, use_bn, use_clstoken, pe)
        self._features, self._out_channels = features, list(out_channels)

    def _engines(self):
        if self._eng is None:
            rt = self._runtime()
            cfg = modules.ENCODERS[self.encoder]
            self._eng = dict(rt=rt, enc=EncoderEngine(rt, self.pretrained, cfg),
                             head=DPTEngine(rt, self.head, cfg["dim"], self._features, self._out_channels, temporal=True))
            if hasattr(self.head, "readout_projects"):   # use_clstoken: per-frame, so it is applied to every tap right after the encoder
                self._eng["enc"].readout = ReadoutEngine(rt, self.head.readout_projects, cfg["dim"])
        return self._eng

    @torch.no_grad()
    def forward(self, x: torch.Tensor, _pre_relu: bool = False) -> torch.Tensor:
        """x [B,T,3,H,W] -> [B,T,H,W] (video_depth.py:58-65). The final resize to (H,W) is the identity
        because H = 14*ph, so it is skipped."""
        e = self._engines()
        rt, enc, head = e["rt"], e["enc"], e["head"]
        B, T, _, H, W = x.shape
        xf = x.to(device=rt.device, dtype=torch.float32).reshape(B * T, 3, H, W).contiguous()
        taps, _, (ph, pw) = enc.run(xf)
        depth = head.run(taps, B * T, ph, pw, T=T, relu=not _pre_relu)
        return depth.reshape(B, T, H, W).clone()

    # ------------------------------------------------------------------ streaming (video_depth_stream.py)
    def reset_stream(self):
        self._stream = None

    @torch.no_grad()
    def stream_step(self, x: torch.Tensor, _pre_relu: bool = False) -> torch.Tensor:
        """One pre-processed frame x [1,1,3,H,W] -> depth [H,W], with the reference's cache bookkeeping
        (video_depth_stream.py:117-118,133-158: slot 0 pinned, 31 cached frames per step, gap 41)."""
        e = self._engines()
        rt, enc, head = e["rt"], e["enc"], e["head"]
        st = getattr(self, "_stream", None)
        if st is None:
            # cache / ids: the reference's lists (one entry per cached frame), here holding RING SLOT numbers: a frame's 8
            # projection sets (2 attention blocks x 4 temporal modules) live in slot k of 8 fixed rings (TemporalEngine)
            st = self._stream = dict(cache=[], ids=[], id=-1, gap=(INFER_LEN - OVERLAP) * 2 - 1 - (OVERLAP - INTERP_LEN),
                                     free=list(range(head.temporal[0].STREAM_SLOTS)))
        st["id"] += 1
        _, _, _, H, W = x.shape
        xf = x.to(device=rt.device, dtype=torch.float32).reshape(1, 3, H, W).contiguous()
        taps, _, (ph, pw) = enc.run(xf)
        new = st["free"].pop(0)
        if not st["cache"]:
            depth = head.run(taps, 1, ph, pw, T=1, relu=not _pre_relu, stream=dict(window=[], new=new))
            st["cache"] = [new] * INFER_LEN
            st["ids"].extend([0] * (INFER_LEN - 1))
        else:
            cur = st["cache"][0:2] + st["cache"][-INFER_LEN + 3:]
            depth = head.run(taps, 1, ph, pw, T=1, relu=not _pre_relu, stream=dict(window=cur, new=new))
            st["cache"].append(new)
        st["ids"].append(st["id"])
        if st["id"] + INFER_LEN > st["gap"] + 1:
            del st["ids"][1]
            gone = st["cache"].pop(1)
            if gone not in st["cache"]:  # the first frame's slot is referenced 32 times at the start
                st["free"].append(gone)
        return depth.reshape(H, W).clone()

    @torch.no_grad()
    def infer_video_depth_one(self, frame: np.ndarray, input_size: int = 518, device: str = "cuda", fp32: bool = False):
        """RGB u8 [h,w,3] -> f32 [h,w] (video_depth_stream.py:76-160)."""
        fh, fw = frame.shape[:2]
        ratio = max(fh, fw) / min(fh, fw)
        if ratio > 1.78:
            input_size = int(input_size * 1.777 / ratio)
            input_size = round(input_size / 14) * 14
        x = self.preprocess_frames(frame[None], input_size)[None]
        d = self.resize_depth(self.stream_step(x)[None], fh, fw)
        util.check_finite(d, "infer_video_depth_one")
        return d[0].cpu().numpy()

    @torch.no_grad()
    def forward_sharded(self, x_local: torch.Tensor, group=None, _pre_relu: bool = False) -> torch.Tensor:
        """One window whose T frames are sharded over the ranks of `group` (this rank passes its
        T/P consecutive frames, [1, T/P, 3, H, W]) -> depth of the local frames [1, T/P, H, W].
        Per-frame work stays local; each temporal module re-shards frames<->pixels with an
        all-to-all over RCCL (vdn/dist.py::FrameShardExchange)."""
        from .dist import FrameShardExchange, world
        e = self._engines()
        rt, enc, head = e["rt"], e["enc"], e["head"]
        B, Tl, _, H, W = x_local.shape
        assert B == 1, "frame sharding is per clip"
        exch = FrameShardExchange(Tl * world(group), group)
        xf = x_local.to(device=rt.device, dtype=torch.float32).reshape(Tl, 3, H, W).contiguous()
        taps, _, (ph, pw) = enc.run(xf)
        depth = head.run(taps, Tl, ph, pw, T=exch.T, relu=not _pre_relu, exch=exch)
        return depth.reshape(1, Tl, H, W).clone()

    # ------------------------------------------------------------------ staged interface of the multi-GPU driver (vdn/dist.py)
    @torch.no_grad()
    def encode_frames(self, x: torch.Tensor):
        """x [k,3,H,W] (k >= 0) -> (planes, rows_per_frame, (H, W)): the 4 final-normed encoder taps of the k frames as
        plain 2-D tensors [k*P, C] (hi and lo planes of tap 0, then tap 1, ...), which is what vdn.dist exchanges."""
        e = self._engines()
        rt, enc = e["rt"], e["enc"]
        k, _, H, W = x.shape
        P, C = (H // 14) * (W // 14), self.pretrained.embed_dim
        out = [rt.hbuf(f"enc_share_tap{j}", (max(k, 1) * P, C)) for j in range(4)]
        xf = x.to(device=rt.device, dtype=torch.float32)
        for c0 in range(0, k, util.INFER_LEN):
            c1 = min(k, c0 + util.INFER_LEN)
            enc.run(xf[c0:c1].contiguous(), tap_out=[t.narrow0(c0 * P, (c1 - c0) * P) for t in out])
        planes = []
        for t in out:
            planes.append(t.hi[:k * P])
            if t.lo is not None:
                planes.append(t.lo[:k * P])
        return planes, P, (H, W)

    def tap_planes(self, x: torch.Tensor):
        """Destination of `encode_into` for the k frames x [k,3,H,W]: (planes, rows_per_frame, (H, W)), the planes as
        encode_frames returns them but not yet written."""
        rt = self._engines()["rt"]
        k, _, H, W = x.shape
        P, C = (H // 14) * (W // 14), self.pretrained.embed_dim
        out = [rt.hbuf(f"enc_share_tap{j}", (max(k, 1) * P, C)) for j in range(4)]
        self._tap_out = out
        planes = []
        for t in out:
            planes.append(t.hi[:k * P])
            if t.lo is not None:
                planes.append(t.lo[:k * P])
        return planes, P, (H, W)

    @torch.no_grad()
    def encode_into(self, x: torch.Tensor, planes, first: int):
        """Encode frames x [c,3,H,W] into rows [first * P, (first + c) * P) of the planes `tap_planes` handed out (the
        multi-GPU driver encodes in chunks so that a chunk's taps travel while the next one is encoded)."""
        e = self._engines()
        rt, enc = e["rt"], e["enc"]
        c, _, H, W = x.shape
        P = (H // 14) * (W // 14)
        xf = x.to(device=rt.device, dtype=torch.float32).contiguous()
        enc.run(xf, tap_out=[t.narrow0(first * P, c * P) for t in self._tap_out])

    @torch.no_grad()
    def head_from_planes(self, planes, Tl: int, T: int, hw, group=None) -> torch.Tensor:
        """The temporal DPT head on the taps of Tl frames (planes as returned by encode_frames, rows of these Tl frames in
        slot order) -> depth [Tl,H,W]. group None: a whole window (Tl == T); else this rank's Tl of the T frames of a
        window that is frame-sharded over `group` (all-to-all around each temporal module)."""
        from .dist import FrameShardExchange
        from .runtime import HL
        e = self._engines()
        rt, head = e["rt"], e["head"]
        H, W = hw
        per = 2 if rt.split else 1
        taps = [HL(planes[per * j], planes[per * j + 1] if rt.split else None) for j in range(4)]
        exch = None if group is None else FrameShardExchange(T, group)
        assert exch is not None or Tl == T
        depth = head.run(taps, Tl, H // 14, W // 14, T=T, exch=exch)
        return depth.reshape(Tl, H, W)

    def preprocess_frames(self, frames: np.ndarray, input_size: int) -> torch.Tensor:
        """u8 RGB [n,h,w,3] -> normalised f32 [n,3,H,W] on the device (video_depth.py:73-99)."""
        rt = self._engines()["rt"]
        return self.preprocess(rt, torch.from_numpy(np.ascontiguousarray(frames)).to(rt.device), input_size)

    def resize_depth(self, d: torch.Tensor, fh: int, fw: int) -> torch.Tensor:
        """[n,H,W] -> [n,fh,fw], bilinear align_corners (video_depth.py:111)."""
        if tuple(d.shape[-2:]) == (fh, fw):
            return d
        rt = self._engines()["rt"]
        o = torch.empty((d.shape[0], fh, fw), dtype=torch.float32, device=rt.device)
        rt.upsample_f32(d.contiguous(), o, d.shape[0], d.shape[-2], d.shape[-1], fh, fw)
        return o

    @torch.no_grad()
    def infer_video_depth(self, frames: np.ndarray, target_fps, input_size: int = 518, device: str = "cuda",
                          fp32: bool = False):
        """frames u8 RGB [N,h,w,3] -> (f32 [N,h,w], target_fps) (video_depth.py:67-156). `fp32` is accepted for
        signature compatibility; operands are always 16-bit with fp32 accumulation."""
        e = self._engines()
        rt = e["rt"]
        fh, fw = frames[0].shape[:2]
        ratio = max(fh, fw) / min(fh, fw)
        if ratio > 1.78:
            input_size = int(input_size * 1.777 / ratio)
            input_size = round(input_size / 14) * 14
        n = frames.shape[0]
        net_in = self.preprocess_frames(frames, input_size)  # [n,3,H,W]
        table = util.window_table(n)
        st = DeviceStitcher(rt, len(table), fh, fw)
        for d in self.window_depths(net_in, table):
            st.push(self.resize_depth(d, fh, fw))  # [32,fh,fw], stays on the device
        res = st.result(n)
        util.check_finite(res, "infer_video_depth")
        return util.to_host(res), target_fps  # the clip's only device-to-host copy (pinned buffer)

    # bytes of encoder taps one frame keeps in the clip-level cache: 4 taps x P tokens x C channels x 16-bit planes
    def _tap_bytes_per_frame(self, H: int, W: int) -> int:
        rt = self._engines()["rt"]
        return 4 * (H // 14) * (W // 14) * self.pretrained.embed_dim * 2 * (2 if rt.split else 1)

    def window_depths(self, net_in: torch.Tensor, table, windows=None):
        """Depth [32,H,W] of the windows `windows` (default: all) of a clip whose pre-processed frames are net_in [n,3,H,W].
        The reference runs the encoder on all 32 slots of every window (video_depth.py:96-103), but 10 of the 32 are
        copies of earlier input frames (Appendix B of SURVEY.md: slot 0 = frame 0, slot 1 = the previous window's
        keyframe, slots 2..9 = the previous window's last 8), and the encoder is per-frame: here every DISTINCT frame
        goes through the encoder once (its 4 final-normed taps stay in HBM, 22 MB per 518x518 ViT-L frame), and a
        window's head reads its 32 slots from that cache — 256 instead of 384 encoder passes for a 256-frame clip,
        same numbers. Falls back to per-window encoding when the cache would not fit (VDN_TAP_CACHE_GB, default 96)."""
        import os
        e = self._engines()
        rt, enc, head = e["rt"], e["enc"], e["head"]
        n, _, H, W = net_in.shape
        windows = list(range(len(table))) if windows is None else list(windows)
        need = sorted({f for w in windows for f in table[w]})
        budget = float(os.environ.get("VDN_TAP_CACHE_GB", "96")) * 2 ** 30
        if len(need) * self._tap_bytes_per_frame(H, W) > budget:
            for w in windows:
                yield self.forward(net_in[torch.tensor(table[w], device=rt.device)][None])[0]
            return
        ph, pw = H // 14, W // 14
        P, C = ph * pw, self.pretrained.embed_dim
        slot = {f: i for i, f in enumerate(need)}        # cache row block of frame f
        cache = [rt.hbuf(f"clip_tap{j}", (len(need) * P, C)) for j in range(4)]
        for c0 in range(0, len(need), util.INFER_LEN):   # encoder batches of 32 distinct frames
            fr = need[c0:c0 + util.INFER_LEN]
            x = net_in[torch.tensor(fr, device=rt.device)] if fr != list(range(fr[0], fr[0] + len(fr))) else net_in[fr[0]:fr[0] + len(fr)]
            enc.run(x.contiguous(), tap_out=[t.narrow0(c0 * P, len(fr) * P) for t in cache])
        T = util.INFER_LEN
        for w in windows:
            rows = [slot[f] for f in table[w]]
            if rows == list(range(rows[0], rows[0] + T)):          # one contiguous run: read the cache in place
                taps = [t.narrow0(rows[0] * P, T * P) for t in cache]
            else:                                                   # copy the (typically 3) runs of slots into a window buffer
                taps = [rt.hbuf(f"win_tap{j}", (T * P, C)) for j in range(4)]
                i = 0
                while i < T:
                    k = i
                    while k + 1 < T and rows[k + 1] == rows[k] + 1:
                        k += 1
                    for src, dst in zip(cache, taps):
                        dst.hi[i * P:(k + 1) * P].copy_(src.hi[rows[i] * P:(rows[k] + 1) * P])
                        if dst.lo is not None:
                            dst.lo[i * P:(k + 1) * P].copy_(src.lo[rows[i] * P:(rows[k] + 1) * P])
                    i = k + 1
            yield head.run(taps, T, ph, pw, T=T, relu=True).reshape(T, H, W)


class DeviceStitcher:
    """video_depth.py:118-156 on the device: every window after the first is affine-aligned to
    the running result on the two alignment frames (least-squares scale/shift, `vdn_stitch_fit`), clamped at 0,
    cross-faded over the 8 interpolation frames and appended (`vdn_stitch_apply`). No host synchronisation until
    `result()` is copied; `vdn/util.stitch` is the host restatement the tests compare against."""

    def __init__(self, rt, n_windows: int, fh: int, fw: int):
        self.rt = rt
        self.align = OVERLAP - INTERP_LEN
        self.step = INFER_LEN - OVERLAP
        self.out = torch.empty((INFER_LEN + self.step * (n_windows - 1), fh, fw), dtype=torch.float32, device=rt.device)
        self.ref = torch.empty((self.align, fh, fw), dtype=torch.float32, device=rt.device)
        self.coef = torch.empty(2, dtype=torch.float32, device=rt.device)
        self.length = 0

    def push(self, d: torch.Tensor):
        d = d.contiguous()
        assert d.shape[0] == INFER_LEN and d.dtype == torch.float32
        if self.length == 0:
            self.out[:INFER_LEN].copy_(d)
            for j, k in enumerate(KEYFRAMES[:self.align]):
                self.ref[j].copy_(d[k])
            self.length = INFER_LEN
            return
        assert self.align == 2, "the reference keeps ref[0] fixed and moves ref[1] (video_depth.py:150-152)"
        self.rt.stitch_fit(d[:self.align], self.ref, self.coef)
        self.rt.stitch_apply(d, self.coef, self.out[self.length - INTERP_LEN:self.length],
                             self.out[self.length:self.length + self.step], self.ref[1], self.align, OVERLAP, KEYFRAMES[1])
        self.length += self.step

    def result(self, org_len: int) -> torch.Tensor:
        return self.out[:org_len]
