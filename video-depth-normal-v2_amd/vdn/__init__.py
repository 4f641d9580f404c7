"""vdn — MI355X-native per-frame depth inference behind the Depth-Anything-V2 / Video-Depth-Anything API.

Importing the model classes loads libvdn_hip.so (hand-written gfx950 kernels); there is no CPU path.
`from vdn import synth, util, modules` stay importable without the library (host logic only)."""

MODEL_CONFIGS = {
    "vits": {"encoder": "vits", "features": 64, "out_channels": [48, 96, 192, 384]},
    "vitb": {"encoder": "vitb", "features": 128, "out_channels": [96, 192, 384, 768]},
    "vitl": {"encoder": "vitl", "features": 256, "out_channels": [256, 512, 1024, 1024]},
    "vitg": {"encoder": "vitg", "features": 384, "out_channels": [1536, 1536, 1536, 1536]},   # DepthAnythingV2 only (run_video.py:32)
}


def __getattr__(name):
    if name == "DepthAnythingV2":
        from .depth_anything_v2 import DepthAnythingV2
        return DepthAnythingV2
    if name == "VideoDepthAnything":
        from .video_depth import VideoDepthAnything
        return VideoDepthAnything
    raise AttributeError(name)
