"""Drop-in for models/video_depth_model_v5.py:124-192 (class name and state-dict keys kept)."""
from .refiner import _DepthRefiner


class VideoDepthAnything(_DepthRefiner):
    VERSION = 5
