"""Drop-in for models/video_depth_model_v4.py:83-148 (class name and state-dict keys kept)."""
from .refiner import _DepthRefiner


class VideoDepthAnything(_DepthRefiner):
    VERSION = 4
