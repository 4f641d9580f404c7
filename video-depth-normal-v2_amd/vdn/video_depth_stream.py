"""Import-path mirror of video_depth_anything/video_depth_stream.py: the reference keeps its streaming driver
(`infer_video_depth_one`, video_depth_stream.py:76-160) in a second class of the same name and constructor; here the
one VideoDepthAnything serves both drivers (vdn/video_depth.py: stream_step / infer_video_depth_one / reset_stream)."""
from .video_depth import VideoDepthAnything  # noqa: F401
