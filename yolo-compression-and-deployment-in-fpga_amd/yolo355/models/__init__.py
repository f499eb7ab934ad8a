from .slim_yolo_v2 import SlimYOLOv2_quantize_bnfuse, AveragedRangeTracker  # noqa: F401
