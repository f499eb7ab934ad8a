from .slim_yolo_v2 import SlimYOLOv2, SlimYOLOv2_quantize_bnfuse, AveragedRangeTracker  # noqa: F401
from .tiny_yolo_v3 import YOLOv3tiny  # noqa: F401
