"""yolo355 -- MI355X-native quantized slim-YOLOv2 inference (drop-in for the reference's
models/slim_yolo_v2.py forward() and utils/modules.py operator API).

Python host code over a C ABI (include/yolo355.h, libyolo355.so: hand-written HIP kernels
for gfx950).  There is no CPU or PyTorch fallback: anything that computes raises if the
library or a GPU is missing.
"""
__all__ = ["synth", "prep", "engine", "shard"]
