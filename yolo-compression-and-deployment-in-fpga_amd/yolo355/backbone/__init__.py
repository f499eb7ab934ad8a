"""Module holders of the reference's backbone package used on the path (backbone/darknet.py)."""
from .darknet import Conv_BN_LeakyReLU, DarkNet_Light, darknet_light  # noqa: F401
