"""Drop-in for the reference's utils/bn_fuse.py (fuse_conv_and_bn, :21-45)."""
from ..prep import fuse_conv_and_bn  # noqa: F401
