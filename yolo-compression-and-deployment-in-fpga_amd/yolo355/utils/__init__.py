from .modules import Conv2d, Conv2d_fuse, Conv2d_fuse_nobias, reorg_layer, SPP  # noqa: F401
from .bn_fuse import fuse_conv_and_bn  # noqa: F401
