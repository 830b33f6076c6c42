"""h-denseunet_amd -- MI355X (gfx950) native hot path of H-DenseUNet.

Python host code mirroring the reference's model-constructor surface (denseunet.py / denseunet3d.py /
hybridnet.py / loss.py / lib/custom_layers.py) over hand-written HIP kernels reached through the C-ABI of
libhdu.so (include/hdu.h).  torch is used for device memory, streams and torch.distributed only.
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
