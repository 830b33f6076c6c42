"""Drop-in for the reference's densenet.py: the same 2D DenseUNet-161 WITHOUT the UNet skip adds / line0
(densenet.py:10-101); takes `args` (args.b, args.input_size)."""
from . import denseunet as _d
from .keras_api import Model, SGD  # noqa: F401
import os


def DenseUNet(nb_dense_block=4, growth_rate=48, nb_filter=96, reduction=0.0, dropout_rate=0.0, weight_decay=1e-4,
              weights_path=None, args=None, dtype=None, nb_layers=(6, 12, 36, 24), seed=4321):
    if args is None:
        raise ValueError("densenet.DenseUNet needs args (args.b, args.input_size) (densenet.py:34)")
    if reduction != 0.5:
        raise NotImplementedError("the scripts always call DenseUNet(reduction=0.5)")
    dtype = dtype or os.environ.get("HDU_DTYPE", "bf16")
    m = Model("2d", args.b, args.input_size, dtype=dtype, variant="densenet", name="denseu161",
              nb_layers2d=tuple(nb_layers), seed=seed)
    if weights_path is not None:
        m.load_weights(weights_path)
    return m
