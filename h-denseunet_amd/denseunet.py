"""Drop-in for the reference's denseunet.py: `DenseUNet(...)` returns a model object with the Keras `Model`
method subset the training scripts use (keras_api.Model), running on the HIP kernels.

Signature = union of denseunet.py:130 and densenet.py:10-11: without `args` the input is the
module-global (batch_size, img_deps, img_rows, 3) of denseunet.py:30-33,153; with `args` it is
(args.b, args.input_size, args.input_size, 3) (densenet.py:34)."""
import os

from .keras_api import Model, SGD, weighted_crossentropy_2ddense as weighted_crossentropy  # noqa: F401

batch_size = 10      # denseunet.py:30
img_deps = 512       # denseunet.py:31
img_rows = 512       # denseunet.py:32
img_cols = 3


def DenseUNet(nb_dense_block=4, growth_rate=48, nb_filter=96, reduction=0.0, dropout_rate=0.0, weight_decay=1e-4,
              classes=1000, weights_path=None, args=None, dtype=None, nb_layers=(6, 12, 36, 24), seed=4321):
    """DenseNet-161 encoder + UNet decoder with skip adds (denseunet.py:130-227).  `weight_decay` is accepted and
    ignored exactly as in the reference (no kernel_regularizer is ever passed, denseunet.py:248,258,285);
    nb_filter / nb_layers are hard-set inside the reference (:159-160)."""
    if nb_dense_block != 4:
        raise ValueError("the reference hard-codes 4 dense blocks (denseunet.py:160)")
    if dropout_rate:
        raise NotImplementedError("dense-block dropout is dead code in the reference (dropout_rate=0.0 everywhere)")
    dtype = dtype or os.environ.get("HDU_DTYPE", "bf16")
    if args is not None:
        b, size = args.b, args.input_size
    else:
        if img_deps != img_rows:
            raise ValueError("square inputs only")
        b, size = batch_size, img_deps
    m = Model("2d", b, size, dtype=dtype, variant=_VARIANT, name="denseu161", nb_layers2d=tuple(nb_layers), seed=seed)
    m.reduction = reduction
    if reduction != 0.5:
        raise NotImplementedError("the scripts always call DenseUNet(reduction=0.5) (train_2ddense.py:178)")
    if weights_path is not None:
        m.load_weights(weights_path)
    return m


_VARIANT = "denseunet"
