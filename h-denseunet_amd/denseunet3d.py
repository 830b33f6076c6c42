"""Drop-in for the reference's denseunet3d.py: `denseunet_3d(args)` (denseunet3d.py:393-439) = frozen 2D
DenseUNet-161 on 2.5D slabs -> 3D DenseNet on (CT, 250*logits2d) -> HFF head; `-arch 3dpart` of train_hybrid.py."""
import os

from .keras_api import Model


def denseunet_3d(args, dtype=None, nb_layers2d=(6, 12, 36, 24), nb_layers3d=(3, 4, 12, 8), seed=4321, shard=None):
    """shard (shard.ShardInfo): this process holds args.input_cols depth planes of ONE volume split over the ranks
    (new capability; the reference has batch towers only)."""
    dtype = dtype or os.environ.get("HDU_DTYPE", "bf16")
    return Model("hybrid", args.b, args.input_size, args.input_cols, dtype=dtype, variant="3dpart",
                 name="auto3d_residual_conv", nb_layers2d=tuple(nb_layers2d), nb_layers3d=tuple(nb_layers3d), seed=seed, shard=shard)
