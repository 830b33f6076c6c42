"""Stand-alone (optionally depth-sharded) 3D DenseNet trainer: the network of `DenseNet3D` (denseunet3d.py:105-190)
with its `3dclassifer` head, for BASELINE configs[4] (one 512^3 volume split on the depth axis over the GPUs of a node
with RCCL halo exchange, sync-BN and a summed gradient).  New capability: the reference has no spatial sharding."""
import os

from .keras_api import Model


def dense_net3d(args, dtype=None, nb_layers3d=(3, 4, 12, 8), seed=4321, shard=None):
    """args.input_size = H = W, args.input_cols = depth planes held by THIS rank (global depth = world * input_cols)."""
    dtype = dtype or os.environ.get("HDU_DTYPE", "bf16")
    return Model("3d", args.b, args.input_size, args.input_cols, dtype=dtype, variant="3dpart", name="densenet3d",
                 nb_layers3d=tuple(nb_layers3d), seed=seed, shard=shard)
