"""Drop-in for the reference's hybridnet.py: `dense_rnn_net(args)` (hybridnet.py:379-423), the end-to-end
H-DenseUNet (2D convs/Scales trainable with frozen BN; 3D dense-block BNs frozen): `-arch end2end`, test.py."""
import os

from .keras_api import Model


def dense_rnn_net(args, dtype=None, nb_layers2d=(6, 12, 36, 24), nb_layers3d=(3, 4, 12, 8), seed=4321, shard=None):
    """shard (shard.ShardInfo): this process holds args.input_cols depth planes of ONE volume split over the ranks
    (new capability; the reference has batch towers only)."""
    dtype = dtype or os.environ.get("HDU_DTYPE", "bf16")
    return Model("hybrid", args.b, args.input_size, args.input_cols, dtype=dtype, variant="end2end",
                 name="auto3d_residual_conv", nb_layers2d=tuple(nb_layers2d), nb_layers3d=tuple(nb_layers3d), seed=seed, shard=shard)
