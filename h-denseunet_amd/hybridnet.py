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


_VARIANT = "end2end"


def DenseNet3D(img_input, nb_dense_block=4, growth_rate=32, nb_filter=96, reduction=0.0, dropout_rate=0.0, weight_decay=1e-4,
               classes=1000, weights_path=None):
    """The reference's sub-builder (hybridnet.py:98-178): 3D DenseNet encoder + decoder on `img_input`, returns `(ac_up4, x)` -- the
    64-channel feature map after `3dbn_up4` + ReLU and the `3dclassifer` logits -- exactly the pair the reference returns
    (its callers use ac_up4 only; `x` stays dead there too).  `img_input` is an `engine.Var` [1][D][H][W][cpad(4)] of an open
    build context (`engine.Ctx(dtype, None)`; `ctx.new_var(1, D, H, W, ops.cpad(4, dtype))`): where the reference threads Keras
    tensors through layer calls, this framework appends launches to the context the input belongs to; `ctx.finalize()` closes it.
    BN modes / trainable flags are this module's (hybridnet.py: dense-block / transition BNs in inference mode (gamma / beta of the transitions trainable), stem / decoder BNs batch-statistics).  Same argument list as the reference; the architecture constants
    it hard-codes (4 dense blocks of (3, 4, 12, 8) layers, 96 stem filters) are checked, `weight_decay` / `classes` /
    `weights_path` are accepted and unused as in the reference."""
    from . import models as _m
    from .engine import ConvLayer, MaterializeLayer
    if nb_dense_block != 4 or nb_filter != 96:
        raise ValueError("DenseNet3D: the reference architecture has nb_dense_block=4, nb_filter=96")
    if dropout_rate:
        raise ValueError("DenseNet3D: the reference builds its 3D branch with dropout_rate 0")
    ctx = img_input.ctx
    feat, bn = _m.build_dense_net_3d(ctx, img_input, variant=_VARIANT, reduction=reduction, growth=growth_rate)
    ac_up4 = MaterializeLayer(ctx, feat, bn).out
    x = ConvLayer(ctx, "3dclassifer", ac_up4, 3, (1, 1, 1), keras_nd=3).out
    return ac_up4, x


def DenseUNet(img_input, nb_dense_block=4, growth_rate=48, nb_filter=96, reduction=0.0, dropout_rate=0.0, weight_decay=1e-4,
              classes=1000, weights_path=None):
    """The reference's 2D sub-builder (hybridnet.py:182-262): DenseNet-161 encoder + 5-stage decoder WITHOUT skip connections on the 2.5D
    slabs `img_input` (engine.Var [D][1][H][W][cpad(3)]), returns `(ac_up4, x)`: the 64-channel feature map and the
    `dense167classifer` logits.  BN modes / trainable flags: convs and Scales trainable, BNs frozen in inference mode, hybridnet.py:210-320.  See DenseNet3D for the build-context convention."""
    from . import models as _m
    if nb_dense_block != 4 or nb_filter != 96:
        raise ValueError("DenseUNet: the reference architecture has nb_dense_block=4, nb_filter=96")
    r = _m.build_dense_unet_2d(img_input.ctx, img_input, variant=_VARIANT, reduction=reduction, growth=growth_rate,
                               materialize_feature=True)
    return r["feat"], r["logits"]
