"""Network builders: the reference's constructors restated as static launch lists on the engine.

Layer names, weight shapes and the trainable / BN-mode matrix follow the reference exactly:
  2D DenseUNet-161 ...... denseunet.py:130-319 (skips), densenet.py:10-101 (no skips),
                          denseunet3d.py:194-363 (frozen), hybridnet.py:182-354 (BN frozen, convs/Scales trainable)
  3D DenseNet ........... denseunet3d.py:18-190, hybridnet.py:11-178
  hybrid + HFF head ..... denseunet3d.py:393-439 (denseunet_3d), hybridnet.py:379-423 (dense_rnn_net)
"""
import torch

from . import ops
from .custom_layers import Scale
from .engine import AvgPoolLayer, BNLayer, ConvLayer, MaterializeLayer, MaxPoolLayer, StatsOp, Var

EPS_DENSE = 1.1e-5


def _FORCE_HALO():
    """HDU_FORCE_DEPTH_HALO=1 (tests / timing): a world-1 ShardInfo builds the depth-halo form of the network as well -- every
    3D layer over an input that stores its halo planes, which stay zero without neighbours, i.e. the unsharded network computed
    by the sharded launch list"""
    import os
    return os.environ.get("HDU_FORCE_DEPTH_HALO") == "1"


def _stats(ctx, var, mode):
    """returns the StatsOp (or None) so that a unique consumer BN can be fused into it"""
    if mode == "batch":
        return StatsOp(ctx, var)
    return None


def _fuse(st, bn):
    if st is not None and bn.mode == "batch":
        st.fuse(bn)


def build_dense_unet_2d(ctx, x_in, variant="denseunet", reduction=0.5, nb_layers=(6, 12, 36, 24), growth=48,
                        materialize_feature=False):
    """x_in: Var [N][1][H][W][cpad(3)].  Returns dict(feat_raw, feat_bn, feat (if materialised), logits)."""
    compression = 1.0 - reduction
    standalone = variant in ("denseunet", "densenet")
    mode = "batch" if standalone else "frozen"
    tr_conv = variant != "3dpart"
    tr_bn = standalone
    tr_scale = variant != "3dpart"
    dec_init = "normal" if standalone else "glorot"
    skips = variant == "denseunet"
    dt = ctx.dtype
    a = x_in.act
    N, H, W = a.N, a.H, a.W
    assert a.D == 1 and H % 32 == 0 and W % 32 == 0, "H and W must be multiples of 32 (five stride-2 stages)"

    def bn_dense(name, C):
        return BNLayer(ctx, name + "_bn", C, EPS_DENSE, 0.99, mode, tr_bn, scale=Scale(axis=3, name=name + "_scale", trainable=tr_scale))

    nb_filter = 96
    conv1 = ConvLayer(ctx, "conv1", x_in, nb_filter, (1, 7, 7), (1, 2, 2), (0, 3, 3), use_bias=False,
                      trainable=tr_conv, cin_logical=3)
    st = _stats(ctx, conv1.out, mode)
    bn1 = bn_dense("conv1", nb_filter)
    _fuse(st, bn1)
    z0 = MaterializeLayer(ctx, conv1.out, bn1).out          # relu1 = box[0]
    box = [z0]

    seg_stats = [None]        # StatsOp of the slab segment written last: its finalize launch also folds the next BN

    def dense_block(stage, nlayers, buf, c0):
        c = c0
        for i in range(nlayers):
            base = "conv%d_%d" % (stage, i + 1)
            bn_a = bn_dense(base + "_x1", c)
            if seg_stats[0] is not None:
                seg_stats[0].then_fold(bn_a)
            c1 = ConvLayer(ctx, base + "_x1", buf.slab(0, c), growth * 4, (1, 1, 1), bn=bn_a, use_bias=False,
                           trainable=tr_conv)
            st = _stats(ctx, c1.out, mode)
            bn_b = bn_dense(base + "_x2", growth * 4)
            _fuse(st, bn_b)
            ConvLayer(ctx, base + "_x2", c1.out, growth, (1, 3, 3), pad=(0, 1, 1), bn=bn_b, use_bias=False,
                      out=buf.slab(c, growth), trainable=tr_conv, producer=c1)
            seg_stats[0] = _stats(ctx, buf.slab(c, growth), mode)
            c += growth
        return c

    h, w = H // 4, W // 4
    buf = ctx.new_var(N, 1, h, w, nb_filter + nb_layers[0] * growth)
    MaxPoolLayer(ctx, z0, out=buf.slab(0, nb_filter))
    _stats(ctx, buf.slab(0, nb_filter), mode)
    stage = 1
    for bi in range(3):
        stage = bi + 2
        nb_filter = dense_block(stage, nb_layers[bi], buf, nb_filter)
        box.append(buf)
        base = "conv%d_blk" % stage
        bn_t = bn_dense(base, nb_filter)
        if seg_stats[0] is not None:
            seg_stats[0].then_fold(bn_t)
        seg_stats[0] = None
        nout = int(nb_filter * compression)
        ct = ConvLayer(ctx, base, buf, nout, (1, 1, 1), bn=bn_t, use_bias=False, trainable=tr_conv)
        h, w = h // 2, w // 2
        nbuf = ctx.new_var(N, 1, h, w, ops.cpad(nout, dt) + nb_layers[bi + 1] * growth)
        AvgPoolLayer(ctx, ct.out, nbuf.slab(0, ct.out.C))
        _stats(ctx, nbuf.slab(0, ct.out.C), mode)
        buf, nb_filter = nbuf, nout
    final_stage = stage + 1
    nb_filter = dense_block(final_stage, nb_layers[-1], buf, nb_filter)
    bn5 = bn_dense("conv%d_blk" % final_stage, nb_filter)
    if seg_stats[0] is not None:
        seg_stats[0].then_fold(bn5)

    # decoder widths equal the skip widths: 768, 384, 96 for DenseNet-161 (denseunet.py:192-204)
    dec = [(box[2].C, "0"), (box[1].C, "1"), (box[0].C, "2"), (96, "3"), (64, "4")]
    assert tuple(nb_layers) != (6, 12, 36, 24) or [d[0] for d in dec] == [768, 384, 96, 96, 64]
    cur, cur_bn = buf, bn5
    for i, (f, tag) in enumerate(dec):
        skip = None
        if skips and i == 0:
            line0 = ConvLayer(ctx, "line0", box[3], nb_filter, (1, 1, 1), init="normal")
            skip = line0.out
        elif skips and i in (1, 2, 3):
            skip = box[3 - i]
        drop = 0.3 if (i == 4 and standalone) else 0.0
        cu = ConvLayer(ctx, "conv_up" + tag, cur, f, (1, 3, 3), pad=(0, 1, 1), bn=cur_bn, up=(0, 1, 1), skip=skip,
                       init=dec_init, trainable=tr_conv, dropout=drop)
        st = _stats(ctx, cu.out, mode)
        cur_bn = BNLayer(ctx, "bn_up" + tag, cu.out.C, 1e-3, 0.99, mode, tr_bn)
        _fuse(st, cur_bn)
        cur = cu.out
    res = dict(feat_raw=cur, feat_bn=cur_bn)
    if materialize_feature:
        feat = MaterializeLayer(ctx, cur, cur_bn).out
        cls = ConvLayer(ctx, "dense167classifer", feat, 3, (1, 1, 1), init=dec_init, trainable=tr_conv)
        res["feat"] = feat
    else:
        cls = ConvLayer(ctx, "dense167classifer", cur, 3, (1, 1, 1), bn=cur_bn, init=dec_init, trainable=tr_conv)
    res["logits"] = cls.out
    return res


def build_dense_net_3d(ctx, x_in, variant="3dpart", reduction=0.5, nb_layers=(3, 4, 12, 8), growth=32):
    """x_in: Var [1][D][H][W][cpad(4)] (depth-major).  Returns (ac_up4 raw Var, its BN)."""
    compression = 1.0 - reduction
    blk_mode = "batch" if variant == "3dpart" else "frozen"
    blk_tr = variant == "3dpart"
    dt = ctx.dtype
    a = x_in.act
    N, D, H, W = a.N, a.D, a.H, a.W
    assert H % 32 == 0 and W % 32 == 0 and D % 4 == 0, "H,W multiples of 32 and D a multiple of 4 (SURVEY.md A.2)"
    nb_filter = 96
    sharded = ctx.shard is not None and (ctx.shard.world > 1 or _FORCE_HALO())
    hl = 1 if sharded else 0      # depth halo of every 3x3x3 conv / of the max pool; the 7x7x7 stride-2 stem needs 3
    if sharded:
        assert D % 4 == 0, "local depth must be a multiple of 4"
    conv1 = ConvLayer(ctx, "3dconv1", x_in, nb_filter, (7, 7, 7), (2, 2, 2), (3, 3, 3), use_bias=False, keras_nd=3,
                      cin_logical=4, halo=3 if sharded else 0)
    st = StatsOp(ctx, conv1.out)
    bn1 = BNLayer(ctx, "3dconv1_bn", nb_filter, EPS_DENSE, 0.99, "batch", True, scale=Scale(axis=4, name="3dconv1_scale"))
    st.fuse(bn1)
    z0 = MaterializeLayer(ctx, conv1.out, bn1, halo=hl).out
    d0 = conv1.out.act.D
    d, h, w = (d0 - 1) // 2 + 1, H // 4, W // 4
    buf = ctx.new_var(N, d, h, w, nb_filter + nb_layers[0] * growth)
    MaxPoolLayer(ctx, z0, out=buf.slab(0, nb_filter), pad_d=0 if sharded else 1)
    _stats(ctx, buf.slab(0, nb_filter), blk_mode)

    seg_stats = [None]

    def dense_block(stage, nlayers, buf, c0):
        c = c0
        for i in range(nlayers):
            base = "3dconv%d_%d" % (stage, i + 1)
            bn_a = BNLayer(ctx, base + "_x1_bn", c, EPS_DENSE, 0.99, blk_mode, blk_tr, scale=Scale(axis=4, name=base + "_x1_scale"))
            if seg_stats[0] is not None:
                seg_stats[0].then_fold(bn_a)
            c1 = ConvLayer(ctx, base + "_x1", buf.slab(0, c), growth * 4, (1, 1, 1), bn=bn_a, use_bias=False, keras_nd=3)
            st = _stats(ctx, c1.out, blk_mode)
            bn_b = BNLayer(ctx, base + "_x2_bn", growth * 4, EPS_DENSE, 0.99, blk_mode, blk_tr, scale=Scale(axis=4, name=base + "_x2_scale"))
            _fuse(st, bn_b)
            ConvLayer(ctx, base + "_x2", c1.out, growth, (3, 3, 3), pad=(1, 1, 1), bn=bn_b, use_bias=False,
                      out=buf.slab(c, growth), keras_nd=3, halo=hl, producer=c1)
            seg_stats[0] = _stats(ctx, buf.slab(c, growth), blk_mode)
            c += growth
        return c

    stage = 1
    for bi in range(3):
        stage = bi + 2
        nb_filter = dense_block(stage, nb_layers[bi], buf, nb_filter)
        base = "3dconv%d_blk" % stage
        bn_t = BNLayer(ctx, base + "_bn", nb_filter, EPS_DENSE, 0.99, blk_mode, True, scale=Scale(axis=4, name=base + "_scale"))
        if seg_stats[0] is not None:
            seg_stats[0].then_fold(bn_t)
        seg_stats[0] = None
        nout = int(nb_filter * compression)
        ct = ConvLayer(ctx, base, buf, nout, (1, 1, 1), bn=bn_t, use_bias=False, keras_nd=3)
        h, w = h // 2, w // 2
        nbuf = ctx.new_var(N, d, h, w, ops.cpad(nout, dt) + nb_layers[bi + 1] * growth)
        AvgPoolLayer(ctx, ct.out, nbuf.slab(0, ct.out.C))
        # the next block's BNs (blk_mode) and, after the last block, the batch-stat final BN read these channels
        StatsOp(ctx, nbuf.slab(0, ct.out.C)) if (blk_mode == "batch" or bi == 2) else None
        buf, nb_filter = nbuf, nout
    final_stage = stage + 1
    c0_last = nb_filter
    nb_filter = dense_block(final_stage, nb_layers[-1], buf, nb_filter)
    if blk_mode != "batch":
        # final BN is batch-stat in both variants: it needs statistics of every slab of the last block
        for i in range(nb_layers[-1]):
            StatsOp(ctx, buf.slab(c0_last + i * growth, growth))
    bn5 = BNLayer(ctx, "3dconv%d_blk_bn" % final_stage, nb_filter, EPS_DENSE, 0.99, "batch", True,
                  scale=Scale(axis=4, name="3dconv%d_blk_scale" % final_stage))
    if seg_stats[0] is not None:
        seg_stats[0].then_fold(bn5)
    ups = [(0, 1, 1), (0, 1, 1), (0, 1, 1), (1, 1, 1), (1, 1, 1)]   # reference (2,2,1)x3 then (2,2,2)x2 over (H,W,D)
    filt = [504 if nb_layers == (3, 4, 12, 8) else nb_filter, 224, 192, 96, 64]
    cur, cur_bn = buf, bn5
    for i in range(5):
        cu = ConvLayer(ctx, "3dconv_up%d" % i, cur, filt[i], (3, 3, 3), pad=(1, 1, 1), bn=cur_bn, up=ups[i], keras_nd=3,
                       halo=hl)
        st = StatsOp(ctx, cu.out)
        cur_bn = BNLayer(ctx, "3dbn_up%d" % i, cu.out.C, 1e-3, 0.99, "batch", True)
        st.fuse(cur_bn)
        cur = cu.out
    return cur, cur_bn


class Slab25DLayer:
    """denseunet3d.py:399-410: slab k = CT slices (k-1,k,k+1), edges replicated, as the channels of 2D sample k."""

    def __init__(self, ctx, vol, out):
        self.vol, self.out = vol, out
        ctx.fwd.append(self.forward)

    def forward(self):
        a = self.out.act
        ops.slab25d(self.vol, a.N, a.H, a.W, a)


class Input3DLayer:
    """denseunet3d.py:423-425: input3d = concat([CT, 250 * logits2d], channels)."""

    def __init__(self, ctx, vol, logits2d, out, scale=250.0):
        self.ctx, self.vol, self.logits2d, self.out, self.scale = ctx, vol, logits2d, out, scale
        ctx.fwd.append(lambda: ops.make_input3d(self.vol, self.logits2d.act, self.scale, self.out.act))
        ctx.bwd.append(self.backward)

    def backward(self):
        if self.out.root.needs_grad and self.logits2d.root.needs_grad:
            acc = self.logits2d.grad_mode()
            ops.make_input3d_bwd(self.out.grad, self.scale, self.logits2d.grad, acc)


def as3d(ctx, v2d):
    """view a 2D-branch tensor [D][1][H][W][C] as the 3D tensor [1][D][H][W][C] (same memory: the reference's
    slice2d / transpose / concat chain, denseunet3d.py:371-420, is a pure re-indexing in depth-major layout)."""
    a = v2d.act
    act = ops.Act(a.buf, a.off, 1, a.N, a.H, a.W, a.C, a.ld, a.dtype)
    return Var(ctx, act, v2d.root, v2d.c0)


def build_hybrid(ctx, vol, D, H, W, variant="3dpart", nb_layers2d=(6, 12, 36, 24), nb_layers3d=(3, 4, 12, 8)):
    """vol: float32 device tensor [D][H][W].  Returns logits Var [1][D][H][W][cpad(3)]."""
    dt = ctx.dtype
    sharded = ctx.shard is not None and (ctx.shard.world > 1 or _FORCE_HALO())
    hl = 1 if sharded else 0
    if sharded:
        # depth-sharded hybrid (third row): this rank holds D planes of the volume plus ONE raw CT
        # plane from each depth neighbour (`vol` has D+2 planes; Model._upload_x exchanges them, global edges replicate
        # their own plane = the reference's edge slabs (0,0,1) / (D-2,D-1,D-1), denseunet3d.py:399-409).  The slabs of the
        # halo'd volume are built for all D+2 planes; the 2D net -- per-slice, BN in inference mode in both hybrids --
        # runs on the D interior ones.
        xall = ctx.new_var(D + 2, 1, H, W, ops.cpad(3, dt))
        Slab25DLayer(ctx, vol, xall)
        a = xall.act
        x2d = Var(ctx, ops.Act(a.buf, a.off + H * W * a.ld, D, 1, H, W, a.C, a.ld, a.dtype))
        ctx.vars.append(x2d)
        vol = vol[H * W:(D + 1) * H * W]
    else:
        x2d = ctx.new_var(D, 1, H, W, ops.cpad(3, dt))
        Slab25DLayer(ctx, vol, x2d)
    ctx.grad_enabled = variant == "end2end"
    r2d = build_dense_unet_2d(ctx, x2d, variant=variant, nb_layers=nb_layers2d, materialize_feature=True)
    ctx.grad_enabled = True
    in3d = ctx.new_var(1, D, H, W, ops.cpad(4, dt))
    if variant == "end2end":
        in3d.require_grad()
    Input3DLayer(ctx, vol, r2d["logits"], in3d)
    feat3d, bn3d = build_dense_net_3d(ctx, in3d, variant=variant, nb_layers=nb_layers3d)
    fea2d = as3d(ctx, r2d["feat"])
    fc = ConvLayer(ctx, "fianl_conv", feat3d, 64, (3, 3, 3), pad=(1, 1, 1), bn=bn3d, skip=fea2d, keras_nd=3,
                   dropout=0.1 if variant == "3dpart" else 0.3, halo=hl)
    st = StatsOp(ctx, fc.out)
    fbn = BNLayer(ctx, "final_bn", fc.out.C, 1e-3, 0.99, "batch", True)
    st.fuse(fbn)
    cls = ConvLayer(ctx, "2d3dclassifer", fc.out, 3, (1, 1, 1), bn=fbn, keras_nd=3)
    return cls.out


def build_dense_net_3d_standalone(ctx, x_in, nb_layers=(3, 4, 12, 8)):
    """DenseNet3D with its own head (`3dclassifer`, denseunet3d.py:187) as a stand-alone segmentation net: the
    network of BASELINE configs[4] (one 512^3 volume, depth-sharded).  All BNs in batch-statistics mode
    (denseunet3d.py variant)."""
    feat, bn = build_dense_net_3d(ctx, x_in, variant="3dpart", nb_layers=nb_layers)
    cls = ConvLayer(ctx, "3dclassifer", feat, 3, (1, 1, 1), bn=bn, keras_nd=3)
    return cls.out
