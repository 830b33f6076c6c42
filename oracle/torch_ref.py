"""oracle/torch_ref.py -- TEST INFRASTRUCTURE: CPU restatement of the reference graphs.

**Pinned to the reference's own code** (round 3; tests/test_oracle_ref.py): oracle/ref_keras/ executes the reference's UNMODIFIED
Keras 2.0.8, model constructors, loss.py and SGD.get_updates on an eager torch `keras.backend`; the committed fixtures
tests/golden/ref_keras_*.{json,npz} (generator: oracle/ref_keras/make_ref_fixtures.py) hold this file's graphs, loss, gradients,
moving-average updates and optimizer step to them in float64 (logits 1e-9, gradients 1e-7).  Residual, stated: the reference's
PRIMITIVE arithmetic lives in TensorFlow 1.x (requirements.txt:71-72), which is neither vendored under /root/reference nor
installable here, and the reference holds no golden vectors for conv / pooling / exact BN -- those ~40
backend ops are restated on torch and cross-checked against oracle/np64.py.  This file restates the graphs of
denseunet.py / densenet.py / denseunet3d.py / hybridnet.py / loss.py layer by layer on plain torch CPU
functional ops (float32 or float64), in the reference's own channels-last layouts (2D: N,H,W,C; 3D:
N,H,W,D,C) and Keras weight shapes.  Op semantics are pinned by oracle/np64.py (independent float64 numpy
loops) and by the few known-answer tests the reference's Keras suite holds (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is the
checker, never the thing measured or shipped.

Layer provider protocol: the graph functions below call P.conv / P.bn / P.scale with the Keras layer name;
`ParamStore` creates Keras-shaped parameters on first use (seeded) and replays them afterwards, so the graph
definition is also the parameter inventory.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LOSS_WEIGHTS = (0.78, 0.65, 8.57)  # loss.py:23


# --------------------------------------------------------------------------- op semantics (channels-last)
def conv_nd(x, kernel, strides, padding, bias):
    """K.layers/convolutional.py:148-182 -> TFB:3128-3165 / 3277-3314.  x: (N,*spatial,C); kernel: (*k,Cin,Cout).
    padding 'valid' or 'same' (stride 1, odd k: symmetric k//2)."""
    nd = x.dim() - 2
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    perm_out = (0,) + tuple(range(2, nd + 2)) + (1,)
    w = kernel.permute((nd + 1, nd) + tuple(range(nd)))
    pad = 0
    if padding == "same":
        assert all(s == 1 for s in strides)
        pad = tuple(k // 2 for k in kernel.shape[:nd])
    fn = F.conv2d if nd == 2 else F.conv3d
    y = fn(x.permute(perm_in), w, bias, stride=strides, padding=pad)
    return y.permute(perm_out)


def zero_pad(x, p):
    """ZeroPadding2D/3D (K.layers/convolutional.py:1584,1702 -> TFB:1989-2071)."""
    nd = x.dim() - 2
    pads = [0, 0]
    for _ in range(nd):
        pads += [p, p]
    return F.pad(x, pads)


def max_pool_valid(x, k, s):
    """MaxPooling2D/3D VALID on the (already zero-padded) input: the pad zeros are ordinary data (TFB:3386,3426)."""
    nd = x.dim() - 2
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    perm_out = (0,) + tuple(range(2, nd + 2)) + (1,)
    fn = F.max_pool2d if nd == 2 else F.max_pool3d
    return fn(x.permute(perm_in), k, s).permute(perm_out)


def avg_pool_valid(x, k):
    nd = x.dim() - 2
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    perm_out = (0,) + tuple(range(2, nd + 2)) + (1,)
    fn = F.avg_pool2d if nd == 2 else F.avg_pool3d
    return fn(x.permute(perm_in), k, k).permute(perm_out)


def upsample_nearest(x, size):
    """UpSampling2D/3D = np.repeat per axis (K.layers/convolutional.py:1359,1432; TFB:1739-1827)."""
    for ax, f in enumerate(size):
        if f > 1:
            x = x.repeat_interleave(f, dim=1 + ax)
    return x


def batch_norm(x, gamma, beta, mov_mean, mov_var, eps, batch_stats):
    """K.layers/normalization.py:126-190.  batch_stats: tf.nn.moments (biased) + tf.nn.batch_normalization
    (TFB:1635-1640); else inference with the moving statistics (TFB:1667-1684).  Returns y, (mean, var)."""
    C = x.shape[-1]
    if batch_stats:
        xf = x.reshape(-1, C)
        mean = xf.mean(0)
        var = ((xf - mean) ** 2).mean(0)
    else:
        mean, var = mov_mean, mov_var
    inv = gamma * torch.rsqrt(var + eps)
    return x * inv + (beta - mean * inv), (mean, var)


def weighted_crossentropy_rows(logits, labels):
    """loss.py:27-46 on (M,3) logits and (M,) labels."""
    p = torch.softmax(logits, 1)
    lp = torch.log(torch.clamp(p, 1e-10, 1.0))
    parts = []
    for c, w in enumerate(LOSS_WEIGHTS):
        parts.append(w * lp[labels == c, c])
    return -torch.cat(parts).mean()


def weighted_crossentropy_2ddense(y_true, y_pred):
    return weighted_crossentropy_rows(y_pred.reshape(-1, 3), y_true.reshape(-1).long())


def weighted_crossentropy(y_true, y_pred):
    """loss.py:5-25: depth slices 1:7 only (axis 3 of N,H,W,D,C)."""
    return weighted_crossentropy_rows(y_pred[:, :, :, 1:7, :].reshape(-1, 3), y_true[:, :, :, 1:7, :].reshape(-1).long())


class _RoundBF16(torch.autograd.Function):
    """value AND gradient rounded to bfloat16 (round-to-nearest-even) and widened back: what storing a tensor and its
    gradient in bf16 does.  Used only by the calibration runs of the bf16 parity tests (ParamStore.store_bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundGradBF16(torch.autograd.Function):
    """identity forward, gradient rounded to bfloat16: one consumer's contribution to a bf16-stored gradient tensor"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


# --------------------------------------------------------------------------- parameters
class ParamStore:
    """Keras-ordered, Keras-shaped weights keyed by layer name.
    conv: [kernel (*k,Cin,Cout)(, bias)]; bn: [gamma, beta, moving_mean, moving_variance]; scale: [gamma, beta]."""

    def __init__(self, seed=4321, dtype=torch.float32, perturb=True):
        self.w = OrderedDict()
        self.kind = OrderedDict()
        self.trainable = {}
        self.bn_cfg = {}
        self.rng = np.random.default_rng(seed)
        self.dtype = dtype
        self.perturb = perturb
        self.bn_batch_means = {}
        self.learning_phase = 1
        self.dropout_off = True
        # calibration mode of the bf16 parity tests: every conv reads a bf16-stored input and bf16 filter copy and
        # writes a bf16-stored output (and the same for the gradients flowing back), accumulation stays float32 --
        # the storage precision of the product's throughput mode applied to the ORACLE's arithmetic.  The distance
        # between this run and the plain float32 run is the noise floor a bf16 implementation cannot beat.
        self.store_bf16 = False
        # storage-point ablation (tests/bf16_storage_ablation.py): with store_bf16 on, `store_policy(tag)` -> bool says
        # whether THIS storage point rounds (tag = the conv's layer name, or the name given at a q / qg call site); None = all round
        self.store_policy = None

    def _rounds(self, tag):
        return self.store_bf16 and (self.store_policy is None or bool(self.store_policy(tag)))

    # -- creation helpers
    def _t(self, a):
        return torch.tensor(np.asarray(a), dtype=self.dtype)

    def _conv_params(self, name, kshape, use_bias, init):
        fan_in = int(np.prod(kshape[:-1]))
        fan_out = int(np.prod(kshape[:-2])) * kshape[-1]
        if init == "normal":  # K.initializers.py:72,429: RandomNormal stddev 0.05
            k = self.rng.normal(0.0, 0.05, kshape)
        else:  # glorot_uniform, K.initializers.py:332
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            k = self.rng.uniform(-lim, lim, kshape)
        ws = [self._t(k)]
        if use_bias:
            b = self.rng.normal(0, 0.05, (kshape[-1],)) if self.perturb else np.zeros(kshape[-1])
            ws.append(self._t(b))
        return ws

    def conv(self, name, x, filters, k, strides=None, padding="valid", use_bias=True, init="glorot", trainable=True):
        nd = x.dim() - 2
        strides = strides or (1,) * nd
        if name not in self.w:
            self.w[name] = self._conv_params(name, tuple(k) + (x.shape[-1], filters), use_bias, init)
            self.kind[name] = "conv"
        self.trainable[name] = trainable
        ws = self.w[name]
        if self._rounds(name):
            w = ws[0] + (ws[0].detach().to(torch.bfloat16).to(ws[0].dtype) - ws[0].detach())   # bf16 copy, f32 gradient
            y = conv_nd(_RoundBF16.apply(x), w, strides, padding, ws[1] if use_bias else None)
            return _RoundBF16.apply(y)
        return conv_nd(x, ws[0], strides, padding, ws[1] if use_bias else None)

    def bn(self, name, x, eps=1e-3, momentum=0.99, mode="batch", trainable=True):
        """mode 'batch': training-phase batch statistics (+ moving update); 'frozen': training=False."""
        C = x.shape[-1]
        if name not in self.w:
            if self.perturb:
                ws = [self.rng.uniform(0.5, 1.5, C), self.rng.normal(0, 0.1, C), self.rng.normal(0, 0.1, C),
                      self.rng.uniform(0.5, 1.5, C)]
            else:
                ws = [np.ones(C), np.zeros(C), np.zeros(C), np.ones(C)]
            self.w[name] = [self._t(a) for a in ws]
            self.kind[name] = "bn"
        self.trainable[name] = trainable
        self.bn_cfg[name] = dict(eps=eps, momentum=momentum, mode=mode)
        g, b, mm, mv = self.w[name]
        batch = (mode == "batch") and self.learning_phase == 1
        y, (mean, var) = batch_norm(x, g, b, mm, mv, eps, batch)
        if batch:
            self.bn_batch_means[name] = (mean.detach(), var.detach())
        return y

    def scale(self, name, x, trainable=True):
        """lib/custom_layers.py:63-69"""
        C = x.shape[-1]
        if name not in self.w:
            if self.perturb:
                ws = [self.rng.uniform(0.5, 1.5, C), self.rng.normal(0, 0.1, C)]
            else:
                ws = [np.ones(C), np.zeros(C)]
            self.w[name] = [self._t(a) for a in ws]
            self.kind[name] = "scale"
        self.trainable[name] = trainable
        g, b = self.w[name]
        return x * g + b

    def q(self, x, tag="q"):
        """calibration mode: a tensor the product STORES (value and its gradient in bf16); identity otherwise"""
        return _RoundBF16.apply(x) if self._rounds(tag) else x

    def qg(self, x, tag="qg"):
        """calibration mode: a consumer's gradient contribution to a stored (bf16) gradient tensor; identity otherwise"""
        return _RoundGradBF16.apply(x) if self._rounds(tag) else x

    def dropout(self, x, rate):
        # parity runs: rate 0 (TF's RNG cannot be reproduced); predict: identity
        return x

    # -- training utilities
    def trainable_tensors(self):
        out = []
        for name, ws in self.w.items():
            if not self.trainable.get(name, True):
                continue
            n = 2 if self.kind[name] == "bn" else len(ws)
            for i in range(n):
                out.append((name, i, ws[i]))
        return out

    def apply_bn_updates(self):
        """moving_average_update, TFB:915-927: m -= (m - batch)*(1-momentum) with the biased batch variance"""
        for name, (mean, var) in self.bn_batch_means.items():
            mom = self.bn_cfg[name]["momentum"]
            ws = self.w[name]
            ws[2] = ws[2] - (ws[2] - mean) * (1 - mom)
            ws[3] = ws[3] - (ws[3] - var) * (1 - mom)
        self.bn_batch_means = {}

    def numpy(self):
        return OrderedDict((k, [w.detach().cpu().numpy() for w in ws]) for k, ws in self.w.items())


# --------------------------------------------------------------------------- 2D DenseUNet-161
def _conv_block2d(P, x, stage, branch, nb_filter, bn_mode, tr_conv, tr_bn, tr_scale):
    """denseunet.py:229-263 (frozen variants denseunet3d.py:276-309, hybridnet.py:264-297)"""
    eps = 1.1e-5
    base = "conv%d_%d" % (stage, branch)
    x = P.bn(base + "_x1_bn", x, eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale(base + "_x1_scale", x, trainable=tr_scale)
    x = torch.relu(x)
    x = P.conv(base + "_x1", x, nb_filter * 4, (1, 1), use_bias=False, trainable=tr_conv)
    x = P.bn(base + "_x2_bn", x, eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale(base + "_x2_scale", x, trainable=tr_scale)
    x = torch.relu(x)
    x = zero_pad(x, 1)
    x = P.conv(base + "_x2", x, nb_filter, (3, 3), use_bias=False, trainable=tr_conv)
    return x


def _dense_block2d(P, x, stage, nb_layers, nb_filter, growth, *flags):
    """denseunet.py:295-319"""
    concat = x
    for i in range(nb_layers):
        y = _conv_block2d(P, P.qg(concat), stage, i + 1, growth, *flags)
        concat = torch.cat([concat, y], -1)
        nb_filter += growth
    return concat, nb_filter


def _transition2d(P, x, stage, nb_filter, compression, bn_mode, tr_conv, tr_bn, tr_scale):
    """denseunet.py:266-292"""
    eps = 1.1e-5
    base = "conv%d_blk" % stage
    x = P.bn(base + "_bn", P.qg(x), eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale(base + "_scale", x, trainable=tr_scale)
    x = torch.relu(x)
    x = P.conv(base, x, int(nb_filter * compression), (1, 1), use_bias=False, trainable=tr_conv)
    return P.q(avg_pool_valid(x, (2, 2)), base + "_pool")


def dense_unet_2d(P, img, variant="denseunet", reduction=0.5, nb_layers=(6, 12, 36, 24), growth_rate=48):
    """variant 'denseunet': denseunet.py:130-227 (UNet skips + line0, decoder init 'normal', Dropout .3);
    'densenet': densenet.py:10-101 (no skips); '3dpart': denseunet3d.py:194-274 (everything frozen);
    'end2end': hybridnet.py:182-262 (BN frozen, convs + Scales trainable).  Returns (ac_up4, logits)."""
    eps = 1.1e-5
    compression = 1.0 - reduction
    standalone = variant in ("denseunet", "densenet")
    bn_mode = "batch" if standalone else "frozen"
    tr_conv = variant != "3dpart"
    tr_bn = standalone
    tr_scale = variant != "3dpart"
    flags = (bn_mode, tr_conv, tr_bn, tr_scale)
    dec_init = "normal" if standalone else "glorot"
    nb_filter = 96
    box = []
    x = zero_pad(img, 3)
    x = P.conv("conv1", x, nb_filter, (7, 7), strides=(2, 2), use_bias=False, trainable=tr_conv)
    x = P.bn("conv1_bn", x, eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale("conv1_scale", x, trainable=tr_scale)
    x = P.q(torch.relu(x), "conv1_relu")
    box.append(x)
    x = zero_pad(x, 1)
    x = max_pool_valid(x, 3, 2)
    stage = 1
    for bi in range(3):
        stage = bi + 2
        x, nb_filter = _dense_block2d(P, x, stage, nb_layers[bi], nb_filter, growth_rate, *flags)
        box.append(x)
        x = _transition2d(P, x, stage, nb_filter, compression, *flags)
        nb_filter = int(nb_filter * compression)
    final_stage = stage + 1
    xb, nb_filter = _dense_block2d(P, x, final_stage, nb_layers[-1], nb_filter, growth_rate, *flags)
    x = P.scale("conv%d_blk_scale" % final_stage, P.bn("conv%d_blk_bn" % final_stage, P.qg(xb), eps=eps, mode=bn_mode, trainable=tr_bn), trainable=tr_scale)
    x = torch.relu(x)
    box.append(x)

    skips = variant == "denseunet"
    # decoder widths equal the skip widths (768, 384, 96 for the real DenseNet-161 blocks, denseunet.py:192-204);
    # reduced-depth test nets keep that relation
    dec = [(box[2].shape[-1], "0"), (box[1].shape[-1], "1"), (box[0].shape[-1], "2"), (96, "3"), (64, "4")]
    cur = x
    for i, (f, tag) in enumerate(dec):
        up = upsample_nearest(cur, (2, 2))
        if skips and i == 0:
            line0 = P.conv("line0", box[3], box[4].shape[-1], (1, 1), padding="same", init="normal")
            up = line0 + up
        elif skips and i in (1, 2, 3):
            up = box[3 - i] + up
        c = P.conv("conv_up" + tag, up, f, (3, 3), padding="same", init=dec_init, trainable=tr_conv)
        if i == 4 and standalone:
            c = P.dropout(c, 0.3)
        c = P.bn("bn_up" + tag, c, mode=bn_mode, trainable=tr_bn)
        cur = torch.relu(c)
    logits = P.conv("dense167classifer", cur, 3, (1, 1), padding="same", init=dec_init, trainable=tr_conv)
    return cur, logits


# --------------------------------------------------------------------------- 3D DenseNet
def _conv_block3d(P, x, stage, branch, nb_filter, bn_mode, tr_bn):
    """denseunet3d.py:18-52; hybridnet.py:11-45 (BN frozen)"""
    eps = 1.1e-5
    base = "3dconv%d_%d" % (stage, branch)
    x = P.bn(base + "_x1_bn", x, eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale(base + "_x1_scale", x)
    x = torch.relu(x)
    x = P.conv(base + "_x1", x, nb_filter * 4, (1, 1, 1), use_bias=False)
    x = P.bn(base + "_x2_bn", x, eps=eps, mode=bn_mode, trainable=tr_bn)
    x = P.scale(base + "_x2_scale", x)
    x = torch.relu(x)
    x = zero_pad(x, 1)
    x = P.conv(base + "_x2", x, nb_filter, (3, 3, 3), use_bias=False)
    return x


def dense_net_3d(P, img, variant="3dpart", reduction=0.5, nb_layers=(3, 4, 12, 8), growth_rate=32):
    """denseunet3d.py:105-190 ('3dpart': all BN batch-stat) / hybridnet.py:98-178 ('end2end': dense-block BNs
    frozen+untrainable, transition BNs inference but trainable).  img: (N,H,W,D,4).  Returns ac_up4."""
    eps = 1.1e-5
    compression = 1.0 - reduction
    blk_mode = "batch" if variant == "3dpart" else "frozen"
    blk_tr = variant == "3dpart"
    nb_filter = 96
    x = zero_pad(img, 3)
    x = P.conv("3dconv1", x, nb_filter, (7, 7, 7), strides=(2, 2, 2), use_bias=False)
    x = P.bn("3dconv1_bn", x, eps=eps)
    x = P.scale("3dconv1_scale", x)
    x = P.q(torch.relu(x), "3dconv1_relu")
    x = zero_pad(x, 1)
    x = max_pool_valid(x, 3, 2)
    stage = 1
    for bi in range(3):
        stage = bi + 2
        concat = x
        for i in range(nb_layers[bi]):
            y = _conv_block3d(P, P.qg(concat), stage, i + 1, growth_rate, blk_mode, blk_tr)
            concat = torch.cat([concat, y], -1)
            nb_filter += growth_rate
        x = concat
        base = "3dconv%d_blk" % stage
        x = P.bn(base + "_bn", P.qg(x), eps=eps, mode=blk_mode, trainable=True)
        x = P.scale(base + "_scale", x)
        x = torch.relu(x)
        x = P.conv(base, x, int(nb_filter * compression), (1, 1, 1), use_bias=False)
        x = P.q(avg_pool_valid(x, (2, 2, 1)), base + "_pool")
        nb_filter = int(nb_filter * compression)
    final_stage = stage + 1
    concat = x
    for i in range(nb_layers[-1]):
        y = _conv_block3d(P, P.qg(concat), final_stage, i + 1, growth_rate, blk_mode, blk_tr)
        concat = torch.cat([concat, y], -1)
        nb_filter += growth_rate
    x = P.qg(concat)
    x = P.bn("3dconv%d_blk_bn" % final_stage, x, eps=eps)
    x = P.scale("3dconv%d_blk_scale" % final_stage, x)
    x = torch.relu(x)
    ups = [(2, 2, 1), (2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2)]
    filt = [504, 224, 192, 96, 64]
    if nb_layers != (3, 4, 12, 8):  # reduced-depth test nets keep the decoder widths proportional
        filt = [x.shape[-1], 224, 192, 96, 64]
    for i in range(5):
        up = upsample_nearest(x, ups[i])
        c = P.conv("3dconv_up%d" % i, up, filt[i], (3, 3, 3), padding="same")
        c = P.bn("3dbn_up%d" % i, c)
        x = torch.relu(c)
    return x


def dense_net_3d_standalone(P, img, nb_layers=(3, 4, 12, 8)):
    """DenseNet3D with its own `3dclassifer` head (denseunet3d.py:105-190): returns the logits `x` of :187"""
    feat = dense_net_3d(P, img, variant="3dpart", nb_layers=nb_layers)
    return P.conv("3dclassifer", feat, 3, (1, 1, 1), padding="same")


# --------------------------------------------------------------------------- hybrid nets
def hybrid_net(P, img, variant="3dpart", nb_layers2d=(6, 12, 36, 24), nb_layers3d=(3, 4, 12, 8)):
    """denseunet3d.py:393-439 (`denseunet_3d`, variant '3dpart') / hybridnet.py:379-423 (`dense_rnn_net`,
    variant 'end2end').  img: (1,H,W,D,1).  Returns logits (1,H,W,D,3)."""
    assert img.shape[0] == 1, "the reference's slice2d indexes the batch axis as the slice index (b must be 1)"
    D = img.shape[3]
    vol = img[0, :, :, :, 0]  # (H,W,D)
    slabs = []
    for k in range(D):  # denseunet3d.py:399-410
        idx = [max(k - 1, 0), k, min(k + 1, D - 1)]
        slabs.append(vol[:, :, idx])
    input2d = torch.stack(slabs, 0)  # (D,H,W,3)
    feature2d, classifer2d = dense_unet_2d(P, input2d, variant=variant, nb_layers=nb_layers2d)
    res2d = classifer2d.permute(1, 2, 0, 3)[None]  # slice2d + concat: (1,H,W,D,3)
    fea2d = P.q(feature2d, "fea2d").permute(1, 2, 0, 3)[None]
    input3d = torch.cat([img, res2d * 250], 4)
    feature3d = dense_net_3d(P, input3d, variant=variant, nb_layers=nb_layers3d)
    final = feature3d + fea2d
    c = P.conv("fianl_conv", final, 64, (3, 3, 3), padding="same")
    c = P.dropout(c, 0.1 if variant == "3dpart" else 0.3)
    c = P.bn("final_bn", c)
    c = torch.relu(c)
    return P.conv("2d3dclassifer", c, 3, (1, 1, 1), padding="same")


# --------------------------------------------------------------------------- one training step
def sgd_nesterov_(p, v, g, lr=1e-3, momentum=0.9):
    """K.optimizers.py:168-185"""
    v_new = momentum * v - lr * g
    return p + momentum * v_new - lr * g, v_new


def train_step(P, forward, loss_fn, x, y, velocities, lr=1e-3, momentum=0.9):
    """K.engine/training.py:948-967,1715-1766: loss, gradients w.r.t. pre-update trainable weights, BN moving
    updates, SGD assigns.  `velocities`: dict (name,i)->tensor (created on first use).  Returns loss, grads."""
    P.learning_phase = 1
    for ws in P.w.values():
        for t in ws:
            t.requires_grad_(False)
    trainable = None
    if P.w:
        trainable = P.trainable_tensors()
        for _, _, t in trainable:
            t.requires_grad_(True)
    out = forward(P, x)
    if trainable is None:  # first call created the parameters
        for ws in P.w.values():
            for t in ws:
                t.requires_grad_(False)
        trainable = P.trainable_tensors()
        for _, _, t in trainable:
            t.requires_grad_(True)
        P.bn_batch_means = {}
        out = forward(P, x)
    loss = loss_fn(y, out)
    grads = torch.autograd.grad(loss, [t for _, _, t in trainable], allow_unused=True)
    gd = {}
    with torch.no_grad():
        for (name, i, t), g in zip(trainable, grads):
            if g is None:
                g = torch.zeros_like(t)
            gd[(name, i)] = g
            v = velocities.get((name, i))
            if v is None:
                v = torch.zeros_like(t)
            pn, vn = sgd_nesterov_(t.detach(), v, g, lr, momentum)
            velocities[(name, i)] = vn
            P.w[name][i] = pn
        P.apply_bn_updates()
        for ws in P.w.values():
            for j in range(len(ws)):
                ws[j] = ws[j].detach()
    return float(loss.detach()), gd, out.detach()


def predict(P, forward, x):
    """K.engine/training.py:1659-1713: learning_phase 0 -> moving statistics everywhere, no dropout."""
    P.learning_phase = 0
    with torch.no_grad():
        out = forward(P, x)
    P.learning_phase = 1
    P.bn_batch_means = {}
    return out


def dice_per_class(pred_lab, ref_lab, classes=(0, 1, 2)):
    """hard Dice on arg-max labels (evaluation quantity)"""
    out = []
    for c in classes:
        a, b = pred_lab == c, ref_lab == c
        den = a.sum() + b.sum()
        out.append(1.0 if den == 0 else float(2.0 * (a & b).sum() / den))
    return out
