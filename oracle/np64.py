"""oracle/np64.py -- TEST INFRASTRUCTURE: float64 numpy restatement of the op semantics of the hot path, written as
plain loops / einsums directly from the Keras layer + TensorFlow-backend definitions (channels-last).  It is the
ground truth that pins oracle/torch_ref.py (tests/test_oracle.py) and is never imported by the product.

**parity unpinned** at the TensorFlow boundary: see the header of oracle/torch_ref.py.
"""
import numpy as np


def conv_nd(x, kernel, strides, pad, bias=None):
    """cross-correlation, K.layers/convolutional.py:148-182 -> TFB:3128-3165 / 3277-3314.
    x (N,*S,Cin), kernel (*k,Cin,Cout), explicit symmetric zero `pad` per spatial axis, VALID afterwards:
    out = floor((L + 2p - k)/s) + 1  (K.utils/conv_utils.py:90-116)."""
    nd = x.ndim - 2
    xp = np.pad(x, [(0, 0)] + [(p, p) for p in pad] + [(0, 0)])
    ks = kernel.shape[:nd]
    out_sp = [(xp.shape[1 + a] - ks[a]) // strides[a] + 1 for a in range(nd)]
    y = np.zeros((x.shape[0],) + tuple(out_sp) + (kernel.shape[-1],))
    for idx in np.ndindex(*ks):
        sl = tuple(slice(idx[a], idx[a] + strides[a] * (out_sp[a] - 1) + 1, strides[a]) for a in range(nd))
        patch = xp[(slice(None),) + sl + (slice(None),)]
        y += np.einsum("...c,co->...o", patch, kernel[idx])
    if bias is not None:
        y = y + bias
    return y


def batch_norm_train(x, gamma, beta, eps):
    """tf.nn.moments (two-pass, biased) + tf.nn.batch_normalization, TFB:1635-1640"""
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    mean = xf.mean(0)
    var = ((xf - mean) ** 2).mean(0)
    inv = gamma / np.sqrt(var + eps)
    return x * inv + (beta - mean * inv), mean, var


def batch_norm_infer(x, gamma, beta, mm, mv, eps):
    inv = gamma / np.sqrt(mv + eps)
    return x * inv + (beta - mm * inv)


def moving_update(m, batch, momentum):
    """TFB:915-927 assign_moving_average(zero_debias=False)"""
    return m - (m - batch) * (1 - momentum)


def scale(x, gamma, beta):
    """lib/custom_layers.py:63-69"""
    return x * gamma + beta


def relu(x):
    return np.maximum(x, 0)


def zero_pad(x, p):
    return np.pad(x, [(0, 0)] + [(p, p)] * (x.ndim - 2) + [(0, 0)])


def max_pool(x, k, s):
    """VALID max pooling (on an explicitly zero-padded input the zeros compete), TFB:3386,3426"""
    nd = x.ndim - 2
    out_sp = [(x.shape[1 + a] - k) // s + 1 for a in range(nd)]
    y = np.full((x.shape[0],) + tuple(out_sp) + (x.shape[-1],), -np.inf)
    for idx in np.ndindex(*([k] * nd)):
        sl = tuple(slice(idx[a], idx[a] + s * (out_sp[a] - 1) + 1, s) for a in range(nd))
        y = np.maximum(y, x[(slice(None),) + sl + (slice(None),)])
    return y


def avg_pool(x, k):
    """VALID average pooling with window = stride = k (tuple per axis), TFB:3388,3428"""
    nd = x.ndim - 2
    out_sp = [x.shape[1 + a] // k[a] for a in range(nd)]
    y = np.zeros((x.shape[0],) + tuple(out_sp) + (x.shape[-1],))
    for idx in np.ndindex(*k):
        sl = tuple(slice(idx[a], idx[a] + k[a] * (out_sp[a] - 1) + 1, k[a]) for a in range(nd))
        y += x[(slice(None),) + sl + (slice(None),)]
    return y / np.prod(k)


def upsample(x, size):
    """UpSampling2D/3D: np.repeat per axis (the reference suite's own known answer, convolutional_test.py:673-681)"""
    for a, f in enumerate(size):
        x = np.repeat(x, f, axis=1 + a)
    return x


def softmax(z):
    e = np.exp(z - z.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def weighted_crossentropy(logits, labels, weights=(0.78, 0.65, 8.57)):
    """loss.py:27-46 and its analytic gradient (tf.clip_by_value passes gradient only inside [1e-10, 1])."""
    z = logits.reshape(-1, 3)
    lab = labels.reshape(-1).astype(int)
    p = softmax(z)
    M = z.shape[0]
    pc = p[np.arange(M), lab]
    w = np.asarray(weights)[lab]
    loss = -(w * np.log(np.clip(pc, 1e-10, 1.0))).sum() / M
    onehot = np.eye(3)[lab]
    inside = ((pc >= 1e-10) & (pc <= 1.0))[:, None]
    grad = (w[:, None] / M) * (p - onehot) * inside
    return loss, grad.reshape(logits.shape)


def sgd_nesterov(p, v, g, lr, momentum):
    """K.optimizers.py:168-185"""
    v_new = momentum * v - lr * g
    return p + momentum * v_new - lr * g, v_new


def slab25d(vol_hwd):
    """denseunet3d.py:399-410: (H,W,D) -> (D,H,W,3) with edge replication"""
    D = vol_hwd.shape[2]
    return np.stack([vol_hwd[:, :, [max(k - 1, 0), k, min(k + 1, D - 1)]] for k in range(D)], 0)
