"""oracle/ref_keras/torch_backend.py -- TEST INFRASTRUCTURE: an eager `keras.backend` for the reference's vendored Keras.

The reference's arithmetic lives in TensorFlow 1.x, which is not vendored and cannot be installed here (SURVEY.md section
8c).  Everything ABOVE TensorFlow, however, is plain Python that ships with the reference: the model constructors
(denseunet.py, densenet.py, denseunet3d.py, hybridnet.py), lib/custom_layers.py, loss.py and the vendored
Keras-2.0.8/keras/{engine,layers,initializers,...}.  This module is installed as `keras.backend` (see
oracle/ref_keras/run_reference.py) so that those files run UNMODIFIED: graph wiring, layer names, weight shapes and
order, trainable flags, BN modes / eps / momentum, padding modes, pool sizes, the loss's slicing -- all come from the
reference's own code.  Only the ~40 primitive functions of Keras-2.0.8/keras/backend/tensorflow_backend.py ("TFB") the
path uses are restated here on torch CPU tensors, each citing the TFB lines it follows.

Eager instead of symbolic: a "placeholder" is a real tensor (the fed input), so building a model with the reference's
constructor IS one forward pass; `Model.__call__` on another tensor re-runs the layers' own `call()`s.  The learning phase
is a Python int (TFB:123-135 `set_learning_phase` semantics), so `in_train_phase` picks its branch at call time.

Only tests/ and the fixture generator may import this.  Nothing on the product path does.
"""
import builtins as _bi
import contextlib
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------- common.py
_FLOATX = "float32"          # K.backend/common.py:4
_EPSILON = 10e-8             # common.py:5
_IMAGE_DATA_FORMAT = "channels_last"   # common.py:6
_LEARNING_PHASE = 1
_UIDS = defaultdict(int)
_RNG = np.random.RandomState(4321)
FEED = []                    # tensors handed to the next placeholder() calls of matching shape (the model inputs)
PENDING_UPDATES = []         # (variable, new_value) pairs produced by moving_average_update during a forward pass

_TDT = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64,
        "bool": torch.bool, "uint8": torch.uint8, "float16": torch.float16}


def epsilon():
    return _EPSILON


def set_epsilon(e):
    global _EPSILON
    _EPSILON = e


def floatx():
    return _FLOATX


def set_floatx(fx):
    global _FLOATX
    if fx not in ("float16", "float32", "float64"):
        raise ValueError("Unknown floatx type: " + str(fx))
    _FLOATX = str(fx)


def cast_to_floatx(x):
    return np.asarray(x, dtype=_FLOATX)


def image_data_format():
    return _IMAGE_DATA_FORMAT


def set_image_data_format(fmt):
    global _IMAGE_DATA_FORMAT
    if fmt not in ("channels_last", "channels_first"):
        raise ValueError("Unknown data_format:", fmt)
    _IMAGE_DATA_FORMAT = str(fmt)


def image_dim_ordering():          # K.backend/common.py:118-131 (legacy names the reference scripts use)
    return "th" if _IMAGE_DATA_FORMAT == "channels_first" else "tf"


def set_image_dim_ordering(dim_ordering):     # common.py:100-115
    global _IMAGE_DATA_FORMAT
    if dim_ordering not in ("tf", "th"):
        raise ValueError("Unknown dim_ordering:", dim_ordering)
    _IMAGE_DATA_FORMAT = "channels_first" if dim_ordering == "th" else "channels_last"


def backend():
    # the vendored layers branch on this string (e.g. Lambda.compute_output_shape, K.layers/core.py:608-621):
    # the reference runs on the TensorFlow backend (train_2ddense.py:18 -> K.set_image_dim_ordering('tf'))
    return "tensorflow"


# ---------------------------------------------------------------------------------------------- bookkeeping
def get_uid(prefix=""):            # TFB:56-75
    _UIDS[prefix] += 1
    return _UIDS[prefix]


def reset_uids():
    _UIDS.clear()


def clear_session():
    reset_uids()
    del PENDING_UPDATES[:]


def manual_variable_initialization(value):
    pass


def learning_phase():              # TFB:102-117 (returns the Python int once set_learning_phase was called)
    return _LEARNING_PHASE


def set_learning_phase(value):     # TFB:120-135
    global _LEARNING_PHASE
    if value not in (0, 1):
        raise ValueError("Expected learning phase to be 0 or 1.")
    _LEARNING_PHASE = value


@contextlib.contextmanager
def name_scope(name):              # TFB:name_scope = tf.name_scope
    yield


def seed_initializers(seed):
    """test hook: the initialisers below draw from this generator"""
    global _RNG
    _RNG = np.random.RandomState(seed)


class KTensor(torch.Tensor):
    """torch tensor with the IDENTITY comparison semantics of a TensorFlow graph tensor: the Keras engine tests
    `x in inputs_ls`, `x == y` and hashes tensors (K.engine/topology.py:611); element-wise comparison goes through
    K.equal / tf.equal.  Every tensor derived from a KTensor is a KTensor (torch's subclass propagation)."""
    __hash__ = lambda self: id(self)

    def __eq__(self, other):
        return self is other

    def __ne__(self, other):
        return self is not other

    # graph tensors are immutable: `output += inputs[i]` (K.layers/merge.py:207-211) makes a NEW tensor
    def __iadd__(self, other):
        return self + other

    def __isub__(self, other):
        return self - other

    def __imul__(self, other):
        return self * other

    def __itruediv__(self, other):
        return self / other

    def get_shape(self):
        """tf.Tensor.get_shape(): the static shape (K.utils2/multi_gpu.py:46 calls .as_list() on it)"""
        shp = tuple(int(s) for s in self.shape)

        class _Shape(tuple):
            def as_list(self_inner):
                return list(self_inner)
        return _Shape(shp)


def _k(t):
    return t.as_subclass(KTensor)


def _dt(dtype):
    if dtype is None:
        dtype = _FLOATX
    if isinstance(dtype, torch.dtype):
        return dtype
    return _TDT[str(dtype)]


def _as_tensor(x, dtype=None):
    if torch.is_tensor(x):
        return x
    return _k(torch.as_tensor(np.asarray(x), dtype=_dt(dtype)))


# ---------------------------------------------------------------------------------------------- variables / placeholders
def variable(value, dtype=None, name=None, constraint=None):       # TFB:300-348
    if torch.is_tensor(value):
        value = value.detach().cpu().numpy()
    v = _k(torch.tensor(np.asarray(value), dtype=_dt(dtype)))
    if v.dtype.is_floating_point:
        v.requires_grad_(True)
    v._keras_shape = tuple(v.shape)          # TFB:342-345
    v._uses_learning_phase = False
    v._kname = name
    v.constraint = constraint
    return v


def constant(value, dtype=None, shape=None, name=None):            # TFB:351-368
    a = np.asarray(value, dtype=np.float64)
    if shape is not None:
        a = np.broadcast_to(a, shape).copy()
    return _k(torch.tensor(a, dtype=_dt(dtype)))


def is_keras_tensor(x):                                            # TFB:371-417
    if not torch.is_tensor(x):
        raise ValueError("Unexpectedly found an instance of type `%s`. Expected a symbolic tensor instance." % type(x))
    return hasattr(x, "_keras_history")


def placeholder(shape=None, ndim=None, dtype=None, sparse=False, name=None):      # TFB:420-459
    if shape is None and ndim:
        shape = tuple([None] * ndim)
    want = tuple(shape)
    for i, t in enumerate(FEED):
        if len(t.shape) == len(want) and _bi.all(w is None or w == s for w, s in zip(want, t.shape)):
            x = FEED.pop(i).to(_dt(dtype))
            break
    else:
        x = torch.zeros([1 if s is None else int(s) for s in want], dtype=_dt(dtype))
    x = _k(x.detach().clone())
    x._keras_shape = want
    x._uses_learning_phase = False
    x._is_placeholder = True
    return x


def is_placeholder(x):
    return getattr(x, "_is_placeholder", False)


def is_sparse(x):
    return False


def shape(x):
    return tuple(x.shape)


def int_shape(x):                                                  # TFB:481-502
    if hasattr(x, "_keras_shape"):
        return x._keras_shape
    return tuple(int(s) for s in x.shape)


def ndim(x):
    return x.dim()


def dtype(x):
    return str(x.dtype).replace("torch.", "")


def eval(x):
    return x.detach().cpu().numpy()


def get_value(x):
    return x.detach().cpu().numpy()


def batch_get_value(xs):
    return [get_value(x) for x in xs]


def set_value(x, value):
    with torch.no_grad():
        x.copy_(torch.as_tensor(np.asarray(value), dtype=x.dtype))


def batch_set_value(tuples):
    for x, value in tuples:
        set_value(x, value)


def count_params(x):
    return int(np.prod(tuple(x.shape)))


def zeros(shape, dtype=None, name=None):
    return variable(np.zeros(shape), dtype, name)


def ones(shape, dtype=None, name=None):
    return variable(np.ones(shape), dtype, name)


def zeros_like(x, dtype=None, name=None):
    return torch.zeros_like(x)


def ones_like(x, dtype=None, name=None):
    return torch.ones_like(x)


def identity(x):
    return x.clone()


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None):          # TFB:3634-3653
    return _k(torch.tensor(_RNG.uniform(minval, maxval, size=tuple(shape)), dtype=_dt(dtype)))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None):            # TFB:3610-3631
    return _k(torch.tensor(_RNG.normal(mean, stddev, size=tuple(shape)), dtype=_dt(dtype)))


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None):
    a = _RNG.normal(mean, stddev, size=tuple(shape))
    bad = np.abs(a - mean) > 2 * stddev
    while bad.any():
        a[bad] = _RNG.normal(mean, stddev, size=int(bad.sum()))
        bad = np.abs(a - mean) > 2 * stddev
    return _k(torch.tensor(a, dtype=_dt(dtype)))


def cast(x, dtype):
    return x.to(_dt(dtype))


# ---------------------------------------------------------------------------------------------- updates
def update(x, new_x):
    return (x, new_x)


def update_add(x, increment):
    return (x, x.detach() + increment)


def update_sub(x, decrement):
    return (x, x.detach() - decrement)


def moving_average_update(x, value, momentum):
    """TFB:915-927 -> moving_averages.assign_moving_average(x, value, momentum, zero_debias=False):
    x -= (x - value) * (1 - momentum).  Eager: the new value is recorded, `apply_pending_updates()` assigns it (the
    reference applies the updates in the same session.run as the optimiser step, K.engine/training.py:961)."""
    new = x.detach() - (x.detach() - value.detach()) * (1.0 - momentum)
    PENDING_UPDATES.append((x, new))
    return (x, new)


def apply_pending_updates():
    with torch.no_grad():
        for x, new in PENDING_UPDATES:
            x.copy_(new)
    del PENDING_UPDATES[:]


# ---------------------------------------------------------------------------------------------- element-wise / reductions
def _axis(axis, x):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(a % x.dim() for a in axis)
    return axis % x.dim()


def dot(x, y):
    return torch.matmul(x, y)


def square(x):
    return x * x


def sqrt(x):
    return torch.sqrt(torch.clamp(x, min=0.0))


def abs(x):
    return torch.abs(x)


def exp(x):
    return torch.exp(x)


def log(x):
    return torch.log(x)


def maximum(x, y):
    return torch.maximum(_as_tensor(x), _as_tensor(y))


def clip(x, min_value, max_value):
    return torch.clamp(x, min_value, max_value)


def equal(x, y):
    return torch.eq(x, y)


def not_equal(x, y):
    return torch.ne(x, y)


def greater_equal(x, y):
    return torch.ge(x, y)


def sum(x, axis=None, keepdims=False):
    return x.sum() if axis is None else x.sum(dim=_axis(axis, x), keepdim=keepdims)


def mean(x, axis=None, keepdims=False):                            # TFB:1375-1392
    if x.dtype == torch.bool:
        x = x.to(_dt(None))
    return x.mean() if axis is None else x.mean(dim=_axis(axis, x), keepdim=keepdims)


def max(x, axis=None, keepdims=False):
    return x.max() if axis is None else x.amax(dim=_axis(axis, x), keepdim=keepdims)


def prod(x, axis=None, keepdims=False):
    return x.prod() if axis is None else x.prod(dim=axis, keepdim=keepdims)


def any(x, axis=None, keepdims=False):
    return x.bool().any() if axis is None else x.bool().any(dim=axis, keepdim=keepdims)


def all(x, axis=None, keepdims=False):
    return x.bool().all() if axis is None else x.bool().all(dim=axis, keepdim=keepdims)


def softmax(x):                                                    # TFB:2697-2708 -> tf.nn.softmax (last axis)
    return torch.softmax(_as_tensor(x), dim=-1)          # (tf.nn.softmax takes numpy arrays too: lib/funcs.py:31 passes one)


def relu(x, alpha=0.0, max_value=None):                            # TFB:2656-2679
    if alpha != 0.0:
        neg = F.relu(-x)
    x = F.relu(x)
    if max_value is not None:
        x = torch.clamp(x, 0.0, max_value)
    if alpha != 0.0:
        x = x - alpha * neg
    return x


def sigmoid(x):
    return torch.sigmoid(x)


def tanh(x):
    return torch.tanh(x)


# ---------------------------------------------------------------------------------------------- shape operations
def concatenate(tensors, axis=-1):                                 # TFB:1689-1709
    return torch.cat(list(tensors), dim=axis)


def reshape(x, shape):                                             # TFB:1712-1722
    return x.reshape(tuple(int(s) for s in shape))


def permute_dimensions(x, pattern):
    return x.permute(tuple(pattern))


def expand_dims(x, axis=-1):
    return x.unsqueeze(axis)


def squeeze(x, axis):
    return x.squeeze(axis)


def stack(x, axis=0):
    return torch.stack(list(x), dim=axis)


def batch_flatten(x):
    return x.reshape(x.shape[0], -1)


def repeat_elements(x, rep, axis):                                 # TFB:1806-1827: like np.repeat
    return x.repeat_interleave(rep, dim=axis)


def resize_images(x, height_factor, width_factor, data_format):    # TFB:1739-1774: tf.image.resize_nearest_neighbor
    # integer factors, align_corners=False: out[i] = in[floor(i / f)] = np.repeat (the Keras suite's own known answer,
    # Keras-2.0.8/tests/keras/layers/convolutional_test.py:673-681)
    if data_format == "channels_last":
        return x.repeat_interleave(height_factor, dim=1).repeat_interleave(width_factor, dim=2)
    return x.repeat_interleave(height_factor, dim=2).repeat_interleave(width_factor, dim=3)


def resize_volumes(x, depth_factor, height_factor, width_factor, data_format):    # TFB:1777-1803
    a = 1 if data_format == "channels_last" else 2
    x = repeat_elements(x, depth_factor, a)
    x = repeat_elements(x, height_factor, a + 1)
    return repeat_elements(x, width_factor, a + 2)


def spatial_2d_padding(x, padding=((1, 1), (1, 1)), data_format=None):            # TFB:2005-2035 -> tf.pad (zeros)
    data_format = data_format or image_data_format()
    assert data_format == "channels_last"
    (t, b), (l, r) = padding
    return F.pad(x, (0, 0, l, r, t, b))


def spatial_3d_padding(x, padding=((1, 1), (1, 1), (1, 1)), data_format=None):    # TFB:2038-2086
    data_format = data_format or image_data_format()
    assert data_format == "channels_last"
    (a0, a1), (b0, b1), (c0, c1) = padding
    return F.pad(x, (0, 0, c0, c1, b0, b1, a0, a1))


# ---------------------------------------------------------------------------------------------- NN ops
def in_train_phase(x, alt, training=None):                         # TFB:2591-2631
    if training is None:
        training = learning_phase()
    if training is 1 or training is True:       # noqa: F632  (the reference's own identity tests)
        return x() if callable(x) else x
    if training is 0 or training is False:      # noqa: F632
        return alt() if callable(alt) else alt
    raise ValueError("eager backend: the learning phase must be a static 0 / 1")


def in_test_phase(x, alt, training=None):
    return in_train_phase(alt, x, training=training)


DROPOUT_IDENTITY = False     # parity fixtures: TensorFlow's RNG stream cannot be reproduced, so Dropout is the identity there


def dropout(x, level, noise_shape=None, seed=None):                # TFB:2869-2888 -> tf.nn.dropout(x, keep)
    if DROPOUT_IDENTITY:
        return x * 1.0
    keep = 1.0 - level
    mask = _k(torch.tensor(_RNG.uniform(size=tuple(x.shape)) < keep, dtype=x.dtype))
    return x * mask / keep


def _same_pads(size, k, s):
    """TensorFlow 'SAME': out = ceil(size / s), total pad = max((out-1)*s + k - size, 0), low = total // 2"""
    out = -(-size // s)
    total = (out - 1) * s + k - size
    total = total if total > 0 else 0
    return total // 2, total - total // 2


def _conv(x, kernel, strides, padding, nd):
    """TFB:3128-3165 (conv2d) / TFB:3277-3314 (conv3d) -> tf.nn.convolution on NHWC / NDHWC: cross-correlation, kernel
    (k..., Cin, Cout), 'VALID' or TensorFlow 'SAME' (asymmetric: extra pad on the high side)."""
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    perm_out = (0,) + tuple(range(2, nd + 2)) + (1,)
    w = kernel.permute((nd + 1, nd) + tuple(range(nd)))
    xi = x.permute(perm_in)
    if padding == "same":
        pads = []
        for a in reversed(range(nd)):
            lo, hi = _same_pads(x.shape[1 + a], kernel.shape[a], strides[a])
            pads += [lo, hi]
        xi = F.pad(xi, pads)
    elif padding != "valid":
        raise ValueError("Invalid padding:", padding)
    fn = F.conv2d if nd == 2 else F.conv3d
    return fn(xi, w, None, stride=tuple(strides)).permute(perm_out)


def conv2d(x, kernel, strides=(1, 1), padding="valid", data_format=None, dilation_rate=(1, 1)):
    assert (data_format or image_data_format()) == "channels_last" and tuple(dilation_rate) == (1, 1)
    return _conv(x, kernel, strides, padding, 2)


def conv3d(x, kernel, strides=(1, 1, 1), padding="valid", data_format=None, dilation_rate=(1, 1, 1)):
    assert (data_format or image_data_format()) == "channels_last" and tuple(dilation_rate) == (1, 1, 1)
    return _conv(x, kernel, strides, padding, 3)


def bias_add(x, bias, data_format=None):                           # TFB:3435-3497 (channels_last: broadcast add)
    assert (data_format or image_data_format()) == "channels_last"
    return x + bias.reshape((1,) * (x.dim() - 1) + (-1,))


def _pool(x, pool_size, strides, padding, pool_mode, nd):
    """TFB:3354-3389 (pool2d) / TFB:3392-3432 (pool3d) -> tf.nn.max_pool / avg_pool (VALID, or SAME where the padded
    positions are excluded: not used by the reference, which pads explicitly)"""
    if padding != "valid":
        raise NotImplementedError("the reference pools with padding='valid' only")
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    perm_out = (0,) + tuple(range(2, nd + 2)) + (1,)
    xi = x.permute(perm_in)
    if pool_mode == "max":
        y = (F.max_pool2d if nd == 2 else F.max_pool3d)(xi, tuple(pool_size), tuple(strides))
    elif pool_mode == "avg":
        y = (F.avg_pool2d if nd == 2 else F.avg_pool3d)(xi, tuple(pool_size), tuple(strides))
    else:
        raise ValueError("Invalid pooling mode:", pool_mode)
    return y.permute(perm_out)


def pool2d(x, pool_size, strides=(1, 1), padding="valid", data_format=None, pool_mode="max"):
    assert (data_format or image_data_format()) == "channels_last"
    return _pool(x, pool_size, strides, padding, pool_mode, 2)


def pool3d(x, pool_size, strides=(1, 1, 1), padding="valid", data_format=None, pool_mode="max"):
    assert (data_format or image_data_format()) == "channels_last"
    return _pool(x, pool_size, strides, padding, pool_mode, 3)


def batch_normalization(x, mean, var, beta, gamma, epsilon=1e-3):
    """TFB:1667-1684 -> tf.nn.batch_normalization: inv = rsqrt(var + eps) * gamma; x * inv + (beta - mean * inv)"""
    inv = torch.rsqrt(var + epsilon)
    if gamma is not None:
        inv = inv * gamma
    return x * inv + ((beta if beta is not None else 0.0) - mean * inv)


def normalize_batch_in_training(x, gamma, beta, reduction_axes, epsilon=1e-3):
    """TFB:1620-1664 -> tf.nn.moments (mean, then mean of squared differences: biased variance) + the above"""
    axes = tuple(reduction_axes)
    m = x.mean(dim=axes)
    bshape = [1 if a in axes else x.shape[a] for a in range(x.dim())]
    v = ((x - m.reshape(bshape)) ** 2).mean(dim=axes)
    if sorted(axes) == list(range(x.dim()))[:-1]:
        normed = batch_normalization(x, m, v, beta, gamma, epsilon)
    else:
        normed = batch_normalization(x, m.reshape(bshape), v.reshape(bshape),
                                     None if beta is None else beta.reshape(bshape),
                                     None if gamma is None else gamma.reshape(bshape), epsilon)
    return normed, m, v


def l2_normalize(x, axis=None):
    return x / torch.sqrt(torch.clamp((x * x).sum(dim=axis, keepdim=True), min=_EPSILON))


def gradients(loss, variables):                                    # TFB:2300-2310 -> tf.gradients
    return torch.autograd.grad(loss, list(variables), allow_unused=True)


def stop_gradient(x):
    return x.detach()


def function(inputs, outputs, updates=None, **kwargs):
    raise NotImplementedError("eager backend: there is no session.run; call the model on tensors instead")
