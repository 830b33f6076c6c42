"""oracle/ref_keras/fake_tf.py -- TEST INFRASTRUCTURE: the handful of `tf.*` functions the reference's model files and
loss.py call directly (loss.py:12-23, denseunet3d.py:374-390, hybridnet.py:362-376), on torch CPU tensors, so that those
files import and run unmodified next to oracle/ref_keras/torch_backend.py.  TensorFlow 1.x semantics are cited per function.
"""
import torch


def transpose(a, perm=None, name=None):
    """tf.transpose: permute the dimensions according to `perm` (default: reverse)"""
    if perm is None:
        perm = tuple(reversed(range(a.dim())))
    return a.permute(tuple(perm))


def expand_dims(input, axis=None, name=None, dim=None):
    return input.unsqueeze(axis if axis is not None else dim)


def reshape(tensor, shape, name=None):
    return tensor.reshape(tuple(shape))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    """tf.clip_by_value = minimum(maximum(t, lo), hi); the gradient passes where lo <= t <= hi"""
    return torch.clamp(t, clip_value_min, clip_value_max)


def where(condition, x=None, y=None, name=None):
    """tf.where(cond) with one argument: coordinates of the true elements, shape [n, rank] (row-major order)"""
    if x is None and y is None:
        return torch.nonzero(condition)
    return torch.where(condition, x, y)


def gather(params, indices, validate_indices=None, name=None, axis=0):
    """tf.gather(params, indices): output shape = indices.shape + params.shape[1:]"""
    return params[indices.long()]


def concat(values, axis, name="concat"):
    return torch.cat(list(values), dim=axis)


def equal(x, y, name=None):
    return torch.eq(x, y)


def log(x, name=None):
    return torch.log(x)


class _NN:
    @staticmethod
    def softmax(logits, dim=-1, name=None):
        return torch.softmax(logits, dim=dim)


nn = _NN()
float32 = torch.float32
float64 = torch.float64
int32 = torch.int32


# ---- K.utils2/multi_gpu.py:7-69 (make_parallel): tower placement is a no-op on one eager device
import contextlib      # noqa: E402


@contextlib.contextmanager
def device(name):
    """tf.device('/gpu:%d'): placement only -- the towers of make_parallel share their variables and differ in the batch slice"""
    yield


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name


def shape(input, name=None, out_type=None):
    """tf.shape: the dynamic shape as a 1-D integer tensor (multi_gpu.py:9 computes it and does not use it)"""
    return torch.tensor(list(input.shape), dtype=torch.int64)
