"""Shared helpers of the model-level parity tests: build the oracle + the product with identical (perturbed)
weights, run one step on both, compare.  TEST INFRASTRUCTURE (imports oracle/)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import torch_ref as R  # noqa: E402

# The oracle's 161-layer nets are thousands of SMALL CPU ops: on the 128-core GPU host torch's default team of 128 threads spends its
# time in fork / join (measured there, round 5, `tools/oracle_pair_timing.py`: one predict + train step of the float32 and the
# bf16-storage oracle of dense_rnn_net at 224 x 224 x 12 -- 59.5 s with 128 threads, 23.6 s with 64, 15.0 s with 32).  The GPU tier's time
# is mostly this oracle, so the tests cap the team at 32 threads; results do not depend on the thread count.
ORACLE_THREADS = 32
if torch.get_num_threads() > ORACLE_THREADS:
    torch.set_num_threads(ORACLE_THREADS)


def pkg(mod=None):
    return importlib.import_module("h-denseunet_amd" + ("." + mod if mod else ""))


def make_args(b, size, cols=None):
    return types.SimpleNamespace(b=b, input_size=size, input_cols=cols)


def oracle_forward_fn(kind, variant, nb2d, nb3d):
    if kind == "2d":
        return lambda P, x: R.dense_unet_2d(P, x, variant=variant, nb_layers=nb2d)[1]
    if kind == "3d":
        return lambda P, x: R.dense_net_3d_standalone(P, x, nb_layers=nb3d)
    return lambda P, x: R.hybrid_net(P, x, variant=variant, nb_layers2d=nb2d, nb_layers3d=nb3d)


def synthetic_batch(kind, b, size, cols, seed=1234):
    if kind == "3d":   # 4-channel volume: CT + three pseudo-probability channels (what the 2D branch would supply)
        x, y = pkg("synth").synthetic_batch("hybrid", b, size, cols, seed)
        rng = np.random.default_rng(seed + 1)
        extra = rng.normal(0.0, 60.0, x.shape[:4] + (3,)).astype(np.float32)
        return np.concatenate([x, extra], -1), y
    return pkg("synth").synthetic_batch(kind, b, size, cols, seed)


def build_pair(kind, variant, b, size, cols, dtype, nb2d, nb3d, seed=4321, odtype=torch.float64, perturb=True):
    """returns (product model, oracle ParamStore, oracle forward fn) with identical weights"""
    P = R.ParamStore(seed=seed, dtype=odtype, perturb=perturb)
    fwd = oracle_forward_fn(kind, variant, nb2d, nb3d)
    x, _ = synthetic_batch(kind, b, size, cols)
    with torch.no_grad():
        fwd(P, torch.tensor(x, dtype=odtype))   # creates the parameters
    P.bn_batch_means = {}
    if kind == "2d":
        mod = pkg("denseunet" if variant == "denseunet" else "densenet")
        m = mod.DenseUNet(reduction=0.5, args=make_args(b, size), dtype=dtype, nb_layers=nb2d)
    elif kind == "3d":
        m = pkg("densenet3d_sharded").dense_net3d(make_args(b, size, cols), dtype=dtype, nb_layers3d=nb3d)
    elif variant == "3dpart":
        m = pkg("denseunet3d").denseunet_3d(make_args(b, size, cols), dtype=dtype, nb_layers2d=nb2d, nb_layers3d=nb3d)
    else:
        m = pkg("hybridnet").dense_rnn_net(make_args(b, size, cols), dtype=dtype, nb_layers2d=nb2d, nb_layers3d=nb3d)
    ow = P.numpy()
    assert list(ow.keys()) == m.layer_names() or set(ow.keys()) == set(m.layer_names()), \
        "layer inventories differ: %s" % (set(ow.keys()) ^ set(m.layer_names()))
    m.set_weights_dict(ow)
    return m, P, fwd


def loss_fn_for(kind):
    # the stand-alone 3D net is trained on every voxel (the 1:7 slice of loss.py:6-7 belongs to the 8-slice hybrid)
    return R.weighted_crossentropy if kind == "hybrid" else R.weighted_crossentropy_2ddense


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def dice_vs_oracle(logits, ref_logits):
    return R.dice_per_class(np.argmax(logits, -1), np.argmax(ref_logits, -1))
