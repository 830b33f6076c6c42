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


# ---- reproducible training of the parity tests' weights.  Every run of the bf16 parity tests trains its own
# weights in the product's float32 mode; with float atomics in the statistics / BN-backward / filter-gradient reductions two runs of
# one commit drew different weights (tests/determinism_probe.py: ~all parameters differ after 30 steps).  This recipe of EXISTING
# switches routes every such reduction through its two-level (per-workgroup partials + fixed-order finalize) or single-writer form;
# measured on MI355X (profiles/r06_determinism.txt): forward outputs, BN / bias gradients AND filter gradients bit-equal between two
# executions, the trained weights bit-equal between two trainings.
ORDERED_ENV = {"HDU_EPILOGUE_STATS": "0",        # batch moments by the two-pass reduction instead of conv-epilogue float atomics
               "HDU_BN_BWD_FUSED": "0",          # BN backward sums by hdu_bn_bwd_reduce_coef (partials + finalize), no slot-table atomics
               "HDU_FUSE_BN_BWD": "0", "HDU_BNB_SUMS_EPILOGUE": "0", "HDU_FUSE_BN_BWD_PW": "0"}     # no BN-backward sums in conv epilogues
ORDERED_TUNING = {2: 1}                          # include/hdu.h HDU_TUNE_WGRAD_TARGET_WGS = 1: every dw element has ONE writer


class ordered_reductions:
    """context manager: a model BUILT and TRAINED inside it runs atomics-free (bit-reproducible) reductions"""

    def __enter__(self):
        self.prev = {k: os.environ.get(k) for k in ORDERED_ENV}
        os.environ.update(ORDERED_ENV)
        lib = pkg("lib").get()
        for k, v in ORDERED_TUNING.items():
            lib.hdu_set_tuning(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib = pkg("lib").get()
        for k in ORDERED_TUNING:
            lib.hdu_set_tuning(k, 0)
