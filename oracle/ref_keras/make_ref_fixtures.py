#!/usr/bin/env python
"""oracle/ref_keras/make_ref_fixtures.py -- TEST INFRASTRUCTURE: golden vectors produced BY THE REFERENCE'S OWN CODE.

    PYTHONDONTWRITEBYTECODE=1 python oracle/ref_keras/make_ref_fixtures.py [variant ...]

For each model constructor of the reference (denseunet.DenseUNet, densenet.DenseUNet, denseunet3d.denseunet_3d,
hybridnet.dense_rnn_net -- imported unmodified from /root/reference over the eager backend of this directory) it
  1. builds the model on a small seeded input (2 x 32 x 32 x 3, or 1 x 32 x 32 x 8 x 1) at FULL depth (6/12/36/24 2D and
     3/4/12/8 3D layers: the constructors hard-code them),
  2. assigns deterministic weights (oracle/ref_keras/weights.py),
  3. runs predict (learning phase 0) and one training-phase forward + the reference's loss.py loss + gradients of every
     trainable weight + the BatchNormalization moving-average updates,
and writes tests/golden/ref_keras_<variant>.{json,npz}: the layer inventory (names, classes, weight shapes, trainable
flags, eps / momentum / call-time `training` flags, conv strides / padding / bias), the inputs, both logit tensors, the loss
and compact digests (norm, sum, sampled entries) of every gradient and updated moving statistic.  float64 throughout, so
tests/test_oracle_ref.py can hold oracle/torch_ref.py to ~1e-9.

Dropout is the identity in these runs (TensorFlow's mask stream cannot be reproduced; the layers and their rates are still
in the inventory).  /root/reference does not exist on the GPU box: only the committed fixtures travel.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.ref_keras import harness as H          # noqa: E402
from oracle.ref_keras.weights import det_weights, digest      # noqa: E402

SHAPES = {"denseunet": (2, 32, 32, 3), "densenet": (2, 32, 32, 3), "3dpart": (1, 32, 32, 8, 1), "end2end": (1, 32, 32, 8, 1)}


def inputs_for(variant, seed=1234):
    rng = np.random.default_rng(seed)
    shp = SHAPES[variant]
    x = rng.normal(0.0, 60.0, shp)
    y = rng.integers(0, 3, shp[:-1] + (1,)).astype(np.float64)
    return x, y


def _new_signature(opt):
    """Keras 2.0.8: get_updates(self, loss, params), wrapped by legacy_get_updates_support (K.legacy/interfaces.py)"""
    import inspect
    return list(inspect.signature(opt.get_updates).parameters)[:1] != ["params"]


def run(variant):
    K = H.setup("float64")
    x, y = inputs_for(variant)
    t0 = time.time()
    model = H.build(variant, torch.tensor(x), learning_phase=1)
    inv = H.inventory(model)
    # call-time arguments of every layer call (the authors freeze BNs with `(x, training=False)`)
    call_args = {}
    for layer in model.layers:
        nodes = getattr(layer, "inbound_nodes", [])
        args = [dict(n.arguments or {}) for n in nodes]
        if any(args):
            call_args[layer.name] = [{k: (v if isinstance(v, (int, float, bool, str, type(None))) else repr(v)) for k, v in a.items()}
                                     for a in args]
    for layer in model.layers:
        if layer.weights:
            layer.set_weights(det_weights(layer.name, layer.__class__.__name__, [tuple(w.shape) for w in layer.weights]))
    def feed():      # a fresh tensor object per call: Container.call caches outputs by id(input) (K.engine/topology.py:2040-2050)
        t = K._k(torch.tensor(x))
        t._keras_shape = tuple(x.shape)
        t._uses_learning_phase = False
        return t
    # ---- predict: learning phase 0 (K.engine/training.py:1659-1713)
    K.set_learning_phase(0)
    del K.PENDING_UPDATES[:]
    with torch.no_grad():
        logits_pred = model(feed()).detach().numpy().copy()
    del K.PENDING_UPDATES[:]
    # ---- one training-phase forward / loss / gradients / BN updates (K.engine/training.py:948-967)
    K.set_learning_phase(1)
    out = model(feed())
    loss = H.loss_fn(variant)(K._k(torch.tensor(y)), out)
    tw = list(model.trainable_weights)
    grads = torch.autograd.grad(loss, tw, allow_unused=True, retain_graph=True)
    owner = {}
    for layer in model.layers:
        for i, w in enumerate(layer.weights):
            owner[id(w)] = (layer.name, i)
    # ---- the reference's optimiser on the same loss: keras.optimizers.SGD(lr=1e-3, momentum=0.9, nesterov=True).get_updates
    # (K.optimizers.py:155-185, train_2ddense.py:181 / train_hybrid.py:149), executed unmodified.  Its moment variables are
    # created by K.zeros inside get_updates; they are handed a deterministic NON-ZERO previous velocity here
    # (weights.det_weights(<layer>/<index>, "Moment")) so that one step exercises the momentum and the Nesterov term.
    from keras.optimizers import SGD
    sgd = SGD(lr=1e-3, momentum=0.9, nesterov=True)
    order = [owner[id(w)] for w in tw]
    calls = []
    zeros_orig = K.zeros

    def zeros_moment(shape, dtype=None, name=None):
        name_i = order[len(calls)]
        calls.append(name_i)
        return K.variable(det_weights("%s/%d" % name_i, "Moment", [tuple(shape)])[0], dtype, name)
    K.zeros = zeros_moment
    try:
        updates = sgd.get_updates(loss, tw) if _new_signature(sgd) else sgd.get_updates(tw, {}, loss)
    finally:
        K.zeros = zeros_orig
    assert calls == order
    sgd_dig = {}
    by_var = {id(var): new for var, new in updates}
    for w, m in zip(tw, sgd.weights[1:]):
        name, i = owner[id(w)]
        sgd_dig["%s/%d" % (name, i)] = {"v": digest(by_var[id(m)].detach().numpy()), "p": digest(by_var[id(w)].detach().numpy())}
    gdig = {}
    for w, g in zip(tw, grads):
        name, i = owner[id(w)]
        gdig["%s/%d" % (name, i)] = digest(np.zeros(tuple(w.shape)) if g is None else g.detach().numpy())
    upd = {}
    for var, new in K.PENDING_UPDATES:
        name, i = owner[id(var)]
        upd["%s/%d" % (name, i)] = digest(new.detach().numpy())
    del K.PENDING_UPDATES[:]
    meta = {
        "variant": variant, "model_name": model.name, "input_shape": list(x.shape), "n_layers": len(model.layers),
        "floatx": "float64", "dropout": "identity", "loss": float(loss.detach()),
        "inventory": inv, "call_args": call_args, "grad_digests": gdig, "bn_update_digests": upd, "sgd_step_digests": sgd_dig,
        "sgd": {"lr": 1e-3, "momentum": 0.9, "nesterov": True, "iterations_after": int(by_var[id(sgd.iterations)])},
        "n_params": int(sum(int(np.prod(tuple(w.shape))) for layer in model.layers for w in layer.weights)),
        "generated_by": "oracle/ref_keras/make_ref_fixtures.py over /root/reference (xmengli/H-DenseUNet, Keras 2.0.8 vendored)",
        "seconds": round(time.time() - t0, 1),
    }
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "ref_keras_%s.json" % variant), "w") as f:
        json.dump(meta, f, indent=0, separators=(",", ":"))
    np.savez_compressed(os.path.join(gold, "ref_keras_%s.npz" % variant), x=x, y=y,
                        logits_train=out.detach().numpy(), logits_predict=logits_pred)
    print("%-10s %4d layers %9d params  loss %.6f  max|logit| train %.3f predict %.3f  %d trainable tensors, %d BN updates  (%.0f s)"
          % (variant, len(model.layers), meta["n_params"], meta["loss"], np.abs(out.detach().numpy()).max(),
             np.abs(logits_pred).max(), len(tw), len(upd), time.time() - t0))
    return meta


if __name__ == "__main__":
    if not H.available():
        sys.exit("the reference tree is not present: fixtures can only be regenerated where /root/reference exists")
    for v in (sys.argv[1:] or H.VARIANTS):
        run(v)
