#!/usr/bin/env python
"""oracle/ref_keras/make_parallel_fixture.py -- TEST INFRASTRUCTURE: what the reference's data parallelism COMPUTES, from the
reference's own code: Keras-2.0.8/keras/utils2/multi_gpu.py:make_parallel (the authors' file; train_2ddense.py:180) wrapped
around denseunet.DenseUNet, both imported unmodified from /root/reference over the eager backend.

    PYTHONDONTWRITEBYTECODE=1 python oracle/ref_keras/make_parallel_fixture.py

make_parallel(model, gpu_count=2, mini_batch=1): tower i runs the SAME model (shared variables) on batch slice i, the tower
outputs are concatenated on axis 0.  Consequences the product's one-process-per-GPU data parallelism has to reproduce
(DESIGN.md section 6): BatchNormalization batch statistics are PER TOWER (each tower's BN sees its slice only), the loss is
loss.py's mean over the concatenated batch (= the mean of the tower means), the gradient is d(that mean)/d(shared weights)
(= the towers' gradients of their own mean, averaged), and every tower emits its own moving-average update of the same
variable from the same old value (TensorFlow runs both assigns in one session.run: one of them survives).
Recorded: the concatenated training-phase logits, the loss, digests of every gradient, and BOTH towers' moving-average
candidates per BN.  Writes tests/golden/ref_keras_make_parallel.{json,npz}.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.ref_keras import harness as H          # noqa: E402
from oracle.ref_keras.weights import det_weights, digest      # noqa: E402


def main():
    K = H.setup("float64")
    rng = np.random.default_rng(4321)
    x = rng.normal(0.0, 60.0, (2, 32, 32, 3))
    y = rng.integers(0, 3, (2, 32, 32, 1)).astype(np.float64)
    model = H.build("denseunet", torch.tensor(x), learning_phase=1)
    for layer in model.layers:
        if layer.weights:
            layer.set_weights(det_weights(layer.name, layer.__class__.__name__, [tuple(w.shape) for w in layer.weights]))
    from keras.utils2.multi_gpu import make_parallel
    K.FEED.append(torch.tensor(x))
    pm = make_parallel(model, 2, mini_batch=1)

    def feed():
        t = K._k(torch.tensor(x))
        t._keras_shape = tuple(x.shape)
        t._uses_learning_phase = False
        return t
    K.set_learning_phase(1)
    del K.PENDING_UPDATES[:]
    out = pm(feed())
    assert tuple(out.shape) == (2, 32, 32, 3)
    loss = H.loss_fn("denseunet")(K._k(torch.tensor(y)), out)
    tw = list(model.trainable_weights)
    grads = torch.autograd.grad(loss, tw, allow_unused=True)
    owner = {}
    for layer in model.layers:
        for i, w in enumerate(layer.weights):
            owner[id(w)] = (layer.name, i)
    gdig = {"%s/%d" % owner[id(w)]: digest(g.detach().numpy()) for w, g in zip(tw, grads)}
    upd = {}
    for var, new in K.PENDING_UPDATES:
        upd.setdefault("%s/%d" % owner[id(var)], []).append(digest(new.detach().numpy()))
    del K.PENDING_UPDATES[:]
    assert all(len(v) == 2 for v in upd.values()), "one moving-average candidate per tower"
    meta = {"gpu_count": 2, "mini_batch": 1, "loss": float(loss.detach()), "grad_digests": gdig, "bn_update_candidates": upd,
            "generated_by": "oracle/ref_keras/make_parallel_fixture.py over /root/reference (K.utils2/multi_gpu.py:make_parallel)"}
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "ref_keras_make_parallel.json"), "w") as f:
        json.dump(meta, f, indent=0, separators=(",", ":"))
    np.savez_compressed(os.path.join(gold, "ref_keras_make_parallel.npz"), x=x, y=y, logits_train=out.detach().numpy())
    print("make_parallel(DenseUNet, 2 towers x 1 slice): loss %.6f, %d gradients, %d BN variables x 2 candidates" %
          (meta["loss"], len(gdig), len(upd)))


if __name__ == "__main__":
    if not H.available():
        sys.exit("the reference tree is not present")
    main()
