"""The oracle (oracle/torch_ref.py) and the product's layer inventory, pinned to the REFERENCE'S OWN CODE.

tests/golden/ref_keras_<variant>.{json,npz} were produced by oracle/ref_keras/make_ref_fixtures.py: the reference's model
constructors (denseunet.py, densenet.py, denseunet3d.py, hybridnet.py), lib/custom_layers.py, loss.py and the vendored
Keras 2.0.8 layer classes, imported UNMODIFIED from /root/reference and executed over an eager torch `keras.backend`
(oracle/ref_keras/torch_backend.py restates only the ~40 primitive tensorflow_backend.py functions).  What these tests pin:

  * graph wiring, layer names, Keras weight shapes and order, trainable flags, BN epsilon / momentum / call-time `training`
    flags, conv strides / padding / bias, pool sizes, the loss's depth slicing -- everything above the primitive ops;
  * float64 outputs of the full-depth nets (predict and training-phase logits, loss, every trainable gradient, every
    moving-average update) against oracle/torch_ref.py with the same deterministic weights.

The fixtures travel (the reference tree does not exist on the GPU box); where /root/reference is present one variant is
also re-generated live and compared with the committed file.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as U                                  # noqa: E402
from oracle import torch_ref as R                         # noqa: E402
from oracle.ref_keras.weights import det_weights, digest   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
VARIANTS = ("denseunet", "densenet", "3dpart", "end2end")
NB2D, NB3D = (6, 12, 36, 24), (3, 4, 12, 8)
WEIGHT_CLASSES = ("Conv2D", "Conv3D", "BatchNormalization", "Scale")


def load_fixture(variant):
    with open(os.path.join(GOLD, "ref_keras_%s.json" % variant)) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLD, "ref_keras_%s.npz" % variant))
    return meta, z


def weight_layers(meta):
    """{layer name: (class, trainable, cfg, [(weight name, shape, trainable)])} for the layers that own weights"""
    return {n: (c, tr, cfg, ws) for n, c, tr, cfg, ws in meta["inventory"] if ws}


def oracle_with_reference_weights(variant, meta, x):
    kind = "2d" if variant in ("denseunet", "densenet") else "hybrid"
    fwd = U.oracle_forward_fn(kind, variant, NB2D, NB3D)
    P = R.ParamStore(seed=1, dtype=torch.float64, perturb=False)
    with torch.no_grad():
        fwd(P, torch.tensor(x))               # the oracle's graph definition creates ITS parameter inventory
    P.bn_batch_means = {}
    wl = weight_layers(meta)
    for name, (cls, _, _, ws) in wl.items():
        if name in P.w:
            arrs = det_weights(name, cls, [s for _, s, _ in ws])
            assert [tuple(a.shape) for a in arrs] == [tuple(t.shape) for t in P.w[name]], name
            P.w[name] = [torch.tensor(a) for a in arrs]
    return P, fwd, kind


def close(a, b, rtol=1e-8, atol=1e-10):
    return abs(a - b) <= atol + rtol * max(abs(a), abs(b))


def digest_matches(d, arr, rtol=1e-7, what=""):
    g = digest(arr)
    assert g["size"] == d["size"], what
    scale = max(d["norm"], 1e-30)
    assert abs(g["norm"] - d["norm"]) <= rtol * scale + 1e-14, (what, g["norm"], d["norm"])
    assert abs(g["sum"] - d["sum"]) <= rtol * scale * np.sqrt(d["size"]) + 1e-14, (what, g["sum"], d["sum"])
    assert g["idx"] == d["idx"]
    for a, b in zip(g["val"], d["val"]):
        assert abs(a - b) <= rtol * scale + 1e-14, (what, a, b)


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_inventory_equals_reference_constructor(variant):
    """names / Keras shapes / order / trainable flags / BN configuration of oracle/torch_ref.py == what the reference's
    constructor built"""
    meta, z = load_fixture(variant)
    P, fwd, kind = oracle_with_reference_weights(variant, meta, z["x"])
    wl = weight_layers(meta)
    # the `3dclassifer` conv is built by DenseNet3D but its output is discarded by both hybrid constructors
    # (denseunet3d.py:187,426; hybridnet.py:176,412): Keras leaves it out of the Model, and so do the oracle and the product
    assert set(wl) == set(P.w), (sorted(set(wl) ^ set(P.w))[:10])
    assert all(c in WEIGHT_CLASSES for c, _, _, _ in wl.values())
    for name, (cls, layer_trainable, cfg, ws) in wl.items():
        assert [tuple(s) for _, s, _ in ws] == [tuple(t.shape) for t in P.w[name]], name
        kind_o = P.kind[name]
        assert kind_o == {"Conv2D": "conv", "Conv3D": "conv", "BatchNormalization": "bn", "Scale": "scale"}[cls], name
        ref_trainable = [t for _, _, t in ws]
        if cls == "BatchNormalization":
            # Keras order gamma, beta, moving_mean, moving_variance (K.layers/normalization.py:97-123); statistics never train
            assert ref_trainable[2:] == [False, False], name
            assert ref_trainable[0] == ref_trainable[1] == P.trainable[name], name
            bc = P.bn_cfg[name]
            assert abs(bc["eps"] - cfg["epsilon"]) < 1e-12, (name, bc["eps"], cfg["epsilon"])
            call = meta["call_args"].get(name, [{}])[0]
            frozen = call.get("training", None) is False
            assert (bc["mode"] == "frozen") == frozen, (name, bc["mode"], call)
            if not frozen:       # momentum only matters where the moving statistics are updated
                assert abs(bc["momentum"] - cfg["momentum"]) < 1e-12, (name, bc["momentum"], cfg["momentum"])
            assert cfg["axis"] == len(meta["input_shape"]) - 1 or cfg["axis"] == -1 or cfg["axis"] in (3, 4)
        else:
            assert all(t == P.trainable[name] for t in ref_trainable), (name, ref_trainable, P.trainable[name])
            if cls != "Scale":
                assert (len(ws) == 2) == bool(cfg["use_bias"]), name


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_outputs_equal_reference_run(variant):
    """float64: predict logits, training-phase logits, loss.py loss, every trainable gradient and every BN moving-average
    update of the oracle == the reference's own code run over the eager backend"""
    meta, z = load_fixture(variant)
    x, y = z["x"], z["y"]
    P, fwd, kind = oracle_with_reference_weights(variant, meta, x)
    pred = R.predict(P, fwd, torch.tensor(x)).numpy()
    scale = max(1.0, float(np.abs(z["logits_predict"]).max()))
    assert np.abs(pred - z["logits_predict"]).max() <= 1e-9 * scale, np.abs(pred - z["logits_predict"]).max()
    before = {n: [t.clone() for t in ws] for n, ws in P.w.items()}
    loss, grads, out = R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), {}, lr=0.0, momentum=0.0)
    scale = max(1.0, float(np.abs(z["logits_train"]).max()))
    assert np.abs(out.numpy() - z["logits_train"]).max() <= 1e-9 * scale
    assert close(loss, meta["loss"], rtol=1e-10), (loss, meta["loss"])
    # exactly the reference's trainable tensors carry a gradient
    gd = meta["grad_digests"]
    assert set("%s/%d" % k for k in grads) == set(gd), sorted(set("%s/%d" % k for k in grads) ^ set(gd))[:10]
    for (name, i), g in grads.items():
        digest_matches(gd["%s/%d" % (name, i)], g.numpy(), what="grad %s/%d" % (name, i))
    # moving statistics: exactly the reference's updated BNs moved, to the reference's values (lr = 0: nothing else moved)
    ud = meta["bn_update_digests"]
    moved = set()
    for name, ws in P.w.items():
        if P.kind[name] != "bn":
            continue
        for i in (2, 3):
            key = "%s/%d" % (name, i)
            if key in ud:
                digest_matches(ud[key], ws[i].numpy(), what="moving statistic " + key)
                moved.add(key)
            else:
                assert torch.equal(ws[i], before[name][i]), "the reference does not update " + key
    assert moved == set(ud)


@pytest.mark.parametrize("variant", VARIANTS)
def test_product_inventory_equals_reference_constructor(variant, emu_lib):
    """the product's constructors (h-denseunet_amd/{denseunet,densenet,denseunet3d,hybridnet}.py) create the reference's
    layers: same names, Keras weight shapes and order, and the same set of trainable tensors"""
    meta, _ = load_fixture(variant)
    shp = meta["input_shape"]
    if variant in ("denseunet", "densenet"):
        mod = U.pkg("denseunet" if variant == "denseunet" else "densenet")
        m = mod.DenseUNet(reduction=0.5, args=U.make_args(shp[0], shp[1]), dtype="f32")
    elif variant == "3dpart":
        m = U.pkg("denseunet3d").denseunet_3d(U.make_args(1, shp[1], shp[3]), dtype="f32")
    else:
        m = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, shp[1], shp[3]), dtype="f32")
    assert m.name == meta["model_name"]
    wl = weight_layers(meta)
    ctx = m.ctx
    assert set(ctx.by_layer) == set(wl), sorted(set(ctx.by_layer) ^ set(wl))[:10]
    for name, (cls, _, _, ws) in wl.items():
        ps = ctx.by_layer[name]
        assert [tuple(p.keras_shape) for p in ps] == [tuple(s) for _, s, _ in ws], name
        assert [bool(p.trainable) for p in ps] == [bool(t) for _, _, t in ws], (name, [p.trainable for p in ps], ws)
    assert m.count_params() == meta["n_params"]


def test_fixture_regenerates_from_reference_tree():
    """where the reference tree exists (this container; not the GPU box) the committed fixture is what its code produces"""
    from oracle.ref_keras import harness as H
    if not H.available():
        pytest.skip("/root/reference is not present on this machine: the committed fixtures are the pin")
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); sys.dont_write_bytecode = True\n"
            "import oracle.ref_keras.make_ref_fixtures as M, numpy as np, os, tempfile\n"
            "d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, 'tests', 'golden')); M.ROOT = d\n"
            "meta = M.run('densenet'); z = np.load(os.path.join(d, 'tests', 'golden', 'ref_keras_densenet.npz'))\n"
            "g = np.load(%r)\n"
            "assert np.abs(z['logits_train'] - g['logits_train']).max() < 1e-12 and np.abs(z['logits_predict'] - g['logits_predict']).max() < 1e-12\n"
            "old = json.load(open(%r)); assert old['inventory'] == json.loads(json.dumps(meta['inventory'])) and abs(old['loss'] - meta['loss']) < 1e-12\n"
            "print('REGEN_OK')\n") % (ROOT, os.path.join(GOLD, "ref_keras_densenet.npz"), os.path.join(GOLD, "ref_keras_densenet.json"))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "REGEN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_sgd_step_equals_reference_optimizer(variant):
    """One optimiser step from a NON-ZERO previous velocity: the reference's own keras.optimizers.SGD(lr=1e-3, momentum=0.9,
    nesterov=True).get_updates (K.optimizers.py:155-185, run unmodified over the eager backend on the reference model's loss)
    produced the fixture's new velocities and new weights; oracle/torch_ref.py's train_step -- the restatement the product's
    hdu_sgd_nesterov is tested against -- must reproduce both for every trainable tensor."""
    meta, z = load_fixture(variant)
    assert meta["sgd"] == {"lr": 1e-3, "momentum": 0.9, "nesterov": True, "iterations_after": 1}
    x, y = z["x"], z["y"]
    P, fwd, kind = oracle_with_reference_weights(variant, meta, x)
    sd = meta["sgd_step_digests"]
    vel = {}
    for key in sd:
        name, i = key.rsplit("/", 1)
        vel[(name, int(i))] = torch.tensor(det_weights(key, "Moment", [tuple(P.w[name][int(i)].shape)])[0])
    R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), vel, lr=1e-3, momentum=0.9)
    assert set("%s/%d" % k for k in vel) == set(sd)
    for key, d in sd.items():
        name, i = key.rsplit("/", 1)
        digest_matches(d["v"], vel[(name, int(i))].numpy(), what="velocity " + key)
        digest_matches(d["p"], P.w[name][int(i)].numpy(), what="updated weight " + key)


def test_oracle_data_parallel_semantics_equal_reference_make_parallel():
    """What data parallelism COMPUTES in the reference, from its own K.utils2/multi_gpu.py:make_parallel run unmodified around
    denseunet.DenseUNet (oracle/ref_keras/make_parallel_fixture.py: 2 towers x 1 slice): BatchNormalization statistics per
    tower, loss.py's mean over the concatenated batch, gradients of that mean w.r.t. the shared weights, and one
    moving-average candidate per tower.  The oracle run tower by tower reproduces all of it -- the semantics the product's
    one-process-per-GPU path (local BN statistics, loss / world, summed gradients; tests/test_dp_gloo.py) is built to."""
    with open(os.path.join(GOLD, "ref_keras_make_parallel.json")) as f:
        pm = json.load(f)
    zp = np.load(os.path.join(GOLD, "ref_keras_make_parallel.npz"))
    meta, _ = load_fixture("denseunet")                       # the same network: its inventory names the weights
    x, y = zp["x"], zp["y"]
    P, fwd, kind = oracle_with_reference_weights("denseunet", meta, x[:1])
    P.learning_phase = 1
    trainable = P.trainable_tensors()
    for _, _, t in trainable:
        t.requires_grad_(True)
    outs, cands = [], []
    for i in range(pm["gpu_count"]):
        P.bn_batch_means = {}
        outs.append(fwd(P, torch.tensor(x[i:i + 1])))
        cands.append({n: (m.clone(), v.clone()) for n, (m, v) in P.bn_batch_means.items()})
    out = torch.cat(outs, 0)
    scale = max(1.0, float(np.abs(zp["logits_train"]).max()))
    assert np.abs(out.detach().numpy() - zp["logits_train"]).max() <= 1e-9 * scale
    loss = U.loss_fn_for(kind)(torch.tensor(y), out)
    assert close(float(loss.detach()), pm["loss"], rtol=1e-10)
    grads = torch.autograd.grad(loss, [t for _, _, t in trainable], allow_unused=True)
    gd = pm["grad_digests"]
    assert set("%s/%d" % (n, i) for n, i, _ in trainable) == set(gd)
    for (name, i, t), g in zip(trainable, grads):
        digest_matches(gd["%s/%d" % (name, i)], g.numpy(), what="make_parallel gradient %s/%d" % (name, i))
    # one moving-average candidate per tower, each from the OLD value (TFB:915-927)
    uc = pm["bn_update_candidates"]
    assert set(k.rsplit("/", 1)[0] for k in uc) == set(cands[0])
    for name in cands[0]:
        mom = P.bn_cfg[name]["momentum"]
        for tower in range(2):
            mean, var = cands[tower][name]
            digest_matches(uc[name + "/2"][tower], (P.w[name][2] - (P.w[name][2] - mean) * (1 - mom)).detach().numpy(),
                           what="moving mean of %s, tower %d" % (name, tower))
            digest_matches(uc[name + "/3"][tower], (P.w[name][3] - (P.w[name][3] - var) * (1 - mom)).detach().numpy(),
                           what="moving variance of %s, tower %d" % (name, tower))
    P.bn_batch_means = {}
