"""N2 export side: Model.save_weights / Model.save / ModelCheckpoint write REAL Keras-2.0.8-layout HDF5 files natively
(h5lite writer; no h5py in this image), and the author's multi-GPU loaders (K.engine/topology.py:3171-3330) read the
nested layouts a `make_parallel` checkpoint has.  Files are read back with the pure-Python reader and, where the conda
interpreter with the real HDF5 library exists, with h5py itself."""
import json
import os
import subprocess

import numpy as np
import pytest

import parity_utils as U

CONDA_PY = "/opt/conda/bin/python3.9"


def _have_h5py():
    return os.path.exists(CONDA_PY) and subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True).returncode == 0


def _mk(seed, dtype="f32"):
    return U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(1, 32), dtype=dtype, nb_layers=(2, 2, 2, 2), seed=seed)


def _same(a, b):
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    assert list(wa) == list(wb)
    for n in wa:
        for x, y in zip(wa[n], wb[n]):
            np.testing.assert_array_equal(x, y)


def test_writer_many_links_and_scalars(tmp_path):
    """> 8 links per symbol node and > 32 symbol nodes per B-tree node: the multi-level group B-tree; rank-0 datasets"""
    h5 = U.pkg("h5lite")
    root = h5.WGroup()
    rng = np.random.default_rng(0)
    layers = [("data", [])] + [("l%03d" % i, [("l%03d/kernel:0" % i, rng.normal(size=(2, 3)).astype(np.float32)),
                                              ("l%03d/bias:0" % i, rng.normal(size=(3,)).astype(np.float32))])
                               for i in range(300)]
    h5.keras_weights_group(root, layers)
    root.group("optimizer_weights").dataset("SGD/iterations:0", np.array(7, dtype=np.int64))
    p = str(tmp_path / "many.h5")
    h5.write_file(p, root)
    d = h5.read_keras_weights(p)
    assert list(d) == [n for n, _ in layers]
    for n, ws in layers:
        for (_, a), b in zip(ws, d[n]):
            np.testing.assert_array_equal(a, b)
    it = np.asarray(h5.File(p)["optimizer_weights/SGD/iterations:0"])
    assert it.shape == () and int(it) == 7
    with pytest.raises(h5.H5Error):                      # HDF5's 64 KB object-header-message limit is reported, not hit blindly
        big = h5.WGroup()
        big.attrs["model_config"] = "x" * 70000
        h5.write_file(str(tmp_path / "big.h5"), big)


def test_save_weights_hdf5_round_trip_and_names(emu_lib, tmp_path):
    m = _mk(3)
    p = str(tmp_path / "w.h5")
    m.save_weights(p)
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    h5 = U.pkg("h5lite")
    f = h5.File(p)
    assert bytes(f.attrs["keras_version"]).rstrip(b"\0") == b"2.0.8"
    wn = [bytes(b).rstrip(b"\0").decode() for b in np.asarray(f["conv1_scale"].attrs["weight_names"])]
    assert wn == ["conv1_scale/conv1_scale_gamma:0", "conv1_scale/conv1_scale_beta:0"]      # lib/custom_layers.py:53-57
    assert np.asarray(f["conv_up4/conv_up4/kernel:0"]).shape == (3, 3, 96, 64)
    m2 = _mk(11)
    m2.load_weights(p)
    _same(m, m2)


def test_full_save_restores_optimizer_state(emu_lib, tmp_path):
    ka = U.pkg("keras_api")
    m = _mk(3)
    m.ctx.dropout_enabled = False
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    x, y = U.synthetic_batch("2d", 1, 32, None)
    m.train_on_batch(x, y)
    m.train_on_batch(x, y)
    p = str(tmp_path / "model_best.hdf5")
    m.save(p)
    f = U.pkg("h5lite").File(p)
    tc = json.loads(bytes(f.attrs["training_config"]).rstrip(b"\0").decode())
    assert tc["optimizer_config"]["config"]["nesterov"] is True and "model_weights" in f and "optimizer_weights" in f
    m2 = _mk(11)
    m2.ctx.dropout_enabled = False
    m2.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    m2.load_weights(p)                                   # /model_weights of a full-model file (topology.py:2621-2622)
    m2.load_optimizer_weights(p)
    assert m2.optimizer.iterations == 2
    np.testing.assert_array_equal(m2.ctx.V.cpu().numpy(), m.ctx.V.cpu().numpy())
    # resumed training continues exactly where the first model would
    l1, l2 = m.train_on_batch(x, y), m2.train_on_batch(x, y)
    assert abs(l1 - l2) <= 1e-6 * abs(l1)
    np.testing.assert_allclose(m2.ctx.P.cpu().numpy(), m.ctx.P.cpu().numpy(), rtol=1e-5, atol=1e-5)   # float atomics order
    # the .npz container carries the same state
    q = str(tmp_path / "model.npz")
    m.save(q)
    m3 = _mk(12)
    m3.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    m3.load_weights(q)
    m3.load_optimizer_weights(q)
    assert m3.optimizer.iterations == 3
    np.testing.assert_array_equal(m3.ctx.V.cpu().numpy(), m.ctx.V.cpu().numpy())


def test_author_multi_gpu_loaders(emu_lib, tmp_path):
    """train_hybrid.py:146 `load_weights(w, by_name=True, by_gpu=True, two_model=True, by_flag=True)`: the 2D DenseUNet
    of a make_parallel pre-training run (layers under /denseu161, no weight_names attributes: link order + swap) into
    the hybrid; `by_gpu` alone reads /model_1.  A name-based load that matches nothing raises (round 1 silently kept
    the random weights)."""
    m = U.pkg("densenet").DenseUNet(reduction=0.5, args=U.make_args(1, 32), dtype="f32", nb_layers=(2, 2, 2, 2), seed=5)
    p1, p2, p3 = (str(tmp_path / n) for n in ("par_model1.h5", "par_denseu161.h5", "par_auto3d.h5"))
    m.save_weights(p1, nested_under="model_1")
    m.save_weights(p2, nested_under="denseu161")
    hy = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 32, 8), dtype="f32", nb_layers2d=(2, 2, 2, 2), nb_layers3d=(1, 1, 2, 1))
    hy.save_weights(p3, nested_under="auto3d_residual_conv")
    f = U.pkg("h5lite").File(p1)
    assert f.keys() == ["model_1"] and "conv1_bn" in f["model_1"].keys()
    assert f["model_1/conv1_bn"].keys() == ["beta:0", "gamma:0", "moving_mean:0", "moving_variance:0"]   # link (name) order

    def fresh():
        return U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 32, 8), dtype="f32", nb_layers2d=(2, 2, 2, 2),
                                                nb_layers3d=(1, 1, 2, 1), seed=99)
    ref = m.get_weights_dict()
    for path, kw in ((p1, dict(by_gpu=True)), (p2, dict(by_gpu=True, two_model=True, by_flag=True))):
        h = fresh()
        before3d = h.get_weights_dict()["3dconv1"][0].copy()
        h.load_weights(path, by_name=True, **kw)
        got = h.get_weights_dict()
        for n in ("conv1", "conv1_bn", "conv1_scale", "conv2_1_x2", "conv_up4", "dense167classifer"):
            for a, b in zip(ref[n], got[n]):
                np.testing.assert_array_equal(a, b)       # incl. the (bias, kernel) / (beta, gamma, ...) swap
        np.testing.assert_array_equal(got["3dconv1"][0], before3d)
    h = fresh()
    h.load_weights(p3, by_name=True, by_gpu=True, two_model=True, by_flag=False)      # /auto3d_residual_conv
    _sameish = hy.get_weights_dict()
    for n in ("3dconv1", "fianl_conv", "final_bn", "2d3dclassifer", "conv1"):
        for a, b in zip(_sameish[n], h.get_weights_dict()[n]):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError, match="nothing would be loaded"):
        fresh().load_weights(p1, by_name=True)            # plain by_name sees one layer called 'model_1'
    with pytest.raises(KeyError):
        fresh().load_weights(p1, by_name=True, by_gpu=True, two_model=True, by_flag=True)   # no /denseu161 in this file


def test_model_checkpoint_writes_keras_hdf5(emu_lib, tmp_path):
    """train_2ddense.py:190-203: ModelCheckpoint('.../weights.{epoch:02d}-{loss:.2f}.hdf5') -> a file Keras can open"""
    ka = U.pkg("keras_api")
    m = _mk(3)
    cb = ka.ModelCheckpoint(str(tmp_path / "model" / "weights.{epoch:02d}-{loss:.2f}.hdf5"), monitor="loss", verbose=1,
                            save_best_only=False, save_weights_only=False, mode="min", period=1)
    cb.set_model(m)
    cb.on_epoch_end(0, {"loss": 1.234})
    p = tmp_path / "model" / "weights.00-1.23.hdf5"
    assert p.exists() and p.read_bytes()[:8] == b"\x89HDF\r\n\x1a\n"
    m2 = _mk(8)
    m2.load_weights(str(p))
    _same(m, m2)


@pytest.mark.skipif(not _have_h5py(), reason="cross-check against the real HDF5 library needs the conda interpreter")
def test_files_open_with_real_libhdf5(emu_lib, tmp_path):
    ka = U.pkg("keras_api")
    m = _mk(3)
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    p = str(tmp_path / "full.hdf5")
    m.save(p)
    np.save(str(tmp_path / "k.npy"), m.get_weights_dict()["conv_up0"][0])
    code = ("import h5py, numpy as np, json, sys\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "g = f['model_weights']\n"
            "names = [n.decode() for n in g.attrs['layer_names']]\n"
            "n = 0\n"
            "for ln in names:\n"
            "    for wn in g[ln].attrs['weight_names']:\n"
            "        g[ln][wn.decode()][()]; n += 1\n"
            "assert np.array_equal(g['conv_up0']['conv_up0/kernel:0'][()], np.load(sys.argv[2]))\n"
            "json.loads(f.attrs['model_config']); json.loads(f.attrs['training_config'])\n"
            "assert f['optimizer_weights']['SGD/iterations:0'][()] == 0\n"
            "print('LIBHDF5_OK', len(names), n)\n")
    out = subprocess.run([CONDA_PY, "-c", code, p, str(tmp_path / "k.npy")], capture_output=True, text=True)
    assert out.returncode == 0 and "LIBHDF5_OK" in out.stdout, out.stderr[-3000:]
