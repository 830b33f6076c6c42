"""N2: Keras-HDF5 weight import without h5py.  The fixtures were written by the real HDF5 library
in the layout of Keras 2.0.8's save_weights / save (tests/golden/make_keras_h5.py); the expected values are re-derived
from that script's seed."""
import importlib.util
import os

import numpy as np
import pytest

import parity_utils as U

GOLD = os.path.join(U.ROOT, "tests", "golden")


def _gen():
    src = open(os.path.join(GOLD, "make_keras_h5.py")).read().replace("import h5py\n", "")   # only LAYERS / values() are used
    ns = {"__file__": os.path.join(GOLD, "make_keras_h5.py"), "__name__": "gen"}
    exec(compile(src, "make_keras_h5", "exec"), ns)
    return ns


@pytest.mark.parametrize("fname", ["keras_weights_tiny.h5", "keras_model_tiny.hdf5"])
def test_reads_keras_files_written_by_libhdf5(fname):
    h5 = U.pkg("h5lite")
    gen = _gen()
    vals = gen["values"]()
    w = h5.read_keras_weights(os.path.join(GOLD, fname))
    assert list(w.keys()) == [n for n, _ in gen["LAYERS"]]
    for lname, ws in gen["LAYERS"]:
        assert len(w[lname]) == len(ws)
        for arr, (wname, shape) in zip(w[lname], ws):
            assert arr.shape == tuple(shape) and arr.dtype == np.float32
            np.testing.assert_array_equal(arr, vals[wname])
    f = h5.File(os.path.join(GOLD, fname))
    g = f["model_weights"] if fname.endswith("hdf5") else f
    kv = g.attrs["keras_version"]
    assert (kv if isinstance(kv, bytes) else bytes(kv)) == b"2.0.8"
    assert g["conv1/conv1/kernel:0"].shape == (7, 7, 3, 8)
    assert "conv1" in g and "nope" not in g
    with pytest.raises(KeyError):
        g["conv1/missing"]


def test_rejects_non_hdf5(tmp_path):
    h5 = U.pkg("h5lite")
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(h5.H5Error):
        h5.File(str(p))


CONDA_PY = "/opt/conda/bin/python3.9"


def _have_h5py():
    import subprocess
    return os.path.exists(CONDA_PY) and subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _have_h5py(), reason="needs an interpreter with h5py to WRITE the Keras file (reading needs none)")
def test_model_round_trip_through_keras_hdf5(emu_lib, tmp_path):
    """save_weights (.npz) -> tools/npz_to_keras_h5.py (real libhdf5, Keras 2.0.8 layout) -> Model.load_weights(.h5):
    every weight comes back bit for bit; by_name semantics for files with extra / missing layers."""
    import subprocess
    mk = lambda seed: U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(1, 32), dtype="f32",
                                                     nb_layers=(2, 2, 2, 2), seed=seed)
    m = mk(3)
    npz, h5 = str(tmp_path / "w.npz"), str(tmp_path / "w.h5")
    m.save_weights(npz)
    out = subprocess.run([CONDA_PY, os.path.join(U.ROOT, "tools", "npz_to_keras_h5.py"), npz, h5], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    m2 = mk(11)
    m2.load_weights(h5)                                   # topological form: every layer must be present
    for n, arrs in m.get_weights_dict().items():
        for a, b in zip(arrs, m2.get_weights_dict()[n]):
            np.testing.assert_array_equal(a, b)
    # a hybrid model shares the 2D layer names: by_name loads them and leaves the 3D layers alone
    hy = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 32, 8), dtype="f32", nb_layers2d=(2, 2, 2, 2), nb_layers3d=(1, 1, 2, 1))
    before3d = hy.get_weights_dict()["3dconv1"][0].copy()
    with pytest.raises(ValueError):
        hy.load_weights(h5)                               # 'line0' is not part of the hybrid's 2D branch
    hy.load_weights(h5, by_name=True)
    np.testing.assert_array_equal(hy.get_weights_dict()["conv2_1_x1"][0], m.get_weights_dict()["conv2_1_x1"][0])
    np.testing.assert_array_equal(hy.get_weights_dict()["3dconv1"][0], before3d)
