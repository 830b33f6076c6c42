"""Generates tests/golden/keras_weights_tiny.h5 and keras_model_tiny.hdf5 -- TEST INFRASTRUCTURE.

Writes, with the real HDF5 library, small files in exactly the layout Keras 2.0.8 produces
(K.engine/topology.py:2847-2873 `save_weights_to_hdf5_group`; `Model.save` puts the same group under
/model_weights, K.models.py): root attributes layer_names / backend / keras_version (fixed-length byte strings), one
group per layer with a weight_names attribute and one dataset per weight whose name contains a '/' (so it lives in a
nested sub-group, e.g. /conv1/conv1/kernel:0).  The pure-Python reader h-denseunet_amd/h5lite.py is tested against these
files (tests/test_h5_import.py).

h5py is NOT installed in the default interpreter of this image; run with the conda one that has it:
    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py
The expected array values are re-derivable from the seed below (numpy default_rng(7), float32), so the test needs no
h5py.
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# (layer name, [(weight name, shape)]) -- a cut-down DenseUNet naming sample incl. a layer without weights
LAYERS = [
    ("data", []),
    ("conv1", [("conv1/kernel:0", (7, 7, 3, 8))]),
    ("conv1_bn", [("conv1_bn/gamma:0", (8,)), ("conv1_bn/beta:0", (8,)), ("conv1_bn/moving_mean:0", (8,)),
                  ("conv1_bn/moving_variance:0", (8,))]),
    ("conv1_scale", [("conv1_scale/conv1_scale_gamma:0", (8,)), ("conv1_scale/conv1_scale_beta:0", (8,))]),
    ("conv2_1_x1", [("conv2_1_x1/kernel:0", (1, 1, 8, 16))]),
    ("conv_up4", [("conv_up4/kernel:0", (3, 3, 16, 8)), ("conv_up4/bias:0", (8,))]),
    ("dense167classifer", [("dense167classifer/kernel:0", (1, 1, 8, 3)), ("dense167classifer/bias:0", (3,))]),
    ("3dconv1", [("3dconv1/kernel:0", (3, 3, 3, 4, 8))]),
    ("scalar_holder", [("scalar_holder/iterations:0", ())]),
] + [("filler_%02d" % i, [("filler_%02d/kernel:0" % i, (2, 3))]) for i in range(40)]   # > 32 links: multi-node B-tree


def values():
    rng = np.random.default_rng(7)
    out = {}
    for lname, ws in LAYERS:
        for wname, shape in ws:
            out[wname] = rng.normal(0, 1, shape).astype(np.float32)
    return out


def write_group(f, vals, fixed):
    """fixed=True: fixed-length byte strings, what h5py 2.x (the Keras 2.0.8 era) wrote for lists of bytes;
    fixed=False: variable-length strings, what h5py 3.x writes for the same Python objects"""
    def S(b):
        return np.bytes_(b) if fixed else b

    def SL(lst):
        return np.array(lst, dtype="S") if (fixed and lst) else (np.array([], dtype="S1") if fixed else lst)
    f.attrs["layer_names"] = SL([n.encode("utf8") for n, _ in LAYERS])
    f.attrs["backend"] = S("tensorflow".encode("utf8"))
    f.attrs["keras_version"] = S("2.0.8".encode("utf8"))
    for lname, ws in LAYERS:
        g = f.create_group(lname)
        g.attrs["weight_names"] = SL([w.encode("utf8") for w, _ in ws])
        for wname, shape in ws:
            d = g.create_dataset(wname, shape, dtype=np.float32)
            if not shape:
                d[()] = vals[wname]
            else:
                d[:] = vals[wname]


def main():
    vals = values()
    with h5py.File(os.path.join(HERE, "keras_weights_tiny.h5"), "w") as f:      # model.save_weights(...)
        write_group(f, vals, fixed=True)
    with h5py.File(os.path.join(HERE, "keras_model_tiny.hdf5"), "w") as f:      # model.save(...): weights under /model_weights
        f.attrs["keras_version"] = "2.0.8".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = ('{"class_name": "Model", "config": {"name": "denseu161"}}').encode("utf8")
        write_group(f.create_group("model_weights"), vals, fixed=False)
    for n in ("keras_weights_tiny.h5", "keras_model_tiny.hdf5"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
