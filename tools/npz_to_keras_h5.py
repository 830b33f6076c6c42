"""Export: weights saved by this package (`Model.save_weights` -> .npz, keys "<layer>/<index>", Keras shapes and per-layer
order) -> a Keras 2.0.8 HDF5 weight file the reference can `load_weights` (K.engine/topology.py:2847-2873 layout).
Needs h5py, which the default interpreter of this image lacks:  /opt/conda/bin/python3.9 tools/npz_to_keras_h5.py in.npz out.h5
(The other direction needs nothing: Model.load_weights reads Keras HDF5 files with the built-in h5lite reader.)"""
import sys
from collections import OrderedDict

import h5py
import numpy as np


def weight_names(layer, arrs):
    if arrs and arrs[0].ndim >= 4:                                   # Conv2D / Conv3D
        return [layer + "/kernel:0", layer + "/bias:0"][:len(arrs)]
    if len(arrs) == 4:                                               # BatchNormalization
        return [layer + "/" + n for n in ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0")]
    if len(arrs) == 2:                                               # Scale (lib/custom_layers.py:53-57)
        return ["%s/%s_gamma:0" % (layer, layer), "%s/%s_beta:0" % (layer, layer)]
    return ["%s/param_%d" % (layer, i) for i in range(len(arrs))]


def main(src, dst):
    z = np.load(src, allow_pickle=False)
    layers = OrderedDict()
    for k in z.files:
        if k == "__model_name__":
            continue
        n, i = k.rsplit("/", 1)
        layers.setdefault(n, {})[int(i)] = z[k]
    with h5py.File(dst, "w") as f:
        f.attrs["layer_names"] = np.array([n.encode("utf8") for n in layers], dtype="S")
        f.attrs["backend"] = np.bytes_(b"tensorflow")
        f.attrs["keras_version"] = np.bytes_(b"2.0.8")
        for n, d in layers.items():
            arrs = [d[i] for i in sorted(d)]
            names = weight_names(n, arrs)
            g = f.create_group(n)
            g.attrs["weight_names"] = np.array([w.encode("utf8") for w in names], dtype="S")
            for w, a in zip(names, arrs):
                g.create_dataset(w, data=np.asarray(a, np.float32))
    print("wrote %s: %d layers" % (dst, len(layers)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
