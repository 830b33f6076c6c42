"""oracle/ref_keras/weights.py -- TEST INFRASTRUCTURE: deterministic, well-conditioned weights keyed by Keras layer name.

Both sides of the reference-pin test use this: the fixture generator assigns these arrays to the layers of the model the
reference's constructor built (Layer.set_weights), the test assigns the same arrays to the oracle's ParamStore and to the
product.  The full nets hold 16-60 M parameters, far too many to commit, so the fixture stores only names / shapes / flags
and the arrays are re-derived from (layer name, weight index, shape).
"""
import zlib

import numpy as np


def det_weights(layer_name, class_name, shapes):
    """arrays (float64) for the weights of one layer, in Keras' own order (layer.weights):
    Conv2D/3D [kernel(, bias)], BatchNormalization [gamma, beta, moving_mean, moving_variance], Scale [gamma, beta]"""
    out = []
    for idx, shape in enumerate(shapes):
        rng = np.random.default_rng([zlib.crc32(layer_name.encode()), idx])
        shape = tuple(shape)
        if class_name in ("Conv2D", "Conv3D"):
            if idx == 0:
                fan_in = int(np.prod(shape[:-1]))
                fan_out = int(np.prod(shape[:-2])) * shape[-1]
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-lim, lim, shape)
            else:
                a = rng.normal(0.0, 0.05, shape)
        elif class_name == "BatchNormalization":
            a = (rng.uniform(0.5, 1.5, shape), rng.normal(0.0, 0.1, shape), rng.normal(0.0, 0.1, shape),
                 rng.uniform(0.5, 1.5, shape))[idx]
        elif class_name == "Scale":
            a = (rng.uniform(0.5, 1.5, shape), rng.normal(0.0, 0.1, shape))[idx]
        elif class_name == "Moment":       # a previous SGD velocity (the optimiser-step fixture starts from a non-zero one)
            a = rng.normal(0.0, 1e-3, shape)
        else:
            raise ValueError("no weight recipe for layer class %s (%s)" % (class_name, layer_name))
        out.append(np.asarray(a, dtype=np.float64))
    return out


def digest(a, n=6):
    """compact fingerprint of a tensor: L2 norm, sum, and n entries at fixed flat positions"""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.unique(np.linspace(0, a.size - 1, n).astype(np.int64))
    return {"norm": float(np.sqrt((a * a).sum())), "sum": float(a.sum()), "idx": [int(i) for i in idx],
            "val": [float(v) for v in a[idx]], "size": int(a.size)}
