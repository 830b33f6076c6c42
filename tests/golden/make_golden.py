"""Generates tests/golden/ops_f64.npz -- TEST INFRASTRUCTURE.

The reference holds no golden vectors for this path and its arithmetic (TensorFlow 1.x) cannot run here (SURVEY.md
section 8c), so these fixtures are NOT reference outputs: they are the outputs of the independent float64 numpy
restatement (oracle/np64.py: plain loops over the Keras / TF-backend definitions) on small seeded inputs, committed so
that (a) the torch oracle used for whole graphs and (b) the HIP kernels are both pinned to fixed numbers, and a silent
change in either the oracle or a kernel shows up as a fixture mismatch.  "parity unpinned" at the TF boundary stands.

    python tests/golden/make_golden.py        # rewrites ops_f64.npz (deterministic: numpy default_rng seeds below)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np64 as O  # noqa: E402


def bf16_round(a):
    """values exactly representable in bfloat16 (so the bf16 kernels see the same inputs as the float64 restatement)"""
    u = np.asarray(a, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32).astype(np.float64)


def main():
    rng = np.random.default_rng(20240924)
    out = {}

    def conv_case(tag, x_shape, k_shape, strides, pad, bias):
        x = bf16_round(rng.normal(0, 1, x_shape))
        k = bf16_round(rng.normal(0, 1.0 / np.sqrt(np.prod(k_shape[:-1])), k_shape))
        b = rng.normal(0, 0.3, k_shape[-1]).astype(np.float32).astype(np.float64) if bias else None
        y = O.conv_nd(x, k, strides, pad, b)
        dy = bf16_round(rng.normal(0, 1, y.shape))
        # gradients of sum(y * dy): dk by correlating padded x with dy, dx by the transposed correlation (loops)
        nd = x.ndim - 2
        xp = np.pad(x, [(0, 0)] + [(p, p) for p in pad] + [(0, 0)])
        dk = np.zeros_like(k)
        dxp = np.zeros_like(xp)
        for idx in np.ndindex(*k.shape[:nd]):
            sl = tuple(slice(idx[a], idx[a] + strides[a] * (y.shape[1 + a] - 1) + 1, strides[a]) for a in range(nd))
            patch = xp[(slice(None),) + sl + (slice(None),)]
            dk[idx] = patch.reshape(-1, patch.shape[-1]).T @ dy.reshape(-1, dy.shape[-1])
            dxp[(slice(None),) + sl + (slice(None),)] += np.einsum("...o,co->...c", dy, k[idx])
        core = tuple(slice(p, dxp.shape[1 + a] - p) for a, p in enumerate(pad))
        dx = dxp[(slice(None),) + core + (slice(None),)]
        for n, v in (("x", x), ("k", k), ("y", y), ("dy", dy), ("dk", dk), ("dx", dx)):
            out["%s/%s" % (tag, n)] = v
        if b is not None:
            out[tag + "/bias"] = b
        out[tag + "/strides"] = np.array(strides)
        out[tag + "/pad"] = np.array(pad)

    conv_case("conv2d_3x3", (2, 9, 11, 16), (3, 3, 16, 24), (1, 1), (1, 1), True)
    conv_case("conv2d_1x1", (1, 8, 8, 40), (1, 1, 40, 48), (1, 1), (0, 0), False)
    conv_case("conv2d_7x7s2", (1, 18, 14, 8), (7, 7, 8, 16), (2, 2), (3, 3), False)
    conv_case("conv3d_3x3x3", (1, 5, 6, 3, 8), (3, 3, 3, 8, 16), (1, 1, 1), (1, 1, 1), True)

    # BatchNormalization (+Scale) training forward / moving update
    x = bf16_round(rng.normal(0.7, 2.0, (2, 6, 5, 16)))
    g, b_, sg, sb = (rng.normal(1, 0.2, 16), rng.normal(0, 0.2, 16), rng.normal(1, 0.2, 16), rng.normal(0, 0.1, 16))
    y, mean, var = O.batch_norm_train(x, g, b_, 1.1e-5)
    z = O.relu(O.scale(y, sg, sb))
    out.update({"bn/x": x, "bn/gamma": g, "bn/beta": b_, "bn/sgamma": sg, "bn/sbeta": sb, "bn/mean": mean, "bn/var": var,
                "bn/z": z, "bn/mov_mean": O.moving_update(np.full(16, 0.5), mean, 0.99),
                "bn/mov_var": O.moving_update(np.full(16, 2.0), var, 0.99)})
    # pools / up-sampling
    xp = bf16_round(rng.normal(0, 1, (1, 9, 8, 8)))
    out["pool/x"] = xp
    out["pool/max3s2_pad1"] = O.max_pool(O.zero_pad(xp, 1), 3, 2)
    out["pool/avg2"] = O.avg_pool(xp[:, :8], (2, 2))
    out["pool/up2"] = O.upsample(xp, (2, 2))
    # weighted cross-entropy + gradient
    lg = bf16_round(rng.normal(0, 2, (1, 6, 7, 3)))
    lab = rng.integers(0, 3, (1, 6, 7, 1))
    loss, grad = O.weighted_crossentropy(lg, lab)
    out.update({"wce/logits": lg, "wce/labels": lab.astype(np.int64), "wce/loss": np.array(loss), "wce/grad": grad})
    # Nesterov SGD
    p, v, gr = rng.normal(0, 1, 40), rng.normal(0, 0.1, 40), rng.normal(0, 1, 40)
    pn, vn = O.sgd_nesterov(p, v, gr, 1e-3, 0.9)
    out.update({"sgd/p": p, "sgd/v": v, "sgd/g": gr, "sgd/p_new": pn, "sgd/v_new": vn})
    # 2.5D slab
    vol = rng.normal(0, 50, (4, 5, 6))
    out.update({"slab/vol": vol, "slab/out": O.slab25d(vol)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops_f64.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %.1f KB" % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
