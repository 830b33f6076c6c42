"""Deterministic synthetic CT phantom used by bench.py, the smoke test and the parity tests:
there is no network access for LiTS, so inputs are generated with the value statistics the reference's
pre-processing produces (HU clip [-200,250], preprocessing.py:15-16; mean 48 subtracted, train_2ddense.py:32,65)."""
import numpy as np


def synthetic_ct(shape_hwd, seed=1234):
    """float32 volume (H,W,D) and int labels {0: background, 1: liver, 2: tumour} of the same geometry."""
    H, W, D = shape_hwd
    rng = np.random.default_rng(seed)
    yy, xx, zz = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), np.linspace(-1, 1, D), indexing="ij")
    vol = np.full((H, W, D), -200.0)
    liver = ((yy + 0.1) / 0.62) ** 2 + ((xx - 0.05) / 0.5) ** 2 + (zz / 1.4) ** 2 < 1.0
    vol[liver] = rng.normal(100.0, 20.0, int(liver.sum()))
    lab = np.zeros((H, W, D), np.int64)
    lab[liver] = 1
    for _ in range(int(rng.integers(1, 4))):
        cy, cx, cz = rng.uniform(-0.3, 0.2), rng.uniform(-0.2, 0.3), rng.uniform(-0.6, 0.6)
        r = rng.uniform(0.12, 0.22)
        tum = ((yy - cy) ** 2 + (xx - cx) ** 2 + ((zz - cz) * 0.6) ** 2 < r * r) & liver
        vol[tum] = rng.normal(60.0, 15.0, int(tum.sum()))
        lab[tum] = 2
    vol += rng.normal(0.0, 10.0, vol.shape)
    vol = np.clip(vol, -200.0, 250.0) - 48.0
    return vol.astype(np.float32), lab


def synthetic_batch(kind, b, size, cols, seed=1234):
    """'2d': b slices of one phantom, 3 adjacent slices as channels (train_2ddense.py:62-66) -> x (b,H,W,3),
    y (b,H,W,1); 'hybrid': one volume -> x (1,H,W,D,1), y (1,H,W,D,1)."""
    if kind == "2d":
        vol, lab = synthetic_ct((size, size, b + 2), seed)
        x = np.stack([vol[:, :, k:k + 3] for k in range(b)], 0).astype(np.float32)
        y = np.stack([lab[:, :, k + 1] for k in range(b)], 0)[..., None]
        return x, y
    vol, lab = synthetic_ct((size, size, cols), seed)
    return vol[None, ..., None].astype(np.float32), lab[None, ..., None]
