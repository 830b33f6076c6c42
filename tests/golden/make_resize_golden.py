"""Generates tests/golden/skimage_resize.npz -- TEST INFRASTRUCTURE.

Outputs of the REAL `skimage.transform.resize` (scikit-image 0.18.3, the only release available offline; the reference
pins 0.13.1) for the two calls the reference's generators make (train_2ddense.py:103-104): labels order=0 mode='edge',
image order=3 mode='constant' cval=0 clip=True preserve_range=True; anti_aliasing=False (0.13 had none).
oracle/augment_ref.py's restatement and the HIP augment kernel are pinned against these arrays (tests/test_augment.py).

    /opt/conda/bin/python3.9 tests/golden/make_resize_golden.py
"""
import os

import numpy as np
import skimage
from skimage.transform import resize

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(21)
out = {"skimage_version": np.array(skimage.__version__)}
for tag, (rows, cols, ch, size) in {"down45": (45, 45, 3, 32), "up26": (26, 26, 3, 32), "same32": (32, 32, 3, 32),
                                    "hyb38": (38, 38, 8, 32), "rect": (40, 29, 3, 32)}.items():
    img = (rng.normal(0, 60, (rows, cols, ch)) + 40 * np.sin(np.arange(rows) / 5.0)[:, None, None]).astype(np.float32)
    lab = rng.integers(0, 3, (rows, cols, ch)).astype(np.float32)
    out[tag + "_img"] = img
    out[tag + "_lab"] = lab
    out[tag + "_img_out"] = resize(img.astype(np.float64), (size, size, ch), order=3, mode="constant", cval=0, clip=True,
                                   preserve_range=True, anti_aliasing=False)
    out[tag + "_lab_out"] = resize(lab.astype(np.float64), (size, size, ch), order=0, mode="edge", cval=0, clip=True,
                                   preserve_range=True, anti_aliasing=False)
# a crop whose value range excludes cval = 0: the 'preserve cval' branch of the clip
img = (rng.uniform(5, 50, (30, 30, 3))).astype(np.float32)
out["poscval_img"] = img
out["poscval_img_out"] = resize(img.astype(np.float64), (32, 32, 3), order=3, mode="constant", cval=0, clip=True,
                                preserve_range=True, anti_aliasing=False)
np.savez_compressed(os.path.join(HERE, "skimage_resize.npz"), **out)
print("wrote", os.path.join(HERE, "skimage_resize.npz"), skimage.__version__)
