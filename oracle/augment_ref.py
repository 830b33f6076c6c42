"""oracle/augment_ref.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's training-sample pipeline.

  load_seq_crop_data_masktumor_try ... train_2ddense.py:40-106 (2D: 3 adjacent slices, label of the middle one)
                                        train_hybrid.py:40-100 (hybrid: input_cols slices, all labels)
  the batch assembly ................. train_2ddense.py:108-133 / train_hybrid.py:102-133

Per sample: random scale in [0.8, 1.2) -> square crop of int(size*scale) around a liver / tumour voxel, clamped into the
liver bounding box -> mean subtraction -> one of 8 flips / rotations -> resize to (size, size): labels nearest (order 0,
mode 'edge'), image bicubic (order 3, mode 'constant', cval 0, clip to the crop's value range, preserve_range).

The resize is scikit-image's `skimage.transform.resize` (requirements.txt pins scikit-image==0.13.1).  Neither that
release nor its source is available offline; what IS available is scikit-image 0.18.3 in /opt/conda (not importable by
the default interpreter): `resize_like_skimage` below restates ITS 2-D fast path (`warp` -> Cython `_warp_fast`:
Catmull-Rom cubic convolution over a 4x4 neighbourhood anchored at floor(coordinate), out-of-image taps read `cval`;
nearest = C `round`), called with anti_aliasing=False (0.13 had no anti-aliasing), and is pinned bit-for-tolerance against
it by tests/golden/skimage_resize.npz (generator: tests/golden/make_resize_golden.py, run with the conda interpreter).
**Pinned against 0.18.3, not 0.13.1**: the 2017 release's bicubic kernel differs.

Only tests/ may import this module.
"""
import numpy as np


def _cubic(x, f0, f1, f2, f3):
    """skimage/_shared/interpolation.pxd cubic_interpolation (Catmull-Rom, a = -0.5)"""
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)))


def resize_like_skimage(image, out_rows, out_cols, order, mode, cval=0.0, clip=True):
    """skimage.transform.resize(image, (out_rows, out_cols, C), order, mode, cval, clip, preserve_range=True,
    anti_aliasing=False) for a (rows, cols, C) array with C unchanged: per-channel 2-D warp."""
    image = np.asarray(image, np.float64)
    rows, cols, C = image.shape
    rs, cs = rows / float(out_rows), cols / float(out_cols)
    r = rs * (np.arange(out_rows) + 0.5) - 0.5
    c = cs * (np.arange(out_cols) + 0.5) - 0.5
    if mode == "constant":
        padv = cval

        def fetch(ri, ci):
            ok = (ri >= 0) & (ri < rows) & (ci >= 0) & (ci < cols)
            v = image[np.clip(ri, 0, rows - 1), np.clip(ci, 0, cols - 1)]
            return np.where(ok[..., None], v, padv)
    elif mode == "edge":
        def fetch(ri, ci):
            return image[np.clip(ri, 0, rows - 1), np.clip(ci, 0, cols - 1)]
    else:
        raise ValueError(mode)
    if order == 0:
        # C round(): half away from zero
        rr = np.where(r >= 0, np.floor(r + 0.5), np.ceil(r - 0.5)).astype(np.int64)
        cr = np.where(c >= 0, np.floor(c + 0.5), np.ceil(c - 0.5)).astype(np.int64)
        return fetch(rr[:, None], cr[None, :])
    if order != 3:
        raise ValueError("order 0 and 3 only (what the reference asks for)")
    r0 = np.floor(r).astype(np.int64)
    c0 = np.floor(c).astype(np.int64)
    xr = (r - r0)[:, None, None]
    xc = (c - c0)[None, :, None]
    rows_interp = []
    for pr in range(4):
        f = [fetch((r0 - 1 + pr)[:, None], (c0 - 1 + pc)[None, :]) for pc in range(4)]
        rows_interp.append(_cubic(xc, *f))
    out = _cubic(xr, *rows_interp)
    if clip:
        lo, hi = image.min(), image.max()
        preserve = mode == "constant" and not (lo <= cval <= hi)
        mask = out == cval if preserve else None
        out = np.clip(out, lo, hi)
        if preserve:
            out[mask] = cval
    return out


FLIPS = 8


def flip_rot(a, flip_num):
    """train_2ddense.py:73-101 (applied to the (rows, cols, slices) crop)"""
    if flip_num == 1:
        return np.flipud(a)
    if flip_num == 2:
        return np.fliplr(a)
    if flip_num == 3:
        return np.rot90(a, k=1, axes=(1, 0))
    if flip_num == 4:
        return np.rot90(a, k=3, axes=(1, 0))
    if flip_num == 5:
        return np.rot90(np.fliplr(a), k=1, axes=(1, 0))
    if flip_num == 6:
        return np.rot90(np.fliplr(a), k=3, axes=(1, 0))
    if flip_num == 7:
        return np.fliplr(np.flipud(a))
    return a


def draw_sample_params(rng, size, cols, centres, minindex, maxindex):
    """the random draws of one sample in the reference's order (train_2ddense.py:49-62): scale, centre line, flip"""
    scale = rng.uniform(0.8, 1.2)
    deps = rows = int(size * scale)
    sed = rng.randint(1, len(centres) + 1) if len(centres) > 1 else 1     # np.random.randint(1, numid), numid = len + 1
    cen = centres[sed - 1]
    a = min(max(minindex[0] + deps // 2, cen[0]), maxindex[0] - deps // 2 - 1)
    b = min(max(minindex[1] + rows // 2, cen[1]), maxindex[1] - rows // 2 - 1)
    c = min(max(minindex[2] + cols // 2, cen[2]), maxindex[2] - cols // 2 - 1)
    flip = int(rng.randint(0, FLIPS))
    return dict(deps=deps, a=int(a), b=int(b), c=int(c), flip=flip)


def make_sample(img, tumor, prm, size, cols, mean, hybrid):
    """the deterministic part of load_seq_crop_data_masktumor_try for drawn parameters `prm`.  2D: returns
    (size, size, 3) image and (size, size) label of the middle slice; hybrid: (size, size, cols) both."""
    d, a, b, c = prm["deps"], prm["a"], prm["b"], prm["c"]
    if hybrid:
        zs = slice(c - cols // 2, c + cols // 2)                  # train_hybrid.py:63-66
    else:
        zs = slice(c - 3 // 2, c + 3 // 2 + 1)                    # train_2ddense.py:64-67 (cols = 3)
    ci = img[a - d // 2:a + d // 2, b - d // 2:b + d // 2, zs].astype(np.float64) - mean
    ct = tumor[a - d // 2:a + d // 2, b - d // 2:b + d // 2, zs].astype(np.float64)
    ci, ct = flip_rot(ci, prm["flip"]), flip_rot(ct, prm["flip"])
    lab = resize_like_skimage(ct, size, size, 0, "edge")
    x = resize_like_skimage(ci, size, size, 3, "constant", 0.0, True)
    return (x, lab) if hybrid else (x, lab[:, :, 1])
