"""N1: host-side post-processing of test.py:57-112 -- thresholds, largest connected component,
hole filling -- against a literal, independent restatement (breadth-first labelling in numpy/python)."""
import itertools
from collections import deque

import numpy as np
import pytest

import parity_utils as U


def bfs_label(binary):
    """26-connected components in raster order (the order skimage.measure.label numbers them)"""
    lab = np.zeros(binary.shape, dtype=np.int64)
    nxt = 0
    offs = [o for o in itertools.product((-1, 0, 1), repeat=binary.ndim) if any(o)]
    for idx in zip(*np.nonzero(binary)):
        if lab[idx]:
            continue
        nxt += 1
        lab[idx] = nxt
        q = deque([idx])
        while q:
            p = q.popleft()
            for o in offs:
                n = tuple(a + b for a, b in zip(p, o))
                if all(0 <= c < s for c, s in zip(n, binary.shape)) and binary[n] and not lab[n]:
                    lab[n] = nxt
                    q.append(n)
    return lab, nxt


def largest(binary):
    lab, num = bfs_label(binary)
    box = [int((lab == i + 1).sum()) for i in range(num)]
    keep = box.index(max(box)) + 1
    return (lab == keep).astype(int)


def fill_holes(binary):
    """background voxels not 6-connected to the border become foreground (ndimage.binary_fill_holes default)"""
    b = np.pad(np.asarray(binary).astype(bool), 1)
    out = np.zeros(b.shape, dtype=bool)
    q = deque([(0,) * b.ndim])
    out[(0,) * b.ndim] = True
    offs = [tuple(int(i == a) * s for i in range(b.ndim)) for a in range(b.ndim) for s in (-1, 1)]
    while q:
        p = q.popleft()
        for o in offs:
            n = tuple(x + y for x, y in zip(p, o))
            if all(0 <= c < s for c, s in zip(n, b.shape)) and not b[n] and not out[n]:
                out[n] = True
                q.append(n)
    core = tuple(slice(1, -1) for _ in range(b.ndim))
    return (~out)[core].astype(int)


def dilate(binary):
    """one iteration with the 6-connected cross (ndimage.binary_dilation default structure)"""
    b = np.pad(np.asarray(binary).astype(bool), 1)
    out = b.copy()
    for a in range(b.ndim):
        for s in (-1, 1):
            out |= np.roll(b, s, axis=a)
    return out[tuple(slice(1, -1) for _ in range(b.ndim))]


def reference_postprocess(score1, score2, mask, tl, tt):
    r1, r2 = score1.copy(), score2.copy()
    r1[r1 >= tl] = 1; r1[r1 < tl] = 0
    r2[r2 >= tt] = 1; r2[r2 < tt] = 0
    r1[r2 == 1] = 1
    seg = r2
    liver_res = largest(r1)
    m = dilate(mask).astype(int)
    liver_labels = fill_holes(largest(m))
    seg = fill_holes(seg * liver_labels).astype(np.uint8)
    liver_res = fill_holes(liver_res.astype(np.uint8))
    liver_res[seg == 1] = 2
    return liver_res.astype(np.uint8)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_postprocess_matches_literal_restatement(seed):
    f = U.pkg("funcs")
    rng = np.random.default_rng(seed)
    shape = (18, 20, 12)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    liver = ((zz - 8) ** 2 / 36.0 + (yy - 9) ** 2 / 49.0 + (xx - 6) ** 2 / 16.0) < 1.0
    blob2 = ((zz - 15) ** 2 + (yy - 17) ** 2 + (xx - 2) ** 2) < 5          # a second, smaller "liver" component
    score1 = np.clip(0.9 * (liver | blob2) + rng.normal(0, 0.15, shape), 0, 1)
    score1[7:9, 8:10, 5:7] = 0.0                                            # a hole inside the liver
    tumor = ((zz - 8) ** 2 + (yy - 11) ** 2 + (xx - 6) ** 2) < 4
    stray = ((zz - 15) ** 2 + (yy - 3) ** 2 + (xx - 9) ** 2) < 3            # tumour-like response outside the liver
    score2 = np.clip(0.95 * (tumor | stray) + rng.normal(0, 0.05, shape), 0, 1)
    coarse = (liver | blob2).astype(np.int16)
    coarse[liver & (rng.random(shape) < 0.05)] = 2                          # coarse mask carries label 2 voxels too
    mask, mini, maxi = f.liver_window_from_mask(coarse)
    ref_mask = coarse.copy(); ref_mask[ref_mask == 2] = 1
    np.testing.assert_array_equal(mask.astype(bool), dilate(ref_mask))
    idx = np.where(dilate(ref_mask))
    np.testing.assert_array_equal(mini, np.min(idx, axis=-1)); np.testing.assert_array_equal(maxi, np.max(idx, axis=-1))
    got = f.segment_liver_tumor(score1, score2, mask, 0.5, 0.8)
    ref = reference_postprocess(score1, score2, mask.astype(int), 0.5, 0.8)
    assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 1, 2} and (got == 2).any() and (got == 1).any()
    np.testing.assert_array_equal(got, ref)
    assert not got[stray & ~dilate(dilate(ref_mask))].any()                # the stray response is outside the liver box


def test_postprocess_empty_inputs_raise():
    f = U.pkg("funcs")
    with pytest.raises(ValueError):
        f.liver_window_from_mask(np.zeros((4, 4, 4), np.int16))
    with pytest.raises(ValueError):
        f.segment_liver_tumor(np.zeros((4, 4, 4)), np.zeros((4, 4, 4)), np.ones((4, 4, 4), np.int16))
