#!/usr/bin/env python
"""oracle/ref_keras/make_funcs_fixture.py -- TEST INFRASTRUCTURE: golden vectors of the z-sliding-window inference loop produced
BY THE REFERENCE'S OWN lib/funcs.py:predict_tumor_inwindow (imported unmodified from /root/reference over the eager backend).

    PYTHONDONTWRITEBYTECODE=1 python oracle/ref_keras/make_funcs_fixture.py

The function is Python 2 (`window_cols = (img_cols/4)` feeds xrange): `args.input_cols` is handed over as an int whose `/`
floors like Python 2's; nothing in the reference is edited.  The model is a deterministic stand-in with the one method the
loop calls (`predict(box, batch_size, verbose)` -> (1, rows, cols, window, 3) logits, a fixed function of the box and of the
voxel position), so the fixture pins the WINDOWING: window starts incl. the clamped last window, the [1:-1] slice trim, the
overlap counts, the final division -- tests/test_sliding_window.py holds its numpy restatement (the thing the product's
HBM-resident sweep is tested against) to these arrays.  Writes tests/golden/ref_funcs_sliding_window.npz.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.ref_keras import harness as H      # noqa: E402


class Py2Int(int):
    """an int whose true division floors, as `/` did for ints in the Python 2 the reference was written for"""

    def __truediv__(self, other):
        return Py2Int(int(self) // int(other))


class StandInModel:
    """logits = fixed per-class affine maps of the CT value plus a position-dependent term (so that a mis-placed window or a
    wrong trim changes the result), float32 like a Keras predict"""

    def predict(self, box, batch_size=None, verbose=0):
        b, d, r, c, _ = box.shape
        x = box[..., 0].astype(np.float64)
        zz = np.arange(c, dtype=np.float64)[None, None, None, :]
        yy = np.arange(r, dtype=np.float64)[None, None, :, None]
        out = np.stack([0.01 * x + 0.05 * zz, -0.02 * x + 0.03 * yy, 0.015 * x - 0.04 * zz + 0.01 * yy], -1)
        return out.astype(np.float32)


CASES = [   # (volume shape, window, mini, maxi): the scripts' call (test.py:48-51) at reduced size; the last window is clamped
    ((16, 16, 29), 8, (0, 0, 6), (15, 15, 22)),
    ((16, 16, 12), 8, (0, 0, 4), (15, 15, 9)),
    ((16, 16, 40), 12, (3, 2, 0), (14, 13, 39)),
]


def main():
    H.setup("float64")
    import funcs as ref_funcs                        # /root/reference/lib/funcs.py
    out = {}
    for i, (shape, win, mini, maxi) in enumerate(CASES):
        rng = np.random.default_rng(100 + i)
        vol = rng.normal(0.0, 40.0, shape).astype(np.float32)
        args = types.SimpleNamespace(b=1, input_size=shape[0], input_cols=Py2Int(win))
        s1, s2 = ref_funcs.predict_tumor_inwindow(StandInModel(), vol, 3, mini, maxi, args)
        out["s1_%d" % i], out["s2_%d" % i] = np.asarray(s1), np.asarray(s2)      # (the volume is re-drawn from its seed by the test)
        out["meta%d" % i] = np.array(list(shape) + [win] + list(mini) + list(maxi) + [100 + i], dtype=np.int64)
        print("case %d: volume %s window %d -> score1 mean %.6f score2 mean %.6f" % (i, shape, win, s1.mean(), s2.mean()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_funcs_sliding_window.npz"), **out)


if __name__ == "__main__":
    if not H.available():
        sys.exit("the reference tree is not present")
    main()
