"""N1: z-sliding-window inference (lib/funcs.py:4-51) -- the HBM-resident implementation equals a literal numpy
restatement of the reference loop driven through Model.predict."""
import numpy as np
import pytest
import torch

import parity_utils as U


def reference_loop(model, imgs_test, num, mini, maxi, args):
    """lib/funcs.py:4-51 restated with numpy (softmax via numpy instead of K.softmax/K.eval); `model` only needs
    .predict(box, batch_size, verbose)"""
    batch, img_deps, img_rows, img_cols = args.b, args.input_size, args.input_size, args.input_cols
    window_cols = img_cols // 4
    box_test = np.zeros((batch, img_deps, img_rows, img_cols, 1), dtype="float32")
    x, y, z = imgs_test.shape
    right_cols = int(min(z, maxi[2] + 10) - img_cols)
    left_cols = max(0, min(mini[2] - 5, right_cols))
    score = np.zeros((x, y, z, num), dtype="float32")
    score_num = np.zeros((x, y, z, num), dtype="int16")
    for cols in range(left_cols, right_cols + window_cols, window_cols):
        c0 = z - img_cols if cols > z - img_cols else cols
        box_test[0, :, :, :, 0] = imgs_test[0:img_deps, 0:img_rows, c0:c0 + img_cols]
        m = model.predict(box_test, batch_size=batch, verbose=0)
        e = np.exp(m - m.max(-1, keepdims=True))
        m = (e / e.sum(-1, keepdims=True))[:, :, :, 1:-1, :]
        score[0:img_deps, 0:img_rows, c0 + 1:c0 + img_cols - 1, :] += m[0]
        score_num[0:img_deps, 0:img_rows, c0 + 1:c0 + img_cols - 1, :] += 1
    score = score / (score_num + 1e-4)
    return score[:, :, :, num - 2], score[:, :, :, num - 1]


def test_sliding_window_matches_reference_loop(emu_lib):
    args = U.make_args(1, 32, 8)
    model = U.pkg("hybridnet").dense_rnn_net(args, dtype="f32", nb_layers2d=(2, 2, 2, 2), nb_layers3d=(1, 1, 2, 1))
    vol, _ = U.pkg("synth").synthetic_ct((32, 32, 12), seed=3)
    mini, maxi = (0, 0, 4), (31, 31, 9)
    s1, s2 = U.pkg("funcs").predict_tumor_inwindow(model, vol, 3, mini, maxi, args)
    r1, r2 = reference_loop(model, vol, 3, mini, maxi, args)
    assert s1.shape == (32, 32, 12)
    np.testing.assert_allclose(s1, r1, atol=2e-6)
    np.testing.assert_allclose(s2, r2, atol=2e-6)
    assert float(np.abs(s1).max()) > 0


class _OraclePredictor:
    """the float32 torch restatement of hybridnet.py behind the one method lib/funcs.py calls"""

    def __init__(self, P, fwd):
        self.P, self.fwd = P, fwd

    def predict(self, box, batch_size=None, verbose=0):
        return U.R.predict(self.P, self.fwd, torch.tensor(box)).numpy()


@pytest.mark.gpu
def test_sliding_window_full_size_vs_torch_oracle(hip_lib):
    """N1 on hardware: the HBM-resident sweep of `dense_rnn_net` (224 x 224 x 12 windows, float32 parity mode) over a
    224 x 224 x 40 phantom against the literal lib/funcs.py loop driven by the TORCH ORACLE's predict (not the
    product's): both averaged score volumes within 1e-3 absolute (probabilities) and thresholded masks that differ
    on at most 1e-4 of the voxels (the host-side post-processing of test.py has its own test, test_postprocess.py)."""
    args = U.make_args(1, 224, 12)
    m, P, fwd = U.build_pair("hybrid", "end2end", 1, 224, 12, "f32", (6, 12, 36, 24), (3, 4, 12, 8),
                             odtype=torch.float32, perturb=False)
    vol, lab = U.pkg("synth").synthetic_ct((224, 224, 40), seed=3)
    funcs = U.pkg("funcs")
    mask, mini, maxi = funcs.liver_window_from_mask((lab > 0).astype(np.uint8)[:, :, :])
    maxi = np.array([maxi[0], maxi[1], min(int(maxi[2]), 22)])        # 6 windows: bounds the oracle's CPU time
    s1, s2 = funcs.predict_tumor_inwindow(m, vol, 3, mini, maxi, args)
    r1, r2 = reference_loop(_OraclePredictor(P, fwd), vol, 3, mini, maxi, args)
    assert s1.shape == r1.shape == (224, 224, 40)
    e1, e2 = float(np.abs(s1 - r1).max()), float(np.abs(s2 - r2).max())
    print("sliding window vs oracle: max abs score err %.2e / %.2e, swept planes %d" % (e1, e2, int((r1.sum((0, 1)) != 0).sum())))
    assert e1 <= 1e-3 and e2 <= 1e-3
    for thr in (0.3, 0.5):
        for a, b in ((s1, r1), (s2, r2)):
            assert float(((a >= thr) != (b >= thr)).mean()) <= 1e-4
    assert float(np.abs(r1).max()) > 0


def test_reference_loop_restatement_equals_the_references_own_function():
    """`reference_loop` above -- what the product's HBM-resident sweep is held to -- against arrays produced by the REFERENCE'S OWN
    lib/funcs.py:predict_tumor_inwindow, run unmodified over the eager backend with a deterministic stand-in model
    (oracle/ref_keras/make_funcs_fixture.py): window starts incl. the clamped last window, the [1:-1] trim, overlap counts"""
    import os
    import sys
    import types
    sys.path.insert(0, U.ROOT)
    from oracle.ref_keras.make_funcs_fixture import StandInModel
    z = np.load(os.path.join(U.ROOT, "tests", "golden", "ref_funcs_sliding_window.npz"))
    n = len([k for k in z.files if k.startswith("meta")])
    assert n == 3
    for i in range(n):
        meta = [int(v) for v in z["meta%d" % i]]
        shape, win, mini, maxi, seed = tuple(meta[0:3]), meta[3], tuple(meta[4:7]), tuple(meta[7:10]), meta[10]
        vol = np.random.default_rng(seed).normal(0.0, 40.0, shape).astype(np.float32)
        args = types.SimpleNamespace(b=1, input_size=shape[0], input_cols=win)
        r1, r2 = reference_loop(StandInModel(), vol, 3, mini, maxi, args)
        np.testing.assert_allclose(r1, z["s1_%d" % i], atol=2e-6)
        np.testing.assert_allclose(r2, z["s2_%d" % i], atol=2e-6)
        assert float(np.abs(z["s1_%d" % i]).max()) > 0.1
