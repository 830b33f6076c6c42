"""Pins the oracle (test infrastructure) before it is trusted:
 1. the known-answer tests the reference's own Keras suite holds for ops on the hot path;
 2. oracle/torch_ref.py (used for whole graphs) against oracle/np64.py (independent numpy loops) per op;
 3. finite-difference checks of the analytic loss gradient and of one BN+conv composite.
The TensorFlow boundary itself stays unpinned (no TF, no golden vectors in the reference) -- see the oracle headers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np64 as N  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

rng = np.random.default_rng(0)
T = lambda a: torch.tensor(a, dtype=torch.float64)


# ---- 1. known answers lifted from Keras-2.0.8/tests
def test_upsampling_is_np_repeat():
    """tests/keras/layers/convolutional_test.py:673-681 (2D), :726-736 (3D); backend_test.py:295-305"""
    x = rng.normal(size=(2, 4, 5, 3))
    exp = np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)
    np.testing.assert_array_equal(N.upsample(x, (2, 2)), exp)
    np.testing.assert_array_equal(R.upsample_nearest(T(x), (2, 2)).numpy(), exp)
    x3 = rng.normal(size=(1, 3, 4, 2, 2))
    exp3 = np.repeat(np.repeat(np.repeat(x3, 2, axis=1), 2, axis=2), 1, axis=3)
    np.testing.assert_array_equal(R.upsample_nearest(T(x3), (2, 2, 1)).numpy(), exp3)


def test_zero_padding_border_and_interior():
    """tests/keras/layers/convolutional_test.py:525-565: zeros on the border, input in the interior"""
    x = np.ones((1, 4, 5, 2))
    for out in (N.zero_pad(x, 2), R.zero_pad(T(x), 2).numpy()):
        assert out.shape == (1, 8, 9, 2)
        assert np.all(out[:, :2] == 0) and np.all(out[:, -2:] == 0) and np.all(out[:, :, :2] == 0) and np.all(out[:, :, -2:] == 0)
        assert np.all(out[:, 2:-2, 2:-2] == 1)


def test_relu_softmax_vs_numpy():
    """tests/keras/activations_test.py:54-68 (softmax), :158-164 (relu)"""
    x = rng.normal(size=(2, 5))
    e = np.exp(x - x.max(1, keepdims=True))
    np.testing.assert_allclose(N.softmax(x), e / e.sum(1, keepdims=True), rtol=1e-12)
    np.testing.assert_allclose(torch.softmax(T(x), 1).numpy(), e / e.sum(1, keepdims=True), rtol=1e-12)
    np.testing.assert_array_equal(N.relu(x), x * (x > 0))


def test_batchnorm_statistics():
    """tests/keras/layers/normalization_test.py:35-49: normalised output has mean ~0, std ~1 (atol 1e-1)"""
    x = rng.normal(5.0, 10.0, size=(64, 4, 4, 3))
    y, _, _ = N.batch_norm_train(x, np.ones(3), np.zeros(3), 1e-3)
    np.testing.assert_allclose(y.mean((0, 1, 2)), 0, atol=1e-1)
    np.testing.assert_allclose(y.std((0, 1, 2)), 1, atol=1e-1)


# ---- 2. torch restatement == numpy restatement, op by op
@pytest.mark.parametrize("nd,k,s,padding", [(2, 3, 1, "same"), (2, 7, 2, "valid"), (2, 1, 1, "same"), (3, 3, 1, "same"),
                                            (3, 7, 2, "valid"), (3, 1, 1, "valid")])
def test_conv_torch_vs_numpy(nd, k, s, padding):
    sp = (9, 8) if nd == 2 else (9, 8, 7)
    x = rng.normal(size=(2,) + sp + (3,))
    kern = rng.normal(size=(k,) * nd + (3, 4))
    b = rng.normal(size=4)
    pad = (k // 2,) * nd if padding == "same" else (0,) * nd
    ref = N.conv_nd(x, kern, (s,) * nd, pad, b)
    got = R.conv_nd(T(x), T(kern), (s,) * nd, padding, T(b)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_pools_bn_torch_vs_numpy():
    x = np.maximum(rng.normal(size=(2, 8, 10, 5)), 0)
    np.testing.assert_allclose(R.max_pool_valid(R.zero_pad(T(x), 1), 3, 2).numpy(), N.max_pool(N.zero_pad(x, 1), 3, 2))
    np.testing.assert_allclose(R.avg_pool_valid(T(x), (2, 2)).numpy(), N.avg_pool(x, (2, 2)), rtol=1e-12)
    x3 = np.maximum(rng.normal(size=(1, 6, 8, 4, 3)), 0)
    np.testing.assert_allclose(R.max_pool_valid(R.zero_pad(T(x3), 1), 3, 2).numpy(), N.max_pool(N.zero_pad(x3, 1), 3, 2))
    np.testing.assert_allclose(R.avg_pool_valid(T(x3), (2, 2, 1)).numpy(), N.avg_pool(x3, (2, 2, 1)), rtol=1e-12)
    g, b = rng.uniform(0.5, 1.5, 5), rng.normal(size=5)
    y, mean, var = N.batch_norm_train(x, g, b, 1.1e-5)
    yt, (mt, vt) = R.batch_norm(T(x), T(g), T(b), None, None, 1.1e-5, True)
    np.testing.assert_allclose(yt.numpy(), y, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(vt.numpy(), var, rtol=1e-12)
    mm, mv = rng.normal(size=5), rng.uniform(0.5, 2, 5)
    np.testing.assert_allclose(R.batch_norm(T(x), T(g), T(b), T(mm), T(mv), 1e-3, False)[0].numpy(),
                               N.batch_norm_infer(x, g, b, mm, mv, 1e-3), rtol=1e-10)
    np.testing.assert_allclose(N.moving_update(mm, mean, 0.99), mm * 0.99 + mean * 0.01, rtol=1e-12)


def test_loss_and_sgd_torch_vs_numpy():
    z = rng.normal(scale=4, size=(300, 3))
    z[:5, 0] = 80.0
    lab = rng.integers(0, 3, 300)
    lab[:5] = 1
    loss, grad = N.weighted_crossentropy(z, lab)
    zt = T(z).requires_grad_(True)
    lt = R.weighted_crossentropy_rows(zt, torch.tensor(lab))
    lt.backward()
    assert abs(float(lt) - loss) < 1e-12
    np.testing.assert_allclose(zt.grad.numpy(), grad, atol=1e-14)
    assert np.all(grad[:5] == 0)      # clipped rows: no gradient (tf.clip_by_value)
    p, v, g = rng.normal(size=7), rng.normal(size=7), rng.normal(size=7)
    pn, vn = N.sgd_nesterov(p, v, g, 1e-3, 0.9)
    pt, vt = R.sgd_nesterov_(T(p), T(v), T(g), 1e-3, 0.9)
    np.testing.assert_allclose(pt.numpy(), pn, rtol=1e-14)
    np.testing.assert_allclose(vt.numpy(), vn, rtol=1e-14)


def test_slab25d_matches_reference_slicing():
    """denseunet3d.py:399-410: first slab (0,0,1), last (D-2,D-1,D-1)"""
    vol = rng.normal(size=(4, 5, 6))
    s = N.slab25d(vol)
    np.testing.assert_array_equal(s[0, :, :, 0], vol[:, :, 0])
    np.testing.assert_array_equal(s[0, :, :, 2], vol[:, :, 1])
    np.testing.assert_array_equal(s[5, :, :, 1], vol[:, :, 5])
    np.testing.assert_array_equal(s[5, :, :, 2], vol[:, :, 5])
    np.testing.assert_array_equal(s[3], vol[:, :, 2:5])


# ---- 3. finite differences
def test_loss_gradient_finite_difference():
    z = rng.normal(size=(20, 3))
    lab = rng.integers(0, 3, 20)
    _, grad = N.weighted_crossentropy(z, lab)
    for (i, j) in [(0, 0), (3, 2), (11, 1)]:
        d = np.zeros_like(z)
        d[i, j] = 1e-6
        num = (N.weighted_crossentropy(z + d, lab)[0] - N.weighted_crossentropy(z - d, lab)[0]) / 2e-6
        assert abs(num - grad[i, j]) < 1e-8


def test_graph_inventory_and_shapes():
    """the oracle graph reproduces the reference's layer inventory: DenseNet-161 blocks 6/12/36/24, growth 48,
    decoder 768/384/96/96/64, 3D blocks 3/4/12/8 growth 32 (A.2)"""
    P = R.ParamStore(dtype=torch.float32, perturb=False)
    with torch.no_grad():
        feat, logits = R.dense_unet_2d(P, torch.zeros(1, 64, 64, 3), variant="denseunet")
    assert tuple(logits.shape) == (1, 64, 64, 3) and tuple(feat.shape) == (1, 64, 64, 64)
    assert P.w["conv1"][0].shape == (7, 7, 3, 96)
    assert P.w["conv5_24_x1"][0].shape == (1, 1, 2160, 192) and P.w["conv5_24_x2"][0].shape == (3, 3, 192, 48)
    assert P.w["conv4_blk"][0].shape == (1, 1, 2112, 1056) and P.w["line0"][0].shape == (1, 1, 2112, 2208)
    assert [P.w["conv_up%d" % i][0].shape[-1] for i in range(5)] == [768, 384, 96, 96, 64]
    n2d = sum(int(np.prod(t.shape)) for k, ws in P.w.items() if P.kind[k] == "conv" for t in ws)
    assert abs(n2d - 49.3e6) < 0.1e6          # 49.3 M (conv kernels + biases)
    P3 = R.ParamStore(dtype=torch.float32, perturb=False)
    with torch.no_grad():
        f3 = R.dense_net_3d(P3, torch.zeros(1, 32, 32, 8, 4), variant="3dpart")
    assert tuple(f3.shape) == (1, 32, 32, 8, 64)
    assert P3.w["3dconv1"][0].shape == (7, 7, 7, 4, 96) and P3.w["3dconv5_8_x1"][0].shape == (1, 1, 1, 472, 128)
    assert [P3.w["3dconv_up%d" % i][0].shape[-1] for i in range(5)] == [504, 224, 192, 96, 64]
