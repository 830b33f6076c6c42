"""GPU tier: the bf16-ONLY kernels of round 5 (conv_halo_wide_kernel, conv_stem_s2_kernel, conv_stem_wgrad_kernel) on the ACTUAL
BASELINE layer shapes, held tighter than the 2e-2 kernel tolerance of tests/test_kernels.py (these kernels
have no float32 instantiation, so no model-level 1e-4 gate ever executes their tiling / halo / depth-tap / up-sampling address code).

Every case runs the SAME descriptor twice on identical bf16 inputs -- once on the halo-tile / stem kernel, once on the im2col
kernels of rounds 1-4 (HDU_TUNE_HALO_WIDE = 1) -- and requires
  * every output element within ONE bf16 ulp of the other kernel's (both accumulate the same bf16 products in float32; only the
    summation order differs, so the float32 sums differ by ~1e-6 relative and at most flip the final bf16 rounding), with a floor of
    2e-6 x max|y| for elements that are cancellations to (nearly) zero, and at most 3 % of the elements differing at all;
  * the conv-epilogue statistics (sum, sum of squares of the stored values) equal to 2e-4 of the sum of magnitudes (the stored
    values differ by the ulp flips above, the atomics by their order);
  * a float64 restatement on a sample of output positions (all corners / borders of the tile grid + random interior ones: patches
    gathered on the host from the bf16 inputs, contracted in float64) within 4e-3 relative to max|y|, for BOTH kernels.
Shapes: conv_up0...4 at 8 x 512^2 (fused up-sampling = the hybrids' 2D branch form; and their data gradients, which are forward
convs over dy with the transposed filter), 3dconv_up1...4 / fianl_conv at 224 x 224 x 12, and 3dconv_up4 / fianl_conv on a 512 x 512 x 8
slab in the plain, depth-"valid" and cropped (depth-sharded) forms.  Match: denseunet.py:189-218, denseunet3d.py:158-184,430.
"""
import ctypes
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = 0


def _ops():
    return importlib.import_module("h-denseunet_amd.ops")


# id, N, D, H, W (stored input), Cin, Cout, K, pad, up
CASES = [
    # ---- 2D decoder at 8 x 512^2, up-sampling fused
    ("conv_up0", 8, 1, 16, 16, 2208, 768, (1, 3, 3), (0, 1, 1), (0, 1, 1)),
    ("conv_up1", 8, 1, 32, 32, 768, 384, (1, 3, 3), (0, 1, 1), (0, 1, 1)),
    ("conv_up2", 8, 1, 64, 64, 384, 96, (1, 3, 3), (0, 1, 1), (0, 1, 1)),
    ("conv_up3", 8, 1, 128, 128, 96, 96, (1, 3, 3), (0, 1, 1), (0, 1, 1)),
    ("conv_up4", 8, 1, 256, 256, 96, 64, (1, 3, 3), (0, 1, 1), (0, 1, 1)),
    # ---- their data gradients: a forward conv over dy (no up-sampling) into the (padded) input channels
    ("conv_up1_dgrad", 8, 1, 64, 64, 384, 768, (1, 3, 3), (0, 1, 1), (0, 0, 0)),
    ("conv_up2_dgrad", 8, 1, 128, 128, 96, 384, (1, 3, 3), (0, 1, 1), (0, 0, 0)),
    ("conv_up3_dgrad", 8, 1, 256, 256, 96, 96, (1, 3, 3), (0, 1, 1), (0, 0, 0)),
    ("conv_up4_dgrad", 8, 1, 512, 512, 64, 96, (1, 3, 3), (0, 1, 1), (0, 0, 0)),
    # ---- 3D decoder + HFF conv at 224 x 224 x 12
    ("3dconv_up1", 1, 3, 14, 14, 504, 224, (3, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("3dconv_up2", 1, 3, 28, 28, 224, 192, (3, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("3dconv_up3", 1, 3, 56, 56, 192, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("3dconv_up4", 1, 6, 112, 112, 96, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("fianl_conv", 1, 12, 224, 224, 64, 64, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    ("3dconv_up4_dgrad", 1, 12, 224, 224, 64, 96, (3, 3, 3), (1, 1, 1), (0, 0, 0)),
    # ---- a 512 x 512 x 8 slab of the configs[4] shard shape: plain, depth "valid" (stored halo planes), cropped (up-sampled halo planes)
    ("slab_3dconv_up4", 1, 4, 256, 256, 96, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("slab_3dconv_up4_cropped", 1, 5, 256, 256, 96, 64, (3, 3, 3), (-1, 1, 1), (1, 1, 1)),
    ("slab_3dconv_up3_valid", 1, 6, 256, 256, 192, 96, (3, 3, 3), (0, 1, 1), (0, 0, 0)),
    ("slab_fianl_conv_valid", 1, 10, 512, 512, 64, 64, (3, 3, 3), (0, 1, 1), (0, 0, 0)),
]


def _rand_bf16(shape, seed, scale):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return ((torch.rand(shape, generator=g, device="cuda", dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


def _ulp_bf16(v):
    """one bf16 unit in the last place at |v| (float64 tensor)"""
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-30)))
    return torch.pow(2.0, e - 7)


def _one_ulp_compare(a, b, what):
    a, b = a.double(), b.double()
    scale = float(torch.maximum(a.abs().max(), b.abs().max()))
    diff = (a - b).abs()
    lim = torch.maximum(_ulp_bf16(torch.maximum(a.abs(), b.abs())), torch.full_like(a, 2e-6 * scale))
    bad = diff > lim * 1.0001
    assert not bool(bad.any()), "%s: %d of %d elements more than one bf16 ulp apart (max %.3e at scale %.3e)" % (
        what, int(bad.sum()), a.numel(), float(diff.max()), scale)
    frac = float((diff > 0).double().mean())
    assert frac <= 0.03, "%s: %.2f %% of the elements differ (summation order alone flips far fewer roundings)" % (what, 100 * frac)
    return frac


def _sample_positions(N, Do, Ho, Wo, n, seed):
    rng = np.random.default_rng(seed)
    pos = set()
    for nn in (0, N - 1):
        for d in (0, Do - 1):
            for h in (0, 1, 7, 8, 15, 16, Ho // 2, Ho - 2, Ho - 1):
                for w in (0, 1, 31, 32, 33, Wo // 2, Wo - 2, Wo - 1):
                    if 0 <= h < Ho and 0 <= w < Wo:
                        pos.add((nn, d, h, w))
    pos = sorted(pos)
    rng.shuffle(pos)
    pos = pos[:max(8, n // 2)]
    while len(pos) < n:
        pos.append((int(rng.integers(N)), int(rng.integers(Do)), int(rng.integers(Ho)), int(rng.integers(Wo))))
    return pos


def _ref_at(x, w, bias, pos, K, pad, up):
    """float64 conv outputs at the sampled positions: x [N,D,H,W,C] stored input (bf16, device), w [Cout,KD,KH,KW,Cin]"""
    N, D, H, W, C = x.shape
    De, He, We = D << up[0], H << up[1], W << up[2]
    patches = torch.zeros((len(pos),) + K + (C,), dtype=torch.float64)
    xc = x.cpu()
    for i, (n, od, oh, ow) in enumerate(pos):
        for kd in range(K[0]):
            for kh in range(K[1]):
                for kw in range(K[2]):
                    d, h, ww = od + kd - pad[0], oh + kh - pad[1], ow + kw - pad[2]
                    if 0 <= d < De and 0 <= h < He and 0 <= ww < We:
                        patches[i, kd, kh, kw] = xc[n, d >> up[0], h >> up[1], ww >> up[2]].double()
    ref = patches.reshape(len(pos), -1) @ w.cpu().double().reshape(w.shape[0], -1).t()
    return ref + bias.cpu().double()


@pytest.mark.parametrize("case", [pytest.param(c, id=c[0]) for c in CASES])
def test_halo_wide_equals_im2col_on_baseline_layer_shapes(hip_lib, case):
    name, N, D, H, W, Cin, Cout, K, pad, up = case
    ops = _ops()
    lib = hip_lib.lib.get()
    De, He, We = D << up[0], H << up[1], W << up[2]
    Do, Ho, Wo = De + 2 * pad[0] - K[0] + 1, He + 2 * pad[1] - K[1] + 1, We + 2 * pad[2] - K[2] + 1
    taps = K[0] * K[1] * K[2]
    x = _rand_bf16((N, D, H, W, Cin), 11, 1.0)
    w = _rand_bf16((Cout,) + K + (Cin,), 12, 1.0 / np.sqrt(taps * Cin))
    bias = ((torch.rand(Cout, device="cuda") - 0.5)).float()
    xa = ops.Act(x.reshape(-1), 0, N, D, H, W, Cin, Cin, BF16)
    wp = ctypes.c_void_p(w.data_ptr())
    slots = 16
    shift = torch.zeros(Cout, dtype=torch.float32, device="cuda")
    outs, stats = {}, {}
    try:
        for mode in (0, 1):         # 0: the shipped choice (must be the halo-tile kernel), 1: im2col kernels of rounds 1-4
            lib.hdu_set_tuning(29, mode)
            ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, BF16)
            part = torch.zeros(slots * 2 * Cout, dtype=torch.float32, device="cuda")
            d = ops.conv_desc(xa, wp, ya, K, (1, 1, 1), pad, up, None, None, True, bias)
            d.stats_partial, d.stats_shift, d.stats_slots = part.data_ptr(), shift.data_ptr(), slots
            kn = ops.conv_kernel_name(d, 0)
            if mode == 0 and not kn.startswith("conv_halo_wide_kernel"):
                pytest.skip("%s stays on %s at this size (fewer than 128 halo tiles): nothing to compare" % (name, kn.split("<")[0]))
            assert kn.startswith("conv_halo_wide_kernel") == (mode == 0), (name, mode, kn)
            ops.conv_fprop(d)
            torch.cuda.synchronize()
            outs[mode] = ya.buf.float().reshape(N, Do, Ho, Wo, Cout)
            stats[mode] = part.double().reshape(slots, 2, Cout).sum(0).cpu()
    finally:
        lib.hdu_set_tuning(29, 0)
    frac = _one_ulp_compare(outs[0], outs[1], name)
    y0 = outs[0].double()
    mag = y0.abs().reshape(-1, Cout).sum(0).cpu()
    assert float(((stats[0][0] - stats[1][0]).abs() / (mag + 1e-9)).max()) <= 2e-4, "epilogue sums"
    mag2 = (y0 * y0).reshape(-1, Cout).sum(0).cpu()
    assert float(((stats[0][1] - stats[1][1]).abs() / (mag2 + 1e-9)).max()) <= 2e-4, "epilogue sums of squares"
    # each kernel's statistics are the sums of ITS stored values
    assert float(((stats[0][0] - y0.reshape(-1, Cout).sum(0).cpu()).abs() / (mag + 1e-9)).max()) <= 1e-4
    # float64 restatement at sampled positions
    npos = int(max(32, min(1024, 1.5e9 // (Cout * taps * Cin))))
    pos = _sample_positions(N, Do, Ho, Wo, npos, 5)
    ref = _ref_at(x, w, bias, pos, K, pad, up)
    idx = torch.tensor(pos, device="cuda")
    scale = float(ref.abs().max())
    for mode in (0, 1):
        got = outs[mode][idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]].double().cpu()
        err = float((got - ref).abs().max())
        assert err <= 4e-3 * scale, "%s (%s): %.3e vs float64 at scale %.3e" % (name, "halo-wide" if mode == 0 else "im2col", err, scale)
    print("%s: %.3f %% of %d elements differ by one bf16 ulp between halo-wide and im2col" % (name, 100 * frac, y0.numel()))


STEM_CASES = [
    # id, N, D, H, W, K, pad -- conv1 at 8 x 512^2, 3dconv1 at 224 x 224 x 12 and on a 512 x 512 x 16 slab (plain and depth-"valid")
    ("conv1_8x512", 8, 1, 512, 512, (1, 7, 7), (0, 3, 3)),
    ("3dconv1_224x224x12", 1, 12, 224, 224, (7, 7, 7), (3, 3, 3)),
    ("3dconv1_512x512x16", 1, 16, 512, 512, (7, 7, 7), (3, 3, 3)),
    ("3dconv1_512x512x22_valid", 1, 22, 512, 512, (7, 7, 7), (0, 3, 3)),
]


@pytest.mark.parametrize("case", [pytest.param(c, id=c[0]) for c in STEM_CASES])
def test_stem_kernels_equal_im2col_on_baseline_shapes(hip_lib, case):
    """conv_stem_s2_kernel / conv_stem_wgrad_kernel against the im2col kernels on the same bf16 operands: forward within one bf16 ulp,
    filter gradient (float32 output, float atomics on both sides) within 2e-5 of the gradient's scale, both against float64 samples."""
    name, N, D, H, W, K, pad = case
    ops = _ops()
    lib = hip_lib.lib.get()
    Cin, Cout, s = 8, 96, (2 if K[0] > 1 else 1, 2, 2)
    Do = (D + 2 * pad[0] - K[0]) // s[0] + 1
    Ho, Wo = (H + 2 * pad[1] - K[1]) // 2 + 1, (W + 2 * pad[2] - K[2]) // 2 + 1
    taps = K[0] * K[1] * K[2]
    x = _rand_bf16((N, D, H, W, Cin), 21, 1.0)
    x[..., 4:] = 0          # the stem stores 3 / 4 logical channels in 8
    w = _rand_bf16((Cout,) + K + (Cin,), 22, 1.0 / np.sqrt(taps * 4))
    dy = _rand_bf16((N, Do, Ho, Wo, Cout), 23, 1.0)
    xa = ops.Act(x.reshape(-1), 0, N, D, H, W, Cin, Cin, BF16)
    dya = ops.Act(dy.reshape(-1), 0, N, Do, Ho, Wo, Cout, Cout, BF16)
    wp = ctypes.c_void_p(w.data_ptr())
    outs, grads = {}, {}
    try:
        for mode in (0, 1):
            lib.hdu_set_tuning(29, mode)
            ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, BF16)
            d = ops.conv_desc(xa, wp, ya, K, s, pad, (0, 0, 0), None, None, True, None)
            assert (ops.conv_kernel_name(d, 0) == "conv_stem_s2_kernel") == (mode == 0), ops.conv_kernel_name(d, 0)
            ops.conv_fprop(d)
            outs[mode] = ya.buf.float().reshape(N, Do, Ho, Wo, Cout)
            dw = torch.zeros((Cout,) + K + (Cin,), dtype=torch.float32, device="cuda")
            dg = ops.conv_desc(xa, wp, dya, K, s, pad, (0, 0, 0), None, None, True, None)
            assert (ops.conv_kernel_name(dg, 1) == "conv_stem_wgrad_kernel") == (mode == 0), ops.conv_kernel_name(dg, 1)
            ops.conv_wgrad(dg, dw)
            torch.cuda.synchronize()
            grads[mode] = dw.double().cpu()
    finally:
        lib.hdu_set_tuning(29, 0)
    _one_ulp_compare(outs[0], outs[1], name + " forward")
    gs = float(grads[1].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) <= 2e-5 * gs, "stem filter gradient vs im2col: %.3e at scale %.3e" % (
        float((grads[0] - grads[1]).abs().max()), gs)
    # float64: forward at sampled positions; filter gradient for sampled output channels / taps
    pos = _sample_positions(N, Do, Ho, Wo, 256, 7)
    xc, wc = x.cpu().double(), w.cpu().double()
    ref = torch.zeros(len(pos), Cout, dtype=torch.float64)
    for i, (n, od, oh, ow) in enumerate(pos):
        for kd in range(K[0]):
            dd = od * s[0] + kd - pad[0]
            if not 0 <= dd < D:
                continue
            h0, w0 = oh * 2 - pad[1], ow * 2 - pad[2]
            hs, ws = max(h0, 0), max(w0, 0)
            he, we = min(h0 + K[1], H), min(w0 + K[2], W)
            patch = xc[n, dd, hs:he, ws:we]                                   # [kh', kw', C]
            ref[i] += torch.einsum("hwc,ohwc->o", patch, wc[:, kd, hs - h0:he - h0, ws - w0:we - w0])
    idx = torch.tensor(pos, device="cuda")
    for mode in (0, 1):
        got = outs[mode][idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]].double().cpu()
        assert float((got - ref).abs().max()) <= 4e-3 * float(ref.abs().max()), (name, mode)
    # filter gradient of tap (kd, kh, kw) = sum over output pixels of dy[o] x[2 o + k - p]: a strided slice product in float64
    dyc = dy.cpu().double()
    rng = np.random.default_rng(3)
    for _ in range(6):
        kd, kh, kw = int(rng.integers(K[0])), int(rng.integers(K[1])), int(rng.integers(K[2]))
        acc = torch.zeros(Cout, Cin, dtype=torch.float64)
        for od in range(Do):
            dd = od * s[0] + kd - pad[0]
            if not 0 <= dd < D:
                continue
            oh = torch.arange(Ho)
            ow = torch.arange(Wo)
            ih, iw = oh * 2 + kh - pad[1], ow * 2 + kw - pad[2]
            mh, mw = (ih >= 0) & (ih < H), (iw >= 0) & (iw < W)
            xs = xc[:, dd][:, ih[mh]][:, :, iw[mw]]                            # [N, h', w', C]
            ds = dyc[:, od][:, oh[mh]][:, :, ow[mw]]                           # [N, h', w', Cout]
            acc += torch.einsum("nhwo,nhwc->oc", ds, xs)
        for mode in (0, 1):
            assert float((grads[mode][:, kd, kh, kw] - acc).abs().max()) <= 1e-4 * max(float(acc.abs().max()), 1e-9) + 1e-3, (name, mode, kd, kh, kw)
