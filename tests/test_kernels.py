"""Op-level parity of every C-ABI kernel against a float64 torch/numpy restatement of the Keras/TF op
semantics.  Each test runs twice: under the x86 emulator build of the kernel
sources (CPU tier) and, marked `gpu`, on the gfx950 library."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

BF16, F32 = 0, 1
DT = [pytest.param(F32, id="f32"), pytest.param(BF16, id="bf16")]


def ops_mod():
    return importlib.import_module("h-denseunet_amd.ops")


def rnd(shape, seed, scale=1.0, dtype=F32):
    g = torch.Generator().manual_seed(seed)
    t = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale
    if dtype == BF16:
        t = t.to(torch.bfloat16).double()
    else:
        t = t.float().double()
    return t


def q(t, dtype):
    return t.to(torch.bfloat16).double() if dtype == BF16 else t.float().double()


def tol(dtype):
    return (2e-2, 2e-2) if dtype == BF16 else (2e-5, 2e-5)


def assert_close(got, ref, dtype, scale=None, what=""):
    rt, at = tol(dtype)
    s = float(ref.abs().max()) if scale is None else scale
    err = (got.double() - ref).abs()
    lim = at * max(s, 1e-6) + rt * ref.abs()
    bad = err > lim
    assert not bad.any(), "%s: %d/%d mismatches, max err %.3e (scale %.3e)" % (what, int(bad.sum()), bad.numel(), float(err.max()), s)


def mkact(ops, t, dtype, ld=None, coff=0):
    """t: [N,D,H,W,C] float64 -> Act (optionally as a slab of a wider buffer)"""
    N, D, H, W, C = t.shape
    if ld is None:
        a = ops.Act.alloc(N, D, H, W, C, dtype)
    else:
        big = ops.Act.alloc(N, D, H, W, ld, dtype, zero=True)
        big.buf.fill_(7.0)  # poison the other channels
        a = big.slab(coff, C)
    a.from_torch(t)
    return a


def dev(ops, t):
    return t.float().contiguous().to(ops.device())


def ref_xeff(x, up, skip, pro, relu, dtype):
    xe = x
    if pro is not None:
        xe = xe * pro[0] + pro[1]
        if relu:
            xe = xe.clamp_min(0)
    for ax, u in zip((1, 2, 3), up):
        if u:
            xe = xe.repeat_interleave(2, dim=ax)
    if skip is not None:
        xe = xe + skip
    return q(xe, dtype)  # the kernel feeds the MFMA in the storage dtype


def ref_conv(xe, w, stride, pad, bias):
    if pad[0] < 0:          # negative depth padding = the outer planes are cropped (depth-sharded decoder layers: engine.ConvLayer halo mode)
        xe = xe[:, -pad[0]:xe.shape[1] + pad[0]]
        pad = (0, pad[1], pad[2])
    y = F.conv3d(xe.permute(0, 4, 1, 2, 3), w.permute(0, 4, 1, 2, 3), stride=stride, padding=pad)
    y = y.permute(0, 2, 3, 4, 1)
    if bias is not None:
        y = y + bias
    return y


CONV_CASES = [
    # N, D, H, W, Cin, Cout, K, stride, pad, up, skip, pro, bias, ld_in, ld_out
    dict(N=2, D=1, H=9, W=11, Cin=16, Cout=48, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=True, bias=False, ldin=40, ldout=64, id="dense3x3_2d_slab"),
    dict(N=1, D=1, H=8, W=8, Cin=72, Cout=192, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=True, bias=False, ldin=96, ldout=None, id="bottleneck1x1"),
    dict(N=1, D=1, H=6, W=5, Cin=24, Cout=40, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 1, 1), skip=True, pro=True, bias=True, ldin=None, ldout=None, id="decoder_up_skip_2d"),
    dict(N=1, D=3, H=5, W=6, Cin=16, Cout=32, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 0, 0), skip=False, pro=True, bias=False, ldin=None, ldout=48, id="dense3x3x3"),
    dict(N=1, D=2, H=4, W=4, Cin=8, Cout=24, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(1, 1, 1), skip=False, pro=True, bias=True, ldin=None, ldout=None, id="decoder_up222_3d"),
    dict(N=1, D=2, H=5, W=4, Cin=8, Cout=16, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 1, 1), skip=True, pro=False, bias=True, ldin=None, ldout=None, id="decoder_up221_skip_3d"),
    dict(N=2, D=1, H=18, W=14, Cin=8, Cout=96, K=(1, 7, 7), s=(1, 2, 2), p=(0, 3, 3), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="stem7x7s2"),
    dict(N=1, D=8, H=10, W=10, Cin=8, Cout=96, K=(7, 7, 7), s=(2, 2, 2), p=(3, 3, 3), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="stem7x7x7s2"),
    dict(N=1, D=1, H=40, W=36, Cin=64, Cout=8, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=True, bias=True, ldin=None, ldout=None, id="classifier_pad8"),
    dict(N=1, D=1, H=70, W=66, Cin=32, Cout=264, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=True, ldin=None, ldout=None, id="wide_bn128"),
    dict(N=1, D=1, H=130, W=128, Cin=8, Cout=128, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=False, bias=True, ldin=None, ldout=None, id="tile128x128_ragged_m"),
    dict(N=2, D=1, H=10, W=37, Cin=64, Cout=48, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=96, ldout=64, id="halo_tile_ragged_slab"),
    dict(N=1, D=1, H=8, W=64, Cin=32, Cout=64, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=True, ldin=None, ldout=None, id="halo_tile_exact"),
    # round 4: 3 x 3 x 3 layers on the halo-tile filter gradient (three plane-shifted 2D problems): ragged tile grid, two volumes
    # (the plane before volume 1's first plane is volume 0's last: it must read as padding), slab input / output
    dict(N=1, D=3, H=6, W=33, Cin=32, Cout=48, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="halo_tile_3d_ragged"),
    dict(N=2, D=2, H=5, W=32, Cin=64, Cout=40, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=True, ldin=80, ldout=56, id="halo_tile_3d_two_volumes_slab"),
    # depth "valid" (pad 0 in depth, two more input than output planes): the depth-sharded layers, whose halo planes are stored
    dict(N=2, D=4, H=5, W=33, Cin=32, Cout=48, K=(3, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="halo_tile_3d_valid_depth"),
    dict(N=1, D=3, H=3, W=16, Cin=32, Cout=24, K=(3, 3, 3), s=(1, 1, 1), p=(-1, 1, 1), up=(1, 1, 1), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="halo_tile_3d_sharded_decoder"),
    # ... and with the decoder's nearest-neighbour up-sampling in front of the conv resolved in the tile addressing
    dict(N=2, D=1, H=5, W=17, Cin=32, Cout=64, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 1, 1), skip=False, pro=False, bias=False, ldin=48, ldout=None, id="halo_tile_up2d"),
    dict(N=1, D=2, H=3, W=16, Cin=32, Cout=24, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(1, 1, 1), skip=False, pro=False, bias=True, ldin=None, ldout=None, id="halo_tile_up222_3d"),
    dict(N=1, D=1, H=7, W=35, Cin=40, Cout=48, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=56, ldout=None, id="halo_tile_ragged_channel_chunk"),
    dict(N=2, D=3, H=4, W=20, Cin=64, Cout=32, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 1, 1), skip=False, pro=False, bias=False, ldin=None, ldout=48, id="halo_tile_up221_3d"),
    # the filter-stationary pointwise kernel (bf16: K = 192 / 128, no prologue / bias, >= 256 output channels): ragged pixel
    # count, a channel count that is not a multiple of its 128-channel groups, slab input and output
    dict(N=2, D=1, H=13, W=11, Cin=192, Cout=328, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=208, ldout=344, id="pw_bstat_k192_ragged"),
    dict(N=1, D=3, H=9, W=10, Cin=128, Cout=256, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="pw_bstat_k128_3d"),
    # round 4: pointwise convs over relu(a * x + b) on the async-DMA kernels (the affine applied to the operand fragments in
    # registers): a grid that takes the two-stage form (> 256 tiles of 128 rows, ragged M, slab input), a contraction wider than
    # the small (1024-channel) LDS table with split-K, a 3D bottleneck (Cout 128 -> 128-wide tile) and a ragged channel count
    dict(N=1, D=1, H=131, W=128, Cin=72, Cout=192, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=True, bias=False, ldin=96, ldout=None, id="pw_pro_two_stage"),
    dict(N=1, D=1, H=7, W=9, Cin=1096, Cout=48, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=True, bias=False, ldin=1112, ldout=64, id="pw_pro_wide_table_splitk"),
    dict(N=1, D=3, H=9, W=10, Cin=104, Cout=128, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=True, bias=True, ldin=None, ldout=None, id="pw_pro_3d_bn128"),
]


def build_conv_case(ops, cs, dtype, seed=0):
    K, s, p, up = cs["K"], cs["s"], cs["p"], cs["up"]
    N, D, H, W, Cin, Cout = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cin"], cs["Cout"]
    x = rnd((N, D, H, W, Cin), seed + 1, 1.0, dtype)
    De, He, We = D << up[0], H << up[1], W << up[2]
    skip = rnd((N, De, He, We, Cin), seed + 2, 0.5, dtype) if cs["skip"] else None
    pro = (rnd((Cin,), seed + 3, 1.0).abs() + 0.5, rnd((Cin,), seed + 4, 0.3)) if cs["pro"] else None
    if pro is not None:
        pro = (pro[0].float().double(), pro[1].float().double())
    w = rnd((Cout,) + K + (Cin,), seed + 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cin), dtype)
    bias = rnd((Cout,), seed + 6, 0.5).float().double() if cs["bias"] else None
    Do = (De + 2 * p[0] - K[0]) // s[0] + 1
    Ho = (He + 2 * p[1] - K[1]) // s[1] + 1
    Wo = (We + 2 * p[2] - K[2]) // s[2] + 1
    xa = mkact(ops, x, dtype, cs["ldin"], 8 if cs["ldin"] else 0)
    sa = mkact(ops, skip, dtype) if skip is not None else None
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    return dict(x=x, skip=skip, pro=pro, w=w, bias=bias, xa=xa, sa=sa, wt=wt, out_dims=(N, Do, Ho, Wo, Cout))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES])
def test_conv_fprop(hdu, cs, dtype):
    ops = ops_mod()
    b = build_conv_case(ops, cs, dtype)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], dtype, zero=True)
        big.buf.fill_(3.0)
        ya = big.slab(8, Cout)
    else:
        big = None
        ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    import ctypes
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro, True, bias)
    ops.conv_fprop(d)
    xe = ref_xeff(b["x"], cs["up"], b["skip"], b["pro"], True, dtype)
    ref = ref_conv(xe, b["w"], cs["s"], cs["p"], b["bias"])
    assert_close(ya.to_torch().cpu(), ref, dtype, what="fprop")
    if big is not None:  # neighbouring slab channels untouched
        full = big.to_torch().cpu()
        assert float((full[..., :8] - 3.0).abs().max()) == 0.0
        assert float((full[..., 8 + Cout:] - 3.0).abs().max()) == 0.0
    # accumulate mode
    d.accumulate = 1
    ops.conv_fprop(d)
    assert_close(ya.to_torch().cpu(), q(ref, dtype) * 2, dtype, scale=2 * float(ref.abs().max()), what="fprop accumulate")
    # output affine (+ReLU) of the BN that follows (hdu_conv_desc.epi_*): relu(a * (conv + bias) + b)
    ea, eb = rnd((Cout,), 31, 1.0).float().double() + 1.5, rnd((Cout,), 32, 0.5).float().double()
    ea_d, eb_d = dev(ops, ea), dev(ops, eb)          # (the descriptor holds raw pointers: keep the tensors alive)
    for relu in (True, False):
        d2 = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro, True,
                           bias, epi=(ea_d, eb_d, relu))
        ops.conv_fprop(d2)
        r2 = ref * ea + eb
        assert_close(ya.to_torch().cpu(), r2.clamp_min(0) if relu else r2, dtype, scale=float(r2.abs().max()), what="fprop + output affine")
    d2.accumulate = 1
    with pytest.raises(hdu.lib.HduError, match="output affine"):
        ops.conv_fprop(d2)


PERS_CASES = [c for c in CONV_CASES if c["up"] == (0, 0, 0) and not c["skip"] and c["K"][0] * c["K"][1] * c["K"][2] <= 32
              and (not c["pro"] or c["K"] == (1, 1, 1))]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("wgs", [3, 256])
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in PERS_CASES])
def test_conv_fprop_persistent_kernel(hdu, cs, dtype, wgs):
    """conv_igemm_pers_kernel (round 4: one 512-thread workgroup per CU, 256-row tiles, a 3-slot operand ring that runs across
    tile boundaries, register epilogue) forced onto every eligible CONV_CASE -- with 3 workgroups every workgroup walks
    several (m-tile, n-tile) items, with 256 most get at most one -- against the float64 reference: plain, accumulate,
    bias, output affine, dropout mask consistency, epilogue statistics."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    b = build_conv_case(ops, cs, dtype)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], dtype, zero=True)
        big.buf.fill_(3.0)
        ya = big.slab(8, Cout)
    else:
        big, ya = None, ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    wp = ctypes.c_void_p(b["wt"].data_ptr())
    d = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, pro, True, bias)
    xe = ref_xeff(b["x"], cs["up"], None, b["pro"], True, dtype)
    ref = ref_conv(xe, b["w"], cs["s"], cs["p"], b["bias"])
    M = N * Do * Ho * Wo
    try:
        lib.hdu_set_tuning(24, 1)          # HDU_TUNE_PERS_MIN_ITEMS: every layer qualifies
        lib.hdu_set_tuning(18, 1)          # HDU_TUNE_BM64_MAX_M: no 64-row tiles (the persistent form replaces the 128-row ones)
        lib.hdu_set_tuning(23, wgs)        # HDU_TUNE_PERS: workgroups
        if ops.conv_kernel_name(d, 0).startswith("conv_pw_bstat") or ops.conv_kernel_name(d, 0).startswith("conv_halo"):
            pytest.skip("taken by a specialised kernel")
        assert ops.conv_kernel_name(d, 0).startswith("conv_igemm_pers_kernel"), ops.conv_kernel_name(d, 0)
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), ref, dtype, what="fprop")
        if big is not None:
            full = big.to_torch().cpu()
            assert float((full[..., :8] - 3.0).abs().max()) == 0.0 and float((full[..., 8 + Cout:] - 3.0).abs().max()) == 0.0
        d.accumulate = 1
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), q(ref, dtype) * 2, dtype, scale=2 * float(ref.abs().max()), what="fprop accumulate")
        ea, eb = rnd((Cout,), 31, 1.0).float().double() + 1.5, rnd((Cout,), 32, 0.5).float().double()
        ea_d, eb_d = dev(ops, ea), dev(ops, eb)
        d2 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, pro, True, bias, epi=(ea_d, eb_d, True))
        ops.conv_fprop(d2)
        r2 = (ref * ea + eb).clamp_min(0)
        assert_close(ya.to_torch().cpu(), r2, dtype, scale=float(r2.abs().max()), what="fprop + output affine")
        # epilogue statistics: sums of (y - shift), (y - shift)^2 of the STORED values over all pixels, spread over the slot rows
        slots = 8
        shift = rnd((Cout,), 33, 0.3).float()
        shift_d = shift.to(ops.device())
        part = torch.zeros(slots * 2 * Cout, dtype=torch.float32, device=ops.device())
        d3 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, pro, True, bias)
        d3.stats_partial, d3.stats_shift, d3.stats_slots = part.data_ptr(), shift_d.data_ptr(), slots
        ops.conv_fprop(d3)
        y = ya.to_torch().cpu().double().reshape(M, Cout)
        got = part.cpu().double().reshape(slots, 2, Cout).sum(0)
        dd = y - shift.double()
        scale1 = float(dd.abs().sum(0).max()) + 1e-9
        assert float((got[0] - dd.sum(0)).abs().max()) <= 1e-4 * scale1
        assert float((got[1] - (dd * dd).sum(0)).abs().max()) <= 1e-4 * float((dd * dd).sum(0).max())
        # dropout: the same mask as the tiled kernels draw (stateless hash of the element index)
        seed_dev = torch.zeros(1, dtype=torch.int32, device=ops.device())
        d4 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, pro, True, bias, False, 0.7, 1234, seed_dev)
        ops.conv_fprop(d4)
        got_drop = ya.to_torch().cpu().double()
        lib.hdu_set_tuning(23, 0)
        ops.conv_fprop(d4)
        assert_close(got_drop, ya.to_torch().cpu().double(), dtype, scale=float(ref.abs().max()) / 0.7, what="dropout mask vs the tiled kernel")
        kept = (got_drop != 0).double().mean()
        assert 0.55 < float(kept) < 0.85
    finally:
        lib.hdu_set_tuning(24, 0)
        lib.hdu_set_tuning(18, 0)
        lib.hdu_set_tuning(23, 0)          # (off by default: measured slower than the two-stage kernel)


# round 5: the halo-tile forward / data-gradient kernel for the wide 3x3 / 3x3x3 layers (csrc/conv_halo_wide.hip): every tile
# configuration forced (HDU_TUNE_HALO_WIDE = 2..6) onto geometries with ragged tile grids, ragged 16-channel stages (Cin % 16 = 8),
# ragged 32-channel output groups, slab input / output, two volumes (the plane in front of volume 1 must read as padding), depth
# "valid" / cropped (the depth-sharded forms) and the decoder's fused nearest-neighbour up-sampling per axis
HALO_WIDE_CASES = [
    dict(N=2, D=1, H=9, W=37, Cin=40, Cout=72, K=(1, 3, 3), p=(0, 1, 1), up=(0, 0, 0), bias=True, ldin=56, ldout=88, id="2d_ragged_slab"),
    dict(N=1, D=1, H=16, W=32, Cin=32, Cout=128, K=(1, 3, 3), p=(0, 1, 1), up=(0, 0, 0), bias=False, ldin=None, ldout=None, id="2d_exact_tile"),
    dict(N=2, D=1, H=5, W=17, Cin=24, Cout=64, K=(1, 3, 3), p=(0, 1, 1), up=(0, 1, 1), bias=True, ldin=None, ldout=None, id="2d_up"),
    dict(N=2, D=3, H=5, W=33, Cin=24, Cout=40, K=(3, 3, 3), p=(1, 1, 1), up=(0, 0, 0), bias=False, ldin=32, ldout=None, id="3d_two_volumes"),
    dict(N=1, D=4, H=4, W=34, Cin=16, Cout=96, K=(3, 3, 3), p=(0, 1, 1), up=(0, 0, 0), bias=True, ldin=None, ldout=None, id="3d_valid_depth"),
    dict(N=1, D=2, H=3, W=16, Cin=32, Cout=24, K=(3, 3, 3), p=(1, 1, 1), up=(1, 1, 1), bias=True, ldin=None, ldout=40, id="3d_up222"),
    dict(N=1, D=3, H=3, W=16, Cin=8, Cout=48, K=(3, 3, 3), p=(-1, 1, 1), up=(1, 1, 1), bias=False, ldin=None, ldout=None, id="3d_up222_cropped_depth"),
    dict(N=1, D=3, H=4, W=18, Cin=40, Cout=32, K=(3, 3, 3), p=(1, 1, 1), up=(0, 1, 1), bias=False, ldin=None, ldout=None, id="3d_up221"),
]
for _c in HALO_WIDE_CASES:
    _c.update(s=(1, 1, 1), skip=False, pro=False)
HALO_WIDE_CFGS = [(2, "8x128"), (3, "16x64"), (4, "16x96"), (5, "8x64"), (6, "8x96"), (7, "16x128"), (8, "16x64p")]


@pytest.mark.parametrize("cfg", [pytest.param(c[0], id=c[1]) for c in HALO_WIDE_CFGS])
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in HALO_WIDE_CASES])
def test_conv_halo_wide(hdu, cs, cfg):
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    dtype = BF16
    b = build_conv_case(ops, cs, dtype)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], dtype, zero=True)
        big.buf.fill_(3.0)
        ya = big.slab(8, Cout)
    else:
        big, ya = None, ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    wp = ctypes.c_void_p(b["wt"].data_ptr())
    d = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias)
    xe = ref_xeff(b["x"], cs["up"], None, None, True, dtype)
    ref = ref_conv(xe, b["w"], cs["s"], cs["p"], b["bias"])
    M = N * Do * Ho * Wo
    try:
        lib.hdu_set_tuning(29, cfg)
        assert ops.conv_kernel_name(d, 0).startswith("conv_halo_wide_kernel"), ops.conv_kernel_name(d, 0)
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), ref, dtype, what="fprop")
        if big is not None:
            full = big.to_torch().cpu()
            assert float((full[..., :8] - 3.0).abs().max()) == 0.0 and float((full[..., 8 + Cout:] - 3.0).abs().max()) == 0.0
        d.accumulate = 1
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), q(ref, dtype) * 2, dtype, scale=2 * float(ref.abs().max()), what="fprop accumulate")
        ea, eb = rnd((Cout,), 31, 1.0).float().double() + 1.5, rnd((Cout,), 32, 0.5).float().double()
        ea_d, eb_d = dev(ops, ea), dev(ops, eb)
        d2 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias, epi=(ea_d, eb_d, True))
        ops.conv_fprop(d2)
        r2 = (ref * ea + eb).clamp_min(0)
        assert_close(ya.to_torch().cpu(), r2, dtype, scale=float(r2.abs().max()), what="fprop + output affine")
        # epilogue statistics of the STORED values
        slots = 8
        shift = rnd((Cout,), 33, 0.3).float()
        shift_d = shift.to(ops.device())
        part = torch.zeros(slots * 2 * Cout, dtype=torch.float32, device=ops.device())
        d3 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias)
        d3.stats_partial, d3.stats_shift, d3.stats_slots = part.data_ptr(), shift_d.data_ptr(), slots
        ops.conv_fprop(d3)
        assert_close(ya.to_torch().cpu(), ref, dtype, what="fprop with statistics")
        y = ya.to_torch().cpu().double().reshape(M, Cout)
        got = part.cpu().double().reshape(slots, 2, Cout).sum(0)
        dd = y - shift.double()
        assert float((got[0] - dd.sum(0)).abs().max()) <= 1e-4 * (float(dd.abs().sum(0).max()) + 1e-9)
        assert float((got[1] - (dd * dd).sum(0)).abs().max()) <= 1e-4 * float((dd * dd).sum(0).max())
        # dropout: the mask the im2col kernels draw (stateless hash of the element index)
        seed_dev = torch.zeros(1, dtype=torch.int32, device=ops.device())
        d4 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias, False, 0.7, 1234, seed_dev)
        ops.conv_fprop(d4)
        got_drop = ya.to_torch().cpu().double()
        lib.hdu_set_tuning(29, 1)
        assert not ops.conv_kernel_name(d4, 0).startswith("conv_halo_wide_kernel")
        ops.conv_fprop(d4)
        assert_close(got_drop, ya.to_torch().cpu().double(), dtype, scale=float(ref.abs().max()) / 0.7, what="dropout mask vs the im2col kernel")
        assert 0.55 < float((got_drop != 0).double().mean()) < 0.85
    finally:
        lib.hdu_set_tuning(29, 0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES
                                if c["id"] in ("dense3x3_2d_slab", "bottleneck1x1", "dense3x3x3", "stem7x7s2")])
def test_conv_fprop_large_tensor_path(hdu, cs, dtype):
    """input tensors of 4 GiB and more cannot be addressed by the 32-bit byte offsets of the buffer-resource DMA: the
    implicit GEMM then gathers through 64-bit pointers and a zero page.  HDU_TUNE_DEBUG bit 4 forces that path."""
    import ctypes
    ops = ops_mod()
    b = build_conv_case(ops, cs, dtype)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, None)
    lib = hdu.lib.get()
    try:
        lib.hdu_set_tuning(4, 16)
        ops.conv_fprop(d)
    finally:
        lib.hdu_set_tuning(4, 0)
    xe = ref_xeff(b["x"], cs["up"], None, None, True, dtype)
    assert_close(ya.to_torch().cpu(), ref_conv(xe, b["w"], cs["s"], cs["p"], None), dtype, what="fprop, 64-bit pointer path")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES])
def test_conv_wgrad(hdu, cs, dtype):
    ops = ops_mod()
    b = build_conv_case(ops, cs, dtype, seed=100)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    dy = rnd((N, Do, Ho, Wo, Cout), 777, 1.0, dtype)
    dya = mkact(ops, dy, dtype, cs["ldout"], 8 if cs["ldout"] else 0)
    pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
    import ctypes
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro, True, None)
    dw = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
    ops.conv_wgrad(d, dw)
    xe = ref_xeff(b["x"], cs["up"], b["skip"], b["pro"], True, dtype).requires_grad_(True)
    wref = b["w"].clone().requires_grad_(True)
    y = ref_conv(xe, wref, cs["s"], cs["p"], None)
    (y * dy).sum().backward()
    assert_close(dw.cpu(), wref.grad, F32 if dtype == F32 else BF16, what="wgrad")
    # second call accumulates
    ops.conv_wgrad(d, dw)
    assert_close(dw.cpu(), 2 * wref.grad, F32 if dtype == F32 else BF16, what="wgrad accumulate")


STEM_CASES = [
    dict(N=2, D=1, H=40, W=70, Cin=8, Cout=96, K=(1, 7, 7), s=(1, 2, 2), p=(0, 3, 3), bias=False, ldout=None, id="2d_ragged"),
    dict(N=1, D=1, H=64, W=64, Cin=8, Cout=96, K=(1, 7, 7), s=(1, 2, 2), p=(0, 3, 3), bias=True, ldout=112, id="2d_exact_slab"),
    dict(N=1, D=8, H=34, W=50, Cin=8, Cout=96, K=(7, 7, 7), s=(2, 2, 2), p=(3, 3, 3), bias=False, ldout=None, id="3d_ragged"),
    dict(N=2, D=4, H=12, W=20, Cin=8, Cout=40, K=(7, 7, 7), s=(2, 2, 2), p=(3, 3, 3), bias=True, ldout=None, id="3d_two_volumes_cout40"),
    dict(N=1, D=10, H=16, W=36, Cin=8, Cout=104, K=(7, 7, 7), s=(2, 2, 2), p=(0, 3, 3), bias=False, ldout=None, id="3d_valid_depth_two_groups"),
]
for _c in STEM_CASES:
    _c.update(up=(0, 0, 0), skip=False, pro=False, ldin=None)


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in STEM_CASES])
def test_conv_stem_kernel(hdu, cs):
    """round 5: the 7 x 7 (x 7) stride-2 stem over 8 stored channels on its own kernel (conv_stem_s2_kernel: contraction ordered
    (kd, kh | kw, c), row segments staged once per (kd, kh)) against the float64 reference: plain, bias, slab output, two volumes,
    depth "valid" (the depth-sharded stem), two output-channel groups, epilogue statistics."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    dtype = BF16
    b = build_conv_case(ops, cs, dtype)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], dtype, zero=True)
        big.buf.fill_(3.0)
        ya = big.slab(8, Cout)
    else:
        big, ya = None, ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    wp = ctypes.c_void_p(b["wt"].data_ptr())
    d = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias)
    ref = ref_conv(ref_xeff(b["x"], cs["up"], None, None, True, dtype), b["w"], cs["s"], cs["p"], b["bias"])
    M = N * Do * Ho * Wo
    try:
        lib.hdu_set_tuning(29, 2)
        assert ops.conv_kernel_name(d, 0) == "conv_stem_s2_kernel", ops.conv_kernel_name(d, 0)
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), ref, dtype, what="stem fprop")
        if big is not None:
            full = big.to_torch().cpu()
            assert float((full[..., :8] - 3.0).abs().max()) == 0.0 and float((full[..., 8 + Cout:] - 3.0).abs().max()) == 0.0
        slots = 8
        shift = rnd((Cout,), 33, 0.3).float()
        shift_d = shift.to(ops.device())
        part = torch.zeros(slots * 2 * Cout, dtype=torch.float32, device=ops.device())
        d3 = ops.conv_desc(b["xa"], wp, ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias)
        d3.stats_partial, d3.stats_shift, d3.stats_slots = part.data_ptr(), shift_d.data_ptr(), slots
        ops.conv_fprop(d3)
        y = ya.to_torch().cpu().double().reshape(M, Cout)
        got = part.cpu().double().reshape(slots, 2, Cout).sum(0)
        dd = y - shift.double()
        assert float((got[0] - dd.sum(0)).abs().max()) <= 1e-4 * (float(dd.abs().sum(0).max()) + 1e-9)
        assert float((got[1] - (dd * dd).sum(0)).abs().max()) <= 1e-4 * float((dd * dd).sum(0).max())
        # the im2col kernels on the same descriptor: equal to rounding
        lib.hdu_set_tuning(29, 1)
        assert ops.conv_kernel_name(d, 0) != "conv_stem_s2_kernel"
        got_stem = ya.to_torch().cpu().double()
        ops.conv_fprop(d)
        assert_close(got_stem, ya.to_torch().cpu().double(), dtype, scale=float(ref.abs().max()), what="stem kernel vs im2col")
    finally:
        lib.hdu_set_tuning(29, 0)


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in STEM_CASES])
def test_conv_stem_wgrad_kernel(hdu, cs):
    """round 5: the stem's filter gradient on its own kernel (conv_stem_wgrad_kernel: one depth tap and a range of output tiles per
    workgroup, wave = kh, pixels as the MFMA k axis, both operands through transposing LDS reads) against autograd of the float64
    reference; a second call accumulates."""
    import ctypes
    ops = ops_mod()
    dtype = BF16
    b = build_conv_case(ops, cs, dtype, seed=100)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    dy = rnd((N, Do, Ho, Wo, Cout), 777, 1.0, dtype)
    dya = mkact(ops, dy, dtype, cs["ldout"], 8 if cs["ldout"] else 0)
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, None)
    assert ops.conv_kernel_name(d, 1) == "conv_stem_wgrad_kernel", ops.conv_kernel_name(d, 1)
    dw = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
    ops.conv_wgrad(d, dw)
    xe = ref_xeff(b["x"], cs["up"], None, None, True, dtype).requires_grad_(True)
    wref = b["w"].clone().requires_grad_(True)
    (ref_conv(xe, wref, cs["s"], cs["p"], None) * dy).sum().backward()
    assert_close(dw.cpu(), wref.grad, BF16, what="stem wgrad")
    ops.conv_wgrad(d, dw)
    assert_close(dw.cpu(), 2 * wref.grad, BF16, what="stem wgrad accumulate")


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES if c["id"].startswith("halo_tile")])
def test_conv_wgrad_halo_large_tensor_path(hdu, cs):
    """round 5: tensors of 4 GiB and more (the whole 512^3 volume on one GPU) stay on the halo-tile filter gradient -- the buffer
    resources are made per input / output plane from 64-bit addresses.  HDU_TUNE_DEBUG bit 4 forces that path on the small cases."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    dtype = BF16
    b = build_conv_case(ops, cs, dtype, seed=100)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    dy = rnd((N, Do, Ho, Wo, Cout), 777, 1.0, dtype)
    dya = mkact(ops, dy, dtype, cs["ldout"], 8 if cs["ldout"] else 0)
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, None)
    dw = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
    try:
        lib.hdu_set_tuning(4, 16)
        assert ops.conv_kernel_name(d, 1).startswith("conv_wgrad_halo"), ops.conv_kernel_name(d, 1)
        ops.conv_wgrad(d, dw)
    finally:
        lib.hdu_set_tuning(4, 0)
    xe = ref_xeff(b["x"], cs["up"], None, None, True, dtype).requires_grad_(True)
    wref = b["w"].clone().requires_grad_(True)
    (ref_conv(xe, wref, cs["s"], cs["p"], None) * dy).sum().backward()
    assert_close(dw.cpu(), wref.grad, BF16, what="wgrad, per-plane resources")


def _halo_geometries():
    rng = np.random.default_rng(20260925)
    out = []
    while len(out) < 24:
        three_d = bool(rng.integers(0, 3))            # two thirds 3D
        up = tuple(int(v) for v in rng.integers(0, 2, 3)) if rng.integers(0, 3) == 0 else (0, 0, 0)
        if not three_d:
            up = (0, up[1], up[2])
        N, D = int(rng.integers(1, 3)), (int(rng.integers(1, 5)) if three_d else 1)
        H, W = int(rng.integers(2, 8)), int(rng.integers(12, 21)) if up[2] else int(rng.integers(24, 41))
        Cin, Cout = int(rng.choice([8, 16, 40, 64, 104])), int(rng.choice([8, 24, 48, 72]))
        pd = 1
        if three_d:
            pd = int(rng.choice([1, 0, -1])) if up[0] else int(rng.choice([1, 0]))
            De = D << up[0]
            if De + 2 * pd - 2 < 1:
                continue
        out.append(dict(N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, K=(3, 3, 3) if three_d else (1, 3, 3), s=(1, 1, 1),
                        p=(pd if three_d else 0, 1, 1), up=up, skip=False, pro=False, bias=False,
                        ldin=(Cin + 16 if rng.integers(0, 2) else None), ldout=None, id="geo%d" % len(out)))
    return out


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in _halo_geometries()])
def test_halo_filter_gradient_random_geometries(hdu, cs):
    """seeded sweep over the geometry space of the halo-tile filter gradient (2D / 3D, depth same / valid / cropped behind a depth
    up-sampling, up-sampling per axis, ragged tiles in H and W, ragged channel chunks, several volumes, slab inputs): every case
    must take the halo kernel and match the float64 reference"""
    import ctypes
    ops = ops_mod()
    b = build_conv_case(ops, cs, BF16, seed=700)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    dy = rnd((N, Do, Ho, Wo, Cout), 779, 1.0, BF16)
    dya = mkact(ops, dy, BF16)
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, None)
    assert ops.conv_kernel_name(d, 1).startswith("conv_wgrad_halo_kernel"), (cs, ops.conv_kernel_name(d, 1))
    dw = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
    ops.conv_wgrad(d, dw)
    xe = ref_xeff(b["x"], cs["up"], None, None, True, BF16).requires_grad_(True)
    wref = b["w"].clone().requires_grad_(True)
    (ref_conv(xe, wref, cs["s"], cs["p"], None) * dy).sum().backward()
    assert_close(dw.cpu(), wref.grad, BF16, what="halo wgrad %s" % (cs,))


def _halo_wide_geometries():
    rng = np.random.default_rng(20260926)
    out = []
    while len(out) < 24:
        three_d = bool(rng.integers(0, 3))            # two thirds 3D
        up = tuple(int(v) for v in rng.integers(0, 2, 3)) if rng.integers(0, 3) == 0 else (0, 0, 0)
        if not three_d:
            up = (0, up[1], up[2])
        N, D = int(rng.integers(1, 3)), (int(rng.integers(1, 5)) if three_d else 1)
        H, W = int(rng.integers(2, 20)), int(rng.integers(5, 21)) if up[2] else int(rng.integers(9, 41))
        Cin, Cout = int(rng.choice([8, 16, 24, 40, 56])), int(rng.choice([8, 32, 40, 72, 104, 136]))
        pd = 1
        if three_d:
            pd = int(rng.choice([1, 0, -1])) if up[0] else int(rng.choice([1, 0]))
            if (D << up[0]) + 2 * pd - 2 < 1:
                continue
        out.append(dict(N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, K=(3, 3, 3) if three_d else (1, 3, 3), s=(1, 1, 1),
                        p=(pd if three_d else 0, 1, 1), up=up, skip=False, pro=False, bias=bool(rng.integers(0, 2)),
                        ldin=(Cin + 16 if rng.integers(0, 2) else None), ldout=(Cout + 24 if rng.integers(0, 2) else None),
                        cfg=int(rng.integers(2, 9)), id="geo%d" % len(out)))
    return out


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in _halo_wide_geometries()])
def test_halo_wide_random_geometries(hdu, cs):
    """seeded sweep over the geometry space of the halo-tile forward / data-gradient kernel (2D / 3D, depth same / valid / cropped behind
    a depth up-sampling, up-sampling per axis, ragged tiles in H and W, ragged 16-channel stages and 32-channel output groups, several
    volumes, slab input / output, a random tile configuration each): plain + accumulate against the float64 reference"""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    b = build_conv_case(ops, cs, BF16, seed=900)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], BF16, zero=True)
        big.buf.fill_(3.0)
        ya = big.slab(8, Cout)
    else:
        big, ya = None, ops.Act.alloc(N, Do, Ho, Wo, Cout, BF16)
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], None, None, True, bias)
    ref = ref_conv(ref_xeff(b["x"], cs["up"], None, None, True, BF16), b["w"], cs["s"], cs["p"], b["bias"])
    try:
        lib.hdu_set_tuning(29, cs["cfg"])
        assert ops.conv_kernel_name(d, 0).startswith("conv_halo_wide_kernel"), (cs, ops.conv_kernel_name(d, 0))
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), ref, BF16, what="halo-wide fprop %s" % (cs,))
        if big is not None:
            full = big.to_torch().cpu()
            assert float((full[..., :8] - 3.0).abs().max()) == 0.0 and float((full[..., 8 + Cout:] - 3.0).abs().max()) == 0.0
        d.accumulate = 1
        ops.conv_fprop(d)
        assert_close(ya.to_torch().cpu(), q(ref, BF16) * 2, BF16, scale=2 * float(ref.abs().max()), what="halo-wide accumulate %s" % (cs,))
    finally:
        lib.hdu_set_tuning(29, 0)


SPLIT_CASES = [c for c in CONV_CASES if c["id"] in ("dense3x3_2d_slab", "bottleneck1x1", "decoder_up_skip_2d", "dense3x3x3", "stem7x7s2",
                                                     "wide_bn128", "pw_pro_two_stage", "pw_pro_wide_table_splitk")]


@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in SPLIT_CASES])
def test_conv_f32_bf16x3_contraction(hdu, cs):
    """HDU_TUNE_F32_SPLIT (lib.set_f32_contraction("bf16x3")): float32 storage, every operand split into bf16 hi + lo and contracted
    as ah.bh + ah.bl + al.bh on the bf16 MFMA with the float32 accumulator.  Per product the dropped terms are <= 3 * 2^-18 =
    1.1e-5 relative, so the result is held to 1.2e-5 * sum|a||b| (the float64 reference of the absolute products) -- and it is
    NOT the exact-float32 result, which the same launches give again after the mode is switched back."""
    import ctypes
    ops = ops_mod()
    b = build_conv_case(ops, cs, F32, seed=300)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, F32)
    pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro, True, None)
    dy = rnd((N, Do, Ho, Wo, Cout), 778, 1.0, F32)
    dya = mkact(ops, dy, F32)
    dg = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro, True, None)

    def run():
        ops.conv_fprop(d)
        dw = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
        ops.conv_wgrad(dg, dw)
        return ya.to_torch().cpu().double(), dw.cpu().double()

    y_exact, dw_exact = run()
    assert hdu.lib.set_f32_contraction("bf16x3") == "exact"
    try:
        y_split, dw_split = run()
    finally:
        assert hdu.lib.set_f32_contraction("exact") == "bf16x3"
    y_again, dw_again = run()
    assert torch.equal(y_again, y_exact)
    xe = ref_xeff(b["x"], cs["up"], b["skip"], b["pro"], True, F32).requires_grad_(True)
    wref = b["w"].clone().requires_grad_(True)
    ref = ref_conv(xe, wref, cs["s"], cs["p"], None)
    (ref * dy).sum().backward()
    mag_y = ref_conv(xe.detach().abs(), b["w"].abs(), cs["s"], cs["p"], None)                  # sum |a| |b| per output
    xa2 = xe.detach().abs().requires_grad_(False)
    wabs = b["w"].abs().clone().requires_grad_(True)
    (ref_conv(xa2, wabs, cs["s"], cs["p"], None) * dy.abs()).sum().backward()
    mag_w = wabs.grad
    for what, got, exact, r, mag in (("fprop", y_split, y_exact, ref.detach(), mag_y), ("wgrad", dw_split, dw_exact, wref.grad, mag_w)):
        err = (got - r).abs()
        lim = 1.2e-5 * mag + 1e-6 * float(r.abs().max())
        assert not (err > lim).any(), "%s: max err %.3e, max bound ratio %.3f" % (what, float(err.max()), float((err / lim).max()))
        assert not torch.equal(got, exact), "%s: the split form did not run" % what
        e_exact = float((exact - r).abs().max())
        assert float(err.max()) > e_exact, (what, float(err.max()), e_exact)


SPLIT3_CASES = [c for c in CONV_CASES if c["id"] in ("dense3x3_2d_slab", "bottleneck1x1", "dense3x3x3", "wide_bn128", "halo_tile_ragged_slab",
                                                      "halo_tile_3d_two_volumes_slab", "halo_tile_up221_3d", "pw_pro_two_stage",
                                                      "pw_pro_wide_table_splitk", "pw_pro_3d_bn128")]


def test_split3_filter_gradients(hdu):
    """hdu_split3_batched + the bf16 filter-gradient kernels on N' = 3 N images (round 6): float32 operands (some behind a BN
    prologue, some slabs of a wider buffer) and float32 output gradients are written ONCE as the bf16 image triples (hi, lo, hi) /
    (hi, hi, lo) by ONE launch; the bf16 per-layer kernel and the batched plan over the triples then give dyh.xh + dyh.xl + dyl.xh --
    held to the same 1.2e-5 * sum |dy| |x| bound as the in-kernel split (test_conv_f32_bf16x3_contraction), and the planes
    themselves to hi + lo = x within 2^-16 |x|."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib
    sp = ops.Split3Plan()
    plan = ops.WgradPlan()
    items = []
    for i, cs in enumerate(SPLIT3_CASES):
        b = build_conv_case(ops, cs, F32, seed=500 + 11 * i)
        N, Do, Ho, Wo, Cout = b["out_dims"]
        dy = rnd((N, Do, Ho, Wo, Cout), 940 + i, 1.0, F32)
        dya = mkact(ops, dy, F32, cs["ldout"], 8 if cs["ldout"] else 0)
        pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
        xs = ops.Act.alloc(3 * cs["N"], cs["D"], cs["H"], cs["W"], cs["Cin"], BF16)
        dys = ops.Act.alloc(3 * N, Do, Ho, Wo, Cout, BF16)
        sp.add(b["xa"], lib.SPLIT3_OPERAND, xs, pro, True)
        sp.add(dya, lib.SPLIT3_GRADIENT, dys)
        d16 = ops.conv_desc(xs, ctypes.c_void_p(b["wt"].data_ptr()), dys, cs["K"], cs["s"], cs["p"], cs["up"])
        dw1 = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
        dw2 = torch.zeros(b["w"].shape, dtype=torch.float32, device=ops.device())
        plan.add(d16, dw2)
        items.append((cs, b, dy, xs, dys, d16, dw1, dw2, pro))
    assert len(sp) == 2 * len(SPLIT3_CASES)
    sp.run()
    plan.run()
    for cs, b, dy, xs, dys, d16, dw1, dw2, pro in items:
        ops.conv_wgrad(d16, dw1)
        xe = ref_xeff(b["x"], (0, 0, 0), None, b["pro"], True, F32)          # what the planes hold (up-sampling stays in the conv)
        t = xs.to_torch().cpu().double()
        n = cs["N"]
        hi, lo, hi2 = t[:n], t[n:2 * n], t[2 * n:]
        # (behind a prologue the kernel's float32 a*x+b may differ from the float64 restatement by a float32 ulp: compare magnitudes)
        assert torch.equal(hi, hi2) and float((hi - xe).abs().max()) <= 2.0 ** -8 * float(xe.abs().max()), cs["id"]
        assert float(((hi + lo) - xe).abs().max()) <= (2.0 ** -16 + (2.0 ** -22 if b["pro"] else 0)) * float(xe.abs().max()), cs["id"]
        if not b["pro"]:
            assert torch.equal(hi, xe.to(torch.bfloat16).double()), cs["id"]
        g = dys.to_torch().cpu().double()
        assert torch.equal(g[:n], g[n:2 * n]) and torch.equal(g[:n], dy.to(torch.bfloat16).double()), cs["id"]
        assert float(((g[:n] + g[2 * n:]) - dy).abs().max()) <= 2.0 ** -16, cs["id"]
        xr = ref_xeff(b["x"], cs["up"], None, b["pro"], True, F32).requires_grad_(False)
        wref = b["w"].clone().requires_grad_(True)
        (ref_conv(xr, wref, cs["s"], cs["p"], None) * dy).sum().backward()
        wabs = b["w"].abs().clone().requires_grad_(True)
        (ref_conv(xr.abs(), wabs, cs["s"], cs["p"], None) * dy.abs()).sum().backward()
        lim = 1.2e-5 * wabs.grad + 1e-6 * float(wref.grad.abs().max())
        for what, got in (("per-layer", dw1), ("plan", dw2)):
            err = (got.cpu().double() - wref.grad).abs()
            assert not (err > lim).any(), "%s %s: max err %.3e, max bound ratio %.3f" % (cs["id"], what, float(err.max()), float((err / lim).max()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES
                                if c["id"] in ("dense3x3_2d_slab", "bottleneck1x1", "dense3x3x3", "stem7x7s2", "wide_bn128",
                                               "tile128x128_ragged_m", "halo_tile_ragged_slab", "halo_tile_exact")])
def test_conv_epilogue_statistics(hdu, cs, dtype):
    """moments taken in the conv epilogue (hdu_conv_desc.stats_*) + hdu_bn_stats_finalize == hdu_bn_stats_fold on the
    stored output: same mean / variance / folded a,b / moving statistics, for every kernel family that has the epilogue
    (VALU prologue form, DMA form, halo-tile form; ragged M / N tiles; slab output)."""
    import ctypes
    ops = ops_mod()
    b = build_conv_case(ops, cs, dtype, seed=40)
    N, Do, Ho, Wo, Cout = b["out_dims"]
    if cs["ldout"]:
        big = ops.Act.alloc(N, Do, Ho, Wo, cs["ldout"], dtype, zero=True)
        ya = big.slab(8, Cout)
    else:
        ya = ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype)
    pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
    bias = dev(ops, b["bias"]) if b["bias"] is not None else None
    slots = 5
    z = lambda n, v=0.0: torch.full((n,), v, dtype=torch.float32, device=ops.device())
    partial = z(slots * 2 * Cout)
    shift = dev(ops, rnd((Cout,), 11, 0.3).float().double())        # any value near the mean works
    d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), ya, cs["K"], cs["s"], cs["p"], cs["up"], b["sa"], pro,
                      True, bias, drop_keep=0.8, drop_seed=77)
    d.stats_partial, d.stats_shift, d.stats_slots = ctypes.c_void_p(partial.data_ptr()), ctypes.c_void_p(shift.data_ptr()), slots
    ops.conv_fprop(d)
    gamma, beta, sg, sb = (dev(ops, (rnd((Cout,), 20 + i, 0.2) + 1.0).float().double()) for i in range(4))
    outs = []
    for fused in (True, False):
        mean, var, a, bb, r = z(Cout), z(Cout), z(Cout), z(Cout), z(Cout)
        mm, mv = z(Cout, 0.5), z(Cout, 2.0)
        if fused:
            mean.copy_(shift)                                           # shift may alias mean
            ops.bn_stats_finalize(partial, slots, ya.M, Cout, mean, mean, var,
                                  (gamma, beta, 1.1e-5, sg, sb, a, bb, r, mm, mv, 0.99))
        else:
            ws = ops.Workspace(ops.reduce_ws_bytes(ya.M, Cout))
            ops.bn_stats_fold(ya, mean, var, gamma, beta, 1.1e-5, sg, sb, a, bb, r, mm, mv, 0.99, ws)
        outs.append([t.cpu().double() for t in (mean, var, a, bb, r, mm, mv)])
    y = ya.to_torch().cpu()
    assert float(y.abs().max()) > 0 and float((y == 0).double().mean()) > 0.1      # dropout is part of what is measured
    names = ("mean", "var", "a", "b", "rstd", "mov_mean", "mov_var")
    for nme, u, v in zip(names, *outs):
        tol = 2e-5 * max(1.0, float(v.abs().max()))
        assert float((u - v).abs().max()) <= tol, (nme, float((u - v).abs().max()))


def test_bn_stats_finalize_fold_next(hdu):
    """hdu_bn_stats_finalize_fold_next == hdu_bn_stats_finalize on the slab segment, then hdu_bn_fold of the next BN over
    the whole slab (exact: same arithmetic, deterministic inputs)"""
    ops = ops_mod()
    dev_ = ops.device()
    C_all, c0, Cseg, slots, M = 77, 48, 24, 5, 1000          # (the slab has 5 more channels after the segment's BN range)
    g = torch.Generator().manual_seed(3)
    rn = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float32)
    partial = (rn(slots, 2, Cseg).abs() * 50).to(dev_).reshape(-1)
    mean0, var0 = rn(C_all + 5), rn(C_all + 5).abs() + 0.1
    gamma, beta, sg, sb = (rn(C_all) * 0.2 + 1.0).to(dev_), rn(C_all).to(dev_), (rn(C_all) * 0.2 + 1.0).to(dev_), rn(C_all).to(dev_)
    outs = []
    for fused in (True, False):
        mean, var = mean0.clone().to(dev_), var0.clone().to(dev_)
        a, b, r = (torch.zeros(C_all, device=dev_) for _ in range(3))
        mm, mv = torch.full((C_all,), 0.5, device=dev_), torch.full((C_all,), 2.0, device=dev_)
        fold = (gamma, beta, 1.1e-5, sg, sb, a, b, r, mm, mv, 0.99)
        if fused:
            ops.bn_stats_finalize_fold_next(partial, slots, M, Cseg, c0, C_all, mean.clone()[:C_all], mean[:C_all], var[:C_all], fold)
        else:
            ops.bn_stats_finalize(partial, slots, M, Cseg, mean[c0:c0 + Cseg], mean[c0:c0 + Cseg], var[c0:c0 + Cseg])
            ops.bn_fold(C_all, mean[:C_all], var[:C_all], gamma, beta, 1.1e-5, sg, sb, a, b, r, mm, mv, 0.99)
        outs.append([t.cpu() for t in (mean, var, a, b, r, mm, mv)])
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert not torch.equal(outs[0][0][c0:c0 + Cseg], mean0[c0:c0 + Cseg]) and torch.equal(outs[0][0][C_all:], mean0[C_all:])
    with pytest.raises(hdu.lib.HduError):
        ops.bn_stats_finalize_fold_next(partial, slots, M, Cseg, C_all - 3, C_all, mean.clone(), mean, var, fold)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("seg", ["whole", "tail"])
@pytest.mark.parametrize("form", ["up-skip", "plain", "plain-rows-loop"])
def test_materialize_stats(hdu, dtype, seg, form):
    """hdu_materialize_stats == hdu_bn_stats_finalize(_fold_next) followed by hdu_materialize: the output tensor and
    everything the finalize launch would have published (a, b, rstd, the segment's moments, the moving averages), with an
    up-sampled input and a skip add, several column groups and row blocks.  Round 6: "plain" = no up-sampling / skip (the dense
    layers' call) -- at this size the launch takes the form that requests its rows before the slot-table round trip
    (materialize_kernel PRE); "plain-rows-loop" = the same call with that form switched off (HDU_TUNE_DEBUG bit 11)."""
    ops = ops_mod()
    up = (0, 1, 1) if form == "up-skip" else (0, 0, 0)
    if form == "plain-rows-loop":
        hdu.lib.get().hdu_set_tuning(4, 2048)
    try:
        _materialize_stats_case(hdu, ops, dtype, seg, up)
    finally:
        hdu.lib.get().hdu_set_tuning(4, 0)


def _materialize_stats_case(hdu, ops, dtype, seg, up):
    dev_ = ops.device()
    N, D, H, W = 2, 1, 12, 20
    C = 304 if seg == "tail" else 192                  # tail: a dense-block slab, the last 48 channels just written
    Cseg, c0 = (48, C - 48) if seg == "tail" else (C, 0)
    slots, M = 32, N * D * H * W
    g = torch.Generator().manual_seed(11)
    rn = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float32)
    x = q(rnd((N, D, H, W, C), 21, 2.0, dtype) + 0.3, dtype)
    skip = q(rnd((N, D, 2 * H, 2 * W, C), 22, 1.0, dtype), dtype) if up != (0, 0, 0) else None
    xa, ska = mkact(ops, x, dtype), (mkact(ops, skip, dtype) if skip is not None else None)
    shift0 = rn(C) * 0.5
    # slot sums of (y - shift), (y - shift)^2 over the segment channels, spread over the slot rows like a conv epilogue
    xs = x.reshape(-1, C)[:, c0:c0 + Cseg] - shift0[c0:c0 + Cseg].double()
    rows = torch.arange(M) % slots
    part = torch.zeros(slots, 2, Cseg, dtype=torch.float64)
    part[:, 0].index_add_(0, rows, xs)
    part[:, 1].index_add_(0, rows, xs * xs)
    partial = part.float().to(dev_).reshape(-1)
    mean0, var0 = rn(C) * 0.3, rn(C).abs() + 0.2
    gamma, beta, sg, sb = (rn(C) * 0.2 + 1.0).to(dev_), rn(C).to(dev_), (rn(C) * 0.2 + 1.0).to(dev_), rn(C).to(dev_)
    outs = []
    for fused in (True, False):
        shift, mean, var = shift0.clone().to(dev_), mean0.clone().to(dev_), var0.clone().to(dev_)
        a, b, r = (torch.zeros(C, device=dev_) for _ in range(3))
        mm, mv = torch.full((C,), 0.5, device=dev_), torch.full((C,), 2.0, device=dev_)
        fold = (gamma, beta, 1.1e-5, sg, sb, a, b, r, mm, mv, 0.99)
        out = ops.Act.alloc(N, D, H << up[1], W << up[2], C, dtype)
        if fused:
            ops.materialize_stats(xa, (partial, slots, M, Cseg, c0, shift, mean, var, fold), True, up, ska, out)
        else:
            if seg == "tail":
                ops.bn_stats_finalize_fold_next(partial, slots, M, Cseg, c0, C, shift, mean, var, fold)
            else:
                ops.bn_stats_finalize(partial, slots, M, C, shift, mean, var, fold)
            ops.materialize(xa, a, b, True, up, ska, out)
        outs.append([t.cpu() for t in (mean, var, a, b, r, mm, mv)] + [out.to_torch().cpu().float()])
    names = ["mean", "var", "a", "b", "rstd", "mov_mean", "mov_var", "out"]
    for nm, u, v in zip(names, *outs):
        tolv = 2e-2 if (nm == "out" and dtype == BF16) else 2e-5
        assert float((u.double() - v.double()).abs().max()) <= tolv * max(1.0, float(v.abs().max())), nm
    # the statistics really are those of the tensor
    xm = x.reshape(-1, C)[:, c0:c0 + Cseg]
    assert float((outs[0][0][c0:c0 + Cseg].double() - xm.mean(0)).abs().max()) < 1e-4
    assert float((outs[0][1][c0:c0 + Cseg].double() - xm.var(0, unbiased=False)).abs().max()) < 1e-3
    with pytest.raises(hdu.lib.HduError):
        ops.materialize_stats(xa, (partial, slots, M, Cseg, c0 + 4, shift, mean, var, fold), True, up, ska, out)


def test_conv_wgrad_batched_plan(hdu):
    """hdu_wgrad_plan_*: ONE launch per kernel family over many layers == hdu_conv_wgrad per layer (bf16,
    materialised inputs: every CONV_CASE without prologue / skip, plus extra 1x1 and 3x3 shapes)"""
    import ctypes
    ops = ops_mod()
    extra = [
        dict(N=2, D=1, H=16, W=16, Cin=64, Cout=192, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="pw192"),
        dict(N=1, D=1, H=12, W=33, Cin=136, Cout=192, K=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=160, ldout=None, id="pw_ragged"),
        dict(N=2, D=1, H=8, W=32, Cin=64, Cout=48, K=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=96, id="halo48"),
        dict(N=1, D=2, H=6, W=6, Cin=16, Cout=32, K=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), up=(0, 0, 0), skip=False, pro=False, bias=False, ldin=None, ldout=None, id="c333"),
    ]
    # (round 4: pointwise layers WITH a BN prologue are batched too -- the affine is recomputed on the x fragments)
    pointwise = lambda c: c["K"] == (1, 1, 1) and c["up"] == (0, 0, 0)
    cases = [c for c in CONV_CASES if (not c["pro"] or pointwise(c)) and not c["skip"]] + extra
    assert sum(1 for c in cases if c["pro"]) >= 4
    plan = ops.WgradPlan()
    items = []
    for i, cs in enumerate(cases):
        b = build_conv_case(ops, cs, BF16, seed=300 + 7 * i)
        N, Do, Ho, Wo, Cout = b["out_dims"]
        dya = mkact(ops, rnd((N, Do, Ho, Wo, Cout), 900 + i, 1.0, BF16), BF16, cs["ldout"], 8 if cs["ldout"] else 0)
        pro = (dev(ops, b["pro"][0]), dev(ops, b["pro"][1])) if b["pro"] else None
        b["pro_dev"] = pro
        d = ops.conv_desc(b["xa"], ctypes.c_void_p(b["wt"].data_ptr()), dya, cs["K"], cs["s"], cs["p"], cs["up"], None, pro, True)
        ref = torch.full(b["w"].shape, 0.25, dtype=torch.float32, device=ops.device())     # dw += ...
        got = ref.clone()
        ops.conv_wgrad(d, ref)
        plan.add(d, got)
        items.append((cs["id"], ref, got, b, dya))
    plan.finalize()
    assert len(plan) == len(cases) and len(plan.by_variant) >= 3      # several kernel families in one plan
    plan.run()
    for name, ref, got, _, _ in items:
        scale = float(ref.abs().max())
        assert scale > 0.3
        # identical tiles and splits; only the order of the float atomics differs
        assert float((ref - got).abs().max()) <= 2e-5 * scale, name
    plan.run()                                                          # accumulates again
    for name, ref, got, _, _ in items:
        assert float((got - 0.25 - 2 * (ref - 0.25)).abs().max()) <= 1e-4 * float(ref.abs().max()), name


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in CONV_CASES if c["s"] == (1, 1, 1)])
def test_conv_dgrad_via_fprop(hdu, cs, dtype, dma_stages):
    """data gradient of a stride-1 conv = fprop with the flipped/transposed filter from hdu_weight_prep"""
    ops = ops_mod()
    hdu.lib.get().hdu_set_tuning(0, dma_stages)
    import ctypes
    K, p, up = cs["K"], cs["p"], cs["up"]
    N, D, H, W, Cin, Cout = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cin"], cs["Cout"]
    De, He, We = D << up[0], H << up[1], W << up[2]
    w = rnd((Cout,) + K + (Cin,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cin), F32)
    T = K[0] * K[1] * K[2]
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    wf = torch.empty(Cout * T * Cin, dtype=tdt, device=ops.device())
    wd = torch.empty(Cout * T * Cin, dtype=tdt, device=ops.device())
    ops.weight_prep(dtype, dev(ops, w).reshape(-1), Cout, T, Cin, wf, wd)
    assert_close(wf.cpu().reshape(w.shape), w, dtype, what="weight_prep fwd copy")
    dy = rnd((N, De, He, We, Cout), 9, 1.0, dtype)   # stride 1 + same-size padding cases only produce De,He,We
    Do, Ho, Wo = De + 2 * p[0] - K[0] + 1, He + 2 * p[1] - K[1] + 1, We + 2 * p[2] - K[2] + 1
    dy = rnd((N, Do, Ho, Wo, Cout), 9, 1.0, dtype)
    dya = mkact(ops, dy, dtype)
    dxe = ops.Act.alloc(N, De, He, We, Cin, dtype)
    d = ops.conv_desc(dya, ctypes.c_void_p(wd.data_ptr()), dxe, K, (1, 1, 1), (K[0] - 1 - p[0], K[1] - 1 - p[1], K[2] - 1 - p[2]))
    ops.conv_fprop(d)
    xe = torch.zeros((N, De, He, We, Cin), dtype=torch.float64, requires_grad=True)
    y = ref_conv(xe, q(w, dtype), (1, 1, 1), p, None)
    (y * dy).sum().backward()
    hdu.lib.get().hdu_set_tuning(0, 2)
    assert_close(dxe.to_torch().cpu(), xe.grad, dtype, what="dgrad")


BNB_CASES = [
    # data-gradient launches whose output is dz of a BN(+Scale)+ReLU: (pixels, dy channels = K side, dz channels = N side)
    dict(N=2, D=1, H=9, W=7, Cdy=192, Cu=144, K=(1, 1, 1), p=(0, 0, 0), ldu=200, acc=True, sums=True, relu=True, id="x1_1x1_slab_acc"),
    dict(N=1, D=1, H=10, W=12, Cdy=48, Cu=192, K=(1, 3, 3), p=(0, 1, 1), ldu=None, acc=False, sums=True, relu=True, id="x2_3x3_two_tiles"),
    dict(N=1, D=3, H=5, W=6, Cdy=32, Cu=128, K=(3, 3, 3), p=(1, 1, 1), ldu=None, acc=False, sums=True, relu=True, id="x2_3x3x3"),
    dict(N=1, D=1, H=16, W=16, Cdy=192, Cu=1056, K=(1, 1, 1), p=(0, 0, 0), ldu=None, acc=True, sums=False, relu=True, id="frozen_no_sums_wide"),
    dict(N=1, D=1, H=6, W=6, Cdy=24, Cu=40, K=(1, 1, 1), p=(0, 0, 0), ldu=56, acc=False, sums=True, relu=False, id="no_relu_ragged_n"),
    # round 4: the filter-stationary streaming kernel (bf16, K = 192 / 128, >= 256 dz channels) with the BN backward in its tile
    # epilogue: ragged pixel count (several tiles per workgroup + a tail), channels that do not fill the last 128-group, slab
    # output with accumulation, a 3D bottleneck, and the overwrite form
    dict(N=2, D=1, H=13, W=11, Cdy=192, Cu=328, K=(1, 1, 1), p=(0, 0, 0), ldu=344, acc=True, sums=True, relu=True, id="pw_bstat_k192_slab_acc"),
    dict(N=1, D=3, H=9, W=10, Cdy=128, Cu=256, K=(1, 1, 1), p=(0, 0, 0), ldu=None, acc=False, sums=True, relu=True, id="pw_bstat_k128_3d"),
    dict(N=1, D=1, H=40, W=33, Cdy=192, Cu=1056, K=(1, 1, 1), p=(0, 0, 0), ldu=None, acc=True, sums=True, relu=True, id="pw_bstat_many_tiles"),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in BNB_CASES])
def test_conv_fused_bn_backward(hdu, cs, dtype):
    """hdu_conv_desc.bnb_*: the epilogue of a data-gradient launch stores a*g (g = dz masked by the forward ReLU) into /
    onto the gradient slab and leaves S1 = sum g, S2 = sum g*uhat in slot rows; hdu_bn_bwd_finalize turns them into the
    parameter gradients and the corr3 / corr4 accumulators, hdu_bn_bwd_correct applies -corr3*u + corr4: together
    exactly tf.gradients of BN(+Scale)+ReLU (TFB:1639) in float64."""
    import ctypes
    ops = ops_mod()
    N, D, H, W, Cdy, Cu, K, p = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cdy"], cs["Cu"], cs["K"], cs["p"]
    M = N * D * H * W
    dy = rnd((N, D, H, W, Cdy), 1, 1.0, dtype)
    w = rnd((Cu,) + K + (Cdy,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cdy), dtype)      # the data-gradient filter [N side][taps][K side]
    u = rnd((N, D, H, W, Cu), 7, 1.0, dtype)
    old = rnd((N, D, H, W, Cu), 8, 0.5, dtype)
    a = (rnd((Cu,), 9, 1.0).abs() + 0.3).float().double()
    b = rnd((Cu,), 10, 0.4).float().double()
    mean = rnd((Cu,), 11, 0.3).float().double()
    rstd = (rnd((Cu,), 12, 0.5).abs() + 0.5).float().double()
    dya = mkact(ops, dy, dtype)
    ua = mkact(ops, u, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
    outa = mkact(ops, old, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    slots = 5
    keep = [dev(ops, t) for t in (a, b, mean, rstd)]
    partial = torch.zeros(slots * 2 * Cu, dtype=torch.float32, device=ops.device())
    d = ops.conv_desc(dya, ctypes.c_void_p(wt.data_ptr()), outa, K, (1, 1, 1), p, accumulate=cs["acc"])
    d.bnb_u, d.bnb_ldu = ua.ptr, ua.ld
    d.bnb_a, d.bnb_b, d.bnb_relu = keep[0].data_ptr(), keep[1].data_ptr(), 1 if cs["relu"] else 0
    if cs["sums"]:
        d.bnb_mean, d.bnb_rstd, d.bnb_partial, d.bnb_slots = keep[2].data_ptr(), keep[3].data_ptr(), partial.data_ptr(), slots
    if cs["id"].startswith("pw_bstat") and dtype == BF16:
        assert ops.conv_kernel_name(d, 0).startswith("conv_pw_bstat_kernel") and ", true" in ops.conv_kernel_name(d, 0)
    ops.conv_fprop(d)
    # reference: dz = conv(dy, w) in the storage dtype (the tile is staged in it), then the masked scale
    dz = q(ref_conv(dy, w, (1, 1, 1), p, None), dtype)
    s = a * u + b
    g = torch.where(s > 0, dz, torch.zeros_like(dz)) if cs["relu"] else dz
    ref = a * g + (q(old, dtype) if cs["acc"] else 0.0)
    assert_close(outa.to_torch().cpu(), ref, dtype, scale=float(ref.abs().max()), what="a*g")
    if cs["ldu"]:       # neighbouring slab channels untouched
        full = ops.Act(outa.buf, 0, N, D, H, W, cs["ldu"], cs["ldu"], dtype).to_torch().cpu()
        assert float((full[..., :8] - 7.0).abs().max()) == 0.0 and float((full[..., 8 + Cu:] - 7.0).abs().max()) == 0.0
    if not cs["sums"]:
        return
    S = partial.cpu().double().reshape(slots, 2, Cu).sum(0)
    S1, S2 = g.reshape(M, Cu).sum(0), (g * (u - mean) * rstd).reshape(M, Cu).sum(0)
    tol = 2e-5 if dtype == F32 else 2e-3
    sc = float(max(S1.abs().max(), S2.abs().max()))
    assert float((S[0] - S1).abs().max()) <= tol * sc and float((S[1] - S2).abs().max()) <= tol * sc
    # finalize: parameter gradients + deferred coefficients; correct: du += -corr3*u + corr4
    gamma = (rnd((Cu,), 13, 0.5).abs() + 0.5).float().double()
    beta = rnd((Cu,), 14, 0.3).float().double()
    sg = (rnd((Cu,), 15, 0.5).abs() + 0.5).float().double()
    z = lambda: torch.zeros(Cu, dtype=torch.float32, device=ops.device())
    dg, db, dsg, dsb, c3, c4 = z(), z(), z(), z(), z(), z()
    c3.fill_(0.25)                                        # the accumulators ADD (other consumers were there first)
    kp = [dev(ops, t) for t in (gamma, beta, sg)]
    ops.bn_bwd_finalize(partial, slots, M, Cu, True, kp[0], kp[1], kp[2], keep[2], keep[3], dg, db, dsg, dsb, c3, c4)
    kk = sg * gamma * rstd
    k2, k3 = kk * S1 / M, kk * rstd * S2 / M
    for got, want, what in ((dg, sg * S2, "dgamma"), (db, sg * S1, "dbeta"), (dsg, gamma * S2 + beta * S1, "dsgamma"),
                            (dsb, S1, "dsbeta"), (c3, 0.25 + k3, "corr3"), (c4, k3 * mean - k2, "corr4")):
        assert float((got.cpu().double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), what
    before = outa.to_torch().cpu()
    ops.bn_bwd_correct(ua, c3, c4, outa)
    want = before - c3.cpu().double() * q(u, dtype) + c4.cpu().double()
    assert_close(outa.to_torch().cpu(), want, dtype, scale=float(want.abs().max()), what="correct")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in BNB_CASES if not c["id"].startswith("pw_bstat") and c["sums"]])
def test_conv_bn_backward_sums_in_epilogue_then_apply(hdu, cs, dtype):
    """Round 6 (hdu_conv_desc.bnb_relu bit 2 + hdu_bn_bwd_apply_sums): the data-gradient launch stores RAW dz and adds S1 = sum g,
    S2 = sum g*uhat to the BN's slot table in its epilogue; the apply launch derives k1 / k2 / k3 from the table and writes
    du = k1*g - k2 - k3*(u - mean) (+ old) and the parameter gradients: together exactly tf.gradients of BN(+Scale)+ReLU (TFB:1639)
    in float64 -- the reduce_rows launch between the two is gone.  Also: the two-launch form hdu_bn_bwd_fused on the same dz."""
    import ctypes
    ops = ops_mod()
    N, D, H, W, Cdy, Cu, K, p = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cdy"], cs["Cu"], cs["K"], cs["p"]
    M = N * D * H * W
    dy = rnd((N, D, H, W, Cdy), 1, 1.0, dtype)
    w = rnd((Cu,) + K + (Cdy,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cdy), dtype)
    u = rnd((N, D, H, W, Cu), 7, 1.0, dtype)
    old = rnd((N, D, H, W, Cu), 8, 0.5, dtype)
    gamma = (rnd((Cu,), 13, 0.5).abs() + 0.5).float().double()
    beta = rnd((Cu,), 14, 0.3).float().double()
    sg = (rnd((Cu,), 15, 0.5).abs() + 0.5).float().double()
    sb = rnd((Cu,), 16, 0.3).float().double()
    mean = rnd((Cu,), 11, 0.3).float().double()
    rstd = (rnd((Cu,), 12, 0.5).abs() + 0.5).float().double()
    a = (sg * gamma * rstd).float().double()                       # the folded BN(+Scale): z = relu(a*u + b)
    b = (sg * (beta - gamma * rstd * mean) + sb).float().double()
    dya = mkact(ops, dy, dtype)
    ua = mkact(ops, u, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
    dza = ops.Act.alloc(N, D, H, W, Cu, dtype)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    slots = 16
    keep = [dev(ops, t) for t in (a, b, mean, rstd, gamma, beta, sg)]
    table = torch.zeros(slots * 2 * Cu, dtype=torch.float32, device=ops.device())
    d = ops.conv_desc(dya, ctypes.c_void_p(wt.data_ptr()), dza, K, (1, 1, 1), p)
    d.bnb_u, d.bnb_ldu = ua.ptr, ua.ld
    d.bnb_a, d.bnb_b, d.bnb_relu = keep[0].data_ptr(), keep[1].data_ptr(), (1 if cs["relu"] else 0) | 4
    d.bnb_mean, d.bnb_rstd, d.bnb_partial, d.bnb_slots = keep[2].data_ptr(), keep[3].data_ptr(), table.data_ptr(), slots
    assert not ops.conv_kernel_name(d, 0).startswith("conv_pw_bstat"), "the streaming kernel has no sums-only form"
    ops.conv_fprop(d)
    dz = q(ref_conv(dy, w, (1, 1, 1), p, None), dtype)
    assert_close(dza.to_torch().cpu(), dz, dtype, scale=float(dz.abs().max()), what="raw dz")
    dzs = dza.to_torch().cpu().double()                        # the STORED dz: what the apply pass reads
    uq = q(u, dtype)
    gmask = (a * uq + b > 0) if cs["relu"] else torch.ones_like(dz, dtype=torch.bool)
    g = torch.where(gmask, dz, torch.zeros_like(dz))
    S = table.cpu().double().reshape(slots, 2, Cu).sum(0)
    S1, S2 = g.reshape(M, Cu).sum(0), (g * (uq - mean) * rstd).reshape(M, Cu).sum(0)
    tol = 2e-5 if dtype == F32 else 2e-3
    sc = float(max(S1.abs().max(), S2.abs().max()))
    assert float((S[0] - S1).abs().max()) <= tol * sc and float((S[1] - S2).abs().max()) <= tol * sc, "slot sums"
    z = lambda: torch.zeros(Cu, dtype=torch.float32, device=ops.device())
    outs = {}
    for form in ("apply_sums", "fused"):
        dg, db, dsg, dsb = z(), z(), z(), z()
        outa = mkact(ops, old, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
        tbl = table.clone() if form == "apply_sums" else torch.zeros_like(table)
        ops.bn_bwd_fused(dza, ua, keep[0], keep[1], cs["relu"], keep[2], keep[3], True, keep[4], keep[5], keep[6], tbl, slots,
                         dg, db, dsg, dsb, outa, accumulate=cs["acc"], sums_ready=(form == "apply_sums"))
        # reference from the stored dz (float64): du = k1*g - k2 - k3*(u - mean)
        gs = torch.where(gmask, dzs, torch.zeros_like(dzs))
        T1, T2 = gs.reshape(M, Cu).sum(0), (gs * (uq - mean) * rstd).reshape(M, Cu).sum(0)
        kk = sg * gamma * rstd
        want = kk * gs - kk * T1 / M - kk * rstd * T2 / M * (uq - mean) + (q(old, dtype) if cs["acc"] else 0.0)
        assert_close(outa.to_torch().cpu(), want, dtype, scale=float(want.abs().max()), what="du (%s)" % form)
        for got, ref_v, what in ((dg, sg * T2, "dgamma"), (db, sg * T1, "dbeta"), (dsg, gamma * T2 + beta * T1, "dsgamma"), (dsb, T1, "dsbeta")):
            assert float((got.cpu().double() - ref_v).abs().max()) <= tol * max(1.0, float(ref_v.abs().max())), (what, form)
        outs[form] = outa.to_torch().cpu().double()
        if cs["ldu"]:
            full = ops.Act(outa.buf, 0, N, D, H, W, cs["ldu"], cs["ldu"], dtype).to_torch().cpu()
            assert float((full[..., :8] - 7.0).abs().max()) == 0.0 and float((full[..., 8 + Cu:] - 7.0).abs().max()) == 0.0
    # the two forms agree to the rounding of their sums (float atomics in different orders)
    assert_close(outs["apply_sums"], outs["fused"], dtype, scale=float(outs["fused"].abs().max()), what="epilogue sums vs reduction pass")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in BNB_CASES if not c["id"].startswith("pw_bstat")])
def test_conv_fused_bn_backward_from_output(hdu, cs, dtype):
    """hdu_conv_desc.bnb_relu bit 1: `bnb_u` holds z = relu(a*u + b) -- what a producer with the BN in its epilogue stored -- and
    the epilogue takes the ReLU mask (z > 0) and the normalised input ((z - (b + a*mean)) * rstd / a) from it: same a*g, and
    S2 equal to the float64 evaluation of that formula on the stored z (and, within the storage rounding of z, to the u form)."""
    import ctypes
    ops = ops_mod()
    N, D, H, W, Cdy, Cu, K, p = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cdy"], cs["Cu"], cs["K"], cs["p"]
    M = N * D * H * W
    dy = rnd((N, D, H, W, Cdy), 1, 1.0, dtype)
    w = rnd((Cu,) + K + (Cdy,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cdy), dtype)
    u = rnd((N, D, H, W, Cu), 7, 1.0, dtype)
    old = rnd((N, D, H, W, Cu), 8, 0.5, dtype)
    a = (rnd((Cu,), 9, 1.0).abs() + 0.3).float().double() * torch.where(torch.arange(Cu) % 3 == 0, -1.0, 1.0)   # both signs
    b = rnd((Cu,), 10, 0.4).float().double()
    mean = rnd((Cu,), 11, 0.3).float().double()
    rstd = (rnd((Cu,), 12, 0.5).abs() + 0.5).float().double()
    s = a * u + b
    z = q(s.clamp_min(0) if cs["relu"] else s, dtype)
    dya = mkact(ops, dy, dtype)
    za = mkact(ops, z, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
    outa = mkact(ops, old, dtype, cs["ldu"], 8 if cs["ldu"] else 0)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    slots = 5
    keep = [dev(ops, t) for t in (a, b, mean, rstd)]
    partial = torch.zeros(slots * 2 * Cu, dtype=torch.float32, device=ops.device())
    d = ops.conv_desc(dya, ctypes.c_void_p(wt.data_ptr()), outa, K, (1, 1, 1), p, accumulate=cs["acc"])
    d.bnb_u, d.bnb_ldu = za.ptr, za.ld
    d.bnb_a, d.bnb_b, d.bnb_relu = keep[0].data_ptr(), keep[1].data_ptr(), (1 if cs["relu"] else 0) | 2
    if cs["sums"]:
        d.bnb_mean, d.bnb_rstd, d.bnb_partial, d.bnb_slots = keep[2].data_ptr(), keep[3].data_ptr(), partial.data_ptr(), slots
    assert not ops.conv_kernel_name(d, 0).startswith("conv_pw_bstat_kernel")
    ops.conv_fprop(d)
    dz = q(ref_conv(dy, w, (1, 1, 1), p, None), dtype)
    g = torch.where(z > 0, dz, torch.zeros_like(dz)) if cs["relu"] else dz
    ref = a * g + (q(old, dtype) if cs["acc"] else 0.0)
    assert_close(outa.to_torch().cpu(), ref, dtype, scale=float(ref.abs().max()), what="a*g from z")
    if not cs["sums"]:
        return
    S = partial.cpu().double().reshape(slots, 2, Cu).sum(0)
    uhat_z = (z - (b + a * mean)) * (rstd / a)
    S1, S2 = g.reshape(M, Cu).sum(0), (g * uhat_z).reshape(M, Cu).sum(0)
    tol = 2e-5 if dtype == F32 else 2e-3
    sc = float(max(S1.abs().max(), S2.abs().max()))
    assert float((S[0] - S1).abs().max()) <= tol * sc and float((S[1] - S2).abs().max()) <= tol * sc
    S2_u = (g * (u - mean) * rstd).reshape(M, Cu).sum(0)            # the form that reads u: equal up to the rounding of the stored z
    assert float((S[1] - S2_u).abs().max()) <= (1e-4 if dtype == F32 else 2e-2) * sc


def test_conv_fused_bn_backward_from_output_dead_channel(hdu):
    """bnb_relu bit 1 with a channel whose folded scale a is exactly 0 (a dead Scale gamma): z = relu(b) carries no information
    about the BN input, so that channel's S2 is defined as 0 -- and nothing may turn into NaN / Inf (rstd / a is guarded)."""
    import ctypes
    ops = ops_mod()
    N, D, H, W, Cdy, Cu, K, p = 1, 1, 10, 12, 48, 64, (1, 3, 3), (0, 1, 1)
    dy = rnd((N, D, H, W, Cdy), 1, 1.0, F32)
    w = rnd((Cu,) + K + (Cdy,), 5, 1.0 / np.sqrt(9 * Cdy), F32)
    u = rnd((N, D, H, W, Cu), 7, 1.0, F32)
    a = (rnd((Cu,), 9, 1.0).abs() + 0.3).float().double()
    a[5] = 0.0
    a[40] = 0.0
    b = rnd((Cu,), 10, 0.4).float().double()
    b[5] = 0.25                                          # relu(b) > 0: the mask is open on the dead channel
    mean = rnd((Cu,), 11, 0.3).float().double()
    rstd = (rnd((Cu,), 12, 0.5).abs() + 0.5).float().double()
    z = q((a * u + b).clamp_min(0), F32)
    dya, za = mkact(ops, dy, F32), mkact(ops, z, F32)
    outa = ops.Act.alloc(N, D, H, W, Cu, F32)
    wt = w.float().contiguous().to(ops.device())
    slots = 3
    keep = [dev(ops, t) for t in (a, b, mean, rstd)]
    partial = torch.zeros(slots * 2 * Cu, dtype=torch.float32, device=ops.device())
    d = ops.conv_desc(dya, ctypes.c_void_p(wt.data_ptr()), outa, K, (1, 1, 1), p)
    d.bnb_u, d.bnb_ldu = za.ptr, za.ld
    d.bnb_a, d.bnb_b, d.bnb_relu = keep[0].data_ptr(), keep[1].data_ptr(), 3
    d.bnb_mean, d.bnb_rstd, d.bnb_partial, d.bnb_slots = keep[2].data_ptr(), keep[3].data_ptr(), partial.data_ptr(), slots
    ops.conv_fprop(d)
    out = outa.to_torch().cpu().double()
    S = partial.cpu().double().reshape(slots, 2, Cu).sum(0)
    assert torch.isfinite(out).all() and torch.isfinite(S).all()
    assert float(out[..., 5].abs().max()) == 0.0 and float(out[..., 40].abs().max()) == 0.0
    assert float(S[1, 5]) == 0.0 and float(S[1, 40]) == 0.0
    dz = ref_conv(dy, w, (1, 1, 1), p, None)
    g = torch.where(z > 0, dz, torch.zeros_like(dz))
    assert abs(float(S[0, 5]) - float(g[..., 5].sum())) <= 2e-5 * float(g[..., 5].abs().sum())     # S1 of the dead channel is still the masked sum
    live = [c for c in range(Cu) if c not in (5, 40)]
    uhat = (z - (b + a * mean)) * torch.where(a != 0, rstd / torch.where(a != 0, a, torch.ones_like(a)), torch.zeros_like(a))
    S2 = (g * uhat).reshape(-1, Cu).sum(0)
    assert float((S[1][live] - S2[live]).abs().max()) <= 2e-5 * float(S2[live].abs().max())


SPLITK_CASES = [
    # dense-block shapes whose output grid cannot fill the chip (include/hdu.h, hdu_conv_desc.splitk_ws)
    dict(N=2, D=1, H=16, W=15, Cin=192, Cout=48, K=(1, 3, 3), p=(0, 1, 1), bias=False, ldout=96, id="block5_3x3_192to48"),
    dict(N=1, D=3, H=7, W=7, Cin=128, Cout=32, K=(3, 3, 3), p=(1, 1, 1), bias=True, ldout=None, id="3d_block5_3x3x3_128to32"),
    dict(N=1, D=1, H=10, W=20, Cin=1056, Cout=192, K=(1, 1, 1), p=(0, 0, 0), bias=False, ldout=None, id="block5_1x1_1056to192"),
    dict(N=1, D=1, H=9, W=12, Cin=96, Cout=128, K=(1, 3, 3), p=(0, 1, 1), bias=True, ldout=None, id="n128_waves2x2"),
    dict(N=1, D=1, H=8, W=8, Cin=160, Cout=64, K=(1, 3, 3), p=(0, 1, 1), bias=True, ldout=None, id="n64_waves2x2"),
    dict(N=1, D=2, H=4, W=4, Cin=64, Cout=40, K=(3, 3, 3), p=(1, 1, 1), bias=True, ldout=None, up=(1, 1, 1), id="upsampled_3d_decoder"),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cs", [pytest.param(c, id=c["id"]) for c in SPLITK_CASES])
def test_conv_splitk(hdu, cs, dtype):
    """K loop dealt to S workgroups per output tile, partial tiles summed by the tile's last arriver: same result as the
    unsplit launch for forced S = 2, 5, the library's own choice, with bias / ragged M / slab output / accumulate and
    the conv-epilogue statistics; the ticket counters are back at zero after every launch."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib.get()
    N, D, H, W, Cin, Cout, K, p = cs["N"], cs["D"], cs["H"], cs["W"], cs["Cin"], cs["Cout"], cs["K"], cs["p"]
    up = cs.get("up", (0, 0, 0))
    x = rnd((N, D, H, W, Cin), 1, 1.0, dtype)
    w = rnd((Cout,) + K + (Cin,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cin), dtype)
    bias = rnd((Cout,), 6, 0.5).float().double() if cs["bias"] else None
    xa = mkact(ops, x, dtype)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    ref = ref_conv(ref_xeff(x, up, None, None, False, dtype), w, (1, 1, 1), p, bias)
    D, H, W = D << up[0], H << up[1], W << up[2]
    slots = 4
    shift = dev(ops, rnd((Cout,), 11, 0.3).float().double())
    outs, stats = {}, {}
    ws, cnt = ops.splitk_scratch()
    for S in (1, 2, 5, 0):
        lib.hdu_set_tuning(13, S)
        try:
            if cs["ldout"]:
                big = ops.Act.alloc(N, D, H, W, cs["ldout"], dtype, zero=True)
                ya = big.slab(8, Cout)
            else:
                ya = ops.Act.alloc(N, D, H, W, Cout, dtype)
            bias_d = dev(ops, bias) if bias is not None else None       # must outlive the launches
            d = ops.conv_desc(xa, ctypes.c_void_p(wt.data_ptr()), ya, K, (1, 1, 1), p, up, None, None, True, bias_d)
            if S == 0:
                assert ops.conv_splitk_ws_bytes(d) > 0, "the library's default should split this shape"
            elif S == 1:
                assert ops.conv_splitk_ws_bytes(d) == 0
            partial = torch.zeros(slots * 2 * Cout, dtype=torch.float32, device=ops.device())
            d.stats_partial, d.stats_shift, d.stats_slots = partial.data_ptr(), shift.data_ptr(), slots
            ops.conv_fprop(d)
            outs[S] = ya.to_torch().cpu()
            stats[S] = partial.cpu().reshape(slots, 2, Cout).sum(0)
            assert int(cnt.abs().sum()) == 0, "split-K tickets must return to zero"
            assert_close(outs[S], ref, dtype, what="fprop S=%d" % S)
            d.stats_partial = None
            d.accumulate = 1
            ops.conv_fprop(d)
            assert_close(ya.to_torch().cpu(), q(ref, dtype) * 2, dtype, scale=2 * float(ref.abs().max()), what="accumulate S=%d" % S)
        finally:
            lib.hdu_set_tuning(13, 0)
    for S in (2, 5, 0):
        # same products, different summation order: float32 roundoff only (one storage ulp in bf16)
        tol = 2e-5 if dtype == F32 else 8e-3
        assert float((outs[S] - outs[1]).abs().max()) <= tol * float(ref.abs().max()), S
        assert float((stats[S] - stats[1]).abs().max()) <= 1e-2 * float(stats[1].abs().max()) + 1e-3, S


@pytest.mark.parametrize("dtype", DT)
def test_conv_dgrad_strided(hdu, dtype):
    ops = ops_mod()
    import ctypes
    N, D, H, W, Cin, Cout, K, s, p = 1, 8, 10, 10, 8, 16, (7, 7, 7), (2, 2, 2), (3, 3, 3)
    w = rnd((Cout,) + K + (Cin,), 5, 0.05, dtype)
    Do, Ho, Wo = [(n + 6 - 7) // 2 + 1 for n in (D, H, W)]
    dy = rnd((N, Do, Ho, Wo, Cout), 9, 1.0, dtype)
    dya = mkact(ops, dy, dtype)
    dx = ops.Act.alloc(N, D, H, W, Cin, dtype)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    d = ops.conv_desc(dx, ctypes.c_void_p(wt.data_ptr()), dya, K, s, p)
    ops.conv_dgrad_strided(d)
    xe = torch.zeros((N, D, H, W, Cin), dtype=torch.float64, requires_grad=True)
    (ref_conv(xe, w, s, p, None) * dy).sum().backward()
    assert_close(dx.to_torch().cpu(), xe.grad, dtype, what="dgrad strided")


@pytest.mark.parametrize("dtype", DT)
def test_dropout_epilogue_consistent(hdu, dtype):
    """conv epilogue dropout mask == mask regenerated by bn_bwd_apply; keep fraction plausible"""
    ops = ops_mod()
    import ctypes
    N, D, H, W, Cin, Cout = 1, 1, 24, 24, 16, 32
    x = rnd((N, D, H, W, Cin), 1, 1.0, dtype)
    w = rnd((Cout, 1, 1, 1, Cin), 2, 0.3, dtype)
    xa = mkact(ops, x, dtype)
    wt = w.to(torch.bfloat16 if dtype == BF16 else torch.float32).contiguous().to(ops.device())
    y0 = ops.Act.alloc(N, D, H, W, Cout, dtype)
    y1 = ops.Act.alloc(N, D, H, W, Cout, dtype)
    ops.conv_fprop(ops.conv_desc(xa, ctypes.c_void_p(wt.data_ptr()), y0, (1, 1, 1)))
    ops.conv_fprop(ops.conv_desc(xa, ctypes.c_void_p(wt.data_ptr()), y1, (1, 1, 1), drop_keep=0.7, drop_seed=123))
    a, b_ = y0.to_torch().cpu().double(), y1.to_torch().cpu().double()
    mask = b_ != 0
    frac = float(mask.double().mean())
    assert 0.66 < frac < 0.74, frac
    assert_close(b_[mask], q(a[mask] / 0.7, dtype), dtype, what="dropout scale")
    # bn_bwd_apply with k1=1,k2=k3=0, a=1,b=0, no relu regenerates the same mask
    C = Cout
    one = torch.ones(C, device=ops.device()); zero = torch.zeros(C, device=ops.device())
    dz = mkact(ops, torch.ones((N, D, H, W, C), dtype=torch.float64), dtype)
    dx = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.bn_bwd_apply(dz, y0, one, zero, False, zero, one, zero, zero, dx, False, 0.7, 123)
    m2 = dx.to_torch().cpu() != 0
    assert bool((m2 == mask).all()) or float((m2 != mask).double().mean()) < 1e-3  # (y0==0 exactly is measure-zero)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 1, 13, 11, 48, 72, 8), (1, 3, 7, 7, 8, None, 0), (1, 1, 33, 37, 264, None, 0), (1, 1, 5, 5, 96, 128, 16),
                                   (1, 1, 96, 97, 16, None, 0)])   # > 8192 pixels: separate finalize kernel
def test_bn_stats_fold(hdu, dtype, shape):
    ops = ops_mod()
    N, D, H, W, C, ld, coff = shape
    x = rnd((N, D, H, W, C), 3, 2.0, dtype) + q(rnd((C,), 4, 30.0), dtype)  # large per-channel offsets
    x = q(x, dtype)
    xa = mkact(ops, x, dtype, ld, coff)
    mean = torch.empty(C, device=ops.device()); var = torch.empty(C, device=ops.device())
    ws = ops.Workspace(ops.reduce_ws_bytes(xa.M, C))
    ops.bn_stats(xa, mean, var, ws)
    xf = x.reshape(-1, C)
    mref = xf.mean(0)
    vref = ((xf - mref) ** 2).mean(0)   # tf.nn.moments: biased
    assert float((mean.cpu().double() - mref).abs().max()) < 1e-4 * (1 + float(mref.abs().max()))
    assert float(((var.cpu().double() - vref) / vref).abs().max()) < 2e-4
    # fold with Scale + moving stats
    g = rnd((C,), 5, 0.5) + 1.0; be = rnd((C,), 6, 0.2); sg = rnd((C,), 7, 0.5) + 1.0; sb = rnd((C,), 8, 0.2)
    mm = rnd((C,), 9, 1.0).float(); mv = (rnd((C,), 10, 0.4) + 1.0).float()
    a = torch.empty(C, device=ops.device()); b = torch.empty(C, device=ops.device()); r = torch.empty(C, device=ops.device())
    mmd, mvd = mm.clone().to(ops.device()), mv.clone().to(ops.device())
    eps = 1.1e-5
    ops.bn_fold(C, mean, var, dev(ops, g), dev(ops, be), eps, dev(ops, sg), dev(ops, sb), a, b, r, mmd, mvd, 0.99)
    rr = 1 / torch.sqrt(vref + eps)
    aref = sg * g * rr
    bref = sg * (be - mref * g * rr) + sb
    assert float(((a.cpu().double() - aref) / aref).abs().max()) < 2e-4
    assert float((b.cpu().double() - bref).abs().max()) < 2e-3 * (1 + float(bref.abs().max()))
    assert float((mmd.cpu().double() - (mm.double() - (mm.double() - mref) * 0.01)).abs().max()) < 1e-4
    assert float((mvd.cpu().double() - (mv.double() - (mv.double() - vref) * 0.01)).abs().max()) < 1e-4
    # normalised output has mean ~ beta', std ~ gamma' (the reference suite's own BN check, normalization_test.py:35-49)
    z = ops.Act.alloc(N, D, H, W, C, dtype)
    one = torch.ones(C, device=ops.device()); zero = torch.zeros(C, device=ops.device())
    ops.bn_fold(C, mean, var, one, zero, eps, None, None, a, b, r)
    ops.affine_act(xa, a, b, False, z)
    zt = z.to_torch().cpu().double().reshape(-1, C)
    assert float(zt.mean(0).abs().max()) < 1e-1 and float((zt.std(0) - 1).abs().max()) < 1e-1


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("batch_stats", [True, False])
def test_bn_backward(hdu, dtype, batch_stats):
    """BN(+Scale)+ReLU backward: reduce + coef + apply vs autograd of the float64 restatement"""
    ops = ops_mod()
    N, D, H, W, C = 2, 1, 9, 7, 24
    eps = 1.1e-5
    x = q(rnd((N, D, H, W, C), 3, 2.0, dtype) + 0.5, dtype)
    dz = rnd((N, D, H, W, C), 4, 1.0, dtype)
    g = (rnd((C,), 5, 0.5) + 1.0).float().double(); be = rnd((C,), 6, 0.2).float().double()
    sg = (rnd((C,), 7, 0.5) + 1.0).float().double(); sb = rnd((C,), 8, 0.2).float().double()
    mm = rnd((C,), 9, 0.5).float().double(); mv = (rnd((C,), 10, 0.4) + 1.0).float().double()
    M = N * D * H * W
    xa, dza = mkact(ops, x, dtype), mkact(ops, dz, dtype)
    ws = ops.Workspace(ops.reduce_ws_bytes(M, C))
    dv = lambda t: dev(ops, t)
    E = lambda: torch.empty(C, device=ops.device())
    mean, var, a, b, r, s1, s2, k1, k2, k3, dg, db, dsg, dsb = [E() for _ in range(14)]
    if batch_stats:
        ops.bn_stats(xa, mean, var, ws)
    else:
        mean.copy_(dv(mm)); var.copy_(dv(mv))
    ops.bn_fold(C, mean, var, dv(g), dv(be), eps, dv(sg), dv(sb), a, b, r)
    ops.bn_bwd_reduce(dza, xa, a, b, True, mean, r, s1, s2, ws)
    ops.bn_bwd_coef(C, M, batch_stats, s1, s2, dv(g), dv(be), dv(sg), r, k1, k2, k3, dg, db, dsg, dsb)
    dx = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.bn_bwd_apply(dza, xa, a, b, True, mean, k1, k2, k3, dx)
    # reference
    xr = x.clone().requires_grad_(True)
    gr, ber, sgr, sbr = [t.clone().requires_grad_(True) for t in (g, be, sg, sb)]
    xf = xr.reshape(-1, C)
    if batch_stats:
        mu = xf.mean(0); v = ((xf - mu) ** 2).mean(0)
    else:
        mu, v = mm, mv
    y = (xr - mu) / torch.sqrt(v + eps) * gr + ber
    z = (sgr * y + sbr).clamp_min(0)
    (z * dz).sum().backward()
    assert_close(dx.to_torch().cpu(), xr.grad, dtype, what="bn dx")
    for got, ref, nm in ((dg, gr.grad, "dgamma"), (db, ber.grad, "dbeta"), (dsg, sgr.grad, "dsgamma"), (dsb, sbr.grad, "dsbeta")):
        assert_close(got.cpu(), ref, dtype, what=nm)
    # accumulate
    ops.bn_bwd_apply(dza, xa, a, b, True, mean, k1, k2, k3, dx, True)
    assert_close(dx.to_torch().cpu(), 2 * xr.grad, dtype, what="bn dx accumulate")
    # the two-launch form (hdu_bn_bwd_fused): slot-table reduction + coefficients / parameter gradients / dx in the apply
    for slots in (16, 3):
        sums = torch.zeros(slots * 2 * C, device=ops.device())
        dg2, db2, dsg2, dsb2 = [E() for _ in range(4)]
        dx2 = ops.Act.alloc(N, D, H, W, C, dtype)
        ops.bn_bwd_fused(dza, xa, a, b, True, mean, r, batch_stats, dv(g), dv(be), dv(sg), sums, slots, dg2, db2, dsg2, dsb2, dx2)
        assert_close(dx2.to_torch().cpu(), xr.grad, dtype, what="fused bn dx")
        for got, ref, nm in ((dg2, gr.grad, "dgamma"), (db2, ber.grad, "dbeta"), (dsg2, sgr.grad, "dsgamma"), (dsb2, sbr.grad, "dsbeta")):
            assert_close(got.cpu(), ref, dtype, what="fused " + nm)
        sums.zero_()
        ops.bn_bwd_fused(dza, xa, a, b, True, mean, r, batch_stats, dv(g), dv(be), dv(sg), sums, slots, None, None, None, None,
                         dx2, True)
        assert_close(dx2.to_torch().cpu(), 2 * xr.grad, dtype, what="fused bn dx accumulate")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("form", ["rows-first", "rows-loop"])
def test_bn_backward_fused_wide(hdu, dtype, form):
    """hdu_bn_bwd_fused over several column groups and many row blocks (C = 328 channels: 2 groups of 32 chunks in bf16, a
    partial last group; M = 4096 rows) against the three-launch form of the same library.  Round 6: at this size the apply launch
    requests its rows before the slot-table round trip (bn_bwd_apply_kernel PRE); "rows-loop" = that form switched off."""
    ops = ops_mod()
    if form == "rows-loop":
        hdu.lib.get().hdu_set_tuning(4, 2048)
    try:
        _bn_backward_fused_wide_case(hdu, ops, dtype)
    finally:
        hdu.lib.get().hdu_set_tuning(4, 0)


def _bn_backward_fused_wide_case(hdu, ops, dtype):
    N, D, H, W, C = 1, 1, 64, 64, 328
    x = q(rnd((N, D, H, W, C), 13, 2.0, dtype) + 0.25, dtype)
    dz = rnd((N, D, H, W, C), 14, 1.0, dtype)
    M = N * D * H * W
    xa, dza = mkact(ops, x, dtype), mkact(ops, dz, dtype)
    ws = ops.Workspace(ops.reduce_ws_bytes(M, C))
    dv = lambda t: dev(ops, t)
    E = lambda: torch.empty(C, device=ops.device())
    g = dv((rnd((C,), 5, 0.5) + 1.0).float().double()); be = dv(rnd((C,), 6, 0.2).float().double())
    mean, var, a, b, r, s1, s2, k1, k2, k3, dg, db, dg2, db2 = [E() for _ in range(14)]
    ops.bn_stats(xa, mean, var, ws)
    ops.bn_fold(C, mean, var, g, be, 1.1e-5, None, None, a, b, r)
    ops.bn_bwd_reduce(dza, xa, a, b, True, mean, r, s1, s2, ws)
    ops.bn_bwd_coef(C, M, True, s1, s2, g, be, None, r, k1, k2, k3, dg, db, None, None)
    dx = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.bn_bwd_apply(dza, xa, a, b, True, mean, k1, k2, k3, dx)
    sums = torch.zeros(16 * 2 * C, device=ops.device())
    dx2 = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.bn_bwd_fused(dza, xa, a, b, True, mean, r, True, g, be, None, sums, 16, dg2, db2, None, None, dx2)
    ref = dx.to_torch().cpu().double()
    assert_close(dx2.to_torch().cpu(), ref, dtype, what="fused wide dx")
    assert_close(dg2.cpu(), dg.cpu().double(), F32, scale=float(dg.abs().max()), what="fused wide dgamma")
    assert_close(db2.cpu(), db.cpu().double(), F32, scale=float(db.abs().max()), what="fused wide dbeta")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dims", [(2, 1, 10, 12, 16), (1, 6, 8, 10, 8), (1, 5, 7, 9, 8)])
def test_maxpool(hdu, dtype, dims):
    ops = ops_mod()
    N, D, H, W, C = dims
    x = rnd((N, D, H, W, C), 3, 1.0, dtype).clamp_min(0)   # post-ReLU input as in the model
    x = x + (x > 0) * q(rnd((N, D, H, W, C), 4, 0.01, dtype).abs(), dtype)
    x = q(x, dtype)
    xa = mkact(ops, x, dtype)
    Do = 1 if D == 1 else (D - 1) // 2 + 1
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = ops.Act.alloc(N, Do, Ho, Wo, C, dtype)
    amax = torch.zeros(N * Do * Ho * Wo * C, dtype=torch.uint8, device=ops.device())
    ops.maxpool_fwd(xa, y, amax)
    xr = x.clone().requires_grad_(True)
    xp = xr.permute(0, 4, 1, 2, 3)
    if D == 1:
        yr = F.max_pool2d(F.pad(xp[:, :, 0], (1, 1, 1, 1)), 3, 2)[:, :, None]
    else:
        yr = F.max_pool3d(F.pad(xp, (1, 1, 1, 1, 1, 1)), 3, 2)
    yr = yr.permute(0, 2, 3, 4, 1)
    assert tuple(yr.shape) == (N, Do, Ho, Wo, C)
    assert float((y.to_torch().cpu().double() - yr).abs().max()) == 0.0
    dy = rnd((N, Do, Ho, Wo, C), 5, 1.0, dtype)
    (yr * dy).sum().backward()
    dx = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.maxpool_bwd(amax, mkact(ops, dy, dtype), dx)
    got = dx.to_torch().cpu().double()
    # ties only happen at exactly 0 (post-ReLU) where the upstream ReLU kills the gradient: compare where x>0
    nz = x > 0
    assert_close(got[nz], xr.grad[nz], dtype, scale=float(xr.grad.abs().max()), what="maxpool bwd")


@pytest.mark.parametrize("dtype", DT)
def test_maxpool_depth_halo_mode(hdu, dtype):
    """pad_d=0: the caller supplies the neighbouring depth planes (depth sharding); equals the padded pool on the
    interior of a volume whose first/last planes play the halo role"""
    ops = ops_mod()
    N, D, H, W, C = 1, 8, 6, 6, 8
    x = q(rnd((N, D, H, W, C), 3, 1.0, dtype).clamp_min(0) + 0.01, dtype)
    full = ops.Act.alloc(N, (D - 1) // 2 + 1, 3, 3, C, dtype)
    ops.maxpool_fwd(mkact(ops, x, dtype), full)                       # reference pooling of the whole volume
    # shard = planes 3..6 with halo plane 3 (low) and plane 8(absent -> zero)... take interior planes 4..7, halo 3 and zeros
    sh = torch.zeros((N, 6, H, W, C), dtype=torch.float64)
    sh[:, 0] = x[:, 3]
    sh[:, 1:5] = x[:, 4:8]
    out = ops.Act.alloc(N, 2, 3, 3, C, dtype)
    am = torch.zeros(2 * 9 * C, dtype=torch.uint8, device=ops.device())
    ops.maxpool_fwd(mkact(ops, sh, dtype), out, am, pad_d=0)
    assert float((out.to_torch().cpu() - full.to_torch().cpu()[:, 2:4]).abs().max()) == 0.0
    dy = rnd((N, 2, 3, 3, C), 5, 1.0, dtype)
    dx = ops.Act.alloc(N, 6, H, W, C, dtype)
    ops.maxpool_bwd(am, mkact(ops, dy, dtype), dx, pad_d=0)
    xr = sh.clone().requires_grad_(True)
    yr = F.max_pool3d(F.pad(xr.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 0, 0)), 3, 2).permute(0, 2, 3, 4, 1)
    (yr * dy).sum().backward()
    nz = sh > 0
    assert_close(dx.to_torch().cpu().double()[nz], xr.grad[nz], dtype, scale=float(xr.grad.abs().max()), what="maxpool halo bwd")


@pytest.mark.parametrize("dtype", DT)
def test_avgpool_upsample(hdu, dtype):
    ops = ops_mod()
    N, D, H, W, C = 1, 3, 8, 6, 16
    x = rnd((N, D, H, W, C), 3, 1.0, dtype)
    xa = mkact(ops, x, dtype)
    y = ops.Act.alloc(N, D, H // 2, W // 2, C, dtype)
    ops.avgpool_fwd(xa, y)
    ref = x.reshape(N, D, H // 2, 2, W // 2, 2, C).mean((3, 5))
    assert_close(y.to_torch().cpu(), ref, dtype, what="avgpool")
    dy = rnd((N, D, H // 2, W // 2, C), 4, 1.0, dtype)
    dx = ops.Act.alloc(N, D, H, W, C, dtype)
    ops.avgpool_bwd(mkact(ops, dy, dtype), dx)
    refdx = dy.repeat_interleave(2, 2).repeat_interleave(2, 3) * 0.25
    assert_close(dx.to_torch().cpu(), refdx, dtype, what="avgpool bwd")
    # nearest up-sampling gradient == sum over children; known answer from the reference suite: UpSampling == np.repeat
    for up in [(0, 1, 1), (1, 1, 1)]:
        g = rnd((N, D << up[0], H << up[1], W << up[2], C), 5, 1.0, dtype)
        dz = ops.Act.alloc(N, D, H, W, C, dtype)
        ops.upsample_bwd(mkact(ops, g, dtype), dz, up)
        r = g.reshape(N, D, 1 << up[0], H, 1 << up[1], W, 1 << up[2], C).sum((2, 4, 6))
        assert_close(dz.to_torch().cpu(), r, dtype, what="upsample bwd")


@pytest.mark.parametrize("dtype", DT)
def test_wce_loss(hdu, dtype):
    ops = ops_mod()
    M = 5000
    z = rnd((1, 1, 1, M, 3), 3, 6.0, dtype)
    z[0, 0, 0, :20, 0] = 60.0  # saturate: p(class 1/2) < 1e-10 -> clipped, zero gradient
    z = q(z, dtype)
    g = torch.Generator().manual_seed(5)
    lab = torch.randint(0, 3, (M,), generator=g)
    lab[:20] = torch.tensor([1, 2] * 10)
    Cp = 8 if dtype == BF16 else 4
    zz = torch.zeros((1, 1, 1, M, Cp), dtype=torch.float64); zz[..., :3] = z
    la = mkact(ops, zz, dtype)
    dl = ops.Act.alloc(1, 1, 1, M, Cp, dtype)
    dl.buf.fill_(5.0)
    loss = torch.zeros(1, device=ops.device()); cnt = torch.zeros(3, device=ops.device())
    ws = ops.Workspace(1 << 16)
    w = (0.78, 0.65, 8.57)
    ops.wce_loss(la, lab.to(torch.uint8).to(ops.device()), 0, M, w, 1.0 / M, dl, loss, cnt, ws)
    zr = z.reshape(M, 3).clone().requires_grad_(True)
    p = torch.softmax(zr, 1)
    lp = torch.log(torch.clamp(p, 1e-10, 1.0))
    wt = torch.tensor(w, dtype=torch.float64)[lab]
    L = -(wt * lp[torch.arange(M), lab]).sum() / M
    L.backward()
    assert abs(float(loss.cpu()) / M - float(L)) < 2e-5 * abs(float(L)) + 1e-6
    assert cnt.cpu().tolist() == [float((lab == i).sum()) for i in range(3)]
    got = dl.to_torch().cpu().double().reshape(M, Cp)
    assert float(got[:, 3:].abs().max()) == 0.0
    rt, at = tol(dtype)
    assert float((got[:, :3] - zr.grad).abs().max()) < at * float(zr.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("up,use_skip,use_pro", [((0, 0, 0), False, True), ((0, 1, 1), True, True), ((1, 1, 1), False, True),
                                                  ((0, 0, 0), True, False), ((0, 1, 1), True, True)])
def test_materialize(hdu, dtype, up, use_skip, use_pro):
    """relu(a*x+b) -> nearest up-sampling (== np.repeat, convolutional_test.py:673-681,726-736) -> + skip"""
    ops = ops_mod()
    N, D, H, W, C = 2, 2, 5, 7, 24
    x = rnd((N, D, H, W, C), 1, 1.0, dtype)
    De, He, We = D << up[0], H << up[1], W << up[2]
    skip = rnd((N, De, He, We, C), 2, 1.0, dtype) if use_skip else None
    a = (rnd((C,), 3, 0.5) + 1.0).float().double(); b = rnd((C,), 4, 0.3).float().double()
    xa = mkact(ops, x, dtype, 40, 8)
    out = ops.Act.alloc(N, De, He, We, C, dtype)
    ops.materialize(xa, dev(ops, a) if use_pro else None, dev(ops, b) if use_pro else None, use_pro, up,
                    mkact(ops, skip, dtype) if use_skip else None, out)
    ref = x
    if use_pro:
        ref = (ref * a + b).clamp_min(0)
    for ax, u in zip((1, 2, 3), up):
        if u:
            ref = ref.repeat_interleave(2, dim=ax)
    if use_skip:
        ref = ref + skip
    assert_close(out.to_torch().cpu(), ref, dtype, what="materialize")


@pytest.mark.parametrize("dtype", DT)
def test_plumbing(hdu, dtype):
    ops = ops_mod()
    D, H, W = 5, 4, 6
    Cp = 8 if dtype == BF16 else 4
    vol = rnd((D, H, W), 1, 100.0).float()
    out = ops.Act.alloc(D, 1, H, W, Cp, dtype)
    ops.slab25d(vol.to(ops.device()), D, H, W, out)
    got = out.to_torch().cpu().reshape(D, H, W, Cp)
    for k in range(D):  # denseunet3d.py:399-409: slab k = slices (k-1,k,k+1), edges replicated
        for j, kk in enumerate((max(k - 1, 0), k, min(k + 1, D - 1))):
            assert_close(got[k, :, :, j], q(vol[kk].double(), dtype), dtype, what="slab25d")
    assert float(got[..., 3:].abs().max()) == 0.0
    lg = rnd((1, D, H, W, Cp), 2, 3.0, dtype); lg[..., 3:] = 0
    lga = mkact(ops, lg, dtype)
    i3 = ops.Act.alloc(1, D, H, W, Cp, dtype)
    ops.make_input3d(vol.to(ops.device()), lga, 250.0, i3)
    g3 = i3.to_torch().cpu().double()
    assert_close(g3[0, ..., 0], q(vol.double(), dtype), dtype, what="input3d ct")
    assert_close(g3[0, ..., 1:4], q(lg[0, ..., :3] * 250, dtype), dtype, what="input3d logits")
    din = rnd((1, D, H, W, Cp), 3, 1.0, dtype)
    dl = ops.Act.alloc(1, D, H, W, Cp, dtype)
    ops.make_input3d_bwd(mkact(ops, din, dtype), 250.0, dl)
    assert_close(dl.to_torch().cpu()[..., :3], din[..., 1:4] * 250, dtype, what="input3d bwd")
    # cast helpers
    src = rnd((7, 3), 4, 5.0).float()
    a = ops.Act.alloc(1, 1, 1, 7, Cp, dtype)
    ops.cast_pad(src.to(ops.device()), 7, 3, a)
    back = torch.empty(7, 3, device=ops.device())
    ops.cast_out(a, 3, back)
    assert_close(back.cpu(), q(src.double(), dtype), dtype, what="cast round trip")


def test_abi_errors(hdu):
    """bad arguments come back as error codes + message, never a crash (include/hdu.h conventions)"""
    ops = ops_mod()
    import ctypes
    x = ops.Act.alloc(1, 1, 4, 4, 8, BF16)
    y = ops.Act.alloc(1, 1, 5, 5, 8, BF16)
    w = torch.zeros(8 * 9 * 8, dtype=torch.bfloat16, device=ops.device())
    # pad 0: 2x2 outputs; up to K-1 = 2 extra positions per axis are legal (implicit high-side padding), 5x5 is not
    d = ops.conv_desc(x, ctypes.c_void_p(w.data_ptr()), y, (1, 3, 3), (1, 1, 1), (0, 0, 0))
    with pytest.raises(hdu.lib.HduError, match="output dims"):
        ops.conv_fprop(d)
    bad = ops.Act(x.buf, 0, 1, 1, 4, 4, 6, 6, BF16)
    with pytest.raises(hdu.lib.HduError):
        ops.affine_act(bad, None, None, True, bad)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [dict(N=1, D=8, H=10, W=12, Cin=8, Cout=16, K=(7, 7, 7), s=(2, 2, 2), p=(3, 3, 3), id="stem7x7x7s2"),
                                   dict(N=2, D=1, H=12, W=8, Cin=8, Cout=24, K=(1, 7, 7), s=(1, 2, 2), p=(0, 3, 3), id="stem7x7s2_2d"),
                                   dict(N=1, D=4, H=6, W=6, Cin=16, Cout=8, K=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), id="k3s2")],
                         ids=lambda c: c["id"])
def test_stride2_dgrad_parity_classes(hdu, dtype, shape):
    """hdu_stride2_dgrad_filters + hdu_conv_fprop per parity class + hdu_parity_interleave == the data gradient of the
    strided convolution (autograd, float64), plain and accumulating -- and == the scalar hdu_conv_dgrad_strided form"""
    ops = ops_mod()
    N, D, H, W, Cin, Cout, K, s, p = (shape[k] for k in ("N", "D", "H", "W", "Cin", "Cout", "K", "s", "p"))
    w = rnd((Cout,) + K + (Cin,), 5, 1.0 / np.sqrt(K[0] * K[1] * K[2] * Cout), dtype)
    Do, Ho, Wo = [(n + 2 * pp - k) // ss + 1 for n, pp, k, ss in zip((D, H, W), p, K, s)]
    dy = rnd((N, Do, Ho, Wo, Cout), 9, 1.0, dtype)
    dya = mkact(ops, dy, dtype)
    wm = dev(ops, q(w, dtype)).reshape(-1)
    xe = torch.zeros((N, D, H, W, Cin), dtype=torch.float64, requires_grad=True)
    (ref_conv(xe, q(w, dtype), s, p, None) * dy).sum().backward()
    dx = ops.Act.alloc(N, D, H, W, Cin, dtype, zero=True)
    s2 = ops.Stride2Dgrad(dtype, wm, dya, (N, D, H, W), Cin, K, s, p)
    s2.run(dx)
    assert_close(dx.to_torch().cpu(), xe.grad, dtype, what="stride-2 dgrad")
    s2.run(dx, accumulate=True)
    assert_close(dx.to_torch().cpu(), 2 * xe.grad, dtype, scale=2 * float(xe.grad.abs().max()), what="stride-2 dgrad accumulate")


def test_weight_prep_batched_equals_per_layer(hdu):
    """hdu_weight_prep_batched (one launch, 64 x 64 tiles, 16-byte reads / 8-byte bf16 writes, a workgroup walks consecutive tiles across
    layer boundaries: round 6; one layer with a channel count that is not a multiple of 4 on the element-wise path) ==
    hdu_weight_prep per layer, bit for bit, for both copies and both dtypes; layers with ragged channel counts, one tile, many
    taps, forward-only and data-gradient-only entries; the bytes around every copy stay untouched."""
    import ctypes
    ops = ops_mod()
    lib = hdu.lib
    layers = [(48, 9, 192, True, True), (192, 1, 2208, True, True), (8, 1, 64, True, False), (96, 343, 8, True, True), (40, 27, 24, False, True),
              (32, 1, 32, True, True), (3 * 8, 9, 8, True, True), (136, 3, 72, True, True), (16, 1, 16, True, True), (64, 27, 96, True, True),
              (70, 3, 66, True, True), (200, 1, 130, True, True)]
    for dtype in (BF16, F32):
        tdt = torch.bfloat16 if dtype == BF16 else torch.float32
        n_master = sum(co * t * ci for co, t, ci, _, _ in layers)
        master = (torch.rand(n_master + 16, generator=torch.Generator().manual_seed(5 + dtype)) * 2 - 1).to(ops.device())
        ents, refs, tiles, moff, woff = [], [], 0, 0, 8
        for co, t, ci, want_f, want_d in layers:
            numel = co * t * ci
            wf = woff if want_f else -1
            woff += (numel + 8) if want_f else 0
            wd = woff if want_d else -1
            woff += (numel + 8) if want_d else 0
            ents.append(lib.PrepEntry(moff, wf, wd, tiles, co, t, ci, 0))
            tiles += t * ((co + 63) // 64) * ((ci + 63) // 64)
            refs.append((moff, wf, wd, co, t, ci, numel))
            moff += numel
        wc = torch.full((woff + 8,), 3.0, dtype=tdt, device=ops.device())
        arr = (lib.PrepEntry * len(ents))(*ents)
        table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(ops.device())
        ops.weight_prep_batched(dtype, table, len(ents), tiles, master, wc)
        covered = torch.zeros(woff + 8, dtype=torch.bool)
        for moff_, wf, wd, co, t, ci, numel in refs:
            rf = torch.zeros(numel, dtype=tdt, device=ops.device())
            rd = torch.zeros(numel, dtype=tdt, device=ops.device())
            ops.weight_prep(dtype, master[moff_:moff_ + numel], co, t, ci, rf, rd)
            if wf >= 0:
                assert torch.equal(wc[wf:wf + numel], rf), (dtype, co, t, ci, "forward copy")
                covered[wf:wf + numel] = True
            if wd >= 0:
                assert torch.equal(wc[wd:wd + numel], rd), (dtype, co, t, ci, "data-gradient copy")
                covered[wd:wd + numel] = True
        assert float((wc.cpu().float()[~covered] - 3.0).abs().max()) == 0.0


def test_zero_regions(hdu):
    """hdu_zero_regions / hdu_zero: the one-launch re-initialisation of a step's accumulators -- every region cleared
    exactly (multi-block regions, a 4-byte-granular tail, a region smaller than one block), neighbours untouched, the
    step counter advanced once"""
    ops = ops_mod()
    d = ops.device()
    sizes = [5, 4, 16384 + 7, 3 * 16384, 1]          # floats; 16384 floats = one 64 KiB block
    pad = 8
    bufs, views = [], []
    for i, n in enumerate(sizes):
        big = torch.full((n + 2 * pad + 3,), float(i + 1), dtype=torch.float32, device=d)
        bufs.append(big)
        off = pad + (-(big.data_ptr() // 4 + pad)) % 4       # 16-byte aligned start
        views.append(big[off:off + n])
    half = torch.full((64,), 3.0, dtype=torch.bfloat16, device=d)
    counter = torch.tensor([41], dtype=torch.int32, device=d)
    plan = ops.ZeroPlan(views + [half])
    plan.run(counter, 1)
    plan.run(counter, 1)
    assert int(counter.item()) == 43
    for i, (big, v) in enumerate(zip(bufs, views)):
        assert float(v.abs().max()) == 0.0
        rest = big.clone()
        off = v.data_ptr() // 4 - big.data_ptr() // 4
        rest[off:off + v.numel()] = float(i + 1)
        assert torch.equal(rest, torch.full_like(big, float(i + 1))), "neighbouring bytes of region %d were touched" % i
    assert float(half.float().abs().max()) == 0.0
    t = torch.full((16384 * 2 + 12,), 2.0, dtype=torch.float32, device=d)
    ops.zero_tensor(t[4:4 + 16384 * 2 + 4])
    assert float(t[4:-4].abs().max()) == 0.0 and float(t[:4].min()) == 2.0 and float(t[-4:].min()) == 2.0
    with pytest.raises(Exception):
        ops.zero_tensor(t[1:9])                        # not 16-byte aligned


def test_stats_sync_combine(hdu):
    """hdu_stats_pack / hdu_stats_unpack (sync-BN of a depth-sharded volume): the statistics of the whole tensor from the
    per-shard (n, mean, var) slots -- including channels whose |mean| is 1e3 standard deviations, where the
    (n mean, n (var + mean^2)) form all-reduced in rounds 1-2 lost the variance entirely"""
    ops = ops_mod()
    lib = hdu.lib.get()
    d = ops.device()
    C, world = 40, 3
    ns = [700, 700, 1100]
    rng = np.random.default_rng(3)
    offs = rng.normal(0, 1, C) * np.where(np.arange(C) % 2 == 0, 1.0, 1e3)          # odd channels: mean >> sigma
    xs = [(rng.normal(0, 1, (n, C)) * 0.7 + offs + 0.05 * r).astype(np.float64) for r, n in enumerate(ns)]
    allx = np.concatenate(xs, 0)
    n_fl = int(lib.hdu_stats_sync_floats(C, world))
    assert n_fl == world * (1 + 2 * C)
    total = torch.zeros(n_fl, dtype=torch.float32, device=d)
    for r in range(world):
        mean = torch.tensor(xs[r].mean(0), dtype=torch.float32, device=d)
        var = torch.tensor(xs[r].var(0), dtype=torch.float32, device=d)
        buf = torch.full((n_fl,), 7.0, dtype=torch.float32, device=d)
        assert lib.hdu_stats_pack(C, ops.fptr(mean), ops.fptr(var), ns[r], r, world, ops.fptr(buf), ops.stream()) == 0
        total += buf                                  # the caller's SUM all-reduce
    mean = torch.empty(C, dtype=torch.float32, device=d)
    var = torch.empty(C, dtype=torch.float32, device=d)
    assert lib.hdu_stats_unpack(C, ops.fptr(total), world, ops.fptr(mean), ops.fptr(var), ops.stream()) == 0
    np.testing.assert_allclose(mean.cpu().numpy(), allx.mean(0), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(var.cpu().numpy(), allx.var(0), rtol=2e-5)
    assert lib.hdu_stats_pack(C, ops.fptr(mean), ops.fptr(var), 5, 3, 3, ops.fptr(total), ops.stream()) != 0      # rank out of range


def test_bn_bwd_finalize_batched(hdu):
    """hdu_bn_bwd_finalize_batched (one launch for many inference-mode BNs) == hdu_bn_bwd_finalize per layer"""
    ops = ops_mod()
    dev_ = ops.device()
    g = torch.Generator().manual_seed(5)
    rn = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float32).to(dev_)
    entries, refs = [], []
    for C, slots, has_scale, tr_bn in ((48, 32, True, False), (200, 32, True, True), (8, 5, False, True), (1208, 32, True, False)):
        part = (rn(slots * 2 * C) * 3).contiguous()
        gamma, beta, sg = rn(C) * 0.2 + 1.0, rn(C), (rn(C) * 0.2 + 1.0) if has_scale else None
        mean, rstd = rn(C), rn(C).abs() + 0.5
        outs = [torch.full((C,), 7.0, device=dev_) for _ in range(4)]
        want = [tr_bn, tr_bn, has_scale, has_scale]
        o = [t if w else None for t, w in zip(outs, want)]
        entries.append((part, slots, C, gamma, beta, sg, o[0], o[1], o[2], o[3]))
        r = [torch.full((C,), 7.0, device=dev_) for _ in range(4)]
        ro = [t if w else None for t, w in zip(r, want)]
        ops.bn_bwd_finalize(part, slots, 1000, C, False, gamma, beta, sg, mean, rstd, ro[0], ro[1], ro[2], ro[3], None, None)
        refs.append(r)
        entries[-1] = entries[-1] + ()
    plan = ops.BnBwdPlan(entries)
    plan.run()
    for e, r in zip(entries, refs):
        got = [t for t in e[6:10]]
        for i, (a, b) in enumerate(zip(got, r)):
            if a is None:
                assert float((b - 7.0).abs().max()) == 0.0       # the per-layer call left an unwanted output alone too
            else:
                assert torch.equal(a.cpu(), b.cpu()), i


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("num", [3, 2])
def test_softmax_accumulate(hdu, dtype, num):
    """hdu_softmax_accumulate (lib/funcs.py:31-34): score += softmax(logits)[:, :num], logits padded to the 16-byte chunk and
    taken from a row offset inside the tensor"""
    ops = ops_mod()
    M, Cp = 1000, 8 if dtype == BF16 else 4
    lg = q(rnd((1, 1, 1, M + 7, Cp), 5, 3.0, dtype), dtype)
    la = mkact(ops, lg, dtype)
    score0 = rnd((M, num), 6, 1.0).float()
    score = score0.clone().to(ops.device()).reshape(-1)
    ops.softmax_accumulate(la, 7, M, num, score)
    ref = score0.double() + torch.softmax(lg.reshape(-1, Cp)[7:, :3].double(), -1)[:, :num]
    assert float((score.cpu().double().reshape(M, num) - ref).abs().max()) < 2e-6
    with pytest.raises(hdu.lib.HduError):
        ops.softmax_accumulate(la, 0, M, 4, torch.zeros(M * 4, device=ops.device()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(1000, 304, 256, 48), (77, 128, 96, 32), (4100, 64, 0, 64)])
def test_bn_bwd_finalize_correct(hdu, dtype, geom):
    """hdu_bn_bwd_finalize_correct == hdu_bn_bwd_finalize (batch statistics) followed by hdu_bn_bwd_correct of the channels
    [cs0, cs0 + Cc): same parameter gradients, same corrected gradient, the accumulators of every OTHER channel updated, those
    of the corrected channels left alone"""
    ops = ops_mod()
    dev_ = ops.device()
    M, C, cs0, Cc = geom
    slots = 32
    g = torch.Generator().manual_seed(9)
    rn = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float32)
    partial = (rn(slots, 2, C) * 3).to(dev_).reshape(-1)
    gamma, beta, sg = (rn(C) * 0.2 + 1.0).to(dev_), rn(C).to(dev_), (rn(C) * 0.2 + 1.0).to(dev_)
    mean, rstd = (rn(C) * 0.3).to(dev_), (rn(C).abs() + 0.5).to(dev_)
    u = q(rnd((1, 1, 1, M, C + 8), 3, 1.0, dtype), dtype)
    du0 = q(rnd((1, 1, 1, M, C + 8), 4, 1.0, dtype), dtype)
    outs = []
    c3_init, c4_init = rn(C).abs() * 0.1, rn(C) * 0.1
    for merged in (True, False):
        ua, dua = mkact(ops, u, dtype), mkact(ops, du0, dtype)
        z = lambda: torch.zeros(C, dtype=torch.float32, device=dev_)
        dg, db, dsg, dsb = z(), z(), z(), z()
        c3, c4 = c3_init.clone().to(dev_), c4_init.clone().to(dev_)
        c3_0, c4_0 = c3.clone(), c4.clone()
        us, dus = ua.slab(cs0, Cc), dua.slab(cs0, Cc)
        if merged:
            ops.bn_bwd_finalize_correct(partial, slots, M, C, gamma, beta, sg, mean, rstd, dg, db, dsg, dsb, c3, c4, cs0, us, dus)
        else:
            ops.bn_bwd_finalize(partial, slots, M, C, True, gamma, beta, sg, mean, rstd, dg, db, dsg, dsb, c3, c4)
            ops.bn_bwd_correct(us, c3[cs0:cs0 + Cc], c4[cs0:cs0 + Cc], dus)
        outs.append([t.cpu().double() for t in (dg, db, dsg, dsb, c3, c4)] + [dua.to_torch().cpu().double(), c3_0.cpu().double(), c4_0.cpu().double()])
    mg, sp = outs
    for i, nm in enumerate(["dgamma", "dbeta", "dsgamma", "dsbeta"]):
        assert float((mg[i] - sp[i]).abs().max()) <= 1e-5 * max(1.0, float(sp[i].abs().max())), nm
    keep = torch.ones(C, dtype=torch.bool)
    keep[cs0:cs0 + Cc] = False
    for i in (4, 5):
        if keep.any():
            assert float((mg[i][keep] - sp[i][keep]).abs().max()) <= 1e-5 * max(1.0, float(sp[i].abs().max()))
        assert torch.equal(mg[i][~keep], mg[7 if i == 4 else 8][~keep])            # corrected channels: accumulators untouched
    tol = 2e-2 if dtype == BF16 else 2e-5
    assert float((mg[6] - sp[6]).abs().max()) <= tol * max(1.0, float(sp[6].abs().max()))
    untouched = torch.cat([mg[6][..., :cs0] - du0[..., :cs0].double(), mg[6][..., cs0 + Cc:] - du0[..., cs0 + Cc:].double()], -1)
    assert float(untouched.abs().max()) == 0.0
    assert float((mg[6][..., cs0:cs0 + Cc] - du0[..., cs0:cs0 + Cc].double()).abs().max()) > 0.01


def test_halo_wide_choice_is_the_whole_layers(emu_lib):
    """A depth shard must run the kernel (family AND tile configuration) that the unsharded layer runs -- the summation order of every
    output element, hence the bit-equality of the sharded step, depends on it (hdu_conv_desc.layer_rows).  Host-side decision only: the
    descriptors describe the configs[4] decoder / dense-block layers at full size over dummy storage."""
    import ctypes
    ops = ops_mod()
    buf = torch.zeros(64, dtype=torch.bfloat16, device=ops.device())
    wt = torch.zeros(64, dtype=torch.bfloat16, device=ops.device())

    def name(D, H, W, Cin, Cout, up, pd, world):
        x = ops.Act(buf, 0, 1, D, H, W, Cin, Cin, BF16)
        De, He, We = D << up[0], H << up[1], W << up[2]
        y = ops.Act(buf, 0, 1, De + 2 * pd - 2, He, We, Cout, Cout, BF16)
        d = ops.conv_desc(x, ctypes.c_void_p(wt.data_ptr()), y, (3, 3, 3), (1, 1, 1), (pd, 1, 1), up, shard_world=world)
        return ops.conv_kernel_name(d, 0)

    seen = set()
    for (Dw, H, W, Cin, Cout, up) in [(32, 256, 256, 96, 64, (1, 1, 1)), (16, 128, 128, 192, 96, (1, 1, 1)), (16, 64, 64, 224, 192, (0, 1, 1)),
                                      (16, 32, 32, 504, 224, (0, 1, 1)), (16, 16, 16, 504, 504, (0, 1, 1)), (64, 512, 512, 64, 96, (0, 0, 0)),
                                      (16, 128, 128, 32, 128, (0, 0, 0)), (16, 128, 128, 128, 32, (0, 0, 0))]:
        whole = name(Dw, H, W, Cin, Cout, up, 1, 1)
        for world in (2, 4, 8):
            # a shard stores its planes + 2 halo planes (one per side, at the STORED resolution); depth padding 0 ("valid"), or -1 behind
            # a depth up-sampling (the up-sampled halo planes are cropped): engine.ConvLayer halo mode
            Dl = Dw // world
            if Dl < 1:
                continue
            shard = name(Dl + 2, H, W, Cin, Cout, up, 0 - up[0], world)
            assert shard == whole, (Dw, H, W, Cin, Cout, up, world, shard, whole)
        seen.add(whole)
    assert any(n.startswith("conv_halo_wide_kernel") for n in seen), seen
