"""ctypes binding of libhdu.so (the C-ABI declared in include/hdu.h).

The product library is `libhdu.so` next to this file, built for gfx950 by `build.sh hip` /
`__graft_entry__.build()`.  If it is missing, loading fails loudly -- there is no CPU fallback on the
product path, and `load()` refuses a library that does not identify itself as the gfx950 build.  (The x86
emulator build of the same kernel sources, which lets the CPU-only test tier execute kernel logic, is bound
by tests/emu_bind.py -- test infrastructure; nothing in this package knows it exists.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

HDU_BF16 = 0
HDU_F32 = 1

c_int, c_i64, c_f, c_p, c_u32, c_sz = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                       ctypes.c_uint32, ctypes.c_size_t)


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("x", c_p), ("ldx", c_i64),
        ("N", c_int), ("Di", c_int), ("Hi", c_int), ("Wi", c_int), ("Cin", c_int),
        ("ud", c_int), ("uh", c_int), ("uw", c_int),
        ("skip", c_p), ("ldskip", c_i64),
        ("pro_a", c_p), ("pro_b", c_p), ("pro_relu", c_int),
        ("w", c_p),
        ("KD", c_int), ("KH", c_int), ("KW", c_int),
        ("sd", c_int), ("sh", c_int), ("sw", c_int),
        ("pd", c_int), ("ph", c_int), ("pw", c_int),
        ("y", c_p), ("ldy", c_i64),
        ("Do", c_int), ("Ho", c_int), ("Wo", c_int), ("Cout", c_int),
        ("bias", c_p),
        ("epi_a", c_p), ("epi_b", c_p), ("epi_relu", c_int),
        ("accumulate", c_int),
        ("drop_keep", c_f),
        ("drop_seed", c_u32),
        ("drop_seed_dev", c_p),
        ("stats_partial", c_p),
        ("stats_shift", c_p),
        ("stats_slots", c_int),
        ("bnb_u", c_p), ("bnb_ldu", c_i64),
        ("bnb_a", c_p), ("bnb_b", c_p), ("bnb_mean", c_p), ("bnb_rstd", c_p),
        ("bnb_relu", c_int),
        ("bnb_partial", c_p), ("bnb_slots", c_int),
        ("splitk_ws", c_p),
        ("splitk_ws_bytes", c_sz),
        ("splitk_counters", c_p),
        ("layer_rows", ctypes.c_int64),
    ]


class PrepEntry(ctypes.Structure):
    _fields_ = [("master_off", c_i64), ("w_f_off", c_i64), ("w_d_off", c_i64), ("tile_begin", c_i64),
                ("Cout", ctypes.c_int32), ("T", ctypes.c_int32), ("Cin", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class AugSample(ctypes.Structure):
    _fields_ = [("img_off", c_i64), ("vrows", ctypes.c_int32), ("vcols", ctypes.c_int32), ("vslices", ctypes.c_int32),
                ("a0", ctypes.c_int32), ("b0", ctypes.c_int32), ("c0", ctypes.c_int32), ("crop", ctypes.c_int32),
                ("flip", ctypes.c_int32)]


class ZeroEntry(ctypes.Structure):
    _fields_ = [("ptr", c_p), ("bytes", ctypes.c_uint64), ("block_begin", ctypes.c_uint32), ("pad_", ctypes.c_uint32),
                ("src", c_p)]


class BnBwdEntry(ctypes.Structure):
    """include/hdu.h: hdu_bnbwd_entry"""
    _fields_ = [("partial", c_p), ("slots", ctypes.c_int32), ("C", ctypes.c_int32), ("gamma", c_p), ("beta", c_p),
                ("sgamma", c_p), ("dgamma", c_p), ("dbeta", c_p), ("dsgamma", c_p), ("dsbeta", c_p)]


class BnStatsFold(ctypes.Structure):
    """include/hdu.h: hdu_stats_fold_desc"""
    _fields_ = [("partial", c_p), ("slots", ctypes.c_int32), ("Cseg", ctypes.c_int32), ("seg_c0", ctypes.c_int32),
                ("pad_", ctypes.c_int32), ("M", c_i64), ("shift", c_p), ("mean", c_p), ("var", c_p), ("gamma", c_p),
                ("beta", c_p), ("sgamma", c_p), ("sbeta", c_p), ("eps", c_f), ("momentum", c_f), ("a", c_p), ("b", c_p),
                ("rstd", c_p), ("mov_mean", c_p), ("mov_var", c_p)]


ZERO_BLOCK_BYTES = 65536      # include/hdu.h HDU_ZERO_BLOCK_BYTES


class Split3Entry(ctypes.Structure):
    """include/hdu.h hdu_split3_entry"""
    _fields_ = [("src", c_p), ("dst", c_p), ("a", c_p), ("b", c_p), ("ld_src", ctypes.c_uint64), ("rows", ctypes.c_uint64),
                ("C", ctypes.c_uint32), ("relu", ctypes.c_uint32), ("pattern", ctypes.c_uint32), ("cols", ctypes.c_uint32),
                ("col_groups", ctypes.c_uint32), ("block_begin", ctypes.c_uint32), ("iters", ctypes.c_uint32),
                ("chl", ctypes.c_uint32)]


SPLIT3_OPERAND, SPLIT3_GRADIENT = 0, 1      # include/hdu.h HDU_SPLIT3_*


class FoldEntry(ctypes.Structure):
    _fields_ = [("mean", c_p), ("var", c_p), ("gamma", c_p), ("beta", c_p), ("sgamma", c_p), ("sbeta", c_p),
                ("a", c_p), ("b", c_p), ("rstd", c_p), ("C", ctypes.c_int32), ("eps", c_f)]


_SIGS = {
    "hdu_last_error": (ctypes.c_char_p, []),
    "hdu_backend": (ctypes.c_char_p, []),
    "hdu_abi_version": (c_int, []),
    "hdu_sizeof_conv_desc": (c_sz, []),
    "hdu_set_tuning": (c_int, [c_int, c_int]),
    "hdu_conv_fprop": (c_int, [ctypes.POINTER(ConvDesc), c_p]),
    "hdu_bn_bwd_finalize": (c_int, [c_p, c_int, c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "hdu_bn_bwd_correct": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_p, c_i64, c_p]),
    "hdu_bn_bwd_finalize_correct": (c_int, [c_int, c_p, c_int, c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                            c_int, c_int, c_p, c_i64, c_p, c_i64, c_p]),
    "hdu_conv_splitk_ws_bytes": (c_sz, [ctypes.POINTER(ConvDesc)]),
    "hdu_conv_wgrad": (c_int, [ctypes.POINTER(ConvDesc), c_p, c_p]),
    "hdu_conv_dgrad_strided": (c_int, [ctypes.POINTER(ConvDesc), c_p]),
    "hdu_stride2_dgrad_filters": (c_int, [c_int, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_p, c_p]),
    "hdu_parity_interleave": (c_int, [c_int, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_i64, c_int,
                                      c_p]),
    "hdu_conv_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.c_char_p, c_sz]),
    "hdu_weight_prep": (c_int, [c_int, c_p, c_int, c_int, c_int, c_p, c_p, c_p]),
    "hdu_weight_prep_batched": (c_int, [c_int, c_p, c_int, c_i64, c_p, c_p, c_p]),
    "hdu_reduce_ws_bytes": (c_sz, [c_i64, c_int]),
    "hdu_bn_stats": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_p, c_sz, c_p]),
    "hdu_bn_fold": (c_int, [c_int, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p]),
    "hdu_augment_batch": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_i64,
                                  c_i64, c_p]),
    "hdu_bn_fold_batched": (c_int, [c_p, c_p, c_int, ctypes.c_uint32, c_p]),
    "hdu_bn_stats_fold": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p,
                                  c_p, c_f, c_p, c_sz, c_p]),
    "hdu_bn_bwd_reduce_coef": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_p, c_int,
                                       c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "hdu_bn_bwd_reduce": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_p, c_p, c_p,
                                  c_p, c_sz, c_p]),
    "hdu_bn_bwd_coef": (c_int, [c_int, c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                c_p, c_p]),
    "hdu_bn_bwd_apply": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_p, c_p, c_p,
                                 c_p, c_i64, c_int, c_f, c_u32, c_p, c_p]),
    "hdu_bn_bwd_fused": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_p,
                                 c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_f, c_u32, c_p, c_p]),
    "hdu_bn_bwd_apply_sums": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_p,
                                      c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_f, c_u32, c_p, c_p]),
    "hdu_affine_act": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_p, c_i64, c_p]),
    "hdu_materialize": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_int, c_int, c_int, c_int,
                                c_p, c_i64, c_p, c_i64, c_p]),
    "hdu_bn_bwd_finalize_batched": (c_int, [c_p, c_p, c_int, c_u32, c_p]),
    "hdu_materialize_stats": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_int, c_int, c_int,
                                      c_p, c_i64, c_p, c_i64, c_p]),
    "hdu_wgrad_plan_entry_bytes": (c_sz, []),
    "hdu_wgrad_plan_fill": (c_int, [ctypes.POINTER(ConvDesc), c_p, c_int, c_int, c_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_uint32)]),
    "hdu_wgrad_plan_shape": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_uint32),
                                     ctypes.POINTER(ctypes.c_uint32)]),
    "hdu_wgrad_plan_run": (c_int, [c_int, c_p, c_p, c_int, ctypes.c_uint32, c_p]),
    "hdu_bn_stats_finalize_fold_next": (c_int, [c_p, c_int, c_i64, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p,
                                                c_p, c_p, c_p, c_p, c_p, c_f, c_p]),
    "hdu_bn_stats_finalize": (c_int, [c_p, c_int, c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_f, c_p]),
    "hdu_stats_sync_floats": (c_sz, [c_int, c_int]),
    "hdu_stats_pack": (c_int, [c_int, c_p, c_p, c_i64, c_int, c_int, c_p, c_p]),
    "hdu_stats_unpack": (c_int, [c_int, c_p, c_int, c_p, c_p, c_p]),
    "hdu_colsum": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_sz, c_p]),
    "hdu_maxpool3s2_fwd": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_i64, c_p, c_int, c_p]),
    "hdu_maxpool3s2_bwd": (c_int, [c_int, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_i64, c_int, c_int, c_p]),
    "hdu_avgpool2_fwd": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_i64, c_p]),
    "hdu_avgpool2_bwd": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p, c_i64, c_int, c_p]),
    "hdu_upsample_bwd": (c_int, [c_int, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p,
                                 c_i64, c_int, c_p]),
    "hdu_wce_loss": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_f, c_f, c_f, c_f, c_p, c_i64, c_int, c_p, c_p, c_p,
                             c_sz, c_p]),
    "hdu_sgd_nesterov": (c_int, [c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_p]),
    "hdu_slab25d": (c_int, [c_int, c_p, c_int, c_int, c_int, c_p, c_int, c_p]),
    "hdu_make_input3d": (c_int, [c_int, c_p, c_p, c_i64, c_f, c_int, c_int, c_int, c_p, c_int, c_p]),
    "hdu_make_input3d_bwd": (c_int, [c_int, c_p, c_int, c_f, c_i64, c_p, c_i64, c_int, c_int, c_p]),
    "hdu_cast_pad": (c_int, [c_int, c_p, c_i64, c_int, c_p, c_i64, c_int, c_p]),
    "hdu_cast_out": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p]),
    "hdu_softmax_accumulate": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p]),
    "hdu_zero_regions": (c_int, [c_p, c_int, c_u32, c_p, c_u32, c_p]),
    "hdu_zero": (c_int, [c_p, ctypes.c_uint64, c_p]),
    "hdu_split3_entry_fill": (c_int, [c_p, c_p, c_i64, c_i64, c_int, c_p, c_p, c_int, c_int, c_p, c_u32, ctypes.POINTER(c_u32)]),
    "hdu_split3_batched": (c_int, [c_p, c_int, c_u32, c_int, c_p]),
    "hdu_profile_begin": (c_int, [c_int]),
    "hdu_profile_count": (c_int, []),
    "hdu_profile_end": (c_int, []),
    "hdu_profile_get": (c_int, [c_int, ctypes.c_char_p, c_sz, ctypes.POINTER(c_f)]),
    "hdu_comm_unique_id": (c_int, [c_p]),
    "hdu_comm_init": (c_int, [ctypes.POINTER(c_p), c_int, c_int, c_p]),
    "hdu_comm_destroy": (c_int, [c_p]),
    "hdu_comm_allreduce_f32": (c_int, [c_p, c_p, c_i64, c_p]),
    "hdu_comm_sendrecv": (c_int, [c_p, c_int, c_p, c_p, c_int, c_p, c_p, c_sz, c_p]),
}

EXPORTS = tuple(_SIGS.keys())

_lib = None
_backend = None


class HduError(RuntimeError):
    pass


ABI_VERSION = 7        # include/hdu.h HDU_ABI_VERSION


def product_library_path():
    return os.path.join(_HERE, "libhdu.so")


def _bind(path):
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be resident BEFORE libhdu.so is loaded so
    # that both share ONE runtime instance (otherwise our launches hit a second, device-less runtime).
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    # a library built from other sources than this binding (stale .so after a pull) must not be driven through structs
    # of a different layout
    if lib.hdu_abi_version() != ABI_VERSION or lib.hdu_sizeof_conv_desc() != ctypes.sizeof(ConvDesc):
        raise HduError("%s has ABI version %d / hdu_conv_desc of %d bytes, this binding expects %d / %d -- rebuild it "
                       "(python -c 'import __graft_entry__ as g; g.build()')" %
                       (path, lib.hdu_abi_version(), lib.hdu_sizeof_conv_desc(), ABI_VERSION, ctypes.sizeof(ConvDesc)))
    return lib


def _apply_env_tuning(lib):
    """developer knobs (A/B runs): HDU_* environment variables -> hdu_set_tuning"""
    if "HDU_DMA_STAGES" in os.environ:      # developer knobs (A/B runs)
        lib.hdu_set_tuning(0, int(os.environ["HDU_DMA_STAGES"]))
    if "HDU_SPLITK" in os.environ:
        lib.hdu_set_tuning(13, int(os.environ["HDU_SPLITK"]))
    if "HDU_BM64_MAX_M" in os.environ:
        lib.hdu_set_tuning(18, int(os.environ["HDU_BM64_MAX_M"]))
    if "HDU_SPLITK_TARGET" in os.environ:
        lib.hdu_set_tuning(16, int(os.environ["HDU_SPLITK_TARGET"]))
    if "HDU_SPLITK_MIN_STEPS" in os.environ:
        lib.hdu_set_tuning(17, int(os.environ["HDU_SPLITK_MIN_STEPS"]))
    if "HDU_HALO_MIN_TILES" in os.environ:
        lib.hdu_set_tuning(15, int(os.environ["HDU_HALO_MIN_TILES"]))
    if "HDU_RING_MIN_K" in os.environ:
        lib.hdu_set_tuning(14, int(os.environ["HDU_RING_MIN_K"]))
    if "HDU_RED_WGS" in os.environ:
        lib.hdu_set_tuning(11, int(os.environ["HDU_RED_WGS"]))
    if "HDU_ROW_WGS" in os.environ:
        lib.hdu_set_tuning(12, int(os.environ["HDU_ROW_WGS"]))
    if "HDU_NO_HALO_FPROP" in os.environ:
        lib.hdu_set_tuning(9, int(os.environ["HDU_NO_HALO_FPROP"]))
    if "HDU_NO_HALO" in os.environ:
        lib.hdu_set_tuning(8, int(os.environ["HDU_NO_HALO"]))
    if "HDU_HALO_WIDE" in os.environ:       # 1 = the im2col kernels of rounds 1-4 for the wide 3x3 / 3x3x3 layers (A/B)
        lib.hdu_set_tuning(29, int(os.environ["HDU_HALO_WIDE"]))
    if "HDU_HALO_MIN_W" in os.environ:
        lib.hdu_set_tuning(28, int(os.environ["HDU_HALO_MIN_W"]))
    if "HDU_HALO_TARGET" in os.environ:
        lib.hdu_set_tuning(7, int(os.environ["HDU_HALO_TARGET"]))
    if "HDU_MAX_BN" in os.environ:
        lib.hdu_set_tuning(6, int(os.environ["HDU_MAX_BN"]))
    if "HDU_NO_FAST" in os.environ:
        lib.hdu_set_tuning(5, int(os.environ["HDU_NO_FAST"]))
    if "HDU_DEBUG_FLAGS" in os.environ:
        lib.hdu_set_tuning(4, int(os.environ["HDU_DEBUG_FLAGS"]))
    if "HDU_XCD_SWIZZLE" in os.environ:
        lib.hdu_set_tuning(3, int(os.environ["HDU_XCD_SWIZZLE"]))
    if "HDU_WGRAD_TARGET" in os.environ:
        lib.hdu_set_tuning(2, int(os.environ["HDU_WGRAD_TARGET"]))
    if "HDU_NO_PW_BSTAT" in os.environ:
        lib.hdu_set_tuning(19, int(os.environ["HDU_NO_PW_BSTAT"]))
    if "HDU_PW_BSTAT_WGS" in os.environ:
        lib.hdu_set_tuning(20, int(os.environ["HDU_PW_BSTAT_WGS"]))
    if "HDU_PW_BSTAT_FORM" in os.environ:   # 1 = 128 channels per workgroup, one per CU (rounds 3-5); 2 = 64 channels, two per CU (round 6)
        lib.hdu_set_tuning(30, int(os.environ["HDU_PW_BSTAT_FORM"]))
    if "HDU_SPLIT3_FORM" in os.environ:
        lib.hdu_set_tuning(31, int(os.environ["HDU_SPLIT3_FORM"]))
    if "HDU_WGRAD_NCT" in os.environ:
        lib.hdu_set_tuning(21, int(os.environ["HDU_WGRAD_NCT"]))
    if "HDU_NO_PRO_DMA" in os.environ:
        lib.hdu_set_tuning(22, int(os.environ["HDU_NO_PRO_DMA"]))
    if "HDU_PERS" in os.environ:
        lib.hdu_set_tuning(23, int(os.environ["HDU_PERS"]))
    if "HDU_NO_PW_BSTAT_BNB" in os.environ:
        lib.hdu_set_tuning(26, int(os.environ["HDU_NO_PW_BSTAT_BNB"]))
    lib.hdu_set_tuning(TUNE_F32_SPLIT, _F32_MODES[_f32_contraction])
    if "HDU_PERS_MIN_ITEMS" in os.environ:
        lib.hdu_set_tuning(24, int(os.environ["HDU_PERS_MIN_ITEMS"]))
    if "HDU_WGRAD_MIN_STEPS" in os.environ:
        lib.hdu_set_tuning(1, int(os.environ["HDU_WGRAD_MIN_STEPS"]))


def load(path=None):
    """Bind the product library (gfx950).  Raises if it has not been built."""
    global _lib, _backend
    path = path or product_library_path()
    if not os.path.exists(path):
        raise HduError(
            "libhdu.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    bound = _bind(path)
    backend = bound.hdu_backend().decode()
    if backend != "hip-gfx950":
        raise HduError("%s is not the gfx950 product library (hdu_backend() = %r).  There is no CPU fallback." % (path, backend))
    _lib, _backend = bound, backend
    _apply_env_tuning(_lib)
    return _lib


class _CallLog:
    """proxy of the bound library used while the launch profiler is armed (profile_begin): every C-ABI call is forwarded
    unchanged, and (entry point, its arguments, index of the first / one-past-last kernel record it produced) is appended to
    `calls`, so that a caller (bench.py) can attach an algorithmic-work model to each recorded launch"""

    def __init__(self, lib):
        self._lib, self.calls = lib, []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("hdu_") or name.startswith("hdu_profile") or name in ("hdu_last_error", "hdu_conv_kernel_name"):
            return fn
        lib, calls = self._lib, self.calls

        def wrapped(*args):
            n0 = lib.hdu_profile_count()
            r = fn(*args)
            n1 = lib.hdu_profile_count()
            if n1 > n0:
                calls.append((name, args, n0, n1))
            return r
        return wrapped


_call_log = None

TUNE_F32_SPLIT = 27     # include/hdu.h: HDU_TUNE_F32_SPLIT


def set_f32_contraction(mode):
    """how the float32 networks (dtype "f32": float32 storage, statistics, row kernels) contract in their convolutions,
    process-wide and read at launch time: "exact" (default) = float32 MFMA, the parity mode; "bf16x3" = every operand split
    into bf16 hi + lo, a.b ~ ah.bh + ah.bl + al.bh on the bf16 MFMA with the float32 accumulator (<= 3 * 2^-18 relative per
    product); "bf16x3_bwd" = exact forward, split backward (the logits are the parity mode's).
    Environment: HDU_F32_CONTRACTION=bf16x3 | bf16x3_bwd.  Returns the previous mode."""
    global _f32_contraction
    prev = _f32_contraction
    check(get().hdu_set_tuning(TUNE_F32_SPLIT, _F32_MODES[_check_f32_mode(mode)]), "hdu_set_tuning")
    _f32_contraction = mode
    return prev


# "bf16x3_bwd" (round 6): the FORWARD convolutions contract in exact float32 -- predict and training-phase logits are those of the
# parity mode, bit for bit -- and only the backward pass (data and filter gradients) uses the split contraction: engine.Ctx.run_backward
# switches HDU_TUNE_F32_SPLIT on for its launches (read at launch time, so a captured step bakes it in per launch).
_F32_MODES = {"exact": 0, "bf16x3": 1, "bf16x3_bwd": 0}


def f32_split_in_backward_only():
    return _f32_contraction == "bf16x3_bwd"


def f32_contraction():
    """the current mode of set_f32_contraction"""
    return _f32_contraction


def set_f32_split_now(on):
    """engine hook of the "bf16x3_bwd" mode: the split contraction for the launches that follow (float32 networks only)"""
    check(get().hdu_set_tuning(TUNE_F32_SPLIT, 1 if on else 0), "hdu_set_tuning")


def _check_f32_mode(mode, what="f32 contraction mode"):
    if mode not in _F32_MODES:
        raise ValueError("%s: 'exact', 'bf16x3' or 'bf16x3_bwd', not %r" % (what, mode))
    return mode


# validated ONCE, at import (an unknown value used to surface as a bare KeyError while the library loaded, and
# set_f32_contraction could hand it back as the 'previous' mode)
_f32_contraction = _check_f32_mode(os.environ.get("HDU_F32_CONTRACTION", "exact"), "HDU_F32_CONTRACTION")


def profile_begin(max_records=1 << 16):
    """arm the library's launch profiler (include/hdu.h: hdu_profile_*) and start logging the C-ABI calls"""
    global _call_log
    lib = get() if _call_log is None else _call_log._lib
    check(lib.hdu_profile_begin(max_records), "hdu_profile_begin")
    _call_log = _CallLog(lib)


def profile_end():
    """-> ([(kernel name, milliseconds)] in launch order, [(entry point, args, first record, one past last record)])"""
    global _call_log
    log, _call_log = _call_log, None
    lib = log._lib
    n = lib.hdu_profile_end()
    if n < 0:
        raise HduError("hdu_profile_end failed: %s" % lib.hdu_last_error().decode())
    buf = ctypes.create_string_buffer(256)
    ms = c_f(0.0)
    recs = []
    for i in range(n):
        check(lib.hdu_profile_get(i, buf, 256, ctypes.byref(ms)), "hdu_profile_get")
        recs.append((buf.value.decode(), float(ms.value)))
    return recs, log.calls


def get():
    if _call_log is not None:
        return _call_log
    if _lib is None:
        load()
    return _lib


def backend():
    get()
    return _backend


def is_emulator():
    return backend() == "emu-x86"


def check(code, what=""):
    if code != 0:
        raise HduError("%s failed (%d): %s" % (what or "hdu call", code, get().hdu_last_error().decode()))
