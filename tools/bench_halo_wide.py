"""Per-layer A/B of the halo-tile forward / data-gradient kernel (csrc/conv_halo_wide.hip) against the im2col kernels on the wide
3x3 / 3x3x3 layers of the BASELINE workloads (developer tool; prints us + TFLOP/s per layer, form and configuration).
Usage: python tools/bench_halo_wide.py [2d|shard|v224|all] [cfgs, e.g. 1,0,2,3,4,5,6]
  HDU_TUNE_HALO_WIDE values: 1 = off (im2col kernels), 0 = library heuristic, 2..6 = forced 8x128 / 16x64 / 16x96 / 8x64 / 8x96"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("h-denseunet_amd")
pkg.lib.load()
ops = importlib.import_module("h-denseunet_amd.ops")
lib = pkg.lib.get()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
ONLY = os.environ.get("HW_ONLY")          # "layer:form" -- one layer / form only (PMC probes)
cfgs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 0, 2, 3, 4, 5, 6, 7]
NAMES = {1: "im2col", 0: "auto", 2: "8x128", 3: "16x64", 4: "16x96", 5: "8x64", 6: "8x96", 7: "16x128", 8: "16x64p"}

# (name, N, D, H, W (stored input), Cin, Cout, K, up)
L2D = [
    ("conv_up0", 8, 1, 32, 32, 2208, 768, (1, 3, 3), (0, 0, 0)),
    ("conv_up1", 8, 1, 64, 64, 768, 384, (1, 3, 3), (0, 0, 0)),
    ("conv_up2", 8, 1, 128, 128, 384, 96, (1, 3, 3), (0, 0, 0)),
    ("conv_up3", 8, 1, 256, 256, 96, 96, (1, 3, 3), (0, 0, 0)),
    ("conv_up4", 8, 1, 256, 256, 96, 64, (1, 3, 3), (0, 1, 1)),
    ("b2_x2", 8, 1, 128, 128, 192, 48, (1, 3, 3), (0, 0, 0)),
    ("b3_x2", 8, 1, 64, 64, 192, 48, (1, 3, 3), (0, 0, 0)),
    ("b4_x2", 8, 1, 32, 32, 192, 48, (1, 3, 3), (0, 0, 0)),
]
LSHARD = [
    ("3dconv_up0", 1, 16, 16, 16, 504, 504, (3, 3, 3), (0, 1, 1)),
    ("3dconv_up1", 1, 16, 32, 32, 504, 224, (3, 3, 3), (0, 1, 1)),
    ("3dconv_up2", 1, 16, 64, 64, 224, 192, (3, 3, 3), (0, 1, 1)),
    ("3dconv_up3", 1, 16, 128, 128, 192, 96, (3, 3, 3), (1, 1, 1)),
    ("3dconv_up4", 1, 32, 256, 256, 96, 64, (3, 3, 3), (1, 1, 1)),
    ("3db2_x2", 1, 16, 128, 128, 128, 32, (3, 3, 3), (0, 0, 0)),
]
L224 = [
    ("3dconv_up1", 1, 3, 14, 14, 504, 224, (3, 3, 3), (0, 1, 1)),
    ("3dconv_up2", 1, 3, 28, 28, 224, 192, (3, 3, 3), (0, 1, 1)),
    ("3dconv_up3", 1, 3, 56, 56, 192, 96, (3, 3, 3), (1, 1, 1)),
    ("3dconv_up4", 1, 6, 112, 112, 96, 64, (3, 3, 3), (1, 1, 1)),
    ("fianl_conv", 1, 12, 224, 224, 64, 64, (3, 3, 3), (0, 0, 0)),
    ("conv_up2@224", 12, 1, 56, 56, 384, 96, (1, 3, 3), (0, 0, 0)),
    ("conv_up3@224", 12, 1, 112, 112, 96, 96, (1, 3, 3), (0, 0, 0)),
    ("conv_up4@224", 12, 1, 112, 112, 96, 64, (1, 3, 3), (0, 1, 1)),
]


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(layers, tag):
    for (name, N, D, H, W, Cin, Cout, K, up) in layers:
        if ONLY and ONLY.split(":")[0] != name:
            continue
        pad = (K[0] // 2, 1, 1)
        x = ops.Act.alloc(N, D, H, W, Cin, 0); x.buf.normal_()
        De, He, We = D << up[0], H << up[1], W << up[2]
        y = ops.Act.alloc(N, De, He, We, Cout, 0); y.buf.normal_()
        T = K[0] * 9
        w = (torch.randn(Cout * T * Cin, device="cuda") * 0.05).to(torch.bfloat16)
        dxe = ops.Act.alloc(N, De, He, We, Cin, 0)
        d_f = ops.conv_desc(x, ctypes.c_void_p(w.data_ptr()), y, K, (1, 1, 1), pad, up)
        d_d = ops.conv_desc(y, ctypes.c_void_p(w.data_ptr()), dxe, K, (1, 1, 1), pad)
        flops = 2.0 * N * De * He * We * Cout * T * Cin
        for form, d in (("fprop", d_f), ("dgrad", d_d)):
            if ONLY and ":" in ONLY and ONLY != name + ":" + form:
                continue
            line = "%-6s %-13s %-5s %6.1f GF |" % (tag, name, form, flops / 1e9)
            for c in cfgs:
                lib.hdu_set_tuning(29, c)
                kn = ops.conv_kernel_name(d, 0)
                if c >= 2 and not kn.startswith("conv_halo_wide"):
                    line += " %s: n/a |" % NAMES[c]
                    continue
                t = timeit(lambda: ops.conv_fprop(d))
                lab = NAMES[c] if c != 0 else "auto[" + kn.replace("conv_halo_wide_kernel", "hw").replace("conv_igemm_", "")[:10] + "]"
                line += " %s %6.0f us %4.0f |" % (lab, t * 1e3, flops / t / 1e9) if c < 2 else " %s %4.0f |" % (lab, flops / t / 1e9)
            lib.hdu_set_tuning(29, 0)
            print(line, flush=True)
        del x, y, dxe, w


def run_stem():
    for (name, N, D, H, W, K, st, pad) in (("2d stem 8x512^2", 8, 1, 512, 512, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
                                           ("3d stem 224^2x12", 1, 12, 224, 224, (7, 7, 7), (2, 2, 2), (3, 3, 3)),
                                           ("3d stem 512^2x64", 1, 64, 512, 512, (7, 7, 7), (2, 2, 2), (3, 3, 3))):
        x = ops.Act.alloc(N, D, H, W, 8, 0); x.buf.normal_()
        Do, Ho, Wo = [(n + 2 * p - k) // s_ + 1 for n, p, k, s_ in zip((D, H, W), pad, K, st)]
        y = ops.Act.alloc(N, Do, Ho, Wo, 96, 0)
        T = K[0] * 49
        w = (torch.randn(96 * T * 8, device="cuda") * 0.05).to(torch.bfloat16)
        d = ops.conv_desc(x, ctypes.c_void_p(w.data_ptr()), y, K, st, pad)
        flops = 2.0 * N * Do * Ho * Wo * 96 * T * 8
        line = "stem   %-18s fprop %7.1f GF (8 stored channels) |" % (name, flops / 1e9)
        for c in (1, 0):
            lib.hdu_set_tuning(29, c)
            t = timeit(lambda: ops.conv_fprop(d))
            line += " %s %7.0f us %5.0f TF |" % (ops.conv_kernel_name(d, 0)[:24], t * 1e3, flops / t / 1e9)
        lib.hdu_set_tuning(29, 0)
        print(line, flush=True)
        y.buf.normal_()
        dw = torch.zeros(96 * T * 8, device="cuda")
        line = "stem   %-18s wgrad %7.1f GF (8 stored channels) |" % (name, flops / 1e9)
        for c in (1, 0):
            lib.hdu_set_tuning(29, c)
            t = timeit(lambda: ops.conv_wgrad(d, dw))
            line += " %s %7.0f us %5.0f TF |" % (ops.conv_kernel_name(d, 1)[:24], t * 1e3, flops / t / 1e9)
        lib.hdu_set_tuning(29, 0)
        print(line, flush=True)


if which in ("stem", "all"):
    run_stem()
if which in ("2d", "all"):
    run(L2D, "2d")
if which in ("shard", "all"):
    run(LSHARD, "shard")
if which in ("v224", "all"):
    run(L224, "v224")
