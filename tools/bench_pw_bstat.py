"""developer micro-benchmark of conv_pw_bstat_kernel (the roofline kernel of the 2D line): the bottleneck data gradient
dx[M x C] = dt[M x 192] . W^T on the dense-block shapes of the 2D step, in three forms -- plain store, accumulate, and the fused
BN backward (u + old read, S1 / S2) -- 20 launches per hipGraph replay.  Prints us per launch and algorithmic TB/s."""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("h-denseunet_amd")
pkg.lib.load()
ops = importlib.import_module("h-denseunet_amd.ops")
BF16 = 0
SHAPES = [(2048, 1632), (8192, 1584), (8192, 624), (32768, 576), (131072, 240)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, C in SHAPES:
    H = 64
    dt = ops.Act.alloc(1, 1, M // H, H, 192, BF16); dt.buf.normal_()
    slab_u = ops.Act.alloc(1, 1, M // H, H, C, BF16); slab_u.buf.normal_()
    out = ops.Act.alloc(1, 1, M // H, H, C, BF16); out.buf.zero_()
    w = (torch.randn(C * 192, device="cuda") * 0.05).to(torch.bfloat16)
    a = torch.rand(C, device="cuda") + 0.5; b = torch.rand(C, device="cuda") - 0.5
    mean = torch.rand(C, device="cuda"); rstd = torch.rand(C, device="cuda") + 0.5
    part = torch.zeros(32 * 2 * C, device="cuda")
    line = "M=%6d C=%4d" % (M, C)
    for form in ("plain", "acc", "bnb_acc"):
        d = ops.conv_desc(dt, ctypes.c_void_p(w.data_ptr()), out, (1, 1, 1), accumulate=(form != "plain"))
        nbytes = M * (192 * 2 + C * 2 * (1 if form == "plain" else 2))
        if form == "bnb_acc":
            d.bnb_u, d.bnb_ldu = slab_u.ptr, slab_u.ld
            d.bnb_a, d.bnb_b, d.bnb_relu = a.data_ptr(), b.data_ptr(), 1
            d.bnb_mean, d.bnb_rstd, d.bnb_partial, d.bnb_slots = mean.data_ptr(), rstd.data_ptr(), part.data_ptr(), 32
            nbytes += M * C * 2
        name = ops.conv_kernel_name(d, 0)
        ops.conv_fprop(d); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.conv_fprop(d)
        us = timeit(g.replay) / 20 * 1e3
        line += " | %s %6.1f us %5.2f TB/s" % (form, us, nbytes / us / 1e6)
        if not name.startswith("conv_pw_bstat"):
            line += " (%s)" % name.split("<")[0]
    print(line, flush=True)
