"""Micro-benchmark of the conv kernels on the distinct implicit-GEMM shapes of SURVEY.md A.4 (developer tool;
prints TFLOP/s per shape for fprop / dgrad-form / wgrad).  Usage: python tools/bench_kernels.py [2d|3d] [bf16|f32]"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("h-denseunet_amd")
pkg.lib.load()
ops = importlib.import_module("h-denseunet_amd.ops")

which = sys.argv[1] if len(sys.argv) > 1 else "2d"
dtype = 0 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else 1
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None

# (name, N, D, H, W, Cin, Cout, K, up, pro)
S2D = [
    ("stem7x7s2", 8, 1, 512, 512, 8, 96, (1, 7, 7), (0, 0, 0), False, (1, 2, 2), (0, 3, 3)),
    ("b2_x1_1x1_336", 8, 1, 128, 128, 336, 192, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
    ("b2_x2_3x3", 8, 1, 128, 128, 192, 48, (1, 3, 3), (0, 0, 0), True, (1, 1, 1), (0, 1, 1)),
    ("b3_x1_1x1_720", 8, 1, 64, 64, 720, 192, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
    ("b3_x2_3x3", 8, 1, 64, 64, 192, 48, (1, 3, 3), (0, 0, 0), True, (1, 1, 1), (0, 1, 1)),
    ("b4_x1_1x1_2064", 8, 1, 32, 32, 2064, 192, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
    ("b4_x2_3x3", 8, 1, 32, 32, 192, 48, (1, 3, 3), (0, 0, 0), True, (1, 1, 1), (0, 1, 1)),
    ("b5_x1_1x1_2160", 8, 1, 16, 16, 2160, 192, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
    ("b5_x2_3x3", 8, 1, 16, 16, 192, 48, (1, 3, 3), (0, 0, 0), True, (1, 1, 1), (0, 1, 1)),
    ("trans4_1x1", 8, 1, 32, 32, 2112, 1056, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
    ("line0_1x1", 8, 1, 32, 32, 2112, 2208, (1, 1, 1), (0, 0, 0), False, (1, 1, 1), (0, 0, 0)),
    ("conv_up0", 8, 1, 16, 16, 2208, 768, (1, 3, 3), (0, 1, 1), True, (1, 1, 1), (0, 1, 1)),
    ("conv_up1", 8, 1, 32, 32, 768, 384, (1, 3, 3), (0, 1, 1), True, (1, 1, 1), (0, 1, 1)),
    ("conv_up2", 8, 1, 64, 64, 384, 96, (1, 3, 3), (0, 1, 1), True, (1, 1, 1), (0, 1, 1)),
    ("conv_up3", 8, 1, 128, 128, 96, 96, (1, 3, 3), (0, 1, 1), True, (1, 1, 1), (0, 1, 1)),
    ("conv_up4", 8, 1, 256, 256, 96, 64, (1, 3, 3), (0, 1, 1), True, (1, 1, 1), (0, 1, 1)),
    ("classifier", 8, 1, 512, 512, 64, 8, (1, 1, 1), (0, 0, 0), True, (1, 1, 1), (0, 0, 0)),
]
S3D = [
    ("3dstem", 1, 12, 224, 224, 8, 96, (7, 7, 7), (0, 0, 0), False, (2, 2, 2), (3, 3, 3)),
    ("3db2_x2", 1, 3, 56, 56, 128, 32, (3, 3, 3), (0, 0, 0), True, (1, 1, 1), (1, 1, 1)),
    ("3db4_x2", 1, 3, 14, 14, 128, 32, (3, 3, 3), (0, 0, 0), True, (1, 1, 1), (1, 1, 1)),
    ("3dconv_up2", 1, 3, 28, 28, 224, 192, (3, 3, 3), (0, 1, 1), True, (1, 1, 1), (1, 1, 1)),
    ("3dconv_up3", 1, 3, 56, 56, 192, 96, (3, 3, 3), (1, 1, 1), True, (1, 1, 1), (1, 1, 1)),
    ("3dconv_up4", 1, 6, 112, 112, 96, 64, (3, 3, 3), (1, 1, 1), True, (1, 1, 1), (1, 1, 1)),
    ("fianl_conv", 1, 12, 224, 224, 64, 64, (3, 3, 3), (0, 0, 0), True, (1, 1, 1), (1, 1, 1)),
]
tdt = torch.bfloat16 if dtype == 0 else torch.float32


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for (name, N, D, H, W, Cin, Cout, K, up, pro, st, pad) in (S2D if which == "2d" else S3D):
    if only and name not in only:
        continue
    x = ops.Act.alloc(N, D, H, W, Cin, dtype); x.buf.normal_()
    De, He, We = D << up[0], H << up[1], W << up[2]
    Do, Ho, Wo = [(n + 2 * p - k) // s + 1 for n, p, k, s in zip((De, He, We), pad, K, st)]
    y = ops.Act.alloc(N, Do, Ho, Wo, Cout, dtype); y.buf.normal_()
    T = K[0] * K[1] * K[2]
    w = (torch.randn(Cout * T * Cin, device="cuda") * 0.05).to(tdt)
    a = torch.rand(Cin, device="cuda") + 0.5; b = torch.rand(Cin, device="cuda") - 0.5
    if os.environ.get("BENCH_NOPRO"):
        pro = False          # model-like: inputs are materialised, every conv is the DMA form
    d = ops.conv_desc(x, ctypes.c_void_p(w.data_ptr()), y, K, st, pad, up, None, (a, b) if pro else None, True)
    flops = 2.0 * N * Do * Ho * Wo * Cout * T * Cin
    t_f = timeit(lambda: ops.conv_fprop(d))
    line = "%-16s M=%8d N=%5d K=%6d | fprop %8.3f ms %7.1f TF" % (name, N * Do * Ho * Wo, Cout, T * Cin, t_f, flops / t_f / 1e9)
    tot["fprop"] += t_f
    if st == (1, 1, 1):
        # dgrad form: input dy (Cout ch) at output res -> dx_eff (Cin ch) at effective res
        dxe = ops.Act.alloc(N, De, He, We, Cin, dtype)
        dd = ops.conv_desc(y, ctypes.c_void_p(w.data_ptr()), dxe, K, (1, 1, 1), tuple(k - 1 - p for k, p in zip(K, pad)))
        t_d = timeit(lambda: ops.conv_fprop(dd))
        tot["dgrad"] += t_d
        line += " | dgrad %8.3f ms %7.1f TF" % (t_d, flops / t_d / 1e9)
    dw = torch.zeros(Cout * T * Cin, device="cuda")
    t_w = timeit(lambda: ops.conv_wgrad(d, dw))
    tot["wgrad"] += t_w
    line += " | wgrad %8.3f ms %7.1f TF" % (t_w, flops / t_w / 1e9)
    if st == (1, 1, 1) and os.environ.get("BENCH_PAIR"):
        # would a merged dgrad+wgrad launch pay?  Upper bound: the two kernels on two streams inside one hipGraph
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

        def pair():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                ops.conv_fprop(dd)
            with torch.cuda.stream(s2):
                ops.conv_wgrad(d, dw)
            cur.wait_stream(s1); cur.wait_stream(s2)

        def seq():
            ops.conv_fprop(dd); ops.conv_wgrad(d, dw)

        res = []
        for fn in (seq, pair):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    fn()
            res.append(timeit(g.replay) / 10)
        line += " | graph seq %6.1f us, 2 streams %6.1f us" % (res[0] * 1e3, res[1] * 1e3)
    print(line, flush=True)
print("sum ms:", tot)
