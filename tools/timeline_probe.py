"""Where does a small implicit-GEMM launch spend its 10-20 us?  (developer tool)

`tools/build_timeline_lib.sh` compiles the same sources with -DHDU_TIMELINE into tools/libhdu_tl.so, whose
conv_igemm_{dma,ring}_kernel stamp the shader clock at: 0 entry, 1 per-row state done, 2 prologue DMAs issued, 3 first
tile landed (wait + barrier), 4 K loop done, 5 split-K combine done (or this split leaves), 6 epilogue done; plus the
100 MHz constant clock at entry / exit.  Prints, per shape: the launch's event duration, the distribution of the
workgroups' entry and exit times on the common clock, and the median cycles of every segment.

Usage: python tools/timeline_probe.py
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("h-denseunet_amd")
lib = pkg.lib.load(os.path.join(ROOT, "tools", "libhdu_tl.so"))
ops = importlib.import_module("h-denseunet_amd.ops")
lib.hdu_timeline_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
WGS = 8192

# (name, N, D, H, W, Cin, Cout, K, pad, stats epilogue)
SHAPES = [
    ("2D b4 3x3 fprop  M=8192 N=48 K=1728", 8, 1, 32, 32, 192, 48, (1, 3, 3), (0, 1, 1), True),
    ("2D b4 3x3 dgrad  M=8192 N=192 K=432", 8, 1, 32, 32, 48, 192, (1, 3, 3), (0, 1, 1), False),
    ("2D b4 1x1 fprop  M=8192 N=192 K=1200", 8, 1, 32, 32, 1200, 192, (1, 1, 1), (0, 0, 0), True),
    ("2D b4 1x1 dgrad  M=8192 N=1200 K=192", 8, 1, 32, 32, 192, 1200, (1, 1, 1), (0, 0, 0), False),
    ("2D b5 3x3 fprop  M=2048 N=48 K=1728", 8, 1, 16, 16, 192, 48, (1, 3, 3), (0, 1, 1), True),
    ("2D b5 1x1 fprop  M=2048 N=192 K=1632", 8, 1, 16, 16, 1632, 192, (1, 1, 1), (0, 0, 0), True),
    ("2D b3 1x1 dgrad  M=32768 N=480 K=192", 8, 1, 64, 64, 192, 480, (1, 1, 1), (0, 0, 0), False),
    ("2D b3 3x3 dgrad  M=32768 N=192 K=432", 8, 1, 64, 64, 48, 192, (1, 3, 3), (0, 1, 1), False),
    ("2D b2 3x3 dgrad  M=131072 N=192 K=432", 8, 1, 128, 128, 48, 192, (1, 3, 3), (0, 1, 1), False),
    ("2D b2 1x1 fprop  M=131072 N=192 K=336", 8, 1, 128, 128, 336, 192, (1, 1, 1), (0, 0, 0), True),
    ("3D b3 3x3x3 fprop M=588 N=32 K=3456", 1, 3, 14, 14, 128, 32, (3, 3, 3), (1, 1, 1), True),
    ("3D b2 1x1 fprop  M=2352 N=128 K=480", 1, 3, 28, 28, 480, 128, (1, 1, 1), (0, 0, 0), True),
]


def main():
    dev = ops.device()
    host = np.zeros(WGS * 10, dtype=np.uint64)
    other = torch.empty(64 << 20, dtype=torch.uint8, device=dev)        # something else runs between two launches
    for (name, N, D, H, W, Cin, Cout, K, pad, stats) in SHAPES:
        x = ops.Act.alloc(N, D, H, W, Cin, 0); x.buf.normal_()
        y = ops.Act.alloc(N, D, H, W, Cout, 0)
        T = K[0] * K[1] * K[2]
        w = (torch.randn(Cout * T * Cin, device=dev) * 0.05).to(torch.bfloat16)
        d = ops.conv_desc(x, ctypes.c_void_p(w.data_ptr()), y, K, (1, 1, 1), pad)
        if stats:
            slots = 32
            part = torch.zeros(slots * 2 * y.ld, dtype=torch.float32, device=dev)
            shift = torch.zeros(y.ld, dtype=torch.float32, device=dev)
            d.stats_partial, d.stats_shift, d.stats_slots = part.data_ptr(), shift.data_ptr(), slots
        kname = ops.conv_kernel_name(d, 0)
        for _ in range(3):
            ops.conv_fprop(d)
            other.zero_()
        torch.cuda.synchronize()
        lib.hdu_timeline_read(host.ctypes.data, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_fprop(d); e1.record()
        torch.cuda.synchronize()
        lib.hdu_timeline_read(host.ctypes.data, 1)
        t = host.reshape(WGS, 10).astype(np.int64)
        live = t[:, 0] != 0
        t = t[live]
        n = len(t)
        if n == 0:
            print("\n%s  -> %s: no stamps (this kernel family is not instrumented)" % (name, kname))
            continue
        w0 = (t[:, 8] - t[:, 8].min()) * 10.0 / 1e3        # us
        w1 = (t[:, 9] - t[:, 8].min()) * 10.0 / 1e3
        last = t[:, 6] != 0                                    # ran the epilogue (the tile's last split, or unsplit)
        print("\n%s  -> %s" % (name, kname))
        print("  workgroups %d (with epilogue %d); launch (events) %.1f us; first entry -> last exit %.1f us" %
              (n, int(last.sum()), e0.elapsed_time(e1) * 1e3, w1.max()))
        print("  entry time  us: min %.1f  median %.1f  max %.1f   |  exit time us: min %.1f  median %.1f  max %.1f" %
              (w0.min(), np.median(w0), w0.max(), w1.min(), np.median(w1), w1.max()))
        wall = (t[:, 9] - t[:, 8]) * 10.0                      # ns inside the kernel
        endc = np.where(last, t[:, 6], t[:, 5])
        cyc = endc - t[:, 0]
        ghz = np.median(cyc[wall > 0] / wall[wall > 0])
        seg = ["per-row state", "prologue issue", "first tile lands", "K loop", "split-K combine", "epilogue"]
        line = "  median us per segment (clock %.2f GHz):" % ghz
        for i, s in enumerate(seg):
            sel = last if i == 5 else np.ones(n, bool)
            dt = (t[sel, i + 1] - t[sel, i]) / ghz / 1e3
            line += "  %s %.2f" % (s, np.median(dt))
        print(line)
        st = last & (t[:, 7] != 0)
        if st.sum():
            print("    epilogue split: accumulators -> LDS staging (incl. both barriers) %.2f us, stores (+ statistics) %.2f us" %
                  (np.median((t[st, 7] - t[st, 5]) / ghz / 1e3), np.median((t[st, 6] - t[st, 7]) / ghz / 1e3)))
        # first-round workgroups (cold instruction / scalar caches on their CU) against later rounds
        early, late = w0 < 2.0, w0 > 4.0
        if late.sum() >= 16:
            for tag, grp in (("entered < 2 us", early), ("entered > 4 us", late)):
                line = "    %s (%d):" % (tag, int(grp.sum()))
                for i, sname in enumerate(seg):
                    sel = grp & (last if i == 5 else np.ones(n, bool))
                    if sel.sum():
                        line += "  %s %.2f" % (sname, np.median((t[sel, i + 1] - t[sel, i]) / ghz / 1e3))
                print(line)
        print("  in-kernel time per workgroup us: median %.2f  max %.2f" % (np.median(wall) / 1e3, wall.max() / 1e3))


if __name__ == "__main__":
    main()
