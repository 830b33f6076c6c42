"""TEST INFRASTRUCTURE ONLY (tests/test_bench_flow_gloo.py): bench.py's control flow on CPU -- the x86 emulator build of the
kernels, gloo instead of RCCL, reduced-depth nets -- so that the multi-rank sequence of collectives of that script is checked
without a multi-GPU node.  Never a measurement; the JSON line says so.  bench.py itself has no switch that binds anything but
the gfx950 library: this wrapper sets its DRYRUN flag and hands it the emulator binder."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import emu_bind  # noqa: E402

bench.DRYRUN = True
bench._dryrun_bind = emu_bind.use_emulator
bench.main()
