#!/bin/bash
# round-2 GPU calls N, S: after the prologue / epilogue rework of the implicit GEMM -- kernel parity tests, timeline probe, benches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/${TAG:-n}_kernels.log 2>&1; tail -3 gpurun_out/${TAG:-n}_kernels.log
timeout 300 python tools/timeline_probe.py > gpurun_out/${TAG:-n}_timeline.txt 2>&1
grep -A4 "b4 3x3 fprop\|b2 3x3 dgrad\|b2 1x1 fprop\|3D b3" gpurun_out/${TAG:-n}_timeline.txt | cut -c1-220
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d; run "A=0" 3dpart; run "A=0" end2end; run "A=0" 2d ) > gpurun_out/${TAG:-n}_ab.log 2>&1
cat gpurun_out/${TAG:-n}_ab.log
