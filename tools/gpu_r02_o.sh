#!/bin/bash
# round-2 GPU call O: 192-column tiles (HDU_MAX_BN=192) A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
HDU_MAX_BN=192 timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv" > gpurun_out/o_kernels.log 2>&1; tail -2 gpurun_out/o_kernels.log
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d; run "HDU_MAX_BN=192" 2d; run "A=0" 3dpart; run "HDU_MAX_BN=192" 3dpart; run "A=0" end2end; run "HDU_MAX_BN=192" end2end; run "A=0" 2d; run "HDU_MAX_BN=192" 2d ) > gpurun_out/o_ab.log 2>&1
cat gpurun_out/o_ab.log
