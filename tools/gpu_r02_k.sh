#!/bin/bash
# round-2 GPU call K: 64-row tiles on large grids (3 workgroups / CU instead of 2)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d
  run "HDU_BM64_MAX_M=40000" 2d
  run "HDU_BM64_MAX_M=140000" 2d
  run "HDU_BM64_MAX_M=600000" 2d
  run "HDU_BM64_MAX_M=2000000000" 2d
  run "A=0" 2d
  run "HDU_BM64_MAX_M=2000000000" end2end
  run "HDU_BM64_MAX_M=2000000000" 3dpart ) > gpurun_out/k_ab.log 2>&1
cat gpurun_out/k_ab.log
