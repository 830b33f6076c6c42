#!/bin/bash
# round-2 GPU call U: grid size of the row / reduction kernels (BN chain), 2D
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d
  run "HDU_ROW_WGS=1024" 2d
  run "HDU_ROW_WGS=4096" 2d
  run "HDU_ROW_WGS=8192" 2d
  run "HDU_RED_WGS=256" 2d
  run "HDU_RED_WGS=1024" 2d
  run "HDU_RED_WGS=2048" 2d
  run "A=0" 2d ) > gpurun_out/u_ab.log 2>&1
cat gpurun_out/u_ab.log
