#!/bin/bash
# round-2 GPU call G: split-K geometry sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( for cfg in 3dpart 2d; do
    run "A=0" $cfg
    run "HDU_SPLITK_TARGET=384" $cfg
    run "HDU_SPLITK_TARGET=512" $cfg
    run "HDU_SPLITK_MIN_STEPS=2" $cfg
    run "HDU_SPLITK_MIN_STEPS=4" $cfg
    run "HDU_SPLITK_TARGET=512 HDU_SPLITK_MIN_STEPS=2" $cfg
    run "HDU_HALO_MIN_TILES=512" $cfg
    run "A=0" $cfg
  done ) > gpurun_out/g_sweep.log 2>&1
cat gpurun_out/g_sweep.log
