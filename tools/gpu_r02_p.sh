#!/bin/bash
# round-2 GPU call P: shallower ring (2 workgroups / CU on small grids) x split-K target
# (HDU_RING_STAGES was an experiment knob of that day: the result is the default now and the knob is gone)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( for c in 3dpart 2d; do
  run "A=0" $c
  run "HDU_RING_STAGES=4" $c
  run "HDU_RING_STAGES=4 HDU_SPLITK_TARGET=512" $c
  run "HDU_RING_STAGES=3" $c
  run "HDU_RING_STAGES=3 HDU_SPLITK_TARGET=512" $c
  run "HDU_RING_STAGES=3 HDU_SPLITK_TARGET=512 HDU_SPLITK_MIN_STEPS=3" $c
  run "A=0" $c
done ) > gpurun_out/p_ab.log 2>&1
cat gpurun_out/p_ab.log
