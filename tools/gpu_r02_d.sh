#!/bin/bash
# round-2 GPU call D: remaining parity cases + N1 on hardware + filter-gradient plan sweep (W5)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q -s -k "3dpart or 3d" 2>&1 | grep "^\[\|recipe\|passed\|failed\|^E" | cut -c1-900 > gpurun_out/d_bf16_parity.log
( timeout 600 python -m pytest tests/test_sliding_window.py -m gpu -q -s 2>&1 | tail -6 ) > gpurun_out/d_sliding.log
run() { echo "== $1" ; env $1 python bench.py --config 2d --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0"
  run "HDU_XCD_SWIZZLE=3"
  run "HDU_BATCH_WGRAD_TARGET=384"
  run "HDU_BATCH_WGRAD_TARGET=192"
  run "HDU_BATCH_WGRAD_TARGET=96"
  run "HDU_BATCH_WGRAD_TARGET=192 HDU_XCD_SWIZZLE=3"
  run "HDU_BATCH_WGRAD_TARGET=96 HDU_XCD_SWIZZLE=3"
  run "HDU_BATCH_WGRAD_TARGET=192 HDU_WGRAD_MIN_STEPS=16"
  run "HDU_HALO_TARGET=128"
  run "HDU_HALO_TARGET=128 HDU_BATCH_WGRAD_TARGET=192"
  run "A=0" ) > gpurun_out/d_wgrad_sweep.log 2>&1
cat gpurun_out/d_bf16_parity.log gpurun_out/d_sliding.log gpurun_out/d_wgrad_sweep.log
