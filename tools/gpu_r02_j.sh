#!/bin/bash
# round-2 GPU call J: knob probes on the large-grid GEMM (deep ring everywhere, tile widths, XCD tile order)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d
  run "HDU_DMA_STAGES=6" 2d
  run "HDU_MAX_BN=96" 2d
  run "HDU_XCD_SWIZZLE=0" 2d
  run "A=0" 2d
  run "A=0" end2end
  run "HDU_DMA_STAGES=6" end2end
  run "A=0" 3dpart
  run "HDU_DMA_STAGES=6" 3dpart ) > gpurun_out/j_ab.log 2>&1
cat gpurun_out/j_ab.log
