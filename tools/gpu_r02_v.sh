#!/bin/bash
# round-2 GPU call V: finalize + fold-next in one launch (HDU_FOLD_NEXT) -- unit test, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "fold_next or epilogue_statistics" > gpurun_out/v_kernels.log 2>&1; tail -2 gpurun_out/v_kernels.log
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 2d; run "HDU_FOLD_NEXT=0" 2d; run "A=0" 3dpart; run "HDU_FOLD_NEXT=0" 3dpart; run "A=0" end2end; run "HDU_FOLD_NEXT=0" end2end; run "A=0" 2d; run "HDU_FOLD_NEXT=0" 2d ) > gpurun_out/v_ab.log 2>&1
cat gpurun_out/v_ab.log
