#!/bin/bash
# round-2 GPU call E: fused BN backward in the data-gradient epilogue -- kernel tests, parity, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/e_kernels.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/e_parity_f32.log
timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q -s 2>&1 | grep "^\[\|recipe\|passed\|failed\|^E" | cut -c1-900 > gpurun_out/e_bf16_parity.log
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( for cfg in 2d 3dpart end2end; do
    run "HDU_FUSE_BN_BWD=0" $cfg
    run "A=0" $cfg
  done
  run "HDU_FUSE_BN_BWD=0" 2d
  run "A=0" 2d ) > gpurun_out/e_ab.log 2>&1
cat gpurun_out/e_kernels.log gpurun_out/e_parity_f32.log; grep "passed\|failed\|^E" gpurun_out/e_bf16_parity.log; cat gpurun_out/e_ab.log
