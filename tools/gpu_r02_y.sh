#!/bin/bash
# round-2 GPU call Y: buffer loads also in the general (up-sampling / > 32 taps) operand path -- tests + same-box A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/y_kernels.log 2>&1; tail -2 gpurun_out/y_kernels.log
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -x -q -k "2d or 3dpart" > gpurun_out/y_parity.log 2>&1; tail -2 gpurun_out/y_parity.log
run() { echo "== $1 / $2" ; python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
cp h-denseunet_amd/libhdu.so /tmp/libhdu_new.so
( run new 2d; run new 3dpart; run new end2end
  cp tools/libhdu_prev.so h-denseunet_amd/libhdu.so
  run prev 2d; run prev 3dpart; run prev end2end
  cp /tmp/libhdu_new.so h-denseunet_amd/libhdu.so ) > gpurun_out/y_ab.log 2>&1
cat gpurun_out/y_ab.log
