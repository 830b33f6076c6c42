#!/bin/bash
# round-2 GPU call X: buffer-resource LDS-DMA in the implicit GEMM -- kernel + model parity, same-box A/B against the
# previous build (tools/libhdu_prev.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/x_kernels.log 2>&1; tail -2 gpurun_out/x_kernels.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_parity" > gpurun_out/x_parity.log 2>&1; tail -2 gpurun_out/x_parity.log
run() { echo "== $1 / $2" ; python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
cp h-denseunet_amd/libhdu.so /tmp/libhdu_new.so
( run new 2d; run new 3dpart; run new end2end
  cp tools/libhdu_prev.so h-denseunet_amd/libhdu.so
  run prev 2d; run prev 3dpart; run prev end2end
  cp /tmp/libhdu_new.so h-denseunet_amd/libhdu.so
  run new 2d; run new 3dpart ) > gpurun_out/x_ab.log 2>&1
cat gpurun_out/x_ab.log
