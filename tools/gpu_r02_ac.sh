#!/bin/bash
# round-2 GPU call AC (last): software-pipelined split-K combine -- split-K tests + same-box A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_kernels.py -m gpu -x -q -k "splitk" > gpurun_out/ac_kernels.log 2>&1; tail -1 gpurun_out/ac_kernels.log
run() { echo "== $1 / $2" ; timeout 60 python bench.py --config $2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
cp h-denseunet_amd/libhdu.so /tmp/libhdu_new.so
( run new 3dpart; run new end2end; run new 2d
  cp tools/libhdu_prev.so h-denseunet_amd/libhdu.so
  run prev 3dpart; run prev end2end; run prev 2d
  cp /tmp/libhdu_new.so h-denseunet_amd/libhdu.so ) > gpurun_out/ac_ab.log 2>&1
cat gpurun_out/ac_ab.log
