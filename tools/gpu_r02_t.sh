#!/bin/bash
# round-2 GPU call T: same-box A/B of two builds of libhdu.so (tools/libhdu_prev.so = the previous commit)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
cp h-denseunet_amd/libhdu.so /tmp/libhdu_new.so
( run new 2d; run new 3dpart; run new end2end
  cp tools/libhdu_prev.so h-denseunet_amd/libhdu.so
  run prev 2d; run prev 3dpart; run prev end2end
  cp /tmp/libhdu_new.so h-denseunet_amd/libhdu.so
  run new 2d; run new 3dpart ) > gpurun_out/t_ab.log 2>&1
cat gpurun_out/t_ab.log
