#!/bin/bash
# round-2 GPU call F: where does the fused BN backward pay?  pixel-count threshold sweep + kernel statistics
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( for cfg in 2d end2end 3dpart; do
    run "HDU_FUSE_BN_BWD=0" $cfg
    run "HDU_FUSE_BN_BWD_MAXM=2400" $cfg
    run "HDU_FUSE_BN_BWD_MAXM=10000" $cfg
    run "HDU_FUSE_BN_BWD_MAXM=40000" $cfg
    run "A=0" $cfg
  done ) > gpurun_out/f_ab.log 2>&1
tools/gpu_profile.sh 2d_fused 0 --config 2d --steps 10 --warmup 3
HDU_FUSE_BN_BWD=0 tools/gpu_profile.sh 2d_unfused 0 --config 2d --steps 10 --warmup 3
( timeout 600 python -m pytest tests/test_augment.py -m gpu -q 2>&1 | tail -3 ) > gpurun_out/f_augment.log
cat gpurun_out/f_ab.log gpurun_out/f_augment.log
