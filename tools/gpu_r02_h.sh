#!/bin/bash
# round-2 GPU call H: fused BN backward with the u / old-gradient loads issued before the staging pass
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "HDU_FUSE_BN_BWD=0" 2d
  run "HDU_FUSE_BN_BWD=1" 2d
  run "HDU_FUSE_BN_BWD=2" 2d
  run "HDU_FUSE_BN_BWD=0" 2d
  run "HDU_FUSE_BN_BWD=2" 2d
  run "HDU_FUSE_BN_BWD=0" end2end
  run "HDU_FUSE_BN_BWD=1" end2end
  run "HDU_FUSE_BN_BWD=2" end2end
  run "HDU_FUSE_BN_BWD=1" 3dpart
  run "HDU_FUSE_BN_BWD=2" 3dpart ) > gpurun_out/h_ab.log 2>&1
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "fused_bn or splitk" 2>&1 | tail -2 ) >> gpurun_out/h_ab.log
cat gpurun_out/h_ab.log
