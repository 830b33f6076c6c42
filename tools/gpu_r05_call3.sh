#!/bin/bash
# round 5, call 3: model-level GPU parity with the halo-wide kernel in the launch lists + same-box A/B of the four workloads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2d-denseunet-8-512 or shard_shape or 3dpart or end2end" 2>&1 | tail -6 ) > gpurun_out/r05c3_parity.log 2>&1
cat gpurun_out/r05c3_parity.log
AB_STEPS=20 tools/gpu_ab.sh r05_halo_wide 2 "2d 3dpart end2end shard3d" "im2col=HDU_HALO_WIDE=1" "halo_wide="
