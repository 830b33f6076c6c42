#!/bin/bash
# round 5, call 4: the new GPU parity cases (shard-shape bf16 training step, depth-halo form vs unsharded)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_parity.py -m gpu -x -q -s -k "3d-shard or depth_halo" 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r05c4_parity.log 2>&1
cut -c1-400 gpurun_out/r05c4_parity.log
