#!/bin/bash
# round 5, call 7: stem kernels in the launch lists -- same-box A/B of the four workloads + shard parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_STEPS=20 tools/gpu_ab.sh r05_stem 2 "2d 3dpart end2end shard3d" "r4_kernels=HDU_HALO_WIDE=1" "r5="
