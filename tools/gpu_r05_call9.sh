#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_STEPS=20 tools/gpu_ab.sh r05_p 3 "shard3d 2d end2end" "no_p=HDU_DEBUG_FLAGS=512" "with_p="
