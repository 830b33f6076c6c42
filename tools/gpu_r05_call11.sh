#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_STEPS=20 tools/gpu_ab.sh r05_deep16 2 "2d end2end 3dpart shard3d" "no_rule=HDU_DEBUG_FLAGS=1024" "rule="
