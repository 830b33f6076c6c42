#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in 384 512 768; do echo "== HDU_PW_BSTAT_WGS=$w"; HDU_PW_BSTAT_WGS=$w python tools/bench_pw_bstat.py 2>&1 | grep -v amdgpu | awk -F'|' '{print $1 "|" $4}'; done | tee gpurun_out/c16_pw_bstat_wgs.txt
AB_STEPS=30 tools/gpu_ab.sh r06_pwb6 2 "2d 3dpart end2end shard3d" "prev=LIB=tools/libhdu_prev.so" "new="
