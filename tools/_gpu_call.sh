cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_STEPS=15 tools/gpu_ab.sh c24 2 "2d" "s32_b16=" "s16_b16=HDU_STATS_SLOTS=16" "s8_b16=HDU_STATS_SLOTS=8" "s32_b32=HDU_BSUM_SLOTS=32" "s32_b8=HDU_BSUM_SLOTS=8" > /dev/null 2>&1
cat gpurun_out/ab_c24.txt
