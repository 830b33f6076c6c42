cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB_STEPS=15 tools/gpu_ab.sh c16 2 "2d 3dpart end2end" "prev5=LIB=tools/libhdu_prev5.so" "nct3=" > /dev/null 2>&1
cat gpurun_out/ab_c16.txt
