cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2 ) > gpurun_out/c5_tests.log 2>&1
AB_STEPS=15 tools/gpu_ab.sh c5 2 "2d 3dpart end2end" "prev=LIB=tools/libhdu_prev.so" "agpr=LIB=tools/libhdu_agpr.so" "vgprform=" > /dev/null 2>&1
cat gpurun_out/c5_tests.log gpurun_out/ab_c5.txt
