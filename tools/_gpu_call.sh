cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/c14_tests.log 2>&1
AB_STEPS=15 tools/gpu_ab.sh c14 2 "2d 3dpart end2end" "prev=LIB=tools/libhdu_prev4.so" "slots_at_once=" > /dev/null 2>&1
cat gpurun_out/c14_tests.log gpurun_out/ab_c14.txt
