cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/c17_tests.log 2>&1
AB_STEPS=15 tools/gpu_ab.sh c17 2 "end2end 3dpart" "each=HDU_DEFER_BNB_FINALIZE=0" "deferred=" > /dev/null 2>&1
cat gpurun_out/c17_tests.log gpurun_out/ab_c17.txt
