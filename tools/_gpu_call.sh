cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py tests/test_comm_abi.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/c7_tests.log 2>&1
AB_STEPS=15 tools/gpu_ab.sh c7 2 "2d 3dpart end2end" "base=" "ring64=HDU_DMA_STAGES=6 HDU_BM64_MAX_M=2000000000" "ring64n128=HDU_DMA_STAGES=6 HDU_BM64_MAX_M=2000000000 HDU_MAX_BN=128" > /dev/null 2>&1
cat gpurun_out/c7_tests.log gpurun_out/ab_c7.txt
