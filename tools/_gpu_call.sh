cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/bf16_parity_figures.txt
( time timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/c2_tests.log 2>&1
AB_STEPS=15 tools/gpu_ab.sh c2 2 "2d 3dpart end2end" "base=" "fuse2=HDU_FUSE_BN_BWD=2" > /dev/null 2>&1
( time HDU_PARITY_MEASURE_ONLY=1 timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_parity.py -m gpu -x -q -s -k "bf16_train_step or f32_absolute" 2>&1 | grep -v "^recipe\|Warning\|warn" | tail -15 ) > gpurun_out/c2_parity.log 2>&1
cat gpurun_out/c2_tests.log gpurun_out/ab_c2.txt; tail -5 gpurun_out/c2_parity.log
