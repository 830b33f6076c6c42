#!/bin/bash
# round-2 GPU call R: BN in the producer's epilogue (hdu_conv_desc.epi_*) -- parity (f32 + bf16, 3dpart) and A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv_fprop" > gpurun_out/r_kernels.log 2>&1; tail -2 gpurun_out/r_kernels.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bf16.py -m gpu -x -q -k "3dpart or dice" > gpurun_out/r_parity.log 2>&1; tail -4 gpurun_out/r_parity.log
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( run "A=0" 3dpart; run "HDU_FUSE_BN_EPILOGUE=0" 3dpart; run "A=0" 3dpart; run "HDU_FUSE_BN_EPILOGUE=0" 3dpart ) > gpurun_out/r_ab.log 2>&1
cat gpurun_out/r_ab.log
