#!/bin/bash
# round-2 GPU call Z2: the GPU tests not yet re-run on the buffer-DMA tree (the others: calls X, Y)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 330 python -m pytest tests -m gpu -q -x --ignore=tests/test_kernels.py -k "not forward_parity and not 2d-2x512 and not (bf16_train_step and 3dpart)" 2>&1 | tail -6 ) > gpurun_out/z2_gputests.log 2>&1
cat gpurun_out/z2_gputests.log
