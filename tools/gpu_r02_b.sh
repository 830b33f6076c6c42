#!/bin/bash
# round-2 GPU call B: all four bf16 full-size parity cases (no -x: every case reports)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q -s 2>&1 | grep -v "^$" | grep "^\[\|recipe\|passed\|failed\|Error\|assert" | cut -c1-900 > gpurun_out/b_bf16_parity.log
tail -40 gpurun_out/b_bf16_parity.log
