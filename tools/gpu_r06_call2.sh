#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_baseline_shapes.py -m gpu -q -s 2>&1 | tail -60 > gpurun_out/c2_baseline_shapes.txt
tail -40 gpurun_out/c2_baseline_shapes.txt
timeout 1500 python tests/bf16_storage_ablation.py > gpurun_out/c2_bf16_storage_ablation.txt 2> gpurun_out/c2_ablation.err
cat gpurun_out/c2_bf16_storage_ablation.txt; tail -5 gpurun_out/c2_ablation.err
