#!/bin/bash
# the GPU tier as the driver runs it (+ durations), figures of the parity tests into gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/bf16_parity_figures.txt
start=$(date +%s)
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=30 2>&1 | tail -60 > gpurun_out/tier_gpu.txt
echo "wall $(( $(date +%s) - start )) s" >> gpurun_out/tier_gpu.txt
tail -50 gpurun_out/tier_gpu.txt
