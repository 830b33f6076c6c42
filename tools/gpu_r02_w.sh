#!/bin/bash
# round-2 GPU call W: the driver's bench command on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
HDU_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cut -c1-1500 gpurun_out/final_bench.json
