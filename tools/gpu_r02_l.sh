#!/bin/bash
# round-2 GPU call L: per-GEMM-shape table of the 2D and 3dpart steps (which launches are short-K / small-N)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
HDU_BENCH_VERBOSE=2 python bench.py --config 2d --steps 3 --warmup 2 --no-cpu-baseline --extras none > gpurun_out/l_2d.json 2> gpurun_out/l_2d_shapes.txt
HDU_BENCH_VERBOSE=2 python bench.py --config 3dpart --steps 3 --warmup 2 --no-cpu-baseline --extras none > gpurun_out/l_3d.json 2> gpurun_out/l_3dpart_shapes.txt
wc -l gpurun_out/l_*shapes.txt
