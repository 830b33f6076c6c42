#!/bin/bash
# round-2 GPU call Z: evidence refresh on the final tree -- smoke, the driver's bench command, rocprofv3 stats + PMC passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/final_smoke.log
HDU_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tools/gpu_profile.sh 2d_bf16 1 --config 2d --steps 10 --warmup 3
tools/gpu_profile.sh 3dpart_bf16 1 --config 3dpart --steps 10 --warmup 3
tools/gpu_profile.sh end2end_bf16 0 --config end2end --steps 10 --warmup 3
cat gpurun_out/final_smoke.log; cut -c1-300 gpurun_out/final_bench.json
