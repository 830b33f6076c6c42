#!/bin/bash
# round-2 GPU call A: bf16 full-size parity tests, the whole-metric bench line, rocprofv3 stats + PMC of 3dpart and 2d
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q -s -x 2>&1 | tail -60 ) > gpurun_out/a_bf16_parity.log
HDU_BENCH_VERBOSE=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
tools/gpu_profile.sh 3dpart_bf16 1 --config 3dpart --steps 10 --warmup 3
tools/gpu_profile.sh 2d_bf16 0 --config 2d --steps 10 --warmup 3
tail -5 gpurun_out/a_bf16_parity.log; cut -c1-600 gpurun_out/a_bench.json
