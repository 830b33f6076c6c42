#!/bin/bash
# Round-4 last refresh (one short gpurun call) after the halo split sizing / valid-depth change: filter-gradient kernel tests, the
# default bench line, rocprofv3 kernel statistics of the three bf16 workloads (30 steps).  Census / PMC: tools/gpu_r04_evidence3.sh.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "wgrad" > gpurun_out/ev4_kernels.log 2>&1
timeout 900 python bench.py > gpurun_out/ev4_bench.json 2> gpurun_out/ev4_bench.err
cp gpurun_out/bench_details.json gpurun_out/ev4_bench_details.json 2>/dev/null
for cfg in 2d 3dpart end2end; do
  tools/gpu_profile.sh ev4_${cfg}_s30 0 --config $cfg --steps 30 --warmup 3
done
tail -2 gpurun_out/ev4_kernels.log
head -c 300 gpurun_out/ev4_bench.json
