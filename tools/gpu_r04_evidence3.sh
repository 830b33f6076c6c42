#!/bin/bash
# Round-4 closing evidence (one gpurun call) of the FINAL tree: the default bench line, rocprofv3 kernel statistics + step census of
# the three bf16 workloads, PMC passes (FETCH / WRITE / SQ) of each.  Outputs under gpurun_out/ev3_*; copied to profiles/r04_*.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/ev3_bench.json 2> gpurun_out/ev3_bench.err
cp gpurun_out/bench_details.json gpurun_out/ev3_bench_details.json 2>/dev/null
for cfg in 2d 3dpart end2end; do
  tools/gpu_profile.sh ev3_${cfg}_s30 0 --config $cfg --steps 30 --warmup 3
  tools/gpu_profile.sh ev3_${cfg}_s10 0 --config $cfg --steps 10 --warmup 3
  python tools/step_census.py gpurun_out/prof_ev3_${cfg}_s10/stats.csv 10 gpurun_out/prof_ev3_${cfg}_s30/stats.csv 30 gpurun_out/ev3_census_${cfg}.txt > /dev/null
  tools/gpu_profile.sh ev3_${cfg}_pmc 1 --config $cfg --steps 10 --warmup 2
done
ls gpurun_out | grep ev3_ | head -40
