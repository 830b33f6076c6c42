#!/bin/bash
# Round-4 evidence run (one gpurun call): the default bench line, rocprofv3 kernel statistics + step census of the three bf16
# workloads, PMC passes (FETCH / WRITE / SQ) of the 2D step.  Outputs under gpurun_out/ev_*; copied to profiles/r04_* afterwards.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err
cp gpurun_out/bench_details.json gpurun_out/ev_bench_details.json 2>/dev/null
for cfg in 2d 3dpart end2end; do
  tools/gpu_profile.sh ev_${cfg}_s30 0 --config $cfg --steps 30 --warmup 3
  tools/gpu_profile.sh ev_${cfg}_s10 0 --config $cfg --steps 10 --warmup 3
  python tools/step_census.py gpurun_out/prof_ev_${cfg}_s10/stats.csv 10 gpurun_out/prof_ev_${cfg}_s30/stats.csv 30 gpurun_out/ev_census_${cfg}.txt > /dev/null
done
tools/gpu_profile.sh ev_2d_pmc 1 --config 2d --steps 10 --warmup 2
ls gpurun_out | grep ev_ | head -40
