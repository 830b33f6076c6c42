#!/bin/bash
# Round-4 closing evidence (one gpurun call): the default bench line of the final tree (all extra workloads incl. the float32
# split-bf16 mode), and rocprofv3 kernel statistics + step census + PMC passes of the workload that changed last (end2end).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/ev2_bench.json 2> gpurun_out/ev2_bench.err
cp gpurun_out/bench_details.json gpurun_out/ev2_bench_details.json 2>/dev/null
for cfg in end2end; do
  tools/gpu_profile.sh ev2_${cfg}_s30 0 --config $cfg --steps 30 --warmup 3
  tools/gpu_profile.sh ev2_${cfg}_s10 0 --config $cfg --steps 10 --warmup 3
  python tools/step_census.py gpurun_out/prof_ev2_${cfg}_s10/stats.csv 10 gpurun_out/prof_ev2_${cfg}_s30/stats.csv 30 gpurun_out/ev2_census_${cfg}.txt > /dev/null
  tools/gpu_profile.sh ev2_${cfg}_pmc 1 --config $cfg --steps 10 --warmup 2
done
ls gpurun_out | grep ev2_ | head -40
head -c 1500 gpurun_out/ev2_bench.json
