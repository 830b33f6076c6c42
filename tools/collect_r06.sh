#!/bin/bash
# after tools/gpu_r06_final.sh: gpurun_out/ -> profiles/r06_* (the files the line and DESIGN section 5 quote)
cd "$(dirname "$0")/.."
cp gpurun_out/bf16_parity_figures.txt profiles/r06_bf16_parity_figures.txt
cp gpurun_out/tier_gpu.txt profiles/r06_gpu_test_tier.log
cp gpurun_out/ev_bench.json profiles/r06_bench_all_workloads.json
cp gpurun_out/ev_bench_details.json profiles/r06_bench_details.json
cp gpurun_out/ev_full512.json profiles/r06_full_512cubed_world1.json
cp gpurun_out/ev_full512_details.json profiles/r06_full_512cubed_world1_details.json
for cfg in 2d 3dpart end2end shard3d; do
  s=s30; [ $cfg = shard3d ] && s=s10
  cp gpurun_out/prof_ev_${cfg}_$s/rocprofv3_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_${cfg}_bf16.csv 2>/dev/null \
    || cp gpurun_out/prof_ev_${cfg}_$s/stats.csv profiles/r06_rocprofv3_kernel_stats_${cfg}_bf16.csv
  cp gpurun_out/ev_census_${cfg}.txt profiles/r06_step_census_${cfg}_bf16.txt 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/prof_ev_${cfg}_pmc/pmc_$ctr.txt profiles/r06_pmc_${ctr}_${cfg}_bf16.txt; done
  cp gpurun_out/prof_ev_${cfg}_pmc/pmc_SQ_WAVE_CYCLES.txt profiles/r06_pmc_SQ_counters_${cfg}_bf16.txt
done
ls -la profiles | grep r06 | wc -l
