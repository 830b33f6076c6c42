#!/bin/bash
# Round-6 closing evidence, ONE gpurun call of the FINAL tree: per workload the PMC passes (FETCH / WRITE / SQ),
# rocprofv3 kernel statistics and the per-step census; THEN the default bench line, which reads this call's FETCH / WRITE summaries
# for `roofline.traffic` (they are copied into profiles/ on the box first, and into profiles/ of the repo afterwards -- same files).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in 2d 3dpart end2end; do
  tools/gpu_profile.sh ev_${cfg}_s30 0 --config $cfg --steps 30 --warmup 3
  tools/gpu_profile.sh ev_${cfg}_s10 0 --config $cfg --steps 10 --warmup 3
  python tools/step_census.py gpurun_out/prof_ev_${cfg}_s10/stats.csv 10 gpurun_out/prof_ev_${cfg}_s30/stats.csv 30 gpurun_out/ev_census_${cfg}.txt > /dev/null
  tools/gpu_profile.sh ev_${cfg}_pmc 1 --config $cfg --steps 10 --warmup 2
done
tools/gpu_profile.sh ev_shard3d_s10 0 --config shard3d --steps 10 --warmup 2
tools/gpu_profile.sh ev_shard3d_pmc 1 --config shard3d --steps 4 --warmup 1
for cfg in 2d 3dpart end2end shard3d; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    cp gpurun_out/prof_ev_${cfg}_pmc/pmc_$ctr.txt profiles/r06_pmc_${ctr}_${cfg}_bf16.txt 2>/dev/null
  done
done
timeout 1500 python bench.py > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err
cp gpurun_out/bench_details.json gpurun_out/ev_bench_details.json 2>/dev/null
# the N = 1 denominator of configs[4]: the whole 512 x 512 x 512 volume on ONE GPU (strong-scaling reference of bench.py --gpus N)
timeout 900 python bench.py --config shard3d --cols 512 --steps 5 --warmup 1 --no-cpu-baseline --extras none > gpurun_out/ev_full512.json 2> gpurun_out/ev_full512.err
cp gpurun_out/bench_details.json gpurun_out/ev_full512_details.json 2>/dev/null
ls gpurun_out | grep "ev_" | head -40
wc -c gpurun_out/ev_bench.json; head -c 600 gpurun_out/ev_bench.json; tail -3 gpurun_out/ev_bench.err
