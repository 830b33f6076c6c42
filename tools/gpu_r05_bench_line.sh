#!/bin/bash
# the default bench line of the final tree (reads profiles/r05_pmc_* of the evidence call for roofline.traffic)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
( time timeout 1500 python bench.py > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err ) 2>&1 | tail -3
cp gpurun_out/bench_details.json gpurun_out/ev_bench_details.json
wc -c gpurun_out/ev_bench.json; head -c 400 gpurun_out/ev_bench.json; echo; tail -2 gpurun_out/ev_bench.err
