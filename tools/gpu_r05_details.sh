#!/bin/bash
# per-kernel tables (library launch profiler) of the workloads named in $1 (default: shard3d 2d 3dpart end2end)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in ${1:-shard3d 2d 3dpart end2end}; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --extras none > gpurun_out/details_$cfg.json 2> gpurun_out/details_$cfg.err
  cp gpurun_out/bench_details.json gpurun_out/bench_details_$cfg.json
  python tools/show_details.py gpurun_out/bench_details_$cfg.json ${2:-16}
done
