#!/bin/bash
# the sharded (depth-halo) launch list on ONE GPU at the configs[4] shard shape, beside the unsharded form (same box)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for arm in 0 1 0 1; do
  ms=$(HDU_FORCE_DEPTH_HALO=$arm timeout 300 python bench.py --config shard3d --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --extras none 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"peak_hbm_gib": [0-9.]*' | tr '\n' ' ')
  echo "HDU_FORCE_DEPTH_HALO=$arm  $ms"
done | tee gpurun_out/r05_shard_form_world1_timing.txt
