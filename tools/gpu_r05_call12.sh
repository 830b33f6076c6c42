#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  HDU_FORCE_DP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$i HDU_COMM=torch timeout 300 python bench.py --steps 3 --warmup 1 --batch 2 --size 64 --no-cpu-baseline --no-roofline --extras none > gpurun_out/c12_$i.out 2> gpurun_out/c12_$i.err
  echo "torch-dp run $i rc=$?"
done
bash tools/gpu_r05_full_tier.sh
