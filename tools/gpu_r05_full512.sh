#!/bin/bash
# round 5: BASELINE configs[4] on ONE GPU -- the whole 512 x 512 x 512 volume, world 1 (the N = 1 denominator of the 8-GPU strong-scaling target)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python bench.py --config shard3d --cols 512 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --extras none ) > gpurun_out/r05_full512_world1.json 2> gpurun_out/r05_full512_world1.err
cat gpurun_out/r05_full512_world1.json | cut -c1-1500; tail -5 gpurun_out/r05_full512_world1.err
