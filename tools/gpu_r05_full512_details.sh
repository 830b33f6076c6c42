#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --config shard3d --cols 512 --steps 3 --warmup 1 --no-cpu-baseline --extras none > gpurun_out/r05_full512_details.json 2> gpurun_out/r05_full512_details.err
python tools/show_details.py gpurun_out/bench_details.json 22
