#!/bin/bash
# round 5, call 5: the per-plane halo-tile filter gradient on hardware (parity) and the whole 512^3 volume at world 1 again
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "large_tensor_path or test_conv_wgrad or halo_filter_gradient" 2>&1 | tail -3 ) 
timeout 900 python bench.py --config shard3d --cols 512 --steps 5 --warmup 2 --no-cpu-baseline --extras none > gpurun_out/r05_full512_world1.json 2> gpurun_out/r05_full512_world1.err
cp gpurun_out/bench_details.json gpurun_out/r05_full512_world1_details.json
cut -c1-900 gpurun_out/r05_full512_world1.json; tail -3 gpurun_out/r05_full512_world1.err
python tools/show_details.py gpurun_out/bench_details.json 12
