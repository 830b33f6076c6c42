#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "test_conv_stem or test_conv_wgrad" 2>&1 | tail -3 )
( timeout 600 python tools/bench_halo_wide.py stem ) 2>&1 | tee gpurun_out/r05c6_stem.txt
