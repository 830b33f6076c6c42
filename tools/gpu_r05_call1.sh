#!/bin/bash
# round 5, call 1: the halo-wide kernel on hardware -- parity cases, then the per-layer A/B against the im2col kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "halo_wide" 2>&1 | tail -5 ) > gpurun_out/r05c1_tests.log 2>&1
cat gpurun_out/r05c1_tests.log
( timeout 900 python tools/bench_halo_wide.py all ) > gpurun_out/r05c1_halo_wide_ab.txt 2>&1
cat gpurun_out/r05c1_halo_wide_ab.txt
