#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_parity.py -m gpu -x -q --durations=6 -k "end2end-mid or (test_f32_absolute and 2d-denseunet) or (test_full_forward_parity_f32 and end2end)" 2>&1 | tail -14 ) | cut -c1-160
