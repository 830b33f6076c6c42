#!/bin/bash
# round-2 GPU call C: split-K / batched fold / ring-for-all-small-grids -- kernel tests, parity, A/B bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/c_kernels.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/c_parity_f32.log
timeout 1500 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q -s 2>&1 | grep -v "^$" | grep "^\[\|recipe\|passed\|failed\|Error\|assert" | cut -c1-900 > gpurun_out/c_bf16_parity.log
OLD="HDU_SPLITK=1 HDU_RING_MIN_K=1024 HDU_HALO_MIN_TILES=1 HDU_BATCH_FOLD=0"
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"value": [0-9.]*, "unit": "slices/s", "n_gpus": 1, "steps": 15, "warmup": 4, "ms_per_step": [0-9.]*' ; }
( for cfg in 2d 3dpart end2end; do
    run "$OLD" $cfg
    run "A=0" $cfg
    run "HDU_SPLITK=1" $cfg
    run "HDU_RING_MIN_K=1024" $cfg
    run "A=0" $cfg
  done ) > gpurun_out/c_ab.log 2>&1
cat gpurun_out/c_kernels.log gpurun_out/c_parity_f32.log; tail -30 gpurun_out/c_bf16_parity.log; cat gpurun_out/c_ab.log
