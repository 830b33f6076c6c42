#!/bin/bash
# round-2 GPU call I: stride-2 stem data gradient on the MFMA path (parity classes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1 / $2" ; env $1 python bench.py --config $2 --steps 15 --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>&1 | grep -o '"ms_per_step": [0-9.]*' ; }
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "stride2 or dgrad" 2>&1 | tail -2
  run "HDU_STRIDE2_DGRAD=0" end2end
  run "A=0" end2end
  run "HDU_STRIDE2_DGRAD=0" end2end
  run "A=0" end2end
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bf16.py -m gpu -q -k "end2end" 2>&1 | tail -3 ) > gpurun_out/i_ab.log 2>&1
tools/gpu_profile.sh end2end_bf16 0 --config end2end --steps 10 --warmup 3
cat gpurun_out/i_ab.log; head -8 gpurun_out/prof_end2end_bf16/stats.csv | cut -c1-150
