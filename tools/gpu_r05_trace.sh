#!/bin/bash
# launch-ordered trace of one eager step (library launch profiler): gpurun_out/step_trace_0.json
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
HDU_BENCH_TRACE=1 timeout 600 python bench.py --config ${1:-2d} --steps 5 --warmup 2 --no-cpu-baseline --extras none > /dev/null 2> gpurun_out/trace.err
python - <<'PY'
import json
t=json.load(open('gpurun_out/step_trace_0.json'))
rows=[r for r in t if r['k'].startswith('conv_halo_wide') or r['k'].startswith('conv_stem') or r['k'].startswith('conv_halo_fprop')]
for r in rows: print(r['k'][:52], r['us'], r['shape'])
PY
