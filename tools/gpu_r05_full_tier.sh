#!/bin/bash
# Round 5: the whole GPU test tier + smoke(), as the driver runs them at round end (timed: the driver's step limit is 1200 s)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r05_gpu_tests.log 2>&1
echo "pytest -m gpu: $(( $(date +%s) - t0 )) s" >> gpurun_out/r05_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" >> gpurun_out/r05_gpu_tests.log 2>&1
tail -25 gpurun_out/r05_gpu_tests.log | cut -c1-200
