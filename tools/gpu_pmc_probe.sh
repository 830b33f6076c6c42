#!/bin/bash
# developer: one rocprofv3 --pmc pass per argument (a quoted counter list) over a short 2D bench run; prints the per-kernel averages
# of the kernels matching $KFILTER.   tools/gpu_pmc_probe.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "$@"; do
  i=$((i+1))
  rm -rf /tmp/rp_probe_$i
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/rp_probe_$i -o res -- python "$ROOT/bench.py" --config ${CFG:-2d} --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --extras none > /dev/null 2> /tmp/rp_probe_$i.err
  db=$(find /tmp/rp_probe_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python "$ROOT/tools/rocpd_pmc.py" "$db" 60 | grep -A12 -E "${KFILTER:-wgrad_halo}" | head -${LINES_OUT:-40}; else echo "pass $i ($ctr): no db"; tail -3 /tmp/rp_probe_$i.err; fi
done > "$ROOT/gpurun_out/pmc_probe.txt" 2>&1
cat "$ROOT/gpurun_out/pmc_probe.txt"
