#!/bin/bash
# Developer script for ONE gpurun call: rocprofv3 kernel statistics (+ optional PMC passes) of a bench.py workload.
#   tools/gpu_profile.sh <tag> <pmc:0|1> <bench args...>
# writes gpurun_out/prof_<tag>/{stats.csv,pmc_*.txt,bench.json}; copy what is judged into profiles/ afterwards.
# rocprofv3 writes a rocpd SQLite database; tools/rocpd_stats.py / rocpd_pmc.py summarise it.
tag=$1; pmc=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=$ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run_prof() {   # $1 = subdir, rest = rocprofv3 options
  sub=$1; shift
  rm -rf "/tmp/rp_${tag}_$sub"
  rocprofv3 "$@" -d "/tmp/rp_${tag}_$sub" -o res -- python "$ROOT/bench.py" "${BENCH_ARGS[@]}" > "$out/bench_$sub.json" 2> "$out/bench_$sub.err"
  find "/tmp/rp_${tag}_$sub" -name "*.db" | head -1
}
BENCH_ARGS=("$@" --no-cpu-baseline --extras none)
db=$(run_prof stats --kernel-trace --stats -f rocpd csv)
find /tmp/rp_${tag}_stats -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats.csv" \;
if [ -n "$db" ]; then python "$ROOT/tools/rocpd_stats.py" "$db" "$out/stats.csv" > /dev/null; else echo "no db for stats" >&2; ls -R /tmp/rp_${tag}_stats | head -30 >&2; fi
if [ "$pmc" = 1 ]; then
  BENCH_ARGS=("$@" --no-cpu-baseline --extras none --no-roofline --steps 3 --warmup 1)
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT"; do
    name=$(echo $ctr | cut -d' ' -f1)
    db=$(run_prof "pmc_$name" --kernel-trace --pmc $ctr)
    if [ -n "$db" ]; then python "$ROOT/tools/rocpd_pmc.py" "$db" 40 > "$out/pmc_$name.txt"; fi
  done
fi
cd "$ROOT"
