#!/bin/bash
# one-call sweep of the host-side tuning knobs on the three 224 / 512 workloads (same box, ms per step; the first three lines are the
# default, repeated: the noise floor)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
out=gpurun_out/r06_knob_sweep.txt; : > $out
run() {   # run <label> <env assignments...>
  label=$1; shift
  for cfg in 2d 3dpart end2end; do
    ms=$(env "$@" timeout 300 python bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --extras none 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    printf "%-34s %-8s %s\n" "$label" $cfg "$ms" >> $out
  done
}
run default X=1; run default X=1; run default X=1
for v in 128 384 512 768; do run SPLITK_TARGET=$v HDU_SPLITK_TARGET=$v; done
for v in 2 8; do run SPLITK_MIN_STEPS=$v HDU_SPLITK_MIN_STEPS=$v; done
for v in 8192 32768 65536; do run BM64_MAX_M=$v HDU_BM64_MAX_M=$v; done
for v in 256 1024; do run ROW_WGS=$v HDU_ROW_WGS=$v; done
for v in 256 1024; do run RED_WGS=$v HDU_RED_WGS=$v; done
for v in 64 256; do run HALO_MIN_TILES=$v HDU_HALO_MIN_TILES=$v; done
for v in 8 32; do run BSUM_SLOTS=$v HDU_BSUM_SLOTS=$v; done
for v in 8 16; do run STATS_SLOTS=$v HDU_STATS_SLOTS=$v; done
for v in 256 1024; do run WGRAD_LAUNCH_WGS=$v HDU_WGRAD_LAUNCH_WGS=$v; done
for v in 2 6; do run DMA_STAGES=$v HDU_DMA_STAGES=$v; done
run default X=1
cat $out
