#!/bin/bash
# ONE parametrised script for the same-box A/B runs of a gpurun call (boxes differ by several percent between calls, so
# every comparison is made inside one call, interleaved):
#   tools/gpu_ab.sh <tag> <rounds> "<configs>" "<label>=<ENV=VAL ENV=VAL ...>" ["<label>=<...>" ...]
#   tools/gpu_ab.sh r03_fuse 2 "2d 3dpart" "base=" "fuse2=HDU_FUSE_BN_BWD=2"
# A label may also name a library: "prev=LIB=tools/libhdu_prev.so" (copied over h-denseunet_amd/libhdu.so for that arm).
# Writes gpurun_out/ab_<tag>.txt: one line per (round, config, label) with ms_per_step.
tag=$1; rounds=$2; configs=$3; shift 3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_$tag.txt
: > "$out"
cp h-denseunet_amd/libhdu.so /tmp/libhdu_cur.so
for r in $(seq 1 "$rounds"); do
  for cfg in $configs; do
    for arm in "$@"; do
      label=${arm%%=*}; envs=${arm#*=}
      libsel=/tmp/libhdu_cur.so
      cleaned=""
      for kv in $envs; do
        if [ "${kv%%=*}" = LIB ]; then libsel=${kv#LIB=}; else cleaned="$cleaned $kv"; fi
      done
      cp "$libsel" h-denseunet_amd/libhdu.so
      ms=$(env $cleaned timeout 300 python bench.py --config "$cfg" --steps ${AB_STEPS:-20} --warmup 4 --no-cpu-baseline --no-roofline --extras none 2>>gpurun_out/ab_$tag.err | grep -o '"ms_per_step": [0-9.]*' | head -1)
      echo "round $r  $cfg  $label  ${ms:-FAILED}" | tee -a "$out"
    done
  done
done
cp /tmp/libhdu_cur.so h-denseunet_amd/libhdu.so
