#!/bin/bash
# Builds libhdu.so (gfx950 product library) and, with "emu", the x86 emulator build of the same sources
# used by the CPU-only test tier.  Usage: ./build.sh [hip|emu|all]
# Every translation unit is compiled to its own object under build/ (in parallel, skipped when the object is newer than
# the source and every header), then linked: a one-file change rebuilds in the time of that file.
set -e
cd "$(dirname "$0")"
SRC=h-denseunet_amd/csrc
what=${1:-all}
UNITS="conv_igemm.hip conv_halo_wide.hip rowops.hip augment.hip hdu_core.cpp hdu_comm.cpp"
HDRS="$SRC/*.h include/hdu.h tests/hipemu/hipemu.h build.sh"
mkdir -p build/hip build/emu

stale() {   # stale <object> <source>: true when the object has to be rebuilt
  [ ! -f "$1" ] && return 0
  for f in $2 $HDRS; do [ "$f" -nt "$1" ] && return 0; done
  return 1
}

compile_all() {   # compile_all <hip|emu>
  local mode=$1 pids="" fail=0
  for u in $UNITS; do
    local o=build/$mode/${u%.*}.o
    if stale "$o" "$SRC/$u"; then
      if [ "$mode" = hip ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-c++20-extensions \
          -mllvm -amdgpu-mfma-vgpr-form=1 -x hip -c "$SRC/$u" -o "$o" &
      else
        /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -DHDU_EMU -Itests/hipemu -pthread -Wno-c++20-extensions \
          -x c++ -c "$SRC/$u" -o "$o" &
      fi
      pids="$pids $!"
    fi
  done
  if [ "$mode" = emu ] && stale build/emu/hipemu.o tests/hipemu/hipemu.cpp; then
    /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -DHDU_EMU -Itests/hipemu -pthread -c tests/hipemu/hipemu.cpp -o build/emu/hipemu.o &
    pids="$pids $!"
  fi
  for p in $pids; do wait "$p" || fail=1; done
  return $fail
}

if [ "$what" = hip ] || [ "$what" = all ]; then
  compile_all hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(for u in $UNITS; do echo build/hip/${u%.*}.o; done) -ldl \
    -o h-denseunet_amd/libhdu.so
  echo "built h-denseunet_amd/libhdu.so"
fi
if [ "$what" = emu ] || [ "$what" = all ]; then
  compile_all emu
  /opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -pthread $(for u in $UNITS; do echo build/emu/${u%.*}.o; done) build/emu/hipemu.o -ldl \
    -o tests/hipemu/libhdu_emu.so
  echo "built tests/hipemu/libhdu_emu.so"
fi
