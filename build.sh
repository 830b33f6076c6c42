#!/bin/bash
# Builds libhdu.so (gfx950 product library) and, with "emu", the x86 emulator build of the same sources
# used by the CPU-only test tier.  Usage: ./build.sh [hip|emu|all]
set -e
cd "$(dirname "$0")"
SRC=h-denseunet_amd/csrc
what=${1:-all}
if [ "$what" = hip ] || [ "$what" = all ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-c++20-extensions -mllvm -amdgpu-mfma-vgpr-form=1 \
    -x hip $SRC/conv_igemm.hip $SRC/rowops.hip $SRC/augment.hip -x hip $SRC/hdu_core.cpp $SRC/hdu_comm.cpp -ldl \
    -o h-denseunet_amd/libhdu.so
  echo "built h-denseunet_amd/libhdu.so"
fi
if [ "$what" = emu ] || [ "$what" = all ]; then
  /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -shared -DHDU_EMU -Itests/hipemu -pthread -Wno-c++20-extensions \
    -x c++ $SRC/conv_igemm.hip $SRC/rowops.hip $SRC/augment.hip $SRC/hdu_core.cpp $SRC/hdu_comm.cpp tests/hipemu/hipemu.cpp \
    -o tests/hipemu/libhdu_emu.so
  echo "built tests/hipemu/libhdu_emu.so"
fi
