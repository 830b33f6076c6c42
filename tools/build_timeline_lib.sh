#!/bin/bash
# builds tools/libhdu_tl.so: the product sources with -DHDU_TIMELINE (s_memtime stamps inside the implicit-GEMM kernels),
# loaded only by tools/timeline_probe.py.  Git-ignored (*.so); travels to the GPU box with the gpurun snapshot.
cd "$(dirname "$0")/.."
SRC=h-denseunet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DHDU_TIMELINE -Wno-c++20-extensions \
  -x hip $SRC/conv_igemm.hip $SRC/conv_halo_wide.hip $SRC/rowops.hip $SRC/augment.hip -x hip $SRC/hdu_core.cpp $SRC/hdu_comm.cpp -ldl -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/libhdu_tl.so && echo "built tools/libhdu_tl.so"
