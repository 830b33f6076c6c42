// probe: empirical lane/element map of ds_read_b64_tr_b16 on gfx950 (developer tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // canonical addressing: 16-lane group g reads a 4x16 row-major tile at g*64; lane i points at row i>>2, cols (i&3)*4
  const int i = lane & 15, g = lane >> 4;
  unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
  unsigned a = (g * 64 + (i >> 2) * 16 + (i & 3) * 4) * 2;
  typedef unsigned short v4 __attribute__((ext_vector_type(4)));
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a + base));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(r >> (16 * j));
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
