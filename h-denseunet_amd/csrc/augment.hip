// augment.hip -- training-sample assembly on the device.
//
// Replaces the host thread pool of train_2ddense.py:40-133 / train_hybrid.py:40-133: the pre-processed CT volumes and
// their label volumes stay resident in HBM, and ONE launch pair per batch crops, mean-subtracts, flips / rotates and
// resizes every sample straight into the model's input and label buffers -- no per-batch host work, no PCIe traffic.
// Interpolation = skimage.transform.resize as the reference calls it (train_2ddense.py:103-104): labels nearest
// (order 0, mode 'edge'), image bicubic (order 3: Catmull-Rom cubic convolution over the 4x4 neighbourhood anchored at
// floor(coordinate), mode 'constant' with cval 0, output clipped to the value range of the crop).
#include "hdu_host.h"

// crop-space source pixel of transformed-crop pixel (i, j): T[i][j] = crop[gi][gj] for the 8 cases of
// train_2ddense.py:73-101 (np.flipud / np.fliplr / np.rot90(.., axes=(1, 0)) on a square crop of side n)
__device__ __forceinline__ void aug_unflip(int flip, int n, int i, int j, int* gi, int* gj) {
  switch (flip) {
    case 1: *gi = n - 1 - i; *gj = j; break;                 // flipud
    case 2: *gi = i; *gj = n - 1 - j; break;                 // fliplr
    case 3: *gi = n - 1 - j; *gj = i; break;                 // rot90 k=1 axes=(1,0)
    case 4: *gi = j; *gj = n - 1 - i; break;                 // rot90 k=3 axes=(1,0)
    case 5: *gi = n - 1 - j; *gj = n - 1 - i; break;         // fliplr, then rot90 k=1 axes=(1,0)
    case 6: *gi = j; *gj = i; break;                         // fliplr, then rot90 k=3 axes=(1,0)
    case 7: *gi = n - 1 - i; *gj = n - 1 - j; break;         // flipud, then fliplr
    default: *gi = i; *gj = j; break;
  }
}

// value range of every sample's (mean-subtracted) crop: skimage clips the bicubic output to it
__global__ __launch_bounds__(256) void augment_minmax_kernel(const float* __restrict__ img, const hdu_aug_sample* __restrict__ smp,
                                                             int nslices, float mean, float* __restrict__ minmax) {
  __shared__ float smin[256], smax[256];
  const hdu_aug_sample s = smp[blockIdx.x];
  const float* vol = img + s.img_off;
  float lo = 3.0e38f, hi = -3.0e38f;
  const long long total = (long long)s.crop * s.crop * nslices;
  for (long long q = threadIdx.x; q < total; q += 256) {
    const int k = (int)(q % nslices);
    const long long pix = q / nslices;
    const int j = (int)(pix % s.crop), i = (int)(pix / s.crop);
    const float v = vol[((long long)(s.a0 + i) * s.vcols + (s.b0 + j)) * s.vslices + s.c0 + k] - mean;
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      smin[threadIdx.x] = smin[threadIdx.x + st] < smin[threadIdx.x] ? smin[threadIdx.x + st] : smin[threadIdx.x];
      smax[threadIdx.x] = smax[threadIdx.x + st] > smax[threadIdx.x] ? smax[threadIdx.x + st] : smax[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { minmax[2 * blockIdx.x] = smin[0]; minmax[2 * blockIdx.x + 1] = smax[0]; }
}

__device__ __forceinline__ float aug_cubic(float x, float f0, float f1, float f2, float f3) {
  return f1 + 0.5f * x * (f2 - f0 + x * (2.0f * f0 - 5.0f * f1 + 4.0f * f2 - f3 + x * (3.0f * (f1 - f2) + f3 - f0)));
}

// one thread per (sample, output pixel); loops over the slices (contiguous in the volume)
__global__ __launch_bounds__(256) void augment_resample_kernel(const float* __restrict__ img, const uint8_t* __restrict__ lab,
                                                               const hdu_aug_sample* __restrict__ smp, int size, int nslices,
                                                               int lab_slice, float mean, const float* __restrict__ minmax,
                                                               float* __restrict__ x_out, long long x_sample, long long x_pix,
                                                               long long x_slice, uint8_t* __restrict__ y_out,
                                                               long long y_sample, long long y_slice) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= size * size) return;
  const int orow = pix / size, ocol = pix - orow * size;
  const hdu_aug_sample s = smp[n];
  const float* vol = img + s.img_off;
  const uint8_t* lvol = lab + s.img_off;
  const float scale = (float)s.crop / (float)size;
  const float r = scale * ((float)orow + 0.5f) - 0.5f;
  const float c = scale * ((float)ocol + 0.5f) - 0.5f;
  // ---- labels: nearest, mode 'edge' (C round(): half away from zero)
  {
    int ri = (int)(r >= 0.f ? floorf(r + 0.5f) : ceilf(r - 0.5f));
    int ci = (int)(c >= 0.f ? floorf(c + 0.5f) : ceilf(c - 0.5f));
    ri = ri < 0 ? 0 : (ri > s.crop - 1 ? s.crop - 1 : ri);
    ci = ci < 0 ? 0 : (ci > s.crop - 1 ? s.crop - 1 : ci);
    int gi, gj;
    aug_unflip(s.flip, s.crop, ri, ci, &gi, &gj);
    const uint8_t* lp = lvol + ((long long)(s.a0 + gi) * s.vcols + (s.b0 + gj)) * s.vslices + s.c0;
    if (lab_slice >= 0) {
      y_out[n * y_sample + pix] = lp[lab_slice];
    } else {
      for (int k = 0; k < nslices; ++k) y_out[n * y_sample + k * y_slice + pix] = lp[k];
    }
  }
  // ---- image: bicubic, mode 'constant' (taps outside the crop read 0 = cval, AFTER the mean subtraction)
  const int r0 = (int)floorf(r), c0 = (int)floorf(c);
  const float xr = r - (float)r0, xc = c - (float)c0;
  const float* tap[4][4];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr)
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int ti = r0 - 1 + pr, tj = c0 - 1 + pc;
      if (ti < 0 || ti >= s.crop || tj < 0 || tj >= s.crop) { tap[pr][pc] = nullptr; continue; }
      int gi, gj;
      aug_unflip(s.flip, s.crop, ti, tj, &gi, &gj);
      tap[pr][pc] = vol + ((long long)(s.a0 + gi) * s.vcols + (s.b0 + gj)) * s.vslices + s.c0;
    }
  const float lo = minmax[2 * n], hi = minmax[2 * n + 1];
  const bool preserve = !(lo <= 0.f && 0.f <= hi);
  for (int k = 0; k < nslices; ++k) {
    float fr[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      float f[4];
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) f[pc] = tap[pr][pc] ? tap[pr][pc][k] - mean : 0.f;
      fr[pr] = aug_cubic(xc, f[0], f[1], f[2], f[3]);
    }
    float v = aug_cubic(xr, fr[0], fr[1], fr[2], fr[3]);
    if (!(preserve && v == 0.f)) v = v < lo ? lo : (v > hi ? hi : v);
    x_out[n * x_sample + (long long)pix * x_pix + k * x_slice] = v;
  }
}

extern "C" int hdu_augment_batch(const float* img, const uint8_t* lab, const hdu_aug_sample* samples, int n, int size,
                                 int nslices, int lab_slice, float mean, float* minmax_ws, float* x_out, int64_t x_sample,
                                 int64_t x_pix, int64_t x_slice, uint8_t* y_out, int64_t y_sample, int64_t y_slice,
                                 void* stream) {
  if (!img || !lab || !samples || !minmax_ws || !x_out || !y_out || n <= 0 || size <= 0 || nslices <= 0 || lab_slice >= nslices)
    return hdu_set_error(HDU_ERR_ARG, "augment_batch: bad args");
  hipStream_t s = (hipStream_t)stream;
  HDU_LAUNCH(augment_minmax_kernel, dim3((unsigned)n), dim3(256), 0, s, img, samples, nslices, mean, minmax_ws);
  HDU_LAUNCH(augment_resample_kernel, dim3((unsigned)((size * size + 255) / 256), (unsigned)n), dim3(256), 0, s, img, lab,
             samples, size, nslices, lab_slice, mean, (const float*)minmax_ws, x_out, (long long)x_sample, (long long)x_pix,
             (long long)x_slice, y_out, (long long)y_sample, (long long)y_slice);
  return hdu_check_launch("augment_batch");
}
