/* hdu.h -- C-ABI of libhdu.so: the MI355X (gfx950) kernels of the H-DenseUNet hot path.
 *
 * The reference (xmengli/H-DenseUNet) has NO native/FFI interface: its device
 * arithmetic is TensorFlow ops reached through the vendored Keras backend
 * (SURVEY.md section 8b).  Each entry point below therefore cites the Keras
 * backend / layer function (reference file:line) whose device work it replaces.
 * "TFB" = Keras-2.0.8/keras/backend/tensorflow_backend.py, "K." = Keras-2.0.8/keras/.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types.
 *  - All pointers are DEVICE pointers owned by the caller (no ownership transfer, no
 *    allocation inside); work is enqueued asynchronously on `stream` (a hipStream_t).
 *  - Return value: 0 = ok, <0 = error; hdu_last_error() gives a thread-local message.
 *    Nothing throws or exits across the ABI.
 *  - Activations are channels-last [N][D][H][W][C] (D = 1 for 2D), element type selected
 *    by `dtype`; `ld*` is the element stride between consecutive pixels (>= C, lets a
 *    tensor be a channel slab of a wider dense-block buffer).  Channel counts and slab
 *    offsets must be multiples of 16 bytes / sizeof(element) (8 for bf16, 4 for f32);
 *    the host pads the 3-/4-channel network inputs and the 3-class head to that.
 *  - Per-channel parameter / statistics vectors are always float32.
 */
#ifndef HDU_H_
#define HDU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDU_BF16 0
#define HDU_F32 1

#define HDU_OK 0
#define HDU_ERR_ARG (-1)
#define HDU_ERR_LAUNCH (-2)
#define HDU_ERR_WORKSPACE (-3)

const char* hdu_last_error(void);
/* "hip-gfx950" for the product library; "emu-x86" for the CPU test build of the same sources. */
const char* hdu_backend(void);
/* Layout version of the structs in this header (hdu_conv_desc, hdu_fold_entry, hdu_aug_sample, hdu_prep_entry) and of the
 * entry-point set.  A binding compares hdu_abi_version() and hdu_sizeof_conv_desc() with what it was written against and
 * refuses a stale library (h-denseunet_amd/lib.py does): 1 = round 1, 2 = round 2 (splitk_*, bnb_*), 3 = epi_*,
 * 4 = round 3 (hdu_zero_regions, hdu_comm_*), 5 = round 4 (hdu_profile_*, pointwise convs with a fused BN prologue on the
 * DMA path, hdu_wgrad_plan_shape / min_steps), 6 = round 6 (hdu_bn_bwd_apply_sums, bnb_relu bit 2), 7 = round 6 (hdu_split3_*). */
#define HDU_ABI_VERSION 7
int hdu_abi_version(void);
size_t hdu_sizeof_conv_desc(void);
/* Launch profiler (measurement only; replaces nothing in the reference -- Keras has `verbose`, the reference was profiled with
 * nvprof from outside).  hdu_profile_begin(max_records) arms it: from then on EVERY kernel launch of this library is made
 * with a start / stop event pair attached to that dispatch (hipExtLaunchKernelGGL), so a record is the kernel's own
 * begin-to-end time -- the figure rocprofv3 --kernel-trace reports -- not a marker-to-marker interval.  hdu_profile_end()
 * disarms it, waits for the last recorded kernel and returns the number of records; hdu_profile_get(i, ...) then returns
 * record i in launch order: the instantiated kernel name (NUL-terminated, truncated to `buflen`) and its duration in
 * milliseconds.  Launches beyond max_records run unprofiled.  Do not arm it during hipGraph capture.  The x86 test
 * build records names with a duration of 0. */
int hdu_profile_begin(int max_records);
int hdu_profile_count(void);
int hdu_profile_end(void);
int hdu_profile_get(int i, char* name_buf, size_t buflen, float* ms);
/* developer tuning knobs (process-wide): key HDU_TUNE_DMA_STAGES: 2 = two LDS stages, deep ring for small grids (default); 6 = deep ring everywhere */
#define HDU_TUNE_DMA_STAGES 0
#define HDU_TUNE_HALO_TARGET_WGS 7    /* workgroups a halo-tile filter-gradient launch aims for */
#define HDU_TUNE_MAX_BN 6            /* widest N tile the dispatcher may pick (default 128) */
#define HDU_TUNE_NO_FAST 5           /* 1 = disable the bitmask/32-bit-offset addressing path (A/B) */
#define HDU_TUNE_DEBUG 4             /* developer experiments: bit0 skip operand DMA, bit1 skip MFMA (wrong results!);
                                        bit4 (16): treat every input tensor as >= 4 GiB (64-bit pointer path; tests) */
#define HDU_TUNE_XCD_SWIZZLE 3       /* bit0 = XCD-aware tile order in the implicit GEMM (default on) */
#define HDU_TUNE_RED_WGS 11          /* workgroups a per-channel reduction aims for (default 512) */
#define HDU_TUNE_ROW_WGS 12          /* workgroups an element-wise row kernel aims for (default 512 = 2 per CU; rounds 1-2: 2048) */
#define HDU_TUNE_NO_HALO_FPROP 9     /* 1 = disable the halo-tile forward / data-gradient kernel (A/B) */
#define HDU_TUNE_NO_HALO 8           /* bit 0 = disable the halo-tile filter-gradient kernel, bit 1 = for 3 x 3 x 3 layers only, bit 2 = for layers with a fused up-sampling only (A/B) */
#define HDU_TUNE_WGRAD_TARGET_WGS 2  /* workgroups a filter-gradient launch aims for */
#define HDU_TUNE_SPLITK 13           /* 0 = library default (split small grids), 1 = never split, N >= 2 = force N splits where possible (tests) */
#define HDU_TUNE_BM64_MAX_M 18        /* layers with at most this many output pixels use 64-row tiles (default 16384) */
#define HDU_TUNE_SPLITK_TARGET 16     /* workgroups a split-K launch aims for (default 256 per co-resident workgroup) */
#define HDU_TUNE_SPLITK_MIN_STEPS 17  /* K steps every split keeps at least (default 4) */
#define HDU_TUNE_HALO_MIN_TILES 15   /* the halo-tile forward kernel needs this many 4x32-pixel tiles (default 128) */
#define HDU_TUNE_RING_MIN_K 14       /* small grids use the deep LDS ring when Ktot > this (default 0: always) */
#define HDU_TUNE_WGRAD_MIN_STEPS 1   /* minimum pixel steps (of 64) per filter-gradient workgroup */
#define HDU_TUNE_NO_PW_BSTAT 19      /* 1 = disable the filter-stationary pointwise kernel (A/B) */
#define HDU_TUNE_PW_BSTAT_WGS 20     /* workgroups that kernel aims for (default 256 / 512 by form) */
#define HDU_TUNE_PW_BSTAT_FORM 30    /* 0 = library's choice, 1 = 128 channels per workgroup, three-slot ring, one workgroup per CU (rounds 3-5),
                                        2 = 64 channels, two slots, 80 KB of LDS: two workgroups per CU (round 6) */
#define HDU_TUNE_NO_PRO_DMA 22       /* 1 = a pointwise conv with a BN prologue takes the VGPR-gather kernels of rounds 1-3 (A/B) */
#define HDU_TUNE_PERS 23             /* 0 = two-stage implicit GEMM on large grids (default: the persistent 256-row kernel was measured 10-20 %
                                        slower per launch, profiles/r04_experiment_persistent_gemm.txt), 1 = persistent kernel, one
                                        workgroup per CU (256), N > 1 = persistent kernel with N workgroups */
#define HDU_TUNE_NO_PW_BSTAT_BNB 26   /* 1 = a bottleneck data gradient with a fused BN backward takes the tiled kernels (round 3; A/B) */
#define HDU_TUNE_PERS_MIN_ITEMS 24   /* (m-tile, n-tile) pairs a layer needs to take the persistent form (default 512) */
#define HDU_TUNE_HALO_MIN_W 28       /* narrowest layer (pixels per row) that takes the halo-tile filter gradient (tiles are 4 x 32 pixels; default 24) */
#define HDU_TUNE_F32_SPLIT 27        /* float32 convolutions: 0 = exact f32 MFMA (default; the parity mode), 1 = every operand split into bf16
                                        hi + lo and contracted as ah.bh + ah.bl + al.bh on the bf16 MFMA with the float32 accumulator
                                        ("bf16 x 3": <= 3 * 2^-18 relative per product; storage, statistics and every row kernel stay float32) */
#define HDU_TUNE_HALO_WIDE 29        /* halo-tile forward / data-gradient kernel for the wide 3x3 / 3x3x3 layers (conv_halo_wide.hip, round 5):
                                        0 = library heuristic (default), 1 = off (A/B: the im2col kernels of rounds 1-4), 2..8 = force
                                        configuration 8x128 / 16x64 / 16x96 / 8x64 / 8x96 / 16x128 / 16x64p (tile rows x output channels) where the shape allows;
                                        any value >= 2 also lets the stem kernels take geometries below their size thresholds (tests) */
#define HDU_TUNE_SPLIT3_FORM 31      /* hdu_split3_entry_fill / hdu_split3_batched (developer A/B, set before a table is filled): low nibble = channels per lane
                                      * (0 = 8, 4), bits 4.. = row groups of 8 per thread (0 = 1) */
#define HDU_TUNE_WGRAD_NCT 21        /* 1 = one filter-row tile per pointwise filter-gradient workgroup (round 2's form; A/B) */
int hdu_set_tuning(int key, int value);

/* ------------------------------------------------------------------ convolution
 * Replaces K.layers/convolutional.py:148-182 (_Conv.call) -> TFB:3128-3165 (conv2d) /
 * TFB:3277-3314 (conv3d) + TFB:3435-3497 (bias_add), with the layers the reference
 * places directly in front of a conv folded into the operand gather:
 *   BatchNormalization(+Scale)+Activation('relu')  -> per-channel  relu(pro_a[c]*x + pro_b[c])
 *       (K.layers/normalization.py:126-190, lib/custom_layers.py:63-69, TFB:2656-2679)
 *   UpSampling2D/3D nearest                        -> ud/uh/uw (K.layers/convolutional.py:1359,1432; TFB:1739-1827)
 *   add([skip, up])                                -> skip      (K.layers/merge.py:207-211)
 *   ZeroPadding2D/3D                               -> pd/ph/pw  (K.layers/convolutional.py:1584,1702; TFB:1989-2071)
 * and Dropout (TFB:2869-2888) folded into the epilogue.
 *   x_eff[n,d,h,w,c] = act(pro_a[c]*x[n,d>>ud,h>>uh,w>>uw,c]+pro_b[c]) + skip[n,d,h,w,c]   (0 outside)
 *   y[n,o,co] = sum_{k,c} x_eff[n, o*s + k - p, c] * w[co][k][c]  (+ bias[co]) (* dropout mask/keep)
 * Filter layout: [Cout][KD][KH][KW][Cin] (Keras' (kd,kh,kw,Cin,Cout) is converted on the host).
 */
typedef struct hdu_conv_desc {
  int dtype;
  const void* x;      int64_t ldx;
  int N, Di, Hi, Wi, Cin;          /* stored (pre-upsample) input dims */
  int ud, uh, uw;                  /* nearest-upsample shift per axis: 0 (x1) or 1 (x2) */
  const void* skip;   int64_t ldskip;   /* optional, at the upsampled resolution, Cin channels */
  const float* pro_a; const float* pro_b; int pro_relu; /* optional input affine (+ReLU) */
  const void* w;                   /* [Cout][KD*KH*KW*Cin], element type = dtype */
  int KD, KH, KW;
  int sd, sh, sw;
  int pd, ph, pw;
  void* y;            int64_t ldy;
  int Do, Ho, Wo, Cout;
  const float* bias;               /* optional [Cout] */
  /* optional output affine (+ReLU) AFTER bias and dropout: y = act(epi_a[co] * y + epi_b[co]).  The folded
   * BatchNormalization(+Scale)+Activation('relu') that FOLLOWS this conv (K.layers/normalization.py:126-190,
   * lib/custom_layers.py:63-69) when it runs on stored statistics and nothing needs the conv's raw output: the conv
   * writes the next conv's operand directly -- one full-width pass per bottleneck less (frozen 2D branch of the hybrids,
   * every predict).  Not with accumulate / epilogue statistics / the fused BN backward. */
  const float* epi_a; const float* epi_b; int epi_relu;
  int accumulate;                  /* y += result instead of y = result */
  float drop_keep;                 /* 1.0 = no dropout; else keep-probability */
  uint32_t drop_seed;
  const uint32_t* drop_seed_dev;   /* optional device word added to drop_seed (lets a captured hipGraph
                                      draw a fresh mask per replay) */
  /* optional (hdu_conv_fprop only): per-channel moments of the stored output, taken in the epilogue while the tile is
   * still in LDS -- tf.nn.moments of the conv output (TFB:1635) without a second pass over it.
   * stats_partial[slot][0][c] += sum_m (y[m][c] - stats_shift[c]); [slot][1][c] += sum_m (y - shift)^2, float atomics,
   * slot = workgroup % stats_slots.  The caller zeroes stats_partial and finishes with hdu_bn_stats_finalize. */
  float* stats_partial;
  const float* stats_shift;
  int stats_slots;
  /* optional fused BatchNormalization(+Scale)+ReLU BACKWARD in the epilogue of a data-gradient launch (x = dy, w = the
   * flipped filter): the launch's output tile is dz, the gradient w.r.t. z = relu(a*u + b) where u (`bnb_u`, same pixel
   * grid and channel count as the output) is the BN input of the forward pass.  Instead of storing dz the epilogue
   *   - stores (or, with `accumulate`, adds to) a[c] * g,  g = dz where a*u+b > 0 (or everywhere when bnb_relu == 0):
   *     the part of du that needs no reduction (tf.gradients of TFB:1639 batch_normalization: du = a*(g - mean(g) -
   *     uhat*mean(g*uhat)); the two mean terms are an affine function of u applied later by hdu_bn_bwd_correct);
   *   - accumulates S1[c] += sum g and S2[c] += sum g*(u - bnb_mean[c])*bnb_rstd[c] into bnb_partial[slot][0|1][c]
   *     (float atomics, slot = workgroup % bnb_slots; caller zeroes it; NULL = no sums wanted), finished by
   *     hdu_bn_bwd_finalize.
   * Replaces, per BN, the dz round trip through HBM, the reduction pass and the full-width apply pass.
   * bnb_relu: bit 0 = the BN is followed by ReLU; bit 1 (value 2) = `bnb_u` holds the BN's OUTPUT z = relu(a*u + b) instead
   * of its input (a producer whose epilogue applied the stored-statistics BN, epi_*, never wrote u): the mask is z > 0 and
   * the normalised input is recovered as (z - (b + a*mean)) * (rstd / a) where z > 0.  Tile kernels only.
   * bit 2 (value 4, ABI 6) = SUMS ONLY: the tile is stored as raw dz and the epilogue only accumulates S1 / S2 into bnb_partial --
   * the reduction half of a batch-statistics BN backward rides in the launch that produces dz, the apply half follows as ONE
   * launch (hdu_bn_bwd_apply_sums with sums = bnb_partial, slots = bnb_slots).  Tile kernels only. */
  const void* bnb_u;  int64_t bnb_ldu;
  const float* bnb_a; const float* bnb_b; const float* bnb_mean; const float* bnb_rstd;
  int bnb_relu;
  float* bnb_partial; int bnb_slots;
  /* optional split-K scratch of hdu_conv_fprop.  Layers whose output grid cannot fill the chip (dense blocks at 1/16 and
   * 1/32 resolution, every 3D dense block: M = 147..9408 pixels) are bound by what ONE compute unit can pull from L2;
   * with scratch the K loop is split over `S` workgroups per output tile (S chosen by the library,
   * hdu_conv_splitk_ws_bytes says how much it wants), partial tiles meet in `splitk_ws` (float32, write-through) and the
   * tile's last-arriving workgroup sums them and runs the ordinary epilogue.  `splitk_counters`: >= 512 zeroed uint32
   * (every launch returns them to zero).  Both may be shared by all launches of one stream.  NULL = never split. */
  void* splitk_ws;
  size_t splitk_ws_bytes;
  uint32_t* splitk_counters;
  /* Depth sharding: this launch computes part of a layer (the local depth planes of one volume split over several ranks,
   * possibly plus halo planes a neighbour also computes).  layer_rows = N*Do*Ho*Wo of the WHOLE unsharded layer; tile-shape
   * and split-K decisions are then taken for that pixel count, so every output element is summed over the same K partition
   * -- bit for bit -- as in the unsharded launch.  0 = this launch is the whole layer. */
  int64_t layer_rows;
} hdu_conv_desc;

/* finishes the sums of a fused BN-backward epilogue (hdu_conv_desc.bnb_partial): S1, S2 totals -> parameter gradients
 * (dgamma = sg*S2, dbeta = sg*S1, dsgamma = g*S2 + beta*S1, dsbeta = S1; NULL = not wanted) and, for a batch-statistics
 * BN (batch_stats != 0), the coefficients of the deferred part of du = -k3*u + k4 ADDED to corr3 / corr4 (one pair of
 * per-channel accumulators per stored tensor: every consumer BN of a dense-block slab adds its own; the caller zeroes
 * them once per backward pass).  k3 = a*rstd*S2/M, k4 = k3*mean - a*S1/M. */
int hdu_bn_bwd_finalize(const float* partial, int slots, int64_t M, int C, int batch_stats, const float* gamma,
                        const float* beta, const float* sgamma, const float* mean, const float* rstd, float* dgamma,
                        float* dbeta, float* dsgamma, float* dsbeta, float* corr3, float* corr4, void* stream);

/* hdu_bn_bwd_finalize of MANY inference-mode BNs (batch_stats = 0: parameter gradients only, nothing the backward chain reads)
 * as ONE launch at the end of the backward pass -- dense_rnn_net trains 216 Scale layers on frozen statistics
 * (hybridnet.py:182-354), i.e. 216 finalize launches of ~4.6 us inside one step.  `begins[i]` = first 8-channel block of entry
 * i (exclusive prefix sum of ceil(C / 8)); total_blocks = the grand total. */
typedef struct hdu_bnbwd_entry {
  const float* partial;      /* [slots][2][C] */
  int32_t slots, C;
  const float* gamma; const float* beta; const float* sgamma;
  float* dgamma; float* dbeta; float* dsgamma; float* dsbeta;      /* any may be NULL */
} hdu_bnbwd_entry;
int hdu_bn_bwd_finalize_batched(const hdu_bnbwd_entry* table, const uint32_t* begins, int n, uint32_t total_blocks, void* stream);

/* du[m][c] += -corr3[c]*u[m][c] + corr4[c]: the deferred, reduction-dependent part of the BN backward of every consumer
 * of these channels, applied ONCE, right before their producer reads the gradient. */
int hdu_bn_bwd_correct(int dtype, const void* u, int64_t ldu, int64_t M, int C, const float* corr3, const float* corr4,
                       void* du, int64_t lddu, void* stream);
/* hdu_bn_bwd_finalize(batch_stats = 1) of one BN and hdu_bn_bwd_correct of the channels [cs0, cs0 + Cc) of the tensor it
 * normalises, in ONE launch (in a dense block the two follow each other: layer i's fused BN backward is finalized, then layer
 * i-1 -- the producer of the last slab that BN reads -- takes its gradient).  corr3 / corr4: the accumulators of the BN's C
 * channels; the corrected channels use (stored value + this BN's share) and their accumulators are NOT updated (they are never
 * read again).  u / du point at channel cs0 of the stored tensor / its gradient. */
int hdu_bn_bwd_finalize_correct(int dtype, const float* partial, int slots, int64_t M, int C, const float* gamma, const float* beta,
                                const float* sgamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                float* dsgamma, float* dsbeta, float* corr3, float* corr4, int cs0, int Cc, const void* u,
                                int64_t ldu, void* du, int64_t lddu, void* stream);

/* bytes of split-K scratch hdu_conv_fprop would use for this descriptor (0 = it would not split) */
size_t hdu_conv_splitk_ws_bytes(const hdu_conv_desc* d);

/* forward conv; also the data-gradient of every stride-1 conv (caller passes dy as x and the
 * flipped/transposed filter from hdu_weight_prep; replaces tf.gradients of TFB:3158/3307). */
int hdu_conv_fprop(const hdu_conv_desc* d, void* stream);

/* filter gradient: dw[co][k][c] += sum_{n,o} x_eff[n,o*s+k-p,c] * dy[n,o,co]; d->y is dy (read only),
 * dw is float32 [Cout][KD*KH*KW*Cin] and is accumulated into (atomics).  x_eff is recomputed from the raw
 * input with the same prologue as the forward (the normalised tensor is never stored). */
int hdu_conv_wgrad(const hdu_conv_desc* d, float* dw, void* stream);

/* Batched filter gradients: ONE launch per kernel family covers many layers.  Filter gradients feed nothing but the
 * optimiser (tf.gradients order is free, K.optimizers.py:168), so a caller may defer them to the end of the backward
 * pass; layers too small to fill the GPU on their own then run together.  bf16, materialised inputs only.
 *   hdu_wgrad_plan_entry_bytes(): size of one opaque table entry.
 *   hdu_wgrad_plan_fill(): fills ONE host-side entry for (desc, dw) as hdu_conv_wgrad(desc, dw) would run it;
 *     *variant = kernel family of the entry (entries of one family go into one table), *nblocks = workgroups it needs.
 *     target_wgs <= 0: the single-launch default split.  min_steps > 0: every workgroup keeps at least that many 64-pixel
 *     steps (DMA families) / 4x32-pixel spatial tiles (halo families) -- the caller sizes it from the WHOLE launch (hdu_wgrad_plan_shape of every layer of the family):
 *     a partial tile costs Cout x 128 float atomics, and in a batched launch the other layers fill the GPU, so a layer needs
 *     far fewer pixel splits than it would alone (measured round 4: 8 -> ~100 steps, 2D step 18.1 -> 17.6 ms).
 *   hdu_wgrad_plan_shape(): *variant as above, *tiles = output tiles of the layer (workgroups per pixel split), *steps = its
 *     64-pixel steps (DMA families) / 4x32-pixel spatial tiles (halo families).
 *   hdu_wgrad_plan_run(): dev_entries = the entries of ONE family copied to device memory, dev_begins[i] = first
 *     workgroup of entry i (exclusive prefix sum of nblocks), total_blocks = their sum.  Same result as calling
 *     hdu_conv_wgrad per layer (dw += ..., float atomics). */
size_t hdu_wgrad_plan_entry_bytes(void);
int hdu_wgrad_plan_fill(const hdu_conv_desc* d, float* dw, int target_wgs, int min_steps, void* entry, int* variant,
                        uint32_t* nblocks);
int hdu_wgrad_plan_shape(const hdu_conv_desc* d, int* variant, uint32_t* tiles, uint32_t* steps);
int hdu_wgrad_plan_run(int variant, const void* dev_entries, const uint32_t* dev_begins, int n, uint32_t total_blocks,
                       void* stream);

/* data-gradient of a strided conv (the 7x7x7 stride-2 stem, needed by hybridnet end2end):
 * dx[n,i,c] (+)= sum_{o,k: o*s+k-p=i} dy[n,o,co]*w[co][k][c].  d->x is the output dx, d->y is dy. */
int hdu_conv_dgrad_strided(const hdu_conv_desc* d, void* stream);

/* A stride-2 data gradient on the MFMA path: input positions of parity r = i mod 2 (per stride-2 axis) only see the taps
 * k = (r + pad) mod 2 of the forward filter, so the gradient is 2^d stride-1 correlations of dy with sub-filters of
 * ceil / floor(K/2) taps per axis.  hdu_stride2_dgrad_filters gathers the sub-filters of all classes from the float32
 * master filter [Cout][KD][KH][KW][Cin] into w_out, class (rd, rh, rw) after class in row-major order, each
 * [Cin][ntd][nth][ntw][Cout] in the compute dtype: tap t of a stride-2 axis is k = kmax_r - 2t, kmax_r the largest tap of
 * the class's parity; a stride-1 axis has one class with the taps reversed.  The caller then runs hdu_conv_fprop per
 * class (x = dy, w = the sub-filter, KD/KH/KW = the class's taps, stride 1, pad = (r + pad - kmax_r)/-2 per stride-2
 * axis, output = the class grid of size Di/sd x Hi/sh x Wi/sw) into a buffer [class][N][Dq][Hq][Wq][Cin], and
 * hdu_parity_interleave scatters (or adds) the class grids into dx.  (hdu_conv_dgrad_strided stays as the general form.) */
int hdu_stride2_dgrad_filters(int dtype, const float* w_master, int Cout, int KD, int KH, int KW, int Cin, int sd, int sh,
                              int sw, int pd, int ph, int pw, void* w_out, void* stream);
int hdu_parity_interleave(int dtype, const void* cls, int N, int Di, int Hi, int Wi, int C, int sd, int sh, int sw, void* dx,
                          int64_t lddx, int accumulate, void* stream);

/* name of the kernel template instance the dispatcher launches for this descriptor (op 0 = fprop/dgrad-form,
 * 1 = wgrad); lets a profiler-side tool group launches exactly like rocprofv3's per-kernel statistics. */
int hdu_conv_kernel_name(const hdu_conv_desc* d, int op, char* buf, size_t buflen);

/* master float32 filters [Cout][T][Cin] -> compute-dtype copies: w_f (same layout) and, if w_d != NULL,
 * the data-gradient filter w_d [Cin][T flipped][Cout]. */
int hdu_weight_prep(int dtype, const float* w_master, int Cout, int T, int Cin, void* w_f, void* w_d, void* stream);

/* all layers of a model in ONE launch: `table` is a device array of n hdu_prep_entry (element offsets into the flat
 * float32 master buffer and into the flat compute-dtype filter buffer; w_f_off / w_d_off < 0 = not wanted).  Work
 * is split in 64x64 (Cout x Cin) tiles per tap (HDU_PREP_TILE; 32 until ABI 6); tile_begin is the running tile count (exclusive
 * prefix sum of T*ceil(Cout/64)*ceil(Cin/64)), total_tiles the grand total.  Offsets and channel counts that are multiples of 4 take
 * 16-byte reads / 8-16-byte writes; others are copied element-wise. */
#define HDU_PREP_TILE 64
typedef struct hdu_prep_entry {
  int64_t master_off, w_f_off, w_d_off, tile_begin;
  int32_t Cout, T, Cin, pad_;
} hdu_prep_entry;
int hdu_weight_prep_batched(int dtype, const hdu_prep_entry* table, int n, int64_t total_tiles,
                            const float* master_base, void* wc_base, void* stream);

/* ------------------------------------------------------------------ training-sample assembly (N3)
 * train_2ddense.py:40-133 / train_hybrid.py:40-133 on the device: the pre-processed CT volumes (float32) and label
 * volumes (uint8), each [rows][cols][slices] with slices fastest, stay resident in HBM; one call crops `crop` x `crop` x
 * nslices voxels at (a0, b0, c0), subtracts `mean`, applies flip / rotation case `flip` (0..7, train_2ddense.py:73-101)
 * and resizes to size x size exactly as skimage.transform.resize does for the reference's two calls (:103-104: labels
 * order 0 mode 'edge'; image order 3 mode 'constant' cval 0 clip=True -- Catmull-Rom cubic convolution, output clipped
 * to the crop's value range), writing sample n / output pixel p / slice k to x_out[n*x_sample + p*x_pix + k*x_slice]
 * and the labels to y_out[n*y_sample + p] (lab_slice >= 0: that slice only, the 2D net's middle slice) or
 * y_out[n*y_sample + k*y_slice + p] (lab_slice < 0: all slices, the hybrid).  minmax_ws: 2*n floats of scratch. */
typedef struct hdu_aug_sample {
  int64_t img_off;                 /* element offset of the case's volume inside img / lab */
  int32_t vrows, vcols, vslices;
  int32_t a0, b0, c0;
  int32_t crop, flip;
} hdu_aug_sample;
int hdu_augment_batch(const float* img, const uint8_t* lab, const hdu_aug_sample* samples, int n, int size, int nslices,
                      int lab_slice, float mean, float* minmax_ws, float* x_out, int64_t x_sample, int64_t x_pix,
                      int64_t x_slice, uint8_t* y_out, int64_t y_sample, int64_t y_slice, void* stream);

/* ------------------------------------------------------------------ batch normalisation
 * K.layers/normalization.py:126-190 -> TFB:1620-1664 (normalize_batch_in_training = tf.nn.moments +
 * tf.nn.batch_normalization), TFB:1667-1684 (inference), TFB:915-927 (moving_average_update);
 * Scale: lib/custom_layers.py:63-69.
 */
size_t hdu_reduce_ws_bytes(int64_t M, int C);

/* Inference-mode folds of MANY BatchNormalization(+Scale) layers in ONE launch (K.layers/normalization.py:173-190 with
 * training=False -> TFB:1667-1684: the moving statistics, no update).  Such folds depend on parameters only, never on
 * activations, so a frozen sub-network (denseunet3d.py:222-224, hybridnet.py:211) or a whole predict pass needs one
 * launch instead of one per layer.  `table`: device array of n entries; `begins`: device array, begins[i] = first
 * 256-channel block of entry i (exclusive prefix sum of ceil(C/256)); total_blocks = the grand total. */
typedef struct hdu_fold_entry {
  const float* mean; const float* var; const float* gamma; const float* beta; const float* sgamma; const float* sbeta;
  float* a; float* b; float* rstd;
  int32_t C; float eps;
} hdu_fold_entry;
int hdu_bn_fold_batched(const hdu_fold_entry* table, const uint32_t* begins, int n, uint32_t total_blocks, void* stream);

/* per-channel mean and biased variance over M pixels (tf.nn.moments, TFB:1635) */
int hdu_bn_stats(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* mean, float* var, void* ws,
                 size_t ws_bytes, void* stream);

/* fold BN (+ optional Scale) into one per-channel affine  y = a*x + b :
 *   rstd = rsqrt(var+eps); a = sg*g*rstd; b = sg*(beta - mean*g*rstd) + sb      (sg=1, sb=0 when NULL)
 * and, when mov_mean != NULL, the training-mode moving-statistics update  m -= (m - batch)*(1-momentum). */
int hdu_bn_fold(int C, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                const float* sgamma, const float* sbeta, float* a, float* b, float* rstd, float* mov_mean,
                float* mov_var, float momentum, void* stream);

/* hdu_bn_stats + hdu_bn_fold in two launches instead of three (the fold runs in the finalize pass of the reduction) */
int hdu_bn_stats_fold(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* mean, float* var,
                      const float* gamma, const float* beta, float eps, const float* sgamma, const float* sbeta,
                      float* a, float* b, float* rstd, float* mov_mean, float* mov_var, float momentum, void* ws,
                      size_t ws_bytes, void* stream);

/* hdu_bn_bwd_reduce + hdu_bn_bwd_coef fused the same way */
int hdu_bn_bwd_reduce_coef(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                           const float* a, const float* b, int relu, const float* mean, const float* rstd,
                           int batch_stats, const float* gamma, const float* beta, const float* sgamma, float* s1,
                           float* s2, float* k1, float* k2, float* k3, float* dgamma, float* dbeta, float* dsgamma,
                           float* dsbeta, void* ws, size_t ws_bytes, void* stream);

/* backward of  z = relu?(a*x+b)  where a,b came from hdu_bn_fold: per-channel sums
 *   s1 = sum g, s2 = sum g*(x-mean)*rstd,  g = dz * [a*x+b > 0]  */
int hdu_bn_bwd_reduce(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                      const float* a, const float* b, int relu, const float* mean, const float* rstd, float* s1,
                      float* s2, void* ws, size_t ws_bytes, void* stream);

/* per-channel coefficients of dx = k1*g - k2 - k3*(x-mean) and the parameter gradients.
 * batch_stats=1: training-mode BN (TFB:1635-1640): k1=sg*g*rstd, k2=k1*s1/M, k3=k1*rstd*s2/M;
 * batch_stats=0: frozen/inference BN: k1=sg*g*rstd, k2=k3=0.  Any of the d* outputs may be NULL. */
int hdu_bn_bwd_coef(int C, int64_t M, int batch_stats, const float* s1, const float* s2, const float* gamma,
                    const float* beta, const float* sgamma, const float* rstd, float* k1, float* k2, float* k3,
                    float* dgamma, float* dbeta, float* dsgamma, float* dsbeta, void* stream);

/* dx (+)= (k1*g - k2 - k3*(x-mean)) * dropmask   (g as above; dropout mask of the conv that produced x) */
int hdu_bn_bwd_apply(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C,
                     const float* a, const float* b, int relu, const float* mean, const float* k1,
                     const float* k2, const float* k3, void* dx, int64_t lddx, int accumulate, float drop_keep,
                     uint32_t drop_seed, const uint32_t* drop_seed_dev, void* stream);

/* The whole BatchNormalization(+Scale)+ReLU backward in TWO launches (hdu_bn_bwd_reduce_coef + hdu_bn_bwd_apply take three).
 * Launch 1 adds every workgroup's column sums (S1 = sum g, S2 = sum g*xhat) to row (workgroup % slots) of `sums`
 * ([slots][2][C] float32, 1 <= slots <= 32, ZERO on entry: hdu_zero_regions) with float atomics; launch 2 sums the slot rows
 * for its own channels, derives k1 / k2 / k3 in registers (formulas of hdu_bn_bwd_coef) and writes dx like hdu_bn_bwd_apply;
 * its first row block writes the parameter gradients (any of them may be NULL).  Per-channel vectors 16-byte aligned.
 * A finalize launch costs ~4.8 us of pure latency per BatchNormalization (161 of them in one DenseUNet-161 step). */
int hdu_bn_bwd_fused(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C, const float* a,
                     const float* b, int relu, const float* mean, const float* rstd, int batch_stats, const float* gamma,
                     const float* beta, const float* sgamma, float* sums, int slots, float* dgamma, float* dbeta,
                     float* dsgamma, float* dsbeta, void* dx, int64_t lddx, int accumulate, float drop_keep,
                     uint32_t drop_seed, const uint32_t* drop_seed_dev, void* stream);

/* ABI 6: launch 2 of hdu_bn_bwd_fused alone.  `sums` ([slots][2][C]) already holds S1 / S2: the data-gradient launch that produced
 * dz took them in its epilogue (hdu_conv_desc.bnb_relu bit 2 with bnb_partial = sums, bnb_slots = slots).  Replaces the
 * reduce_rows launch of every dense-block BN_b (denseunet.py:249-251 backward): 78 launches of a DenseUNet-161 step. */
int hdu_bn_bwd_apply_sums(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, int64_t M, int C, const float* a,
                          const float* b, int relu, const float* mean, const float* rstd, int batch_stats, const float* gamma,
                          const float* beta, const float* sgamma, float* sums, int slots, float* dgamma, float* dbeta,
                          float* dsgamma, float* dsbeta, void* dx, int64_t lddx, int accumulate, float drop_keep,
                          uint32_t drop_seed, const uint32_t* drop_seed_dev, void* stream);

/* materialise z = relu?(a*x+b) (needed where the activation is consumed by pooling / as a skip / HFF operand) */
int hdu_affine_act(int dtype, const void* x, int64_t ldx, int64_t M, int C, const float* a, const float* b,
                   int relu, void* z, int64_t ldz, void* stream);

/* materialise the input of a conv:  out[n,d,h,w,c] = relu?(a[c]*x[n,d>>ud,h>>uh,w>>uw,c]+b[c]) + skip[n,d,h,w,c]
 * (BN(+Scale)+ReLU, nearest up-sampling and the skip add in ONE streaming pass; a/b and skip optional).
 * N,D,H,W are the stored (low-res) dims of x; out and skip have dims (D<<ud, H<<uh, W<<uw).  With the input
 * materialised, the conv itself runs as a pure async-DMA implicit GEMM (no VALU work on the operands). */
int hdu_materialize(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, const float* a,
                    const float* b, int relu, int ud, int uh, int uw, const void* skip, int64_t ldskip, void* out,
                    int64_t ldout, void* stream);

/* second half of hdu_bn_stats / hdu_bn_stats_fold when the first half ran in a conv epilogue (stats_partial of
 * hdu_conv_desc): sums the `slots` partial rows, mean = shift + S1/M, var = S2/M - (S1/M)^2 (biased, as tf.nn.moments),
 * and -- when a/b are given -- folds the BN(+Scale) and updates the moving statistics exactly like hdu_bn_stats_fold.
 * `shift` must be the array the epilogue used; it may alias `mean` (each channel is read before it is written). */
int hdu_bn_stats_finalize(const float* partial, int slots, int64_t M, int C, const float* shift, float* mean, float* var,
                          const float* gamma, const float* beta, float eps, const float* sgamma, const float* sbeta,
                          float* a, float* b, float* rstd, float* mov_mean, float* mov_var, float momentum, void* stream);

/* hdu_bn_stats_finalize of a slab SEGMENT (channels [seg_c0, seg_c0 + Cseg) of the moment arrays mean_all / var_all; the
 * epilogue's shift is mean_all + seg_c0 itself) followed by hdu_bn_fold of the NEXT BN over the whole slab [0, C_all) --
 * layer l+1's first BatchNormalization of a dense block (denseunet.py:229-262) reads the concatenation that layer l's
 * 3x3 conv has just extended.  One launch; results identical to the two separate calls. */
int hdu_bn_stats_finalize_fold_next(const float* partial, int slots, int64_t M, int Cseg, int seg_c0, int C_all,
                                    const float* shift_all, float* mean_all, float* var_all, const float* gamma,
                                    const float* beta, float eps, const float* sgamma, const float* sbeta, float* a, float* b,
                                    float* rstd, float* mov_mean, float* mov_var, float momentum, void* stream);

/* hdu_materialize with the BatchNormalization folded INSIDE the launch from a conv epilogue's statistics -- no finalize launch
 * between the producing conv and the pass that applies the BN (a finalize costs ~4.7 us of pure latency; DenseUNet-161 has
 * 162 of them per training step).  The BN's C input channels are the channels of x; the segment [seg_c0, seg_c0 + Cseg) of
 * them was just written by a conv whose epilogue left sum(y - shift), sum((y - shift)^2) in `partial` ([slots][2][Cseg],
 * slots <= 32), the other channels take the stored moments mean[c] / var[c] (dense blocks: the slab written by earlier
 * layers).  Every workgroup derives a / b for its own channels in registers; the first row block writes a, b, rstd, the
 * segment's mean / var and the moving averages, exactly as hdu_bn_stats_finalize(_fold_next) would have.
 * `shift` must not alias `mean` (the launch reads one while it writes the other): the caller refreshes shift <- mean between
 * steps (hdu_zero_regions copy entries).  Segment bounds: multiples of the 16-byte chunk; vectors 16-byte aligned. */
typedef struct hdu_stats_fold_desc {
  const float* partial;
  int32_t slots, Cseg, seg_c0, pad_;
  int64_t M;                 /* pixels the sums run over */
  const float* shift;        /* [C] */
  float* mean;               /* [C] in (non-segment channels) / out (segment) */
  float* var;
  const float* gamma; const float* beta; const float* sgamma; const float* sbeta;     /* [C]; any may be NULL */
  float eps, momentum;
  float* a; float* b; float* rstd; float* mov_mean; float* mov_var;                     /* [C]; rstd / mov_* may be NULL */
} hdu_stats_fold_desc;
int hdu_materialize_stats(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, const hdu_stats_fold_desc* f,
                          int relu, int ud, int uh, int uw, const void* skip, int64_t ldskip, void* out, int64_t ldout,
                          void* stream);

/* sync-BN over the depth shards of one volume.  buf: hdu_stats_sync_floats(C, world) floats = [world][1 + 2C].
 * hdu_stats_pack writes (n_local, mean, var) into slot `rank` and zeros into the other slots; the caller SUM-all-reduces buf
 * over the ranks (adding zeros is exact: an all-gather); hdu_stats_unpack combines the slots in rank order with
 *   mean = sum n_i mean_i / N,   var = sum n_i (var_i + (mean_i - mean)^2) / N      (no E[x^2] - E[x]^2 cancellation) */
size_t hdu_stats_sync_floats(int C, int world);
int hdu_stats_pack(int C, const float* mean, const float* var, int64_t n_local, int rank, int world, float* buf, void* stream);
int hdu_stats_unpack(int C, const float* buf, int world, float* mean, float* var, void* stream);

/* per-channel column sum: out[c] = sum_m x[m][c]   (bias gradients) */
int hdu_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int C, float* out, void* ws, size_t ws_bytes,
               void* stream);

/* ------------------------------------------------------------------ pooling / resampling
 * K.layers/pooling.py:166-433 -> TFB:3354-3432.  Max pool: ZeroPadding(1) + 3x3(x3) stride 2 VALID; the zero
 * padding takes part in the max (denseunet.py:169-170, denseunet3d.py:135-136).  D==1 selects the 2D window.
 * Avg pool: 2x2 stride 2 over (H,W); depth is not pooled (denseunet3d.py:102). */
int hdu_maxpool3s2_fwd(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, void* y,
                       int64_t ldy, uint8_t* argmax /* optional [N*Do*Ho*Wo][C]: winning tap, 255 = padding */,
                       int pad_d /* 1 = reference; 0 = depth halo planes supplied by the caller (depth sharding) */,
                       void* stream);
int hdu_maxpool3s2_bwd(int dtype, const uint8_t* argmax, const void* dy, int64_t lddy, int N, int D, int H, int W,
                       int C, void* dx, int64_t lddx, int accumulate, int pad_d, void* stream);
int hdu_avgpool2_fwd(int dtype, const void* x, int64_t ldx, int N, int D, int H, int W, int C, void* y,
                     int64_t ldy, void* stream);
int hdu_avgpool2_bwd(int dtype, const void* dy, int64_t lddy, int N, int D, int H, int W, int C, void* dx,
                     int64_t lddx, int accumulate, void* stream);
/* gradient of nearest up-sampling: dz[n,d,h,w,c] = sum over the 2^(ud+uh+uw) children of dxe (low-res dims given) */
int hdu_upsample_bwd(int dtype, const void* dxe, int64_t lddxe, int N, int D, int H, int W, int C, int ud, int uh,
                     int uw, void* dz, int64_t lddz, int accumulate, void* stream);

/* ------------------------------------------------------------------ loss  (loss.py:5-46)
 * rows i with label c_i in {0,1,2}: L = -(1/count) * sum_i w[c_i]*log(clip(softmax(z_i)[c_i],1e-10,1));
 * dlogits[i][j] = grad_scale*w[c_i]*(p_j - [j==c_i]) inside the clip range, else 0 (tf.clip_by_value gradient).
 * loss_sum receives sum_i w[c_i]*(-log p) (the caller divides by the global count); class_count[3] += #rows.
 * logits/dlogits have ld >= 3 (channels beyond 3 of dlogits are written as 0 up to C_pad). */
int hdu_wce_loss(int dtype, const void* logits, int64_t ldl, const uint8_t* labels, int64_t M, float w0, float w1,
                 float w2, float grad_scale, void* dlogits, int64_t lddl, int C_pad, float* loss_sum,
                 float* class_count, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ optimiser (K.optimizers.py:155-186)
 * v = momentum*v - lr*g ; p = p + momentum*v - lr*g  (Nesterov), g = grad_scale * grad. */
int hdu_sgd_nesterov(float* p, float* v, const float* g, int64_t n, float lr, float momentum, float grad_scale,
                     void* stream);

/* ------------------------------------------------------------------ 2.5D <-> 3D plumbing
 * denseunet3d.py:396-425 / hybridnet.py:382-411 (slice/concat/transpose lambdas).
 * vol: [D][H][W] float32 (one volume).  slab25d: out[k][h][w][0..2] = vol[clamp(k-1)], vol[k], vol[clamp(k+1)];
 * channels 3..Cpad-1 = 0. */
int hdu_slab25d(int dtype, const float* vol, int D, int H, int W, void* out, int Cpad, void* stream);
/* input3d[d][h][w] = (vol, scale*logit0, scale*logit1, scale*logit2, 0...) ; logits2d is [D][H][W][ldl] */
int hdu_make_input3d(int dtype, const float* vol, const void* logits2d, int64_t ldl, float scale, int D, int H,
                     int W, void* out, int Cpad, void* stream);
/* backward of the above w.r.t. logits2d: dlogits2d[...,j] (+)= scale * dinput3d[...,1+j] */
int hdu_make_input3d_bwd(int dtype, const void* dinput3d, int Cpad, float scale, int64_t M, void* dlogits2d,
                         int64_t lddl, int Cpad_l, int accumulate, void* stream);

/* dtype conversion / layout helpers for the boundary (float32 channels-last in, compute dtype out, and back) */
int hdu_cast_pad(int dtype, const float* src, int64_t M, int C, void* dst, int64_t lddst, int Cpad, void* stream);
int hdu_cast_out(int dtype, const void* src, int64_t ldsrc, int64_t M, int C, float* dst, void* stream);

/* z-sliding-window inference, per window (lib/funcs.py:31-34: `result = K.softmax(result); score[...] += K.eval(result)`):
 * score[m][j] += softmax(logits[m][0..2])[j] for j < num (num = 3 in test.py), logits [M][ldl] in the compute dtype (3 classes
 * in the first channels), score float32 [M][num].  One pass over the logits; nothing leaves the device. */
int hdu_softmax_accumulate(int dtype, const void* logits, int64_t ldl, int64_t M, int num, float* score, void* stream);

/* ------------------------------------------------------------------ per-step re-initialisation
 * The accumulators a training step adds into (epilogue statistics, fused BN-backward slot rows, the flat gradient buffer the
 * filter-gradient kernels atomically accumulate into, slab gradient buffers whose first writer covers only some channels,
 * the loss sums) are cleared by ONE launch at the head of the step, which also advances the device-side step counter the
 * dropout masks are seeded with.  Replaces what TensorFlow does implicitly by materialising fresh tensors every
 * session.run (K.engine/training.py:1715-1766) -- and the per-buffer torch fills the engine used before.
 * table: DEVICE array of n entries sorted by block_begin; entry i is cleared by blocks [block_begin[i], block_begin[i+1])
 * of HDU_ZERO_BLOCK_BYTES each; ptr 16-byte aligned, bytes a multiple of 4.  counter may be NULL. */
#define HDU_ZERO_BLOCK_BYTES 65536
typedef struct hdu_zero_entry {
  void* ptr;
  uint64_t bytes;
  uint32_t block_begin;
  uint32_t pad_;
  const void* src;            /* NULL: the region is cleared; else `bytes` are copied from src (16-byte aligned, no overlap) --
                               * the shifts of the epilogue statistics take last step's means in the same launch */
} hdu_zero_entry;
int hdu_zero_regions(const hdu_zero_entry* dev_table, int n, uint32_t total_blocks, uint32_t* counter, uint32_t counter_inc,
                     void* stream);
/* one region (ptr 16-byte aligned, bytes a multiple of 4) */
int hdu_zero(void* ptr, uint64_t bytes, void* stream);

/* ------------------------------------------------------------------ float32 operands as bf16 hi / lo image triples
 * Filter gradients of a float32 network in the split-bf16 modes (HDU_TUNE_F32_SPLIT; the reference is float32 throughout,
 * K.backend/common.py:4, and its filter gradient is tf.nn.conv2d_backprop_filter / conv3d_backprop_filter_v2, TFB:3097-3327).
 * x = hi + lo + r with hi, lo bfloat16 (round to nearest even), |r| <= 2^-17 |x|;  dy . x ~ dyh.xh + dyh.xl + dyl.xh, and because
 * the filter gradient contracts over PIXELS the three products are one product over three times the images:
 *   HDU_SPLIT3_OPERAND:  dst = [hi ; lo ; hi]   (the convolution input, optionally relu(a*x+b) of a BN prologue first)
 *   HDU_SPLIT3_GRADIENT: dst = [hi ; hi ; lo]   (the output gradient)
 * dst: bfloat16 [3][rows][C] dense (pixel stride C), src: float32 [rows][ld_src].  hdu_conv_wgrad / hdu_wgrad_plan_* then take a
 * bfloat16 descriptor with N' = 3 N images over the two dst buffers and add the float32 filter gradient as for a bfloat16 layer.
 * hdu_split3_entry_fill() fills ONE host-side table entry (block_begin = blocks of all entries before it) and returns its block
 * count; hdu_split3_batched() runs a DEVICE copy of the table (sorted by block_begin) as one launch, chl = the entries' `chl`
 * (one value per table: the lane layout hdu_split3_entry_fill chose).
 * C a multiple of 8, ld_src a multiple of 4, pointers 16-byte aligned, a / b both NULL (identity) or both set. */
#define HDU_SPLIT3_OPERAND 0
#define HDU_SPLIT3_GRADIENT 1
typedef struct hdu_split3_entry {
  const float* src;
  void* dst;
  const float* a;
  const float* b;
  uint64_t ld_src;
  uint64_t rows;
  uint32_t C;
  uint32_t relu;
  uint32_t pattern;
  uint32_t cols;           /* filled by hdu_split3_entry_fill: 8-channel columns per workgroup, column groups per row block */
  uint32_t col_groups;
  uint32_t block_begin;
  uint32_t iters;          /* row groups of 8 a thread walks */
  uint32_t chl;            /* channels per lane (8 or 4) */
} hdu_split3_entry;
int hdu_split3_entry_fill(hdu_split3_entry* e, const float* src, int64_t ld_src, int64_t rows, int C, const float* a,
                          const float* b, int relu, int pattern, void* dst, uint32_t block_begin, uint32_t* nblocks);
int hdu_split3_batched(const hdu_split3_entry* dev_table, int n, uint32_t total_blocks, int chl, void* stream);

/* ------------------------------------------------------------------ collectives (RCCL over xGMI)
 * The reference's only multi-GPU mechanism is in-graph tower replication (K.utils2/multi_gpu.py:7-69: the gradient sum is
 * implicit in tf.gradients, TFB:2310).  One process per GPU needs two exchanges, both on the caller's stream:
 *   - hdu_comm_allreduce_f32: in-place sum of the flat gradient buffer (data parallelism; the partial filter gradients of
 *     a depth-sharded volume);
 *   - hdu_comm_sendrecv: one grouped send/receive pair with each depth neighbour (halo planes forward, halo gradients
 *     back, the raw CT plane of the 2.5D slabs); a neighbour rank of -1 = the volume's edge.
 * Bootstrap: rank 0 calls hdu_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the launcher
 * has (h-denseunet_amd/parallel.py broadcasts them through the torch.distributed store); every rank then calls
 * hdu_comm_init.  RCCL is bound with dlopen at first use: a process that never calls these needs no librccl. */
typedef struct hdu_comm hdu_comm;
int hdu_comm_unique_id(void* id128);
int hdu_comm_init(hdu_comm** comm, int rank, int world, const void* id128);
int hdu_comm_destroy(hdu_comm* comm);
int hdu_comm_allreduce_f32(hdu_comm* comm, float* buf, int64_t n, void* stream);
int hdu_comm_sendrecv(hdu_comm* comm, int lo_rank, const void* send_lo, void* recv_lo, int hi_rank, const void* send_hi,
                      void* recv_hi, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDU_H_ */
