// ResNet-50 stem in one pass: conv 7x7 / stride 2 / pad 3 (3 -> 64, BatchNorm folded by the caller) + bias + ReLU + max-pool 3x3 /
// stride 2 / pad 1, NCHW float32 image in, NHWC float32 [N, H/4, W/4, 64] out (torchvision ResNet.forward's conv1 / bn1 / relu /
// maxpool as used by models/resnet.py:139-150 -> models/egohmr/egohmr.py:183).  Replaces a library convolution (2.3 ms at
// B = 256), an NCHW max-pool (1.0 ms), a bias/ReLU pass (0.4 ms) and the NCHW -> NHWC permute of the pooled tensor.
//
// Ci = 3 leaves nothing for an implicit GEMM to tile (K = 147), so the conv runs on the vector ALU in float32 with the roles
// chosen so that nothing but FMAs sits in the inner loop:
//   * lane = output channel (64 lanes = the 64 channels); the lane's 147 weights live in VGPRs for the wave's whole life;
//   * the image is wave-uniform data: a (ci, kh) slice of the input row segment is fetched with SCALAR loads (s_load_dwordx16,
//     K-cache) into SGPRs and enters v_pk_fma_f32 as an SGPR-pair operand (two taps of one pixel per instruction, two partial
//     sums per pixel) - no LDS, no per-lane address arithmetic; 1428 packed FMAs per conv row of 17 pixels;
//   * a wave walks down 17 conv rows of a 17-pixel-wide strip (16 + 1 halo each way for the pool), keeps the 17 accumulators of
//     the current row in registers, folds bias / ReLU / the 3-wide column maximum and carries the 3-row maximum in 8 registers:
//     the pool is register-local because a lane owns one channel; every pooled pixel leaves as one 256-byte row of 64 lanes.
// A small pre-pass copies the image into a zero-padded [N,3,H+8,W+8] scratch (5 left / top, 3 right / bottom) so that every
// scalar load is in bounds and 32-byte aligned and the inner loop has no border conditions; conv pixels that are POOL padding
// (row / column -1) are zero, which equals torch's -inf padding because every window also holds a real ReLU output (>= 0).
#include "common.h"
#include "egohmr_hip.h"
#include "internal.h"

namespace {

constexpr int SP_L = 5, SP_X = 8;     // left / top padding, total extra columns / rows of the scratch image

// one thread = four consecutive columns of the padded image (W + 8 is a multiple of 4: one 16-byte store)
__global__ __launch_bounds__(256) void stem_pad_kernel(const float* __restrict__ img, float* __restrict__ pad, int H, int W, long long total4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int Wp4 = (W + SP_X) / 4, Hp = H + SP_X;
  const int c0 = 4 * (int)(i % Wp4), r = (int)((i / Wp4) % Hp);
  const long long plane = i / ((long long)Wp4 * Hp);
  const int sr = r - SP_L;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (sr >= 0 && sr < H) {
    const float* src = img + (plane * H + sr) * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int sc = c0 + k - SP_L;
      if (sc >= 0 && sc < W) v[k] = src[sc];
    }
  }
  *(f32x4*)(pad + 4 * i) = v;
}

typedef const float __attribute__((address_space(4))) cfloat;

// wave task = (image, strip of 16 conv columns, chunk of 16 conv rows); 4 tasks per block
__global__ __launch_bounds__(256, 2) void stem_conv_pool_kernel(const float* __restrict__ pad, const float* __restrict__ Wt,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                                 int strips, int chunks, int tasks, int out_x2) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int Hc = H / 2, Wc = W / 2, Hq = H / 4, Wq = W / 4;      // conv / pooled extents
  const int Wp = W + SP_X, Hp = H + SP_X;
  int b = 4 * (int)blockIdx.x + wave;
  if (b >= tasks) return;
  const int chunk = b % chunks; b /= chunks;                      // pooled rows 8 chunk .. +7, conv rows 16 chunk - 1 .. + 15
  const int strip = b % strips;
  const int n = b / strips;
  (void)Hc; (void)Wc;

  // the lane's weights as 21 slices (ci, kh) of four register PAIRS (kw 0|1, 2|3, 4|5, 6|zero): one v_pk_fma_f32 multiplies two taps
  // of one output pixel into two partial sums
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 w[21][4];
#pragma unroll
  for (int sl = 0; sl < 21; ++sl)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      w[sl][h][0] = Wt[(sl * 7 + 2 * h) * 64 + lane];
      w[sl][h][1] = h < 3 ? Wt[(sl * 7 + 2 * h + 1) * 64 + lane] : 0.f;
    }
  const float bs = bias[lane];

  // conv pixel i of the strip = conv column 16 strip - 1 + i; its taps are padded columns 32 strip + 2 i + kw (kw = 0..6)
  const float* img_n = pad + (size_t)n * 3 * Hp * Wp + 32 * strip;
  float m[8];                                                      // running 3-row maximum of the column-pooled values
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = 0.f;
  float* yrow = y + (((size_t)n * Hq + 8 * chunk) * Wq + 8 * strip) * 64 + lane;

  for (int i = 0; i < 17; ++i) {
    const int r = 16 * chunk - 1 + i;                             // conv row; its taps are padded rows 2 r + 2 + kh
    float cm[8];
    if (r < 0) {                                                   // pool padding
#pragma unroll
      for (int j = 0; j < 8; ++j) cm[j] = 0.f;
    } else {
      f32x2 acc2[17];
#pragma unroll
      for (int p = 0; p < 17; ++p) acc2[p] = f32x2{bs, 0.f};
      // slice sl = (ci, kh): 40 wave-uniform floats (padded columns 32 strip .. + 39) in SGPR pairs, double buffered: the next slice is
      // requested before the current one is multiplied and a scheduling barrier keeps it there (left alone, hipcc sinks every slice's
      // loads down to its FMAs and each slice eats a scalar-cache round trip: 1.43 vs 1.14 ms).  The loads stay compiler-visible -
      // its own lgkmcnt bookkeeping then also covers any SGPR it decides to spill (hand-written s_load + s_waitcnt asm did not: a spill
      // of the not-yet-arrived registers corrupted pixels as soon as one more kernel argument raised the SGPR pressure).
      typedef const f32x2 __attribute__((address_space(4))) cf32x2;
      const float* row0 = img_n + (size_t)(2 * r + 2) * Wp;
      auto load_slice = [&](f32x2 (&x)[20], int sl) {
        cf32x2* q = (cf32x2*)(uintptr_t)(row0 + ((size_t)(sl / 7) * Hp + (sl % 7)) * Wp);
#pragma unroll
        for (int j = 0; j < 20; ++j) x[j] = q[j];
      };
      auto fma_slice = [&](const f32x2 (&x)[20], int sl, int first, int last) {   // FMAs number first .. last - 1 of the slice's 68
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
          for (int p = 0; p < 17; ++p)                             // taps kw = 2 h, 2 h + 1 of pixel p: columns 2 p + 2 h, + 1 (column 39 only meets the zero weight)
            if (17 * h + p >= first && 17 * h + p < last)
              asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[p]) : "s"(x[p + h]), "v"(w[sl][h]));
      };
      // Scalar loads return out of order, so every wait is lgkmcnt(0): the wait for slice k must come BEFORE slice k + 1 is requested
      // or it waits for that one too.  Order per slice: first FMA of k (the compiler puts the wait in front of it), request k + 1,
      // the other 67 FMAs.
      f32x2 xa[20], xb[20];
      load_slice(xa, 0);
#pragma unroll
      for (int sl = 0; sl < 21; sl += 2) {
        __builtin_amdgcn_sched_barrier(0);
        fma_slice(xa, sl, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (sl + 1 < 21) load_slice(xb, sl + 1);
        __builtin_amdgcn_sched_barrier(0);
        fma_slice(xa, sl, 1, 68);
        if (sl + 1 < 21) {
          __builtin_amdgcn_sched_barrier(0);
          fma_slice(xb, sl + 1, 0, 1);
          __builtin_amdgcn_sched_barrier(0);
          if (sl + 2 < 21) load_slice(xa, sl + 2);
          __builtin_amdgcn_sched_barrier(0);
          fma_slice(xb, sl + 1, 1, 68);
        }
      }
      float acc[17];
#pragma unroll
      for (int p = 0; p < 17; ++p) acc[p] = fmaxf(acc2[p][0] + acc2[p][1], 0.f);
      if (strip == 0) acc[0] = 0.f;                                // conv column -1: pool padding
#pragma unroll
      for (int j = 0; j < 8; ++j) cm[j] = fmaxf(fmaxf(acc[2 * j], acc[2 * j + 1]), acc[2 * j + 2]);
    }
    if ((i & 1) == 0) {                                            // row 2 j: closes pooled row j - 1, opens pooled row j
      if (i > 0) {
        float* dst = yrow + (size_t)(i / 2 - 1) * Wq * 64;
        if (out_x2) {                                              // X2 rows for ehm_conv_x2: per 32 channels 32 hi halves | 32 lo halves
          _Float16* d2 = (_Float16*)(dst - lane) + ((lane >> 5) * 64 + (lane & 31));
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = fmaxf(m[j], cm[j]);
            const _Float16 hi = (_Float16)fminf(v, 65504.f);
            d2[j * 128] = hi;
            d2[j * 128 + 32] = (_Float16)(v - (float)hi);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) dst[j * 64] = fmaxf(m[j], cm[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = cm[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], cm[j]);
    }
  }
}

}  // namespace

extern "C" size_t ehm_resnet_stem_scratch_bytes(int N, int H, int W) {
  return (size_t)N * 3 * (H + SP_X) * (W + SP_X) * sizeof(float);
}

extern "C" int ehm_resnet_stem(const float* img, const float* Wt, const float* bias, float* scratch, float* y, int N, int H, int W,
                               int out_x2, void* stream) {
  EHM_CHECK_ARG(img && Wt && bias && scratch && y && N > 0 && H > 0 && W > 0);
  if (H % 32 != 0 || W % 32 != 0) {
    ehm_set_error("ehm_resnet_stem needs H %% 32 == 0 and W %% 32 == 0 (got %d x %d)", H, W);
    return EHM_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long total4 = (long long)N * 3 * (H + SP_X) * ((W + SP_X) / 4);
  hipLaunchKernelGGL(stem_pad_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, st, img, scratch, H, W, total4);
  EHM_LAUNCH_CHECK();
  const int strips = W / 32, chunks = H / 32, tasks = N * strips * chunks;
  hipLaunchKernelGGL(stem_conv_pool_kernel, dim3((unsigned)ceil_div(tasks, 4)), dim3(256), 0, st, scratch, Wt, bias, y, H, W, strips, chunks,
                     tasks, out_x2);
  EHM_LAUNCH_CHECK();
  return 0;
}
