// ResNet-50 stem in one pass: conv 7x7 / stride 2 / pad 3 (3 -> 64, BatchNorm folded by the caller) + bias + ReLU + max-pool 3x3 /
// stride 2 / pad 1, NCHW float32 image in, NHWC [N, H/4, W/4, 64] out as float32 or X2 split rows (torchvision ResNet.forward's conv1 /
// bn1 / relu / maxpool as used by models/resnet.py:139-150 -> models/egohmr/egohmr.py:183).  Replaces a library convolution (2.3 ms at
// B = 256), an NCHW max-pool (1.0 ms), a bias/ReLU pass (0.4 ms) and the NCHW -> NHWC permute of the pooled tensor.
// A small pre-pass copies the image into a zero-padded [N,3,H+8,W+8] scratch (5 left / top, 3 right / bottom) so that every load is in
// bounds and the inner loop has no border conditions; conv pixels that are POOL padding (row / column -1) are zero, which equals torch's
// -inf padding because every window also holds a real ReLU output (>= 0).
// (Rounds 2-3 ran the conv on the vector ALU - lane = output channel, image slices through scalar loads, v_pk_fma_f32: 1.07 ms; that kernel
//  was superseded by the matrix-core one below in round 4 and removed in round 5, git history has it.)
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------------------------------------------
// The same stem on the MATRIX cores (round 4): conv 7x7 / 2 as an implicit GEMM with K = (ci, kh) x 8 = 21 rows of eight taps (kw 0..6 and a
// zero weight) padded to 22 = 11 k-steps of v_mfma_f32_32x32x16_f16, both operands split hi + lo (three MFMAs per product, f32 accumulate:
// f32-grade like the trunk).  Lane (pixel p, half g) of k-step s reads the eight consecutive padded-image floats of row (ci, kh) = 2 s + g that
// start at the pixel's first tap - two 16-byte global loads, L1 / L2 absorb the overlap between neighbouring pixels - and splits them in
// registers; the weights sit in LDS as ready-made B fragments.  A block = 7 x 14 pooled pixels = a 15 x 29 patch of conv pixels (16 MFMA row
// tiles of 4 x 8 pixels, four per wave); bias + ReLU + the register-local part of the pool leave the accumulators (lane = channel) as 9
// unsigned-max LDS atomics per lane and tile into the pooled tile (ReLU outputs are >= 0: float order = bit-pattern order; pool padding
// contributes nothing), which then leaves as X2 / float32 NHWC rows.  (One atomic per conv pixel and pooled cell - 64 per lane - made the
// LDS atomics the bound: 0.67 ms.)
// 0.19 TFLOP issued -> ~0.1 ms of matrix time against 1.14 ms (alone) / 2.1-2.3 ms (beside the PointNet) of the vector-ALU kernel of rounds 2-3.
typedef _Float16 st_half8 __attribute__((ext_vector_type(8)));
constexpr int MP_R = 7, MP_C = 14;                       // pooled tile of a block
constexpr int MC_R = 2 * MP_R + 1, MC_C = 2 * MP_C + 1;  // conv patch 15 x 29
constexpr int M_TI = (MC_R + 3) / 4, M_TJ = (MC_C + 7) / 8;   // MFMA row tiles = 4 x 8 conv patches: 4 x 4 of them
constexpr int M_TILES = M_TI * M_TJ;                     // 16
constexpr int M_KSTEPS = 11;
constexpr int M_WFRAG = M_KSTEPS * 2 * 2 * 64;           // half8 fragments: [k-step][n-tile][hi / lo][lane]

// Wt [147][64] (k = (ci * 7 + kh) * 7 + kw) -> B fragments: lane l of (s, nt, hl): channel 32 nt + (l & 31), row q = 2 s + (l >> 5), taps kw 0..7
__global__ void stem_pack_w_kernel(const float* __restrict__ Wt, st_half8* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M_KSTEPS * 2 * 64) return;
  const int lane = i & 63, nt = (i >> 6) & 1, s = i >> 7;
  const int q = 2 * s + (lane >> 5), ch = 32 * nt + (lane & 31);
  st_half8 hi, lo;
#pragma unroll
  for (int kw = 0; kw < 8; ++kw) {
    const float w = (q < 21 && kw < 7) ? Wt[(q * 7 + kw) * 64 + ch] : 0.f;
    hi[kw] = (_Float16)w;
    lo[kw] = (_Float16)(w - (float)hi[kw]);
  }
  out[((s * 2 + nt) * 2 + 0) * 64 + lane] = hi;
  out[((s * 2 + nt) * 2 + 1) * 64 + lane] = lo;
}

__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const float* __restrict__ pad, const st_half8* __restrict__ wfrag, const float* __restrict__ bias,
                                                            float* __restrict__ y, int H, int W, int tiles_c, int n_patches, int out_x2) {
  __shared__ __attribute__((aligned(16))) st_half8 sW[M_WFRAG];          // 45 KiB
  __shared__ unsigned int pool[MP_R * MP_C * 64];                        // 28 KiB: pooled maxima as bit patterns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hq = H / 4, Wq = W / 4, Hc = H / 2, Wc = W / 2, Wp = W + SP_X, Hp = H + SP_X;
  const int tiles_r = (Hq + MP_R - 1) / MP_R;
  for (int i = tid; i < M_WFRAG; i += 256) sW[i] = wfrag[i];             // once per (persistent) block
  const int g = lane >> 5, mi = lane & 31;
  const float b0 = bias[mi], b1 = bias[32 + mi];
  for (int patch = blockIdx.x; patch < n_patches; patch += gridDim.x) {
  int b = patch;
  const int tc = b % tiles_c; b /= tiles_c;
  const int tr = b % tiles_r;
  const int n = b / tiles_r;
  const int pr0 = MP_R * tr, pc0 = MP_C * tc;                            // first pooled pixel of the patch
  __syncthreads();                                                        // (the previous patch's pooled tile has been written out)
  for (int i = tid; i < MP_R * MP_C * 64; i += 256) pool[i] = 0u;
  __syncthreads();
  const float* img = pad + (size_t)n * 3 * Hp * Wp;
  for (int t = wave; t < M_TILES; t += 4) {
    // An MFMA row tile = a 4 x 8 patch of conv pixels: tile pixel i -> (row i >> 3, column 4 ((i >> 2) & 1) + (i & 3)), so that with the C
    // layout (register k <-> tile pixel (k & 3) + 8 (k >> 2) + 4 g) a lane holds a 4 x 4 sub-patch of ONE channel: row k >> 2, column k & 3.
    // The 3 x 3 / stride-2 pool of that sub-patch is register-local up to its rim: 9 partial maxima per lane instead of 64 LDS atomics.
    const int ti = t / M_TJ, tj = t - ti * M_TJ;
    const int arow = 4 * ti + (mi >> 3), acol = 8 * tj + 4 * ((mi >> 2) & 1) + (mi & 3);     // my A-operand pixel (patch coordinates)
    int r = 2 * pr0 - 1 + arow, c = 2 * pc0 - 1 + acol;
    r = r < 0 ? 0 : (r >= Hc ? Hc - 1 : r);                              // (clamped for the loads; masked below)
    c = c < 0 ? 0 : (c >= Wc ? Wc - 1 : c);
    const float* px = img + (size_t)(2 * r + 2) * Wp + (2 * c + 2);      // tap (ci = 0, kh = 0, kw = 0) of the padded image
    f32x16 acc0, acc1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int s = 0; s < M_KSTEPS; ++s) {
      int q = 2 * s + g;
      q = q > 20 ? 20 : q;                                               // (row 21 is padding: its weights are zero)
      const float* src = px + ((size_t)(q / 7) * Hp + (q % 7)) * Wp;
      const f32x4u x0 = *(const f32x4u*)src, x1 = *(const f32x4u*)(src + 4);
      const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      st_half8 ah, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ah[e] = (_Float16)xv[e];
        al[e] = (_Float16)(xv[e] - (float)ah[e]);
      }
      const st_half8 bh0 = sW[((s * 2 + 0) * 2 + 0) * 64 + lane], bl0 = sW[((s * 2 + 0) * 2 + 1) * 64 + lane];
      const st_half8 bh1 = sW[((s * 2 + 1) * 2 + 0) * 64 + lane], bl1 = sW[((s * 2 + 1) * 2 + 1) * 64 + lane];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, acc1, 0, 0, 0);
    }
    // my 4 x 4 sub-patch: patch rows 4 ti + kk, columns c0 + dc (c0 = 8 tj + 4 g); bias + ReLU, pool padding / out-of-image pixels -> 0
    // (neutral: every pooled window also holds a real ReLU output >= 0)
    const int c0 = 8 * tj + 4 * g, r0 = 4 * ti;
    float v0[4][4], v1[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int dc = 0; dc < 4; ++dc) {
        const int cr = 2 * pr0 - 1 + r0 + kk, cc = 2 * pc0 - 1 + c0 + dc;
        const bool ok = r0 + kk < MC_R && c0 + dc < MC_C && cr >= 0 && cr < Hc && cc >= 0 && cc < Wc;
        v0[kk][dc] = ok ? fmaxf(acc0[4 * kk + dc] + b0, 0.f) : 0.f;
        v1[kk][dc] = ok ? fmaxf(acc1[4 * kk + dc] + b1, 0.f) : 0.f;
      }
    // pooled cell (rr, cq) = max over patch rows 2 rr .. 2 rr + 2, columns 2 cq .. 2 cq + 2.  Of my rows r0 .. r0 + 3 (r0 % 4 == 0):
    // cell row r0/2 - 1 sees {r0}, r0/2 sees {r0, r0+1, r0+2}, r0/2 + 1 sees {r0+2, r0+3}; columns likewise.
    auto pool3 = [&](const float (&v)[4][4], int choff) {
      float cm[4][3];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        cm[kk][0] = v[kk][0];
        cm[kk][1] = fmaxf(fmaxf(v[kk][0], v[kk][1]), v[kk][2]);
        cm[kk][2] = fmaxf(v[kk][2], v[kk][3]);
      }
#pragma unroll
      for (int a3 = 0; a3 < 3; ++a3) {
        const int rr = r0 / 2 - 1 + a3;
        if (rr < 0 || rr >= MP_R) continue;
#pragma unroll
        for (int b3 = 0; b3 < 3; ++b3) {
          const int cq = c0 / 2 - 1 + b3;
          if (cq < 0 || cq >= MP_C) continue;
          const float m = a3 == 0 ? cm[0][b3] : (a3 == 1 ? fmaxf(fmaxf(cm[0][b3], cm[1][b3]), cm[2][b3]) : fmaxf(cm[2][b3], cm[3][b3]));
          atomicMax(&pool[(rr * MP_C + cq) * 64 + choff + mi], __builtin_bit_cast(unsigned int, m));
        }
      }
    };
    pool3(v0, 0);
    pool3(v1, 32);
  }
  __syncthreads();
  // the pooled tile as NHWC rows: thread = (pooled pixel, 8 consecutive channels)
  for (int i = tid; i < MP_R * MP_C * 8; i += 256) {
    const int pix = i >> 3, c8 = (i & 7) * 8;
    const int prow = pr0 + pix / MP_C, pcol = pc0 + pix % MP_C;
    if (prow >= Hq || pcol >= Wq) continue;
    const size_t row = ((size_t)n * Hq + prow) * Wq + pcol;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(float, pool[pix * 64 + c8 + e]);
    if (out_x2) {
      st_half8 hh, ll;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        hh[e] = (_Float16)fminf(v[e], 65504.f);
        ll[e] = (_Float16)(v[e] - (float)hh[e]);
      }
      _Float16* d2 = (_Float16*)y + row * 128 + (c8 >> 5) * 64 + (c8 & 31);      // X2 rows of 64 channels: per 32 channels 32 hi | 32 lo
      *(st_half8*)d2 = hh;
      *(st_half8*)(d2 + 32) = ll;
    } else {
      float* d = y + row * 64 + c8;
      *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  }   // patches
}

}  // namespace

extern "C" size_t ehm_resnet_stem_scratch_bytes(int N, int H, int W) {
  return round_up((int64_t)N * 3 * (H + SP_X) * (W + SP_X) * sizeof(float), 256) + (size_t)M_WFRAG * 16;   // padded image | weight fragments
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
  st_half8* wfrag = (st_half8*)((char*)scratch + round_up((int64_t)N * 3 * (H + SP_X) * (W + SP_X) * sizeof(float), 256));
  hipLaunchKernelGGL(stem_pack_w_kernel, dim3((unsigned)ceil_div(M_KSTEPS * 2 * 64, 256)), dim3(256), 0, st, Wt, wfrag);
  const int tiles_r = (H / 4 + MP_R - 1) / MP_R, tiles_c = (W / 4 + MP_C - 1) / MP_C;
  const int n_patches = N * tiles_r * tiles_c, slots = 2 * ehm_num_cus();
  hipLaunchKernelGGL(stem_mfma_kernel, dim3((unsigned)(n_patches < slots ? n_patches : slots)), dim3(256), 0, st, (const float*)scratch, (const st_half8*)wfrag, bias,
                     y, H, W, tiles_c, n_patches, out_x2);
  EHM_LAUNCH_CHECK();
  return 0;
}
