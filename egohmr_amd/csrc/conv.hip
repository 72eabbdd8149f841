// NHWC convolution (1x1 / 3x3, stride 1 or 2) as an implicit GEMM on the f16 matrix cores with f32-grade accuracy
// (split-f16 "f16x3": both operands as hi + lo f16, three v_mfma_f32_32x32x16_f16 per product, f32 accumulate - gcn_f16r.hip),
// with the BatchNorm-folded bias, the bottleneck's identity add and the ReLU fused into the epilogue.
//
// Serves the ResNet-50 backbone of the conditioning path (torchvision Bottleneck as used by models/egohmr/egohmr.py:183): the
// library path spent 11.3 ms in f32 Tensile GEMMs for the 1x1 convolutions, 11.5 ms in MIOpen's f32 3x3 kernels and 5.7 ms in the
// bias / add / ReLU passes per B=256 batch.
//
//   y[n,ho,wo,co] = act( sum_{kh,kw,ci} x[n, ho*s - p + kh, wo*s - p + kw, ci] * w[co,kh,kw,ci] + bias[co] (+ res[n,ho,wo,co]) )
//
// GEMM view: M = N*Ho*Wo rows, K = KH*KW*Ci (tap-major, Ci % 32 == 0 so a 32-wide K tile never straddles a tap), N = Co.
// Tile 128 x 128 x 32, 4 waves (2 x 2, 64 x 64 each), 2 blocks per CU.
//   A (activations, float32 in HBM): gathered by the threads (one float4 per thread and 32 rows; out-of-image taps and rows past M
//     read as zero), split into hi / lo halves in registers and written to the same XOR-swizzled LDS image the DMA-fed kernels use;
//     the loads of tile k+1 are issued before the MFMAs of tile k and converted after them.
//   B (weights, pre-split X2<32> [Co_pad][K], scaled by a power of two): global_load_lds DMA, double buffered.
//   Epilogue: through a float [128][128] LDS tile so that rows leave as 16-byte stores with the residual read the same way.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int CBM = 128, CBN = 128, CBK = 32;
constexpr int C_STAGE = (CBM + CBN) * CBK;   // floats: 32 KiB

struct ConvArgs {
  const float* x; const half_t* W; const float* bias; const float* res; float* y;
  int N, H, Wd, Ci, Ho, Wo, Co;
  int KH, KW, stride, pad, relu;
  float inv_scale;
  long long M;
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv_nhwc_split_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * C_STAGE];   // 64 KiB, the only LDS object

  const int n_tiles = (p.Co + CBN - 1) / CBN;
  const long long m_tiles = (p.M + CBM - 1) / CBM;
  const long long total = m_tiles * n_tiles;
  const long long bid = blockIdx.x;
  const long long lin = ((total & 7) == 0) ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;   // XCD b%8 owns a contiguous run of row tiles
  const long long m_tile = lin / n_tiles;
  const int n_tile = (int)(lin % n_tiles);
  const long long m0 = m_tile * CBM;
  const int n0 = n_tile * CBN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int K = p.KH * p.KW * p.Ci;
  const int cpt = p.Ci / CBK;                  // K tiles per tap
  const int KT = p.KH * p.KW * cpt;

  // ---- A gather: thread -> row (tid>>3) + 32 i, float4 q = tid & 7 of the 32-wide K tile
  const int q = tid & 7, ar = tid >> 3;
  long long abase[4];                          // element offset of x[n, ho*s - p, wo*s - p, 4q] (may point before the image)
  unsigned int amask[4];                       // bit (kh*KW + kw): that tap lies inside the image (0 for rows past M)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long gm = m0 + ar + 32 * i;
    amask[i] = 0u;
    abase[i] = 0;
    if (gm < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(gm / hw), rem = (int)(gm % hw);
      const int hi0 = (rem / p.Wo) * p.stride - p.pad, wi0 = (rem % p.Wo) * p.stride - p.pad;
      abase[i] = (((long long)n * p.H + hi0) * p.Wd + wi0) * p.Ci + 4 * q;
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if (hi0 + kh >= 0 && hi0 + kh < p.H && wi0 + kw >= 0 && wi0 + kw < p.Wd) amask[i] |= 1u << (kh * p.KW + kw);
    }
  }
  // LDS byte offsets of this thread's hi / lo 8-byte slots inside a stage (row r: 128 B; chunk c at (c ^ key(r)) * 16)
  int woff_hi[4], woff_lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ar + 32 * i, key = (r >> 1) & 7;
    woff_hi[i] = r * 128 + (((q >> 1) ^ key) << 4) + ((q & 1) << 3);
    woff_lo[i] = r * 128 + (((4 + (q >> 1)) ^ key) << 4) + ((q & 1) << 3);
  }
  f32x4 areg[4];
  auto a_load = [&](int kt) {
    const int tap = kt / cpt, ci0 = (kt % cpt) * CBK;
    const int kh = tap / p.KW, kw = tap % p.KW;
    const long long toff = ((long long)kh * p.Wd + kw) * p.Ci + ci0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      areg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((amask[i] >> tap) & 1u) areg[i] = *(const f32x4*)(p.x + abase[i] + toff);
    }
  };
  auto a_store = [&](int buf) {
    char* base = (char*)(lds + buf * C_STAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      half4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = fminf(fmaxf(areg[i][e], -65504.f), 65504.f);
        hi[e] = (half_t)c;
        lo[e] = (half_t)(areg[i][e] - (float)hi[e]);
      }
      *(u32x2*)(base + woff_hi[i]) = __builtin_bit_cast(u32x2, hi);
      *(u32x2*)(base + woff_lo[i]) = __builtin_bit_cast(u32x2, lo);
    }
  };

  // ---- B DMA: 4 wave-instructions of 8 rows per wave; row r_i = r0 + 32 i -> one swizzle key
  const int ld_r = lane >> 3, ld_c = lane & 7;
  const int r0 = 8 * wave + ld_r;
  const int swz = (ld_c ^ ((r0 >> 1) & 7)) << 2;
  const float* pW = (const float*)p.W + ((size_t)n0 + r0) * K + swz;
  auto b_stage = [&](int buf, int kt) {
    float* base = lds + buf * C_STAGE + CBM * CBK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pW + (size_t)i * 32 * K + (size_t)kt * CBK), (AS3 void*)(base + (wave + 4 * i) * 256), 16, 0, 0);
  };

  const int mi = lane & 31, g = lane >> 5;
  const int rA = 64 * wm + mi, rB = 64 * wn + mi;
  const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;

  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  b_stage(0, 0);
  a_load(0);
  a_store(0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // B of tile kt has landed
    __syncthreads();                                     // ... everybody's A rows of tile kt are written; stage (kt+1)&1 is free
    if (kt + 1 < KT) {
      b_stage((kt + 1) & 1, kt + 1);
      a_load(kt + 1);                                    // in flight under this tile's MFMAs
    }
    const float* As = lds + (kt & 1) * C_STAGE;
    const float* Bs = As + CBM * CBK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = 2 * s + g, cl = 4 + 2 * s + g;
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *(const half8*)(As + (rA + 32 * t) * CBK + ((ch ^ keyA) << 2));
        al[t] = *(const half8*)(As + (rA + 32 * t) * CBK + ((cl ^ keyA) << 2));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bh[u] = *(const half8*)(Bs + (rB + 32 * u) * CBK + ((ch ^ keyB) << 2));
        bl[u] = *(const half8*)(Bs + (rB + 32 * u) * CBK + ((cl ^ keyB) << 2));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // small cross terms first, leading term last
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) a_store((kt + 1) & 1);              // the other stage: everybody left it at the barrier above
  }

  // ---- epilogue: accumulators (lane = output channel, registers = rows) -> float [128][128] LDS tile -> 16-byte rows
  __syncthreads();                                       // every wave is done with the operand stages
  float* T = lds;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int col = 64 * wn + 32 * u + mi;
    const float add = (n0 + col < p.Co && p.bias) ? p.bias[n0 + col] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 64 * wm + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
        T[row * CBN + col] = fmaf(acc[t][u][r], p.inv_scale, add);
      }
  }
  __syncthreads();
  // work item = (row, 8 consecutive channels): 128 x 16 items, 8 per thread; lanes with c8 >= 8 read their two 16-byte halves in
  // the opposite order, which makes every ds_read_b128 lane group hit 16 distinct 16-byte slots of the 256-byte bank window
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = tid + 256 * i, row = u >> 4, c8 = u & 15;
    const long long gm = m0 + row;
    const int co = n0 + 8 * c8;
    const float* src = T + row * CBN + 8 * c8;
    const int flip = c8 >> 3;
    const f32x4 va = *(const f32x4*)(src + 4 * flip), vb = *(const f32x4*)(src + 4 * (1 - flip));
    f32x4 v0 = flip ? vb : va, v1 = flip ? va : vb;
    if (gm >= p.M || co >= p.Co) continue;
    float* dst = p.y + gm * p.Co + co;
    if (p.res) {
      const float* rs = p.res + gm * p.Co + co;
      v0 += *(const f32x4*)rs;
      v1 += *(const f32x4*)(rs + 4);
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
    }
    *(f32x4*)dst = v0;
    *(f32x4*)(dst + 4) = v1;
  }
}

}  // namespace

extern "C" int ehm_conv_nhwc_split(const ehm_conv_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->x && d->W && d->y);
  EHM_CHECK_ARG(d->N > 0 && d->H > 0 && d->Wd > 0 && d->Ci > 0 && d->Co > 0 && d->Ci % CBK == 0 && d->Co % 8 == 0);
  EHM_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 32 && (d->stride == 1 || d->stride == 2) && d->pad >= 0);
  EHM_CHECK_ARG(d->w_scale > 0.f);
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Wd + 2 * d->pad - d->KW) / d->stride + 1;
  EHM_CHECK_ARG(Ho > 0 && Wo > 0);
  ConvArgs a;
  a.x = d->x; a.W = (const half_t*)d->W; a.bias = d->bias; a.res = d->residual; a.y = d->y;
  a.N = d->N; a.H = d->H; a.Wd = d->Wd; a.Ci = d->Ci; a.Ho = Ho; a.Wo = Wo; a.Co = d->Co;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.relu = d->relu;
  a.inv_scale = 1.f / d->w_scale;
  a.M = (long long)d->N * Ho * Wo;
  const long long blocks = ceil_div(a.M, CBM) * ceil_div(d->Co, CBN);
  EHM_CHECK_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(conv_nhwc_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  EHM_LAUNCH_CHECK();
  return 0;
}
