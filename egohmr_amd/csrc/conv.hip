// NHWC convolution (1x1 / 3x3, stride 1 or 2) as an implicit GEMM on the f16 matrix cores with f32-grade accuracy
// (split-f16 "f16x3": both operands as hi + lo f16, three v_mfma_f32_32x32x16_f16 per product, f32 accumulate - gcn_tile.hip),
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
#include <type_traits>

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


// ------------------------------------------------------------------------------------------------------------------------------
// The same convolution on activations that STAY in the X2 split format between the layers ([pixels][C], 128 bytes = 32 hi halves |
// 32 lo halves per 32 channels; rows padded to the 192-row tile plus ONE all-zero row).  One K tile of the implicit GEMM =
// (tap, 32-channel group) = exactly one 128-byte chunk of an input pixel, so the activation operand needs no registers at all: each
// lane of a global_load_lds instruction points at its (output row, tap) pixel - or at the zero row when the tap falls outside the
// image or the row is past M - and the DMA gathers the tile.  The float32 path above re-reads and re-SPLITS every input value once
// per tap in the vector ALU.  Engine = linear.hip's linear_tile_kernel: persistent blocks, 192 x 128 tiles, next tile's DMA before
// the epilogue, barrier-free epilogue through each wave's own pieces of stage 1, residual read and output written as X2 rows.
constexpr int XBM = 192, XRK = 32;
constexpr int XA_T = XBM * XRK;                                          // floats of a stage's activation region (24 KiB)
constexpr int x_stage_floats(int nu) { return XA_T + 64 * nu * XRK; }    // + 8 KiB of weights per 64 output channels

struct ConvX2Args {
  const float* x; const float* W; const float* bias; const char* res; char* y;   // x, W, res, y: X2 rows addressed as floats / bytes
  int N, H, Wd, Ci, Ho, Wo, Co;
  int KH, KW, stride, pad, relu;
  float inv_scale;
  long long M;
  unsigned int zero_off;       // float offset of the all-zero row of x
  // stream-K (SK = true): the tiles' K tiles are dealt out to the blocks as ONE sequence in equal contiguous runs (pairs of K tiles), so a
  // tile may be computed by two or three blocks; the block that holds a tile's FIRST K tiles finishes it (adds the others' partial sums in k
  // order, epilogue), the others leave their accumulators in sk_part and count in sk_flags.
  unsigned int* sk_flags;      // [cut tiles] arrival counters: zero at launch, zeroed again by the block that consumed them (bit 31: poisoned by a time-out)
  float* sk_part;              // [tiles][kSkMaxParts][acc floats per thread][256]
  int sk_per;                  // pairs of K tiles per block
  int sk_rounds;               // whole tiles first: rounds [0, sk_rounds) of the grid run one whole tile per block (the plain schedule), only the tiles BEHIND them - the
                               // last, partly filled round: e.g. 12 of 524 tiles on 512 slots - are dealt out as K runs (0: every tile is, the long-K plan)
  // second K segment (DS = true): a 1 x 1 convolution of ANOTHER tensor x2 [N, H2, W2, Ci2] with stride stride2, accumulated into the same
  // output - a bottleneck's projection shortcut inside its last conv (torchvision Bottleneck.forward: out = conv3(.) + downsample(x)): the
  // weights are [W | W2] along K, the K tiles past the first segment gather x2's pixels (2 stride2 ho, stride2 wo)
  const float* x2;
  int H2, W2, Ci2, stride2;
  unsigned int zero_off2;
};
constexpr int kSkMaxParts = 3;
constexpr int kSkFlagBytes = 4096;         // arrival counters of up to 1023 cut tiles at the head of the workspace ...
constexpr int kSkPoisonWord = kSkFlagBytes / 4 - 1;   // ... and, in its last word, the sticky count of hand-off time-outs (ehm_conv_x2_workspace_status)
constexpr int kSkHandoffKTiles = 14;   // what a cut tile's hand-off costs, in K-tile times (measured, see sk_plan)

template <int NU>
struct XFrags {
  half8 ah[3], al[3], bh[NU], bl[NU];
};

// NU = 2: 192 x 128 tiles (96 x 64 per wave); NU = 1: 192 x 64 tiles (96 x 32 per wave) for Co = 64 layers (no padding columns through
// the matrix cores).
// HO ("hi only", ehm_conv_x2_desc.hi_only): the plain-f16 tier of the encoders (BASELINE config 5's fp16 tier; NOT parity grade: 0.4 - 1.4 mm of final
// vertex, docs/EXPERIMENTS.md 3.3) on the SAME buffers - only the hi halves of activations and weights are fetched (half of every 128-byte chunk: the lanes
// of an operand piece that carry lo chunks are masked off), multiplied (one MFMA per product instead of three), and written; lo halves are don't-care.
template <int NU, bool SK, bool DS = false, bool HO = false>
__global__ __launch_bounds__(256, 2) void conv_x2_tile_kernel(ConvX2Args p) {
  constexpr int XBN = 64 * NU, XSTG = x_stage_floats(NU);
  __shared__ __attribute__((aligned(16))) float lds[2 * XSTG];   // 80 / 64 KiB; the ONLY LDS object
  // MODE.FP16_OVFL = 1 for the life of the wave: every f32 -> f16 conversion of the epilogue clamps to +-65504 instead of producing inf - the
  // same results on finite values as the explicit clamps it replaces (4 of the ~8 vector-ALU instructions per output value: v_med3 + its
  // canonicalising v_max, twice), as in the GCN tile engine (gcn_tile.hip)
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);

  constexpr int KS = 2, NM = 9 * NU, NR = 6 + 2 * NU, NBD = 2 * NU;
  // split-f16 products on v_mfma_f32_16x16x32_f16 (a K tile = ONE k-step; the K loop of gcn_tile.hip's split mode: four (row half, column half) phases in
  // snake order, no operand half double-buffered); the hi-only tier keeps the 32 x 32 x 16 form
  constexpr bool M16 = !HO;
  const int tid = threadIdx.x;
  const int K = p.KH * p.KW * p.Ci + (DS ? p.Ci2 : 0);
  const int cpt = p.Ci / XRK;                    // K tiles per tap
  const int KT1 = p.KH * p.KW * cpt;             // K tiles of the first segment
  const int KT = KT1 + (DS ? p.Ci2 / XRK : 0);
  const int n_tiles = (p.Co + XBN - 1) / XBN;
  const int m_tiles = (int)((p.M + XBM - 1) / XBM);

  int lane, wave, wm, wn, mi, g, r0, swz;
  bool hi_lane;                                   // my 16-byte chunk of an operand piece holds hi halves (logical chunks 0-3 of the 128-byte K tile)
  int oA[KS][2], oB[KS][2];
  [[maybe_unused]] int oA16[2], oB16[2];
  auto thread_consts = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63;
    wave = __builtin_amdgcn_readfirstlane(t >> 6);
    wm = wave >> 1; wn = wave & 1;
    mi = lane & 31; g = lane >> 5;
    r0 = 8 * wave + (lane >> 3);
    swz = ((lane & 7) ^ ((r0 >> 1) & 7)) << 2;
    hi_lane = swz < 16;
    const int rA = 96 * wm + mi, rB = 32 * NU * wn + mi;
    const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const int c = 4 * hl + 2 * s + g;
        oA[s][hl] = rA * XRK + ((c ^ keyA) << 2);
        oB[s][hl] = XA_T + rB * XRK + ((c ^ keyB) << 2);
      }
    if constexpr (M16) {      // lane (i = l & 15, kg = l >> 4): row i of a 16-row tile, logical chunk kg (hi) / 4 + kg (lo) of the 128-byte K tile
      const int i16 = lane & 15, kg = lane >> 4, key = (i16 >> 1) & 7;
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        oA16[hl] = (96 * wm + i16) * XRK + (((4 * hl + kg) ^ key) << 2);
        oB16[hl] = XA_T + (32 * NU * wn + i16) * XRK + (((4 * hl + kg) ^ key) << 2);
      }
    }
  };
  thread_consts();

  const int G = gridDim.x, b = blockIdx.x;
  const bool xcd_order = (G % 8 == 0) && ((G / 8) % n_tiles == 0);
  auto tile_of = [&](int it, int& m, int& n) -> bool {
    if (xcd_order) {
      const int x = b & 7, j = b >> 3, per = (G >> 3) / n_tiles;
      m = (it * per + j / n_tiles) * 8 + x;
      n = j % n_tiles;
    } else {
      const long long t = (long long)it * G + b;
      m = (int)(t / n_tiles);
      n = (int)(t % n_tiles);
    }
    return m < m_tiles;
  };

  // my six activation rows of the tile: float offset of the (kh = 0, kw = 0) tap pixel's channel 0 (+ my swizzled chunk) and the taps
  // that lie inside the image
  // operand pieces in buffer form (as in gcn_tile.hip): the activation tensor / the tile's weight rows behind SGPR descriptors, one 32-bit BYTE
  // offset per lane and piece (the gathered pixel, or the zero row), the weights' piece in the scalar offset
  unsigned int pbase[6];        // byte offset (mod 2^32: border pixels start below zero) of the (kh = 0, kw = 0) tap pixel's channel 0 + my swizzled chunk
  unsigned int pmask[6];
  [[maybe_unused]] unsigned int pbase2[6];   // DS: byte offset of my rows' pixels in x2 (or its zero row)
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsX2 = ehm_buffer_rsrc_4g(DS ? p.x2 : p.x);
  const __amdgpu_buffer_rsrc_t rsX = ehm_buffer_rsrc_4g(p.x);      // lane offsets address the whole activation tensor (< 2^30 elements = 4 GiB, checked by ehm_conv_x2)
  __amdgpu_buffer_rsrc_t rsB;
  int voB;
  const int brow32 = 32 * K;
  auto set_tile = [&](int m, int n) {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const long long gm = (long long)m * XBM + r0 + 32 * i;
      pmask[i] = 0u;
      pbase[i] = 0u;
      if constexpr (DS) pbase2[i] = (p.zero_off2 + (unsigned int)swz) * 4u;
      if (gm < p.M) {
        const int nimg = (int)(gm / hw), rem = (int)(gm % hw);
        if constexpr (DS) pbase2[i] = (unsigned int)(((nimg * p.H2 + (rem / p.Wo) * p.stride2) * p.W2 + (rem % p.Wo) * p.stride2) * p.Ci2 + swz) * 4u;
        const int hi0 = (rem / p.Wo) * p.stride - p.pad, wi0 = (rem % p.Wo) * p.stride - p.pad;
        pbase[i] = (unsigned int)(((nimg * p.H + hi0) * p.Wd + wi0) * p.Ci + swz) * 4u;
        for (int kh = 0; kh < p.KH; ++kh)
          for (int kw = 0; kw < p.KW; ++kw)
            if (hi0 + kh >= 0 && hi0 + kh < p.H && wi0 + kw >= 0 && wi0 + kw < p.Wd) pmask[i] |= 1u << (kh * p.KW + kw);
      }
    }
    rsB = ehm_buffer_rsrc(p.W + (size_t)n * XBN * K);
    voB = (r0 * K + swz) * 4;
  };
  auto dma_a = [&](int buf, int kt, int i) {
    if (HO && !hi_lane) return;                   // (half of the lanes of every piece: the instruction is still issued, the wait counts stand)
    if constexpr (DS) {
      if (kt >= KT1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, (AS3 void*)(lds + buf * XSTG + (wave + 4 * i) * 256), 16, (int)pbase2[i], (kt - KT1) * XRK * 4, 0, 0);
        return;
      }
    }
    const int tap = kt / cpt, cg = kt - tap * cpt;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const unsigned int toff = (unsigned int)((kh * p.Wd + kw) * p.Ci + cg * XRK) * 4u;   // (wave-uniform, bytes)
    const unsigned int off = ((pmask[i] >> tap) & 1u) ? pbase[i] + toff : (p.zero_off + (unsigned int)swz) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (AS3 void*)(lds + buf * XSTG + (wave + 4 * i) * 256), 16, (int)off, 0, 0, 0);
  };
  auto dma_b = [&](int buf, int kt, int i) {
    if (HO && !hi_lane) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (AS3 void*)(lds + buf * XSTG + XA_T + (wave + 4 * i) * 256), 16, voB, (i * brow32 + kt * XRK) * 4, 0, 0);
  };
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_a(buf, kt, i);
#pragma unroll
    for (int i = 0; i < NBD; ++i) dma_b(buf, kt, i);
  };
  auto read_frags = [&](XFrags<NU>& f, int buf, int s) {
    const float* S = lds + buf * XSTG;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f.ah[t] = *(const half8*)(S + oA[s][0] + 32 * t * XRK);
      if constexpr (!HO) f.al[t] = *(const half8*)(S + oA[s][1] + 32 * t * XRK);
    }
 #pragma unroll
    for (int u = 0; u < NU; ++u) {
      f.bh[u] = *(const half8*)(S + oB[s][0] + 32 * u * XRK);
      if constexpr (!HO) f.bl[u] = *(const half8*)(S + oB[s][1] + 32 * u * XRK);
    }
  };
  f32x16 acc[3][NU];
  auto mfmas = [&](const XFrags<NU>& f) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u) {                             // small cross terms first, leading term last
        if constexpr (!HO) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[u], acc[t][u], 0, 0, 0);
        }
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[u], acc[t][u], 0, 0, 0);
      }
  };
  auto pin_reads = [&]() {
    if constexpr (HO) return;                     // (the hi-only instruction mix is left to the scheduler)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  };
  auto pin_reads_dma = [&]() {
    if constexpr (HO) return;
    if constexpr (NU == 2) {                                      // 18 MFMAs, 10 reads, 10 DMAs
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    } else {                                                      // 9 MFMAs, 8 reads, 8 DMAs
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };

  if (blockIdx.x == 0)                                           // the output's own all-zero row (the next conv's out-of-image taps)
    for (int c = tid; c < p.Co; c += 256) ((float*)p.y)[(size_t)m_tiles * XBM * p.Co + c] = 0.f;
  // ---- work pieces: a whole tile (k0 = 0, k1 = KT), or with stream-K a contiguous run of K tiles of one tile
  struct Piece { int m, n, k0, k1, tile; };
  const int P2 = KT / 2;                                         // pairs of K tiles per tile (stream-K deals in pairs: every piece has >= 2 K tiles)
  long long sk_cur = 0, sk_end = 0;
  [[maybe_unused]] int sk_t0 = 0;                                // first tile (linear index m * n_tiles + n) of the K-run part
  if constexpr (SK) {
    sk_t0 = p.sk_rounds * G;                                     // (round `it` of tile_of covers the linear tiles [it G, (it + 1) G) in either block order)
    const long long U2 = ((long long)m_tiles * n_tiles - sk_t0) * P2;
    sk_cur = (long long)b * p.sk_per;
    sk_end = sk_cur + p.sk_per < U2 ? sk_cur + p.sk_per : U2;
  }
  auto piece_of = [&](int it, Piece& o) -> bool {
    if constexpr (SK) {
      if (it < p.sk_rounds) {                                    // a whole tile of the plain schedule (tile = -1: no hand-off)
        o.k0 = 0; o.k1 = KT; o.tile = -1;
        return tile_of(it, o.m, o.n);
      }
      if (sk_cur >= sk_end) return false;
      o.tile = (int)(sk_cur / P2);                               // (index within the K-run part: flags and partial sums are kept per such tile)
      const int kp0 = (int)(sk_cur % P2);
      const long long left = sk_end - sk_cur;
      const int kp1 = left < P2 - kp0 ? kp0 + (int)left : P2;
      o.m = (sk_t0 + o.tile) / n_tiles; o.n = (sk_t0 + o.tile) % n_tiles;
      o.k0 = 2 * kp0; o.k1 = 2 * kp1;
      return true;
    } else {
      o.k0 = 0; o.k1 = KT; o.tile = 0;
      return tile_of(it, o.m, o.n);
    }
  };
  auto advance = [&](const Piece& o) { if constexpr (SK) { if (o.tile >= 0) sk_cur += (o.k1 - o.k0) / 2; } };
  int it = 0;
  Piece pc;
  if (!piece_of(0, pc)) return;
  advance(pc);
  set_tile(pc.m, pc.n);
#pragma unroll
  for (int i = 0; i < NBD; ++i) dma_b(0, pc.k0, i);
#pragma unroll
  for (int i = 0; i < NBD; ++i) dma_b(1, pc.k0 + 1, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_a(0, pc.k0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_a(1, pc.k0 + 1, i);

  while (true) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // everything but stage 1's six activation pieces (the last instructions issued)
    __syncthreads();
    thread_consts();
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    XFrags<NU> f0, f1;
    if constexpr (!M16) read_frags(f0, 0, 0);
    const int KL = pc.k1 - pc.k0;                                // K tiles of this piece (>= 2)
    // ---- 16 x 16 x 32: operand halves A[rh] (row tiles 3 rh .. + 2 of 16 rows), B[ch] (column tiles NU ch .. + NU - 1 of 16), 6 x 2 NU accumulators
    [[maybe_unused]] half8 Ah[2][3], Al[2][3], Bh[2][NU], Bl[2][NU];
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    [[maybe_unused]] f32x4a c16[6][2 * NU];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    [[maybe_unused]] auto ldA = [&](auto rhc, int buf) __attribute__((always_inline)) {
      constexpr int rh = decltype(rhc)::value;
      const float* S = lds + buf * XSTG;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        Ah[rh][t] = *(const half8*)(S + oA16[0] + 16 * (3 * rh + t) * XRK);
        Al[rh][t] = *(const half8*)(S + oA16[1] + 16 * (3 * rh + t) * XRK);
      }
    };
    [[maybe_unused]] auto ldB = [&](auto chc, int buf) __attribute__((always_inline)) {
      constexpr int ch = decltype(chc)::value;
      const float* S = lds + buf * XSTG;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        Bh[ch][u] = *(const half8*)(S + oB16[0] + 16 * (NU * ch + u) * XRK);
        Bl[ch][u] = *(const half8*)(S + oB16[1] + 16 * (NU * ch + u) * XRK);
      }
    };
    [[maybe_unused]] auto mm = [&](auto rhc, auto chc) __attribute__((always_inline)) {      // 9 NU MFMAs: small cross terms first, 3 NU independent accumulators per term
      constexpr int rh = decltype(rhc)::value, ch = decltype(chc)::value;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) c16[3 * rh + t][NU * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[rh][t], Bh[ch][u], c16[3 * rh + t][NU * ch + u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) c16[3 * rh + t][NU * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[rh][t], Bl[ch][u], c16[3 * rh + t][NU * ch + u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) c16[3 * rh + t][NU * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[rh][t], Bh[ch][u], c16[3 * rh + t][NU * ch + u], 0, 0, 0);
    };
    [[maybe_unused]] auto pin16 = [&](int reads, int dmas) __attribute__((always_inline)) {   // reads one per MFMA from the start, DMAs behind them
#pragma unroll
      for (int i = 0; i < 9 * NU; ++i) {                          // reads behind every second MFMA, DMAs in the gaps (as in gcn_tile.hip)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((i & 1) == 0 && (i >> 1) < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if ((i & 1) == 1 && (i >> 1) < dmas) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
    // K tile j of the piece (parity PAR = j & 1 = its stage).  Phases (0, cf) (0, cs) | barrier | (1, cs) (1, cf) with cf = PAR: the next tile's first phase
    // is (0, cs) - the halves that are free to be refilled during this tile's last two phases.  Returns false behind the barrier of the piece's LAST K tile
    // (its last two phases run below); K tile j + 2 is staged while there is one.
    [[maybe_unused]] auto tile16 = [&](auto parc, int j) __attribute__((always_inline)) -> bool {
      constexpr int PAR = decltype(parc)::value;
      constexpr int buf = PAR;
      typedef std::integral_constant<int, PAR> CF;
      typedef std::integral_constant<int, 1 - PAR> CS;
      ldB(CS{}, buf);
      mm(I0{}, CF{});
      pin16(2 * NU, 0);
      __builtin_amdgcn_sched_barrier(0);
      ldA(I1{}, buf);
      mm(I0{}, CS{});
      pin16(6, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (j == KL - 1) return false;
      const bool more = j + 2 < KL;                            // (block-uniform)
      ldA(I0{}, buf ^ 1);
      __builtin_amdgcn_s_setprio(2);
      if (more) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_a(buf, pc.k0 + j + 2, i);
      }
      mm(I1{}, CS{});
      pin16(6, NU == 2 ? 6 : 3);
      __builtin_amdgcn_sched_barrier(0);
      ldB(CS{}, buf ^ 1);
      if (more) {
#pragma unroll
        for (int i = 0; i < NBD; ++i) dma_b(buf, pc.k0 + j + 2, i);
      }
      mm(I1{}, CF{});
      pin16(2 * NU, NU == 2 ? NBD : NBD + 3);
      __builtin_amdgcn_s_setprio(0);
      return true;
    };
    if constexpr (M16) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int u = 0; u < 2 * NU; ++u) c16[t][u] = f32x4a{0.f, 0.f, 0.f, 0.f};
      ldA(I0{}, 0);
      ldB(I0{}, 0);
      for (int j = 0;; j += 2) {
        if (!tile16(I0{}, j)) break;
        if (!tile16(I1{}, j + 1)) break;
      }                                                           // (the last K tile's second half - phases (1, 0) and (1, 1) - runs below)
    } else {
    auto first_phase = [&](int buf) {
      read_frags(f1, buf, 1);
      mfmas(f0);
      pin_reads();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    for (int j = 0; j < KL - 2; ++j) {
      const int buf = j & 1;
      first_phase(buf);
      read_frags(f0, buf ^ 1, 0);
      stage(buf, pc.k0 + j + 2);
      mfmas(f1);
      pin_reads_dma();
    }
    {
      const int buf = (KL - 2) & 1;
      first_phase(buf);
      read_frags(f0, buf ^ 1, 0);
      mfmas(f1);
      pin_reads();
    }
    first_phase((KL - 1) & 1);
    }

    const int n0 = pc.n * XBN;
    const size_t row0 = (size_t)pc.m * XBM;
    const bool cols_live = n0 + 32 * NU * wn < p.Co;             // (NU = 2 with Co = 64: the upper column half of the tile is padding)
    constexpr int NADD = M16 ? 2 * NU : NU;                      // M16: my columns are 32 NU wn + 16 ct + (lane & 15), ct < 2 NU
    float add[NADD];
#pragma unroll
    for (int u = 0; u < NADD; ++u) {
      const int col = M16 ? n0 + 32 * NU * wn + 16 * u + (lane & 15) : n0 + 32 * NU * wn + 32 * u + mi;
      add[u] = (p.bias && col < p.Co) ? p.bias[col] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M16) {
      mm(I1{}, I0{});
      mm(I1{}, I1{});
    } else {
      mfmas(f1);
    }
    Piece nx;
    const bool have_next = piece_of(it + 1, nx);
    if (have_next) {
      advance(nx);
      set_tile(nx.m, nx.n);
#pragma unroll
      for (int i = 0; i < NBD; ++i) dma_b(0, nx.k0, i);
#pragma unroll
      for (int i = 0; i < NBD; ++i) dma_b(1, nx.k0 + 1, i);
#pragma unroll
      for (int i = 0; i < 6; ++i) dma_a(0, nx.k0, i);
    }

    // ---- stream-K hand-off.  A piece that does not start its tile leaves its accumulators for the block that does; the block that starts a
    //      tile (always the LAST piece of its run) collects the later pieces - they were the FIRST pieces of their blocks' runs - in k order.
    bool finish = true;
    if (SK && pc.tile >= 0) {
      constexpr int NACC = 3 * NU * 16;
      const int owner = (int)(((long long)pc.tile * P2) / p.sk_per), last_blk = (int)(((long long)pc.tile * P2 + P2 - 1) / p.sk_per);
      float* part = p.sk_part + (size_t)pc.tile * kSkMaxParts * NACC * 256;
      if (pc.k0 != 0) {
        finish = false;
        float* mine = part + (size_t)(b - owner - 1) * NACC * 256 + tid;
        if constexpr (M16) {
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int u = 0; u < 2 * NU; ++u)
#pragma unroll
              for (int r = 0; r < 4; ++r) mine[((t * 2 * NU + u) * 4 + r) * 256] = c16[t][u][r];
        } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[((t * NU + u) * 16 + r) * 256] = acc[t][u][r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(p.sk_flags + pc.tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (last_blk > owner) {
        const unsigned int expect = (unsigned int)(last_blk - owner);
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.sk_flags + pc.tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && ++spins < (1 << 24))
            __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        // never hang the device - but never hand out a sum with a missing K range either: a tile whose partners did not arrive in ~1 s (a
        // preempted / shared GPU) comes out as NaN, which the callers' finite checks and every downstream consumer make loud.  Every thread
        // looks at the (monotonic) counter itself and poisons through the bias term (every output is acc * inv_scale + add): __syncthreads_or
        // brings its own LDS word, and with 80 KiB + 4 bytes the kernel lost its second block per CU (hipcc then spent 183 VGPRs + 96 AGPRs);
        // writing NaN into the 96 accumulators has the same effect on the register budget.
        const unsigned int seen = __hip_atomic_load(p.sk_flags + pc.tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool missing = seen < expect || (seen & 0x80000000u) != 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // the counter goes back to zero for the next conv on this workspace (no memset node per launch: ehm_conv_x2_desc.workspace_clean); after a time-out it
        // stays poisoned - a partner may still arrive late, and a later launch must not take its count for its own
        __syncthreads();
        if (tid == 0) {
          __hip_atomic_store(p.sk_flags + pc.tile, missing ? 0x80000000u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (missing) __hip_atomic_fetch_add(p.sk_flags + kSkPoisonWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky: the host's status call reports it and zeroes the counters
        }
        if (missing) {
#pragma unroll
          for (int u = 0; u < NADD; ++u) add[u] = __builtin_nanf("");
        }
        for (int q = 0; q < last_blk - owner; ++q) {
          const float* theirs = part + (size_t)q * NACC * 256 + tid;
          if constexpr (M16) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
              for (int u = 0; u < 2 * NU; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) c16[t][u][r] += theirs[((t * 2 * NU + u) * 4 + r) * 256];
          } else {
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[t][u][r] += theirs[((t * NU + u) * 16 + r) * 256];
          }
        }
      }
    }

    // Epilogue per wave through its six 1 KiB pieces of stage 1's activation region.  Accumulator register r of acc[t][u] is row
    // 32 t + 8 (r >> 2) + 4 g + (r & 3): row groups Gq = 4 t + (r >> 2) of 8 rows.  NU = 2: a pass turns 3 groups x 64 columns (piece
    // 2 gi + u), 4 passes; NU = 1: 6 groups x 32 columns (piece gi), 2 passes.  Read items = (row, 8 columns), three per lane.
    if constexpr (M16) {
     if (cols_live && finish) {
      // accumulator layout: lane (i = l & 15, rg = l >> 4) owns columns 32 NU wn + 16 ct + i; register r of c16[rt][ct] is row 16 rt + 4 rg + r.  One pass per
      // row tile (16 rows x 32 NU columns through 2 NU of the wave's six 1 KiB pieces): piece NU (row >> 3) + (col >> 5) holds [8 rows][32 columns]; inside a piece
      // row (row & 7) sits at (row & 7) ^ (row >> 3) and its two 16-column halves swap places for odd rg (four lane groups of a write -> four 16-bank groups).
      // Read items = (row, 8 columns): NU per lane.
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const unsigned int yrow = (unsigned int)p.Co * 4u;
      const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc(p.y + (row0 + 96 * wm) * (size_t)yrow);
      const __amdgpu_buffer_rsrc_t rB = ehm_buffer_rsrc((p.res ? p.res : p.y) + (row0 + 96 * wm) * (size_t)yrow);
      const bool relu = p.relu != 0, has_res = p.res != nullptr;
      const int i16 = lane & 15, rg = lane >> 4;
      int wb[2][2];                                                // write address of (r, ct) = wb[r & 1][ct & 1] + (ct >> 1) 1024 + 32 r
      {
        const int base = XSTG + wave * 256 + (rg >> 1) * (NU * 1024) + (rg & 1) * 128 + i16, hoff = 32 * (rg >> 1), ooff = 16 * (rg & 1);
        wb[0][0] = base + hoff + ooff;       wb[0][1] = base + hoff + 16 - ooff;
        wb[1][0] = base - hoff + ooff;       wb[1][1] = base - hoff + 16 - ooff;
      }
      // items: NU = 2: item k = row (l >> 3) + 8 k, columns 8 (l & 7) .. + 7 of 64;  NU = 1: one item, row l >> 2, columns 8 (l & 3) .. + 7 of 32
      const int irow0 = NU == 2 ? lane >> 3 : lane >> 2, oct = NU == 2 ? lane & 7 : lane & 3;
      const int colw = n0 + 32 * NU * wn + 8 * oct;
      const unsigned int col_off = (unsigned int)(((colw >> 5) * 64 + (colw & 31)) * 2);
      int rbase[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int rho = NU == 2 ? irow0 + 8 * k : irow0;          // row of the pass (0..15): piece row (rho & 7) ^ (rho >> 3), column halves swapped for odd rho >> 2
        rbase[k] = XSTG + wave * 256 + (NU * (rho >> 3) + (oct >> 2)) * 1024 + ((rho & 7) ^ (rho >> 3)) * 32 + ((8 * (oct & 3)) ^ (16 * ((rho >> 2) & 1)));
      }
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
        u32x4_t rq[NU][2];
        if (has_res) {
#pragma unroll
          for (int k = 0; k < NU; ++k) {
            const unsigned int vo = (unsigned int)(16 * rt + (NU == 2 ? irow0 + 8 * k : irow0)) * yrow + col_off;
            rq[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo, 0, 0);
            rq[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo + 64u, 0, 0);
          }
        }
#pragma unroll
        for (int ct = 0; ct < 2 * NU; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[wb[r & 1][ct & 1] + (ct >> 1) * 1024 + 32 * r] = fmaf(c16[rt][ct][r], p.inv_scale, add[ct]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 tq[NU][2];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
          tq[k][0] = *(const f32x4*)(lds + rbase[k]);
          tq[k][1] = *(const f32x4*)(lds + rbase[k] + 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NU; ++k) {
          float v[8] = {tq[k][0][0], tq[k][0][1], tq[k][0][2], tq[k][0][3], tq[k][1][0], tq[k][1][1], tq[k][1][2], tq[k][1][3]};
          if (has_res) {
            const half8 rh = __builtin_bit_cast(half8, rq[k][0]), rl = __builtin_bit_cast(half8, rq[k][1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += (float)rh[c] + (float)rl[c];
          }
          half8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            // (stream-K: a poisoned tile must stay NaN through the ReLU - v_max_f32 returns its non-NaN operand)
            if (relu) v[c] = SK ? (v[c] < 0.f ? 0.f : v[c]) : fmaxf(v[c], 0.f);
            hh[c] = (half_t)v[c];                              // (MODE.FP16_OVFL: the conversions saturate at +-65504, see the kernel's head)
            ll[c] = (half_t)(v[c] - (float)hh[c]);
          }
          const unsigned int vo = (unsigned int)(16 * rt + (NU == 2 ? irow0 + 8 * k : irow0)) * yrow + col_off;
          if (colw < p.Co) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, 0);
          }
        }
      }
     }
    } else
    if (cols_live && finish) {
      constexpr int GP = NU == 2 ? 3 : 6, NPASS = 12 / GP;       // row groups per pass
      const unsigned int yrow = (unsigned int)p.Co * 4u;
      const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc(p.y + (row0 + 96 * wm) * (size_t)yrow);
      const __amdgpu_buffer_rsrc_t rB = ehm_buffer_rsrc((p.res ? p.res : p.y) + (row0 + 96 * wm) * (size_t)yrow);
      const bool relu = p.relu != 0, has_res = p.res != nullptr;
      const int wbase = XSTG + wave * 256 + (4 * g) * 32 + mi;
      int rbase, colw, irow;                                     // LDS offset of my item 0, its first column, its row inside the pass
      if constexpr (NU == 2) {
        const int rr = lane >> 3, oct = lane & 7;               // item it = group it: row rr, columns 8 oct .. + 7 of 64
        rbase = XSTG + wave * 256 + (oct >> 2) * 1024 + rr * 32 + 8 * (oct & 3);
        colw = n0 + 64 * wn + 8 * oct;
        irow = rr;
      } else {
        const int lr = lane >> 2;                                // item it: row 16 it + lr of the pass's 48, columns 8 (lane & 3) .. + 7 of 32
        rbase = XSTG + wave * 256 + (lr >> 3) * 1024 + (lr & 7) * 32 + 8 * (lane & 3);
        colw = n0 + 32 * wn + 8 * (lane & 3);
        irow = lr;
      }
      constexpr int ISTEP = NU == 2 ? 8 : 16;                    // rows between my consecutive items
      const unsigned int col_off = (unsigned int)(((colw >> 5) * 64 + (colw & 31)) * 2);
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        u32x4_t rq[3][2];
        if (has_res) {
#pragma unroll
          for (int it3 = 0; it3 < 3; ++it3) {
            const unsigned int vo = (unsigned int)(8 * GP * ps + ISTEP * it3 + irow) * yrow + col_off;
            rq[it3][0] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo, 0, 0);
            if constexpr (!HO) rq[it3][1] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo + 64u, 0, 0);
          }
        }
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
          const int Gq = GP * ps + gi, t = Gq >> 2, q = Gq & 3;
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[wbase + (NU * gi + u) * 1024 + j * 32] = fmaf(acc[t][u][4 * q + j], p.inv_scale, add[u]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 tq[3][2];
#pragma unroll
        for (int it3 = 0; it3 < 3; ++it3) {
          tq[it3][0] = *(const f32x4*)(lds + rbase + 2048 * it3);
          tq[it3][1] = *(const f32x4*)(lds + rbase + 2048 * it3 + 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it3 = 0; it3 < 3; ++it3) {
          float v[8] = {tq[it3][0][0], tq[it3][0][1], tq[it3][0][2], tq[it3][0][3], tq[it3][1][0], tq[it3][1][1], tq[it3][1][2], tq[it3][1][3]};
          if (has_res) {
            const half8 rh = __builtin_bit_cast(half8, rq[it3][0]);
            if constexpr (HO) {
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] += (float)rh[c];
            } else {
              const half8 rl = __builtin_bit_cast(half8, rq[it3][1]);
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] += (float)rh[c] + (float)rl[c];
            }
          }
          half8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            // (stream-K: a poisoned tile must stay NaN through the ReLU - v_max_f32 returns its non-NaN operand)
            if (relu) v[c] = SK ? (v[c] < 0.f ? 0.f : v[c]) : fmaxf(v[c], 0.f);
            hh[c] = (half_t)v[c];                              // (MODE.FP16_OVFL: the conversions saturate at +-65504, see the kernel's head)
            ll[c] = (half_t)(v[c] - (float)hh[c]);
          }
          const unsigned int vo = (unsigned int)(8 * GP * ps + ISTEP * it3 + irow) * yrow + col_off;
          if (colw < p.Co) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, 0);
            if constexpr (!HO) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, 0);
          }
        }
      }
    }
    if (!have_next) break;
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_a(1, nx.k0 + 1, i);
    pc = nx; ++it;
  }
}

// X2 [rows, C] -> float32 mean over groups of `hw` consecutive rows: the global average pool behind the last bottleneck
__global__ void x2_group_mean_kernel(const half_t* __restrict__ X, float* __restrict__ Y, int hw, int C, int hi_only) {
  const int img = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;        // thread = channel: a wave reads 64 consecutive halves per row
  if (c >= C) return;
  float s = 0.f;
  if (hi_only) {
    for (int r = 0; r < hw; ++r) s += (float)X[split_off<32>((size_t)img * hw + r, c, C)];
  } else
  for (int r = 0; r < hw; ++r) s += split_load<32>(X, (size_t)img * hw + r, c, C);
  Y[(size_t)img * C + c] = s / (float)hw;
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

extern "C" int64_t ehm_conv_x2_rows(int64_t pixels) { return round_up(pixels, XBM) + 1; }

namespace {
// stream-K plans of a conv (whole tiles only: the last round of the grid is as long as a full one however few tiles it holds - 524 tiles on 512
// slots are two rounds for 1.02 rounds of work, and at N = 256 images most of ResNet-50's layers 2 - 4 land just above a multiple of the slots):
//  * rounds = 0, "every tile": the K tiles of ALL tiles are one sequence dealt out in equal runs - for K loops of >= 64 K tiles whose tile count is
//    below one round (layer 4) or just above it (the 3 x 3 convs of layer 3); with shorter K loops nearly every tile would be cut and the 96 KiB
//    partial-sum hand-off of a cut tile costs more than the idle slots (measured: 1 x 1 convs 0.14 -> 0.175 ms);
//  * rounds = R >= 1, "tail only": R whole rounds run the plain schedule, only the L < slots tiles behind them are cut, into <= kSkMaxParts runs each,
//    so that the last round lasts a third of a K loop + one hand-off instead of a whole K loop.
struct SkPlan { bool on; int per, rounds; int64_t tiles, cut_tiles, blocks, flag_bytes, bytes; int nacc; };
SkPlan sk_plan(const ehm_conv_x2_desc* d, int Ho, int Wo) {
  SkPlan s{};
  const int64_t M = (int64_t)d->N * Ho * Wo, slots = 2 * (int64_t)ehm_num_cus();
  const bool narrow = d->Co % 128 != 0;
  const int KT = d->KH * d->KW * (d->Ci / XRK) + (d->x2 != nullptr ? d->Ci2 / XRK : 0);
  s.tiles = ceil_div(M, XBM) * ceil_div(d->Co, narrow ? 64 : 128);
  s.nacc = 3 * (narrow ? 1 : 2) * 16;
  if (KT % 2 != 0 || KT < 4) return s;
  const int P2 = KT / 2;
  const int64_t R = s.tiles / slots, L = s.tiles - R * slots;
  const int64_t rounds = ceil_div(s.tiles, slots);
  const double eff = (double)s.tiles / (double)(rounds * slots);
  if (KT >= 64 && d->x2 == nullptr && eff < 0.875 && s.tiles >= slots / 2 && R <= 1) {          // every tile
    s.per = (int)ceil_div(s.tiles * P2, slots);
    if (s.per < 2 || ceil_div(P2, s.per) > kSkMaxParts) return s;
    s.rounds = 0; s.cut_tiles = s.tiles;
    s.blocks = ceil_div(s.tiles * P2, (int64_t)s.per);
  } else if (R >= 1 && L > 0) {                                                                    // tail only
    const int64_t parts = std::min<int64_t>(std::min<int64_t>(kSkMaxParts, slots / L), P2);
    if (parts < 2) return s;
    s.per = (int)ceil_div(P2, parts);
    // worth it when the cut tail - its K run + the hand-off - is shorter than a whole K loop.  Measured per conv at N = 256 (tools/enc_layers.py, same box,
    // against whole tiles): K loops of 32 / 36 K tiles gain 13 - 19 us (5 x layer 3 c1 0.121 -> 0.102 ms, layer 2's 3 x 3 convs 0.210 -> 0.196), K loops of 16 / 18
    // K tiles LOSE 10 - 28 us: the hand-off (partial sums out, agent-scope release / acquire, partial sums in) costs about 14 K tiles' time
    if (2 * s.per + kSkHandoffKTiles > KT - 2 || ceil_div(P2, s.per) > kSkMaxParts) return s;
    s.rounds = (int)R; s.cut_tiles = L;
    s.blocks = slots;
  } else {
    return s;
  }
  s.on = true;
  if (s.cut_tiles > kSkPoisonWord) { s.on = false; return s; }
  s.flag_bytes = kSkFlagBytes;                 // (a fixed region at the head of the workspace: convs of different shapes share one workspace and one zeroing)
  s.bytes = s.flag_bytes + s.cut_tiles * kSkMaxParts * s.nacc * 256 * 4;
  return s;
}
}  // namespace

extern "C" int64_t ehm_conv_x2_workspace_bytes(const ehm_conv_x2_desc* d) {
  if (!d || d->Ci <= 0 || d->Co <= 0 || d->KH <= 0 || d->KW <= 0 || (d->stride != 1 && d->stride != 2)) return 0;
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Wd + 2 * d->pad - d->KW) / d->stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  return sk_plan(d, Ho, Wo).bytes;
}

extern "C" int ehm_conv_x2(const ehm_conv_x2_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->x && d->W && d->y);
  EHM_CHECK_ARG(d->N > 0 && d->H > 0 && d->Wd > 0 && d->Ci > 0 && d->Co > 0 && d->Ci % XRK == 0 && d->Co % 32 == 0);
  EHM_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 32 && (d->stride == 1 || d->stride == 2) && d->pad >= 0);
  EHM_CHECK_ARG(d->w_scale > 0.f);
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Wd + 2 * d->pad - d->KW) / d->stride + 1;
  EHM_CHECK_ARG(Ho > 0 && Wo > 0 && d->KH * d->KW * (d->Ci / XRK) >= 2);
  const bool dual = d->x2 != nullptr;
  if (dual) {
    EHM_CHECK_ARG(d->Ci2 > 0 && d->Ci2 % XRK == 0 && (d->stride2 == 1 || d->stride2 == 2) && d->H2 > 0 && d->W2 > 0 && d->Co % 128 == 0);
    EHM_CHECK_ARG((d->H2 - 1) / d->stride2 + 1 == Ho && (d->W2 - 1) / d->stride2 + 1 == Wo);
    if (d->x2_rows < ehm_conv_x2_rows((int64_t)d->N * d->H2 * d->W2) || d->x2_rows * d->Ci2 >= ((int64_t)1 << 30)) {
      ehm_set_error("ehm_conv_x2: x2_rows = %lld, need ehm_conv_x2_rows(N*H2*W2) = %lld rows and x2 below 2^30 elements", (long long)d->x2_rows,
                    (long long)ehm_conv_x2_rows((int64_t)d->N * d->H2 * d->W2));
      return EHM_EINVAL;
    }
  }
  const int64_t in_rows = (int64_t)d->N * d->H * d->Wd;
  if (d->x_rows < ehm_conv_x2_rows(in_rows) || (d->x_rows * d->Ci) >= ((int64_t)1 << 30) ||       // (32-bit BYTE offsets into x: the operand pieces use buffer addressing)
      ehm_conv_x2_rows((int64_t)d->N * Ho * Wo) * d->Co >= ((int64_t)1 << 31)) {
    ehm_set_error("ehm_conv_x2: x_rows = %lld, need ehm_conv_x2_rows(N*H*W) = %lld rows (tile padding + the zero row), x below 2^30 and y below 2^31 elements",
                  (long long)d->x_rows, (long long)ehm_conv_x2_rows(in_rows));
    return EHM_EINVAL;
  }
  ConvX2Args a;
  a.x = (const float*)d->x; a.W = (const float*)d->W; a.bias = d->bias; a.res = (const char*)d->residual; a.y = (char*)d->y;
  a.N = d->N; a.H = d->H; a.Wd = d->Wd; a.Ci = d->Ci; a.Ho = Ho; a.Wo = Wo; a.Co = d->Co;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.relu = d->relu;
  a.inv_scale = 1.f / d->w_scale;
  a.M = (long long)d->N * Ho * Wo;
  a.zero_off = (unsigned int)((d->x_rows - 1) * d->Ci);
  a.x2 = (const float*)d->x2; a.H2 = d->H2; a.W2 = d->W2; a.Ci2 = dual ? d->Ci2 : 0; a.stride2 = d->stride2;
  a.zero_off2 = dual ? (unsigned int)((d->x2_rows - 1) * d->Ci2) : 0u;
  // 64-wide column tiles only when Co is not a multiple of 128 (Co = 64: no padding columns through the matrix cores; 0.42 -> 0.29 ms
  // on the 3x3 convs of layer 1).  For layer 4's 264-tile convs they LOSE although they would give every CU two blocks (0.26 ->
  // 0.31 ms): a 96 x 32 wave tile reads 8 fragments per 9 MFMAs.
  const int64_t slots = 2 * (int64_t)ehm_num_cus();
  const bool narrow = d->Co % 128 != 0;
  const int64_t tiles = ceil_div(a.M, XBM) * ceil_div(d->Co, narrow ? 64 : 128);
  const SkPlan sk = sk_plan(d, Ho, Wo);
  a.sk_flags = nullptr; a.sk_part = nullptr; a.sk_per = 0; a.sk_rounds = 0;
  if (sk.on && d->workspace && d->workspace_bytes >= sk.bytes) {
    // stream-K (every tile, or the tail round only): a tile cut by a run boundary is finished by the block that started it
    a.sk_flags = (unsigned int*)d->workspace;
    a.sk_part = (float*)((char*)d->workspace + sk.flag_bytes);
    a.sk_per = sk.per;
    a.sk_rounds = sk.rounds;
    if (!d->workspace_clean) EHM_HIP(hipMemsetAsync(a.sk_flags, 0, (size_t)sk.flag_bytes, (hipStream_t)stream));
    const dim3 grid((unsigned)sk.blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (d->hi_only) {
      if (dual) hipLaunchKernelGGL((conv_x2_tile_kernel<2, true, true, true>), grid, blk, 0, st, a);
      else if (narrow) hipLaunchKernelGGL((conv_x2_tile_kernel<1, true, false, true>), grid, blk, 0, st, a);
      else hipLaunchKernelGGL((conv_x2_tile_kernel<2, true, false, true>), grid, blk, 0, st, a);
    } else if (dual) hipLaunchKernelGGL((conv_x2_tile_kernel<2, true, true>), grid, blk, 0, st, a);
    else if (narrow) hipLaunchKernelGGL((conv_x2_tile_kernel<1, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_x2_tile_kernel<2, true>), grid, blk, 0, st, a);
    EHM_LAUNCH_CHECK();
    return 0;
  }
  const int64_t blocks = tiles < slots ? tiles : slots;
  if (d->hi_only) {
    if (dual) hipLaunchKernelGGL((conv_x2_tile_kernel<2, false, true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if (narrow) hipLaunchKernelGGL((conv_x2_tile_kernel<1, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_x2_tile_kernel<2, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  } else if (dual) hipLaunchKernelGGL((conv_x2_tile_kernel<2, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (narrow) hipLaunchKernelGGL((conv_x2_tile_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv_x2_tile_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_conv_x2_workspace_status(void* workspace, uint32_t* host_flag, void* stream) {
  EHM_CHECK_ARG(workspace);
  const unsigned int* word = (const unsigned int*)workspace + kSkPoisonWord;
  if (host_flag) {                               // stream-ordered copy: the caller looks at *host_flag behind an event and calls again with NULL when it is non-zero
    EHM_HIP(hipMemcpyAsync(host_flag, word, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
  }
  unsigned int n = 0;
  EHM_HIP(hipMemcpyAsync(&n, word, sizeof(n), hipMemcpyDeviceToHost, (hipStream_t)stream));
  EHM_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (n == 0) return 0;
  EHM_HIP(hipMemsetAsync(workspace, 0, (size_t)kSkFlagBytes, (hipStream_t)stream));   // poisoned arrival counters + the count itself: the workspace is usable again
  ehm_set_error("ehm_conv_x2: %u stream-K tile hand-off(s) on this workspace timed out since the last status call (GPU shared / preempted / under a profiler?): "
                "those tiles - and every stream-K conv behind them - came out as NaN; the counters are zeroed again, re-run the batch", n);
  return EHM_EIO;
}

extern "C" int ehm_x2_group_mean(const void* X, float* Y, int groups, int rows_per_group, int C, int hi_only, void* stream) {
  EHM_CHECK_ARG(X && Y && groups > 0 && rows_per_group > 0 && C > 0 && C % 32 == 0);
  hipLaunchKernelGGL(x2_group_mean_kernel, dim3((unsigned)groups, (unsigned)ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)X, Y, rows_per_group, C,
                     hi_only);
  EHM_LAUNCH_CHECK();
  return 0;
}
