// Modulated-GCN hidden conv, split-f16 ('f16x3') arithmetic, WIDE wave tile: the same conv, the same bits as gcn_tile.hip's P = 3 engine
// (modulated_gcn_conv.py:39-50 + the residual of modulated_gcn.py:38-42), with half the operand traffic per matrix instruction.
//
// gcn_tile.hip's wave owns 96 rows x 32 channels (x 2 branches): 10 fragment reads and 10 one-KiB operand pieces per 36 MFMAs.  Its K loop is
// bound by what it takes to FEED the matrix pipe - issuing the pieces holds the SIMD's issue for the partner wave too (docs/EXPERIMENTS.md 3.2: the
// same schedule without an operand stream runs the pipe at 0.84 instead of 0.70).  Here a wave owns 96 rows x 64 channels (x 2 branches) =
// 192 accumulator registers, a block (4 waves, 2 x 2) 192 rows x 128 channels, and a K tile is ONE 16-wide k-step (64 bytes per row: 16 hi
// halves | 16 lo halves), so that two 28 KiB stages (A 12 KiB + B 16 KiB) still let two blocks share a CU:
//     per 36 MFMAs: 14 fragment reads (A 6, B 8) and 7 operand pieces (A 3, B 4)  -  was 20 and 10.
// Registers: 192 accumulators + the six A fragments of the k-step (24) + the four B fragments of ONE (channel tile, branch pair) in flight
// (16), refilled for the second channel tile while the first one's MFMAs issue.
//
// LDS image of a stage: rows of 64 bytes = 4 chunks of 16 bytes (logical chunk c = 2 * hl + g: hl = hi / lo halves, g = k 0-7 / 8-15);
// physical chunk = c ^ key(row), key = (row >> 3) & 3 for the activation rows (whose fragment rows follow the epilogue's row permutation),
// (row >> 2) & 3 for the weight rows: every 16-lane group of a ds_read_b128 then covers the 64 banks exactly once.  The DMA applies the
// swizzle on the SOURCE address (a piece = 16 rows x 64 bytes, lane -> (row, physical chunk)); a wave's pieces are 64 rows apart, which
// keeps both keys, so one lane offset per operand serves all its pieces.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

#include <type_traits>

namespace {

#define AS3 __attribute__((address_space(3)))

constexpr int WRK = 16;                 // floats per row and K tile (64 bytes)
constexpr int WA_T = 192 * WRK;         // activation region of a stage: 3072 floats = 12 KiB
constexpr int WB_T = 256 * WRK;         // weight region: 128 channels x 2 branches: 4096 floats = 16 KiB
constexpr int WSTG = WA_T + WB_T;       // 28 KiB
constexpr int kWLoadAux = 16, kWStoreAux = 16;   // sc1, as in gcn_tile.hip

struct WideArgs {
  LayerDev L;
  const void* X;
  const void* Res;
  void* Y;
  int m_tiles, out_f32;
};

typedef float f32x2w __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void gcn_hidden_wide_kernel(WideArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * WSTG];   // 56 KiB: two blocks per CU
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);                   // MODE.FP16_OVFL: saturating f32 -> f16 conversions (gcn_tile.hip)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, mi = lane & 31, g = lane >> 5;
  const int K = a.L.K, N = a.L.N;
  const int rowf = K;                   // floats per X2 row
  const int KT = K / 16;
  const int n_tiles = N / 128, tot = a.m_tiles * n_tiles, bid = blockIdx.x;
  const int lin = ((tot & 7) == 0) ? (bid & 7) * (tot >> 3) + (bid >> 3) : bid;   // XCD-aware tile order
  const int m_tile = lin / n_tiles, n_tile = lin % n_tiles;

  // ---- operand DMA: buffer form; lane -> (row of the piece, physical chunk)
  const __amdgpu_buffer_rsrc_t rsA = ehm_buffer_rsrc((const float*)a.X + (size_t)m_tile * 192 * rowf);
  const __amdgpu_buffer_rsrc_t rsB = ehm_buffer_rsrc((const float*)a.L.Ws + (size_t)n_tile * 256 * rowf);
  const int rl = lane >> 2, ph = lane & 3;
  const int cA = ph ^ ((2 * wave + (rl >> 3)) & 3), cB = ph ^ ((rl >> 2) & 3);
  const int voA = ((16 * wave + rl) * rowf) * 4 + (cA >> 1) * 64 + (cA & 1) * 16;
  const int voB = ((16 * wave + rl) * rowf) * 4 + (cB >> 1) * 64 + (cB & 1) * 16;
  const int row64 = 64 * rowf * 4;       // bytes between a wave's consecutive pieces
  auto stage = [&](int buf, int kt) {
    const int ko = (kt >> 1) * 128 + (kt & 1) * 32;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (AS3 void*)(lds + buf * WSTG + (wave + 4 * q) * 256), 16, voA, q * row64 + ko, 0, kWLoadAux);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (AS3 void*)(lds + buf * WSTG + WA_T + (wave + 4 * q) * 256), 16, voB, q * row64 + ko, 0, 0);
  };

  // ---- fragment offsets (floats inside a stage).  Activation rows follow the epilogue's row permutation (gcn_tile.hip): MFMA row mi of row
  //      tile t = wave row 48 ((mi>>2)&1) + 24 (mi&1) + ((mi>>1)&1) + 2 (mi>>3) + 8 t
  const int rA = 96 * wm + 48 * ((mi >> 2) & 1) + 24 * (mi & 1) + ((mi >> 1) & 1) + 2 * (mi >> 3);
  int oA[3][2], oB[2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int row = rA + 8 * t, c = 2 * hl + g;
      oA[t][hl] = row * WRK + ((c ^ ((row >> 3) & 3)) << 2);
    }
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const int row = 128 * wn + mi, c = 2 * hl + g;                  // + 64 br + 32 cc: multiples of 16 rows, the key ((row >> 2) & 3) stays
    oB[hl] = WA_T + row * WRK + ((c ^ ((mi >> 2) & 3)) << 2);
  }

  f32x16 acc[3][2][2];                   // [row tile][channel tile][branch]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int br = 0; br < 2; ++br)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][cc][br][r] = 0.f;

  stage(0, 0);
  stage(1, 1);
  asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // stage 0 has landed (stage 1's seven pieces may still be in flight)
  __syncthreads();

  // one K tile (compile-time stage: every LDS offset is an immediate off six + two base registers)
  auto ktile = [&](auto bufc, int kt) {
    constexpr int buf = decltype(bufc)::value;
    const float* S = lds + buf * WSTG;
    half8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) { ah[t] = *(const half8*)(S + oA[t][0]); al[t] = *(const half8*)(S + oA[t][1]); }
#pragma unroll
    for (int br = 0; br < 2; ++br) { bh[br] = *(const half8*)(S + oB[0] + 64 * br * WRK); bl[br] = *(const half8*)(S + oB[1] + 64 * br * WRK); }
    // channel tile 0: branch 0, then its fragments are replaced by channel tile 1's while branch 1 issues
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc[t][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[0], acc[t][0][0], 0, 0, 0);
      acc[t][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[0], acc[t][0][0], 0, 0, 0);
      acc[t][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[0], acc[t][0][0], 0, 0, 0);
    }
    const half8 ch0 = *(const half8*)(S + oB[0] + 32 * WRK), cl0 = *(const half8*)(S + oB[1] + 32 * WRK);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc[t][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[1], acc[t][0][1], 0, 0, 0);
      acc[t][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[1], acc[t][0][1], 0, 0, 0);
      acc[t][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[1], acc[t][0][1], 0, 0, 0);
    }
    const half8 ch1 = *(const half8*)(S + oB[0] + (64 + 32) * WRK), cl1 = *(const half8*)(S + oB[1] + (64 + 32) * WRK);
    // every fragment of this K tile is in registers (or on its way): the next K tile must have landed, and this stage may be refilled
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const bool more = kt + 2 < KT;
    if (more) {
      __builtin_amdgcn_s_setprio(2);
      stage(buf, kt + 2);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc[t][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], ch0, acc[t][1][0], 0, 0, 0);
      acc[t][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], cl0, acc[t][1][0], 0, 0, 0);
      acc[t][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], ch0, acc[t][1][0], 0, 0, 0);
    }
    if (more) {                               // one operand piece behind each of the first seven MFMAs
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc[t][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], ch1, acc[t][1][1], 0, 0, 0);
      acc[t][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], cl1, acc[t][1][1], 0, 0, 0);
      acc[t][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], ch1, acc[t][1][1], 0, 0, 0);
    }
  };
  for (int kt = 0; kt < KT; kt += 2) {        // (K % 32 == 0: an even number of K tiles)
    ktile(std::integral_constant<int, 0>{}, kt);
    ktile(std::integral_constant<int, 1>{}, kt + 1);
  }

  // ---- epilogue, one channel tile at a time (gcn_tile.hip's, value for value): dp = D h0 + shift, gp = M1 h1, exact-f32 24 x 24 adjacency mix,
  //      ReLU, through the wave's own six 1 KiB pieces of stage 1 into rows of 8 consecutive channels per lane, residual, X2 / float32 stores
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  typedef const float __attribute__((address_space(4))) cfloat;
  const unsigned int arow = (unsigned int)N * 4u, tblrow = (unsigned int)N * 4u;
  const bool out_f32 = a.out_f32 != 0, has_res = a.Res != nullptr, relu = a.L.relu != 0;
  const float floor_v = relu ? 0.f : -3.4e38f;
  const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc((char*)a.Y + ((size_t)m_tile * 192 + 96 * wm) * (size_t)arow);
  const __amdgpu_buffer_rsrc_t resB = ehm_buffer_rsrc((has_res ? (const char*)a.Res : (const char*)a.Y) + ((size_t)m_tile * 192 + 96 * wm) * arow);
  const __amdgpu_buffer_rsrc_t dsB = ehm_buffer_rsrc(a.L.Ds), m1B = ehm_buffer_rsrc(a.L.M1s), shB = ehm_buffer_rsrc(a.L.shift);
  const int lr = lane >> 2, c8 = 8 * (lane & 3);
  const unsigned int orow = out_f32 ? tblrow : arow;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int n = 128 * n_tile + 64 * wn + 32 * cc + mi;
    const int ch0 = 128 * n_tile + 64 * wn + 32 * cc + c8;
    const unsigned int col_in = (unsigned int)(((ch0 >> 5) * 64 + (ch0 & 31)) * 2);
    const unsigned int col_out = out_f32 ? (unsigned int)ch0 * 4u : col_in;
    float dj[kJ], mj[kJ];
    const float sh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shB, (unsigned int)n * 4u, 0, 0));
    const unsigned int nrow = (unsigned int)n * (unsigned int)(kJ * 4);
#pragma unroll
    for (int q4 = 0; q4 < kJ / 4; ++q4) {
      const u32x4_t d4 = __builtin_amdgcn_raw_buffer_load_b128(dsB, nrow, 16 * q4, 0);
      const u32x4_t m4 = __builtin_amdgcn_raw_buffer_load_b128(m1B, nrow, 16 * q4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned int du = d4[i], mu = m4[i];
        dj[4 * q4 + i] = __builtin_bit_cast(float, du);
        mj[4 * q4 + i] = __builtin_bit_cast(float, mu);
      }
    }
    auto item_vrow = [&](int p, int it) -> unsigned int {
      const int rlq = 16 * it + lr;
      return (unsigned int)(24 * p + rlq + (rlq >= 24 ? 24 : 0));
    };
    u32x4_t rq[6];
    auto load_res_pass = [&](int p) {
      if (has_res) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
          const unsigned int vo = item_vrow(p, it) * arow + col_in;
          rq[2 * it] = __builtin_amdgcn_raw_buffer_load_b128(resB, vo, 0, kWLoadAux);
          rq[2 * it + 1] = __builtin_amdgcn_raw_buffer_load_b128(resB, vo + 64u, 0, kWLoadAux);
        }
      }
    };
    f32x2w dp[kJ], gp[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const f32x2w a0 = f32x2w{acc[j >> 3][cc][0][2 * (j & 7)], acc[j >> 3][cc][0][2 * (j & 7) + 1]};
      const f32x2w a1 = f32x2w{acc[j >> 3][cc][1][2 * (j & 7)], acc[j >> 3][cc][1][2 * (j & 7) + 1]};
      dp[j] = __builtin_elementwise_fma(f32x2w{dj[j], dj[j]}, a0, f32x2w{sh, sh});
      gp[j] = a1 * f32x2w{mj[j], mj[j]};
    }
    __builtin_amdgcn_sched_barrier(0);
    load_res_pass(0);
    float V[2][24];
    constexpr int GR = 4;
#pragma unroll
    for (int j0 = 0; j0 < kJ; j0 += GR) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long ag = (unsigned long long)(uintptr_t)(a.L.Aoff + j0 * kJ);
      unsigned int ag_lo = __builtin_amdgcn_readfirstlane((unsigned int)ag), ag_hi = __builtin_amdgcn_readfirstlane((unsigned int)(ag >> 32));
      asm volatile("" : "+s"(ag_lo), "+s"(ag_hi));
      const cfloat* Ag = (const cfloat*)(uintptr_t)(((unsigned long long)ag_hi << 32) | ag_lo);
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const int j = j0 + i;
        float s0 = dp[j][0], s1 = dp[j][1];
#pragma unroll
        for (int jp = 0; jp < kJ; ++jp) {
          const float c = Ag[i * kJ + jp];
          s0 = fmaf(c, gp[jp][0], s0);
          s1 = fmaf(c, gp[jp][1], s1);
        }
        asm volatile("" : "+v"(s0), "+v"(s1));
        V[0][j] = fmaxf(s0, floor_v);
        V[1][j] = fmaxf(s1, floor_v);
      }
    }
    const int wbase = WSTG + wave * 256 + 3072 * g + mi;             // scratch row 24 g + joint: pieces 3 g + (joint >> 3)  (WA_T = 3 pieces: the weight region's pieces follow)
    const int rbase = WSTG + wave * 256 + (lr & 7) * 32 + c8;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p == 1) load_res_pass(1);
#pragma unroll
      for (int k = 0; k < 24; ++k) lds[wbase + (k >> 3) * 1024 + (k & 7) * 32] = V[p][k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 tq[3][2];
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int ro = rbase + (2 * it + (lr >> 3)) * 1024;
        tq[it][0] = *(const f32x4*)(lds + ro);
        tq[it][1] = *(const f32x4*)(lds + ro + 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        float v[8] = {tq[it][0][0], tq[it][0][1], tq[it][0][2], tq[it][0][3], tq[it][1][0], tq[it][1][1], tq[it][1][2], tq[it][1][3]};
        if (has_res) {
          const half8 rh = __builtin_bit_cast(half8, rq[2 * it]), rlo = __builtin_bit_cast(half8, rq[2 * it + 1]);
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] += (float)rh[c] + (float)rlo[c];
        }
        const unsigned int vo = item_vrow(p, it) * orow + col_out;
        if (out_f32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v[0], v[1], v[2], v[3]}), yB, vo, 0, kWStoreAux);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v[4], v[5], v[6], v[7]}), yB, vo + 16u, 0, kWStoreAux);
        } else {
          half8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            hh[c] = (half_t)v[c];
            ll[c] = (half_t)(v[c] - (float)hh[c]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, kWStoreAux);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, kWStoreAux);
        }
      }
    }
  }
}

}  // namespace

// One conv per launch (ehm_gcn_hidden_layer, split-f16 mode, hid % 128 == 0).
int ehm_gcn_wide_layer_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_f32,
                            hipStream_t st) {
  if (h->precision != EHM_PREC_F16X3 || h->hid % 128 != 0 || rows_pad % 192 != 0 || h->hid < 32) {
    ehm_set_error("ehm_gcn_wide_layer_impl: split-f16 mode, hid %% 128 == 0 and rows_pad %% 192 == 0 only");
    return EHM_EINVAL;
  }
  WideArgs a;
  a.L = h->hidden[layer];
  a.X = X; a.Res = residual; a.Y = out;
  a.m_tiles = (int)(rows_pad / 192);
  a.out_f32 = out_f32 ? 1 : 0;
  hipLaunchKernelGGL(gcn_hidden_wide_kernel, dim3((unsigned)(a.m_tiles * (h->hid / 128))), dim3(256), 0, st, a);
  EHM_LAUNCH_CHECK();
  return 0;
}
