// Modulated-GCN hidden conv on the f16 matrix cores of gfx950, float32-grade accuracy by operand splitting.
//
// Same conv as gcn.hip's gcn_hidden_kernel (modulated_gcn.py:21-28,38-42; modulated_gcn_conv.py:39-50) and the same
// block / wave tiling, LDS swizzle, XCD-aware order and in-register epilogue.  What changes is the MFMA:
// v_mfma_f32_32x32x16_f16 runs at 16x the rate of the f32-input MFMA (2.5 PFLOP/s dense vs 157 TFLOP/s), and an f32
// value x is carried as two f16 numbers  x = hi + lo,  hi = rn16(x), lo = rn16(x - hi)  (22 significant bits).
//   PASSES == 3:  A.B ~ Ahi.Bhi + Ahi.Blo + Alo.Bhi   (dropped term ~2^-22 relative; f16 products are exact in the
//                 f32 accumulator) -> "f16x3": f32-grade results at up to 5.3x the f32 MFMA roofline.
//   PASSES == 1:  Ahi.Bhi only -> plain f16 denoiser (BASELINE config 5's "fp16 denoiser"); NOT parity-grade.
// Activations and weights live in the "X2" format of gcn_dev.h: per row, every 32-k group is 32 hi halves then
// 32 lo halves = the same 128-byte line a float32 row tile occupies, so HBM/LDS traffic per element is unchanged
// and the global_load_lds staging is byte-identical to the f32 kernel.  Weights are pre-scaled by a power of two
// (w_scale) so that their lo parts stay out of the f16 subnormal range; the epilogue constants carry 1/w_scale.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// WM x WN = how many 96-row groups x 32-channel groups one wave owns.  Block = 2 x 2 waves, so the block tile is
// (192*WM) rows x (64*WN) channels (x2 branches):
//   <1,1>: 192 x 64,  80 KiB LDS, 2 blocks/CU (2 waves/SIMD)          - small launches
//   <1,2>: 192 x 128, 112 KiB LDS, 1 block/CU (1 wave/SIMD)          - 30 % less L2->LDS traffic, measured slower (see launch())
//   <2,2>: 384 x 128 would halve the traffic but hipcc spills its 384 accumulator registers inside the K loop.
template <int PASSES, bool RES, bool OUT_SPLIT, int WM, int WN, int EXP = 8, int NWM = 2, int NWN = 2>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN == 8 || WM * WN == 1) ? 2 : 1) void gcn_hidden_f16_kernel(const half_t* __restrict__ X, LayerDev L,
                                                                                     const half_t* __restrict__ Res,
                                                                                     float* __restrict__ Y, int m_tiles) {
  constexpr int NW = NWN * NWM;              // waves per block: NWM along rows x NWN along channels
  constexpr int BM_ = 96 * NWM * WM;         // rows per block
  constexpr int BR_ = 64 * WN * NWN;         // weight rows per block (2 branches x 32*WN*NWN channels) = consecutive packed 64-channel tiles
  constexpr int A_T = BM_ * BK, B_T = BR_ * BK, STG = A_T + B_T;   // floats
  constexpr int NLA = BM_ / (8 * NW), NLB = BR_ / (8 * NW);       // staging wave-instructions (8 rows each) per wave
  __shared__ __attribute__((aligned(16))) float lds[2 * STG];

  const int K = L.K, N = L.N;
  const int n_tiles = N / (32 * WN * NWN);
  const int total = m_tiles * n_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // Persistent blocks: the grid is capped at what is co-resident (2 blocks per CU) and every block walks tiles
  // bid, bid + grid, ... .  With 1024 tiles on 512 slots each block does exactly two: no CU ends up with 3 or 5 of the
  // 4-per-CU average (measured: kernel span 2.5x one block's life time instead of 2x with one block per tile).
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
  // XCD-aware order: tile % 8 == blockIdx % 8 == XCD, which owns a contiguous run of (row tile, all channel tiles)
  const int lin = ((total & 7) == 0 && (gridDim.x & 7) == 0) ? (tile & 7) * (total >> 3) + (tile >> 3) : tile;
  const int m_tile = lin / n_tiles, n_tile = lin % n_tiles;
  const size_t m0 = (size_t)m_tile * BM_;

  // ---- global -> LDS DMA: rows of 128 B (= K-tile of 32: 4 hi chunks + 4 lo chunks), physical chunk c holds logical c ^ key(row)
  const int ld_r = lane >> 3, ld_c = lane & 7;
  const float* Xf = (const float*)X;                 // row stride K floats == K*4 bytes in both formats
  const float* Wf = (const float*)L.Ws;
  // row r_i = 8*(wave + 4i) + ld_r = r_0 + 32 i  =>  same swizzle key for every i: keep ONE pointer per operand (the
  // 2 x 20 per-instruction pointers cost 40 VGPRs and pushed the 384 accumulator registers of the big tile into scratch)
  const int r0 = 8 * wave + ld_r;
  const int swz = (ld_c ^ ((r0 >> 1) & 7)) << 2;
  const float* pA = Xf + (m0 + r0) * K + swz;
  const float* pB = Wf + ((size_t)n_tile * BR_ + r0) * K + swz;
  const size_t row32 = (size_t)8 * NW * K;   // rows between consecutive staging instructions of one wave
  auto stage = [&](int buf, int kt) {
    float* base = lds + buf * STG;
#pragma unroll
    for (int i = 0; i < NLA; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pA + i * row32 + kt * BK), (AS3 void*)(base + (wave + NW * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NLB; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pB + i * row32 + kt * BK), (AS3 void*)(base + A_T + (wave + NW * i) * 256), 16, 0, 0);
  };

  // register-staged alternative (EXP & 8): global_load_dwordx4 -> VGPR -> ds_write_b128 to the same LDS image.  Measured
  // (rocprofv3 SQ_WAVE_CYCLES, with vs without the loads): one global_load_lds costs the issuing wave ~134 cycles, ten of
  // them per K tile = more than the tile's 1152 MFMA cycles; a plain load + LDS write is several times cheaper to issue.
  f32x4 sreg[(EXP & 8) ? NLA + NLB : 1];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) sreg[i] = *(const f32x4*)(pA + i * row32 + kt * BK);
#pragma unroll
    for (int i = 0; i < NLB; ++i) sreg[NLA + i] = *(const f32x4*)(pB + i * row32 + kt * BK);
  };
  auto lwrite = [&](int buf) {
    float* base = lds + buf * STG + lane * 4;
#pragma unroll
    for (int i = 0; i < NLA; ++i) *(f32x4*)(base + (wave + NW * i) * 256) = sreg[i];
#pragma unroll
    for (int i = 0; i < NLB; ++i) *(f32x4*)(base + A_T + (wave + NW * i) * 256) = sreg[NLA + i];
  };

  // ---- fragments: v_mfma_f32_32x32x16_f16 lane l holds row l&31, k = 8*(l>>5) .. +7 of a 16-wide step
  const int mi = lane & 31, g = lane >> 5;
  // row permutation inside a 96-row group (same as the f32 kernel): MFMA row i of tile t <-> 48*((i>>2)&1) + 16t + (i&3) + 4*(i>>3)
  const int rA = 96 * WM * wm + 48 * ((mi >> 2) & 1) + (mi & 3) + 4 * (mi >> 3);
  // channel group cg of this wave: channels 64*WN*n_tile + 32*(WN*wn + cg) + mi; packed weight tile = (32*(WN*wn+cg))/64
  int rB[WN];
#pragma unroll
  for (int cg = 0; cg < WN; ++cg) {
    const int ch0 = 32 * (WN * wn + cg);
    rB[cg] = (ch0 >> 6) * 128 + (ch0 & 63) + mi;
  }
  const int keyA = (rA >> 1) & 7;                     // +16t, +96 do not change (row>>1)&7
  int keyB[WN];
#pragma unroll
  for (int cg = 0; cg < WN; ++cg) keyB[cg] = (rB[cg] >> 1) & 7;   // +64 (branch) does not change it

  f32x16 acc0[WM][WN][3], acc1[WM][WN][3];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[a][c][t][r] = 0.f; acc1[a][c][t][r] = 0.f; }

  const int KT = K / BK;
  if (EXP & 8) {
    gload(0);
    lwrite(0);
    __syncthreads();
  } else {
    stage(0, 0);
  }
  for (int kt = 0; kt < KT; ++kt) {
    if (EXP & 8) {
      if (kt + 1 < KT) gload(kt + 1);                      // in flight during this tile's MFMAs
    } else {
      if (!(EXP & 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (!(EXP & 1) && kt + 1 < KT) stage((kt + 1) & 1, kt + 1);
    }
    const float* As = lds + ((EXP & 1) ? 0 : (kt & 1)) * STG + ((EXP & 4) ? (kt & 0) : 0);
    const float* Bs = As + A_T;
#pragma unroll
    for (int s = 0; s < 2; ++s) {                          // two 16-wide k steps per 32-wide tile
      const int ch = 2 * s + g, cl = 4 + 2 * s + g;         // logical hi / lo chunk (8 halves = 16 B each)
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        half8 ah[3], al[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          ah[t] = *(const half8*)(As + (rA + 96 * a + 16 * t) * BK + ((ch ^ keyA) << 2));
          if (PASSES == 3) al[t] = *(const half8*)(As + (rA + 96 * a + 16 * t) * BK + ((cl ^ keyA) << 2));
        }
#pragma unroll
        for (int c = 0; c < WN; ++c) {
          half8 bh[2], bl[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            bh[u] = *(const half8*)(Bs + (rB[c] + 64 * u) * BK + ((ch ^ keyB[c]) << 2));
            if (PASSES == 3) bl[u] = *(const half8*)(Bs + (rB[c] + 64 * u) * BK + ((cl ^ keyB[c]) << 2));
          }
          if (EXP & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            if (PASSES == 3) {                              // small cross terms first, leading term last
              acc0[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[0], acc0[a][c][t], 0, 0, 0);
              acc1[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[1], acc1[a][c][t], 0, 0, 0);
              acc0[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[0], acc0[a][c][t], 0, 0, 0);
              acc1[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[1], acc1[a][c][t], 0, 0, 0);
            }
            acc0[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[0], acc0[a][c][t], 0, 0, 0);
            acc1[a][c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[1], acc1[a][c][t], 0, 0, 0);
          }
          if (EXP & 16) __builtin_amdgcn_s_setprio(0);
          if (WM * WN > 1) __builtin_amdgcn_sched_barrier(0);   // big tile: keep hipcc from hoisting every fragment load (spills)
        }
      }
    }
    if (EXP & 8) {
      if (kt + 1 < KT) lwrite((kt + 1) & 1);               // the other buffer: everyone left it at the previous barrier
      __syncthreads();
    }
  }

  // ---- epilogue (Ds/M1s carry 1/w_scale), one (channel group, row group) at a time; staged with sched_barrier so that at most
  //      the accumulators + one 24-wide batch of loads are live (the staging registers of the K loop are dead here) ----
#pragma unroll
  for (int c = 0; c < WN; ++c) {
    const int n = 32 * WN * NWN * n_tile + 32 * (WN * wn + c) + mi;
    {
      float dj[kJ], mj[kJ];
      const float sh = L.shift[n];
#pragma unroll
      for (int j = 0; j < kJ; ++j) { dj[j] = L.Ds[j * N + n]; mj[j] = L.M1s[j * N + n]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int q = 0; q < 48; ++q) {   // fold modulation / BatchNorm scale into the accumulators in place
          acc0[a][c][q >> 4][q & 15] = fmaf(dj[q % 24], acc0[a][c][q >> 4][q & 15], sh);
          acc1[a][c][q >> 4][q & 15] *= mj[q % 24];
        }
    }
#pragma unroll
    for (int a = 0; a < WM; ++a) {
#pragma unroll
      for (int beta = 0; beta < 2; ++beta) {
        const size_t rowb = m0 + 96 * (WM * wm + a) + 48 * g + 24 * beta;
        __builtin_amdgcn_sched_barrier(0);
        float res[kJ];
#pragma unroll
        for (int j = 0; j < kJ; ++j) res[j] = RES ? split_load_pair(Res, rowb + j, n, N) : 0.f;
        float d0[kJ], g1[kJ];
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
          const int q = 24 * beta + j;
          d0[j] = acc0[a][c][q >> 4][q & 15];
          g1[j] = acc1[a][c][q >> 4][q & 15];
        }
        gcn_mix_store<OUT_SPLIT>(d0, g1, res, n, N, rowb, L.Aoff, Y, L.relu != 0);
      }
    }
  }
  }   // persistent tile loop
}

// float32 [rows][K] <-> X2 split format (tests / interop; the sampler never needs them)
template <int G>
__global__ void pack_x2_kernel(const float* __restrict__ X, half_t* __restrict__ Y, int64_t rows, int K) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  split_store<G>(Y, (size_t)(i / K), (int)(i % K), K, X[i]);
}
template <int G>
__global__ void unpack_x2_kernel(const half_t* __restrict__ X, float* __restrict__ Y, int64_t rows, int K) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  Y[i] = split_load<G>(X, (size_t)(i / K), (int)(i % K), K);
}

template <int PASSES, int WM, int WN, int EXP, int NWM = 2, int NWN = 2>
int launch_cfg(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_split,
               hipStream_t st) {
  const int m_tiles = (int)(rows_pad / (96 * NWM * WM));
  const int tiles = m_tiles * (h->hid / (32 * WN * NWN));
  const int slots = ehm_num_cus() * ((WM * WN == 1 && NWM * NWN == 4) ? 2 : 1);      // co-resident blocks (LDS-limited)
  const int blocks = (tiles < slots || !h->persistent) ? tiles : slots;
  const LayerDev& L = h->hidden[layer];
  const half_t* x = (const half_t*)X;
  const half_t* r = (const half_t*)residual;
  float* y = (float*)out;
  if (residual) {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, true, true, WM, WN, EXP, NWM, NWN>), dim3(blocks), dim3(64 * NWM * NWN), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, true, false, WM, WN, EXP, NWM, NWN>), dim3(blocks), dim3(64 * NWM * NWN), 0, st, x, L, r, y, m_tiles);
  } else {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, WM, WN, EXP, NWM, NWN>), dim3(blocks), dim3(64 * NWM * NWN), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, false, WM, WN, EXP, NWM, NWN>), dim3(blocks), dim3(64 * NWM * NWN), 0, st, x, L, r, y, m_tiles);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

template <int PASSES>
int launch(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_split, hipStream_t st) {
  // Measured on MI355X at B=256 (tools/bench_hidden.py): 192x64 tiles at 2 blocks/CU 191 us, 192x128 tiles at 1 block/CU
  // (1 wave/SIMD) 217 us - halving the L2->LDS traffic does not pay for the lost latency hiding, so the small tile is the
  // default and the big one stays selectable for experiments.
  const int force = h->tile_override;   // 0 / 1 = 192x64, 2 = 192x128
  if (force >= 3 && force <= 7) {   // timing experiments (results are wrong on purpose): bit0 no DMA in the K loop, bit1 no barrier, bit2 loop-invariant LDS reads
    const int m_tiles = (int)(rows_pad / 192);
    const dim3 grid(m_tiles * (h->hid / 64));
    const half_t* x = (const half_t*)X;
    const half_t* r0 = nullptr;
    if (force == 3) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, 1, 1, 1>), grid, dim3(256), 0, st, x, h->hidden[layer], r0, (float*)out, m_tiles);
    if (force == 4) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, 1, 1, 3>), grid, dim3(256), 0, st, x, h->hidden[layer], r0, (float*)out, m_tiles);
    if (force == 5) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, 1, 1, 5>), grid, dim3(256), 0, st, x, h->hidden[layer], r0, (float*)out, m_tiles);
    if (force == 6) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, 1, 1, 7>), grid, dim3(256), 0, st, x, h->hidden[layer], r0, (float*)out, m_tiles);
    if (force == 7) hipLaunchKernelGGL((gcn_hidden_f16_kernel<PASSES, false, true, 1, 1, 0>), grid, dim3(256), 0, st, x, h->hidden[layer], r0, (float*)out, m_tiles);   // global_load_lds staging
    EHM_LAUNCH_CHECK();
    return 0;
  }
  if (force == 2 && h->hid % 128 == 0)
    return launch_cfg<PASSES, 1, 2, 8>(h, layer, X, residual, out, rows_pad, out_split, st);
  if (force == 10 && h->hid % 128 == 0)    // 8 waves as 2 x 4: 192 rows x 128 channels, 96x32 per wave, 1 block/CU
    return launch_cfg<PASSES, 1, 1, 0, 2, 4>(h, layer, X, residual, out, rows_pad, out_split, st);
  if (force == 11 && rows_pad % 384 == 0)   // 8 waves as 4 x 2: 384 rows x 64 channels
    return launch_cfg<PASSES, 1, 1, 0, 4, 2>(h, layer, X, residual, out, rows_pad, out_split, st);
  if (force == 9) return launch_cfg<PASSES, 1, 1, 16>(h, layer, X, residual, out, rows_pad, out_split, st);   // s_setprio around the MFMA clusters
  if (force == 8 && h->hid % 128 == 0 && rows_pad % 384 == 0)      // 8 waves: 384 rows x 128 channels, DMA staging
    return launch_cfg<PASSES, 1, 2, 0, 4>(h, layer, X, residual, out, rows_pad, out_split, st);
  if (h->reg_staging) return launch_cfg<PASSES, 1, 1, 8>(h, layer, X, residual, out, rows_pad, out_split, st);
  return launch_cfg<PASSES, 1, 1, 0>(h, layer, X, residual, out, rows_pad, out_split, st);
}

}  // namespace

int ehm_gcn_hidden_f16_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad,
                            bool out_split, hipStream_t st) {
  if (h->pipelined == 1) return ehm_gcn_hidden_f16p_impl(h, layer, X, residual, out, rows_pad, out_split, st);
  if (h->pipelined == 2 && h->tile_override == 0) return ehm_gcn_hidden_f16r_impl(h, layer, X, residual, out, rows_pad, out_split, st);
  if (h->precision == EHM_PREC_F16X3) return launch<3>(h, layer, X, residual, out, rows_pad, out_split, st);
  return launch<1>(h, layer, X, residual, out, rows_pad, out_split, st);
}

extern "C" int ehm_gcn_pack_activations(const float* X, void* X2, int64_t rows, int K, int group, void* stream) {
  EHM_CHECK_ARG(X && X2 && rows > 0 && K > 0 && K % 32 == 0 && (group == 16 || group == 32));
  const dim3 grid((unsigned)ceil_div(rows * K, 256));
  if (group == 16) hipLaunchKernelGGL(pack_x2_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, rows, K);
  else hipLaunchKernelGGL(pack_x2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, rows, K);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_unpack_activations(const void* X2, float* X, int64_t rows, int K, int group, void* stream) {
  EHM_CHECK_ARG(X && X2 && rows > 0 && K > 0 && K % 32 == 0 && (group == 16 || group == 32));
  const dim3 grid((unsigned)ceil_div(rows * K, 256));
  if (group == 16) hipLaunchKernelGGL(unpack_x2_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)X2, X, rows, K);
  else hipLaunchKernelGGL(unpack_x2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)X2, X, rows, K);
  EHM_LAUNCH_CHECK();
  return 0;
}
