// Split-f16 Modulated-GCN hidden conv, 4-stage software pipeline (experiment, EHM_F16_PIPELINED=1; measured 3-5 % slower than gcn_f16.hip).
//
// Same maths, tile (192 rows x 64 channels x 2 branches, 4 waves, 2 blocks/CU), LDS swizzle idea and in-register epilogue as
// gcn_f16.hip.  What changes is the K loop.  Measured on the 2-stage BK=32 kernel (rocprofv3 SQ counters, tools/bench_hidden.py):
// with the global->LDS traffic removed the loop runs in 3150 cycles per K tile (2 co-resident waves x 1152 MFMA cycles = 2304),
// with it 4500 - the loads of tile k+1 are issued one tile ahead only, their L2/MALL latency under load exceeds one tile of
// MFMA work, and every wave sits in `s_waitcnt vmcnt(0)` + barrier.  Bigger tiles, register staging, persistent blocks and
// s_setprio did not move it.  Here:
//   * K tile = 16 (one MFMA k-step), FOUR LDS stages of 20 KiB (same 80 KiB per block): loads run THREE tiles ahead;
//   * counted waits: `s_waitcnt vmcnt(10)` = "my loads of tile k have landed, the 2 x 5 younger ones may stay in flight",
//     then a raw `s_barrier` (a __syncthreads() would make hipcc drain vmcnt(0) and serialise the pipeline again);
//   * operands in X2<16>: every 16-k group of a row is 16 hi + 16 lo halves = one 64-byte segment per row and K tile.
// Hazards: RAW - a wave waits for its own DMA of tile k (vmcnt) before the barrier that precedes the tile's first ds_read;
// WAR - stage (k+3)%4 == (k-1)%4 is refilled only after the barrier that every wave reaches after its last read of tile k-1.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int PK = 16;                       // K per pipeline stage
constexpr int PA_T = 192 * PK, PB_T = 128 * PK, PSTG = PA_T + PB_T;   // floats per stage: 5120 = 20 KiB

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

template <int PASSES, bool RES, bool OUT_SPLIT, int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gcn_hidden_f16p_kernel(const half_t* __restrict__ X, LayerDev L,
                                                                  const half_t* __restrict__ Res, float* __restrict__ Y,
                                                                  int m_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[NST * PSTG];   // 80 KiB, the only LDS object

  const int K = L.K, N = L.N;
  const int n_tiles = N / 64;
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  const int lin = ((total & 7) == 0) ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;   // XCD-aware tile order
  const int m_tile = lin / n_tiles, n_tile = lin % n_tiles;
  const size_t m0 = (size_t)m_tile * 192;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- DMA: one wave instruction = 16 rows x 64 B; lane -> row lane>>2, 16-byte chunk lane&3.  Physical chunk c of row r holds
  //      logical chunk c ^ ((r>>2)&3); with r = 16*(wave + 4i) + (lane>>2) that key is (lane>>4)&3 for every instruction.
  const int ld_r = lane >> 2;
  const int swz = ((lane & 3) ^ ((lane >> 4) & 3)) << 2;             // floats
  const float* pA = (const float*)X + (m0 + 16 * wave + ld_r) * K + swz;
  const float* pB = (const float*)L.Ws16 + ((size_t)n_tile * 128 + 16 * wave + ld_r) * K + swz;
  const size_t row64 = (size_t)64 * K;                                // 4 waves x 16 rows between a wave's consecutive instructions
  auto stage = [&](int kt) {
    float* base = lds + (kt % NST) * PSTG;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pA + i * row64 + kt * PK), (AS3 void*)(base + (wave + 4 * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pB + i * row64 + kt * PK), (AS3 void*)(base + PA_T + (wave + 4 * i) * 256), 16, 0, 0);
  };

  // ---- fragments (v_mfma_f32_32x32x16_f16: lane l holds row l&31, k = 8*(l>>5) .. +7)
  const int mi = lane & 31, g = lane >> 5;
  const int rA = 96 * wm + 48 * ((mi >> 2) & 1) + (mi & 3) + 4 * (mi >> 3);   // row permutation of the in-register epilogue
  const int rB = 32 * wn + mi;
  const int keyA = (rA >> 2) & 3, keyB = (rB >> 2) & 3;                       // unchanged by +16t / +64u
  const int oAh = rA * PK + ((g ^ keyA) << 2), oAl = rA * PK + (((2 + g) ^ keyA) << 2);   // float offsets inside a stage
  const int oBh = PA_T + rB * PK + ((g ^ keyB) << 2), oBl = PA_T + rB * PK + (((2 + g) ^ keyB) << 2);

  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }

  const int KT = K / PK;
  stage(0);
  if (NST == 4) { stage(1); stage(2); }
  for (int kt = 0; kt < KT; ++kt) {
    if (NST == 4) {
      // my DMA of tile kt has landed (two younger tiles = 10 instructions may still fly; fewer at the tail)
      if (kt + 2 < KT) wait_vmcnt<10>();
      else if (kt + 1 < KT) wait_vmcnt<5>();
      else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();                      // everyone's tile kt is in LDS; everyone left stage (kt-1)%NST
    __builtin_amdgcn_sched_barrier(0);
    if (NST == 4) { if (kt + 3 < KT) stage(kt + 3); }  // refills stage (kt-1)%4
    else if (kt + 1 < KT) stage(kt + 1);               // 2 stages (20 KiB each, 3 blocks per CU): refills the stage everyone just left
    const float* S = lds + (kt % NST) * PSTG;
    half8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      ah[t] = *(const half8*)(S + oAh + 16 * t * PK);
      if (PASSES == 3) al[t] = *(const half8*)(S + oAl + 16 * t * PK);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bh[u] = *(const half8*)(S + oBh + 64 * u * PK);
      if (PASSES == 3) bl[u] = *(const half8*)(S + oBl + 64 * u * PK);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (PASSES == 3) {                                // small cross terms first, leading term last
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[1], acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[1], acc1[t], 0, 0, 0);
      }
      acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[0], acc0[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[1], acc1[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);                  // keep the next iteration's wait / barrier behind this tile's MFMAs
  }

  // ---- epilogue (Ds/M1s carry 1/w_scale); X2<16> input / output ----
  const int n = 64 * n_tile + 32 * wn + mi;
  {
    float dj[kJ], mj[kJ];
    const float sh = L.shift[n];
#pragma unroll
    for (int j = 0; j < kJ; ++j) { dj[j] = L.Ds[j * N + n]; mj[j] = L.M1s[j * N + n]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 48; ++q) {   // fold modulation / BatchNorm scale into the accumulators in place
      acc0[q >> 4][q & 15] = fmaf(dj[q % 24], acc0[q >> 4][q & 15], sh);
      acc1[q >> 4][q & 15] *= mj[q % 24];
    }
  }
#pragma unroll
  for (int beta = 0; beta < 2; ++beta) {
    const size_t rowb = m0 + 96 * wm + 48 * g + 24 * beta;
    __builtin_amdgcn_sched_barrier(0);
    float res[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) res[j] = RES ? split_load_pair<16>(Res, rowb + j, n, N) : 0.f;
    float d0[kJ], g1[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const int q = 24 * beta + j;
      d0[j] = acc0[q >> 4][q & 15];
      g1[j] = acc1[q >> 4][q & 15];
    }
    gcn_mix_store<OUT_SPLIT, 16>(d0, g1, res, n, N, rowb, L.Aoff, Y, L.relu != 0);
  }
}

template <int PASSES, int NST, int OCC>
int launch(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_split, hipStream_t st) {
  const int m_tiles = (int)(rows_pad / 192);
  const int blocks = m_tiles * (h->hid / 64);
  const LayerDev& L = h->hidden[layer];
  const half_t* x = (const half_t*)X;
  const half_t* r = (const half_t*)residual;
  float* y = (float*)out;
  if (residual) {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16p_kernel<PASSES, true, true, NST, OCC>), dim3(blocks), dim3(256), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16p_kernel<PASSES, true, false, NST, OCC>), dim3(blocks), dim3(256), 0, st, x, L, r, y, m_tiles);
  } else {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16p_kernel<PASSES, false, true, NST, OCC>), dim3(blocks), dim3(256), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16p_kernel<PASSES, false, false, NST, OCC>), dim3(blocks), dim3(256), 0, st, x, L, r, y, m_tiles);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int ehm_gcn_hidden_f16p_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad,
                             bool out_split, hipStream_t st) {
  if (h->hid % 64 != 0 || h->hidden[layer].K % PK != 0 || h->hidden[layer].K / PK < 3) {
    ehm_set_error("pipelined split-f16 conv needs hid %% 64 == 0 and K >= 48");
    return EHM_EINVAL;
  }
  if (h->tile_override == 3) {   // experiment: 2 stages of 20 KiB, 3 blocks per CU (3 waves per SIMD, <= 168 VGPRs)
    if (h->precision == EHM_PREC_F16X3) return launch<3, 2, 3>(h, layer, X, residual, out, rows_pad, out_split, st);
    return launch<1, 2, 3>(h, layer, X, residual, out, rows_pad, out_split, st);
  }
  if (h->precision == EHM_PREC_F16X3) return launch<3, 4, 2>(h, layer, X, residual, out, rows_pad, out_split, st);
  return launch<1, 4, 2>(h, layer, X, residual, out, rows_pad, out_split, st);
}
