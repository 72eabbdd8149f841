// Split-f16 Modulated-GCN hidden conv with register double-buffered MFMA fragments (EHM_F16_PIPELINED=2).
//
// Same maths, tile (192 rows x 64 channels x 2 branches, 4 waves, 2 blocks/CU), X2<32> operands, LDS image / swizzle and
// in-register epilogue as gcn_f16.hip.  What changes is the schedule of the K loop.  hipcc's schedule of that kernel keeps the
// fragment registers minimal: it issues 2-4 ds_read_b128, waits lgkmcnt(0), issues 2-12 MFMAs, and repeats - six exposed LDS
// round trips per K tile, and the ten global_load_lds of the next tile in one burst right behind the barrier.  Here the loop
// is written in half-tile phases with TWO fragment sets:
//   phase A(k): MFMAs of (tile k, k-step 0) from set 0  |  ds_reads of (tile k, k-step 1) into set 1
//   -- s_waitcnt vmcnt(0) lgkmcnt(0) + barrier: every wave has all of tile k in registers, tile k+1 is complete in LDS --
//   phase B(k): MFMAs of (tile k, k-step 1) from set 1  |  ds_reads of (tile k+1, k-step 0) into set 0  |  DMA of tile k+2
// so every LDS read has half a tile (576 MFMA cycles) to land, every DMA a whole tile, and the only exposed wait is the one
// barrier per tile.  __builtin_amdgcn_sched_group_barrier pins the interleaving (1 MFMA : 1-2 reads : 1 DMA).
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// aux bits of the output stores (16 = sc1, write-through).  Tried to spread the end-of-kernel L2 write-back of the 48 MiB output
// over the kernel's run time: launch cadence unchanged (156.1 vs 156.6 us), so plain stores stay the default.
#ifndef EHM_STORE_AUX
#define EHM_STORE_AUX 0
#endif
constexpr int kStoreAux = EHM_STORE_AUX;
#ifndef EHM_MIX_MFMA
#define EHM_MIX_MFMA 0   // 1: the 24x24 adjacency mix of the epilogue as split-f16 MFMA with v_permlane32_swap-built operands instead of
                         // 1152 v_fmac per wave.  Parity-green, but measured neutral (144.5 vs 144.9 ms per DDPM-100 call, A/B in one
                         // session): the epilogue is bound by load latency behind the other block's DMA stream, not by VALU issue
#endif
constexpr int RK = 32;                                        // K per tile
// Block shape: NWN = waves along the channel axis.  2: 4 waves, 192 rows x 64 channels, 40 KiB per stage, 2 blocks per CU (default);
// 4: 8 waves, 192 rows x 128 channels, 56 KiB per stage, 1 block per CU - 30 % fewer DMA bytes and instructions per MFMA.
template <int NWN>
struct Shape {
  static constexpr int NW = 2 * NWN;              // waves per block (2 along rows x NWN along channels, 96 x 32 per wave)
  static constexpr int CH = 32 * NWN;             // channels per block
  static constexpr int BROWS = 2 * CH;            // weight rows per block (both branches)
  static constexpr int A_T = 192 * RK, B_T = BROWS * RK, STG = A_T + B_T;   // floats
  static constexpr int NA = 24 / NW, NB = 4;      // DMA instructions per wave and K tile (8 rows each)
  static constexpr int UPR = CH / 8;              // epilogue work items (8 channels) per row
};

// Build with EHM_HIPCC_FLAGS=-DEHM_STAMPS to record per-block phase time stamps (tools/stamp_hidden.py): slot i of block b at
// g_dbg[16 b + i]; STAMP = s_memrealtime (100 MHz), STAMPC = s_memtime (shader clock).
#ifdef EHM_STAMPS
__device__ unsigned long long* g_dbg = nullptr;
#define STAMP(i) do { if (g_dbg && threadIdx.x == 0) g_dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMPC(i) do { if (g_dbg && threadIdx.x == 0) g_dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPC(i) do { } while (0)
#endif

template <int PASSES>
struct Frags {
  half8 ah[3], al[3], bh[2], bl[2];
};

// One 192-row x 64-channel output tile of one hidden conv.  `lds` = the block's 80 KiB; no barrier at the end (the caller
// must put one between this tile's last LDS reads and its next use of `lds`).  RES / OUT_SPLIT are compile-time constants in
// the one-launch-per-conv kernel and runtime (block-uniform) values in the chained kernel.
// AUX = cache-policy bits of the output stores, IN_AUX = of the activation loads (A-operand DMA, residual); 16 = sc1 = agent scope
// (stores write through, loads bypass the CU's L1), used by the chained kernel.  `ready()` is called after the weight DMA of
// the first two K tiles has been issued and before the first activation byte is requested.
template <int PASSES, int AUX, int IN_AUX, int NWN, class Ready>
__device__ __forceinline__ void f16r_tile(float* lds, const half_t* __restrict__ X, const LayerDev& L, const half_t* __restrict__ Res,
                                          float* __restrict__ Y, int m_tile, int n_tile, const bool RES, const bool OUT_SPLIT, Ready ready) {
  STAMP(0); STAMPC(4);
  typedef Shape<NWN> SH;
  constexpr int NW = SH::NW, CH = SH::CH, RA_T = SH::A_T, RSTG = SH::STG, NA = SH::NA;
  const int K = L.K, N = L.N;
  const size_t m0 = (size_t)m_tile * 192;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- DMA: one wave instruction = 8 rows x 128 B; physical 16-byte chunk c of row r holds logical chunk c ^ ((r>>1)&7)
  const int ld_r = lane >> 3, ld_c = lane & 7;
  const int r0 = 8 * wave + ld_r;
  const int swz = (ld_c ^ ((r0 >> 1) & 7)) << 2;             // r0 + 32 i keeps the key
  const float* pA = (const float*)X + (m0 + r0) * K + swz;
  const float* pB = (const float*)L.Ws + ((size_t)n_tile * SH::BROWS + r0) * K + swz;
  const size_t row32 = (size_t)8 * NW * K;            // rows between a wave's consecutive DMA instructions (keeps the swizzle key: multiple of 16)
  auto dma_a = [&](int buf, int kt, int i) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)(pA + i * row32 + kt * RK), (AS3 void*)(lds + buf * RSTG + (wave + NW * i) * 256), 16, 0, IN_AUX);
  };
  auto dma_b = [&](int buf, int kt, int i) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)(pB + i * row32 + kt * RK), (AS3 void*)(lds + buf * RSTG + RA_T + (wave + NW * i) * 256), 16, 0, 0);
  };
  auto stage = [&](int buf, int kt) {
#ifndef EHM_EXP_NO_A_DMA      // timing experiments only (wrong results): how much of the K loop is DMA issue?
#pragma unroll
    for (int i = 0; i < NA; ++i) dma_a(buf, kt, i);
#endif
#ifndef EHM_EXP_NO_B_DMA
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_b(buf, kt, i);
#endif
  };

  // ---- fragments (v_mfma_f32_32x32x16_f16: lane l holds row l&31, k = 8*(l>>5) .. +7 of a 16-wide step)
  const int mi = lane & 31, g = lane >> 5;
  // Row permutation of the in-register epilogue: MFMA row i of tile t <-> tile row 48*((i>>2)&1) + 24*(i&1) + ((i>>1)&1) + 2*(i>>3) + 8t
  // of the wave's 96 rows.  With the C layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) lane (mi, g) then owns, for ONE channel, all
  // 24 joints of bodies 2g and 2g+1, and joint j of the two bodies sits in the ADJACENT registers 2*(j&7), 2*(j&7)+1 of accumulator j>>3:
  // the 24x24 adjacency mix runs as v_pk_fma_f32 on register pairs without a single move.
  const int rA = 96 * wm + 48 * ((mi >> 2) & 1) + 24 * (mi & 1) + ((mi >> 1) & 1) + 2 * (mi >> 3);
  const int rB = ((32 * wn) >> 6) * 128 + ((32 * wn) & 63) + mi;   // packed weights: per 64 channels, 64 rows of W0 then 64 rows of W1
  // swizzle key (row>>1)&7: +8t flips bit 2 for t = 1 (12*(mi&1) + (mi>>3) + 4t), +64u leaves it.  Conflict-free for ds_read_b128's
  // lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}: their 16 rows carry every key twice, once on an even and once on an odd row.
  const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;
  // float offsets inside a stage: [k-step][hi/lo][t odd] for A, [k-step][hi/lo] for B; logical chunk 2s+g (hi), 4+2s+g (lo)
  int oA[2][2][2], oB[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int c = 4 * hl + 2 * s + g;
#pragma unroll
      for (int o = 0; o < 2; ++o) oA[s][hl][o] = rA * RK + (((c ^ keyA) ^ (4 * o)) << 2);
      oB[s][hl] = RA_T + rB * RK + ((c ^ keyB) << 2);
    }
  auto read_frags = [&](Frags<PASSES>& f, int buf, int s) {
    const float* S = lds + buf * RSTG;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f.ah[t] = *(const half8*)(S + oA[s][0][t & 1] + 8 * t * RK);
      if (PASSES == 3) f.al[t] = *(const half8*)(S + oA[s][1][t & 1] + 8 * t * RK);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f.bh[u] = *(const half8*)(S + oB[s][0] + 64 * u * RK);
      if (PASSES == 3) f.bl[u] = *(const half8*)(S + oB[s][1] + 64 * u * RK);
    }
  };

  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }

  auto mfmas = [&](const Frags<PASSES>& f) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (PASSES == 3) {                                // small cross terms first, leading term last
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[1], acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[1], acc1[t], 0, 0, 0);
      }
      acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[0], acc0[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[1], acc1[t], 0, 0, 0);
    }
  };
  constexpr int NM = PASSES == 3 ? 18 : 6;     // MFMAs per phase
  constexpr int NR = PASSES == 3 ? 10 : 5;     // ds_read_b128 per phase
  // sched_group_barrier masks: 0x008 MFMA, 0x100 DS read, 0x010 VMEM
  auto pin_reads = [&]() {                      // MFMA, read, MFMA, read, ... then the remaining MFMAs
#pragma unroll
    for (int i = 0; i < (NR < NM ? NR : NM); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  };
  auto pin_reads_dma = [&]() {                  // first the reads (one per MFMA), then the ten DMAs spread over the remaining MFMAs
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (PASSES == 3 && NA == 6) {     // 8 MFMAs left: 2,2,1,1,1,1,1,1 DMAs behind them
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    } else if constexpr (PASSES == 3) {         // wide tile: 7 DMAs, one behind each of the next 7 MFMAs
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 10, 0);
    }
  };

  const int KT = K / RK;
  Frags<PASSES> f0, f1;
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_b(0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_b(1, 1, i);
  ready();
#pragma unroll
  for (int i = 0; i < NA; ++i) dma_a(0, 0, i);
#pragma unroll
  for (int i = 0; i < NA; ++i) dma_a(1, 1, i);
  if constexpr (NA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile 0 landed (the activation half of tile 1 may still fly)
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  __syncthreads();
  read_frags(f0, 0, 0);
  STAMP(1); STAMPC(5);

  for (int kt = 0; kt < KT - 2; ++kt) {
    const int buf = kt & 1;
    // phase A
    read_frags(f1, buf, 1);
    mfmas(f0);
    pin_reads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // tile kt is in everyone's registers; tile kt+1 is complete in LDS
    // phase B
    read_frags(f0, buf ^ 1, 0);                        // before the DMA in program order: hipcc cannot tell the two stages apart
    stage(buf, kt + 2);
    mfmas(f1);
    pin_reads_dma();
  }
  // ---- the last two tiles: nothing left to fetch; the per-channel epilogue constants are fetched under their MFMAs
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int n = CH * n_tile + 32 * wn + mi;
  const unsigned int rowbytes = (unsigned int)N * 4u;               // X2 rows and float rows have the same size
  const __amdgpu_buffer_rsrc_t resB = ehm_buffer_rsrc((const char*)Res + m0 * rowbytes);
  const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc((const char*)Y + m0 * rowbytes);
  const __amdgpu_buffer_rsrc_t dsB = ehm_buffer_rsrc(L.Ds);
  const __amdgpu_buffer_rsrc_t m1B = ehm_buffer_rsrc(L.M1s);
  float dj[kJ], mj[kJ], sh;
  const unsigned int n4 = (unsigned int)n * 4u;
  {
    const int buf = (KT - 2) & 1;
    read_frags(f1, buf, 1);
    mfmas(f0);
    pin_reads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(f0, buf ^ 1, 0);
    mfmas(f1);
    pin_reads();
    read_frags(f1, buf ^ 1, 1);
    mfmas(f0);
    pin_reads();
    __builtin_amdgcn_sched_barrier(0);
    sh = L.shift[n];                                   // fragment set 0 is dead: room for the two tables
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      dj[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsB, n4, j * rowbytes, 0));
      mj[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(m1B, n4, j * rowbytes, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1);
  }

  STAMP(2); STAMPC(6);
  // ---- epilogue (Ds/M1s carry 1/w_scale), in two parts around an LDS transposition:
  //   P1  per lane (one channel, 24 joints of two bodies in adjacent registers): modulation / BatchNorm fold, 24x24 adjacency mix with the coefficients in SGPRs
  //       (plain v_fmac: v_pk_fma_f32 measured 45 % slower here), ReLU, ds_write_b32 into a float [192][64] tile;
  //   P2  per thread 6 x (one row, 8 consecutive channels): residual add, hi/lo split, 16-byte stores.
  // Measured with s_memrealtime stamps: the one-dword-per-lane epilogue (96 loads + 96 stores of 4 B per lane) took 25k cycles of a
  // 110k-cycle block; 16-byte accesses cut its memory instruction count 8x.  The residual is fetched under the mix.
  // Addresses = block-uniform buffer descriptor + 32-bit lane offset: a 192-row tile spans < 2 GiB.
  // P2 work item: rows urow + 32 i (i < 6), channels 64*n_tile + 8*uc .. +7
  const int urow = tid / SH::UPR, uc = tid % SH::UPR;
  const int uc0 = CH * n_tile + 8 * uc;                                                    // first channel of the work item
  const unsigned int ucolx = (unsigned int)(uc0 >> 5) * 128u + (unsigned int)(uc0 & 31) * 2u;   // X2: 8 hi halves here, 8 lo halves 64 B on
  const unsigned int ucolf = (unsigned int)uc0 * 4u;
  // The mix's coefficient fragments are requested NOW, behind the last MFMA: an epilogue load queues behind the DMA stream of the
  // CU's other block and takes 1-3 us to come back (stamps).  (Requesting the residual here as well spills ~25 registers: slower.)
#if EHM_MIX_MFMA
  const half8* AF = (const half8*)L.AoffF + lane;
  const float invS = ((const float*)((const half8*)L.AoffF + 6 * 64))[0];
  half8 a_hi[3], a_lo[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) { a_hi[s3] = AF[(2 * s3) * 64]; a_lo[s3] = AF[(2 * s3 + 1) * 64]; }
#endif
  __builtin_amdgcn_sched_barrier(0);
  f32x2 dp[kJ], gp[kJ];
#pragma unroll
  for (int j = 0; j < kJ; ++j) {   // fold modulation / BatchNorm scale
    const f32x2 a0 = f32x2{acc0[j >> 3][2 * (j & 7)], acc0[j >> 3][2 * (j & 7) + 1]};
    const f32x2 a1 = f32x2{acc1[j >> 3][2 * (j & 7)], acc1[j >> 3][2 * (j & 7) + 1]};
    dp[j] = __builtin_elementwise_fma(f32x2{dj[j], dj[j]}, a0, f32x2{sh, sh});
    gp[j] = a1 * f32x2{mj[j], mj[j]};
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0);
  // transposition tile: element (row, ch) at float row*64 + (((ch>>2) ^ ((row>>1)&1)) << 2) + (ch&3).  The XOR keeps the
  // ds_read_b128 of P2 conflict-free (lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} cover 4 rows x 4 of the 8 chunks).
  STAMP(8);
  __syncthreads();                                                   // every wave is done reading operand fragments
  STAMP(9);
  float* T = lds;
#if EHM_MIX_MFMA
  {
    // 24x24 adjacency mix on the matrix cores.  Per body: out[j][ch] = sum_k Aoff[j][k] gp[k][ch] + dp[j][ch] = [S Aoff | S I] (24 x 48)
    // times [gp; dp] (48 x 32 channels), split-f16 (3 MFMA per product), K = 48 = three k-steps.  B fragments: a lane owns, for
    // ITS channel, all 24 joints of the two bodies of its half-wave, but the MFMA wants lanes 0-31 to carry k-block 2s and lanes 32-63
    // k-block 2s+1 of ONE body: v_permlane32_swap exchanges the upper half of P (= both lane halves' block 2s) with the lower half
    // of Q (= block 2s+1), leaving P = body of the lower half-wave, Q = body of the upper one.  The VALU version was 1152 v_fmac
    // per wave (~4.4 us per tile, measured); this is 36 MFMA + ~400 VALU.
    const int ch = 32 * wn + mi;
    const bool relu = L.relu != 0;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int beta = 0; beta < 2; ++beta) {
      f32x16 dA, dB;                               // body (half-wave 0, beta) and body (half-wave 1, beta)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dA[r] = 0.f; dB[r] = 0.f; }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        half8 ph, pl, qh, ql;                      // own k-blocks 2 s3 (P) and 2 s3 + 1 (Q), hi / lo
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kp = 16 * s3 + e, kq = kp + 8;   // < 24: gp[k], else dp[k - 24]
          const float vp = kp < kJ ? gp[kp][beta] : dp[kp - kJ][beta];
          const float vq = kq < kJ ? gp[kq][beta] : dp[kq - kJ][beta];
          const float cp = fminf(fmaxf(vp, -65504.f), 65504.f), cq = fminf(fmaxf(vq, -65504.f), 65504.f);
          ph[e] = (half_t)cp; pl[e] = (half_t)(vp - (float)ph[e]);
          qh[e] = (half_t)cq; ql[e] = (half_t)(vq - (float)qh[e]);
        }
        u32x4 Ph = __builtin_bit_cast(u32x4, ph), Pl = __builtin_bit_cast(u32x4, pl);
        u32x4 Qh = __builtin_bit_cast(u32x4, qh), Ql = __builtin_bit_cast(u32x4, ql);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const auto sh = __builtin_amdgcn_permlane32_swap(Ph[w], Qh[w], false, false);
          Ph[w] = sh[0]; Qh[w] = sh[1];
          const auto sl = __builtin_amdgcn_permlane32_swap(Pl[w], Ql[w], false, false);
          Pl[w] = sl[0]; Ql[w] = sl[1];
        }
        const half8 bAh = __builtin_bit_cast(half8, Ph), bAl = __builtin_bit_cast(half8, Pl);
        const half8 bBh = __builtin_bit_cast(half8, Qh), bBl = __builtin_bit_cast(half8, Ql);
        dA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[s3], bAh, dA, 0, 0, 0);
        dB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[s3], bBh, dB, 0, 0, 0);
        dA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[s3], bAl, dA, 0, 0, 0);
        dB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[s3], bBl, dB, 0, 0, 0);
        dA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[s3], bAh, dA, 0, 0, 0);
        dB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[s3], bBh, dB, 0, 0, 0);
      }
      // D: column = my channel, rows = joints (r&3) + 8 (r>>2) + 4 g (g = lane half; r >> 2 == 3 is padding)
      float* ta = T + (96 * wm + 24 * beta) * CH;        // body of half-wave 0;   + 48 rows: body of half-wave 1
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        const int j0 = (r & 3) + 8 * (r >> 2);             // + 4 g
        const int sw = (j0 >> 1) & 1;                       // (row >> 1) & 1 of the transposition tile's swizzle: 4 g, 24 beta, 48, 96 wm leave it
        float va = dA[r] * invS, vb = dB[r] * invS;
        if (relu) { va = fmaxf(va, 0.f); vb = fmaxf(vb, 0.f); }
        float* t = ta + (j0 + 4 * g) * CH + (((ch >> 2) ^ sw) << 2) + (ch & 3);
        t[0] = va;
        t[48 * CH] = vb;
      }
    }
  }
#else
  {
    const int ch = 32 * wn + mi;
    const int lrow = 96 * wm + 48 * g;                               // body a = rows lrow.., body b = lrow + 24..; (row>>1)&1 == (j>>1)&1
    float* t0 = T + lrow * CH + (((ch >> 2) ^ 0) << 2) + (ch & 3);
    float* t1 = T + lrow * CH + (((ch >> 2) ^ 1) << 2) + (ch & 3);
    gcn_mix2(dp, gp, L.Aoff, L.relu != 0, [&](int j, float s0, float s1) {
      float* t = ((j >> 1) & 1) ? t1 : t0;
      t[j * CH] = s0;
      t[(24 + j) * CH] = s1;
    });
  }
#endif
  __builtin_amdgcn_sched_barrier(0);     // the mix needs the registers; the residual travels under the LDS writes and the barrier
  u32x4 rh[6], rl[6];
  if (RES) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      rh[i] = __builtin_amdgcn_raw_buffer_load_b128(resB, (urow + 32 * i) * rowbytes + ucolx, 0, IN_AUX);
      rl[i] = __builtin_amdgcn_raw_buffer_load_b128(resB, (urow + 32 * i) * rowbytes + ucolx + 64u, 0, IN_AUX);
    }
  }
  STAMP(10);
  __syncthreads();
  STAMP(11);
  {
    const int p = (urow >> 1) & 1;
    const float* src = T + urow * CH;
    const int o0 = ((2 * uc) ^ p) << 2, o1 = ((2 * uc + 1) ^ p) << 2;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      // (wide tile: 16 work items per row span two 256-byte bank windows; items 8..15 read their halves in the opposite order so
      //  that a ds_read_b128 lane group still touches 16 distinct 16-byte slots)
      const bool flip = (CH == 128) && (uc & 8);
      const f32x4 va = *(const f32x4*)(src + 32 * i * CH + (flip ? o1 : o0));
      const f32x4 vb = *(const f32x4*)(src + 32 * i * CH + (flip ? o0 : o1));
      const f32x4 v0 = flip ? vb : va, v1 = flip ? va : vb;
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (RES) {
        const half8 h = __builtin_bit_cast(half8, rh[i]), l = __builtin_bit_cast(half8, rl[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)h[k] + (float)l[k];
      }
      const unsigned int rowoff = (unsigned int)(urow + 32 * i) * rowbytes;
      if (OUT_SPLIT) {
        half8 h, l;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float c = fminf(fmaxf(v[k], -65504.f), 65504.f);
          h[k] = (half_t)c;
          l[k] = (half_t)(v[k] - (float)h[k]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), yB, rowoff + ucolx, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l), yB, rowoff + ucolx + 64u, 0, AUX);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), yB, rowoff + ucolf, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), yB, rowoff + ucolf + 16u, 0, AUX);
      }
    }
  }
  STAMP(3); STAMPC(7);
}

template <int PASSES, bool RES, bool OUT_SPLIT, int NWN>
__global__ __launch_bounds__(128 * NWN, 2) void gcn_hidden_f16r_kernel(const half_t* __restrict__ X, LayerDev L,
                                                                        const half_t* __restrict__ Res, float* __restrict__ Y,
                                                                        int m_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Shape<NWN>::STG];   // 80 KiB (112 KiB wide), the only LDS object
  const int n_tiles = L.N / Shape<NWN>::CH;
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  const int lin = ((total & 7) == 0) ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;   // XCD-aware tile order
  f16r_tile<PASSES, kStoreAux, 0, NWN>(lds, X, L, Res, Y, lin / n_tiles, lin % n_tiles, RES, OUT_SPLIT, [] {});
}

// ---------------------------------------------------------------------------------------------------------------------------
// All hidden convs of one GCN forward in ONE launch (ehm_gcn_hidden_stack).  Work item = (layer, row tile, channel tile); a
// conv's tile needs all 16 channel tiles of the previous conv for the SAME 192 rows and nothing else, so the convs are chained
// per row tile with counters instead of kernel boundaries:
//   * every XCD owns the row tiles m = xcc (mod 8) for ALL layers (the block reads its own XCC_ID), so producer and consumer of
//     a row tile share one L2; each XCD has a ticket counter handing out its items in (layer, m, n) order - a consumer's
//     producers always hold smaller tickets, so waiting cannot deadlock whatever the residency;
//   * publish = sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> barrier -> one relaxed agent-scope add on
//     done[layer][m]; consume = one lane polls relaxed until done[layer-1][m] == n_tiles, barrier, agent acquire fence (drops
//     the CU's stale L1 lines; or, default, agent-scope sc1 loads for the activations, which never hit in L1), then the tile's
//     activation loads (cdna_hip_programming.md, counter hand-off recipe).  The weight DMA of the first two K tiles and the
//     next ticket are issued before the poll, so its round trip is covered.  No work stealing across queues: a stolen tile
//     would put producer and consumer on different L2s.  nq = number of XCDs of the device (CUs / 32).
// What it buys: no launch ramp / two-round tail / inter-kernel gap per conv, and the blocks drift apart so one block's
// VALU-bound epilogue overlaps its CU neighbour's MFMA loop.
struct ChainArgs {
  const LayerDev* layers;   // device array [nl]
  half_t* buf[3];           // activation buffers (X2<32>): conv 2b reads cur -> writes buf[1]; conv 2b+1 reads buf[1] (+ residual cur) -> nxt
  int nl, m_tiles, n_tiles;
  unsigned int* tickets;    // [8]
  unsigned int* done;       // [nl][m_tiles]
  unsigned int* err;        // set when a wait timed out (results invalid)
  int flags;                // bit 0 = agent acquire fence after the wait (needed only when the loads are not sc1)
  int nq;                   // queues = XCDs
  int stagger_ticks;        // experiment (EHM_CHAIN_STAGGER, 100 MHz ticks): delay of the upper half of the grid at start
};

template <int PASSES, int AUX, int IN_AUX, int NWN>
__global__ __launch_bounds__(128 * NWN, 2) void gcn_hidden_chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Shape<NWN>::STG];   // 80 KiB (112 KiB wide), the only LDS object
  const int tid = threadIdx.x;
  const unsigned int q = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % (unsigned int)a.nq;   // HW_REG_XCC_ID[3:0] -> my queue
  const int cm = (a.m_tiles - (int)q + a.nq - 1) / a.nq;   // row tiles of this queue: q, q + nq, ...
  if (cm <= 0) return;
  const unsigned int ipl = (unsigned int)(cm * a.n_tiles), total = ipl * (unsigned int)a.nl;
  volatile unsigned int* slot = (volatile unsigned int*)lds;
  if (a.stagger_ticks > 0 && blockIdx.x >= gridDim.x / 2) {   // experiment: start the second block of every CU half a tile late
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.stagger_ticks) __builtin_amdgcn_s_sleep(32);
  }
  if (tid == 0) slot[0] = __hip_atomic_fetch_add(&a.tickets[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned int t = __builtin_amdgcn_readfirstlane(slot[0]);
  while (t < total) {
    __syncthreads();                                     // everybody has the ticket before this tile's DMA overwrites the slot
    unsigned int t_next = 0;                             // the next ticket travels under this tile
    if (tid == 0) t_next = __hip_atomic_fetch_add(&a.tickets[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int layer = (int)(t / ipl), r = (int)(t % ipl);
    const int m_tile = (int)q + a.nq * (r / a.n_tiles), n_tile = r % a.n_tiles;
    const int blk = layer >> 1, cur = (blk & 1) ? 2 : 0, nxt = (blk & 1) ? 0 : 2;
    const bool odd = layer & 1;
    const half_t* X = odd ? a.buf[1] : a.buf[cur];
    const half_t* Res = odd ? a.buf[cur] : nullptr;
    float* Y = (float*)(odd ? a.buf[nxt] : a.buf[1]);
    f16r_tile<PASSES, AUX, IN_AUX, NWN>(lds, X, a.layers[layer], Res, Y, m_tile, n_tile, odd, layer != a.nl - 1, [&] {
      if (layer == 0) return;                            // (block-uniform)
      if (tid == 0) {
        const unsigned int* f = a.done + (size_t)(layer - 1) * a.m_tiles + m_tile;
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)a.n_tiles) {
          __builtin_amdgcn_s_sleep(4);
          ++spins;                                       // never hang the device: give up after ~1 s (or at once when
          if (spins > (1 << 22) || ((spins & 255) == 0 &&  // somebody else already has) and flag the launch as failed
                                    __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
#ifdef EHM_STAMPS
        if (spins) { __hip_atomic_fetch_add(a.err + 1, (unsigned int)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(a.err + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
      }
      __syncthreads();
      if (a.flags & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores have reached L2 / memory
    __syncthreads();                                     // ... everybody's have, and nobody reads LDS any more
    if (tid == 0) {
      __hip_atomic_fetch_add(&a.done[(size_t)layer * a.m_tiles + m_tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      slot[0] = t_next;
    }
    __syncthreads();
    t = __builtin_amdgcn_readfirstlane(slot[0]);
  }
}

template <int PASSES, int NWN>
int launch(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_split, hipStream_t st) {
  const int m_tiles = (int)(rows_pad / 192);
  const int blocks = m_tiles * (h->hid / Shape<NWN>::CH);
  const LayerDev& L = h->hidden[layer];
  const half_t* x = (const half_t*)X;
  const half_t* r = (const half_t*)residual;
  float* y = (float*)out;
  if (residual) {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16r_kernel<PASSES, true, true, NWN>), dim3(blocks), dim3(128 * NWN), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16r_kernel<PASSES, true, false, NWN>), dim3(blocks), dim3(128 * NWN), 0, st, x, L, r, y, m_tiles);
  } else {
    if (out_split) hipLaunchKernelGGL((gcn_hidden_f16r_kernel<PASSES, false, true, NWN>), dim3(blocks), dim3(128 * NWN), 0, st, x, L, r, y, m_tiles);
    else hipLaunchKernelGGL((gcn_hidden_f16r_kernel<PASSES, false, false, NWN>), dim3(blocks), dim3(128 * NWN), 0, st, x, L, r, y, m_tiles);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef EHM_STAMPS
extern "C" int ehm_dbg_chain_stats(ehm_gcn* h, unsigned int* out3) {   // err, total spins, waits that had to spin (last chained launch)
  if (!h || !h->chain_sync) return EHM_EINVAL;
  EHM_HIP(hipDeviceSynchronize());
  EHM_HIP(hipMemcpy(out3, h->chain_sync + h->chain_err_off, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int ehm_dbg_set(void* p) { unsigned long long* q = (unsigned long long*)p; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &q, sizeof(q)); }
#endif

int ehm_gcn_hidden_chain_impl(ehm_gcn* h, void* const bufs[3], int64_t rows_pad, hipStream_t st) {
  const int nl = h->num_hidden;
  if (h->hid % 64 != 0 || nl < 2 || (nl & 1) || h->hidden[0].K % RK != 0 || h->hidden[0].K / RK < 2 || rows_pad % 192 != 0) {
    ehm_set_error("chained split-f16 convs need hid %% 64 == 0, K >= 64, an even number of hidden convs and rows_pad %% 192 == 0");
    return EHM_EINVAL;
  }
  const bool wide = h->wide_tile && h->hid % 128 == 0;            // 8-wave 192 x 128 blocks, one per CU
  const int m_tiles = (int)(rows_pad / 192), n_tiles = h->hid / (wide ? 128 : 64);
  const size_t need = 8 + (size_t)nl * m_tiles + 8;   // tickets | done | err, (stamps build: total spins, waits that spun)
  if (h->chain_sync_words < need) {                    // grows on the first call of a new shape; never inside a steady loop
    if (h->chain_sync) EHM_HIP(hipFree(h->chain_sync));
    h->chain_sync = nullptr;
    h->chain_sync_words = 0;
    EHM_HIP(hipMalloc(&h->chain_sync, need * sizeof(unsigned int)));
    h->chain_sync_words = need;
  }
  EHM_HIP(hipMemsetAsync(h->chain_sync, 0, need * sizeof(unsigned int), st));
  ChainArgs a;
  a.layers = h->hidden_dev;
  for (int i = 0; i < 3; ++i) a.buf[i] = (half_t*)bufs[i];
  a.nl = nl;
  a.m_tiles = m_tiles;
  a.n_tiles = n_tiles;
  a.tickets = h->chain_sync;
  a.done = h->chain_sync + 8;
  h->chain_err_off = 8 + (size_t)nl * m_tiles;
  a.err = h->chain_sync + h->chain_err_off;
  const int total = nl * m_tiles * n_tiles;
  int blocks = (wide ? 1 : 2) * ehm_num_cus();         // what is co-resident (80 KiB LDS per block, 112 KiB wide)
  if (blocks > total) blocks = total;
  a.nq = ehm_num_cus() / 32;
  if (a.nq < 1) a.nq = 1;
  if (a.nq > 8) a.nq = 8;
  const int mode = getenv("EHM_CHAIN_MODE") ? atoi(getenv("EHM_CHAIN_MODE")) : 0;   // 0 = sc1 loads (default), 1 = plain loads + acquire fence, 2 = neither (timing experiment only)
  a.flags = mode == 1 ? 1 : 0;
  a.stagger_ticks = getenv("EHM_CHAIN_STAGGER") ? atoi(getenv("EHM_CHAIN_STAGGER")) : 0;
  if (h->precision == EHM_PREC_F16X3) {
    if (wide) hipLaunchKernelGGL((gcn_hidden_chain_kernel<3, 16, 16, 4>), dim3(blocks), dim3(512), 0, st, a);
    else if (mode == 0) hipLaunchKernelGGL((gcn_hidden_chain_kernel<3, 16, 16, 2>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gcn_hidden_chain_kernel<3, 16, 0, 2>), dim3(blocks), dim3(256), 0, st, a);
  } else {
    if (wide) hipLaunchKernelGGL((gcn_hidden_chain_kernel<1, 16, 16, 4>), dim3(blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gcn_hidden_chain_kernel<1, 16, 16, 2>), dim3(blocks), dim3(256), 0, st, a);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

int ehm_gcn_chain_error(const ehm_gcn* h, hipStream_t st, unsigned int* flag) {   // debugging aid: did a wait time out in the last chain launch?
  *flag = 0;
  if (!h->chain_sync) return 0;
  EHM_HIP(hipMemcpyAsync(flag, h->chain_sync + h->chain_err_off, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  EHM_HIP(hipStreamSynchronize(st));
  return 0;
}

int ehm_gcn_hidden_f16r_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad,
                             bool out_split, hipStream_t st) {
  if (h->hid % 64 != 0 || h->hidden[layer].K % RK != 0 || h->hidden[layer].K / RK < 2) {
    ehm_set_error("register-pipelined split-f16 conv needs hid %% 64 == 0 and K >= 64");
    return EHM_EINVAL;
  }
  if (h->wide_tile && h->hid % 128 == 0) {
    if (h->precision == EHM_PREC_F16X3) return launch<3, 4>(h, layer, X, residual, out, rows_pad, out_split, st);
    return launch<1, 4>(h, layer, X, residual, out, rows_pad, out_split, st);
  }
  if (h->precision == EHM_PREC_F16X3) return launch<3, 2>(h, layer, X, residual, out, rows_pad, out_split, st);
  return launch<1, 2>(h, layer, X, residual, out, rows_pad, out_split, st);
}
