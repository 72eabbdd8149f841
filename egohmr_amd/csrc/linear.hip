// Fused linear layer on the f16 matrix cores with f32-grade accuracy (split-f16 "f16x3", see gcn_tile.hip):
//     Y = act_out( [act_in(A0) | A1] . W^T * (1/w_scale) + bias + group_bias[row / rows_per_group] )   (+ column max per group)
// Used for the scene PointNet of the conditioning path (models/respointnet.py:33-97: ResnetPointnet / ResnetBlockFC), where
// the reference runs ~40 eager torch kernels over [B,N,256..1024] float32 tensors per call.  Restructuring used by the host
// (egohmr_amd/encoders.py), all exact up to float re-association:
//   * the pooled half of every block input (`cat([net, pooled])`, respointnet.py:41-51) is constant per body, so its
//     contribution to fc_0 and to the shortcut is a per-body bias vector (group_bias) and the per-point GEMMs shrink to 256 wide;
//   * shortcut and fc_1 of a block accumulate into the same output, so they are ONE GEMM over the concatenated K
//     ([relu(h) | net] . [W1 | S]^T): the loader switches source pointer at K0 (dual-source A operand);
//   * the global max-pool over the N points is fused into the epilogue (wave shuffle -> LDS -> one atomic per column and block).
// Operands are in the X2 split format of gcn_dev.h (32 hi halves + 32 lo halves per 32-k group).  Tile 192 x 128 x 32,
// 4 waves (2 x 2, 96 x 64 each), persistent blocks, see linear_tile_kernel.
#include <type_traits>

#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int LBM = 192, LBN = 128;       // output tile; 4 waves as 2 x 2, 96 x 64 per wave (6 x 4 MFMA blocks of 16 x 16; hi-only tier: 3 x 2 of 32 x 32)
constexpr int RK = BK;                    // floats per operand row and K tile (X2: 32 hi halves | 32 lo halves = 128 bytes)
constexpr int LA_T = LBM * RK, LB_T = LBN * RK, LSTG = LA_T + LB_T;   // floats; one stage = 40 KiB

struct LinArgs {
  const float* A0; const float* A1; const float* W;    // X2 rows addressed as floats (K floats per row)
  const float* bias; const float* gbias;
  char* Y; float* colmax;
  const float* pts; const float* W4;                    // LIFT: A0 is generated, see linear_tile_kernel
  int K0, K1, M, N;
  int rows_per_group, valid_rows_per_group;
  int relu_in0, relu_out;
  float inv_scale;
  int gstride;                                          // floats between the rows of gbias
};

struct LFrags {
  half8 ah[3], al[3], bh[2], bl[2];
};

// ReLU of a split value: hi' = max(hi, 0), lo' = hi > 0 ? lo : 0 - on packed halves, 4 VALU per two elements (written as
// instructions: from C the u16 minimum came back as 800 v_cmp + v_cndmask and their lane masks spilled 480 SGPRs)
__device__ __forceinline__ void relu_split(half8& hi, half8& lo) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned int r, k;
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(h[e]));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(k) : "v"(r), "s"(0x00010001u));     // 1 where hi' != 0 (hi' >= 0: its bits order like integers)
    asm("v_pk_sub_u16 %0, 0, %1" : "=v"(k) : "v"(k));                         // 0xffff there, 0 elsewhere
    h[e] = r;
    l[e] &= k;
  }
  hi = __builtin_bit_cast(half8, h);
  lo = __builtin_bit_cast(half8, l);
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// Persistent blocks (2 per CU), software-pipelined across tiles like the GCN tile engine (gcn_tile.hip): K loop with register
// double-buffered fragments and one barrier per K tile; after a tile's last barrier the NEXT tile's operand DMA goes out and only
// then the epilogue runs, per wave and without a block barrier - each wave turns its 96 x 64 accumulators through the six 1 KiB
// pieces of stage 1's activation region that its own DMA instructions fill (wave-private scratch) so that the X2 rows leave as
// 16-byte stores.  With K = 256 .. 544 a tile has only 8 - 17 K tiles: in the one-tile-per-block kernel this replaces, prologue
// and epilogue were more than half of a block's life (435 TFLOP/s issued on the PointNet at B = 256 x 4096 points).
// Tile order: block b sits on XCD b % 8; the N / 128 column tiles of one row tile go to neighbouring blocks of ONE XCD in the same
// iteration, so the row tile's activations come from HBM once.
//
// LIFT = the first layer of the PointNet folded into the loader (respointnet.py:35,:90: net = fc_pos(p); fc_0(actvn(net))): the
// operand A0 = relu(points . Wpos^T + bpos) [M, K0] is a function of 12 bytes per row, so instead of writing it to HBM in X2 format
// (2.3 GB at 256 x 4096 points, the former pointnet_lift kernel: 0.9 ms) and reading it back, every K tile of it is produced in
// place: wave w evaluates k-chunk w (8 channels, weights and bias wave-uniform -> scalar loads, operands of the FMAs) for the
// tile's 192 rows (three per lane, their points stay in nine registers for the whole tile), splits the values exactly like
// split_store and writes the hi / lo chunks at the loader's swizzled positions.  Stage 1 is written behind the head barrier of the
// tile (its rows are other waves' epilogue scratch until then); its values are visible after the first K tile's barrier.
// HO ("hi only", ehm_linear_desc.hi_only): the plain-f16 tier (NOT parity grade, see conv.hip): only hi halves are fetched, multiplied and written.
template <bool RELU_A, bool LIFT, bool HO = false>
__global__ __launch_bounds__(256, 2) void linear_tile_kernel(LinArgs p) {
  static_assert(!(RELU_A && LIFT), "the generated operand is already rectified");
  __shared__ __attribute__((aligned(16))) float lds[2 * LSTG];   // 80 KiB; the ONLY LDS object
  // MODE.FP16_OVFL = 1 for the life of the wave: every f32 -> f16 conversion of the epilogue clamps to +-65504 instead of producing inf - the
  // same results on finite values as the explicit clamps it replaces (4 of the ~8 vector-ALU instructions per output value: v_med3 + its
  // canonicalising v_max, twice), as in the GCN tile engine (gcn_tile.hip)
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);

  constexpr int KS = 2, NM = 18, NR = 10;
  // split-f16 products on v_mfma_f32_16x16x32_f16 (a K tile = ONE k-step; the K loop of gcn_tile.hip's split mode: four (row half, column half) phases in
  // snake order, no operand half double-buffered); the hi-only tier keeps the 32 x 32 x 16 form
  constexpr bool M16 = !HO;
  const int tid = threadIdx.x;
  const int K = p.K0 + p.K1;
  const int KT0 = p.K0 / RK, KT = K / RK;
  const int n_tiles = p.N / LBN, m_tiles = p.M / LBM;

  int lane, wave, wm, wn, mi, g, r0, swz;
  bool hi_lane;                                                // my 16-byte chunk of an operand piece holds hi halves
  int oA[KS][2], oB[KS][2];
  [[maybe_unused]] int oA16[2], oB16[2];
  auto thread_consts = [&]() {                                 // re-derived per tile: nothing of this stays live across the epilogue
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63;
    wave = __builtin_amdgcn_readfirstlane(t >> 6);
    wm = wave >> 1; wn = wave & 1;
    mi = lane & 31; g = lane >> 5;
    r0 = 8 * wave + (lane >> 3);                               // DMA: one wave instruction = 8 rows x 128 B, rows r0 + 32 i share a key
    swz = ((lane & 7) ^ ((r0 >> 1) & 7)) << 2;
    hi_lane = swz < 16;
    const int rA = 96 * wm + mi, rB = 64 * wn + mi;            // (+ 32 t / + 32 u leave the swizzle key alone)
    const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const int c = 4 * hl + 2 * s + g;                      // logical 16-byte chunk: [hi k0-31 | lo k0-31]
        oA[s][hl] = rA * RK + ((c ^ keyA) << 2);
        oB[s][hl] = LA_T + rB * RK + ((c ^ keyB) << 2);
      }
    if constexpr (M16) {      // lane (i = l & 15, kg = l >> 4): row i of a 16-row tile, logical chunk kg (hi) / 4 + kg (lo) of the 128-byte K tile
      const int i16 = lane & 15, kg = lane >> 4, key = (i16 >> 1) & 7;
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        oA16[hl] = (96 * wm + i16) * RK + (((4 * hl + kg) ^ key) << 2);
        oB16[hl] = LA_T + (64 * wn + i16) * RK + (((4 * hl + kg) ^ key) << 2);
      }
    }
  };
  thread_consts();

  // ---- tiles of this block: iteration it -> (row tile, column tile)
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

  // operand pieces in buffer form (as in gcn_tile.hip): the tile's base in an SGPR descriptor, the lane's row / swizzled chunk in one 32-bit
  // offset per operand, the piece in the scalar offset - no vector-ALU address arithmetic per piece
  __amdgpu_buffer_rsrc_t rsA0, rsA1, rsB;
  int voA0, voA1, voB;
  const int a0row32 = 32 * p.K0, a1row32 = 32 * p.K1, brow32 = 32 * K;
  auto set_tile_ptrs = [&](int m, int n) {
    rsA0 = ehm_buffer_rsrc(p.A0 + (size_t)m * LBM * p.K0);
    rsA1 = ehm_buffer_rsrc(p.K1 ? p.A1 + (size_t)m * LBM * p.K1 : p.A0);
    rsB = ehm_buffer_rsrc(p.W + (size_t)n * LBN * K);
    voA0 = (r0 * p.K0 + swz) * 4;
    voA1 = (r0 * p.K1 + swz) * 4;
    voB = (r0 * K + swz) * 4;
  };
  auto dma_a = [&](int buf, int kt, int i) {
    if (HO && !hi_lane) return;                                // (half of the lanes of every piece; the instruction is still issued: wait counts stand)
    AS3 void* dst = (AS3 void*)(lds + buf * LSTG + (wave + 4 * i) * 256);
    if (kt < KT0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, dst, 16, voA0, (i * a0row32 + kt * RK) * 4, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, dst, 16, voA1, (i * a1row32 + (kt - KT0) * RK) * 4, 0, 0);
  };
  auto dma_b = [&](int buf, int kt, int i) {
    if (HO && !hi_lane) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (AS3 void*)(lds + buf * LSTG + LA_T + (wave + 4 * i) * 256), 16, voB, (i * brow32 + kt * RK) * 4, 0, 0);
  };
  float px[3], py[3], pz[3];                                    // LIFT: the points of rows lane + 64 j of the current tile
  auto load_points = [&](int m) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const size_t row = (size_t)m * LBM + (tid & 63) + 64 * j;
      const int grp = (int)(row / p.rows_per_group), i = (int)(row % p.rows_per_group);
      px[j] = py[j] = pz[j] = 0.f;                               // padding rows: relu(bias), like the unfused kernel; they never reach the maximum
      if (i < p.valid_rows_per_group) {
        const float* q = p.pts + ((size_t)grp * p.valid_rows_per_group + i) * 3;
        px[j] = q[0]; py[j] = q[1]; pz[j] = q[2];
      }
    }
  };
  auto gen_a = [&](int buf, int kt) {
    typedef const f32x4 __attribute__((address_space(4))) cf32x4;
    cf32x4* w4 = (cf32x4*)(uintptr_t)(p.W4 + (size_t)(kt * RK + 8 * wave) * 4);   // (w_x, w_y, w_z, bias) of my eight channels
    f32x4 wk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wk[e] = w4[e];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int row = lane + 64 * j, key = (row >> 1) & 7;
      half8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = fmaxf(fmaf(wk[e][2], pz[j], fmaf(wk[e][1], py[j], fmaf(wk[e][0], px[j], wk[e][3]))), 0.f);
        hi[e] = (half_t)v;                                   // (MODE.FP16_OVFL: saturating conversions)
        lo[e] = (half_t)(v - (float)hi[e]);
      }
      float* dst = lds + buf * LSTG + row * RK;
      *(half8*)(dst + ((wave ^ key) << 2)) = hi;                  // logical chunk 2 s + g = wave: k = 8 wave .. + 7
      if constexpr (!HO) *(half8*)(dst + (((wave + 4) ^ key) << 2)) = lo;
    }
  };
  auto stage = [&](int buf, int kt) {
    if constexpr (LIFT) {
      gen_a(buf, kt);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) dma_a(buf, kt, i);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_b(buf, kt, i);
  };

  auto read_frags = [&](LFrags& f, int buf, int s) {
    const float* S = lds + buf * LSTG;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f.ah[t] = *(const half8*)(S + oA[s][0] + 32 * t * RK);
      if constexpr (!HO) f.al[t] = *(const half8*)(S + oA[s][1] + 32 * t * RK);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f.bh[u] = *(const half8*)(S + oB[s][0] + 32 * u * RK);
      if constexpr (!HO) f.bl[u] = *(const half8*)(S + oB[s][1] + 32 * u * RK);
    }
    if constexpr (RELU_A) {                                       // (the ReLU'd operand is the only K segment: K1 == 0, checked by the host)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if constexpr (HO) {
          typedef _Float16 half2_r __attribute__((ext_vector_type(2)));
          typedef unsigned int u32x4_r __attribute__((ext_vector_type(4)));
          u32x4_r h = __builtin_bit_cast(u32x4_r, f.ah[t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned int he = h[e];
            const half2_r m = __builtin_elementwise_max(__builtin_bit_cast(half2_r, he), half2_r{(_Float16)0.f, (_Float16)0.f});
            h[e] = __builtin_bit_cast(unsigned int, m);
          }
          f.ah[t] = __builtin_bit_cast(half8, h);
        } else {
          relu_split(f.ah[t], f.al[t]);
        }
      }
    }
  };
  f32x16 acc[3][2];
  auto mfmas = [&](const LFrags& f) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {                             // small cross terms first, leading term last
        if constexpr (!HO) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[u], acc[t][u], 0, 0, 0);
        }
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[u], acc[t][u], 0, 0, 0);
      }
  };
  // sched_group_barrier masks: 0x008 MFMA, 0x100 DS read, 0x010 VMEM
  auto pin_reads = [&]() {
    if constexpr (HO) return;                                   // (the hi-only instruction mix is left to the scheduler)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  };
  auto pin_reads_dma = [&]() {                                  // 10 x (MFMA, read), then the ten DMAs behind the last 8 MFMAs
    if constexpr (HO) return;
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
    for (int i = 0; i < (LIFT ? 0 : 6); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
  };

  int it = 0, m, n;
  if (!tile_of(0, m, n)) return;
  set_tile_ptrs(m, n);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_b(0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_b(1, 1, i);
  if constexpr (LIFT) {
    load_points(m);
    gen_a(0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_a(0, 0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_a(1, 1, i);
  }

  while (true) {
    // ---- head: stage 1's six activation pieces are the last memory instructions this wave issued; everything older (stage 0,
    //      the previous tile's stores) must be complete
    if constexpr (LIFT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    thread_consts();
    if constexpr (LIFT) gen_a(1, 1);                   // every wave is past its epilogue: stage 1's rows are free
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    LFrags f0, f1;
    if constexpr (!M16) read_frags(f0, 0, 0);

    // ---- 16 x 16 x 32: operand halves A[rh] (row tiles 3 rh .. + 2 of 16 rows), B[ch] (column tiles 2 ch, 2 ch + 1 of 16), 6 x 4 accumulators
    [[maybe_unused]] half8 Ah[2][3], Al[2][3], Bh[2][2], Bl[2][2];
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    [[maybe_unused]] f32x4a c16[6][4];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    [[maybe_unused]] auto ldA = [&](auto rhc, int buf) __attribute__((always_inline)) {
      constexpr int rh = decltype(rhc)::value;
      const float* S = lds + buf * LSTG;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        Ah[rh][t] = *(const half8*)(S + oA16[0] + 16 * (3 * rh + t) * RK);
        Al[rh][t] = *(const half8*)(S + oA16[1] + 16 * (3 * rh + t) * RK);
        if constexpr (RELU_A) relu_split(Ah[rh][t], Al[rh][t]);
      }
    };
    [[maybe_unused]] auto ldB = [&](auto chc, int buf) __attribute__((always_inline)) {
      constexpr int ch = decltype(chc)::value;
      const float* S = lds + buf * LSTG;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        Bh[ch][u] = *(const half8*)(S + oB16[0] + 16 * (2 * ch + u) * RK);
        Bl[ch][u] = *(const half8*)(S + oB16[1] + 16 * (2 * ch + u) * RK);
      }
    };
    [[maybe_unused]] auto mm = [&](auto rhc, auto chc) __attribute__((always_inline)) {            // 18 MFMAs: small cross terms first, six independent accumulators per term
      constexpr int rh = decltype(rhc)::value, ch = decltype(chc)::value;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) c16[3 * rh + t][2 * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[rh][t], Bh[ch][u], c16[3 * rh + t][2 * ch + u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) c16[3 * rh + t][2 * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[rh][t], Bl[ch][u], c16[3 * rh + t][2 * ch + u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) c16[3 * rh + t][2 * ch + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[rh][t], Bh[ch][u], c16[3 * rh + t][2 * ch + u], 0, 0, 0);
    };
    [[maybe_unused]] auto pin16 = [&](int reads, int dmas) __attribute__((always_inline)) {          // reads one per MFMA from the start, DMAs one per MFMA behind them
      if constexpr (RELU_A) return;                               // (the rectifying loads are left to the scheduler)
#pragma unroll
      for (int i = 0; i < 18; ++i) {                          // reads behind every second MFMA, DMAs in the gaps (as in gcn_tile.hip)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((i & 1) == 0 && (i >> 1) < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if ((i & 1) == 1 && (i >> 1) < dmas) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
    // one K tile of parity PAR = kt & 1 (stage kt & 1).  Phases (0, cf) (0, cs) | barrier | (1, cs) (1, cf) with cf = PAR: the next tile's first phase is
    // (0, cs) - exactly the halves that are free to be refilled during this tile's last two phases.  Returns false behind the barrier of the LAST K tile
    // (its last two phases run below, under the epilogue's loads); K tile kt + 2 is staged while there is one.
    [[maybe_unused]] auto tile16 = [&](auto parc, int kt) __attribute__((always_inline)) -> bool {
      constexpr int PAR = decltype(parc)::value;
      constexpr int buf = PAR;
      typedef std::integral_constant<int, PAR> CF;
      typedef std::integral_constant<int, 1 - PAR> CS;
      ldB(CS{}, buf);
      mm(I0{}, CF{});
      pin16(4, 0);
      __builtin_amdgcn_sched_barrier(0);
      ldA(I1{}, buf);
      mm(I0{}, CS{});
      pin16(6, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                         // tile kt is in everyone's registers; tile kt + 1 is complete in LDS
      if (kt == KT - 1) return false;
      const bool more = kt + 2 < KT;                           // (block-uniform)
      ldA(I0{}, buf ^ 1);
      __builtin_amdgcn_s_setprio(2);
      if (more) {
        if constexpr (LIFT) {
          gen_a(buf, kt + 2);
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i) dma_a(buf, kt + 2, i);
        }
      }
      mm(I1{}, CS{});
      pin16(6, LIFT ? 0 : 6);
      __builtin_amdgcn_sched_barrier(0);
      ldB(CS{}, buf ^ 1);
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b(buf, kt + 2, i);
      }
      mm(I1{}, CF{});
      pin16(4, 4);
      __builtin_amdgcn_s_setprio(0);
      return true;
    };
    if constexpr (M16) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) c16[t][u] = f32x4a{0.f, 0.f, 0.f, 0.f};
      ldA(I0{}, 0);
      ldB(I0{}, 0);
      for (int kt = 0;; kt += 2) {
        if (!tile16(I0{}, kt)) break;
        if (!tile16(I1{}, kt + 1)) break;
      }                                                           // (the last K tile's second half - phases (1, 0) and (1, 1) in either order - runs below)
    }

    auto first_phase = [&](int buf) {                  // k-step 0 of K tile kt: multiply f0 while f1 fills with k-step 1
      read_frags(f1, buf, 1);
      mfmas(f0);
      pin_reads();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                         // tile kt is in everyone's registers; tile kt + 1 is complete in LDS
    };
    if constexpr (!M16) {
    for (int kt = 0; kt < KT - 2; ++kt) {
      const int buf = kt & 1;
      first_phase(buf);
      read_frags(f0, buf ^ 1, 0);
      __builtin_amdgcn_s_setprio(2);             // the staging phase at raised priority (as in gcn_tile.hip)
      stage(buf, kt + 2);
      mfmas(f1);
      pin_reads_dma();
      __builtin_amdgcn_s_setprio(0);
    }
    {
      const int buf = (KT - 2) & 1;
      first_phase(buf);
      read_frags(f0, buf ^ 1, 0);
      mfmas(f1);
      pin_reads();
    }
    first_phase((KT - 1) & 1);                         // ends with a barrier: every fragment is in registers, all LDS is dead
    }

    // ---- per-column constants, then the next tile's operand DMA, then this tile's epilogue
    const int n0 = n * LBN;
    const size_t row0 = (size_t)m * LBM;
    const int group = (int)(row0 / p.rows_per_group);
    const int lim = p.valid_rows_per_group - (int)(row0 % p.rows_per_group) - 96 * wm;   // rows of this wave that count for the maximum
    float add[M16 ? 4 : 2];                                  // M16: my columns are 64 wn + 16 ct + (lane & 15), ct = 0..3
#pragma unroll
    for (int u = 0; u < (M16 ? 4 : 2); ++u) {
      const int col = M16 ? n0 + 64 * wn + 16 * u + (lane & 15) : n0 + 64 * wn + 32 * u + mi;
      add[u] = p.bias ? p.bias[col] : 0.f;
      if (p.gbias) add[u] += p.gbias[(size_t)group * p.gstride + col];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M16) {
      mm(I1{}, I0{});
      mm(I1{}, I1{});
    } else {
      mfmas(f1);
    }
    int m_next, n_next;
    const bool have_next = tile_of(it + 1, m_next, n_next);
    if (have_next) {
      set_tile_ptrs(m_next, n_next);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_b(0, 0, i);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_b(1, 1, i);
      if constexpr (LIFT) {
        load_points(m_next);
        gen_a(0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_a(0, 0, i);            // stage 1's activation pieces are the epilogue's scratch first
      }
    }

    const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc(p.Y ? p.Y + (row0 + 96 * wm) * (size_t)p.N * 4 : (char*)p.A0);
    const unsigned int yrow = (unsigned int)p.N * 4u;
    const bool relu_out = p.relu_out != 0, has_y = p.Y != nullptr;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    if constexpr (M16) {
      // accumulator layout: lane (i = l & 15, rg = l >> 4) owns columns 64 wn + 16 ct + i; register r of c16[rt][ct] is row 16 rt + 4 rg + r.
      // One pass per row tile (16 rows x 64 columns through four of the wave's six 1 KiB pieces): piece 2 (row >> 3) + (col >> 5) holds [8 rows][32 columns];
      // inside a piece row (row & 7) sits at (row & 7) ^ (row >> 3) and its two 16-column halves swap places for odd rg, so that the four lane groups of a
      // write go to four different 16-bank groups.  Read items = (row, 8 columns): two per lane, rows (l >> 3) and (l >> 3) + 8.
      const int i16 = lane & 15, rg = lane >> 4;
      // write address of (r, ct) = wb[r & 1][ct & 1] + (ct >> 1) 1024 + 32 r:  row-in-piece (4 (rg & 1) + r) ^ (rg >> 1), column (16 (ct & 1) + i) ^ 16 (rg & 1)
      int wb[2][2];
      {
        const int base = LSTG + wave * 256 + (rg >> 1) * 2048 + (rg & 1) * 128 + i16, hoff = 32 * (rg >> 1), ooff = 16 * (rg & 1);
        wb[0][0] = base + hoff + ooff;       wb[0][1] = base + hoff + 16 - ooff;
        wb[1][0] = base - hoff + ooff;       wb[1][1] = base - hoff + 16 - ooff;
      }
      const int rr = lane >> 3, oct = lane & 7;
      const int colw = n0 + 64 * wn + 8 * oct;
      const unsigned int col_off = (unsigned int)(((colw >> 5) * 64 + (colw & 31)) * 2);   // X2: 8 hi halves here, the 8 lo halves 64 B on
      int rbase[2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        rbase[k] = LSTG + wave * 256 + (2 * k + (oct >> 2)) * 1024 + (rr ^ k) * 32 + ((8 * (oct & 3)) ^ (16 * ((lane >> 5) & 1)));
      float cmax[4] = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = fmaf(c16[rt][ct][r], p.inv_scale, add[ct]);
            if (relu_out) v = fmaxf(v, 0.f);
            cmax[ct] = fmaxf(cmax[ct], v);
            if (has_y) lds[wb[r & 1][ct & 1] + (ct >> 1) * 1024 + 32 * r] = v;
          }
        if (has_y) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private scratch: program order is enough
          f32x4 tq[2][2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            tq[k][0] = *(const f32x4*)(lds + rbase[k]);
            tq[k][1] = *(const f32x4*)(lds + rbase[k] + 4);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float v[8] = {tq[k][0][0], tq[k][0][1], tq[k][0][2], tq[k][0][3], tq[k][1][0], tq[k][1][1], tq[k][1][2], tq[k][1][3]};
            half8 hh, ll;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              hh[c] = (half_t)v[c];                              // (MODE.FP16_OVFL: the conversions saturate at +-65504, see the kernel's head)
              ll[c] = (half_t)(v[c] - (float)hh[c]);
            }
            const unsigned int vo = (unsigned int)(16 * rt + 8 * k + rr) * yrow + col_off;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, 0);
          }
        }
      }
      if (p.colmax) {
        if (lim < 96) {                                           // (rare: the group's last row tile) redo the maximum over the valid rows only
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            cmax[ct] = -3.4e38f;
#pragma unroll
            for (int rt = 0; rt < 6; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float v = fmaf(c16[rt][ct][r], p.inv_scale, add[ct]);
                if (relu_out) v = fmaxf(v, 0.f);
                if (16 * rt + 4 * rg + r < lim) cmax[ct] = fmaxf(cmax[ct], v);
              }
          }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          float cm = fmaxf(cmax[ct], __shfl_xor(cmax[ct], 16));
          cm = fmaxf(cm, __shfl_xor(cm, 32));
          if (rg == 0 && lim > 0) atomic_max_float(p.colmax + (size_t)group * p.N + n0 + 64 * wn + 16 * ct + i16, cm);
        }
      }
    } else {
    // accumulator layout: lane (mi, g) owns column 64 wn + 32 u + mi; register r of acc[t][u] is row 32 t + 8 (r >> 2) + 4 g + (r & 3).
    // Row group Gq = 4 t + (r >> 2) = 8 rows; a pass turns three groups (24 rows x 64 columns = the wave's six 1 KiB pieces):
    // piece 2 gi + u holds [8 rows][32 columns] of group 3 pass + gi.
    const int wbase = LSTG + wave * 256 + (4 * g) * 32 + mi;
    const int rr = lane >> 3, oct = lane & 7;                  // my read items: row rr of group gi = item index, columns 8 oct .. + 7 of the wave's 64
    const int rbase = LSTG + wave * 256 + (oct >> 2) * 1024 + rr * 32 + 8 * (oct & 3);
    const int colw = n0 + 64 * wn + 8 * oct;                   // first column of my items
    const unsigned int col_off = (unsigned int)(((colw >> 5) * 64 + (colw & 31)) * 2);   // X2: 8 hi halves here, the 8 lo halves 64 B on
    float cmax[2] = {-3.4e38f, -3.4e38f};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int Gq = 3 * ps + gi, t = Gq >> 2, q = Gq & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = fmaf(acc[t][u][4 * q + j], p.inv_scale, add[u]);
            if (relu_out) v = fmaxf(v, 0.f);
            cmax[u] = fmaxf(cmax[u], v);
            if (has_y) lds[wbase + (2 * gi + u) * 1024 + j * 32] = v;
          }
      }
      if (has_y) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private scratch: program order is enough
        f32x4 tq[3][2];
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
          tq[gi][0] = *(const f32x4*)(lds + rbase + 2048 * gi);
          tq[gi][1] = *(const f32x4*)(lds + rbase + 2048 * gi + 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
          const int Gq = 3 * ps + gi;
          const float v[8] = {tq[gi][0][0], tq[gi][0][1], tq[gi][0][2], tq[gi][0][3], tq[gi][1][0], tq[gi][1][1], tq[gi][1][2], tq[gi][1][3]};
          half8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            hh[c] = (half_t)v[c];                              // (MODE.FP16_OVFL: the conversions saturate at +-65504, see the kernel's head)
            ll[c] = (half_t)(v[c] - (float)hh[c]);
          }
          const unsigned int vo = (unsigned int)(8 * Gq + rr) * yrow + col_off;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, 0);
          if constexpr (!HO) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, 0);
        }
      }
    }
    if (p.colmax) {
      if (lim < 96) {                                           // (rare: the group's last row tile) redo the maximum over the valid rows only
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          cmax[u] = -3.4e38f;
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = fmaf(acc[t][u][r], p.inv_scale, add[u]);
              if (relu_out) v = fmaxf(v, 0.f);
              if (32 * t + 8 * (r >> 2) + 4 * g + (r & 3) < lim) cmax[u] = fmaxf(cmax[u], v);
            }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float cm = fmaxf(cmax[u], __shfl_xor(cmax[u], 32));
        if (g == 0 && lim > 0) atomic_max_float(p.colmax + (size_t)group * p.N + n0 + 64 * wn + 32 * u + mi, cm);
      }
    }
    }
    if (!have_next) break;
    if constexpr (!LIFT) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dma_a(1, 1, i);              // the scratch is free again: stage 1 of the next tile
    }
    m = m_next; n = n_next; ++it;
  }
}

// Y[M,N] = act(X[M,K] . W[K,N] + bias) in exact float32 on the matrix cores (v_mfma_f32_32x32x2_f32) for SHORT M (a batch of feature
// vectors): the conditioning projections of FusedSampler.prepare - the image / scene slices of the input graph conv
// (modulated_gcn_conv.py:39-50 applied to the step-invariant features) and the beta head (egohmr.py:263-265).  The BLAS picked a
// 256 x 256 macro-tile for M = 256, i.e. 4 - 8 workgroups and 170 - 450 us per GEMM.  Here one block owns a 32 x 32 output tile, its
// four waves split K and reduce through LDS (deterministic, no atomics): 512 - 2048 blocks, a few microseconds.
// A fragment: lane (row = l & 31, h = l >> 5) loads X[row][k + 4 h .. + 3]; MFMA i of an 8-k group contracts k + i and k + 4 + i.
// NW waves split K (in units of 8-k groups); NW = 16 for long K: a wave's whole K range is in flight at once, 32 waves per CU hide the latency
// of the (strided, first-touch) weight loads - with 4 waves the 2048-deep projections of FusedSampler.prepare took 120 - 150 us for 15 us of
// matrix time.
template <int NW>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_f32_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                                  float* __restrict__ Y, int M, int K, int N, int relu) {
  __shared__ float red[NW - 1][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mi = lane & 31, h = lane >> 5;
  const int n0 = 32 * blockIdx.x, m0 = 32 * blockIdx.y;
  const int G = K / 8;
  const int k_begin = 8 * (int)((long long)wave * G / NW), kq = 8 * (int)((long long)(wave + 1) * G / NW) - k_begin;
  const int row = m0 + mi;
  const bool row_ok = row < M;
  const float* xa = X + (size_t)(row_ok ? row : 0) * K + k_begin + 4 * h;
  const float* wb = W + (size_t)(k_begin + 4 * h) * N + n0 + mi;
  // `relu`: bit 0 = max(., 0) on the output; relu >> 1 = the number of leading output columns whose INPUT is rectified first (two products of one
  // vector in one launch: [relu(x) . Wa | x . Wb], the per-body vectors of a PointNet block)
  const bool relu_in = n0 < (relu >> 1);                          // (block-uniform)
  relu &= 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8                                                   // 40 loads in flight per wave: the weights stream from HBM once
  for (int k = 0; k < kq; k += 8) {
    f32x4 a = *(const f32x4*)(xa + k);
    if (!row_ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
    if (relu_in) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = fmaxf(a[i], 0.f);
    }
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = wb[(size_t)(k + i) * N];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  }
  // accumulator layout: column = mi, row = (r & 3) + 8 (r >> 2) + 4 h
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][(r & 3) + 8 * (r >> 2) + 4 * h][mi] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    const float add = bias ? bias[n0 + mi] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[r];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) v += red[w][rr][mi];          // (k order: wave 0, 1, 2, ...)
      v += add;
      if (relu) v = fmaxf(v, 0.f);
      if (m0 + rr < M) Y[(size_t)(m0 + rr) * N + n0 + mi] = v;
    }
  }
}

// The raw points zero-padded to 32 columns in X2 format (input of the folded stage-0 shortcut) and, when R0 is given, relu(fc_pos(p))
// in X2 format (the PointNet itself no longer needs it: ehm_linear_desc.lift_points).  32 lanes per row, 8 rows per block.
__global__ __launch_bounds__(256) void pointnet_lift_kernel(const float* __restrict__ pts, const float* __restrict__ Wpos,
                                                            const float* __restrict__ bpos, half_t* __restrict__ R0, half_t* __restrict__ P32,
                                                            int N, int Npad, int C, size_t rows) {
  const size_t row = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);       // b * Npad + i
  if (row >= rows) return;
  const int c0 = threadIdx.x & 31;
  const int b = (int)(row / Npad), i = (int)(row % Npad);
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < N) {
    const float* q = pts + ((size_t)b * N + i) * 3;
    x = q[0]; y = q[1]; z = q[2];
  }
  split_store(P32, row, c0, 32, c0 == 0 ? x : (c0 == 1 ? y : (c0 == 2 ? z : 0.f)));
  if (R0)
    for (int c = c0; c < C; c += 32) {
      const float v = fmaf(Wpos[c * 3 + 2], z, fmaf(Wpos[c * 3 + 1], y, fmaf(Wpos[c * 3], x, bpos[c])));   // torch addmm order is free
      split_store(R0, row, c, C, fmaxf(v, 0.f));
    }
}

__global__ void pack_scaled_kernel(const float* __restrict__ X, half_t* __restrict__ Y, size_t rows, int K, int Kpad, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Kpad) return;
  const size_t r = i / Kpad;
  const int k = (int)(i % Kpad);
  split_store(Y, r, k, Kpad, k < K ? X[r * K + k] * scale : 0.f);
}

// Embedded-Gaussian attention over the 24 joints of one body (NONLocalBlock2D, nets/non_local_embedded_gaussian.py:68-79):
// qkv [bodies*24, 3*Ci] = [theta | phi | g] rows -> y [bodies*24, Ci] = softmax(theta phi^T) g.  One block per body; the 24 x 24
// logits are accumulated over 64-channel slabs staged in LDS.  (The block is off in every shipped config: correctness first.)
__global__ __launch_bounds__(256) void nonlocal_attention_kernel(const float* __restrict__ qkv, float* __restrict__ y, int Ci) {
  __shared__ float sT[kJ][65], sP[kJ][65], sF[kJ][kJ + 1];
  const int tid = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * kJ;
  const int ld = 3 * Ci;
  float f[3] = {0.f, 0.f, 0.f};                      // logits of pairs tid, tid + 256, tid + 512 (< 576)
  for (int c0 = 0; c0 < Ci; c0 += 64) {
    for (int i = tid; i < kJ * 64; i += 256) {
      const int j = i >> 6, c = i & 63;
      const bool ok = c0 + c < Ci;
      sT[j][c] = ok ? qkv[(row0 + j) * ld + c0 + c] : 0.f;
      sP[j][c] = ok ? qkv[(row0 + j) * ld + Ci + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pr = tid + 256 * k;
      if (pr < kJ * kJ) {
        const int a = pr / kJ, b = pr % kJ;
        float s = f[k];
        for (int c = 0; c < 64; ++c) s = fmaf(sT[a][c], sP[b][c], s);
        f[k] = s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pr = tid + 256 * k;
    if (pr < kJ * kJ) sF[pr / kJ][pr % kJ] = f[k];
  }
  __syncthreads();
  if (tid < kJ) {                                     // softmax over the key joints (dim = -1)
    float m = -3.4e38f;
    for (int b = 0; b < kJ; ++b) m = fmaxf(m, sF[tid][b]);
    float z = 0.f;
    for (int b = 0; b < kJ; ++b) { const float e = expf(sF[tid][b] - m); sF[tid][b] = e; z += e; }
    const float inv = 1.f / z;
    for (int b = 0; b < kJ; ++b) sF[tid][b] *= inv;
  }
  __syncthreads();
  for (int i = tid; i < kJ * Ci; i += 256) {
    const int a = i / Ci, c = i % Ci;
    float s = 0.f;
#pragma unroll 4
    for (int b = 0; b < kJ; ++b) s = fmaf(sF[a][b], qkv[(row0 + b) * ld + 2 * Ci + c], s);
    y[(row0 + a) * Ci + c] = s;
  }
}

}  // namespace

extern "C" int ehm_split_pack(const float* X, void* X2, int64_t rows, int K, int K_padded, float scale, void* stream) {
  EHM_CHECK_ARG(X && X2 && rows > 0 && K > 0 && K_padded >= K && K_padded % 32 == 0 && scale > 0.f);
  hipLaunchKernelGGL(pack_scaled_kernel, dim3((unsigned)ceil_div(rows * K_padded, 256)), dim3(256), 0, (hipStream_t)stream, X,
                     (half_t*)X2, (size_t)rows, K, K_padded, scale);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_linear_split(const ehm_linear_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->W && (d->Y || d->colmax));
  const bool lift = d->lift_points != nullptr;
  EHM_CHECK_ARG(lift ? (!d->A0 && d->lift_W4 && d->K1 == 0 && !d->relu_in0) : d->A0 != nullptr);
  EHM_CHECK_ARG(d->M > 0 && d->M % LBM == 0 && d->N > 0 && d->N % LBN == 0);
  EHM_CHECK_ARG(d->K0 > 0 && d->K0 % BK == 0 && d->K1 >= 0 && d->K1 % BK == 0 && (d->K1 == 0 || d->A1) && d->K0 + d->K1 >= 2 * BK);
  EHM_CHECK_ARG(d->rows_per_group > 0 && d->rows_per_group % LBM == 0 && d->M % d->rows_per_group == 0);
  EHM_CHECK_ARG(d->w_scale > 0.f);
  EHM_CHECK_ARG(d->M < (int64_t)1 << 31 && (int64_t)LBM * d->N * 4 < ((int64_t)1 << 31));
  LinArgs a;
  a.A0 = (const float*)d->A0; a.A1 = (const float*)d->A1; a.W = (const float*)d->W;
  a.bias = d->bias; a.gbias = d->group_bias; a.Y = (char*)d->Y; a.colmax = d->colmax;
  a.pts = d->lift_points; a.W4 = d->lift_W4;
  a.K0 = d->K0; a.K1 = d->K1; a.M = (int)d->M; a.N = d->N;
  a.rows_per_group = d->rows_per_group;
  a.valid_rows_per_group = d->valid_rows_per_group > 0 ? d->valid_rows_per_group : d->rows_per_group;
  a.relu_in0 = d->relu_in0; a.relu_out = d->relu_out; a.inv_scale = 1.f / d->w_scale;
  EHM_CHECK_ARG(d->group_bias_stride == 0 || d->group_bias_stride >= d->N);
  a.gstride = d->group_bias_stride > 0 ? d->group_bias_stride : d->N;
  const int64_t tiles = (d->M / LBM) * (d->N / LBN);
  int64_t blocks = 2 * (int64_t)ehm_num_cus();            // what is co-resident (80 KiB of LDS per block)
  if (blocks > tiles) blocks = tiles;
  if (d->relu_in0 && d->K1 != 0) {
    ehm_set_error("ehm_linear_split: relu_in0 needs K1 == 0 (the ReLU'd operand must be the only K segment)");
    return EHM_EINVAL;
  }
  if (d->hi_only) {
    if (lift) hipLaunchKernelGGL((linear_tile_kernel<false, true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if (d->relu_in0) hipLaunchKernelGGL((linear_tile_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear_tile_kernel<false, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  } else if (lift) hipLaunchKernelGGL((linear_tile_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (d->relu_in0) hipLaunchKernelGGL((linear_tile_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((linear_tile_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_skinny_gemm_f32(const float* X, const float* W, const float* bias, float* Y, int M, int K, int N, int relu, void* stream) {
  EHM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0);
  if (K % 32 != 0 || N % 32 != 0 || ((uintptr_t)X % 16) != 0) {
    ehm_set_error("ehm_skinny_gemm_f32 needs K %% 32 == 0, N %% 32 == 0 and a 16-byte aligned X (K = %d, N = %d)", K, N);
    return EHM_EINVAL;
  }
  const dim3 grid((unsigned)(N / 32), (unsigned)ceil_div(M, 32));
  if (K >= 1024) hipLaunchKernelGGL(skinny_gemm_f32_kernel<16>, grid, dim3(1024), 0, (hipStream_t)stream, X, W, bias, Y, M, K, N, relu);
  else hipLaunchKernelGGL(skinny_gemm_f32_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, M, K, N, relu);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_pointnet_lift(const float* pts, const float* Wpos, const float* bpos, void* R0, void* P32, int B, int N, int N_padded,
                                 int C, void* stream) {
  EHM_CHECK_ARG(pts && P32 && (!R0 || (Wpos && bpos)) && B > 0 && N > 0 && N_padded >= N && C > 0 && C % 32 == 0);
  const size_t rows = (size_t)B * N_padded;
  hipLaunchKernelGGL(pointnet_lift_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, pts, Wpos, bpos, (half_t*)R0,
                     (half_t*)P32, N, N_padded, C, rows);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_nonlocal_attention(const float* qkv, float* y, int64_t bodies, int Ci, void* stream) {
  if (bodies == 0) return 0;
  EHM_CHECK_ARG(qkv && y && bodies > 0 && bodies < (1ll << 31) && Ci > 0);
  hipLaunchKernelGGL(nonlocal_attention_kernel, dim3((unsigned)bodies), dim3(256), 0, (hipStream_t)stream, qkv, y, Ci);
  EHM_LAUNCH_CHECK();
  return 0;
}
