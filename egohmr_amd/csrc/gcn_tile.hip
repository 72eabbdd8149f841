// Modulated-GCN hidden conv (_GraphConv hid -> hid, modulated_gcn.py:21-28 / modulated_gcn_conv.py:39-50, + the residual of
// _ResGraphConv, modulated_gcn.py:38-42) on the f16 matrix cores of gfx950, two arithmetic modes from one tile engine:
//   P = 3  "f16x3": both GEMM operands stored as hi + lo f16 pairs (X2<32>, gcn_dev.h), three MFMAs per product
//          (lo*hi + hi*lo + hi*hi), f32 accumulate: 22-bit operands, f32-grade results (the parity path).  Since round 5 on
//          v_mfma_f32_16x16x32_f16: at the socket's power cap - where this kernel runs - the matrix pipe sustains 2.26 PFLOP/s
//          in that form against 1.86 as 32x32x16 on the same operands (tools/mfma_ceiling.py: a quarter of the accumulator
//          traffic per MAC); chain kernel 1014 -> 905 us per launch, same box.  See "16 x 16 x 32" in run_tiles;
//   P = 1  "f16":   plain f16 storage [rows][hid] and one MFMA per product (BASELINE config 5's fp16 denoiser, and the early
//          steps of the precision schedule).
// Tile = 192 rows (8 bodies x 24 joints) x 64 channels x both branches (W0 | W1); 4 waves as 2 x 2, 96 x 32(x2) per wave;
// operands stream L2 -> LDS with 16-byte global_load_lds DMA, one K tile = 128 bytes per row (64 k in f16, 32 k hi|lo in X2),
// two 40 KiB stages, XOR-swizzled on the source address so every ds_read_b128 fragment read is bank-conflict free; register
// double-buffered fragments, one barrier per K tile.
//
// What is new relative to round 1's kernel (gcn_f16r.hip, deleted): the tile loop is software-pipelined ACROSS tiles and the epilogue no longer
// touches LDS or a barrier.
//   * After the last barrier of a tile's K loop every operand fragment is in registers, so both LDS stages are dead: the block
//     immediately issues the operand DMA of its NEXT tile (weights always; activations when the producers of that tile were seen
//     complete a few K tiles earlier) and only then runs the epilogue, whose ~1500 VALU instructions and ~100 memory
//     instructions per wave cover the DMA latency.  Round 1 paid prologue 4.3k + epilogue 16-19k cycles per 110k-cycle tile
//     with the matrix pipe of that block idle.
//   * Epilogue: the MFMA row map gives every lane all 24 joints of two bodies for ONE channel (gcn.hip header), so the
//     modulated adjacency mix is register-local; results leave straight from registers - adjacent lanes own adjacent channels
//     and exchange halves with one DPP move so that each store is a full dword and a half-wave covers whole 64 / 128-byte
//     segments of two rows.  No transposition tile, no __syncthreads, waves of a block drift apart freely.
//   * One barrier + vmcnt(0) at the head of the next K loop both publishes the finished tile (chained launch) and hands the
//     DMA-written stages to all waves.
#include <stdlib.h>

#include <vector>

#include <type_traits>

#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"
#include "smpl_dev.h"
#include "step_dev.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int RK = 32;                                            // floats per row and K tile = 128 bytes
constexpr int A_T = 192 * RK;                                      // floats of a stage's activation region (24 KiB)
constexpr int stage_floats(int nw) { return A_T + 32 * nw * RK; }  // + weights: 64 channels x 2 branches per 4 waves -> 40 KiB (4 waves) / 56 KiB (8 waves)
constexpr int kStoreAux = 16;                                     // sc1: activation stores write through to L2 (chained launch hand-off)
constexpr int kLoadAux = 16;                                      // sc1: activation loads never hit a stale CU-L1 line

#ifdef EHM_STAMPS
__device__ unsigned long long* g_tdbg = nullptr;
// slot i of the block's tile number `stamp_it` (ring of 64 tiles per block, 8 slots per tile); tools/stamp_tiles.py
#define TSTAMP(i) do { if (g_tdbg && threadIdx.x == 0) g_tdbg[((size_t)blockIdx.x * 64 + (stamp_it & 63)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

template <int P>
struct Frags {
  half8 ah[3], al[P == 3 ? 3 : 1], bh[2], bl[P == 3 ? 2 : 1];
};

// Block-uniform identity of one output tile; everything else is derived from the launch arguments when it is needed, so that
// only three SGPRs per tile stay live across the K loop / the epilogue.
struct Tile {
  int layer, m_tile, n_tile;
};
// What the epilogue and the operand DMA need to know about a tile's conv.
struct TileIO {
  const float* X;       // activation matrix of the conv's input (rows of `rowf` floats)
  const float* W;       // packed weights of the layer (n-tile major, 128 rows per n-tile, rows of `rowf` floats)
  const char* Res;      // residual matrix (same format as X) or nullptr
  char* Y;              // output matrix
  const float *Ds, *M1s, *shift, *Aoff;
  const half_t* AoffH;
  int relu, out_f32;
};

struct ChainArgs {
  const LayerDev* layers;   // device array [nl]
  void* buf[3];             // activation buffers: conv 2b reads cur -> writes buf[1]; conv 2b+1 reads buf[1] (+ residual cur) -> nxt
  int nl, m_tiles, n_tiles;
  unsigned int* tickets;    // [8]
  unsigned int* done;       // [nl][m_tiles]
  unsigned int* err;        // this launch: a wait timed out / a tile was never produced (results invalid)
  unsigned int* sticky;     // never cleared by a launch: accumulates err over a whole sampling loop (ehm_gcn_stack_status): bit 0 = a wait timed out / a tile is
                            // missing, bit 2 (kStickySaturated) = an activation reached the f16 range in an epilogue store and was clamped
  unsigned int* finished;   // blocks that ran out of tickets; the last one audits done[nl-1][*]
  int nq;                   // queues = XCDs
  unsigned int rows_valid;  // rows in front of the tile padding (range guard of the epilogue stores)
};

struct OneArgs {
  LayerDev L;
  const void* X;
  const void* Res;
  void* Y;
  int m_tiles, out_f32;
  unsigned int* sticky;     // the handle's status word (ehm_gcn_stack_status): bit 2 = an f16 store saturated
  unsigned int rows_valid;  // rows in front of the tile padding: the range guard looks at these only (padding rows hold don't-care values)
};

template <int P>
__device__ __forceinline__ const float* layer_weights(const LayerDev& L) {
  return P == 3 ? (const float*)L.Ws : (const float*)L.Wh;
}

// The engine.  CHAIN = true: persistent blocks, per-XCD ticket queues over (layer, row tile, channel tile) with per-row-tile
// completion counters (see the header of ehm_gcn_tile_chain_impl); CHAIN = false: one tile per block, one conv per launch.
// NW = waves per block.  4: 192 rows x 64 channels per block, two blocks per CU.  8 (f16 chain only): ONE block per CU owns 192 rows x 128
// channels - the activation tile is staged once for both channel halves, 56 KiB instead of 80 KiB of operands per K tile and CU; with
// one MFMA per product the K loop is bound by LDS bandwidth (writes + fragment reads), see DESIGN.md 3.2.

// MODE 0: one tile per block, one conv per launch; 1: chained convs of one step.  (The one-launch sampling loop that was MODE 2 in rounds 4 - 5 - bit-equal,
// 12 % slower - is recorded in docs/EXPERIMENTS.md 3.7 and lives in the history: git show c4b8e19:egohmr_amd/csrc/gcn_loop_dev.h.)
template <int P, int MODE, int NW, class Args>
__device__ __forceinline__ void run_tiles(float* lds, const Args& a) {
  constexpr bool CHAIN = MODE == 1;
  constexpr bool M16 = P == 3;       // split-f16 mode: v_mfma_f32_16x16x32_f16 (see "16 x 16 x 32" below); plain f16: v_mfma_f32_32x32x16_f16
  static_assert(NW == 4 || (NW == 8 && P == 1 && CHAIN), "the 8-wave tile exists for the chained f16 kernel (f16x3 is power-bound: 8 waves measured 150 vs 152 us)");
  // MODE.FP16_OVFL = 1 for the life of the wave: every f32 -> f16 conversion of the epilogue clamps to +-65504 instead of producing inf
  // (hwreg MODE = 1, bit 23).  The explicit clamps this replaces were 144 v_med3_f32 + their canonicalising v_max_f32 per wave and tile,
  // in an epilogue during which the block's matrix pipes idle.
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  constexpr int NWN = NW / 2;                 // waves across the channels
  constexpr int NT = 32 * NWN;                // channels per tile
  constexpr int BROWS = 2 * NT;               // weight rows per stage (W0 | W1 per 64 channels)
  constexpr int STG = stage_floats(NW);
  constexpr int NDA = 24 / NW, NDB = BROWS / (8 * NW);   // DMA instructions per wave and stage: activations 6 / 3, weights 4 / 4
  constexpr int KS = P == 3 ? 2 : 4;          // 16-wide k-steps per K tile
  constexpr int NM = P == 3 ? 18 : 6;         // MFMAs per k-step
  constexpr int NR = P == 3 ? 10 : 5;         // ds_read_b128 per k-step
  const int tid = threadIdx.x;
  // Per-thread constants are RE-DERIVED at the head of every tile from an opaque copy of the thread id (thread_consts below), so
  // that none of them has to stay in a register across the epilogue - with them live the chained kernel spilled ~120 VGPRs.
  int lane, wave, wm, wn, mi, g;
  bool odd;

  int K, N;
  if constexpr (CHAIN) { K = a.layers[0].K; N = a.layers[0].N; } else { K = a.L.K; N = a.L.N; }
  const int rowf = P == 3 ? K : K / 2;        // floats per operand row
  const int KT = rowf / RK;

  // ---- DMA: one wave instruction = 8 rows x 128 B; physical 16-byte chunk c of row r holds logical chunk c ^ ((r>>1)&7)
  int r0, swz;                                               // r0 = 8 wave + lane / 8; swz = ((lane & 7) ^ ((r0 >> 1) & 7)) << 2; r0 + 32 i keeps the key
  const size_t row32 = (size_t)(8 * NW) * rowf;              // a wave's consecutive DMA instructions are 32 (64) rows apart
  // operand stream addressing: buffer form - a tile's base in an SGPR descriptor, the lane's row / chunk in ONE 32-bit VGPR offset that is
  // the same for every piece of the tile, the piece (rows + K tile) in the scalar offset: no vector-ALU address arithmetic in the K loop
  // (with 64-bit global pointers every piece cost two v_add: 20 VALU per K tile and wave, and VALU issue is time the matrix pipe of the
  // SIMD does not get)
  __amdgpu_buffer_rsrc_t rsA, rsB;
  int voAB;
  [[maybe_unused]] int voA16 = 0;                            // M16: my activation row is a permuted one (below)
  auto io_of = [&](const Tile& t) -> TileIO {
    TileIO o;
    if constexpr (CHAIN) {
      const int blk = t.layer >> 1, cb = (blk & 1) ? 2 : 0, nb = (blk & 1) ? 0 : 2;
      const bool oddl = t.layer & 1;
      const LayerDev& L = a.layers[t.layer];
      o.X = (const float*)(oddl ? a.buf[1] : a.buf[cb]);
      o.W = layer_weights<P>(L);
      o.Res = oddl ? (const char*)a.buf[cb] : nullptr;
      o.Y = (char*)(oddl ? a.buf[nb] : a.buf[1]);
      o.Ds = L.Ds; o.M1s = L.M1s; o.shift = L.shift; o.Aoff = L.Aoff; o.AoffH = L.AoffH;
      o.relu = L.relu;
      o.out_f32 = P == 3 && t.layer == a.nl - 1;   // (f16 rows are half as long: a float32 row of the last conv would land on two f16 rows of OTHER row tiles still being read)
    } else {
      o.X = (const float*)a.X; o.W = layer_weights<P>(a.L); o.Res = (const char*)a.Res; o.Y = (char*)a.Y;
      o.Ds = a.L.Ds; o.M1s = a.L.M1s; o.shift = a.L.shift; o.Aoff = a.L.Aoff; o.AoffH = a.L.AoffH;
      o.relu = a.L.relu; o.out_f32 = P == 3 && a.out_f32;
    }
    return o;
  };
  auto set_tile_ptrs = [&](const Tile& t) {
    const TileIO o = io_of(t);
    rsA = ehm_buffer_rsrc(o.X + (size_t)t.m_tile * 192 * rowf);
    rsB = ehm_buffer_rsrc(o.W + (size_t)t.n_tile * BROWS * rowf);
    voAB = (r0 * rowf + swz) * 4;
    if constexpr (M16) {
      // 16 x 16 x 32: LDS row rho = 16 rt + i of a wave's 96 (MFMA row i of row tile rt) holds the tile row 24 (i >> 2) + 4 rt + (i & 3), so that the C
      // layout (row = 4 (lane >> 4) + reg) hands lane group rg = lane >> 4 the 24 joints of body rg: joint 4 rt + reg.  My LDS rows are r0 + 32 i
      // (i = 0..5; r0 < 32): row tiles (r0 >> 4) + 2 (i % 3) of wave-row i / 3 - the tile row moves by 96 (i / 3) + 8 (i % 3), the same for every lane.
      const int i16 = r0 & 15;
      voA16 = ((24 * (i16 >> 2) + 4 * (r0 >> 4) + (i16 & 3)) * rowf + swz) * 4;
    }
  };
  auto dma_a = [&](int buf, int kt, int i) {
    if constexpr (M16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (AS3 void*)(lds + buf * STG + (wave + NW * i) * 256), 16, voA16, ((96 * (i / 3) + 8 * (i % 3)) * rowf + kt * RK) * 4, 0, kLoadAux);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (AS3 void*)(lds + buf * STG + (wave + NW * i) * 256), 16, voAB, (i * (int)row32 + kt * RK) * 4, 0, kLoadAux);
  };
  auto dma_b = [&](int buf, int kt, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (AS3 void*)(lds + buf * STG + A_T + (wave + NW * i) * 256), 16, voAB, (i * (int)row32 + kt * RK) * 4, 0, 0);
  };
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < NDA; ++i) dma_a(buf, kt, i);
#pragma unroll
    for (int i = 0; i < NDB; ++i) dma_b(buf, kt, i);
  };
  // The first two K tiles of a tile are fetched around the previous tile's epilogue: both weight stages and activation stage 0 before it,
  // activation stage 1 (whose pieces are the epilogue's scratch) after it = "late" = what a wave issues last = what the head of the K
  // loop leaves in flight.  (NW = 8: a wave owns only three 1 KiB activation pieces; the other half of its scratch is a dedicated 24 KiB
  // region behind the stages - 136 KiB of LDS per block - so that the weights of stage 1 need not wait for the epilogue.)
  constexpr bool B1_LATE = false;
  constexpr int LATE = NDA + (B1_LATE ? NDB : 0);
  auto issue_b_early = [&]() {
#pragma unroll
    for (int i = 0; i < NDB; ++i) dma_b(0, 0, i);
    if constexpr (!B1_LATE) {
#pragma unroll
      for (int i = 0; i < NDB; ++i) dma_b(1, 1, i);
    }
  };
  auto issue_a0 = [&]() {
#pragma unroll
    for (int i = 0; i < NDA; ++i) dma_a(0, 0, i);
  };
  auto issue_late = [&]() {
#pragma unroll
    for (int i = 0; i < NDA; ++i) dma_a(1, 1, i);
    if constexpr (B1_LATE) {
#pragma unroll
      for (int i = 0; i < NDB; ++i) dma_b(1, 1, i);
    }
  };

  // ---- fragments (v_mfma_f32_32x32x16_f16: lane l holds row l&31, k = 8*(l>>5) .. +7 of a 16-wide step = one 16-byte chunk)
  // Row permutation of the in-register epilogue: MFMA row i of row tile t <-> wave row 48*((i>>2)&1) + 24*(i&1) + ((i>>1)&1) + 2*(i>>3) + 8t.
  // With the C layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) lane (mi, g) then owns, for ONE channel, all 24 joints of the
  // wave's bodies 2g and 2g+1, joint j of the two bodies in the ADJACENT registers 2*(j&7), 2*(j&7)+1 of accumulator j>>3.
  //   rA = 96 wm + 48 ((mi>>2)&1) + 24 (mi&1) + ((mi>>1)&1) + 2 (mi>>3);  rB = 32 wn + mi (+ 64 for the W1 branch)
  //   keyA = (rA>>1)&7 (+8t flips its bit 2 for odd t), keyB = (rB>>1)&7 (+64 leaves it)
  // logical chunk of (k-step s, hi/lo hl, lane half g): X2 tile = [hi k0-31 | lo k0-31] -> 4 hl + 2 s + g;  f16 tile = k0-63 -> 2 s + g
  int oA[KS][P == 3 ? 2 : 1][2], oB[KS][P == 3 ? 2 : 1];
  [[maybe_unused]] int oA16[2], oB16[2];
  auto thread_consts = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));                              // opaque: keeps hipcc from hoisting what follows out of the tile loop
    lane = t & 63;
    wave = __builtin_amdgcn_readfirstlane(t >> 6);
    wm = wave / NWN; wn = wave % NWN;
    mi = lane & 31; g = lane >> 5;
    odd = lane & 1;
    r0 = 8 * wave + (lane >> 3);
    swz = ((lane & 7) ^ ((r0 >> 1) & 7)) << 2;
    const int rA = 96 * wm + 48 * ((mi >> 2) & 1) + 24 * (mi & 1) + ((mi >> 1) & 1) + 2 * (mi >> 3);
    const int rB = NW == 8 ? 128 * (wn >> 1) + 32 * (wn & 1) + mi : 32 * wn + mi;   // 8 waves: 64-channel weight group wn >> 1 of the pair
    const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int hl = 0; hl < (P == 3 ? 2 : 1); ++hl) {
        const int c = (P == 3 ? 4 * hl : 0) + 2 * s + g;
#pragma unroll
        for (int o = 0; o < 2; ++o) oA[s][hl][o] = rA * RK + (((c ^ keyA) ^ (4 * o)) << 2);
        oB[s][hl] = A_T + rB * RK + ((c ^ keyB) << 2);
      }
    if constexpr (M16) {
      const int i16 = lane & 15, kg = lane >> 4, key = (i16 >> 1) & 7;
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        oA16[hl] = (96 * wm + i16) * RK + (((4 * hl + kg) ^ key) << 2);
        oB16[hl] = A_T + (32 * wn + i16) * RK + (((4 * hl + kg) ^ key) << 2);
      }
    }
  };
  thread_consts();
  auto read_frags = [&](Frags<P>& f, int buf, int s) {
    const float* S = lds + buf * STG;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f.ah[t] = *(const half8*)(S + oA[s][0][t & 1] + 8 * t * RK);
      if constexpr (P == 3) f.al[t] = *(const half8*)(S + oA[s][1][t & 1] + 8 * t * RK);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f.bh[u] = *(const half8*)(S + oB[s][0] + 64 * u * RK);
      if constexpr (P == 3) f.bl[u] = *(const half8*)(S + oB[s][1] + 64 * u * RK);
    }
  };

  f32x16 acc0[3], acc1[3];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }
  };
  auto mfmas = [&](const Frags<P>& f) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if constexpr (P == 3) {                           // small cross terms first, leading term last
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[t], f.bh[1], acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[0], acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bl[1], acc1[t], 0, 0, 0);
      }
      acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[0], acc0[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[t], f.bh[1], acc1[t], 0, 0, 0);
    }
  };
  // sched_group_barrier masks: 0x008 MFMA, 0x100 DS read, 0x010 VMEM
  auto pin_reads = [&]() {                      // MFMA, read, MFMA, read, ... then the remaining MFMAs
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  };
  auto pin_reads_dma = [&]() {                  // the reads one per MFMA, the ten DMAs spread over the phase
    if constexpr (P == 3) {                     // 18 MFMAs: 10 x (MFMA, read), then 2,2,1,1,1,1,1,1 DMAs behind the last 8
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    } else if constexpr (NW == 4) {             // 6 MFMAs: 5 x (MFMA, read, 2 DMA), MFMA
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    } else {                                    // 7 DMAs: 2 x (MFMA, read, 2 DMA), 3 x (MFMA, read, DMA), MFMA
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
      }
#pragma unroll
      for (int i = 2; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };

  // ---- chained launch: tickets, dependencies
  unsigned int q = 0, ipl = 0, total = 0;
  volatile unsigned int* slot = (volatile unsigned int*)(lds + ((KT - 2) & 1) * STG);   // dead LDS between barrier(KT-2) and the next tile's DMA
  auto decode = [&](unsigned int t, Tile& o) {
    if constexpr (CHAIN) {
      const int layer = (int)(t / ipl), r = (int)(t % ipl);
      o.m_tile = (int)q + a.nq * (r / a.n_tiles);
      o.n_tile = r % a.n_tiles;
      o.layer = layer;
    }
  };
  // what a conv tile waits for: a monotone counter and the value it must have reached (nullptr: nothing)
  auto dep_of = [&](const Tile& t, unsigned int& target) -> const unsigned int* {
    if constexpr (CHAIN) {
      target = (unsigned int)a.n_tiles;
      return t.layer > 0 ? a.done + (size_t)(t.layer - 1) * a.m_tiles + t.m_tile : nullptr;
    } else {
      target = 0;
      return nullptr;
    }
  };
  auto wait_counter = [&](const unsigned int* f, unsigned int target) {   // one lane; never hangs the device (see below)
    if constexpr (CHAIN) {
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if (spins > (1 << 22) || ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_or(a.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  };
  auto poll_deps = [&](const Tile& t) {       // (whole block; ends with a barrier)
    if constexpr (CHAIN) {
      unsigned int target = 0;
      const unsigned int* f = dep_of(t, target);
      if (f != nullptr && tid == 0) {
        // never hang the device: give up after ~1 s (or at once when somebody else already has) and flag the launch - and the whole
        // loop - as failed
        wait_counter(f, target);
      }
      __syncthreads();
    }
  };
  auto publish = [&](const Tile& t) {
    if constexpr (CHAIN)
      if (tid == 0) __hip_atomic_fetch_add(&a.done[(size_t)t.layer * a.m_tiles + t.m_tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  Tile cur{}, nxt{};
  unsigned int t_cur = 0;
  if constexpr (CHAIN) {
    q = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % (unsigned int)a.nq;   // HW_REG_XCC_ID[3:0] -> my queue
    const int cm = (a.m_tiles - (int)q + a.nq - 1) / a.nq;                          // row tiles of this queue: q, q + nq, ...
    ipl = (unsigned int)((cm > 0 ? cm : 0) * a.n_tiles);
    total = ipl * (unsigned int)a.nl;
    if (tid == 0) slot[0] = __hip_atomic_fetch_add(&a.tickets[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    t_cur = __builtin_amdgcn_readfirstlane(slot[0]);
    __syncthreads();
  } else {
    const int n_tiles = N / NT, tot = a.m_tiles * n_tiles, bid = blockIdx.x;
    const int lin = ((tot & 7) == 0) ? (bid & 7) * (tot >> 3) + (bid >> 3) : bid;   // XCD-aware tile order
    cur.m_tile = lin / n_tiles; cur.n_tile = lin % n_tiles; cur.layer = 0;
  }

  bool pending_publish = false;     // the previous tile's stores are issued but its counter is not bumped yet
  Tile prev{};
  if (!CHAIN || t_cur < total) {
    if constexpr (CHAIN) decode(t_cur, cur);
    set_tile_ptrs(cur);
    issue_b_early();
    poll_deps(cur);
    issue_a0();
    issue_late();
  }

#ifdef EHM_STAMPS
  int stamp_it = -1;
#endif
  while (!CHAIN || t_cur < total) {
#ifdef EHM_STAMPS
    ++stamp_it;
#endif
    TSTAMP(0);
    // ---- head: stages 0 / 1 of `cur` are in flight (plus, possibly, the previous tile's stores).  Stage 1's six activation pieces are
    //      always the LAST memory instructions a wave has issued: wait for everything older (vmcnt counts in issue order) - my stores
    //      have reached L2, stage 0 has landed - and leave those six to the first K tile's own wait.
    //      When the previous tile's stores are still in flight (early path) they were issued AFTER stage 0's DMA and before those six:
    //      the K loop only needs stage 0, so leave the stores outstanding too and publish behind the first K tile's vmcnt(0) + barrier.
#ifdef EHM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the stamp stores of lane 0 sit behind the six DMAs)
#else
    if constexpr (NW == 8) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                         // everything but the three late pieces (the stores are older)
    } else if (pending_publish) {
      if constexpr (P == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // 12 sixteen-byte stores per wave and tile (X2 hi + lo, or float32)
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                   // 6
    } else {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
#endif
    __syncthreads();
    TSTAMP(5);
    thread_consts();
    unsigned int t_next = 0xffffffffu;
    if constexpr (CHAIN)
      if (tid == 0) t_next = __hip_atomic_fetch_add(&a.tickets[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int dep_seen = 0;
    zero_acc();
    Frags<P> f0, f1;
    if constexpr (!M16) read_frags(f0, 0, 0);
    TSTAMP(1);
    // ---- 16 x 16 x 32 (split-f16 mode).  A K tile (32 k) is ONE k-step of this instruction: lane (i = l & 15, kg = l >> 4) holds row i of a 16-row tile and
    // the 16-byte chunk kg (hi halves) / 4 + kg (lo halves) of the tile's 128-byte rows - one ds_read_b128 per 16 x 32 operand block, 20 per K tile and wave
    // as before, for 72 MFMAs instead of 36.  A wave's 96 x 32 (x 2 branches) tile = 6 row tiles x 4 column tiles = 24 f32x4 accumulators (the same 96
    // registers).  Holding a whole K tile's fragments twice would cost 160 registers; instead a K tile runs as four phases (row half rh, branch ch) of 18
    // MFMAs in SNAKE order - (0,0) (0,1) (1,1) (1,0), then (0,1) (0,0) (1,0) (1,1) for the next K tile, and so on - so that consecutive phases share one
    // operand half and the other is refilled from LDS while it is not in use: 80 fragment registers, nothing double-buffered.
    // Operand halves A[rh] (row tiles 3 rh .. + 2 of 16 rows), B[ch] (branch ch: two 16-channel column tiles), 6 x 4 accumulators
    [[maybe_unused]] half8 Ah[2][3], Al[2][3], Bh[2][2], Bl[2][2];
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    [[maybe_unused]] f32x4a c16[6][4];
    [[maybe_unused]] auto ldA = [&](auto rhc, int buf) {
      constexpr int rh = decltype(rhc)::value;
      const float* S = lds + buf * STG;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        Ah[rh][t] = *(const half8*)(S + oA16[0] + 16 * (3 * rh + t) * RK);
        Al[rh][t] = *(const half8*)(S + oA16[1] + 16 * (3 * rh + t) * RK);
      }
    };
    [[maybe_unused]] auto ldB = [&](auto chc, int buf) {
      constexpr int ch = decltype(chc)::value;
      const float* S = lds + buf * STG;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        Bh[ch][u] = *(const half8*)(S + oB16[0] + (16 * u + 64 * ch) * RK);
        Bl[ch][u] = *(const half8*)(S + oB16[1] + (16 * u + 64 * ch) * RK);
      }
    };
    [[maybe_unused]] auto mm = [&](auto rhc, auto chc) {            // 18 MFMAs: small cross terms first, six independent accumulators per term
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
    // LDS reads behind every second MFMA of a phase, DMA instructions in the gaps between them (same box: 918 -> 903 us per launch against
    // "reads one per MFMA from the start, DMAs behind them"; no pinning at all measured like the latter)
    [[maybe_unused]] auto pin16 = [&](int reads, int dmas) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if ((i & 1) == 0 && (i >> 1) < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if ((i & 1) == 1 && (i >> 1) < dmas) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // one K tile of parity PAR (snake order of the four (row half, branch) phases: only one operand half changes between consecutive phases, so
    // no half is ever double-buffered).  mode 0: steady (stages K tile kt + 2 into `buf`), 1: second-last, 2: last (no next-tile loads)
    [[maybe_unused]] auto tile16 = [&](auto parc, int buf, int kt, int mode) {
      constexpr int PAR = decltype(parc)::value;
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
      __syncthreads();
      if (pending_publish) { publish(prev); pending_publish = false; }
      if (mode == 2) return;
      ldA(I0{}, buf ^ 1);
      if (mode == 0) {
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < NDA; ++i) dma_a(buf, kt + 2, i);
      }
      mm(I1{}, CS{});
      pin16(6, mode == 0 ? NDA : 0);
      __builtin_amdgcn_sched_barrier(0);
      ldB(CS{}, buf ^ 1);
      if (mode == 0) {
#pragma unroll
        for (int i = 0; i < NDB; ++i) dma_b(buf, kt + 2, i);
      }
      mm(I1{}, CF{});
      pin16(4, mode == 0 ? NDB : 0);
      if (mode == 0) __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (M16) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) c16[t][u] = f32x4a{0.f, 0.f, 0.f, 0.f};
      ldA(I0{}, 0);
      ldB(I0{}, 0);
    }

    // One K tile, phases s = 0 .. KS-1 on alternating fragment sets.  MODE 0: steady state (fetches tile kt + 2 into the stage it
    // just emptied), 1: second-last tile (nothing left to fetch), 2: last tile (stops after the barrier; the caller issues the
    // last k-step's MFMAs together with the epilogue's table loads).
    auto phases_before_barrier = [&](int buf) {
#pragma unroll
      for (int s = 0; s < KS - 1; ++s) {
        Frags<P>& fr = (s & 1) ? f0 : f1;      // set to fill: k-step s + 1
        Frags<P>& fm = (s & 1) ? f1 : f0;      // set to multiply: k-step s
        read_frags(fr, buf, s + 1);
        mfmas(fm);
        pin_reads();
        if (s < KS - 2) __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                         // this tile is in everyone's registers; the next one is complete in LDS
    };
    Frags<P>& f_last = ((KS - 1) & 1) ? f1 : f0;   // set of a tile's last k-step

    if constexpr (M16) {
      for (int kt = 0; kt < KT - 2; kt += 2) {
        tile16(I0{}, 0, kt, 0);
        tile16(I1{}, 1, kt + 1, 0);
      }
    } else
    for (int kt = 0; kt < KT - 2; ++kt) {
      const int buf = kt & 1;
      phases_before_barrier(buf);
      if (pending_publish) { publish(prev); pending_publish = false; }     // every wave's stores of the previous tile have landed (vmcnt(0) + barrier)
      read_frags(f0, buf ^ 1, 0);              // (KS even: the next tile starts on set 0 again)
      // the phase that stages the operand pieces runs at raised priority: a wave in piece issue is the one its SIMD partner has to wait for
      // least (split-f16 conv 144.1 -> 142.1 us, f16 conv 59.8 -> 58.0 us, same-box A/B; priority 3 and priority on the OTHER phase: no gain)
      __builtin_amdgcn_s_setprio(2);
      stage(buf, kt + 2);
      mfmas(f_last);
      pin_reads_dma();
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (CHAIN) {
      if (tid == 0 && t_next < total) {        // were the producers of my NEXT tile complete already?
        Tile tn;
        decode(t_next, tn);
        // one relaxed read, no waiting: a next tile that depends on the tile I am still computing (next layer, same rows) simply reads
        // an incomplete counter and takes the late path; nobody ever WAITS while holding an unpublished tile, so no cycle can form
        unsigned int target = 0;
        const unsigned int* f = dep_of(tn, target);
        dep_seen = (f == nullptr || __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) ? 1u : 0u;
      }
    }
    if constexpr (M16) {
      // second-last K tile (stage 0): its first half up to the barrier by hand so that the slot words can be written behind the barrier
      ldB(I1{}, 0);
      mm(I0{}, I0{});
      pin16(4, 0);
      __builtin_amdgcn_sched_barrier(0);
      ldA(I1{}, 0);
      mm(I0{}, I1{});
      pin16(6, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (pending_publish) { publish(prev); pending_publish = false; }
      if constexpr (CHAIN)
        if (tid == 0) {                                           // stage 0 is dead from here on
          slot[0] = t_next; slot[1] = dep_seen;
        }
      ldA(I0{}, 1);
      mm(I1{}, I1{});
      pin16(6, 0);
      __builtin_amdgcn_sched_barrier(0);
      ldB(I1{}, 1);
      mm(I1{}, I0{});
      pin16(4, 0);
    } else
    {
      const int buf = (KT - 2) & 1;
      phases_before_barrier(buf);
      if (pending_publish) { publish(prev); pending_publish = false; }     // (KT == 2: the loop above did not run)
      if constexpr (CHAIN)
        if (tid == 0) {                                           // stage `buf` is dead from here on
          slot[0] = t_next; slot[1] = dep_seen;
        }
      read_frags(f0, buf ^ 1, 0);
      mfmas(f_last);
      pin_reads();
    }
    // ---- last K tile; the per-channel epilogue constants are fetched under its MFMAs
    const TileIO io = io_of(cur);
    const int n = NT * cur.n_tile + 32 * wn + (M16 ? (lane & 15) : mi);
    const unsigned int n4 = (unsigned int)n * 4u;
    const unsigned int tblrow = (unsigned int)N * 4u;
    float dj[kJ], mj[kJ], sh;
    [[maybe_unused]] float djb[kJ], mjb[kJ], shb = 0.f;          // M16: a lane owns TWO channels (n and n + 16) of one body
    half8 af[3];                       // P == 1: [Aoff | I] fragments of the matrix-core adjacency mix
    unsigned int nt = 0xffffffffu, ready = 0;
    if constexpr (M16) tile16(I1{}, 1, KT - 1, 2);            // last K tile (stage 1): first half + barrier; its second half runs under the table loads below
    else phases_before_barrier((KT - 1) & 1);
    if constexpr (CHAIN) { nt = slot[0]; ready = slot[1]; }
    {
      Frags<P>& fm = f_last;
      const __amdgpu_buffer_rsrc_t dsB = ehm_buffer_rsrc(io.Ds);
      const __amdgpu_buffer_rsrc_t m1B = ehm_buffer_rsrc(io.M1s);
      __builtin_amdgcn_sched_barrier(0);
      sh = io.shift[n];
      if constexpr (P == 1) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) af[s3] = ((const half8*)io.AoffH)[s3 * 64 + lane];
      }
      typedef unsigned int u32x4_tbl __attribute__((ext_vector_type(4)));
      const unsigned int nrow = (unsigned int)n * (unsigned int)(kJ * 4);      // tables are [N][24]: my channel's 96 bytes
#pragma unroll
      for (int q4 = 0; q4 < kJ / 4; ++q4) {
        const u32x4_tbl d4 = __builtin_amdgcn_raw_buffer_load_b128(dsB, nrow, 16 * q4, 0);
        const u32x4_tbl m4 = __builtin_amdgcn_raw_buffer_load_b128(m1B, nrow, 16 * q4, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned int du = d4[i], mu = m4[i];       // (hipcc: __builtin_bit_cast of a vector ELEMENT expression reads element 0)
          dj[4 * q4 + i] = __builtin_bit_cast(float, du);
          mj[4 * q4 + i] = __builtin_bit_cast(float, mu);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (M16) {
        mm(I1{}, I0{});
        mm(I1{}, I1{});
        // the second channel's constants: requested here, behind the last MFMAs (their fragments are dead), consumed behind the next tile's DMA issue
        __builtin_amdgcn_sched_barrier(0);
        shb = io.shift[n + 16];
#pragma unroll
        for (int q4 = 0; q4 < kJ / 4; ++q4) {
          const u32x4_tbl d4 = __builtin_amdgcn_raw_buffer_load_b128(dsB, nrow + 16u * (kJ * 4), 16 * q4, 0);
          const u32x4_tbl m4 = __builtin_amdgcn_raw_buffer_load_b128(m1B, nrow + 16u * (kJ * 4), 16 * q4, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned int du = d4[i], mu = m4[i];
            djb[4 * q4 + i] = __builtin_bit_cast(float, du);
            mjb[4 * q4 + i] = __builtin_bit_cast(float, mu);
          }
        }
      } else {
        mfmas(fm);
      }
    }
    TSTAMP(2);
    // ---- next tile's operand DMA goes out before this tile's epilogue
    bool have_next = false, a_issued = false;
    if constexpr (CHAIN) {
      nt = __builtin_amdgcn_readfirstlane(nt);
      ready = __builtin_amdgcn_readfirstlane(ready);
      have_next = nt < total;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                        // everybody has read the slot (and, long ago, its last fragments): all LDS is dead
      if (have_next) {
        decode(nt, nxt);
        set_tile_ptrs(nxt);
        issue_b_early();
        if (ready) {                        // stage 0 now; stage 1's pieces serve as the epilogue's scratch first
          issue_a0();
          a_issued = true;
        }
      }
    }
    TSTAMP(3);

    // ---- epilogue (Ds / M1s carry 1 / w_scale).  Values are produced with lane = channel, registers = rows (the MFMA layout), but
    //      dword accesses in that layout cost 500-800 cycles per wave instruction here (measured: even / odd lanes on different rows
    //      defeat the quad coalescer; 24 + 24 of them made a 27k-cycle epilogue).  So each wave turns ITS 96 x 32 sub-tile through a
    //      wave-private 6 KiB scratch in LDS - the six 1 KiB pieces of stage 1's activation region that its own DMA instructions
    //      fill, so no other wave ever touches them and no barrier is needed - into rows of 4 consecutive channels per lane:
    //      residual and output move as 8 / 16-byte accesses, 8 lanes per 64-byte row segment.  Two passes of 48 rows.
    const unsigned int arow = (unsigned int)N * (P == 3 ? 4u : 2u);       // bytes per activation row
    const bool out_f32 = io.out_f32 != 0, has_res = io.Res != nullptr, relu = io.relu != 0;
    const float floor_v = relu ? 0.f : -3.4e38f;              // one v_max per value instead of a v_max and a select (the values are finite)
    const __amdgpu_buffer_rsrc_t yB = ehm_buffer_rsrc(io.Y + ((size_t)cur.m_tile * 192 + 96 * wm) * (out_f32 ? (size_t)N * 4 : (size_t)arow));
    const __amdgpu_buffer_rsrc_t resB = ehm_buffer_rsrc((has_res ? io.Res : io.Y) + ((size_t)cur.m_tile * 192 + 96 * wm) * arow);
    // item (it, lane), it = 0..2 per pass: scratch row rl = 16 it + (lane>>2), channels 8 (lane&3) .. +7 of the wave's 32: 16 bytes of f16
    // (32 of float32) per lane, 4 lanes per 64-byte row segment, 16 rows per wave instruction
    const int lr = lane >> 2, c8 = 8 * (lane & 3);
    const int ch0 = NT * cur.n_tile + 32 * wn + c8;                        // first channel of my items
    unsigned int col_in, col_out;                                           // byte offsets of the item's columns: residual / output
    if constexpr (P == 3) col_in = (unsigned int)(((ch0 >> 5) * 64 + (ch0 & 31)) * 2);   // X2: 8 hi halves here, 8 lo halves 64 B on
    else col_in = (unsigned int)ch0 * 2u;
    col_out = out_f32 ? (unsigned int)ch0 * 4u : col_in;
    const unsigned int orow = out_f32 ? tblrow : arow;
    // wave row of scratch row rl in pass p:  P == 1: pass = half-wave x -> 48 p + rl;  P == 3: pass = body beta, scratch rows = 24 g + joint
    // -> 48 (rl / 24) + 24 p + rl % 24
    // M16: pass = joints 12 p .. + 11 of the wave's four bodies, scratch row = 4 (joint - 12 p) + body -> wave row 24 (rl & 3) + 12 p + (rl >> 2)
    auto item_vrow = [&](int p, int it) -> unsigned int {
      const int rl = 16 * it + lr;
      if constexpr (M16) return (unsigned int)(24 * (lr & 3) + 12 * p + 4 * it + (lr >> 2));
      return (unsigned int)(P == 1 ? 48 * p + rl : 24 * p + rl + (rl >= 24 ? 24 : 0));
    };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    u32x4_t rq[6];                                                         // residual: P == 1 both passes (3 + 3 items), P == 3 the current pass (hi, lo per item)
    auto load_res_pass = [&](int p) {
      if (has_res) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
          const unsigned int vo = item_vrow(p, it) * arow + col_in;
          if constexpr (P == 3) {
            rq[2 * it] = __builtin_amdgcn_raw_buffer_load_b128(resB, vo, 0, kLoadAux);
            rq[2 * it + 1] = __builtin_amdgcn_raw_buffer_load_b128(resB, vo + 64u, 0, kLoadAux);
          } else {
            rq[3 * p + it] = __builtin_amdgcn_raw_buffer_load_b128(resB, vo, 0, kLoadAux);
          }
        }
      }
    };

    f32x2 dp[kJ], gp[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      if constexpr (M16) {                     // the pair = my two channels (n, n + 16) of ONE body; joint j = accumulator row tile j >> 2, register j & 3
        const f32x2 a0 = f32x2{c16[j >> 2][0][j & 3], c16[j >> 2][1][j & 3]};
        const f32x2 a1 = f32x2{c16[j >> 2][2][j & 3], c16[j >> 2][3][j & 3]};
        dp[j] = __builtin_elementwise_fma(f32x2{dj[j], djb[j]}, a0, f32x2{sh, shb});
        gp[j] = a1 * f32x2{mj[j], mjb[j]};
      } else {
        const f32x2 a0 = f32x2{acc0[j >> 3][2 * (j & 7)], acc0[j >> 3][2 * (j & 7) + 1]};
        const f32x2 a1 = f32x2{acc1[j >> 3][2 * (j & 7)], acc1[j >> 3][2 * (j & 7) + 1]};
        dp[j] = __builtin_elementwise_fma(f32x2{dj[j], dj[j]}, a0, f32x2{sh, sh});
        gp[j] = a1 * f32x2{mj[j], mj[j]};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(6);
    load_res_pass(0);                                                      // (the tables' registers are free now)
    if constexpr (P == 1) load_res_pass(1);

    // V[p][k]: value k of pass p; scratch offset of value k = lane part + vimm(k) floats
    float V[2][24];
    int wbase;                                                              // lane part of the scratch write offset (floats)
    // a wave's scratch = the 1 KiB pieces of stage 1 that its own late DMA instructions fill: piece p at float offset poff(p) from
    // STG + wave * 256 (4 waves: six activation pieces; 8 waves: three activation pieces + three of the dedicated region)
    auto poff = [](int p) { return NW == 8 ? (p < 3 ? p * 2048 : STG + (p - 3) * 2048) : p * 1024; };   // (8 waves: pieces 3..5 in the region behind the stages)
    auto vimm = [&](int k) {
      if constexpr (P == 1) { const int beta = k / 12, r = k % 12, c = 24 * beta + (r & 3) + 8 * (r >> 2); return poff(c >> 3) + (c & 7) * 32; }
      else return poff(k >> 3) + (k & 7) * 32;
    };
    if constexpr (P == 1) {
      // ---- 'f16' mode: the 24 x 24 adjacency mix on the matrix cores.  Per body: out[j][ch] = sum_k [Aoff | I][j][k] * [gp; dp][k][ch], K = 48 =
      // three k-steps.  B fragments: lane (ch, half h) must carry k-block 2s (h = 0) / 2s+1 (h = 1) of ONE body, but a lane owns all 48 k of
      // the two bodies of ITS half-wave: v_permlane32_swap of P = own block 2s with Q = own block 2s+1 leaves P = the lower half-wave's body,
      // Q = the upper one's.  ~300 VALU + 12 MFMA per wave instead of 1152 v_fmac.
      f32x16 D[2][2];                                            // [half-wave x][body beta of that half-wave's pair]
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 16; ++r) D[x][y][r] = 0.f;
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
        for (int beta = 0; beta < 2; ++beta) {
          u32x4_t Pw, Qw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kp = 16 * s3 + 2 * e, kq = kp + 8;           // < 24: gp[k], else dp[k - 24]
            const float p0 = kp < kJ ? gp[kp][beta] : dp[kp - kJ][beta], p1 = kp + 1 < kJ ? gp[kp + 1][beta] : dp[kp + 1 - kJ][beta];
            const float q0 = kq < kJ ? gp[kq][beta] : dp[kq - kJ][beta], q1 = kq + 1 < kJ ? gp[kq + 1][beta] : dp[kq + 1 - kJ][beta];
            const half2_t hp = {(half_t)p0, (half_t)p1};          // (MODE.FP16_OVFL: conversions saturate at +-65504, see run_tiles)
            const half2_t hq = {(half_t)q0, (half_t)q1};
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned int, hp), __builtin_bit_cast(unsigned int, hq), false, false);
            Pw[e] = sw[0];
            Qw[e] = sw[1];
          }
          D[0][beta] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s3], __builtin_bit_cast(half8, Pw), D[0][beta], 0, 0, 0);
          D[1][beta] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s3], __builtin_bit_cast(half8, Qw), D[1][beta], 0, 0, 0);
        }
      }
      // D[x][beta][r]: channel = mine, joint (r&3) + 8 (r>>2) + 4 g of body (half-wave x, beta); r >> 2 == 3 is padding
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int k = 0; k < 24; ++k) {
          const float v = D[x][k / 12][k % 12];
          V[x][k] = fmaxf(v, floor_v);
        }
      wbase = STG + wave * 256 + 128 * g + mi;                   // scratch row 24 beta + joint0 + 4 g
    } else {
      // ---- 'f16x3' mode: the mix on the VALU, exact f32 (coefficients in SGPRs through the constant address space)
      typedef const float __attribute__((address_space(4))) cfloat;
      constexpr int GR = 4;                                      // coefficient rows in flight: 96 SGPRs
#pragma unroll
      for (int j0 = 0; j0 < kJ; j0 += GR) {
        __builtin_amdgcn_sched_barrier(0);
        // an opaque copy of the pointer per group of rows bounds the number of coefficient rows hipcc keeps in SGPRs (it loaded all
        // 576 up front and spilled them)
        const unsigned long long ag = (unsigned long long)(uintptr_t)(io.Aoff + j0 * kJ);
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
          asm volatile("" : "+v"(s0), "+v"(s1));                // pins both bodies' chains here (hipcc sank body b's to its use in pass 1 and
          V[0][j] = fmaxf(s0, floor_v);                         // parked 500+ coefficients in VGPR lanes for it)
          V[1][j] = fmaxf(s1, floor_v);
        }
      }
      wbase = STG + wave * 256 + (NW == 8 ? STG : 3072) * g + mi;   // scratch row 24 g + joint: pieces 3 g + (joint >> 3) (8 waves: poff(3 + x) - poff(x) = STG)
      // M16: V[c][j] = channel n + 16 c, joint j of body rg = lane >> 4.  Scratch row of (joint 12 p + jj, body rg) = 4 jj + rg: piece jj >> 1, row
      // 4 (jj & 1) + rg of its eight; the two 16-channel halves of a row swap places for bodies 2, 3, so that the four lane groups of a write hit
      // four different 16-bank groups
      if constexpr (M16) wbase = STG + wave * 256 + (lane >> 4) * 32 + ((lane & 15) ^ (16 * (lane >> 5)));
    }
    TSTAMP(7);
    const int rbase = STG + wave * 256 + (lr & 7) * 32 + (M16 ? (c8 ^ (16 * ((lr & 3) >> 1))) : c8);   // item it: + piece 2 it + (lr >> 3)
    auto roff = [&](int it) { return rbase + ((lr >> 3) ? poff(2 * it + 1) : poff(2 * it)); };
    // range guard: the f16 conversions below SATURATE (MODE.FP16_OVFL) - |v| >= 65504 clamps the hi half (the X2 value is then good to ~1e-4
    // relative only, beyond 131008 it is lost; plain f16 rows lose it at once).  Nothing becomes inf / NaN, so nothing downstream would notice:
    // the tile's largest |v| (one v_max3_f32 per two stored values) raises bit 2 of the handle's status word instead
    float vmax = 0.f;
    const unsigned int wave_row0 = (unsigned int)cur.m_tile * 192u + 96u * (unsigned int)wm;
    const unsigned int valid_wave_rows = a.rows_valid > wave_row0 ? a.rows_valid - wave_row0 : 0u;   // rows of this wave's 96 that are real
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (P == 3) { if (p == 1) load_res_pass(1); }
      if constexpr (M16) {
#pragma unroll
        for (int jj = 0; jj < 12; ++jj) {
          lds[wbase + poff(jj >> 1) + (jj & 1) * 128] = V[0][12 * p + jj];
          lds[(wbase ^ 16) + poff(jj >> 1) + (jj & 1) * 128] = V[1][12 * p + jj];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 24; ++k) lds[wbase + vimm(k)] = V[p][k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private: program order is enough
      f32x4 t[3][2];
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        t[it][0] = *(const f32x4*)(lds + roff(it));
        t[it][1] = *(const f32x4*)(lds + roff(it) + 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        float v[8] = {t[it][0][0], t[it][0][1], t[it][0][2], t[it][0][3], t[it][1][0], t[it][1][1], t[it][1][2], t[it][1][3]};
        if (has_res) {
          if constexpr (P == 3) {
            const half8 rh = __builtin_bit_cast(half8, rq[2 * it]), rl = __builtin_bit_cast(half8, rq[2 * it + 1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += (float)rh[c] + (float)rl[c];
          } else {
            const half8 rh = __builtin_bit_cast(half8, rq[3 * p + it]);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += (float)rh[c];
          }
        }
        {
          float im = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
          im = fmaxf(im, fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
          vmax = fmaxf(vmax, item_vrow(p, it) < valid_wave_rows ? im : 0.f);      // (padding rows carry don't-care values: not looked at)
        }
        const unsigned int vo = item_vrow(p, it) * orow + col_out;
        if (out_f32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v[0], v[1], v[2], v[3]}), yB, vo, 0, kStoreAux);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v[4], v[5], v[6], v[7]}), yB, vo + 16u, 0, kStoreAux);
        } else {
          half8 hh, ll;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            hh[c] = (half_t)v[c];                                  // saturating (MODE.FP16_OVFL): |v| > 131008 saturates both halves, never inf
            if constexpr (P == 3) ll[c] = (half_t)(v[c] - (float)hh[c]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, hh), yB, vo, 0, kStoreAux);
          if constexpr (P == 3) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ll), yB, vo + 64u, 0, kStoreAux);
        }
      }
    }
    // (finite values only: an item whose INPUT was inf / NaN is made loud by the NaN rule of the output packer, for that item alone)
    if (!out_f32 && vmax >= 65504.f && vmax <= 3.0e38f) __hip_atomic_fetch_or(a.sticky, kStickySaturated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // stage 1's activation pieces are free again: fetch them for the next tile
    if constexpr (CHAIN) {
      if (a_issued) issue_late();
    }
    TSTAMP(4);

    if constexpr (!CHAIN) {
      break;
    } else {
      prev = cur;
      if (!have_next) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores have reached L2
        __syncthreads();
        publish(prev);
        t_cur = nt;
        break;
      }
      if (!a_issued) {                                        // producers were not complete yet (or the next tile is in a later layer)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        publish(prev);
        poll_deps(nxt);
        issue_a0();
        issue_late();
      } else {
        pending_publish = true;                               // bumped at the head of the next tile, behind its vmcnt(0) + barrier
      }
      cur = nxt;
      t_cur = nt;
    }
  }

  if constexpr (CHAIN) {
    // audit + reset: the last block to run out of tickets checks that every row tile of the last conv was produced by all its channel
    // tiles (a queue whose XCD received no block - CU masking - would otherwise go unnoticed), then zeroes the tickets and counters
    // for the NEXT launch on this handle (stream order): no memset node between the steps of a sampling loop.
    __syncthreads();
    if (tid == 0) {
      const unsigned int old = __hip_atomic_fetch_add(a.finished, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      slot[0] = old == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (slot[0]) {
      bool ok = true;
      for (int m = tid; m < a.m_tiles; m += 64 * NW)
        ok = ok && __hip_atomic_load(&a.done[(size_t)(a.nl - 1) * a.m_tiles + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)a.n_tiles;
      if (!ok) {
        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(a.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      for (int i = tid; i < a.nl * a.m_tiles; i += 64 * NW) a.done[i] = 0u;
      if (tid < 8) a.tickets[tid] = 0u;
      if (tid == 0) *a.finished = 0u;
    }
  }
}

template <int P>
__global__ __launch_bounds__(256, 2) void gcn_hidden_tile_kernel(OneArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * stage_floats(4)];   // 80 KiB, the only LDS object: 2 blocks per CU
  run_tiles<P, 0, 4>(lds, a);
}

template <int P, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void gcn_hidden_chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * stage_floats(NW) + (NW == 8 ? 24 * 256 : 0)];   // 80 KiB x 2 blocks, or 112 + 24 KiB x 1 block per CU
  run_tiles<P, 1, NW>(lds, a);
}

// float32 [rows, K] <-> plain f16 [rows, K]
__global__ void pack_half_kernel(const float* __restrict__ X, half_t* __restrict__ Y, size_t n, float scale, unsigned int* sticky = nullptr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = X[i] * scale;
  Y[i] = (half_t)fminf(fmaxf(v, -65504.f), 65504.f);
  if (sticky && fabsf(v) >= 65504.f && fabsf(v) <= 3.0e38f) __hip_atomic_fetch_or(sticky, kStickySaturated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void unpack_half_kernel(const half_t* __restrict__ X, float* __restrict__ Y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Y[i] = (float)X[i];
}
template <int G>
__global__ void pack_x2_kernel(const float* __restrict__ X, half_t* __restrict__ Y, int64_t rows, int K, unsigned int* sticky = nullptr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * K) return;
  split_store<G>(Y, i / K, (int)(i % K), K, X[i]);
  if (sticky && fabsf(X[i]) >= 65504.f && fabsf(X[i]) <= 3.0e38f) __hip_atomic_fetch_or(sticky, kStickySaturated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int G>
__global__ void unpack_x2_kernel(const half_t* __restrict__ X, float* __restrict__ Y, int64_t rows, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * K) return;
  Y[i] = split_load<G>(X, i / K, (int)(i % K), K);
}

bool shape_ok(const ehm_gcn* h, int64_t rows_pad) {
  const int P = h->precision == EHM_PREC_F16X3 ? 3 : 1;
  const int kt = P == 3 ? h->hid / 32 : h->hid / 64;
  if (h->hid % 64 != 0 || kt < 2 || rows_pad % 192 != 0) {
    ehm_set_error("f16 matrix-core convs need hid %% 64 == 0, hid >= %d and rows_pad %% 192 == 0 (hid = %d, rows_pad = %lld)", P == 3 ? 64 : 128,
                  h->hid, (long long)rows_pad);
    return false;
  }
  return true;
}

}  // namespace

#ifdef EHM_STAMPS
extern "C" int ehm_dbg_set(void* p) { unsigned long long* q = (unsigned long long*)p; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tdbg), &q, sizeof(q)); }
#endif

void ehm_pack_half(const float* X, void* Y, size_t n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(pack_half_kernel, dim3((unsigned)ceil_div((int64_t)n, 256)), dim3(256), 0, st, X, (half_t*)Y, n, scale, (unsigned int*)nullptr);
}

// One conv per launch (ehm_gcn_hidden_layer in the f16 modes).
int ehm_gcn_tile_layer_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_f32,
                            hipStream_t st) {
  if (!shape_ok(h, rows_pad)) return EHM_EINVAL;
  OneArgs a;
  a.L = h->hidden[layer];
  a.X = X;
  a.Res = residual;
  a.Y = out;
  a.m_tiles = (int)(rows_pad / 192);
  a.out_f32 = out_f32 ? 1 : 0;
  a.sticky = h->chain_sticky;
  a.rows_valid = (unsigned int)(h->valid_rows > 0 && h->valid_rows < rows_pad ? h->valid_rows : rows_pad);
  const int blocks = a.m_tiles * (h->hid / 64);
  if (h->precision == EHM_PREC_F16X3) hipLaunchKernelGGL(gcn_hidden_tile_kernel<3>, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gcn_hidden_tile_kernel<1>, dim3(blocks), dim3(256), 0, st, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

// All hidden convs of one GCN forward in ONE launch (ehm_gcn_hidden_stack).  Work item = (layer, row tile, channel tile); a conv's
// tile needs all channel tiles of the previous conv for the SAME 192 rows and nothing else, so the convs are chained per row tile
// with counters instead of kernel boundaries:
//   * every XCD owns the row tiles m = xcc (mod nq) of ALL layers (the block reads its own XCC_ID), so producer and consumer of a
//     row tile share one L2; each XCD has a ticket counter handing out its items in (layer, m, n) order - a consumer's producers
//     always hold smaller tickets, and a block never waits for a tile of its own or a later layer while one of its tiles is
//     unpublished, so waiting cannot deadlock whatever the residency;
//   * publish = sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> barrier -> one relaxed agent-scope add on
//     done[layer][m]; consume = done[layer-1][m] == n_tiles seen by one lane, barrier, then agent-scope (sc1) activation loads,
//     which never hit in the CU's L1.
// The sync words live in the handle (sized at ehm_gcn_create for max_rows; ehm_gcn_reserve grows them outside any capture).
int ehm_gcn_tile_chain_impl(ehm_gcn* h, void* const bufs[3], int64_t rows_pad, hipStream_t st) {
  const int nl = h->num_hidden;
  if (!shape_ok(h, rows_pad)) return EHM_EINVAL;
  if (nl < 2 || (nl & 1)) {
    ehm_set_error("chained hidden convs need an even number (>= 2) of them");
    return EHM_EINVAL;
  }
  const bool wide = h->precision != EHM_PREC_F16X3 && h->hid % 128 == 0;   // f16 mode: one 8-wave block per CU, 128-channel tiles
  const int m_tiles = (int)(rows_pad / 192), n_tiles = h->hid / (wide ? 128 : 64);
  const size_t need = 8 + (size_t)nl * m_tiles + 8;   // tickets | done | err, finished
  if (h->chain_sync_words < need) {
    const int rc = ehm_gcn_reserve_rows(h, rows_pad);
    if (rc != 0) return rc;
  }
  if (!h->chain_sync_clean || h->chain_sync_shape != (int64_t)nl * m_tiles) {   // otherwise the previous launch left the words zeroed
    EHM_HIP(hipMemsetAsync(h->chain_sync, 0, h->chain_sync_words * sizeof(unsigned int), st));
    h->chain_sync_shape = (int64_t)nl * m_tiles;
  }
  h->chain_sync_clean = 1;
  ChainArgs a;
  a.layers = h->hidden_dev;
  for (int i = 0; i < 3; ++i) a.buf[i] = bufs[i];
  a.nl = nl;
  a.m_tiles = m_tiles;
  a.n_tiles = n_tiles;
  a.tickets = h->chain_sync;
  a.done = h->chain_sync + 8;
  h->chain_err_off = 8 + (size_t)nl * m_tiles;
  a.err = h->chain_sync + h->chain_err_off;
  a.finished = a.err + 1;
  a.sticky = h->chain_sticky;
  a.rows_valid = (unsigned int)(h->valid_rows > 0 && h->valid_rows < rows_pad ? h->valid_rows : rows_pad);
  const int total = nl * m_tiles * n_tiles;
  int blocks = (wide ? 1 : 2) * ehm_num_cus();         // what is co-resident (80 KiB LDS per 4-wave block, 112 KiB per 8-wave block)
#ifdef EHM_STAMPS
  if (const char* e = getenv("EHM_CHAIN_BLOCKS")) blocks = atoi(e);   // stamp builds only: e.g. one block per CU to time a tile without a co-resident partner
#endif
  if (blocks > total) blocks = total;
  a.nq = ehm_num_cus() / 32;
  if (a.nq < 1) a.nq = 1;
  if (a.nq > 8) a.nq = 8;
  if (h->precision == EHM_PREC_F16X3) hipLaunchKernelGGL((gcn_hidden_chain_kernel<3, 4>), dim3(blocks), dim3(256), 0, st, a);
  else if (wide) hipLaunchKernelGGL((gcn_hidden_chain_kernel<1, 8>), dim3(blocks), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((gcn_hidden_chain_kernel<1, 4>), dim3(blocks), dim3(256), 0, st, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_pack_activations(const float* X, void* X2, int64_t rows, int K, int group, void* stream) {
  EHM_CHECK_ARG(X && X2 && rows > 0 && K > 0 && K % 32 == 0 && (group == 0 || group == 32));
  const dim3 grid((unsigned)ceil_div(rows * K, 256));
  if (group == 0) hipLaunchKernelGGL(pack_half_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, (size_t)rows * K, 1.f, (unsigned int*)nullptr);
  else hipLaunchKernelGGL(pack_x2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, rows, K, (unsigned int*)nullptr);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_pack_activations_checked(ehm_gcn* h, const float* X, void* X2, int64_t rows, void* stream) {
  EHM_CHECK_ARG(h && X && X2 && rows > 0 && h->precision != EHM_PREC_F32);
  h->valid_rows = rows;
  const int K = h->hid;
  const dim3 grid((unsigned)ceil_div(rows * K, 256));
  if (h->precision == EHM_PREC_F16) hipLaunchKernelGGL(pack_half_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, (size_t)rows * K, 1.f, h->chain_sticky);
  else hipLaunchKernelGGL(pack_x2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, X, (half_t*)X2, rows, K, h->chain_sticky);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_unpack_activations(const void* X2, float* X, int64_t rows, int K, int group, void* stream) {
  EHM_CHECK_ARG(X && X2 && rows > 0 && K > 0 && K % 32 == 0 && (group == 0 || group == 32));
  const dim3 grid((unsigned)ceil_div(rows * K, 256));
  if (group == 0) hipLaunchKernelGGL(unpack_half_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)X2, X, (size_t)rows * K);
  else hipLaunchKernelGGL(unpack_x2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)X2, X, rows, K);
  EHM_LAUNCH_CHECK();
  return 0;
}
