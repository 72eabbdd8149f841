// Device-side structures of the Modulated-GCN kernels, shared by gcn.hip (f32 MFMA) and gcn_tile.hip (f16 / split-f16 MFMA).
#pragma once
#include "common.h"
#include "egohmr_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int BM = 192;   // rows per block = 8 bodies x 24 joints
constexpr int BNH = 64;   // output channels per block (x2: W0 and W1 branch)
constexpr int BK = 32;    // K tile (128 B per row)
constexpr int A_TILE = BM * BK;        // floats
constexpr int B_TILE = 2 * BNH * BK;   // floats
constexpr int STAGE = A_TILE + B_TILE; // 10240 floats = 40 KiB

struct LayerDev {
  float* Wp;     // [N/64][128][K]   packed W0|W1 columns, K contiguous
  half_t* Ws;    // same tiles in split-f16 format X2<32> (below), values scaled by w_scale
  half_t* Wh;    // the same tiles as plain f16 (values scaled by w_scale): operands of the 'f16' mode (gcn_tile.hip)
  float* Ds;     // D / w_scale, transposed: [N][24] (tile engine only)
  float* M1s;    // M1 / w_scale, transposed: [N][24]
  float w_scale; // power of two
  float* D;      // [24][N]  A[j][j] * M[j][n] * scale[n]
  float* M1;     // [24][N]  M[j][n] * scale[n]
  float* shift;  // [N]      (bias - mean) * scale + beta      (bias when no BN)
  float* Aoff;   // [24][24] symmetrised adjacency, zero diagonal
  const half_t* AoffH;  // [Aoff | I] (24 x 48, rows padded to 32) as f16 MFMA A-operand fragments [k-step 3][lane 64][8 halves]: the adjacency mix
                        // of the 'f16' mode runs on the matrix cores (gcn_tile.hip)
  int K, N;
  int relu;
};

struct OutDev {
  float* Wt;    // [12][K]  rows 0-5: W0 columns, rows 6-11: W1 columns
  float* M;     // [24][6]
  float* A;     // [24][24] symmetrised adjacency (diagonal kept)
  float* bias;  // [6]
  int K;
};


enum { EHM_PREC_F32 = 0, EHM_PREC_F16X3 = 1, EHM_PREC_F16 = 2 };

struct ehm_gcn {
  int hid = 0;
  int num_hidden = 0;
  int precision = EHM_PREC_F32;
  const int32_t* mask_items = nullptr;   // exact pass pruning (ehm_gcn_set_pass_map): items that need the second pass, [num_masked] ...
  const int32_t* mask_slot = nullptr;    // ... and for every item its slot in that list or -1, [B]; nullptr = every item has a second pass
  int num_masked = -1;
  int uncond_masks_all = 0;              // second ("unconditional") pass of diffuse_fuse: 0 = image features masked, 1 = whole condition masked
  LayerDev input{};
  LayerDev hidden[16]{};
  LayerDev* hidden_dev = nullptr;        // device copy of hidden[] for the chained kernel (gcn_tile.hip)
  unsigned int* chain_sync = nullptr;    // tickets[8] | done[nl][m_tiles] | err | finished, zeroed before every chained launch
  int64_t valid_rows = 0;                // rows the last input conv / checked pack produced (0: unknown = all): what the epilogues' range guard looks at - the
                                         // tile padding behind them holds don't-care values (float32 rows of a last conv re-read as X2 halves, uninitialised scratch)
  unsigned int* chain_sticky = nullptr;  // one word, zeroed at create / by ehm_gcn_stack_status: accumulates the launches' err flags
  size_t chain_sync_words = 0;
  int chain_sync_clean = 0;              // 1: the last chained launch zeroed tickets / done / finished itself (its last block does)
  int64_t chain_sync_shape = 0;          // nl * m_tiles the words were last used with (err sits right behind done[])
  size_t chain_err_off = 0;              // word offset of err in chain_sync for the last launch
  int chain = 1;                         // ehm_gcn_hidden_stack: 1 = all hidden convs in one chained launch (f16 modes), 0 = one launch per conv (EHM_F16_CHAIN=0)
  OutDev out{};
  float* arena = nullptr;
  float* hs = nullptr;      // [hs_rows,12] responses of the output conv (gcn_out_dot_kernel -> gcn_out_mix_kernel)
  int64_t hs_rows = 0;
  int64_t reserved_rows = 0;   // rows_pad the sync words / hs scratch were sized for (ehm_gcn_reserve)
  ehm_nonlocal_params nonlocal{};   // optional non-local block of the one-call loop (ehm_gcn_set_nonlocal); Ci == 0: none
};

// Split-f16 activation / weight format ("X2<G>"): row-major rows of K values, every group of G consecutive k stored as
// G f16 "hi" followed by G f16 "lo" (value = hi + lo, hi = rn_f16(x), lo = rn_f16(x - hi)).  Same bytes per row as
// float32.  G = 32: one 128-byte line per 32-k tile.
template <int G = 32>
static __device__ __forceinline__ size_t split_off(size_t row, int n, int N) {
  return row * (size_t)N * 2 + (size_t)(n / G) * (2 * G) + (n % G);
}
template <int G = 32>
static __device__ __forceinline__ void split_store(half_t* base, size_t row, int n, int N, float v) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  const half_t hi = (half_t)c;
  const half_t lo = (half_t)fminf(fmaxf(v - (float)hi, -65504.f), 65504.f);   // |v| > 131008 saturates both halves (never inf)
  half_t* p = base + split_off<G>(row, n, N);
  p[0] = hi;
  p[G] = lo;
}
template <int G = 32>
static __device__ __forceinline__ float split_load(const half_t* base, size_t row, int n, int N) {
  const half_t* p = base + split_off<G>(row, n, N);
  return (float)p[0] + (float)p[G];
}

// Lane-pair versions for epilogues in which adjacent lanes own adjacent columns (lane parity == parity of n): the even
// lane moves the dword {hi[n], hi[n+1]}, the odd lane the dword {lo[n-1], lo[n]}, halves are exchanged with one
// DPP move.  One 4-byte access per lane and row instead of two 2-byte ones (the store tail is issue-bound).
static __device__ __forceinline__ unsigned int split_pack_bits(float v) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  const half_t hi = (half_t)c;
  const half_t lo = (half_t)fminf(fmaxf(v - (float)hi, -65504.f), 65504.f);
  return (unsigned int)__builtin_bit_cast(unsigned short, hi) | ((unsigned int)__builtin_bit_cast(unsigned short, lo) << 16);
}
static __device__ __forceinline__ unsigned int split_pair_word(float v, bool odd) {   // the dword split_store_pair writes
  const unsigned int own = split_pack_bits(v);
  const unsigned int nbr = dpp_xor1_u32(own);
  return odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
}
template <int G = 32>
static __device__ __forceinline__ void split_store_pair(half_t* base, size_t row, int n, int N, float v) {
  const unsigned int own = split_pack_bits(v);                       // {hi, lo} of my column
  const unsigned int nbr = dpp_xor1_u32(own);                         // {hi, lo} of the neighbouring column
  const bool odd = n & 1;
  // even lane: hi[n] | hi[n+1] << 16  at &hi[n];   odd lane: lo[n-1] | lo[n] << 16  at &lo[n-1]
  const unsigned int word = odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
  half_t* p = base + split_off<G>(row, n & ~1, N) + (odd ? G : 0);
  *(unsigned int*)p = word;
}
template <int G = 32>
static __device__ __forceinline__ unsigned int split_load_pair_raw(const half_t* base, size_t row, int n, int N) {
  const half_t* p = base + split_off<G>(row, n & ~1, N) + ((n & 1) ? G : 0);
  return *(const unsigned int*)p;                                    // even: {hi[n], hi[n+1]}   odd: {lo[n-1], lo[n]}
}
static __device__ __forceinline__ float split_pair_decode(unsigned int own, int n) {
  const bool odd = n & 1;
  const unsigned int nbr = dpp_xor1_u32(own);
  const unsigned short hb = odd ? (unsigned short)(nbr >> 16) : (unsigned short)(own & 0xffffu);
  const unsigned short lb = odd ? (unsigned short)(own >> 16) : (unsigned short)(nbr & 0xffffu);
  return (float)__builtin_bit_cast(half_t, hb) + (float)__builtin_bit_cast(half_t, lb);
}
template <int G = 32>
static __device__ __forceinline__ float split_load_pair(const half_t* base, size_t row, int n, int N) {
  const bool odd = n & 1;
  const half_t* p = base + split_off<G>(row, n & ~1, N) + (odd ? G : 0);
  const unsigned int own = *(const unsigned int*)p;                  // even: {hi[n], hi[n+1]}   odd: {lo[n-1], lo[n]}
  const unsigned int nbr = dpp_xor1_u32(own);
  const unsigned short hb = odd ? (unsigned short)(nbr >> 16) : (unsigned short)(own & 0xffffu);
  const unsigned short lb = odd ? (unsigned short)(own >> 16) : (unsigned short)(nbr & 0xffffu);
  return (float)__builtin_bit_cast(half_t, hb) + (float)__builtin_bit_cast(half_t, lb);
}

// d0[j] = D[j][n]*h0[j] + shift[n] (diagonal branch, bias and BatchNorm folded), g1[j] = M1[j][n]*h1[j];
// res[j] = residual input (already loaded, zeros when unused).
template <bool SPLIT_OUT, int G = 32>
static __device__ __forceinline__ void gcn_mix_store(const float (&d0)[kJ], const float (&g1)[kJ], const float (&res)[kJ], int n, int N,
                                              size_t row0, const float* __restrict__ Aoff, float* __restrict__ Y, bool relu) {
  // Aoff is wave-uniform and read-only, but hipcc cannot prove either through the by-value LayerDev and emitted 288
  // global_load_dwordx4 per wave for it; the constant address space makes them s_load (SGPR operands of the v_fmac).
  typedef const float __attribute__((address_space(4))) cfloat;
  const cfloat* Ac = (const cfloat*)(uintptr_t)Aoff;
#pragma unroll
  for (int j = 0; j < kJ; ++j) {
    float s = d0[j];
#pragma unroll
    for (int jp = 0; jp < kJ; ++jp) s = fmaf(Ac[j * kJ + jp], g1[jp], s);
    if (relu) s = fmaxf(s, 0.f);
    if (SPLIT_OUT) split_store_pair<G>((half_t*)Y, row0 + j, n, N, s + res[j]);
    else Y[(row0 + j) * (size_t)N + n] = s + res[j];
  }
}

// Two 24-joint bodies per lane at once (rows rowa.. and rowb..): joint j outermost, so one s_load of Aoff row j (24 SGPRs) feeds
// both bodies through v_pk_fma_f32 and only a few rows of coefficients are live (with one body per call hipcc kept all 576
// coefficients of the first call alive for the second and spilled SGPRs into VGPR lanes).  The residual is fetched four joints
// at a time right before the fmas that hide its latency.  Same fma order per output as gcn_mix_store.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// dp[j] / gp[j] = {body a, body b} values of joint j (diagonal branch incl. shift / off-diagonal branch): one v_pk_fma_f32 per
// coefficient.  Kernels whose accumulator layout already holds the two bodies in adjacent registers pass sub-vectors of the
// accumulators and pay no register moves (gcn_tile.hip).
template <class Store>
static __device__ __forceinline__ void gcn_mix2(const f32x2 (&dp)[kJ], const f32x2 (&gp)[kJ], const float* __restrict__ Aoff, bool relu,
                                                Store store) {
  typedef const float __attribute__((address_space(4))) cfloat;
  const cfloat* Ac = (const cfloat*)(uintptr_t)Aoff;
#pragma unroll
  for (int j0 = 0; j0 < kJ; j0 += 4) {
    __builtin_amdgcn_sched_barrier(0);           // bounds the number of coefficient rows hipcc keeps in SGPRs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i;
      float s0 = dp[j][0], s1 = dp[j][1];
#pragma unroll
      for (int jp = 0; jp < kJ; ++jp) {
        const float a = Ac[j * kJ + jp];
        s0 = fmaf(a, gp[jp][0], s0);
        s1 = fmaf(a, gp[jp][1], s1);
      }
      if (relu) { s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f); }
      store(j, s0, s1);                           // (joint, body a value, body b value) before the residual
    }
  }
}

// ------------------------------------------------------------------------------------------------
// input conv with the step-invariant projections hoisted (see ehm_gcn_input_layer in the header)
// one wave = one virtual body x 64 channels; 4 waves per block = 4 channel groups
// ------------------------------------------------------------------------------------------------
// Arguments of the hoisted input conv (gcn.hip: gcn_input_kernel; also run as part of smpl.hip's fused skinning + input-conv launch)
constexpr unsigned int kStickySaturated = 4u;   // bit of ehm_gcn::chain_sticky: an activation reached +-65504 in an f16 / X2 store and was clamped

struct GcnInputArgs {
  const float* h_img; const float* h_oth; const uint8_t* vis; const float* x; const float* Wx; const float* tvec;
  LayerDev L;
  float* Y;
  int B, passes, mask_all;
  const int32_t* mask_items;
  int total_vb, ny;          // grid of the standalone launch: total_vb x ny blocks of 256 threads
  unsigned int* sticky = nullptr;   // the handle's status word: bit 2 (kStickySaturated) when an f16 store of this conv clamped (OUT != 0)
  const float* pre = nullptr;   // PRE instantiation only (ehm_gcn_input_layer_rows): the rows' pre-activations x @ [W0 | W1] as [bodies * 24][2][N]
};

// One block of the input conv: virtual body `bx`, channels 256 * by .. + 255.  T = 24 * 256 floats of LDS.  `tid` = 0..255 (the calling
// 256 threads; a 512-thread block runs two of these side by side on two T regions), `xb_in` = the body's 144 x_t values when the caller has
// staged them itself (the one-launch loop reads them with agent-scope loads: another block of the SAME launch wrote them), else nullptr.
// PRE: the conv's two GEMM results come ready-made per row from a.pre (the general form of modulated_gcn_conv.py:39-41 for an arbitrary input feature: ModulatedGCN.forward
// standalone) instead of being assembled from the hoisted conditioning slices; everything behind them - modulation, adjacency mix, BN, ReLU, stores - is this function's.
template <int OUT, bool X_LDS = false, bool PRE = false>   // OUT: 0 = float32 rows, 1 = X2<32> split rows, 2 = plain f16 rows; X_LDS: xb_in is an LDS pointer (the caller staged x itself)
__device__ __forceinline__ void gcn_input_body(float* T, int tid, int bx, int by, const GcnInputArgs& a, const float* xb_in) {
  const float* __restrict__ h_img = a.h_img; const float* __restrict__ h_oth = a.h_oth; const uint8_t* __restrict__ vis = a.vis;
  const float* __restrict__ x = a.x; const float* __restrict__ Wx = a.Wx; const float* __restrict__ tvec = a.tvec;
  const LayerDev& L = a.L;
  float* __restrict__ Y = a.Y;
  const int B = a.B, passes = a.passes, mask_all = a.mask_all;
  const int32_t* __restrict__ mask_items = a.mask_items;
  (void)passes;
  const int N = L.N;
  const int vb = bx;                   // virtual body: [0, B) = conditional pass of item vb; B + k = second pass of item mask_items[k] (or k)
  const int p = vb >= B ? 1 : 0, b = p ? (mask_items ? mask_items[vb - B] : vb - B) : vb;
  const int n_raw = by * 256 + tid;
  const int n = n_raw < N ? n_raw : N - 1;                        // lanes past N recompute the last channel; their stores are dropped
  float base[2] = {0.f, 0.f}, img[2] = {0.f, 0.f}, wx[2][6] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  if constexpr (!PRE)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    base[k] = ((p == 1 && mask_all) ? 0.f : h_oth[((size_t)b * 2 + k) * N + n]) + tvec[k * N + n];   // egohmr.py:150-158 force_mask: image part only / whole condition
    img[k] = (p == 0) ? h_img[((size_t)b * 2 + k) * N + n] : 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) wx[k][c] = Wx[(k * 6 + c) * N + n];
  }
  // x_t of the body: wave-uniform.  Standalone launches: scalar loads from the global row.  X_LDS: 36 sixteen-byte LDS reads through an explicit
  // LDS pointer (as `xb_in ? xb_in : global row` the pointer was generic and the reads FLAT loads: aperture check per access, both wait counters)
  const float* xb = xb_in ? xb_in : (PRE ? nullptr : x + (size_t)b * kPoseDim);
  float xr[X_LDS ? kPoseDim : 1];
  if constexpr (X_LDS) {
    typedef const f32x4 __attribute__((address_space(3))) lf32x4;
    lf32x4* xl = (lf32x4*)xb_in;
#pragma unroll
    for (int i = 0; i < kPoseDim / 4; ++i) {
      const f32x4 v = xl[i];
      xr[4 * i] = v[0]; xr[4 * i + 1] = v[1]; xr[4 * i + 2] = v[2]; xr[4 * i + 3] = v[3];
    }
  }
  const uint8_t* vb_ = PRE ? nullptr : vis + (size_t)b * kJ;
  float h0[kJ], h1[kJ];
  const float sh = L.shift[n];
  // the channel's 24 + 24 table values from the [N][24] copies: twelve 16-byte loads instead of 48 strided dword loads (a unit's vector-memory
  // instructions: 1280 bodies 303 -> 254 us per fused step launch when the second pass found them in registers - measured with the tables held
  // across both passes at the cost of 88 spilled registers; as 16-byte loads they are cheap enough to fetch per unit)
  float dj[kJ], mj[kJ];
  {
    typedef unsigned int u32x4_tb __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t dsB = ehm_buffer_rsrc(L.Ds), m1B = ehm_buffer_rsrc(L.M1s);   // (buffer form: as plain pointers out of the argument struct these became flat loads)
    const unsigned int nrow = (unsigned int)n * (unsigned int)(kJ * 4);
#pragma unroll
    for (int q4 = 0; q4 < kJ / 4; ++q4) {
      const u32x4_tb d4 = __builtin_amdgcn_raw_buffer_load_b128(dsB, nrow, 16 * q4, 0);
      const u32x4_tb m4 = __builtin_amdgcn_raw_buffer_load_b128(m1B, nrow, 16 * q4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned int du = d4[i], mu = m4[i];       // (hipcc: __builtin_bit_cast of a vector ELEMENT expression reads element 0)
        dj[4 * q4 + i] = __builtin_bit_cast(float, du);
        mj[4 * q4 + i] = __builtin_bit_cast(float, mu);
      }
    }
  }
  // the two branches' sums as the halves of packed FMAs (v_pk_fma_f32: the same fused operations per half, two per instruction - this
  // kernel is bound by vector-ALU issue: ~940 scalar FMAs per lane before, 4 cycles apiece for a wave)
  const f32x2 img2 = {img[0], img[1]}, base2 = {base[0], base[1]};
  f32x2 wx2[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) wx2[c] = f32x2{wx[0][c], wx[1][c]};
#pragma unroll
  for (int j = 0; j < kJ; ++j) {
    f32x2 s;
    if constexpr (PRE) {
      const float* pr = a.pre + ((size_t)vb * kJ + j) * 2 * (size_t)N + n;
      s = f32x2{pr[0], pr[N]};
    } else {
    const float v = vb_[j] ? 1.f : 0.f;
    s = __builtin_elementwise_fma(f32x2{v, v}, img2, base2);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float xv = X_LDS ? xr[X_LDS ? j * 6 + c : 0] : xb[j * 6 + c];
      s = __builtin_elementwise_fma(f32x2{xv, xv}, wx2[c], s);
    }
    }
    h0[j] = fmaf(dj[j], s[0], sh);
    h1[j] = mj[j] * s[1];
  }
  // 24x24 adjacency mix per lane (one channel), then through a float [24][256] LDS tile so that the rows leave as 16-byte stores
  // (one dword per lane and joint is store-issue bound: 30 us for 48 MiB).
  if constexpr (OUT == 2) {
    // 'f16' mode: the mix on the matrix cores, [Aoff | I] (24 x 48) x [h1; h0] (48 x channels) like the hidden convs' epilogue (gcn_tile.hip).
    // A lane owns all 48 k of ONE channel; v_permlane32_swap of its k-blocks 2s / 2s+1 yields the B fragments of the wave's lower 32
    // channels (P) and upper 32 channels (Q).  6 MFMA + ~100 VALU per wave instead of 576 v_fmac per lane.
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const int lane = tid & 63, wv = tid >> 6;
    f32x16 DA, DB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { DA[r] = 0.f; DB[r] = 0.f; }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const half8 af = ((const half8*)L.AoffH)[s3 * 64 + lane];
      u32x4_t Pw, Qw;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kp = 16 * s3 + 2 * e, kq = kp + 8;             // < 24: h1[k], else h0[k - 24]
        const float p0 = kp < kJ ? h1[kp] : h0[kp - kJ], p1 = kp + 1 < kJ ? h1[kp + 1] : h0[kp + 1 - kJ];
        const float q0 = kq < kJ ? h1[kq] : h0[kq - kJ], q1 = kq + 1 < kJ ? h1[kq + 1] : h0[kq + 1 - kJ];
        // (+ 0 * v: an inf / NaN operand stays NaN through the clamp, see the range-guard note at the float path's store below)
        const half2_t hp = {(half_t)fmaf(p0, 0.f, fminf(fmaxf(p0, -65504.f), 65504.f)), (half_t)fmaf(p1, 0.f, fminf(fmaxf(p1, -65504.f), 65504.f))};
        const half2_t hq = {(half_t)fmaf(q0, 0.f, fminf(fmaxf(q0, -65504.f), 65504.f)), (half_t)fmaf(q1, 0.f, fminf(fmaxf(q1, -65504.f), 65504.f))};
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned int, hp), __builtin_bit_cast(unsigned int, hq), false, false);
        Pw[e] = sw[0];
        Qw[e] = sw[1];
      }
      DA = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(half8, Pw), DA, 0, 0, 0);
      DB = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(half8, Qw), DB, 0, 0, 0);
    }
    const bool relu = L.relu != 0;
    const int cA = 64 * wv + (lane & 31), half = lane >> 5;
#pragma unroll
    for (int r = 0; r < 12; ++r) {                               // D row = (r&3) + 8 (r>>2) + 4 half = joint; r >> 2 == 3 is padding
      const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float va = relu ? fmaxf(DA[r], 0.f) : DA[r], vb = relu ? fmaxf(DB[r], 0.f) : DB[r];
      T[j * 256 + cA] = fmaf(DA[r], 0.f, va);                       // (non-finite stays NaN through the ReLU and the saturating store: see below)
      T[j * 256 + cA + 32] = fmaf(DB[r], 0.f, vb);
    }
  } else {
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* Ac = (const cfloat*)(uintptr_t)L.Aoff;
    const bool relu = L.relu != 0;
    // two output joints per packed FMA: the coefficient pair (Aoff[j][jp], Aoff[j+1][jp]) = (Aoff[jp][j], Aoff[jp][j+1]) - the matrix is exactly
    // symmetric (gcn.hip: (a_ij + a_ji) / 2) - sits in an aligned SGPR pair of row jp's scalar load; jp outermost (one row of 24 coefficients
    // live at a time - with j outermost the strided pairs spilled ~1000 SGPRs through v_writelane / v_readlane); every output's FMA chain
    // still runs over jp in ascending order: bit-equal to the scalar form
    f32x2 sacc[kJ / 2];
#pragma unroll
    for (int jj = 0; jj < kJ / 2; ++jj) sacc[jj] = f32x2{h0[2 * jj], h0[2 * jj + 1]};
#pragma unroll
    for (int jp = 0; jp < kJ; ++jp) {
      const f32x2 hh = {h1[jp], h1[jp]};
#pragma unroll
      for (int jj = 0; jj < kJ / 2; ++jj) sacc[jj] = __builtin_elementwise_fma(f32x2{Ac[jp * kJ + 2 * jj], Ac[jp * kJ + 2 * jj + 1]}, hh, sacc[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < kJ / 2; ++jj) {
      // An inf / NaN in the body's x_t (a non-finite noise draw: the item's outputs are NaN by the packer's rule) must not come out of the saturating
      // f16 stores below as a FINITE +-65504: the hidden convs would turn that into large finite activations and raise the range guard for an item that
      // is already accounted for.  v + 0 * v_pre_relu is v for finite values and NaN otherwise (f16 modes only; one FMA per value); the first hidden
      // conv's ReLU then maps the NaN rows to 0.
      f32x2 o = sacc[jj];
      if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
      if constexpr (OUT != 0) o = __builtin_elementwise_fma(sacc[jj], f32x2{0.f, 0.f}, o);
      T[(2 * jj) * 256 + tid] = o[0];
      T[(2 * jj + 1) * 256 + tid] = o[1];
    }
  }
  __syncthreads();
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int nb = by * 256;                               // first channel of this block (N % 256 may leave a partial block)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = tid + 256 * i, j = u >> 5, c8 = (u & 31) * 8;   // (joint, 8 consecutive channels)
    if (nb + c8 >= N) continue;

    const f32x4 v0 = *(const f32x4*)(T + j * 256 + c8), v1 = *(const f32x4*)(T + j * 256 + c8 + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const size_t row = (size_t)vb * kJ + j;
    if (OUT != 0) {
      if (a.sticky) {                                     // range guard of the f16 stores (see run_tiles' epilogue, gcn_tile.hip)
        float vm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k += 2) vm = fmaxf(vm, fmaxf(fabsf(v[k]), fabsf(v[k + 1])));
        if (vm >= 65504.f && vm <= 3.0e38f) __hip_atomic_fetch_or(a.sticky, kStickySaturated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (finite only: inf / NaN inputs have their own rule)
      }
      half8 hh, ll;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float c = fmaf(v[k], 0.f, fminf(fmaxf(v[k], -65504.f), 65504.f));   // (+ 0 * v: inf / NaN stay NaN - the clamp alone would store +-65504)
        hh[k] = (half_t)c;
        ll[k] = (half_t)fminf(fmaxf(v[k] - (float)hh[k], -65504.f), 65504.f);
      }
      if (OUT == 1) {
        half_t* p = (half_t*)Y + split_off<32>(row, nb + c8, N);   // the eight hi halves are contiguous, the lo halves 32 further
        *(u32x4*)p = __builtin_bit_cast(u32x4, hh);
        *(u32x4*)(p + 32) = __builtin_bit_cast(u32x4, ll);
      } else {
        *(u32x4*)((half_t*)Y + row * (size_t)N + nb + c8) = __builtin_bit_cast(u32x4, hh);
      }
    } else {
      float* p = Y + row * (size_t)N + nb + c8;
      *(f32x4*)p = v0;
      *(f32x4*)(p + 4) = v1;
    }
  }
}
template <int OUT>
__device__ __forceinline__ void gcn_input_body(float* T, int bx, int by, const GcnInputArgs& a) {
  gcn_input_body<OUT>(T, (int)threadIdx.x, bx, by, a, nullptr);
}

// ---- output conv responses of 16 rows (gcn.hip: gcn_out_dot_kernel) --------------
constexpr int OUT_ROWS_PER_BLOCK = 16;
// [rows, K] x [K, 12] on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: 16 rows x 16 columns, 12 used).  256 threads = 4 waves split K; a
// wave's lane (row = l&15, q = l>>4) streams float4 X[row][kw + 16 i + 4 q ..+3] - the k order inside an MFMA step is a
// permutation applied to both operands, which a sum does not see - and the matching float4 of the 12 x K weights (48 KiB, L2 hits).
// The four partial 16x16 tiles meet in 4 KiB of LDS (`part`).  AUX = cache policy of the activation loads (0: plain; 16 = sc1).
// One wave's share of a 16-row tile: K quarter `wave` of row `r` (lane: row = l & 15, q = l >> 4) -> the wave's partial 16 x 16 tile in MFMA C
// layout (column = lane & 15, row = 4 (lane >> 4) + reg).  Shared by gcn_out_dot_rows16 and the fused step kernel (step.hip).
template <bool HALF_IN, int AUX>
__device__ __forceinline__ f32x4 gcn_out_dot_quarter(const float* __restrict__ X, const OutDev& O, int64_t r, int wave, int lane) {
  const int K = O.K;
  const int row = lane & 15, q = lane >> 4;
  const int kq = K / 4;                                              // this wave's K range (hid % 64 == 0: a multiple of 16)
  // a lane owns 8 consecutive k of every 32-k group (16 bytes of f16 / 32 bytes of float32 per load: the four lanes of a row cover a
  // 64 / 128-byte segment); MFMA c of the group contracts element c of all lanes, i.e. k = c, 8 + c, 16 + c, 24 + c
  // Both operands in buffer form (descriptor in SGPRs, one 32-bit lane offset, the k step as an immediate): as plain pointers taken out of the
  // kernel-argument structs the loads were FLAT instructions (address-space check per lane, both wait counters).  AUX != 0: the rows were
  // written by other blocks of THIS launch - cache-bypassing loads.
  typedef unsigned int u32x4_od __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2_od __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsX = ehm_buffer_rsrc_4g(X), rsW = ehm_buffer_rsrc(O.Wt);
  const unsigned int vox = (unsigned int)((r * K + (size_t)wave * kq + 8 * q) * (HALF_IN ? 2 : 4));
  const unsigned int vow = (unsigned int)(((size_t)(row < 12 ? row : 0) * K + (size_t)wave * kq + 8 * q) * 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains: the dependent-accumulator latency (40 cyc) exceeds the issue interval
#pragma unroll 4
  for (int k = 0; k + 32 <= kq; k += 32) {
    float xv[8];
    if (HALF_IN) {
      const half8 hv = __builtin_bit_cast(half8, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsX, vox, k * 2, AUX));
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = (float)hv[c];
    } else {
      const f32x4 x0 = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsX, vox, k * 4, AUX));
      const f32x4 x1 = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsX, vox, k * 4 + 16, AUX));
#pragma unroll
      for (int c = 0; c < 4; ++c) { xv[c] = x0[c]; xv[4 + c] = x1[c]; }
    }
    f32x4 w0 = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsW, vow, k * 4, 0));
    f32x4 w1 = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsW, vow, k * 4 + 16, 0));
    if (row >= 12) { w0 = f32x4{0.f, 0.f, 0.f, 0.f}; w1 = w0; }
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c], w0[c], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c + 1], w0[c + 1], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[4 + c], w1[c], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[5 + c], w1[c + 1], acc2, 0, 0, 0);
    }
  }
  if (kq & 16) {                                                      // hid % 128 != 0: one last 16-k group, four k per lane
    const int k = kq - 16 - 4 * q;                                    // (undo the 8 q of the lane offsets: this group's lane stride is 4)
    f32x4 xv;
    if (HALF_IN) {
      typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
      const half4_t hv = __builtin_bit_cast(half4_t, (u32x2_od)__builtin_amdgcn_raw_buffer_load_b64(rsX, vox + (unsigned int)(k * 2), 0, AUX));
      xv = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    } else {
      xv = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsX, vox + (unsigned int)(k * 4), 0, AUX));
    }
    f32x4 wv = __builtin_bit_cast(f32x4, (u32x4_od)__builtin_amdgcn_raw_buffer_load_b128(rsW, vow + (unsigned int)(k * 4), 0, 0));
    if (row >= 12) wv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c], wv[c], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c + 1], wv[c + 1], acc2, 0, 0, 0);
    }
  }
  acc += acc2;
  return acc;
}

template <bool HALF_IN, int AUX>
__device__ __forceinline__ void gcn_out_dot_rows16(const float* __restrict__ X, const OutDev& O, float* __restrict__ hs, int64_t r0, int64_t rows,
                                                   float (*part)[16][16], int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int row = lane & 15, q = lane >> 4;
  const int64_t r = r0 + row < rows ? r0 + row : rows - 1;          // tail block: clamp the load, drop the store
  // (the quarter's lane offsets are 32-bit: hand it the matrix rebased at this block's first row)
  const float* Xb = (const float*)((const char*)X + (size_t)r0 * O.K * (HALF_IN ? 2 : 4));
  const f32x4 acc = gcn_out_dot_quarter<HALF_IN, AUX>(Xb, O, r - r0, wave, lane);
  // C layout of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int c = 0; c < 4; ++c) part[wave][4 * q + c][row] = acc[c];
  __syncthreads();
  if (tid < 16 * 12) {
    const int rr = tid / 12, cc = tid % 12;
    if (r0 + rr < rows) hs[(r0 + rr) * 12 + cc] = (part[0][rr][cc] + part[1][rr][cc]) + (part[2][rr][cc] + part[3][rr][cc]);
  }
}
