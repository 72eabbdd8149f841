// Device-side structures of the Modulated-GCN kernels, shared by gcn.hip (f32 MFMA) and gcn_f16.hip (split-f16 MFMA).
#pragma once
#include "common.h"

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
  half_t* Ws;    // same tiles in split-f16 format (see gcn_f16.hip), values scaled by w_scale
  float* Ds;     // D / w_scale
  float* M1s;    // M1 / w_scale
  float w_scale; // power of two
  float* D;      // [24][N]  A[j][j] * M[j][n] * scale[n]
  float* M1;     // [24][N]  M[j][n] * scale[n]
  float* shift;  // [N]      (bias - mean) * scale + beta      (bias when no BN)
  float* Aoff;   // [24][24] symmetrised adjacency, zero diagonal
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
  int reg_staging = 0;     // split-f16 convs: 1 = global_load -> VGPR -> ds_write staging, 0 = global_load_lds DMA
  int persistent = 0;      // split-f16 convs: 1 = grid capped at the co-resident slots, blocks loop over tiles
  int tile_override = 0;   // split-f16 convs: 0 = pick by size, 1 = 192x64 tiles, 2 = 384x128 tiles
  LayerDev input{};
  LayerDev hidden[16]{};
  OutDev out{};
  float* arena = nullptr;
  float* hs = nullptr;      // [hs_rows,12] responses of the output conv (gcn_out_dot_kernel -> gcn_out_mix_kernel)
  int64_t hs_rows = 0;
};

// Split-f16 activation / weight format ("X2"): row-major rows of K values, each group of 32 consecutive k stored as
// 32 f16 "hi" followed by 32 f16 "lo" (128 bytes, value = hi + lo, hi = rn_f16(x), lo = rn_f16(x - hi)).
// Same bytes per row as float32, and every 32-k tile of a row is one 128-byte line.
static __device__ __forceinline__ void split_store(half_t* base, size_t row, int n, int N, float v) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  const half_t hi = (half_t)c;
  const half_t lo = (half_t)(v - (float)hi);
  half_t* p = base + row * (size_t)N * 2 + (size_t)(n >> 5) * 64 + (n & 31);
  p[0] = hi;
  p[32] = lo;
}
static __device__ __forceinline__ float split_load(const half_t* base, size_t row, int n, int N) {
  const half_t* p = base + row * (size_t)N * 2 + (size_t)(n >> 5) * 64 + (n & 31);
  return (float)p[0] + (float)p[32];
}

// Lane-pair versions for epilogues in which adjacent lanes own adjacent columns (lane parity == parity of n): the even
// lane moves the dword {hi[n], hi[n+1]}, the odd lane the dword {lo[n-1], lo[n]}, halves are exchanged with one
// cross-lane move.  One 4-byte access per lane and row instead of two 2-byte ones (the store tail is issue-bound).
static __device__ __forceinline__ unsigned int split_pack_bits(float v) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  const half_t hi = (half_t)c;
  const half_t lo = (half_t)(v - (float)hi);
  return (unsigned int)__builtin_bit_cast(unsigned short, hi) | ((unsigned int)__builtin_bit_cast(unsigned short, lo) << 16);
}
static __device__ __forceinline__ void split_store_pair(half_t* base, size_t row, int n, int N, float v) {
  const unsigned int own = split_pack_bits(v);                       // {hi, lo} of my column
  const unsigned int nbr = dpp_xor1_u32(own);                         // {hi, lo} of the neighbouring column
  const bool odd = n & 1;
  // even lane: hi[n] | hi[n+1] << 16  at &hi[n];   odd lane: lo[n-1] | lo[n] << 16  at &lo[n-1]
  const unsigned int word = odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
  half_t* p = base + row * (size_t)N * 2 + (size_t)(n >> 5) * 64 + (n & 30) + (odd ? 32 : 0);
  *(unsigned int*)p = word;
}
static __device__ __forceinline__ float split_load_pair(const half_t* base, size_t row, int n, int N) {
  const bool odd = n & 1;
  const half_t* p = base + row * (size_t)N * 2 + (size_t)(n >> 5) * 64 + (n & 30) + (odd ? 32 : 0);
  const unsigned int own = *(const unsigned int*)p;                  // even: {hi[n], hi[n+1]}   odd: {lo[n-1], lo[n]}
  const unsigned int nbr = dpp_xor1_u32(own);
  const unsigned short hb = odd ? (unsigned short)(nbr >> 16) : (unsigned short)(own & 0xffffu);
  const unsigned short lb = odd ? (unsigned short)(own >> 16) : (unsigned short)(nbr & 0xffffu);
  return (float)__builtin_bit_cast(half_t, hb) + (float)__builtin_bit_cast(half_t, lb);
}

// d0[j] = D[j][n]*h0[j] + shift[n] (diagonal branch, bias and BatchNorm folded), g1[j] = M1[j][n]*h1[j];
// res[j] = residual input (already loaded, zeros when unused).
template <bool SPLIT_OUT>
static __device__ __forceinline__ void gcn_mix_store(const float (&d0)[kJ], const float (&g1)[kJ], const float (&res)[kJ], int n, int N,
                                              size_t row0, const float* __restrict__ Aoff, float* __restrict__ Y, bool relu) {
#pragma unroll
  for (int j = 0; j < kJ; ++j) {
    float s = d0[j];
#pragma unroll
    for (int jp = 0; jp < kJ; ++jp) s = fmaf(Aoff[j * kJ + jp], g1[jp], s);  // Aoff: wave-uniform -> scalar loads
    if (relu) s = fmaxf(s, 0.f);
    if (SPLIT_OUT) split_store_pair((half_t*)Y, row0 + j, n, N, s + res[j]);
    else Y[(row0 + j) * (size_t)N + n] = s + res[j];
  }
}

