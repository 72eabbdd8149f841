// Fused linear layer on the f16 matrix cores with f32-grade accuracy (split-f16 "f16x3", see gcn_f16.hip):
//     Y = act_out( [act_in(A0) | A1] . W^T * (1/w_scale) + bias + group_bias[row / rows_per_group] )   (+ column max per group)
// Used for the scene PointNet of the conditioning path (models/respointnet.py:33-97: ResnetPointnet / ResnetBlockFC), where
// the reference runs ~40 eager torch kernels over [B,N,256..1024] float32 tensors per call.  Restructuring used by the host
// (egohmr_amd/encoders.py), all exact up to float re-association:
//   * the pooled half of every block input (`cat([net, pooled])`, respointnet.py:41-51) is constant per body, so its
//     contribution to fc_0 and to the shortcut is a per-body bias vector (group_bias) and the per-point GEMMs shrink to 256 wide;
//   * shortcut and fc_1 of a block accumulate into the same output, so they are ONE GEMM over the concatenated K
//     ([relu(h) | net] . [W1 | S]^T): the loader switches source pointer at K0 (dual-source A operand);
//   * the global max-pool over the N points is fused into the epilogue (wave shuffle -> LDS -> one atomic per column and block).
// Operands are in the X2 split format of gcn_dev.h (32 hi halves + 32 lo halves per 32-k group).  Tile 128 x 128 x 32,
// 4 waves (2x2, 64x64 each), global_load_lds double buffering, the same conflict-free XOR swizzle as the GCN kernels.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int LBM = 128, LBN = 128;
constexpr int L_STAGE = (LBM + LBN) * BK;   // floats: 32 KiB

struct LinArgs {
  const half_t* A0; const half_t* A1; const half_t* W;
  const float* bias; const float* gbias;
  half_t* Y; float* colmax;
  int K0, K1, M, N;
  int rows_per_group, valid_rows_per_group;
  int relu_in0, relu_out;
  float inv_scale;
};

__device__ __forceinline__ void relu_split(half8& hi, half8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool pos = hi[e] > (half_t)0;
    hi[e] = pos ? hi[e] : (half_t)0;
    lo[e] = pos ? lo[e] : (half_t)0;
  }
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

__global__ __launch_bounds__(256, 2) void linear_split_kernel(LinArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * L_STAGE];   // 64 KiB; the ONLY LDS object (a second one makes hipcc
                                                                    // drain vmcnt before every fragment read of the DMA pipeline)

  const int n_tiles = p.N / LBN, m_tiles = p.M / LBM;
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  const int lin = ((total & 7) == 0) ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;   // XCD b%8 owns a contiguous run of row tiles
  const int m_tile = lin / n_tiles, n_tile = lin % n_tiles;
  const size_t m0 = (size_t)m_tile * LBM;
  const int n0 = n_tile * LBN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int K = p.K0 + p.K1;

  // DMA: 4 wave-instructions of 8 rows per wave and operand; row r_i = r0 + 32 i -> one swizzle key
  const int ld_r = lane >> 3, ld_c = lane & 7;
  const int r0 = 8 * wave + ld_r;
  const int swz = (ld_c ^ ((r0 >> 1) & 7)) << 2;
  const float* pA0 = (const float*)p.A0 + (m0 + r0) * p.K0 + swz;
  const float* pA1 = p.K1 ? (const float*)p.A1 + (m0 + r0) * p.K1 + swz : nullptr;
  const float* pW = (const float*)p.W + ((size_t)n0 + r0) * K + swz;
  const int KT0 = p.K0 / BK, KT = K / BK;
  auto stage = [&](int buf, int kt) {
    float* base = lds + buf * L_STAGE;
    const bool second = kt >= KT0;
    const float* a = second ? pA1 + (size_t)(kt - KT0) * BK : pA0 + (size_t)kt * BK;
    const size_t arow = (size_t)32 * (second ? p.K1 : p.K0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(a + i * arow), (AS3 void*)(base + (wave + 4 * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(pW + (size_t)i * 32 * K + (size_t)kt * BK),
                                       (AS3 void*)(base + LBM * BK + (wave + 4 * i) * 256), 16, 0, 0);
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

  stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) stage((kt + 1) & 1, kt + 1);
    const float* As = lds + (kt & 1) * L_STAGE;
    const float* Bs = As + LBM * BK;
    const bool relu_a = p.relu_in0 && kt < KT0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = 2 * s + g, cl = 4 + 2 * s + g;
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *(const half8*)(As + (rA + 32 * t) * BK + ((ch ^ keyA) << 2));
        al[t] = *(const half8*)(As + (rA + 32 * t) * BK + ((cl ^ keyA) << 2));
        if (relu_a) relu_split(ah[t], al[t]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bh[u] = *(const half8*)(Bs + (rB + 32 * u) * BK + ((ch ^ keyB) << 2));
        bl[u] = *(const half8*)(Bs + (rB + 32 * u) * BK + ((cl ^ keyB) << 2));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
    }
  }

  // ---- epilogue.  Accumulator layout: lane = output column, registers = rows.  Bias / ReLU / the per-group column maximum are
  //      taken there; the values then cross a float [128][128] LDS tile so that the rows leave as 16-byte X2 stores (8 hi halves,
  //      8 lo halves) - with K = 256..512 a block has only 8-16 K tiles, and 64 dword stores per lane were most of its life time.
  __syncthreads();                                    // every wave is done reading the last K tile: the stages become the tile
  float* T = lds;
  const int group = (int)(m0 / p.rows_per_group);
  const int row_in_group0 = (int)(m0 % p.rows_per_group);
  float cmaxs[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int col = 64 * wn + 32 * u + mi;
    const int n = n0 + col;
    float add = p.bias ? p.bias[n] : 0.f;
    if (p.gbias) add += p.gbias[(size_t)group * p.N + n];
    float cmax = -3.4e38f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 64 * wm + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
        float v = fmaf(acc[t][u][r], p.inv_scale, add);
        if (p.relu_out) v = fmaxf(v, 0.f);
        T[row * LBN + col] = v;
        if (row_in_group0 + row < p.valid_rows_per_group) cmax = fmaxf(cmax, v);
      }
    }
    cmaxs[u] = fmaxf(cmax, __shfl_xor(cmax, 32));
  }
  __syncthreads();
  if (p.Y) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // work item = (row, 8 consecutive columns): 128 x 16 items, 8 per thread; items 8..15 of a row read their two 16-byte halves in
    // the opposite order, so that every ds_read_b128 lane group touches 16 distinct 16-byte slots of the 256-byte bank window
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int uu = tid + 256 * i, row = uu >> 4, c8 = uu & 15;
      const float* src = T + row * LBN + 8 * c8;
      const int flip = c8 >> 3;
      const f32x4 va = *(const f32x4*)(src + 4 * flip), vb = *(const f32x4*)(src + 4 * (1 - flip));
      const f32x4 v0 = flip ? vb : va, v1 = flip ? va : vb;
      const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      half8 hh, ll;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float c = fminf(fmaxf(v[k], -65504.f), 65504.f);
        hh[k] = (half_t)c;
        ll[k] = (half_t)fminf(fmaxf(v[k] - (float)hh[k], -65504.f), 65504.f);
      }
      half_t* dst = p.Y + split_off<32>(m0 + row, n0 + 8 * c8, p.N);   // 8 | 32: the eight hi halves are contiguous, the lo halves 32 further
      *(u32x4*)dst = __builtin_bit_cast(u32x4, hh);
      *(u32x4*)(dst + 32) = __builtin_bit_cast(u32x4, ll);
    }
  }
  if (p.colmax) {
    __syncthreads();                                  // the tile has been read: its memory carries the column-max exchange now
    float (*smax)[LBN] = (float (*)[LBN])lds;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (g == 0) smax[wm][64 * wn + 32 * u + mi] = cmaxs[u];
    __syncthreads();
    if (tid < LBN) atomic_max_float(p.colmax + (size_t)group * p.N + n0 + tid, fmaxf(smax[0][tid], smax[1][tid]));
  }
}

// relu(fc_pos(p)) in X2 format plus the raw points zero-padded to 32 columns (X2) for the folded stage-0 shortcut
__global__ void pointnet_lift_kernel(const float* __restrict__ pts, const float* __restrict__ Wpos, const float* __restrict__ bpos,
                                     half_t* __restrict__ R0, half_t* __restrict__ P32, int B, int N, int Npad, int C) {
  const size_t row = blockIdx.x;                       // b * Npad + i
  const int b = (int)(row / Npad), i = (int)(row % Npad);
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < N) {
    const float* q = pts + ((size_t)b * N + i) * 3;
    x = q[0]; y = q[1]; z = q[2];
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float v = fmaf(Wpos[c * 3 + 2], z, fmaf(Wpos[c * 3 + 1], y, fmaf(Wpos[c * 3], x, bpos[c])));   // torch addmm order is free
    split_store(R0, row, c, C, fmaxf(v, 0.f));
  }
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    split_store(P32, row, c, 32, c == 0 ? x : (c == 1 ? y : (c == 2 ? z : 0.f)));
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

// y = act(y + bias[c] (+ residual)), NCHW, in place; float4 when a quad never straddles a channel (HW % 4 == 0).
template <bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ res,
                                                       int64_t n, int C, int HW, int relu) {
  const int64_t stride = (int64_t)gridDim.x * 256 * (VEC ? 4 : 1);
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1); i < n; i += stride) {
    if (VEC) {
      const float b = bias[(i / HW) % C];
      f32x4 v = *(const f32x4*)(y + i);
      if (res) v += *(const f32x4*)(res + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] += b;
        if (relu) v[k] = fmaxf(v[k], 0.f);
      }
      *(f32x4*)(y + i) = v;
    } else {
      float v = y[i] + bias[(i / HW) % C];
      if (res) v += res[i];
      y[i] = relu ? fmaxf(v, 0.f) : v;
    }
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
  EHM_CHECK_ARG(d && d->A0 && d->W && (d->Y || d->colmax));
  EHM_CHECK_ARG(d->M > 0 && d->M % LBM == 0 && d->N > 0 && d->N % LBN == 0);
  EHM_CHECK_ARG(d->K0 > 0 && d->K0 % BK == 0 && d->K1 >= 0 && d->K1 % BK == 0 && (d->K1 == 0 || d->A1));
  EHM_CHECK_ARG(d->rows_per_group > 0 && d->rows_per_group % LBM == 0 && d->M % d->rows_per_group == 0);
  EHM_CHECK_ARG(d->w_scale > 0.f);
  LinArgs a;
  a.A0 = (const half_t*)d->A0; a.A1 = (const half_t*)d->A1; a.W = (const half_t*)d->W;
  a.bias = d->bias; a.gbias = d->group_bias; a.Y = (half_t*)d->Y; a.colmax = d->colmax;
  a.K0 = d->K0; a.K1 = d->K1; a.M = (int)d->M; a.N = d->N;
  a.rows_per_group = d->rows_per_group;
  a.valid_rows_per_group = d->valid_rows_per_group > 0 ? d->valid_rows_per_group : d->rows_per_group;
  a.relu_in0 = d->relu_in0; a.relu_out = d->relu_out; a.inv_scale = 1.f / d->w_scale;
  EHM_CHECK_ARG(d->M < (int64_t)1 << 31);
  const int blocks = (int)(d->M / LBM) * (d->N / LBN);
  hipLaunchKernelGGL(linear_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_pointnet_lift(const float* pts, const float* Wpos, const float* bpos, void* R0, void* P32, int B, int N, int N_padded,
                                 int C, void* stream) {
  EHM_CHECK_ARG(pts && Wpos && bpos && R0 && P32 && B > 0 && N > 0 && N_padded >= N && N_padded % LBM == 0 && C > 0 && C % 32 == 0);
  hipLaunchKernelGGL(pointnet_lift_kernel, dim3((unsigned)((size_t)B * N_padded)), dim3(256), 0, (hipStream_t)stream, pts, Wpos, bpos,
                     (half_t*)R0, (half_t*)P32, B, N, N_padded, C);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_bias_act(float* y, const float* bias, const float* residual, int64_t n, int C, int HW, int relu, void* stream) {
  if (n == 0) return 0;
  EHM_CHECK_ARG(y && bias && n > 0 && C > 0 && HW > 0 && n % ((int64_t)C * HW) == 0);
  const bool vec = HW % 4 == 0 && ((uintptr_t)y % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0);
  const int64_t work = vec ? n / 4 : n;
  int64_t blocks = ceil_div(work, 256);
  const int64_t cap = (int64_t)ehm_num_cus() * 16;       // grid-stride: enough waves to keep HBM busy, not one block per 1 KiB
  if (blocks > cap) blocks = cap;
  if (vec) hipLaunchKernelGGL((bias_act_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, residual, n, C, HW, relu);
  else hipLaunchKernelGGL((bias_act_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, residual, n, C, HW, relu);
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
