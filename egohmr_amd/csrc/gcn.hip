// Modulated-GCN denoiser for gfx950 (MI355X): the per-step part of
//   ModulatedGCN.forward           models/egohmr/modulated_gcn/modulated_gcn.py:99-116
//   _GraphConv / _ResGraphConv     modulated_gcn.py:21-28, :38-42
//   ModulatedGraphConv.forward     models/egohmr/modulated_gcn/modulated_gcn_conv.py:39-50
// of the reference, as called from EgoHMR.forward (models/egohmr/egohmr.py:236-254).
//
// Design (not a translation - the reference is ~12 eager torch ops per conv):
//   * one kernel per hid->hid conv: f32 MFMA GEMM  [rows,K] x [K, W0|W1]  with the whole
//     "M (.) h -> 24x24 adjacency mix -> +bias -> BatchNorm(eval) -> ReLU -> (+residual)" epilogue done
//     IN REGISTERS.  The MFMA row->body/joint map is chosen so that after the K loop every lane owns
//     all 24 joints of two bodies for one output channel (v_mfma_f32_32x32x2_f32 C layout:
//     col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); the adjacency mix is then 24x24
//     scalar-broadcast FMAs per body with no LDS / cross-lane traffic.
//   * weights are re-packed once ([n_tile][W0 cols | W1 cols][K], K contiguous) so both MFMA operands
//     are "row = m or n, contiguous k" and stream through LDS with 16-byte global_load_lds DMA,
//     double buffered, one barrier per 32-wide K tile; LDS images are XOR-swizzled on the SOURCE
//     address so the ds_read_b128 fragment reads are bank-conflict free.
//   * BatchNorm is folded into per-channel scale/shift, the adjacency is symmetrised once.
#include <stdlib.h>

#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// parameter packing (runs once per model)
// ------------------------------------------------------------------------------------------------
__global__ void pack_w_kernel(const float* __restrict__ W, float* __restrict__ Wp, int K, int N) {
  // W [2][K][N] -> Wp[nt][u*64 + c][k] = W[u][k][nt*64 + c]; 32x32 LDS transpose tiles
  __shared__ float tile[32][33];
  int u = blockIdx.z;
  int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    int k = k0 + r, n = n0 + tx;
    tile[r][tx] = (k < K && n < N) ? W[((size_t)u * K + k) * N + n] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int n = n0 + r, k = k0 + tx;
    if (n < N && k < K) {
      int nt = n >> 6, c = n & 63;
      Wp[((size_t)nt * 128 + u * 64 + c) * K + k] = tile[tx][r];
    }
  }
}

__global__ void pack_epilogue_kernel(const float* __restrict__ adj, ehm_gconv_params p, float* D, float* M1, float* shift,
                                     float* Aoff) {
  int N = p.out_dim;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < kJ * kJ) {
    int i = threadIdx.x / kJ, j = threadIdx.x % kJ;
    float aij = adj[i * kJ + j] + p.adj2[i * kJ + j];
    float aji = adj[j * kJ + i] + p.adj2[j * kJ + i];
    Aoff[i * kJ + j] = (i == j) ? 0.f : (aji + aij) / 2;   // (adj.T + adj)/2, modulated_gcn_conv.py:44
  }
  if (n >= N) return;
  float scale = 1.f, sh = p.bias ? p.bias[n] : 0.f;
  if (p.bn_weight) {
    scale = p.bn_weight[n] / sqrtf(p.bn_var[n] + 1e-5f);   // BatchNorm1d eval, eps = 1e-5
    sh = (sh - p.bn_mean[n]) * scale + p.bn_bias[n];
  }
  shift[n] = sh;
  for (int j = 0; j < kJ; ++j) {
    const float ajj = adj[j * kJ + j] + p.adj2[j * kJ + j];   // diagonal of (A.T + A)/2
    float m = p.M[j * N + n];
    D[j * N + n] = ajj * m * scale;
    M1[j * N + n] = m * scale;
  }
}

__global__ void pack_out_kernel(const float* __restrict__ adj, ehm_gconv_params p, float* Wt, float* M, float* A, float* bias) {
  int K = p.in_dim;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 12 * K) {
    int r = t / K, k = t % K;
    int u = r / 6, c = r % 6;
    Wt[t] = p.W[((size_t)u * K + k) * 6 + c];
  }
  if (t < kJ * kJ) {
    int i = t / kJ, j = t % kJ;
    A[t] = ((adj[j * kJ + i] + p.adj2[j * kJ + i]) + (adj[i * kJ + j] + p.adj2[i * kJ + j])) / 2;
  }
  if (t < kJ * 6) M[t] = p.M[t];
  if (t < 6) bias[t] = p.bias ? p.bias[t] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// shared epilogue: one lane = one output channel n of one body; h0/h1 = the 24 joints' W0/W1 responses
// ------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ w, size_t n, unsigned int* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bit patterns
}

// Wp (f32, packed tiles) -> X2 split-f16 tiles scaled by `scale`; Ds/M1s = D/scale, M1/scale
template <int G>
__global__ void pack_ws_kernel(const float* __restrict__ Wp, half_t* __restrict__ Ws, size_t rows, int K, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  split_store<G>(Ws, i / K, (int)(i % K), K, Wp[i] * scale);
}
// (D, M1 are [24][N]; the tile engine wants a channel's 24 values contiguous: Ds, M1s are [N][24] - six 16-byte loads per table and lane)
__global__ void scale_epilogue_kernel(const float* __restrict__ D, const float* __restrict__ M1, float* __restrict__ Ds,
                                      float* __restrict__ M1s, int N, float inv_scale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kJ * N) return;
  const int j = i / N, n = i % N;
  Ds[n * kJ + j] = D[i] * inv_scale;
  M1s[n * kJ + j] = M1[i] * inv_scale;
}

// ------------------------------------------------------------------------------------------------
// hid -> hid conv: f32 MFMA GEMM + in-register epilogue
// ------------------------------------------------------------------------------------------------
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

template <bool RES>
__global__ __launch_bounds__(256, 2) void gcn_hidden_kernel(const float* __restrict__ X, LayerDev L,
                                                             const float* __restrict__ Res, float* __restrict__ Y,
                                                             int m_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];  // 80 KiB: 2 blocks per CU

  const int K = L.K, N = L.N;
  const int n_tiles = N / BNH;
  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of (m_tile, all n_tiles)
  int bid = blockIdx.x;
  const int total = m_tiles * n_tiles;
  int lin = ((total & 7) == 0) ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;
  const int m_tile = lin / n_tiles, n_tile = lin % n_tiles;
  const size_t m0 = (size_t)m_tile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- global -> LDS DMA addressing (global_load_lds: LDS dest = wave-uniform base + lane*16 B) ----
  // a wave instruction fills 8 rows x 128 B; physical (row r, chunk c) holds logical chunk c ^ ((r>>1)&7)
  const int ld_r = lane >> 3, ld_c = lane & 7;
  const float* gA[6];
  const float* gB[4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int r = 8 * (wave + 4 * i) + ld_r;
    gA[i] = X + (m0 + r) * K + ((ld_c ^ ((r >> 1) & 7)) << 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = 8 * (wave + 4 * i) + ld_r;
    gB[i] = L.Wp + ((size_t)n_tile * 128 + r) * K + ((ld_c ^ ((r >> 1) & 7)) << 2);
  }
  auto stage = [&](int buf, int kt) {
    float* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(gA[i] + kt * BK), (AS3 void*)(base + (wave + 4 * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const AS1 void*)(gB[i] + kt * BK),
                                       (AS3 void*)(base + A_TILE + (wave + 4 * i) * 256), 16, 0, 0);
  };

  // ---- MFMA fragment addressing ----
  // MFMA row i of row-tile t  <->  LDS row 48*((i>>2)&1) + 16t + (i&3) + 4*(i>>3)  (+96*wm)
  // => lane half h = lane>>5 ends up with accumulator index q = 16t + reg  <->  body 2h + q/24, joint q%24
  const int mi = lane & 31, h = lane >> 5;
  const int rA = 96 * wm + 48 * ((mi >> 2) & 1) + (mi & 3) + 4 * (mi >> 3);
  const int rB = 32 * wn + mi;
  const int keyA = (rA >> 1) & 7, keyB = (rB >> 1) & 7;

  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }
  }

  const int KT = K / BK;
  stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; everyone is done reading the other buffer
    if (kt + 1 < KT) stage((kt + 1) & 1, kt + 1);
    const float* As = lds + (kt & 1) * STAGE;
    const float* Bs = As + A_TILE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cA = ((2 * kk + h) ^ keyA) << 2, cB = ((2 * kk + h) ^ keyB) << 2;
      f32x4 a[3], b[2];
#pragma unroll
      for (int t = 0; t < 3; ++t) a[t] = *(const f32x4*)(As + (rA + 16 * t) * BK + cA);
#pragma unroll
      for (int u = 0; u < 2; ++u) b[u] = *(const f32x4*)(Bs + (rB + 64 * u) * BK + cB);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          acc0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[0][s], acc0[t], 0, 0, 0);
          acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[1][s], acc1[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: lane = channel n, two bodies (beta = 0,1) of 24 joints each in registers ----
  // Hand-staged so the register allocator never sees more than acc (96) + two 24-wide load batches:
  // sched_barrier keeps hipcc from hoisting all ~100 global loads to the top and spilling the accumulators.
  const int n = n_tile * BNH + 32 * wn + mi;
  const size_t rowb = m0 + 96 * wm + 48 * h;
  float res0[kJ], res1[kJ];
#pragma unroll
  for (int j = 0; j < kJ; ++j) res0[j] = RES ? Res[(rowb + j) * (size_t)N + n] : 0.f;
  {
    float dj[kJ], mj[kJ];
    const float sh = L.shift[n];
#pragma unroll
    for (int j = 0; j < kJ; ++j) { dj[j] = L.D[j * N + n]; mj[j] = L.M1[j * N + n]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < kJ; ++j) {   // fold modulation / BatchNorm scale into the accumulators in place
#pragma unroll
      for (int beta = 0; beta < 2; ++beta) {
        const int q = 24 * beta + j;
        acc0[q >> 4][q & 15] = fmaf(dj[j], acc0[q >> 4][q & 15], sh);
        acc1[q >> 4][q & 15] *= mj[j];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < kJ; ++j) res1[j] = RES ? Res[(rowb + 24 + j) * (size_t)N + n] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  {
    float d0[kJ], g1[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) { d0[j] = acc0[j >> 4][j & 15]; g1[j] = acc1[j >> 4][j & 15]; }
    gcn_mix_store<false>(d0, g1, res0, n, N, rowb, L.Aoff, Y, L.relu != 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float d0[kJ], g1[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) { d0[j] = acc0[(24 + j) >> 4][(24 + j) & 15]; g1[j] = acc1[(24 + j) >> 4][(24 + j) & 15]; }
    gcn_mix_store<false>(d0, g1, res1, n, N, rowb + 24, L.Aoff, Y, L.relu != 0);
  }
}

// ------------------------------------------------------------------------------------------------
// input conv with the step-invariant projections hoisted (see ehm_gcn_input_layer in the header)
// one wave = one virtual body x 64 channels; 4 waves per block = 4 channel groups
// ------------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(256) void gcn_input_kernel(GcnInputArgs a) {
  __shared__ __attribute__((aligned(16))) float T[kJ * 256];
  gcn_input_body<OUT>(T, blockIdx.x, blockIdx.y, a);
}

// the same epilogue on ready-made pre-activations [bodies * 24][2][hid] (ehm_gcn_input_layer_rows: ModulatedGCN.forward on an arbitrary input feature)
template <int OUT>
__global__ __launch_bounds__(256) void gcn_input_rows_kernel(GcnInputArgs a) {
  __shared__ __attribute__((aligned(16))) float T[kJ * 256];
  gcn_input_body<OUT, false, true>(T, (int)threadIdx.x, blockIdx.x, blockIdx.y, a, nullptr);
}

// ------------------------------------------------------------------------------------------------
// output conv (hid -> 6, both branches) + visibility fuse, two kernels:
//   gcn_out_dot_kernel   HBM-bound: every activation row is read once (float4); [rows,K] x [K,12] on the exact-f32 MFMA.
//   gcn_out_mix_kernel   per body: modulated adjacency mix of the [24 x 12] responses, bias, pass selection by visibility.
// ------------------------------------------------------------------------------------------------

// [rows, K] x [K, 12] on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: 16 rows x 16 columns, 12 used).  Block = 16 rows, its 4 waves
// split K; a wave's lane (row = l&15, q = l>>4) streams float4 X[row][kw + 16 i + 4 q ..+3] - the k order inside an MFMA step is a
// permutation applied to both operands, which a sum does not see - and the matching float4 of the 12 x K weights (48 KiB, L2 hits).
// The four partial 16x16 tiles meet in 4 KiB of LDS.  HBM-bound by design: every activation row is read once, 16-byte loads,
// 4 x 64 B per row and instruction.  (The previous VALU version re-read the weights from LDS for every row: 39 us per launch.)
template <bool HALF_IN>   // HALF_IN: X holds f16 rows (the 'f16' mode's last hidden conv), converted on load
__global__ __launch_bounds__(256) void gcn_out_dot_kernel(const float* __restrict__ X, OutDev O, float* __restrict__ hs, int64_t rows) {
  __shared__ float part[4][16][16];
  gcn_out_dot_rows16<HALF_IN, 0>(X, O, hs, (int64_t)blockIdx.x * OUT_ROWS_PER_BLOCK, rows, part, (int)threadIdx.x);
}

__global__ __launch_bounds__(192) void gcn_out_mix_kernel(const float* __restrict__ hs, OutDev O, const uint8_t* __restrict__ vis,
                                                          float* __restrict__ x0, int B, int passes, const int32_t* __restrict__ mask_slot) {
  __shared__ float sh[2][kJ][12];
  __shared__ float outs[2][kJ][6];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int slot = mask_slot ? mask_slot[b] : b;              // row block of my second pass: B + slot (slot < 0: pruned, all joints visible)
  for (int i = tid; i < passes * kJ * 12; i += 192) {
    const int p = i / (kJ * 12), rem = i % (kJ * 12);
    sh[p][rem / 12][rem % 12] = (p == 1 && slot < 0) ? 0.f : hs[((size_t)(p ? B + slot : b) * kJ) * 12 + rem];
  }
  __syncthreads();
  for (int e = tid; e < passes * kJ * 6; e += 192) {
    const int p = e / (kJ * 6), j = (e / 6) % kJ, c = e % 6;
    // modulated_gcn_conv.py:47: (adj*E) @ (M*h0) + (adj*(1-E)) @ (M*h1) + bias
    const float s = O.A[j * kJ + j] * (O.M[j * 6 + c] * sh[p][j][c]);
    float t = 0.f;
    for (int jp = 0; jp < kJ; ++jp)
      if (jp != j) t = fmaf(O.A[j * kJ + jp], O.M[jp * 6 + c] * sh[p][jp][6 + c], t);
    outs[p][j][c] = s + t + O.bias[c];
  }
  __syncthreads();
  if (tid < kPoseDim) {
    const int j = tid / 6, c = tid % 6;
    float v = outs[0][j][c];
    if (passes == 2 && !vis[(size_t)b * kJ + j]) v = outs[1][j][c];   // egohmr.py:249-254
    x0[(size_t)b * kPoseDim + tid] = v;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// [Aoff | I] (rows = output joint, 48 columns: 24 neighbour joints of the off-diagonal branch, then the identity that carries the
// diagonal branch through the same MFMA) as v_mfma_f32_32x32x16_f16 A fragments: lane l holds row l&31 (zero for rows >= 24),
// columns 16 s + 8 (l>>5) .. +7 of k-step s.
__global__ void pack_aoff_half_kernel(const float* __restrict__ Aoff, half8* __restrict__ out) {
  const int lane = threadIdx.x, i = lane & 31;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    half8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int kk = 16 * s + 8 * (lane >> 5) + c;
      float x = 0.f;
      if (i < kJ) x = kk < kJ ? Aoff[i * kJ + kk] : (kk - kJ == i ? 1.f : 0.f);
      v[c] = (half_t)x;
    }
    out[s * 64 + lane] = v;
  }
}

static int pack_layer(const float* adj, const ehm_gconv_params& p, LayerDev& L, float*& cursor, bool with_w, hipStream_t st) {
  const int N = p.out_dim, K = p.in_dim;
  L.K = K;
  L.N = N;
  L.relu = p.bn_weight != nullptr;
  if (with_w) {
    L.Wp = cursor;
    cursor += (size_t)2 * K * N;
    L.Ws = (half_t*)cursor;
    cursor += (size_t)2 * K * N;       // X2 format has the same byte size as float32
    L.Wh = (half_t*)cursor;
    cursor += (size_t)K * N;           // 2*K*N halves
    L.Ds = cursor;   cursor += (size_t)kJ * N;
    L.M1s = cursor;  cursor += (size_t)kJ * N;
  }
  if (!with_w) {     // the input conv reads its tables per channel too (gcn_dev.h: gcn_input_body): [N][24] copies, unscaled
    L.Ds = cursor;   cursor += (size_t)kJ * N;
    L.M1s = cursor;  cursor += (size_t)kJ * N;
    L.w_scale = 1.f;
  }
  L.D = cursor;      cursor += (size_t)kJ * N;
  L.M1 = cursor;     cursor += (size_t)kJ * N;
  L.shift = cursor;  cursor += N;
  L.Aoff = cursor;   cursor += kJ * kJ;
  L.AoffH = (const half_t*)cursor;  cursor += 768;     // 3 x 64 x 8 halves
  if (with_w) {
    dim3 grid((unsigned)ceil_div(N, 32), (unsigned)ceil_div(K, 32), 2);
    hipLaunchKernelGGL(pack_w_kernel, grid, dim3(256), 0, st, p.W, L.Wp, K, N);
  }
  const int threads = 1024;  // block 0 also fills the 24x24 adjacency (576 <= 1024 threads)
  hipLaunchKernelGGL(pack_epilogue_kernel, dim3((unsigned)ceil_div(N, threads)), dim3(threads), 0, st, adj, p, L.D, L.M1,
                     L.shift, L.Aoff);
  hipLaunchKernelGGL(pack_aoff_half_kernel, dim3(1), dim3(64), 0, st, L.Aoff, (half8*)L.AoffH);
  if (!with_w) hipLaunchKernelGGL(scale_epilogue_kernel, dim3((unsigned)ceil_div(kJ * N, 256)), dim3(256), 0, st, L.D, L.M1, L.Ds, L.M1s, N, 1.f);
  EHM_LAUNCH_CHECK();
  if (with_w) {
    // power-of-two weight scale that keeps |W|*scale well inside f16 and pushes the lo parts out of the subnormal range
    unsigned int* d_max = (unsigned int*)cursor;   // scratch word at the current arena cursor (overwritten by later packing)
    EHM_HIP(hipMemsetAsync(d_max, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, p.W, (size_t)2 * K * N, d_max);
    unsigned int bits = 0;
    EHM_HIP(hipMemcpyAsync(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost, st));
    EHM_HIP(hipStreamSynchronize(st));
    float wmax;
    memcpy(&wmax, &bits, sizeof(wmax));
    int e = 0;
    if (wmax > 0.f && wmax < 3.0e38f) {
      (void)frexpf(wmax, &e);                 // wmax = m * 2^e, m in [0.5,1)
      e = 12 - e;                             // |W| * 2^e < 4096
      if (e > 24) e = 24;
      if (e < -24) e = -24;
    }
    L.w_scale = ldexpf(1.f, e);
    hipLaunchKernelGGL(pack_ws_kernel<32>, dim3((unsigned)ceil_div((int64_t)2 * K * N, 256)), dim3(256), 0, st, L.Wp, L.Ws, (size_t)2 * N, K,
                       L.w_scale);
    ehm_pack_half(L.Wp, L.Wh, (size_t)2 * K * N, L.w_scale, st);
    hipLaunchKernelGGL(scale_epilogue_kernel, dim3((unsigned)ceil_div(kJ * N, 256)), dim3(256), 0, st, L.D, L.M1, L.Ds, L.M1s, N,
                       1.f / L.w_scale);
    EHM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int ehm_gcn_row_tile(void) { return BM; }

extern "C" int ehm_gcn_create(ehm_gcn** out, const float* adj, const ehm_gconv_params* input_conv,
                              const ehm_gconv_params* hidden, int num_hidden, const ehm_gconv_params* output_conv,
                              int hid_dim, void* stream) {
  EHM_CHECK_ARG(out && adj && input_conv && hidden && output_conv);
  EHM_CHECK_ARG(num_hidden >= 0 && num_hidden <= 16);
  EHM_CHECK_ARG(hid_dim > 0 && hid_dim % 64 == 0 && hid_dim % BK == 0);
  EHM_CHECK_ARG(input_conv->out_dim == hid_dim && output_conv->in_dim == hid_dim && output_conv->out_dim == 6);
  for (int i = 0; i < num_hidden; ++i) EHM_CHECK_ARG(hidden[i].in_dim == hid_dim && hidden[i].out_dim == hid_dim && hidden[i].W);
  hipStream_t st = (hipStream_t)stream;
  auto* g = new ehm_gcn();
  if (const char* e = getenv("EHM_F16_CHAIN")) g->chain = atoi(e);   // 0: one launch per hidden conv (debugging aid; bit-identical results)
  g->hid = hid_dim;
  g->num_hidden = num_hidden;
  const size_t epi = (size_t)2 * kJ * hid_dim + hid_dim + kJ * kJ + 768;
  size_t floats = epi * (1 + num_hidden) + (size_t)num_hidden * (5 * (size_t)hid_dim * hid_dim + 2 * kJ * hid_dim) + 2 * (size_t)kJ * hid_dim +
                  12 * (size_t)hid_dim + kJ * 6 + kJ * kJ + 8 + 64;
  if (hipMalloc(&g->arena, floats * sizeof(float)) != hipSuccess) {
    delete g;
    ehm_set_error("ehm_gcn_create: hipMalloc of %zu bytes failed", floats * sizeof(float));
    return EHM_ENOMEM;
  }
  float* cur = g->arena;
  int rc = pack_layer(adj, *input_conv, g->input, cur, false, st);
  for (int i = 0; i < num_hidden && rc == 0; ++i) rc = pack_layer(adj, hidden[i], g->hidden[i], cur, true, st);
  if (rc == 0) {
    g->out.K = hid_dim;
    g->out.Wt = cur;   cur += 12 * (size_t)hid_dim;
    g->out.M = cur;    cur += kJ * 6;
    g->out.A = cur;    cur += kJ * kJ;
    g->out.bias = cur; cur += 8;
    hipLaunchKernelGGL(pack_out_kernel, dim3((unsigned)ceil_div(12 * hid_dim, 256)), dim3(256), 0, st, adj, *output_conv,
                       g->out.Wt, g->out.M, g->out.A, g->out.bias);
    if (hipGetLastError() != hipSuccess) rc = EHM_EIO;
  }
  if (rc == 0 && num_hidden > 0) {
    if (hipMalloc(&g->hidden_dev, sizeof(LayerDev) * num_hidden) != hipSuccess ||
        hipMemcpyAsync(g->hidden_dev, g->hidden, sizeof(LayerDev) * num_hidden, hipMemcpyHostToDevice, st) != hipSuccess)
      rc = EHM_ENOMEM;
  }
  if (rc == 0 && (hipMalloc(&g->chain_sticky, 64) != hipSuccess || hipMemsetAsync(g->chain_sticky, 0, 64, st) != hipSuccess)) rc = EHM_ENOMEM;
  if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = EHM_EIO;
  if (rc == 0) rc = ehm_gcn_reserve_rows(g, 2 * 256 * kJ);   // the benchmark shape; larger batches grow it on first use (ehm_gcn_reserve)
  if (rc != 0) {
    if (g->hidden_dev) (void)hipFree(g->hidden_dev);
    if (g->chain_sticky) (void)hipFree(g->chain_sticky);
    if (g->chain_sync) (void)hipFree(g->chain_sync);
    if (g->hs) (void)hipFree(g->hs);
    (void)hipFree(g->arena);
    delete g;
    ehm_set_error("ehm_gcn_create: packing kernels failed");
    return rc;
  }
  *out = g;
  return 0;
}

extern "C" void ehm_gcn_destroy(ehm_gcn* h) {
  if (!h) return;
  (void)hipFree(h->arena);
  if (h->hs) (void)hipFree(h->hs);
  if (h->hidden_dev) (void)hipFree(h->hidden_dev);
  if (h->chain_sync) (void)hipFree(h->chain_sync);
  if (h->chain_sticky) (void)hipFree(h->chain_sticky);
  delete h;
}

// fills the launch arguments of the input conv for the handle's current precision / pass map (shared with smpl.hip's fused launch)
int ehm_gcn_input_args(ehm_gcn* h, const float* h_img, const float* h_oth, const uint8_t* vis, const float* x, const float* Wx, const float* tvec,
                       float* out, int B, int passes, GcnInputArgs* a) {
  EHM_CHECK_ARG(h && h_img && h_oth && vis && x && Wx && tvec && out && a);
  EHM_CHECK_ARG(B > 0 && (passes == 1 || passes == 2));
  a->h_img = h_img; a->h_oth = h_oth; a->vis = vis; a->x = x; a->Wx = Wx; a->tvec = tvec;
  a->L = h->input;
  a->Y = out;
  a->B = B; a->passes = passes; a->mask_all = h->uncond_masks_all;
  a->mask_items = (passes == 2 && h->num_masked >= 0) ? h->mask_items : nullptr;
  a->sticky = h->chain_sticky;
  a->total_vb = ehm_gcn_virtual_bodies(h, B, passes);
  h->valid_rows = (int64_t)a->total_vb * kJ;
  a->ny = (int)ceil_div(h->hid, 256);
  return 0;
}

extern "C" int ehm_gcn_input_layer(ehm_gcn* h, const float* h_img, const float* h_oth, const uint8_t* vis, const float* x,
                                   const float* Wx, const float* tvec, float* out, int B, int passes, void* stream) {
  GcnInputArgs a;
  const int rc = ehm_gcn_input_args(h, h_img, h_oth, vis, x, Wx, tvec, out, B, passes, &a);
  if (rc != 0) return rc;
  dim3 grid((unsigned)a.total_vb, (unsigned)a.ny);
  if (h->precision == EHM_PREC_F32) hipLaunchKernelGGL(gcn_input_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (h->precision == EHM_PREC_F16X3) hipLaunchKernelGGL(gcn_input_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);   // X2 split rows
  else hipLaunchKernelGGL(gcn_input_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);                                        // plain f16 rows
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_input_layer_rows(ehm_gcn* h, const float* pre, float* out, int bodies, void* stream) {
  EHM_CHECK_ARG(h && pre && out && bodies > 0);
  GcnInputArgs a{};
  a.L = h->input;
  a.Y = out;
  a.B = bodies; a.passes = 1; a.mask_all = 0;
  a.mask_items = nullptr;
  a.total_vb = bodies;
  a.ny = (int)ceil_div(h->hid, 256);
  a.sticky = h->chain_sticky;
  h->valid_rows = (int64_t)bodies * kJ;
  a.pre = pre;
  dim3 grid((unsigned)a.total_vb, (unsigned)a.ny);
  if (h->precision == EHM_PREC_F32) hipLaunchKernelGGL(gcn_input_rows_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (h->precision == EHM_PREC_F16X3) hipLaunchKernelGGL(gcn_input_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gcn_input_rows_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_hidden_layer(ehm_gcn* h, int layer, const float* X, const float* residual, float* out,
                                    int64_t rows_pad, void* stream) {
  EHM_CHECK_ARG(h && X && out);
  EHM_CHECK_ARG(layer >= 0 && layer < h->num_hidden);
  EHM_CHECK_ARG(rows_pad > 0 && rows_pad % BM == 0);
  EHM_CHECK_ARG(X != out);
  if (h->precision != EHM_PREC_F32)   // X / residual / out in the mode's activation format, except that in mode 1 the last hidden conv writes float32 for the output conv
    return ehm_gcn_tile_layer_impl(h, layer, X, residual, out, rows_pad, layer == h->num_hidden - 1, (hipStream_t)stream);
  const int m_tiles = (int)(rows_pad / BM);
  const int blocks = m_tiles * (h->hid / BNH);
  if (residual)
    hipLaunchKernelGGL(gcn_hidden_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, h->hidden[layer], residual,
                       out, m_tiles);
  else
    hipLaunchKernelGGL(gcn_hidden_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, h->hidden[layer],
                       (const float*)nullptr, out, m_tiles);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_gcn_hidden_stack(ehm_gcn* h, float* const bufs[3], int64_t rows_pad, int* result_index, void* stream) {
  EHM_CHECK_ARG(h && bufs && bufs[0] && bufs[1] && bufs[2] && result_index && rows_pad > 0 && rows_pad % BM == 0);
  EHM_CHECK_ARG(h->num_hidden % 2 == 0);
  const int nblk = h->num_hidden / 2;
  *result_index = (nblk & 1) ? 2 : 0;
  if (nblk == 0) return 0;
  if (h->chain && h->precision != EHM_PREC_F32) return ehm_gcn_tile_chain_impl(h, (void* const*)bufs, rows_pad, (hipStream_t)stream);
  int in = 0;
  for (int blk = 0; blk < nblk; ++blk) {     // the same buffer rotation, one launch per conv
    const int y2 = in == 0 ? 2 : 0;
    int rc = ehm_gcn_hidden_layer(h, 2 * blk, bufs[in], nullptr, bufs[1], rows_pad, stream);
    if (rc == 0) rc = ehm_gcn_hidden_layer(h, 2 * blk + 1, bufs[1], bufs[in], bufs[y2], rows_pad, stream);
    if (rc != 0) return rc;
    in = y2;
  }
  return 0;
}

extern "C" int ehm_gcn_stack_status(ehm_gcn* h, void* stream) {
  EHM_CHECK_ARG(h);
  unsigned int flag = 0;
  if (h->chain_sticky) {
    EHM_HIP(hipMemcpyAsync(&flag, h->chain_sticky, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream));
    EHM_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
      EHM_HIP(hipMemsetAsync(h->chain_sticky, 0, sizeof(flag), (hipStream_t)stream));
      h->chain_sync_clean = 0;            // a launch that gave up may have left tickets / counters behind
    }
  }
  if (flag & ~kStickySaturated) {
    ehm_set_error("ehm_gcn_hidden_stack: a chained launch since the last status call timed out waiting for a producer tile, or left tiles "
                  "unproduced (GPU shared / preempted / CU-masked?); the results of that sampling loop are invalid - re-run, or set EHM_F16_CHAIN=0");
    return EHM_EIO;
  }
  if (flag & kStickySaturated) {
    ehm_set_error("ehm_gcn: an activation of the denoiser reached the f16 range (|x| >= 65504) since the last status call and was CLAMPED in its X2 / f16 "
                  "store - the split format holds |x| <= 131008 and is f32-grade only below 65504 (nothing became inf / NaN, the results are finite but not "
                  "parity grade).  This checkpoint needs ehm_gcn_set_precision(h, 0) (EgoHMR.gcn_precision = 'f32': float32 activations, exact-f32 MFMA)");
    return EHM_ERANGE;
  }
  return 0;
}

extern "C" int ehm_gcn_stack_status_async(ehm_gcn* h, uint32_t* host_flag, void* stream) {
  EHM_CHECK_ARG(h && host_flag);
  if (h->chain_sticky) EHM_HIP(hipMemcpyAsync(host_flag, h->chain_sticky, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  else *host_flag = 0u;
  return 0;
}

int ehm_gcn_reserve_rows(ehm_gcn* h, int64_t rows_pad) {
  if (rows_pad <= h->reserved_rows) return 0;
  const size_t need = 8 + (size_t)((h->num_hidden > 0 ? h->num_hidden : 1) + 3) * (size_t)ceil_div(rows_pad, BM) + 32;   // (+ 3 x m_tiles: the one-launch loop's INPUT / OUT / BODY counters)
  if (h->chain_sync) EHM_HIP(hipFree(h->chain_sync));
  h->chain_sync = nullptr;
  h->chain_sync_words = 0;
  EHM_HIP(hipMalloc(&h->chain_sync, need * sizeof(unsigned int)));
  h->chain_sync_words = need;
  h->chain_sync_clean = 0;
  if (h->hs) EHM_HIP(hipFree(h->hs));
  h->hs = nullptr;
  h->hs_rows = 0;
  EHM_HIP(hipMalloc(&h->hs, (size_t)round_up(rows_pad, 4096) * 12 * sizeof(float)));
  h->hs_rows = round_up(rows_pad, 4096);
  h->reserved_rows = rows_pad;
  return 0;
}

extern "C" int ehm_gcn_reserve(ehm_gcn* h, int max_bodies, int passes) {
  EHM_CHECK_ARG(h && max_bodies > 0 && (passes == 1 || passes == 2));
  return ehm_gcn_reserve_rows(h, round_up((int64_t)max_bodies * passes * kJ, BM));
}

extern "C" int ehm_gcn_output_layer(ehm_gcn* h, const float* X, const uint8_t* vis, float* x0, int B, int passes,
                                    void* stream) {
  EHM_CHECK_ARG(h && X && x0);
  EHM_CHECK_ARG(B > 0 && (passes == 1 || (passes == 2 && vis)));
  const int64_t rows = (int64_t)ehm_gcn_virtual_bodies(h, B, passes) * kJ;
  if (rows > h->hs_rows) {      // [rows,12] scratch of the two-kernel output conv: sized by ehm_gcn_create / ehm_gcn_reserve, grown here only
    const int rc = ehm_gcn_reserve_rows(h, round_up(rows, BM));   // for a batch larger than reserved (allocates: call ehm_gcn_reserve before a capture)
    if (rc != 0) return rc;
  }
  if (h->precision == EHM_PREC_F16)
    hipLaunchKernelGGL(gcn_out_dot_kernel<true>, dim3((unsigned)ceil_div(rows, OUT_ROWS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, X, h->out,
                       h->hs, rows);
  else
    hipLaunchKernelGGL(gcn_out_dot_kernel<false>, dim3((unsigned)ceil_div(rows, OUT_ROWS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, X, h->out,
                       h->hs, rows);
  hipLaunchKernelGGL(gcn_out_mix_kernel, dim3(B), dim3(192), 0, (hipStream_t)stream, h->hs, h->out, vis, x0, B, passes,
                     (passes == 2 && h->num_masked >= 0) ? h->mask_slot : nullptr);
  EHM_LAUNCH_CHECK();
  return 0;
}

int ehm_gcn_virtual_bodies(const ehm_gcn* h, int B, int passes) {
  return passes == 2 ? B + (h->num_masked >= 0 ? h->num_masked : B) : B;
}
const int32_t* ehm_gcn_mask_slot(const ehm_gcn* h, int passes) { return (passes == 2 && h->num_masked >= 0) ? h->mask_slot : nullptr; }

extern "C" int ehm_gcn_set_pass_map(ehm_gcn* h, const int32_t* mask_items, const int32_t* mask_slot, int num_masked) {
  EHM_CHECK_ARG(h && (num_masked < 0 || (mask_slot && (num_masked == 0 || mask_items))));
  h->mask_items = num_masked >= 0 ? mask_items : nullptr;
  h->mask_slot = num_masked >= 0 ? mask_slot : nullptr;
  h->num_masked = num_masked < 0 ? -1 : num_masked;
  return 0;
}

extern "C" int ehm_gcn_set_nonlocal(ehm_gcn* h, const ehm_nonlocal_params* p) {
  EHM_CHECK_ARG(h);
  if (!p) { h->nonlocal = ehm_nonlocal_params{}; return 0; }
  EHM_CHECK_ARG(p->Wqkv && p->bqkv && p->Wo && p->bo && p->Ci > 0 && p->Ci % 8 == 0 && p->qkv_scale > 0.f && p->o_scale > 0.f);
  h->nonlocal = *p;
  return 0;
}
int ehm_gcn_nonlocal_ci(const ehm_gcn* h) { return h->nonlocal.Ci; }
const ehm_nonlocal_params* ehm_gcn_nonlocal(const ehm_gcn* h) { return &h->nonlocal; }

int ehm_gcn_output_dot_impl(ehm_gcn* h, const float* X, int B, int passes, const float** hs, const void** out_dev, hipStream_t st) {
  const int64_t rows = (int64_t)ehm_gcn_virtual_bodies(h, B, passes) * kJ;
  if (rows > h->hs_rows) {
    const int rc = ehm_gcn_reserve_rows(h, round_up(rows, BM));
    if (rc != 0) return rc;
  }
  if (h->precision == EHM_PREC_F16)
    hipLaunchKernelGGL(gcn_out_dot_kernel<true>, dim3((unsigned)ceil_div(rows, OUT_ROWS_PER_BLOCK)), dim3(256), 0, st, X, h->out, h->hs, rows);
  else
    hipLaunchKernelGGL(gcn_out_dot_kernel<false>, dim3((unsigned)ceil_div(rows, OUT_ROWS_PER_BLOCK)), dim3(256), 0, st, X, h->out, h->hs, rows);
  EHM_LAUNCH_CHECK();
  *hs = h->hs;
  *out_dev = &h->out;
  return 0;
}

const void* ehm_gcn_out_dev(const ehm_gcn* h) { return &h->out; }
int ehm_gcn_hid(const ehm_gcn* h) { return h->hid; }
int ehm_gcn_num_hidden(const ehm_gcn* h) { return h->num_hidden; }
int ehm_gcn_chain_enabled(const ehm_gcn* h) { return h->chain != 0 && h->precision != EHM_PREC_F32; }

extern "C" int ehm_gcn_set_precision(ehm_gcn* h, int mode) {
  EHM_CHECK_ARG(h && (mode == EHM_PREC_F32 || mode == EHM_PREC_F16X3 || mode == EHM_PREC_F16));
  h->precision = mode;
  return 0;
}
extern "C" int ehm_gcn_get_precision(const ehm_gcn* h) { return h ? h->precision : EHM_EINVAL; }
extern "C" int ehm_gcn_set_uncond_mode(ehm_gcn* h, int masks_whole_condition) {
  EHM_CHECK_ARG(h && (masks_whole_condition == 0 || masks_whole_condition == 1));
  h->uncond_masks_all = masks_whole_condition;
  return 0;
}
extern "C" int ehm_gcn_activation_group(const ehm_gcn* h) { return h ? (h->precision == EHM_PREC_F16 ? 0 : 32) : EHM_EINVAL; }
