// What does the matrix pipe of an MI355X sustain at the socket power cap when NOTHING but v_mfma_f32_32x32x16_f16 runs, on operands that
// toggle like the hidden convs' (tools/mfma_ceiling.py)?  8 waves per CU (2 per SIMD, like the chained conv kernels), every wave keeps 8 A
// and 4 B fragments in registers and cycles through them: no LDS, no global traffic inside the loop, accumulators 3 x 2 like the conv's wave
// tile.  Variant 0: A x B as loaded (the f16 conv); variant 1: the split-f16 sequence (lo*hi, hi*lo, hi*hi per product).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void mfma_loop(const half8* __restrict__ A, const half8* __restrict__ B, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * 8 + wave) * 64 + lane;
  const size_t stride = (size_t)gridDim.x * 8 * 64;
  half8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = A[base + i * stride];       // VARIANT 1: a[0..3] = hi fragments, a[4..7] = their lo parts
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = B[base + i * stride];       //            b[0..1] = hi, b[2..3] = lo
  f32x16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if constexpr (VARIANT == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(t + 2 * p) & 7], b[(u + p) & 3], acc[t][u], 0, 0, 0);
        } else {
          const int ah = (t + p) & 3, al = 4 + ah;
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[al], b[(u + p) & 1], acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ah], b[2 + ((u + p) & 1)], acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ah], b[(u + p) & 1], acc[t][u], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[base] = s;
}

// The same questions for v_mfma_f32_16x16x32_f16 (half the MACs per instruction, a quarter of the accumulators: 4 instead of 16 per lane): variant 2 = A x B as
// loaded, variant 3 = the split-f16 sequence.  A wave's 3 x 2 tile of 32 x 32 outputs becomes 6 x 4 tiles of 16 x 16 (24 f32x4 accumulators = the same 96 registers);
// per "product" of the 32 x 32 x 16 form, four 16 x 16 x 32 instructions over twice the K depth do the same number of MACs.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int VARIANT>
__global__ __launch_bounds__(512, 2) void mfma_loop16(const half8* __restrict__ A, const half8* __restrict__ B, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * 8 + wave) * 64 + lane;
  const size_t stride = (size_t)gridDim.x * 8 * 64;
  half8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = A[base + i * stride];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = B[base + i * stride];
  f32x4v acc[6][4];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if constexpr (VARIANT == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(t + 3 * p) & 7], b[(u + p) & 3], acc[t][u], 0, 0, 0);
        } else {
          const int ah = (t + p) & 3, al = 4 + ah;
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[al], b[(u + p) & 1], acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ah], b[2 + ((u + p) & 1)], acc[t][u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ah], b[(u + p) & 1], acc[t][u], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[t][u][r];
  out[base] = s;
}
// 16 x 16 x 32 MFMAs per wave and iteration: variant 2: 2 * 6 * 4 = 48; variant 3: 2 * 6 * 12 = 144 (each 16 * 16 * 32 * 2 flop)

extern "C" int mfma_ceiling_launch(int variant, const void* A, const void* B, float* out, int blocks, int iters, void* stream) {
  if (variant == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const half8*)A, (const half8*)B, out, iters);
  else if (variant == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const half8*)A, (const half8*)B, out, iters);
  else if (variant == 2) hipLaunchKernelGGL(mfma_loop16<2>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const half8*)A, (const half8*)B, out, iters);
  else hipLaunchKernelGGL(mfma_loop16<3>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const half8*)A, (const half8*)B, out, iters);
  return (int)hipGetLastError();
}
// MFMAs per wave and iteration: variant 0: 4 * 3 * 2 = 24; variant 1: 4 * 3 * 6 = 72
