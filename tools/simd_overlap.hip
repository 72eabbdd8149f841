// Does a wave that issues operand loads get in the way of the MFMAs of the OTHER wave on its SIMD?  (tools/simd_overlap.py)
// One 8-wave block per CU: waves 0-3 ("compute", one per SIMD) issue 36 v_mfma_f32_32x32x16_f16 per iteration (optionally with 20
// ds_read_b128 fragment reads in between, like the split-f16 K tile); waves 4-7 ("loader", the second wave of each SIMD) issue P operand
// pieces per iteration (global_load_lds plain / sc1, or global_load_dwordx4 into registers); one s_barrier per iteration keeps the two
// roles in step (the conv kernel has one per K tile too).  Timed: compute alone, loader alone, both.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LOAD, bool READS, bool PACKED = true>   // LOAD 0 none, 1 global_load_lds, 2 global_load_lds sc1, 3 global_load_dwordx4 -> VGPR
__global__ __launch_bounds__(512) void overlap_loop(const half8* __restrict__ A, const char* __restrict__ src, unsigned int window_bytes,
                                                    int iters, int pieces, int do_compute, float* __restrict__ out, int valu) {
  __shared__ __attribute__((aligned(16))) char lds[80 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool compute = wave < 4;
  const unsigned int mask = window_bytes - 1;
  if (compute) {
    if (valu < 0) __builtin_amdgcn_s_setprio(3);                             // valu < 0: the same |valu| with the compute waves at raised priority
    half8 a[6], b[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = A[(size_t)(blockIdx.x * 4 + wave) * 64 * 10 + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = A[(size_t)(blockIdx.x * 4 + wave) * 64 * 10 + (6 + i) * 64 + lane];
    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const half8* frag = (const half8*)(lds + 40 * 1024) + wave * 640 + lane;   // 10 KiB of "fragments" per wave
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if (do_compute) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          half8 f[10];
          if (READS) {
#pragma unroll
            for (int i = 0; i < 10; ++i) f[i] = frag[i * 64];
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3 + t], b[u], acc[t][u], 0, 0, 0);
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[2 + u], acc[t][u], 0, 0, 0);
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[u], acc[t][u], 0, 0, 0);
            }
          }
          if (READS) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = a[i] + f[i] * (_Float16)0.0f;      // consume the reads without changing the operands
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = b[i] + f[6 + i] * (_Float16)0.0f;
          }
        }
      }
      __syncthreads();
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][u][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = (threadIdx.x == 0) ? (float)(t_end - t_begin) : s;   // thread 0: shader cycles of the loop
  } else {
    unsigned int off = ((blockIdx.x * 4 + (wave - 4)) * 16384u + lane * 16u) & mask;
    char* my = lds + (wave - 4) * 10240;
    u32x4 x = {0, 0, 0, 0};
    float e[8];                                                              // `valu` independent-chain FMAs per iteration: the epilogue slice a helper wave would run
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = lane * 0.001f + i;
    const float m1 = 0.999f + lane * 1e-9f, m2 = 0.001f;
    for (int it = 0; it < iters; ++it) {
      for (int v = 0; v < (valu < 0 ? -valu : valu); v += 64) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (PACKED) e[i] = __builtin_fmaf(e[i], 0.999f, 0.001f * (r + 1));          // hipcc SLP-packs these into v_pk_fma_f32
            else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(e[i]) : "v"(m1), "v"(m2));   // the conv epilogue's adjacency mix is 1128 plain v_fmac_f32
          }
      }
      if (LOAD != 0) {
        for (int p = 0; p < pieces; ++p) {
          const char* g = src + ((off + p * 1024u) & mask);
          if (LOAD == 1) __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)(my + (p % 10) * 1024), 16, 0, 0);
          else if (LOAD == 2) __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)(my + (p % 10) * 1024), 16, 0, 16);
          else { const u32x4 v = *(const u32x4*)g; x ^= v; }
        }
        off = (off + 4 * 64 * 16384u + 16384u) & mask;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // like the K tile: everything staged before the barrier
      }
      __syncthreads();
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]) + e[0] + e[1] + e[2] + e[3] + e[4] + e[5] + e[6] + e[7];
  }
}

extern "C" int overlap_launch(int load, int reads, const void* A, const void* src, unsigned int window_bytes, int iters, int pieces, int do_compute,
                              float* out, int blocks, void* stream, int valu) {
  hipStream_t st = (hipStream_t)stream;
#define L(LD, RD) hipLaunchKernelGGL((overlap_loop<LD, RD>), dim3(blocks), dim3(512), 0, st, (const half8*)A, (const char*)src, window_bytes, iters, pieces, do_compute, out, valu)
  if (reads) { if (load == 0) L(0, true); else if (load == 1) L(1, true); else if (load == 2) L(2, true); else L(3, true); }
  else { if (load == 0) L(0, false); else if (load == 1) L(1, false); else if (load == 2) L(2, false); else L(3, false); }
  return (int)hipGetLastError();
}

// the same, with the partner wave's vector-ALU work as plain v_fmac_f32 (LOAD: 0 none, 1 global_load_lds; fragment reads on)
extern "C" int overlap_launch_plain_valu(int load, const void* A, const void* src, unsigned int window_bytes, int iters, int pieces, int do_compute,
                                         float* out, int blocks, void* stream, int valu) {
  hipStream_t st = (hipStream_t)stream;
  if (load == 0) hipLaunchKernelGGL((overlap_loop<0, true, false>), dim3(blocks), dim3(512), 0, st, (const half8*)A, (const char*)src, window_bytes, iters, pieces, do_compute, out, valu);
  else hipLaunchKernelGGL((overlap_loop<1, true, false>), dim3(blocks), dim3(512), 0, st, (const half8*)A, (const char*)src, window_bytes, iters, pieces, do_compute, out, valu);
  return (int)hipGetLastError();
}
