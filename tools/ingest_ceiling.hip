// How many bytes per clock can one CU of an MI355X take in from its XCD's L2 - through global_load_lds (the LDS-DMA path the conv
// kernels stage their operands with) and through plain global_load_dwordx4 into registers - when nothing else runs?  (tools/ingest_ceiling.py)
// 8 waves per CU; every wave walks a 2 MiB window (L2-resident, far larger than the CU's 32 KiB L1) in 1 KiB wave-instructions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: global_load_lds plain; 1: global_load_lds sc1; 2: global_load_dwordx4 -> VGPR; 3: alternate 0 and 2
template <int MODE>
__global__ __launch_bounds__(512) void ingest_loop(const char* __restrict__ src, unsigned int window_bytes, int iters, unsigned int* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char lds[8 * 8 * 1024];          // 8 KiB ring per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int mask = window_bytes - 1;
  unsigned int off = ((blockIdx.x * 8 + wave) * 8192u + lane * 16u) & mask;
  u32x4 acc = {0, 0, 0, 0};
  char* my = lds + wave * 8192;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const char* a = src + ((off + p * 1024u) & mask);
      if (MODE == 0 || (MODE == 3 && (p & 1) == 0)) __builtin_amdgcn_global_load_lds((const AS1 void*)a, (AS3 void*)(my + p * 1024), 16, 0, 0);
      else if (MODE == 1) __builtin_amdgcn_global_load_lds((const AS1 void*)a, (AS3 void*)(my + p * 1024), 16, 0, 16);
      else { const u32x4 v = *(const u32x4*)a; acc ^= v; }
    }
    off = (off + 8 * 64 * 1024u + 8192u) & mask;                            // next 8 KiB chunk, far from the previous one
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                       // one chunk in flight behind the one being issued
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc[0] ^= ((const unsigned int*)lds)[threadIdx.x];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;     // keeps the loads alive
}

extern "C" int ingest_launch(int mode, const void* src, unsigned int window_bytes, int iters, void* sink, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(ingest_loop<0>, dim3(blocks), dim3(512), 0, st, (const char*)src, window_bytes, iters, (unsigned int*)sink); break;
    case 1: hipLaunchKernelGGL(ingest_loop<1>, dim3(blocks), dim3(512), 0, st, (const char*)src, window_bytes, iters, (unsigned int*)sink); break;
    case 2: hipLaunchKernelGGL(ingest_loop<2>, dim3(blocks), dim3(512), 0, st, (const char*)src, window_bytes, iters, (unsigned int*)sink); break;
    default: hipLaunchKernelGGL(ingest_loop<3>, dim3(blocks), dim3(512), 0, st, (const char*)src, window_bytes, iters, (unsigned int*)sink); break;
  }
  return (int)hipGetLastError();
}
