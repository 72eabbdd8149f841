// Shared helpers for the gfx950 kernels behind include/egohmr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define EHM_EINVAL (-22)
#define EHM_ENOMEM (-12)
#define EHM_EIO (-5)
#define EHM_ERANGE (-34)

extern "C" const char* ehm_last_error(void);
void ehm_set_error(const char* fmt, ...);

#define EHM_CHECK_ARG(cond)                                              \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ehm_set_error("%s:%d bad argument: %s", __func__, __LINE__, #cond); \
      return EHM_EINVAL;                                                 \
    }                                                                    \
  } while (0)

#define EHM_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      ehm_set_error("%s:%d %s -> %s", __func__, __LINE__, #call, hipGetErrorString(e__));     \
      return e__ == hipErrorOutOfMemory ? EHM_ENOMEM : EHM_EIO;                              \
    }                                                                                        \
  } while (0)

#define EHM_LAUNCH_CHECK() EHM_HIP(hipGetLastError())

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kJ = 24;        // SMPL joints = graph nodes
constexpr int kPoseDim = 144;  // 24 x 6

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- cross-lane helpers on the DPP path (hipcc lowers __shfl_xor to ds_bpermute_b32, an LDS-crossbar instruction) ----
template <int CTRL>
static __device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
static __device__ __forceinline__ unsigned int dpp_xor1_u32(unsigned int v) {          // value of lane ^ 1 (quad_perm [1,0,3,2])
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
static __device__ __forceinline__ float row16_sum(float v) {   // every lane ends with the sum over its 16-lane row
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror
  v += dpp_move<0x140>(v);   // row_mirror
  return v;
}
static __device__ __forceinline__ float wave_sum(float v) {    // wave-uniform sum over the 64 lanes
  const int b = __builtin_bit_cast(int, row16_sum(v));   // readlane is an integer builtin: move the bits, not the value
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

// Buffer descriptor over [p, p + 2 GiB) for raw_buffer_load/store: SGPR descriptor + 32-bit lane offset + scalar offset, i.e. no
// VALU address arithmetic.  readfirstlane pins the (wave-uniform) base into SGPRs; without it hipcc wraps every access in a
// waterfall loop.
// (num_records = 2 GiB - 1: the default; ehm_buffer_rsrc_4g covers [p, p + 4 GiB) for the one caller whose lane offsets span a whole tensor)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t ehm_buffer_rsrc_4g(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)a);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)0xffffffffu, 0x00020000);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t ehm_buffer_rsrc(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)a);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
