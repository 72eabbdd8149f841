// Shared helpers for the gfx950 kernels behind include/egohmr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define EHM_EINVAL (-22)
#define EHM_ENOMEM (-12)
#define EHM_EIO (-5)

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
