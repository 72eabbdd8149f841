// Collision guidance for gfx950: placeholder translation unit (filled in by the guidance milestone).
#include "common.h"
#include "egohmr_hip.h"

#define EHM_ENOSYS (-38)

int ehm_guidance_impl(ehm_smpl*, const float*, const float*, const float*, const float*, const float*, int, int, float, float,
                      float*, float*, float*, float*, float*, float*, float*, float*, hipStream_t) {
  ehm_set_error("collision guidance kernels are not built in this library version");
  return EHM_ENOSYS;
}
extern "C" int ehm_collision_proxy(const float*, const float*, float*, float*, int, int, int, float, void*) {
  ehm_set_error("ehm_collision_proxy: not built in this library version");
  return EHM_ENOSYS;
}
extern "C" int ehm_smpl_backward_rot6d(ehm_smpl*, const float*, const float*, const float*, const float*, const float*, float*,
                                       int, void*) {
  ehm_set_error("ehm_smpl_backward_rot6d: not built in this library version");
  return EHM_ENOSYS;
}
extern "C" int ehm_guidance_grad_finish(const float*, const float*, float*, int, float, void*) {
  ehm_set_error("ehm_guidance_grad_finish: not built in this library version");
  return EHM_ENOSYS;
}
extern "C" int ehm_rot6d_to_rotmat_bwd(const float*, const float*, float*, int64_t, int, void*) {
  ehm_set_error("ehm_rot6d_to_rotmat_bwd: not built in this library version");
  return EHM_ENOSYS;
}
