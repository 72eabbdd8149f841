// Per-item scalar work around the sampling loop, as two + one small launches instead of ~150 eager tensor ops:
//   ehm_item_prep     joint visibility (models/egohmr/egohmr.py:186-188), which items need the image-masked pass (:249-254),
//                     the camera features (:195-205), TranslEnc (:217, Linear 3 -> 64 -> ReLU -> 128), the per-item "inputs are finite" flag
//   ehm_pack_outputs  the output garnish of EgoHMR.forward (:283-301: focal length, camera centre, full-frame 3-D / 2-D keypoints through
//                     utils/geometry.py:78-116), the global_orient / body_pose split, and the NaN rule of a float32 graph (an item with a
//                     non-finite input comes out as NaN; the last step's draw only poisons its own element of `sample`,
//                     diffusion/gaussian_diffusion.py:357-359)
// A sampling call spent 0.8 ms (host-bound, GPU idle) in front of the encoders and 1.5 ms of 4-microsecond launches behind them on this.
#include "common.h"
#include "egohmr_hip.h"
#include "internal.h"

namespace {

constexpr int kPrepThreads = 128;

__global__ __launch_bounds__(kPrepThreads) void item_prep_kernel(ehm_item_prep_desc d, uint8_t* need) {
  __shared__ float hid[128];
  __shared__ int all_vis;
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) all_vis = 1;
  __syncthreads();
  // ---- visibility: confidence > 0, OpenPose joint `force_visible` always on, gathered into SMPL joint order
  if (t < kJ) {
    const int k = d.joint_map[t];
    const bool v = k == d.force_visible || d.keypoints_2d[((size_t)b * d.NK + k) * 3 + 2] > 0.f;
    d.vis[(size_t)b * kJ + t] = v ? 1 : 0;
    if (!v) atomicAnd(&all_vis, 0);
  }
  // ---- TranslEnc
  const float tx = d.transl[3 * b], ty = d.transl[3 * b + 1], tz = d.transl[3 * b + 2];
  for (int o = t; o < d.t_hidden; o += kPrepThreads) {
    float s = d.tb1[o];
    s = fmaf(d.tW1[3 * o], tx, s);
    s = fmaf(d.tW1[3 * o + 1], ty, s);
    s = fmaf(d.tW1[3 * o + 2], tz, s);
    hid[o] = fmaxf(s, 0.f);
  }
  __syncthreads();
  float* orow = d.other + (size_t)b * d.other_ld;
  for (int o = t; o < d.t_out; o += kPrepThreads) {
    float s = d.tb2[o];
    const float* w = d.tW2 + (size_t)o * d.t_hidden;
    for (int k = 0; k < d.t_hidden; ++k) s = fmaf(w[k], hid[k], s);
    orow[d.other_col0 + o] = s;
  }
  // ---- camera features, in the order the reference prepends them: [cx, cy] / ofx | [bcx, bcy, bs] / ofx | fx
  if (t == 0) {
    const float fx = d.fx[b], ofx = __fmul_rn(fx, d.fx_norm);
    int c = d.other_col0 + d.t_out;
    bool ok = isfinite(d.transl[3 * b] + d.transl[3 * b + 1] + d.transl[3 * b + 2]) && isfinite(fx);
    if (d.with_cam_center) {
      orow[c++] = __fdiv_rn(d.cx[b], ofx);
      orow[c++] = __fdiv_rn(d.cy[b], ofx);
      ok = ok && isfinite(d.cx[b]) && isfinite(d.cy[b]);
    }
    if (d.with_bbox) {
      orow[c++] = __fdiv_rn(d.box_center[2 * b], ofx);
      orow[c++] = __fdiv_rn(d.box_center[2 * b + 1], ofx);
      orow[c++] = __fdiv_rn(d.box_size[b], ofx);
      ok = ok && isfinite(d.box_center[2 * b] + d.box_center[2 * b + 1]) && isfinite(d.box_size[b]);
    }
    orow[c++] = fx;
    for (; c < d.other_ld; ++c) orow[c] = 0.f;
    if (d.img_rowsum) ok = ok && isfinite(d.img_rowsum[b]);
    if (d.scene_rowsum) ok = ok && isfinite(d.scene_rowsum[b]);
    d.finite[b] = ok ? 1 : 0;
    need[b] = all_vis ? 0 : 1;
  }
}

// second pass map: slots in ascending item order.  group > 1: an item needs the pass when ANY item of its group of `group`
// consecutive items does (row tiles of whole body groups for the one-launch loop; the extra passes are computed and never selected)
__global__ __launch_bounds__(1024) void pass_map_kernel(const uint8_t* __restrict__ need, int32_t* __restrict__ mask_slot,
                                                        int32_t* __restrict__ mask_items, int32_t* __restrict__ count, int B, int group) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + t;
    int n = 0;
    if (b < B) {
      const int g0 = b / group * group;
      for (int i = g0; i < g0 + group && i < B; ++i) n |= need[i];
    }
    int x = n;                                            // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = base;
    for (int i = 0; i < w; ++i) off += wsum[i];
    if (b < B) {
      const int slot = off + x - 1;
      mask_slot[b] = n ? slot : -1;
      if (n) mask_items[slot] = b;
    }
    __syncthreads();
    if (t == 1023) base = off + x;
    __syncthreads();
  }
  if (t == 0) *count = base;
}

constexpr int kPackThreads = 256;

__global__ __launch_bounds__(kPackThreads) void pack_outputs_kernel(ehm_pack_desc d) {
  __shared__ int bad_s;
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) bad_s = d.finite ? (d.finite[b] ? 0 : 1) : 0;
  __syncthreads();
  // every row of chk (x_T and the draws that feed a denoiser evaluation; x_t of a single forward) must be finite for the item
  if (d.chk) {
    bool bad = false;
    for (int r = 0; r < d.chk_rows; ++r)
      for (int i = t; i < kPoseDim; i += kPackThreads) bad |= !isfinite(d.chk[((size_t)r * d.B + b) * kPoseDim + i]);
    if (bad) atomicOr(&bad_s, 1);
  }
  __syncthreads();
  const bool bad = bad_s != 0;
  const float nan = __builtin_nanf("");
  if (t == 0 && d.finite_out) d.finite_out[b] = bad ? 0 : 1;
  if (d.x_final)
    for (int i = t; i < kPoseDim; i += kPackThreads) {
      const size_t o = (size_t)b * kPoseDim + i;
      if (bad || (d.last_noise && !isfinite(d.last_noise[o]))) d.x_final[o] = nan;
    }
  if (bad) {
    for (int i = t; i < kPoseDim; i += kPackThreads) { d.x0[(size_t)b * kPoseDim + i] = nan; d.pose6d[(size_t)b * kPoseDim + i] = nan; }
    for (int i = t; i < kJ * 9; i += kPackThreads) d.R[(size_t)b * kJ * 9 + i] = nan;
    for (int i = t; i < d.V * 3; i += kPackThreads) d.verts[(size_t)b * d.V * 3 + i] = nan;
    for (int i = t; i < d.J * 3; i += kPackThreads) d.joints[(size_t)b * d.J * 3 + i] = nan;
  }
  __syncthreads();                                                     // (the block's own global writes, re-read below by other threads)
  __threadfence_block();
  for (int i = t; i < kJ * 9; i += kPackThreads) {
    const float v = bad ? nan : d.R[(size_t)b * kJ * 9 + i];
    if (i < 9) d.global_orient[(size_t)b * 9 + i] = v;
    else d.body_pose[(size_t)b * (kJ - 1) * 9 + (i - 9)] = v;
  }
  for (int i = t; i < 10; i += kPackThreads) d.betas_out[(size_t)b * 10 + i] = bad ? nan : d.betas_in[(size_t)b * 10 + i];
  // egohmr.py:283-301 (float32 op for op: focal = fx * FX_NORM; p = joints + transl; p / p.z; u = f p.x + c p.z; u / 1920 - 0.5)
  const float f = __fmul_rn(d.fx[b], d.fx_norm), ccx = d.cx[b], ccy = d.cy[b];
  if (t == 0) {
    d.focal[2 * b] = f; d.focal[2 * b + 1] = f;
    d.center[2 * b] = ccx; d.center[2 * b + 1] = ccy;
  }
  const float tx = d.transl[3 * b], ty = d.transl[3 * b + 1], tz = d.transl[3 * b + 2];
  for (int j = t; j < d.J; j += kPackThreads) {
    const size_t o = ((size_t)b * d.J + j) * 3;
    const float jx = bad ? nan : d.joints[o], jy = bad ? nan : d.joints[o + 1], jz = bad ? nan : d.joints[o + 2];
    const float px = __fadd_rn(jx, tx), py = __fadd_rn(jy, ty), pz = __fadd_rn(jz, tz);
    d.kp3d_full[o] = px; d.kp3d_full[o + 1] = py; d.kp3d_full[o + 2] = pz;
    const float qx = __fdiv_rn(px, pz), qy = __fdiv_rn(py, pz), qz = __fdiv_rn(pz, pz);
    const float u = __fadd_rn(__fmul_rn(f, qx), __fmul_rn(ccx, qz)), v = __fadd_rn(__fmul_rn(f, qy), __fmul_rn(ccy, qz));
    d.kp2d_full[((size_t)b * d.J + j) * 2] = __fsub_rn(__fdiv_rn(u, 1920.f), 0.5f);
    d.kp2d_full[((size_t)b * d.J + j) * 2 + 1] = __fsub_rn(__fdiv_rn(v, 1080.f), 0.5f);
  }
}

}  // namespace

extern "C" int ehm_item_prep(const ehm_item_prep_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->B > 0 && d->keypoints_2d && d->joint_map && d->NK > 0 && d->fx && d->transl);
  EHM_CHECK_ARG(d->tW1 && d->tb1 && d->tW2 && d->tb2 && d->t_hidden > 0 && d->t_hidden <= 128 && d->t_out > 0);
  EHM_CHECK_ARG(!d->with_cam_center || (d->cx && d->cy));
  EHM_CHECK_ARG(!d->with_bbox || (d->box_center && d->box_size));
  EHM_CHECK_ARG(d->other && d->other_col0 >= 0 && d->other_ld >= d->other_col0 + d->t_out + 1 + (d->with_cam_center ? 2 : 0) + (d->with_bbox ? 3 : 0));
  EHM_CHECK_ARG(d->vis && d->mask_slot && d->mask_items && d->count && d->finite && d->need_scratch && d->pass_group >= 1);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(item_prep_kernel, dim3((unsigned)d->B), dim3(kPrepThreads), 0, st, *d, d->need_scratch);
  EHM_LAUNCH_CHECK();
  hipLaunchKernelGGL(pass_map_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)d->need_scratch, d->mask_slot, d->mask_items, d->count, d->B, d->pass_group);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_pack_outputs(const ehm_pack_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->B > 0 && d->J > 0 && d->V > 0);
  EHM_CHECK_ARG(d->x0 && d->pose6d && d->R && d->verts && d->joints && d->betas_in && d->betas_out);
  EHM_CHECK_ARG(d->transl && d->fx && d->cx && d->cy && d->global_orient && d->body_pose && d->kp3d_full && d->kp2d_full && d->focal && d->center);
  EHM_CHECK_ARG(!d->chk || d->chk_rows > 0);
  hipLaunchKernelGGL(pack_outputs_kernel, dim3((unsigned)d->B), dim3(kPackThreads), 0, (hipStream_t)stream, *d);
  EHM_LAUNCH_CHECK();
  return 0;
}
