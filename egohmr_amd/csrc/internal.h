// Cross-translation-unit internals of libegohmr_hip (not part of the C ABI).
#pragma once
#include "common.h"
#include "egohmr_hip.h"

// smpl.hip
int ehm_smpl_forward_impl(ehm_smpl* h, const float* betas, const float* rot_or_x, bool from_rot6d, const float* mean,
                          const float* std_, float* verts, float* joints, float* Rws, float* Aws, float* pose6d_out, int B,
                          hipStream_t st, float* vposed = nullptr);   // vposed [B,V,3]: the blended rest vertices, written by the matrix-core skinning only:
int ehm_smpl_writes_vposed(const ehm_smpl* h, int B);             //   1 when a forward of B bodies takes that path
// rot6d -> R, joint regression, kinematic chain only (no skinning): R [B,24,9], A [B,24,12], joints24 into jws [B,(24+n_extra),3]
int ehm_smpl_pose_impl(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std_, float* Rws, float* Aws,
                       float* jws, int B, hipStream_t st);
// output-conv mix + sampler update + pose chain + blend-coefficient fragments in one launch, then skinning (sampling loop only)
int ehm_step_body_impl(ehm_smpl* h, const float* hs, const void* out_dev, const uint8_t* vis, const float* x, const float* noise,
                       const float* grad, float* x_next, float* x0, const ehm_step_coefs* c, int ddim, int passes, const int32_t* mask_slot, int do_pose,
                       const float* betas, const float* mean, const float* std_, float* verts, float* joints, float* Rws, float* Aws,
                       float* pose6d, int B, hipStream_t st, const struct GcnInputArgs* next_input = nullptr, int next_prec = 0,
                       int* fused = nullptr, void* defer_pf = nullptr);   // defer_pf: write the blend fragments there and launch NO skinning (ehm_skin_steps_impl later)
int ehm_skin_steps_impl(ehm_smpl* h, const float* A_steps, const void* pf_steps, int nsteps, int final_step, int B, float* verts, float* joints,
                        float* scratch_verts, float* scratch_joints, hipStream_t st);
// step.hip: a step's tail (output responses, x0, x_{t-1}) + the next step's input conv in one launch; the poses of pending steps in one launch
int ehm_step_fused_impl(const void* out_dev, const float* X, int prec, const uint8_t* vis, const float* x, const float* noise, const float* grad,
                        float* x_next, float* x0, const ehm_step_coefs* c, int ddim, int passes, const int32_t* mask_slot, int B,
                        const struct GcnInputArgs* next_input, int next_prec, hipStream_t st);
int ehm_pose_steps_impl(ehm_smpl* smpl, const float* x0_steps, int nsteps, int final_step, int B, const float* betas, const float* mean,
                        const float* std_, float* A_steps, void* pf_steps, float* R, float* joints, float* pose6d, float* x0_final,
                        float* scratch_joints, hipStream_t st);
int ehm_skin_min_bodies();
int ehm_smpl_has_mfma_skin(const ehm_smpl* h);   // 0: a body model with more than four skinning weights per vertex (no MFMA fragments): VALU skinning inside every step
int64_t ehm_skin_pf_bytes_per_step(int B);
void ehm_smpl_dev(const ehm_smpl* h, void* out);   // copies the handle's SmplDev (smpl_dev.h) into *out
size_t ehm_smpl_dev_size();
int ehm_smpl_num_verts(const ehm_smpl* h);
int ehm_smpl_num_extra(const ehm_smpl* h);
// gcn.hip
struct GcnInputArgs;
int ehm_gcn_input_args(ehm_gcn* h, const float* h_img, const float* h_oth, const uint8_t* vis, const float* x, const float* Wx, const float* tvec,
                       float* out, int B, int passes, GcnInputArgs* a);
int ehm_gcn_hid(const ehm_gcn* h);
int ehm_gcn_nonlocal_ci(const ehm_gcn* h);
const ehm_nonlocal_params* ehm_gcn_nonlocal(const ehm_gcn* h);
int ehm_gcn_num_hidden(const ehm_gcn* h);
int ehm_gcn_chain_enabled(const ehm_gcn* h);   // the hidden convs run as chained launches in the handle's precision (f16 modes, EHM_F16_CHAIN != 0)
int ehm_gcn_virtual_bodies(const ehm_gcn* h, int B, int passes);   // B + second passes after pruning (ehm_gcn_set_pass_map)
const int32_t* ehm_gcn_mask_slot(const ehm_gcn* h, int passes);
// output conv, first half only: responses hs [passes*B*24, 12] = X . [W0 | W1] (scratch owned by the handle); *out_dev = the OutDev block
int ehm_gcn_output_dot_impl(ehm_gcn* h, const float* X, int B, int passes, const float** hs, const void** out_dev, hipStream_t st);
const void* ehm_gcn_out_dev(const ehm_gcn* h);   // the OutDev block (gcn_dev.h) of the output conv
// sampler.hip
int ehm_num_cus();   // multiProcessorCount of the current device (cached)
// gcn_tile.hip (f16 matrix-core hidden convs: 'f16x3' split operands and plain 'f16')
int ehm_gcn_tile_layer_impl(const ehm_gcn* h, int layer, const void* X, const void* residual, void* out, int64_t rows_pad, bool out_f32,
                            hipStream_t st);
int ehm_gcn_tile_chain_impl(ehm_gcn* h, void* const bufs[3], int64_t rows_pad, hipStream_t st);
void ehm_pack_half(const float* X, void* Y, size_t n, float scale, hipStream_t st);
// gcn.hip
int ehm_gcn_reserve_rows(ehm_gcn* h, int64_t rows_pad);   // sync words + output-conv scratch for up to rows_pad rows (allocates: not inside a capture)
// guidance.hip
int64_t ehm_guidance_scratch_bytes(int B, int N);
int ehm_guidance_impl(ehm_smpl* smpl, const float* betas, const float* x, const float* mean, const float* std_,
                      const float* scene, int B, int N, float tau, float denom, float margin, float* verts_ws, float* joints_ws, float* R_ws,
                      float* A_ws, float* gverts, float* loss, float* gpose, float* grad, void* scratch, hipStream_t st,
                      float* vposed_ws = nullptr);   // [B,V,3] scratch or nullptr: the forward's blended rest vertices for the skinning VJP (else recomputed there)
// sampler.hip: launch-class timing for bench.py (ehm_profile_begin / ehm_profile_end); a no-op unless a profile is open
unsigned long long* ehm_prof_evals_ptr();   // device counter of the collision search's distance evaluations, or nullptr when no profile is open
struct EhmProfScope {
  int cls;
  hipStream_t st;
  void* rec;
  EhmProfScope(int cls_, hipStream_t st_);
  ~EhmProfScope();
};
