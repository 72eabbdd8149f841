// DDPM / DDIM sampler steps and the whole stream-ordered sampling loop for gfx950.
//
// Replaces, for the hot path, diffusion/gaussian_diffusion.py:
//   p_mean_variance :233-276 + q_posterior_mean_variance :209-231   (mean = coef1*x0 + coef2*x)
//   p_sample :298-337, p_sample_with_grad :340-388, ddim_sample :511-556 (+ _predict_eps_from_xstart :286-290)
//   p_sample_loop_progressive :449-508, ddim_sample_loop_progressive :661-718
// The reference's loop is Python with >= 6 tiny host->device copies and ~12 eager kernels per step plus a
// host sync on t[0] in guided steps; here the host enqueues the T steps back to back on one HIP stream
// with every schedule coefficient passed by value (no device-side table, no sync, graph-capturable).
#include <stdarg.h>

#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"
#include "smpl_dev.h"
#include "step_dev.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void ehm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ehm_last_error(void) { return g_err; }
extern "C" const char* ehm_target_arch(void) { return "gfx950"; }
extern "C" const char* ehm_build_features(void) {
  return ""
#ifdef EHM_STAMPS
         "stamps "
#endif
      ;
}

int ehm_num_cus() {   // of the CURRENT device (cached per ordinal)
  static int n[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (n[dev] == 0) {
    hipDeviceProp_t prop;
    n[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;   // MI355X
  }
  return n[dev];
}

// ------------------------------------------------------------------------------------------------ launch-class timing
#include <vector>
namespace {
struct ProfRec { int cls; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec*> g_prof;
unsigned long long* g_prof_evals = nullptr;     // device word: distance evaluations of the collision proxy's search while a profile is open
}  // namespace
unsigned long long* ehm_prof_evals_ptr() { return g_prof_on ? g_prof_evals : nullptr; }
EhmProfScope::EhmProfScope(int cls_, hipStream_t st_) : cls(cls_), st(st_), rec(nullptr) {
  if (!g_prof_on) return;
  ProfRec* r = new ProfRec{cls, nullptr, nullptr};
  if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess || hipEventRecord(r->a, st) != hipSuccess) {
    delete r;
    return;
  }
  rec = r;
}
EhmProfScope::~EhmProfScope() {
  if (!rec) return;
  ProfRec* r = (ProfRec*)rec;
  (void)hipEventRecord(r->b, st);
  g_prof.push_back(r);
}
extern "C" int ehm_profile_begin(void) {
  for (ProfRec* r : g_prof) { (void)hipEventDestroy(r->a); (void)hipEventDestroy(r->b); delete r; }
  g_prof.clear();
  if (!g_prof_evals && hipMalloc(&g_prof_evals, sizeof(unsigned long long)) != hipSuccess) g_prof_evals = nullptr;
  if (g_prof_evals) (void)hipMemset(g_prof_evals, 0, sizeof(unsigned long long));
  g_prof_on = true;
  return 0;
}
extern "C" int ehm_profile_end(double* ms, int64_t* launches, int n) {
  EHM_CHECK_ARG(ms && launches && n > 0 && n <= EHM_PROF_N);
  g_prof_on = false;
  for (int i = 0; i < n; ++i) { ms[i] = 0.0; launches[i] = 0; }
  int rc = 0;
  for (ProfRec* r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r->b) != hipSuccess || hipEventElapsedTime(&t, r->a, r->b) != hipSuccess) rc = EHM_EIO;
    else if (r->cls >= 0 && r->cls < n) { ms[r->cls] += t; launches[r->cls] += 1; }
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
  }
  g_prof.clear();
  if (n > EHM_PROF_G_NEAREST_EVALS && g_prof_evals) {      // a COUNT, not a launch class: distance evaluations of nearest_grid_kernel during the profile
    unsigned long long ev = 0;
    if (hipMemcpy(&ev, g_prof_evals, sizeof(ev), hipMemcpyDeviceToHost) == hipSuccess) launches[EHM_PROF_G_NEAREST_EVALS] = (int64_t)ev;
  }
  if (rc) ehm_set_error("ehm_profile_end: an event of the profile could not be read");
  return rc;
}

namespace {

// torch evaluates these chains as separate rounded float32 ops; keep the same roundings (no FMA contraction)
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ noise,
                                 const float* __restrict__ grad, float* __restrict__ out, float c1, float c2, float logvar,
                                 float nz, float gscale, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float mean = __fadd_rn(__fmul_rn(c1, x0[i]), __fmul_rn(c2, x[i]));                 // :217-220
  if (grad) mean = __fadd_rn(mean, __fmul_rn(gscale, grad[i]));                      // :381 / :385
  const float sd = expf(__fmul_rn(0.5f, logvar));
  out[i] = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sd), noise[i]));                  // :336
}

__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ noise,
                                 float* __restrict__ out, float sr, float srm1, float sap, float dir, float sigma, float nz,
                                 int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(sr, x[i]), x0[i]), srm1);          // :286-290
  const float mean = __fadd_rn(__fmul_rn(x0[i], sap), __fmul_rn(dir, eps));          // :548-551
  out[i] = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sigma), noise[i]));               // :555
}

}  // namespace

extern "C" int ehm_ddpm_step(const float* x, const float* x0, const float* noise, const float* grad, float* x_next, float coef1,
                             float coef2, float log_variance, float nonzero, float grad_scale, int64_t n, void* stream) {
  EHM_CHECK_ARG(x && x0 && noise && x_next && n >= 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x0, noise, grad,
                     x_next, coef1, coef2, log_variance, nonzero, grad_scale, n);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_ddim_step(const float* x, const float* x0, const float* noise, float* x_next, float sqrt_recip_ac,
                             float sqrt_recipm1_ac, float sqrt_ac_prev, float dir_coef, float sigma, float nonzero, int64_t n,
                             void* stream) {
  EHM_CHECK_ARG(x && x0 && noise && x_next && n >= 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x0, noise, x_next,
                     sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_prev, dir_coef, sigma, nonzero, n);
  EHM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ whole loop
namespace {
constexpr int kSkinSegment = 128;      // steps per deferred skinning launch (bounds the per-step transform / fragment slots)
struct Workspace {
  float* X[3];      // activation ping-pong [rows_pad, hid]
  float* x_cur;     // [B,144]
  float* A;         // [B,24,12]
  float* g_verts_in;   // guidance: body decoded from x_t           [B,V,3]
  float* g_joints;     //                                            [B,45,3]
  float* g_R;          //                                            [B,24,9]
  float* g_A;          //                                            [B,24,12]
  float* g_gverts;     // d loss / d verts                           [B,V,3]
  float* g_vposed;     // blended rest vertices of the guided bodies  [B,V,3]  (left by the forward for the skinning VJP)
  float* g_loss;       // [B]
  float* g_gpose;      // [B,144]
  float* g_grad;       // [B,144]
  void* g_scratch;     // bbox / selection / dA / dpose-feature (guidance.hip)
  float* nl_qkv;       // non-local block: [rows, 3 Ci] theta | phi | g
  float* nl_y;         //                  [rows, Ci]   attention output
  // deferred skinning: up to skin_seg steps leave these behind for ONE pose + ONE skinning launch
  float* loop_A;                // [skin_seg, B, 24, 12]
  void* loop_pf;                // [skin_seg, ceil(B/32), 14, 2, 64] x 16 B
  float* loop_verts;            // [B, V, 3]   vertices of the segment's intermediate steps (computed like the final ones, not consumed)
  float* loop_joints;           // [B, J, 3]
  float* loop_x0;               // [skin_seg, B, 144]  x0 of the pending steps (their poses are computed in front of the skinning launch, step.hip)
  int skin_seg;
  int64_t rows, rows_pad;
  int64_t total_bytes;
};

Workspace carve(const ehm_sample_desc* d, int hid, int V, int n_joints, char* base) {
  Workspace w{};
  w.rows = (int64_t)(d->B + (d->passes == 2 ? (d->num_masked >= 0 ? d->num_masked : d->B) : 0)) * kJ;   // virtual bodies after pass pruning
  w.rows_pad = round_up(w.rows, ehm_gcn_row_tile());
  int64_t off = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += round_up(floats * 4, 256);
    return p;
  };
  for (int i = 0; i < 3; ++i) w.X[i] = take(w.rows_pad * hid);
  w.x_cur = take((int64_t)d->B * kPoseDim);
  w.A = take((int64_t)d->B * kJ * 12);
  const bool guided = d->num_scene_points > 0;
  if (guided) {
    w.g_verts_in = take((int64_t)d->B * V * 3);
    w.g_joints = take((int64_t)d->B * n_joints * 3);
    w.g_R = take((int64_t)d->B * kJ * 9);
    w.g_A = take((int64_t)d->B * kJ * 12);
    w.g_gverts = take((int64_t)d->B * V * 3);
    w.g_vposed = take((int64_t)d->B * V * 3);
    w.g_loss = take(d->B);
    w.g_gpose = take((int64_t)d->B * kPoseDim);
    w.g_grad = take((int64_t)d->B * kPoseDim);
    w.g_scratch = take(ehm_guidance_scratch_bytes(d->B, d->num_scene_points) / 4 + 64);
  }
  if (d->nonlocal_ci > 0) {
    w.nl_qkv = take(w.rows * 3 * d->nonlocal_ci);
    w.nl_y = take(w.rows * d->nonlocal_ci);
  }
  // deferred skinning: with lbs_every_step the steps leave their transforms / blend fragments in per-step slots and ONE skinning launch per
  // `skin_seg` steps computes their vertices (the same arithmetic; 14 us per step instead of a 25-30 us launch inside every step)
  if (d->B >= ehm_skin_min_bodies() && d->lbs_every_step) {
    w.skin_seg = d->num_steps < kSkinSegment ? d->num_steps : kSkinSegment;
    w.loop_A = take((int64_t)w.skin_seg * d->B * kJ * 12);
    w.loop_pf = take((int64_t)w.skin_seg * (ehm_skin_pf_bytes_per_step(d->B) / 4));
    w.loop_verts = take((int64_t)d->B * V * 3);
    w.loop_joints = take((int64_t)d->B * n_joints * 3);
    w.loop_x0 = take((int64_t)w.skin_seg * d->B * kPoseDim);
  }
  w.total_bytes = off;
  return w;
}
}  // namespace

extern "C" int64_t ehm_sample_workspace_bytes(const ehm_sample_desc* d, int hid_dim, int num_verts) {
  EHM_CHECK_ARG(d && d->B > 0 && (d->passes == 1 || d->passes == 2) && hid_dim > 0 && num_verts > 0);
  return carve(d, hid_dim, num_verts, 64 + kJ, nullptr).total_bytes;
}

extern "C" int ehm_sample_loop(ehm_gcn* gcn, ehm_smpl* smpl, const ehm_sample_desc* d, const ehm_step_coefs* steps,
                               const float* h_img, const float* h_oth, const uint8_t* vis, const float* Wx, const float* tvecs,
                               const float* noise, const float* scene, const float* betas, const float* mean, const float* std_,
                               float* x_final, float* x0_final, float* verts, float* joints, float* R, float* pose6d,
                               float* trace, void* workspace, int64_t workspace_bytes, void* stream) {
  EHM_CHECK_ARG(gcn && smpl && d && steps && h_img && h_oth && vis && Wx && tvecs && noise && betas && mean && std_);
  EHM_CHECK_ARG(x_final && x0_final && verts && joints && R && pose6d && workspace);
  EHM_CHECK_ARG(d->B > 0 && d->num_steps > 0 && (d->passes == 1 || d->passes == 2));
  EHM_CHECK_ARG(d->passes == 1 || ehm_gcn_virtual_bodies(gcn, d->B, 2) == d->B + (d->num_masked >= 0 ? d->num_masked : d->B));   // desc and ehm_gcn_set_pass_map agree
  const int hid = ehm_gcn_hid(gcn), nh = ehm_gcn_num_hidden(gcn), V = ehm_smpl_num_verts(smpl);
  EHM_CHECK_ARG(nh % 2 == 0);
  bool any_guided = false;
  for (int k = 0; k < d->num_steps; ++k) any_guided |= steps[k].grad_scale != 0.f;
  EHM_CHECK_ARG(!any_guided || (scene && d->num_scene_points > 0));      // (ddim rows with grad_scale != 0: ddim_sample_with_grad, gaussian_diffusion.py:559-614)
  const ehm_nonlocal_params* nlp = ehm_gcn_nonlocal(gcn);
  EHM_CHECK_ARG(d->nonlocal_ci == nlp->Ci);                                            // the descriptor sized the workspace for the block that is set
  EHM_CHECK_ARG(nlp->Ci == 0 || (ehm_gcn_get_precision(gcn) != 2 && d->lowprec_steps == 0));   // the block reads float32 features
  Workspace w = carve(d, hid, V, 64 + kJ, (char*)workspace);
  EHM_CHECK_ARG(workspace_bytes >= w.total_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int B = d->B;
  const int64_t n = (int64_t)B * kPoseDim;

  for (int i = 0; i < 3; ++i)
    if (w.rows_pad > w.rows)
      EHM_HIP(hipMemsetAsync(w.X[i] + w.rows * hid, 0, (size_t)(w.rows_pad - w.rows) * hid * sizeof(float), st));
  EHM_HIP(hipMemcpyAsync(w.x_cur, noise, n * sizeof(float), hipMemcpyDeviceToDevice, st));   // x_T, :476-478

  int rc = 0;
  const int base_prec = ehm_gcn_get_precision(gcn);
  const int lowprec = base_prec == 1 /* f16x3 */ ? d->lowprec_steps : 0;
  struct PrecisionGuard {      // the per-step kernel choice is host-side state of the handle: put it back on EVERY exit path (EHM_HIP returns early)
    ehm_gcn* g;
    int prec;
    bool armed;
    ~PrecisionGuard() { if (armed) ehm_gcn_set_precision(g, prec); }
  } guard{gcn, base_prec, lowprec > 0};
  auto prec_of = [&](int k) { return (lowprec > 0 && k < lowprec) ? 2 : base_prec; };
  // ---- deferred skinning of the per-step launches (see carve): slots filled since the last skinning launch
  // (a body model with dense skinning weights has no MFMA fragments: its steps keep the VALU skinning launch of ehm_step_body_impl - the workspace
  //  was sized without looking at the handle, the slots simply stay unused)
  const bool defer_skin = w.skin_seg > 0 && d->lbs_every_step && ehm_smpl_has_mfma_skin(smpl);
  const int64_t pf_bytes_step = ehm_skin_pf_bytes_per_step(B);
  int pending = 0;
  bool pending_poses = false;       // the pending slots hold x0 only (fused step launches): their poses are still to be computed
  // ---- fused step launches (step.hip): output responses + per-body update + the next step's input conv as ONE launch per step.  Needs the
  //      step's pose off the per-step path: deferred skinning (poses computed per flush), or no per-step skinning at all (then the last step
  //      takes the per-step launches below, which end in the pose and the skinning launch).
  const bool fused_steps = !d->per_step_launches && hid % 64 == 0 && (defer_skin || !d->lbs_every_step) &&
                           w.rows_pad * hid * 4 < ((int64_t)1 << 32);   // (its row loads carry 32-bit byte offsets into the activation matrix)   // (per_step_launches: A/B runs, the bit-equality tests)
  if (defer_skin && B % 32 != 0) EHM_HIP(hipMemsetAsync(w.loop_pf, 0, (size_t)w.skin_seg * pf_bytes_step, st));   // padding bodies of every slot's last 32-body tile
  auto flush_skin = [&](bool final_step_inside) -> int {
    if (pending == 0) return 0;
    int r = 0;
    if (pending_poses)
      r = ehm_pose_steps_impl(smpl, w.loop_x0, pending, final_step_inside ? pending - 1 : -1, B, betas, mean, std_, w.loop_A, w.loop_pf, R, joints, pose6d,
                              x0_final, w.loop_joints, st);
    if (r == 0)
      r = ehm_skin_steps_impl(smpl, w.loop_A, w.loop_pf, pending, final_step_inside ? pending - 1 : -1, B, verts, joints, w.loop_verts, w.loop_joints, st);
    pending = 0;
    pending_poses = false;
    return r;
  };
  bool input_done = false;          // step k's input conv already ran inside step k-1's skinning launch
  for (int k = 0; k < d->num_steps && rc == 0; ++k) {
    const ehm_step_coefs& c = steps[k];
    if (lowprec > 0) ehm_gcn_set_precision(gcn, prec_of(k));   // host-side kernel choice only; same X2 buffers
    const bool last = k == d->num_steps - 1;
    if (trace) EHM_HIP(hipMemcpyAsync(trace + (int64_t)k * n, w.x_cur, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    // ---- collision guidance on x_t (gaussian_diffusion.py:378-385, egohmr.py:517-570): depends on x_t and betas only, so it runs first
    const float* grad = nullptr;
    if (c.grad_scale != 0.f) {
      EhmProfScope ps(EHM_PROF_GUIDANCE, st);
      rc = ehm_guidance_impl(smpl, betas, w.x_cur, mean, std_, scene, B, d->num_scene_points, d->tau, d->guide_denom, d->guide_all_points ? d->tau : 0.f,
                             w.g_verts_in, w.g_joints, w.g_R, w.g_A, w.g_gverts, w.g_loss, w.g_gpose, w.g_grad, w.g_scratch, st, w.g_vposed);
      grad = w.g_grad;
    }
    // ---- denoiser: EgoHMR.forward's per-step part (egohmr.py:232-257): input conv, chained hidden convs, output conv responses ----
    if (rc == 0 && !input_done) {
      EhmProfScope ps(EHM_PROF_INPUT, st);
      rc = ehm_gcn_input_layer(gcn, h_img, h_oth, vis, w.x_cur, Wx, tvecs + (int64_t)k * 2 * hid, w.X[0], B, d->passes, st);
    }
    input_done = false;
    int in = 0;
    if (rc == 0) {
      const int p = prec_of(k);
      EhmProfScope ps(p == 1 ? EHM_PROF_CHAIN_F16X3 : p == 2 ? EHM_PROF_CHAIN_F16 : EHM_PROF_HIDDEN_F32, st);
      rc = ehm_gcn_hidden_stack(gcn, w.X, w.rows_pad, &in, st);
    }
    const float* hs = nullptr;
    const void* out_dev = nullptr;
    const float* feat = w.X[in];
    if (rc == 0 && nlp->Ci > 0) {
      // optional non-local block (modulated_gcn.py:104-110): z = BN(W (softmax(theta phi^T) g)) + x over the 24 joints of a body; float32 features in X[in]
      // (the last hidden conv writes float32), result into X[1] (free between the chain and the next step's convs)
      const int ci = nlp->Ci, nrows = (int)w.rows;
      ehm_conv_desc c1{feat, nlp->Wqkv, nlp->bqkv, nullptr, w.nl_qkv, nrows, 1, 1, hid, 3 * ci, 1, 1, 1, 0, 0, nlp->qkv_scale};
      rc = ehm_conv_nhwc_split(&c1, st);
      if (rc == 0) rc = ehm_nonlocal_attention(w.nl_qkv, w.nl_y, w.rows / kJ, ci, st);
      ehm_conv_desc c2{w.nl_y, nlp->Wo, nlp->bo, feat, w.X[1], nrows, 1, 1, ci, hid, 1, 1, 1, 0, 0, nlp->o_scale};
      if (rc == 0) rc = ehm_conv_nhwc_split(&c2, st);
      feat = w.X[1];
    }
    // ---- fused: responses + x0 + x_{t-1} + the next step's input conv in one launch (the same arithmetic as the launches below).  Not for a
    //      step whose successor reads another activation format (the next rows would land on other bodies' rows of `feat`), nor - without
    //      deferred skinning - for the last step (its pose and skinning follow at once).
    if (rc == 0 && fused_steps && (last ? defer_skin : prec_of(k) == prec_of(k + 1))) {
      if (defer_skin && pending > 0 && !pending_poses) rc = flush_skin(false);      // (slots filled by per-step launches carry their poses already)
      const float* eps = noise + (int64_t)(1 + k) * n;
      float* dst = last ? x_final : w.x_cur;
      GcnInputArgs nin;
      if (rc == 0 && !last) rc = ehm_gcn_input_args(gcn, h_img, h_oth, vis, dst, Wx, tvecs + (int64_t)(k + 1) * 2 * hid, w.X[0], B, d->passes, &nin);
      if (rc == 0)
        rc = ehm_step_fused_impl(ehm_gcn_out_dev(gcn), feat, prec_of(k), vis, w.x_cur, eps, grad, dst, defer_skin ? w.loop_x0 + (int64_t)pending * n : x0_final,
                                 &c, d->ddim, d->passes, ehm_gcn_mask_slot(gcn, d->passes), B, last ? nullptr : &nin, prec_of(k + 1), st);
      input_done = !last;
      if (rc == 0 && defer_skin) {
        pending_poses = true;
        ++pending;
        if (pending == w.skin_seg || last) rc = flush_skin(last);
      }
      continue;
    }
    if (rc == 0 && pending_poses) rc = flush_skin(false);        // this step fills its slot WITH its pose
    if (rc == 0) {
      EhmProfScope ps(EHM_PROF_OUT_DOT, st);
      rc = ehm_gcn_output_dot_impl(gcn, feat, B, d->passes, &hs, &out_dev, st);
    }
    // ---- per body, one launch: output-conv mix + visibility fuse -> x0 (egohmr.py:247-256), x_{t-1} (gaussian_diffusion.py:298-337 /
    //      :511-556), de-normalise + rot6d + kinematic chain (egohmr.py:258-260); then the skinning launch (egohmr.py:276) ----
    //      When the step is not the last one, the NEXT step's input conv (it needs x_{t-1} only) rides in the skinning launch.
    if (rc == 0) {
      const float* eps = noise + (int64_t)(1 + k) * n;
      float* dst = last ? x_final : w.x_cur;
      GcnInputArgs nin;
      const GcnInputArgs* pnin = nullptr;
      if (!last && d->lbs_every_step && !defer_skin) {   // X[0] is free: the chain has consumed it and, if its result landed there, so has the output conv
        if (lowprec > 0) ehm_gcn_set_precision(gcn, prec_of(k + 1));   // the activation format the next step's convs will read
        rc = ehm_gcn_input_args(gcn, h_img, h_oth, vis, dst, Wx, tvecs + (int64_t)(k + 1) * 2 * hid, w.X[0], B, d->passes, &nin);
        if (lowprec > 0) ehm_gcn_set_precision(gcn, prec_of(k));
        pnin = &nin;
      }
      int fused = 0;
      if (rc == 0 && defer_skin) {
        // the step leaves its transforms and fragments in slot `pending`; no skinning (and no fused input conv) launch now
        rc = ehm_step_body_impl(smpl, hs, out_dev, vis, w.x_cur, eps, grad, dst, x0_final, &c, d->ddim, d->passes, ehm_gcn_mask_slot(gcn, d->passes), 1,
                                betas, mean, std_, verts, joints, R, w.loop_A + (int64_t)pending * B * kJ * 12, pose6d, B, st, nullptr, 0, nullptr,
                                (char*)w.loop_pf + (int64_t)pending * pf_bytes_step);
        ++pending;
        if (rc == 0 && (pending == w.skin_seg || last)) rc = flush_skin(last);
      } else if (rc == 0) {
        rc = ehm_step_body_impl(smpl, hs, out_dev, vis, w.x_cur, eps, grad, dst, x0_final, &c, d->ddim, d->passes, ehm_gcn_mask_slot(gcn, d->passes),
                                (d->lbs_every_step || last) ? 1 : 0,
                                betas, mean, std_, verts, joints, R, w.A, pose6d, B, st, pnin, prec_of(k + 1), &fused);
      }
      input_done = fused != 0;
    }
  }
  return rc;      // (PrecisionGuard restores base_prec)
}
