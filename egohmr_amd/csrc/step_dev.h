// Per-body step of the sampling loop (output-conv mix + sampler update + pose chain + blend-coefficient fragments), shared by smpl.hip
// (step_body_kernel: one launch per step) and gcn_tile.hip (the one-launch loop runs it as a work item of the persistent kernel).
#pragma once
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "smpl_dev.h"

typedef _Float16 sk_half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ pose + chain
// One wave per body, lane = joint.  `Rl` (may be nullptr): LDS copy [24][9] of the rotations for a caller that goes on to pack them.
template <bool FROM_ROT6D>
__device__ __forceinline__ void pose_chain_body(int b, int lane, const float* __restrict__ betas, const float* rot_or_x /* this body's row */,
                                                const float* __restrict__ mean, const float* __restrict__ std_, const SmplDev& S,
                                                float* __restrict__ Rws, float* __restrict__ Aout, float* __restrict__ joints,
                                                float* __restrict__ pose6d_out, int joints_stride, float* Rl) {
  const int j = lane < kJ ? lane : 0;
  float R[9];
  if (FROM_ROT6D) {
    float p[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int e = j * 6 + c;
      p[c] = rot_or_x[e] * std_[e] + mean[e];                          // egohmr.py:258
      if (pose6d_out && lane < kJ) pose6d_out[(size_t)b * kPoseDim + e] = p[c];
    }
    rot6d_to_R(p[0], p[2], p[4], p[1], p[3], p[5], R);                 // 'diffusion' layout
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rot_or_x[j * 9 + k];
  }
  // joint regression: J = J_template + J_shape . beta
  float Jx[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 10; ++l) s = fmaf(S.J_shape[j * 30 + c * 10 + l], betas[(size_t)b * 10 + l], s);
    Jx[c] = S.J_template[j * 3 + c] + s;
  }
  const int par = S.tree.parent[j];
  const int plane = par < 0 ? 0 : par;
  float t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float pj = __shfl(Jx[c], plane);
    t[c] = par < 0 ? Jx[c] : Jx[c] - pj;     // rel_joints
  }
  // G = [R | t] for the root; children: G = G_parent * [R | t]
  float G[12];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    G[r * 4 + 0] = R[r * 3 + 0]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2]; G[r * 4 + 3] = t[r];
  }
  const int my_depth = S.tree.depth[j];
  for (int d = 1; d <= S.tree.max_depth; ++d) {
    float P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = __shfl(G[k], plane);
    if (my_depth == d) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float p0 = P[r * 4 + 0], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2], p3 = P[r * 4 + 3];
        G[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
        G[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
        G[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
        G[r * 4 + 3] = p0 * t[0] + p1 * t[1] + p2 * t[2] + p3;
      }
    }
  }
  if (lane >= kJ) return;
  const size_t o = (size_t)b * kJ + j;
  if (Rws) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rws[o * 9 + k] = R[k];
  }
  if (Rl) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rl[j * 9 + k] = R[k];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    joints[(size_t)b * joints_stride + j * 3 + r] = G[r * 4 + 3];           // posed joint = chain translation
    const float gj = G[r * 4 + 0] * Jx[0] + G[r * 4 + 1] * Jx[1] + G[r * 4 + 2] * Jx[2];
    Aout[o * 12 + r * 4 + 0] = G[r * 4 + 0];
    Aout[o * 12 + r * 4 + 1] = G[r * 4 + 1];
    Aout[o * 12 + r * 4 + 2] = G[r * 4 + 2];
    Aout[o * 12 + r * 4 + 3] = G[r * 4 + 3] - gj;                              // rel_transforms
  }
}


struct StepBodyArgs {
  const float* hs;          // [passes*B*24, 12] responses of the output conv (gcn_out_dot_kernel)
  OutDev O;
  const uint8_t* vis;       // [B,24]
  const float* x;           // x_t [B,144]
  const float* noise;       // [B,144]
  const float* grad;        // [B,144] or nullptr
  float* x_next;            // may alias x
  float* x0;                // [B,144]
  ehm_step_coefs c;
  int ddim, passes, B, do_pose;
  const int32_t* mask_slot; // pass pruning (ehm_gcn_set_pass_map) or nullptr
  const float *betas, *mean, *std_;
  float *Rws, *Aws, *joints, *pose6d;
  int jstride;
  sk_half8* pf;             // nullptr: VALU skinning path (B < 24), no fragments
  float* trace;             // [B,144] or nullptr: receives x_t as this step read it
};


struct StepBodyLds {
  float sh[2][kJ][12];
  float x0s[kPoseDim];
  float Rl[kJ * 9];
  float sAo[kJ * kJ], sMo[kJ * 6];
  float xn[kPoseDim];       // x_{t-1} as written to x_next (a caller that goes on with the next step's input conv reads it here)
};

// One wave = one body `b`.  `sync` orders the wave's LDS traffic: __syncthreads() in the 64-thread kernel, a wave-local fence when several
// waves of a larger block run different bodies side by side.
// The body's pose from its x0 (LDS or global row `x0row`): de-normalise, rot6d -> R, kinematic chain, then the blend-coefficient fragments.
template <class Sync>
__device__ __forceinline__ void step_pose_part(int b, int lane, const StepBodyArgs& a, const SmplDev& S, StepBodyLds& L, Sync sync, const float* x0row) {
  float (&Rl)[kJ * 9] = L.Rl;
  pose_chain_body<true>(b, lane, a.betas, x0row, a.mean, a.std_, S, a.Rws, a.Aws, a.joints, a.pose6d, a.jstride, Rl);
  if (!a.pf) return;
  sync();
  if (lane < 2 * kBlendSteps) {                                 // 28 lanes: (k-step s, lane half h) -> 8 coefficients, hi and lo fragments
    const int s = lane >> 1, h = lane & 1;
    sk_half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * h + e;
      float x = 0.f;
      if (k < kPoseBasis) x = Rl[9 + k] - ((k % 9 == 0 || k % 9 == 4 || k % 9 == 8) ? 1.f : 0.f);   // R[1:] - I
      else if (k < kPoseBasis + 10) x = a.betas[(size_t)b * 10 + (k - kPoseBasis)];
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
    const size_t base = ((size_t)(b >> 5) * kBlendSteps + s) * 2;
    a.pf[(base + 0) * 64 + (b & 31) + 32 * h] = hi;
    a.pf[(base + 1) * 64 + (b & 31) + 32 * h] = lo;
  }
}

// HS_LDS: the caller has already put the body's responses into L.sh (the fused step kernel of step.hip computes them in the same block).
// Stage the body's output-conv responses (unless HS_LDS: the caller computes them into L.sh) and the (tiny) adjacency / modulation tables: every
// global load of the step's first phase is requested before the first one is consumed (as an element-wise loop this was nine dependent round
// trips, and the mix fetched its coefficients from global memory inside the inner loop).  The caller orders this against step_body_one's reads.
template <bool HS_LDS>
__device__ __forceinline__ void step_stage_tables(int b, int lane, const StepBodyArgs& a, StepBodyLds& L) {
  float (&sh)[2][kJ][12] = L.sh;
  float (&sAo)[kJ * kJ] = L.sAo;
  float (&sMo)[kJ * 6] = L.sMo;
  const int slot = a.mask_slot ? a.mask_slot[b] : b;            // row block of my second pass: B + slot (slot < 0: pruned, every joint visible)
  float tmp[9], ta[9], tm[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int i = lane + 64 * k, p = i / (kJ * 12), rem = i % (kJ * 12);
    tmp[k] = (!HS_LDS && i < a.passes * kJ * 12 && !(p == 1 && slot < 0)) ? a.hs[((size_t)(p ? a.B + slot : b) * kJ) * 12 + rem] : 0.f;
    ta[k] = a.O.A[i < kJ * kJ ? i : 0];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) tm[k] = a.O.M[lane + 64 * k < kJ * 6 ? lane + 64 * k : 0];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int i = lane + 64 * k;
    if (!HS_LDS && i < a.passes * kJ * 12) (&sh[0][0][0])[i] = tmp[k];
    if (i < kJ * kJ) sAo[i] = ta[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (lane + 64 * k < kJ * 6) sMo[lane + 64 * k] = tm[k];
}

// WITH_POSE = false: the pose part is not even compiled in (the caller leaves it to pose_steps_kernel, step.hip).  STAGED: the caller has run
// step_stage_tables itself (and ordered it).
// (wi, nw): the body's 144 elements are dealt out to `nw` cooperating waves, this one being number `wi` (the arithmetic per element does not
// depend on who computes it; a caller with nw > 1 orders the waves' LDS results itself and must not ask for the pose).
template <bool HS_LDS = false, bool WITH_POSE = true, bool STAGED = false, class Sync>
__device__ __forceinline__ void step_body_one(int b, int lane, const StepBodyArgs& a, const SmplDev& S, StepBodyLds& L, Sync sync, int wi = 0, int nw = 1) {
  float (&sh)[2][kJ][12] = L.sh;
  float (&x0s)[kPoseDim] = L.x0s;
  float (&sAo)[kJ * kJ] = L.sAo;
  float (&sMo)[kJ * 6] = L.sMo;
  if constexpr (!STAGED) step_stage_tables<HS_LDS>(b, lane, a, L);
  sync();
  for (int e = lane + 64 * wi; e < kPoseDim; e += 64 * nw) {
    const int j = e / 6, c = e % 6;
    const int p = (a.passes == 2 && !a.vis[(size_t)b * kJ + j]) ? 1 : 0;          // egohmr.py:249-254
    const float s = sAo[j * kJ + j] * (sMo[j * 6 + c] * sh[p][j][c]);
    float t = 0.f;
    for (int jp = 0; jp < kJ; ++jp)
      if (jp != j) t = fmaf(sAo[j * kJ + jp], sMo[jp * 6 + c] * sh[p][jp][6 + c], t);
    const float x0 = s + t + a.O.bias[c];
    const size_t i = (size_t)b * kPoseDim + e;
    a.x0[i] = x0;
    x0s[e] = x0;
    const float xv = a.x[i], nz = a.noise[i];
    if (a.trace) a.trace[i] = xv;
    float out;
    if (a.ddim) {
      float x0u = x0;                                         // x0 the UPDATE runs on (a.x0 / the pose keep the model's own x0, as the reference's other_outputs do)
      if (a.grad) {
        // ddim_sample_with_grad, gaussian_diffusion.py:580-592 (the last four respaced steps): eps -= sqrt(1 - alpha_bar) * grad * 1.0, x0 re-derived from it;
        // c.grad_scale carries float32 sqrt(1 - alpha_bar); one rounding per torch op
        float e1 = __fdiv_rn(__fsub_rn(__fmul_rn(a.c.sqrt_recip_ac, xv), x0), a.c.sqrt_recipm1_ac);       // :582 _predict_eps_from_xstart
        e1 = __fsub_rn(e1, __fmul_rn(a.c.grad_scale, a.grad[i]));                                          // :585-586
        x0u = __fsub_rn(__fmul_rn(a.c.sqrt_recip_ac, xv), __fmul_rn(a.c.sqrt_recipm1_ac, e1));             // :587 _predict_xstart_from_eps
      }
      const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(a.c.sqrt_recip_ac, xv), x0u), a.c.sqrt_recipm1_ac);
      const float mean = __fadd_rn(__fmul_rn(x0u, a.c.sqrt_ac_prev), __fmul_rn(a.c.dir_coef, eps));
      out = __fadd_rn(mean, __fmul_rn(__fmul_rn(a.c.nonzero, a.c.sigma), nz));
    } else {
      float mean = __fadd_rn(__fmul_rn(a.c.coef1, x0), __fmul_rn(a.c.coef2, xv));
      if (a.grad) mean = __fadd_rn(mean, __fmul_rn(a.c.grad_scale, a.grad[i]));
      const float sd = expf(__fmul_rn(0.5f, a.c.log_variance));
      out = __fadd_rn(mean, __fmul_rn(__fmul_rn(a.c.nonzero, sd), nz));
    }
    a.x_next[i] = out;
    L.xn[e] = out;
  }
  if constexpr (WITH_POSE) {
    if (!a.do_pose) return;
    sync();
    step_pose_part(b, lane, a, S, L, sync, x0s);
  }
}
