// What lies between two chained hidden-conv launches of the sampling loop, as ONE launch with one block per body:
//     tail of step t : output-conv responses of the body's rows (gcn_dev.h: gcn_out_dot_quarter) -> output mix + visibility fuse -> x0,
//                      sampler update x_t -> x_{t-1} (step_dev.h: step_body_one)
//     head of step t+1: the hoisted input conv of the body's rows in both passes (gcn_dev.h: gcn_input_body), fed with x_{t-1} from LDS
// (models/egohmr/egohmr.py:232-257 around diffusion/gaussian_diffusion.py:298-337 / :511-556).  Everything here is per body: the per-step
// launches it replaces (gcn_out_dot_kernel 15 us, step_body_kernel 16 us, gcn_input_kernel 31 us at 256 bodies, with three launch boundaries)
// were each a single dependency chain of global-memory latencies with the chip mostly idle.  The pose of the step (rot6d -> R -> kinematic
// chain -> blend fragments), which only the deferred skinning consumes, is NOT on this path any more: the step leaves its x0 in a slot and
// pose_steps_kernel computes the poses of all pending steps in one launch in front of the skinning launch (sampler.hip: flush_skin).
// The arithmetic per output is that of the kernels replaced (same device functions, same operation order): results are bit-equal.
#include "common.h"
#include "egohmr_hip.h"
#include "gcn_dev.h"
#include "internal.h"
#include "smpl_dev.h"
#include "step_dev.h"

namespace {

struct StepFusedArgs {
  const float* X;          // result of the step's last hidden conv: float32 rows (f32 / split-f16 modes) or f16 rows, [rows, K]
  StepBodyArgs sb;         // hs unused (the responses stay in LDS), do_pose = 0
  GcnInputArgs in;         // the NEXT step's input conv (x unused); only read when do_input
  int do_input;
};

constexpr int kFusedT = kJ * 256;              // floats of one 256-channel group's transpose tile

// NT threads per block = NT / 256 groups of 256 channels for the input conv, NT / 64 waves for the 12 (row tile, K quarter) pairs of the responses.
//   NT = 1024 (batches of up to one body per CU): the round-4 shape - with one body per CU its chain of phases IS the launch's duration, and four
//             groups walk the body's 8 input-conv units in two rounds (LDS 96 KiB: one block per CU);
//   NT = 512  (bigger batches): 48 KiB of LDS and 128 registers let TWO blocks share a CU, so that one body's global-memory round trips hide behind
//             another's vector-ALU work (1280 bodies: 297 -> 254 us per launch);
//   NT = 256  (more than two bodies per CU): FOUR independent bodies per CU - the same 16 waves, barriers among four waves instead of eight, a body's eight
//             input-conv units one after the other (1280 bodies: 246 -> 222 us per launch, profiles/r06l_step_fused_nt256_ab.txt).
// (Two half-blocks per body - each with half of the input-conv units, both recomputing the responses - were built too: 62 -> 70 us at 256 bodies,
//  and a race: the chain's result and the next input rows share one buffer, which is only safe while ONE block reads a body's rows before it
//  overwrites them.)  The update runs on three waves (48 elements each) instead of one.
// Per output the arithmetic is that of the per-step kernels (same device functions, same order): bit-equal.
template <bool HALF_IN, int OUT, int NT>
__global__ __launch_bounds__(NT, 4) void step_fused_kernel(StepFusedArgs a) {
  constexpr int G = NT / 256, NWV = NT / 64;                        // 256-thread groups, waves
  __shared__ __attribute__((aligned(16))) float T[G * kFusedT];     // G transpose tiles (24 KiB each); the response partials alias the first 12 KiB
  __shared__ __attribute__((aligned(16))) StepBodyLds L;          // (L.xn is read back as 16-byte vectors)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, B = a.sb.B;
  const int slot = a.sb.passes == 2 ? (a.sb.mask_slot ? a.sb.mask_slot[b] : b) : -1;   // second pass of this body: rows of virtual body B + slot (< 0: none)
  const int nrows = slot >= 0 ? 2 * kJ : kJ;

  // ---- responses of the body's rows: 16-row tiles over [pass 0 rows | pass 1 rows], four K quarters per tile: 12 tasks
  float (*part)[4][16][16] = (float (*)[4][16][16])T;
  for (int task = wave; task < 12; task += NWV) {
    const int tile = task >> 2, wq = task & 3;
    if (16 * tile < nrows) {
      int rr = 16 * tile + (lane & 15);
      if (rr >= nrows) rr = nrows - 1;                                   // rows past the body's: recompute its last row, never stored
      const int64_t r = (int64_t)(rr >= kJ ? B + slot : b) * kJ + (rr % kJ);
      const f32x4 acc = gcn_out_dot_quarter<HALF_IN, 0>(a.X, a.sb.O, r, wq, lane);
#pragma unroll
      for (int c = 0; c < 4; ++c) part[tile][wq][4 * (lane >> 4) + c][lane & 15] = acc[c];
    }
  }
  if (wave == NWV - 1) step_stage_tables<true>(b, lane, a.sb, L);   // the update's tables arrive while the responses are computed (the last wave has at most one task)
  __syncthreads();
  for (int i = tid; i < 2 * kJ * 12; i += NT) {
    const int rr = i / 12, cc12 = i % 12, tile = rr >> 4, tr = rr & 15;
    (&L.sh[0][0][0])[i] = rr < nrows ? (part[tile][0][tr][cc12] + part[tile][1][tr][cc12]) + (part[tile][2][tr][cc12] + part[tile][3][tr][cc12]) : 0.f;
  }
  __syncthreads();
  // ---- x0 and x_{t-1} of the body (three waves, 48 elements each; the step's pose is left to pose_steps_kernel)
  if (wave < 3) {
    SmplDev unused;                                                        // (WITH_POSE = false: never read)
    step_body_one<true, false, true>(b, lane, a.sb, unused, L, [] {}, wave, 3);
  }
  if (!a.do_input) return;
  __syncthreads();                                                         // L.xn is complete
  // ---- the next step's input conv of the body: units (pass, 256-channel block), one per 256-thread group and round
  //      (one flat loop with one call site: as two nested loops hipcc kept both passes' tables live and spilled 90 registers)
  //      a group keeps its channel block for both passes (the second pass finds the block's tables in the caches)
  const int ny = a.in.ny, np = slot >= 0 ? 2 : 1, grp = tid >> 8;
  const int rounds = np * ((ny + G - 1) / G);                             // block-uniform: every thread meets every barrier
  for (int k = 0; k < rounds; ++k) {
    const int cb = grp + G * (k / np), p = k % np;
    if (cb < ny) gcn_input_body<OUT, true>(T + grp * kFusedT, tid & 255, p ? B + slot : b, cb, a.in, L.xn);   // (one __syncthreads inside)
    else __syncthreads();
    __syncthreads();                                                      // the tile is read back after that barrier: keep the next round's writes behind it
  }
}

struct PoseStepsArgs {
  const float* x0_steps;      // [nsteps, B, 144]
  const float *betas, *mean, *std_;
  float* A_steps; char* pf_steps; int64_t pf_bytes;
  float *R, *joints, *pose6d, *x0_final, *scratch_joints;
  int B, final_step, jstride;
};

// The poses of `nsteps` steps' bodies from their x0 (block = one body of one step, one wave): what step_body_kernel did inside every step.
__global__ __launch_bounds__(64) void pose_steps_kernel(PoseStepsArgs p, SmplDev S) {
  __shared__ StepBodyLds L;
  const int b = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
  const bool fin = s == p.final_step;
  StepBodyArgs a{};
  a.betas = p.betas; a.mean = p.mean; a.std_ = p.std_; a.B = p.B; a.jstride = p.jstride;
  a.Rws = fin ? p.R : nullptr;
  a.joints = fin ? p.joints : p.scratch_joints;                          // (intermediate steps: written, never consumed)
  a.pose6d = fin ? p.pose6d : nullptr;
  a.Aws = p.A_steps + (size_t)s * p.B * kJ * 12;
  a.pf = (sk_half8*)(p.pf_steps + (size_t)s * p.pf_bytes);
  const float* x0row = p.x0_steps + ((size_t)s * p.B + b) * kPoseDim;
  for (int e = lane; e < kPoseDim; e += 64) {
    const float v = x0row[e];
    L.x0s[e] = v;
    if (fin) p.x0_final[(size_t)b * kPoseDim + e] = v;
  }
  __syncthreads();
  step_pose_part(b, lane, a, S, L, [] { __syncthreads(); }, L.x0s);
}

template <bool HALF_IN, int NT>
void launch_fused2(int next_prec, int B, hipStream_t st, const StepFusedArgs& a) {
  const dim3 grid((unsigned)B), blk(NT);
  if (next_prec == EHM_PREC_F32) hipLaunchKernelGGL((step_fused_kernel<HALF_IN, 0, NT>), grid, blk, 0, st, a);
  else if (next_prec == EHM_PREC_F16X3) hipLaunchKernelGGL((step_fused_kernel<HALF_IN, 1, NT>), grid, blk, 0, st, a);
  else hipLaunchKernelGGL((step_fused_kernel<HALF_IN, 2, NT>), grid, blk, 0, st, a);
}
template <bool HALF_IN>
void launch_fused(int next_prec, int B, hipStream_t st, const StepFusedArgs& a) {
  if (B <= ehm_num_cus()) launch_fused2<HALF_IN, 1024>(next_prec, B, st, a);     // one body per CU: the wide block
  else if (B > 2 * ehm_num_cus()) launch_fused2<HALF_IN, 256>(next_prec, B, st, a);   // four blocks per CU (round 6: 1280 bodies 246 -> 222 us, same box)
  else launch_fused2<HALF_IN, 512>(next_prec, B, st, a);                          // two blocks per CU
}

}  // namespace

// The tail of a step + the head of the next one (next_input == nullptr: the loop's last step).  `X` = the step's last hidden conv's rows in
// precision `prec` (f16 rows for EHM_PREC_F16, float32 rows otherwise); the next step's rows are written in `next_prec`'s format.
int ehm_step_fused_impl(const void* out_dev, const float* X, int prec, const uint8_t* vis, const float* x, const float* noise, const float* grad,
                        float* x_next, float* x0, const ehm_step_coefs* c, int ddim, int passes, const int32_t* mask_slot, int B,
                        const GcnInputArgs* next_input, int next_prec, hipStream_t st) {
  StepFusedArgs a{};
  a.X = X;
  a.sb.hs = nullptr; a.sb.O = *(const OutDev*)out_dev; a.sb.vis = vis; a.sb.x = x; a.sb.noise = noise; a.sb.grad = grad; a.sb.x_next = x_next;
  a.sb.x0 = x0; a.sb.c = *c; a.sb.ddim = ddim; a.sb.passes = passes; a.sb.B = B; a.sb.do_pose = 0; a.sb.mask_slot = mask_slot;
  a.sb.trace = nullptr; a.sb.pf = nullptr;
  if (a.sb.O.K % 64 != 0) {
    ehm_set_error("ehm_step_fused_impl: hidden width %d is not a multiple of 64", a.sb.O.K);
    return EHM_EINVAL;
  }
  a.do_input = next_input ? 1 : 0;
  if (next_input) a.in = *next_input;
  EhmProfScope ps(EHM_PROF_STEP_FUSED, st);
  if (prec == EHM_PREC_F16) launch_fused<true>(next_prec, B, st, a);
  else launch_fused<false>(next_prec, B, st, a);
  EHM_LAUNCH_CHECK();
  return 0;
}

// Poses of the pending steps (slots 0 .. nsteps-1 of x0_steps) into the skinning launch's per-step transforms / fragments; slot `final_step`
// (or -1) also fills the loop's outputs R, joints (the 24 chain joints), pose6d, x0_final.
int ehm_pose_steps_impl(ehm_smpl* smpl, const float* x0_steps, int nsteps, int final_step, int B, const float* betas, const float* mean,
                        const float* std_, float* A_steps, void* pf_steps, float* R, float* joints, float* pose6d, float* x0_final,
                        float* scratch_joints, hipStream_t st) {
  SmplDev sd;
  ehm_smpl_dev(smpl, &sd);
  PoseStepsArgs p{x0_steps, betas, mean, std_, A_steps, (char*)pf_steps, ehm_skin_pf_bytes_per_step(B), R, joints, pose6d, x0_final, scratch_joints,
                  B, final_step, (kJ + ehm_smpl_num_extra(smpl)) * 3};
  EhmProfScope ps(EHM_PROF_STEP_BODY, st);
  hipLaunchKernelGGL(pose_steps_kernel, dim3(B, nsteps), dim3(64), 0, st, p, sd);
  EHM_LAUNCH_CHECK();
  return 0;
}
