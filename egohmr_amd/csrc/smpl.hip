// SMPL linear blend skinning for gfx950 (MI355X).
//
// Replaces smplx.SMPL.forward / smplx.lbs.lbs (pip smplx==0.1.28; third-party, absent from the
// reference tree - environment.yml:197) as called by the reference at
//   models/egohmr/egohmr.py:276 (every denoising step), :492, :537; test_egohmr.py:291 (final decode)
// and utils/geometry.py:47-66 (rot6d_to_rotmat, fused in front for the hot path, egohmr.py:258-260).
//
// Design: two kernels per call.
//   pose_chain_kernel  one wave per body, lane j = joint j: rot6d -> R, joint regression from betas
//                      (J = J_template + J_shapedirs.beta, the regressor pre-contracted with the shape
//                      basis at create time), then the 24-node kinematic chain resolved level by level
//                      with wavefront shuffles (child lane pulls its parent's 3x4 transform); writes the
//                      skinning transforms A[b,24,3,4] and the 24 posed joints.
//   skin_kernel        thread = vertex, block = 256 vertices x 8 bodies.  Shape blend, pose-corrective
//                      blend (207 basis rows streamed once per block, reused across the 8 bodies held in
//                      registers), 24-joint weighted transform, apply.  Per-body constants (betas,
//                      R - I, A) sit in LDS and are read as broadcasts.  Blocks that share a vertex tile
//                      are placed on the same XCD so the 17 MB pose basis is fetched from HBM/MALL once
//                      per XCD-resident tile and re-used out of that XCD's L2.
// HBM traffic per body-step: 82.7 KB vertices out (+ ~1.5 KB small tensors); shared constants 19.3 MB per launch.
#include <vector>

#include "common.h"
#include "egohmr_hip.h"
#include "internal.h"
#include "gcn_dev.h"
#include "smpl_dev.h"
#include "step_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------ setup
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  // in [rows][cols] -> out [cols][rows]
  __shared__ float tile[32][33];
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(size_t)(r0 + r) * cols + c0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < cols && r0 + tx < rows) out[(size_t)(c0 + r) * rows + r0 + tx] = tile[tx][r];
}

__global__ void joint_basis_kernel(const float* __restrict__ Jr, const float* __restrict__ v_template,
                                   const float* __restrict__ shapedirs, float* __restrict__ Jt, float* __restrict__ Js,
                                   int V) {
  // block (j, m): m in [0,33): m<3 -> J_template[j][m]; else J_shape[j][c][l] with m-3 = c*10+l
  const int j = blockIdx.x, m = blockIdx.y;
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float x = (m < 3) ? v_template[v * 3 + m] : shapedirs[(size_t)v * 30 + (m - 3)];
    s = fmaf(Jr[(size_t)j * V + v], x, s);
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (m < 3) Jt[j * 3 + m] = red[0];
    else Js[j * 30 + (m - 3)] = red[0];
  }
}

// ------------------------------------------------------------------------------------------------ rot6d
__global__ void rot6d_kernel(const float* __restrict__ x, float* __restrict__ Rout, int64_t n, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = x + i * 6;
  float R[9];
  if (mode == 1) rot6d_to_R(p[0], p[2], p[4], p[1], p[3], p[5], R);
  else rot6d_to_R(p[0], p[1], p[2], p[3], p[4], p[5], R);
#pragma unroll
  for (int k = 0; k < 9; ++k) Rout[i * 9 + k] = R[k];
}

template <bool FROM_ROT6D>
__global__ __launch_bounds__(64) void pose_chain_kernel(const float* __restrict__ betas, const float* __restrict__ rot_or_x,
                                                        const float* __restrict__ mean, const float* __restrict__ std_,
                                                        SmplDev S, float* __restrict__ Rws, float* __restrict__ Aout,
                                                        float* __restrict__ joints, float* __restrict__ pose6d_out,
                                                        int joints_stride) {
  pose_chain_body<FROM_ROT6D>(blockIdx.x, threadIdx.x, betas, rot_or_x + (size_t)blockIdx.x * (FROM_ROT6D ? kPoseDim : kJ * 9), mean, std_, S, Rws, Aout,
                              joints, pose6d_out, joints_stride, nullptr);
}

// ------------------------------------------------------------------------------------------------ skinning
__global__ __launch_bounds__(kVT, 2) void skin_kernel(const float* __restrict__ betas, const float* __restrict__ Rws,
                                                   const float* __restrict__ A, SmplDev S, float* __restrict__ verts,
                                                   int B, int v_tiles, int b_groups) {
  __shared__ __attribute__((aligned(16))) float sA[kBGF][kJ][12];
  __shared__ float sPF[kBGF][kPoseBasis + 1];
  __shared__ float sBeta[kBGF][10];

  // XCD-aware order: blocks of one XCD (bid % 8) walk body groups of the same vertex tile back to back
  const int bid = blockIdx.x;
  const int xcd = bid & 7, k = bid >> 3;
  const int vt = (k / b_groups) * 8 + xcd;
  const int bg = k % b_groups;
  if (vt >= v_tiles) return;
  const int b0 = bg * kBGF;
  const int nb = min(kBGF, B - b0);
  const int tid = threadIdx.x;

  for (int i = tid; i < kBGF * kJ * 12; i += kVT) {
    const int bb = i / (kJ * 12);
    (&sA[0][0][0])[i] = bb < nb ? A[(size_t)b0 * kJ * 12 + i] : 0.f;
  }
  for (int i = tid; i < kBGF * kPoseBasis; i += kVT) {
    const int bb = i / kPoseBasis, p = i % kPoseBasis;
    const int e = p % 9;
    float v = 0.f;
    if (bb < nb) v = Rws[((size_t)(b0 + bb) * kJ + 1) * 9 + p] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);  // R[1:] - I
    sPF[bb][p] = v;
  }
  if (tid < kBGF * 10) sBeta[tid / 10][tid % 10] = (tid / 10) < nb ? betas[(size_t)b0 * 10 + tid] : 0.f;
  __syncthreads();

  const int v = vt * kVT + tid;
  if (v >= S.V) return;
  const int V3 = S.V * 3;

  // shape blend: v_shaped = v_template + shapedirs . beta   (accumulated straight into the posed position)
  float po[kBGF][3];
  {
    const float t0 = S.v_template[v * 3 + 0], t1 = S.v_template[v * 3 + 1], t2 = S.v_template[v * 3 + 2];
#pragma unroll
    for (int bb = 0; bb < kBGF; ++bb) { po[bb][0] = 0.f; po[bb][1] = 0.f; po[bb][2] = 0.f; }
#pragma unroll 2
    for (int l = 0; l < 10; ++l) {
      const float s0 = S.shape_t[(size_t)l * V3 + v * 3 + 0], s1 = S.shape_t[(size_t)l * V3 + v * 3 + 1],
                  s2 = S.shape_t[(size_t)l * V3 + v * 3 + 2];
#pragma unroll
      for (int bb = 0; bb < kBGF; ++bb) {
        const float be = sBeta[bb][l];
        po[bb][0] = fmaf(be, s0, po[bb][0]); po[bb][1] = fmaf(be, s1, po[bb][1]); po[bb][2] = fmaf(be, s2, po[bb][2]);
      }
    }
#pragma unroll
    for (int bb = 0; bb < kBGF; ++bb) { po[bb][0] += t0; po[bb][1] += t1; po[bb][2] += t2; }
  }
  // pose-corrective blend: + pose_feature . posedirs (207 basis rows, each reused by the 8 bodies in registers)
  const float* pd = S.posedirs + (size_t)v * 3;
#pragma unroll 2
  for (int p = 0; p < kPoseBasis; ++p) {
    const float d0 = pd[(size_t)p * V3 + 0], d1 = pd[(size_t)p * V3 + 1], d2 = pd[(size_t)p * V3 + 2];
#pragma unroll
    for (int bb = 0; bb < kBGF; ++bb) {
      const float f = sPF[bb][p];
      po[bb][0] = fmaf(f, d0, po[bb][0]); po[bb][1] = fmaf(f, d1, po[bb][1]); po[bb][2] = fmaf(f, d2, po[bb][2]);
    }
  }
  const float* wv = S.w_t + v;   // [24][V]: re-read per body, L1/L2 resident (keeps 24 registers free)
  float wsp[4] = {0.f, 0.f, 0.f, 0.f};
  int jsp[4] = {0, 0, 0, 0};
  if (S.sparse4) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { wsp[s4] = S.w_val[(size_t)s4 * S.V + v]; jsp[s4] = S.w_idx[(size_t)s4 * S.V + v]; }
  }
#pragma unroll
  for (int bb = 0; bb < kBGF; ++bb) {      // fully unrolled (po[] must stay in registers): no break, predicate the store instead
    __builtin_amdgcn_sched_barrier(0);   // one body at a time: keeps the FMAs of different bodies from being interleaved (register pressure)
    const float px = po[bb][0], py = po[bb][1], pz = po[bb][2];
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    if (S.sparse4) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float wj = wsp[s4];
        const int j = jsp[s4];
        const f32x4 r0 = *(const f32x4*)&sA[bb][j][0], r1 = *(const f32x4*)&sA[bb][j][4], r2 = *(const f32x4*)&sA[bb][j][8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          T[e] = fmaf(wj, r0[e], T[e]); T[4 + e] = fmaf(wj, r1[e], T[4 + e]); T[8 + e] = fmaf(wj, r2[e], T[8 + e]);
        }
      }
    } else {
#pragma unroll 4
      for (int j = 0; j < kJ; ++j) {
        const float wj = wv[(size_t)j * S.V];
        const f32x4 r0 = *(const f32x4*)&sA[bb][j][0], r1 = *(const f32x4*)&sA[bb][j][4], r2 = *(const f32x4*)&sA[bb][j][8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          T[e] = fmaf(wj, r0[e], T[e]); T[4 + e] = fmaf(wj, r1[e], T[4 + e]); T[8 + e] = fmaf(wj, r2[e], T[8 + e]);
        }
      }
    }
    if (bb < nb) {
      float* o = verts + ((size_t)(b0 + bb) * S.V + v) * 3;
      const float ox = T[0] * px + T[1] * py + T[2] * pz + T[3], oy = T[4] * px + T[5] * py + T[6] * pz + T[7],
                  oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
      o[0] = ox; o[1] = oy; o[2] = oz;
    }
  }
}

// ------------------------------------------------------------------------------------------------ skinning on the matrix cores
// The shape + pose-corrective blend is a GEMM  [bodies, 224] x [224, V*3]  (coefficients = [R[1:] - I | betas]).  On the VALU it
// was 70 % of skin_kernel (101 us per B=256 launch, 35 TFLOP/s); here it runs as split-f16 MFMA (hi/lo operands, 3 MFMA per
// product, f32 accumulate - the same f32-grade scheme as the GCN convs): one wave = 32 bodies x 32 vertices, three accumulators
// (x, y, z) so that afterwards lane (vertex, half) owns the blended position of its vertex for 16 bodies and finishes the
// skinning (sparse-4 weights, transforms in LDS) without any transposition.  Both operands are stored in fragment order - every
// load instruction reads 1 KiB contiguous - and go straight from L2 to registers.

__global__ void pd_pack_kernel(const float* __restrict__ posedirs, const float* __restrict__ shapedirs, sk_half8* __restrict__ out,
                               int V, int v_tiles, float scale) {
  // one thread per (vertex tile, k-step, coord, lane): writes the hi and the lo fragment
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)v_tiles * kBlendSteps * 3 * 64) return;
  const int lane = (int)(i & 63), c = (int)((i >> 6) % 3), s = (int)((i / 192) % kBlendSteps), vt = (int)(i / (192 * kBlendSteps));
  const int v = 32 * vt + (lane & 31);
  sk_half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * s + 8 * (lane >> 5) + e;
    float x = 0.f;
    if (v < V) {
      if (k < kPoseBasis) x = posedirs[(size_t)k * V * 3 + (size_t)v * 3 + c];
      else if (k < kPoseBasis + 10) x = shapedirs[((size_t)v * 3 + c) * 10 + (k - kPoseBasis)];
    }
    x *= scale;
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  const size_t base = (((size_t)vt * kBlendSteps + s) * 3 + c) * 2;
  out[(base + 0) * 64 + lane] = hi;
  out[(base + 1) * 64 + lane] = lo;
}

__global__ void pf_pack_kernel(const float* __restrict__ Rws, const float* __restrict__ betas, sk_half8* __restrict__ out, int B, int b_tiles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b_tiles * kBlendSteps * 64) return;
  const int lane = i & 63, s = (i >> 6) % kBlendSteps, bt = i / (64 * kBlendSteps);
  const int b = 32 * bt + (lane & 31);
  sk_half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * s + 8 * (lane >> 5) + e;
    float x = 0.f;
    if (b < B) {
      if (k < kPoseBasis) x = Rws[((size_t)b * kJ + 1) * 9 + k] - ((k % 9 == 0 || k % 9 == 4 || k % 9 == 8) ? 1.f : 0.f);   // R[1:] - I
      else if (k < kPoseBasis + 10) x = betas[(size_t)b * 10 + (k - kPoseBasis)];
    }
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  const size_t base = ((size_t)bt * kBlendSteps + s) * 2;
  out[(base + 0) * 64 + lane] = hi;
  out[(base + 1) * 64 + lane] = lo;
}

// `joints` (may be nullptr): rows [24 + n_extra][3] per body; the VertexJointSelector's extra joints (smplx vertex_joint_selector.py: 21
// picked vertices) are written by the lane that owns the vertex - no separate gather launch.
#ifdef EHM_STAMPS
__device__ unsigned long long* g_sdbg = nullptr;
#define SSTAMP(i) do { if (g_sdbg && (threadIdx.x & 63) == 0) g_sdbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

constexpr int kSkinMfmaMinBodies = 24;   // below: the VALU skinning kernel (a 32-body MFMA tile would be mostly padding)

struct SkinArgs {
  const sk_half8* PF; const float* A; SmplDev S;
  float* verts; float* joints;
  int B, v_tiles, vt_groups;
  // several steps of a sampling loop in ONE launch (ehm_skin_steps_impl): PF / A hold `nsteps` consecutive [B]-body sets, the set `final_step`
  // writes verts / joints, the others scratch_verts / scratch_joints (same arithmetic, results of the intermediate steps are not consumed)
  int nsteps = 1, final_step = 0;
  float* scratch_verts = nullptr; float* scratch_joints = nullptr;
  float* vposed = nullptr;     // [B, V, 3] or nullptr: the blended REST vertex (shape + pose-corrective blend, before skinning) - what the skinning VJP
                               // of the collision guidance needs per vertex and body (guidance.hip: skin_bwd_kernel), which otherwise recomputes it
};

// One block (256 threads) of the matrix-core skinning; sA = 32 * 24 * 12 floats (36 KiB) of LDS: the skinning transforms of the block's
// 32 bodies.  `bid` = block index within the skinning grid (also launched as part of skin_input_kernel below).
__device__ __forceinline__ void skin_mfma_body(float (*sA)[kJ][12], int bid, const SkinArgs& a) {
  const sk_half8* __restrict__ PF = a.PF; const float* __restrict__ A = a.A; const SmplDev& S = a.S;
  float* __restrict__ verts = a.verts; float* __restrict__ joints = a.joints;
  const int B = a.B, v_tiles = a.v_tiles, vt_groups = a.vt_groups;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: the blocks of one XCD walk the body tiles of the same vertex-tile group back to back (basis fragments from L2)
  const int xcd = bid & 7, kk = bid >> 3;
  const int b_tiles_step = (B + 31) / 32, b_tiles = b_tiles_step * a.nsteps;
  const int vg = (kk / b_tiles) * 8 + xcd, bt = kk % b_tiles;
  if (vg >= vt_groups) return;
  SSTAMP(0);
  const int step = bt / b_tiles_step;
  const int b0 = 32 * (bt - step * b_tiles_step), nb = min(32, B - b0);
  if (a.scratch_verts) {                                       // (legacy single-step launches: the pointers as given)
    A += (size_t)step * B * kJ * 12;
    if (step != a.final_step) { verts = a.scratch_verts; joints = joints ? a.scratch_joints : nullptr; }
  }
  const int vt = min(4 * vg + wave, v_tiles - 1);             // (a surplus wave of the last group recomputes the last tile and stores nothing)
  const bool live = 4 * vg + wave < v_tiles;

  // blend GEMM operands: fragment sets three k-steps deep (a set = 8 x 1 KiB coalesced loads from L2; nine MFMAs = 288 cycles cover
  // a third of that latency), the first two requested before the transforms are staged
  const sk_half8* pa = PF + ((size_t)bt * kBlendSteps * 2) * 64 + lane;
  const sk_half8* pb = (const sk_half8*)S.PDf + ((size_t)vt * kBlendSteps * 6) * 64 + lane;
  struct FragSet { sk_half8 a_hi, a_lo, b_hi[3], b_lo[3]; } F[3];
  auto load_set = [&](FragSet& f, int s) {
    f.a_hi = pa[(2 * s) * 64]; f.a_lo = pa[(2 * s + 1) * 64];
#pragma unroll
    for (int c = 0; c < 3; ++c) { f.b_hi[c] = pb[(6 * s + 2 * c) * 64]; f.b_lo[c] = pb[(6 * s + 2 * c + 1) * 64]; }
  };
  load_set(F[0], 0);
  load_set(F[1], 1);

  // the 32 bodies' skinning transforms: 2304 float4, nine per thread, all requested before the first is stored (the element-wise
  // copy loop this replaces was 36 dependent round trips = 23.6 k of the wave's 57.7 k cycles)
  {
    const f32x4* src = (const f32x4*)(A + (size_t)b0 * kJ * 12);
    f32x4 tmp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = tid + 256 * k;
      tmp[k] = (i / (kJ * 3)) < nb ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) ((f32x4*)&sA[0][0][0])[tid + 256 * k] = tmp[k];
  }
  __syncthreads();
  if (!live) return;
  SSTAMP(1);

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
  for (int s = 0; s < kBlendSteps; ++s) {
    if (s + 2 < kBlendSteps) load_set(F[(s + 2) % 3], s + 2);
    const FragSet& f = F[s % 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // small cross terms first, leading term last
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_lo, f.b_hi[c], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi, f.b_lo[c], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a_hi, f.b_hi[c], acc[c], 0, 0, 0);
    }
  }

  SSTAMP(2);
  // ---- skinning: lane (vertex = lane & 31, half = lane >> 5) holds bodies (r&3) + 8*(r>>2) + 4*half, r = 0..15
  const int v = 32 * vt + (lane & 31), half = lane >> 5;
  if (v >= S.V) return;
  const float inv = 1.f / S.pd_scale;
  const float t0 = S.v_template[v * 3 + 0], t1 = S.v_template[v * 3 + 1], t2 = S.v_template[v * 3 + 2];
  float wsp[4];
  int jsp[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { wsp[s4] = S.w_val[(size_t)s4 * S.V + v]; jsp[s4] = S.w_idx[(size_t)s4 * S.V + v]; }
  unsigned long long slots = 0;                               // extra joints that are copies of my vertex (none for almost every lane)
  if (joints)
    for (int e = 0; e < S.n_extra; ++e) slots |= (unsigned long long)(S.extra_idx[e] == v) << e;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bb = (r & 3) + 8 * (r >> 2) + 4 * half;
    const float px = fmaf(acc[0][r], inv, t0), py = fmaf(acc[1][r], inv, t1), pz = fmaf(acc[2][r], inv, t2);
    typedef float sk_f32x2 __attribute__((ext_vector_type(2)));
    sk_f32x2 T2[6];                                              // the blended 3 x 4 transform, two entries per register pair (v_pk_fma_f32)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const sk_f32x2 wj = {wsp[s4], wsp[s4]};
      const int j = jsp[s4];
      const f32x4 r0 = *(const f32x4*)&sA[bb][j][0], r1 = *(const f32x4*)&sA[bb][j][4], r2 = *(const f32x4*)&sA[bb][j][8];
      const sk_f32x2 q[6] = {{r0[0], r0[1]}, {r0[2], r0[3]}, {r1[0], r1[1]}, {r1[2], r1[3]}, {r2[0], r2[1]}, {r2[2], r2[3]}};
#pragma unroll
      for (int e = 0; e < 6; ++e) T2[e] = s4 == 0 ? wj * q[e] : __builtin_elementwise_fma(wj, q[e], T2[e]);
    }
    const float T[12] = {T2[0][0], T2[0][1], T2[1][0], T2[1][1], T2[2][0], T2[2][1], T2[3][0], T2[3][1], T2[4][0], T2[4][1], T2[5][0], T2[5][1]};
    if (bb < nb) {
      float* o = verts + ((size_t)(b0 + bb) * S.V + v) * 3;
      const float ox = T[0] * px + T[1] * py + T[2] * pz + T[3], oy = T[4] * px + T[5] * py + T[6] * pz + T[7],
                  oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
      typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
      *(f32x3u*)o = f32x3u{ox, oy, oz};                          // one 12-byte store per lane: a half-wave covers 384 contiguous bytes
      if (a.vposed) *(f32x3u*)(a.vposed + ((size_t)(b0 + bb) * S.V + v) * 3) = f32x3u{px, py, pz};
      for (unsigned long long m = slots; m; m &= m - 1) {
        float* q = joints + ((size_t)(b0 + bb) * (kJ + S.n_extra) + kJ + __builtin_ctzll(m)) * 3;
        q[0] = ox; q[1] = oy; q[2] = oz;
      }
    }
  }
  SSTAMP(3);
}

// (four blocks per CU - 128 registers, 36 KiB of LDS each: the kernel is a chain of L2 round trips per block, and with hipcc's own choice of 168 VGPRs + 48 AGPRs
//  only two were resident.  Same box: 1.40 -> 1.29 ms per 100 steps at 256 bodies, 8.55 -> 7.81 ms at 1280; three blocks measured like two)
__global__ __launch_bounds__(256, 4) void skin_mfma_kernel(SkinArgs a) {
  __shared__ __attribute__((aligned(16))) float sA[32][kJ][12];
  skin_mfma_body(sA, blockIdx.x, a);
}

// The skinning of step t and the input conv of step t + 1 in ONE launch: both depend only on step_body_kernel(t), neither on the other,
// and both are latency-shaped (25 us and 20 us at B = 256 for 21 MB / 25 MB of output).  Blocks [0, skin_blocks) skin, the rest
// run the input conv (gcn_dev.h: gcn_input_body); one 36 KiB LDS buffer serves either.  Separate streams for the two cost more in
// cross-stream events than they overlapped (measured: -7 %), a second kernel boundary costs 2.6 us.
template <int OUT>
__global__ __launch_bounds__(256, 4) void skin_input_kernel(SkinArgs a, GcnInputArgs g, int skin_blocks) {
  __shared__ __attribute__((aligned(16))) float sbuf[32 * kJ * 12];
  if ((int)blockIdx.x < skin_blocks) {
    skin_mfma_body((float (*)[kJ][12])sbuf, blockIdx.x, a);
  } else {
    const int i = blockIdx.x - skin_blocks;
    gcn_input_body<OUT>(sbuf, i / g.ny, i % g.ny, g);
  }
}
#ifdef EHM_STAMPS
extern "C" int ehm_dbg_set_skin(void* p) { unsigned long long* q = (unsigned long long*)p; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sdbg), &q, sizeof(q)); }
#endif

__global__ void extra_joints_kernel(const float* __restrict__ verts, const int32_t* __restrict__ idx, float* __restrict__ joints,
                                    int B, int V, int n_extra) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_extra * 3) return;
  const int c = i % 3, e = (i / 3) % n_extra, b = i / (3 * n_extra);
  joints[((size_t)b * (kJ + n_extra) + kJ + e) * 3 + c] = verts[((size_t)b * V + idx[e]) * 3 + c];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
static int smpl_scratch(ehm_smpl* h, int B) {
  if (B <= h->ws_cap) return 0;
  if (h->ws) (void)hipFree(h->ws);
  h->ws = nullptr;
  h->ws_cap = 0;
  const int cap = (int)round_up(B, 64);
  EHM_HIP(hipMalloc(&h->ws, (size_t)cap * kJ * 21 * sizeof(float)));
  h->ws_cap = cap;
  return 0;
}

extern "C" int ehm_smpl_create(ehm_smpl** out, const float* v_template, const float* shapedirs, const float* posedirs,
                               const float* J_regressor, const float* lbs_weights, const int32_t* parents,
                               const int32_t* extra_joint_vertex_ids, int num_verts, int n_extra, void* stream) {
  EHM_CHECK_ARG(out && v_template && shapedirs && posedirs && J_regressor && lbs_weights && parents);
  EHM_CHECK_ARG(num_verts > 0 && n_extra >= 0 && n_extra <= 64 && (n_extra == 0 || extra_joint_vertex_ids));
  EHM_CHECK_ARG(parents[0] < 0);
  hipStream_t st = (hipStream_t)stream;
  auto* h = new ehm_smpl();
  SmplDev& d = h->d;
  d.V = num_verts;
  d.n_extra = n_extra;
  d.tree.max_depth = 0;
  for (int j = 0; j < kJ; ++j) {
    if (j > 0 && !(parents[j] >= 0 && parents[j] < j)) {
      delete h;
      ehm_set_error("ehm_smpl_create: parents[%d]=%d must satisfy 0 <= parent < child", j, parents[j]);
      return EHM_EINVAL;
    }
    d.tree.parent[j] = (int8_t)parents[j];
    d.tree.depth[j] = j == 0 ? 0 : (int8_t)(d.tree.depth[parents[j]] + 1);
    if (d.tree.depth[j] > d.tree.max_depth) d.tree.max_depth = d.tree.depth[j];
  }
  const size_t V = num_verts;
  const size_t floats = V * 3 + 10 * V * 3 + kJ * V + kJ * 3 + kJ * 30 + 64 + 64 + 8 * V;
  if (hipMalloc(&h->arena, floats * sizeof(float)) != hipSuccess) {
    delete h;
    ehm_set_error("ehm_smpl_create: hipMalloc failed");
    return EHM_ENOMEM;
  }
  float* cur = h->arena;
  d.v_template = cur;  cur += V * 3;
  d.shape_t = cur;     cur += 10 * V * 3;
  d.w_t = cur;         cur += kJ * V;
  d.J_template = cur;  cur += kJ * 3;
  d.J_shape = cur;     cur += kJ * 30;
  d.extra_idx = (int32_t*)cur;  cur += 64;
  d.w_idx = (int32_t*)cur;      cur += 4 * V;
  d.w_val = cur;                cur += 4 * V;
  d.posedirs = posedirs;
  {  // SMPL skinning weights have <= 4 non-zeros per vertex: detect it once and keep a compressed copy
    std::vector<float> hw(V * kJ);
    std::vector<int32_t> hi(4 * V, 0);
    std::vector<float> hv(4 * V, 0.f);
    d.sparse4 = 0;
    if (hipMemcpy(hw.data(), lbs_weights, hw.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) {
      d.sparse4 = 1;
      for (size_t v = 0; v < V && d.sparse4; ++v) {
        int n = 0;
        for (int j = 0; j < kJ; ++j) {
          const float w = hw[v * kJ + j];
          if (w != 0.f) {
            if (n == 4) { d.sparse4 = 0; break; }
            hi[(size_t)n * V + v] = j;
            hv[(size_t)n * V + v] = w;
            ++n;
          }
        }
      }
      if (d.sparse4) {
        (void)hipMemcpy(d.w_idx, hi.data(), hi.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        (void)hipMemcpy(d.w_val, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice);
      }
    }
  }
  int rc = 0;
  if (hipMemcpyAsync(d.v_template, v_template, V * 3 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = EHM_EIO;
  if (n_extra && hipMemcpyAsync(d.extra_idx, extra_joint_vertex_ids, n_extra * sizeof(int32_t), hipMemcpyHostToDevice, st) != hipSuccess) rc = EHM_EIO;
  for (int e = 0; e < n_extra; ++e)
    if (extra_joint_vertex_ids[e] < 0 || extra_joint_vertex_ids[e] >= num_verts) rc = EHM_EINVAL;
  if (rc == 0) {
    // shapedirs [V*3][10] -> [10][V*3];  lbs_weights [V][24] -> [24][V]
    hipLaunchKernelGGL(transpose_kernel, dim3(1, (unsigned)ceil_div(V * 3, 32)), dim3(256), 0, st, shapedirs, d.shape_t, (int)(V * 3), 10);
    hipLaunchKernelGGL(transpose_kernel, dim3(1, (unsigned)ceil_div(V, 32)), dim3(256), 0, st, lbs_weights, d.w_t, (int)V, kJ);
    hipLaunchKernelGGL(joint_basis_kernel, dim3(kJ, 33), dim3(256), 0, st, J_regressor, v_template, shapedirs, d.J_template,
                       d.J_shape, (int)V);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = EHM_EIO;
  }
  if (rc == 0 && d.sparse4) {   // blend basis as split-f16 MFMA fragments (skin_mfma_kernel); scale = power of two that keeps the lo halves normal
    std::vector<float> hb((size_t)kPoseBasis * V * 3 + V * 30);
    float amax = 0.f;
    if (hipMemcpy(hb.data(), posedirs, (size_t)kPoseBasis * V * 3 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(hb.data() + (size_t)kPoseBasis * V * 3, shapedirs, V * 30 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) {
      for (float x : hb) amax = fmaxf(amax, fabsf(x));
      float scale = 1.f;
      if (amax > 0.f && amax < 1e30f) {
        int e;
        frexpf(2048.f / amax, &e);          // 2048/amax = m * 2^e, m in [0.5, 1)  ->  2^(e-1) <= 2048/amax
        scale = ldexpf(1.f, e - 1);
      }
      const int v_tiles = (int)ceil_div(V, 32);
      const size_t bytes = (size_t)v_tiles * kBlendSteps * 6 * 64 * 16;
      if (hipMalloc(&h->pdf, bytes) == hipSuccess) {
        const int64_t n = (int64_t)v_tiles * kBlendSteps * 3 * 64;
        hipLaunchKernelGGL(pd_pack_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, posedirs, shapedirs, (sk_half8*)h->pdf,
                           (int)V, v_tiles, scale);
        if (hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
          d.PDf = h->pdf;
          d.pd_scale = scale;
        } else {
          (void)hipFree(h->pdf);
          h->pdf = nullptr;
        }
      }
    }
  }
  if (rc != 0) {
    (void)hipFree(h->arena);
    delete h;
    ehm_set_error("ehm_smpl_create: setup failed (rc=%d)", rc);
    return rc;
  }
  *out = h;
  return 0;
}

extern "C" void ehm_smpl_destroy(ehm_smpl* h) {
  if (!h) return;
  (void)hipFree(h->arena);
  if (h->ws) (void)hipFree(h->ws);
  if (h->pdf) (void)hipFree(h->pdf);
  if (h->pf) (void)hipFree(h->pf);
  delete h;
}

// internal entry shared with sampler.hip: caller provides R/A scratch (no allocation -> graph safe)
int ehm_smpl_forward_impl(ehm_smpl* h, const float* betas, const float* rot_or_x, bool from_rot6d, const float* mean,
                          const float* std_, float* verts, float* joints, float* Rws, float* Aws, float* pose6d_out, int B,
                          hipStream_t st, float* vposed) {
  const SmplDev& d = h->d;
  const int jstride = (kJ + d.n_extra) * 3;
  if (from_rot6d)
    hipLaunchKernelGGL(pose_chain_kernel<true>, dim3(B), dim3(64), 0, st, betas, rot_or_x, mean, std_, d, Rws, Aws, joints,
                       pose6d_out, jstride);
  else
    hipLaunchKernelGGL(pose_chain_kernel<false>, dim3(B), dim3(64), 0, st, betas, rot_or_x, (const float*)nullptr,
                       (const float*)nullptr, d, Rws, Aws, joints, (float*)nullptr, jstride);
  constexpr int mfma_min = kSkinMfmaMinBodies;
  if (d.PDf && B >= mfma_min) {
    const int b_tiles = (int)ceil_div(B, 32);
    if (32 * b_tiles > h->pf_cap) {                      // grows on the first call with a larger batch only
      if (h->pf) EHM_HIP(hipFree(h->pf));
      h->pf = nullptr;
      h->pf_cap = 0;
      EHM_HIP(hipMalloc(&h->pf, (size_t)b_tiles * kBlendSteps * 2 * 64 * 16));
      h->pf_cap = 32 * b_tiles;
    }
    hipLaunchKernelGGL(pf_pack_kernel, dim3((unsigned)ceil_div(b_tiles * kBlendSteps * 64, 256)), dim3(256), 0, st, Rws, betas,
                       (sk_half8*)h->pf, B, b_tiles);
    const int v_tiles = (int)ceil_div(d.V, 32), vt_groups = (int)ceil_div(v_tiles, 4);
    const int blocks = (int)round_up(vt_groups, 8) * b_tiles;
    SkinArgs sa{(const sk_half8*)h->pf, Aws, d, verts, d.n_extra ? joints : nullptr, B, v_tiles, vt_groups};
    sa.vposed = vposed;               // (only this path writes it: ehm_smpl_writes_vposed)
    hipLaunchKernelGGL(skin_mfma_kernel, dim3(blocks), dim3(256), 0, st, sa);
  } else {
    const int v_tiles = (int)ceil_div(d.V, kVT), b_groups = (int)ceil_div(B, kBGF);
    const int blocks = (int)round_up(v_tiles, 8) * b_groups;
    hipLaunchKernelGGL(skin_kernel, dim3(blocks), dim3(kVT), 0, st, betas, Rws, Aws, d, verts, B, v_tiles, b_groups);
    if (d.n_extra)
      hipLaunchKernelGGL(extra_joints_kernel, dim3((unsigned)ceil_div((int64_t)B * d.n_extra * 3, 256)), dim3(256), 0, st, verts,
                         d.extra_idx, joints, B, d.V, d.n_extra);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ fused per-body step (sampling loop)
// One launch per denoising step for everything that is per body and tiny - in the round-1 loop four launches (gcn_out_mix, ddpm / ddim
// step, pose_chain, pf_pack: ~5 us each, all latency):
//   (1) output conv: modulated adjacency mix of the [24 x 12] responses + bias, visibility fuse of the two passes -> x0 [144]
//       (modulated_gcn_conv.py:47, egohmr.py:247-256; same operation order as gcn_out_mix_kernel)
//   (2) sampler update x_t -> x_{t-1} (gaussian_diffusion.py:217-220,:333-336,:378-385 / :286-290,:539-555; roundings ordered like the
//       eager torch ops, as ddpm_step_kernel / ddim_step_kernel)
//   (3) de-normalise, rot6d -> R, joint regression, 24-joint kinematic chain by wave shuffles -> R, skinning transforms A, joints
//   (4) the body's blend coefficients [R[1:] - I | betas] as split-f16 MFMA fragments for skin_mfma_kernel.
// One wave per body.
__global__ __launch_bounds__(64) void step_body_kernel(StepBodyArgs a, SmplDev S) {
  __shared__ StepBodyLds L;
  step_body_one(blockIdx.x, threadIdx.x, a, S, L, [] { __syncthreads(); });
}

// sampler.hip's per-step call: output-conv mix + sampler update + pose chain + fragment pack in one launch, then the skinning launch.
int ehm_step_body_impl(ehm_smpl* h, const float* hs, const void* out_dev, const uint8_t* vis, const float* x, const float* noise,
                       const float* grad, float* x_next, float* x0, const ehm_step_coefs* c, int ddim, int passes, const int32_t* mask_slot, int do_pose,
                       const float* betas, const float* mean, const float* std_, float* verts, float* joints, float* Rws, float* Aws,
                       float* pose6d, int B, hipStream_t st, const GcnInputArgs* next_input, int next_prec, int* fused, void* defer_pf) {
  const SmplDev& d = h->d;
  if (fused) *fused = 0;
  constexpr int mfma_min = kSkinMfmaMinBodies;
  const bool mfma = d.PDf && B >= mfma_min;
  const int b_tiles = (int)ceil_div(B, 32);
  if (defer_pf && !mfma) {
    ehm_set_error("ehm_step_body_impl: deferred skinning needs the matrix-core skinning path");
    return EHM_EINVAL;
  }
  if (mfma && !defer_pf && 32 * b_tiles > h->pf_cap) {                    // grows on the first call with a larger batch only
    if (h->pf) EHM_HIP(hipFree(h->pf));
    h->pf = nullptr;
    h->pf_cap = 0;
    EHM_HIP(hipMalloc(&h->pf, (size_t)b_tiles * kBlendSteps * 2 * 64 * 16));
    EHM_HIP(hipMemsetAsync(h->pf, 0, (size_t)b_tiles * kBlendSteps * 2 * 64 * 16, st));   // padding bodies of the last tile
    h->pf_cap = 32 * b_tiles;
  }
  StepBodyArgs a;
  a.hs = hs; a.O = *(const OutDev*)out_dev; a.vis = vis; a.x = x; a.noise = noise; a.grad = grad; a.x_next = x_next; a.x0 = x0;
  a.c = *c; a.ddim = ddim; a.passes = passes; a.B = B; a.do_pose = do_pose; a.mask_slot = mask_slot;
  a.betas = betas; a.mean = mean; a.std_ = std_; a.Rws = Rws; a.Aws = Aws; a.joints = joints; a.pose6d = pose6d;
  a.jstride = (kJ + d.n_extra) * 3;
  a.pf = defer_pf ? (sk_half8*)defer_pf : (mfma ? (sk_half8*)h->pf : nullptr);
  {
    EhmProfScope ps(EHM_PROF_STEP_BODY, st);
    a.trace = nullptr;
    hipLaunchKernelGGL(step_body_kernel, dim3(B), dim3(64), 0, st, a, d);
  }
  if (do_pose && !defer_pf) {
    EhmProfScope ps(EHM_PROF_SKIN_INPUT, st);
    if (mfma) {
      const int v_tiles = (int)ceil_div(d.V, 32), vt_groups = (int)ceil_div(v_tiles, 4);
      const int blocks = (int)round_up(vt_groups, 8) * b_tiles;
      SkinArgs sa{(const sk_half8*)h->pf, Aws, d, verts, d.n_extra ? joints : nullptr, B, v_tiles, vt_groups};
      if (next_input) {                                       // + the next step's input conv in the same launch
        const int in_blocks = next_input->total_vb * next_input->ny;
        if (next_prec == EHM_PREC_F32) hipLaunchKernelGGL(skin_input_kernel<0>, dim3(blocks + in_blocks), dim3(256), 0, st, sa, *next_input, blocks);
        else if (next_prec == EHM_PREC_F16X3) hipLaunchKernelGGL(skin_input_kernel<1>, dim3(blocks + in_blocks), dim3(256), 0, st, sa, *next_input, blocks);
        else hipLaunchKernelGGL(skin_input_kernel<2>, dim3(blocks + in_blocks), dim3(256), 0, st, sa, *next_input, blocks);
        *fused = 1;
      } else {
        hipLaunchKernelGGL(skin_mfma_kernel, dim3(blocks), dim3(256), 0, st, sa);
      }
    } else {
      const int v_tiles = (int)ceil_div(d.V, kVT), b_groups = (int)ceil_div(B, kBGF);
      hipLaunchKernelGGL(skin_kernel, dim3((int)round_up(v_tiles, 8) * b_groups), dim3(kVT), 0, st, betas, Rws, Aws, d, verts, B, v_tiles, b_groups);
      if (d.n_extra)
        hipLaunchKernelGGL(extra_joints_kernel, dim3((unsigned)ceil_div((int64_t)B * d.n_extra * 3, 256)), dim3(256), 0, st, verts, d.extra_idx,
                           joints, B, d.V, d.n_extra);
    }
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

// The skinning of `nsteps` consecutive steps of a sampling loop in one launch (the one-launch loop of gcn_tile.hip leaves every step's
// transforms A_steps [nsteps,B,24,12] and blend-coefficient fragments pf_steps [nsteps, ceil(B/32), 14, 2, 64] behind): step `final_step`
// writes verts / joints, the others the scratch buffers (any negative final_step: all of them).
int ehm_skin_steps_impl(ehm_smpl* h, const float* A_steps, const void* pf_steps, int nsteps, int final_step, int B, float* verts, float* joints,
                        float* scratch_verts, float* scratch_joints, hipStream_t st) {
  const SmplDev& d = h->d;
  if (!d.PDf || B < kSkinMfmaMinBodies || nsteps < 1) {
    ehm_set_error("ehm_skin_steps_impl: needs the matrix-core skinning path (B >= %d) and nsteps >= 1", kSkinMfmaMinBodies);
    return EHM_EINVAL;
  }
  const int b_tiles = (int)ceil_div(B, 32) * nsteps;
  const int v_tiles = (int)ceil_div(d.V, 32), vt_groups = (int)ceil_div(v_tiles, 4);
  const int64_t blocks = round_up(vt_groups, 8) * (int64_t)b_tiles;
  SkinArgs sa{(const sk_half8*)pf_steps, A_steps, d, verts, d.n_extra ? joints : nullptr, B, v_tiles, vt_groups};
  sa.nsteps = nsteps; sa.final_step = final_step; sa.scratch_verts = scratch_verts; sa.scratch_joints = scratch_joints;
  EhmProfScope ps(EHM_PROF_SKIN_INPUT, st);
  hipLaunchKernelGGL(skin_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sa);
  EHM_LAUNCH_CHECK();
  return 0;
}
int ehm_skin_min_bodies() { return kSkinMfmaMinBodies; }
int ehm_smpl_has_mfma_skin(const ehm_smpl* h) { return h->d.PDf != nullptr ? 1 : 0; }
int ehm_smpl_writes_vposed(const ehm_smpl* h, int B) { return h->d.PDf != nullptr && B >= kSkinMfmaMinBodies ? 1 : 0; }
int64_t ehm_skin_pf_bytes_per_step(int B) { return (int64_t)ceil_div(B, 32) * kBlendSteps * 2 * 64 * 16; }
void ehm_smpl_dev(const ehm_smpl* h, void* out) { memcpy(out, &h->d, sizeof(SmplDev)); }
size_t ehm_smpl_dev_size() { return sizeof(SmplDev); }

int ehm_smpl_pose_impl(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std_, float* Rws, float* Aws,
                       float* jws, int B, hipStream_t st) {
  const SmplDev& d = h->d;
  hipLaunchKernelGGL(pose_chain_kernel<true>, dim3(B), dim3(64), 0, st, betas, x, mean, std_, d, Rws, Aws, jws, (float*)nullptr,
                     (kJ + d.n_extra) * 3);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_smpl_forward(ehm_smpl* h, const float* betas, const float* rotmats, float* verts, float* joints, float* A_out,
                                int B, void* stream) {
  EHM_CHECK_ARG(h && betas && rotmats && verts && joints && B > 0);
  int rc = smpl_scratch(h, B);
  if (rc) return rc;
  float* Rws = h->ws;
  float* Aws = A_out ? A_out : h->ws + (size_t)h->ws_cap * kJ * 9;
  // the skinning kernel reads R from scratch: copy the caller's matrices through the pose kernel (Rws written there)
  return ehm_smpl_forward_impl(h, betas, rotmats, false, nullptr, nullptr, verts, joints, Rws, Aws, nullptr, B, (hipStream_t)stream);
}

extern "C" int ehm_smpl_forward_rot6d(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std_,
                                      float* verts, float* joints, float* R_out, float* pose6d_out, float* A_out, int B,
                                      void* stream) {
  EHM_CHECK_ARG(h && betas && x && mean && std_ && verts && joints && B > 0);
  int rc = smpl_scratch(h, B);
  if (rc) return rc;
  float* Rws = R_out ? R_out : h->ws;
  float* Aws = A_out ? A_out : h->ws + (size_t)h->ws_cap * kJ * 9;
  return ehm_smpl_forward_impl(h, betas, x, true, mean, std_, verts, joints, Rws, Aws, pose6d_out, B, (hipStream_t)stream);
}

extern "C" int ehm_rot6d_to_rotmat(const float* x6d, float* R, int64_t n, int mode, void* stream) {
  EHM_CHECK_ARG(n >= 0 && (mode == 0 || mode == 1));
  if (n == 0) return 0;   // empty batch: nothing to do, pointers may be null
  EHM_CHECK_ARG(x6d && R);
  hipLaunchKernelGGL(rot6d_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x6d, R, n, mode);
  EHM_LAUNCH_CHECK();
  return 0;
}

int ehm_smpl_num_verts(const ehm_smpl* h) { return h->d.V; }
int ehm_smpl_num_extra(const ehm_smpl* h) { return h->d.n_extra; }
