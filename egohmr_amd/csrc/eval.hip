// Evaluation block of the reference's driver on gfx950 (test_egohmr.py:399-494) and the axis-angle conversion of the
// COAP / VolSMPL call sites (utils/konia_transform.py:316-340, called at models/egohmr/egohmr.py:495, :540).
//
//   rotmat_to_angle_axis (+ VJP)   one thread per 3 x 3 matrix: rotation_matrix_to_quaternion's four branches (:349-443) and quaternion_to_angle_axis
//                                  (:560-630) in registers; the VJP differentiates the branch the forward took (what autograd does through torch.where)
//   point_errors                   one 256-thread block per (item, sample): |pred - gt| per point (optionally origin-aligned), mean over the points,
//                                  sums over the visible / invisible points (G-MPJPE :399-407, MPJPE :409-417, V2V :441-449)
//   procrustes                     one thread per (item, sample), float64 in registers as the reference's numpy loop (utils/pose_utils.py:10-66): centred
//                                  24 x 3 clouds, K = X1^T X2, 3 x 3 SVD by one-sided Jacobi, R = V Z U^T, scale, translation -> per-joint error :419-437
//   diversity                      one wave per item, lane = joint: unbiased std over the samples and the pairwise-distance sum (:453-494), masked means
//
// These are latency-bound reductions over a few MB (the largest, V2V at B x S = 1280 bodies, reads 106 MB once: HBM-bound, ~25 us at 5 TB/s); what the
// reference does with them is dozens of eager ops + a per-sample numpy SVD loop on the CPU with a device -> host copy in front.
#include "common.h"
#include "egohmr_hip.h"

namespace {

constexpr float kEps = 1.0e-6f;   // konia_transform.py: eps of rotation_matrix_to_quaternion / quaternion_to_angle_axis / safe_zero_division

__device__ __forceinline__ float safe_div(float num, float den) {   // konia_transform.py:343-346
  if (fabsf(den) < kEps) den += kEps;
  return num / den;
}

// branch: 0 = trace > 0; 1, 2, 3 = the x / y / z leading cases.  lead = the clamped radicand's argument, n0..n2 = the three numerators in quaternion
// order with the leading component left out.
struct QuatBranch { int br; float lead, n[3]; };

__device__ __forceinline__ QuatBranch quat_branch(const float* m) {
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
  const float trace = m00 + m11 + m22;
  QuatBranch b;
  if (trace > 0.f) { b.br = 0; b.lead = trace + 1.f; b.n[0] = m21 - m12; b.n[1] = m02 - m20; b.n[2] = m10 - m01; }                       // (w) x y z
  else if (m00 > m11 && m00 > m22) { b.br = 1; b.lead = 1.f + m00 - m11 - m22; b.n[0] = m21 - m12; b.n[1] = m01 + m10; b.n[2] = m02 + m20; }   // w (x) y z
  else if (m11 > m22) { b.br = 2; b.lead = 1.f + m11 - m00 - m22; b.n[0] = m02 - m20; b.n[1] = m01 + m10; b.n[2] = m12 + m21; }              // w x (y) z
  else { b.br = 3; b.lead = 1.f + m22 - m00 - m11; b.n[0] = m10 - m01; b.n[1] = m02 + m20; b.n[2] = m12 + m21; }                           // w x y (z)
  return b;
}

// q = (w, x, y, z) from a branch
__device__ __forceinline__ void quat_of(const QuatBranch& b, float& sq, float* q) {
  sq = sqrtf(fmaxf(b.lead, kEps)) * 2.f;
  const float lead = 0.25f * sq;
  const float a = safe_div(b.n[0], sq), c = safe_div(b.n[1], sq), d = safe_div(b.n[2], sq);
  switch (b.br) {
    case 0: q[0] = lead; q[1] = a; q[2] = c; q[3] = d; break;
    case 1: q[0] = a; q[1] = lead; q[2] = c; q[3] = d; break;
    case 2: q[0] = a; q[1] = c; q[2] = lead; q[3] = d; break;
    default: q[0] = a; q[1] = c; q[2] = d; q[3] = lead; break;
  }
}

__device__ __forceinline__ float safe_atan2(float y, float x) {   // konia_transform.py:44-47
  if (fabsf(y) < kEps && fabsf(x) < kEps) y += kEps;
  return atan2f(y, x);
}

__global__ void rotmat_to_aa_kernel(const float* __restrict__ R, float* __restrict__ aa, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = R[i * 9 + k];
  float sq, q[4];
  quat_of(quat_branch(m), sq, q);
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float sn = sqrtf(fmaxf(s2, kEps));
  const float two_theta = 2.f * (q[0] < 0.f ? safe_atan2(-sn, -q[0]) : safe_atan2(sn, q[0]));
  const float k = s2 > 0.f ? safe_div(two_theta, sn) : 2.f;
  aa[i * 3 + 0] = q[1] * k; aa[i * 3 + 1] = q[2] * k; aa[i * 3 + 2] = q[3] * k;
}

__global__ void rotmat_to_aa_bwd_kernel(const float* __restrict__ R, const float* __restrict__ gaa, float* __restrict__ gR, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = R[i * 9 + k];
  const QuatBranch b = quat_branch(m);
  float sq, q[4];
  quat_of(b, sq, q);
  const float g0 = gaa[i * 3 + 0], g1 = gaa[i * 3 + 1], g2 = gaa[i * 3 + 2];
  // ---- aa = (x, y, z) k(w, sn(x, y, z))
  const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float sn = sqrtf(fmaxf(s2, kEps));
  float gq[4] = {0.f, 0.f, 0.f, 0.f};
  if (s2 > 0.f) {
    const float w = q[0];
    const float two_theta = 2.f * (w < 0.f ? safe_atan2(-sn, -w) : safe_atan2(sn, w));
    float den = sn;
    if (fabsf(den) < kEps) den += kEps;
    const float k = two_theta / den;
    const float gk = g0 * q[1] + g1 * q[2] + g2 * q[3];
    const float r2 = sn * sn + w * w;
    // atan2(+-sn, +-w): d/dsn = w / r2, d/dw = -sn / r2 in both sign branches
    const float dk_dsn = (2.f * w / r2) / den - two_theta / (den * den);
    const float dk_dw = (-2.f * sn / r2) / den;
    const float gsn = gk * dk_dsn;
    gq[0] = gk * dk_dw;
    const float dsn = s2 > kEps ? 1.f / sn : 0.f;            // sqrt(clamp_min(s2, eps)): the clamp passes no gradient below eps
    gq[1] = g0 * k + gsn * q[1] * dsn;
    gq[2] = g1 * k + gsn * q[2] * dsn;
    gq[3] = g2 * k + gsn * q[3] * dsn;
  } else {
    gq[1] = 2.f * g0; gq[2] = 2.f * g1; gq[3] = 2.f * g2;
  }
  // ---- q from (lead, n0, n1, n2): leading component sq / 4, the others n_j / sq (sq >= 2e-3: safe_div's guard never fires)
  float gn[3], glead_q;
  switch (b.br) {
    case 0: glead_q = gq[0]; gn[0] = gq[1]; gn[1] = gq[2]; gn[2] = gq[3]; break;
    case 1: glead_q = gq[1]; gn[0] = gq[0]; gn[1] = gq[2]; gn[2] = gq[3]; break;
    case 2: glead_q = gq[2]; gn[0] = gq[0]; gn[1] = gq[1]; gn[2] = gq[3]; break;
    default: glead_q = gq[3]; gn[0] = gq[0]; gn[1] = gq[1]; gn[2] = gq[2]; break;
  }
  float gsq = 0.25f * glead_q;
#pragma unroll
  for (int j = 0; j < 3; ++j) { gsq -= gn[j] * b.n[j] / (sq * sq); gn[j] /= sq; }
  const float glead = b.lead >= kEps ? gsq * 2.f / sq : 0.f;   // sq = 2 sqrt(clamp_min(lead, eps))
  float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // index: m00 0, m01 1, m02 2, m10 3, m11 4, m12 5, m20 6, m21 7, m22 8
  switch (b.br) {
    case 0: g[0] += glead; g[4] += glead; g[8] += glead;
            g[7] += gn[0]; g[5] -= gn[0]; g[2] += gn[1]; g[6] -= gn[1]; g[3] += gn[2]; g[1] -= gn[2]; break;
    case 1: g[0] += glead; g[4] -= glead; g[8] -= glead;
            g[7] += gn[0]; g[5] -= gn[0]; g[1] += gn[1]; g[3] += gn[1]; g[2] += gn[2]; g[6] += gn[2]; break;
    case 2: g[4] += glead; g[0] -= glead; g[8] -= glead;
            g[2] += gn[0]; g[6] -= gn[0]; g[1] += gn[1]; g[3] += gn[1]; g[5] += gn[2]; g[7] += gn[2]; break;
    default: g[8] += glead; g[0] -= glead; g[4] -= glead;
            g[3] += gn[0]; g[1] -= gn[0]; g[2] += gn[1]; g[6] += gn[1]; g[5] += gn[2]; g[7] += gn[2]; break;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) gR[i * 9 + k] = g[k];
}

// ------------------------------------------------------------------------------------------------ point errors
__device__ __forceinline__ float block_sum_256(float v, float* red) {   // 4 waves; every thread returns the block's sum
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void point_errors_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ opred,
                                                           const float* __restrict__ ogt, const uint8_t* __restrict__ mask, float* __restrict__ per_point,
                                                           float* __restrict__ mean, float* __restrict__ vis_sum, float* __restrict__ invis_sum, int S, int P,
                                                           int pred_stride, int gt_stride, int origin_point) {
  __shared__ float red[4];
  const int bs = blockIdx.x, b = bs / S, tid = threadIdx.x;
  const float* p = pred + (size_t)bs * pred_stride * 3;
  const float* g = gt + (size_t)b * gt_stride * 3;
  float op[3] = {0.f, 0.f, 0.f}, og[3] = {0.f, 0.f, 0.f};
  if (opred) { op[0] = opred[(size_t)bs * 3]; op[1] = opred[(size_t)bs * 3 + 1]; op[2] = opred[(size_t)bs * 3 + 2]; }
  if (ogt) { og[0] = ogt[(size_t)b * 3]; og[1] = ogt[(size_t)b * 3 + 1]; og[2] = ogt[(size_t)b * 3 + 2]; }
  if (origin_point >= 0) {                                   // each cloud's own point `origin_point` (the pelvis, test_egohmr.py:409: joint 0)
    for (int c = 0; c < 3; ++c) { op[c] = p[3 * origin_point + c]; og[c] = g[3 * origin_point + c]; }
  }
  float s_all = 0.f, s_vis = 0.f;
  for (int i = tid; i < P; i += 256) {
    const float dx = (p[3 * i] - op[0]) - (g[3 * i] - og[0]), dy = (p[3 * i + 1] - op[1]) - (g[3 * i + 1] - og[1]), dz = (p[3 * i + 2] - op[2]) - (g[3 * i + 2] - og[2]);
    const float e = sqrtf(dx * dx + dy * dy + dz * dz);
    if (per_point) per_point[(size_t)bs * P + i] = e;
    s_all += e;
    if (mask && mask[(size_t)b * P + i]) s_vis += e;
  }
  s_all = block_sum_256(s_all, red);
  s_vis = block_sum_256(s_vis, red);
  if (tid == 0) {
    mean[bs] = s_all / (float)P;
    if (vis_sum) vis_sum[bs] = mask ? s_vis : s_all;
    if (invis_sum) invis_sum[bs] = mask ? s_all - s_vis : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ Procrustes
constexpr int PJ_MAX = 32;

// One-sided Jacobi on the columns of a 3 x 3 matrix: K V = B with orthogonal columns (B = U diag(s)).  float64: converges in 4 - 6 sweeps.
__device__ void svd3(const double K[3][3], double U[3][3], double s[3], double V[3][3]) {
  double Bm[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { Bm[r][c] = K[r][c]; V[r][c] = r == c ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int r = 0; r < 3; ++r) { al += Bm[r][p] * Bm[r][p]; be += Bm[r][q] * Bm[r][q]; ga += Bm[r][p] * Bm[r][q]; }
        if (ga == 0.0 || fabs(ga) <= 1e-300) continue;
        off = fmax(off, fabs(ga) / sqrt(fmax(al * be, 1e-300)));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          const double bp = Bm[r][p], bq = Bm[r][q];
          Bm[r][p] = c * bp - sn * bq; Bm[r][q] = sn * bp + c * bq;
          const double vp = V[r][p], vq = V[r][q];
          V[r][p] = c * vp - sn * vq; V[r][q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double smax = 0.0;
  for (int c = 0; c < 3; ++c) {
    s[c] = sqrt(Bm[0][c] * Bm[0][c] + Bm[1][c] * Bm[1][c] + Bm[2][c] * Bm[2][c]);
    smax = fmax(smax, s[c]);
  }
  int good[3], ng = 0;
  for (int c = 0; c < 3; ++c) {
    if (s[c] > 1e-13 * smax && s[c] > 0.0) { for (int r = 0; r < 3; ++r) U[r][c] = Bm[r][c] / s[c]; good[ng++] = c; }
  }
  if (ng == 2) {               // rank 2: the missing left vector completes the frame (its sign is absorbed by Z = diag(1, 1, det))
    const int m = 3 - good[0] - good[1], a = good[0], bq = good[1];
    U[0][m] = U[1][a] * U[2][bq] - U[2][a] * U[1][bq];
    U[1][m] = U[2][a] * U[0][bq] - U[0][a] * U[2][bq];
    U[2][m] = U[0][a] * U[1][bq] - U[1][a] * U[0][bq];
  } else if (ng < 2) {         // rank <= 1: the rotation is not determined by the data; any orthonormal completion (the reference's is LAPACK's)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) U[r][c] = V[r][c];
  }
}

__device__ __forceinline__ double det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}

__global__ __launch_bounds__(64) void procrustes_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                                                        float* __restrict__ aligned, float* __restrict__ per_joint, float* __restrict__ mean,
                                                        float* __restrict__ vis_sum, float* __restrict__ invis_sum, int n, int S, int J) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const int b = i / S;
  const float* p = pred + (size_t)i * J * 3;
  const float* g = gt + (size_t)b * J * 3;
  double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
  for (int j = 0; j < J; ++j)
    for (int c = 0; c < 3; ++c) { mu1[c] += (double)p[3 * j + c]; mu2[c] += (double)g[3 * j + c]; }
  for (int c = 0; c < 3; ++c) { mu1[c] /= J; mu2[c] /= J; }
  double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
  for (int j = 0; j < J; ++j) {
    double x1[3], x2[3];
    for (int c = 0; c < 3; ++c) { x1[c] = (double)p[3 * j + c] - mu1[c]; x2[c] = (double)g[3 * j + c] - mu2[c]; var1 += x1[c] * x1[c]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K[r][c] += x1[r] * x2[c];                  // K = X1 X2^T in the reference's 3 x N layout (pose_utils.py:36)
  }
  double U[3][3], s[3], V[3][3];
  svd3(K, U, s, V);
  // Z = diag(1, 1, sign(det(U V^T))) acts on the SMALLEST singular value (numpy sorts them in descending order: pose_utils.py:42-46)
  int mn = 0;
  if (s[1] < s[mn]) mn = 1;
  if (s[2] < s[mn]) mn = 2;
  const double dsign = det3(U) * det3(V) >= 0.0 ? 1.0 : -1.0;               // det(U V^T) = det U det V
  double Rm[3][3], tr = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += (k == mn ? dsign : 1.0) * V[r][k] * U[c][k];   // R = V Z U^T
      Rm[r][c] = acc;
    }
  for (int k = 0; k < 3; ++k) tr += (k == mn ? dsign : 1.0) * s[k];          // trace(R K) = sum_k z_k s_k
  const double scale = tr / var1;
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = mu2[r] - scale * (Rm[r][0] * mu1[0] + Rm[r][1] * mu1[1] + Rm[r][2] * mu1[2]);
  double s_all = 0.0, s_vis = 0.0;
  for (int j = 0; j < J; ++j) {
    double e2 = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double h = scale * (Rm[r][0] * (double)p[3 * j] + Rm[r][1] * (double)p[3 * j + 1] + Rm[r][2] * (double)p[3 * j + 2]) + t[r];
      if (aligned) aligned[((size_t)i * J + j) * 3 + r] = (float)h;
      const double d = h - (double)g[3 * j + r];
      e2 += d * d;
    }
    const double e = sqrt(e2);
    if (per_joint) per_joint[(size_t)i * J + j] = (float)e;
    s_all += e;
    if (mask && mask[(size_t)b * J + j]) s_vis += e;
  }
  if (mean) mean[i] = (float)(s_all / J);
  if (vis_sum) vis_sum[i] = (float)(mask ? s_vis : s_all);
  if (invis_sum) invis_sum[i] = (float)(mask ? s_all - s_vis : 0.0);
}

// ------------------------------------------------------------------------------------------------ diversity
__global__ __launch_bounds__(64) void diversity_kernel(const float* __restrict__ joints, const uint8_t* __restrict__ mask, int invert, float* __restrict__ std_out,
                                                       float* __restrict__ apd_out, int S, int J) {
  const int b = blockIdx.x, j = threadIdx.x;
  const float* a = joints + (size_t)b * S * J * 3;
  float sd = 0.f, pd = 0.f, sel = 0.f;
  if (j < J) {
    bool on = true;
    if (mask) on = (mask[(size_t)b * J + j] != 0) != (invert != 0);
    sel = on ? 1.f : 0.f;
    // unbiased std over the samples per coordinate (two passes), mean over the three coordinates (test_egohmr.py:453-455)
    float mu[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s)
      for (int c = 0; c < 3; ++c) mu[c] += a[((size_t)s * J + j) * 3 + c];
    for (int c = 0; c < 3; ++c) mu[c] /= (float)S;
    float var[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s)
      for (int c = 0; c < 3; ++c) { const float d = a[((size_t)s * J + j) * 3 + c] - mu[c]; var[c] += d * d; }
    sd = (sqrtf(var[0] / (float)(S - 1)) + sqrtf(var[1] / (float)(S - 1)) + sqrtf(var[2] / (float)(S - 1))) / 3.f;      // (S = 1: 0 / 0 = NaN, as torch.std)
    // sum over ORDERED sample pairs of the joint distance (:471-476)
    for (int s = 0; s < S; ++s)
      for (int t = s + 1; t < S; ++t) {
        const float* u = a + ((size_t)s * J + j) * 3;
        const float* v = a + ((size_t)t * J + j) * 3;
        const float dx = u[0] - v[0], dy = u[1] - v[1], dz = u[2] - v[2];
        pd += 2.f * sqrtf(dx * dx + dy * dy + dz * dz);
      }
  }
  const float cnt = wave_sum(sel);
  const float ssd = wave_sum(sel > 0.f ? sd : 0.f), spd = wave_sum(sel > 0.f ? pd : 0.f);
  if (j == 0) {
    if (std_out) std_out[b] = ssd / cnt;                                        // no selected joint: 0 / 0 = NaN like the reference's empty mean
    if (apd_out) apd_out[b] = spd / cnt / (float)S / (float)(S - 1) / 2.f;
  }
}

}  // namespace

extern "C" int ehm_rotmat_to_angle_axis(const float* R, float* aa, int64_t n, void* stream) {
  EHM_CHECK_ARG(R && aa && n >= 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(rotmat_to_aa_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, R, aa, n);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_rotmat_to_angle_axis_bwd(const float* R, const float* gaa, float* gR, int64_t n, void* stream) {
  EHM_CHECK_ARG(R && gaa && gR && n >= 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(rotmat_to_aa_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, R, gaa, gR, n);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_eval_point_errors(const ehm_eval_points_desc* d, void* stream) {
  EHM_CHECK_ARG(d && d->pred && d->gt && d->mean && d->B > 0 && d->S > 0 && d->P > 0);
  EHM_CHECK_ARG(d->pred_points >= d->P && d->gt_points >= d->P && d->origin_point < d->P);
  EHM_CHECK_ARG(d->origin_point < 0 || (!d->pred_origin && !d->gt_origin));
  hipLaunchKernelGGL(point_errors_kernel, dim3((unsigned)(d->B * d->S)), dim3(256), 0, (hipStream_t)stream, d->pred, d->gt, d->pred_origin, d->gt_origin, d->mask,
                     d->per_point, d->mean, d->vis_sum, d->invis_sum, d->S, d->P, d->pred_points, d->gt_points, d->origin_point);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_eval_procrustes(const float* pred, const float* gt, const uint8_t* mask, float* aligned, float* per_joint, float* mean, float* vis_sum,
                                   float* invis_sum, int B, int S, int J, void* stream) {
  EHM_CHECK_ARG(pred && gt && B > 0 && S > 0 && J >= 3 && J <= PJ_MAX && (aligned || per_joint || mean));
  const int n = B * S;
  hipLaunchKernelGGL(procrustes_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, pred, gt, mask, aligned, per_joint, mean, vis_sum,
                     invis_sum, n, S, J);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_eval_diversity(const float* joints, const uint8_t* mask, int invert, float* std_out, float* apd_out, int B, int S, int J, void* stream) {
  EHM_CHECK_ARG(joints && (std_out || apd_out) && B > 0 && S > 0 && J > 0 && J <= 64);
  hipLaunchKernelGGL(diversity_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, joints, mask, invert, std_out, apd_out, S, J);
  EHM_LAUNCH_CHECK();
  return 0;
}
