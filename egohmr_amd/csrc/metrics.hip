// Post-loop geometry metric for gfx950: nearest-neighbour squared distance between two point sets,
// the only arithmetic of the reference's contact score (utils/pytorch3d_chamfer_distance.py:152-156 ->
// pytorch3d.ops.knn_points(K=1), a CUDA extension absent here; call site test_egohmr.py:496-505 with
// x = posed body vertices [B*S,6890,3] and y = scene points [B*S,20000,3]).
// Brute force on purpose (P1 x P2 is only 1.4e8 pairs per body): the reference set streams through LDS in SoA tiles,
// every thread owns one query point and reads the tile as wave-wide broadcasts (ds_read_b128 = 4 points per 3 reads).
#include "common.h"
#include "egohmr_hip.h"

namespace {

constexpr int NN_TILE = 2048;   // reference points per LDS tile: 3 * 2048 * 4 B = 24 KiB

__global__ __launch_bounds__(256) void nn_dist2_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       float* __restrict__ dist, int32_t* __restrict__ idx, int P1, int P2) {
  __shared__ __attribute__((aligned(16))) float s[3 * NN_TILE];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int q = blockIdx.x * 256 + tid;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (q < P1) {
    const float* p = x + ((size_t)b * P1 + q) * 3;
    px = p[0]; py = p[1]; pz = p[2];
  }
  float best = 3.4e38f;
  int bi = 0;
  for (int t0 = 0; t0 < P2; t0 += NN_TILE) {
    const int n = min(NN_TILE, P2 - t0);
    __syncthreads();
    for (int i = tid; i < NN_TILE; i += 256) {
      const bool ok = i < n;
      const float* p = y + ((size_t)b * P2 + t0 + (ok ? i : 0)) * 3;
      s[i] = ok ? p[0] : 3.0e18f;                   // padding is infinitely far away
      s[NN_TILE + i] = ok ? p[1] : 3.0e18f;
      s[2 * NN_TILE + i] = ok ? p[2] : 3.0e18f;
    }
    __syncthreads();
    const int n4 = (n + 3) & ~3;
    for (int i = 0; i < n4; i += 4) {
      const f32x4 X = *(const f32x4*)(s + i), Y = *(const f32x4*)(s + NN_TILE + i), Z = *(const f32x4*)(s + 2 * NN_TILE + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dx = px - X[e], dy = py - Y[e], dz = pz - Z[e];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) { best = d2; bi = t0 + i + e; }      // strict <: first minimum wins
      }
    }
  }
  if (q < P1) {
    dist[(size_t)b * P1 + q] = best;
    if (idx) idx[(size_t)b * P1 + q] = bi;
  }
}

}  // namespace

extern "C" int ehm_nn_dist2(const float* x, const float* y, float* dist2, int32_t* idx, int B, int P1, int P2, void* stream) {
  EHM_CHECK_ARG(x && y && dist2 && B > 0 && P1 > 0 && P2 > 0);
  hipLaunchKernelGGL(nn_dist2_kernel, dim3((unsigned)ceil_div(P1, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, dist2, idx, P1, P2);
  EHM_LAUNCH_CHECK();
  return 0;
}
