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

// one fixed operation order for the squared distance, so that the brute-force and the grid search return the same bits
__device__ __forceinline__ float nn_d2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

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
        const float d2 = nn_d2(px - X[e], py - Y[e], pz - Z[e]);
        if (d2 < best) { best = d2; bi = t0 + i + e; }      // strict <: first minimum wins
      }
    }
  }
  if (q < P1) {
    dist[(size_t)b * P1 + q] = best;
    if (idx) idx[(size_t)b * P1 + q] = bi;
  }
}


// ------------------------------------------------------------------------------------------------ the same search on a uniform grid
// For the driver's contact score (test_egohmr.py:496-505: 6890 body vertices against 20 000 scene points per body, B x S bodies) brute force is
// 1.4e8 distance evaluations per body.  The reference set is binned once per cloud into a uniform grid (~4 points per cell: counting sort, as
// guidance.hip does for the collision term) and a query walks Chebyshev shells of cells around its own until no unvisited cell can hold a closer
// point - exact, same distance bits as nn_dist2_kernel (nn_d2), ties -> the smallest point index (what "first minimum wins" gives there).
struct NNGrid {            // per cloud, in the workspace
  float ox, oy, oz, inv_h, h;
  int nx, ny, nz;
};
constexpr int kNNMaxCells = 32768;     // cells per cloud (one block scans them)
constexpr int kNNMaxDim = 64;

__global__ __launch_bounds__(256) void nn_grid_bbox_kernel(const float* __restrict__ y, NNGrid* __restrict__ grids, int P2) {
  __shared__ float red[6][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = tid; i < P2; i += 256) {
    const float* p = y + ((size_t)b * P2 + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = p[c];
      if (v == v && fabsf(v) < 3.0e38f) { lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }      // (non-finite points take no part in the box; they land in a border cell)
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float l = lo[c], h = hi[c];
    for (int o = 32; o > 0; o >>= 1) { l = fminf(l, __shfl_xor(l, o)); h = fmaxf(h, __shfl_xor(h, o)); }
    if ((tid & 63) == 0) { red[c][tid >> 6] = l; red[3 + c][tid >> 6] = h; }
  }
  __syncthreads();
  if (tid == 0) {
    float e[3], o3[3];
    for (int c = 0; c < 3; ++c) {
      const float l = fminf(fminf(red[c][0], red[c][1]), fminf(red[c][2], red[c][3]));
      const float h = fmaxf(fmaxf(red[3 + c][0], red[3 + c][1]), fmaxf(red[3 + c][2], red[3 + c][3]));
      o3[c] = l <= h ? l : 0.f;
      e[c] = l <= h ? fmaxf(h - l, 1.0e-6f) : 1.0e-6f;
    }
    // ~4 points per cell if the points filled the box; surfaces fill it less, so cells hold fewer
    float hcell = cbrtf(e[0] * e[1] * e[2] / fmaxf((float)P2 * 0.25f, 1.f));
    hcell = fmaxf(hcell, fmaxf(e[0], fmaxf(e[1], e[2])) / (float)kNNMaxDim);
    int n[3];
    for (;;) {
      for (int c = 0; c < 3; ++c) n[c] = min(kNNMaxDim, max(1, (int)ceilf(e[c] / hcell)));
      if ((long long)n[0] * n[1] * n[2] <= kNNMaxCells) break;
      hcell *= 1.26f;
    }
    NNGrid g;
    g.ox = o3[0]; g.oy = o3[1]; g.oz = o3[2]; g.h = hcell; g.inv_h = 1.f / hcell; g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    grids[b] = g;
  }
}

__device__ __forceinline__ int nn_cell_coord(float v, float o, float inv_h, int n) {
  const float f = (v - o) * inv_h;
  int c = f == f ? (int)fminf(fmaxf(floorf(f), 0.f), (float)(n - 1)) : 0;      // clamped: points / queries outside the box use the border cells
  return c;
}

__global__ void nn_grid_count_kernel(const float* __restrict__ y, const NNGrid* __restrict__ grids, int* __restrict__ counts, int P2) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P2) return;
  const NNGrid g = grids[b];
  const float* p = y + ((size_t)b * P2 + i) * 3;
  const int c = (nn_cell_coord(p[2], g.oz, g.inv_h, g.nz) * g.ny + nn_cell_coord(p[1], g.oy, g.inv_h, g.ny)) * g.nx + nn_cell_coord(p[0], g.ox, g.inv_h, g.nx);
  atomicAdd(counts + (size_t)b * (kNNMaxCells + 1) + c, 1);
}

// exclusive scan of a cloud's cell counts (one 1024-thread block per cloud); counts become starts, `fill` a second copy for the scatter
__global__ __launch_bounds__(1024) void nn_grid_scan_kernel(int* __restrict__ counts, int* __restrict__ fill) {
  __shared__ int part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  int* c = counts + (size_t)b * (kNNMaxCells + 1);
  int* f = fill + (size_t)b * (kNNMaxCells + 1);
  constexpr int PER = kNNMaxCells / 1024;
  int loc[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { loc[k] = c[tid * PER + k]; s += loc[k]; }
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
#pragma unroll
  for (int k = 0; k < PER; ++k) { c[tid * PER + k] = run; f[tid * PER + k] = run; run += loc[k]; }
  if (tid == 1023) c[kNNMaxCells] = run;
}

__global__ void nn_grid_scatter_kernel(const float* __restrict__ y, const NNGrid* __restrict__ grids, int* __restrict__ fill, f32x4* __restrict__ slots, int P2) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P2) return;
  const NNGrid g = grids[b];
  const float* p = y + ((size_t)b * P2 + i) * 3;
  const int c = (nn_cell_coord(p[2], g.oz, g.inv_h, g.nz) * g.ny + nn_cell_coord(p[1], g.oy, g.inv_h, g.ny)) * g.nx + nn_cell_coord(p[0], g.ox, g.inv_h, g.nx);
  const int pos = atomicAdd(fill + (size_t)b * (kNNMaxCells + 1) + c, 1);
  slots[(size_t)b * P2 + pos] = f32x4{p[0], p[1], p[2], __builtin_bit_cast(float, i)};
}

__global__ __launch_bounds__(256) void nn_grid_query_kernel(const float* __restrict__ x, const NNGrid* __restrict__ grids, const int* __restrict__ starts,
                                                            const f32x4* __restrict__ slots, float* __restrict__ dist, int32_t* __restrict__ idx, int P1, int P2,
                                                            unsigned long long* __restrict__ evals) {
  const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  unsigned long long ne = 0;
  if (q < P1) {
    const NNGrid g = grids[b];
    const int* st = starts + (size_t)b * (kNNMaxCells + 1);
    const f32x4* sl = slots + (size_t)b * P2;
    const float* p = x + ((size_t)b * P1 + q) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const int cx = nn_cell_coord(px, g.ox, g.inv_h, g.nx), cy = nn_cell_coord(py, g.oy, g.inv_h, g.ny), cz = nn_cell_coord(pz, g.oz, g.inv_h, g.nz);
    float best = 3.4e38f;
    int bi = 0x7fffffff;
    const int rmax = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    auto visit = [&](int cell) __attribute__((always_inline)) {
      const int s0 = st[cell], s1 = st[cell + 1];
      for (int k = s0; k < s1; ++k) {
        const f32x4 v = sl[k];
        const float d2 = nn_d2(px - v[0], py - v[1], pz - v[2]);
        const int vi = __builtin_bit_cast(int, v[3]);
        if (d2 < best || (d2 == best && vi < bi)) { best = d2; bi = vi; }
      }
      ne += (unsigned long long)(s1 - s0);
    };
    for (int r = 0; r <= rmax; ++r) {
      // the shell of Chebyshev radius r around my cell, clipped to the grid
      const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1), y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1), x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
      for (int z = z0; z <= z1; ++z)
        for (int yy = y0; yy <= y1; ++yy) {
          const int row = (z * g.ny + yy) * g.nx;
          if (z == cz - r || z == cz + r || yy == cy - r || yy == cy + r) {          // a face row: every x of the cube is on the shell
            for (int xx = x0; xx <= x1; ++xx) visit(row + xx);
          } else {                                                                    // an inner row: its two ends
            if (cx - r >= 0) visit(row + cx - r);
            if (cx + r <= g.nx - 1) visit(row + cx + r);
          }
        }
      // every unvisited point lies outside the cube of cells [c - r, c + r]: at least `gap` away along some axis (faces on the grid's border do not
      // count - nothing lies beyond them).  1e-5 of a cell covers the rounding of the points' cell assignment.
      float gap = 3.4e38f;
      if (cx - r > 0) gap = fminf(gap, px - (g.ox + (float)(cx - r) * g.h));
      if (cx + r < g.nx - 1) gap = fminf(gap, (g.ox + (float)(cx + r + 1) * g.h) - px);
      if (cy - r > 0) gap = fminf(gap, py - (g.oy + (float)(cy - r) * g.h));
      if (cy + r < g.ny - 1) gap = fminf(gap, (g.oy + (float)(cy + r + 1) * g.h) - py);
      if (cz - r > 0) gap = fminf(gap, pz - (g.oz + (float)(cz - r) * g.h));
      if (cz + r < g.nz - 1) gap = fminf(gap, (g.oz + (float)(cz + r + 1) * g.h) - pz);
      if (gap > 3.0e38f) break;                       // the cube covers the grid
      const float gs = gap - 1.0e-5f * g.h;
      if (gs > 0.f && best <= gs * gs) break;
    }
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

extern "C" int64_t ehm_nn_grid_workspace_bytes(int B, int P2) {
  if (B <= 0 || P2 <= 0) return 0;
  // grids | starts [B][cells + 1] | fill [B][cells + 1] | slots [B][P2] float4 | evaluation counter
  return round_up((int64_t)B * sizeof(NNGrid), 256) + 2 * round_up((int64_t)B * (kNNMaxCells + 1) * 4, 256) + (int64_t)B * P2 * 16;
}

extern "C" int ehm_nn_dist2_grid(const float* x, const float* y, float* dist2, int32_t* idx, int B, int P1, int P2, void* workspace, int64_t workspace_bytes,
                                 uint64_t* evals, void* stream) {
  EHM_CHECK_ARG(x && y && dist2 && B > 0 && P1 > 0 && P2 > 0 && workspace && workspace_bytes >= ehm_nn_grid_workspace_bytes(B, P2));
  hipStream_t st = (hipStream_t)stream;
  const int64_t tab = round_up((int64_t)B * (kNNMaxCells + 1) * 4, 256);
  char* w = (char*)workspace;
  NNGrid* grids = (NNGrid*)w;
  w += round_up((int64_t)B * sizeof(NNGrid), 256);
  int* starts = (int*)w;
  w += tab;
  int* fill = (int*)w;
  w += tab;
  f32x4* slots = (f32x4*)w;
  EHM_HIP(hipMemsetAsync(starts, 0, (size_t)B * (kNNMaxCells + 1) * 4, st));
  hipLaunchKernelGGL(nn_grid_bbox_kernel, dim3(B), dim3(256), 0, st, y, grids, P2);
  const dim3 pg((unsigned)ceil_div(P2, 256), B);
  hipLaunchKernelGGL(nn_grid_count_kernel, pg, dim3(256), 0, st, y, grids, starts, P2);
  hipLaunchKernelGGL(nn_grid_scan_kernel, dim3(B), dim3(1024), 0, st, starts, fill);
  hipLaunchKernelGGL(nn_grid_scatter_kernel, pg, dim3(256), 0, st, y, grids, fill, slots, P2);
  hipLaunchKernelGGL(nn_grid_query_kernel, dim3((unsigned)ceil_div(P1, 256), B), dim3(256), 0, st, x, grids, starts, slots, dist2, idx, P1, P2,
                     (unsigned long long*)evals);
  EHM_LAUNCH_CHECK();
  return 0;
}
