// Collision guidance for gfx950: the per-step gradient of EgoHMR.guide_coll
//   models/egohmr/egohmr.py:517-570 (VolSMPL twin: models/egohmr/egohmr_volsmpl.py:582-629)
// as consumed by p_sample_with_grad (diffusion/gaussian_diffusion.py:378-385).
//
// The reference differentiates a learned collision network (COAP / VolumetricSMPL - unavailable offline, DESIGN.md
// section 5) with torch autograd through SMPL LBS and rot6d_to_rotmat, in a Python loop over the batch.  Here:
//   bbox_kernel          per-body vertex bounding box                                       (egohmr.py:550-551)
//   select_kernel        ordered compaction of the scene points inside the box              (egohmr.py:552)
//   nearest_kernel       build-defined proxy: nearest body vertex of every selected point from an LDS-resident copy of
//                        the body (SoA, broadcast reads), hinge relu(tau - d)^2, scatter of d loss / d vertex
//   skin_bwd_kernel      VJP of the skinning + pose-corrective blend w.r.t. posed rest vertices and the 24 transforms
//   posefeat_bwd_mfma_kernel  VJP of the 207-basis pose blend (a [B,20670] x [20670,207] contraction, exact-f32 MFMA)
//   chain_bwd_kernel     one wave per body, lane = joint: reverse kinematic chain through LDS (children -> parent in
//                        fixed order, deterministic), then the Gram-Schmidt (rot6d) VJP
//   finish_kernel        -1/denom scaling, x2 for joints 3..23, zeroing of the upper-body joints (egohmr.py:562-567)
// The gradient is w.r.t. the DE-NORMALISED 6-D pose, as in the reference (egohmr.py:523-528 rebinds x_t before autograd.grad).
#include "common.h"
#include "egohmr_hip.h"
#include "internal.h"
#include "smpl_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------ selection
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ verts, float* __restrict__ bbox, int V) {
  const int b = blockIdx.x;
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int v = threadIdx.x; v < V; v += 256) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = verts[((size_t)b * V + v) * 3 + c];
      lo[c] = fminf(lo[c], x);
      hi[c] = fmaxf(hi[c], x);
    }
  }
  __shared__ float red[6][256];
#pragma unroll
  for (int c = 0; c < 3; ++c) { red[c][threadIdx.x] = lo[c]; red[3 + c][threadIdx.x] = hi[c]; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        red[c][threadIdx.x] = fminf(red[c][threadIdx.x], red[c][threadIdx.x + o]);
        red[3 + c][threadIdx.x] = fmaxf(red[3 + c][threadIdx.x], red[3 + c][threadIdx.x + o]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) bbox[b * 6 + threadIdx.x] = red[threadIdx.x][0];
}

// ordered stream compaction: idx[b][0..count) = indices of scene points with bb_min <= p <= bb_max (all three coords)
// margin = 0: egohmr.py:550-552; margin = tau: "all points" of the VolSMPL twin (egohmr_volsmpl.py:609-612) - a point farther than tau from
// the box is farther than tau from every vertex and contributes exactly nothing to the hinge, so the selection stays exact
__global__ __launch_bounds__(1024) void select_kernel(const float* __restrict__ scene, const float* __restrict__ bbox,
                                                      int* __restrict__ idx, int* __restrict__ count, int N, float margin) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int wsum[16];
  __shared__ int base;
  if (tid == 0) base = 0;
  float lo[3], hi[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { lo[c] = bbox[b * 6 + c] - margin; hi[c] = bbox[b * 6 + 3 + c] + margin; }
  __syncthreads();
  for (int i0 = 0; i0 < N; i0 += 1024) {
    const int i = i0 + tid;
    bool in = false;
    if (i < N) {
      const float* p = scene + ((size_t)b * N + i) * 3;
      in = p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2];
    }
    const unsigned long long m = __ballot(in);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) woff += wsum[w];
      tot += wsum[w];
    }
    const int bs = base;
    if (in) idx[(size_t)b * N + bs + woff + before] = i;
    __syncthreads();
    if (tid == 0) base = bs + tot;
    __syncthreads();
  }
  if (tid == 0) count[b] = base;
}

// ------------------------------------------------------------------------------------------------ proxy loss
// grid (point chunks of 1024, B); dynamic LDS: body vertices as SoA x[Vp] y[Vp] z[Vp] (Vp = V rounded up to 4)
__global__ __launch_bounds__(1024) void nearest_kernel(const float* __restrict__ verts, const float* __restrict__ scene,
                                                       const int* __restrict__ idx, const int* __restrict__ count,
                                                       float* __restrict__ loss, float* __restrict__ gverts, int* __restrict__ hits,
                                                       int V, int N, float tau) {
  extern __shared__ __attribute__((aligned(16))) float sv[];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int cnt = count[b];
  if (chunk * 1024 >= cnt) return;                       // block-uniform
  const int Vp = (V + 3) & ~3;
  float* sx = sv;
  float* sy = sv + Vp;
  float* sz = sv + 2 * Vp;
  for (int v = tid; v < Vp; v += 1024) {
    const bool ok = v < V;
    const float* p = verts + ((size_t)b * V + (ok ? v : 0)) * 3;
    sx[v] = ok ? p[0] : 3.0e18f;                         // padding vertices are infinitely far away
    sy[v] = ok ? p[1] : 3.0e18f;
    sz[v] = ok ? p[2] : 3.0e18f;
  }
  __syncthreads();
  const int k = chunk * 1024 + tid;
  float contrib = 0.f;
  int nhit = 0;
  if (k < cnt) {
    const float* p = scene + ((size_t)b * N + idx[(size_t)b * N + k]) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float best = 3.4e38f;
    int bi = 0;
    for (int v = 0; v < Vp; v += 4) {
      const f32x4 x = *(const f32x4*)(sx + v), y = *(const f32x4*)(sy + v), z = *(const f32x4*)(sz + v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dx = px - x[e], dy = py - y[e], dz = pz - z[e];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) { best = d2; bi = v + e; }         // strict < keeps the first minimum, like torch.min
      }
    }
    const float d = sqrtf(best + 1e-12f);
    const float h = tau - d;
    if (h > 0.f) {
      contrib = h * h;
      ++nhit;
      if (gverts) {
        const float s = 2.f * h / d;                      // d(h^2)/dv = 2h (p - v)/d
        float* g = gverts + ((size_t)b * V + bi) * 3;
        atomicAdd(g + 0, s * (px - sx[bi]));
        atomicAdd(g + 1, s * (py - sy[bi]));
        atomicAdd(g + 2, s * (pz - sz[bi]));
      }
    }
  }
  // block sum of the hinge terms
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { contrib += __shfl_xor(contrib, o); nhit += __shfl_xor(nhit, o); }
  __shared__ float wred[16];
  __shared__ int hred[16];
  if ((tid & 63) == 0) { wred[tid >> 6] = contrib; hred[tid >> 6] = nhit; }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    int nh = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s += wred[w]; nh += hred[w]; }
    if (s != 0.f) atomicAdd(loss + b, s);
    if (hits && nh) atomicAdd(hits + b, nh);
  }
}

// The same proxy through a uniform grid (exact): the hinge relu(tau - d) is zero unless some body vertex lies within tau of the
// point, and with cells of edge >= tau such a vertex sits in one of the 27 cells around the point's cell - so only those are
// searched (~100 distances instead of 6890).  Points without a vertex in reach contribute exactly nothing in both versions; ties
// resolve to the lowest vertex index like torch.min.  One block per body: vertices (SoA), cell offsets and the cell-sorted vertex
// list live in LDS; the block then walks all selected points of its body.
constexpr int kMaxCells = 4096;
__global__ __launch_bounds__(1024) void nearest_grid_kernel(const float* __restrict__ verts, const float* __restrict__ scene,
                                                            const int* __restrict__ idx, const int* __restrict__ count,
                                                            const float* __restrict__ bbox, float* __restrict__ loss,
                                                            float* __restrict__ gverts, int* __restrict__ hits, int V, int N, float tau,
                                                            unsigned long long* __restrict__ evals) {
  // grid = (bodies, slices): the selected points of a body are dealt out to gridDim.y blocks (wave w of slice s takes points w + 16 s,
  // w + 16 s + 16 slices, ...), each of which builds the body's (cheap: ~11 us) cell grid for itself - one block per body left half of the chip
  // idle at B = 128
  extern __shared__ __attribute__((aligned(16))) float sv[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int slice = blockIdx.y, slices = gridDim.y;
  const int cnt = count[b];
  if (cnt <= 16 * slice) return;                         // block-uniform: nothing for this slice (cnt == 0 included)

  const int Vp = (V + 3) & ~3;
  f32x4* sp = (f32x4*)sv;                                // [Vp] slot i = (x, y, z, vertex id as bits) of the i-th vertex in CELL order
  int* cstart = (int*)(sv + 4 * Vp);                     // [kMaxCells + 1] exclusive offsets
  int* cursor = cstart + kMaxCells + 1;                  // [kMaxCells]
  // The vertices sit SORTED BY CELL, one 16-byte slot each: a candidate costs ONE ds_read_b128.  (Round 2: positions in vertex order = two
  // dependent LDS round trips per candidate; rounds 3-4: three SoA arrays + an id array = three to four 4-byte reads per candidate at addresses
  // that differ from lane to lane - the search, 93 % of the kernel by in-kernel stamps, was bound by LDS bank conflicts.)
  __shared__ int part[1024];
  __shared__ float wred[16];
  __shared__ int hred[16];

  float lo[3], ext[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { lo[c] = bbox[b * 6 + c]; ext[c] = fmaxf(bbox[b * 6 + 3 + c] - lo[c], 1e-6f); }
  float h = fmaxf(tau, cbrtf(ext[0] * ext[1] * ext[2] / 3000.f));
  int nx, ny, nz;
  for (;;) {                                             // uniform: every thread runs the same few iterations
    nx = (int)(ext[0] / h) + 1; ny = (int)(ext[1] / h) + 1; nz = (int)(ext[2] / h) + 1;
    if ((long long)nx * ny * nz <= kMaxCells) break;
    h *= 1.1f;
  }
  const int nc = nx * ny * nz;
  const float ih = 1.f / h;
  auto cell_of = [&](float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = min(nx - 1, max(0, (int)((x - lo[0]) * ih)));
    cy = min(ny - 1, max(0, (int)((y - lo[1]) * ih)));
    cz = min(nz - 1, max(0, (int)((z - lo[2]) * ih)));
  };
  constexpr int kVPT = 8;                                // vertices per thread kept in registers across the build (V <= 8192)
  float vx[kVPT], vy[kVPT], vz[kVPT];
  int vc[kVPT];
#pragma unroll
  for (int i = 0; i < kVPT; ++i) {
    const int v = tid + 1024 * i;
    vc[i] = -1;
    if (v < V) {
      const float* p = verts + ((size_t)b * V + v) * 3;
      vx[i] = p[0]; vy[i] = p[1]; vz[i] = p[2];
      int cx, cy, cz;
      cell_of(vx[i], vy[i], vz[i], cx, cy, cz);
      vc[i] = cx + nx * (cy + ny * cz);
    }
  }
  for (int c = tid; c < nc; c += 1024) cursor[c] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kVPT; ++i)                         // histogram
    if (vc[i] >= 0) atomicAdd(&cursor[vc[i]], 1);
  __syncthreads();
  {                                                      // exclusive scan of up to 4096 counts: 4 per thread + block scan
    int loc[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = 4 * tid + i; loc[i] = c < nc ? cursor[c] : 0; sum += loc[i]; }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    int run = part[tid] - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = 4 * tid + i; if (c < nc) { cstart[c] = run; run += loc[i]; } }
    if (tid == 1023) cstart[nc] = part[1023];
  }
  __syncthreads();
  for (int c = tid; c < nc; c += 1024) cursor[c] = cstart[c];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kVPT; ++i)                         // scatter (order inside a cell is arbitrary; the search below does not care)
    if (vc[i] >= 0) {
      const int slot = atomicAdd(&cursor[vc[i]], 1);
      sp[slot] = f32x4{vx[i], vy[i], vz[i], __builtin_bit_cast(float, tid + 1024 * i)};
    }
  __syncthreads();

  // ---- the search: one WAVE per point.  (Rounds 2-4: one LANE per point - a body's ~1000 selected points have 20 candidates on average, but
  //      a point next to a dense part of the body has a thousand or more, and the block waited for those few lanes: 0.9 ms per guided step at
  //      1280 bodies for 24 M distance evaluations, 0.13 % of the vector rate.)  The (up to) nine x-runs of the point's 27 cells are nine slot
  //      ranges; lanes 0-8 fetch their bounds, a 9-step prefix turns them into ONE candidate list that the 64 lanes stride through; the wave
  //      minimum breaks ties by the lower vertex index like torch.min.
  float contrib = 0.f;
  int nhit = 0;
  unsigned int nev = 0;                                  // distance evaluations (bench.py's roofline of the search)
  const int lane = tid & 63, wave = tid >> 6;
  // a wave's points arrive 64 at a time, one per lane (two dependent global loads - index, then position - per 64 points instead of per point:
  // as a per-point load they were 1.3 us of each point's 1.4), and are handed round with v_readlane
  for (int k0 = wave + 16 * slice; k0 < cnt; k0 += 64 * 16 * slices) {
    const int kl = k0 + lane * 16 * slices;
    float lx = 0.f, ly = 0.f, lz = 0.f;
    if (kl < cnt) {
      const float* p = scene + ((size_t)b * N + idx[(size_t)b * N + kl]) * 3;
      lx = p[0]; ly = p[1]; lz = p[2];
    }
    const int npts = min(64, (cnt - k0 + 16 * slices - 1) / (16 * slices));
  for (int pi = 0; pi < npts; ++pi) {
    const float px = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lx), pi));
    const float py = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ly), pi));
    const float pz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lz), pi));
    int cx, cy, cz;
    cell_of(px, py, pz, cx, cy, cz);
    int rs = 0, rl = 0;                                  // my range (lanes 0-8): first slot, length
    if (lane < 9) {
      const int dz = cz - 1 + lane / 3, dy = cy - 1 + lane % 3;
      if (dz >= 0 && dz < nz && dy >= 0 && dy < ny) {
        const int c0 = max(cx - 1, 0) + nx * (dy + ny * dz), c1 = min(cx + 1, nx - 1) + nx * (dy + ny * dz);
        rs = cstart[c0];
        rl = cstart[c1 + 1] - rs;
      }
    }
    int start[9], pref[10];                              // wave-uniform copies (scalar registers)
    pref[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      start[r] = __builtin_amdgcn_readlane(rs, r);
      pref[r + 1] = pref[r] + __builtin_amdgcn_readlane(rl, r);
    }
    const int total = pref[9];
    if (total == 0) continue;                            // (wave-uniform) no body vertex within reach of this point: most points of a bounding box
    nev += lane == 0 ? (unsigned int)total : 0u;
    float best = 3.4e38f;
    int bi = 0x7fffffff, bslot = 0;
    for (int i = lane; i < total; i += 64) {
      int slot = start[0] + i;
#pragma unroll
      for (int r = 1; r < 9; ++r) slot = i >= pref[r] ? start[r] + (i - pref[r]) : slot;
      const f32x4 q = sp[slot];
      const float ex = px - q[0], ey = py - q[1], ez = pz - q[2];
      const float d2 = ex * ex + ey * ey + ez * ez;
      const float qw = q[3];                             // (hipcc: __builtin_bit_cast of a vector ELEMENT expression reads element 0 - go through a scalar)
      const int v = __builtin_bit_cast(int, qw);
      if (d2 < best || (d2 == best && v < bi)) { best = d2; bi = v; bslot = slot; }
    }
    // wave minimum of (distance, vertex index): four DPP steps inside the 16-lane rows (__shfl_xor is ds_bpermute, an LDS round trip per value and
    // step: 18 of them were ~900 of a point's ~2300 cycles), then the four row results through v_readlane
    auto take = [&](float ob, int oi, int os) { if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; bslot = os; } };
#define EHM_DPP_MIN(CTRL) take(dpp_move<CTRL>(best), __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, bslot, CTRL, 0xF, 0xF, true))
    EHM_DPP_MIN(0xB1); EHM_DPP_MIN(0x4E); EHM_DPP_MIN(0x141); EHM_DPP_MIN(0x140);
#undef EHM_DPP_MIN
#pragma unroll
    for (int rrow = 16; rrow < 64; rrow += 16)
      take(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best), rrow)), __builtin_amdgcn_readlane(bi, rrow),
           __builtin_amdgcn_readlane(bslot, rrow));    // (lane 0 ends with the wave's minimum; the other lanes' values are not used)
    if (lane == 0 && bi != 0x7fffffff) {
      const float d = sqrtf(best + 1e-12f);
      const float hh = tau - d;
      if (hh > 0.f) {
        contrib += hh * hh;
        ++nhit;
        if (gverts) {
          const float s = 2.f * hh / d;                   // d(h^2)/dv = 2h (p - v)/d
          float* g = gverts + ((size_t)b * V + bi) * 3;
          const f32x4 q = sp[bslot];
          atomicAdd(g + 0, s * (px - q[0]));
          atomicAdd(g + 1, s * (py - q[1]));
          atomicAdd(g + 2, s * (pz - q[2]));
        }
      }
    }
  }
  }
  if (evals) {                                           // (a profile is open: one atomic per wave)
    unsigned int ne = nev;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ne += __shfl_xor(ne, o);
    if ((tid & 63) == 0 && ne) atomicAdd(evals, (unsigned long long)ne);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { contrib += __shfl_xor(contrib, o); nhit += __shfl_xor(nhit, o); }
  if ((tid & 63) == 0) { wred[tid >> 6] = contrib; hred[tid >> 6] = nhit; }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    int nh = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s += wred[w]; nh += hred[w]; }
    if (s != 0.f) atomicAdd(loss + b, s);
    if (hits && nh) atomicAdd(hits + b, nh);
  }
}


// ------------------------------------------------------------------------------------------------ LBS backward
// thread = vertex, 8 bodies per block (same tiling as the forward skin_kernel).  In: d loss/d verts (gv, overwritten
// in place by d loss/d posed-rest-vertex).  Out: gA[b][24][12] += sum_v w[v,j] [gv (x) vp | gv]  (LDS, then global atomics)
// `vposed` (may be nullptr): the blended rest vertices [B,V,3] as the forward's matrix-core skinning left them - then they are READ (24 bytes per
// vertex and body with a gradient) instead of recomputed from the 217-row blend basis per 8 bodies (2.6 KB of basis per vertex: 2.9 GB of L2
// reads per guided step at 1280 bodies, the kernel's 0.75 ms).
__global__ __launch_bounds__(kVT, 4) void skin_bwd_kernel(const float* __restrict__ betas, const float* __restrict__ Rws,
                                                          const float* __restrict__ A, SmplDev S, float* __restrict__ gv_io,
                                                          float* __restrict__ gA, int B, int v_tiles, int b_groups,
                                                          const float* __restrict__ vposed) {
  __shared__ __attribute__((aligned(16))) float sA[kBG][kJ][12];
  __shared__ float sG[kBG][kJ][12];
  __shared__ float sPF[kBG][kPoseBasis + 1];
  __shared__ float sBeta[kBG][10];
  __shared__ int any_grad;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, k = bid >> 3;
  const int vt = (k / b_groups) * 8 + xcd, bg = k % b_groups;
  if (vt >= v_tiles) return;
  const int b0 = bg * kBG, nb = min(kBG, B - b0), tid = threadIdx.x;
  const int v = vt * kVT + tid;
  const bool vok = v < S.V;

  // which bodies have a non-zero incoming gradient on this vertex?
  float gv[kBG][3];
  bool mine = false;
#pragma unroll
  for (int bb = 0; bb < kBG; ++bb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gv[bb][c] = (vok && bb < nb) ? gv_io[((size_t)(b0 + bb) * S.V + v) * 3 + c] : 0.f;
      mine |= gv[bb][c] != 0.f;
    }
  }
  if (tid == 0) any_grad = 0;
  __syncthreads();
  if (mine) any_grad = 1;
  __syncthreads();
  if (!any_grad) return;                                  // the hinge touches few vertices: most blocks stop here

  for (int i = tid; i < kBG * kJ * 12; i += kVT) {
    (&sA[0][0][0])[i] = (i / (kJ * 12)) < nb ? A[(size_t)b0 * kJ * 12 + i] : 0.f;
    (&sG[0][0][0])[i] = 0.f;
  }
  if (!vposed) {
    for (int i = tid; i < kBG * kPoseBasis; i += kVT) {
      const int bb = i / kPoseBasis, p = i % kPoseBasis, e = p % 9;
      sPF[bb][p] = bb < nb ? Rws[((size_t)(b0 + bb) * kJ + 1) * 9 + p] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f) : 0.f;
    }
    if (tid < kBG * 10) sBeta[tid / 10][tid % 10] = (tid / 10) < nb ? betas[(size_t)b0 * 10 + tid] : 0.f;
  }
  __syncthreads();

  if (mine) {
    const int V3 = S.V * 3;
    // the blended rest vertex of the bodies with gradient: read (the forward left it), or recomputed (forward skin_kernel)
    float vp[kBG][3];
    if (vposed) {
#pragma unroll
      for (int bb = 0; bb < kBG; ++bb) {
        const bool live = bb < nb && (gv[bb][0] != 0.f || gv[bb][1] != 0.f || gv[bb][2] != 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) vp[bb][c] = live ? vposed[((size_t)(b0 + bb) * S.V + v) * 3 + c] : 0.f;
      }
    } else {
      const float t0 = S.v_template[v * 3 + 0], t1 = S.v_template[v * 3 + 1], t2 = S.v_template[v * 3 + 2];
#pragma unroll
      for (int bb = 0; bb < kBG; ++bb) { vp[bb][0] = 0.f; vp[bb][1] = 0.f; vp[bb][2] = 0.f; }
#pragma unroll 2
      for (int l = 0; l < 10; ++l) {
        const float s0 = S.shape_t[(size_t)l * V3 + v * 3 + 0], s1 = S.shape_t[(size_t)l * V3 + v * 3 + 1],
                    s2 = S.shape_t[(size_t)l * V3 + v * 3 + 2];
#pragma unroll
        for (int bb = 0; bb < kBG; ++bb) {
          const float be = sBeta[bb][l];
          vp[bb][0] = fmaf(be, s0, vp[bb][0]); vp[bb][1] = fmaf(be, s1, vp[bb][1]); vp[bb][2] = fmaf(be, s2, vp[bb][2]);
        }
      }
#pragma unroll
      for (int bb = 0; bb < kBG; ++bb) { vp[bb][0] += t0; vp[bb][1] += t1; vp[bb][2] += t2; }
      const float* pd = S.posedirs + (size_t)v * 3;
#pragma unroll 2
      for (int p = 0; p < kPoseBasis; ++p) {
        const float d0 = pd[(size_t)p * V3 + 0], d1 = pd[(size_t)p * V3 + 1], d2 = pd[(size_t)p * V3 + 2];
#pragma unroll
        for (int bb = 0; bb < kBG; ++bb) {
          const float f = sPF[bb][p];
          vp[bb][0] = fmaf(f, d0, vp[bb][0]); vp[bb][1] = fmaf(f, d1, vp[bb][1]); vp[bb][2] = fmaf(f, d2, vp[bb][2]);
        }
      }
    }
    const float* wv = S.w_t + v;
#pragma unroll
    for (int bb = 0; bb < kBG; ++bb) {
      const float g0 = gv[bb][0], g1 = gv[bb][1], g2 = gv[bb][2];
      if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
      __builtin_amdgcn_sched_barrier(0);
      float T[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) T[e] = 0.f;
      for (int j = 0; j < kJ; ++j) {
        const float wj = wv[(size_t)j * S.V];
        if (wj == 0.f) continue;
        const float* a = &sA[bb][j][0];
        T[0] = fmaf(wj, a[0], T[0]); T[1] = fmaf(wj, a[1], T[1]); T[2] = fmaf(wj, a[2], T[2]);
        T[3] = fmaf(wj, a[4], T[3]); T[4] = fmaf(wj, a[5], T[4]); T[5] = fmaf(wj, a[6], T[5]);
        T[6] = fmaf(wj, a[8], T[6]); T[7] = fmaf(wj, a[9], T[7]); T[8] = fmaf(wj, a[10], T[8]);
        // d/dA_j : w [gv (x) vp | gv]
        float* g = &sG[bb][j][0];
        const float w0 = wj * g0, w1 = wj * g1, w2 = wj * g2;
        atomicAdd(g + 0, w0 * vp[bb][0]); atomicAdd(g + 1, w0 * vp[bb][1]); atomicAdd(g + 2, w0 * vp[bb][2]); atomicAdd(g + 3, w0);
        atomicAdd(g + 4, w1 * vp[bb][0]); atomicAdd(g + 5, w1 * vp[bb][1]); atomicAdd(g + 6, w1 * vp[bb][2]); atomicAdd(g + 7, w1);
        atomicAdd(g + 8, w2 * vp[bb][0]); atomicAdd(g + 9, w2 * vp[bb][1]); atomicAdd(g + 10, w2 * vp[bb][2]); atomicAdd(g + 11, w2);
      }
      // d/d vp = T.R^T gv   (in place)
      float* o = gv_io + ((size_t)(b0 + bb) * S.V + v) * 3;
      o[0] = T[0] * g0 + T[3] * g1 + T[6] * g2;
      o[1] = T[1] * g0 + T[4] * g1 + T[7] * g2;
      o[2] = T[2] * g0 + T[5] * g1 + T[8] * g2;
    }
  }
  __syncthreads();
  for (int i = tid; i < nb * kJ * 12; i += kVT) {
    const float s = (&sG[0][0][0])[i];
    if (s != 0.f) atomicAdd(gA + (size_t)b0 * kJ * 12 + i, s);
  }
}

// gpf[b][p] = sum_col posedirs[p][col] * gvp[b][col];  block = (joint-1: 9 basis rows) x (8 bodies)
// The same contraction on the matrix cores in exact float32 (v_mfma_f32_32x32x2_f32): gpf[b][k] = sum_c gverts[b][c] * posedirs[k][c], a
// [B, 3V] x [3V, 207] GEMM whose BOTH operands are k-contiguous rows - a lane (row / column = l & 31, h = l >> 5) loads 16 bytes of its row at
// k + 4 h, MFMA i of an 8-k group contracts k + i and k + 4 + i.  A wave owns 32 bodies x ALL 224 (207) columns (7 accumulator tiles: the
// gradient rows, 106 MB at 1280 bodies, are read exactly once; the 17 MB basis once per 32 bodies, from L2) over 1 / (4 kPfSplit) of K; the four
// waves of a block reduce through LDS, the kPfSplit partial tiles are added by posefeat_sum_kernel in a fixed order (deterministic, no
// atomics).  11 GFLOP dense: the vector-ALU kernel below (which skips the gradient's zero columns) took 0.38 - 0.58 ms at 1280 bodies.
constexpr int kPfSplit = 8, kPfTiles = 7;
__global__ __launch_bounds__(256, 2) void posefeat_bwd_mfma_kernel(const float* __restrict__ gv, const float* __restrict__ posedirs,
                                                                float* __restrict__ part, int B, int V3) {
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mi = lane & 31, h = lane >> 5;
  const int m0 = 32 * blockIdx.x, sp = blockIdx.y;
  const int G = (V3 + 7) / 8, Gfull = V3 / 8, c = sp * 4 + wave;
  const int g0 = (int)((long long)c * G / (4 * kPfSplit)), g1 = (int)((long long)(c + 1) * G / (4 * kPfSplit));
  const int row = m0 + mi;
  const bool row_ok = row < B;
  const float* xa = gv + (size_t)(row_ok ? row : 0) * V3 + 4 * h;
  const float* wb[kPfTiles];
  bool col_ok[kPfTiles];
#pragma unroll
  for (int t = 0; t < kPfTiles; ++t) {
    col_ok[t] = 32 * t + mi < kPoseBasis;
    wb[t] = posedirs + (size_t)(col_ok[t] ? 32 * t + mi : 0) * V3 + 4 * h;
  }
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef f32x4_t f32x4_u __attribute__((aligned(4)));   // rows of 3 V floats and a caller-owned basis: only dword-aligned (global dwordx4 loads accept that; the TYPE must say so)
  f32x16_t acc[kPfTiles];
#pragma unroll
  for (int t = 0; t < kPfTiles; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  const int ge = g1 < Gfull ? g1 : Gfull;
#pragma unroll 2
  for (int g = g0; g < ge; ++g) {
    f32x4_t a = *(const f32x4_u*)(xa + 8 * g);                       // (rows start at 4-byte multiples only: unaligned 16-byte loads)
    if (!row_ok) a = zero;
#pragma unroll
    for (int t = 0; t < kPfTiles; ++t) {
      f32x4_t b = *(const f32x4_u*)(wb[t] + 8 * g);
      if (!col_ok[t]) b = zero;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[t], 0, 0, 0);
    }
  }
  if (g1 > Gfull) {                                                    // the last, partial 8-k group (3V % 8 != 0)
    f32x4_t a = zero;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 8 * Gfull + 4 * h + i;
      if (k < V3 && row_ok) a[i] = gv[(size_t)row * V3 + k];
    }
#pragma unroll
    for (int t = 0; t < kPfTiles; ++t) {
      f32x4_t b = zero;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 8 * Gfull + 4 * h + i;
        if (k < V3 && col_ok[t]) b[i] = posedirs[(size_t)(32 * t + mi) * V3 + k];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[t], 0, 0, 0);
    }
  }
  // accumulator layout: column = mi, row = (r & 3) + 8 (r >> 2) + 4 h; one 32 x 32 tile at a time through the LDS reduction
#pragma unroll
  for (int t = 0; t < kPfTiles; ++t) {
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][(r & 3) + 8 * (r >> 2) + 4 * h][mi] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0 && 32 * t + mi < 208) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m0 + rr < B) part[((size_t)sp * B + m0 + rr) * 208 + 32 * t + mi] = ((acc[t][r] + red[0][rr][mi]) + red[1][rr][mi]) + red[2][rr][mi];
      }
    }
    __syncthreads();
  }
}
__global__ void posefeat_sum_kernel(const float* __restrict__ part, float* __restrict__ gpf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = part[i];
#pragma unroll
  for (int q = 1; q < kPfSplit; ++q) s += part[(size_t)q * n + i];
  gpf[i] = s;
}


// VJP of R = rot6d_to_rotmat(a1, a2) (utils/geometry.py:59-66): gR [9] row-major -> (ga1, ga2)
__device__ __forceinline__ void rot6d_bwd(float a1x, float a1y, float a1z, float a2x, float a2y, float a2z, const float (&gR)[9],
                                          float (&ga1)[3], float (&ga2)[3]) {
  const float n1r = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), n1 = fmaxf(n1r, 1e-12f);
  const float b1[3] = {a1x / n1, a1y / n1, a1z / n1};
  const float a2[3] = {a2x, a2y, a2z};
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float n2r = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), n2 = fmaxf(n2r, 1e-12f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float g1[3] = {gR[0], gR[3], gR[6]}, g2[3] = {gR[1], gR[4], gR[7]};
  const float g3[3] = {gR[2], gR[5], gR[8]};
  // b3 = b1 x b2
  g1[0] += b2[1] * g3[2] - b2[2] * g3[1]; g1[1] += b2[2] * g3[0] - b2[0] * g3[2]; g1[2] += b2[0] * g3[1] - b2[1] * g3[0];
  g2[0] += g3[1] * b1[2] - g3[2] * b1[1]; g2[1] += g3[2] * b1[0] - g3[0] * b1[2]; g2[2] += g3[0] * b1[1] - g3[1] * b1[0];
  // b2 = u / max(|u|, eps)
  float gu[3];
  if (n2r > 1e-12f) {
    const float t = b2[0] * g2[0] + b2[1] * g2[1] + b2[2] * g2[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) gu[c] = (g2[c] - b2[c] * t) / n2;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) gu[c] = g2[c] / n2;
  }
  // u = a2 - (b1.a2) b1
  const float gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    ga2[c] = gu[c] - gub1 * b1[c];
    g1[c] += -d * gu[c] - gub1 * a2[c];
  }
  // b1 = a1 / max(|a1|, eps)
  if (n1r > 1e-12f) {
    const float t = b1[0] * g1[0] + b1[1] * g1[1] + b1[2] * g1[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) ga1[c] = (g1[c] - b1[c] * t) / n1;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) ga1[c] = g1[c] / n1;
  }
}

__global__ void rot6d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gR, float* __restrict__ gx, int64_t n, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = x + i * 6;
  float g[9], ga1[3], ga2[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = gR[i * 9 + k];
  if (mode == 1) {
    rot6d_bwd(p[0], p[2], p[4], p[1], p[3], p[5], g, ga1, ga2);
    gx[i * 6 + 0] = ga1[0]; gx[i * 6 + 2] = ga1[1]; gx[i * 6 + 4] = ga1[2];
    gx[i * 6 + 1] = ga2[0]; gx[i * 6 + 3] = ga2[1]; gx[i * 6 + 5] = ga2[2];
  } else {
    rot6d_bwd(p[0], p[1], p[2], p[3], p[4], p[5], g, ga1, ga2);
    gx[i * 6 + 0] = ga1[0]; gx[i * 6 + 1] = ga1[1]; gx[i * 6 + 2] = ga1[2];
    gx[i * 6 + 3] = ga2[0]; gx[i * 6 + 4] = ga2[1]; gx[i * 6 + 5] = ga2[2];
  }
}

// one wave per body, lane = joint: forward chain (as pose_chain_kernel), reverse chain, rot6d VJP
__global__ __launch_bounds__(64) void chain_bwd_kernel(const float* __restrict__ betas, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ std_,
                                                       SmplDev S, const float* __restrict__ gA, const float* __restrict__ gpf,
                                                       float* __restrict__ gpose) {
  __shared__ float sC[kJ][12];      // contribution of child c to its parent's dG
  const int b = blockIdx.x, lane = threadIdx.x;
  const int j = lane < kJ ? lane : 0;
  float p6[6], R[9];
#pragma unroll
  for (int c = 0; c < 6; ++c) p6[c] = x[(size_t)b * kPoseDim + j * 6 + c] * std_[j * 6 + c] + mean[j * 6 + c];
  rot6d_to_R(p6[0], p6[2], p6[4], p6[1], p6[3], p6[5], R);
  float Jx[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 10; ++l) s = fmaf(S.J_shape[j * 30 + c * 10 + l], betas[(size_t)b * 10 + l], s);
    Jx[c] = S.J_template[j * 3 + c] + s;
  }
  const int par = S.tree.parent[j], plane = par < 0 ? 0 : par, my_depth = S.tree.depth[j];
  float t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pj = __shfl(Jx[c], plane);
    t[c] = par < 0 ? Jx[c] : Jx[c] - pj;
  }
  float G[12], GP[9];               // G: own global transform; GP: parent's global rotation
#pragma unroll
  for (int r = 0; r < 3; ++r) { G[r * 4] = R[r * 3]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2]; G[r * 4 + 3] = t[r]; }
#pragma unroll
  for (int e = 0; e < 9; ++e) GP[e] = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
  for (int d = 1; d <= S.tree.max_depth; ++d) {
    float P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = __shfl(G[k], plane);
    if (my_depth == d) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float p0 = P[r * 4], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2], p3 = P[r * 4 + 3];
        GP[r * 3] = p0; GP[r * 3 + 1] = p1; GP[r * 3 + 2] = p2;
        G[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
        G[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
        G[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
        G[r * 4 + 3] = p0 * t[0] + p1 * t[1] + p2 * t[2] + p3;
      }
    }
  }
  // dG from dA:  A.R = G.R,  A.t = G.t - G.R J   =>  dG.R = dA.R - dA.t (x) J,  dG.t = dA.t
  float dG[12];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float gt = lane < kJ ? gA[((size_t)b * kJ + j) * 12 + r * 4 + 3] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dG[r * 4 + c] = (lane < kJ ? gA[((size_t)b * kJ + j) * 12 + r * 4 + c] : 0.f) - gt * Jx[c];
    dG[r * 4 + 3] = gt;
  }
  // reverse levels: a child's dG reaches its parent as  dGp.R += dG.R R^T + dG.t (x) t ,  dGp.t += dG.t
  for (int d = S.tree.max_depth; d >= 1; --d) {
    if (lane < kJ && my_depth == d) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          sC[j][r * 4 + c] = dG[r * 4] * R[c * 3] + dG[r * 4 + 1] * R[c * 3 + 1] + dG[r * 4 + 2] * R[c * 3 + 2] + dG[r * 4 + 3] * t[c];
        sC[j][r * 4 + 3] = dG[r * 4 + 3];
      }
    }
    __syncthreads();
    if (lane < kJ && my_depth == d - 1) {
      for (int c = j + 1; c < kJ; ++c) {                     // children in index order: deterministic sum
        if (S.tree.parent[c] == j) {
#pragma unroll
          for (int e = 0; e < 12; ++e) dG[e] += sC[c][e];
        }
      }
    }
    __syncthreads();
  }
  if (lane >= kJ) return;
  // local rotation gradient: dR = Gp.R^T dG.R (+ pose-feature path for joints 1..23)
  float gR[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = GP[0 * 3 + r] * dG[0 * 4 + c] + GP[1 * 3 + r] * dG[1 * 4 + c] + GP[2 * 3 + r] * dG[2 * 4 + c];
      if (j > 0) s += gpf[(size_t)b * 208 + (j - 1) * 9 + r * 3 + c];
      gR[r * 3 + c] = s;
    }
  float ga1[3], ga2[3];
  rot6d_bwd(p6[0], p6[2], p6[4], p6[1], p6[3], p6[5], gR, ga1, ga2);
  float* o = gpose + (size_t)b * kPoseDim + j * 6;
  o[0] = ga1[0]; o[2] = ga1[1]; o[4] = ga1[2];
  o[1] = ga2[0]; o[3] = ga2[1]; o[5] = ga2[2];
}

__global__ void finish_kernel(const float* __restrict__ gpose, float* __restrict__ grad, int B, float denom) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * kPoseDim) return;
  const int j = (i % kPoseDim) / 6;
  // egohmr.py:562-567: grad of -loss.mean() (or .sum()); joints 0..2 x1, 3..23 x2; joints {0,3,6,9,12..23} zeroed
  const bool keep = (j == 1 || j == 2 || j == 4 || j == 5 || j == 7 || j == 8 || j == 10 || j == 11);
  const float scale = keep ? (j >= 3 ? 2.f : 1.f) : 0.f;
  grad[i] = keep ? -gpose[i] / denom * scale : 0.f;
}

struct Scratch {
  float* bbox;   // [B][6]
  int* count;    // [B]
  int* idx;      // [B][N]
  float* gA;     // [B][24][12]
  float* gpf;    // [B][208]
  float* gpf_part;   // [kPfSplit][B][208] partial sums of posefeat_bwd_mfma_kernel
};

Scratch carve_scratch(void* base, int B, int N) {
  char* p = (char*)base;
  Scratch s;
  s.bbox = (float*)p;  p += round_up((int64_t)B * 6 * 4, 256);
  s.count = (int*)p;   p += round_up((int64_t)B * 4, 256);
  s.idx = (int*)p;     p += round_up((int64_t)B * N * 4, 256);
  s.gA = (float*)p;    p += round_up((int64_t)B * kJ * 12 * 4, 256);
  s.gpf = (float*)p;   p += round_up((int64_t)B * 208 * 4, 256);
  s.gpf_part = (float*)p;
  return s;
}

// Scratch of the stand-alone entry points (ehm_collision_proxy / _query, ehm_smpl_backward_rot6d): one buffer per (host thread, device).
// Calls on ONE device from one thread must be stream-ordered with respect to each other (same stream, or externally synchronised):
// ehm_sample_loop does not use this - it carves its scratch out of the caller's workspace.
constexpr int kMaxDevices = 16;
thread_local void* g_scratch[kMaxDevices] = {};
thread_local int64_t g_scratch_bytes[kMaxDevices] = {};
int own_scratch(int64_t bytes, void** out) {
  int dev = 0;
  EHM_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) { ehm_set_error("device ordinal %d out of range", dev); return EHM_EINVAL; }
  if (bytes > g_scratch_bytes[dev]) {
    if (g_scratch[dev]) { EHM_HIP(hipDeviceSynchronize()); (void)hipFree(g_scratch[dev]); }
    g_scratch[dev] = nullptr;
    g_scratch_bytes[dev] = 0;
    EHM_HIP(hipMalloc(&g_scratch[dev], bytes));
    g_scratch_bytes[dev] = bytes;
  }
  *out = g_scratch[dev];
  return 0;
}

int collision_impl(const float* verts, const float* scene, float* loss, float* gverts, int* hits, int B, int V, int N, float tau,
                   float margin, const Scratch& s, hipStream_t st) {
  if (gverts) EHM_HIP(hipMemsetAsync(gverts, 0, (size_t)B * V * 3 * sizeof(float), st));
  if (hits) EHM_HIP(hipMemsetAsync(hits, 0, (size_t)B * sizeof(int), st));
  EHM_HIP(hipMemsetAsync(loss, 0, (size_t)B * sizeof(float), st));
  hipLaunchKernelGGL(bbox_kernel, dim3(B), dim3(256), 0, st, verts, s.bbox, V);
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(1024), 0, st, scene, s.bbox, s.idx, s.count, N, margin);
  const int Vp = (V + 3) & ~3;
  const size_t lds = (size_t)3 * Vp * sizeof(float);
  // (per device, and cheap: set unconditionally rather than once per process)
  EHM_HIP(hipFuncSetAttribute((const void*)nearest_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  if (lds > 160 * 1024 - 256) {
    ehm_set_error("collision proxy: %d vertices do not fit the 160 KiB LDS", V);
    return EHM_EINVAL;
  }
  const size_t lds_grid = (size_t)4 * Vp * sizeof(float) + (size_t)(2 * kMaxCells + 1) * sizeof(int) + 16;   // 16-byte slots (x, y, z, id) + the cell offsets
  if (V <= 8192 && lds_grid <= 160 * 1024 - 4608) {          // (the grid kernel keeps 8 vertices per thread in registers)
    EHM_HIP(hipFuncSetAttribute((const void*)nearest_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4608));
    int slices = ehm_num_cus() / (B > 0 ? B : 1);         // one block per CU (the vertex arrays take half of its LDS)
    slices = slices < 1 ? 1 : (slices > 4 ? 4 : slices);
    if (slices > (N + 1023) / 1024) slices = (N + 1023) / 1024;
    hipLaunchKernelGGL(nearest_grid_kernel, dim3(B, slices), dim3(1024), lds_grid, st, verts, scene, s.idx, s.count, s.bbox, loss, gverts, hits, V,
                       N, tau, ehm_prof_evals_ptr());
  } else {   // bodies too large for the in-LDS grid: brute force over an LDS-resident copy of the vertices
    hipLaunchKernelGGL(nearest_kernel, dim3((unsigned)ceil_div(N, 1024), B), dim3(1024), lds, st, verts, scene, s.idx, s.count, loss,
                       gverts, hits, V, N, tau);
  }
  EHM_LAUNCH_CHECK();
  return 0;
}

int backward_impl(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std_, const float* Rws,
                  const float* Aws, float* gverts, float* gpose, int B, const Scratch& s, hipStream_t st, const float* vposed = nullptr) {
  const SmplDev& d = h->d;
  EHM_HIP(hipMemsetAsync(s.gA, 0, (size_t)B * kJ * 12 * sizeof(float), st));
  const int v_tiles = (int)ceil_div(d.V, kVT), b_groups = (int)ceil_div(B, kBG);
  {
    EhmProfScope ps(EHM_PROF_G_SKIN_BWD, st);
    hipLaunchKernelGGL(skin_bwd_kernel, dim3((unsigned)(round_up(v_tiles, 8) * b_groups)), dim3(kVT), 0, st, betas, Rws, Aws, d, gverts,
                       s.gA, B, v_tiles, b_groups, vposed);
  }
  {
    EhmProfScope ps(EHM_PROF_G_POSEFEAT_BWD, st);
    hipLaunchKernelGGL(posefeat_bwd_mfma_kernel, dim3((unsigned)ceil_div(B, 32), kPfSplit), dim3(256), 0, st, gverts, d.posedirs, s.gpf_part, B, d.V * 3);
    hipLaunchKernelGGL(posefeat_sum_kernel, dim3((unsigned)ceil_div((int64_t)B * 208, 256)), dim3(256), 0, st, s.gpf_part, s.gpf, B * 208);
  }
  hipLaunchKernelGGL(chain_bwd_kernel, dim3(B), dim3(64), 0, st, betas, x, mean, std_, d, s.gA, s.gpf, gpose);
  EHM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int64_t ehm_guidance_scratch_bytes(int B, int N) {
  return round_up((int64_t)B * 6 * 4, 256) + round_up((int64_t)B * 4, 256) + round_up((int64_t)B * N * 4, 256) +
         round_up((int64_t)B * kJ * 12 * 4, 256) + round_up((int64_t)B * 208 * 4, 256) + round_up((int64_t)kPfSplit * B * 208 * 4, 256);
}

int ehm_guidance_impl(ehm_smpl* smpl, const float* betas, const float* x, const float* mean, const float* std_,
                      const float* scene, int B, int N, float tau, float denom, float margin, float* verts_ws, float* joints_ws, float* R_ws,
                      float* A_ws, float* gverts, float* loss, float* gpose, float* grad, void* scratch, hipStream_t st, float* vposed_ws) {
  const int V = smpl->d.V;
  const Scratch s = carve_scratch(scratch, B, N);
  if (vposed_ws && !ehm_smpl_writes_vposed(smpl, B)) vposed_ws = nullptr;       // (small batches / dense skinning weights: the VALU forward does not leave them)
  int rc = ehm_smpl_forward_impl(smpl, betas, x, true, mean, std_, verts_ws, joints_ws, R_ws, A_ws, nullptr, B, st, vposed_ws);   // egohmr.py:528-537
  if (rc == 0) {
    EhmProfScope ps(EHM_PROF_G_NEAREST, st);      // memsets + bbox + select + nearest_grid_kernel (the search dominates)
    rc = collision_impl(verts_ws, scene, loss, gverts, nullptr, B, V, N, tau, margin, s, st);
  }
  if (rc == 0) rc = backward_impl(smpl, betas, x, mean, std_, R_ws, A_ws, gverts, gpose, B, s, st, vposed_ws);
  if (rc == 0) {
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)ceil_div((int64_t)B * kPoseDim, 256)), dim3(256), 0, st, gpose, grad, B, denom);
    EHM_LAUNCH_CHECK();
  }
  return rc;
}

extern "C" int ehm_collision_proxy(const float* verts, const float* scene, float* loss, float* gverts, int B, int V, int N, float tau,
                                   void* stream) {
  EHM_CHECK_ARG(verts && scene && loss && gverts && B > 0 && V > 0 && N > 0 && tau > 0.f);
  void* sc = nullptr;
  int rc = own_scratch(ehm_guidance_scratch_bytes(B, N), &sc);
  if (rc) return rc;
  return collision_impl(verts, scene, loss, gverts, nullptr, B, V, N, tau, 0.f, carve_scratch(sc, B, N), (hipStream_t)stream);
}

extern "C" int ehm_collision_query(const float* verts, const float* scene, float* loss, float* gverts, int32_t* hits, int B, int V, int N,
                                   float tau, int all_points, void* stream) {
  EHM_CHECK_ARG(verts && scene && loss && B > 0 && V > 0 && N > 0 && tau > 0.f);
  void* sc = nullptr;
  int rc = own_scratch(ehm_guidance_scratch_bytes(B, N), &sc);
  if (rc) return rc;
  return collision_impl(verts, scene, loss, gverts, hits, B, V, N, tau, all_points ? tau : 0.f, carve_scratch(sc, B, N), (hipStream_t)stream);
}

extern "C" int ehm_smpl_backward_rot6d(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std_,
                                       const float* gverts, float* gpose6d, int B, void* stream) {
  EHM_CHECK_ARG(h && betas && x && mean && std_ && gverts && gpose6d && B > 0);
  hipStream_t st = (hipStream_t)stream;
  const int V = h->d.V;
  // scratch: guidance block + R [B,24,9] + A [B,24,12] + joints + a private copy of gverts (the VJP works in place)
  const int64_t gs = ehm_guidance_scratch_bytes(B, 1);
  const int64_t extra = round_up((int64_t)B * kJ * 9 * 4, 256) + round_up((int64_t)B * kJ * 12 * 4, 256) +
                        round_up((int64_t)B * (kJ + 64) * 3 * 4, 256) + round_up((int64_t)B * V * 3 * 4, 256);
  void* sc = nullptr;
  int rc = own_scratch(gs + extra, &sc);
  if (rc) return rc;
  char* p = (char*)sc + gs;
  float* Rws = (float*)p;     p += round_up((int64_t)B * kJ * 9 * 4, 256);
  float* Aws = (float*)p;     p += round_up((int64_t)B * kJ * 12 * 4, 256);
  float* jws = (float*)p;     p += round_up((int64_t)B * (kJ + 64) * 3 * 4, 256);
  float* gcopy = (float*)p;
  EHM_HIP(hipMemcpyAsync(gcopy, gverts, (size_t)B * V * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  rc = ehm_smpl_pose_impl(h, betas, x, mean, std_, Rws, Aws, jws, B, st);
  if (rc) return rc;
  return backward_impl(h, betas, x, mean, std_, Rws, Aws, gcopy, gpose6d, B, carve_scratch(sc, B, 1), st);
}

extern "C" int ehm_guidance_grad_finish(const float* gpose6d, const float* loss, float* grad, int B, float denom, void* stream) {
  EHM_CHECK_ARG(gpose6d && grad && B > 0 && denom > 0.f);
  (void)loss;  // an all-zero loss batch has an all-zero gradient already (egohmr.py:561,569-570)
  hipLaunchKernelGGL(finish_kernel, dim3((unsigned)ceil_div((int64_t)B * kPoseDim, 256)), dim3(256), 0, (hipStream_t)stream, gpose6d,
                     grad, B, denom);
  EHM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ehm_rot6d_to_rotmat_bwd(const float* x6d, const float* gR, float* gx, int64_t n, int mode, void* stream) {
  EHM_CHECK_ARG(n >= 0 && (mode == 0 || mode == 1));
  if (n == 0) return 0;
  EHM_CHECK_ARG(x6d && gR && gx);
  hipLaunchKernelGGL(rot6d_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x6d, gR, gx, n, mode);
  EHM_LAUNCH_CHECK();
  return 0;
}
