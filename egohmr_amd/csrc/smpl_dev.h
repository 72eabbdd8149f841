// Device-side view of the SMPL constants, shared by smpl.hip (forward) and guidance.hip (backward).
#pragma once
#include "common.h"

constexpr int kBG = 8;       // bodies per block of the skinning VJP (guidance.hip)
constexpr int kBGF = 16;     // bodies per block of the forward skinning kernel: halves the L2 re-reads of the 17 MB pose basis
constexpr int kVT = 256;     // vertices per skinning block
constexpr int kPoseBasis = 207;

struct Tree {
  int8_t parent[kJ];
  int8_t depth[kJ];
  int max_depth;
};

struct SmplDev {
  int V;
  int n_extra;
  float* v_template;   // [V*3]
  float* shape_t;      // [10][V*3]   shapedirs transposed (basis-major like posedirs)
  const float* posedirs;  // [207][V*3] caller-owned (smplx layout already streams well)
  float* w_t;          // [24][V]     lbs_weights transposed
  int sparse4;         // 1 when every vertex has <= 4 non-zero skinning weights (true for SMPL): use w_idx / w_val
  int32_t* w_idx;      // [4][V]      joint ids of the non-zero weights (padding: joint 0 with weight 0)
  float* w_val;        // [4][V]
  float* J_template;   // [24][3]     J_regressor . v_template
  float* J_shape;      // [24][3][10] J_regressor . shapedirs
  int32_t* extra_idx;  // [n_extra]
  // blend basis [posedirs; shapedirs] (K = 207 + 10 -> 224) as split-f16 MFMA B fragments (skin_mfma_kernel):
  // PDf[vertex tile of 32][k-step of 16][coord][hi/lo][lane][8 halves], values x pd_scale (power of two)
  const void* PDf;
  float pd_scale;
  Tree tree;
};

constexpr int kBlendK = 224;          // 207 pose-corrective + 10 shape coefficients, padded to 14 MFMA k-steps of 16
constexpr int kBlendSteps = kBlendK / 16;


static __device__ __forceinline__ void rot6d_to_R(float a1x, float a1y, float a1z, float a2x, float a2y, float a2z, float (&R)[9]) {
  // utils/geometry.py:61-66; F.normalize = x / max(||x||_2, 1e-12).  Every product / sum is rounded separately
  // (no FMA contraction) like the eager torch ops, so the cancellation in a2 - (b1.a2) b1 for nearly parallel
  // a1, a2 behaves as in the reference.
  const float n1 = fmaxf(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(a1x, a1x), __fmul_rn(a1y, a1y)), __fmul_rn(a1z, a1z))), 1e-12f);
  const float b1x = __fdiv_rn(a1x, n1), b1y = __fdiv_rn(a1y, n1), b1z = __fdiv_rn(a1z, n1);
  const float d = __fadd_rn(__fadd_rn(__fmul_rn(b1x, a2x), __fmul_rn(b1y, a2y)), __fmul_rn(b1z, a2z));
  const float ux = __fsub_rn(a2x, __fmul_rn(d, b1x)), uy = __fsub_rn(a2y, __fmul_rn(d, b1y)), uz = __fsub_rn(a2z, __fmul_rn(d, b1z));
  const float n2 = fmaxf(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz))), 1e-12f);
  const float b2x = __fdiv_rn(ux, n2), b2y = __fdiv_rn(uy, n2), b2z = __fdiv_rn(uz, n2);
  const float b3x = __fsub_rn(__fmul_rn(b1y, b2z), __fmul_rn(b1z, b2y));
  const float b3y = __fsub_rn(__fmul_rn(b1z, b2x), __fmul_rn(b1x, b2z));
  const float b3z = __fsub_rn(__fmul_rn(b1x, b2y), __fmul_rn(b1y, b2x));
  R[0] = b1x; R[1] = b2x; R[2] = b3x;
  R[3] = b1y; R[4] = b2y; R[5] = b3y;
  R[6] = b1z; R[7] = b2z; R[8] = b3z;
}


struct ehm_smpl {
  SmplDev d{};
  float* arena = nullptr;      // packed constants
  float* ws = nullptr;         // per-call scratch: R [cap,24,9] + A [cap,24,12] (+ backward scratch)
  int ws_cap = 0;
  void* pdf = nullptr;         // SmplDev::PDf storage
  void* pf = nullptr;          // blend coefficients of the current batch as MFMA A fragments [ceil(B/32)][14][hi/lo][64][8 halves]
  int pf_cap = 0;              // bodies
};
