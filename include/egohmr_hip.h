/*
 * egohmr_hip.h - C ABI of libegohmr_hip.so: the MI355X (gfx950) kernels of the EgoHMR stage-2
 * diffusion-sampling hot path.
 *
 * The reference (sanweiliti/EgoHMR) is single-process eager PyTorch and has NO FFI boundary of
 * its own.  Each entry point below therefore names the reference Python
 * interface (file:line under /root/reference) whose arithmetic it replaces; the Python host side
 * (egohmr_amd/) keeps those interfaces' names and calls in here through ctypes.
 *
 * Conventions
 *   - every `const float*` / `float*` is a DEVICE pointer, float32, contiguous row-major, owned by
 *     the caller, unless the parameter comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     stream-ordered, never synchronises and never allocates after *_create, so a sequence of
 *     calls may be captured into a hipGraph;
 *   - return value: 0 on success, negative on error (-22 bad argument, -12 out of memory,
 *     -5 HIP runtime error).  ehm_last_error() returns a thread-local description;
 *   - handles are opaque; create/destroy are host-synchronous.
 */
#ifndef EGOHMR_HIP_H
#define EGOHMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EHM_NUM_JOINTS 24
#define EHM_NUM_BETAS 10
#define EHM_POSE_DIM 144 /* 24 joints x 6-D rotation */

const char* ehm_last_error(void);
/* compile-time facts, for the loader's sanity check: returns "gfx950" */
const char* ehm_target_arch(void);
/* optional parts this library was built with, space separated ("" for the default build): "stamps" (-DEHM_STAMPS: in-kernel time stamps for
 * tools/stamp_*.py).  (The experiment engines of rounds 4 - 5 - the one-launch sampling loop, the 96 x 64 wave tile - are gone from the tree: their
 * records are docs/EXPERIMENTS.md 3.2 / 3.7, their code is in the history.) */
const char* ehm_build_features(void);

/* ------------------------------------------------------------------ geometry ------------------ */
/* utils/geometry.py:47-66 rot6d_to_rotmat(x, rot6d_mode).  x6d [n,6] -> R [n,3,3].
 * mode 0 = 'prohmr' (a1 = x[0:3], a2 = x[3:6]); 1 = 'diffusion' (a1 = x[0::2], a2 = x[1::2]). */
int ehm_rot6d_to_rotmat(const float* x6d, float* R, int64_t n, int mode, void* stream);
/* vector-Jacobian product of the above (autograd of geometry.py:47-66 as used under
 * torch.enable_grad() in models/egohmr/egohmr.py:518-529): gR [n,3,3] -> gx [n,6]. */
int ehm_rot6d_to_rotmat_bwd(const float* x6d, const float* gR, float* gx, int64_t n, int mode, void* stream);
/* utils/konia_transform.py:316-340 rotation_matrix_to_angle_axis (-> rotation_matrix_to_quaternion :349-443 in WXYZ order, quaternion_to_angle_axis
 * :560-630; eps = 1e-6 in the clamps, safe_zero_division :343-346, torch_safe_atan2 :44-47): R [n,3,3] -> aa [n,3].  The `full_pose` feed of the
 * COAP / VolSMPL collision models (models/egohmr/egohmr.py:495, :540; egohmr_volsmpl.py:555, :596). */
int ehm_rotmat_to_angle_axis(const float* R, float* aa, int64_t n, void* stream);
/* vector-Jacobian product of the above through the branch the forward took (autograd through torch.where / clamp_min, as under
 * torch.enable_grad() in egohmr.py:518-545): gaa [n,3] -> gR [n,3,3]. */
int ehm_rotmat_to_angle_axis_bwd(const float* R, const float* gaa, float* gR, int64_t n, void* stream);

/* ------------------------------------------------------------------ SMPL body model ----------- */
typedef struct ehm_smpl ehm_smpl;

/* smplx.create('data/smpl', model_type='smpl', ...) (models/egohmr/egohmr.py:105-107): uploads /
 * re-packs the model constants.  v_template [V,3], shapedirs [V,3,10], posedirs [207,V*3],
 * J_regressor [24,V], lbs_weights [V,24] are device pointers (smplx buffer layouts);
 * parents [24] and extra_joint_vertex_ids [n_extra] are HOST int32 arrays. */
int ehm_smpl_create(ehm_smpl** out, const float* v_template, const float* shapedirs, const float* posedirs,
                    const float* J_regressor, const float* lbs_weights, const int32_t* parents,
                    const int32_t* extra_joint_vertex_ids, int num_verts, int n_extra, void* stream);
void ehm_smpl_destroy(ehm_smpl* h);

/* smplx SMPL.forward(betas, body_pose, global_orient, pose2rot=False) as called at
 * models/egohmr/egohmr.py:276,492,537 and test_egohmr.py:291.
 * betas [B,10]; rotmats [B,24,3,3] (global_orient then body_pose); verts [B,V,3];
 * joints [B,24+n_extra,3]; A_out (may be NULL) [B,24,3,4] = the skinning transforms. */
int ehm_smpl_forward(ehm_smpl* h, const float* betas, const float* rotmats, float* verts, float* joints,
                     float* A_out, int B, void* stream);

/* egohmr.py:258-260 + :276 fused: x [B,144] (normalised 6-D pose), pose6d = x*std+mean,
 * R = rot6d_to_rotmat(pose6d,'diffusion'), then SMPL.forward.  mean/std [144]; R_out (may be NULL)
 * [B,24,3,3]; pose6d_out (may be NULL) [B,144]. */
int ehm_smpl_forward_rot6d(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std,
                           float* verts, float* joints, float* R_out, float* pose6d_out, float* A_out, int B,
                           void* stream);

/* ------------------------------------------------------------------ Modulated-GCN denoiser ---- */
/* One ModulatedGraphConv (+ optional BatchNorm1d) in the reference's parameter layout
 * (models/egohmr/modulated_gcn/modulated_gcn_conv.py:16-37, modulated_gcn.py:12-13). */
typedef struct {
  const float* W;         /* [2, in_dim, out_dim] */
  const float* M;         /* [24, out_dim]        */
  const float* adj2;      /* [24, 24]             */
  const float* bias;      /* [out_dim]            */
  const float* bn_weight; /* [out_dim] or NULL (no BatchNorm/ReLU after this conv) */
  const float* bn_bias;
  const float* bn_mean;
  const float* bn_var;
  int in_dim;
  int out_dim;
} ehm_gconv_params;

typedef struct ehm_gcn ehm_gcn;

/* ModulatedGCN(adj, in_dim, hid_dim, out_dim=6, num_layers) (modulated_gcn.py:60-97) with the
 * step-invariant part of the input conv hoisted (SURVEY.md section 0 finding 5):
 *   adj        [24,24]  fixed adjacency (egohmr.py:86-93)
 *   input_conv          epilogue parameters of gconv_input (its W is consumed by the caller's
 *                       one-off projections, so .W may be NULL; in_dim is ignored)
 *   hidden              2*num_blocks convs hid->hid, order gconv_layers.{b}.gconv1, gconv2
 *   output_conv         hid->6, no batch norm
 * HOST array of structs holding DEVICE pointers. */
int ehm_gcn_create(ehm_gcn** out, const float* adj, const ehm_gconv_params* input_conv,
                   const ehm_gconv_params* hidden, int num_hidden, const ehm_gconv_params* output_conv,
                   int hid_dim, void* stream);
void ehm_gcn_destroy(ehm_gcn* h);

/* Arithmetic of the hidden convs (docs/EXPERIMENTS.md 3.2).  0 = f32-input MFMA (exact f32 products; what a fresh handle is in);
 * 1 = "f16x3": f16 MFMA on hi/lo-split operands, three products per term, f32 accumulate (22-bit operands,
 * f32-grade results); 2 = plain f16 operands and f16 activation storage (NOT parity-grade on its own - BASELINE config 5's
 * fp16 denoiser, and the early steps of ehm_sample_desc.lowprec_steps).
 * Activation matrices exchanged between ehm_gcn_input_layer -> ehm_gcn_hidden_layer / _stack: mode 0 float32 [rows_pad,hid];
 * mode 1 the opaque "X2" split format (same byte size), except that the LAST hidden conv writes float32; mode 2 f16
 * [rows_pad,hid] throughout.  ehm_gcn_output_layer reads what the handle's mode produces.  ehm_gcn_pack/unpack_activations convert float32 <-> the mode's format (tests, interop). */
int ehm_gcn_set_precision(ehm_gcn* h, int mode);
int ehm_gcn_get_precision(const ehm_gcn* h);
/* group = ehm_gcn_activation_group(h): 32 = X2 split format (mode 1), 0 = plain f16 (mode 2); pass it to pack / unpack */
int ehm_gcn_activation_group(const ehm_gcn* h);
/* Size the handle's internal scratch (chained-launch counters, output-conv responses) for batches of up to max_bodies bodies x
 * passes.  ehm_gcn_create reserves 256 x 2; a larger batch grows the scratch on first use (a hipMalloc - so call this first when
 * the launches are going to be captured into a hipGraph). */
int ehm_gcn_reserve(ehm_gcn* h, int max_bodies, int passes);
int ehm_gcn_pack_activations(const float* X, void* X2, int64_t rows, int K, int group, void* stream);
int ehm_gcn_unpack_activations(const void* X2, float* X, int64_t rows, int K, int group, void* stream);
/* ehm_gcn_pack_activations into the handle's own activation format (modes 1 / 2; K = hid_dim) WITH the range guard of the conv kernels: a value with
 * |x| >= 65504 (clamped: see ehm_gcn_stack_status) raises bit 2 of the handle's status word. */
int ehm_gcn_pack_activations_checked(ehm_gcn* h, const float* X, void* X2, int64_t rows, void* stream);

/* rows of the activation matrices must be padded to a multiple of this many rows (zero-filled) */
int ehm_gcn_row_tile(void);

/* gconv_input (modulated_gcn.py:21-28 applied to egohmr.py:236/:245's concatenated feature) with
 * the conditioning/timestep part pre-projected.  For virtual body vb = p*B + b (p = 0 conditional
 * pass, p = 1 image-masked pass, egohmr.py:239-246), joint j, branch k (W[0] / W[1]):
 *   pre_k = (p==0 ? vis[b,j] * h_img[b,k,:] : 0) + h_oth[b,k,:] + tvec[k,:] + x[b,j,0:6] @ Wx[k]
 * then the modulated adjacency mix, bias, BatchNorm(eval), ReLU.
 *   h_img, h_oth [B,2,hid]; vis [B,24] uint8; x [B,144]; Wx [2,6,hid]; tvec [2,hid];
 *   out [rows_pad,hid], rows = passes*B*24 ((B + num_masked)*24 under ehm_gcn_set_pass_map). */
/* What the second pass of diffuse_fuse masks (EgoHMR.mask_cond(force_mask=True), egohmr.py:150-158): 0 (default) = the image features only
 * (only_mask_img_cond=True, the shipped test configuration) -> pre_k of p = 1 drops h_img; 1 = the whole condition (only_mask_img_cond=False)
 * -> it drops h_img and h_oth. */
int ehm_gcn_set_uncond_mode(ehm_gcn* h, int masks_whole_condition);
/* Exact pass pruning (from egohmr.py:239-254): an item whose 24 joints are ALL visible takes every output entry from the
 * conditional pass, so its second pass is dead work.  mask_items [num_masked] int32 = the items that still need it (ascending),
 * mask_slot [B] int32 = each item's index in that list or -1; DEVICE arrays that must stay alive while the map is set.  From then
 * on, with passes == 2, the activation matrices hold (B + num_masked) * 24 rows: rows [0, B*24) the conditional pass, then the second
 * pass of mask_items[0], mask_items[1], ...  num_masked < 0 clears the map (every item has a second pass: B * 2 * 24 rows). */
int ehm_gcn_set_pass_map(ehm_gcn* h, const int32_t* mask_items, const int32_t* mask_slot, int num_masked);
/* The optional non-local block behind the last hidden conv (ModulatedGCN(nonlocal_layer=True), modulated_gcn.py:104-110;
 * nets/non_local_embedded_gaussian.py:60-85) for the ONE-CALL sampling loop (ehm_sample_loop): two 1x1-conv GEMMs in the operand format
 * of ehm_conv_nhwc_split - Wqkv = [theta | phi | g] as [3*Ci (padded to 128), hid], Wo = W.0 with BatchNorm(eval) folded as [hid, Ci] -
 * with their biases and power-of-two scales, and ehm_nonlocal_attention between them.  The arrays must stay alive while set;
 * NULL removes the block.  ehm_sample_desc.nonlocal_ci must carry Ci (it sizes the workspace). */
typedef struct {
  const void* Wqkv; const float* bqkv; float qkv_scale;
  const void* Wo; const float* bo; float o_scale;
  int Ci;
} ehm_nonlocal_params;
int ehm_gcn_set_nonlocal(ehm_gcn* h, const ehm_nonlocal_params* p);
int ehm_gcn_input_layer(ehm_gcn* h, const float* h_img, const float* h_oth, const uint8_t* vis, const float* x,
                        const float* Wx, const float* tvec, float* out, int B, int passes, void* stream);
/* The same conv in its general form (modulated_gcn_conv.py:39-50 + the BatchNorm / ReLU of modulated_gcn.py:21-28) for an ARBITRARY input feature -
 * ModulatedGCN.forward called on its own (modulated_gcn.py:99-116), where nothing is hoisted: the caller hands over the two GEMM results
 *   pre [bodies*24][2][hid] float32,  pre[r][k][:] = x[r, :] @ W[k]      (one ehm_conv_nhwc_split call with H = W = 1 on the weights [W[0] | W[1]])
 * and this call applies the modulation M, the adjacency mix, bias, BatchNorm(eval), ReLU and writes out [rows_pad, hid] in the handle's activation format. */
int ehm_gcn_input_layer_rows(ehm_gcn* h, const float* pre, float* out, int bodies, void* stream);

/* _GraphConv hid->hid (modulated_gcn.py:21-28): out = ReLU(BN(mix(X W0, X W1))) [+ residual,
 * modulated_gcn.py:42].  X, out, residual [rows_pad,hid]; residual may be NULL. */
int ehm_gcn_hidden_layer(ehm_gcn* h, int layer, const float* X, const float* residual, float* out,
                         int64_t rows_pad, void* stream);

/* All hidden convs of one ModulatedGCN.forward (the `for i in range(self.num_layers): out = self.gconv_layers[i](out)` loop,
 * modulated_gcn.py:108-109; each _ResGraphConv = two _GraphConv + residual, :31-43).  bufs[0] holds the input conv's output,
 * bufs[1] and bufs[2] are scratch of the same size; *result_index says which buffer holds the result (0 or 2).  With the
 * f16 operands (modes 1, 2) this is ONE chained launch (per-row-tile counters instead of kernel boundaries); otherwise it
 * loops over ehm_gcn_hidden_layer with the same buffer rotation.  Env EHM_F16_CHAIN=0 forces the loop. */
int ehm_gcn_hidden_stack(ehm_gcn* h, float* const bufs[3], int64_t rows_pad, int* result_index, void* stream);
/* Synchronises the stream and reports (and clears) the handle's status word: whether ANY chained launch since the previous status call flagged a
 * timed-out producer wait or an unproduced tile (never expected; the kernel gives up instead of hanging the device, and audits its own completion):
 * -5 (EIO), the results computed since the previous call are invalid; or whether an activation of the input / hidden convs reached the f16 range
 * (|x| >= 65504) in an X2 / f16 store and was clamped: -34 (ERANGE) - finite results that are not parity grade; that checkpoint needs
 * ehm_gcn_set_precision(h, 0).  0 = fine.  (The reference's float32 activations have no such limit: modulated_gcn.py:99-116.)  The guard covers the
 * rows the handle's last ehm_gcn_input_layer / _rows / ehm_gcn_pack_activations_checked call produced; the tile padding behind them is don't-care. */
int ehm_gcn_stack_status(ehm_gcn* h, void* stream);
/* The same word copied to *host_flag (pinned host memory owned by the caller) in stream order WITHOUT synchronising: a pipeline that
 * keeps batches in flight looks at *host_flag once the stream has passed this point (an event), and calls ehm_gcn_stack_status to
 * report and clear when it is non-zero (bit 0: chain failure, bit 2: saturation).  The flag is sticky on the device, so nothing is lost by looking late. */
int ehm_gcn_stack_status_async(ehm_gcn* h, uint32_t* host_flag, void* stream);

/* gconv_output (modulated_gcn.py:113) + the visibility fuse of egohmr.py:247-256:
 *   x0[b, j*6+c] = vis[b,j] ? out_cond[b,j,c] : out_uncond[b,j,c]      (passes == 2)
 *   x0 = out_cond                                                      (passes == 1)
 * X [rows_pad,hid] -> x0 [B,144]. */
int ehm_gcn_output_layer(ehm_gcn* h, const float* X, const uint8_t* vis, float* x0, int B, int passes,
                         void* stream);

/* ------------------------------------------------------------------ scene PointNet (conditioning) ----------- */
/* Building blocks of ResnetPointnet.forward (models/respointnet.py:33-59, ResnetBlockFC :89-97) on the split-f16
 * matrix-core path (f32-grade, see ehm_gcn_set_precision mode 1).  All activation / weight operands are in the "X2"
 * split format (ehm_split_pack); the host side (egohmr_amd/encoders.py) chains them, see csrc/linear.hip. */
typedef struct {
  const void* A0;          /* X2 [M, K0]                                                           */
  const void* A1;          /* X2 [M, K1] or NULL: second K segment (dual-source A operand)          */
  const void* W;           /* X2 [N, K0+K1], values pre-multiplied by w_scale (ehm_split_pack)      */
  const float* bias;       /* [N] or NULL                                                           */
  const float* group_bias; /* [M/rows_per_group, N] or NULL: per-body bias (pooled half of a block) */
  void* Y;                 /* X2 [M, N] or NULL                                                     */
  float* colmax;           /* [M/rows_per_group, N] or NULL: running max over the group's valid rows
                              (caller initialises to -inf); fused torch.max(dim=1), respointnet.py:38 */
  int64_t M;               /* rows, multiple of 192 (the row tile)                                  */
  int N, K0, K1;           /* N multiple of 128; K0, K1 multiples of 32, K0 + K1 >= 64              */
  int rows_per_group;      /* padded points per body, multiple of 192                               */
  int valid_rows_per_group;/* real points per body (<= rows_per_group); 0 = all                     */
  int relu_in0;            /* apply ReLU to A0 on load (ResnetBlockFC's actvn before fc_0); needs K1 == 0 */
  int relu_out;            /* apply ReLU before storing                                             */
  float w_scale;           /* the power-of-two scale baked into W                                   */
  const float* lift_points;/* NULL, or [M/rows_per_group, valid_rows_per_group, 3] float32: A0 is not read but generated on
                              the fly as relu(points . Wpos^T + bpos), K0 wide - fc_pos + the first block's actvn
                              (respointnet.py:35,:90) folded into the loader; then A0 == NULL, K1 == 0, relu_in0 == 0 */
  const float* lift_W4;    /* [K0][4] float32 rows (w_x, w_y, w_z, bias) of fc_pos                   */
  int hi_only;             /* 1 = the plain-f16 tier (NOT parity grade): hi halves of A0 / A1 / W only, hi halves of Y only; 0 = split-f16 (f32 grade) */
  int group_bias_stride;   /* floats between the rows of group_bias; 0 = N (dense).  Lets two blocks' per-body vectors come out of one small GEMM */
} ehm_linear_desc;
int ehm_linear_split(const ehm_linear_desc* d, void* stream);
/* float32 [rows,K] -> X2 [rows,K_padded] (zero padded), values multiplied by `scale` (1 for activations). */
int ehm_split_pack(const float* X, void* X2, int64_t rows, int K, int K_padded, float scale, void* stream);
/* The raw points zero-padded to 32 columns, P32 X2 [B*N_padded, 32] (input of the folded block_0 shortcut), and - when R0 is not
 * NULL - fc_pos + ReLU (respointnet.py:35,:90): pts [B,N,3] -> R0 = relu(pts W^T + b) as X2 [B*N_padded, C].  (The PointNet
 * generates R0 inside its first GEMM instead: ehm_linear_desc.lift_points; R0 here serves tests and other callers.) */
int ehm_pointnet_lift(const float* pts, const float* Wpos, const float* bpos, void* R0, void* P32, int B, int N, int N_padded,
                      int C, void* stream);

/* Y[M,N] = act(X[M,K] . W[K,N] + bias[N]) in exact float32 on the matrix cores, for short M (batches of feature vectors): the
 * step-invariant slices of the input graph conv (models/egohmr/modulated_gcn/modulated_gcn_conv.py:39-50 on the image / scene
 * features) and the beta head's first layer (models/egohmr/egohmr.py:263-265, fc_head_beta).  K % 32 == 0, N % 32 == 0, any M;
 * X 16-byte aligned; bias may be NULL.  relu: bit 0 applies max(., 0) to the output; relu >> 1 = number of leading output columns (a multiple of 32)
 * whose INPUT row is rectified first - [relu(x) . Wa | x . Wb] in one launch (the pooled halves of a ResnetBlockFC's fc_0 and shortcut,
 * models/respointnet.py:41-51).  Deterministic (no atomics). */
int ehm_skinny_gemm_f32(const float* X, const float* W, const float* bias, float* Y, int M, int K, int N, int relu, void* stream);


/* ResNet-50 stem in one pass (torchvision ResNet.forward conv1 / bn1 / relu / maxpool, models/resnet.py:139-150 as used at
 * models/egohmr/egohmr.py:183): conv 7x7 stride 2 pad 3 (3 -> 64) + bias + ReLU + max-pool 3x3 stride 2 pad 1.
 *   img [N,3,H,W] float32 (NCHW), H % 32 == 0, W % 32 == 0;   y [N,H/4,W/4,64] float32 (NHWC)
 *   Wt [147][64]: the BatchNorm-folded weights transposed, row k = (ci*7 + kh)*7 + kw;   bias [64]
 *   scratch: ehm_resnet_stem_scratch_bytes(N,H,W) bytes of device memory (zero-padded copy of the image)
 *   out_x2 != 0: y is written in the X2 split format instead (same byte size; the input format of ehm_conv_x2) */
size_t ehm_resnet_stem_scratch_bytes(int N, int H, int W);
int ehm_resnet_stem(const float* img, const float* Wt, const float* bias, float* scratch, float* y, int N, int H, int W,
                    int out_x2, void* stream);

/* NHWC convolution + bias (+ identity) + ReLU as an implicit GEMM on the f16 matrix cores with f32-grade accuracy (hi/lo split
 * operands, f32 accumulate): y[n,ho,wo,co] = act(sum x[n, ho*s-p+kh, wo*s-p+kw, ci] w[co,kh,kw,ci] + bias[co] (+ residual)).
 * Stands in for torchvision's Bottleneck conv1/conv2/conv3/downsample + BatchNorm(eval, folded by the caller) + ReLU of the
 * ResNet-50 backbone (models/egohmr/egohmr.py:183).
 *   x [N,H,W,Ci] float32, Ci % 32 == 0;   y [N,Ho,Wo,Co] float32, Co % 8 == 0;   residual NULL or like y;   bias NULL or [Co]
 *   W: the weights as [Co_pad][KH*KW*Ci] (tap-major: k = (kh*KW + kw)*Ci + ci), Co_pad = Co rounded up to 128 with zero rows,
 *      multiplied by w_scale (a power of two) and packed with ehm_split_pack(..., K, K, w_scale, ...);  KH*KW <= 32 */
typedef struct ehm_conv_desc {
  const float* x; const void* W; const float* bias; const float* residual; float* y;
  int N, H, Wd, Ci, Co;
  int KH, KW, stride, pad, relu;
  float w_scale;
} ehm_conv_desc;
int ehm_conv_nhwc_split(const ehm_conv_desc* d, void* stream);

/* The same convolution with the activations kept in the X2 split format between the layers (the ResNet-50 trunk of
 * models/resnet.py:139-150 / models/egohmr/egohmr.py:183 end to end): x, residual, y are X2 [rows, C] matrices (ehm_split_pack layout,
 * pixel-major NHWC) whose row count is ehm_conv_x2_rows(N*H*W) = pixels rounded up to the 192-row tile + ONE extra row; the LAST
 * row of x must be all zero (out-of-image taps read it).  Ci % 32 == 0, Co % 32 == 0, KH*KW*Ci >= 64; W, bias, relu, w_scale as
 * in ehm_conv_desc.  y's padding rows receive don't-care values; its last row is cleared by the call. */
typedef struct ehm_conv_x2_desc {
  const void* x; int64_t x_rows; const void* W; const float* bias; const void* residual; void* y;
  int N, H, Wd, Ci, Co;
  int KH, KW, stride, pad, relu;
  float w_scale;
  void* workspace;           /* ehm_conv_x2_workspace_bytes(d) bytes of device scratch, or NULL (then whole tiles only) */
  int64_t workspace_bytes;
  /* Optional second K segment (x2 != NULL): y = act( conv(x) + conv1x1_stride2(x2) + bias (+ residual) ) - a bottleneck's projection
   * shortcut accumulated inside its last convolution (torchvision Bottleneck.forward: out = bn3(conv3(.)) + downsample(x)), so that the
   * shortcut tensor never exists.  x2 is X2 [x2_rows, Ci2] over [N, H2, W2] pixels with its own zero row, (H2 - 1) / stride2 + 1 == Ho
   * (same for W); W then holds [Co_pad][KH*KW*Ci + Ci2] with the shortcut's weights behind the main ones (one common w_scale), bias
   * the SUM of the two folded biases; Co % 128 == 0. */
  const void* x2; int64_t x2_rows;
  int H2, W2, Ci2, stride2;
  int hi_only;               /* 0 = split-f16 arithmetic (f32 grade, the parity path).  1 = the plain-f16 tier (NOT parity grade): only the hi halves of x, W,
                                residual are read and only hi halves written - one MFMA per product, half the bytes; the lo halves of y are don't-care  */
  int workspace_clean;       /* 1 = the caller zeroed the first 4096 bytes of `workspace` once and nothing but ehm_conv_x2 has written to it since: the call
                                skips its memset node (every call leaves the arrival counters zeroed - or poisoned, after a hand-off time-out: then every later
                                stream-K conv on that workspace comes out as NaN until the caller zeroes it again).  0 = the call clears them itself */
} ehm_conv_x2_desc;
int64_t ehm_conv_x2_rows(int64_t pixels);
/* Scratch of the stream-K schedule: when a conv's tile count would leave a large share of the GPU's block slots idle in its last round
 * (ResNet-50 layers 3 / 4 at B = 256: 524 or 264 tiles on 512 slots), the tiles' K loops are dealt out to the blocks in equal runs and a tile
 * cut by a run boundary is finished by the block that started it (fixed summation order: deterministic).  0 = the conv runs whole tiles. */
int64_t ehm_conv_x2_workspace_bytes(const ehm_conv_x2_desc* d);
int ehm_conv_x2(const ehm_conv_x2_desc* d, void* stream);
/* Did a stream-K tile hand-off on `workspace` time out since the last call (a partner block did not deliver its partial sums within ~1 s: GPU shared,
 * preempted, under a profiler)?  Such a tile is written as NaN (through the ReLU as well), its arrival counter stays poisoned so that every later
 * stream-K conv on the workspace is NaN too, and the LAST word of the workspace's first 4096 bytes counts the time-outs (sticky on the device).
 *   host_flag != NULL : enqueue a stream-ordered copy of that count to *host_flag (pinned host memory) and return 0 - the asynchronous form;
 *   host_flag == NULL : wait for the stream, and when the count is non-zero zero the 4096 counter bytes again (the workspace is clean) and return -5 (EIO).
 * The torchvision convolution this replaces cannot fail (models/resnet.py:139-150); this is the hand-off protocol's own failure mode made loud. */
int ehm_conv_x2_workspace_status(void* workspace, uint32_t* host_flag, void* stream);
/* Y[g, c] = mean over the rows_per_group consecutive rows of group g of the X2 matrix X [groups*rows_per_group (+ padding), C]:
 * the global average pool behind the last bottleneck (models/resnet.py:148-149). */
int ehm_x2_group_mean(const void* X, float* Y, int groups, int rows_per_group, int C, int hi_only /* 1: X was written by a hi_only call */, void* stream);

/* Attention core of the optional non-local block of ModulatedGCN (nonlocal_layer=True, modulated_gcn.py:93-110;
 * nets/non_local_embedded_gaussian.py:68-79): per body, y = softmax(theta phi^T, dim=-1) g over the 24 joints.
 *   qkv [bodies*24, 3*Ci] float32 rows = [theta | phi | g] (one 1x1-conv GEMM, e.g. ehm_conv_nhwc_split with H = W = 1);
 *   y [bodies*24, Ci].  The W conv + BatchNorm + residual that follow are another ehm_conv_nhwc_split call. */
int ehm_nonlocal_attention(const float* qkv, float* y, int64_t bodies, int Ci, void* stream);

/* ------------------------------------------------------------------ sampler steps ------------- */
/* diffusion/gaussian_diffusion.py:217-220 + :333-336 (p_sample) and :378-385 (p_sample_with_grad):
 *   mean = coef1*x0 + coef2*x  [+ grad_scale * grad]
 *   x_next = mean + nonzero * exp(0.5*log_variance) * noise
 * grad may be NULL.  All [B,144]; x_next may alias x. */
int ehm_ddpm_step(const float* x, const float* x0, const float* noise, const float* grad, float* x_next,
                  float coef1, float coef2, float log_variance, float nonzero, float grad_scale, int64_t n,
                  void* stream);
/* gaussian_diffusion.py:286-290 + :539-555 (ddim_sample):
 *   eps = (sqrt_recip_ac*x - x0) / sqrt_recipm1_ac
 *   x_next = x0*sqrt_ac_prev + dir_coef*eps + nonzero*sigma*noise,  dir_coef = sqrt(1-ac_prev-sigma^2) */
int ehm_ddim_step(const float* x, const float* x0, const float* noise, float* x_next, float sqrt_recip_ac,
                  float sqrt_recipm1_ac, float sqrt_ac_prev, float dir_coef, float sigma, float nonzero,
                  int64_t n, void* stream);

/* ------------------------------------------------------------------ collision guidance -------- */
/* Build-defined proxy for smpl.coap.collision_loss (egohmr.py:555; COAP is unavailable offline,
 * see DESIGN.md): with the bbox selection of egohmr.py:550-552,
 *   loss[b] = sum over scene points p inside bbox(verts[b]) of relu(tau - min_v |p - v|)^2
 * and gverts[b,v,:] = d(loss[b])/d(verts[b,v,:]).  scene [B,N,3] (already canonicalised,
 * egohmr.py:211); loss [B]; gverts [B,V,3] (overwritten). */
int ehm_collision_proxy(const float* verts, const float* scene, float* loss, float* gverts, int B, int V, int N,
                        float tau, void* stream);

/* The same proxy with the knobs of the reference's other call sites:
 *   all_points != 0   no bbox selection - the batched `volume.collision_loss(scene, smpl_output)` of the VolSMPL twin
 *                     (models/egohmr/egohmr_volsmpl.py:609-612); 0 = egohmr.py:550-552 as above
 *   hits [B] int32    (may be NULL) number of selected scene points closer than tau to the body: the count behind
 *                     EgoHMR.eval_coll (egohmr.py:487-514, `occupancy > 0.5`) / eval_coll_volsmpl (egohmr_volsmpl.py:548-579, `sdf < 0`)
 *   gverts            may be NULL (metric only) */
int ehm_collision_query(const float* verts, const float* scene, float* loss, float* gverts, int32_t* hits, int B, int V, int N,
                        float tau, int all_points, void* stream);

/* VJP of ehm_smpl_forward_rot6d w.r.t. the DE-NORMALISED 6-D pose (the quirk of egohmr.py:523-528:
 * autograd.grad is taken w.r.t. x_t*std+mean): gverts [B,V,3] -> gpose6d [B,144].
 * Needs the forward's x/mean/std/betas again (recomputes the chain). */
int ehm_smpl_backward_rot6d(ehm_smpl* h, const float* betas, const float* x, const float* mean, const float* std,
                            const float* gverts, float* gpose6d, int B, void* stream);

/* egohmr.py:561-570: g = -(1/denom) * gpose6d (denom = B for loss.mean(), 1 for loss.sum()),
 * joints 3..23 scaled by 2, joints {0,3,6,9,12..23} zeroed; all-zero loss -> zeros. */
int ehm_guidance_grad_finish(const float* gpose6d, const float* loss, float* grad, int B, float denom, void* stream);

/* ------------------------------------------------------------------ post-loop metric ---------- */
/* pytorch3d.ops.knn_points(x, y, K=1) as used by utils/pytorch3d_chamfer_distance.py:152-156 (contact score,
 * test_egohmr.py:496-505): for every x[b,i] the SQUARED distance to its nearest y[b,:] and (optionally) its index.
 * x [B,P1,3], y [B,P2,3] -> dist2 [B,P1], idx [B,P1] int32 or NULL. */
int ehm_nn_dist2(const float* x, const float* y, float* dist2, int32_t* idx, int B, int P1, int P2, void* stream);

/* ------------------------------------------------------------------ evaluation block ---------- */
/* test_egohmr.py:399-449: Euclidean error per point of S samples per item against ONE ground truth per item, its mean over the points and its sums over
 * the visible / invisible points (G-MPJPE :399-407: joints in the camera frame; MPJPE :409-417 and V2V :441-449: pelvis-aligned).
 *   pred [B,S,pred_points,3] and gt [B,gt_points,3]: the first P points of each are compared (SMPL hands over 45 joints, the metrics use 24);
 *   pred_origin [B,S,3] / gt_origin [B,3]: subtracted first (NULL = nothing: the caller aligned already, or G-MPJPE); or origin_point >= 0: each cloud's own
 *     point of that index is its origin (joint 0 = the pelvis, :409) and both pointers are NULL; origin_point < 0: pointers only;
 *   mask [B,P] (NULL: everything visible);  per_point [B,S,P] (may be NULL);  mean [B,S];  vis_sum, invis_sum [B,S] (may be NULL). */
typedef struct ehm_eval_points_desc {
  const float* pred; const float* gt; const float* pred_origin; const float* gt_origin; const uint8_t* mask;
  float* per_point; float* mean; float* vis_sum; float* invis_sum;
  int B, S, P, pred_points, gt_points, origin_point;
} ehm_eval_points_desc;
int ehm_eval_point_errors(const ehm_eval_points_desc* d, void* stream);
/* utils/pose_utils.py:10-66 (compute_similarity_transform) + :109-126 (reconstruction_error) as called at test_egohmr.py:419-437: the similarity transform
 * (scale, rotation, translation) of each pred[b,s] [J,3] closest to gt[b] [J,3] - float64 in registers like the reference's per-sample numpy loop, the
 * 3 x 3 SVD by one-sided Jacobi - then the per-joint error.  aligned [B,S,J,3], per_joint [B,S,J], mean [B,S], vis_sum / invis_sum [B,S] with mask [B,J]:
 * each may be NULL (one of aligned / per_joint / mean must not be).  3 <= J <= 32. */
int ehm_eval_procrustes(const float* pred, const float* gt, const uint8_t* mask, float* aligned, float* per_joint, float* mean, float* vis_sum,
                        float* invis_sum, int B, int S, int J, void* stream);
/* test_egohmr.py:453-494: sample diversity of joints [B,S,J,3] over the joints selected by mask [B,J] (NULL: all; invert != 0: the unselected ones) -
 * std_out [B] = mean over the selected joints and the 3 coordinates of the unbiased std over the samples; apd_out [B] = sum over ordered sample pairs and
 * selected joints of the joint distance / n_selected / S / (S - 1) / 2 (the reference's normalisation).  No selected joint -> NaN, as the reference. */
int ehm_eval_diversity(const float* joints, const uint8_t* mask, int invert, float* std_out, float* apd_out, int B, int S, int J, void* stream);

/* ------------------------------------------------------------------ per-item scalars ---------- */
/* The per-item scalar work of EgoHMR.forward in front of the encoders, in two launches:
 *   vis [B,24] u8       models/egohmr/egohmr.py:186-188: confidence > 0, OpenPose joint `force_visible` (8) always on, gathered by joint_map
 *   mask_slot [B], mask_items [count], count [1] (int32)   which items need the image-masked second pass (:249-254: any invisible
 *                       joint), as ehm_gcn_set_pass_map wants them: slot of the item among those that do (ascending) or -1, and the
 *                       items in slot order.  pass_group > 1 rounds the need up to groups of pass_group consecutive items.
 *   other[:, other_col0 ...] = TranslEnc(transl) (:217, Linear(3,t_hidden) -> ReLU -> Linear(t_hidden,t_out), torch [out,in] weights)
 *                       | camera features (:195-205: [cx, cy] / (fx*fx_norm), [box_center, box_size] / (fx*fx_norm), fx), zero-filled
 *                       up to other_ld (row stride of `other`; columns [0, other_col0) are the caller's: the scene features)
 *   finite [B] u8       0 when transl / fx / cx / cy / box / img_rowsum / scene_rowsum of the item hold a NaN or Inf (the float32
 *                       graph of the reference turns every output of such an item into NaN; see ehm_pack_outputs)
 * need_scratch: B bytes of device memory. */
typedef struct ehm_item_prep_desc {
  const float* keypoints_2d; /* [B, NK, 3] */
  const int32_t* joint_map;  /* [24] device */
  int NK, force_visible;
  const float *fx, *cx, *cy, *box_center, *box_size, *transl; /* [B], [B], [B], [B,2], [B], [B,3]; cx/cy/box may be NULL when unused */
  float fx_norm;
  int with_bbox, with_cam_center;
  const float *tW1, *tb1, *tW2, *tb2;
  int t_hidden, t_out;
  const float *img_rowsum, *scene_rowsum; /* [B] or NULL */
  float* other;
  int other_ld, other_col0;
  uint8_t* vis;
  int32_t *mask_slot, *mask_items, *count;
  uint8_t* finite;
  uint8_t* need_scratch;
  int pass_group;
  int B;
} ehm_item_prep_desc;
int ehm_item_prep(const ehm_item_prep_desc* d, void* stream);

/* The output garnish of EgoHMR.forward (egohmr.py:283-301) in one launch, in place on the loop's outputs:
 *   bad item = !finite[b] or a non-finite value in chk[0..chk_rows)[b] (x_T and the draws that feed a denoiser evaluation, or the
 *   x_t of a single forward): x0, pose6d, R, verts, joints, betas of a bad item become NaN; x_final (may be NULL) additionally takes
 *   NaN where last_noise (the last step's draw, multiplied by nonzero_mask = 0: gaussian_diffusion.py:357-359) is not finite.
 *   global_orient [B,1,3,3] / body_pose [B,23,3,3] = the split of R; focal [B,2] = fx*fx_norm; center [B,2]; kp3d_full [B,J,3] =
 *   joints + transl; kp2d_full [B,J,2] = utils/geometry.py:78-116 perspective_projection, then x / 1920 - 0.5, y / 1080 - 0.5. */
typedef struct ehm_pack_desc {
  int B, J, V;
  const uint8_t* finite; /* [B] or NULL */
  const float* chk;      /* [chk_rows, B, 144] or NULL */
  int chk_rows;
  const float* last_noise; /* [B,144] or NULL */
  float* x_final;          /* [B,144] or NULL */
  float *x0, *pose6d, *R, *verts, *joints;
  const float* betas_in;
  float* betas_out;
  const float *transl, *fx, *cx, *cy;
  float fx_norm;
  float *global_orient, *body_pose, *kp3d_full, *kp2d_full, *focal, *center;
  uint8_t* finite_out; /* [B] or NULL */
} ehm_pack_desc;
int ehm_pack_outputs(const ehm_pack_desc* d, void* stream);

/* ------------------------------------------------------------------ whole sampling loop ------- */
/* One executed step of the loop (host-side table lookup already done, float32 like
 * _extract_into_tensor, gaussian_diffusion.py:794). */
typedef struct {
  float coef1, coef2, log_variance, variance; /* DDPM */
  float sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_prev, dir_coef, sigma; /* DDIM */
  float nonzero;   /* 0 at respaced index 0 */
  float grad_scale; /* 0 = unguided step; p_sample_with_grad: cond_grad_weight*variance or cond_grad_weight*0.01 (gaussian_diffusion.py:378-385);
                       ddim_sample_with_grad (ddim = 1): float32 sqrt(1 - alpha_bar) on the last four respaced steps (:580-592, scale 1.0) */
} ehm_step_coefs;

typedef struct {
  int B;              /* bodies                                                     */
  int passes;         /* 2 = diffuse_fuse (conditional + image-masked pass), 1      */
  int num_steps;      /* executed steps T                                           */
  int ddim;           /* 0 = p_sample loop, 1 = ddim_sample loop (eta = 0)          */
  int lbs_every_step; /* 1 = decode the body every step like EgoHMR.forward does    */
  int num_scene_points; /* N (guidance only)                                        */
  float guide_denom;  /* B for COAP-style loss.mean(), 1 for VolSMPL-style sum()    */
  float tau;          /* collision proxy contact distance                          */
  int num_masked;     /* second passes after pruning = what ehm_gcn_set_pass_map was given; -1 = no map (B)   */
  int guide_all_points; /* 1 = VolSMPL-style guidance over ALL scene points (egohmr_volsmpl.py:609-612), 0 = bbox-selected (egohmr.py:550-552) */
  int lowprec_steps;  /* precision schedule: the FIRST lowprec_steps executed steps run the hidden convs on plain f16 operands
                         (ehm_gcn_set_precision mode 2), the remaining ones in the handle's mode; 0 = off.  docs/EXPERIMENTS.md 3.6  */
  int nonlocal_ci;    /* inter_channels of the non-local block set with ehm_gcn_set_nonlocal, 0 = none (needs float32 features:
                         handle mode 0 or 1 and lowprec_steps == 0)                                                              */
  int per_step_launches; /* 0 (default) = two launches per step: chained hidden convs, then step_fused_kernel (output responses + per-body
                         update + the next step's input conv); 1 = the separate launches of rounds 2-3 (input conv, chain, responses, per-body
                         step) - same bits, kept for A/B runs and the bit-equality tests                                                      */
} ehm_sample_desc;

/* GaussianDiffusion.p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:391-508, :618-718)
 * around EgoHMR.forward's per-step part (egohmr.py:232-278), everything step-invariant hoisted:
 *   steps [num_steps] HOST; tvecs [num_steps,2,hid] (timestep-embedding projection per step);
 *   noise [num_steps+1,B,144] (row 0 = x_T); scene [B,N,3] or NULL; betas [B,10];
 *   outputs: x_final [B,144], x0_final [B,144] (pred_x_start), verts [B,V,3], joints [B,45,3],
 *   R [B,24,3,3], pose6d [B,144] of the LAST step; trace (may be NULL) [num_steps,B,144] = x_t fed
 *   to each step.  workspace from ehm_sample_workspace_bytes(). */
int64_t ehm_sample_workspace_bytes(const ehm_sample_desc* d, int hid_dim, int num_verts);
int ehm_sample_loop(ehm_gcn* gcn, ehm_smpl* smpl, const ehm_sample_desc* d, const ehm_step_coefs* steps,
                    const float* h_img, const float* h_oth, const uint8_t* vis, const float* Wx, const float* tvecs,
                    const float* noise, const float* scene, const float* betas, const float* mean, const float* std,
                    float* x_final, float* x0_final, float* verts, float* joints, float* R, float* pose6d,
                    float* trace, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- measurement aid (bench.py's roofline objects; SURVEY.md 8d asks for per-kernel figures).  No reference counterpart.
 * Between ehm_profile_begin() and ehm_profile_end() every launch of ehm_sample_loop is bracketed by a pair of HIP events ON THE
 * LAUNCH STREAM, tagged with its class; ehm_profile_end waits for them and adds the elapsed times up per class
 * (ms[c] = sum of (stop - start), launches[c] = pairs), c < n <= EHM_PROF_N.  The event pairs serialise nothing that is not
 * already serial (the loop is one dependency chain), but they cost a few microseconds each: never inside a timed region.
 * Not thread-safe, one profile at a time. */
enum {
  EHM_PROF_INPUT = 0,        /* gcn_input_kernel (only the first step when the skinning launch carries it)              */
  EHM_PROF_CHAIN_F16X3 = 1,  /* gcn_hidden_chain_kernel<3, 4>: the 8 hidden convs of a step, split-f16                   */
  EHM_PROF_CHAIN_F16 = 2,    /* gcn_hidden_chain_kernel<1, 8>: the same on plain f16 operands                            */
  EHM_PROF_HIDDEN_F32 = 3,   /* gcn_hidden_kernel x 8 (f32-input MFMA) or per-conv tile launches                         */
  EHM_PROF_OUT_DOT = 4,      /* gcn_out_dot_kernel                                                                       */
  EHM_PROF_STEP_BODY = 5,    /* step_body_kernel; with the fused step launches: pose_steps_kernel (the pending steps' poses, per flush) */
  EHM_PROF_SKIN_INPUT = 6,   /* skin_input_kernel (skinning of step t + input conv of step t+1) / skin_mfma_kernel       */
  EHM_PROF_GUIDANCE = 7,     /* the collision-guidance kernel sequence of a guided step                                  */
  EHM_PROF_RESERVED_8 = 8,   /* (was the one-launch loop experiment's class; kept so that the indices behind it do not move)           */
  EHM_PROF_RESERVED_9 = 9,
  EHM_PROF_G_NEAREST = 10,   /* inside EHM_PROF_GUIDANCE: bbox + select + nearest_grid_kernel (the collision proxy's search) */
  EHM_PROF_G_SKIN_BWD = 11,  /* inside EHM_PROF_GUIDANCE: skin_bwd_kernel (VJP of the skinning)                          */
  EHM_PROF_G_POSEFEAT_BWD = 12, /* inside EHM_PROF_GUIDANCE: posefeat_bwd_kernel ([B, 20670] x [20670, 207] contraction) */
  EHM_PROF_STEP_FUSED = 13,  /* step_fused_kernel: a step's output responses + per-body update + the NEXT step's input conv, one block per body */
  EHM_PROF_G_NEAREST_EVALS = 14, /* not a launch class: launches[14] = point-to-vertex distance evaluations of nearest_grid_kernel while the profile was open (ms[14] = 0) */
  EHM_PROF_N = 15
};
int ehm_profile_begin(void);
int ehm_profile_end(double* ms, int64_t* launches, int n);

#ifdef __cplusplus
}
#endif
#endif /* EGOHMR_HIP_H */
