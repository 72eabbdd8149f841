"""Seeded synthetic stand-ins for assets that cannot ship with this repository.

Nothing here is a reference algorithm: it only manufactures *inputs* of the right shape so
the hot path can be exercised, parity-tested and benchmarked without the licensed SMPL model
files, the EgoBody dataset or trained checkpoints (none of which exist offline):

* :func:`make_smpl_asset`   - an SMPL-shaped body model (6890 verts, 24 joints, 10 betas,
  207 pose-corrective directions, <=4 non-zero skinning weights per vertex), tensor names
  as registered by ``smplx.SMPL`` (v_template, shapedirs, posedirs, J_regressor, lbs_weights,
  parents, faces) - the module built at ``models/egohmr/egohmr.py:105``.
* :func:`egohmr_manifest` / :func:`make_state_dict` - every parameter/buffer of the stage-2
  model under the reference's ``state_dict`` names (``models/egohmr/egohmr.py:58-102``), with
  "trained-like" magnitudes so every layer matters numerically (default torch init makes the
  GCN's GEMMs numerically invisible: xavier on a [2,in,out] tensor gives |W|~3e-3).
* :func:`make_batch` - the batch dict schema ``dataloaders/egobody_dataset.py:241-277`` feeds
  to ``EgoHMR.forward`` (SURVEY.md section 8d lists the distributions).
* :func:`make_noise_stack` - explicit N(0,1) draws in the reference's draw order
  (``diffusion/gaussian_diffusion.py:478,331``): row 0 is x_T, row 1+k is the k-th step's noise.

All generators use numpy ``Generator(PCG64(seed))`` so CPU oracle, golden fixtures and the GPU
path see bit-identical inputs.
"""
from __future__ import annotations

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207  # 23 joints x 9 rotation entries
NUM_FACES = 13776
NUM_EXTRA_JOINTS = 21

# Kinematic tree of SMPL (child -> parent); same tree as utils/other_utils.py:86-108 (SMPL_EDGES).
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int64,
)
SMPL_EDGES = [(int(SMPL_PARENTS[c]), c) for c in range(1, NUM_JOINTS)]

# smplx VertexJointSelector for model_type='smpl' (vertex_ids['smplh']): 5 face, 6 feet,
# 10 finger-tip vertices appended after the 24 kinematic joints -> 45 joints.
EXTRA_JOINT_VERTEX_IDS = np.array(
    [332, 6260, 2800, 4071, 583,                 # nose, reye, leye, rear, lear
     3216, 3226, 3387, 6617, 6624, 6787,         # LBigToe LSmallToe LHeel RBigToe RSmallToe RHeel
     2746, 2319, 2445, 2556, 2673,               # l thumb/index/middle/ring/pinky
     6191, 5782, 5905, 6016, 6133],              # r thumb/index/middle/ring/pinky
    dtype=np.int64,
)

# Rough T-pose joint locations (metres) so the synthetic body has human proportions.
_REST_JOINTS = np.array([
    [0.00, -0.22, 0.03], [0.07, -0.31, 0.02], [-0.07, -0.31, 0.02], [0.00, -0.10, 0.00],
    [0.10, -0.69, 0.02], [-0.10, -0.69, 0.02], [0.00, 0.04, 0.02], [0.09, -1.09, -0.02],
    [-0.09, -1.09, -0.02], [0.00, 0.09, 0.03], [0.11, -1.15, 0.10], [-0.11, -1.15, 0.10],
    [0.00, 0.31, -0.01], [0.08, 0.21, 0.00], [-0.08, 0.21, 0.00], [0.00, 0.39, 0.04],
    [0.17, 0.24, -0.01], [-0.17, 0.24, -0.01], [0.43, 0.23, -0.03], [-0.43, 0.23, -0.03],
    [0.68, 0.24, -0.03], [-0.68, 0.24, -0.03], [0.77, 0.23, -0.04], [-0.77, 0.23, -0.04],
], dtype=np.float64)


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def make_smpl_asset(seed: int = 0) -> dict:
    """SMPL-shaped synthetic body model. Returns float32/int64 numpy arrays (smplx buffer names)."""
    g = _rng(1000 + seed)
    V, J = NUM_VERTS, NUM_JOINTS
    # every vertex hangs on one bone (child joint c, parent p): point on the segment + radial offset
    bone = g.integers(1, J, size=V)
    par = SMPL_PARENTS[bone]
    u = g.random(V)
    centre = _REST_JOINTS[par] * (1 - u[:, None]) + _REST_JOINTS[bone] * u[:, None]
    radius = np.where(np.isin(bone, [3, 6, 9]), 0.13, 0.05)  # torso thicker than limbs
    d = g.normal(size=(V, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    v_template = centre + d * radius[:, None] * (0.6 + 0.4 * g.random((V, 1)))

    # skinning weights: <=4 non-zeros per vertex among {bone, parent, grandparent, a child}
    lbs_weights = np.zeros((V, J))
    child_of = {p: [c for c in range(1, J) if SMPL_PARENTS[c] == p] for p in range(J)}
    for v in range(V):
        c, p = int(bone[v]), int(par[v])
        cand = [c, p]
        gp = int(SMPL_PARENTS[p])
        if gp >= 0:
            cand.append(gp)
        if child_of[c]:
            cand.append(child_of[c][int(g.integers(len(child_of[c])))])
        w = g.random(len(cand)) ** 2 + 1e-3
        w[0] += u[v]
        w[1] += 1 - u[v]
        lbs_weights[v, cand] = w / w.sum()

    # joint regressor: each joint = convex combination of ~40 vertices nearest to its rest position
    J_regressor = np.zeros((J, V))
    for j in range(J):
        dist = np.linalg.norm(v_template - _REST_JOINTS[j], axis=1)
        idx = np.argsort(dist)[:40]
        w = g.random(40) + 0.1
        J_regressor[j, idx] = w / w.sum()

    shapedirs = g.normal(scale=0.012, size=(V, 3, NUM_BETAS))
    posedirs = g.normal(scale=0.004, size=(NUM_POSE_BASIS, V * 3))
    faces = g.integers(0, V, size=(NUM_FACES, 3))
    return {
        "v_template": v_template.astype(np.float32),
        "shapedirs": shapedirs.astype(np.float32),
        "posedirs": posedirs.astype(np.float32),
        "J_regressor": J_regressor.astype(np.float32),
        "lbs_weights": lbs_weights.astype(np.float32),
        "parents": SMPL_PARENTS.copy(),
        "faces": faces.astype(np.int64),
        "extra_joints_idxs": EXTRA_JOINT_VERTEX_IDS.copy(),
    }


# ----------------------------------------------------------------------------------------------
# stage-2 model parameters
# ----------------------------------------------------------------------------------------------

def _resnet50_manifest(prefix: str) -> list:
    """conv/bn names of models/resnet.py:97-136 (Bottleneck [3,4,6,3])."""
    out = []

    def bn(name, c):
        out.extend([(f"{name}.weight", (c,)), (f"{name}.bias", (c,)), (f"{name}.running_mean", (c,)),
                    (f"{name}.running_var", (c,)), (f"{name}.num_batches_tracked", ())])

    out.append((f"{prefix}conv1.weight", (64, 3, 7, 7)))
    bn(f"{prefix}bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for b in range(blocks):
            p = f"{prefix}layer{li}.{b}."
            out.append((p + "conv1.weight", (planes, inplanes, 1, 1)))
            bn(p + "bn1", planes)
            out.append((p + "conv2.weight", (planes, planes, 3, 3)))
            bn(p + "bn2", planes)
            out.append((p + "conv3.weight", (planes * 4, planes, 1, 1)))
            bn(p + "bn3", planes * 4)
            if b == 0 and (stride != 1 or inplanes != planes * 4):
                out.append((p + "downsample.0.weight", (planes * 4, inplanes, 1, 1)))
                bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    return out


def nonlocal_manifest(hid_dim: int = 1024, prefix: str = "diffusion_model.non_local.") -> list:
    """NONLocalBlock2D(in_channels=hid, sub_sample=False, bn_layer=True) of ModulatedGCN(nonlocal_layer=True)
    (modulated_gcn.py:93-94, nets/non_local_embedded_gaussian.py:6-58): g / theta / phi 1x1 convs hid -> hid/2, W = conv hid/2 -> hid
    + BatchNorm2d.  Listed apart from egohmr_manifest (the shipped configs leave the block off) and always drawn AFTER it, so the
    seeded values of every other tensor do not move."""
    ci = max(hid_dim // 2, 1)
    m = []
    for n in ("g", "theta", "phi"):
        m += [(prefix + n + ".weight", (ci, hid_dim, 1, 1)), (prefix + n + ".bias", (ci,))]
    m += [(prefix + "W.0.weight", (hid_dim, ci, 1, 1)), (prefix + "W.0.bias", (hid_dim,)),
          (prefix + "W.1.weight", (hid_dim,)), (prefix + "W.1.bias", (hid_dim,)), (prefix + "W.1.running_mean", (hid_dim,)),
          (prefix + "W.1.running_var", (hid_dim,)), (prefix + "W.1.num_batches_tracked", ())]
    return m


def egohmr_manifest(hid_dim: int = 1024, num_blocks: int = 4, scene_feat_dim: int = 512,
                    img_feat_dim: int = 2048, with_backbone: bool = True, nonlocal_layer: bool = False, cam_dim: int = 6) -> list:
    """(name, shape) of every learnable tensor / buffer of the stage-2 model, reference names.

    Conditioning width = img 2048 + scene 512 + transl 128 + cam (2+3+1) = 2694
    (models/egohmr/egohmr.py:76-83); GCN input = 2694 + 512 (x_t embed) + 512 (timestep) = 3718.
    SMPL / COAP buffers that the reference also keeps in its state_dict are not listed: the build
    loads the body model from its own asset and reference checkpoints are read with strict=False.
    """
    ctx = img_feat_dim + cam_dim + scene_feat_dim + 128      # cam_dim = 1 (fx) + 3 (bbox, with_bbox_info) + 2 (centre, with_cam_center)
    in_dim = ctx + 512 + 512
    m = [
        ("input_process.poseEmbedding.weight", (512, 6)),
        ("input_process.poseEmbedding.bias", (512,)),
        ("sequence_pos_encoder.pe", (5000, 1, 512)),
        ("embed_timestep.sequence_pos_encoder.pe", (5000, 1, 512)),
        ("embed_timestep.time_embed.0.weight", (512, 512)),
        ("embed_timestep.time_embed.0.bias", (512,)),
        ("embed_timestep.time_embed.2.weight", (512, 512)),
        ("embed_timestep.time_embed.2.bias", (512,)),
    ]
    if with_backbone:
        m += _resnet50_manifest("backbone.")
    hd = 256
    m += [("scene_enc.fc_pos_0.weight", (2 * hd, 3)), ("scene_enc.fc_pos_0.bias", (2 * hd,))]
    for b in range(4):
        p = f"scene_enc.block_{b}."
        m += [(p + "fc_0.weight", (hd, 2 * hd)), (p + "fc_0.bias", (hd,)),
              (p + "fc_1.weight", (hd, hd)), (p + "fc_1.bias", (hd,)),
              (p + "shortcut.weight", (hd, 2 * hd))]
    m += [("scene_enc.fc_c.weight", (scene_feat_dim, hd)), ("scene_enc.fc_c.bias", (scene_feat_dim,))]
    m += [("transl_enc.layers.0.weight", (64, 3)), ("transl_enc.layers.0.bias", (64,)),
          ("transl_enc.layers.2.weight", (128, 64)), ("transl_enc.layers.2.bias", (128,))]

    def gconv(name, cin, cout):
        return [(name + ".W", (2, cin, cout)), (name + ".M", (NUM_JOINTS, cout)),
                (name + ".adj2", (NUM_JOINTS, NUM_JOINTS)), (name + ".bias", (cout,))]

    def bn(name, c):
        return [(name + ".weight", (c,)), (name + ".bias", (c,)), (name + ".running_mean", (c,)),
                (name + ".running_var", (c,)), (name + ".num_batches_tracked", ())]

    m += gconv("diffusion_model.gconv_input.0.gconv", in_dim, hid_dim) + bn("diffusion_model.gconv_input.0.bn", hid_dim)
    for b in range(num_blocks):
        for k in (1, 2):
            p = f"diffusion_model.gconv_layers.{b}.gconv{k}"
            m += gconv(p + ".gconv", hid_dim, hid_dim) + bn(p + ".bn", hid_dim)
    m += gconv("diffusion_model.gconv_output", hid_dim, 6)
    m += [("beta_layer.layers.0.weight", (1024, ctx)), ("beta_layer.layers.0.bias", (1024,)),
          ("beta_layer.layers.2.weight", (10, 1024)), ("beta_layer.layers.2.bias", (10,)),
          ("beta_layer.init_betas", (1, 10))]
    if nonlocal_layer:
        m += nonlocal_manifest(hid_dim)
    return m


def positional_table(max_len: int = 5000, d_model: int = 512) -> np.ndarray:
    """sin/cos table of models/egohmr/egohmr.py:614-619, float32 arithmetic like torch's."""
    position = np.arange(max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d_model))
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[:, None, :]


def make_state_dict(seed: int = 0, manifest: list | None = None, **manifest_kw) -> dict:
    """Seeded 'trained-like' float32 weights for every name of :func:`egohmr_manifest`."""
    if manifest is None:
        manifest = egohmr_manifest(**manifest_kw)
    g = _rng(2000 + seed)
    sd = {}
    for name, shape in manifest:
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(100, dtype=np.int64)
        elif leaf == "pe":
            sd[name] = positional_table(shape[0], shape[2])
        elif leaf == "running_mean":
            sd[name] = g.normal(scale=0.1, size=shape).astype(np.float32)
        elif leaf == "running_var":
            sd[name] = g.uniform(0.6, 1.4, size=shape).astype(np.float32)
        elif leaf == "init_betas":
            sd[name] = g.normal(scale=0.5, size=shape).astype(np.float32)
        elif leaf == "adj2":
            sd[name] = g.normal(scale=0.02, size=shape).astype(np.float32)
        elif leaf == "M":
            sd[name] = (1.0 + g.normal(scale=0.15, size=shape)).astype(np.float32)
        elif leaf == "W":  # [2, in, out] - both branches scaled so that the layer roughly preserves rms
            cin = shape[1]
            if shape[2] <= 8:          # output conv: keep the x_0 prediction O(1) and contractive in x_t
                gain = 0.12
            elif cin == shape[2]:      # hidden convs: a residual block grows rms by ~1.15x, not 2x
                gain = 0.55
            else:                      # input conv
                gain = 0.8
            sd[name] = g.normal(scale=gain / np.sqrt(cin), size=shape).astype(np.float32)
        elif leaf == "weight" and len(shape) == 1:  # batch-norm gamma
            last_bn = name.endswith("bn3.weight") or "downsample.1" in name
            lo, hi = (0.3, 0.6) if last_bn else (0.8, 1.2)
            sd[name] = g.uniform(lo, hi, size=shape).astype(np.float32)
        elif leaf == "bias":
            sd[name] = g.normal(scale=0.05, size=shape).astype(np.float32)
        elif leaf == "weight" and len(shape) == 4:  # conv
            fan_in = shape[1] * shape[2] * shape[3]
            scale = np.sqrt(2.0 / fan_in)
            if ".non_local.theta." in name or ".non_local.phi." in name:
                scale *= 0.25          # attention logits O(3) instead of O(50): a trained block is not one-hot, and parity tests stay well-conditioned
            sd[name] = g.normal(scale=scale, size=shape).astype(np.float32)
        elif leaf == "weight" and len(shape) == 2:  # linear [out, in]
            scale = np.sqrt(1.0 / shape[1])
            if name.startswith("beta_layer.layers.2"):
                scale *= 0.1
            sd[name] = g.normal(scale=scale, size=shape).astype(np.float32)
        else:
            raise KeyError(f"no synthetic rule for {name} {shape}")
    return sd


def mmse_gain(tau, prior_var: float = 0.3):
    """d x0 / d x_t of the MMSE denoiser of a Gaussian prior N(mu, prior_var) (normalised pose space) under the cosine schedule
    abar(tau) = cos^2(tau pi / 2): J = sqrt(abar) v / (abar v + 1 - abar).  -> 1 at low noise, -> 0 at tau = 1."""
    tau = np.asarray(tau, dtype=np.float64)
    abar = np.cos((tau + 0.008) / 1.008 * np.pi / 2) ** 2
    return np.sqrt(abar) * prior_var / (abar * prior_var + 1.0 - abar)


SENSITIVE_CHANNELS = 12      # hidden channels 2d / 2d+1 carry +x_t[:, j, d] / -x_t[:, j, d] of the joint's 6-D rotation


def _norm_cdf_inv(p):
    from statistics import NormalDist
    nd = NormalDist()
    return np.array([nd.inv_cdf(float(min(max(q, 1e-4), 1 - 1e-4))) for q in np.ravel(p)]).reshape(np.shape(p))


def make_sensitive_state_dict(seed: int = 0, num_diffusion_timesteps: int = 100, prior_var: float = 0.3, gain: float = 1.0,
                              **manifest_kw) -> dict:
    """`make_state_dict(seed)` plus an **x_t-sensitive skip path** through the denoiser, so that the network behaves like a trained
    START_X denoiser: d x0 / d x_t ~ gain * mmse_gain(t / n) - close to `gain` at low noise, a few percent at t ~ n - instead of the
    ~0.06 at every t of the plain random network (whose output conv is scaled down on purpose).  Why it matters: the sampler's
    posterior mean x_{t-1} = c1 x0(x_t) + c2 x_t contracts an early step's rounding error only if c1 J + c2 < 1; for the ideal
    denoiser J = 1 / sqrt(abar) at low noise and c1 J + c2 = 1 / sqrt(alpha_t) >= 1, i.e. an error is carried to the output.  A
    precision schedule tuned on a network that ignores x_t says nothing about a checkpoint.

    Construction (exact in the reference's own ModulatedGCN arithmetic, modulated_gcn.py:99-116 - nothing here is a new operator):
      * hidden channels 2d, 2d+1 (d = 0..5) of the input conv receive p = +x_t[j,d] + s(t), n = -x_t[j,d] + s(t): the 512-d
        InputProcess slice of W[0] is the minimum-norm solution of Wp^T w = +-e_d, bp . w = 0; the timestep slice is a ridge fit of
        s(t) on the TimestepEmbedder's features for t in [0, n); the conditioning slices, W[1] (neighbour branch), the bias and the
        BatchNorm of these channels are neutral (0 / 1);
      * every residual block passes them through unchanged (gconv2 writes relu(0) = 0 on top of the skip), the other channels
        still read them through their random weights;
      * the output conv reads g_out * (relu(p) - relu(n)): gain 2 g_out when s >> |x|, g_out at s = 0, 0 when s << -|x|;
        s(t) = Phi^-1(target(t) / (2 g_out)) makes the average gain over x ~ N(0, 1) follow the target profile.
    `num_diffusion_timesteps` = the n of `create_gaussian_diffusion` the weights are "trained" for (the gate is a function of the
    ORIGINAL timestep the model is conditioned on, respace.py:124-129)."""
    sd = make_state_dict(seed, **manifest_kw)
    n = int(num_diffusion_timesteps)
    S = SENSITIVE_CHANNELS
    ch = np.arange(S)
    g_out = 0.5 * max(gain, 1e-3) * 1.02
    # --- timestep features phi(t) = time_embed.2(silu(time_embed.0(pe[t])))  (egohmr.py:636-643), float64
    pe = sd["embed_timestep.sequence_pos_encoder.pe"][:n, 0].astype(np.float64)
    h = pe @ sd["embed_timestep.time_embed.0.weight"].astype(np.float64).T + sd["embed_timestep.time_embed.0.bias"]
    h = h / (1.0 + np.exp(-h))
    phi = h @ sd["embed_timestep.time_embed.2.weight"].astype(np.float64).T + sd["embed_timestep.time_embed.2.bias"]   # [n,512]
    tau = np.arange(n) / max(n - 1, 1)
    target = gain * mmse_gain(tau, prior_var)
    s = np.clip(_norm_cdf_inv(target / (2.0 * g_out)), -3.5, 3.5)                      # [n]
    lam = 1e-6 * np.trace(phi.T @ phi) / phi.shape[1]
    v_t = np.linalg.solve(phi.T @ phi + lam * np.eye(phi.shape[1]), phi.T @ s)        # ridge: phi v ~ s
    # --- x_t slice: W[0, b:c, ch] with Wp^T w = +-e_d and bp . w = 0 (minimum norm)
    Wp = sd["input_process.poseEmbedding.weight"].astype(np.float64)                  # [512, 6]
    bp = sd["input_process.poseEmbedding.bias"].astype(np.float64)                    # [512]
    Amat = np.concatenate([Wp, bp[:, None]], axis=1)                                  # [512, 7]
    pinv = Amat @ np.linalg.inv(Amat.T @ Amat)                                        # w = pinv @ rhs solves Amat^T w = rhs
    p = "diffusion_model.gconv_input.0."
    W = sd[p + "gconv.W"]
    in_dim = W.shape[1]
    b, c, d_ = in_dim - 1024, in_dim - 512, in_dim
    W[:, :, ch] = 0.0
    for k in range(S):
        rhs = np.zeros(7)
        rhs[k // 2] = 1.0 if k % 2 == 0 else -1.0
        W[0, b:c, k] = (pinv @ rhs).astype(np.float32)
        W[0, c:d_, k] = v_t.astype(np.float32)
    sd[p + "gconv.M"][:, ch] = 1.0
    sd[p + "gconv.bias"][ch] = 0.0
    for leaf, val in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
        sd[p + "bn." + leaf][ch] = val
    blocks = sorted({k.split(".")[2] for k in sd if k.startswith("diffusion_model.gconv_layers.")})
    for blk in blocks:                                   # the residual carries the channels; the block adds relu(0) = 0 to them
        q = f"diffusion_model.gconv_layers.{blk}.gconv2."
        sd[q + "gconv.W"][:, :, ch] = 0.0
        sd[q + "gconv.bias"][ch] = 0.0
        sd[q + "bn.bias"][ch] = 0.0
        sd[q + "bn.running_mean"][ch] = 0.0
    Wo = sd["diffusion_model.gconv_output.W"]                                           # [2, hid, 6]
    Wo[:, ch, :] = 0.0
    for k in range(S):
        Wo[0, k, k // 2] = g_out if k % 2 == 0 else -g_out
    return sd


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------

def make_batch(batch_size: int, num_scene_points: int = 4096, seed: int = 0, vis_prob: float = 0.6) -> dict:
    """Synthetic batch dict (numpy, float32). Distributions: SURVEY.md section 8d."""
    g = _rng(3000 + seed)
    B, N = batch_size, num_scene_points
    transl = (np.array([0.0, 0.0, 3.0]) + g.uniform(-0.5, 0.5, size=(B, 3))).astype(np.float32)
    n_floor = int(0.3 * N)
    pts = g.uniform(-1.0, 1.0, size=(B, N, 3))
    pts[:, :n_floor, 1] = -1.0  # floor plane 1 m below the body centre
    scene = (pts + transl[:, None, :]).astype(np.float32)
    kp = np.zeros((B, 25, 3), dtype=np.float32)
    kp[:, :, 0] = g.uniform(0, 1920, size=(B, 25))
    kp[:, :, 1] = g.uniform(0, 1080, size=(B, 25))
    kp[:, :, 2] = (g.random((B, 25)) < vis_prob).astype(np.float32) * g.uniform(0.3, 1.0, size=(B, 25))
    return {
        "img": g.normal(size=(B, 3, 224, 224)).astype(np.float32),
        "orig_keypoints_2d": kp,
        "fx": np.ones(B, dtype=np.float32),
        "cam_cx": np.full(B, 960.0, dtype=np.float32),
        "cam_cy": np.full(B, 540.0, dtype=np.float32),
        "box_center": np.stack([g.uniform(400, 1500, size=B), g.uniform(200, 900, size=B)], -1).astype(np.float32),
        "box_size": g.uniform(150, 600, size=B).astype(np.float32),
        "smpl_params": {"transl": transl},
        "scene_pcd_verts_full": scene,
    }


def make_noise_stack(num_steps: int, batch_size: int, seed: int = 0, dim: int = 144) -> np.ndarray:
    """[num_steps+1, B, dim] float32 N(0,1): row 0 = x_T, row 1+k = noise of the k-th executed step."""
    return _rng(4000 + seed).normal(size=(num_steps + 1, batch_size, dim)).astype(np.float32)


def make_body_rep_stats(seed: int = 0, identity: bool = False):
    """Xmean/Xstd [144] like preprocess_stats.npz (dataloaders/egobody_dataset.py:100-117: the
    std is one scalar for dims 0:6 and one for dims 6:144)."""
    if identity:
        return np.zeros(144, dtype=np.float32), np.ones(144, dtype=np.float32)
    g = _rng(5000 + seed)
    mean = g.normal(scale=0.3, size=144).astype(np.float32)
    std = np.concatenate([np.full(6, 0.55), np.full(138, 0.42)]).astype(np.float32)
    return mean, std


def make_gt_annotations(batch_size: int, seed: int = 0) -> dict:
    """Ground-truth body annotations of a batch as the EgoBody loader delivers them (dataloaders/egobody_dataset.py:241-277:
    axis-angle ``global_orient`` [B,3] / ``body_pose`` [B,69], ``betas`` [B,10], ``gender`` [B] with 0 = male, 1 = female),
    for the driver block of test_egohmr.py:268-318.  Its own random stream: ``make_batch`` stays bit-identical."""
    g = _rng(7000 + seed)
    B = batch_size
    return {"global_orient": g.normal(scale=0.4, size=(B, 3)).astype(np.float32),
            "body_pose": g.normal(scale=0.25, size=(B, 69)).astype(np.float32),
            "betas": g.normal(scale=0.8, size=(B, 10)).astype(np.float32),
            "gender": (g.random(B) < 0.5).astype(np.int64)}
