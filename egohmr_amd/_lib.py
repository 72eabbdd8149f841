"""ctypes binding of libegohmr_hip.so (C ABI: include/egohmr_hip.h).

The library is the product: there is no eager / CPU fallback.  If it is missing or cannot be
loaded this module raises - loudly - at first use.

``import torch`` happens before the dlopen on purpose: PyTorch-ROCm ships its own
``libamdhip64.so`` (SONAME libamdhip64.so.7, same as /opt/rocm's) and the dynamic loader then
binds our library to that already-loaded runtime, so device pointers / streams handed over
from torch tensors are valid inside the kernels' launches.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EHM_LIB_PATH") or os.path.join(_HERE, "libegohmr_hip.so")   # EHM_LIB_PATH: A/B a second build (experiments)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
SOURCES = ["gcn.hip", "gcn_tile.hip", "linear.hip", "conv.hip", "stem.hip", "metrics.hip", "smpl.hip", "sampler.hip", "guidance.hip", "prep.hip", "step.hip", "eval.hip"]


class EgoHMRHipError(RuntimeError):
    pass


class EgoHMRRangeError(EgoHMRHipError):
    """rc = -34: an activation of the denoiser reached the f16 range and was clamped in its X2 / f16 store (ehm_gcn_stack_status).  The results are finite
    but not parity grade; the remedy is float32 activations for that checkpoint: EgoHMR.gcn_precision = 'f32' (EgoHMR.on_saturation = 'f32' does it
    and re-runs the call)."""


HEADERS = ["common.h", "smpl_dev.h", "gcn_dev.h", "step_dev.h", "internal.h"]


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile the gfx950 library in-tree with hipcc (cross-compiles without a GPU): one object per source, the sources in parallel, objects
    kept under egohmr_amd/build/ and reused while neither their source, a header nor the flags changed."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("EHM_HIPCC_FLAGS", "").split()   # e.g. -DEHM_STAMPS (tools/stamp_*.py)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", *extra]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "egohmr_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    tag = hashlib.sha1(" ".join([hipcc] + flags).encode()).hexdigest()[:10]
    objdir = os.path.join(_HERE, "build", os.path.basename(LIB_PATH) + "." + tag)
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    sources = SOURCES
    for s in sources:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in sources]
    # every flag set links into the same LIB_PATH: the tag of the object directory the library was linked from is kept beside it, and a library
    # linked from ANOTHER flag set (say -DEHM_STAMPS, then a default build whose objects were already up to date) is relinked, never reused
    tag_path = LIB_PATH + ".tag"
    linked_tag = open(tag_path).read().strip() if os.path.exists(tag_path) else None
    if not jobs and os.path.exists(LIB_PATH) and linked_tag == tag and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        cmd = [hipcc, *flags, "-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=int(os.environ.get("EHM_BUILD_JOBS", min(len(jobs) or 1, os.cpu_count() or 1)))) as ex:
        for (src, _), r in zip(jobs, ex.map(compile_one, jobs)):
            if r.returncode != 0:
                raise EgoHMRHipError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
            if "failed to meet occupancy target" in r.stderr:
                # __launch_bounds__(256, 2) is a wish, not a limit: a kernel that lost its second block per CU still compiles (with this warning) and
                # every test stays green at half the speed - the host-side plans count on the occupancy, so treat it as a build error
                os.remove(os.path.join(objdir, os.path.basename(src).replace(".hip", ".o")))
                raise EgoHMRHipError(f"hipcc: a kernel of {src} missed its occupancy target:\n{r.stderr}")
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise EgoHMRHipError(f"hipcc (link) failed:\n{r.stdout}\n{r.stderr}")
    with open(tag_path, "w") as f:
        f.write(tag + "\n")
    return LIB_PATH


def build_features() -> set:
    """Optional parts the loaded library was built with (ehm_build_features): 'stamps'."""
    return set(lib().ehm_build_features().decode().split())


class GConvParams(C.Structure):
    """ehm_gconv_params"""
    _fields_ = [("W", C.c_void_p), ("M", C.c_void_p), ("adj2", C.c_void_p), ("bias", C.c_void_p),
                ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p),
                ("in_dim", C.c_int), ("out_dim", C.c_int)]


class LinearDesc(C.Structure):
    """ehm_linear_desc"""
    _fields_ = [("A0", C.c_void_p), ("A1", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("group_bias", C.c_void_p),
                ("Y", C.c_void_p), ("colmax", C.c_void_p), ("M", C.c_int64), ("N", C.c_int), ("K0", C.c_int), ("K1", C.c_int),
                ("rows_per_group", C.c_int), ("valid_rows_per_group", C.c_int), ("relu_in0", C.c_int), ("relu_out", C.c_int),
                ("w_scale", C.c_float), ("lift_points", C.c_void_p), ("lift_W4", C.c_void_p), ("hi_only", C.c_int), ("group_bias_stride", C.c_int)]


class ConvDesc(C.Structure):
    """ehm_conv_desc"""
    _fields_ = [("x", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("Ci", C.c_int), ("Co", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("relu", C.c_int),
                ("w_scale", C.c_float)]


class ConvX2Desc(C.Structure):
    """ehm_conv_x2_desc"""
    _fields_ = [("x", C.c_void_p), ("x_rows", C.c_int64), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("Ci", C.c_int), ("Co", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("relu", C.c_int),
                ("w_scale", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("x2", C.c_void_p), ("x2_rows", C.c_int64), ("H2", C.c_int), ("W2", C.c_int), ("Ci2", C.c_int), ("stride2", C.c_int), ("hi_only", C.c_int),
                ("workspace_clean", C.c_int)]


class EvalPointsDesc(C.Structure):
    """ehm_eval_points_desc"""
    _fields_ = [("pred", C.c_void_p), ("gt", C.c_void_p), ("pred_origin", C.c_void_p), ("gt_origin", C.c_void_p), ("mask", C.c_void_p),
                ("per_point", C.c_void_p), ("mean", C.c_void_p), ("vis_sum", C.c_void_p), ("invis_sum", C.c_void_p),
                ("B", C.c_int), ("S", C.c_int), ("P", C.c_int), ("pred_points", C.c_int), ("gt_points", C.c_int), ("origin_point", C.c_int)]


class ItemPrepDesc(C.Structure):
    """ehm_item_prep_desc"""
    _fields_ = [("keypoints_2d", C.c_void_p), ("joint_map", C.c_void_p), ("NK", C.c_int), ("force_visible", C.c_int),
                ("fx", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p), ("box_center", C.c_void_p), ("box_size", C.c_void_p), ("transl", C.c_void_p),
                ("fx_norm", C.c_float), ("with_bbox", C.c_int), ("with_cam_center", C.c_int),
                ("tW1", C.c_void_p), ("tb1", C.c_void_p), ("tW2", C.c_void_p), ("tb2", C.c_void_p), ("t_hidden", C.c_int), ("t_out", C.c_int),
                ("img_rowsum", C.c_void_p), ("scene_rowsum", C.c_void_p), ("other", C.c_void_p), ("other_ld", C.c_int), ("other_col0", C.c_int),
                ("vis", C.c_void_p), ("mask_slot", C.c_void_p), ("mask_items", C.c_void_p), ("count", C.c_void_p), ("finite", C.c_void_p),
                ("need_scratch", C.c_void_p), ("pass_group", C.c_int), ("B", C.c_int)]


class PackDesc(C.Structure):
    """ehm_pack_desc"""
    _fields_ = [("B", C.c_int), ("J", C.c_int), ("V", C.c_int), ("finite", C.c_void_p), ("chk", C.c_void_p), ("chk_rows", C.c_int),
                ("last_noise", C.c_void_p), ("x_final", C.c_void_p), ("x0", C.c_void_p), ("pose6d", C.c_void_p), ("R", C.c_void_p),
                ("verts", C.c_void_p), ("joints", C.c_void_p), ("betas_in", C.c_void_p), ("betas_out", C.c_void_p),
                ("transl", C.c_void_p), ("fx", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p), ("fx_norm", C.c_float),
                ("global_orient", C.c_void_p), ("body_pose", C.c_void_p), ("kp3d_full", C.c_void_p), ("kp2d_full", C.c_void_p),
                ("focal", C.c_void_p), ("center", C.c_void_p), ("finite_out", C.c_void_p)]


class StepCoefs(C.Structure):
    """ehm_step_coefs"""
    _fields_ = [(n, C.c_float) for n in ("coef1", "coef2", "log_variance", "variance", "sqrt_recip_ac", "sqrt_recipm1_ac",
                                         "sqrt_ac_prev", "dir_coef", "sigma", "nonzero", "grad_scale")]


class NonlocalParams(C.Structure):
    """ehm_nonlocal_params"""
    _fields_ = [("Wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("qkv_scale", C.c_float), ("Wo", C.c_void_p), ("bo", C.c_void_p), ("o_scale", C.c_float),
                ("Ci", C.c_int)]


class SampleDesc(C.Structure):
    """ehm_sample_desc"""
    _fields_ = [("B", C.c_int), ("passes", C.c_int), ("num_steps", C.c_int), ("ddim", C.c_int), ("lbs_every_step", C.c_int),
                ("num_scene_points", C.c_int), ("guide_denom", C.c_float), ("tau", C.c_float), ("num_masked", C.c_int), ("guide_all_points", C.c_int), ("lowprec_steps", C.c_int), ("nonlocal_ci", C.c_int), ("per_step_launches", C.c_int)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/egohmr_hip.h declares
PROTOTYPES = {
    "ehm_last_error": (C.c_char_p, []),
    "ehm_target_arch": (C.c_char_p, []),
    "ehm_build_features": (C.c_char_p, []),
    "ehm_rot6d_to_rotmat": (_I, [_P, _P, _L, _I, _P]),
    "ehm_rot6d_to_rotmat_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "ehm_rotmat_to_angle_axis": (_I, [_P, _P, _L, _P]),
    "ehm_rotmat_to_angle_axis_bwd": (_I, [_P, _P, _P, _L, _P]),
    "ehm_smpl_create": (_I, [C.POINTER(_P), _P, _P, _P, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _P]),
    "ehm_smpl_destroy": (None, [_P]),
    "ehm_smpl_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "ehm_smpl_forward_rot6d": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "ehm_gcn_create": (_I, [C.POINTER(_P), _P, C.POINTER(GConvParams), C.POINTER(GConvParams), _I, C.POINTER(GConvParams), _I, _P]),
    "ehm_gcn_destroy": (None, [_P]),
    "ehm_gcn_row_tile": (_I, []),
    "ehm_gcn_set_precision": (_I, [_P, _I]),
    "ehm_gcn_get_precision": (_I, [_P]),
    "ehm_gcn_set_uncond_mode": (_I, [_P, _I]),
    "ehm_gcn_set_pass_map": (_I, [_P, _P, _P, _I]),
    "ehm_gcn_set_nonlocal": (_I, [_P, C.POINTER(NonlocalParams)]),
    "ehm_gcn_reserve": (_I, [_P, _I, _I]),
    "ehm_gcn_activation_group": (_I, [_P]),
    "ehm_gcn_pack_activations": (_I, [_P, _P, _L, _I, _I, _P]),
    "ehm_gcn_unpack_activations": (_I, [_P, _P, _L, _I, _I, _P]),
    "ehm_gcn_pack_activations_checked": (_I, [_P, _P, _P, _L, _P]),
    "ehm_gcn_input_layer": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "ehm_gcn_input_layer_rows": (_I, [_P, _P, _P, _I, _P]),
    "ehm_gcn_hidden_layer": (_I, [_P, _I, _P, _P, _P, _L, _P]),
    "ehm_gcn_hidden_stack": (_I, [_P, C.POINTER(C.c_void_p), _L, C.POINTER(C.c_int), _P]),
    "ehm_gcn_stack_status": (_I, [_P, _P]),
    "ehm_gcn_stack_status_async": (_I, [_P, _P, _P]),
    "ehm_gcn_output_layer": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "ehm_linear_split": (_I, [C.POINTER(LinearDesc), _P]),
    "ehm_split_pack": (_I, [_P, _P, _L, _I, _I, _F, _P]),
    "ehm_skinny_gemm_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ehm_resnet_stem_scratch_bytes": (C.c_size_t, [_I, _I, _I]),
    "ehm_resnet_stem": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ehm_conv_x2_rows": (C.c_int64, [C.c_int64]),
    "ehm_conv_x2": (_I, [C.POINTER(ConvX2Desc), _P]),
    "ehm_conv_x2_workspace_bytes": (C.c_int64, [C.POINTER(ConvX2Desc)]),
    "ehm_conv_x2_workspace_status": (_I, [_P, _P, _P]),
    "ehm_x2_group_mean": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ehm_conv_nhwc_split": (_I, [C.POINTER(ConvDesc), _P]),
    "ehm_nonlocal_attention": (_I, [_P, _P, _L, _I, _P]),
    "ehm_pointnet_lift": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ehm_ddpm_step": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _L, _P]),
    "ehm_ddim_step": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _L, _P]),
    "ehm_collision_proxy": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ehm_collision_query": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "ehm_smpl_backward_rot6d": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "ehm_guidance_grad_finish": (_I, [_P, _P, _P, _I, _F, _P]),
    "ehm_nn_dist2": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ehm_eval_point_errors": (_I, [C.POINTER(EvalPointsDesc), _P]),
    "ehm_eval_procrustes": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ehm_eval_diversity": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "ehm_sample_workspace_bytes": (_L, [C.POINTER(SampleDesc), _I, _I]),
    "ehm_sample_loop": (_I, [_P, _P, C.POINTER(SampleDesc), C.POINTER(StepCoefs)] + [_P] * 18 + [_L, _P]),
    "ehm_item_prep": (_I, [C.POINTER(ItemPrepDesc), _P]),
    "ehm_pack_outputs": (_I, [C.POINTER(PackDesc), _P]),
    "ehm_profile_begin": (_I, []),
    "ehm_profile_end": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_int64), _I]),
}
PROF_CLASSES = ("input", "chain_f16x3", "chain_f16", "hidden_f32", "out_dot", "step_body", "skin_input", "guidance", "reserved_8", "reserved_9",
                "guid_nearest", "guid_skin_bwd", "guid_posefeat_bwd", "step_fused", "guid_nearest_evals")   # EHM_PROF_* of the header (the last one is a COUNT in `launches`)

_lib = None


def lib() -> C.CDLL:
    """The loaded library.  Raises EgoHMRHipError when it is absent - never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EgoHMRHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  egohmr_amd has no CPU/eager fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise EgoHMRHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise EgoHMRHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        if handle.ehm_target_arch() != b"gfx950":
            raise EgoHMRHipError("libegohmr_hip.so was not built for gfx950")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ehm_last_error()
        if rc == -34:
            raise EgoHMRRangeError(f"{what or 'libegohmr_hip'} (rc={rc}): {msg.decode() if msg else ''}")
        raise EgoHMRHipError(f"{what or 'libegohmr_hip'} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a contiguous float32/uint8 CUDA(HIP) tensor (0 for None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EgoHMRHipError("egohmr_amd kernels need tensors on a HIP device (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise EgoHMRHipError("tensor handed to the C ABI must be contiguous")
    return t.data_ptr()


class TensorKey:
    """(data_ptr, _version) of every parameter and buffer of some modules, as a cache key, without walking the module tree each time:
    `Module.parameters()` costs ~0.9 ms for ResNet-50 + the PointNet and was evaluated several times per sampling call with the GPU
    idle behind it.  The slots (owner dict, name) are collected once; a replaced Parameter object, an in-place update (_version) and a
    move to another device / dtype (data_ptr) all change the key.  Modules or parameters ADDED later are not seen: call refresh()."""

    def __init__(self, *modules):
        self.modules = modules
        self.refresh()

    def refresh(self):
        self.slots = []
        seen = set()
        for mod in self.modules:
            for m in mod.modules():
                if id(m) in seen:
                    continue
                seen.add(id(m))
                self.slots += [(m._parameters, n) for n in m._parameters] + [(m._buffers, n) for n in m._buffers]

    def __call__(self):
        out = []
        for d, n in self.slots:
            t = d.get(n)
            if t is not None:
                out.append((t.data_ptr(), t._version))
        return tuple(out)


def f32(t, device=None):
    """contiguous float32 view/copy on the device."""
    t = t.detach()
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def on_device(dev):
    """Context for a native call: the HIP device of the tensors it is given becomes the current one (the library launches on the CURRENT
    device's stream; a model living on cuda:1 of a process whose current device is cuda:0 must not launch there).  CPU devices pass through -
    the call sites reject CPU tensors themselves (EgoHMRHipError)."""
    import contextlib
    dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
    return torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()
