"""SMPL body model with the ``smplx`` call surface the reference uses, running on libegohmr_hip.

Reference seams honoured:
  smplx.create('data/smpl', model_type='smpl', gender=..., [create_transl=False, batch_size=...])
      models/egohmr/egohmr.py:105-107, test_egohmr.py:143-145
  model(betas=, body_pose=, global_orient=, transl=, return_full_pose=, pose2rot=False)
      egohmr.py:276,492,537; test_egohmr.py:291  -> .vertices [B,6890,3] .joints [B,45,3] .full_pose
  registered buffers v_template / shapedirs / posedirs / J_regressor / lbs_weights / parents /
  faces_tensor (smplx names, so COAP-like consumers and reference checkpoints keep working).

The arithmetic (smplx/lbs.py of pip smplx==0.1.28, absent from the reference tree) runs in
csrc/smpl.hip; there is no torch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib, synthetic


class SMPLOutput:
    """Default-constructible bag of attributes (smplx.utils.SMPLOutput as used at egohmr.py:393-404)."""

    def __init__(self, vertices=None, joints=None, full_pose=None, betas=None, body_pose=None, global_orient=None, **kw):
        self.vertices, self.joints, self.full_pose = vertices, joints, full_pose
        self.betas, self.body_pose, self.global_orient = betas, body_pose, global_orient
        for k, v in kw.items():
            setattr(self, k, v)


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, asset: dict, gender: str = "neutral", batch_size: int = 1, create_transl: bool = True, **_):
        super().__init__()
        self.gender = gender
        self.batch_size = batch_size
        f = lambda k: torch.as_tensor(np.asarray(asset[k]), dtype=torch.float32).contiguous()
        self.register_buffer("v_template", f("v_template"))
        self.register_buffer("shapedirs", f("shapedirs")[:, :, :10].contiguous())
        self.register_buffer("posedirs", f("posedirs"))
        self.register_buffer("J_regressor", f("J_regressor"))
        self.register_buffer("lbs_weights", f("lbs_weights"))
        parents = torch.as_tensor(np.asarray(asset["parents"]), dtype=torch.long).clone()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("faces_tensor", torch.as_tensor(np.asarray(asset["faces"]), dtype=torch.long))
        self.faces = np.asarray(asset["faces"])
        self.register_buffer("extra_joints_idxs", torch.as_tensor(np.asarray(asset["extra_joints_idxs"]), dtype=torch.long))
        self._handle = None
        self._handle_key = None

    @property
    def num_verts(self) -> int:
        return self.v_template.shape[0]

    @property
    def num_joints_out(self) -> int:
        return 24 + self.extra_joints_idxs.numel()

    def get_num_verts(self):
        return self.num_verts

    # ------------------------------------------------------------------ native handle
    def handle(self):
        """ehm_smpl* for the current device placement of the buffers (re-created after .to())."""
        bufs = (self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.lbs_weights)
        key = tuple((b.data_ptr(), b._version) for b in bufs) + (str(self.v_template.device),)   # in-place loads (load_state_dict) re-pack too
        if self._handle is None or self._handle_key != key:
            self._free()
            L = _lib.lib()
            if not self.v_template.is_cuda:
                raise _lib.EgoHMRHipError("SMPL buffers are on the CPU: move the module to a HIP device (.to('cuda'))")
            par = (C.c_int32 * 24)(*[int(p) for p in self.parents.tolist()])
            idx = self.extra_joints_idxs.tolist()
            ext = (C.c_int32 * max(len(idx), 1))(*idx)
            h = C.c_void_p()
            with torch.cuda.device(self.v_template.device):
                _lib.check(L.ehm_smpl_create(C.byref(h), _lib.ptr(self.v_template), _lib.ptr(self.shapedirs), _lib.ptr(self.posedirs),
                                             _lib.ptr(self.J_regressor), _lib.ptr(self.lbs_weights), par, ext, self.num_verts,
                                             len(idx), _lib.stream_ptr()), "ehm_smpl_create")
            self._handle, self._handle_key = h, key
        return self._handle

    def _free(self):
        if self._handle is not None:
            try:
                _lib.lib().ehm_smpl_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def __del__(self):
        self._free()

    # ------------------------------------------------------------------ forward
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True,
                return_full_pose=False, pose2rot=True, **kwargs):
        dev = self.v_template.device
        if pose2rot:        # axis-angle inputs (ground-truth bodies of the driver, test_egohmr.py:307-310): Rodrigues like smplx's batch_rodrigues
            from .geometry import aa_to_rotmat
            B0 = max(body_pose.shape[0], global_orient.shape[0])
            global_orient = aa_to_rotmat(_lib.f32(global_orient, dev).reshape(-1, 3)).reshape(B0, 1, 3, 3)
            body_pose = aa_to_rotmat(_lib.f32(body_pose, dev).reshape(-1, 3)).reshape(B0, 23, 3, 3)
        if dev.type != "cuda":
            raise _lib.EgoHMRHipError("SMPL.forward needs the module on a HIP device (.to('cuda')); egohmr_amd has no CPU path")
        B = max(betas.shape[0], body_pose.shape[0], global_orient.shape[0])
        betas = _lib.f32(betas, dev)
        if betas.shape[0] != B:
            betas = betas.expand(B, -1).contiguous()
        full = torch.cat([_lib.f32(global_orient, dev).reshape(B, 1, 3, 3), _lib.f32(body_pose, dev).reshape(B, 23, 3, 3)], dim=1).contiguous()
        verts = torch.empty(B, self.num_verts, 3, device=dev, dtype=torch.float32)
        joints = torch.empty(B, self.num_joints_out, 3, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ehm_smpl_forward(self.handle(), _lib.ptr(betas), _lib.ptr(full), _lib.ptr(verts), _lib.ptr(joints),
                                                   None, B, _lib.stream_ptr()), "ehm_smpl_forward")
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            verts = verts + transl.unsqueeze(1)
        return SMPLOutput(vertices=verts if return_verts else None, joints=joints, betas=betas, body_pose=body_pose,
                          global_orient=global_orient, full_pose=full if return_full_pose else None)


def load_smpl_asset(model_path: str) -> dict:
    """Read an official SMPL_*.pkl without chumpy (arrays are unwrapped by attribute)."""
    class _Stub:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.startswith("chumpy") or module.startswith("scipy.sparse"):
                if module.startswith("scipy.sparse"):
                    return super().find_class(module, name)
                return _Stub
            return super().find_class(module, name)

    with open(model_path, "rb") as f:
        d = _Unpickler(f, encoding="latin1").load()

    def arr(v):
        if isinstance(v, _Stub):
            v = v.__dict__.get("x", v.__dict__.get("r"))
        if hasattr(v, "toarray"):
            v = v.toarray()
        return np.asarray(v)

    V = arr(d["v_template"]).shape[0]
    posedirs = arr(d["posedirs"]).reshape(V * 3, -1).T          # smplx: [P, V*3]
    kin = arr(d["kintree_table"]).astype(np.int64)
    return {
        "v_template": arr(d["v_template"]).astype(np.float32),
        "shapedirs": arr(d["shapedirs"])[:, :, :10].astype(np.float32),
        "posedirs": np.ascontiguousarray(posedirs).astype(np.float32),
        "J_regressor": arr(d["J_regressor"]).astype(np.float32),
        "lbs_weights": arr(d["weights"]).astype(np.float32),
        "parents": kin[0],
        "faces": arr(d["f"]).astype(np.int64),
        "extra_joints_idxs": synthetic.EXTRA_JOINT_VERTEX_IDS.copy(),
    }


_GENDER_FILE = {"neutral": "SMPL_NEUTRAL.pkl", "male": "SMPL_MALE.pkl", "female": "SMPL_FEMALE.pkl"}


def resolve_model_file(model_path: str, model_type: str, gender: str) -> str | None:
    """smplx.create's path resolution (pip smplx 0.1.28 body_models.create): a directory gets ``model_type`` appended, then the
    gender file name - the documented layout is ``data/smpl/smpl/SMPL_NEUTRAL.pkl`` (reference README.md:60-65).  A flat
    ``data/smpl/SMPL_NEUTRAL.pkl`` and a direct file path are accepted as well."""
    if os.path.isdir(model_path):
        for cand in (os.path.join(model_path, model_type, _GENDER_FILE[gender]), os.path.join(model_path, _GENDER_FILE[gender])):
            if os.path.isfile(cand):
                return cand
        return None
    return model_path if os.path.isfile(model_path) else None


def create(model_path: str = "data/smpl", model_type: str = "smpl", gender: str = "neutral", asset: dict | None = None,
           allow_synthetic: bool = False, **kwargs) -> SMPL:
    """Drop-in for ``smplx.create`` (egohmr.py:105-107).  The licensed model files cannot ship, so tests / benchmarks pass a
    synthetic ``asset`` (or ``allow_synthetic=True``) explicitly; a missing file is otherwise an error, never a silent
    substitute - every vertex and metric would come from a fake body."""
    if model_type != "smpl":
        raise ValueError("only model_type='smpl' is on the EgoHMR path")
    if asset is None:
        path = resolve_model_file(model_path, model_type, gender)
        if path is not None:
            asset = load_smpl_asset(path)
        elif allow_synthetic:
            asset = synthetic.make_smpl_asset({"neutral": 0, "male": 1, "female": 2}[gender])
        else:
            raise FileNotFoundError(
                f"no SMPL model for gender '{gender}' under '{model_path}' (looked for {model_type}/{_GENDER_FILE[gender]} and "
                f"{_GENDER_FILE[gender]}); pass asset=... or allow_synthetic=True to run on the synthetic SMPL-shaped asset")
    return SMPL(asset, gender=gender, **kwargs)
