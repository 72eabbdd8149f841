"""egohmr_amd - MI355X-native EgoHMR stage-2 diffusion sampling hot path.

Drop-in surface (same names as the reference, see DESIGN.md / INTEGRATION.md):
    from egohmr_amd import create_gaussian_diffusion, EgoHMR, smpl, rot6d_to_rotmat
Importing the package does not load the HIP library; the first kernel call does, and raises
``EgoHMRHipError`` if ``libegohmr_hip.so`` is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"

_LAZY = {
    "create_gaussian_diffusion": ("diffusion", "create_gaussian_diffusion"),
    "SpacedDiffusion": ("diffusion", "SpacedDiffusion"),
    "GaussianDiffusion": ("diffusion", "GaussianDiffusion"),
    "EgoHMR": ("model", "EgoHMR"),
    "EgoHMRVolsmpl": ("model", "EgoHMRVolsmpl"),
    "rot6d_to_rotmat": ("geometry", "rot6d_to_rotmat"),
    "rotmat_to_rot6d": ("geometry", "rotmat_to_rot6d"),
    "EgoHMRHipError": ("_lib", "EgoHMRHipError"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    if name in ("smpl", "synthetic", "diffusion", "model", "geometry", "dist", "_lib", "encoders"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
