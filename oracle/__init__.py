"""CPU oracle for the EgoHMR stage-2 sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy fp64 for the schedule tables, eager torch-CPU for the
tensor maths, dtype selectable fp32/fp64), the algorithm of the reference path

    GaussianDiffusion.val_losses -> p_sample_loop / ddim_sample_loop -> p_sample / ddim_sample
      -> EgoHMR.forward -> ModulatedGCN -> rot6d_to_rotmat -> smplx.SMPL.forward (LBS)

Every function cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only
as the checker / the timed CPU baseline - never as the product path.  ``egohmr_amd`` does not
import this package and raises if its HIP library is missing.

Pinning status (details in DESIGN.md section "Oracle"):
* schedule, samplers, EgoHMR.forward, Modulated-GCN, encoders, rot6d, rotmat->axis-angle:
  PINNED - checked against golden vectors produced in the build container by importing the
  reference itself (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
* SMPL linear blend skinning: the arithmetic lives in pip ``smplx==0.1.28`` which is absent from
  /root/reference and from this image; restated from the published smplx algorithm
  (``smplx/lbs.py``: blend_shapes, vertices2joints, batch_rigid_transform, lbs;
  ``smplx/body_models.py``: SMPL.forward; ``smplx/vertex_joint_selector.py``).  PARITY UNPINNED
  by the reference (it has no tests); pinned only by algebraic properties (tests/test_oracle_smpl.py).
* collision guidance: COAP / VolumetricSMPL are learned networks that cannot be obtained offline.
  The plumbing around them (bbox selection, reduction, gradient masks, mean shift) is pinned by
  goldens; the collision term itself is a build-defined proxy.  PARITY UNPINNED.
"""
