"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/egohmr_hip.h declares
(no compute calls - there is no GPU here); host-side schedule logic equals the reference's goldens."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "egohmr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ehm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from egohmr_amd import _lib
    _lib.build()
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/egohmr_hip.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == declared
    assert L.ehm_target_arch() == b"gfx950"
    assert L.ehm_gcn_row_tile() == 192


def test_header_is_plain_c_and_a_c_program_binds_the_library(tmp_path):
    """include/egohmr_hip.h is the drop-in boundary: it must compile as strict C99 (and C++17) on its own, and a C program that includes it
    must link against the library and call through it (no torch, no Python types) - here the GPU-free entry points only."""
    import shutil
    import subprocess
    from egohmr_amd import _lib
    _lib.build()
    inc = os.path.join(REPO, "include")
    so = _lib.library_path() if hasattr(_lib, "library_path") else os.path.join(REPO, "egohmr_amd", "libegohmr_hip.so")
    src = tmp_path / "bind.c"
    src.write_text(
        '#include "egohmr_hip.h"\n#include <stdio.h>\n#include <string.h>\n'
        "int main(void) {\n"
        "  ehm_sample_desc d; memset(&d, 0, sizeof d);\n"
        '  if (strcmp(ehm_target_arch(), "gfx950") != 0) return 1;\n'
        "  if (ehm_gcn_row_tile() != 192) return 2;\n"
        "  if (ehm_sample_workspace_bytes(NULL, 1024, 6890) != -22) return 3;          /* EHM_EINVAL without touching a GPU */\n"
        '  if (strstr(ehm_last_error(), "bad argument") == NULL) return 4;\n'
        '  printf("%d %s %d %d %d %d %d %d %d %d %d\\n", (int)sizeof d, ehm_target_arch(), (int)sizeof(ehm_gconv_params), (int)sizeof(ehm_linear_desc),\n'
        "         (int)sizeof(ehm_conv_desc), (int)sizeof(ehm_conv_x2_desc), (int)sizeof(ehm_step_coefs), (int)sizeof(ehm_nonlocal_params),\n"
        "         (int)sizeof(ehm_item_prep_desc), (int)sizeof(ehm_pack_desc), (int)sizeof(ehm_eval_points_desc));\n"
        "  return 0;\n}\n")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)], check=True)
    exe = tmp_path / "bind"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), so, "-o", str(exe), "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.split()[1] == "gfx950"
    import ctypes
    f = out.stdout.split()
    mirrors = [_lib.SampleDesc, None, _lib.GConvParams, _lib.LinearDesc, _lib.ConvDesc, _lib.ConvX2Desc, _lib.StepCoefs, _lib.NonlocalParams,
               _lib.ItemPrepDesc, _lib.PackDesc, _lib.EvalPointsDesc]
    for i, cls in enumerate(mirrors):
        if cls is not None:
            assert int(f[i]) == ctypes.sizeof(cls), f"the ctypes mirror {cls.__name__} has drifted from the header ({f[i]} vs {ctypes.sizeof(cls)})"


def test_cabi_rejects_bad_arguments_without_a_gpu():
    from egohmr_amd import _lib
    L = _lib.lib()
    assert L.ehm_gcn_hidden_layer(None, 0, None, None, None, 192, None) == -22
    assert b"bad argument" in L.ehm_last_error()
    assert L.ehm_ddpm_step(None, None, None, None, None, 0, 0, 0, 0, 0, 10, None) == -22
    assert L.ehm_sample_workspace_bytes(None, 1024, 6890) == -22


def test_product_path_fails_loudly_on_cpu_tensors():
    import torch
    from egohmr_amd import _lib
    from egohmr_amd.geometry import rot6d_to_rotmat
    with pytest.raises(_lib.EgoHMRHipError):
        rot6d_to_rotmat(torch.zeros(4, 6), "diffusion")


@pytest.mark.parametrize("n,rs", [(50, ""), (50, "ddim5"), (100, "ddim10"), (1000, "ddim50")])
def test_product_schedule_equals_reference_golden(golden_dir, n, rs):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    g = np.load(os.path.join(golden_dir, "g1_schedules.npz"))
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    tag = f"n{n}_{rs or 'ddpm'}"
    for f in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(d, f), g[f"{tag}__{f}"])
    assert d.timestep_map == list(g[f"{tag}__timestep_map"])
    with pytest.raises(ValueError):
        create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim49")


def test_state_dict_names_match_reference_manifest():
    """egohmr_manifest() was asserted equal to the reference's own state_dict keys/shapes when the goldens
    were generated (oracle/make_golden.py: build_reference_model); the product module must match it."""
    from egohmr_amd import synthetic as syn
    from egohmr_amd.factory import build_synthetic_model
    m = build_synthetic_model("cpu", 0)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("smpl.")}
    assert mine == dict(syn.egohmr_manifest())


def test_oracle_is_test_infrastructure_only():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/: nothing under egohmr_amd/ or tools/ does, and the
    two allowed files import it inside exactly those functions."""
    import ast
    import glob
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef):
                for sub in ast.walk(node):
                    if isinstance(sub, (ast.Import, ast.ImportFrom)):
                        names = [a.name for a in sub.names] if isinstance(sub, ast.Import) else [sub.module or ""]
                        if any(n == "oracle" or n.startswith("oracle.") for n in names):
                            hits.append(node.name)
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        for sub in top:
            names = [a.name for a in sub.names] if isinstance(sub, ast.Import) else [sub.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                hits.append("<module>")
        return hits

    for f in glob.glob(os.path.join(repo, "egohmr_amd", "*.py")) + glob.glob(os.path.join(repo, "tools", "*.py")):
        assert oracle_imports(f) == [], f
    assert set(oracle_imports(os.path.join(repo, "bench.py"))) == {"cpu_baseline"}
    assert set(oracle_imports(os.path.join(repo, "__graft_entry__.py"))) == {"smoke"}


def test_default_library_has_no_experiment_and_no_env_switch_on_the_launch_path():
    """the experiment engines (the one-launch sampling loop, the 96 x 64 wave tile, the 32 x 32 x 16 form of the split-f16
    chain) are GONE from the tree - no source file, no kernel in the library, no second code path through the tile engine - and no environment variable may
    change which kernels a caller's process runs: the only getenv left in the product sources is EHM_F16_CHAIN, read once in ehm_gcn_create (handle creation);
    -DEHM_STAMPS (in-kernel time stamps for tools/stamp_*.py) is the one build-time instrumentation flag."""
    import glob
    from egohmr_amd import _lib
    assert not os.environ.get("EHM_HIPCC_FLAGS") and not os.environ.get("EHM_LIB_PATH"), "this test is about the DEFAULT build"
    _lib.build()
    assert _lib.build_features() == set()
    with open(_lib.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"gcn_loop_kernel" not in blob and b"gcn_hidden_wide_kernel" not in blob and b"gcn_hidden_chain_kernel" in blob
    sites = []
    for path in sorted(glob.glob(os.path.join(REPO, "egohmr_amd", "csrc", "*"))):
        assert os.path.basename(path) not in ("gcn_loop_host.inc", "gcn_loop_dev.h", "gcn_wide.hip")
        depth = []                                                  # stack of "is this #if block the instrumentation flag"
        for ln, line in enumerate(open(path), 1):
            st = line.strip()
            assert "EHM_WITH_LOOP_ENGINE" not in st and "EHM_WITH_WIDE_TILE" not in st and "EHM_P3_MFMA32" not in st and "EHM_LOOPSTAT" not in st, (path, ln)
            if st.startswith("#if"):
                depth.append("EHM_STAMPS" in st and not st.startswith("#ifndef"))
            elif st.startswith("#endif") and depth:
                depth.pop()
            elif "getenv(" in line and not any(depth):
                sites.append((os.path.basename(path), ln, st))
    assert [(f, s.split('getenv("')[1].split('"')[0]) for f, _, s in sites] == [("gcn.hip", "EHM_F16_CHAIN")], sites
    assert not os.path.exists(os.path.join(REPO, "tools", "jobs"))


def test_conv_x2_stream_k_plans_at_the_benchmark_batch():
    """Host logic of csrc/conv.hip::sk_plan, no GPU needed (ehm_conv_x2_workspace_bytes only plans; without a device the library assumes an MI355X: 256 CUs = 512
    block slots).  At N = 256 images ResNet-50's layers 2 - 4 land just above a multiple of the slots; which convs are cut into K runs, and how much scratch that takes:
      * K loops of >= 64 K tiles with about one round of tiles or less: every tile is cut (layer 3 / 4 3x3 convs, layer 4's 1x1 2048 -> 512);
      * otherwise, one whole round or more + a partly filled one: only the tail is cut, and only where the hand-off (~14 K-tile times) pays: K loops of >= 32 K tiles;
      * everything else runs whole tiles (0 bytes)."""
    import ctypes as C
    from egohmr_amd import _lib
    L = _lib.lib()

    def need(H, Ci, Co, k, stride=1, Ci2=0):
        d = _lib.ConvX2Desc(None, 0, None, None, None, None, 256, H, H, Ci, Co, k, k, stride, k // 2, 1, 256.0, None, 0)
        if Ci2:
            d.x2, d.H2, d.W2, d.Ci2, d.stride2 = 1, 2 * H - 1, 2 * H - 1, Ci2, 2          # (any non-NULL pointer: planning does not touch it)
        return int(L.ehm_conv_x2_workspace_bytes(C.byref(d)))

    flags, part = 4096, 3 * 96 * 256 * 4                      # arrival counters | partial sums per cut tile (three parts of 96 accumulators x 256 threads)
    # every tile cut: 524 tiles x 72 K tiles (layer 3, 3x3), 264 x 144 (layer 4, 3x3), 264 x 64 (layer 4, 1x1 2048 -> 512)
    assert need(14, 256, 256, 3) == flags + 524 * part
    assert need(7, 512, 512, 3) == flags + 264 * part
    assert need(7, 2048, 512, 1) == flags + 264 * part
    # tail only: 524 tiles x 32 K tiles (layer 3, 1x1 1024 -> 256): 12 tiles behind one whole round; 1046 x 36 (layer 2, 3x3): 22 behind two rounds
    assert need(14, 1024, 256, 1) == flags + 12 * part
    assert need(28, 128, 128, 3) == flags + 22 * part
    # the projection shortcut inside layer 4's first block (K = 512 + 1024: 48 K tiles, 1056 tiles): 32 behind two rounds
    assert need(7, 512, 2048, 1, Ci2=1024) == flags + 32 * part
    # whole tiles: short K loops (16 / 18 / 8 K tiles) lose more in the hand-off than the idle slots cost; full rounds have nothing to cut
    for args in ((28, 512, 128, 1), (56, 64, 64, 3), (14, 256, 1024, 1), (7, 512, 2048, 1), (56, 64, 256, 1)):
        assert need(*args) == 0, args
