"""CPU: the LAST stdout line of bench.py is one compact strict-JSON object (the driver parsed round 3's 9 KB line and not
round 4's 24 KB one).  `bench.compact_line` is a pure function of the detail dict, so it is tested here on canned detail objects: the committed
default line of round 4 (24 KB), a worst case with every string blown up and non-finite numbers, and an N > 1 line."""
import copy
import glob
import json
import math
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline")


def _canned():
    with open(os.path.join(REPO, "profiles", "r04e_bench_default_line.json")) as f:
        return json.load(f)


def _strict(line):
    def bad(c):
        raise ValueError(c)
    return json.loads(line, parse_constant=bad)          # NaN / Infinity / -Infinity are not JSON


def test_compact_line_of_the_round4_detail_object():
    d = _canned()
    assert len(json.dumps(d)) > 20000                     # the object the driver did not parse
    line = bench.compact_line(d)
    assert "\n" not in line and len(line) < 4096
    o = _strict(line)
    for k in CONTRACT_KEYS:
        assert k in o, k
    assert o["metric"] == d["metric"] and abs(o["value"] - d["value"]) < 1e-2 * d["value"]
    r = o["roofline"]
    assert set(r) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"}
    assert r["bound"] == "mfma" and r["kernel"].startswith("gcn_hidden_chain_kernel")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # one launch = the 8 chained convs: the launch duration is what rocprofv3's per-kernel average shows
    assert abs(r["avg_launch_ms"] - 8 * d["roofline"]["avg_launch_ms"]) < 1e-2
    assert abs(r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 - r["achieved"]) < 0.01 * r["achieved"]
    c = o["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample"} and c["kind"] in ("port", "reference") and len(c["sample"]) <= 200
    assert set(o["configs"]) == {"c2_ddim10", "c3_guided"}
    for sub in o["configs"].values():
        assert set(sub) >= {"value", "ms_per_step", "roofline_frac"}
    assert o["config"]["name"] == "ddpm100" and "workload" in o["config"]


def test_compact_line_worst_case_still_fits_and_is_strict_json():
    d = _canned()
    d["value"] = float("nan")
    d["roofline"]["traffic"] = float("inf")
    d["dtype"] = "x" * 5000
    d["config"]["workload"] = "w" * 5000
    d["cpu_baseline"]["sample"] = "s" * 5000
    d["roofline"]["kernel"] = "k" * 3000
    for i in range(40):                                   # many sub-configs: optional blocks are dropped until the line fits
        d["configs"][f"extra_{i}"] = copy.deepcopy(d["configs"]["c2_ddim10"])
    line = bench.compact_line(d)
    assert len(line) <= bench.COMPACT_LIMIT
    o = _strict(line)
    assert o["value"] is None and o["roofline"]["traffic"] is None
    for k in CONTRACT_KEYS:
        assert k in o, k


def test_compact_line_multi_gpu_keys():
    d = _canned()
    d.update(n_gpus=8, n_ranks_seen=8, per_rank_bodies_per_s=[2000.123456] * 8, all_gather_ms=0.1234567, cpu_baseline=None)
    d.pop("configs")
    o = _strict(bench.compact_line(d))
    assert o["n_gpus"] == 8 and o["n_ranks_seen"] == 8 and len(o["per_rank_bodies_per_s"]) == 8 and o["all_gather_ms"] is not None
    assert o["cpu_baseline"] is None and len(json.dumps(o)) < 4096


def test_every_committed_round5_line_is_compact():
    """builder-run compact lines kept under profiles/ (r05*_line.json) obey the same bound"""
    for p in glob.glob(os.path.join(REPO, "profiles", "r05*_line.json")):
        with open(p) as f:
            txt = f.read().strip()
        assert len(txt) < 4096, p
        o = _strict(txt)
        assert o["roofline"] and math.isfinite(o["value"]), p
