"""CPU: `python bench.py --gpus N` starts its N ranks itself (it used to run on one rank and print n_gpus: 1), and
refuses a rank count that does not match --gpus.  `--launch-check` stops after the rendezvous, so no GPU is needed (gloo)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env["EGOHMR_DIST_BACKEND"] = "gloo"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=600)


def test_bench_gpus_2_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["backend"] == "gloo"
    assert "launching 2 ranks" in r.stderr


def test_bench_gpus_8_launch_check():
    """the shape of the driver's SCALE run: eight ranks rendezvous over 127.0.0.1 and agree on the world size"""
    r = _run(["--gpus", "8", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8


def test_bench_refuses_mismatched_world_size():
    r = _run(["--gpus", "2", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2 but 1 rank" in (r.stderr + r.stdout)
