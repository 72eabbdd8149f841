"""CPU, world_size 2, gloo: the item sharding and the single end-of-batch gather (egohmr_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egohmr_amd import dist as edist


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 256, 257, 1023):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in edist.shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(edist.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    g = torch.Generator().manual_seed(0)
    p = {"betas": torch.randn(5, 10, generator=g), "global_orient": torch.randn(5, 1, 3, 3, generator=g),
         "body_pose": torch.randn(5, 23, 3, 3, generator=g)}
    packed = edist.pack_params(p)
    assert packed.shape == (5, edist.PACKED_WIDTH)
    q = edist.unpack_params(packed)
    for k in p:
        assert torch.equal(p[k], q[k])


def _worker(rank, world, port, n_items, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = edist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    items = edist.shard_range(n_items, rank, world)
    counts = [len(edist.shard_range(n_items, q, world)) for q in range(world)]
    # stand-in for the sampler: row i of the global result is a deterministic function of the item id
    local = torch.stack([torch.full((edist.PACKED_WIDTH,), float(i)) + torch.arange(edist.PACKED_WIDTH) * 1e-3 for i in items]) \
        if len(items) else torch.zeros(0, edist.PACKED_WIDTH)
    full = edist.gather_packed(local, counts)
    edist.barrier()
    t = edist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    assert t == float(world)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_two_rank_gather_gloo(tmp_path, n_items):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    expect = np.stack([np.full(edist.PACKED_WIDTH, float(i), dtype=np.float32) + np.arange(edist.PACKED_WIDTH, dtype=np.float32) * 1e-3
                       for i in range(n_items)])
    for r in range(2):
        np.testing.assert_allclose(np.load(tmp_path / f"r{r}.npy"), expect, rtol=0, atol=1e-6)


class _FakeSampler:
    """Stands in for FusedSampler: rank r would calibrate k = 10 + 7 r on its own shard."""

    def __init__(self, rank):
        self.rank, self.installed, self.calibrated = rank, None, 0

    def calibrate_schedule(self, diffusion, batch, **kw):
        self.calibrated += 1
        return {"k": 10 + 7 * self.rank, "T": 50, "tol_m": 1e-5, "trials": [], "denom": kw.get("denom_items")}

    def install_schedule(self, diffusion, info, **kw):
        self.installed = dict(info)


def _agree_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    edist.init_from_env("gloo")
    fs = _FakeSampler(rank)
    info = edist.agree_schedule(fs, None, {"img": torch.zeros(6 + rank, 3)}, ddim=True)
    np.save(os.path.join(out_dir, f"k{rank}.npy"), np.array([info["k"], fs.calibrated, -1 if fs.installed is None else fs.installed["k"], info["denom"]]))
    dist.destroy_process_group()


def test_agree_schedule_every_rank_adopts_rank0_calibration(tmp_path):
    """dist.agree_schedule: ONE calibration per job - rank 0 measures on its batch (the first items of the data set under contiguous
    sharding), every rank installs that k; nobody else runs a calibration.  Single-process: it simply calibrates."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "k0.npy"), np.load(tmp_path / "k1.npy")
    assert list(r0) == [10, 1, 10, 6] and list(r1) == [10, 0, 10, 6]          # k, calibrations run here, installed k, rank 0's batch size
    fs = _FakeSampler(3)
    assert edist.agree_schedule(fs, None, {"img": torch.zeros(4, 3)})["k"] == 31 and fs.calibrated == 1 and fs.installed is None
