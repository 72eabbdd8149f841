"""Multi-GPU layer: one process per GPU, items sharded, ONE collective at the end.

Every (image, scene, sample) item is independent in sampling - the reference itself
is single-process (test_egohmr.py:92) and only suggests "running multiple jobs" (README.md:154-156).  So:
contiguous blocks of items per rank, weights and SMPL constants replicated, no per-step communication,
and a single all-gather (RCCL over xGMI when the backend is "nccl") of the packed result rows
    [betas 10 | global_orient 9 | body_pose 207] = 226 float32 per body  (what test_egohmr.py:261-266 collects).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

PACKED_WIDTH = 226


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises torch.distributed when launched by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("EGOHMR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split (first n_items % world ranks get one extra item)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def pack_params(pred_smpl_params: dict) -> torch.Tensor:
    """{'betas' [B,10], 'global_orient' [B,1,3,3], 'body_pose' [B,23,3,3]} -> [B,226]."""
    B = pred_smpl_params["betas"].shape[0]
    return torch.cat([pred_smpl_params["betas"].reshape(B, 10), pred_smpl_params["global_orient"].reshape(B, 9),
                      pred_smpl_params["body_pose"].reshape(B, 207)], dim=1).contiguous()


def unpack_params(packed: torch.Tensor) -> dict:
    B = packed.shape[0]
    return {"betas": packed[:, :10], "global_orient": packed[:, 10:19].reshape(B, 1, 3, 3),
            "body_pose": packed[:, 19:].reshape(B, 23, 3, 3)}


def gather_packed(packed: torch.Tensor, counts: list[int] | None = None) -> torch.Tensor:
    """All ranks receive every rank's rows in rank order.  ``counts`` = rows per rank when ragged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return packed
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and packed.is_cuda:             # functional multi-rank runs on one GPU box: stage through the host
        return gather_packed(packed.cpu(), counts).to(packed.device)
    if counts is None or len(set(counts)) == 1:
        out = torch.empty(world * packed.shape[0], packed.shape[1], dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(out, packed)
        return out
    width, cap = packed.shape[1], max(counts)                       # ragged: pad to the largest shard, one collective
    padded = torch.zeros(cap, width, dtype=packed.dtype, device=packed.device)
    padded[: packed.shape[0]] = packed
    out = torch.empty(world * cap, width, dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * cap: r * cap + counts[r]] for r in range(world)], dim=0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device) -> list:
    """Every rank's scalar, in rank order (bench.py: per-rank throughput in the N > 1 line)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def agree_schedule(fused_sampler, diffusion, batch, ddim=False, guided=False, cond_grad_weight=1.0, denom_items=None, **kw):
    """One precision-schedule calibration for the whole job: rank 0 measures k on ITS batch (FusedSampler.calibrate_schedule - with the
    contiguous sharding of `shard_range` these are the first items of the data set whatever the world size), every rank installs that k.
    Without this each rank would calibrate on its own shard at first use and the result rows would depend on how the items were
    sharded.  Single-process: just calibrates.  Returns the info dict."""
    B = int(denom_items or next(v for v in batch.values() if torch.is_tensor(v)).shape[0])
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    info = None
    if not multi or dist.get_rank() == 0:
        try:
            info = fused_sampler.calibrate_schedule(diffusion, batch, ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, denom_items=B, **kw)
        except Exception as e:                      # the other ranks are waiting in the broadcast below: tell them instead of leaving them there
            if not multi:
                raise
            info = {"error": f"{type(e).__name__}: {e}"}
    if multi:
        box = [info]
        dist.broadcast_object_list(box, src=0)
        info = box[0]
        if "error" in info:
            raise RuntimeError(f"agree_schedule: rank 0's calibration failed: {info['error']}")
        fused_sampler.install_schedule(diffusion, info, ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, denom_items=B)
    return info
