"""Scene-sharded data parallelism for the sampling path (new in this build; the reference is
single-device, `utils/trainer_utils.py:122-141`).

Every agent is independent inside the denoising loop (GroupNorm is per sample, no cross-agent op
in `dm_model.py:103-142`), so scenes are partitioned over one process per GPU with the weights
replicated and NO collective during the 100 steps.  The only exchange is at a rollout-step
boundary: an all-gather of the decoded trajectories [B_local, 52, 6] so that every rank can build
the next observation (neighbour positions).  `torch.distributed` backend "nccl" is RCCL over xGMI
on ROCm; payloads are <= ~10 MB per rank, i.e. latency-bound, so one fused all-gather per step.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_scenes(n_scenes: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of scene indices: returns [lo, hi) for `rank`.
    The first (n_scenes % world) ranks take one extra scene."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_scenes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_agents(n_scenes: int, agents_per_scene: int, world: int, rank: int) -> Tuple[int, int]:
    """Agent-row range [lo, hi) of this rank when rows are ordered scene-major."""
    lo, hi = shard_scenes(n_scenes, world, rank)
    return lo * agents_per_scene, hi * agents_per_scene


def gather_trajectories(local: torch.Tensor, out: torch.Tensor = None, group=None) -> torch.Tensor:
    """All-gather equal-sized per-rank trajectory blocks [B_local, 52, 6] -> [world * B_local, 52, 6]
    (rank-major = scene-major order under `shard_scenes` with equal shards)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal backend: stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def gather_ragged(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """All-gather per-rank blocks of different row counts (uneven scene split): pads to the largest,
    gathers once, and strips the padding."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    assert len(sizes) == world
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    full = gather_trajectories(pad, group=group)
    return torch.cat([full[r * m: r * m + sizes[r]] for r in range(world)], dim=0)
