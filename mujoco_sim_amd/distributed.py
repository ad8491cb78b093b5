"""Multi-GPU sharding of the environment batch: one process per GPU, no data-path collective.

Every env is an independent unit, so global env ``i`` lives on rank
``i // (N / world)`` and keeps its global seed ``base_seed + i`` — results are identical for
any world size. The only collective is an optional gather of a rollout block
(``[T, N/world, ...] -> [T, N, ...]``) once per rollout chunk: RCCL (backend "nccl") over xGMI
on GPUs, gloo in the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(num_envs_global: int, world_size: int, rank: int) -> tuple[int, int]:
    """[start, stop) of the global env indices owned by ``rank`` (contiguous, balanced)."""
    if num_envs_global % world_size != 0:
        raise ValueError(f"num_envs ({num_envs_global}) must be divisible by world_size ({world_size})")
    per = num_envs_global // world_size
    return rank * per, (rank + 1) * per


def init_process_group(backend: str | None = None):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_rollout(block: torch.Tensor, env_dim: int = 1) -> torch.Tensor:
    """All-gather per-rank rollout blocks along the env axis: [T, n_local, ...] -> [T, n_global, ...]."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return block
    world = dist.get_world_size()
    moved = block.movedim(env_dim, 0).contiguous()  # [n_local, T, ...]
    out = torch.empty((world * moved.shape[0],) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
    dist.all_gather_into_tensor(out, moved)
    return out.movedim(0, env_dim)


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
