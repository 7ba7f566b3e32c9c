"""Multi-GPU scale-out = env sharding: environments never interact (there is no cross-env
state anywhere in the reference's miniworld.py), so rank r of W simply owns envs
[r*n, (r+1)*n) with its own engine, textures and RNG seeds — no data-path collective.
The only cross-rank operations are the timing barrier and a MAX-reduction of elapsed time."""
from __future__ import annotations


def shard_plan(rank: int, world: int, envs_per_rank: int, base_seed: int = 0) -> dict:
    return {"rank": rank, "world": world, "num_envs": envs_per_rank,
            "global_env_offset": rank * envs_per_rank,
            "first_seed": base_seed + rank * envs_per_rank}


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX over ranks of a python float (the slowest rank defines the step time)."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
