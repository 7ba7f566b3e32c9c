"""Multi-GPU scale-out = env sharding, one process per GPU.

Environments never interact (there is no cross-env state anywhere in the reference's miniworld.py; its own
answer to throughput is "multiple processes", README.md:34), so rank r of W owns envs [r*n, (r+1)*n) with its own
engine, textures, meshes and RNG seeds: stepping needs NO data-path collective.  What crosses ranks:

* the timing barrier and a MAX-reduction of the elapsed time (`max_over_ranks`), small host objects
  (`gather_objects`: per-rank kernel timings for the bench line);
* optionally the observations themselves, for a single-process trainer that wants every env's frame on every GPU
  (`ObsAllGather`): one RCCL all-gather of the uint8 [n, H, W, 3] shard per step (14.7 MB per rank for 1024 Maze
  envs, 29.5 MB for 2048 PickupObjects envs, 59 MB for 4096 Hallway envs).  xGMI is point to point (7 links of
  ~153 GB/s per GPU), an all-gather of equal shards sends each shard once over each link, so the step costs about
  shard_bytes / 153 GB/s (0.1 - 0.4 ms) and is issued on its own stream so that it overlaps the next step's raster
  pass; the gathered tensor is double-buffered for that reason.

`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on gloo for the CPU tests
(tests/test_sharding_gloo.py, world size 2).
"""
from __future__ import annotations


def shard_plan(rank: int, world: int, envs_per_rank: int, base_seed: int = 0) -> dict:
    """Which envs (and which reset seeds: env i of the whole job is seeded base_seed + i, like a batch of that size on
    one GPU would be) rank `rank` of `world` owns."""
    if not (0 <= rank < world) or envs_per_rank <= 0:
        raise ValueError(f"bad shard: rank {rank} of {world}, {envs_per_rank} envs per rank")
    return {"rank": rank, "world": world, "num_envs": envs_per_rank,
            "global_env_offset": rank * envs_per_rank,
            "first_seed": base_seed + rank * envs_per_rank}


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX over ranks of a python float (the slowest rank defines the step time)."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(dist, obj) -> list:
    """Every rank's small python object, in rank order, on every rank."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


class ObsAllGather:
    """All-gather of the per-rank observation shards into [world * n, ...] on every rank, overlapped with compute.

    gather(obs) enqueues the collective on a private stream behind the work already queued on the caller's stream
    (i.e. behind the step that produced `obs`) and returns at once; wait() makes the caller's stream wait for it and
    returns the gathered tensor.  Two result buffers alternate, so step t+1 may run (and gather) while the consumer
    still reads step t's result.  The engine writes `obs` in place every step: the shard is snapshotted into a
    staging buffer on the caller's stream first (one device copy of the shard, the price of overlapping)."""

    def __init__(self, dist, shard_like, group=None):
        import torch
        self.dist, self.group, self.torch = dist, group, torch
        self.world = dist.get_world_size(group)
        self.cuda = shard_like.is_cuda
        self.out = [torch.empty((self.world,) + tuple(shard_like.shape), dtype=shard_like.dtype, device=shard_like.device)
                    for _ in range(2)]
        self.stage = [torch.empty_like(shard_like) for _ in range(2)]
        self.stream = torch.cuda.Stream(device=shard_like.device) if self.cuda else None
        self.done = [torch.cuda.Event() for _ in range(2)] if self.cuda else None
        self.k = 0
        self._pending = None
        self._used = [False, False]

    def gather(self, obs):
        torch = self.torch
        k = self.k
        self.k ^= 1
        if self.cuda and self._used[k]:
            # buffer k's previous collective (two gathers ago, on the private stream) must be through before its staging
            # copy is overwritten
            torch.cuda.current_stream(obs.device).wait_event(self.done[k])
        self._used[k] = True
        self.stage[k].copy_(obs, non_blocking=True)
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream(obs.device))
            with torch.cuda.stream(self.stream):
                self.dist.all_gather_into_tensor(self.out[k], self.stage[k], group=self.group)
                self.done[k].record(self.stream)
        else:
            parts = list(self.out[k].unbind(0))
            self.dist.all_gather(parts, self.stage[k], group=self.group)     # gloo: list form, views into the buffer
        self._pending = k
        return self

    def wait(self):
        """[world, n, ...] tensor of the last gather(); the caller's stream is ordered behind the collective."""
        k = self._pending
        if k is None:
            raise RuntimeError("wait() without a gather()")
        if self.cuda:
            self.torch.cuda.current_stream(self.out[k].device).wait_event(self.done[k])
        return self.out[k]

    def flat(self):
        """wait() viewed as [world * n, ...]: global env index = rank * n + local index (shard_plan's offset)."""
        g = self.wait()
        return g.view((g.shape[0] * g.shape[1],) + tuple(g.shape[2:]))
