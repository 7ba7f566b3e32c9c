"""The multi-GPU path is plain env sharding (no data-path collective): every rank owns its own
envs, seeds are disjoint, and the bench takes the MAX elapsed over ranks.  Covered here with a
world_size-2 gloo job on CPU."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["MW_ROOT"])
import torch, torch.distributed as dist
from miniworld_amd.sharding import shard_plan, max_over_ranks
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
plan = shard_plan(rank, world, envs_per_rank=8, base_seed=100)
t = max_over_ranks(dist, 1.0 + rank)          # slowest rank defines the step time
gathered = [None] * world
dist.all_gather_object(gathered, plan)
if rank == 0:
    print(json.dumps({"t": t, "plans": gathered}))
dist.destroy_process_group()
"""


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MW_ROOT=ROOT)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
        capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["t"] == 2.0
    seeds = [set(range(p["first_seed"], p["first_seed"] + p["num_envs"])) for p in res["plans"]]
    assert seeds[0].isdisjoint(seeds[1]) and len(seeds[0] | seeds[1]) == 16
    assert [p["global_env_offset"] for p in res["plans"]] == [0, 8]
