"""The multi-GPU path is plain env sharding (no data-path collective): every rank owns its own envs, seeds are
disjoint, the bench takes the MAX elapsed over ranks; optionally the observations are all-gathered.  Covered here on
CPU with world_size-2 gloo jobs: the sharding helpers, the observation all-gather, and bench.py's own launcher
(`--gpus 2 --dry`: self-spawned ranks, barrier, reductions, one JSON line with n_gpus = 2 — and loud failures when
the rank count is not what was asked)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["MW_ROOT"])
import torch, torch.distributed as dist
from miniworld_amd.sharding import ObsAllGather, gather_objects, max_over_ranks, shard_plan
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
plan = shard_plan(rank, world, envs_per_rank=8, base_seed=100)
t = max_over_ranks(dist, 1.0 + rank)          # slowest rank defines the step time
plans = gather_objects(dist, plan)
# observation all-gather: rank r's env i shows the value 10 * r + i; two steps through the double buffer
n = plan["num_envs"]
obs = torch.zeros((n, 6, 8, 3), dtype=torch.uint8)
g = ObsAllGather(dist, obs)
sums = []
for step in range(3):
    obs[:] = (torch.arange(n, dtype=torch.uint8) + 10 * rank + step).view(n, 1, 1, 1)
    g.gather(obs)
    obs[:] = 255                                # the engine overwrites its buffer in the next step: the gather took a snapshot
    flat = g.flat()
    assert flat.shape == (world * n, 6, 8, 3)
    want = torch.cat([torch.arange(n) + 10 * r + step for r in range(world)])
    assert torch.equal(flat[:, 0, 0, 0].long(), want), (flat[:, 0, 0, 0], want)
    sums.append(int(flat.long().sum()))
if rank == 0:
    print(json.dumps({"t": t, "plans": plans, "sums": sums}))
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
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["t"] == 2.0
    seeds = [set(range(p["first_seed"], p["first_seed"] + p["num_envs"])) for p in res["plans"]]
    assert seeds[0].isdisjoint(seeds[1]) and len(seeds[0] | seeds[1]) == 16
    assert [p["global_env_offset"] for p in res["plans"]] == [0, 8]
    assert len(res["sums"]) == 3 and res["sums"][1] > res["sums"][0]


def _bench(*argv, env=None):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if env is None:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          env=e, timeout=300, cwd=ROOT)


def test_bench_launcher_dry_run_spawns_the_ranks_it_reports():
    """`bench.py --gpus 2 --dry` with no WORLD_SIZE: the script launches its own two ranks (torch.distributed.run on
    127.0.0.1), shards the envs, reduces the time, gathers per-rank entries and prints n_gpus = 2."""
    out = _bench("--gpus", "2", "--dry", "--steps", "5", "--warmup", "1", "--config", "maze", "--gather-obs")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["envs_per_gpu"] == 1024 and res["config"]["parallelism"] == "env-shard x2"
    assert res["config"]["obs_allgather"] is True
    assert [r["rank"] for r in res["roofline"]["per_rank"]] == [0, 1]
    assert res["value"] > 0 and "dry-run" in res["data"]
    # five timed windows of K steps each by default, `value` the first one's (MAX over the ranks of each)
    w = res["windows"]
    assert w["n"] == 5 and w["steps_each"] == 5 and len(w["values"]) == 5 and w["values"][0] == res["value"]
    assert w["min"] <= w["median"] <= w["max"] and min(w["values"]) > 0


def test_bench_launcher_dry_run_with_eight_ranks():
    """`bench.py --gpus 8 --dry`: the rank count of BASELINE.json's 8-GPU configs through the launcher, the sharding plan, the
    barrier / MAX-reduction / object gather (gloo) — everything of an 8-GPU run but the engine and the wire."""
    out = _bench("--gpus", "8", "--dry", "--steps", "3", "--warmup", "1", "--config", "pickup_dr")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["scaling"] == "weak" and res["config"]["envs_per_gpu"] == 2048
    assert res["config"]["parallelism"] == "env-shard x8" and [r["rank"] for r in res["roofline"]["per_rank"]] == list(range(8))
    assert res["value"] > 0 and "cpu_baseline" not in res and "also" not in res


def test_bench_refuses_a_rank_count_it_was_not_asked_for():
    # torch.distributed.run gave us one rank, the command line says two: no line, non-zero exit
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = _bench("--gpus", "2", "--dry", "--steps", "2", "--warmup", "0", env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # more GPUs asked than the box has (this container has none): refuses before spawning anything
    import torch
    if torch.cuda.device_count() < 2:
        out = _bench("--gpus", "2", "--steps", "2", "--warmup", "0")
        assert out.returncode != 0 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
