"""The multi-rank code path on ONE GPU (GPU box): torch.distributed's "nccl" backend is RCCL on ROCm, and a process group of
world size 1 still goes through RCCL's communicator set-up, the collectives' launch on a side stream and the event
hand-over — everything of miniworld_amd.sharding but the wire.  (N > 1 is covered on CPU with gloo, world size 2:
tests/test_sharding_gloo.py; the driver's scaling runs launch bench.py on 2 / 4 / 8 GPUs.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = r"""
import sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from miniworld_amd.sharding import ObsAllGather, gather_objects, max_over_ranks, shard_plan
from miniworld_amd.vec_env import MiniWorldVecEnv
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
plan = shard_plan(0, 1, 64)
vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", 64, seed=plan["first_seed"])
vec.reset()
g = ObsAllGather(dist, vec.obs)
gen = torch.Generator(device="cuda").manual_seed(3)
ok = True
prev = None
for t in range(6):
    vec.step(torch.randint(0, 3, (64,), generator=gen, device="cuda", dtype=torch.int32))
    want = vec.obs.clone()
    g.gather(vec.obs)
    if prev is not None:
        # the previous step's result is still intact while this step's collective is in flight (double buffering)
        ok &= bool((prev[0] == prev[1]).all())
    got = g.flat()
    ok &= tuple(got.shape) == (64, 60, 80, 3) and bool((got == want).all())
    prev = (got, want)
torch.cuda.synchronize()
m = max_over_ranks(dist, 1.25, device="cuda")
objs = gather_objects(dist, {{"rank": 0}})
dist.barrier()
dist.destroy_process_group()
print(json.dumps({{"ok": ok, "max": m, "objs": objs}}))
"""


def test_obs_allgather_over_rccl_world_size_one():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT, port=_free_port())], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])      # (RCCL prints its library path on the way out)
    assert res == {"ok": True, "max": 1.25, "objs": [{"rank": 0}]}


def test_bench_force_dist_runs_the_rank_path_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gather-obs", "--steps", "10", "--warmup", "2",
                          "--envs-per-gpu", "256", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["torch_distributed"] is True and line["config"]["obs_allgather"] is True
    assert line["value"] > 0 and line["parity_checked"] >= 1
    assert np.isfinite(line["ms_per_step"])
