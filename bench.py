#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step+render hot path (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): MiniWorld-Hallway-v0, 4096 batched envs, 80x60 RGB
on one MI355X; synthetic actions uniform{0,1,2}, pre-generated on the device; episodes
auto-reset on the device (same-step).  One "step" = one pass of the hot path over the whole
batch: physics + collision + reward/flags + auto-reset + one rendered observation per env.
For N>1 the driver launches one rank per GPU (torch.distributed.run); envs shard trivially —
every rank owns its own 4096 envs, no data-path collective (weak scaling).

Prints ONE JSON line on rank 0 (see the task contract): value = total env-steps / wall time,
plus `roofline` (algorithmic HBM bytes of the dominant kernel / its HIP-event duration) and, at
N=1, `cpu_baseline` (the C oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ENV_ID = "MiniWorld-Hallway-v0"
# the other BASELINE.json configs, runnable with --config (not the headline line):
#   name -> (env id, envs per GPU, depth, domain_rand, n_actions, algorithmic bytes per env-step)
OTHER_CONFIGS = {
    "oneroom_rgbd": ("MiniWorld-OneRoom-v0", 4096, True, False, 3, 33740),
    "maze": ("MiniWorld-Maze-v0", 1024, False, False, 3, 30860),
    "pickup_dr": ("MiniWorld-PickupObjects-v0", 2048, False, True, 5, 14800),
}
# algorithmic bytes per env-step, RGB, shared geometry:
# obs 14400 + action 4 + agent/episode state r+w 64 + entity 64 + reward/flags 6 (+2 rounding)
ALGO_BYTES_PER_ENV_STEP = 14540
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def pmc_traffic(kernel, config, n):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*/pmc_hbm_summary.json,
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this same command, KiB per launch).
    Only meaningful for the workload the passes were run on (the headline config); None otherwise."""
    if config != "hallway" or n != ENVS_PER_GPU:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_hbm_summary.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1])).get(kernel)
        return (d["FETCH_SIZE_KiB_mean"] + d["WRITE_SIZE_KiB_mean"]) * 1024.0, os.path.relpath(files[-1], ROOT)
    except Exception:  # noqa: BLE001
        return None, None


def _cpu_worker(steps):
    """One host process of the CPU baseline: `steps` env-steps of one Hallway env in the C oracle; returns seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    env = envs.Hallway(host_only=True)
    env.reset(seed=0)
    sc = scene_from_env(env)
    pyoracle.bench_loop(sc, 1, 250, 3, 50)         # warm-up (page in textures, build mips)
    return pyoracle.bench_loop(sc, 1, 250, 3, steps)


def cpu_baseline():
    """The CPU oracle (oracle/, a port of the reference path) on the host: a bounded sample of
    the same workload — one Hallway env stepped + rendered in a C loop on one core (`value`), and the same loop
    in one process per host core at once (`all_cores`: the reference's own answer to throughput is "multiple
    processes", README.md:34)."""
    steps = 4000                                   # ~4 s on one core
    sec = _cpu_worker(steps)
    out = {"value": steps / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} steps of 1 Hallway env (step + 80x60x8spp render), C oracle, 1 thread"}
    # whole-host figure: independent interpreter processes (nothing shared, like the reference's one-GL-context-per-
    # process recipe), bounded in number and in time so that the default run stays within minutes
    import subprocess
    n = min(os.cpu_count() or 1, 32)
    if n > 1:
        code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; print(bench._cpu_worker({steps}))"
        procs = []
        t0 = time.perf_counter()
        try:
            for _ in range(n):
                procs.append(subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE,
                                              stderr=subprocess.DEVNULL, text=True))
            secs = [float(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
            wall = time.perf_counter() - t0
            # rate of the timed loops themselves (interpreter start-up and texture loading excluded, as in `value`)
            out["all_cores"] = {"value": sum(steps / s_ for s_ in secs), "unit": "env-steps/s", "cores": n,
                                "sample": f"{n} processes x {steps} steps at once ({wall:.1f} s wall incl. start-up)"}
        except Exception as exc:  # noqa: BLE001 — the single-core figure stands on its own
            for p in procs:
                if p.poll() is None:
                    p.kill()
            out["all_cores"] = {"error": repr(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="hallway", help="hallway (headline) | " + " | ".join(OTHER_CONFIGS))
    args = ap.parse_args()
    env_id, want_depth, dr, n_act, algo_bytes = ENV_ID, False, False, 3, ALGO_BYTES_PER_ENV_STEP
    if args.config != "hallway":
        env_id, n_default, want_depth, dr, n_act, algo_bytes = OTHER_CONFIGS[args.config]
        if args.envs_per_gpu == ENVS_PER_GPU:
            args.envs_per_gpu = n_default
        args.no_cpu_baseline = True

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0

    from miniworld_amd.sharding import max_over_ranks, shard_plan
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = args.envs_per_gpu
    plan = shard_plan(rank, world, n)
    vec = MiniWorldVecEnv(env_id, n, device_id=local, seed=plan["first_seed"], want_depth=want_depth, domain_rand=dr)
    vec.reset()
    total = args.steps + args.warmup
    g = torch.Generator(device=f"cuda:{local}").manual_seed(1234 + rank)
    actions = torch.randint(0, n_act, (total, n), generator=g, device=f"cuda:{local}", dtype=torch.int32)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for t in range(args.warmup):
        vec.step(actions[t])
    vec.engine.kernel_time_ms()         # enable + clear the HIP-event timing of the kernels
    barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        vec.step(actions[t])
    barrier()
    elapsed = time.perf_counter() - t0
    raster_ms, setup_ms, launches = vec.engine.kernel_time_ms()
    vec.engine.check()
    # sanity: the frames are real (a static or empty frame would be an invalid measurement)
    m = float(vec.obs.float().mean())
    assert 1.0 < m < 254.0, f"degenerate observation tensor (mean {m})"

    if dist is not None:
        elapsed = max_over_ranks(dist, elapsed, device=f"cuda:{local}")
    if rank == 0:
        steps_per_s = world * n * args.steps / elapsed
        achieved = algo_bytes * n / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else None
        dominant = "mw_raster_mesh_kernel" if vec.mesh_ids else ("mw_raster_depth_kernel" if want_depth else "mw_raster_kernel")
        if env_id == "MiniWorld-Maze-v0":
            dominant = "mw_raster_big_kernel"
        traffic, traffic_src = pmc_traffic(dominant, args.config, n)
        out = {
            "metric": "env-steps/s (batched, 80x60 RGB)",
            "value": steps_per_s,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{env_id}, {n} batched envs per GPU, 80x60 RGB{'-D' if want_depth else ''}, 8x MSAA, "
                                   f"random actions, {'domain_rand, ' if dr else ''}auto-reset",
                       "envs_per_gpu": n, "parallelism": f"env-shard x{world}"},
            "samples_per_s": steps_per_s * 80 * 60 * 8,
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes * n,
                "kernel_ms": raster_ms,
                "setup_kernel_ms": setup_ms,
                "launches_timed": launches,      # one launch in 8 is bracketed with HIP events (mwengine.h)
                "note": "path is raster/texture VALU bound, not HBM bound (SURVEY.md section 8d); traffic = PMC "
                        "FETCH_SIZE + WRITE_SIZE per launch, bytes",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
