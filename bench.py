#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step+render hot path (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): MiniWorld-Hallway-v0, 4096 batched envs, 80x60 RGB
on one MI355X; synthetic actions uniform{0,1,2}, pre-generated on the device; episodes
auto-reset on the device (same-step).  One "step" = one pass of the hot path over the whole
batch: physics + collision + reward/flags + auto-reset + one rendered observation per env.

Multi-GPU (`--gpus N`): envs shard trivially — one process per GPU, every rank owns its own envs
(4096 Hallway / 4096 OneRoom RGB-D / 1024 Maze / 2048 PickupObjects-DR per GPU, BASELINE.json configs[1..4]), no
data-path collective (weak scaling); only the timing barrier and a MAX-reduce of the elapsed time cross ranks
(RCCL).  Launched by the driver under torch.distributed.run (RANK / WORLD_SIZE in the env), or by itself: when
WORLD_SIZE is unset and N > 1 this script re-executes itself under torch.distributed.run with N ranks on
127.0.0.1.  It never prints a line for fewer ranks than asked: WORLD_SIZE != --gpus, or fewer visible GPUs than
ranks, is an error.  `--dry` runs the same launcher / sharding / reduction path on CPU (gloo) with the engine left out
(tests/test_sharding_gloo.py).

Prints ONE JSON line on rank 0 (see the task contract): value = total env-steps / wall time,
plus `roofline` (algorithmic HBM bytes of the dominant kernel / its HIP-event duration, per rank), at
N=1 `cpu_baseline` (the C oracle on the host cores, bounded sample), and `parity_checked` (frames of the timed
region's last step compared bit for bit with the CPU oracle, after the clock has stopped).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs: name -> (env id, host class, envs per GPU, depth, domain_rand, n_actions, task,
#                                 algorithmic bytes per env-step, dominant kernel)
# hallway = configs[1], the configuration the metric is quoted on; the others via --config
CONFIGS = {
    # obs 14400 + action 4 + agent/episode state r+w 64 + entity 64 + reward/flags 6 (+2 rounding)
    "hallway": ("MiniWorld-Hallway-v0", "Hallway", 4096, False, False, 3, 1, 14540, "mw_rasterq_kernel"),
    "oneroom_rgbd": ("MiniWorld-OneRoom-v0", "OneRoom", 4096, True, False, 3, 1, 33740, "mw_rasterq_kernel"),
    "maze": ("MiniWorld-Maze-v0", "Maze", 1024, False, False, 3, 1, 30860, "mw_raster_big_kernel"),
    "pickup_dr": ("MiniWorld-PickupObjects-v0", "PickupObjects", 2048, False, True, 5, 2, 14800, "mw_rasterq_kernel"),
}
# The kernels between the engine's second and third timing event (`roofline.kernel_ms`: everything of a frame behind the
# geometry kernel).  roofline.traffic is the SUM of their PMC bytes, with the per-kernel breakdown beside it: with mesh
# entities the raster phase is several kernels on two streams, and the dominant one alone would under-report the waste.
RASTER_PHASE = {
    "hallway": ("mw_rasterq_kernel",),
    "oneroom_rgbd": ("mw_rasterq_kernel",),
    "maze": ("mw_raster_big_kernel",),
    "pickup_dr": ("mw_rasterq_kernel", "mw_mesh_entity_kernel", "mw_mesh_slow_kernel", "mw_raster_mesh_kernel"),
}
ALSO = ("oneroom_rgbd", "maze", "pickup_dr")      # BASELINE.json configs[2..4], timed after the headline in the default run
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PREWARM_S = 0.5             # untimed steps before the W warm-up steps: clocks and caches of a cold box (reported)


def _traffic_sum(per_kernel, source, extra=None):
    """per_kernel: {name: (fetch_bytes_raw, write_bytes)} -> the traffic dict of a bench line (FETCH doubled: gfx950)."""
    if not per_kernel:
        return None
    out = {"bytes_per_launch": sum(2.0 * f + w for f, w in per_kernel.values()),
           "fetch_bytes": sum(2.0 * f for f, w in per_kernel.values()), "write_bytes": sum(w for f, w in per_kernel.values()),
           "fetch_bytes_raw_counter": sum(f for f, w in per_kernel.values()),
           "per_kernel": {k: {"bytes": 2.0 * f + w, "fetch_bytes": 2.0 * f, "write_bytes": w} for k, (f, w) in per_kernel.items()},
           "source": source}
    out.update(extra or {})
    return out


def pmc_profiled(kernel, config, n):
    """What the committed rocprofv3 PMC passes say about `kernel` (profiles/<round>/pmc_hbm_summary[_<config>].json,
    pmc_all_summary.json: FETCH_SIZE, WRITE_SIZE and the SQ counters, each collected in its own --pmc pass of this same
    command, means per launch).  These are PROFILED numbers of the commit recorded in the file, not measurements of this
    run — counters cannot be read without the profiler attached; only meaningful for the workload the passes were run on.
    Returns (traffic dict or None, valu dict or None)."""
    if n != CONFIGS[config][2]:
        return None, None
    import glob
    name = "pmc_hbm_summary.json" if config == "hallway" else f"pmc_hbm_summary_{config}.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", name)))
    traffic = valu = None
    if files:
        try:
            j = json.load(open(files[-1]))
            per = {k: (j[k]["FETCH_SIZE_KiB_mean"] * 1024.0, j[k]["WRITE_SIZE_KiB_mean"] * 1024.0) for k in RASTER_PHASE[config] if k in j}
            traffic = _traffic_sum(per, os.path.relpath(files[-1], ROOT) + " (FETCH_SIZE x 2: gfx950)", {"commit": (j.get("_meta") or {}).get("commit")})
        except Exception:  # noqa: BLE001
            traffic = None
    name = "pmc_all_summary.json" if config == "hallway" else f"pmc_all_summary_{config}.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", name)))
    if files:
        try:
            j = json.load(open(files[-1]))
            d = j.get(kernel)
            tiles = n * 75
            valu = {"valu_insts_per_tile": d["SQ_INSTS_VALU_mean"] / tiles, "salu_insts_per_tile": d["SQ_INSTS_SALU_mean"] / tiles,
                    "valu_active_over_wave_cycles": d["SQ_ACTIVE_INST_VALU_mean"] / d["SQ_WAVE_CYCLES_mean"],
                    "wait_over_wave_cycles": d["SQ_WAIT_ANY_mean"] / d["SQ_WAVE_CYCLES_mean"],
                    "source": os.path.relpath(files[-1], ROOT), "commit": (j.get("_meta") or {}).get("commit")}
        except Exception:  # noqa: BLE001
            valu = None
    return traffic, valu


# ------------------------------------------------------------------ CPU baseline (oracle = checker, timed beside the GPU)

def _cpu_worker(seconds, config="hallway"):
    """One host process of the CPU baseline: about `seconds` of env-steps of one env of `config` in the C oracle
    (step + 80x60x8spp render per step); returns (steps, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    _, cls, _, _, dr, n_act, task, _, _ = CONFIGS[config]
    env = getattr(envs, cls)(host_only=True, domain_rand=dr)
    env.reset(seed=0)
    sc = scene_from_env(env)
    meshes = {}
    if len(sc["mesh_names"]):
        from miniworld_amd.objmesh import ObjMesh
        for name in [str(m) for m in sc["mesh_names"]]:
            m = ObjMesh.get(name)
            meshes[name] = {"verts": m.verts, "norms": m.norms, "texcs": m.texcs, "colors": m.colors}
    mes = int(min(float(env.max_episode_steps), 2 ** 30))
    pyoracle.bench_loop(sc, task, mes, n_act, 20, meshes)          # warm-up (page in textures, build mips)
    t = pyoracle.bench_loop(sc, task, mes, n_act, 40, meshes)      # calibration
    steps = int(max(40, min(20000, seconds / max(t / 40, 1e-6))))
    return steps, pyoracle.bench_loop(sc, task, mes, n_act, steps, meshes)


def reference_llvmpipe(config):
    """The REFERENCE ITSELF (/root/reference/miniworld, unmodified, scripts/benchmark.py:22-42's loop) on Mesa llvmpipe: run
    here when the reference tree and the software GL driver exist (the build container), otherwise the figure recorded
    there by the same tool (tools/ref_benchmark.py -> profiles/*/ref_llvmpipe.jsonl), labelled as carried over — a GPU
    box has neither /root/reference nor a GL driver."""
    import glob
    env_id = CONFIGS[config][0]
    name = env_id.split("-")[1]
    if os.path.isdir("/root/reference") and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_benchmark.py"), "--steps", "300", "--envs", name],
                               capture_output=True, text=True, timeout=600)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            return {"value": j["steps_per_s"], "unit": "env-steps/s", "cores": j["host_cores"], "where": "this host, this run",
                    "sample": f"{j['steps']} steps of 1 {env_id} env, the reference on llvmpipe (4 samples: its GL_MAX_SAMPLES)", "driver": j["driver"]["renderer"]}
        except Exception as exc:  # noqa: BLE001
            return {"error": repr(exc)}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "ref_llvmpipe.jsonl")))
    for f in reversed(files):
        for ln in open(f):
            j = json.loads(ln)
            if j["env"] == env_id:
                return {"value": j["steps_per_s"], "unit": "env-steps/s", "cores": j["host_cores"],
                        "where": "carried over: measured in the build container (no /root/reference, no GL driver on this box)",
                        "source": os.path.relpath(f, ROOT),
                        "sample": f"{j['steps']} steps of 1 {env_id} env, the reference on llvmpipe (4 samples: its GL_MAX_SAMPLES)",
                        "driver": j["driver"]["renderer"]}
    return None


# counter passes of the live child runs: HBM bytes (one counter per pass, as MI355X_MICROARCH.md prescribes), then the SQ counters of the
# VALU picture (separate passes too; --pmc with --kernel-trace only)
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"),
              ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32"),
              ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"))
N_SIMD = 256 * 4            # MI355X: 256 CUs x 4 SIMDs


def pmc_live(config, n, dominant=None, kernel_ms=None):
    """Counters of the raster-phase kernels of `config` (RASTER_PHASE), read during this run: short child runs of this script
    under rocprofv3 (--pmc with --kernel-trace only, one pass per counter group: PMC_PASSES) when the profiler is on the box.
    Returns (traffic dict like pmc_profiled's — sum + per-kernel breakdown —, valu dict of the dominant kernel), either may be None."""
    import csv
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    # (not inside a profiler's child: this script's own child runs, or a run somebody else is profiling)
    if prof is None or os.environ.get("MW_BENCH_CHILD") or any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):
        return None, None
    import signal
    out = tempfile.mkdtemp(prefix="mwpmc_", dir="/tmp")
    env = dict(os.environ, MW_BENCH_CHILD="1", TMPDIR="/tmp")
    kernels = RASTER_PHASE[config]
    vals = {c: {k: [] for k in kernels} for grp in PMC_PASSES for c in grp}
    try:
        for gi, grp in enumerate(PMC_PASSES):        # bounded in time
            p = subprocess.Popen([prof, "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, f"pass{gi}"), "-o", "pmc", "--",
                                  sys.executable, os.path.abspath(__file__), "--config", config, "--envs-per-gpu", str(n), "--steps", "6", "--warmup", "2",
                                  "--prewarm-steps", "64", "--windows", "1", "--no-cpu-baseline", "--no-parity-check", "--no-also"], cwd="/tmp", env=env,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=75)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)        # the profiler and the run under it (its own process group)
                p.wait()
                if gi < 2:
                    return None, None
                break                                   # (the traffic passes are in: report what there is)
        for root, _, files in os.walk(out):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    for r in csv.DictReader(open(os.path.join(root, f))):
                        if r["Counter_Name"] in vals:
                            for k in kernels:       # (exact name or its signature: mw_rasterq_kernel must not take mw_rasterq4_kernel's rows)
                                if r["Kernel_Name"] == k or r["Kernel_Name"].startswith(k + "(") or r["Kernel_Name"].startswith(k + " "):
                                    vals[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
        mean = lambda c, k: (sum(vals[c][k]) / len(vals[c][k])) if vals[c][k] else None       # noqa: E731
        per = {}
        for k in kernels:
            fv, wv = mean("FETCH_SIZE", k), mean("WRITE_SIZE", k)
            if fv is not None and wv is not None:           # KiB per launch (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KiB)
                per[k] = (fv * 1024.0, wv * 1024.0)
        if kernels[0] not in per:
            return None, None
        # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes — doubled in the sum
        traffic = _traffic_sum(per, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) --kernel-trace, child runs of this command; "
                                    "FETCH_SIZE x 2 (gfx950); summed over the raster-phase kernels kernel_ms spans",
                               {"launches": len(vals["FETCH_SIZE"][kernels[0]])})
        valu = None
        k = dominant or kernels[0]
        iv = mean("SQ_INSTS_VALU", k) if k in kernels else None
        if iv:
            g = lambda c: mean(c, k)        # noqa: E731
            fp = [g("SQ_INSTS_VALU_FMA_F32"), g("SQ_INSTS_VALU_MUL_F32"), g("SQ_INSTS_VALU_ADD_F32")]
            trans, i32, cvt = g("SQ_INSTS_VALU_TRANS_F32"), g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_VALU_CVT")
            act, wav = g("SQ_ACTIVE_INST_VALU"), g("SQ_WAVE_CYCLES")
            valu = {"kernel": k, "insts_per_launch": iv, "insts_per_env": iv / n, "insts_per_64px_tile": iv / (n * 75),
                    # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD's VALU works on an instruction (1.009 per instruction in
                    # every kernel here, whatever its class); over the kernel's duration at the probe's clock that is the share of
                    # SIMD time the VALU is busy
                    "active_quadcycles_per_launch": act, "wave_quadcycles_per_launch": wav,
                    "active_over_wave_cycles": (act / wav) if act and wav else None,
                    # the classes of tools/ubench/clock_probe.hip: fp32 fma / mul / add issue in 2.4 cycles per wave64 instruction per SIMD,
                    # transcendentals in 8.2, conversions and the integer / compare / select / packed / DPP group in 4.2 (part of the INT32
                    # group — mov, and, add, arithmetic shifts — is full rate too: full_rate_share is a lower bound)
                    "full_rate_share": (sum(fp) / iv) if all(x is not None for x in fp) else None,
                    "trans_share": (trans / iv) if trans is not None else None,
                    "int32_share": (i32 / iv) if i32 is not None else None,
                    "cvt_share": (cvt / iv) if cvt is not None else None,
                    "salu_insts_per_launch": g("SQ_INSTS_SALU"),
                    "source": "rocprofv3 --pmc (SQ counters, one group per pass) --kernel-trace, child runs of this command"}
            if valu["full_rate_share"] is not None and trans is not None and kernel_ms:
                full = sum(fp)
                cyc = (full * 2.4 + trans * 8.2 + (iv - full - trans) * 4.2) / N_SIMD        # cycles per SIMD if the VALU never idled
                valu["model"] = {"cycles_per_simd": cyc, "kernel_ms": kernel_ms, "clock_ghz_for_100pct_valu": cyc / (kernel_ms * 1e-3) / 1e9,
                                 "valu_issue_share_at_2p2_ghz": cyc / (kernel_ms * 1e-3) / 2.2e9,
                                 "note": "issue cycles per SIMD from the class mix (2.4 / 4.2 / 8.2 cycles per wave64 instruction: profiles/r06/clock_probe.txt) "
                                         "/ the kernel's HIP-event duration = the shader clock at which the kernel would be 100 % VALU-issue bound; "
                                         "the probe measures 1.8-2.4 GHz under VALU load"}
        return traffic, valu
    except Exception:  # noqa: BLE001
        return None, None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def cpu_baseline(config):
    """The CPU oracle (oracle/, a port of the reference path) on the host: a bounded sample of
    the same workload — one env stepped + rendered in a C loop on one core (`value`), and the same loop
    in one process per host core at once (`all_cores`: the reference's own answer to throughput is "multiple
    processes", README.md:34)."""
    steps, sec = _cpu_worker(4.0, config)
    env_id = CONFIGS[config][0]
    out = {"value": steps / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} steps of 1 {env_id} env (step + 80x60x8spp render), C oracle, 1 thread"}
    # whole-host figure: independent interpreter processes (nothing shared, like the reference's one-GL-context-per-
    # process recipe), one per host core, bounded in time so that the default run stays within minutes
    n = os.cpu_count() or 1
    if n > 1:
        code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; print(*bench._cpu_worker(4.0, {config!r}))"
        procs = []
        t0 = time.perf_counter()
        try:
            for _ in range(n):
                procs.append(subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE,
                                              stderr=subprocess.DEVNULL, text=True))
            res = [p.communicate(timeout=240)[0].strip().splitlines()[-1].split() for p in procs]
            wall = time.perf_counter() - t0
            # rate of the timed loops themselves (interpreter start-up and texture loading excluded, as in `value`)
            out["all_cores"] = {"value": sum(float(s) / float(t) for s, t in res), "unit": "env-steps/s", "cores": n,
                                "sample": f"{n} processes x ~4 s of steps at once ({wall:.1f} s wall incl. start-up)"}
        except Exception as exc:  # noqa: BLE001 — the single-core figure stands on its own
            for p in procs:
                if p.poll() is None:
                    p.kill()
            out["all_cores"] = {"error": repr(exc)}
    ref = reference_llvmpipe(config)
    if ref is not None:
        out["reference_llvmpipe"] = ref
    return out


# ------------------------------------------------------------------ parity spot-check (after the clock has stopped)

def parity_spot_check(vec, last_actions, config, n_check=8):
    """The frames the timed region's LAST step left in the observation tensor, for n_check envs (first / last block, all
    8 XCD residues), against the CPU oracle's render of the state the device holds; then, without domain randomisation,
    one more step of those envs against the oracle's dynamics.  Returns the number of envs compared; raises on any
    difference.  (PickupObjects removes a picked-up object after the frame was drawn, pickupobjects.py:86-88: envs
    whose last action was `pickup` are skipped for the frame comparison.)"""
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import helpers
    import pyoracle
    _, _, _, depth, dr, n_act, task, _, _ = CONFIGS[config]
    n = vec.num_envs
    la = last_actions.cpu().numpy()
    cand = [i for i in ([0, 1, 2, 3] + [n // 2 + 4, n // 2 + 5] + [n - 2, n - 1] + list(range(8, n, max(1, n // 61))))
            if 0 <= i < n and not (task == 2 and la[i] == 4)]
    pick, seen = [], set()
    for i in cand:                                   # one env per residue i mod 8 first
        if i % 8 not in seen:
            seen.add(i % 8)
            pick.append(i)
    pick = sorted(set(pick + cand[:n_check]))[:max(n_check, len(pick))][:n_check]
    st = vec.engine.get_state()
    meshes = helpers.vec_env_meshes(vec)
    rgb = vec.obs[torch.tensor(pick, device=vec.obs.device)].cpu().numpy()
    dep = vec.depth[torch.tensor(pick, device=vec.obs.device)].cpu().numpy() if depth else None
    scenes = {}
    for j, i in enumerate(pick):
        sc = scenes[i] = helpers.scene_of_vec_env(vec, st, i)
        want = pyoracle.render(sc, meshes=meshes)
        bad = int(np.count_nonzero(rgb[j] != want["rgb"]))
        if bad:
            raise AssertionError(f"parity: env {i}: {bad} RGB values of the timed region's last frame differ from the oracle")
        if depth and not np.array_equal(dep[j], want["depth"]):
            raise AssertionError(f"parity: env {i}: depth differs from the oracle")
    if not dr:
        act = torch.randint(0, n_act, (n,), device=vec.obs.device, dtype=torch.int32)
        _, rew, term, trunc = vec.step(act)
        st2 = vec.engine.get_state()
        a, rew, term, trunc = act.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy()
        for i in pick:
            sc = scenes[i]
            alive = sc["ents_kind"] != 0
            dyn = pyoracle.Dynamics(sc, task, int(vec.engine.cfg.max_episode_steps), num_objs=int(alive.sum()),
                                    max_forward_step=float(vec.template.max_forward_step),
                                    agent_radius=float(vec.template.agent.radius))
            dyn.ag.step_count, dyn.ag.carrying = int(st["step_count"][i]), int(st["carrying"][i])
            dyn.ag.num_picked_up = int(st["num_picked_up"][i])
            for k in range(len(alive)):
                dyn.ents[k].alive = int(alive[k])
            r, te, tr = dyn.step(int(a[i]))
            if not (np.float32(r) == rew[i] and te == bool(term[i]) and tr == bool(trunc[i])):
                raise AssertionError(f"parity: env {i}: step outcome {(rew[i], term[i], trunc[i])} != oracle {(r, te, tr)}")
            if not (te or tr):
                err = max(np.abs(st2["agent_pos"][i] - np.array(dyn.ag.pos[:])).max(), abs(st2["agent_dir"][i] - dyn.ag.dir))
                if not err < 1e-12:
                    raise AssertionError(f"parity: env {i}: pose differs from the oracle by {err}")
    return len(pick)


# ------------------------------------------------------------------ launcher

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """WORLD_SIZE unset and --gpus N > 1: re-execute under torch.distributed.run, one rank per GPU."""
    if not args.dry:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible: refusing to run fewer ranks than asked")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


class _DryVec:
    """--dry: stands in for the engine so that the launcher / sharding / reduction path runs on a CPU-only box."""

    def __init__(self, n):
        import numpy as np
        self.num_envs, self.x = n, np.zeros(n)

    def step(self, _):
        self.x += 1.0


def measure(args, ctx, config, n, steps, warmup, headline):
    """One BASELINE config through the whole protocol: engine of `n` envs per rank, pre-generated random actions, pre-warm,
    W warm-up steps, K timed steps between barriers (MAX over ranks), HIP-event kernel times (one launch in 4 / 8), the parity
    spot check after the clock has stopped, PMC traffic of the raster-phase kernels read by child runs (single GPU).
    Returns the JSON object of the line (the headline's, or an entry of its `also` list) on rank 0, None elsewhere."""
    import torch
    from miniworld_amd.sharding import ObsAllGather, gather_objects, max_over_ranks, shard_plan
    rank, world, local, dist, device = ctx["rank"], ctx["world"], ctx["local"], ctx["dist"], ctx["device"]
    env_id, _, _, want_depth, dr, n_act, _, algo_bytes, dominant = CONFIGS[config]
    plan = shard_plan(rank, world, n)
    if args.dry:
        vec = _DryVec(n)
    else:
        from miniworld_amd.vec_env import MiniWorldVecEnv
        vec = MiniWorldVecEnv(env_id, n, device_id=local, seed=plan["first_seed"], want_depth=want_depth, domain_rand=dr)
        vec.reset()
    total = steps + warmup
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    actions = torch.randint(0, n_act, (total, n), generator=g, device=device, dtype=torch.int32)

    gath = None
    if headline and args.gather_obs and dist is not None:
        gath = ObsAllGather(dist, torch.zeros((n, 60, 80, 3), dtype=torch.uint8) if args.dry else vec.obs)

    def step(t):
        vec.step(actions[t])
        if gath is not None:
            gath.gather(gath.stage[0] if args.dry else vec.obs)

    def sync():
        if not args.dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    # pre-warm: a cold box (fresh lease, idle clocks, cold L2 / instruction caches) is not what the metric describes;
    # untimed and reported, separate from the W warm-up steps of the contract
    t_pre = time.perf_counter()
    k = 0
    while not args.dry and (k < args.prewarm_steps if args.prewarm_steps else time.perf_counter() - t_pre < PREWARM_S):
        for _ in range(16):
            vec.step(actions[k % total])
            k += 1
        sync()
    prewarm_s = time.perf_counter() - t_pre
    for t in range(warmup):
        step(t)
    if not args.dry:
        # HIP-event timing of the kernels on the launch stream: one launch in 4 for short runs, one in 8 otherwise
        # (three event records around a launch cost ~7 us, i.e. ~3 % of the step rate on every launch, measured)
        vec.engine.kernel_time_ms(4 if steps <= 64 else 8)
    barrier()
    t0 = time.perf_counter()
    for t in range(warmup, total):
        step(t)
    if gath is not None:
        gath.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    # more windows of the same K steps behind the contract's one (each between barriers): `value` stays the first window's; the
    # list shows how far a region of a few milliseconds scatters
    window_s = [elapsed]
    for _ in range(max(0, args.windows - 1)):
        barrier()
        tw = time.perf_counter()
        for t in range(warmup, total):
            step(t)
        if gath is not None:
            gath.wait()
        barrier()
        window_s.append(time.perf_counter() - tw)
    raster_ms = setup_ms = 0.0
    launches = parity = 0
    if not args.dry:
        raster_ms, setup_ms, launches = vec.engine.kernel_time_ms(-1)
        vec.engine.check()
        # sanity: the frames are real (a static or empty frame would be an invalid measurement)
        m = float(vec.obs.float().mean())
        assert 1.0 < m < 254.0, f"degenerate observation tensor (mean {m})"
        if not args.no_parity_check:
            parity = parity_spot_check(vec, actions[total - 1], config)
        vec.close()
    del vec

    elapsed_max = max_over_ranks(dist, elapsed, device=device) if dist is not None else elapsed
    window_max = [elapsed_max] + [max_over_ranks(dist, w, device=device) if dist is not None else w for w in window_s[1:]]       # (every rank: a collective)
    achieved = algo_bytes * n / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else None
    mine = {"rank": rank, "device": local, "envs": n, "first_seed": plan["first_seed"], "elapsed_s": elapsed,
            "kernel_ms": raster_ms, "setup_kernel_ms": setup_ms, "launches_timed": launches, "achieved": achieved,
            "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "parity_checked": parity}
    per_rank = gather_objects(dist, mine) if dist is not None else [mine]
    if rank != 0:
        return None
    if len(per_rank) != args.gpus or sorted(r["rank"] for r in per_rank) != list(range(args.gpus)):
        sys.exit(f"bench.py: {len(per_rank)} rank(s) reported, {args.gpus} asked")
    steps_per_s = world * n * steps / elapsed_max
    slow = max(per_rank, key=lambda r: r["kernel_ms"])          # the roofline entry is the slowest rank's
    traffic, valu = pmc_profiled(dominant, config, n)
    # counters read during THIS run when the profiler is on the box (short child runs under rocprofv3), else the committed profile's
    live, valu_live = pmc_live(config, n, dominant, slow["kernel_ms"]) if (world == 1 and not args.dry and not args.no_pmc) else (None, None)
    if live is not None:
        traffic = live
    out = {
        "metric": "env-steps/s (batched, 80x60 RGB)",
        "value": steps_per_s,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed_max / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.dry else "dry-run: no GPU work, launcher / sharding path only",
        "config": {"workload": f"{env_id}, {n} batched envs per GPU, 80x60 RGB{'-D' if want_depth else ''}, 8x MSAA, "
                               f"random actions, {'domain_rand, ' if dr else ''}auto-reset",
                   "name": config, "envs_per_gpu": n, "parallelism": f"env-shard x{world}",
                   "obs_allgather": gath is not None, "torch_distributed": dist is not None},
        "samples_per_s": steps_per_s * 80 * 60 * 8,
        "windows": (lambda v: {"n": len(v), "steps_each": steps, "values": v, "median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v),
                               "note": "env-steps/s of consecutive timed windows of K steps each (rank 0's clock); `value` is the first — the contract's — window"})(
            [world * n * steps / w for w in window_max]),
        "prewarm_s": prewarm_s,
        "parity_checked": sum(r["parity_checked"] for r in per_rank),
        "roofline": {
            "bound": "hbm",
            "kernel": dominant,
            "kernels_timed": list(RASTER_PHASE[config]),
            "achieved": slow["achieved"],
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": slow["frac"],
            "traffic": traffic["bytes_per_launch"] if traffic else None,
            "traffic_profiled": traffic,
            "valu": valu_live,
            "valu_profiled": valu,
            "algorithmic_bytes_per_launch": algo_bytes * n,
            "kernel_ms": slow["kernel_ms"],
            "setup_kernel_ms": slow["setup_kernel_ms"],
            "launches_timed": slow["launches_timed"],
            "per_rank": [{k: r[k] for k in ("rank", "kernel_ms", "setup_kernel_ms", "achieved", "frac", "elapsed_s")}
                         for r in sorted(per_rank, key=lambda r: r["rank"])],
            "traffic_source": "live" if live is not None else ("committed profile" if traffic else None),
            "note": "per GPU; kernel_ms = HIP events around the raster phase of a step (kernels_timed; with mesh entities several kernels "
                    "on two streams), achieved = algorithmic bytes / kernel_ms.  The path is VALU bound (coverage, depth, texture "
                    "filtering, resolve), not HBM bound (SURVEY.md section 8d): see valu_profiled.  traffic = PMC FETCH_SIZE x 2 + "
                    "WRITE_SIZE per launch SUMMED over kernels_timed (traffic_profiled.per_kernel): read during this run by child runs "
                    "under rocprofv3 (traffic_source = live) or, without the profiler, from the committed profile of the named commit",
        },
    }
    return out


def view800(args):
    """Not the headline: `render()` / vis_fb of the reference (miniworld.py:518, 1340-1362: one env, 800 x 600, 16 samples, agent
    view and map view) through mw_render_view — the generic-resolution kernels.  One JSON line of its own."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    out = []
    for env_id in ("MiniWorld-Hallway-v0", "MiniWorld-PickupObjects-v0"):
        vec = MiniWorldVecEnv(env_id, 16, seed=0)
        vec.reset()
        eng = vec.engine
        for top in (False, True):
            for _ in range(3):
                eng.render_view(0, 800, 600, 16, top=top, render_agent=top)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = max(5, args.steps // 10)
            for i in range(k):
                eng.render_view(i % 16, 800, 600, 16, top=top, render_agent=top)
            torch.cuda.synchronize()
            out.append({"env": env_id, "view": "top" if top else "agent", "ms_per_frame": (time.perf_counter() - t0) / k * 1e3, "frames": k})
        vec.close()
    print(json.dumps({"metric": "render() ms per 800x600x16spp frame (one env; not the headline)", "unit": "ms", "higher_is_better": False,
                      "n_gpus": 1, "data": "synthetic", "frames": out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view800", action="store_true", help="time render() at 800 x 600 x 16 samples instead (a line of its own; not the headline)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="default: the BASELINE.json size of the config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--config", default="hallway", choices=list(CONFIGS), help="hallway = the headline workload")
    ap.add_argument("--prewarm-steps", type=int, default=0,
                    help="pre-warm with exactly this many steps instead of PREWARM_S seconds (A/B runs: the timed region then "
                         "covers the same episode phases in both)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra single-GPU configs timed after the headline (the `also` list)")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of K steps each; `value` is the first one's (the contract's), the others show the spread")
    ap.add_argument("--no-pmc", action="store_true", help="skip the child runs under rocprofv3 that read roofline.traffic during the run")
    ap.add_argument("--dry", action="store_true", help="CPU dry run of the multi-rank path (gloo, no engine)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still go through torch.distributed (RCCL, world size 1) — barrier, MAX-reduction, object "
                         "gather and, with --gather-obs, the observation all-gather — so that the multi-rank code path runs on one GPU")
    ap.add_argument("--gather-obs", action="store_true",
                    help="also all-gather every rank's observations onto every rank each step (RCCL over xGMI, overlapped "
                         "with the next step; for a single-process trainer).  Not part of the headline workload")
    args = ap.parse_args()
    if args.view800:
        return view800(args)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])
    n = args.envs_per_gpu or CONFIGS[args.config][2]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line for a "
                 "different number of ranks than asked")
    import torch
    dist = None
    if not args.dry and torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} needs GPU {local} but {torch.cuda.device_count()} are visible")
    if world > 1:
        import torch.distributed as dist
        if args.dry:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.force_dist:
        import torch.distributed as dist
        if not args.dry:
            torch.cuda.set_device(0)
        local = 0
        dist.init_process_group("gloo" if args.dry else "nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                **({} if args.dry else {"device_id": torch.device("cuda", 0)}))
    elif not args.dry:
        torch.cuda.set_device(0)
        local = 0
    device = "cpu" if args.dry else f"cuda:{local}"
    ctx = {"rank": rank, "world": world, "local": local, "dist": dist, "device": device}

    head = measure(args, ctx, args.config, n, args.steps, args.warmup, headline=True)
    # BASELINE.json configs[2..4] in the same command (same protocol, same pre-warm, their BASELINE sizes per GPU; under N
    # ranks every rank runs its shard of each): before the CPU baseline, whose one-process-per-core storm leaves the GPU
    # idle for ~20 s.  Never touches `value`.
    also = []
    if not args.no_also and args.config == "hallway" and not args.envs_per_gpu:
        for name in ALSO:
            try:
                also.append(measure(args, ctx, name, CONFIGS[name][2], args.steps, args.warmup, headline=False))
            except Exception as exc:  # noqa: BLE001 — never at the headline's expense
                if dist is not None:
                    raise                       # (a rank that dropped out of a collective cannot be papered over)
                also.append({"config": {"name": name}, "error": repr(exc)})
    if rank == 0:
        out = head
        if world == 1 and not args.no_cpu_baseline and not args.dry:
            out["cpu_baseline"] = cpu_baseline(args.config)
        if also:
            out["also"] = also
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
