"""Times the REFERENCE ITSELF — /root/reference/miniworld, unmodified — on this container's CPU with Mesa llvmpipe
(BASELINE tooling; build container only: needs /root/reference and the GL loader of tools/refshim_gl.py).

The reference publishes no number for its own scripts/benchmark.py on any hardware, and an MI355X box has
neither /root/reference nor a GL driver, so this is the one place where the reference's step + render_obs loop can be
run at all: the protocol of /root/reference/scripts/benchmark.py:18-41 (reset, then env.step(random action) in a loop,
reset on episode end), here for the BASELINE.json environments, 80x60 observations.  llvmpipe gives the frame buffers 4
samples instead of the 8 the reference asks for (opengl.py:229-231), and rasterises with LP_NUM_THREADS worker threads
(reported).  One process, like the reference.

Usage:  python tools/ref_benchmark.py [--steps 300] [--envs Hallway OneRoom Maze PickupObjects]
Prints one JSON line per environment; the lines are quoted in BASELINE.md section 2.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim_gl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", nargs="*", default=["Hallway", "OneRoom", "Maze", "PickupObjects"])
    args = ap.parse_args()
    if not refshim_gl.gl_available():
        sys.exit("no GL driver / reference tree here: this tool runs in the build container only")
    info = refshim_gl.driver_info()
    for name in args.envs:
        kw = {"domain_rand": True} if name == "PickupObjects" else {}
        env = refshim_gl.make_env(name, **kw)
        env.reset(seed=0)
        n_act = {"PickupObjects": 5}.get(name, 3)
        import numpy as np
        rng = np.random.default_rng(1234)
        acts = rng.integers(0, n_act, args.steps + args.warmup)
        t0 = None
        for t, a in enumerate(acts):
            if t == args.warmup:
                t0 = time.perf_counter()
            obs, rew, term, trunc, _ = env.step(int(a))
            if term or trunc:
                env.reset()
        dt = time.perf_counter() - t0
        print(json.dumps({"env": f"MiniWorld-{name}-v0", "steps": args.steps, "steps_per_s": args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                          "obs": list(obs.shape), "driver": info, "lp_num_threads": os.environ.get("LP_NUM_THREADS", "default (one per core)"),
                          "host_cores": os.cpu_count(), "domain_rand": bool(kw)}), flush=True)
        env.close()


if __name__ == "__main__":
    main()
