"""GPU-side debugging aid: engine vs oracle on the golden scenes, mismatch statistics instead of assertions.
usage (GPU box): python tools/gpu_debug_render.py [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import helpers  # noqa: E402
import pyoracle  # noqa: E402
from conftest import golden_cases  # noqa: E402

cases = sys.argv[1:] or golden_cases()
tot = dict(frames=0, rgb_px=0, z_px=0, top_px=0)
for case in cases:
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes), agent_radius=float(meta.get("agent_radius", 0.4)))
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    top = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    eng.render_top(top, None, True)
    try:
        eng.check()
    except Exception as ex:  # noqa
        print(case, "check:", ex)
    rgb, depth, top = rgb.cpu().numpy(), depth.cpu().numpy(), top.cpu().numpy()
    for i, f in enumerate(frames):
        want = obs[f]
        bad = (rgb[i] != want["rgb"]).any(axis=2)
        zbad = depth[i, :, :, 0] != helpers.depth_from_z16(want["z16"])
        tbad = (top[i] != want["top_rgb"]).any(axis=2)
        tot["frames"] += 1; tot["rgb_px"] += int(bad.sum()); tot["z_px"] += int(zbad.sum()); tot["top_px"] += int(tbad.sum())
        d = np.abs(rgb[i].astype(int) - want["rgb"].astype(int))
        print(f"{case} frame {f}: rgb bad px {int(bad.sum())} (max {int(d.max())}), depth bad {int(zbad.sum())}, top bad {int(tbad.sum())}")
        if bad.any() and os.environ.get("SHOW"):
            ys, xs = np.nonzero(bad)
            for y, x in list(zip(ys, xs))[:int(os.environ["SHOW"])]:
                print("    px", y, x, "engine", rgb[i, y, x], "oracle", want["rgb"][y, x])
    if os.environ.get("DUMP"):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "dbg_" + case + ".npz"), rgb=rgb, depth=depth, top=top)
    eng.close()
print("TOTAL", tot)
