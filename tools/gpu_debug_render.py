"""Debug aid for the GPU box: renders the golden frames with the HIP engine and writes
diff statistics + PNGs under gpurun_out/debug/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import helpers  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "debug")
os.makedirs(out, exist_ok=True)
cases = sys.argv[1:] or ["hallway_s0", "oneroom_s0", "mazes3_s0", "maze_s0", "pickup_s0", "pickup_dr_s1"]
for case in cases:
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes))
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    try:
        eng.check()
    except Exception as ex:
        print(case, "check:", ex)
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    for i, f in enumerate(frames):
        d = np.abs(rgb[i].astype(int) - obs[f]["rgb"].astype(int))
        dz = depth[i, :, :, 0] != helpers.depth_from_z16(obs[f]["z16"])
        print(f"{case} frame {f}: rgb max diff {d.max()} n_diff_px {np.count_nonzero(d.max(axis=2))} depth mismatches {np.count_nonzero(dz)}")
        Image.fromarray(rgb[i]).save(os.path.join(out, f"{case}_{f}_hip.png"))
        Image.fromarray(obs[f]["rgb"]).save(os.path.join(out, f"{case}_{f}_oracle.png"))
        if d.max() > 0:
            ys, xs = np.nonzero(d.max(axis=2))
            print("   first diffs (y,x,hip,oracle):", [(int(y), int(x), rgb[i][y, x].tolist(), obs[f]["rgb"][y, x].tolist()) for y, x in list(zip(ys, xs))[:6]])
    eng.close()
