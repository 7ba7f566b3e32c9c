import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import helpers, torch
for case in sys.argv[1:]:
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes))
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    eng.render_top(rgb, None, True); eng.check()
    rgb = rgb.cpu().numpy()
    for i, f in enumerate(frames):
        d = rgb[i].astype(int) - obs[f]["top_rgb"].astype(int)
        ys, xs = np.nonzero(np.abs(d).max(axis=2))
        print(case, f, "ndiff", len(ys), [(int(y), int(x), rgb[i][y, x].tolist(), obs[f]["top_rgb"][y, x].tolist()) for y, x in list(zip(ys, xs))[:5]],
              "kinds", scenes[i]["ents_kind"].tolist(), "agent", scenes[i]["agent_pos"].tolist())
