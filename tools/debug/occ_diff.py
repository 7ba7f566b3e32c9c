"""Which of the two (MW_OCCLUSION=0 / 1) frames of a differing env equals the oracle's?  Debug helper."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np, torch
import helpers, pyoracle
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 96
res = {}
for flag in ("0", "1"):
    os.environ["MW_OCCLUSION"] = flag
    vec = MiniWorldVecEnv("MiniWorld-Maze-v0", n, seed=11, want_depth=True, max_episode_steps=40, obs_width=160, obs_height=120)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(42):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    st = vec.engine.get_state()
    sc = helpers.scene_of_vec_env(vec, st, 49)
    want = pyoracle.render(sc, 160, 120, 8, want_prim=True)
    got = vec.obs[49].cpu().numpy()
    d = (got != want["rgb"]).any(-1)
    print("MW_OCCLUSION", flag, "pixels differing from the oracle:", int(d.sum()), np.argwhere(d)[:6].tolist())
    res[flag] = (got, want, sc)
    np.savez(os.path.join(root, "gpurun_out", "occ_scene.npz"), **{k: np.asarray(v) for k, v in sc.items() if not k.startswith("_")})
    vec.close()
g0, g1 = res["0"][0], res["1"][0]
d = (g0 != g1).any(-1)
print("pixels differing between the two runs:", np.argwhere(d).tolist()[:10])
for (y, x) in np.argwhere(d)[:4]:
    want, sc = res["1"][1], res["1"][2]
    print("pixel", y, x, "off", g0[y, x], "on", g1[y, x], "oracle", want["rgb"][y, x], "oracle prim ids of the samples", want["prim"][y, x].tolist() if want["prim"].ndim == 3 else None)
    for p in set(int(v) for v in np.ravel(want["prim"][y, x])):
        if 0 <= p < len(sc["polys_nv"]):
            print("   poly", p, "nv", int(sc["polys_nv"][p]), "verts", np.round(sc["polys_v"][p][:int(sc["polys_nv"][p])], 3).tolist())
print("agent", res["1"][2]["agent_pos"], res["1"][2]["agent_dir"], "cam", st["cam"][49])
