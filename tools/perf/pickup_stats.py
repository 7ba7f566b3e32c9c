"""How many PickupObjects envs have a mesh entity in view (what the mesh kernel has to draw)?"""
import sys, math
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 2048
vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", n, domain_rand=True, seed=0)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for t in range(60):
    vec.step(torch.randint(0, 5, (n,), generator=g, device="cuda", dtype=torch.int32))
st = vec.engine.get_state()
kind, pos, geom = st["ent_kind"], st["ent_pos"], st["ent_geom"]
ap, ad, cam = st["agent_pos"], st["agent_dir"], st["cam"]
carry = st["carrying"]
print("carrying:", (carry >= 0).mean(), "carrying a mesh:", np.mean([(c >= 0 and kind[i, c] == 2) for i, c in enumerate(carry)]))
inview = np.zeros(n, int)
for i in range(n):
    d = np.array([math.cos(ad[i]), 0, -math.sin(ad[i])])
    eye = ap[i] + np.array([0, cam[i, 0], 0])
    for k in range(kind.shape[1]):
        if kind[i, k] != 2:
            continue
        v = pos[i, k] + np.array([0, geom[i, k, 8] / 2, 0]) - eye
        dist = np.linalg.norm(v)
        ang = math.degrees(math.acos(np.clip(v @ d / max(dist, 1e-9), -1, 1)))
        if ang < 50 or dist < 1.0:
            inview[i] += 1
print("envs with >=1 mesh roughly in view:", (inview > 0).mean(), "mean meshes in view:", inview.mean())
print("mesh ents per env:", (kind == 2).sum(1).mean(), "alive ents:", (kind != 0).sum(1).mean())
