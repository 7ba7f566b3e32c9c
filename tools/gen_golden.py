"""Generates tests/golden/*.npz from the reference itself (FIXTURE TOOLING).

Runs the reference's own world generation and dynamics (under tools/refshim.py GL stubs)
for the BASELINE configs, and stores per case:
  * the neutral scene right after reset(seed)                    (keys "s0/<name>")
  * a random-action trajectory of the reference                  (keys "tr/<name>")
  * oracle renders (rgb, z16) of selected trajectory frames      (keys "obs/<k>/rgb|z16")
plus tests/golden/meshes.npz with the per-face arrays produced by the reference's ObjMesh.

The trajectory part pins the oracle's dynamics and the product's world generation to the
reference; the obs part is produced by the CPU oracle (pixels are "parity unpinned", see
oracle/mwo.h) and is what the HIP engine is compared with on the GPU box, where
/root/reference does not exist.

Usage (build container only):  python tools/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import pyoracle  # noqa: E402
import refscene  # noqa: E402
import refshim  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")

# (case name, env class, kwargs, seed, n_actions, steps, frames to render)
CASES = [
    ("hallway_s0", "Hallway", {}, 0, 3, 120, [0, 7, 40, 119]),
    ("hallway_s1", "Hallway", {}, 1, 3, 120, [0, 25]),
    ("hallway_dr_s3", "Hallway", {"domain_rand": True}, 3, 3, 120, [0, 30, 60]),
    ("oneroom_s0", "OneRoom", {}, 0, 3, 100, [0, 33, 99]),
    ("oneroom_dr_s5", "OneRoom", {"domain_rand": True}, 5, 3, 60, [0, 20]),
    ("mazes3_s0", "MazeS3", {}, 0, 3, 150, [0, 50, 149]),
    ("maze_s0", "Maze", {}, 0, 3, 200, [0, 100, 199]),
    ("maze_s2", "Maze", {}, 2, 3, 60, [0, 59]),
    ("maze_dr_s3", "Maze", {"domain_rand": True}, 3, [0.2, 0.2, 0.6], 90, [0, 45, 89]),
    ("mazes3_dr_s1", "MazeS3", {"domain_rand": True}, 1, [0.2, 0.2, 0.6], 120, [0, 119]),
    ("pickup_s0", "PickupObjects", {}, 0, 5, 200, [0, 60, 199]),
    ("pickup_dr_s1", "PickupObjects", {"domain_rand": True}, 1, 5, 300, [0, 100, 299]),
    ("pickup_dr_s4", "PickupObjects", {"domain_rand": True}, 4, 5, 300, [0, 150]),
    # forward-biased policies so that termination / reward / pickup paths are exercised
    ("hallway_fwd_s2", "Hallway", {}, 2, [0.1, 0.1, 0.8], 250, [0, 20]),
    ("hallway_dr_fwd_s4", "Hallway", {"domain_rand": True}, 4, [0.1, 0.1, 0.8], 250, [0, 15]),
    ("oneroom_fwd_s3", "OneRoom", {}, 3, [0.2, 0.1, 0.7], 180, [0, 40]),
    ("oneroom_trunc_s7", "OneRoom", {}, 7, [0.5, 0.5, 0.0], 180, [0, 180]),
    ("pickup_fwd_s2", "PickupObjects", {}, 2, [0.15, 0.1, 0.45, 0.05, 0.25], 400, [0, 80, 200]),
    ("pickup_dr_fwd_s6", "PickupObjects", {"domain_rand": True}, 6, [0.15, 0.1, 0.45, 0.05, 0.25], 400, [0, 120]),
    # section 8(f) rank 4: more env families on the same primitives
    ("fourrooms_s0", "FourRooms", {}, 0, [0.15, 0.15, 0.7], 250, [0, 60, 150]),
    ("fourrooms_dr_s2", "FourRooms", {"domain_rand": True}, 2, 3, 120, [0, 119]),
    ("tmaze_s1", "TMaze", {}, 1, [0.1, 0.1, 0.8], 280, [0, 40]),
    ("tmazeleft_s0", "TMazeLeft", {}, 0, [0.12, 0.08, 0.8], 280, [0, 70]),
    ("putnext_s0", "PutNext", {}, 0, [0.12, 0.1, 0.38, 0.05, 0.2, 0.15, 0.0, 0.0], 250, [0, 100, 249]),
    # "poke": before step 40 the red box is teleported 1 m from the yellow one, so that the
    # reference's success rule (putnext.py:74-78) fires and is pinned
    ("putnext_poke_s1", "PutNext", {}, 1, [0.12, 0.1, 0.38, 0.05, 0.2, 0.15, 0.0, 0.0], 250, [0, 40]),
    ("putnext_dr_s3", "PutNext", {"domain_rand": True}, 3, [0.12, 0.1, 0.38, 0.05, 0.2, 0.15, 0.0, 0.0], 250, [0, 125]),
    ("ymaze_s0", "YMaze", {}, 0, [0.1, 0.1, 0.8], 280, [0, 30, 90]),
    ("ymazeleft_s1", "YMazeLeft", {"domain_rand": True}, 1, [0.12, 0.08, 0.8], 280, [0, 60]),
    # textured static meshes (building, cones), ImageFrame / TextFrame, host-side task rules
    ("wallgap_s0", "WallGap", {}, 0, [0.15, 0.15, 0.7], 300, [0, 12, 60]),
    ("wallgap_dr_s1", "WallGap", {"domain_rand": True}, 1, [0.2, 0.2, 0.6], 120, [0, 80]),
    ("sidewalk_s0", "Sidewalk", {}, 0, [0.12, 0.08, 0.8], 150, [0, 10, 40]),
    ("sidewalk_s3", "Sidewalk", {}, 3, [0.3, 0.1, 0.6], 150, [0, 5]),
    # seed 13: two kits are consumed and respawn (114 steps survived)
    ("collecthealth_s13", "CollectHealth", {}, 13, [0.1, 0.1, 0.45, 0.05, 0.3, 0.0, 0.0, 0.0], 120, [0, 30, 100]),
    ("threerooms_s0", "ThreeRooms", {}, 0, [0.2, 0.2, 0.6], 160, [0, 50, 159]),
    ("threerooms_dr_s2", "ThreeRooms", {"domain_rand": True}, 2, [0.25, 0.15, 0.6], 90, [0, 89]),
    ("sign_s0", "Sign", {}, 0, [0.3, 0.3, 0.4, 0.0], 20, [0, 3, 9]),
    ("sign_green_key_s1", "Sign", {"color_index": 2, "goal": 1}, 1, [0.2, 0.2, 0.55, 0.05], 20, [0, 6]),
    ("roomobjects_s0", "RoomObjects", {}, 0, [0.15, 0.1, 0.4, 0.05, 0.2, 0.1, 0.0, 0.0], 150, [0, 75, 149]),
]


def run_case(name, cls, kwargs, seed, n_actions, steps, frames, meshes):
    env = refshim.make_env(cls, **kwargs)
    log = []
    params = env.params.copy()
    orig = params.sample

    def sample(rng, pname):
        v = orig(rng, pname)
        log.append((pname, v))
        return v
    params.sample = sample
    env.params = params
    env.reset(seed=seed)
    s0 = refscene.scene_from_ref_env(env)
    ents0 = [e for e in env.entities if e is not env.agent]
    for e in ents0:
        if hasattr(e, "mesh"):
            mname = refscene.mesh_name_of(e)
            base = mname.split("_")[0]
            if base not in meshes:
                meshes[base] = refscene.ref_mesh_arrays(e.mesh)
            meshes["kd:" + mname] = meshes.get("kd:" + mname, np.array(
                refscene.ref_mesh_arrays(e.mesh)["colors"][0, 0], np.float64))
    E = len(ents0)
    rng = np.random.default_rng(1000 + seed)
    tr = {k: [] for k in ("action", "pos", "dir", "carrying", "ents_pos", "ents_dir", "ents_alive",
                          "reward", "term", "trunc", "fwd_step", "fwd_drift", "turn_step")}
    scenes = {0: s0}
    poke = np.array([-1, 0, 0, 0, 0], np.float64)
    for t in range(steps):
        if name.startswith("putnext_poke") and t == 40:
            env.red_box.pos = env.yellow_box.pos + np.array([1.0, 0.0, 0.0])
            poke = np.array([t, ents0.index(env.red_box), *env.red_box.pos], np.float64)
        if isinstance(n_actions, list):
            a = int(rng.choice(len(n_actions), p=n_actions))
        else:
            a = int(rng.integers(0, n_actions))
        del log[:]
        obs, rew, term, trunc, info = env.step(a)
        step_params = dict(log[:3])
        tr["action"].append(a)
        tr["pos"].append(np.array(env.agent.pos, np.float64))
        tr["dir"].append(float(env.agent.dir))
        tr["carrying"].append(ents0.index(env.agent.carrying) if env.agent.carrying is not None else -1)
        tr["ents_pos"].append(np.array([e.pos for e in ents0], np.float64).reshape(E, 3))
        tr["ents_dir"].append(np.array([e.dir for e in ents0], np.float64))
        tr["ents_alive"].append(np.array([any(e is x for x in env.entities) for e in ents0], np.int32))
        tr["reward"].append(float(rew))
        tr["term"].append(bool(term))
        tr["trunc"].append(bool(trunc))
        tr["fwd_step"].append(float(step_params["forward_step"]))
        tr["fwd_drift"].append(float(step_params["forward_drift"]))
        tr["turn_step"].append(float(step_params["turn_step"]))
        if (t + 1) in frames:
            sc = refscene.scene_from_ref_env(env)
            # keep the original entity indexing: dead entities get kind 0
            full = dict(s0)
            full.update({k: sc[k] for k in ("agent_pos", "agent_dir", "agent_carrying", "step_count")})
            alive = tr["ents_alive"][-1]
            full["ents_pos"] = tr["ents_pos"][-1]
            full["ents_dir"] = tr["ents_dir"][-1]
            full["ents_kind"] = np.where(alive > 0, s0["ents_kind"], 0).astype(np.int32)
            # CollectHealth re-places consumed kits at the end of self.entities: keep the live order
            scenes[t + 1] = sc if cls == "CollectHealth" else full
        if term or trunc:
            break
    out = {}
    for k, v in s0.items():
        out["s0/" + k] = v
    for k, v in tr.items():
        out["tr/" + k] = np.array(v)
    out["meta/n_actions"] = np.int32(len(n_actions) if isinstance(n_actions, list) else n_actions)
    out["meta/seed"] = np.int32(seed)
    out["meta/domain_rand"] = np.int32(bool(kwargs.get("domain_rand", False)))
    out["meta/env"] = np.array(cls)
    # which layer implements the env's reward / termination rule: "engine" (K1 task rules), "host" (Python on top
    # of the engine's physics: compare poses only at the C level) or "api_only" (the host also moves entities)
    out["meta/rule"] = np.array({"Sidewalk": "host", "Sign": "host", "CollectHealth": "api_only"}.get(cls, "engine"))
    out["meta/kwargs"] = np.array(repr({k: v for k, v in kwargs.items() if k != "domain_rand"}))
    out["meta/poke"] = poke
    goal, goal2 = 0, -1
    if cls == "PutNext":
        goal, goal2 = ents0.index(env.red_box), ents0.index(env.yellow_box)
    out["meta/goal_ent"], out["meta/goal_ent2"] = np.int32(goal), np.int32(goal2)
    out["meta/agent_radius"] = np.float64(env.agent.radius)
    mesh_arrays = {}
    for mname in [str(m) for m in s0["mesh_names"]]:
        base = mname.split("_")[0]
        m = dict(meshes[base])
        m["colors"] = np.broadcast_to(meshes["kd:" + mname].astype(np.float32), m["verts"].shape).copy()
        mesh_arrays[mname] = m
    for k, sc in scenes.items():
        r = pyoracle.render(sc, meshes=mesh_arrays)
        out[f"obs/{k}/rgb"] = r["rgb"]
        out[f"obs/{k}/z16"] = r["z16"]
        # render_top_view(render_agent=True) of the same state, at the observation resolution
        out[f"obs/{k}/top_rgb"] = pyoracle.render(sc, meshes=mesh_arrays, view="top", render_agent=True)["rgb"]
        for key in ("agent_pos", "agent_dir", "ents_pos", "ents_dir", "ents_kind"):
            out[f"obs/{k}/{key}"] = sc[key]
    out["meta/frames"] = np.array(sorted(scenes.keys()), np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: steps={len(tr['action'])} ents={E} polys={len(s0['polys_nv'])} "
          f"segs={len(s0['wall_segs'])} frames={sorted(scenes.keys())} "
          f"reward_sum={sum(tr['reward']):.4f}")


def main():
    os.makedirs(OUT, exist_ok=True)
    meshes = {}
    only = set(sys.argv[1:])               # optional: names of the cases to (re)generate
    for case in CASES:
        if not only or case[0] in only:
            run_case(*case, meshes)
    mout = {}
    mpath = os.path.join(OUT, "meshes.npz")
    if only and os.path.exists(mpath):       # partial run: keep the meshes of the other cases
        old = np.load(mpath)
        mout.update({k: old[k] for k in old.files})
    for k, v in meshes.items():
        if k.startswith("kd:"):
            mout[k] = v
        else:
            for kk, vv in v.items():
                if kk != "colors":
                    mout[f"{k}/{kk}"] = vv
    np.savez_compressed(os.path.join(OUT, "meshes.npz"), **mout)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
