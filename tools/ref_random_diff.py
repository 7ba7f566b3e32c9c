"""Out-of-sample differential check: the CPU oracle against the REFERENCE ITSELF on Mesa llvmpipe (FIXTURE TOOLING / test worker).

Draws (family, seed, steps, domain_rand, view) triples from a seeded generator that is NOT the fixture table of
tools/gen_golden.py, runs /root/reference/miniworld unmodified on the headless GL context (tools/refshim_gl.py) and compares
every frame with oracle/pyoracle.py bit for bit: RGB, the resolved 16-bit depth buffer, and for the agent view the float32
depth map.  The env families are the registered ids of /root/reference/miniworld/envs/__init__.py:44-157.

usage: ref_random_diff.py --cases N --rng-seed S [--one-spp]      (prints one JSON object; exit code 1 on any difference)
Only works where /root/reference and the Mesa swrast driver exist (the build container); tests/test_oracle_vs_reference_gl_random.py
runs it in a process of its own (the GL shim and the stub-GL shim of the other CPU tests cannot share one).
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))

# class name -> number of actions a random policy draws from (the reference's Actions enum: 0-2 move, 3 back, 4 pickup, 5 drop)
FAMILIES = {
    "CollectHealth": 3, "FourRooms": 3, "Hallway": 3, "Maze": 3, "MazeS2": 3, "MazeS3": 3, "MazeS3Fast": 3, "OneRoom": 3,
    "OneRoomS6": 3, "OneRoomS6Fast": 3, "PickupObjects": 5, "PutNext": 6, "RoomObjects": 3, "Sidewalk": 3, "Sign": 3,
    "TMaze": 3, "TMazeLeft": 3, "TMazeRight": 3, "ThreeRooms": 3, "WallGap": 3, "YMaze": 3, "YMazeLeft": 3, "YMazeRight": 3,
}


def draw_cases(n, rng_seed):
    rng = np.random.default_rng(rng_seed)
    names = sorted(FAMILIES)
    out = []
    for i in range(n):
        cls = names[i % len(names)] if i < len(names) else names[int(rng.integers(0, len(names)))]
        c = {"cls": cls, "seed": int(rng.integers(1000, 1000000)), "steps": int(rng.integers(0, 70)),
             "domain_rand": bool(rng.integers(0, 2)), "top": bool(rng.integers(0, 4) == 0)}
        if cls == "Sign":
            c["domain_rand"] = False        # (sign.py:89-96 passes domain_rand=False itself)
        out.append(c)
    return out


def run_case(c, nsamples):
    import pyoracle
    import refscene
    import refshim_gl
    env = refshim_gl.make_env(c["cls"], **({"domain_rand": True} if c["domain_rand"] else {}))
    env.reset(seed=c["seed"])
    rng = np.random.default_rng(c["seed"] + 7)
    for _ in range(c["steps"]):
        _, _, term, trunc, _ = env.step(int(rng.integers(0, FAMILIES[c["cls"]])))
        if term or trunc:
            break
    gl = refshim_gl.gl
    rgb = env.render_top_view(env.obs_fb) if c["top"] else env.render_obs()
    z16 = np.zeros((env.obs_fb.height, env.obs_fb.width), np.uint16)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, env.obs_fb.final_fbo)
    gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, env.obs_fb.width, env.obs_fb.height, gl.GL_DEPTH_COMPONENT, gl.GL_UNSIGNED_SHORT, z16.ctypes.data)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, 0)
    z16 = np.ascontiguousarray(z16[::-1])
    depth = None if c["top"] else env.render_depth()
    sc = refscene.scene_from_ref_env(env)
    meshes = {}
    for e in env.entities:
        if hasattr(e, "mesh"):
            meshes[refscene.mesh_name_of(e)] = refscene.ref_mesh_arrays(e.mesh)
    r = pyoracle.render(sc, nsamples=nsamples, meshes=meshes, view="top" if c["top"] else "agent", render_agent=c["top"])
    res = dict(c)
    res["rgb_bad"] = int(np.count_nonzero(rgb != r["rgb"]))
    res["rgb_max"] = int(np.abs(rgb.astype(int) - r["rgb"].astype(int)).max())
    res["z_bad"] = int(np.count_nonzero(z16 != r["z16"]))
    res["depth_bad"] = 0 if depth is None else int(np.count_nonzero(np.asarray(depth, np.float32).view(np.uint32) != r["depth"].view(np.uint32)))
    res["mean"] = float(rgb.mean())
    env.close() if hasattr(env, "close") else None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--rng-seed", type=int, default=20260930)
    ap.add_argument("--one-spp", action="store_true", help="the reference's single-sampled fallback (opengl.py:263-284)")
    args = ap.parse_args()
    if args.one_spp:
        os.environ["MW_REF_FORCE_1SPP"] = "1"
    import refshim_gl
    if not refshim_gl.gl_available():
        print(json.dumps({"skipped": "needs /root/reference and Mesa's swrast_dri.so"}))
        return 0
    results = [run_case(c, 1 if args.one_spp else 4) for c in draw_cases(args.cases, args.rng_seed)]
    bad = [r for r in results if r["rgb_bad"] or r["z_bad"] or r["depth_bad"]]
    print(json.dumps({"cases": len(results), "bad": bad, "families": sorted({r["cls"] for r in results}),
                      "driver": refshim_gl.gl.gl_info.get_renderer(), "samples": 1 if args.one_spp else 4,
                      "results": results}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
