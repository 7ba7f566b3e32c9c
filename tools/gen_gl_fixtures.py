"""Generates tests/golden/gl_*.npz: frames of the REFERENCE ITSELF on real OpenGL (FIXTURE TOOLING).

tools/refshim_gl.py runs /root/reference/miniworld unmodified on Mesa llvmpipe (the reference's CI driver family); this
script replays the trajectories of tools/gen_golden.py's cases and stores, for the frames listed there:
  * the neutral scene of that state (tools/refscene.py; what the oracle and the engine render from),
  * render_obs()            -> gl/<k>/rgb   uint8[60,80,3]
  * the resolved depth      -> gl/<k>/z16   uint16[60,80]   (glReadPixels of final_fbo, flipped like get_depth_map)
  * render_depth()          -> gl/<k>/depth float32[60,80,1]
  * render_top_view()       -> gl/<k>/top   uint8[60,80,3]
  * get_visible_ents()      -> gl/<k>/vis   bool[E]
and, for a few cases, render() at 800x600 (vis_fb).  llvmpipe has GL_MAX_SAMPLES = 4, so the reference's frame buffers
fall back to 4 samples (opengl.py:229-231): these are msaa = 4 fixtures.  Driver strings and sample positions are stored
in gl_meta.npz.  While generating, every frame is compared with the oracle and the statistics are printed.

Usage (build container only):  python tools/gen_gl_fixtures.py [case ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import pyoracle  # noqa: E402
import refscene  # noqa: E402
import refshim_gl  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
BIG = {"hallway_s0": [7], "pickup_dr_s1": [100], "sidewalk_s0": [10], "maze_s0": [100]}      # 800x600 render() frames
# --one-spp: the reference's OTHER fallback (opengl.py:263-284) — a driver whose glTexImage2DMultisample fails gets a plain
# single-sampled GL_RGBA texture and a 16-bit depth renderbuffer.  tools/refshim_gl.py makes that call raise
# (MW_REF_FORCE_1SPP), the unmodified FrameBuffer takes its own `except` branch, and these cases are stored as gl1_*.npz.
ONE_SPP_CASES = ["hallway_s0", "pickup_dr_s1", "maze_s0"]
NS, PREFIX = 4, "gl_"


def cases():
    # the case table lives in gen_golden.py; importing it would pull in the GL-stub shim, so parse it instead
    import ast
    src = open(os.path.join(HERE, "gen_golden.py")).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "CASES":
            return ast.literal_eval(node.value)
    raise RuntimeError("CASES not found")


def read_z16(env, fb):
    gl = refshim_gl.gl
    z = np.zeros((fb.height, fb.width), np.uint16)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fb.final_fbo)
    gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, fb.width, fb.height, gl.GL_DEPTH_COMPONENT, gl.GL_UNSIGNED_SHORT, z.ctypes.data)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, 0)
    return np.ascontiguousarray(z[::-1])


def mesh_arrays(env):
    meshes = {}
    for e in env.entities:
        if hasattr(e, "mesh"):
            meshes[refscene.mesh_name_of(e)] = refscene.ref_mesh_arrays(e.mesh)
    return meshes


def capture(env, out, k, stats, big=False):
    sc = refscene.scene_from_ref_env(env)
    meshes = mesh_arrays(env)
    ents = [e for e in env.entities if e is not env.agent]
    rgb = env.render_obs().copy()
    z16 = read_z16(env, env.obs_fb)
    depth = env.render_depth().copy()
    top = env.render_top_view(env.obs_fb).copy()
    visible = env.get_visible_ents()
    vis = np.array([any(e is v for v in visible) for e in ents], bool)
    for key, val in sc.items():
        out[f"gl/{k}/scene/{key}"] = val
    out[f"gl/{k}/rgb"], out[f"gl/{k}/z16"], out[f"gl/{k}/depth"], out[f"gl/{k}/top"], out[f"gl/{k}/vis"] = rgb, z16, depth, top, vis
    # oracle against the driver, right here
    r = pyoracle.render(sc, nsamples=NS, meshes=meshes)
    t = pyoracle.render(sc, nsamples=NS, meshes=meshes, view="top", render_agent=True)
    v = pyoracle.visible_ents(sc, nsamples=NS)
    stats["frames"] += 1
    stats["rgb_bad"] += int((r["rgb"] != rgb).any(axis=2).sum())
    stats["rgb_max"] = max(stats["rgb_max"], int(np.abs(r["rgb"].astype(int) - rgb.astype(int)).max()))
    stats["z_bad"] += int((r["z16"] != z16).sum())
    stats["depth_bad"] += int((r["depth"].view(np.uint32) != depth.view(np.uint32)).sum())
    stats["top_bad"] += int((t["rgb"] != top).any(axis=2).sum())
    stats["top_max"] = max(stats["top_max"], int(np.abs(t["rgb"].astype(int) - top.astype(int)).max()))
    stats["vis_bad"] += int((v != vis).sum())
    if big:
        env.render_mode = "rgb_array"
        for view in ("agent", "top"):
            env.view = view
            img = env.render().copy()
            out[f"gl/{k}/view_{view}"] = img
            rr = pyoracle.render(sc, width=800, height=600, nsamples=NS, meshes=meshes, view=view, render_agent=(view == "top"))
            stats["view_bad"] += int((rr["rgb"] != img).any(axis=2).sum())
        env.view = "agent"


# Beyond the registered ids: rooms whose outline has more than four corners (Room accepts any polygon, miniworld.py:127-176;
# floor and ceiling are then GL_POLYGONs of that many vertices).  A hexagon and a heptagon joined by nothing, a box, the
# reference's own classes and methods throughout.
EXTRA = [("ngon_s0", "NGonRooms", {}, 0, 3, 40, [0, 13, 27, 40]), ("ngon_dr_s2", "NGonRooms", {"domain_rand": True}, 2, 3, 30, [0, 17, 30])]


def make_ngon_env(**kwargs):
    refshim_gl.load_reference()
    from miniworld.entity import Box
    from miniworld.miniworld import MiniWorldEnv

    class NGonRooms(MiniWorldEnv):
        def __init__(self, **kw):
            MiniWorldEnv.__init__(self, max_episode_steps=200, **kw)

        def _gen_world(self):
            def ring(cx, cz, r, n, t0):
                # counter-clockwise seen from above (x east, z south), like add_rect_room's outline (miniworld.py:737-750)
                return np.array([[cx + r * np.cos(t0 + 2 * np.pi * k / n), cz - r * np.sin(t0 + 2 * np.pi * k / n)] for k in range(n)])
            self.add_room(outline=ring(0.0, 0.0, 4.5, 6, 0.3))
            self.add_room(outline=ring(12.0, 1.0, 3.5, 7, 1.1), wall_tex="brick_wall", floor_tex="asphalt", no_ceiling=True)
            self.box = self.place_entity(Box(color="red"), room=self.rooms[0])
            self.box2 = self.place_entity(Box(color="blue", size=0.5), room=self.rooms[1])
            self.place_agent(room=self.rooms[seed_room[0]])

        def step(self, action):
            return MiniWorldEnv.step(self, action)

    seed_room = [0]
    env = NGonRooms(**kwargs)
    env._seed_room = seed_room
    return env


def run_case(name, cls, kwargs, seed, n_actions, steps, frames, totals):
    env = make_ngon_env(**kwargs) if cls == "NGonRooms" else refshim_gl.make_env(cls, **kwargs)
    if cls == "NGonRooms":
        env._seed_room[0] = seed % 2 and 1 or 0        # the agent starts in the hexagon (seed 0) or the heptagon (seed 2 -> 0 ... see EXTRA)
    env.reset(seed=seed)
    rng = np.random.default_rng(1000 + seed)            # the action stream of tools/gen_golden.py
    out = {}
    stats = dict(frames=0, rgb_bad=0, rgb_max=0, z_bad=0, depth_bad=0, top_bad=0, top_max=0, vis_bad=0, view_bad=0)
    done_frames = []
    if 0 in frames:
        capture(env, out, 0, stats, big=0 in BIG.get(name, []))
        done_frames.append(0)
    for t in range(steps):
        if name.startswith("putnext_poke") and t == 40:
            env.red_box.pos = env.yellow_box.pos + np.array([1.0, 0.0, 0.0])
        if isinstance(n_actions, list):
            a = int(rng.choice(len(n_actions), p=n_actions))
        else:
            a = int(rng.integers(0, n_actions))
        obs, rew, term, trunc, info = env.step(a)
        if (t + 1) in frames or (t + 1) in BIG.get(name, []):
            capture(env, out, t + 1, stats, big=(t + 1) in BIG.get(name, []))
            done_frames.append(t + 1)
        if term or trunc:
            break
    out["meta/frames"] = np.array(done_frames, np.int32)
    out["meta/env"] = np.array(cls)
    np.savez_compressed(os.path.join(OUT, PREFIX + name + ".npz"), **out)
    print(f"{name}: {stats}")
    for k, v in stats.items():
        totals[k] = max(totals.get(k, 0), v) if k.endswith("_max") else totals.get(k, 0) + v


def gl_mip_checksums():
    """CRC32 of every mip level of every texture variant of the asset pack, as the driver's glGenerateMipmap built it
    through the reference's own Texture.load (opengl.py:148-184); rows padded with zeros to 12 levels."""
    import zlib
    from ctypes import byref, c_int
    import miniworld.opengl as ogl
    from miniworld.utils import get_file_path
    gl = refshim_gl.gl
    pack = np.load(os.path.join(HERE, "..", "miniworld_amd", "assets", "assets_v1.npz"))
    names, crcs = [], []
    for key in sorted(k for k in pack.files if k.startswith("tex:")):
        name = key[4:]
        if name.startswith("mesh:"):
            path = get_file_path("meshes", name[5:], "png")
        else:
            path = get_file_path("textures", name, "png")
        tex = ogl.Texture.load(path)
        gl.glBindTexture(gl.GL_TEXTURE_2D, tex.id)
        row, lvl = [], 0
        while True:
            w, h = c_int(0), c_int(0)
            gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_WIDTH, byref(w))
            gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_HEIGHT, byref(h))
            if w.value == 0:
                break
            buf = np.zeros((h.value, w.value, 4), np.uint8)
            gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
            gl.glGetTexImage(gl.GL_TEXTURE_2D, lvl, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, buf.ctypes.data)
            row.append(zlib.crc32(np.ascontiguousarray(buf[:, :, :3]).tobytes()))
            lvl += 1
        names.append(name)
        crcs.append(row + [0] * (12 - len(row)))
    return names, crcs


def main():
    global NS, PREFIX
    args = sys.argv[1:]
    if "--one-spp" in args:
        args.remove("--one-spp")
        os.environ["MW_REF_FORCE_1SPP"] = "1"
        NS, PREFIX = 1, "gl1_"
        BIG.clear()
        totals = {}
        for case in cases():
            if case[0] in (set(args) or set(ONE_SPP_CASES)):
                name, cls, kwargs, seed, n_actions, steps, frames = case
                run_case(name, cls, kwargs, seed, n_actions, steps, frames[:3], totals)
        print("TOTAL (1 sample)", totals)
        return
    only = set(args)
    totals = {}
    for case in list(cases()) + EXTRA:
        if not only or case[0] in only:
            run_case(*case, totals)
    info = refshim_gl.driver_info()
    names, crcs = gl_mip_checksums()
    np.savez_compressed(os.path.join(OUT, "gl_meta.npz"), renderer=np.array(info["renderer"]), version=np.array(info["version"]),
                        sample_positions_4=np.array(info["sample_positions_4"], np.float32),
                        mip_names=np.array(names), mip_crc=np.array(crcs, np.int64))
    print("TOTAL", totals, info)


if __name__ == "__main__":
    main()
