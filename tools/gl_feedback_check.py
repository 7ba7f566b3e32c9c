"""FIXTURE TOOLING: compares the vertex half of the oracle (oracle/mwo_geom.c) with the driver, bit for bit.

GL feedback mode returns, for every triangle that leaves the draw module's clipper, the window coordinates, clip w,
lit colour and texture coordinates of its three vertices as floats.  The reference's own frame (display list 1 + the
dynamic entities, exactly what MiniWorldEnv._render_world issues) is replayed in feedback mode and matched against the
oracle's triangle stream for the same state.

Usage: python tools/gl_feedback_check.py [EnvClass seed steps [top]] ...   (no arguments: a standard set)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import refshim_gl  # noqa: E402
import refscene  # noqa: E402
import pyoracle  # noqa: E402


def gl_feedback(env, top=False):
    gl = refshim_gl.gl
    if top:
        env.render_top_view(env.obs_fb)
    else:
        env.render_obs()
    env.obs_fb.bind()
    n_max = 4_000_000
    buf = (C.c_float * n_max)()
    gl.glFeedbackBuffer(n_max, gl.GL_4D_COLOR_TEXTURE, buf)
    gl.glRenderMode(gl.GL_FEEDBACK)
    env._render_world(env.obs_fb, render_agent=top)          # the resolve at its end draws nothing
    n = gl.glRenderMode(gl.GL_RENDER)
    assert n >= 0, "feedback buffer overflow"
    a = np.frombuffer(buf, np.float32, n).copy()
    tris, i = [], 0
    while i < n:
        tok = int(a[i]); i += 1
        if tok == gl.GL_POLYGON_TOKEN:
            nv = int(a[i]); i += 1
            assert nv == 3
            tris.append(a[i:i + 36].reshape(3, 12)); i += 36
        elif tok in (gl.GL_BITMAP_TOKEN, gl.GL_DRAW_PIXEL_TOKEN, gl.GL_COPY_PIXEL_TOKEN):
            i += 12
        else:
            raise RuntimeError("unexpected token %x" % tok)
    return np.array(tris, np.float32).reshape(-1, 3, 12)


def oracle_tris(env, top=False):
    sc = refscene.scene_from_ref_env(env)
    meshes = {}
    for e in env.entities:
        if hasattr(e, "mesh"):
            meshes[refscene.mesh_name_of(e)] = refscene.ref_mesh_arrays(e.mesh)
    s, keep = pyoracle.pack_scene(sc, nsamples=4, meshes=meshes, view="top" if top else "agent", render_agent=top)
    L = pyoracle.lib()
    L.mwo_debug_geometry.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    cap = 400000
    buf = np.zeros((cap, 32), np.float32)
    n = L.mwo_debug_geometry(C.byref(s), 0, buf.ctypes.data, cap)
    assert 0 <= n <= cap, n
    return buf[:n]


def compare(env, top=False, verbose=True):
    fb = gl_feedback(env, top)
    orc = oracle_tris(env, top)
    j, skipped, bad, worst = 0, 0, 0, 0.0
    fields = {"xyz": [0, 1, 2], "oow": [3], "rgb": [4, 5, 6], "st": [8, 9]}
    nbad = {k: 0 for k in fields}
    for t in fb:
        # feedback vertex: x y z w r g b a s t r q ; oracle: win[4] (x y z 1/w) col[4] st[2]
        while j < len(orc):
            o = orc[j, :30].reshape(3, 10)
            if np.abs(o[:, :2] - t[:, :2]).max() < 0.05:
                break
            j += 1; skipped += 1
        if j >= len(orc):
            bad += 1
            continue
        o = orc[j, :30].reshape(3, 10); j += 1
        got = np.concatenate([t[:, :3], 1.0 / t[:, 3:4], t[:, 4:8], t[:, 8:10]], axis=1)
        got[:, 3] = t[:, 3]          # GL returns clip w
        exp = o.copy()
        exp[:, 3] = np.float32(1.0) / o[:, 3] if False else o[:, 3]
        for k, idx in fields.items():
            if k == "oow":
                # feedback's w is the clip-space w; the oracle stores 1 / w: compare through the division the driver does
                ok = np.array_equal((np.float32(1.0) / t[:, 3]).astype(np.float32), o[:, 3]) or \
                     np.array_equal(t[:, 3], (np.float32(1.0) / o[:, 3]).astype(np.float32))
            else:
                ok = np.array_equal(got[:, idx].view(np.uint32), exp[:, idx].view(np.uint32))
            if k == "st" and orc[j - 1, 30] < 0:
                ok = True                       # untextured: GL reports the stale current texcoord
            if not ok:
                nbad[k] += 1
                if verbose and sum(nbad.values()) <= 6:
                    print("  mismatch", k, "tri", j - 1, "draw", int(orc[j - 1, 31]))
                    print("    gl ", [float.hex(float(x)) for x in got[:, idx].ravel()])
                    print("    orc", [float.hex(float(x)) for x in exp[:, idx].ravel()])
    return {"gl_tris": len(fb), "oracle_tris": len(orc), "skipped": skipped + len(orc) - j, "unmatched": bad, **nbad}


if __name__ == "__main__":
    cases = [("Hallway", 0, 0), ("Hallway", 1, 17), ("OneRoom", 0, 5), ("Maze", 0, 30), ("PickupObjects", 0, 20),
             ("FourRooms", 0, 40), ("Sidewalk", 0, 10), ("Sign", 0, 2), ("YMaze", 0, 25), ("PutNext", 0, 30)]
    if len(sys.argv) > 1:
        cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
    top = "top" in sys.argv
    for cls, seed, steps in cases:
        env = refshim_gl.make_env(cls)
        env.reset(seed=seed)
        rng = np.random.default_rng(seed)
        for _ in range(steps):
            env.step(int(rng.integers(0, 3)))
        print(cls, seed, steps, compare(env, top))
