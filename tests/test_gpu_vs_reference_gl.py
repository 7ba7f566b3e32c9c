"""The HIP engine against the REFERENCE ITSELF on real OpenGL (GPU box; no /root/reference needed).

tests/golden/gl_*.npz are frames of /root/reference/miniworld, unmodified, on Mesa llvmpipe (see
test_oracle_vs_reference_gl.py): 4-sample frame buffers, because that driver clamps GL_MAX_SAMPLES (opengl.py:229-231).
The engine is configured the same way (msaa = 4) and must reproduce, through the C ABI: render_obs(), the depth map,
render_top_view(), get_visible_ents() and render() at 800x600.

Bar (BASELINE.json north_star): depth pixel-exact, RGB within +-1 LSB.  The engine shares the pinned arithmetic with the
oracle, so the frames are expected to be identical and the count of differing values is asserted too.
"""
import numpy as np
import pytest

import helpers
from test_oracle_vs_reference_gl import gl_cases, load_gl

pytestmark = pytest.mark.gpu


def _groups(frames):
    """Frames of one fixture that share their static geometry (one engine each)."""
    groups = {}
    for k, (sc, fr) in frames.items():
        key = (sc["polys_v"].tobytes(), sc["polys_tex"].tobytes(), sc["polys_uv"].tobytes(), sc["wall_segs"].tobytes(),
               tuple(str(t) for t in sc["tex_names"]), len(sc["ents_kind"]), sc["ents_mesh"].tobytes() if "ents_mesh" in sc else b"")
        groups.setdefault(key, []).append(k)
    return list(groups.values())


@pytest.mark.parametrize("case", gl_cases())
def test_engine_equals_the_reference_on_opengl(case):
    import torch
    frames = load_gl(case)
    off_by_one = 0
    for ks in _groups(frames):
        scenes = [frames[k][0] for k in ks]
        s0 = scenes[0]
        eng = helpers.make_engine_for_scene(s0, len(scenes), agent_radius=float(s0.get("agent_radius", 0.4)), msaa=4)
        eng.set_state(helpers.scene_state_arrays(scenes))
        rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
        depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
        eng.render(rgb, depth)
        top = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
        eng.render_top(top, None, True)
        vis = eng.visible_ents()
        eng.check()
        rgb, depth, top, vis = rgb.cpu().numpy(), depth.cpu().numpy(), top.cpu().numpy(), vis.cpu().numpy()
        for i, k in enumerate(ks):
            fr = frames[k][1]
            # depth: pixel-exact (the float32 map get_depth_map derives from the 16-bit buffer)
            assert np.array_equal(depth[i, :, :, 0].view(np.uint32), fr["depth"].reshape(60, 80).view(np.uint32)), f"{case} frame {k}: depth map"
            assert np.array_equal(depth[i, :, :, 0], helpers.depth_from_z16(fr["z16"])), f"{case} frame {k}: depth buffer"
            diff = np.abs(rgb[i].astype(int) - fr["rgb"].astype(int))
            assert diff.max() <= 1, f"{case} frame {k}: RGB differs by {diff.max()}"
            off_by_one += np.count_nonzero(diff)
            tdiff = np.abs(top[i].astype(int) - fr["top"].astype(int))
            assert tdiff.max() <= 1, f"{case} frame {k}: top view differs by {tdiff.max()}"
            off_by_one += np.count_nonzero(tdiff)
            n = len(fr["vis"])
            assert np.array_equal(vis[i, :n].astype(bool), np.asarray(fr["vis"]).astype(bool)), f"{case} frame {k}: visible entities"
            if "view_agent" in fr:
                for view in ("agent", "top"):
                    out = eng.render_view(i, 800, 600, msaa=4, top=(view == "top"), render_agent=(view == "top")).cpu().numpy()
                    vdiff = np.abs(out.astype(int) - fr["view_" + view].astype(int))
                    assert vdiff.max() <= 1, f"{case} frame {k} {view} view: differs by {vdiff.max()}"
                    assert np.count_nonzero(vdiff) <= 2, f"{case} frame {k} {view} view: {np.count_nonzero(vdiff)} values off by one"
        eng.close()
    assert off_by_one == 0, f"{case}: {off_by_one} channel values off by one"
