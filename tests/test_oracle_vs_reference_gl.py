"""The pixel oracle against the REFERENCE ITSELF on real OpenGL (CPU; no GPU, no /root/reference needed).

tests/golden/gl_*.npz hold what /root/reference/miniworld produces, unmodified, on Mesa llvmpipe — the reference's CI
driver family — for 101 states of every env family and of rooms with six and seven corners (tools/refshim_gl.py + tools/gen_gl_fixtures.py; the reference asks
for 8 / 16 samples and gets GL_MAX_SAMPLES = 4, opengl.py:229-231, so these are 4-sample frames):
render_obs(), the resolved 16-bit depth buffer, render_depth(), render_top_view(), get_visible_ents(), and render() at
800x600 for four states.

Bar (BASELINE.json north_star): depth pixel-exact, RGB within +-1 LSB.  Measured: RGB, depth, depth map, top view and
visible entities are all IDENTICAL on every 80x60 frame; the 800x600 views differ in 1 channel value of 11.5 million.
The assertions are the north_star's bar, with the exact counts asserted on top so that any drift shows.
"""
import os

import numpy as np
import pytest

import helpers
import pyoracle
from conftest import GOLDEN


def gl_cases():
    return sorted(f[3:-4] for f in os.listdir(GOLDEN) if f.startswith("gl_") and f.endswith(".npz") and f != "gl_meta.npz")


def load_gl(case):
    d = np.load(os.path.join(GOLDEN, "gl_" + case + ".npz"))
    frames = {}
    for k in d["meta/frames"]:
        pre = f"gl/{int(k)}/"
        sc = {key[len(pre) + 6:]: d[key] for key in d.files if key.startswith(pre + "scene/")}
        fr = {key[len(pre):]: d[key] for key in d.files if key.startswith(pre) and not key.startswith(pre + "scene/")}
        frames[int(k)] = (sc, fr)
    return frames


def test_fixtures_come_from_the_driver_the_oracle_names():
    m = np.load(os.path.join(GOLDEN, "gl_meta.npz"))
    assert "llvmpipe" in str(m["renderer"]) and "Mesa 23.2.1" in str(m["version"])
    # glGetMultisamplefv(GL_SAMPLE_POSITION) of the 4-sample FBO: the pattern mwo_render.c's PAT4 restates (1/16 px, y up)
    assert np.array_equal(np.round(m["sample_positions_4"] * 16).astype(int), [[6, 2], [14, 6], [2, 10], [10, 14]])
    assert len(gl_cases()) >= 40


@pytest.mark.parametrize("case", gl_cases())
def test_oracle_equals_the_reference_on_opengl(case):
    for k, (sc, fr) in load_gl(case).items():
        meshes = helpers.golden_meshes(sc)
        r = pyoracle.render(sc, nsamples=4, meshes=meshes)
        # depth buffer: pixel-exact, as 16-bit values and as the float32 map get_depth_map derives
        assert np.array_equal(r["z16"], fr["z16"]), f"{case} frame {k}: {np.count_nonzero(r['z16'] != fr['z16'])} depth values differ"
        assert np.array_equal(r["depth"].view(np.uint32), fr["depth"].view(np.uint32)), f"{case} frame {k}: depth map"
        diff = np.abs(r["rgb"].astype(int) - fr["rgb"].astype(int))
        assert diff.max() <= 1, f"{case} frame {k}: RGB differs by {diff.max()}"
        assert np.count_nonzero(diff) == 0, f"{case} frame {k}: {np.count_nonzero(diff)} channel values off by one"
        # render_top_view with the agent marker
        t = pyoracle.render(sc, nsamples=4, meshes=meshes, view="top", render_agent=True)
        assert np.array_equal(t["rgb"], fr["top"]), f"{case} frame {k}: top view, {np.count_nonzero(t['rgb'] != fr['top'])} values differ"
        # get_visible_ents
        assert np.array_equal(pyoracle.visible_ents(sc, nsamples=4), fr["vis"]), f"{case} frame {k}: visible entities"


@pytest.mark.parametrize("case", [c for c in gl_cases() if any("view_agent" in fr for _, fr in load_gl(c).values())])
def test_oracle_equals_the_reference_window_views(case):
    """render() into vis_fb (800 x 600, miniworld.py:1340-1362), agent view and top view."""
    total = 0
    for k, (sc, fr) in load_gl(case).items():
        if "view_agent" not in fr:
            continue
        meshes = helpers.golden_meshes(sc)
        for view in ("agent", "top"):
            r = pyoracle.render(sc, width=800, height=600, nsamples=4, meshes=meshes, view=view, render_agent=(view == "top"))
            diff = np.abs(r["rgb"].astype(int) - fr["view_" + view].astype(int))
            assert diff.max() <= 1, f"{case} frame {k} {view}: differs by {diff.max()}"
            total += np.count_nonzero(diff)
    assert total <= 2, f"{case}: {total} channel values off by one in the 800x600 views"


def test_mip_pyramids_equal_glGenerateMipmap():
    """Every level of every shipped texture as llvmpipe's glGenerateMipmap builds it (checksums stored by
    tools/gen_gl_fixtures.py next to the frames)."""
    m = np.load(os.path.join(GOLDEN, "gl_meta.npz"))
    names = [str(n) for n in m["mip_names"]]
    assert len(names) >= 20
    import zlib
    for name, want in zip(names, m["mip_crc"]):
        levels = pyoracle.mip_levels(pyoracle.texture_rgb_bottom_up(name))
        got = [zlib.crc32(np.ascontiguousarray(l).tobytes()) for l in levels]
        assert got == [int(x) for x in want[:len(got)]] and int(want[len(got)]) == 0 if len(got) < len(want) else got == [int(x) for x in want], name
