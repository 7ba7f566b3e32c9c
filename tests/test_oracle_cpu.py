"""CPU tests of the oracle itself (no GPU): pinned against the reference-derived fixtures."""
import math

import numpy as np
import pytest

import helpers
import pyoracle
from conftest import golden_cases


def test_sincos_within_one_ulp_of_libm():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-600, 600, 20000), rng.uniform(-4, 4, 5000),
                         np.arange(-200, 200) * math.pi / 12, [0.0, 1e-300, 1e-8]])
    worst = 0.0
    for x in xs:
        s, c = pyoracle.sincos(float(x))
        for got, want in ((s, math.sin(x)), (c, math.cos(x))):
            if want != 0.0:
                worst = max(worst, abs(got - want) / np.spacing(abs(want)))
            else:
                assert got == 0.0
    assert worst <= 1.0


@pytest.mark.parametrize("case", golden_cases())
def test_dynamics_match_reference_trajectory(case):
    """mwo_step vs the reference's own miniworld.py run under GL stubs (tools/gen_golden.py)."""
    s0, tr, meta, obs = helpers.load_case(case)
    rule = helpers.rule_of(meta)
    if rule == "api_only":
        pytest.skip("the env's Python rule moves entities (CollectHealth respawn): covered through the env API")
    E = len(s0["ents_kind"])
    g0, g1 = helpers.goals_of(meta)
    dyn = pyoracle.Dynamics(s0, helpers.task_of(meta), int(min(float(s0["max_episode_steps"]), 2 ** 30)), goal_ent=g0,
                            goal_ent2=g1, num_objs=E, max_forward_step=float(s0["max_forward_step"]),
                            agent_radius=float(meta.get("agent_radius", 0.4)))
    worst = 0.0
    poke = meta.get("poke", np.array([-1.0]))
    for t in range(len(tr["action"])):
        if int(poke[0]) == t:
            dyn.ents[int(poke[1])].pos[:] = [float(x) for x in poke[2:5]]
        r, te, tu = dyn.step(tr["action"][t], tr["fwd_step"][t], tr["fwd_drift"][t], tr["turn_step"][t])
        if rule == "engine":
            assert r == tr["reward"][t] and te == tr["term"][t], (case, t)
        assert tu == tr["trunc"][t], (case, t)
        assert dyn.ag.carrying == tr["carrying"][t]
        worst = max(worst, np.abs(np.array(dyn.ag.pos[:]) - tr["pos"][t]).max(), abs(dyn.ag.dir - tr["dir"][t]))
        for i in range(E):
            assert dyn.ents[i].alive == tr["ents_alive"][t][i]
            if dyn.ents[i].alive:
                worst = max(worst, np.abs(np.array(dyn.ents[i].pos[:]) - tr["ents_pos"][t][i]).max())
    assert worst < 1e-12


@pytest.mark.parametrize("case", golden_cases())
def test_render_reproduces_committed_golden(case):
    s0, tr, meta, obs = helpers.load_case(case)
    meshes = helpers.golden_meshes(s0)
    for f, fr in obs.items():
        out = pyoracle.render(helpers.frame_scene(s0, fr), meshes=meshes)
        assert np.array_equal(out["rgb"], fr["rgb"]) and np.array_equal(out["z16"], fr["z16"])
        # get_depth_map: the reference's own numpy expression applied to the resolved depth buffer
        assert np.array_equal(out["depth"][:, :, 0], helpers.depth_from_z16(fr["z16"]))
        top = pyoracle.render(helpers.frame_scene(s0, fr), meshes=meshes, view="top", render_agent=True)
        assert np.array_equal(top["rgb"], fr["top_rgb"])


def _empty_scene():
    s0, *_ = helpers.load_case("hallway_s0")
    sc = dict(s0)
    for k in ("polys_v", "polys_uv", "polys_n", "polys_nv", "polys_tex", "polys_rgb"):
        sc[k] = s0[k][:0]
    for k in [k for k in s0 if k.startswith("ents_")]:
        sc[k] = s0[k][:0]
    return sc, s0


def test_sky_only_frame_is_sky_colour():
    sc, _ = _empty_scene()
    sc["sky"] = np.array([0.25, 0.82, 1.0])
    out = pyoracle.render(sc)
    assert (out["rgb"] == np.array([64, 209, 255], np.uint8)).all()
    assert (out["z16"] == 65535).all()
    assert np.allclose(out["depth"], 100.0, rtol=1e-3)                   # far plane


def test_fronto_parallel_white_quad_is_lit_factor_times_255():
    sc, s0 = _empty_scene()
    # a huge untextured wall facing the camera 2 m ahead (+x), normal -x => ambient only: 0.65
    sc["agent_pos"], sc["agent_dir"] = np.array([0.0, 0.0, 0.0]), np.float64(0.0)
    v = np.array([[[2, -50, -50], [2, -50, 50], [2, 50, 50], [2, 50, -50]]], np.float32)
    sc["polys_v"], sc["polys_uv"] = v, np.zeros((1, 4, 2), np.float32)
    sc["polys_n"] = np.array([[-1, 0, 0]], np.float32)
    sc["polys_nv"], sc["polys_tex"] = np.array([4], np.int32), np.array([-1], np.int32)
    sc["polys_rgb"] = np.ones((1, 3), np.float32)
    out = pyoracle.render(sc)
    if (out["z16"] == 65535).all():       # winding the other way round -> culled; flip it
        sc["polys_v"] = v[:, ::-1].copy()
        out = pyoracle.render(sc)
    assert (out["rgb"] == round(0.65 * 255)).all()
    # depth of the centre pixel ~ 2 m (16-bit quantisation)
    assert abs(out["depth"][30, 40, 0] - 2.0) < 0.01


def test_default_light_direction_and_face_factors():
    """Appendix A.3: directional light along light_pos + 1; floor 1.0, ceiling 0.65, walls 0.8354 / 0.65."""
    s0, tr, meta, obs = helpers.load_case("hallway_s0")
    lp = np.array([0, 2.5, 0]) + 1
    L = lp / np.linalg.norm(lp)
    assert np.allclose(L, [0.26490647, 0.92717265, 0.26490647])
    # far-field pixel of the hallway's long walls in the frame looking down the hallway:
    # concrete mean 152.9 * 0.65 ~ 99 / * 0.8354 ~ 128
    sc = dict(s0)
    sc["agent_pos"], sc["agent_dir"] = np.array([0.5, 0.0, 0.3]), np.float64(0.1)
    out = pyoracle.render(sc)
    assert abs(int(out["rgb"][30, 2, 0]) - 128) <= 4       # left wall, normal +z, lit
    assert abs(int(out["rgb"][30, 77, 0]) - 99) <= 4       # right wall, ambient only


def test_mip_chain_even_and_odd():
    rgb = np.arange(6 * 4 * 3, dtype=np.uint8).reshape(4, 6, 3)       # h=4, w=6 -> 3x2 -> 1x1
    lv = pyoracle.mip_levels(rgb)
    assert [l.shape[:2] for l in lv] == [(4, 6), (2, 3), (1, 1)]
    want = (rgb[0::2, 0::2].astype(int) + rgb[0::2, 1::2] + rgb[1::2, 0::2] + rgb[1::2, 1::2] + 2) >> 2
    assert np.array_equal(lv[1], want)
    # odd axis 3 -> 1: equal thirds; even axis 2 -> 1: halves
    a = lv[1].astype(np.int64)
    acc = (a[0] + a[1]).sum(axis=0)
    assert np.array_equal(lv[2][0, 0], (2 * acc + 6) // 12)
    big = pyoracle.mip_levels(pyoracle.texture_rgb_bottom_up("concrete_tiles_1"))    # 768: 768..3,1
    assert [l.shape[0] for l in big] == [768, 384, 192, 96, 48, 24, 12, 6, 3, 1]


def test_oracle_visible_ents_geometry():
    """get_visible_ents restatement (miniworld.py:1238-1333): a box in front of the camera passes its
    query, one behind the agent or behind a wall does not, and a nearer proxy hides a farther one."""
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    env = envs.OneRoom(host_only=True)
    env.reset(seed=0)
    sc = scene_from_env(env)
    sc["agent_pos"], sc["agent_dir"] = np.array([5.0, 0.0, 5.0]), 0.0           # looking along +x
    for pos, want in (([9.0, 0.0, 5.0], True), ([1.0, 0.0, 5.0], False), ([7.0, 0.0, 5.0 + 4.9], False),
                      ([12.0, 0.0, 5.0], False)):                                 # 12 m: beyond the wall at x = 10
        sc["ents_pos"] = np.array([pos])
        assert bool(pyoracle.visible_ents(sc)[0]) == want, pos
    # two entities in line: the far proxy is completely covered by the near one only if it is drawn later
    two = dict(sc)
    for k in ("ents_kind", "ents_mesh", "ents_dir", "ents_size", "ents_color", "ents_scale", "ents_radius", "ents_height", "ents_static"):
        two[k] = np.concatenate([sc[k], sc[k]])
    two["ents_pos"] = np.array([[5.6, 1.4, 5.0], [9.0, 1.4, 5.0]])            # near one first, at eye height
    assert pyoracle.visible_ents(two).tolist() == [True, False]
    two["ents_pos"] = two["ents_pos"][::-1].copy()                              # far one drawn first: both pass
    assert pyoracle.visible_ents(two).tolist() == [True, True]


def test_polygon_fragments_stay_inside_their_vertex_depth_range():
    """llvmpipe's depth plane comes from the (clipped) vertices' window coordinates, so the
    16-bit depth of every fragment lies between the depths of the polygon's nearest and farthest vertex — also for polygons
    seen edge-on (wall stubs a fraction of a pixel wide, far floors a pixel high).  The geometry kernel's occlusion culling
    relies on it (mw_geom.hip)."""
    sc, _ = _empty_scene()
    rng = np.random.default_rng(7)
    W, H, ch, fov = 160, 120, 1.5, 60.0
    t = math.tan(math.radians(fov) / 2)

    def z16_of(w):
        zn, zf = 0.04, 100.0
        return math.floor((0.5 * ((zf + zn) / (zf - zn) - 2 * zf * zn / ((zf - zn) * w)) + 0.5) * 65535 + 0.5)
    checked = thin = 0
    for trial in range(160):
        d = rng.uniform(-math.pi, math.pi)
        F2, S2 = np.array([math.cos(d), -math.sin(d)]), np.array([math.sin(d), math.cos(d)])
        dist = rng.uniform(2, 25)
        c = np.array([10.0, 10.0]) + F2 * dist + S2 * rng.uniform(-0.5, 0.5) * dist
        if trial % 2:
            width, ang = rng.choice([0.25, 0.5, 3.0]), math.radians(10 ** rng.uniform(-2, 1.5))
            view = (c - 10.0) / np.linalg.norm(c - 10.0)
            wd = np.array([view[0] * math.cos(ang) - view[1] * math.sin(ang), view[0] * math.sin(ang) + view[1] * math.cos(ang)])
            A, B = c - wd * width / 2, c + wd * width / 2
            quad = np.array([[A[0], 0, A[1]], [A[0], 2.74, A[1]], [B[0], 2.74, B[1]], [B[0], 0, B[1]]], np.float32)
        else:
            s1, s2, y = rng.choice([0.25, 3.0]), rng.choice([0.25, 3.0]), rng.choice([0.0, 2.74])
            x0, z0 = np.round(c[0]), np.round(c[1])
            quad = np.array([[x0, y, z0], [x0 + s1, y, z0], [x0 + s1, y, z0 + s2], [x0, y, z0 + s2]], np.float32)
        eye, F = np.array([10.0, ch, 10.0]), np.array([F2[0], 0, F2[1]])
        for order in (quad, quad[::-1].copy()):
            w = (order.astype(np.float64) - eye) @ F
            if np.cross(order[1] - order[0], order[2] - order[0]).astype(np.float64) @ (eye - order[0]) <= 0 or (w < 0.2).any():
                continue
            sc["polys_v"], sc["polys_uv"] = order[None], np.zeros((1, 4, 2), np.float32)
            sc["polys_n"], sc["polys_rgb"] = np.array([[0, 1, 0]], np.float32), np.ones((1, 3), np.float32)
            sc["polys_nv"], sc["polys_tex"] = np.array([4], np.int32), np.array([-1], np.int32)
            sc["agent_pos"], sc["agent_dir"] = np.array([10.0, 0.0, 10.0]), np.float64(d)
            sc["cam_height"], sc["cam_fov_y"], sc["cam_pitch"], sc["cam_fwd_disp"] = ch, fov, 0.0, 0.0
            out = pyoracle.render(sc, W, H, 8, want_prim=True)
            own = out["prim"][:, :, 0] == 0
            if not own.any():
                continue
            z = out["z16"][own].astype(int)
            assert z16_of(w.min()) - 1 <= z.min() and z.max() <= z16_of(w.max()) + 1, (trial, z.min(), z.max(), z16_of(w.min()), z16_of(w.max()))
            checked += 1
            ys, xs = np.nonzero(own)
            thin += min(xs.max() - xs.min(), ys.max() - ys.min()) <= 1
    assert checked >= 40 and thin >= 10, (checked, thin)
