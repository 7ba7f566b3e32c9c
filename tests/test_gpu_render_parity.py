"""HIP render_obs vs the CPU oracle on the reference-derived golden scenes (GPU box).

Bar (BASELINE.json north_star): depth buffer pixel-exact, RGB within +-1 LSB; the engine
and the oracle implement the same pinned rules, so RGB is in fact
expected to be bit-exact and the test records how many pixels differ at all.
"""
import numpy as np
import pytest

import helpers
from conftest import golden_cases

pytestmark = pytest.mark.gpu

ALL_CASES = golden_cases()          # Hallway / OneRoom / Maze (quads) and PickupObjects (ball / key meshes)


@pytest.mark.parametrize("case", ALL_CASES)
def test_render_matches_golden_and_oracle(case):
    import torch
    import pyoracle
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes))
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    eng.check()
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    for i, f in enumerate(frames):
        want = pyoracle.render(scenes[i], meshes=helpers.golden_meshes(s0))
        # the committed golden equals the live oracle (same machine-independent arithmetic)
        assert np.array_equal(want["rgb"], obs[f]["rgb"]) and np.array_equal(want["z16"], obs[f]["z16"])
        # depth: exact
        assert np.array_equal(depth[i, :, :, 0], helpers.depth_from_z16(obs[f]["z16"])), f"{case} frame {f}: depth differs"
        assert np.array_equal(depth[i], want["depth"])
        diff = np.abs(rgb[i].astype(int) - obs[f]["rgb"].astype(int))
        assert diff.max() <= 1, f"{case} frame {f}: max RGB diff {diff.max()}"
        assert np.count_nonzero(diff) == 0, f"{case} frame {f}: {np.count_nonzero(diff)} channel values off by one"
    eng.close()


@pytest.mark.parametrize("msaa", [8, 4])
@pytest.mark.parametrize("flags,what", [(4, "every quad through the exact path (packed keys)"), (8, "every quad through the fallback class (all triangles)"),
                                        (0x80, "records hold 8 triangles: the tile code (8 samples) / the scratch-staged exact path (4)")])
@pytest.mark.parametrize("case", ["hallway_s0", "oneroom_s0", "putnext_s0", "pickup_s0"])
def test_quad_kernel_side_paths_equal_the_oracle(case, flags, what, msaa, monkeypatch):
    """mw_rasterq.hip draws most quads through its trivial and painter classes; the exact, fallback and over-capacity paths
    are forced here (MW_DEBUG_FLAGS) and must give the same frames: results do not depend on the class a quad is filed under."""
    import torch
    import pyoracle
    from miniworld_amd import engine as E
    if case not in ALL_CASES:
        pytest.skip("fixture not present")
    monkeypatch.setenv("MW_DEBUG_FLAGS", str(flags))
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)[:3]
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes), msaa=msaa)
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    eng.check()
    if msaa == 8 or len(eng._test_mesh_map) == 0:
        assert eng.raster_path() in (E.PATH_QUAD, E.PATH_QUAD_MESH), eng.raster_path()
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    for i, f in enumerate(frames):
        want = pyoracle.render(scenes[i], nsamples=msaa, meshes=helpers.golden_meshes(s0))
        assert np.array_equal(depth[i], want["depth"]), f"{case} frame {f} ({what}): depth differs"
        assert np.array_equal(rgb[i], want["rgb"]), f"{case} frame {f} ({what}): {np.count_nonzero(rgb[i] != want['rgb'])} RGB values differ"
    eng.close()


@pytest.mark.parametrize("case", ["hallway_s0", "oneroom_s0", "putnext_s0", "pickup_dr_s1", "fourrooms_s0"])
def test_tile_kernels_on_small_scenes_equal_the_oracle(case, monkeypatch):
    """MW_K2Q=0: small scenes through the tile kernels (mw_raster.hip: the painter pass, the pairs classification of up to 32 triangles,
    records staged in LDS) instead of the quad kernel — the quad kernel's A/B baseline, and the code its over-capacity envs and the mesh
    tiles run.  Same frames, with and without a depth channel (two instantiations)."""
    import torch
    import pyoracle
    from miniworld_amd import engine as E
    if case not in ALL_CASES:
        pytest.skip("fixture not present")
    monkeypatch.setenv("MW_K2Q", "0")
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)[:3]
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    meshes = helpers.golden_meshes(s0)
    want = [pyoracle.render(sc, meshes=meshes) for sc in scenes]
    for with_depth in (True, False):
        eng = helpers.make_engine_for_scene(s0, len(scenes))
        eng.set_state(helpers.scene_state_arrays(scenes))
        rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
        depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda") if with_depth else None
        eng.render(rgb, depth)
        eng.check()
        assert eng.raster_path() == E.PATH_TILE, eng.raster_path()
        got = rgb.cpu().numpy()
        for i, f in enumerate(frames):
            assert np.array_equal(got[i], want[i]["rgb"]), f"{case} frame {f} (depth {with_depth}): {np.count_nonzero(got[i] != want[i]['rgb'])} RGB values differ"
            if with_depth:
                assert np.array_equal(depth[i].cpu().numpy(), want[i]["depth"]), f"{case} frame {f}: depth differs"
        eng.close()


# (agent_pos, agent_dir, cam_height, cam_pitch): the Hallway seen from outside and above, with just the tip of one wall
# triangle inside the frustum — display lists of ONE triangle (found with tests/hostcheck's mwhost_list_length)
ONE_TRIANGLE_POSES = [
    ([-23.755122431102166, 0.0, -4.84849794907086], 0.5573117328223667, 7.6530537330811415, 7.006483955274433),
    ([-21.8762906765359, 0.0, -5.551504590760579], 0.523737958532958, 9.054171015388388, 8.643154696824396),
    ([-37.036598647190225, 0.0, -5.126312156465551], 0.5736695910795755, 8.601372821369111, 8.00823826055818),
    ([-6.833958180747146, 0.0, 3.16828247379388], -0.6439531693747176, 5.902227755047577, -23.47320820641254),
    ([-15.00865342721077, 0.0, 0.6401171993273138], 0.8174925421874895, 3.968046259341109, 7.530872200512874),
    ([-10.960166012995469, 0.0, -2.203745795975623], 0.773035741135232, 11.776101639278126, -23.04932772447699),
]


@pytest.mark.parametrize("msaa", [8, 4])
def test_display_list_of_one_triangle(msaa):
    """A frame whose display list holds ONE triangle (a clipped wall quad is a fan of two, so this takes a view from
    outside with one triangle's tip in the frustum and sky around it): the quad kernel's (tile, triangle) index
    arithmetic divides by the list length, and the magic-number division has no 32-bit multiplier for 1 — every tile
    but the first came out as sky."""
    import torch
    import pyoracle
    from miniworld_amd import engine as E
    s0, tr, meta, obs = helpers.load_case("hallway_s0")
    base = helpers.frame_scene(s0, obs[sorted(obs)[0]])
    base["ents_kind"] = np.zeros_like(base["ents_kind"])
    scenes = []
    for pos, d, ch, cp in ONE_TRIANGLE_POSES:
        sc = dict(base)
        sc["agent_pos"], sc["agent_dir"] = np.array(pos), np.float64(d)
        sc["cam_height"], sc["cam_pitch"] = np.float64(ch), np.float64(cp)
        scenes.append(sc)
    eng = helpers.make_engine_for_scene(s0, len(scenes), msaa=msaa)
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    eng.check()
    assert eng.raster_path() == E.PATH_QUAD
    assert eng.list_lengths().tolist() == [1] * len(scenes)
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    shown = 0
    for i in range(len(scenes)):
        want = pyoracle.render(scenes[i], nsamples=msaa)
        assert np.array_equal(rgb[i], want["rgb"]), f"pose {i}: {np.count_nonzero(rgb[i] != want['rgb'])} RGB values differ"
        assert np.array_equal(depth[i], want["depth"]), f"pose {i}: depth differs"
        shown += int((want["z16"] != 65535).sum())
    assert shown > 0          # the triangle does cover pixels somewhere
    eng.close()


@pytest.mark.parametrize("size", [(16, 12), (16, 60), (32, 4), (128, 128)])
@pytest.mark.parametrize("msaa", [8, 4])
def test_observation_sizes_at_the_edges_of_the_tile_kernels(size, msaa):
    """mw_create accepts any multiple of the 16 x 4 tile; up to 128 x 96 pixels the tile / quad kernels draw it: one tile column
    (16 pixels wide: the tile index arithmetic divides by tiles_x = 1), one tile row; 128 x 128 is past their 32-bit edge sums
    (|c| <= 2 W H 2^16: a wall across the whole frame lost a triangle there) and takes the generic-resolution kernels."""
    import torch
    import pyoracle
    W, H = size
    s0, tr, meta, obs = helpers.load_case("hallway_s0")
    frames = sorted(obs)[:3]
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes), msaa=msaa, width=W, height=H)
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), H, W, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), H, W, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    eng.check()
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    for i, f in enumerate(frames):
        want = pyoracle.render(scenes[i], width=W, height=H, nsamples=msaa)
        assert np.array_equal(rgb[i], want["rgb"]), f"{W}x{H} frame {f}: {np.count_nonzero(rgb[i] != want['rgb'])} RGB values differ"
        assert np.array_equal(depth[i], want["depth"]), f"{W}x{H} frame {f}: depth differs"
    eng.close()


@pytest.mark.parametrize("offset", [0, 4, 1])
def test_observation_buffer_of_any_alignment(offset):
    """The C ABI takes any device pointer for the observations: the quad kernel's frame leaves as 16-byte stores when the
    buffer allows it, as dwords or bytes otherwise (a view into a larger allocation) — the same frames each way."""
    import torch
    import pyoracle
    s0, tr, meta, obs = helpers.load_case("hallway_s0")
    frames = sorted(obs)[:3]
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes))
    eng.set_state(helpers.scene_state_arrays(scenes))
    n = len(scenes) * 60 * 80 * 3
    big = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    rgb = big[offset:offset + n].view(len(scenes), 60, 80, 3)
    assert rgb.data_ptr() % 16 == offset
    eng.render(rgb, None)
    eng.check()
    out = rgb.cpu().numpy()
    for i, f in enumerate(frames):
        assert np.array_equal(out[i], pyoracle.render(scenes[i])["rgb"]), (offset, f)
    assert int(big[:offset].sum()) == 0 and int(big[offset + n:].sum()) == 0          # nothing written outside the frames
    eng.close()


@pytest.mark.parametrize("case", ALL_CASES)
def test_top_view_matches_oracle(case):
    """render_top_view (miniworld.py:1088-1175): orthographic map + the agent marker lit by GL's
    stale current normal; HIP == oracle == committed golden, bit for bit."""
    import torch
    s0, tr, meta, obs = helpers.load_case(case)
    frames = sorted(obs)
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes), agent_radius=float(meta.get("agent_radius", 0.4)))
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    eng.render_top(rgb, None, True)
    eng.check()
    rgb = rgb.cpu().numpy()
    for i, f in enumerate(frames):
        assert np.array_equal(rgb[i], obs[f]["top_rgb"]), f"{case} frame {f}: {np.count_nonzero(rgb[i] != obs[f]['top_rgb'])} values differ"
    eng.close()


def test_mesh_without_a_vertex_table_takes_the_per_triangle_vertex_stage():
    """The mesh entity kernel runs the vertex stage once per distinct position (a table in LDS, at most MW_MESH_VCAP = 3584
    positions); a mesh with more keeps no table and every triangle transforms its own three vertices.  Here every ball of
    the PickupObjects fixture is cracked open — each face vertex moved by an offset of its own, 15 576 distinct positions —
    and the frames must still be the oracle's of the same arrays."""
    import torch
    import pyoracle
    s0, tr, meta, obs = helpers.load_case("pickup_s0")
    frames = sorted(obs)[:4]
    scenes = [helpers.frame_scene(s0, obs[f]) for f in frames]
    eng = helpers.make_engine_for_scene(s0, len(scenes))
    meshes = helpers.golden_meshes(s0)
    rng = np.random.default_rng(11)
    cracked = 0
    for i, name in enumerate([str(m) for m in s0["mesh_names"]]):
        m = meshes[name]
        if len(m["verts"]) < 2000:
            continue
        m["verts"] = (m["verts"] + rng.uniform(-2e-3, 2e-3, m["verts"].shape)).astype(np.float32)
        assert len(np.unique(m["verts"].reshape(-1, 3), axis=0)) > 3584
        eng.upload_mesh(eng._test_mesh_map[i], m["verts"], m["norms"], m["texcs"], m["colors"], -1)
        cracked += 1
    assert cracked > 0
    eng.set_state(helpers.scene_state_arrays(scenes))
    rgb = torch.zeros((len(scenes), 60, 80, 3), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((len(scenes), 60, 80, 1), dtype=torch.float32, device="cuda")
    eng.render(rgb, depth)
    eng.check()
    rgb, depth = rgb.cpu().numpy(), depth.cpu().numpy()
    shown = 0
    for i, f in enumerate(frames):
        want = pyoracle.render(scenes[i], meshes=meshes)
        assert np.array_equal(depth[i], want["depth"]), f"frame {f}: depth differs"
        assert np.array_equal(rgb[i], want["rgb"]), f"frame {f}: {np.count_nonzero(rgb[i] != want['rgb'])} RGB values differ"
        shown += int(np.count_nonzero(want["rgb"] != pyoracle.render(scenes[i], meshes=helpers.golden_meshes(s0))["rgb"]))
    assert shown > 0          # a cracked ball is in view somewhere
    eng.close()
