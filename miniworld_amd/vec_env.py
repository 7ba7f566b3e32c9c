"""MiniWorldVecEnv — N environments stepped and rendered in lockstep on one MI355X.

This is the performance path: world state lives on the device as Structure-of-Arrays, one
``step()`` is three kernel launches (step, geometry, raster; two more with mesh entities) that write the
``uint8[N,60,80,3]`` observation tensor (and optionally ``float32[N,60,80,1]`` depth) straight into torch memory.
Episodes auto-reset on the device (same-step semantics: the observation returned together
with ``terminated|truncated`` is the first one of the next episode; the reference leaves the
reset to the caller, scripts/benchmark.py:36-37).

All 23 env ids are generated, ruled and auto-reset on the device, on the reference's own numpy PCG64 stream: Hallway,
OneRoom*, Maze* and PickupObjects through their own generators, the fixed-floorplan families through placement programs
(`genprog.py`).  `host_generate()` (the host world generator + `mw_set_state`) remains as an injection API.
"""
from __future__ import annotations

import math

import numpy as np

from . import engine as eng
from . import envs as _envs
from .scene import base_config, polys_array, scene_from_env, state_arrays, upload_scene_meshes

_KIND = {
    "MiniWorld-Hallway-v0": ("Hallway", eng.GEN_HALLWAY, eng.TASK_GOTO, 3),
    "MiniWorld-OneRoom-v0": ("OneRoom", eng.GEN_ONEROOM, eng.TASK_GOTO, 3),
    "MiniWorld-OneRoomS6-v0": ("OneRoomS6", eng.GEN_ONEROOM, eng.TASK_GOTO, 3),
    "MiniWorld-OneRoomS6Fast-v0": ("OneRoomS6Fast", eng.GEN_ONEROOM, eng.TASK_GOTO, 3),     # S6 + its own step / turn sizes
    "MiniWorld-Maze-v0": ("Maze", eng.GEN_MAZE, eng.TASK_GOTO, 3),
    "MiniWorld-MazeS2-v0": ("MazeS2", eng.GEN_MAZE, eng.TASK_GOTO, 3),
    "MiniWorld-MazeS3-v0": ("MazeS3", eng.GEN_MAZE, eng.TASK_GOTO, 3),
    "MiniWorld-MazeS3Fast-v0": ("MazeS3Fast", eng.GEN_MAZE, eng.TASK_GOTO, 3),
    "MiniWorld-PickupObjects-v0": ("PickupObjects", eng.GEN_PICKUP, eng.TASK_PICKUP, 5),
    # fixed floorplans: _gen_world compiled into a placement program the device generator runs (genprog.py)
    "MiniWorld-FourRooms-v0": ("FourRooms", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-TMaze-v0": ("TMaze", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-TMazeLeft-v0": ("TMazeLeft", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-TMazeRight-v0": ("TMazeRight", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-YMaze-v0": ("YMaze", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-YMazeLeft-v0": ("YMazeLeft", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-YMazeRight-v0": ("YMazeRight", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-WallGap-v0": ("WallGap", eng.GEN_PROGRAM, eng.TASK_GOTO, 3),
    "MiniWorld-ThreeRooms-v0": ("ThreeRooms", eng.GEN_PROGRAM, eng.TASK_NONE, 3),
    # the forbidden street / the touch table + end-of-episode action are K1 task rules fed by the program's tables
    "MiniWorld-Sidewalk-v0": ("Sidewalk", eng.GEN_PROGRAM, eng.TASK_SIDEWALK, 3),
    "MiniWorld-Sign-v0": ("Sign", eng.GEN_PROGRAM, eng.TASK_SIGN, 4),
    "MiniWorld-PutNext-v0": ("PutNext", eng.GEN_PROGRAM, eng.TASK_PUTNEXT, 8),
    "MiniWorld-RoomObjects-v0": ("RoomObjects", eng.GEN_PROGRAM, eng.TASK_NONE, 8),
    # health bookkeeping and the respawn of a consumed kit (place_entity at the END of the entity list,
    # collecthealth.py:79-98) are a K1 task rule too
    "MiniWorld-CollectHealth-v0": ("CollectHealth", eng.GEN_PROGRAM, eng.TASK_COLLECT, 8),
}


class MiniWorldVecEnv:
    def __init__(self, env_id: str, num_envs: int, device_id: int = 0, domain_rand: bool = False,
                 want_depth: bool = False, seed: int = 0, autoreset: bool = True, obs_layout: str = "hwc",
                 rng: str = "auto", msaa: int = 8, **env_kwargs):
        """obs_layout: "hwc" uint8[N,H,W,3] (the env's observation), "cwh" uint8[N,3,W,H]
        (PyTorchObsWrapper, wrappers.py:24) or "grey" float64[N,H,W,1] (GreyscaleWrapper, wrappers.py:44):
        the raster kernel stores the frame in that layout, there is no extra pass.
        msaa: samples per pixel, 8 like the reference's FrameBuffer(80, 60, 8) (miniworld.py:515); 4 or 1 reproduce what the
        reference renders on a driver that clamps GL_MAX_SAMPLES (opengl.py:229-231) — same semantics, not the tuned path.
        rng: stream of the device-side resets. "pcg64" = numpy's own Generator(PCG64(SeedSequence(seed + i))) drawn in
        the reference's call order, so that env i IS the reference's env.reset(seed=seed + i) and its later episodes
        continue like env.reset(), per-step domain-randomisation draws included (every device generator);
        "philox" = the engine's counter-based stream; "auto" = pcg64 where implemented."""
        import torch
        self.torch = torch
        if obs_layout not in ("hwc", "cwh", "grey"):
            raise ValueError(f"obs_layout must be 'hwc', 'cwh' or 'grey', not {obs_layout!r}")
        self.obs_layout = obs_layout
        if env_id not in _KIND:
            raise KeyError(f"{env_id!r} is not available in the batched engine yet; have {sorted(_KIND)}")
        cls_name, generator, task, n_actions = _KIND[env_id]
        if cls_name == "Sign":
            domain_rand = False             # sign.py:92-98 fixes it
        self.env_id, self.num_envs, self.n_actions = env_id, num_envs, n_actions
        self.domain_rand, self.want_depth = domain_rand, want_depth
        self.generator = generator
        cls = getattr(_envs, cls_name)
        # template world: geometry, textures, capacities (host-side world generation only)
        # (Sign fixes domain_rand=False itself and does not take the argument, sign.py:92-98)
        self._dr_kw = (lambda dr: {}) if cls_name == "Sign" else (lambda dr: {"domain_rand": dr})
        self.template = cls(host_only=True, **self._dr_kw(False), **env_kwargs)
        self.template.reset(seed=seed)
        self._cls, self._env_kwargs = cls, env_kwargs
        sc = scene_from_env(self.template)
        # texture domain randomisation needs one geometry set per env (texcoords depend on the variant)
        from . import assets as _assets
        room0 = self.template.rooms[0]
        tex_slots = [room0.wall_tex_name, room0.floor_tex_name, room0.ceil_tex_name]
        variants = [_assets.texture_variants(t) for t in tex_slots]
        tex_dr = bool(domain_rand) and generator in (eng.GEN_HALLWAY, eng.GEN_ONEROOM, eng.GEN_PICKUP, eng.GEN_MAZE) and any(len(v) > 1 for v in variants)
        # placement programs: every room may name its own textures
        prog_names = sorted({n for r in self.template.rooms for n in (r.wall_tex_name, r.floor_tex_name, r.ceil_tex_name)})
        prog_tex_dr = bool(domain_rand) and generator == eng.GEN_PROGRAM and any(len(_assets.texture_variants(n)) > 1 for n in prog_names)
        shared = generator != eng.GEN_MAZE and not tex_dr and not prog_tex_dr
        P, S, E = len(sc["polys_nv"]), len(sc["wall_segs"]), max(1, len(sc["ents_kind"]))
        pickup_meshes = None
        if cls_name == "PickupObjects":
            # any mix of kinds can be generated on the device: all 12 ball / key meshes are resident,
            # ids ball_<colour> = 0..5, key_<colour> = 6..11 in sorted colour order
            from .entity import COLOR_NAMES, COLORS, Ball, Box, Key
            E = max(E, self.template.num_objs)
            pickup_meshes = [f"ball_{c}" for c in COLOR_NAMES] + [f"key_{c}" for c in COLOR_NAMES]
            protos = (Ball(COLOR_NAMES[0], size=0.9), Box(COLOR_NAMES[0], size=0.9), Key(COLOR_NAMES[0]))
            first_mesh = (0, -1, 6)
        cfg = base_config(num_envs, self.template.obs_width, self.template.obs_height, E, P, S,
                          max_visible=-(-(P + 6 * E) // 16) * 16,
                          params_ranges=self.template.params.as_ranges(), device_id=device_id)
        cfg.shared_geometry = int(shared)
        cfg.msaa = int(msaa)
        cfg.task, cfg.goal_ent, cfg.num_objs = task, 0, len(sc["ents_kind"])
        ents = [e for e in self.template.entities if e is not self.template.agent]
        if task in (eng.TASK_GOTO, eng.TASK_SIDEWALK) and hasattr(self.template, "box"):
            cfg.goal_ent = ents.index(self.template.box)
        if task == eng.TASK_PUTNEXT:
            cfg.goal_ent, cfg.goal_ent2 = ents.index(self.template.red_box), ents.index(self.template.yellow_box)
        cfg.max_episode_steps = int(min(float(self.template.max_episode_steps), 2 ** 30))
        cfg.domain_rand = int(domain_rand)
        cfg.generator = generator
        cfg.autoreset = eng.AUTORESET_SAME_STEP if autoreset else eng.AUTORESET_OFF
        self.autoreset = bool(autoreset)
        cfg.agent_radius = float(self.template.agent.radius)
        if generator in (eng.GEN_HALLWAY, eng.GEN_ONEROOM):
            room = self.template.rooms[0]
            args = [room.min_x, room.max_x, room.min_z, room.max_z]
            if generator == eng.GEN_HALLWAY:
                args += [room.max_x - 2, room.max_x - 2, math.pi / 4, 0.8]
            else:
                args += [room.min_x, room.max_x, math.pi, 0.8]
            for i, v in enumerate(args):
                cfg.gen_args[i] = float(v)
        if generator == eng.GEN_MAZE:
            t = self.template
            r0 = t.rooms[0]
            texs = [r0.floor_tex, r0.ceil_tex, r0.wall_tex]
            names = [str(v) for v in sc["tex_names"]]
            vals = [t.num_rows, t.num_cols, t.room_size, t.gap_size, r0.wall_height] + [names.index(x.variant) for x in texs]
            for i, v in enumerate(vals):
                cfg.gen_tab[i] = float(v)
            for k, x in enumerate(texs):
                cfg.gen_colors[2 * k] = 512 / x.width
                cfg.gen_colors[2 * k + 1] = 512 / x.height
        if generator == eng.GEN_PICKUP:
            room = self.template.rooms[0]
            for i, v in enumerate([room.min_x, room.max_x, room.min_z, room.max_z]):
                cfg.gen_args[i] = float(v)
            for k, (proto, fm) in enumerate(zip(protos, first_mesh)):
                scale = float(getattr(proto, "scale", 1.0))
                for j, v in enumerate([float(proto.radius), float(proto.height), scale, float(fm)]):
                    cfg.gen_tab[k * 4 + j] = v
            for ci, cname in enumerate(COLOR_NAMES):
                for j in range(3):
                    cfg.gen_colors[ci * 3 + j] = float(COLORS[cname][j])
        self._tex_dr_variants = None
        if prog_tex_dr:     # every variant of every room texture is resident; the program's tables name them
            names = [str(v) for v in sc["tex_names"]]
            for n in prog_names:
                names += [v for v in _assets.texture_variants(n) if v not in names]
            self._tex_dr_variants = names
        if tex_dr:
            order, nid = [], len(sc["tex_names"])
            names = [str(v) for v in sc["tex_names"]]
            for k, vs in enumerate(variants):
                cfg.tex_nvar[k] = len(vs)
                for j, v in enumerate(vs):
                    if v not in names:
                        names.append(v)
                    cfg.tex_var_id[k][j] = names.index(v)
                    w, h = _assets.texture_size(v)
                    cfg.tex_var_scale[k][j][0], cfg.tex_var_scale[k][j][1] = 512 / w, 512 / h
            self._tex_dr_variants = names
            cfg.room_wall_height = float(room0.wall_height)
            cfg.room_no_ceiling = int(bool(room0.no_ceiling))
        pcg_ok = True       # every device generator draws numpy's PCG64 stream
        if rng not in ("auto", "pcg64", "philox") or (rng == "pcg64" and not pcg_ok):
            raise ValueError(f"rng={rng!r} is not available for {env_id} (domain_rand={domain_rand})")
        cfg.rng_mode = eng.RNG_PCG64 if (pcg_ok and rng != "philox") else eng.RNG_PHILOX
        self.rng_mode = "pcg64" if cfg.rng_mode == eng.RNG_PCG64 else "philox"
        self.engine = eng.Engine(cfg)
        self.host_autoreset = False         # every registered id is generated, ruled and auto-reset on the device
        self._upload_assets(sc)
        if cls_name == "RoomObjects":       # any colour of ball / key can be drawn: all twelve meshes are resident
            from .entity import COLOR_NAMES
            pickup_meshes = [f"ball_{c}" for c in COLOR_NAMES] + [f"key_{c}" for c in COLOR_NAMES]
        if pickup_meshes:
            from .objmesh import ObjMesh
            for name in pickup_meshes:
                self.mesh_ids[name] = len(self.mesh_ids)
                m = ObjMesh.get(name)
                self.engine.upload_mesh(self.mesh_ids[name], m.verts, m.norms, m.texcs, m.colors)
        if generator == eng.GEN_PROGRAM:
            from . import genprog
            mesh_map = upload_scene_meshes(self.engine, sc, self.mesh_ids, self.tex_ids)
            slot_of = lambda e: ents.index(e)                       # noqa: E731
            room_of = lambda r: self.template.rooms.index(r)        # noqa: E731
            if cls_name == "RoomObjects":
                ops = genprog.room_objects_ops(0, 1, 2, self.mesh_ids["ball_" + COLOR_NAMES[0]], self.mesh_ids["key_" + COLOR_NAMES[0]])
            else:
                ops = genprog.family_ops(self.template, slot_of, room_of)
            self.engine.set_gen_program(*genprog.compile_program(self.template, sc, self.tex_ids, mesh_map, ops))
        dev = self.engine.device
        H, W = self.template.obs_height, self.template.obs_width
        self.engine.set_obs_layout({"hwc": eng.OBS_HWC_U8, "cwh": eng.OBS_CWH_U8, "grey": eng.OBS_GREY_F64}[obs_layout])
        self.obs = self.engine.obs_buffer()
        self.depth = torch.zeros((num_envs, H, W, 1), dtype=torch.float32, device=dev) if want_depth else None
        self.reward = torch.zeros(num_envs, dtype=torch.float32, device=dev)
        self.terminated = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self._host_envs = None
        self._next_seed = seed
        # what the env's step() reports in `info` beside the observation (collecthealth.py:100, tmaze.py:89, ymaze.py:125)
        self._info_kind = {"CollectHealth": "health", "TMaze": "goal_pos", "TMazeLeft": "goal_pos", "TMazeRight": "goal_pos",
                           "YMaze": "goal_pos", "YMazeLeft": "goal_pos", "YMazeRight": "goal_pos"}.get(cls_name)
        self._info_slot = int(cfg.goal_ent)
        self._info_buf = None
        self._final_info_buf = None

    # ------------------------------------------------------------------ assets / worlds
    def _upload_assets(self, sc):
        from . import assets
        self.tex_ids, self.mesh_ids = {}, {}
        for i, variant in enumerate(self._tex_dr_variants or [str(v) for v in sc["tex_names"]]):
            self.tex_ids[variant] = i
            self.engine.upload_texture(i, assets.texture_rgb_bottom_up(variant))
        if self.engine.cfg.shared_geometry:
            self.engine.set_geometry(-1, polys_array(sc), sc["wall_segs"])

    def host_generate(self, indices, seeds):
        return self._host_generate(indices, seeds)

    def _host_generate(self, indices, seeds):
        """Host world generation (reference-compatible stream) for envs without a device generator."""
        if self._host_envs is None:
            self._host_envs = [None] * self.num_envs
        for i, s in zip(indices, seeds):
            env = self._host_envs[i]
            if env is None:
                env = self._cls(host_only=True, **self._dr_kw(self.domain_rand), **self._env_kwargs)
                self._host_envs[i] = env
            env.reset(seed=int(s))
            sc = scene_from_env(env)
            # with domain randomisation a world may draw a texture variant no earlier world used (concrete_2 ...):
            # upload it on first sight
            for v in [str(v) for v in sc["tex_names"]]:
                if v not in self.tex_ids:
                    from . import assets
                    self.tex_ids[v] = len(self.tex_ids)
                    self.engine.upload_texture(self.tex_ids[v], assets.texture_rgb_bottom_up(v))
            if not self.engine.cfg.shared_geometry:
                tex_map = {k: self.tex_ids[str(v)] for k, v in enumerate(sc["tex_names"])}
                self.engine.set_geometry(i, polys_array(sc, tex_map), sc["wall_segs"])
            mm = upload_scene_meshes(self.engine, sc, self.mesh_ids, self.tex_ids)
            self.engine.set_state(state_arrays([sc], self.engine.E, [mm]), first=i, count=1)
        if self.domain_rand:
            # the per-step forward_step / drift / turn_step draws (miniworld.py:677-680) come from the env's device
            # stream: seed it with the env's seed (not with the env index it got at creation)
            mask = np.zeros(self.num_envs, np.uint8)
            full = np.zeros(self.num_envs, np.uint64)
            for i, s_ in zip(indices, seeds):
                mask[i], full[i] = 1, int(s_)
            self.engine.reset(mask, full)

    # ------------------------------------------------------------------ API
    def reset(self, seed: int | None = None):
        """Reset every env (env i is seeded with seed + i); returns the observation tensor."""
        if seed is not None:
            self._next_seed = seed
        seeds = np.arange(self.num_envs, dtype=np.uint64) + np.uint64(self._next_seed)
        self._next_seed += self.num_envs
        self.engine.reset(None, seeds)
        self.engine.render(self.obs, self.depth)
        return self.obs

    def step(self, actions):
        """actions: integer torch tensor [N] (converted to contiguous int32 on the engine's device if needed)."""
        self.engine.step(actions, self.obs, self.depth, self.reward, self.terminated, self.truncated)
        return self.obs, self.reward, self.terminated, self.truncated

    def infos(self):
        """The batched `info` of the last step as device tensors: {"health": int32[N]} for CollectHealth (collecthealth.py:100),
        {"goal_pos": float64[N, 3]} for TMaze / YMaze (the box's position, tmaze.py:89, ymaze.py:125), {} for the other envs
        (miniworld.py:730 returns an empty dict).  One small gather kernel on the engine's stream; the values are those of the
        state the device holds (with the same-step auto-reset an env that just finished reports its new episode; the finished
        episode's own values: final_infos)."""
        if self._info_kind is None:
            return {}
        torch = self.torch
        if self._info_buf is None:
            dev = self.engine.device
            self._info_buf = (torch.zeros(self.num_envs, dtype=torch.int32, device=dev) if self._info_kind == "health"
                              else torch.zeros((self.num_envs, 3), dtype=torch.float64, device=dev))
        if self._info_kind == "health":
            self.engine.get_info(health=self._info_buf)
        else:
            self.engine.get_info(ent_pos=self._info_buf, ent_slot=self._info_slot)
        return {self._info_kind: self._info_buf}

    def final_infos(self):
        """The `info` each env's last FINISHED episode ended with ({"health": …} / {"goal_pos": …} / {}), as the step kernel kept it
        before the same-step auto-reset installed the next world (mw_get_final_info); rows of envs that have not finished an episode yet
        are undefined (mask them with terminated | truncated of the step)."""
        if self._info_kind is None:
            return {}
        torch = self.torch
        if self._final_info_buf is None:
            dev = self.engine.device
            self._final_info_buf = (torch.zeros(self.num_envs, dtype=torch.int32, device=dev) if self._info_kind == "health"
                                    else torch.zeros((self.num_envs, 3), dtype=torch.float64, device=dev))
        if self._info_kind == "health":
            self.engine.get_final_info(health=self._final_info_buf)
        else:
            self.engine.get_final_info(goal_pos=self._final_info_buf)
        return {self._info_kind: self._final_info_buf}

    def render_top_view(self, render_agent=True):
        """uint8[N,H,W,3] map views (render_top_view, miniworld.py:1088-1175) of every env."""
        layout = self.engine.obs_layout
        self.engine.set_obs_layout(eng.OBS_HWC_U8)
        out = self.engine.obs_buffer()
        self.engine.render_top(out, None, render_agent)
        self.engine.set_obs_layout(layout)
        return out

    def get_visible_ents(self):
        """bool[N, max_ents]: which entity slots each agent currently sees (get_visible_ents,
        miniworld.py:1238-1333: occlusion queries around 0.2 m proxy boxes)."""
        return self.engine.visible_ents().bool()

    def close(self):
        self.engine.close()
