"""Rooms, world generation and the single-environment Gymnasium API.

Public surface of the reference's ``miniworld.miniworld`` (``Room``, ``MiniWorldEnv``,
``gen_texcs_wall`` / ``gen_texcs_floor``; miniworld.py:76-1443) with the GL half removed:
world generation (rooms, portals, static geometry, rejection-sampled placement, domain
randomisation) runs on the host with the same numpy Generator call sequence as the
reference — so ``reset(seed=s)`` builds the same world — while *every* step and every
rendered frame is computed by the HIP engine (``miniworld_amd.engine``).  A
``MiniWorldEnv`` is a batch-of-one view of that engine: before each step the Python object
state is pushed to the device, afterwards it is read back, so ``env.agent.pos``,
``env.entities`` ... behave like the reference's attributes.  The batched, device-resident
API is ``miniworld_amd.vec_env.MiniWorldVecEnv``.
"""
from __future__ import annotations

import math
from enum import IntEnum
from typing import Optional

import numpy as np

from . import engine as _eng
from .entity import Agent, Box, Entity, MeshEnt
from .gymshim import gym, spaces
from .math import Y_VEC, intersect_circle_segs
from .params import DEFAULT_PARAMS
from .texture import Texture

DEFAULT_WALL_HEIGHT = 2.74      # miniworld.py:76
TEX_DENSITY = 512               # texels per metre, miniworld.py:79


def gen_texcs_wall(tex, min_x, min_y, width, height):
    """Texture coordinates of a wall quad: metres * 512 / texture size, float32 (miniworld.py:82-103)."""
    ku, kv = TEX_DENSITY / tex.width, TEX_DENSITY / tex.height
    u0, u1 = min_x * ku, (min_x + width) * ku
    v0, v1 = min_y * kv, (min_y + height) * kv
    return np.array([[u0, v0], [u0, v1], [u1, v1], [u1, v0]], dtype=np.float32)


def gen_texcs_floor(tex, poss):
    """Floor / ceiling texture coordinates straight from x, z (miniworld.py:106-119)."""
    scale = np.array([TEX_DENSITY / tex.width, TEX_DENSITY / tex.height], dtype=float)
    return np.stack([poss[:, 0], poss[:, 2]], axis=1) * scale


class Room:
    """A room: outline polygon, portals, and the static geometry derived from them."""

    def __init__(self, outline, wall_height=DEFAULT_WALL_HEIGHT, floor_tex="floor_tiles_bw",
                 wall_tex="concrete", ceil_tex="concrete_tiles", no_ceiling=False):
        assert outline.ndim == 2 and outline.shape[1] == 2 and outline.shape[0] >= 3
        self.outline = np.insert(outline, 1, 0, axis=1)         # (x, z) -> (x, 0, z)
        self.num_walls = self.outline.shape[0]
        xs, zs = self.outline[:, 0], self.outline[:, 2]
        self.min_x, self.max_x, self.min_z, self.max_z = xs.min(), xs.max(), zs.min(), zs.max()
        self.mid_x = (self.max_x + self.min_x) / 2
        self.mid_z = (self.max_z + self.min_z) / 2
        self.area = (self.max_x - self.min_x) * (self.max_z - self.min_z)
        # unit edge directions and inward normals (outline is counter-clockwise seen from above)
        edges = np.roll(self.outline, -1, axis=0) - self.outline
        self.edge_dirs = (edges.T / np.linalg.norm(edges, axis=1)).T
        norms = -np.cross(self.edge_dirs, Y_VEC)
        self.edge_norms = (norms.T / np.linalg.norm(norms, axis=1)).T
        self.wall_height = wall_height
        self.no_ceiling = no_ceiling
        self.wall_tex_name, self.floor_tex_name, self.ceil_tex_name = wall_tex, floor_tex, ceil_tex
        self.portals = [[] for _ in range(self.num_walls)]
        self.neighbors = []

    def add_portal(self, edge, start_pos=None, end_pos=None, min_x=None, max_x=None, min_z=None,
                   max_z=None, min_y=0, max_y=None):
        """Open a portal in wall ``edge``; extents along the wall, or by world x / z range."""
        if max_y is None:
            max_y = self.wall_height
        assert edge <= self.num_walls and max_y > min_y
        p0 = self.outline[edge]
        p1 = self.outline[(edge + 1) % self.num_walls]
        length = np.linalg.norm(p1 - p0)
        direction = (p1 - p0) / length
        if min_x is not None:
            assert min_z is None and max_z is None and start_pos is None and end_pos is None
            assert p0[0] != p1[0]
            span = sorted(((min_x - p0[0]) / direction[0], (max_x - p0[0]) / direction[0]))
            start_pos, end_pos = span
        elif min_z is not None:
            assert min_x is None and max_x is None and start_pos is None and end_pos is None
            assert p0[2] != p1[2]
            span = sorted(((min_z - p0[2]) / direction[2], (max_z - p0[2]) / direction[2]))
            start_pos, end_pos = span
        else:
            assert min_x is None and max_x is None and min_z is None and max_z is None
        assert end_pos > start_pos
        assert start_pos >= 0, "portal outside of wall extents"
        assert end_pos <= length, "portal outside of wall extents"
        self.portals[edge].append({"start_pos": start_pos, "end_pos": end_pos, "min_y": min_y, "max_y": max_y})
        self.portals[edge].sort(key=lambda p: p["start_pos"])
        return start_pos, end_pos

    def point_inside(self, p):
        """Strictly inside every wall (miniworld.py:272-284)."""
        return np.all(np.sum(self.edge_norms * (p - self.outline), axis=1) > 0)

    def _wall_spans(self, wall_idx, wall_width):
        """(start, end, y0, y1) pieces of one wall, in the reference's emission order
        (miniworld.py:346-386): run up to the first portal, then per portal the piece under the
        sill, the piece above the lintel, and the run to the next portal / the wall's end."""
        portals = self.portals[wall_idx]
        h = self.wall_height
        yield 0, (portals[0]["start_pos"] if portals else wall_width), 0, h
        for i, p in enumerate(portals):
            yield p["start_pos"], p["end_pos"], 0, p["min_y"]
            yield p["start_pos"], p["end_pos"], p["max_y"], h
            nxt = portals[i + 1]["start_pos"] if i + 1 < len(portals) else wall_width
            yield p["end_pos"], nxt, 0, h

    def _gen_static_data(self, params, rng):
        """Floor / ceiling / wall polygons, texture coordinates and collision segments."""
        self.wall_tex = Texture.get(self.wall_tex_name, rng)
        self.floor_tex = Texture.get(self.floor_tex_name, rng)
        self.ceil_tex = Texture.get(self.ceil_tex_name, rng)
        self.floor_verts = self.outline
        self.floor_texcs = gen_texcs_floor(self.floor_tex, self.floor_verts)
        self.ceil_verts = np.flip(self.outline, axis=0) + self.wall_height * Y_VEC     # flipped: faces down
        self.ceil_texcs = gen_texcs_floor(self.ceil_tex, self.ceil_verts)
        verts, norms, texcs, segs = [], [], [], []
        for w in range(self.num_walls):
            p0 = self.outline[w, :]
            p1 = self.outline[(w + 1) % self.num_walls, :]
            width = np.linalg.norm(p1 - p0)
            side = (p1 - p0) / width
            for start, end, y0, y1 in self._wall_spans(w, width):
                if end == start or y0 == y1:
                    continue
                a, b = p0 + start * side, p0 + end * side
                if y0 == 0:
                    segs.append(np.array([b, a]))           # collidable at ground level
                verts += [a + y0 * Y_VEC, a + y1 * Y_VEC, b + y1 * Y_VEC, b + y0 * Y_VEC]
                n = np.cross(b - a, Y_VEC)
                n = -n / np.linalg.norm(n)
                norms += [n] * 4
                texcs.append(gen_texcs_wall(self.wall_tex, start, y0, end - start, y1 - y0))
        self.wall_verts = np.array(verts)
        self.wall_norms = np.array(norms)
        self.wall_segs = np.array(segs) if segs else np.array([]).reshape(0, 2, 3)
        self.wall_texcs = np.concatenate(texcs) if texcs else np.array([]).reshape(0, 2)


class FrameBuffer:
    """Size / sample-count descriptor with the reference's FrameBuffer constructor signature
    (opengl.py:202: FrameBuffer(width, height, num_samples)); the pixels live on the GPU."""

    def __init__(self, width, height, num_samples=1):
        assert 0 < num_samples <= 16
        self.width, self.height, self.num_samples = width, height, num_samples


class MiniWorldEnv(gym.Env):
    """Base class of all environments: world generation + engine-backed simulation."""

    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 30,
                "render_modes": ["human", "rgb_array"], "render_fps": 30}

    class Actions(IntEnum):
        turn_left = 0
        turn_right = 1
        move_forward = 2
        move_back = 3
        pickup = 4
        drop = 5
        toggle = 6
        done = 7

    def __init__(self, max_episode_steps: int = 1500, obs_width: int = 80, obs_height: int = 60,
                 window_width: int = 800, window_height: int = 600, params=DEFAULT_PARAMS,
                 domain_rand: bool = False, render_mode: Optional[str] = None, view: str = "agent",
                 device_id: int = 0, host_only: bool = False):
        self.actions = MiniWorldEnv.Actions
        self.action_space = spaces.Discrete(len(self.actions))
        self.observation_space = spaces.Box(low=0, high=255, shape=(obs_height, obs_width, 3), dtype=np.uint8)
        self.reward_range = (-math.inf, math.inf)
        self.max_episode_steps = max_episode_steps
        self.params = params
        self.domain_rand = domain_rand
        self.render_mode = render_mode
        assert view in ["agent", "top"]
        self.view = view
        self.obs_width, self.obs_height = obs_width, obs_height
        self.window_width, self.window_height = window_width, window_height
        self.device_id = device_id
        self.obs_fb = FrameBuffer(obs_width, obs_height, 8)              # miniworld.py:515
        self.vis_fb = FrameBuffer(window_width, window_height, 16)       # miniworld.py:518
        # host_only: generate worlds but never touch the GPU (used by MiniWorldVecEnv, which owns
        # a batched engine, and by the CPU tests of world generation); step/render then raise.
        self._host_only = host_only
        self._engine = None
        self._engine_key = None
        self.reset()

    # ------------------------------------------------------------------ episode start
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self.step_count = 0
        self.agent = Agent()
        self.entities = []
        self.rooms = []
        self.wall_segs = []
        self._gen_world()
        rand = self.np_random if self.domain_rand else None
        self.params.sample_many(rand, self, ["sky_color", "light_pos", "light_color", "light_ambient"])
        self.max_forward_step = self.params.get_max("forward_step")
        for ent in self.entities:
            ent.randomize(self.params, rand)
        self.min_x = min(r.min_x for r in self.rooms)
        self.max_x = max(r.max_x for r in self.rooms)
        self.min_z = min(r.min_z for r in self.rooms)
        self.max_z = max(r.max_z for r in self.rooms)
        if len(self.wall_segs) == 0:
            self._gen_static_data()
        if self._host_only:
            return None, {}
        self._upload_world()
        return self.render_obs(), {}

    def _gen_world(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ world building
    def add_rect_room(self, min_x, max_x, min_z, max_z, **kwargs):
        """Rectangular room; outline counter-clockwise seen from above: E, N, W, S walls."""
        outline = np.array([[max_x, max_z], [max_x, min_z], [min_x, min_z], [min_x, max_z]])
        return self.add_room(outline=outline, **kwargs)

    def add_room(self, **kwargs):
        assert len(self.wall_segs) == 0, "cannot add rooms after static data is generated"
        room = Room(**kwargs)
        self.rooms.append(room)
        return room

    def connect_rooms(self, room_a, room_b, min_x=None, max_x=None, min_z=None, max_z=None, max_y=None):
        """Join two rooms along facing walls with portals (+ a connecting room across a gap)."""
        pair = None
        for ia in range(room_a.num_walls):
            na = room_a.edge_norms[ia]
            for ib in range(room_b.num_walls):
                if np.dot(na, room_b.edge_norms[ib]) > -0.9:        # not facing each other
                    continue
                if np.dot(na, room_b.outline[ib] - room_a.outline[ia]) > 0.05:   # not touching
                    continue
                pair = (ia, ib)
                break
            if pair:
                break
        assert pair is not None, "matching edges not found in connect_rooms"
        ia, ib = pair
        sa, ea = room_a.add_portal(edge=ia, min_x=min_x, max_x=max_x, min_z=min_z, max_z=max_z, max_y=max_y)
        sb, eb = room_b.add_portal(edge=ib, min_x=min_x, max_x=max_x, min_z=min_z, max_z=max_z, max_y=max_y)
        a = room_a.outline[ia] + room_a.edge_dirs[ia] * sa
        b = room_a.outline[ia] + room_a.edge_dirs[ia] * ea
        c = room_b.outline[ib] + room_b.edge_dirs[ib] * sb
        d = room_b.outline[ib] + room_b.edge_dirs[ib] * eb
        if np.linalg.norm(a - d) < 0.001:       # portals coincide: nothing in between
            return
        len_a, len_b = np.linalg.norm(b - a), np.linalg.norm(d - c)
        quad = np.stack([c, b, a, d])
        outline = np.stack([quad[:, 0], quad[:, 2]], axis=1)
        max_y = max_y if max_y is not None else room_a.wall_height
        link = Room(outline, wall_height=max_y, wall_tex=room_a.wall_tex_name, floor_tex=room_a.floor_tex_name,
                    ceil_tex=room_a.ceil_tex_name, no_ceiling=room_a.no_ceiling)
        self.rooms.append(link)
        link.add_portal(1, start_pos=0, end_pos=len_a)
        link.add_portal(3, start_pos=0, end_pos=len_b)

    def place_entity(self, ent, room=None, pos=None, dir=None, min_x=None, max_x=None, min_z=None, max_z=None):  # noqa: A002
        """Put an entity at ``pos`` or at a random free spot (rejection sampling with the
        reference's exact Generator call sequence, miniworld.py:872-905)."""
        assert len(self.rooms) > 0, "create rooms before calling place_entity"
        assert ent.radius is not None, "entity must have physical size defined"
        if len(self.wall_segs) == 0:
            self._gen_static_data()
        if pos is not None:
            ent.dir = dir if dir is not None else self.np_random.uniform(-math.pi, math.pi)
            ent.pos = pos
            self.entities.append(ent)
            return ent
        while True:
            r = room if room else self.rooms[self.np_random.choice(len(self.rooms), p=self.room_probs)]
            lx = r.min_x if min_x is None else min_x
            hx = r.max_x if max_x is None else max_x
            lz = r.min_z if min_z is None else min_z
            hz = r.max_z if max_z is None else max_z
            pos = self.np_random.uniform(low=[lx - ent.radius, 0, lz - ent.radius],
                                         high=[hx + ent.radius, 0, hz + ent.radius])
            if not r.point_inside(pos):
                continue
            if self.intersect(ent, pos, ent.radius):
                continue
            ent.pos = pos
            ent.dir = dir if dir is not None else self.np_random.uniform(-math.pi, math.pi)
            break
        self.entities.append(ent)
        return ent

    def place_agent(self, room=None, pos=None, dir=None, min_x=None, max_x=None, min_z=None, max_z=None):  # noqa: A002
        return self.place_entity(self.agent, room=room, pos=pos, dir=dir, min_x=min_x, max_x=max_x,
                                 min_z=min_z, max_z=max_z)

    def intersect(self, ent, pos, radius):
        """Host-side collision query used by placement: True for a wall, the entity hit, or None."""
        p = np.array([pos[0], 0, pos[2]])
        if intersect_circle_segs(p, radius, self.wall_segs):
            return True
        for other in self.entities:
            if other is ent:
                continue
            q = np.array([other.pos[0], 0, other.pos[2]])
            if np.linalg.norm(q - p) < radius + other.radius:
                return other
        return None

    def near(self, ent0, ent1=None):
        if ent1 is None:
            ent1 = self.agent
        return np.linalg.norm(ent0.pos - ent1.pos) < ent0.radius + ent1.radius + 1.1 * self.max_forward_step

    # Host-side counterparts of the two motion primitives (miniworld.py:606-668).  step() runs them on the GPU; these
    # are for callers that move the agent themselves — the next step / render pushes the host state to the engine.
    def _get_carry_pos(self, agent_pos, ent):
        dist = self.agent.radius + ent.radius + self.max_forward_step
        pos = agent_pos + self.agent.dir_vec * 1.05 * dist
        lift = max(self.agent.cam_height - ent.height - 0.3, 0)      # keeps the carried object in view
        return pos + np.array([0.0, 1.0, 0.0]) * lift

    def move_agent(self, fwd_dist, fwd_drift):
        """Forward by fwd_dist, sideways by fwd_drift; blocked (no sliding) by walls and entities.  True if moved."""
        target = self.agent.pos + self.agent.dir_vec * fwd_dist + self.agent.right_vec * fwd_drift
        if self.intersect(self.agent, target, self.agent.radius):
            return False
        held = self.agent.carrying
        if held:
            held_target = self._get_carry_pos(target, held)
            if self.intersect(held, held_target, held.radius):
                return False
            held.pos = held_target
        self.agent.pos = target
        return True

    def turn_agent(self, turn_angle):
        """Turn by turn_angle degrees (positive = left); undone if the carried object would collide.  True if turned."""
        before = self.agent.dir
        self.agent.dir += turn_angle * (math.pi / 180)
        held = self.agent.carrying
        if held:
            held_target = self._get_carry_pos(self.agent.pos, held)
            if self.intersect(held, held_target, held.radius):
                self.agent.dir = before
                return False
            held.pos = held_target
            held.dir = self.agent.dir
        return True

    def _gen_static_data(self):
        rng = self.np_random if self.domain_rand else None
        for room in self.rooms:
            room._gen_static_data(self.params, rng)
        self.wall_segs = np.concatenate([r.wall_segs for r in self.rooms])
        self.room_probs = np.array([r.area for r in self.rooms], dtype=float)
        self.room_probs /= np.sum(self.room_probs)

    def _reward(self):
        return 1.0 - 0.2 * (self.step_count / self.max_episode_steps)

    # ------------------------------------------------------------------ engine plumbing
    def scene(self) -> dict:
        """The world as plain arrays (the same "neutral scene" layout the tests use)."""
        from .scene import scene_from_env
        return scene_from_env(self)

    def _upload_world(self):
        from .scene import EngineBinding
        if self._host_only:
            raise RuntimeError("this environment was created with host_only=True: it has no engine")
        if self._engine is None:
            self._engine = EngineBinding(self)
        self._engine.upload_world(self)

    def _sync_to_device(self):
        self._engine.push_state(self)

    # ------------------------------------------------------------------ simulation
    def step(self, action):
        """One action: physics + collision on the GPU, then the rendered observation."""
        self.step_count += 1
        rand = self.np_random if self.domain_rand else None
        fwd_step = self.params.sample(rand, "forward_step")
        fwd_drift = self.params.sample(rand, "forward_drift")
        turn_step = self.params.sample(rand, "turn_step")
        if self._engine is None:
            raise RuntimeError("no engine: environment created with host_only=True")
        obs = self._engine.step(self, int(action), fwd_step, fwd_drift, turn_step)
        truncation = self.step_count >= self.max_episode_steps
        return obs, 0, False, truncation, {}

    def _is_obs_fb(self, fb):
        return fb is None or (fb.width == self.obs_width and fb.height == self.obs_height and fb.num_samples == 8)

    def _render_into(self, fb, top, render_agent):
        self._engine.push_state(self)
        msaa = 16 if fb.num_samples > 8 else 8
        return self._engine.engine.render_view(0, fb.width, fb.height, msaa, top=top, render_agent=render_agent).cpu().numpy()

    def render_obs(self, frame_buffer=None):
        if not self._is_obs_fb(frame_buffer):
            return self._render_into(frame_buffer, False, False)
        return self._engine.render(self)["rgb"]

    def render_top_view(self, frame_buffer=None, render_agent=True, return_scale=False):
        """Orthographic map of the whole floorplan at the observation resolution (the reference's
        default frame buffer for this call is obs_fb, miniworld.py:1093-1094)."""
        if self._is_obs_fb(frame_buffer):
            img = self._engine.render(self, top_view=True, render_agent=render_agent)["rgb"]
        else:
            img = self._render_into(frame_buffer, True, render_agent)
        if not return_scale:
            return img
        min_x, max_x, min_z, max_z = self.min_x - 1, self.max_x + 1, self.min_z - 1, self.max_z + 1
        width, height = max_x - min_x, max_z - min_z
        aspect, fb_aspect = width / height, self.obs_width / self.obs_height
        if aspect > fb_aspect:
            diff = width / fb_aspect - height
            min_z, max_z = min_z - diff / 2, max_z + diff / 2
        elif aspect < fb_aspect:
            diff = height * fb_aspect - width
            min_x, max_x = min_x - diff / 2, max_x + diff / 2
        xs, zs = self.obs_width / (max_x - min_x), self.obs_height / (max_z - min_z)
        return img, {"x_scale": xs, "z_scale": zs, "x_offset": int(0 - min_x * xs), "z_offset": int(0 - min_z * zs)}

    def render_depth(self, frame_buffer=None):
        """Depth map in metres at the frame buffer's size and sample count (miniworld.py:1223-1236)."""
        if not self._is_obs_fb(frame_buffer):
            self._engine.push_state(self)
            msaa = 16 if frame_buffer.num_samples > 8 else 8
            _, dep = self._engine.engine.render_view(0, frame_buffer.width, frame_buffer.height, msaa, want_depth=True)
            return dep.cpu().numpy()
        return self._engine.render(self, want_depth=True)["depth"]

    def get_visible_ents(self):
        """Set of entities whose 0.2 m proxy box passes an occlusion query from the agent's camera
        (miniworld.py:1238-1333)."""
        return self._engine.visible_ents(self)

    def render(self):
        if self.render_mode is None:
            gym.logger.warn("You are calling render method without specifying any render mode.")
            return None
        # rgb_array / human: the 800 x 600 x 16-sample visualisation buffer (miniworld.py:1354-1362)
        return self.render_top_view(self.vis_fb) if self.view == "top" else self.render_obs(self.vis_fb)

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
