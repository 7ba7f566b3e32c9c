"""Entities: the host-side object model (pos, dir, radius, ...) of the reference's
``miniworld.entity`` (entity.py:43-551).  There is no ``render()`` here: drawing is done by
the HIP engine from the state these objects carry (Box: size + colour, MeshEnt: mesh + scale).
"""
from __future__ import annotations

import math

import numpy as np

from .math import X_VEC, Y_VEC, Z_VEC, gen_rot_matrix
from .objmesh import ObjMesh
from .texture import Texture

COLORS = {
    "red": np.array([1.0, 0.0, 0.0]),
    "green": np.array([0.0, 1.0, 0.0]),
    "blue": np.array([0.0, 0.0, 1.0]),
    "purple": np.array([0.44, 0.15, 0.76]),
    "yellow": np.array([1.00, 1.00, 0.00]),
    "grey": np.array([0.39, 0.39, 0.39]),
}
COLOR_NAMES = sorted(COLORS.keys())


class Entity:
    def __init__(self):
        self.pos = None         # world position (floor level for most entities)
        self.dir = None         # heading in radians
        self.radius = 0         # bounding cylinder
        self.height = 0

    def randomize(self, params, rng):
        pass

    def step(self, delta_time):
        pass

    @property
    def dir_vec(self):
        return np.array([math.cos(self.dir), 0, -math.sin(self.dir)])

    @property
    def right_vec(self):
        return np.array([math.sin(self.dir), 0, math.cos(self.dir)])

    @property
    def is_static(self):
        return False


class MeshEnt(Entity):
    """Entity drawn from an OBJ mesh scaled to ``height`` (entity.py:124-165)."""

    def __init__(self, mesh_name, height, static=True):
        super().__init__()
        self.static = static
        self.mesh_name = mesh_name
        self.mesh = ObjMesh.get(mesh_name)
        sx, sy, sz = self.mesh.max_coords
        self.scale = height / sy
        self.radius = math.sqrt(sx * sx + sz * sz) * self.scale
        self.height = height

    @property
    def is_static(self):
        return self.static


class _Frame(Entity):
    """Common part of ImageFrame / TextFrame: a static quad strip on a wall, facing +x of its own
    frame, plus a black border.  The reference draws it into display list 1 with
    glTranslatef(pos) glRotatef(dir) around raw quads (entity.py:193-259, 303-383); here `quads()`
    returns the same quads as data, for scene_from_env to append to the static polygon list."""

    @property
    def is_static(self):
        return True

    def _front(self):
        """[(texture or None, z_0, z_1)] front faces, left to right in texture space."""
        raise NotImplementedError

    def _sx(self):
        return self.depth

    def quads(self):
        """[(verts[4][3] local, texcs[4][2], normal[3] local, rgb[3], texture or None)] in draw order."""
        sx, hz, hy = self._sx(), self.width / 2, self.height / 2
        out = []
        for tex, z0, z1 in self._front():
            out.append(([(sx, +hy, z0), (sx, +hy, z1), (sx, -hy, z1), (sx, -hy, z0)],
                        [(1, 1), (0, 1), (0, 0), (1, 0)], (1, 0, 0), (1, 1, 1), tex))
        black, uv0 = (0, 0, 0), [(0, 0)] * 4
        out.append(([(0, +hy, -hz), (+sx, +hy, -hz), (+sx, -hy, -hz), (0, -hy, -hz)], uv0, (0, 0, -1), black, None))
        out.append(([(+sx, +hy, +hz), (0, +hy, +hz), (0, -hy, +hz), (+sx, -hy, +hz)], uv0, (0, 0, 1), black, None))
        out.append(([(+sx, +hy, +hz), (+sx, +hy, -hz), (0, +hy, -hz), (0, +hy, +hz)], uv0, (0, 1, 0), black, None))
        out.append(([(+sx, -hy, -hz), (+sx, -hy, +hz), (0, -hy, +hz), (0, -hy, -hz)], uv0, (0, -1, 0), black, None))
        return out


class ImageFrame(_Frame):
    """Frame to display an image on a wall; pos is the middle of the frame, on the wall (entity.py:168-259)."""

    def __init__(self, pos, dir, tex_name, width, depth=0.05):  # noqa: A002
        super().__init__()
        self.pos = pos
        self.dir = dir
        self.tex = Texture.get(tex_name)
        self.width = width
        self.depth = depth
        self.height = (float(self.tex.height) / self.tex.width) * self.width

    def _front(self):
        hz = self.width / 2
        return [(self.tex, -hz, +hz)]


class TextFrame(_Frame):
    """Frame to display text or numbers on a wall (entity.py:262-383): one textured quad per character."""

    def __init__(self, pos, dir, str, height=0.15, depth=0.05):  # noqa: A002
        super().__init__()
        self.pos = pos
        self.dir = dir
        self.str = str
        self.depth = depth
        self.height = height
        self.width = len(str) * height
        self.texs = []

    def randomize(self, params, rng):
        self.texs = []
        for ch in self.str:
            try:
                self.texs.append(None if ch == " " else Texture.get(f"chars/ch_0x{ord(ch)}", rng))
            except Exception:
                raise ValueError("only alphanumerical characters supported in TextFrame")

    def _sx(self):
        return 0.05             # the reference stores `depth` but draws with sx = 0.05 (entity.py:311)

    def _front(self):
        hz, cw = self.width / 2, self.height
        out = []
        for idx in range(len(self.str)):
            z0 = hz - cw * (idx + 1)
            out.append((self.texs[idx], z0, z0 + cw))
        return out


class Box(Entity):
    """Coloured box (entity.py:386-432)."""

    def __init__(self, color, size=0.8):
        super().__init__()
        if type(size) is int or type(size) is float:
            size = np.array([size, size, size])
        size = np.array(size)
        sx, sy, sz = size
        self.color = color
        self.size = size
        self.radius = math.sqrt(sx * sx + sz * sz) / 2
        self.height = sy
        self.color_vec = COLORS[color]

    def randomize(self, params, rng):
        self.color_vec = np.clip(COLORS[self.color] + params.sample(rng, "obj_color_bias"), 0, 1)


class Key(MeshEnt):
    def __init__(self, color):
        assert color in COLOR_NAMES
        super().__init__(mesh_name=f"key_{color}", height=0.35, static=False)


class Ball(MeshEnt):
    def __init__(self, color, size=0.6):
        assert color in COLOR_NAMES
        super().__init__(mesh_name=f"ball_{color}", height=size, static=False)


class Agent(Entity):
    """The camera-carrying agent (entity.py:455-551)."""

    def __init__(self):
        super().__init__()
        self.cam_height = 1.5
        self.cam_pitch = 0          # degrees, positive looks up
        self.cam_fov_y = 60
        self.cam_fwd_disp = 0
        self.radius = 0.4
        self.height = 1.6
        self.carrying = None

    @property
    def cam_pos(self):
        disp = np.dot(np.array([self.cam_fwd_disp, self.cam_height, 0]), gen_rot_matrix(Y_VEC, self.dir))
        return self.pos + disp

    @property
    def cam_dir(self):
        d = np.dot(X_VEC, gen_rot_matrix(Z_VEC, self.cam_pitch * math.pi / 180))
        return np.dot(d, gen_rot_matrix(Y_VEC, self.dir))

    def randomize(self, params, rng):
        params.sample_many(rng, self, ["cam_height", "cam_fwd_disp", "cam_pitch", "cam_fov_y"])
