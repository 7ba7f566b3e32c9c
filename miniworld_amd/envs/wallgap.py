"""MiniWorld-WallGap-v0: two outdoor rooms joined by a gap in a wall, a red box in the far one and
a (textured) building in the background (wallgap.py:9-89)."""
import math

import numpy as np

from ..entity import Box, MeshEnt
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class WallGap(MiniWorldEnv, EzPickle):
    def __init__(self, **kwargs):
        MiniWorldEnv.__init__(self, max_episode_steps=300, **kwargs)
        EzPickle.__init__(self, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        room0 = self.add_rect_room(min_x=-7, max_x=7, min_z=0.5, max_z=8, wall_tex="brick_wall",
                                   floor_tex="asphalt", no_ceiling=True)
        room1 = self.add_rect_room(min_x=-7, max_x=7, min_z=-8, max_z=-0.5, wall_tex="brick_wall",
                                   floor_tex="asphalt", no_ceiling=True)
        self.connect_rooms(room0, room1, min_x=-1.5, max_x=1.5)
        self.box = self.place_entity(Box(color="red"), room=room1)
        self.place_entity(MeshEnt(mesh_name="building", height=30), pos=np.array([30, 0, 30]), dir=-math.pi)
        self.place_agent(room=room0)

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info
