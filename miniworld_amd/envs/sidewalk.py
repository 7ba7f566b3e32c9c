"""MiniWorld-Sidewalk-v0: walk along a sidewalk to a red box; stepping into the street ends the
episode (sidewalk.py:9-104).  Cones and a building are textured static meshes."""
import math

import numpy as np

from ..entity import Box, MeshEnt
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class Sidewalk(MiniWorldEnv, EzPickle):
    def __init__(self, **kwargs):
        MiniWorldEnv.__init__(self, max_episode_steps=150, **kwargs)
        EzPickle.__init__(self, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        sidewalk = self.add_rect_room(min_x=-3, max_x=0, min_z=0, max_z=12, wall_tex="brick_wall",
                                      floor_tex="concrete_tiles", no_ceiling=True)
        self.street = self.add_rect_room(min_x=0, max_x=6, min_z=-80, max_z=80, floor_tex="asphalt", no_ceiling=True)
        self.connect_rooms(sidewalk, self.street, min_z=0, max_z=12)
        self.place_entity(MeshEnt(mesh_name="building", height=30), pos=np.array([30, 0, 30]), dir=-math.pi)
        for i in range(1, sidewalk.max_z // 2):
            self.place_entity(MeshEnt(mesh_name="cone", height=0.75), pos=np.array([1, 0, 2 * i]))
        self.box = self.place_entity(Box(color="red"), room=sidewalk, min_z=sidewalk.max_z - 2, max_z=sidewalk.max_z)
        self.place_agent(room=sidewalk, min_z=0, max_z=1.5)

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.street.point_inside(self.agent.pos):     # walking into the street ends the episode
            reward = 0
            termination = True
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info
