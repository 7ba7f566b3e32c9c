"""MiniWorld-ThreeRooms-v0: two small rooms off a large one with five objects and a picture on the
wall (threerooms.py:7-73): Box, ImageFrame, textured duckie mesh, Key, Ball."""
import math

from ..entity import Ball, Box, ImageFrame, Key, MeshEnt
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class ThreeRooms(MiniWorldEnv, EzPickle):
    def __init__(self, **kwargs):
        MiniWorldEnv.__init__(self, max_episode_steps=400, **kwargs)
        EzPickle.__init__(self, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        room0 = self.add_rect_room(min_x=-7, max_x=7, min_z=0.5, max_z=7)
        room1 = self.add_rect_room(min_x=-7, max_x=-1, min_z=-7, max_z=-0.5)
        room2 = self.add_rect_room(min_x=1, max_x=7, min_z=-7, max_z=-0.5)
        self.connect_rooms(room0, room1, min_x=-5.25, max_x=-2.75)
        self.connect_rooms(room0, room2, min_x=2.75, max_x=5.25)
        self.box = self.place_entity(Box(color="red"))
        self.place_entity(Box(color="green", size=0.6))
        self.entities.append(ImageFrame(pos=[0, 1.35, 7], dir=math.pi / 2, width=1.8, tex_name="logo_mila"))
        self.place_entity(MeshEnt(mesh_name="duckie", height=0.25, static=False))
        self.place_entity(Key(color="blue"))
        self.place_entity(Ball(color="green"))
        self.place_agent()

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        return obs, reward, termination, truncation, info
