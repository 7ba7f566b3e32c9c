"""MiniWorld-FourRooms-v0: four rooms joined by openings, go to the red box (fourrooms.py:8-73)."""
from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class FourRooms(MiniWorldEnv, EzPickle):
    def __init__(self, **kwargs):
        MiniWorldEnv.__init__(self, max_episode_steps=250, **kwargs)
        EzPickle.__init__(self, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        top_left = self.add_rect_room(min_x=-7, max_x=-1, min_z=1, max_z=7)
        top_right = self.add_rect_room(min_x=1, max_x=7, min_z=1, max_z=7)
        bottom_right = self.add_rect_room(min_x=1, max_x=7, min_z=-7, max_z=-1)
        bottom_left = self.add_rect_room(min_x=-7, max_x=-1, min_z=-7, max_z=-1)
        # door-height openings (2.2 m) between neighbouring rooms, clockwise
        self.connect_rooms(top_left, top_right, min_z=3, max_z=5, max_y=2.2)
        self.connect_rooms(top_right, bottom_right, min_x=3, max_x=5, max_y=2.2)
        self.connect_rooms(bottom_right, bottom_left, min_z=-5, max_z=-3, max_y=2.2)
        self.connect_rooms(bottom_left, top_left, min_x=-5, max_x=-3, max_y=2.2)
        self.box = self.place_entity(Box(color="red"))
        self.place_agent()

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info
