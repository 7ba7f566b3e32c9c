"""MiniWorld-Hallway-v0: reach the red box at the end of a straight hallway (hallway.py:8-74)."""
import math

from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class Hallway(MiniWorldEnv, EzPickle):
    def __init__(self, length=12, **kwargs):
        assert length >= 2
        self.length = length
        MiniWorldEnv.__init__(self, max_episode_steps=250, **kwargs)
        EzPickle.__init__(self, length, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)    # left / right / forward

    def _gen_world(self):
        room = self.add_rect_room(min_x=-1, max_x=-1 + self.length, min_z=-2, max_z=2)
        self.box = self.place_entity(Box(color="red"), min_x=room.max_x - 2)
        # note the argument order: the heading is drawn before the position
        self.place_agent(dir=self.np_random.uniform(-math.pi / 4, math.pi / 4), max_x=room.max_x - 2)

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info
