"""MiniWorld-TMaze(-Left/-Right)-v0: a corridor ending in a T; the box is in one arm (tmaze.py:9-101)."""
import math

from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class TMaze(MiniWorldEnv, EzPickle):
    def __init__(self, goal_pos=None, **kwargs):
        self.goal_pos = goal_pos
        MiniWorldEnv.__init__(self, max_episode_steps=280, **kwargs)
        EzPickle.__init__(self, goal_pos, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        stem = self.add_rect_room(min_x=-1, max_x=8, min_z=-2, max_z=2)
        bar = self.add_rect_room(min_x=8, max_x=12, min_z=-8, max_z=8)
        self.connect_rooms(stem, bar, min_z=-2, max_z=2)
        self.box = Box(color="red")
        if self.goal_pos is not None:
            g = self.goal_pos
            self.place_entity(self.box, min_x=g[0], max_x=g[0], min_z=g[2], max_z=g[2])
        elif self.np_random.integers(0, 2) == 0:
            self.place_entity(self.box, room=bar, max_z=bar.min_z + 2)
        else:
            self.place_entity(self.box, room=bar, min_z=bar.max_z - 2)
        self.place_agent(dir=self.np_random.uniform(-math.pi / 4, math.pi / 4), room=stem)

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        info["goal_pos"] = self.box.pos
        return obs, reward, termination, truncation, info


class TMazeLeft(TMaze):
    def __init__(self, goal_pos=[10, 0, -6], **kwargs):
        super().__init__(goal_pos=goal_pos, **kwargs)


class TMazeRight(TMaze):
    def __init__(self, goal_pos=[10, 0, 6], **kwargs):
        super().__init__(goal_pos=goal_pos, **kwargs)
