"""MiniWorld-YMaze(-Left/-Right)-v0: three corridors at 120 degrees around a triangular hub; the box
is at the end of the left or the right arm (ymaze.py:12-127).  Non-rectangular rooms (rotated
outlines, a triangular hub) on the same polygon path as every other room."""
import math

import numpy as np

from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..math import gen_rot_matrix
from ..miniworld import MiniWorldEnv


class YMaze(MiniWorldEnv, EzPickle):
    def __init__(self, goal_pos=None, **kwargs):
        self.goal_pos = goal_pos
        MiniWorldEnv.__init__(self, max_episode_steps=280, **kwargs)
        EzPickle.__init__(self, goal_pos, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        # the starting arm; its outline rotated by -+120 degrees about Y gives the other two (ymaze.py:57-93)
        main_outline = np.array([[-9.15, 0, -2], [-9.15, 0, +2], [-1.15, 0, +2], [-1.15, 0, -2]])
        main_arm = self.add_room(outline=np.delete(main_outline, 1, 1))
        hub_room = self.add_room(outline=np.array([[-1.15, -2], [-1.15, +2], [2.31, 0]]))
        m = gen_rot_matrix(np.array([0, 1, 0]), -120 * (math.pi / 180))
        left_arm = self.add_room(outline=np.delete(np.dot(main_outline, m), 1, 1))
        m = gen_rot_matrix(np.array([0, 1, 0]), +120 * (math.pi / 180))
        right_arm = self.add_room(outline=np.delete(np.dot(main_outline, m), 1, 1))
        self.connect_rooms(main_arm, hub_room, min_z=-2, max_z=2)
        self.connect_rooms(left_arm, hub_room, min_z=-1.995, max_z=0)
        self.connect_rooms(right_arm, hub_room, min_z=0, max_z=1.995)
        self.box = Box(color="red")
        if self.goal_pos is not None:
            g = self.goal_pos
            self.place_entity(self.box, min_x=g[0], max_x=g[0], min_z=g[2], max_z=g[2])
        elif self.np_random.integers(0, 2) == 0:
            self.place_entity(self.box, room=left_arm, max_z=left_arm.min_z + 2.5)
        else:
            self.place_entity(self.box, room=right_arm, min_z=right_arm.max_z - 2.5)
        self.place_agent(dir=self.np_random.uniform(-math.pi / 4, math.pi / 4), room=main_arm)

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        info["goal_pos"] = self.box.pos
        return obs, reward, termination, truncation, info


class YMazeLeft(YMaze):
    def __init__(self, goal_pos=[3.9, 0, -7.0], **kwargs):
        super().__init__(goal_pos=goal_pos, **kwargs)


class YMazeRight(YMaze):
    def __init__(self, goal_pos=[3.9, 0, 7.0], **kwargs):
        super().__init__(goal_pos=goal_pos, **kwargs)
