"""MiniWorld-RoomObjects-v0: a single room with a box, a ball and a key; no reward (roomobjects.py:8-82)."""
import math

from ..entity import COLOR_NAMES, Ball, Box, Key
from ..gymshim import EzPickle
from ..miniworld import MiniWorldEnv


class RoomObjects(MiniWorldEnv, EzPickle):
    def __init__(self, size=10, **kwargs):
        assert size >= 2
        self.size = size
        MiniWorldEnv.__init__(self, max_episode_steps=math.inf, **kwargs)
        EzPickle.__init__(self, size, **kwargs)

    def _gen_world(self):
        self.add_rect_room(min_x=0, max_x=self.size, min_z=0, max_z=self.size, wall_tex="brick_wall",
                           floor_tex="asphalt", no_ceiling=True)
        self.agent.radius = 1.5         # keeps the objects far enough from the spawn point to be seen
        colors = list(COLOR_NAMES)
        self.place_entity(Box(color=colors[self.np_random.choice(len(colors))], size=0.9))
        self.place_entity(Ball(color=colors[self.np_random.choice(len(colors))], size=0.9))
        self.place_entity(Key(color=colors[self.np_random.choice(len(colors))]))
        self.place_agent()
