"""MiniWorld-PutNext-v0: carry the red box next to the yellow box (putnext.py:7-80)."""
from ..entity import COLOR_NAMES, Box
from ..gymshim import EzPickle
from ..miniworld import MiniWorldEnv


class PutNext(MiniWorldEnv, EzPickle):
    def __init__(self, size=12, **kwargs):
        assert size >= 2
        self.size = size
        MiniWorldEnv.__init__(self, max_episode_steps=250, **kwargs)
        EzPickle.__init__(self, size, **kwargs)

    def _gen_world(self):
        self.add_rect_room(min_x=0, max_x=self.size, min_z=0, max_z=self.size)
        for color in COLOR_NAMES:
            box = Box(color=color, size=self.np_random.uniform(0.6, 0.85))      # size drawn before the placement
            self.place_entity(box)
            if color == "red":
                self.red_box = box
            elif color == "yellow":
                self.yellow_box = box
        self.place_agent()

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if not self.agent.carrying and self.near(self.red_box, self.yellow_box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info
