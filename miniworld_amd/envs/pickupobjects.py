"""MiniWorld-PickupObjects-v0: collect the objects scattered in a big room (pickupobjects.py:8-95)."""
from ..entity import COLOR_NAMES, Ball, Box, Key
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv


class PickupObjects(MiniWorldEnv, EzPickle):
    def __init__(self, size=12, num_objs=5, **kwargs):
        assert size >= 2
        self.size, self.num_objs = size, num_objs
        MiniWorldEnv.__init__(self, max_episode_steps=400, **kwargs)
        EzPickle.__init__(self, size, num_objs, **kwargs)
        self.action_space = spaces.Discrete(self.actions.pickup + 1)

    def _gen_world(self):
        self.add_rect_room(min_x=0, max_x=self.size, min_z=0, max_z=self.size,
                           wall_tex="brick_wall", floor_tex="asphalt", no_ceiling=True)
        kinds = [Ball, Box, Key]
        colors = list(COLOR_NAMES)
        for _ in range(self.num_objs):
            kind = kinds[self.np_random.choice(len(kinds))]
            color = colors[self.np_random.choice(len(colors))]
            if kind is Box:
                self.place_entity(Box(color=color, size=0.9))
            elif kind is Ball:
                self.place_entity(Ball(color=color, size=0.9))
            else:
                self.place_entity(Key(color=color))
        self.place_agent()
        self.num_picked_up = 0

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.agent.carrying:
            # the observation above still shows the object at its carry pose (miniworld.py:711-717)
            self.entities.remove(self.agent.carrying)
            self.agent.carrying = None
            self.num_picked_up += 1
            reward = 1
            if self.num_picked_up == self.num_objs:
                termination = True
        return obs, reward, termination, truncation, info
