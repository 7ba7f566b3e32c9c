"""MiniWorld-OneRoom-v0 and variants: one square room, go to the red box (oneroom.py:7-94)."""
from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv
from ..params import DEFAULT_PARAMS


class OneRoom(MiniWorldEnv, EzPickle):
    def __init__(self, size=10, max_episode_steps=180, **kwargs):
        assert size >= 2
        self.size = size
        MiniWorldEnv.__init__(self, max_episode_steps=max_episode_steps, **kwargs)
        EzPickle.__init__(self, size=size, max_episode_steps=max_episode_steps, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        self.add_rect_room(min_x=0, max_x=self.size, min_z=0, max_z=self.size)
        self.box = self.place_entity(Box(color="red"))
        self.place_agent()

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info


class OneRoomS6(OneRoom):
    def __init__(self, size=6, max_episode_steps=100, **kwargs):
        super().__init__(size=size, max_episode_steps=max_episode_steps, **kwargs)


# larger movement steps for fast stepping (oneroom.py:79-82)
default_params = DEFAULT_PARAMS.no_random()
default_params.set("forward_step", 0.7)
default_params.set("turn_step", 45)


class OneRoomS6Fast(OneRoomS6):
    def __init__(self, max_episode_steps=50, params=default_params, domain_rand=False, **kwargs):
        super().__init__(max_episode_steps=max_episode_steps, params=params, domain_rand=domain_rand, **kwargs)
