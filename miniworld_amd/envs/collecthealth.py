"""MiniWorld-CollectHealth-v0: collect med-kits to stay alive (collecthealth.py:6-107).  The kits are
textured meshes; a picked-up kit is re-placed with the episode's own random stream."""
from ..entity import MeshEnt
from ..gymshim import EzPickle
from ..miniworld import MiniWorldEnv


class CollectHealth(MiniWorldEnv, EzPickle):
    def __init__(self, size=16, **kwargs):
        assert size >= 2
        self.size = size
        MiniWorldEnv.__init__(self, max_episode_steps=1000, **kwargs)
        EzPickle.__init__(self, size, **kwargs)

    def _gen_world(self):
        self.add_rect_room(min_x=0, max_x=self.size, min_z=0, max_z=self.size, wall_tex="cinder_blocks", floor_tex="slime")
        for _ in range(18):
            self.box = self.place_entity(MeshEnt(mesh_name="medkit", height=0.40, static=False))
        self.place_agent()
        self.health = 100

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        self.health -= 2
        if action == self.actions.pickup:
            if self.agent.carrying:     # the kit is consumed and respawns somewhere else (collecthealth.py:86-90)
                self.entities.remove(self.agent.carrying)
                self.place_entity(self.agent.carrying)
                self.agent.carrying = None
                self.health = 100
        if self.health > 0:
            reward = 2
        else:
            reward = -100
            termination = True
        info["health"] = self.health
        return obs, reward, termination, truncation, info
