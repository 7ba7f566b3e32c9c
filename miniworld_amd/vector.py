"""Gymnasium VectorEnv-style adapter over MiniWorldVecEnv.

    envs = MiniWorldVectorEnv("MiniWorld-Hallway-v0", num_envs=4096)
    obs, infos = envs.reset(seed=0)
    obs, rewards, terminations, truncations, infos = envs.step(actions)

follows gymnasium.vector.VectorEnv's calling convention (num_envs, single_* / batched spaces,
reset -> (obs, infos), step -> 5-tuple, autoreset mode "same-step": the observation returned with
a finished episode is the first one of the next episode).  Observations, rewards and flags stay
torch tensors on the engine's GPU by default (`to_numpy=True` copies them to the host like a
classic VectorEnv); actions may be a torch tensor, a numpy array or a list.

**What a same-step consumer does NOT get: `final_obs`.**  Gymnasium's SAME_STEP mode promises the finished episode's last
observation under info["final_obs"]; the engine never draws it (the step's only frame is the next episode's first one —
the reference leaves resets to the caller and renders once per step, miniworld.py:670-730).  Code that bootstraps a value
from final_obs on truncation must take the returned observation's predecessor instead.  What IS there: info["_final_info"]
(bool[N], for every env family, also those without info keys) and — for the families that have info keys
(CollectHealth's health, TMaze / YMaze's goal_pos) — info["final_info"], the finished episodes' own values.  The arrays
under final_info are copies: they stay valid after the next step.
"""
from __future__ import annotations

import numpy as np

from .gymshim import AUTORESET_SAME_STEP, VectorEnvBase, batch_action_space, spaces
from .vec_env import MiniWorldVecEnv


class MiniWorldVectorEnv(VectorEnvBase):
    """A gymnasium.vector.VectorEnv when gymnasium is importable (a plain class with the same surface otherwise)."""
    metadata = {"autoreset_mode": AUTORESET_SAME_STEP, "render_modes": ["rgb_array"]}

    def __init__(self, env_id: str, num_envs: int, to_numpy: bool = False, **kwargs):
        self.vec = MiniWorldVecEnv(env_id, num_envs, **kwargs)
        self.num_envs = num_envs
        self.to_numpy = to_numpy
        shape, dtype = tuple(self.vec.obs.shape[1:]), {"grey": np.float64}.get(self.vec.obs_layout, np.uint8)
        self.single_observation_space = spaces.Box(0, 255, shape, dtype=dtype)
        self.observation_space = spaces.Box(0, 255, (num_envs,) + shape, dtype=dtype)
        self.single_action_space = spaces.Discrete(self.vec.n_actions)
        self.action_space = batch_action_space(self.single_action_space, num_envs)
        self.render_mode = "rgb_array"
        self.closed = False

    def _out(self, t):
        return t.cpu().numpy() if self.to_numpy else t

    def _infos(self):
        """gymnasium's batched info convention: one array per key (health / goal_pos: MiniWorldVecEnv.infos)."""
        return {k: self._out(v) for k, v in self.vec.infos().items()}

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        """Env i is seeded with seed + i (gymnasium's convention for an integer seed)."""
        obs = self.vec.reset(seed)
        return self._out(obs), {}

    def step(self, actions):
        torch = self.vec.torch
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.vec.engine.device)
        actions = actions.to(device=self.vec.engine.device, dtype=torch.int32)
        obs, rew, term, trunc = self.vec.step(actions)
        info = self._infos()
        if self.vec.autoreset:
            # gymnasium's same-step convention: "_final_info" masks the envs whose episode ended with this step (every family); the
            # finished episodes' own info under "final_info" where the family has info keys — clones, the engine's buffers are
            # rewritten by the next step.  (No "final_obs": see the module's docstring.)
            done = self._out((term | trunc).bool())
            info["_final_info"] = done
            if info.keys() - {"_final_info"}:
                final = {k: (self._out(v) if self.to_numpy else v.clone()) for k, v in self.vec.final_infos().items()}
                final.update({"_" + k: done for k in list(final)})
                info["final_info"] = final
        return self._out(obs), self._out(rew), self._out(term.bool()), self._out(trunc.bool()), info

    def render(self):
        """Tuple-free batched render: the map view of every env (uint8[N, H, W, 3])."""
        return self._out(self.vec.render_top_view())

    def get_visible_ents(self):
        return self._out(self.vec.get_visible_ents())

    def close(self):
        if not self.closed:
            self.vec.close()
            self.closed = True
