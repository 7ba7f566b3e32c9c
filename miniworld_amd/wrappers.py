"""Observation / action wrappers (the reference's miniworld/wrappers.py).

Single-env wrappers keep the reference's classes and semantics on the host.  For the batched
engine the two observation wrappers are not a post-pass: MiniWorldVecEnv(obs_layout="cwh" | "grey")
makes the raster kernel store the frame directly in the wrapper's layout (mw_set_obs_layout), and
`stochastic_actions` is the batched StochasticActionWrapper.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .gymshim import gym


class PyTorchObsWrapper(gym.ObservationWrapper):
    """Transpose the observation image tensors for PyTorch: (H, W, C) -> (C, W, H) (wrappers.py:7-25)."""

    def __init__(self, env):
        super().__init__(env)
        obs_shape = self.observation_space.shape
        self.observation_space = gym.spaces.Box(
            self.observation_space.low[0, 0, 0], self.observation_space.high[0, 0, 0],
            [obs_shape[2], obs_shape[1], obs_shape[0]], dtype=self.observation_space.dtype)

    def observation(self, observation):
        return observation.transpose(2, 1, 0)


class GreyscaleWrapper(gym.ObservationWrapper):
    """RGB -> greyscale, 0.30 R + 0.59 G + 0.11 B as float64 of shape (H, W, 1) (wrappers.py:28-46)."""

    def __init__(self, env):
        super().__init__(env)
        obs_shape = self.observation_space.shape
        self.observation_space = gym.spaces.Box(
            self.observation_space.low[0, 0, 0], self.observation_space.high[0, 0, 0],
            (obs_shape[0], obs_shape[1], 1), dtype=self.observation_space.dtype)

    def observation(self, obs):
        obs = 0.30 * obs[:, :, 0] + 0.59 * obs[:, :, 1] + 0.11 * obs[:, :, 2]
        return np.expand_dims(obs, axis=2)


class StochasticActionWrapper(gym.ActionWrapper):
    """With probability `prob` the given action is kept; otherwise `random_action`, or a uniform
    draw from the 6 first actions when that is None (wrappers.py:49-73)."""

    def __init__(self, env, prob: float = 0.9, random_action: Optional[int] = None):
        super().__init__(env)
        self.prob = prob
        self.random_action = random_action

    def action(self, action):
        if self.np_random.uniform() < self.prob:
            return action
        if self.random_action is None:
            return self.np_random.integers(0, 6)
        return self.random_action


def stochastic_actions(actions, prob: float = 0.9, random_action: Optional[int] = None, generator=None):
    """Batched StochasticActionWrapper.action for an int32 device tensor [N]; draws come from the
    given torch.Generator (device stream, not numpy's: batched envs have no per-env numpy rng)."""
    import torch
    keep = torch.rand(actions.shape, generator=generator, device=actions.device) < prob
    if random_action is None:
        other = torch.randint(0, 6, actions.shape, generator=generator, device=actions.device, dtype=actions.dtype)
    else:
        other = torch.full_like(actions, int(random_action))
    return torch.where(keep, actions, other)
