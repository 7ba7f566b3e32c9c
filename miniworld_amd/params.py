"""Simulation parameters with domain-randomisation ranges.

Same table and API as the reference's ``miniworld.params`` (params.py:7-130:
``DomainParams.set / get_max / sample / sample_many / copy / no_random`` and
``DEFAULT_PARAMS``); ``sample`` consumes the numpy Generator exactly like the reference
(``uniform(min, max)`` for floats, ``integers(min, max + 1)`` for ints), which keeps
``reset(seed)`` stream-compatible.
"""
from __future__ import annotations

import copy as _copy
from collections import namedtuple

import numpy as np

DomainParam = namedtuple("DomainParam", ["default", "min", "max", "type"])


def _as_value(v, kind):
    if isinstance(v, (list, tuple)):
        v = np.array(v)
    if isinstance(v, np.ndarray) and kind == "float":
        v = v.astype("float")
    return v


class DomainParams:
    DomainParam = DomainParam

    def __init__(self):
        self.params = {}

    def copy(self):
        return _copy.deepcopy(self)

    def no_random(self):
        """Copy with every range collapsed onto its default."""
        out = self.copy()
        out.params = {k: DomainParam(p.default, p.default, p.default, p.type) for k, p in out.params.items()}
        return out

    def set(self, name, default, min=None, max=None, type="float"):  # noqa: A002
        default = _as_value(default, type)
        lo = default if min is None else _as_value(min, type)
        hi = default if max is None else _as_value(max, type)
        if isinstance(default, np.ndarray):
            assert lo.shape == hi.shape == default.shape
            assert np.all(hi >= default) and np.all(default >= lo)
        else:
            assert hi >= default >= lo
        prev = self.params.get(name)
        if prev is not None:
            assert prev.type == type
            if isinstance(prev.default, np.ndarray):
                assert default.shape == prev.default.shape
        self.params[name] = DomainParam(default, lo, hi, type)

    def get_max(self, name):
        return self.params[name].max

    def sample(self, rng, name):
        """Default value when ``rng`` is None (domain randomisation off), else one draw."""
        p = self.params[name]
        if rng is None:
            return p.default
        if p.type == "float":
            return rng.uniform(p.min, p.max)
        if p.type == "int":
            return rng.integers(p.min, p.max + 1)
        raise AssertionError(p.type)

    def sample_many(self, rng, target_obj, param_names):
        for name in param_names:
            setattr(target_obj, name, self.sample(rng, name))

    def as_ranges(self):
        """{name: (default, min, max)} for the engine configuration."""
        return {k: (p.default, p.min, p.max) for k, p in self.params.items()}


DEFAULT_PARAMS = DomainParams()
for _name, _d, _lo, _hi in [
    ("sky_color", [0.25, 0.82, 1], [0.1, 0.1, 0.1], [1.0, 1.0, 1.0]),
    ("light_pos", [0, 2.5, 0], [-40, 2.5, -40], [40, 5, 40]),
    ("light_color", [0.7, 0.7, 0.7], [0.45, 0.45, 0.45], [0.8, 0.8, 0.8]),
    ("light_ambient", [0.45, 0.45, 0.45], [0.35, 0.35, 0.35], [0.55, 0.55, 0.55]),
    ("obj_color_bias", [0, 0, 0], [-0.2, -0.2, -0.2], [0.2, 0.2, 0.2]),
    ("forward_step", 0.15, 0.12, 0.17),
    ("forward_drift", 0, -0.05, 0.05),
    ("turn_step", 15, 10, 20),
    ("bot_radius", 0.4, 0.38, 0.42),
    ("cam_pitch", 0, -5, 5),
    ("cam_fov_y", 60, 55, 65),
    ("cam_height", 1.5, 1.45, 1.55),
    ("cam_fwd_disp", 0, -0.05, 0.10),
]:
    DEFAULT_PARAMS.set(_name, _d, _lo, _hi)
