"""Texture registry — the role of ``miniworld.opengl.Texture`` (opengl.py:102-198) without GL.

``Texture.get(name, rng)`` resolves the variants ``name_1 .. name_k`` and, when a Generator is
given (domain randomisation), draws ``rng.integers(0, k)`` exactly like the reference
(opengl.py:136-138) so that seeded resets stay stream-compatible.  The pixels are uploaded to
the engine (mw_upload_texture) by whoever renders; this object only carries identity and size.
"""
from __future__ import annotations

from . import assets


class Texture:
    tex_cache: dict = {}

    def __init__(self, variant: str, tex_name: str):
        self.variant = variant
        self.name = tex_name
        self.width, self.height = assets.texture_size(variant)

    @classmethod
    def get(cls, tex_name, rng=None):
        variants = assets.texture_variants(tex_name)
        variant = variants[int(rng.integers(0, len(variants)))] if rng else variants[0]
        if variant not in cls.tex_cache:
            cls.tex_cache[variant] = Texture(variant, tex_name)
        return cls.tex_cache[variant]

    @classmethod
    def load(cls, variant: str):
        """A texture addressed directly instead of through the variant lookup: the map_Kd image of a
        mesh, ``mesh:<name>`` (objmesh.py:211 calls Texture.load(path))."""
        if variant not in cls.tex_cache:
            cls.tex_cache[variant] = Texture(variant, variant)
        return cls.tex_cache[variant]

    def rgb_bottom_up(self):
        return assets.texture_rgb_bottom_up(self.variant)

    def __repr__(self):
        return f"Texture({self.variant!r}, {self.width}x{self.height})"
