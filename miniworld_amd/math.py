"""Host-side geometry helpers with the reference's ``miniworld.math`` API (math.py:1-62).

Used by world generation (placement runs once per episode on the host for the single-env
API); the per-step collision test lives in the HIP step kernel (csrc/mw_setup.hip).
"""
import math

import numpy as np

X_VEC = np.array([1, 0, 0])
Y_VEC = np.array([0, 1, 0])
Z_VEC = np.array([0, 0, 1])


def gen_rot_matrix(axis, angle):
    """Counter-clockwise rotation about ``axis`` by ``angle`` radians (Euler-Rodrigues)."""
    axis = axis / math.sqrt(np.dot(axis, axis))
    a = math.cos(angle / 2.0)
    b, c, d = -axis * math.sin(angle / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    return np.array([
        [aa + bb - cc - dd, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), aa + cc - bb - dd, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), aa + dd - bb - cc],
    ])


def intersect_circle_segs(point, radius, segs):
    """True if the circle (ground-plane projection) touches any segment, else None."""
    p = np.array([point[0], 0, point[2]])
    a, b = segs[:, 0, :], segs[:, 1, :]
    ab, ap = b - a, p - a
    t = np.clip(np.sum(ap * ab, axis=1) / np.sum(ab * ab, axis=1), 0, 1)
    closest = a + t[:, None] * ab
    if np.any(np.linalg.norm(closest - p, axis=1) < radius):
        return True
    return None
