"""OBJ/MTL loader producing the per-face vertex arrays the engine uploads (mw_upload_mesh).

Behavioural twin of the reference's ``ObjMesh`` (objmesh.py:11-292): triangles only, faces
grouped by material (stable sort on the material name), vertex colour = material ``Kd``,
and the re-centring step including its quirk — the "maximum" used for the x/z midpoint is
``verts.max(axis=0).min(axis=0)`` (objmesh.py:175), i.e. the smallest of the per-corner
maxima, which is what determines ``max_coords`` and therefore MeshEnt radius / scale.
"""
from __future__ import annotations

import numpy as np

from . import assets


class ObjMesh:
    cache: dict = {}

    @classmethod
    def get(cls, mesh_name: str) -> "ObjMesh":
        if mesh_name not in cls.cache:
            cls.cache[mesh_name] = ObjMesh(mesh_name)
        return cls.cache[mesh_name]

    def __init__(self, mesh_name: str):
        self.name = mesh_name
        text, materials = assets.mesh_sources(mesh_name)
        materials = dict(materials)
        materials.setdefault("", {"Kd": np.array([1, 1, 1])})
        pos, tex, nrm, faces = [], [], [], []
        current = ""
        for raw in text.splitlines():
            line = raw.rstrip(" \r\n")
            if not line or line.startswith("#"):
                continue
            tok = [t for t in line.split(" ") if t.strip(" ") != ""]
            head, rest = tok[0], tok[1:]
            if head == "v":
                pos.append([float(t) for t in rest])
            elif head == "vt":
                tex.append([float(t) for t in rest])
            elif head == "vn":
                nrm.append([float(t) for t in rest])
            elif head == "usemtl":
                current = rest[0] if rest[0] in materials else ""
            elif head == "f":
                assert len(rest) == 3, "only triangle faces are supported"
                corners = []
                for t in rest:
                    idx = [int(i) for i in t.split("/") if i != ""]
                    assert len(idx) in (2, 3)
                    corners.append(idx)
                faces.append((corners, current))
        faces.sort(key=lambda f: f[1])
        n = len(faces)
        verts = np.zeros((n, 3, 3), np.float32)
        norms = np.zeros((n, 3, 3), np.float32)
        texcs = np.zeros((n, 3, 2), np.float32)
        colors = np.zeros((n, 3, 3), np.float32)
        chunks = []
        for f, (corners, mtl) in enumerate(faces):
            if not chunks or chunks[-1]["mtl_name"] != mtl:
                if chunks:
                    chunks[-1]["end_idx"] = f
                chunks.append({"mtl_name": mtl, "mtl": materials[mtl], "start_idx": f, "end_idx": None})
            kd = materials[mtl].get("Kd", np.array((1, 1, 1)))
            for k, idx in enumerate(corners):
                if len(idx) == 3:
                    v, t, nn = idx
                    texcs[f, k] = tex[t - 1]
                else:
                    v, nn = idx
                    texcs[f, k] = (0, 0)
                verts[f, k] = pos[v - 1]
                norms[f, k] = nrm[nn - 1]
                colors[f, k] = kd
        chunks[-1]["end_idx"] = n
        lo = verts.min(axis=0).min(axis=0)
        hi_quirk = verts.max(axis=0).min(axis=0)        # sic — see module docstring
        mid = (lo + hi_quirk) / 2
        verts[:, :, 1] -= lo[1]
        verts[:, :, 0] -= mid[0]
        verts[:, :, 2] -= mid[2]
        self.min_coords = verts.min(axis=0).min(axis=0)
        self.max_coords = verts.max(axis=0).max(axis=0)
        self.verts, self.norms, self.texcs, self.colors = verts, norms, texcs, colors
        self.chunks = chunks
        self.num_faces = n
        # map_Kd texture of the chunks (objmesh.py:209-216).  The engine keeps one texture per mesh:
        # every mesh the environments use is a single chunk.
        texs = {c["mtl"].get("map_Kd") for c in chunks}
        if len(texs) > 1:
            raise NotImplementedError(f"mesh {mesh_name!r}: chunks with different textures are not supported")
        self.tex_variant = texs.pop() if texs else None
