"""Asset lookup for the engine: texture variants and mesh files.

Replaces ``miniworld.utils.get_file_path`` + the file probing in ``Texture.get``
(utils.py:14-37, opengl.py:124-140).  Search order:
  1. the packed container ``miniworld_amd/assets/assets_v1.npz`` (subset used by the BASELINE
     configs, produced by tools/pack_assets.py from the reference's Apache-2.0 data files);
  2. directories listed in ``$MINIWORLD_ASSET_PATH`` (``:``-separated), each laid out like the
     reference package (``textures/<name>_<i>.png``, ``meshes/<name>.obj|mtl``);
  3. an installed ``miniworld`` package, if present.
"""
from __future__ import annotations

import os

import numpy as np

_PACK_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "assets_v1.npz")
_pack = None
_tex_cache: dict = {}


def _pack_file():
    global _pack
    if _pack is None:
        _pack = np.load(_PACK_PATH) if os.path.exists(_PACK_PATH) else {}
    return _pack


def _asset_dirs():
    dirs = [d for d in os.environ.get("MINIWORLD_ASSET_PATH", "").split(":") if d]
    try:
        import importlib.util
        spec = importlib.util.find_spec("miniworld")
        if spec and spec.submodule_search_locations:
            dirs.extend(spec.submodule_search_locations)
    except Exception:
        pass
    return dirs


def _find_file(sub_dir, file_name):
    for d in _asset_dirs():
        p = os.path.join(d, sub_dir, file_name)
        if os.path.exists(p):
            return p
    return None


def texture_variants(tex_name: str) -> list:
    """Names ``tex_name_1 .. tex_name_k`` that exist, consecutive from 1 (opengl.py:127-132)."""
    out = []
    pack = _pack_file()
    for i in range(1, 10):
        v = f"{tex_name}_{i}"
        if ("tex:" + v) in pack or _find_file("textures", v + ".png"):
            out.append(v)
        else:
            break
    if not out:
        raise ValueError(f'failed to load textures for name "{tex_name}"')
    return out


def texture_rgb_top_down(variant: str) -> np.ndarray:
    """``variant`` is a texture variant ``name_i`` (textures/, may contain a sub-directory such as
    ``chars/ch_0x66_1``) or ``mesh:<name>`` for the map_Kd image of a mesh (meshes/<name>.png)."""
    if variant not in _tex_cache:
        pack = _pack_file()
        key = "tex:" + variant
        if key in pack:
            arr = pack[key]
        else:
            if variant.startswith("mesh:"):
                path = _find_file("meshes", variant[5:] + ".png")
            else:
                path = _find_file("textures", variant + ".png")
            if path is None:
                raise FileNotFoundError(f"texture {variant!r} not found in the asset pack or $MINIWORLD_ASSET_PATH")
            from PIL import Image
            with Image.open(path) as im:
                arr = np.asarray(im.convert("RGBA"))[:, :, :3]   # alpha dropped: glTexImage2D(GL_RGB) opengl.py:161-171
        _tex_cache[variant] = np.ascontiguousarray(arr, np.uint8)
    return _tex_cache[variant]


def texture_rgb_bottom_up(variant: str) -> np.ndarray:
    """Rows bottom-up, the order pyglet hands them to glTexImage2D (opengl.py:156-171)."""
    return np.ascontiguousarray(texture_rgb_top_down(variant)[::-1])


def texture_size(variant: str):
    a = texture_rgb_top_down(variant)
    return a.shape[1], a.shape[0]


def mesh_sources(mesh_name: str):
    """(obj_text, materials) for ``mesh_name`` where materials maps name -> {"Kd": array}.
    Mirrors ObjMesh._load_mtl's lookup rule: the MTL next to the OBJ with the OBJ's own base
    name (objmesh.py:234), not the ``mtllib`` line."""
    pack = _pack_file()
    base = mesh_name.split("_")[0]
    if ("obj:" + base) in pack and ("kd:" + mesh_name) in pack:
        text = bytes(pack["obj:" + base]).decode()
        return text, {"TheMaterial": {"Kd": np.array(pack["kd:" + mesh_name], np.float64)}}
    if ("obj:" + mesh_name) in pack:
        # meshes without an MTL: only the default material, textured iff meshes/<name>.png exists
        # (objmesh.py:222-231)
        default = {"Kd": np.array([1.0, 1.0, 1.0])}
        if ("tex:mesh:" + mesh_name) in pack:
            default["map_Kd"] = "mesh:" + mesh_name
        return bytes(pack["obj:" + mesh_name]).decode(), {"": default}
    path = _find_file("meshes", mesh_name + ".obj")
    if path is None:
        raise FileNotFoundError(f"mesh {mesh_name!r} not found in the asset pack or $MINIWORLD_ASSET_PATH")
    with open(path) as f:
        text = f.read()
    mats = {"": {"Kd": np.array([1.0, 1.0, 1.0])}}
    if os.path.exists(os.path.splitext(path)[0] + ".png"):      # default material's texture (objmesh.py:227-231)
        mats[""]["map_Kd"] = "mesh:" + mesh_name
    mtl = os.path.splitext(path)[0] + ".mtl"
    if os.path.exists(mtl):
        cur = None
        with open(mtl) as f:
            for line in f:
                tok = line.split()
                if not tok or tok[0].startswith("#"):
                    continue
                if tok[0] == "newmtl":
                    cur = {}
                    mats[tok[1]] = cur
                elif tok[0] == "Kd" and cur is not None:
                    cur["Kd"] = np.array([float(t) for t in tok[1:4]])
                elif tok[0] == "map_Kd" and cur is not None:
                    cur["map_Kd"] = "mesh:" + os.path.splitext(tok[-1])[0]
    return text, mats
