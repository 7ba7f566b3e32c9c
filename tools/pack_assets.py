"""Builds miniworld_amd/assets/assets_v1.npz from the reference's data assets.

The engine needs the same texture / mesh *data* the reference ships (Apache-2.0,
/root/reference/miniworld/{textures,meshes}) to draw the same worlds; those files do not
exist on the GPU box, so the subset used by the BASELINE configs is re-packed here into
one engine-side container:
    tex:<name>   uint8[h, w, 3]  decoded RGB, rows top-down as stored in the PNG
                                 (alpha dropped exactly as glTexImage2D(GL_RGB) does,
                                  opengl.py:161-171)
    tex:mesh:<name>              the map_Kd image of a textured mesh (meshes/<name>.png)
    obj:<name>   uint8[...]      OBJ text (ball / key geometry is shared by all colours)
    kd:<name>    float64[3]      diffuse colour of <name>.mtl (objmesh.py:234 looks the MTL
                                 up by the OBJ's own name, not by its mtllib line)
Run in the build container only:  python tools/pack_assets.py
"""
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("MINIWORLD_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(__file__), "..", "miniworld_amd", "assets", "assets_v1.npz")

TEXTURES = [
    "floor_tiles_bw_1", "concrete_1", "concrete_2", "concrete_3", "concrete_4",
    "concrete_tiles_1", "brick_wall_1", "asphalt_1",
    # section 8(f) rank 4 environments: CollectHealth room, ThreeRooms picture, Sign letters (variant 1:
    # Sign never domain-randomises, sign.py:92-98)
    "cinder_blocks_1", "slime_1", "logo_mila_1",
] + [f"chars/ch_0x{ord(c)}_1" for c in "BLUERDGN"]
# textured single-chunk meshes (default material + meshes/<name>.png, objmesh.py:222-231)
TEXTURED_MESHES = ["building", "cone", "medkit", "duckie"]
COLORS = ["blue", "green", "grey", "purple", "red", "yellow"]


def main():
    items = {}
    for name in TEXTURES:
        path = os.path.join(REF, "miniworld", "textures", name + ".png")
        with Image.open(path) as im:
            rgba = np.asarray(im.convert("RGBA"))
        items["tex:" + name] = np.ascontiguousarray(rgba[:, :, :3])
    for base in ("ball", "key"):
        # geometry identical for every colour (checked with cmp); store it once
        with open(os.path.join(REF, "miniworld", "meshes", f"{base}_red.obj"), "rb") as f:
            items["obj:" + base] = np.frombuffer(f.read(), np.uint8)
        for col in COLORS:
            kd = None
            with open(os.path.join(REF, "miniworld", "meshes", f"{base}_{col}.mtl")) as f:
                for line in f:
                    tok = line.split()
                    if tok and tok[0] == "Kd":
                        kd = np.array([float(t) for t in tok[1:4]])
            items[f"kd:{base}_{col}"] = kd
    for name in TEXTURED_MESHES:
        with open(os.path.join(REF, "miniworld", "meshes", name + ".obj"), "rb") as f:
            items["obj:" + name] = np.frombuffer(f.read(), np.uint8)
        with Image.open(os.path.join(REF, "miniworld", "meshes", name + ".png")) as im:
            items["tex:mesh:" + name] = np.ascontiguousarray(np.asarray(im.convert("RGBA"))[:, :, :3])
    np.savez_compressed(OUT, **items)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    sys.exit(main())
