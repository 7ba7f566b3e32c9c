#!/usr/bin/env python3
"""Turns the reference's own screenshots into small numeric fixtures (tests/golden/screenshots.npz).

The only pixels in /root/reference that a real OpenGL driver produced are the JPEG screenshots of the
manual_control window under images/ (render(), miniworld.py:1340-1443): the 800x600 vis_fb view on the left
(x 1..800 behind the window frame), the agent's 80x60 observation blown up to 256x192 with GL_LINEAR at the top right (blit at
x = img_width, :1409-1421), and a text label with the pose (":1424-1430": pos to 2 decimals, angle in whole
degrees, step count).  This script crops the two views, box-filters them down (main view 4x4 -> 200x150, inset
-> 80x60) and stores them with the printed pose.  tests/test_oracle_vs_reference_screenshots.py renders the
CPU oracle at that pose and compares — the one external anchor the "parity unpinned" pixel oracle has.

maze_0.jpg is not used: its maze is random and unseeded (no floorplan to register against), and its aliased inset shows it
predates the mip-mapped textures of v2.1.0.  ymaze_0.jpg shows the top view (render_top_view) in the main pane.
Three more screenshots were tried and are older than the code under /root/reference: fourrooms_0 (its label prints
"angle: 488", i.e. it predates the "% 360" of miniworld.py:1427, and its walls are shaded by a positional light),
wallgap_0 (another sky colour, the same positional light) and collecthealth_0 (a smaller room than size=16 gives).

Run here (needs /root/reference); the npz travels with the repo.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "screenshots.npz")

# file -> (env class, printed pos, printed angle in degrees, printed step count): read off the label in the image
SHOTS = {
    "hallway_0": ("Hallway", (-0.02, 0.00, -0.03), 1, 20),
    "oneroom_0": ("OneRoom", (0.63, 0.00, 8.42), 30, 10),
    "pickupobjs_0": ("PickupObjects", (3.03, 0.00, 4.12), 289, 0),
    "sidewalk_0": ("Sidewalk", (-1.89, 0.00, 0.41), 298, 82),       # textured static meshes: the building and the cones
    "tmaze_0": ("TMaze", (3.41, 0.00, -0.31), 359, 124),
    "ymaze_0": ("YMaze", (-5.64, 0.00, 0.48), 5, 37),              # main view = render_top_view (manual_control --top_view)
}
TOP_PATCH = (300, 250, 400, 350)      # x0, y0, x1, y1 inside ymaze_0's main view: the horizontal arm, beside the agent
TITLE_BAR = 24        # rows of window decoration above the GL area (the JPEGs are 1058 x 625 = 1+800+256+1 x 24+600+1)
BORDER = 1            # the window frame left of the GL area (ymaze_0's 4-px checker correlates 0.999 at this offset, 0.81 at 0)


def main():
    out = {}
    for name, (cls, pos, ang, steps) in SHOTS.items():
        im = Image.open(os.path.join(REF, name + ".jpg")).convert("RGB")
        assert im.size == (1058, 625), im.size
        main_view = im.crop((BORDER, TITLE_BAR, BORDER + 800, TITLE_BAR + 600))
        inset = im.crop((801, TITLE_BAR, 801 + 256, TITLE_BAR + 192))
        out[f"{name}/main"] = np.asarray(main_view.resize((200, 150), Image.BOX), np.uint8)
        out[f"{name}/inset"] = np.asarray(inset.resize((80, 60), Image.BOX), np.uint8)
        if name == "ymaze_0":       # top view: a full-resolution patch of the floor (the checker is ~4 px wide there)
            out[f"{name}/patch"] = np.asarray(main_view.crop(TOP_PATCH), np.uint8)
            out[f"{name}/patch_box"] = np.array(TOP_PATCH, np.int64)
        out[f"{name}/env"] = np.array(cls)
        out[f"{name}/pos"] = np.array(pos, np.float64)
        out[f"{name}/angle_deg"] = np.array(ang, np.int64)
        out[f"{name}/steps"] = np.array(steps, np.int64)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
