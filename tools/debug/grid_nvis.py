"""How many primitives does the big-scene K1 list for the host-built grids of test_big_host_built_worlds_match_the_oracle?
(MW_K1_PROF's count of the last frame, per pose; MW_OCCLUSION=0 / 1.)  Debug helper."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import numpy as np
out = os.path.join(root, "gpurun_out", "k1grid.bin")
os.environ["MW_K1_PROF"] = out
from miniworld_amd.entity import Box
from miniworld_amd.miniworld import MiniWorldEnv

def make(rows, cols, open_plan=False):
    class Grid(MiniWorldEnv):
        def __init__(self, **kwargs):
            MiniWorldEnv.__init__(self, max_episode_steps=500, **kwargs)
        def _gen_world(self):
            rooms = [[self.add_rect_room(min_x=3.25 * i, max_x=3.25 * i + 3, min_z=3.25 * j, max_z=3.25 * j + 3) for i in range(cols)] for j in range(rows)]
            for j in range(rows):
                for i in range(cols):
                    if open_plan:
                        if i + 1 < cols: self.connect_rooms(rooms[j][i], rooms[j][i + 1], min_z=3.25 * j + 0.1, max_z=3.25 * j + 2.9)
                        if j + 1 < rows: self.connect_rooms(rooms[j][i], rooms[j + 1][i], min_x=3.25 * i + 0.1, max_x=3.25 * i + 2.9)
                        continue
                    if i + 1 < cols and (i + j) % 3 != 0:
                        self.connect_rooms(rooms[j][i], rooms[j][i + 1], min_z=3.25 * j + 0.5, max_z=3.25 * j + 2.5, **({"max_y": 2.2} if (i + j) % 2 else {}))
                    if j + 1 < rows and (i * 2 + j) % 4 != 0:
                        self.connect_rooms(rooms[j][i], rooms[j + 1][i], min_x=3.25 * i + 0.75, max_x=3.25 * i + 2.25, **({"max_y": 2.2} if (i + j) % 2 == 0 else {}))
            self.box = self.place_entity(Box(color="red"))
            if open_plan:
                import math
                self.place_agent(pos=np.array([0.6, 0.0, 0.6]), dir=-math.pi / 4)
            else:
                self.place_agent()
    return Grid

for occ in ("1", "0"):
    os.environ["MW_OCCLUSION"] = occ
    for rows, op in ((6, False), (8, False), (11, False), (6, True), (9, True), (11, True)):
        counts = []
        for steps in (0, 3, 7, 12, 18, 24):
            env = make(rows, rows, op)()
            env.reset(seed=3)
            g = np.random.default_rng(rows)
            for t in range(steps):
                env.step(int(g.choice([0, 1, 2, 2, 2])))
            env.close()
            counts.append(int(np.fromfile(out, np.uint64).reshape(-1, 8)[0, 5]))
        print("MW_OCCLUSION", occ, "grid", rows, "open" if op else "doors", "listed primitives at 6 poses:", counts)
