"""MiniWorld-Maze-v0 and variants: recursive-backtracker maze of square rooms (maze.py:9-199)."""
from ..entity import Box
from ..gymshim import EzPickle, spaces
from ..miniworld import MiniWorldEnv
from ..params import DEFAULT_PARAMS


class Maze(MiniWorldEnv, EzPickle):
    def __init__(self, num_rows=8, num_cols=8, room_size=3, max_episode_steps=None, **kwargs):
        self.num_rows, self.num_cols, self.room_size = num_rows, num_cols, room_size
        self.gap_size = 0.25
        MiniWorldEnv.__init__(self, max_episode_steps=max_episode_steps or num_rows * num_cols * 24, **kwargs)
        EzPickle.__init__(self, num_rows=num_rows, num_cols=num_cols, room_size=room_size,
                          max_episode_steps=max_episode_steps, **kwargs)
        self.action_space = spaces.Discrete(self.actions.move_forward + 1)

    def _gen_world(self):
        pitch = self.room_size + self.gap_size
        grid = [[self.add_rect_room(min_x=i * pitch, max_x=i * pitch + self.room_size,
                                    min_z=j * pitch, max_z=j * pitch + self.room_size, wall_tex="brick_wall")
                 for i in range(self.num_cols)] for j in range(self.num_rows)]
        visited = set()

        def carve(i, j):
            """Depth-first carving; the visiting order of the 4 neighbours is drawn without
            replacement with np_random.choice, one draw per pick (maze.py:112-120)."""
            room = grid[j][i]
            visited.add(room)
            pool = [(0, 1), (0, -1), (-1, 0), (1, 0)]
            order = []
            while len(order) < 4:
                pick = pool[self.np_random.choice(len(pool))]
                pool.remove(pick)
                order.append(pick)
            for dj, di in order:
                ni, nj = i + di, j + dj
                if not (0 <= nj < self.num_rows and 0 <= ni < self.num_cols):
                    continue
                neighbor = grid[nj][ni]
                if neighbor in visited:
                    continue
                if di == 0:
                    self.connect_rooms(room, neighbor, min_x=room.min_x, max_x=room.max_x)
                elif dj == 0:
                    self.connect_rooms(room, neighbor, min_z=room.min_z, max_z=room.max_z)
                carve(ni, nj)

        carve(0, 0)
        self.box = self.place_entity(Box(color="red"))
        self.place_agent()

    def step(self, action):
        obs, reward, termination, truncation, info = super().step(action)
        if self.near(self.box):
            reward += self._reward()
            termination = True
        return obs, reward, termination, truncation, info


class MazeS2(Maze):
    def __init__(self, num_rows=2, num_cols=2, **kwargs):
        Maze.__init__(self, num_rows=num_rows, num_cols=num_cols, **kwargs)


class MazeS3(Maze):
    def __init__(self, num_rows=3, num_cols=3, **kwargs):
        Maze.__init__(self, num_rows=num_rows, num_cols=num_cols, **kwargs)


default_params = DEFAULT_PARAMS.no_random()
default_params.set("forward_step", 0.7)
default_params.set("turn_step", 45)


class MazeS3Fast(Maze):
    def __init__(self, num_rows=3, num_cols=3, max_episode_steps=300, params=default_params,
                 domain_rand=False, **kwargs):
        Maze.__init__(self, num_rows=num_rows, num_cols=num_cols, max_episode_steps=max_episode_steps,
                      params=params, domain_rand=domain_rand, **kwargs)
