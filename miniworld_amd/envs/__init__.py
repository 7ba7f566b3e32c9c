"""Environment classes and their Gymnasium ids (all 24 ids of envs/__init__.py:44-157)."""
from ..gymshim import gym
from .fourrooms import FourRooms
from .hallway import Hallway
from .maze import Maze, MazeS2, MazeS3, MazeS3Fast
from .oneroom import OneRoom, OneRoomS6, OneRoomS6Fast
from .pickupobjects import PickupObjects
from .putnext import PutNext
from .roomobjects import RoomObjects
from .tmaze import TMaze, TMazeLeft, TMazeRight
from .ymaze import YMaze, YMazeLeft, YMazeRight
from .collecthealth import CollectHealth
from .sidewalk import Sidewalk
from .sign import Sign
from .threerooms import ThreeRooms
from .wallgap import WallGap

__all__ = ["CollectHealth", "Sidewalk", "Sign", "ThreeRooms", "WallGap", "YMaze", "YMazeLeft", "YMazeRight", "FourRooms", "PutNext", "RoomObjects", "TMaze", "TMazeLeft", "TMazeRight", "Hallway", "Maze", "MazeS2", "MazeS3", "MazeS3Fast", "OneRoom", "OneRoomS6", "OneRoomS6Fast",
           "PickupObjects"]

ENV_IDS = {
    "MiniWorld-CollectHealth-v0": "CollectHealth",
    "MiniWorld-Sidewalk-v0": "Sidewalk",
    "MiniWorld-Sign-v0": "Sign",
    "MiniWorld-ThreeRooms-v0": "ThreeRooms",
    "MiniWorld-WallGap-v0": "WallGap",
    "MiniWorld-YMaze-v0": "YMaze",
    "MiniWorld-YMazeLeft-v0": "YMazeLeft",
    "MiniWorld-YMazeRight-v0": "YMazeRight",
    "MiniWorld-FourRooms-v0": "FourRooms",
    "MiniWorld-PutNext-v0": "PutNext",
    "MiniWorld-RoomObjects-v0": "RoomObjects",
    "MiniWorld-TMaze-v0": "TMaze",
    "MiniWorld-TMazeLeft-v0": "TMazeLeft",
    "MiniWorld-TMazeRight-v0": "TMazeRight",
    "MiniWorld-Hallway-v0": "Hallway",
    "MiniWorld-Maze-v0": "Maze",
    "MiniWorld-MazeS2-v0": "MazeS2",
    "MiniWorld-MazeS3-v0": "MazeS3",
    "MiniWorld-MazeS3Fast-v0": "MazeS3Fast",
    "MiniWorld-OneRoom-v0": "OneRoom",
    "MiniWorld-OneRoomS6-v0": "OneRoomS6",
    "MiniWorld-OneRoomS6Fast-v0": "OneRoomS6Fast",
    "MiniWorld-PickupObjects-v0": "PickupObjects",
}

_MODULE_OF = {"CollectHealth": "collecthealth", "Sidewalk": "sidewalk", "Sign": "sign", "ThreeRooms": "threerooms", "WallGap": "wallgap", "YMaze": "ymaze", "YMazeLeft": "ymaze", "YMazeRight": "ymaze", "FourRooms": "fourrooms", "PutNext": "putnext", "RoomObjects": "roomobjects", "TMaze": "tmaze",
              "TMazeLeft": "tmaze", "TMazeRight": "tmaze","Hallway": "hallway", "Maze": "maze", "MazeS2": "maze", "MazeS3": "maze", "MazeS3Fast": "maze",
              "OneRoom": "oneroom", "OneRoomS6": "oneroom", "OneRoomS6Fast": "oneroom",
              "PickupObjects": "pickupobjects"}

for _id, _cls in ENV_IDS.items():
    try:
        gym.register(id=_id, entry_point=f"miniworld_amd.envs.{_MODULE_OF[_cls]}:{_cls}")
    except Exception:  # already registered (re-import)
        pass
