"""Mirror of hrl_pybullet_envs/utils.py for what the env constructors take from it.

The reference's spawn / debug-draw helpers (`get_player_cube`, `get_cube`, `get_sphere`, `debug_draw_point`,
utils.py:17-63) create pybullet bodies and GUI overlays; the batched simulator has no per-object scene graph, so only
`PositionEncoding` (utils.py:66-68, a constructor argument of the maze envs) has a counterpart here."""
from enum import Enum


class PositionEncoding(Enum):
    normed_vec = 0  # the two target entries of the observation are the unit vector robot -> target (ant_maze_bullet_env.py:128-129)
    angle = 1       # they are (sin, cos) of the target's bearing relative to the robot's heading (:130-131)
