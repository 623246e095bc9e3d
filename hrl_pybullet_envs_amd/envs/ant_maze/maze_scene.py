"""MazeScene -- the bounding lines of hrl_pybullet_envs/envs/ant_maze/maze_scene.py:8-38: a 10 x 18 arena and three of the four sides of the
6 x 4 box that sits against its x = -5 wall (the side at y = +2 is not among the reference's sensor lines, :18-21; the kernel's collision box
has all of its faces)."""
from ..intersection_utils import Point
from ..sizeable_enclosed_scene import SizeableEnclosedScene


class MazeScene(SizeableEnclosedScene):
    def __init__(self):
        super().__init__((10, 18), (0, 0))
        self.box_size = (6, 4)
        self.box_pos = (-(self.size[0] - self.box_size[0]) / 2, 0)
        hi = Point(self.box_pos[0] + self.box_size[0] / 2, self.box_pos[1] + self.box_size[1] / 2)
        lo = Point(self.box_pos[0] - self.box_size[0] / 2, self.box_pos[1] - self.box_size[1] / 2)
        self.box_bounds = [(hi, Point(hi.x, lo.y)), (lo, Point(lo.x, hi.y)), (lo, Point(hi.x, lo.y))]
