"""SizeableEnclosedScene -- what hrl_pybullet_envs/envs/sizeable_enclosed_scene.py:14-97 leaves on `env.scene` / `env.stadium_scene` for a user:
the bounding lines of the arena and `sense_walls`.  The walls the robots collide with and the wall sensor of the observation live in the kernel
(csrc/host_cfg.h: build_devcfg; csrc/step_core.h: wall_sensor_bin); this host-side mirror is for user code that reads the scene, and its
`sense_walls` is checked against the reference's own outputs (tests/golden/sense_walls.json)."""
import math

from .intersection_utils import Point, inf_intersection, pol2cart, quadrant


class SizeableEnclosedScene:
    def __init__(self, world_size, world_center=(0, 0)):
        self.size, self.center = tuple(world_size), tuple(world_center)
        hi = Point(self.size[0] / 2 + self.center[0], self.size[1] / 2 + self.center[1])
        lo = Point(-self.size[0] / 2 + self.center[0], -self.size[1] / 2 + self.center[1])
        # the arena's four sides in the reference's order (:28-34): from the (+, +) corner along y = +, along x = +, from (-, -) along x = -, along y = -
        self.world_bounds = [(hi, Point(lo.x, hi.y)), (hi, Point(hi.x, lo.y)), (lo, Point(lo.x, hi.y)), (lo, Point(hi.x, lo.y))]
        self.box_bounds = []

    @property
    def bounds(self):
        return self.world_bounds + self.box_bounds

    def sense_walls(self, s_bins, s_span, s_range, robot_pos, rob_yaw, debug=False):
        """One reading per ray, 1 - distance / range of the nearest bounding LINE the ray's line meets within range on the ray's side (:63-97):
        both the ray and the walls are infinite lines, "on the ray's side" is equality of quadrants, a span of 2 pi spaces the rays by
        (i + 1) / n, any other span by i / (n - 1)."""
        x0, y0 = float(robot_pos[0]), float(robot_pos[1])
        here = Point(x0, y0)
        full_turn = s_span == 2 * math.pi
        out = []
        for i in range(s_bins):
            frac = (i + 1) / s_bins if full_turn else i / (s_bins - 1)
            dx, dy = pol2cart(s_range, math.pi / 2 + rob_yaw + frac * s_span)
            tip = Point(x0 + dx, y0 + dy)
            side = quadrant(Point(tip.x - x0, tip.y - y0))
            best = 0
            for a, b in self.bounds:
                hit = inf_intersection(here, tip, a, b)
                if hit is None:
                    continue
                ex, ey = x0 - hit.x, y0 - hit.y
                dist = math.sqrt(ex * ex + ey * ey)
                if dist > s_range or quadrant(Point(hit.x - x0, hit.y - y0)) != side:
                    continue
                best = max(best, 1. - dist / s_range)
            out.append(best)
        return out
