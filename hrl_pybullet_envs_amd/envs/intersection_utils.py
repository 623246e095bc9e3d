"""Host-side mirror of hrl_pybullet_envs/envs/intersection_utils.py (Point, segment_intersection, inf_intersection, quadrant, cart2pol,
pol2cart): the 2-D helpers behind the wall and target sensors.  The sensors themselves run in the kernel (csrc/step_core.h: wall_sensor_bin,
segment_intersection) and in the oracle; these functions are for user code that imports them from the reference's module path, and they are
checked against values the reference's own functions produced (tests/golden/intersection.json)."""
import math
from collections import namedtuple

Point = namedtuple('Point', ['x', 'y'])


def _turn(a, b, c):
    """sign of the turn a -> b -> c: +1 clockwise, -1 counter-clockwise, 0 collinear (intersection_utils.py:22-38)"""
    v = float(b.y - a.y) * (c.x - b.x) - float(b.x - a.x) * (c.y - b.y)
    return (v > 0) - (v < 0)


def _within_box(a, q, b):
    """q inside the axis-aligned box spanned by a and b (for collinear triples: q on the segment a-b)"""
    return min(a.x, b.x) <= q.x <= max(a.x, b.x) and min(a.y, b.y) <= q.y <= max(a.y, b.y)


def segment_intersection(p1, q1, p2, q2):
    """True if the segments p1-q1 and p2-q2 share a point (intersection_utils.py:41-74)."""
    t1, t2, t3, t4 = _turn(p1, q1, p2), _turn(p1, q1, q2), _turn(p2, q2, p1), _turn(p2, q2, q1)
    if t1 != t2 and t3 != t4:
        return True   # the end points of each segment lie on different sides of the other
    touching = ((t1, p1, p2, q1), (t2, p1, q2, q1), (t3, p2, p1, q2), (t4, p2, q1, q2))   # a collinear end point lying on the other segment
    return any(t == 0 and _within_box(a, q, b) for t, a, q, b in touching)


def inf_intersection(p1, p2, p3, p4):
    """Where the infinite lines p1-p2 and p3-p4 meet, or None when they are parallel (an EXACT zero test, :86)."""
    dx12, dy12, dx34, dy34 = p1.x - p2.x, p1.y - p2.y, p3.x - p4.x, p3.y - p4.y
    det = dx12 * dy34 - dy12 * dx34
    if det == 0:
        return None
    c12, c34 = p1.x * p2.y - p1.y * p2.x, p3.x * p4.y - p3.y * p4.x
    return Point((c12 * dx34 - dx12 * c34) / det, (c12 * dy34 - dy12 * c34) / det)


def quadrant(p):
    """1 / 2 / 3 / 4 with the axes belonging to the first match in the order 1, 4, 2, 3 (:95-102); a NaN coordinate raises as in the reference."""
    for q, ok in ((1, p.x >= 0 and p.y >= 0), (4, p.x >= 0 and p.y <= 0), (2, p.x <= 0 and p.y >= 0), (3, p.x <= 0 and p.y <= 0)):
        if ok:
            return q
    raise Exception(f'This should never happen, attempting to get quadrant for point:{p}')


def cart2pol(x, y):
    return math.sqrt(x ** 2 + y ** 2), math.atan2(y, x)


def pol2cart(rho, phi):
    return rho * math.cos(phi), rho * math.sin(phi)
