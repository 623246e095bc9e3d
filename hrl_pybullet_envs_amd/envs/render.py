"""Top-down `rgb_array` picture of one env, drawn on the host with numpy from the packed state record
(include/hrl_envs.h layout).  Debugging aid only: nothing here is on the step path."""
import numpy as np

from .. import _capi as K

_LEG = np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], float)       # assets/ant.xml:15-58 leg directions
_ANK = np.array([[-1, 1], [1, 1], [-1, 1], [1, 1]], float) / np.sqrt(2)  # ankle axes


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _axis_rot(a, th):
    a = np.asarray(a, float)
    c, s = np.cos(th), np.sin(th)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(a, a) + s * Kx


def ant_points(qpos):
    """World positions of the torso centre and, per leg, hip point, ankle point and foot tip."""
    p0, R0 = np.asarray(qpos[:3], float), _rot(qpos[3:7])
    legs = []
    for l in range(4):
        d = np.array([_LEG[l, 0], _LEG[l, 1], 0.0])
        Rx = R0 @ _axis_rot([0, 0, 1], qpos[7 + 2 * l])
        Rf = Rx @ _axis_rot([_ANK[l, 0], _ANK[l, 1], 0], qpos[8 + 2 * l])
        hip = p0 + R0 @ (0.2 * d)
        ank = hip + Rx @ (0.2 * d)
        legs.append((hip, ank, ank + Rf @ (0.4 * d)))
    return p0, legs


def draw_env(cfg, st, items, size=256):
    kind = cfg.env_kind
    if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ):
        hx, hy = 5.0, 9.0
    elif kind == K.HRL_ANT_FLAT:
        hx = hy = 6.0
    else:
        hx, hy = cfg.world_size[0] / 2, cfg.world_size[1] / 2
    cx, cy = (st[0], st[1]) if kind == K.HRL_ANT_FLAT else (0.0, 0.0)
    scale = (size - 1) / (2 * max(hx, hy))
    img = np.full((size, size, 3), 255, np.uint8)

    def px(x, y):
        return int(round((x - cx + max(hx, hy)) * scale)), int(round((max(hx, hy) - (y - cy)) * scale))

    def rect(x0, y0, x1, y1, col):
        (a, b), (c, d) = px(x0, y1), px(x1, y0)
        img[max(b, 0):max(d + 1, 0), max(a, 0):max(c + 1, 0)] = col

    def line(p, q, col, w=1):
        (a, b), (c, d) = px(*p), px(*q)
        n = max(abs(c - a), abs(d - b), 1)
        for t in np.linspace(0, 1, n + 1):
            u, v = int(round(a + (c - a) * t)), int(round(b + (d - b) * t))
            img[max(v - w, 0):v + w + 1, max(u - w, 0):u + w + 1] = col

    def disc(x, y, r, col):
        u, v = px(x, y)
        rr = max(int(round(r * scale)), 1)
        yy, xx = np.ogrid[:size, :size]
        img[(xx - u) ** 2 + (yy - v) ** 2 <= rr * rr] = col

    if kind != K.HRL_ANT_FLAT:  # walls 0.1 thick centred on +-size/2 (sizeable_enclosed_scene.py:46-57)
        for a, b in (((-hx, -hy), (hx, -hy)), ((hx, -hy), (hx, hy)), ((hx, hy), (-hx, hy)), ((-hx, hy), (-hx, -hy))):
            line(a, b, (60, 60, 60), 1)
    if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ):
        rect(-5, -2, 1, 2, (170, 170, 170))  # box.xml:19 at (-2, 0), maze_scene.py:12-13
    if items is not None:
        n = cfg.n_food + cfg.n_poison
        for i in range(n):
            x, y = items[2 * i], items[2 * i + 1]
            rect(x - 0.125, y - 0.125, x + 0.125, y + 0.125, (0, 170, 0) if i < cfg.n_food else (210, 0, 0))
    if kind == K.HRL_POINT_GATHER:
        R = _rot(st[3:7])
        c = [st[:3] + R @ (0.35 * np.array(s)) for s in ((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0))]
        for a, b in zip(c, c[1:] + c[:1]):
            line(a[:2], b[:2], (0, 50, 200), 1)
    else:
        p0, legs = ant_points(st[:15])
        for hip, ank, tip in legs:
            line(p0[:2], hip[:2], (120, 80, 20), 1); line(hip[:2], ank[:2], (150, 100, 30), 1); line(ank[:2], tip[:2], (200, 140, 40), 1)
        disc(p0[0], p0[1], 0.25, (0, 50, 200))
    return img
