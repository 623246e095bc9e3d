"""States in which a leg capsule meets a convex box by its MID-SECTION (assets/ant.xml:16-55 capsules against assets/food.xml:12 cubes and the
assets/box.xml:12 maze box): shared by the emulator (CPU) and device parity tests.  Test infrastructure only."""
import ctypes as C

import numpy as np

import orc


def leg_points(model, q):
    """[leg][hip point, ankle point, tip][xyz] in world coordinates (fp64 oracle kinematics)"""
    out = np.zeros(36)
    orc.lib().orc_ant_leg_points_f64(C.byref(model), orc.ptr(np.ascontiguousarray(q, np.float64)), orc.ptr(out))
    return out.reshape(4, 3, 3)


def end_sphere_clearance(model, q, lo, hi):
    """distance of the 13 end-point spheres of pose q to the box [lo, hi]: what the pre-capsule collider set tested"""
    ends = np.vstack([np.asarray(q[:3], np.float64)[None], leg_points(model, q).reshape(12, 3)])
    return np.sqrt(((ends - np.clip(ends, lo, hi)) ** 2).sum(1)) - np.r_[0.25, np.full(12, 0.08)]


def cubes_under_the_feet(o, rng):
    """Every env of the gather OracleEnv `o` (already reset): level torso, ankles near their stops, the first four items (food) under the foot
    capsules 25-50 % of the way from ankle to tip, the torso 0-3 cm below the height at which the feet touch them; random yaw, small joint and
    velocity noise.  Returns the number of envs (all) -- state and items are written in place."""
    n = o.N
    ni = o.cfg.n_food + o.cfg.n_poison
    for i in range(n):
        q = np.zeros(15); q[6] = 1.0
        yaw = rng.uniform(-np.pi, np.pi)
        q[5], q[6] = np.sin(yaw / 2), np.cos(yaw / 2)
        q[:2] = rng.uniform(-4, 4, 2)
        q[7:] = np.radians([0, 30, 0, -30, 0, -30, 0, 30]) + rng.uniform(-0.03, 0.03, 8) + np.radians([0, 4, 0, -4, 0, -4, 0, 4])
        q[2] = 1.0
        pts = leg_points(o.cfg.model, q)
        frac = rng.uniform(0.25, 0.5, 4)[:, None]
        cen = pts[:, 1, :2] + frac * (pts[:, 2, :2] - pts[:, 1, :2])
        # height at which the lowest foot axis is 0.08 above its cube: scan downwards over the axis samples
        ts = np.linspace(0, 1, 201)[:, None]
        best = 0.0
        for l in range(4):
            P = pts[l, 1] + ts * (pts[l, 2] - pts[l, 1])
            lo = np.r_[cen[l] - 0.125, -0.025]; hi = np.r_[cen[l] + 0.125, 0.225]
            inside = np.all((P[:, :2] >= lo[:2] - 0.08) & (P[:, :2] <= hi[:2] + 0.08), axis=1)
            drop = (P[inside, 2] - 0.225 - 0.08).min()  # how far the torso can come down before this foot touches (flat-top estimate)
            best = drop if l == 0 else min(best, drop)
        q[2] = 1.0 - best - rng.uniform(0.0, 0.03)
        items = np.full((ni, 2), 0.0)
        far = rng.uniform(5.5, 7.0, (ni, 2)) * np.where(rng.rand(ni, 2) < 0.5, -1, 1)
        items[:] = far
        items[:4] = cen
        o.state[i, :15] = q.astype(np.float32)
        o.state[i, 15:29] = rng.normal(size=14).astype(np.float32) * 0.05
        o.state[i, 30] = 0.75
        o.items[i, :2 * ni] = items.reshape(-1).astype(np.float32)
    return n


def foot_across_the_maze_corner(o, rng):
    """Every env of the maze OracleEnv `o`: leg 0's foot capsule lies across one of the four vertical edges of the maze box [-5, 1] x [-2, 2]
    (maze_scene.py:12-13), its axis passing -4 .. +6 cm outside the edge, the end spheres clear; the leg that would poke into the box folded."""
    n = o.N
    corners = np.array([[1.0, -2.0], [1.0, 2.0], [-5.0, 2.0], [-5.0, -2.0]])
    outs = np.array([[1.0, -1.0], [1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0]]) / np.sqrt(2)
    for i in range(n):
        c = rng.randint(4)
        # torso yaw such that leg 0 (torso-frame direction (1,1)/sqrt 2) runs along the tangent of the corner, +- 0.3 rad
        tang = np.array([-outs[c][1], outs[c][0]])
        yaw = np.arctan2(tang[1], tang[0]) - np.pi / 4 + rng.uniform(-0.3, 0.3)
        q = np.zeros(15); q[5], q[6] = np.sin(yaw / 2), np.cos(yaw / 2)
        q[7:] = np.radians([0, 30, 0, -100, 0, -100, 0, 100]) + rng.uniform(-0.02, 0.02, 8) + np.radians([0, 3, 0, 3, 0, 3, 0, -3])
        q[2] = rng.uniform(0.7, 1.2)
        pts = leg_points(o.cfg.model, q)
        mid = pts[0, 1, :2] + rng.uniform(0.35, 0.65) * (pts[0, 2, :2] - pts[0, 1, :2])
        q[:2] = corners[c] + rng.uniform(-0.04, 0.06) * outs[c] - (mid - q[:2])
        o.state[i, :15] = q.astype(np.float32)
        o.state[i, 15:29] = rng.normal(size=14).astype(np.float32) * 0.05
        o.state[i, 30] = 0.25
    return n


def count_mid_section_contacts(o, rows, boxes_of):
    """Among envs `rows`: contacts of the first substep (fp64 oracle) with a box that NO end-point sphere is within contact distance of.
    boxes_of(i) -> {surface code: (lo, hi)}."""
    seen = 0
    gather = o.cfg.n_food + o.cfg.n_poison if o.items is not None and o.cfg.env_kind in (1, 3) else 0
    for i in rows:
        q = o.state[i, :15].astype(np.float64); info = np.zeros(3, np.int32); dbg = np.zeros(13, np.int32)
        it = o.items[i, :2 * gather].astype(np.float64) if gather else None
        orc.lib().orc_ant_substeps_items_f64(C.byref(o.cfg), orc.ptr(q.copy()), orc.ptr(np.zeros(14)), orc.ptr(np.zeros(8)), 1,
                                             orc.ptr(it) if gather else None, gather, orc.ptr(info), orc.ptr(dbg), None)
        boxes = boxes_of(i)
        for s in {int(s) for s in dbg[1:1 + info[2]]}:
            if s in boxes and np.all(end_sphere_clearance(o.cfg.model, q, *boxes[s]) >= o.cfg.model.contact_dist):
                seen += 1
    return seen


def count_second_points(o, rows):
    """kept contacts of the first substep's collision pass (fp64 oracle kinematics of the fp32 state) that are SECOND support points"""
    gather = o.cfg.n_food + o.cfg.n_poison if o.cfg.env_kind in (1, 3) else 0
    n = 0
    for i in rows:
        q = o.state[i, :15].astype(np.float64)
        it = o.items[i, :2 * gather].astype(np.float64) if gather else None
        n += orc.lib().orc_ant_second_points_f64(C.byref(o.cfg), orc.ptr(q), orc.ptr(it) if gather else None, gather)
    return n


def feet_flat_against_the_maze_box(o, rng):
    """Every env of the maze OracleEnv `o`: the ant beside one of the four vertical faces of the maze box [-5, 1] x [-2, 2] x [0, 2] (maze_scene.py:12-13,
    assets/box.xml:12), the feet of the two legs that point towards the box hanging (nearly) straight down -- their axes run alongside the face, up to
    1 cm inside its contact shell or 1.5 cm outside, a degree or two off parallel: capsules that lie FLAT on a face (second support points)."""
    n = o.N
    for i in range(n):
        k = rng.randint(4)   # face: x = 1 (ant to the east), y = 2 (north), x = -5 (west), y = -2 (south): the configuration turned by k * 90 degrees
        yaw = k * np.pi / 2 + rng.uniform(-0.02, 0.02)
        q = np.zeros(15); q[5], q[6] = np.sin(yaw / 2), np.cos(yaw / 2)
        off = rng.uniform(-1.5, 1.5, 4)
        q[7:] = np.radians([0, 45, 0, -90 + off[0], 0, -90 + off[1], 0, 45]) + np.r_[rng.uniform(-0.02, 0.02), 0, rng.uniform(-0.02, 0.02), 0, rng.uniform(-0.02, 0.02), 0, rng.uniform(-0.02, 0.02), 0]
        q[2] = rng.uniform(0.9, 1.6)
        gap = rng.uniform(-0.01, 0.015)
        d = 0.08 + gap + 0.4   # the two ankle points are 0.4 towards the box from the torso centre (level torso)
        along = rng.uniform(-1.2, 1.2)
        q[:2] = [(1 + d, along), (along * 1.5 - 2.0, 2 + d), (-5 - d, along), (along * 1.5 - 2.0, -2 - d)][k]
        o.state[i, :15] = q.astype(np.float32)
        o.state[i, 15:29] = rng.normal(size=14).astype(np.float32) * 0.05
        o.state[i, 30] = 0.25
    return n


def feet_flat_on_cubes(o, rng):
    """Every env of the gather OracleEnv `o`: legs stretched out level (ankles near 0 -- outside their range, a state like any other: the limit rows push
    back), the torso so low that the foot capsules LIE on the tops of cubes put under them (assets/food.xml:12: tops at z = 0.225), within the contact
    shell: flat capsule-on-face contacts with second support points at the far edge of the cube's top."""
    n = o.N
    ni = o.cfg.n_food + o.cfg.n_poison
    for i in range(n):
        yaw = rng.uniform(-np.pi, np.pi)
        q = np.zeros(15); q[5], q[6] = np.sin(yaw / 2), np.cos(yaw / 2)
        q[:2] = rng.uniform(-4, 4, 2)
        sg = np.array([1, -1, -1, 1.0])   # the ankles' positive sense, leg by leg (assets/ant.xml:21,32,43,54)
        q[7::2] = rng.uniform(-0.05, 0.05, 4)
        q[8::2] = sg * rng.uniform(-0.015, 0.03, 4)
        q[2] = 0.225 + 0.08 + rng.uniform(-0.008, 0.012)
        pts = leg_points(o.cfg.model, q)
        frac = rng.uniform(0.3, 0.7, 4)[:, None]
        cen = pts[:, 1, :2] + frac * (pts[:, 2, :2] - pts[:, 1, :2])
        items = rng.uniform(5.5, 7.0, (ni, 2)) * np.where(rng.rand(ni, 2) < 0.5, -1, 1)
        items[:4] = cen
        o.state[i, :15] = q.astype(np.float32)
        o.state[i, 15:29] = rng.normal(size=14).astype(np.float32) * 0.05
        o.state[i, 30] = 0.75
        o.items[i, :2 * ni] = items.reshape(-1).astype(np.float32)
    return n
