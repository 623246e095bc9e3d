"""Host-side restatement of the step's counter-based draws that a reference ATTRIBUTE needs (csrc/step_core.h: philox4x32,
u01, flag_goal): the shared goal list of a non-manual AntFlagrunBulletEnv is never stored -- the kernel regenerates goal k
of episode ep on demand -- so `env.goals` / `env.goal` evaluate the same function here, in numpy, fp32 operation by
operation.  Nothing on the step path calls this."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(seed, env, index, w2, w3):
    """key = (seed_lo, seed_hi ^ env_hi), counter = (env_lo, index, w2, w3); arguments broadcast; returns 4 uint32 arrays."""
    env = np.asarray(env, np.int64).astype(np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(env & _MASK, np.asarray(index, np.uint64) & _MASK, np.asarray(w2, np.uint64) & _MASK,
                                         np.asarray(w3, np.uint64) & _MASK)
    k0 = np.uint64(int(seed) & 0xFFFFFFFF)
    k1 = np.uint64((int(seed) >> 32) & 0xFFFFFFFF) ^ (env >> np.uint64(32))
    for r in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        c0, c1, c2, c3 = n0 & _MASK, p1 & _MASK, n2 & _MASK, p0 & _MASK
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def u01(x):
    return (np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-08)


def flag_goal(seed, flag_size, ep, k):
    """Goal k of episode ep of the shared list (ant_flagrun_env.py:71-78 with counter-based draws; step_core.h flag_goal):
    uniform in the square, redrawn while within 0.5 of the origin, at most 64 attempts.  ep, k broadcast; returns [..., 2] fp32."""
    ep, k = np.broadcast_arrays(np.asarray(ep, np.int64).astype(np.uint64), np.asarray(k, np.int64).astype(np.uint64))
    size = np.float32(flag_size)
    gx = np.zeros(ep.shape, np.float32); gy = np.zeros(ep.shape, np.float32)
    todo = np.ones(ep.shape, bool)
    for a in range(64):
        r0, r1, _, _ = philox4x32(seed, np.zeros(ep.shape, np.int64), ep, (np.uint64(4) << np.uint64(16)) | k, np.full(ep.shape, a, np.uint64))
        x = -size / np.float32(2) + size * u01(r0)
        y = -size / np.float32(2) + size * u01(r1)
        gx = np.where(todo, x, gx); gy = np.where(todo, y, gy)
        todo = todo & (np.sqrt(gx * gx + gy * gy, dtype=np.float32) < np.float32(0.5))
        if not todo.any():
            break
    return np.stack([gx, gy], axis=-1)
