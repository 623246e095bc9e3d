"""GPU box: BASELINE config 1, the README loop of the reference (one env through the gym-style class, numpy in / numpy out): host-latency-bound by construction."""
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from hrl_pybullet_envs_amd import AntGatherBulletEnv
env = AntGatherBulletEnv()
env.seed(0)
obs = env.reset()
rng = np.random.RandomState(0)
acts = rng.uniform(-1, 1, (1000, 8)).astype(np.float32)
for a in acts[:50]: env.step(a)
torch.cuda.synchronize()
t = time.perf_counter(); n = 0
for a in acts:
    obs, r, d, info = env.step(a); n += 1
    if d: obs = env.reset()
dt = time.perf_counter() - t
print(f'config 1 (README loop, 1 env through the gym-style class): {n / dt:.0f} steps/s, {dt / n * 1e6:.1f} us per step; obs {obs.shape} {obs.dtype}')
