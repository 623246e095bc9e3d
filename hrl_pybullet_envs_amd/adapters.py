"""API adapters around the reference-shaped envs.

The reference speaks old-gym (<= 0.21): reset() -> obs, step() -> (obs, rew, done, info) (README.md:24-34).
`GymnasiumAdapter` exposes the Gymnasium 5-tuple on top of it, for one env (numpy) or a batch (torch tensors)."""


class GymnasiumAdapter:
    def __init__(self, env):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.num_envs = getattr(env, 'num_envs', 1)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.env.seed(seed)
        return self.env.reset(), {}

    def step(self, action):
        obs, rew, done, info = self.env.step(action)
        if self.num_envs == 1:
            truncated = bool(info.get('TimeLimit.truncated', False))
            return obs, rew, bool(done) and not truncated, truncated, info
        limit = getattr(self.env, 'max_episode_steps', 0)
        d = done.bool()
        truncated = d & (info['episode_length'] >= limit) if limit > 0 else d & False
        return obs, rew, d & ~truncated, truncated, info

    def close(self):
        self.env.close()
