"""GatherScene -- what hrl_pybullet_envs/envs/gather/gather_scene.py:14-36 leaves on `env.stadium_scene` for a user to read: the item
dictionaries and the scene's constructor arguments.  Spawning, respawning and the collision world live in the kernel (csrc/step_core.h:
respawn_item, phase_items); this class is a read-only view over the env's `items` tensor."""


from ..sizeable_enclosed_scene import SizeableEnclosedScene


class GatherScene(SizeableEnclosedScene):
    def __init__(self, env):
        super().__init__(env.world_size)      # the arena's bounding lines and a host-side sense_walls (sizeable_enclosed_scene.py)
        self._e = env
        self.n_food, self.n_poison = env.n_food, env.n_poison
        self.spacing, self.respawn = env.spacing, env.respawn

    def _xy(self):
        b = self._e._backend()
        n = self.n_food + self.n_poison
        return b.items[:, :2 * n].reshape(self._e.num_envs, n, 2)

    def _dict(self, lo, hi):
        xy = self._xy()
        if self._e.num_envs > 1:     # a batch: [N, count, 2] tensor on the GPU
            return xy[:, lo:hi]
        p = xy[0, lo:hi].double().cpu().numpy()
        return {lo + i: [float(p[i, 0]), float(p[i, 1]), 0.1] for i in range(hi - lo)}   # id -> [x, y, 0.1] (gather_scene.py:62); the id is the item's index

    food = property(lambda self: self._dict(0, self.n_food))
    poison = property(lambda self: self._dict(self.n_food, self.n_food + self.n_poison))

    @property
    def all_items(self):
        if self._e.num_envs > 1:
            return self._xy()
        return {**self.food, **self.poison}

