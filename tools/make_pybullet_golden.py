#!/usr/bin/env python3
"""Generate rigid-body golden vectors from the REAL reference: pybullet + pybullet_envs + gym + hrl_pybullet_envs.

The rigid-body half of the oracle (oracle/orc_impl.h PART 2/3) is "parity unpinned": its arithmetic lives in the pybullet wheel
(/root/reference/requirements.txt:1 `pybullet>=3.0.0`), which neither the build image nor the GPU image has, and the reference holds no fixture for
it.  This script is the road from there to a pin.  It CANNOT run in the build container (ModuleNotFoundError: pybullet, gym) and nothing in the
test suite needs it; run it on any machine where

    pip install pybullet "gym<0.22" hrl_pybullet_envs      # + pybullet-gym from GitHub for the Mj variants (reference README.md:7)

works, from a checkout of this repository:

    python tools/make_pybullet_golden.py [--steps 200] [--seeds 8] [--out tests/golden]

and commit the files it writes, tests/golden/pybullet_<env>.json.  They are DATA ONLY -- inputs and outputs of the reference's own step() -- :
per env id, per seed, per step
    qpos[15] (x y z, quaternion x y z w, 8 joint angles in hip_1, ankle_1, ... hip_4, ankle_4 order), qvel[14] (v world, omega world, 8 joint rates),
    items[n][2] (food then poison, gather kinds), target (maze), `task` (what the step reads besides those: potential, initial_z, the feet-contact
    flags of the step before, the flagrun goal bookkeeping -- task_state()), the action, and after `env.step(action)`: qpos', qvel', items', obs,
    rew, done, info, the contact points pybullet reports (link pair, position, normal, distance, normal force);
plus once per env: getDynamicsInfo of every link (mass, friction, local inertia diagonal, restitution, damping), the joint table (name, type,
limits, axis, parent frame), getPhysicsEngineParameters(), the collision shapes, and the versions of the packages.
tests/test_pybullet_golden.py picks the files up when they exist: it replays every recorded step on the CPU oracle from the identical
(qpos, qvel, items, action), reports the deviation per quantity, and settles SURVEY Appendix A.4 (density 1000 vs 5) from the recorded masses.
Nothing here is ever fabricated: without pybullet the script exits with the list of missing modules and writes nothing.
The whole road -- this script, the replay, the fit -- is rehearsed end to end by tests/test_pin_road_dry_run.py against stand-in packages backed by the
CPU oracle under a perturbed model (tests/pybullet_standin.py; scratch directories only, its records carry `versions.standin` and are refused as fixtures)."""
import argparse
import json
import os
import sys

ENV_IDS = ['AntGatherBulletEnv-v0', 'AntMazeBulletEnv-v0', 'PointGatherBulletEnv-v0', 'AntFlagrunBulletEnv-v0']
# the pybulletgym flavour (reference README.md:7: installed from GitHub, not in requirements.txt): recorded where it imports, skipped with the reason
# where it does not.  AntMjEnv is never registered (hrl_pybullet_envs/__init__.py:9): it is constructed from its module (envs/MjAnt.py:31-34).
OPTIONAL_ENV_IDS = ['AntMazeMjEnv-v0', 'hrl_pybullet_envs.envs.MjAnt:AntMjEnv']
JOINT_ORDER = ['hip_1', 'ankle_1', 'hip_2', 'ankle_2', 'hip_3', 'ankle_3', 'hip_4', 'ankle_4']   # URDF link order (assets/ant.xml:18-54)


def need(mods):
    missing = []
    for m in mods:
        try:
            __import__(m)
        except Exception as e:  # noqa: BLE001
            missing.append(f'{m} ({type(e).__name__}: {e})')
    return missing


def tolist(x):
    import numpy as np
    if isinstance(x, dict):
        return {str(k): tolist(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.astype(float).tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, bytes):
        return x.decode(errors='replace')
    return x


class Probe:
    """Reads the packed state of include/hrl_envs.h straight from the physics client (no reference code involved)."""

    def __init__(self, env):
        u = env.unwrapped
        self.u, self.p = u, u._p
        robot = u.robot
        self.body = robot.robot_body.bodies[robot.robot_body.bodyIndex]   # pybullet_envs BodyPart: unique id of the multibody
        self.joints = {}
        for j in range(self.p.getNumJoints(self.body)):
            info = self.p.getJointInfo(self.body, j)
            self.joints[info[1].decode()] = j
        self.ant = all(n in self.joints for n in JOINT_ORDER)

    def qpos_qvel(self):
        pos, orn = self.p.getBasePositionAndOrientation(self.body)
        lin, ang = self.p.getBaseVelocity(self.body)
        q, qd = [], []
        if self.ant:
            for n in JOINT_ORDER:
                s = self.p.getJointState(self.body, self.joints[n])
                q.append(s[0]); qd.append(s[1])
        return list(pos) + list(orn) + q, list(lin) + list(ang) + qd

    def items(self):
        sc = getattr(self.u, 'stadium_scene', None)
        if sc is None or not hasattr(sc, 'food'):
            return None
        return [list(v)[:2] for v in sc.food.values()] + [list(v)[:2] for v in sc.poison.values()]

    def contacts(self):
        out = []
        for c in self.p.getContactPoints(bodyA=self.body):
            out.append({'bodyB': c[2], 'linkA': c[3], 'linkB': c[4], 'posA': list(c[5]), 'posB': list(c[6]), 'normalB': list(c[7]),
                        'distance': c[8], 'normal_force': c[9]})
        return out

    def model(self):
        p, b = self.p, self.body
        links = []
        for l in range(-1, p.getNumJoints(b)):
            d = p.getDynamicsInfo(b, l)
            name = p.getBodyInfo(b)[0].decode() if l < 0 else p.getJointInfo(b, l)[12].decode()
            shapes = [{'type': s[2], 'dims': list(s[3]), 'pos': list(s[5]), 'orn': list(s[6])} for s in p.getCollisionShapeData(b, l)]
            links.append({'link': l, 'name': name, 'mass': d[0], 'lateral_friction': d[1], 'local_inertia_diag': list(d[2]),
                          'inertial_pos': list(d[3]), 'inertial_orn': list(d[4]), 'restitution': d[5], 'rolling_friction': d[6],
                          'spinning_friction': d[7], 'contact_damping': d[8], 'contact_stiffness': d[9], 'collision_shapes': shapes})
        joints = []
        for j in range(p.getNumJoints(b)):
            i = p.getJointInfo(b, j)
            joints.append({'index': j, 'name': i[1].decode(), 'type': i[2], 'damping': i[6], 'friction': i[7], 'lower': i[8], 'upper': i[9],
                           'max_force': i[10], 'max_velocity': i[11], 'link_name': i[12].decode(), 'axis': list(i[13]),
                           'parent_frame_pos': list(i[14]), 'parent_frame_orn': list(i[15]), 'parent_index': i[16]})
        statics = []
        for other in range(p.getNumBodies()):
            if other == b:
                continue
            d = p.getDynamicsInfo(other, -1)
            pos, orn = p.getBasePositionAndOrientation(other)
            statics.append({'body': other, 'name': p.getBodyInfo(other)[0].decode(), 'mass': d[0], 'lateral_friction': d[1], 'restitution': d[5],
                            'pos': list(pos), 'orn': list(orn),
                            'collision_shapes': [{'type': s[2], 'dims': list(s[3])} for s in p.getCollisionShapeData(other, -1)]})
        return {'links': links, 'joints': joints, 'total_mass': sum(x['mass'] for x in links), 'static_bodies': statics,
                'engine': tolist(p.getPhysicsEngineParameters())}


def task_state(u):
    """What a step() of the reference reads BESIDES (qpos, qvel, items, action) -- the env's and the robot's own bookkeeping --, taken before the
    step so that a replay can start from the identical state: upstream WalkerBaseBulletEnv's `potential` (the progress term is measured from it),
    `robot.initial_z` (obs[0] = z - initial_z) and `robot.feet_contact` (calc_state() shows the flags the step BEFORE left); AntMazeBulletEnv.t
    (ant_maze_bullet_env.py:78); AntFlagrunBulletEnv's goal bookkeeping (ant_flagrun_env.py:41-52,98-120: steps_since_goal_change, _rewarded,
    _sq_dist_goal, _goal_start_pos, how many goals are pending and the one next_target() would pop)."""
    out = {}
    for k in ('potential', 't', 'steps_since_goal_change', '_rewarded', '_sq_dist_goal', '_goal_start_pos'):
        if hasattr(u, k):
            out[k] = tolist(getattr(u, k))
    if hasattr(u, 'goals'):
        goals = list(u.goals)
        out['n_goals_pending'] = len(goals)
        out['next_goal'] = tolist(list(goals[-1])) if goals else None   # next_target(): self.goals.pop() takes the LAST one (:116)
    for k in ('initial_z', 'feet_contact'):
        if hasattr(u.robot, k):
            out[k] = tolist(getattr(u.robot, k))
    return out


def make_env(env_id):
    """a registered id through gym.make (with its TimeLimit, hrl_pybullet_envs/__init__.py:15), or `module:Class` constructed directly"""
    import gym
    if ':' not in env_id:
        return gym.make(env_id)
    import importlib
    mod, cls = env_id.split(':')
    return getattr(importlib.import_module(mod), cls)()


def record(env_id, seeds, steps):
    import numpy as np
    out = {'env_id': env_id.split(':')[-1], 'episodes': []}
    for seed in range(seeds):
        env = make_env(env_id)
        env.seed(seed)
        obs0 = env.reset()
        pr = Probe(env)
        if seed == 0:
            out['model'] = pr.model()
            out['obs_dim'], out['act_dim'] = int(np.asarray(obs0).shape[0]), int(env.action_space.shape[0])
        rng = np.random.RandomState(1000 + seed)
        ep = {'seed': seed, 'reset_obs': tolist(np.asarray(obs0)), 'steps': []}
        for t in range(steps):
            qpos, qvel = pr.qpos_qvel()
            rec = {'qpos': qpos, 'qvel': qvel, 'items': pr.items(), 'action': None}
            u = env.unwrapped
            rec['task'] = task_state(u)
            if hasattr(u, 'target'):
                rec['target'] = tolist(np.asarray(u.target))
            if hasattr(u, 'walk_target_x'):
                rec['walk_target'] = [float(u.walk_target_x), float(u.walk_target_y)]
            a = rng.uniform(-1, 1, env.action_space.shape)
            rec['action'] = a.tolist()
            obs, rew, done, info = env.step(a)
            q1, v1 = pr.qpos_qvel()
            rec.update({'qpos_after': q1, 'qvel_after': v1, 'items_after': pr.items(), 'obs': tolist(np.asarray(obs)), 'rew': float(rew),
                        'done': bool(done), 'info': tolist({k: v for k, v in info.items() if isinstance(v, (int, float, bool, np.floating, np.integer))}),
                        'contacts_after': pr.contacts(),
                        'robot': {k: tolist(getattr(u.robot, k)) for k in ('body_xyz', 'body_rpy', 'initial_z', 'walk_target_dist', 'joints_at_limit', 'feet_contact')
                                  if hasattr(u.robot, k)}})
            ep['steps'].append(rec)
            if done:
                break
        out['episodes'].append(ep)
        env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
    ap.add_argument('--envs', nargs='*', default=ENV_IDS)
    ap.add_argument('--no-optional', action='store_true', help='do not try the pybulletgym flavour (AntMazeMjEnv-v0, AntMjEnv)')
    a = ap.parse_args()
    missing = need(['numpy', 'pybullet', 'pybullet_envs', 'gym', 'hrl_pybullet_envs'])
    if missing:
        sys.exit('make_pybullet_golden: cannot run here, nothing written.  Missing: ' + '; '.join(missing))
    import gym
    import pybullet
    import hrl_pybullet_envs  # noqa: F401  (registers the ids, hrl_pybullet_envs/__init__.py:11-16)
    versions = {'pybullet_api': pybullet.getAPIVersion(), 'gym': gym.__version__, 'python': sys.version.split()[0]}
    if getattr(hrl_pybullet_envs, 'standin', False):
        # the dry run of this road (tests/test_pin_road_dry_run.py): the packages are tests/pybullet_standin.py's, backed by the CPU oracle.  Its
        # records say so and are refused as fixtures; they never go under tests/golden
        versions['standin'] = True
        golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
        if os.path.realpath(a.out) == os.path.realpath(golden):
            sys.exit('make_pybullet_golden: stand-in packages are installed (a dry run): give --out a scratch directory, tests/golden is for records of the real pybullet')
    os.makedirs(a.out, exist_ok=True)
    for env_id in list(a.envs) + ([] if a.no_optional else OPTIONAL_ENV_IDS):
        try:
            data = record(env_id, a.seeds, a.steps)
        except Exception as e:  # noqa: BLE001
            if env_id not in OPTIONAL_ENV_IDS:
                raise
            print(f'{env_id}: skipped ({type(e).__name__}: {e}) -- the pybulletgym flavour is optional')
            continue
        data['versions'] = versions
        name = 'pybullet_' + data['env_id'].split('-')[0] + '.json'
        with open(os.path.join(a.out, name), 'w') as f:
            json.dump(data, f, allow_nan=True)
        print(name, sum(len(e['steps']) for e in data['episodes']), 'steps, total mass', data['model']['total_mass'])


if __name__ == '__main__':
    main()
