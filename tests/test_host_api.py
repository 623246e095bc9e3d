"""Host logic that needs no GPU: the C-ABI library loads and exports every symbol of include/hrl_envs.h, the ctypes
mirror matches the C structs, the env classes mirror the reference's constructor API, sharding helpers."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import orc
from hrl_pybullet_envs_amd import _capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_exports_every_declared_symbol():
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.build import build
    build()  # hipcc cross-compiles for gfx950 without a GPU
    hdr = open(os.path.join(ROOT, 'include', 'hrl_envs.h')).read()
    declared = set(re.findall(r'\b(hrl_[a-z_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s)
    assert L.hrl_backend() == b'hip-gfx950'


def test_no_cpu_fallback_without_gpu():
    import torch
    from hrl_pybullet_envs_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    cfg = _lib.default_config(K.HRL_ANT_GATHER, num_envs=4)
    h = C.c_void_p()
    assert _lib.lib().hrl_create(C.byref(cfg), C.byref(h)) == K.HRL_ERR_NO_DEVICE
    assert b'no HIP device' in _lib.lib().hrl_last_error()
    import hrl_pybullet_envs_amd as H
    with pytest.raises(_lib.HrlError):
        H.make('AntGatherBulletEnv-v0').reset()


def test_struct_layout_matches_header():
    from hrl_pybullet_envs_amd import _lib
    # sizes as the C compiler sees them (the oracle is compiled from the same header)
    src = '#include "include/hrl_envs.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu", sizeof(hrl_config), sizeof(hrl_model), sizeof(hrl_buffers));return 0;}'
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', ROOT, '-o', os.path.join(d, 't'), os.path.join(d, 't.c')], cwd=ROOT)
        out = subprocess.check_output([os.path.join(d, 't')]).decode().split()
    assert [int(x) for x in out] == [C.sizeof(K.hrl_config), C.sizeof(K.hrl_model), C.sizeof(K.hrl_buffers)]
    # product defaults == oracle defaults, byte for byte, for every kind
    for kind in range(6):
        assert bytes(_lib.default_config(kind)) == bytes(orc.default_config(kind))


def test_bad_config_is_rejected_with_a_reason():
    from hrl_pybullet_envs_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    for kw, frag in ((dict(n_food=40, n_poison=25), b'n_food'), (dict(num_envs=0), b'num_envs'),
                     (dict(abi_version=99), b'abi_version'), (dict(robot_coll_dist=0, model_item_collision=0), b'robot_coll_dist')):
        cfg = _lib.default_config(K.HRL_ANT_GATHER, **kw)
        assert L.hrl_create(C.byref(cfg), C.byref(h)) == K.HRL_ERR_BAD_ARG
        assert frag in L.hrl_last_error()
    assert L.hrl_step(None, None, None) == K.HRL_ERR_BAD_ARG


def test_env_classes_mirror_reference_constructor_api():
    import hrl_pybullet_envs_amd as H
    e = H.AntGatherBulletEnv()
    assert e.observation_space.shape == (46,) and e.action_space.shape == (8,)   # ant_gather_env.py:53-55
    assert H.AntGatherBulletEnv(n_bins=6).observation_space.shape == (38,)
    assert H.PointGatherBulletEnv().observation_space.shape == (18,)             # gather_base.py:54-55
    assert H.PointGatherBulletEnv().action_space.shape == (2,)                   # point_bot.py:15
    assert H.AntMazeBulletEnv().observation_space.shape == (38,)                 # ant_maze_bullet_env.py:54-57
    assert H.AntMazeBulletEnv(sense_target=True, n_bins=8).observation_space.shape == (26 + 8 + 8,)
    assert H.AntMazeBulletEnv(sense_walls=False).observation_space.shape == (28,)
    assert H.AntMjEnv().observation_space.shape == (29,)                         # MjAnt.py:15
    assert H.AntMazeMjEnv().observation_space.shape == (60,)                     # ant_maze_mj_env.py:50
    assert H.AntMazeMjEnv()._cfg.n_targets == 5                                  # ant_maze_mj_env.py:13-14
    assert H.AntFlagrunBulletEnv().observation_space.shape == (28,)              # ant_flagrun_env.py:53-55
    assert H.AntFlagrunBulletEnv(use_sensor=True).observation_space.shape == (36,)
    with pytest.raises(AssertionError):
        H.AntFlagrunBulletEnv(max_targets=5, max_target_dist=3)                  # ant_flagrun_env.py:17-18
    assert H.AntFlagrunBulletEnv(max_targets=0, max_target_dist=3)._cfg.flag_max_target_dist == 3.0   # :80-89 goals near the robot
    with pytest.raises(ValueError):
        H.AntMazeBulletEnv(target_encoding=5)                                    # PositionEncoding(5), utils.py:66-68
    assert isinstance(H.make('AntMazeBulletEnv-v0', tol=2.0), H.AntMazeBulletEnv)
    with pytest.raises(KeyError):
        H.make('AntMjBulletEnv-0')  # README.md:13 names an id that is never registered (SURVEY C-12)
    c = H.AntGatherBulletEnv(n_food=4, n_poison=3, world_size=(9, 11), dying_cost=-3, seed=5)._cfg
    assert (c.n_food, c.n_poison, c.world_size[0], c.world_size[1], c.dying_cost, c.seed) == (4, 3, 9.0, 11.0, -3.0, 5)
    assert c.max_episode_steps == 0 and c.auto_reset == 0                        # the class has no step limit (gym.make adds it, __init__.py:15)
    assert H.make('AntGatherBulletEnv-v0')._cfg.max_episode_steps == 2000


def test_shard_range_partitions_all_envs():
    from hrl_pybullet_envs_amd.dist import shard_range
    for total, world in ((32768, 8), (4096, 3), (10, 4), (7, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1


def test_gymnasium_adapter_five_tuple():
    import torch
    from hrl_pybullet_envs_amd.adapters import GymnasiumAdapter

    class Fake:  # old-gym shaped, like the env classes
        observation_space = action_space = None
        max_episode_steps = 5

        def __init__(self, n):
            self.num_envs, self.t = n, 0

        def seed(self, s):
            self.seeded = s

        def reset(self):
            self.t = 0
            return 'obs0'

        def step(self, a):
            self.t += 1
            if self.num_envs == 1:
                return 'o', 1.0, self.t >= 5, ({'TimeLimit.truncated': True} if self.t >= 5 else {})
            done = torch.tensor([self.t >= 5, self.t == 2], dtype=torch.uint8)
            return 'o', torch.ones(2), done, {'episode_length': torch.tensor([float(self.t), 2.0])}

    g = GymnasiumAdapter(Fake(1))
    assert g.reset(seed=3) == ('obs0', {}) and g.env.seeded == 3
    outs = [g.step(None) for _ in range(5)]
    assert outs[0][2:4] == (False, False) and outs[-1][2:4] == (False, True)
    gb = GymnasiumAdapter(Fake(2))
    gb.reset()
    o = [gb.step(None) for _ in range(5)]
    assert o[1][2].tolist() == [False, True] and o[1][3].tolist() == [False, False]   # env 1 terminated at t=2
    assert o[4][2].tolist() == [False, False] and o[4][3].tolist() == [True, False]   # env 0 truncated at the limit

    class WithFlag(Fake):  # the batched envs of this package hand the kernel's flag over: it wins over the length heuristic
        def step(self, a):
            ob, r, done, info = super().step(a)
            info['TimeLimit.truncated'] = torch.tensor([0, 0], dtype=torch.uint8)   # e.g. died exactly at the limit: not a truncation
            return ob, r, done, info
    gf = GymnasiumAdapter(WithFlag(2))
    gf.reset()
    o = [gf.step(None) for _ in range(5)]
    assert o[4][2].tolist() == [True, False] and o[4][3].tolist() == [False, False]


def test_bench_spawns_its_own_ranks(monkeypatch):
    """bench.py --gpus N without WORLD_SIZE starts the N ranks itself (torch.distributed.run on 127.0.0.1) before it
    imports anything that could touch the GPU, and hands back their exit status."""
    import importlib
    import bench
    importlib.reload(bench)
    calls = {}

    def fake_call(cmd, env=None):
        calls['cmd'], calls['env'] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--kind', 'mixed'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = calls['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '3', '--kind', 'mixed']
    assert calls['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert 'torch' not in bench.__dict__  # the parent's module level never imports torch


def test_bench_roofline_reports_hbm_and_valu(monkeypatch):
    import bench
    from hrl_pybullet_envs_amd.build import kernel_source_hash
    monkeypatch.setattr(bench, 'pmc_summary', lambda kind: {'tag': 't', 'source_sha256': kernel_source_hash(), 'fetch_bytes_per_env': 250.0,
                                                            'write_bytes_per_env': 480.0, 'valu_insts_per_env': 6000.0,
                                                            'wave_cycles_per_env': 100000.0, 'kernel_us_profiled': 50.0})
    r = bench.roofline('gather', 4096, 70e-6)
    assert r['bound'] == 'hbm' and abs(r['achieved'] - 581 * 4096 / 70e-6 / 1e9) < 1e-9 and r['peak'] == 8000.0
    v = r['valu']
    assert v['issue_cycles_per_simd'] == v['insts_per_env'] * 4096 / 1024 * 2
    assert 0 < v['frac'] < v['frac_at_held_clock'] < 1


def test_default_config_rejects_unknown_fields():
    from hrl_pybullet_envs_amd import _lib
    with pytest.raises(TypeError):
        _lib.default_config(K.HRL_ANT_GATHER, n_foods=3)
    with pytest.raises(TypeError):
        _lib.default_config(K.HRL_ANT_GATHER, model_gravty=3.0)


def test_mirrored_module_paths_and_reference_ids():
    """same names and module paths under ...envs as the reference, same registered ids (__init__.py:11-16)."""
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd.envs.ant_maze.ant_maze_bullet_env import PositionEncoding as PE2
    from hrl_pybullet_envs_amd.envs.gather.gather_base import GatherBulletEnv
    from hrl_pybullet_envs_amd.envs.gather.point_bot import PointBot
    from hrl_pybullet_envs_amd.utils import PositionEncoding
    assert PositionEncoding is PE2 and PositionEncoding.angle.value == 1 and PositionEncoding(0) is PositionEncoding.normed_vec  # utils.py:66-68
    bot = PointBot()
    assert bot.start_pos == [0, 0, 0.5] and bot.initial_z == 1 and bot.action_space.shape == (2,) and bot.observation_space.shape == (8,)
    g = GatherBulletEnv(bot, n_bins=7)                                # gather_base.py:14-28: the robot comes first, n_bins defaults to 5
    assert g.observation_space.shape == (8 + 14,) and GatherBulletEnv(bot).observation_space.shape == (18,)
    assert issubclass(H.PointGatherBulletEnv, GatherBulletEnv) and isinstance(H.PointGatherBulletEnv().robot, PointBot)  # point_gather_env.py:7-24
    with pytest.raises(TypeError):
        GatherBulletEnv(object())

    class FakeGym:
        class envs:
            calls = []

            @staticmethod
            def register(**kw):
                FakeGym.envs.calls.append(kw)
    ids = H.register_with(FakeGym)
    assert ids == ['AntGatherBulletEnv-v0', 'AntMazeMjEnv-v0', 'AntMazeBulletEnv-v0', 'AntFlagrunBulletEnv-v0', 'PointGatherBulletEnv-v0']
    assert all(c['max_episode_steps'] == 2000 for c in FakeGym.envs.calls)
    mod, cls = FakeGym.envs.calls[0]['entry_point'].split(':')
    import importlib
    assert getattr(importlib.import_module(mod), cls) is H.AntGatherBulletEnv
    # the reference imported in the same process has registered its ids already (an A/B script): those are skipped, nothing raises
    class TakenGym:
        class envs:
            registry = {'AntGatherBulletEnv-v0': object(), 'PointGatherBulletEnv-v0': object()}
            calls = []

            @staticmethod
            def register(**kw):
                if kw['id'] in TakenGym.envs.registry:
                    raise RuntimeError('Cannot re-register id: ' + kw['id'])
                TakenGym.envs.calls.append(kw)
    assert H.register_with(TakenGym) == ['AntMazeMjEnv-v0', 'AntMazeBulletEnv-v0', 'AntFlagrunBulletEnv-v0']
    # contact-based pickup and manual goals are constructor options now, as in the reference
    assert H.AntGatherBulletEnv(robot_coll_dist=0)._cfg.robot_coll_dist == 0.0
    f = H.AntFlagrunBulletEnv(manual_goal_creation=True)
    assert f._cfg.flag_manual_goals == 1
    with pytest.raises(RuntimeError):
        H.AntFlagrunBulletEnv().set_goals([[1, 1]])


def test_rgb_array_render_draws_the_scene():
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd.envs.render import ant_points, draw_env
    st = np.zeros(32, np.float32); st[2] = 0.75; st[6] = 1; st[7:15] = [0, 1, 0, -1, 0, -1, 0, 1]
    items = np.random.RandomState(0).uniform(-7, 7, 32).astype(np.float32)
    img = draw_env(H.AntGatherBulletEnv()._cfg, st, items, 200)
    assert img.shape == (200, 200, 3) and img.dtype == np.uint8
    assert (img == (0, 170, 0)).all(axis=2).any() and (img == (210, 0, 0)).all(axis=2).any() and (img == (0, 50, 200)).all(axis=2).any()
    p0, legs = ant_points(st[:15])
    assert np.allclose(legs[0][0], [0.2, 0.2, 0.75]) and np.allclose(legs[0][1], [0.4, 0.4, 0.75])  # ant.xml:17-20
    assert abs(np.linalg.norm(legs[0][2] - legs[0][1]) - 0.4 * np.sqrt(2)) < 1e-6  # the float32 quaternion is not exactly unit
    assert draw_env(H.AntMazeBulletEnv()._cfg, st, None, 64).shape == (64, 64, 3)
    assert draw_env(H.PointGatherBulletEnv()._cfg, st, items, 64).shape == (64, 64, 3)


def test_host_philox_restates_the_shared_goal_list():
    """`env.goals` / `env.goal` of a non-manual flagrun env evaluate the kernel's goal function on the host
    (hrl_pybullet_envs_amd/_philox.py): same bits as the oracle's flag_goal for every (seed, episode, k) tried."""
    import ctypes as C
    import orc
    from hrl_pybullet_envs_amd._philox import flag_goal
    for seed, size in ((123, 10.0), (0, 1.2), (2 ** 40 + 17, 6.0)):
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, seed=seed, flag_size=size)
        ep, k = np.meshgrid(np.arange(1, 6), np.arange(1, 41), indexing='ij')
        got = flag_goal(seed, size, ep, k)
        want = np.zeros(got.shape, np.float32)
        for i in range(ep.shape[0]):
            for j in range(ep.shape[1]):
                g = np.zeros(2, np.float32)
                orc.lib().orc_flag_goal_f32(C.byref(cfg), int(ep[i, j]), int(k[i, j]), orc.ptr(g))
                want[i, j] = g
        assert np.array_equal(got, want)
        assert np.all(np.linalg.norm(got, axis=-1) >= 0.5 - 1e-6) and np.all(np.abs(got) <= size / 2)


def test_bench_withholds_counters_of_other_code(monkeypatch):
    """bench.py reports `traffic` / `valu` from the committed PMC summary only while the summary's source hash equals the kernel
    sources being benched (roofline.pmc_stale otherwise, counters withheld)."""
    import bench
    from hrl_pybullet_envs_amd.build import kernel_source_hash
    fresh = {'tag': 't', 'source_sha256': kernel_source_hash(), 'envs_per_launch': 4096, 'fetch_bytes_per_env': 250.0, 'write_bytes_per_env': 480.0,
             'valu_insts_per_env': 6000.0, 'wave_cycles_per_env': 100000.0, 'kernel_us_profiled': 50.0}
    monkeypatch.setattr(bench, 'pmc_summary', lambda kind: dict(fresh))
    r = bench.roofline('gather', 4096, 50e-6)
    assert r['pmc_stale'] is False and r['traffic'] == (250.0 + 480.0) * 4096 and r['valu']['insts_per_env'] == 6000.0
    assert r['achieved'] == pytest.approx(581 * 4096 / 50e-6 / 1e9) and r['frac'] == pytest.approx(r['achieved'] / 8000.0)
    monkeypatch.setattr(bench, 'pmc_summary', lambda kind: dict(fresh, source_sha256='0' * 64))
    r = bench.roofline('gather', 4096, 50e-6)
    assert r['pmc_stale'] is True and r['traffic'] is None and 'valu' not in r
    monkeypatch.setattr(bench, 'pmc_summary', lambda kind: dict(fresh))
    r = bench.roofline('gather', 32768, 400e-6)   # another batch size than the counters were taken at: the fetch side has a fixed part per launch, nothing is scaled
    assert r['traffic'] is None and '4096' in r['traffic_note'] and r['valu']['insts_per_env'] == 6000.0
    monkeypatch.setattr(bench, 'pmc_summary', lambda kind: {})
    assert bench.roofline('gather', 4096, 50e-6)['pmc_stale'] is False   # no summary committed for this kernel: nothing to be stale


def test_committed_counter_summary_describes_the_committed_kernels():
    """profiles/pmc_summary.json (what bench.py reports as roofline.traffic / roofline.valu) was collected from exactly the kernel sources in the
    tree: every kind's entry carries the sha256 of csrc/step_core.h + csrc/hrl_hip.hip (tools/summarize_profile.py), equal to the tree's."""
    import json
    import os
    from hrl_pybullet_envs_amd.build import kernel_source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    summ = json.load(open(os.path.join(root, 'profiles', 'pmc_summary.json')))
    assert {'gather', 'flat', 'maze', 'point', 'maze_mj', 'flagrun'} <= set(summ)
    stale = [k for k, v in summ.items() if v.get('source_sha256') != kernel_source_hash()]
    assert not stale, f're-profile (tools/profile_round.sh + tools/summarize_profile.py): stale counter summaries for {stale}'


def test_constructor_surfaces_equal_the_references():
    """SURVEY 8b, pinned by data taken from the reference itself (`tests/golden/constructor_signatures.json`: inspect.signature of its classes and the
    keyword arguments its __init__.py:11-16 registered, recorded by make_golden.py): every class sits at the mirrored module path, takes the
    reference's parameters under the same names, in the same order, with the same defaults (this package's own -- num_envs, device, seed where the
    reference has none -- come after them), and registers the same ids in the same order with the same step limit."""
    import enum
    import importlib
    import inspect
    import json
    import os
    import hrl_pybullet_envs_amd as H
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'constructor_signatures.json')))

    def plain(v):
        if isinstance(v, enum.Enum): return {'enum': type(v).__name__, 'name': v.name}
        if isinstance(v, (tuple, list)): return [plain(x) for x in v]
        if isinstance(v, (bool, int, str)) or v is None: return v
        if isinstance(v, (float, np.floating, np.integer)): return float(v)
        return {'repr': repr(v)}

    assert len(g['classes']) == 9
    for path, params in g['classes'].items():
        mod, name = path.split(':')
        cls = getattr(importlib.import_module(mod.replace('hrl_pybullet_envs', 'hrl_pybullet_envs_amd', 1)), name)
        ours = [(n, q) for n, q in inspect.signature(cls.__init__).parameters.items() if n != 'self']
        assert len(ours) >= len(params), path
        for (n, q), ref in zip(ours, params):
            assert n == ref['name'] and q.kind.name == ref['kind'], (path, n, ref)
            d = '<required>' if q.default is inspect.Parameter.empty else plain(q.default)
            assert d == ref['default'] and type(d) is type(ref['default']) or (isinstance(d, (int, float)) and not isinstance(d, bool) and d == ref['default']), (path, n, d, ref['default'])
        extra = [n for n, _ in ours[len(params):]]
        assert set(extra) <= {'num_envs', 'device', 'seed', 'goal_capacity'}, (path, extra)   # goal_capacity: the length bound of `env.goals` (ABI v6)

    class FakeGym:
        class envs:
            calls = []

            @staticmethod
            def register(**kw):
                FakeGym.envs.calls.append(kw)
    H.register_with(FakeGym)
    assert [(c['id'], c['max_episode_steps']) for c in FakeGym.envs.calls] == [(r['id'], r['max_episode_steps']) for r in g['registered']]
    assert [c['entry_point'] for c in FakeGym.envs.calls] == [r['entry_point'].replace('hrl_pybullet_envs', 'hrl_pybullet_envs_amd', 1) for r in g['registered']]


def test_who_sets_the_step_limit():
    """hrl_pybullet_envs/__init__.py:15: the limit belongs to the registration, not to the classes.  make() = gym.make: 2000; a single env
    constructed directly: none (the reference's object; gym's own TimeLimit may wrap it); a batch constructed directly: 2000 (its own vector env)."""
    import hrl_pybullet_envs_amd as H
    for cls in (H.AntGatherBulletEnv, H.PointGatherBulletEnv, H.AntMazeBulletEnv, H.AntMazeMjEnv, H.AntFlagrunBulletEnv, H.AntMjEnv):
        assert cls().max_episode_steps == 0 and cls(num_envs=8).max_episode_steps == 2000
        assert H.make(f'{cls.__name__}-v0').max_episode_steps == 2000 and H.make(f'{cls.__name__}-v0', num_envs=8).max_episode_steps == 2000
    e = H.AntGatherBulletEnv()
    e.max_episode_steps = 7
    assert e._cfg.max_episode_steps == 7 and e.max_episode_steps == 7
    with pytest.raises(ValueError):
        e.max_episode_steps = -1


def test_constructor_arguments_reach_the_config():
    """Every constructor argument of every env class, drawn at random, ends up in the hrl_config field the kernel reads (same
    kwarg names; the mapping to the C-ABI is this package's).  No GPU: the backend is created on first use."""
    import hrl_pybullet_envs_amd as H
    rng = np.random.RandomState(12)
    f32 = lambda x: float(np.float32(x))
    for _ in range(40):
        kw = dict(n_food=int(rng.randint(0, 30)), n_poison=int(rng.randint(0, 30)), world_size=(f32(rng.uniform(4, 30)), f32(rng.uniform(4, 30))),
                  n_bins=int(rng.randint(1, 60)), sensor_range=f32(rng.uniform(1, 40)), sensor_span=f32(rng.uniform(0.5, 6)),
                  robot_coll_dist=f32(rng.choice([1.0, 0.0, -1.0, 3.5])), robot_object_spacing=f32(rng.uniform(0.5, 3)), dying_cost=f32(rng.uniform(-20, 5)),
                  use_sensor=bool(rng.randint(2)), respawn=bool(rng.randint(2)))
        for cls, kind in ((H.AntGatherBulletEnv, K.HRL_ANT_GATHER), (H.PointGatherBulletEnv, K.HRL_POINT_GATHER)):
            c = cls(num_envs=3, seed=9, **kw)._cfg
            assert c.env_kind == kind and (c.n_food, c.n_poison, c.n_bins, c.use_sensor, c.respawn) == (kw['n_food'], kw['n_poison'], kw['n_bins'], int(kw['use_sensor']), int(kw['respawn']))
            assert (c.world_size[0], c.world_size[1], c.sensor_range, c.sensor_span, c.robot_coll_dist, c.robot_object_spacing, c.dying_cost) == \
                (*kw['world_size'], kw['sensor_range'], kw['sensor_span'], kw['robot_coll_dist'], kw['robot_object_spacing'], kw['dying_cost'])
            assert (c.num_envs, c.seed, c.auto_reset, c.max_episode_steps) == (3, 9, 1, 2000)
        nt = int(rng.randint(1, 20))
        mk = dict(n_bins=int(rng.randint(2, 60)), sensor_range=f32(rng.uniform(1, 40)), sensor_span=f32(rng.uniform(0.5, 6)),
                  targets=[(f32(rng.uniform(-4, 4)), f32(rng.uniform(-8, 8))) for _ in range(nt)], target_encoding=int(rng.randint(2)),
                  tol=f32(rng.uniform(0.2, 4)), inner_rew_weight=f32(rng.uniform(-1, 1)))
        extra = dict(sense_target=bool(rng.randint(2)), sense_walls=bool(rng.randint(2)), done_at_target=bool(rng.randint(2)), max_steps=int(rng.randint(-1, 50)),
                     targ_dist_rew=bool(rng.randint(2)))
        for cls, kind, kws in ((H.AntMazeBulletEnv, K.HRL_ANT_MAZE, dict(mk, **extra)), (H.AntMazeMjEnv, K.HRL_ANT_MAZE_MJ, mk)):
            c = cls(seed=4, **kws)._cfg
            assert c.env_kind == kind and (c.n_bins, c.sensor_range, c.sensor_span, c.target_encoding, c.tol, c.inner_rew_weight) == \
                (mk['n_bins'], mk['sensor_range'], mk['sensor_span'], mk['target_encoding'], mk['tol'], mk['inner_rew_weight'])
            assert c.n_targets == nt and [(c.targets[i][0], c.targets[i][1]) for i in range(nt)] == mk['targets']
            assert (c.start_pos[0], c.start_pos[1], c.start_pos[2]) == (-2.0, -5.0, 0.25) and (c.num_envs, c.seed, c.auto_reset, c.max_episode_steps) == (1, 4, 0, 0)
            if cls is H.AntMazeBulletEnv:
                assert (c.sense_target, c.sense_walls, c.done_at_target, c.max_steps, c.targ_dist_rew) == tuple(int(extra[k]) for k in ('sense_target', 'sense_walls', 'done_at_target', 'max_steps', 'targ_dist_rew'))
        close = bool(rng.randint(2))
        fk = dict(size=f32(rng.uniform(2, 30)), tolerance=f32(rng.uniform(0.2, 1.0)), max_targets=0 if close else int(rng.randint(1, 200)),
                  max_target_dist=f32(rng.uniform(2.5, 8)) if close else 0, timeout=int(rng.randint(0, 500)), enclosed=bool(rng.randint(2)),
                  use_sensor=bool(rng.randint(2)), sensor_bins=int(rng.randint(2, 60)), sensor_span=f32(rng.uniform(0.5, 6)), sensor_range=f32(rng.uniform(1, 10)),
                  switch_flag_on_collision=bool(rng.randint(2)), manual_goal_creation=bool(rng.randint(2)))
        c = H.AntFlagrunBulletEnv(num_envs=2, goal_capacity=33, **fk)._cfg
        assert c.env_kind == K.HRL_ANT_FLAGRUN and (c.flag_size, c.tol, c.flag_max_targets, c.flag_max_target_dist, c.flag_timeout) == \
            (fk['size'], fk['tolerance'], fk['max_targets'], fk['max_target_dist'], fk['timeout'])
        assert (c.flag_enclosed, c.use_sensor, c.n_bins, c.sensor_span, c.sensor_range, c.flag_switch_on_collision, c.flag_manual_goals, c.flag_goal_capacity) == \
            (int(fk['enclosed']), int(fk['use_sensor']), fk['sensor_bins'], fk['sensor_span'], fk['sensor_range'], int(fk['switch_flag_on_collision']), int(fk['manual_goal_creation']), 33)
        assert c.world_size[0] == c.world_size[1] == np.float32(fk['size'] + 2) and c.seed == 123                       # ant_flagrun_env.py:59-61; seed=123 is the reference's default (:16)
        walls = fk['enclosed'] or fk['use_sensor']
        assert (c.centroid_n_static, c.centroid_static_sum[0]) == ((2, np.float32(-(fk['size'] + 2) / 2)) if walls else (1, 0.0))
    with pytest.raises(AssertionError, match='max_targets and max_target_dist'):
        H.AntFlagrunBulletEnv(max_targets=5, max_target_dist=3.0)    # ant_flagrun_env.py:17-18
    with pytest.raises(ValueError):
        H.AntMazeBulletEnv(target_encoding=2)                        # PositionEncoding(2), utils.py:66-68


def test_alias_makes_the_references_import_lines_work():
    """`import hrl_pybullet_envs_amd.alias`: the import lines of a script written against the reference (README.md:19-22, the module paths of
    hrl_pybullet_envs/__init__.py:3-7) resolve to this package.  Run in a child process: the alias is process-wide."""
    import subprocess
    import sys
    code = '''
import hrl_pybullet_envs_amd.alias
import hrl_pybullet_envs
from hrl_pybullet_envs.envs.gather.ant_gather_env import AntGatherBulletEnv
from hrl_pybullet_envs.envs.gather.point_gather_env import PointGatherBulletEnv
from hrl_pybullet_envs.envs.ant_maze.ant_maze_bullet_env import AntMazeBulletEnv
from hrl_pybullet_envs.envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv
from hrl_pybullet_envs.envs.ant_flagrun.ant_flagrun_env import AntFlagrunBulletEnv
from hrl_pybullet_envs.envs.MjAnt import AntMjEnv, MjAnt
from hrl_pybullet_envs.utils import PositionEncoding
import hrl_pybullet_envs_amd as H
assert hrl_pybullet_envs is H and AntGatherBulletEnv is H.AntGatherBulletEnv and AntMazeMjEnv is H.AntMazeMjEnv and AntFlagrunBulletEnv is H.AntFlagrunBulletEnv
assert AntGatherBulletEnv(n_food=3).observation_space.shape == (46,) and hrl_pybullet_envs.make('PointGatherBulletEnv-v0').max_episode_steps == 2000
print('alias-ok')
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and 'alias-ok' in r.stdout, r.stderr[-2000:]
    # with the real reference already imported the alias refuses instead of shadowing it
    code2 = "import sys, types; sys.modules['hrl_pybullet_envs'] = types.ModuleType('hrl_pybullet_envs')\ntry:\n    import hrl_pybullet_envs_amd.alias\nexcept ImportError as e:\n    print('refused', e)"
    r = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert 'refused' in r.stdout, (r.stdout, r.stderr[-1000:])


def test_intersection_utils_mirror_equals_the_references_values():
    """hrl_pybullet_envs_amd/envs/intersection_utils.py (the host-side mirror a user script may import from the reference's path) against
    `tests/golden/intersection.json`, produced by the reference's own functions."""
    import json
    from hrl_pybullet_envs_amd.envs import intersection_utils as iu
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'intersection.json')))
    for c in g['lines']:
        p = [iu.Point(*q) for q in c['p']]
        r = iu.inf_intersection(*p)
        assert (r is None) == (c['inf'] is None) and (r is None or (r.x, r.y) == tuple(c['inf']))
        assert iu.segment_intersection(*p) == c['seg']
    for c in g['quadrant']:
        assert iu.quadrant(iu.Point(*c['p'])) == c['q']
    for c in g['pol2cart']:
        assert np.allclose(iu.pol2cart(c['rho'], c['phi']), c['xy'], rtol=0, atol=1e-15)
    with pytest.raises(Exception, match='never happen'):
        iu.quadrant(iu.Point(float('nan'), 1.0))
    assert len(g['lines']) > 50


def test_scene_mirrors_equal_the_references_bounds_and_wall_sensor():
    """`env.scene` / `env.stadium_scene` of every env: the bounding lines equal the reference scenes' (`sense_walls.json`: maze_bounds,
    arena_bounds; `random_config.json`: flagrun arenas of other sizes) and the host-side `sense_walls` reproduces the 182 readings vectors the
    reference's own method returned, to 1e-12."""
    import json
    import hrl_pybullet_envs_amd as H
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = json.load(open(os.path.join(gd, 'sense_walls.json')))
    as_lists = lambda b: [[[p.x, p.y], [q.x, q.y]] for p, q in b]
    maze, arena = H.AntMazeBulletEnv().scene, H.AntGatherBulletEnv().stadium_scene
    assert as_lists(maze.bounds) == g['maze_bounds'] and as_lists(H.AntMazeMjEnv().stadium_scene.bounds) == g['maze_bounds'] and list(maze.box_pos) == g['maze_box_pos']
    assert as_lists(arena.bounds) == g['arena_bounds'] and as_lists(H.PointGatherBulletEnv().scene.bounds) == g['arena_bounds']
    worst = 0.0
    for c in g['cases']:
        out = (maze if c['scene'] == 'maze' else arena).sense_walls(c['bins'], c['span'], c['range'], np.array(c['pos']), c['yaw'])
        worst = max(worst, float(np.abs(np.array(out, float) - np.array(c['out'])).max()))
    assert worst <= 1e-12 and len(g['cases']) > 150, worst
    for c in json.load(open(os.path.join(gd, 'random_config.json')))['flagrun_step']:
        f = H.AntFlagrunBulletEnv(size=c['size'], use_sensor=True, sensor_bins=c['n_bins'], sensor_span=c['span'], sensor_range=c['range'])
        assert as_lists(f.scene.bounds) == c['arena_bounds']
        if c['use_sensor']:
            assert np.abs(np.array(f.scene.sense_walls(c['n_bins'], c['span'], c['range'], c['pos'], c['yaw']), float) - np.array(c['sensor'])).max() <= 1e-12
    assert H.AntFlagrunBulletEnv(enclosed=False).scene is None


def test_evidence_index_points_at_files_that_exist():
    """profiles/INDEX.md maps claims to raw files: every `r4_*` / `pmc_summary.json` name it cites is committed (globs expand to at least one file),
    and so is every tool it names."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'profiles', 'INDEX.md')).read()
    names = set(re.findall(r'`((?:r4_|pmc_summary)[A-Za-z0-9_{},.*]*\.(?:json|txt|csv|patch))`', text))
    assert len(names) > 20
    for n in names:
        alts = [n]
        m = re.search(r'\{([^}]*)\}', n)
        if m:
            alts = [n[:m.start()] + a + n[m.end():] for a in m.group(1).split(',')]
        for a in alts:
            assert glob.glob(os.path.join(root, 'profiles', a)), a
    for t in set(re.findall(r'`(?:python |bash )?((?:tests/)?tools/[a-z_]+\.(?:py|sh))', text)):
        assert os.path.exists(os.path.join(root, t)), t


def test_readme_and_design_cite_tools_and_tests_that_exist():
    """Every `tools/...`, `tests/tools/...`, `tests/...py` path and every `tests/...::test_name` the top-level documents cite is in the tree."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for doc in ('README.md', 'DESIGN.md', 'INTEGRATION.md', os.path.join('profiles', 'INDEX.md'), os.path.join('profiles', 'EXPERIMENTS.md')):
        text = open(os.path.join(root, doc)).read()
        for path in set(re.findall(r'((?:tests/)?tools/(?:micro/)?[A-Za-z_0-9]+\.(?:py|sh|hip))', text)) | set(re.findall(r'(tests/[A-Za-z_0-9/]+\.(?:py|c|cpp|json))', text)) \
                | set(re.findall(r'(profiles/[A-Za-z_0-9]+\.(?:json|txt|csv|patch))', text)):
            assert os.path.exists(os.path.join(root, path)), (doc, path)
        for path, name in set(re.findall(r'(tests/[A-Za-z_0-9]+\.py)::([A-Za-z_0-9]+)', text)):
            src = open(os.path.join(root, path)).read()
            assert re.search(r'def %s[A-Za-z_0-9]*\(' % re.escape(name.rstrip('_')), src), (doc, path, name)


def test_make_leaves_what_gym_make_leaves_on_the_object():
    """`gym.make('<Class>-v0')` of the reference hands back a TimeLimit-wrapped env (hrl_pybullet_envs/__init__.py:11-16): trainers read
    `env._max_episode_steps` and `env.spec.id` / `env.spec.max_episode_steps` off it.  make() of this package leaves the same (no GPU touched)."""
    import hrl_pybullet_envs_amd as H
    env = H.make('AntMazeBulletEnv-v0', num_envs=4)
    assert env._max_episode_steps == 2000 and env.spec.id == 'AntMazeBulletEnv-v0' and env.spec.max_episode_steps == 2000
    assert env.spec.entry_point.endswith(':AntMazeBulletEnv') and env.spec.kwargs == {'num_envs': 4}
    env._max_episode_steps = 500            # the idiom of trainers that shorten episodes
    assert env.max_episode_steps == 500
    one = H.AntGatherBulletEnv()            # constructed directly: no wrapper, no limit, no spec -- like the reference's bare class
    assert one._max_episode_steps is None and one.spec is None


def _fake_kfd(root, simds):
    for i, s in enumerate(simds):
        d = root / str(i)
        d.mkdir()
        (d / 'properties').write_text(f'cpu_cores_count {0 if s else 64}\nsimd_count {s}\nmem_banks_count 1\ngfx_target_version {90500 if s else 0}\n')


def test_bench_preflight_counts_gpus_before_anything_touches_one(tmp_path, monkeypatch):
    """`bench.py --gpus N` on a machine with fewer GPUs: one line and exit code 2 from the PARENT (and from each rank of the driver's own
    torch.distributed.run, before the rendezvous), instead of N ranks waiting in init_process_group until the watchdog ends them 900 s later.
    GPUs = KFD topology nodes with simd_count != 0 (a fake sysfs root here), narrowed by ROCR_ / HIP_VISIBLE_DEVICES; gloo rehearsals are let through."""
    import importlib
    import subprocess
    import bench
    importlib.reload(bench)
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    root = tmp_path / 'nodes'
    root.mkdir()
    _fake_kfd(root, [0, 0, 1024, 1024])   # two CPU nodes, two GPUs
    assert bench.count_gpus(str(root)) == 2
    assert bench.count_gpus(str(tmp_path / 'absent')) is None
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '1')
    assert bench.count_gpus(str(root)) == 1
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    monkeypatch.setenv('HRL_KFD_ROOT', str(root))
    called = []
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: called.append(cmd) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and not called            # not started
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '3'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(called) == 1      # enough GPUs: the ranks are spawned
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--backend', 'gloo'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(called) == 2      # a gloo rehearsal shares the card on purpose
    # a rank of the driver's own launch: same check before the rendezvous, in a real process, within seconds, nothing imported from torch
    env = dict(os.environ, WORLD_SIZE='8', RANK='3', LOCAL_RANK='3', MASTER_ADDR='127.0.0.1', MASTER_PORT='1', HRL_KFD_ROOT=str(root))
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3'], capture_output=True, text=True, env=env, timeout=60)
    assert p.returncode == 2 and 'shows 2 GPU(s)' in p.stderr and p.stdout == ''
