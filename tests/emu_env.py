"""ctypes binding of tests/emu/libhrl_emu.so (lock-step host executor of the product's wave phases).
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from hrl_pybullet_envs_amd import _capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIBS = {}


def lib(asan=False):
    name = 'libhrl_emu_asan.so' if asan else 'libhrl_emu.so'
    if name not in _LIBS:
        d = os.path.join(ROOT, 'tests', 'emu')
        subprocess.check_call(['make', '-s', '-C', d, name])
        _LIBS[name] = C.CDLL(os.path.join(d, name))
        _LIBS[name].emu_validate.restype = C.c_char_p
    return _LIBS[name]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class EmuEnv:
    def __init__(self, cfg, reverse=False, asan=False):
        self.cfg, self.reverse, self.asan = cfg, int(reverse), asan
        L = lib(asan)
        n = cfg.num_envs
        self.N, self.od, self.ad = n, L.emu_obs_dim(C.byref(cfg)), L.emu_act_dim(C.byref(cfg))
        f = np.float32
        self.state = np.zeros((n, K.HRL_STATE_STRIDE), f)
        self.items = np.zeros((n, L.emu_items_stride(C.byref(cfg))), f)
        self.aux = np.zeros((n, K.HRL_AUX_STRIDE), np.int32)
        self.obs = np.zeros((n, self.od), f)
        self.rew = np.zeros(n, f)
        self.done = np.zeros(n, np.uint8)
        self.info = np.zeros((n, K.HRL_INFO_STRIDE), f)
        self.act = np.zeros((n, self.ad), f)
        self.final_obs = np.zeros((n, self.od), f)
        self.truncated = np.zeros(n, np.uint8)
        self.goal = np.zeros((n, K.HRL_GOAL_STRIDE), f)
        self.solver_rows = np.zeros(n, np.int32)

    def _bufs(self):
        return K.make_buffers(ptr(self.state), ptr(self.items), ptr(self.aux), ptr(self.act), ptr(self.obs),
                              ptr(self.rew), ptr(self.done), ptr(self.info), ptr(self.final_obs), ptr(self.truncated), ptr(self.goal), ptr(self.solver_rows))

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        b = self._bufs()
        rc = lib(self.asan).emu_reset(C.byref(self.cfg), C.byref(b), ptr(m), self.reverse)
        assert rc == 0, lib(self.asan).emu_validate(C.byref(self.cfg))
        return self.obs

    def observe(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        b = self._bufs()
        rc = lib(self.asan).emu_observe(C.byref(self.cfg), C.byref(b), ptr(m), self.reverse)
        assert rc == 0
        return self.obs

    def step(self, actions):
        self.act[...] = np.asarray(actions, np.float32).reshape(self.N, self.ad)
        b = self._bufs()
        rc = lib(self.asan).emu_step(C.byref(self.cfg), C.byref(b), self.reverse)
        assert rc == 0
        return self.obs, self.rew, self.done, self.info
