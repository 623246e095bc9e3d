"""Batched environment on one GPU: torch-ROCm tensors for storage and streams, the HIP C-ABI for the step.

Mirrors the reference's Env API batched over N envs (README.md:24-34):
    obs = env.reset(); obs, rew, done, info = env.step(actions)
All tensors stay resident in HBM.  Layout: include/hrl_envs.h.
"""
import ctypes as C

import torch

from . import _capi as K
from . import _lib


class BatchedEnv:
    def __init__(self, cfg, device='cuda:0'):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != 'cuda' or not torch.cuda.is_available():
            raise _lib.HrlError('BatchedEnv needs an MI355X (torch.cuda.is_available() is False or a CPU device was '
                                'requested): the env step has no CPU path')
        L = _lib.lib()
        self.num_envs = cfg.num_envs
        self.obs_dim, self.act_dim = L.hrl_obs_dim(C.byref(cfg)), L.hrl_act_dim(C.byref(cfg))
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            _lib.check(L.hrl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        n, dev, f32 = self.num_envs, self.device, torch.float32
        self.state = torch.zeros(n, K.HRL_STATE_STRIDE, dtype=f32, device=dev)
        self.items = torch.zeros(n, K.HRL_ITEMS_STRIDE, dtype=f32, device=dev)
        self.aux = torch.zeros(n, K.HRL_AUX_STRIDE, dtype=torch.int32, device=dev)
        self.obs = torch.zeros(n, self.obs_dim, dtype=f32, device=dev)
        self.reward = torch.zeros(n, dtype=f32, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.info = torch.zeros(n, K.HRL_INFO_STRIDE, dtype=f32, device=dev)
        self._host = None  # pinned host buffers of step_host(), made on first use
        self._bufs = K.hrl_buffers(self.state.data_ptr(), self.items.data_ptr(), self.aux.data_ptr(), None,
                                   self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                   self.info.data_ptr())

    def close(self):
        if getattr(self, '_h', None):
            _lib.lib().hrl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_reset(self._h, C.byref(self._bufs), None if m is None else m.data_ptr(),
                                            self._stream()))
        return self.obs

    def step(self, actions):
        """actions: float32 [N, act_dim] on this device.  Returns (obs, reward, done, info) -- views of the env's
        own output tensors, overwritten by the next step."""
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous() \
                or tuple(actions.shape) != (self.num_envs, self.act_dim):
            actions = actions.to(device=self.device, dtype=torch.float32).reshape(self.num_envs, self.act_dim).contiguous()
        self._bufs.actions = actions.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_step(self._h, C.byref(self._bufs), self._stream()))
        self._last_actions = actions  # keep alive until the stream has consumed it
        info = {'food_rew': self.info[:, 0], 'dead_rew': self.info[:, 1], 'episode_return': self.info[:, 2],
                'episode_length': self.info[:, 3]}
        return self.obs, self.reward, self.done, info

    def step_host(self, actions):
        """One step with HOST input and outputs -- what the reference's numpy-in / numpy-out `env.step(a)` hands over
        (README.md:24-34), for the one-env classes: the action is written into pinned host memory that the kernel reads
        directly, and observations / reward / done / info land in pinned host memory the kernel writes directly, so a step is
        one launch and one stream synchronisation -- no staging copies, no packing kernel.  The simulation state stays in HBM.
        Returns numpy views (obs [N, obs_dim], reward [N], done [N] uint8, info [N, 4]) that the next call overwrites; the
        device tensors `obs` / `reward` / `done` / `info` are NOT updated by this path."""
        if self._host is None:
            f32 = torch.float32
            t = {'act': torch.zeros(self.num_envs, self.act_dim, dtype=f32).pin_memory(),
                 'obs': torch.zeros(self.num_envs, self.obs_dim, dtype=f32).pin_memory(),
                 'rew': torch.zeros(self.num_envs, dtype=f32).pin_memory(),
                 'done': torch.zeros(self.num_envs, dtype=torch.uint8).pin_memory(),
                 'info': torch.zeros(self.num_envs, K.HRL_INFO_STRIDE, dtype=f32).pin_memory()}
            self._host = t
            self._host_np = {k: v.numpy() for k, v in t.items()}
            self._hbufs = K.hrl_buffers(self.state.data_ptr(), self.items.data_ptr(), self.aux.data_ptr(), t['act'].data_ptr(),
                                        t['obs'].data_ptr(), t['rew'].data_ptr(), t['done'].data_ptr(), t['info'].data_ptr())
        h = self._host_np
        h['act'][...] = actions
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_step(self._h, C.byref(self._hbufs), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        return h['obs'], h['rew'], h['done'], h['info']

    def set_goals(self, goals, mask=None):
        """AntFlagrun with flag_manual_goals: goals float32 [N, n_goals, 2] on this device (include/hrl_envs.h: hrl_set_goals)."""
        goals = goals.to(device=self.device, dtype=torch.float32).contiguous()
        if goals.dim() != 3 or goals.shape[0] != self.num_envs or goals.shape[2] != 2:
            raise ValueError(f'goals must be [num_envs={self.num_envs}, n_goals, 2], got {tuple(goals.shape)}')
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_set_goals(self._h, C.byref(self._bufs), goals.data_ptr(), int(goals.shape[1]),
                                                None if m is None else m.data_ptr(), self._stream()))
        self._last_goals = goals  # keep alive until the stream has consumed it
        return self.obs

    def next_target(self, mask=None):
        """AntFlagrun with flag_manual_goals: `env.next_target()` for every (masked) env (include/hrl_envs.h: hrl_next_target).
        Returns (obs, ok): ok[i] == 0 where the reference raises IndexError (empty list; that env is unchanged)."""
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        ok = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_next_target(self._h, C.byref(self._bufs), None if m is None else m.data_ptr(), ok.data_ptr(), self._stream()))
        return self.obs, ok

    # state access (identical-state parity tests)
    @property
    def qpos(self):
        return self.state[:, K.HRL_QPOS_OFF:K.HRL_QPOS_OFF + 15]

    @property
    def qvel(self):
        return self.state[:, K.HRL_QVEL_OFF:K.HRL_QVEL_OFF + 14]

    def get_state(self):
        qpos = torch.empty(self.num_envs, 15, dtype=torch.float32, device=self.device)
        qvel = torch.empty(self.num_envs, 14, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_get_state(self._h, C.byref(self._bufs), qpos.data_ptr(), qvel.data_ptr(), self._stream()))
        return qpos, qvel

    def set_state(self, qpos, qvel):
        qpos = qpos.to(device=self.device, dtype=torch.float32).contiguous()
        qvel = qvel.to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_set_state(self._h, C.byref(self._bufs), qpos.data_ptr(), qvel.data_ptr(), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
