"""Batched environment on one GPU: torch-ROCm tensors for storage and streams, the HIP C-ABI for the step.

Mirrors the reference's Env API batched over N envs (README.md:24-34):
    obs = env.reset(); obs, rew, done, info = env.step(actions)
All tensors stay resident in HBM.  Layout: include/hrl_envs.h.
"""
import ctypes as C

import numpy as np
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
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        L = _lib.lib()
        self.num_envs = cfg.num_envs
        self.obs_dim, self.act_dim = L.hrl_obs_dim(C.byref(cfg)), L.hrl_act_dim(C.byref(cfg))
        self.items_stride = L.hrl_items_stride(C.byref(cfg))  # 32 unless the config holds more than 16 items / 15 manual goals
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            _lib.check(L.hrl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        n, dev, f32 = self.num_envs, self.device, torch.float32
        self.state = torch.zeros(n, K.HRL_STATE_STRIDE, dtype=f32, device=dev)
        self.items = torch.zeros(n, self.items_stride, dtype=f32, device=dev)
        self.aux = torch.zeros(n, K.HRL_AUX_STRIDE, dtype=torch.int32, device=dev)
        # the step's outputs; `final_obs` = the observation of the step that ended an episode (rows of envs that did not finish keep their
        # last terminal observation) and `truncated` = the step limit alone ended it: what a trainer bootstraps a truncated episode from
        # (include/hrl_envs.h).  Read them through the properties below: after step_host() they are refreshed from the host buffers first.
        self._out = {'obs': torch.zeros(n, self.obs_dim, dtype=f32, device=dev), 'reward': torch.zeros(n, dtype=f32, device=dev),
                     'done': torch.zeros(n, dtype=torch.uint8, device=dev), 'info': torch.zeros(n, K.HRL_INFO_STRIDE, dtype=f32, device=dev),
                     'final_obs': torch.zeros(n, self.obs_dim, dtype=f32, device=dev), 'truncated': torch.zeros(n, dtype=torch.uint8, device=dev)}
        if cfg.env_kind == K.HRL_ANT_FLAGRUN:  # the goal being chased and whether this step switched to it: info['target'] (ant_flagrun_env.py:191,199)
            self._out['goal'] = torch.zeros(n, K.HRL_GOAL_STRIDE, dtype=f32, device=dev)
        self._host = None         # pinned host buffers of step_host(), made on first use
        self._host_fresh = False  # the last step wrote its outputs to the host buffers: the device tensors are stale until read
        self._bind()

    def _bind(self):
        """The buffer record handed to the library and what step() hands out as `info`, from the tensors as they are now.  Both are made once:
        the tensors never move, a view costs a microsecond or two of host time, and an eager rollout loop is host-bound long before the GPU is."""
        o = self._out
        self._info_views = {'food_rew': o['info'][:, 0], 'dead_rew': o['info'][:, 1], 'episode_return': o['info'][:, 2], 'episode_length': o['info'][:, 3],
                            # valid where done: the terminal observation (the returned obs of such an env is already the next episode's first
                            # when auto_reset is on) and gym's TimeLimit flag
                            'final_observation': o['final_obs'], 'TimeLimit.truncated': o['truncated']}
        # the items record is state of the gather kinds (item positions) and of every flagrun env (the bookkeeping of set_target(); the goal itself in the
        # manual and near-the-robot modes); the other kinds keep nothing in it, and the library takes NULL for it (include/hrl_envs.h): 128 B per
        # env and step less to read and to write back
        c = self.cfg
        self._uses_items = c.env_kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER, K.HRL_ANT_FLAGRUN)
        goal = o['goal'].data_ptr() if 'goal' in o else None
        if 'goal' in o:  # `target`: the goal after the step; `retargeted`: the rows whose step switched to it (the reference sets info['target'] on those steps only)
            self._info_views['target'] = o['goal'][:, 0:2]
            self._info_views['retargeted'] = o['goal'][:, 2]
        self._bufs = K.make_buffers(self.state.data_ptr(), self.items.data_ptr() if self._uses_items else None, self.aux.data_ptr(), None,
                                    o['obs'].data_ptr(), o['reward'].data_ptr(), o['done'].data_ptr(),
                                    o['info'].data_ptr(), o['final_obs'].data_ptr(), o['truncated'].data_ptr(), goal)
        self._bufs_ref = C.byref(self._bufs)
        # set_goals always hands the items record over: an env that has no use for it is refused by the library with the reason (not a manual env)
        self._bufs_with_items = K.make_buffers(self.state.data_ptr(), self.items.data_ptr(), self.aux.data_ptr(), None,
                                               o['obs'].data_ptr(), o['reward'].data_ptr(), o['done'].data_ptr(),
                                               o['info'].data_ptr(), o['final_obs'].data_ptr(), o['truncated'].data_ptr(), goal)

    def _device_out(self, name):
        """Output tensor `name` on the device.  step_host() leaves the step's outputs in pinned host memory only (that is its point: one
        launch, one synchronisation); whoever then reads a device tensor -- ReturnGatherer's snapshot of info[:, 2], code that mixes step()
        and step_host() -- gets the host values copied over first, once, instead of silently stale ones."""
        if self._host_fresh:
            self._host_fresh = False
            h = self._host
            for k, hk in (('obs', 'obs'), ('reward', 'rew'), ('done', 'done'), ('info', 'info')):
                self._out[k].copy_(h[hk], non_blocking=True)
            # final_obs: the kernel writes rows only where an episode ended; the rows that ended in any host step since the last refresh
            if self._host_ended.any():
                d = torch.from_numpy(self._host_ended)
                self._out['final_obs'][d.to(self.device)] = h['final_obs'][d].to(self.device)
                self._host_ended[:] = False
            self._out['truncated'].copy_(h['trunc'], non_blocking=True)
            if 'goal' in h:
                self._out['goal'].copy_(h['goal'], non_blocking=True)
        return self._out[name]

    def _before_device_launch(self):
        """A launch that writes the device output tensors comes after a step_host(): bring the host outputs over first (they would
        otherwise be copied over the NEW device values at the next read)."""
        if self._host_fresh:
            self._device_out('obs')

    obs = property(lambda self: self._device_out('obs'))
    reward = property(lambda self: self._device_out('reward'))
    done = property(lambda self: self._device_out('done'))
    info = property(lambda self: self._device_out('info'))
    final_obs = property(lambda self: self._device_out('final_obs'))
    truncated = property(lambda self: self._device_out('truncated'))
    goal = property(lambda self: self._device_out('goal'))  # AntFlagrun only: [N, 4] = goal x, y | switched to it in the last step | steps since the goal changed

    def update_config(self, cfg):
        """A changed copy of this env's config takes effect for the launches that follow on the current stream (hrl_update_config): the step limit,
        reward weights, tolerances, engine parameters ... of a LIVE env -- the simulation, every tensor and the pinned host buffers stay as they
        are.  What the tensors' shapes depend on cannot change (the library refuses and nothing happens)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_update_config(self._h, C.byref(cfg), self._stream()))
        self.cfg = cfg

    def count_solver_rows(self, on=True):
        """Diagnostic: from now on every step ADDS to `self.solver_rows` [N] (int32, zeroed here) the constraint rows each env's solver held
        -- joint limits + 3 per contact over the step's substeps (hrl_buffers.solver_rows).  What a launch costs depends on it."""
        self.solver_rows = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device) if on else None
        p = self.solver_rows.data_ptr() if on else None
        self._bufs.solver_rows = p
        self._bufs_with_items.solver_rows = p
        if self._host is not None:   # step_host()'s record too
            self._hbufs.solver_rows = p
        return self.solver_rows

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
        self._before_device_launch()
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
        if self._host_fresh:
            self._before_device_launch()
        if torch.cuda.current_device() == self.device.index:   # the usual case; the context manager costs more host time than the launch
            rc = _lib.lib().hrl_step(self._h, self._bufs_ref, self._stream())
        else:
            with torch.cuda.device(self.device):
                rc = _lib.lib().hrl_step(self._h, self._bufs_ref, self._stream())
        if rc != K.HRL_OK:
            _lib.check(rc)
        self._last_actions = actions  # keep alive until the stream has consumed it
        o = self._out                 # (current: a pending host refresh was brought over before the launch)
        return o['obs'], o['reward'], o['done'], dict(self._info_views)

    def step_host(self, actions):
        """One step with HOST input and outputs -- what the reference's numpy-in / numpy-out `env.step(a)` hands over
        (README.md:24-34), for the one-env classes: the action is written into pinned host memory that the kernel reads
        directly, and observations / reward / done / info land in pinned host memory the kernel writes directly, so a step is
        one launch and one stream synchronisation -- no staging copies, no packing kernel.  The simulation state stays in HBM.
        Returns numpy views (obs [N, obs_dim], reward [N], done [N] uint8, info [N, 4]) that the next call overwrites (the terminal
        observation / truncation flag of this path: `host_final_obs()`); the device tensors `obs` / `reward` / `done` / `info` are
        refreshed from these host buffers when they are next read (`_device_out`), not by this call."""
        if self._host is None:
            f32 = torch.float32
            t = {'act': torch.zeros(self.num_envs, self.act_dim, dtype=f32).pin_memory(),
                 'obs': torch.zeros(self.num_envs, self.obs_dim, dtype=f32).pin_memory(),
                 'rew': torch.zeros(self.num_envs, dtype=f32).pin_memory(),
                 'done': torch.zeros(self.num_envs, dtype=torch.uint8).pin_memory(),
                 'info': torch.zeros(self.num_envs, K.HRL_INFO_STRIDE, dtype=f32).pin_memory(),
                 'final_obs': torch.zeros(self.num_envs, self.obs_dim, dtype=f32).pin_memory(),
                 'trunc': torch.zeros(self.num_envs, dtype=torch.uint8).pin_memory()}
            if 'goal' in self._out:
                t['goal'] = torch.zeros(self.num_envs, K.HRL_GOAL_STRIDE, dtype=f32).pin_memory()
            self._host = t
            self._host_np = {k: v.numpy() for k, v in t.items()}
            self._host_ended = np.zeros(self.num_envs, bool)  # envs whose episode ended in a host step since the device tensors were refreshed
            self._hbufs = K.make_buffers(self.state.data_ptr(), self.items.data_ptr() if self._uses_items else None, self.aux.data_ptr(), t['act'].data_ptr(),
                                         t['obs'].data_ptr(), t['rew'].data_ptr(), t['done'].data_ptr(), t['info'].data_ptr(),
                                         t['final_obs'].data_ptr(), t['trunc'].data_ptr(), t['goal'].data_ptr() if 'goal' in t else None)
            self._hbufs.solver_rows = self._bufs.solver_rows   # the diagnostic counter, when count_solver_rows() switched it on
        h = self._host_np
        h['act'][...] = actions
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_step(self._h, C.byref(self._hbufs), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        self._host_fresh = True
        self._host_ended |= h['done'] != 0
        return h['obs'], h['rew'], h['done'], h['info']

    def host_goal(self):
        """AntFlagrun: [N, 4] numpy view of the last step_host(): goal x, y | 1 if that step switched to it | steps since the goal changed."""
        return self._host_np['goal']

    def host_final_obs(self):
        """(final_obs [N, obs_dim], truncated [N] uint8) numpy views of the last step_host(): valid where its `done` was set."""
        return self._host_np['final_obs'], self._host_np['trunc']

    def set_goals(self, goals, mask=None):
        """AntFlagrun with flag_manual_goals: goals float32 [N, n_goals, 2] on this device (include/hrl_envs.h: hrl_set_goals)."""
        goals = goals.to(device=self.device, dtype=torch.float32).contiguous()
        if goals.dim() != 3 or goals.shape[0] != self.num_envs or goals.shape[2] != 2:
            raise ValueError(f'goals must be [num_envs={self.num_envs}, n_goals, 2], got {tuple(goals.shape)}')
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._before_device_launch()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_set_goals(self._h, C.byref(self._bufs_with_items), goals.data_ptr(), int(goals.shape[1]),
                                                None if m is None else m.data_ptr(), self._stream()))
        self._last_goals = goals  # keep alive until the stream has consumed it
        return self.obs

    def next_target(self, mask=None):
        """AntFlagrun with flag_manual_goals: `env.next_target()` for every (masked) env (include/hrl_envs.h: hrl_next_target).
        Returns (obs, ok): ok[i] == 0 where the reference raises IndexError (empty list; that env is unchanged)."""
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        ok = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
        self._before_device_launch()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_next_target(self._h, C.byref(self._bufs), None if m is None else m.data_ptr(), ok.data_ptr(), self._stream()))
        return self.obs, ok

    # checkpoint / resume: everything an env carries from one step to the next is three tensors -- `state` (pose, velocities, episode
    # return, potential), `items` (item positions / pending goals) and `aux` (step, pickup, episode and goal counters: the keys of the counter-based
    # random streams) -- plus the config (seed, env_id_offset, constructor arguments).  torch.save()-able; resuming continues bit for bit.
    def state_dict(self):
        self._before_device_launch()
        out = {k: getattr(self, k).detach().cpu().clone() for k in ('state', 'items', 'aux', 'obs', 'reward', 'done', 'info', 'final_obs', 'truncated') + (('goal',) if 'goal' in self._out else ())}
        out['config'] = bytes(self.cfg)
        out['abi_version'] = K.HRL_ABI_VERSION
        return out

    def load_state_dict(self, sd, strict=True):
        """Restores a state_dict() of an env with the same config.  strict: the whole config (seed and env_id_offset included: they key the
        random streams) must be equal; strict=False only insists on what the buffers' shapes depend on."""
        if sd.get('abi_version') != K.HRL_ABI_VERSION:
            raise _lib.HrlError(f"checkpoint of ABI v{sd.get('abi_version')}, this library is v{K.HRL_ABI_VERSION}")
        if strict and sd['config'] != bytes(self.cfg):
            raise _lib.HrlError('checkpoint was taken from an env with another config (strict=False skips this check)')
        for k in ('state', 'items', 'aux'):
            if tuple(sd[k].shape) != tuple(getattr(self, k).shape):
                raise _lib.HrlError(f'checkpoint {k} is {tuple(sd[k].shape)}, this env holds {tuple(getattr(self, k).shape)}')
        self._before_device_launch()
        for k in ('state', 'items', 'aux'):
            getattr(self, k).copy_(sd[k])
        for k in ('obs', 'reward', 'done', 'info', 'final_obs', 'truncated', 'goal'):
            if k in sd and k in self._out and tuple(sd[k].shape) == tuple(self._out[k].shape):
                self._out[k].copy_(sd[k])
        return self.obs

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

    def set_state(self, qpos, qvel, observe=True):
        """Teleport: writes qpos [N, 15] / qvel [N, 14] into the state records and -- as the reference does after
        `resetBasePositionAndOrientation` (ant_maze_bullet_env.py:117-121: calc_state(), _get_obs()) -- recomputes the observations of the new
        state (hrl_observe).  Returns them."""
        qpos = qpos.to(device=self.device, dtype=torch.float32).contiguous()
        qvel = qvel.to(device=self.device, dtype=torch.float32).contiguous()
        self._before_device_launch()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_set_state(self._h, C.byref(self._bufs), qpos.data_ptr(), qvel.data_ptr(), self._stream()))
            if observe:
                _lib.check(_lib.lib().hrl_observe(self._h, C.byref(self._bufs), None, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        return self.obs

    def observe(self, mask=None):
        """The observations of the state / items / aux tensors AS THEY ARE (after writing into them: a teleport, moved items, another target):
        hrl_observe.  Nothing else changes.  Rows with mask == 0 keep their observation."""
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._before_device_launch()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hrl_observe(self._h, C.byref(self._bufs), None if m is None else m.data_ptr(), self._stream()))
        return self.obs
