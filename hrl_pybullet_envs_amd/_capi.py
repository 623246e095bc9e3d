"""ctypes mirror of include/hrl_envs.h (structs + constants only; no library loading here)."""
import ctypes as C

HRL_ABI_VERSION = 7
HRL_ANT_FLAT, HRL_ANT_GATHER, HRL_ANT_MAZE, HRL_POINT_GATHER, HRL_ANT_MAZE_MJ, HRL_ANT_FLAGRUN = 0, 1, 2, 3, 4, 5
HRL_STATE_STRIDE = 32
HRL_QPOS_OFF, HRL_QVEL_OFF, HRL_EPRET_OFF, HRL_INITZ_OFF, HRL_POTENTIAL_OFF = 0, 15, 29, 30, 31
HRL_ITEMS_STRIDE = 32  # the default configs; in general hrl_items_stride(cfg)
HRL_MAX_ITEMS = 64
HRL_MAX_BINS = 64
HRL_MAX_OBS = 256
HRL_AUX_STRIDE = 4
HRL_INFO_STRIDE = 4
HRL_MAX_TARGETS = 64
HRL_MAX_GOALS = 61
HRL_FLAG_GOAL_OFF, HRL_FLAG_START_OFF, HRL_FLAG_SQDIST_OFF, HRL_FLAG_PENDING_OFF = 0, 2, 4, 6  # flagrun items record
HRL_GOAL_STRIDE = 4
HRL_OK, HRL_ERR_BAD_ARG, HRL_ERR_HIP, HRL_ERR_NO_DEVICE = 0, 1, 2, 3


class hrl_model(C.Structure):
    _fields_ = [('gravity', C.c_float), ('timestep', C.c_float), ('frame_skip', C.c_int32),
                ('solver_iters', C.c_int32), ('density', C.c_float), ('torque_scale', C.c_float),
                ('contact_erp', C.c_float), ('limit_erp', C.c_float), ('friction_ground', C.c_float),
                ('friction_robot', C.c_float), ('contact_dist', C.c_float), ('limit_margin', C.c_float),
                ('max_joint_vel', C.c_float), ('limit_max_impulse', C.c_float), ('ground_z', C.c_float),
                ('point_force', C.c_float), ('self_collision', C.c_int32), ('item_collision', C.c_int32), ('step_group', C.c_int32),
                ('linear_damping', C.c_float), ('angular_damping', C.c_float), ('restitution', C.c_float), ('restitution_threshold', C.c_float),
                ('max_contacts', C.c_int32), ('joint_damping', C.c_float), ('joint_armature', C.c_float)]


class hrl_config(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('env_kind', C.c_int32), ('num_envs', C.c_int32),
                ('max_episode_steps', C.c_int32), ('env_id_offset', C.c_int64), ('seed', C.c_uint64),
                ('auto_reset', C.c_int32),
                ('n_food', C.c_int32), ('n_poison', C.c_int32), ('n_bins', C.c_int32),
                ('use_sensor', C.c_int32), ('respawn', C.c_int32),
                ('world_size', C.c_float * 2),
                ('sensor_range', C.c_float), ('sensor_span', C.c_float), ('robot_coll_dist', C.c_float),
                ('robot_object_spacing', C.c_float), ('dying_cost', C.c_float),
                ('target_encoding', C.c_int32), ('sense_target', C.c_int32), ('sense_walls', C.c_int32),
                ('done_at_target', C.c_int32), ('max_steps', C.c_int32), ('targ_dist_rew', C.c_int32),
                ('n_targets', C.c_int32),
                ('tol', C.c_float), ('inner_rew_weight', C.c_float),
                ('targets', (C.c_float * 2) * HRL_MAX_TARGETS),
                ('start_pos', C.c_float * 3),
                ('centroid_n_static', C.c_int32), ('centroid_static_sum', C.c_float * 2),
                ('walk_target', C.c_float * 2),
                ('flag_size', C.c_float), ('flag_max_targets', C.c_int32), ('flag_timeout', C.c_int32),
                ('flag_switch_on_collision', C.c_int32), ('flag_enclosed', C.c_int32), ('flag_max_target_dist', C.c_float),
                ('flag_manual_goals', C.c_int32), ('flag_goal_capacity', C.c_int32),
                ('flag_ant_env_rew_weight', C.c_float), ('flag_path_rew_weight', C.c_float), ('flag_dist_rew_weight', C.c_float), ('flag_goal_reach_rew', C.c_float),
                ('walker_electricity_cost', C.c_float), ('walker_stall_torque_cost', C.c_float), ('walker_joints_at_limit_cost', C.c_float),
                ('model', hrl_model)]

    def copy(self):
        c = hrl_config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(hrl_config))
        return c


class hrl_buffers(C.Structure):
    """Build it with keywords (`hrl_buffers(state=..., obs=...)`) or through `make_buffers`: struct_size is filled in either way."""
    _fields_ = [('struct_size', C.c_uint64),
                ('state', C.c_void_p), ('items', C.c_void_p), ('aux', C.c_void_p), ('actions', C.c_void_p),
                ('obs', C.c_void_p), ('reward', C.c_void_p), ('done', C.c_void_p), ('info', C.c_void_p),
                ('final_obs', C.c_void_p), ('truncated', C.c_void_p), ('goal', C.c_void_p), ('solver_rows', C.c_void_p)]

    def __init__(self, *args, **kw):
        if args:
            raise TypeError('hrl_buffers takes keywords only (its first field is struct_size)')
        super().__init__(**kw)
        self.struct_size = C.sizeof(hrl_buffers)


def make_buffers(state, items, aux, actions, obs, reward, done, info, final_obs=None, truncated=None, goal=None, solver_rows=None):
    """An initialised buffer record from addresses (ints / c_void_p / None), in the order of include/hrl_envs.h."""
    return hrl_buffers(state=state, items=items, aux=aux, actions=actions, obs=obs, reward=reward, done=done, info=info,
                       final_obs=final_obs, truncated=truncated, goal=goal, solver_rows=solver_rows)
