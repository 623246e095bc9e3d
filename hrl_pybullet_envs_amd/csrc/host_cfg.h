/*
 * host_cfg.h -- host side of the C-ABI that does not touch the GPU: reference constructor defaults,
 * observation/action widths, argument validation and the derivation of the kernel constants (DevCfg).
 */
#pragma once
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/hrl_envs.h"
#include "step_core.h"

namespace hrl {

/* reference constructor defaults: ant_gather_env.py:16-29, point_gather_env.py:8-21,
 * ant_maze_bullet_env.py:13-27, MjAnt.py:31-34; scene constants ant_gather_env.py:58 */
inline int default_config(int32_t kind, hrl_config *c) {
    if (!c || kind < HRL_ANT_FLAT || kind > HRL_ANT_FLAGRUN) return HRL_ERR_BAD_ARG;
    memset(c, 0, sizeof(*c));
    c->abi_version = HRL_ABI_VERSION;
    c->env_kind = kind;
    c->num_envs = 1;
    c->max_episode_steps = 2000; /* hrl_pybullet_envs/__init__.py:15 */
    c->n_food = 8; c->n_poison = 8;
    c->n_bins = kind == HRL_POINT_GATHER ? 5 : 10;
    c->use_sensor = 1; c->respawn = 1;
    c->world_size[0] = 15.f; c->world_size[1] = 15.f;
    c->sensor_range = 20.f;
    c->sensor_span = 3.14159265358979323846f;
    c->robot_coll_dist = 1.f; c->robot_object_spacing = 2.f; c->dying_cost = -10.f;
    c->centroid_n_static = 2; c->centroid_static_sum[0] = -7.5f; /* floor (0,0) + last wall (-size_x/2, 0) */
    hrl_model &m = c->model;
    m.gravity = 9.8f; m.timestep = 0.0165f / 4; m.frame_skip = 4; /* ant_gather_env.py:58 */
    m.solver_iters = 5; m.density = 1000.f; m.torque_scale = 250.f;
    m.contact_erp = 0.9f; m.limit_erp = 0.2f;
    m.friction_ground = 0.8f;                                 /* sizeable_enclosed_scene.py:60 */
    m.friction_robot = kind == HRL_POINT_GATHER ? 0.1f : 1.5f; /* player_cube.xml:8, ant.xml:9 */
    m.contact_dist = 0.02f; m.limit_margin = 0.25f; m.max_joint_vel = 100.f; m.limit_max_impulse = 100.f;
    m.ground_z = 0.005f;  /* plane.xml:19 */
    m.point_force = 500.f; /* point_bot.py:29 */
    m.self_collision = kind != HRL_POINT_GATHER; /* URDF_USE_SELF_COLLISION | ..._EXCLUDE_ALL_PARENTS */
    m.item_collision = kind == HRL_ANT_GATHER || kind == HRL_POINT_GATHER; /* food.xml / poison.xml are collidable boxes */
    m.linear_damping = 0.04f; m.angular_damping = 0.04f; /* btMultiBody's built-in damping, pybullet's documented default */
    m.restitution = 0.f; m.restitution_threshold = 0.2f; m.max_contacts = MAXC; m.joint_damping = 0.f; m.joint_armature = 0.f; /* DESIGN.md 3.9 */
    if (kind != HRL_ANT_GATHER && kind != HRL_POINT_GATHER) c->walk_target[0] = 1000.f; /* upstream WalkerBase default walk target (1e3, 0) until the env sets one */
    if (kind == HRL_ANT_MAZE) {
        static const float t[4][2] = {{2, -3}, {2, 0}, {2, 3}, {-2, 4}}; /* ant_maze_bullet_env.py:13-14 */
        c->sensor_range = 5.f; c->sensor_span = 6.28318530717958647692f; c->n_targets = 4;
        for (int i = 0; i < 4; ++i) { c->targets[i][0] = t[i][0]; c->targets[i][1] = t[i][1]; }
        c->sense_walls = 1; c->done_at_target = 1; c->max_steps = -1; c->tol = 1.5f;
        c->start_pos[0] = -2.f; c->start_pos[1] = -5.f; c->start_pos[2] = 0.25f; /* :27 */
        c->centroid_n_static = 3; c->centroid_static_sum[0] = -7.f; /* floor + wall (-5,0) + obstacle (-2,0) */
    }
    if (kind == HRL_ANT_MAZE_MJ) { /* ant_maze_mj_env.py:13-27 */
        static const float t[5][2] = {{2, -4}, {2, 0}, {2, 4}, {0, 4}, {-2, 4}};
        c->sensor_range = 5.f; c->sensor_span = 6.28318530717958647692f; c->n_targets = 5;
        for (int i = 0; i < 5; ++i) { c->targets[i][0] = t[i][0]; c->targets[i][1] = t[i][1]; }
        c->sense_walls = 1; c->done_at_target = 1; c->max_steps = -1; c->tol = 1.5f;
        c->start_pos[0] = -2.f; c->start_pos[1] = -5.f; c->start_pos[2] = 0.25f;
        c->centroid_n_static = 3; c->centroid_static_sum[0] = -7.f;
    }
    c->flag_size = 10.f; c->flag_max_targets = 100; c->flag_timeout = 200; c->flag_switch_on_collision = 1; c->flag_enclosed = 1;
    c->flag_goal_capacity = 15;
    c->flag_ant_env_rew_weight = 1.f; c->flag_path_rew_weight = 0.f; c->flag_dist_rew_weight = 0.f; c->flag_goal_reach_rew = 5000.f; /* ant_flagrun_env.py:157-160 */
    c->walker_electricity_cost = -2.0f; c->walker_stall_torque_cost = -0.1f; c->walker_joints_at_limit_cost = -0.1f; /* upstream WalkerBaseBulletEnv */
    if (kind == HRL_ANT_FLAGRUN) { c->walker_electricity_cost = 0.f; c->walker_stall_torque_cost = 0.f; c->walker_joints_at_limit_cost = 0.f; } /* ant_flagrun_env.py:133-135 */
    if (kind == HRL_ANT_FLAGRUN) { /* ant_flagrun_env.py:14-16; arena (size+2)^2 :59-61; start (0,0,0.25) :144 */
        c->use_sensor = 0; c->n_bins = 8; c->sensor_span = 3.14159265358979323846f; c->sensor_range = 4.f; c->tol = 0.5f;
        c->world_size[0] = 12.f; c->world_size[1] = 12.f; c->start_pos[2] = 0.25f;
        c->centroid_n_static = 2; c->centroid_static_sum[0] = -6.f;
    }
    if (kind == HRL_ANT_FLAT) {
        c->walk_target[0] = 1000.f; c->centroid_n_static = 0; c->centroid_static_sum[0] = 0.f; m.ground_z = 0.f;
    }
    return HRL_OK;
}

inline int obs_dim(const hrl_config *c) {
    const int nfo = c->use_sensor ? 2 * c->n_bins
                                  : 2 * ((c->n_food < c->n_bins ? c->n_food : c->n_bins) + (c->n_poison < c->n_bins ? c->n_poison : c->n_bins));
    switch (c->env_kind) {
        case HRL_ANT_FLAT: return 29;                /* MjAnt.py:15 */
        case HRL_ANT_GATHER: return 28 - 2 + nfo;    /* ant_gather_env.py:53-55 */
        case HRL_POINT_GATHER: return 8 + nfo;       /* gather_base.py:53-55, point_bot.py:16 */
        case HRL_ANT_MAZE: return 28 - 2 + (c->sense_walls ? c->n_bins : 0) + (c->sense_target ? c->n_bins : 2); /* ant_maze_bullet_env.py:54-57 */
        case HRL_ANT_MAZE_MJ: return 29 + 3 * c->n_bins + 1; /* ant_maze_mj_env.py:50 */
        case HRL_ANT_FLAGRUN: return 28 + (c->use_sensor ? c->n_bins : 0); /* ant_flagrun_env.py:53-55 */
    }
    return -1;
}
inline int act_dim(const hrl_config *c) { return c->env_kind == HRL_POINT_GATHER ? 2 : 8; }
/* floats per env of the items buffer: 32 (HRL_ITEMS_STRIDE) unless the config holds more than 16 items or 15 manual goals */
inline int items_stride(const hrl_config *c) {
    int words = 0;
    if (c->env_kind == HRL_ANT_GATHER || c->env_kind == HRL_POINT_GATHER) words = 2 * (c->n_food + c->n_poison);
    if (c->env_kind == HRL_ANT_FLAGRUN && c->flag_manual_goals) words = HRL_FLAG_PENDING_OFF + 2 * c->flag_goal_capacity;
    const int s = (words + 31) / 32 * 32;
    return s < HRL_ITEMS_STRIDE ? HRL_ITEMS_STRIDE : s;
}

/* returns "" when the config is usable, else the reason */
inline std::string validate(const hrl_config *c) {
    char buf[256];
    if (!c) return "null config";
    if (c->abi_version != HRL_ABI_VERSION) return "abi_version mismatch";
    if (c->env_kind < HRL_ANT_FLAT || c->env_kind > HRL_ANT_FLAGRUN) return "unknown env_kind";
    if (c->num_envs <= 0) return "num_envs must be positive";
    const bool gather = c->env_kind == HRL_ANT_GATHER || c->env_kind == HRL_POINT_GATHER;
    if (gather) {
        if (c->n_food < 0 || c->n_poison < 0 || c->n_food + c->n_poison > HRL_MAX_ITEMS) return "n_food + n_poison must be within 0..64 (an item is a lane of the env's wave)";
        if (c->n_bins < 1 || c->n_bins > HRL_MAX_BINS) return "n_bins must be within 1..64";
        if (!(c->robot_coll_dist > 0) && !c->model.item_collision) return "robot_coll_dist <= 0 (contact based pickup, ant_gather_env.py:113-116) needs model.item_collision";
        if (!(c->world_size[0] > 1 && c->world_size[1] > 1 && c->world_size[0] < 50 && c->world_size[1] < 50)) return "world_size must be within (1, 50)";
        if (!(c->sensor_range > 0) || !(c->sensor_span > 0)) return "sensor_range and sensor_span must be positive";
    }
    if (c->env_kind == HRL_ANT_MAZE || c->env_kind == HRL_ANT_MAZE_MJ) {
        if (c->n_targets < 1 || c->n_targets > HRL_MAX_TARGETS) return "n_targets must be within 1..64";
        if (c->n_bins < 1 || c->n_bins > HRL_MAX_BINS) return "n_bins must be within 1..64";
        if (c->target_encoding != 0 && c->target_encoding != 1) return "target_encoding must be 0 (normed_vec) or 1 (angle)"; /* utils.py:66-68 */
        const bool walls = c->env_kind == HRL_ANT_MAZE_MJ || c->sense_walls; /* the Mj variant always senses walls (ant_maze_mj_env.py:58) */
        if ((walls || c->sense_target) && !(c->sensor_span > 0)) return "sensor_span must be positive";
        /* sizeable_enclosed_scene.py:68-71: a span other than 2 pi spaces the rays by i / (n_bins - 1) */
        if (walls && c->n_bins < 2 && c->sensor_span != 6.28318530717958647692f) return "n_bins must be >= 2 unless sensor_span == 2 pi (the wall sensor divides by n_bins - 1)";
        if (!(c->sensor_range > 0)) return "sensor_range must be positive";
    }
    if (c->env_kind == HRL_ANT_FLAGRUN) {
        /* ant_flagrun_env.py:17-18: a goal list (max_targets > 0) or goals near the robot (max_target_dist > 0), never both */
        const bool list_mode = c->flag_max_target_dist == 0.f && c->flag_max_targets > 0, close_mode = c->flag_max_targets <= 0 && c->flag_max_target_dist > 0.f;
        if (!list_mode && !close_mode) return "exactly one of flag_max_targets > 0 (with flag_max_target_dist == 0) and flag_max_target_dist > 0 (with flag_max_targets <= 0) must hold";
        if (list_mode && c->flag_max_targets > 65535) return "flag_max_targets must be <= 65535";
        if (close_mode && !(c->flag_max_target_dist / 2 > c->tol)) return "flag_max_target_dist / 2 must exceed tol (the per-axis offset is drawn from U(tol, max_target_dist / 2))";
        if (c->flag_timeout > 32767) return "flag_timeout must be <= 32767";
        if (!(c->flag_size > 1.0f)) return "flag_size must exceed 1 (targets are rejected within 0.5 of the origin)";
        if (c->use_sensor && (c->n_bins < 1 || c->n_bins > HRL_MAX_BINS)) return "sensor_bins must be within 1..64";
        if (c->flag_manual_goals && (c->flag_goal_capacity < 1 || c->flag_goal_capacity > HRL_MAX_GOALS)) return "flag_goal_capacity must be within 1..61";
        if (c->use_sensor && c->n_bins < 2 && c->sensor_span != 6.28318530717958647692f) return "sensor_bins must be >= 2 unless sensor_span == 2 pi (the wall sensor divides by n_bins - 1)";
    }
    if (obs_dim(c) > HRL_MAX_OBS) { snprintf(buf, sizeof buf, "observation width %d exceeds %d", obs_dim(c), HRL_MAX_OBS); return buf; }
    const hrl_model &m = c->model;
    if (!(m.timestep > 0) || m.frame_skip < 1 || m.frame_skip > 64 || m.solver_iters < 1 || m.solver_iters > 64) return "bad timestep / frame_skip / solver_iters";
    if (!(m.density > 0)) return "density must be positive";
    if (m.max_contacts < 1 || m.max_contacts > MAXC) return "model.max_contacts must be within 1..12";
    if (!(m.linear_damping >= 0) || !(m.angular_damping >= 0) || !(m.restitution >= 0) || !(m.restitution_threshold >= 0) || !(m.joint_damping >= 0) || !(m.joint_armature >= 0))
        return "model damping / restitution / armature parameters must be >= 0";
    if (m.step_group != 0 && m.step_group != 1) return "model.step_group must be 0 (four env-waves per workgroup) or 1 (one wave per env)";
    return "";
}

/* kernel constants; the mass model is computed in double and rounded once to fp32 */
/* Solver rows per substep of an ant that STANDS under config c -- what the straggler rule of ant_env_block compares an env's rows with (scheduling only,
 * no effect on results).  Its four feet rest on the floor: one contact each, three rows per contact, as far as the contact cap lets them in; every
 * hinge sits at a stop (the ankles are outside their range at the reset pose already, assets/ant.xml:21-54; under load the hips follow): one limit
 * row per joint, two where the margin spans the joint's whole range (then both limits are within reach at once).  Defaults: 4 x 3 + 8 = 20. */
inline int standing_rows(const hrl_config &c, const DevCfg &d) {
    const int feet = c.model.max_contacts < 4 ? (c.model.max_contacts < 0 ? 0 : c.model.max_contacts) : 4;
    int rows = 3 * feet;
    for (int j = 0; j < NJ; ++j) rows += (c.model.limit_margin >= d.jhi[j] - d.jlo[j]) ? 2 : 1;
    return rows;
}

inline void build_devcfg(const hrl_config &c, DevCfg &d) {
    memset(&d, 0, sizeof(d));
    d.kind = c.env_kind; d.n_envs = c.num_envs; d.max_episode_steps = c.max_episode_steps; d.auto_reset = c.auto_reset;
    d.env_id_offset = c.env_id_offset; d.seed_lo = (unsigned)c.seed; d.seed_hi = (unsigned)(c.seed >> 32);
    d.n_food = c.n_food; d.n_poison = c.n_poison; d.n_bins = c.n_bins; d.use_sensor = c.use_sensor; d.respawn = c.respawn;
    d.world_sx = c.world_size[0]; d.world_sy = c.world_size[1];
    d.sensor_range = c.sensor_range; d.sensor_span = c.sensor_span; d.coll_dist = c.robot_coll_dist;
    d.spacing = c.robot_object_spacing; d.dying_cost = c.dying_cost;
    d.target_encoding = c.target_encoding; d.sense_target = c.sense_target; d.sense_walls = c.sense_walls;
    d.done_at_target = c.done_at_target; d.max_steps = c.max_steps; d.targ_dist_rew = c.targ_dist_rew; d.n_targets = c.n_targets;
    d.tol = c.tol; d.inner_rew_weight = c.inner_rew_weight;
    for (int i = 0; i < HRL_MAX_TARGETS; ++i) { d.targets[i][0] = c.targets[i][0]; d.targets[i][1] = c.targets[i][1]; }
    for (int k = 0; k < 3; ++k) d.start_pos[k] = c.start_pos[k];
    d.centroid_n_static = c.centroid_n_static; d.centroid_sx = c.centroid_static_sum[0]; d.centroid_sy = c.centroid_static_sum[1];
    d.walk_tx = c.walk_target[0]; d.walk_ty = c.walk_target[1];
    d.span_is_2pi = c.sensor_span == 6.28318530717958647692f; /* sizeable_enclosed_scene.py:68 `s_span == 2*pi` */
    const hrl_model &m = c.model;
    d.h = m.timestep; d.inv_h = 1.0f / m.timestep; d.g = m.gravity; d.erp_c = m.contact_erp; d.erp_l = m.limit_erp;
    d.mu = m.friction_ground * m.friction_robot; d.cdist = m.contact_dist; d.lmargin = m.limit_margin;
    d.vmax = m.max_joint_vel; d.limp_max = m.limit_max_impulse; d.ground_z = m.ground_z;
    d.torque_scale = m.torque_scale; d.point_force = m.point_force;
    d.iters = m.solver_iters; d.nsub = m.frame_skip;
    d.dt = d.h * (float)d.nsub;
    /* Ant bodies (assets/ant.xml:12-58): torso sphere r .25; capsules r .08 of length |(.2,.2)| and |(.4,.4)| */
    const double pi = 3.14159265358979323846, s2 = 1.41421356237309504880;
    const double rho = m.density, rt = 0.25, rc = 0.08, Ls[2] = {0.2 * s2, 0.4 * s2};
    const double msph = rho * 4.0 / 3.0 * pi * rt * rt * rt, Isph = 0.4 * msph * rt * rt;
    double mk[2], Ia[2], It[2];
    for (int k = 0; k < 2; ++k) {
        const double mc = rho * pi * rc * rc * Ls[k], ms = rho * 4.0 / 3.0 * pi * rc * rc * rc, L = Ls[k];
        mk[k] = mc + ms;
        Ia[k] = mc * rc * rc / 2 + ms * 0.4 * rc * rc;
        It[k] = mc * (L * L / 12 + rc * rc / 4) + ms * (0.4 * rc * rc + L * L / 4 + 3 * L * rc / 8);
    }
    const double cl = Ls[0] / 2, common = Isph + 4 * (It[0] + mk[0] * cl * cl), dz = Ia[0] - It[0] - mk[0] * cl * cl;
    d.m0 = (float)(msph + 4 * mk[0]); d.a0 = (float)(common + 2 * dz); d.b0 = (float)(common - (common + 2 * dz));
    d.m1 = (float)mk[0]; d.a1 = (float)It[0]; d.b1 = (float)(Ia[0] - It[0]);
    d.m2 = (float)mk[1]; d.a2 = (float)It[1]; d.b2 = (float)(Ia[1] - It[1]);
    d.L1 = (float)Ls[0]; d.L2 = (float)Ls[1]; d.r_torso = (float)rt; d.r_caps = (float)rc;
    const double lo[NJ] = {-40, 30, -40, -100, -40, -100, -40, 30}, hi[NJ] = {40, 100, 40, -30, 40, -30, 40, 100}; /* ant.xml:18-54 */
    const double d2r = pi / 180.0;
    for (int j = 0; j < NJ; ++j) {
        d.jlo[j] = (float)(lo[j] * d2r); d.jhi[j] = (float)(hi[j] * d2r);
        d.jmid[j] = 0.5f * (d.jlo[j] + d.jhi[j]); d.jscale[j] = 2.0f / (d.jhi[j] - d.jlo[j]);
    }
    /* static world: walls 0.1 thick centred on +-size/2 (sizeable_enclosed_scene.py:46-57, wall.xml:19); maze box (box.xml:19) */
    float hx = 0.f, hy = 0.f;
    if (c.env_kind == HRL_ANT_GATHER || c.env_kind == HRL_POINT_GATHER) { hx = c.world_size[0] / 2; hy = c.world_size[1] / 2; }
    if (c.env_kind == HRL_ANT_FLAGRUN && (c.flag_enclosed || c.use_sensor)) { hx = c.world_size[0] / 2; hy = c.world_size[1] / 2; } /* ant_flagrun_env.py:59-61 */
    const bool maze_world = c.env_kind == HRL_ANT_MAZE || c.env_kind == HRL_ANT_MAZE_MJ;
    if (maze_world) { hx = 5.f; hy = 9.f; } /* maze_scene.py:10 */
    if (hx > 0.f) {
        const float t = 0.05f, n[4][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}};
        const float dd[4] = {-(hx - t), -(hx - t), -(hy - t), -(hy - t)};
        d.n_planes = 4;
        for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) d.plane_n[i][k] = n[i][k]; d.plane_d[i] = dd[i]; }
    }
    if (maze_world) { d.n_boxes = 1; d.box_lo[0] = -5; d.box_lo[1] = -2; d.box_lo[2] = 0; d.box_hi[0] = 1; d.box_hi[1] = 2; d.box_hi[2] = 2; }
    d.flag_size = c.flag_size; d.flag_max_targets = c.flag_max_targets; d.flag_timeout = c.flag_timeout;
    d.flag_switch = c.flag_switch_on_collision; d.flag_mtd = c.flag_max_target_dist; d.flag_manual = c.flag_manual_goals;
    d.self_collision = m.self_collision; d.item_collision = m.item_collision; d.mu_self = m.friction_robot * m.friction_robot;
    d.flag_path_on = c.env_kind == HRL_ANT_FLAGRUN; /* set_target() keeps `_goal_start_pos` / `_sq_dist_goal` whatever the weights are (ant_flagrun_env.py:98-103): a path reward weight switched on for a live env finds them */
    d.w_elec = c.walker_electricity_cost; d.w_stall = c.walker_stall_torque_cost; d.w_jal = c.walker_joints_at_limit_cost;
    d.flag_w_env = c.flag_ant_env_rew_weight; d.flag_w_path = c.flag_path_rew_weight; d.flag_w_dist = c.flag_dist_rew_weight; d.flag_goal_rew = c.flag_goal_reach_rew;
    d.max_contacts = m.max_contacts;
    d.damping_on = (m.linear_damping != 0.f) || (m.angular_damping != 0.f);
    d.damp_lin = m.linear_damping; d.damp_ang = m.angular_damping;
    d.restitution = m.restitution; d.rest_thr = m.restitution_threshold;
    d.jdamp = m.joint_damping; d.armature = m.joint_armature;
    d.obs_dim = obs_dim(&c); d.act_dim = act_dim(&c);
    d.items_stride = items_stride(&c);
    d.item_shift = c.n_food + c.n_poison > 16 ? 6 : 4;
    d.hot_rows = standing_rows(c, d);
}

}  // namespace hrl
