/*
 * hrl_envs.h -- C-ABI of the MI355X-native batched Ant/Point environment step
 * (hot path of sash-a/hrl_pybullet_envs).
 *
 * The reference has NO FFI/plugin ABI: its boundary is the old-gym `gym.Env` class API
 * (reset() -> obs ; step(a) -> (obs, rew, done, info) ; seed(s)), see
 *   hrl_pybullet_envs/__init__.py:11-16, README.md:24-34,
 *   envs/gather/ant_gather_env.py:68-119, envs/ant_maze/ant_maze_bullet_env.py:77-121,
 *   envs/gather/gather_base.py:67-109, envs/MjAnt.py:36-97.
 * Beneath that class API the reference calls pybullet's C-API client (stepSimulation & getters).
 * This header is the ABI a maintainer would bind in place of those pybullet calls: plain
 * pointers + sizes, caller-owned buffers, int status returns, no C++/torch types.
 *
 * Every entry point documents the reference call site it replaces.
 * All `float*`/`uint8_t*`/`int32_t*` buffer arguments of the hip library are DEVICE pointers;
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on
 * that stream.  `actions`, `obs`, `reward`, `done` and `info` may instead be pinned, device-mapped HOST
 * memory (hipHostMalloc): the kernels read / write them in place, which makes a numpy-in / numpy-out
 * step of a few envs one launch and one stream synchronisation (the one-env classes do this,
 * vec_env.py: step_host); `state`, `items` and `aux` belong in HBM.  The library owns nothing but the immutable config copied at hrl_create() (host copy plus one
 * ~0.6 KB device copy of the derived constants, freed by hrl_destroy()).
 */
#ifndef HRL_ENVS_H
#define HRL_ENVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRL_ABI_VERSION 7

/* env kinds */
#define HRL_ANT_FLAT 0     /* AntMjEnv: flat ground, obs 29 (envs/MjAnt.py:31-97)                        */
#define HRL_ANT_GATHER 1   /* AntGatherBulletEnv: obs 26 + 2*n_bins (envs/gather/ant_gather_env.py:12-200) */
#define HRL_ANT_MAZE 2     /* AntMazeBulletEnv: obs 26 + 2 + n_bins (envs/ant_maze/ant_maze_bullet_env.py) */
#define HRL_POINT_GATHER 3 /* PointGatherBulletEnv: obs 8 + 2*n_bins (envs/gather/point_gather_env.py)     */
#define HRL_ANT_MAZE_MJ 4  /* AntMazeMjEnv: obs 29 + 3*n_bins + 1 (envs/ant_maze/ant_maze_mj_env.py:17-78)    */
#define HRL_ANT_FLAGRUN 5  /* AntFlagrunBulletEnv: obs 28 (+ sensor_bins) (envs/ant_flagrun/ant_flagrun_env.py) */

/* buffer geometry (floats / ints per env) */
#define HRL_STATE_STRIDE 32 /* state record: qpos[15] | qvel[14] | ep_return | initial_z | potential */
#define HRL_QPOS_OFF 0      /* x,y,z, qx,qy,qz,qw, hip_1,ankle_1,...,hip_4,ankle_4 (MjAnt.py:19-20)  */
#define HRL_QVEL_OFF 15     /* vx,vy,vz, wx,wy,wz (world), joint rates (MjAnt.py:22-23)              */
#define HRL_EPRET_OFF 29
#define HRL_INITZ_OFF 30
#define HRL_POTENTIAL_OFF 31
#define HRL_ITEMS_STRIDE 32 /* items record of the default configs: 16 items x (x,y), food slots first, then poison slots (the reference's
                               defaults: 8 + 8).  A config with more items (or a longer manual goal list) has a longer record: hrl_items_stride() */
#define HRL_MAX_ITEMS 64    /* n_food + n_poison (ant_gather_env.py:16-17 takes any counts; an item is a lane of the wave here) */
#define HRL_MAX_BINS 64     /* n_bins / sensor_bins */
#define HRL_MAX_OBS 256     /* widest observation: AntMazeMj with 64 bins = 29 + 3 * 64 + 1 = 222 */
#define HRL_AUX_STRIDE 4    /* int32: t_episode, t_lifetime (AntMaze / AntFlagrun: low 28 bits; bits 28..31 = the feet contacts of the last step, what upstream's
                               robot.feet_contact holds when the next step's calc_state() reads it), episode_index, target_index (flagrun: goal index | steps_since_goal_change << 16 | rewarded << 31) */
#define HRL_INFO_STRIDE 4   /* float: food_rew, dead_rew (gather kinds; the locomotion kinds: alive, progress -- `self.rewards[0:2]`, MjAnt.py:82-84), episode_return (running; final when done), episode_len */
#define HRL_MAX_TARGETS 64  /* maze kinds: `targets` of the constructor (ant_maze_bullet_env.py:23) */
#define HRL_MAX_GOALS 61    /* flagrun manual goals (flag_goal_capacity) */
/* flagrun items record (the modes that keep one): current goal | where the robot stood when it got that goal | squared distance to it then
 * (ant_flagrun_env.py:100-103 `_goal_start_pos`, `_sq_dist_goal`: the path reward of :174-176) | spare | the pending list in list order */
#define HRL_FLAG_GOAL_OFF 0
#define HRL_FLAG_START_OFF 2
#define HRL_FLAG_SQDIST_OFF 4
#define HRL_FLAG_PENDING_OFF 6
#define HRL_GOAL_STRIDE 4   /* optional flagrun output `goal`: current goal x, y | 1 if this step switched to it (info['target']) | steps since the goal changed */

/* status codes */
#define HRL_OK 0
#define HRL_ERR_BAD_ARG 1
#define HRL_ERR_HIP 2
#define HRL_ERR_NO_DEVICE 3

/* Rigid-body model + solver parameters.  Defaults: hrl_default_config().  Physics constants the
 * reference fixes in-tree: gravity 9.8, timestep 0.0165/4, frame_skip 4 (ant_gather_env.py:58,
 * ant_maze_bullet_env.py:60); ground/wall friction 0.8, restitution 0.5
 * (sizeable_enclosed_scene.py:60); Ant geometry/friction 1.5 (assets/ant.xml:9-58).  Everything
 * else restates upstream pybullet behaviour from memory (unverified). */
typedef struct hrl_model {
    float gravity;           /* 9.8 */
    float timestep;          /* 0.0165/4 (one physics substep) */
    int32_t frame_skip;      /* 4 substeps per env step */
    int32_t solver_iters;    /* 5 PGS iterations per substep */
    float density;           /* 1000 kg/m^3 */
    float torque_scale;      /* 2.5 * 100 N*m per unit action */
    float contact_erp;       /* 0.9 */
    float limit_erp;         /* 0.2 */
    float friction_ground;   /* 0.8 */
    float friction_robot;    /* 1.5 ant, 0.1 point */
    float contact_dist;      /* 0.02: contacts closer than this become solver rows */
    float limit_margin;      /* 0.25 rad: joint limits closer than this become solver rows */
    float max_joint_vel;     /* 100 rad/s */
    float limit_max_impulse; /* 100 */
    float ground_z;          /* 0.005 = top of the 50x50x0.01 plane box (assets/plane.xml:19) */
    float point_force;       /* 500 N (point_bot.py:29) */
    /* 1: capsules of different legs collide (links that are not ancestors of one another: upstream loads the robot with
     * URDF_USE_SELF_COLLISION | URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS); friction between two
     * ant links = friction_robot^2.  Default 1 for the ant kinds. */
    int32_t self_collision;
    /* 1: the food / poison cubes are static colliders (assets/food.xml:12,19: 0.25 m boxes, centre z = 0.1,
     * gather_scene.py:62).  Default 1 for the gather kinds. */
    int32_t item_collision;
    /* Launch shape of hrl_step for the ant kinds -- no effect on results.  0 (default): four env-waves per 256-thread workgroup, the
     * articulated-body phases and the contact phase of the four envs lane-packed on one wave each.  1: one 64-thread workgroup per env,
     * every phase on the env's own wave (the measurement reference of DESIGN.md 4; same arithmetic, bit for bit). */
    int32_t step_group;
    /* ---- ABI v7: model choices nothing in the reference tree decides (the arithmetic lives in the absent pybullet wheel), as parameters, so
     * that fitting recorded pybullet steps (tools/make_pybullet_golden.py) is a config change.  The defaults are the build's specification
     * (DESIGN.md 3.9); restitution and a tighter contact cap are skipped by a wave-uniform test at their defaults. ---- */
    /* Damping of the bodies' motion in the form Bullet's multibodies apply it (recalled, SURVEY A.3: btMultiBody adds, for the base and every
     * link, the bias force m v k_l (1 + |v|) at the body's centre of mass and the bias torque (I omega) k_a (1 + |omega|), v / omega the body's
     * velocity at the start of the substep; its constructor sets k_l = k_a = 0.04 -- next to the maximal coordinate velocity of 100 that
     * max_joint_vel restates -- and pybullet documents 0.04 as the default of changeDynamics(linearDamping / angularDamping), which the
     * reference never calls).  Default 0.04 / 0.04: a body moving at 1 m/s loses 0.13 % of its speed per env step.  0 / 0 switches the term
     * off (a wave-uniform test; the launch is then 1.2 % shorter). */
    float linear_damping, angular_damping;
    /* restitution of a contact whose bodies approach faster than restitution_threshold along the normal: the normal row then asks for a
     * separating velocity of restitution * (approach speed) (Bullet combines the two bodies' restitutions by their product, SURVEY A.3:
     * ground / walls 0.5 (sizeable_enclosed_scene.py:60) x robot 0 = 0).  Default 0: every contact is plastic. */
    float restitution, restitution_threshold; /* 0, 0.2 m/s */
    /* contacts kept per substep, 1..12 (candidates beyond it are dropped in candidate order: ground, walls, the maze box, cubes, then the second support
     * points of capsules lying flat on a box face, then capsule pairs).
     * Default 12, the most the solver's 44 rows hold. */
    int32_t max_contacts;
    /* assets/ant.xml:8 gives every joint `armature="1" damping="1"` (MuJoCo's rotor inertia, kg m^2, and viscous damping, N m s / rad).  Whether
     * Bullet's MJCF importer honours either is not decidable from the reference tree (the torque scale was argued with the
     * armature ignored); the specification leaves both out.  As parameters: joint_armature is added to every hinge's diagonal entry of the
     * mass matrix (the articulated-body D_j = S_j . I^A S_j + armature), joint_damping * rate is subtracted from the joint torque each substep.
     * Defaults 0, 0. */
    float joint_damping, joint_armature;
} hrl_model;

typedef struct hrl_config {
    int32_t abi_version;       /* HRL_ABI_VERSION */
    int32_t env_kind;          /* HRL_ANT_FLAT .. HRL_ANT_FLAGRUN */
    int32_t num_envs;          /* envs owned by this handle (this GPU's shard) */
    int32_t max_episode_steps; /* 2000 (hrl_pybullet_envs/__init__.py:15); <= 0 disables */
    int64_t env_id_offset;     /* global id of local env 0: RNG streams are keyed by global id */
    uint64_t seed;
    int32_t auto_reset;        /* 1: envs that finish are reset inside hrl_step */
    /* gather task (ant_gather_env.py:16-29, point_gather_env.py:8-21) */
    int32_t n_food, n_poison, n_bins;
    int32_t use_sensor, respawn;
    float world_size[2];
    /* robot_coll_dist > 0: pickup when the SQUARED planar distance is below it (ant_gather_env.py:88-92);
     * <= 0: pickup by contact -- +-1 per contact point between the robot and an item cube in the step's last collision
     * pass (ant_gather_env.py:113-116, gather_base.py:103-106; needs model.item_collision). */
    float sensor_range, sensor_span, robot_coll_dist, robot_object_spacing, dying_cost;
    /* maze task (ant_maze_bullet_env.py:23-25) */
    int32_t target_encoding, sense_target, sense_walls, done_at_target, max_steps, targ_dist_rew;
    int32_t n_targets;
    float tol, inner_rew_weight;
    float targets[HRL_MAX_TARGETS][2];
    float start_pos[3];        /* maze: (-2,-5,0.25) (ant_maze_bullet_env.py:27) */
    /* upstream WalkerBase.calc_state averages x,y over robot.parts, which after the first reset also
     * holds the scene's static bodies: count and summed xy of those statics. */
    int32_t centroid_n_static;
    float centroid_static_sum[2];
    float walk_target[2];      /* flat: (1e3, 0) upstream default; maze: overwritten by the episode's target */
    /* flagrun task (ant_flagrun_env.py:14-16); tolerance -> tol, sensor_bins -> n_bins, use_sensor/sensor_* shared */
    float flag_size;           /* 10: targets ~ U(-size/2, size/2)^2, arena (size+2)^2 */
    int32_t flag_max_targets;  /* 100 goals per episode; the episode ends when they run out (<= 0 with flag_max_target_dist > 0) */
    int32_t flag_timeout;      /* 200 steps without reaching the goal -> next goal */
    int32_t flag_switch_on_collision, flag_enclosed;
    /* > 0 (with flag_max_targets <= 0): every goal is drawn near the robot, per axis +-U(tol, max_target_dist / 2) around
     * its position, redrawn until it lies inside the arena (ant_flagrun_env.py:80-89); the episode then never runs out
     * of goals.  The current goal is kept in items[0..1], so `items` must be provided. */
    float flag_max_target_dist;
    /* manual_goal_creation (ant_flagrun_env.py:27,150-153): reset neither draws goals nor changes the current one; goals
     * come from outside: hrl_set_goals() (`env.goals = [...]; env.next_target()`) and hrl_next_target().  The current goal
     * lives in items[0..1], the pending list from items[HRL_FLAG_PENDING_OFF] on, so `items` must be provided.  As in the reference, next_target()
     * pops the list when flag_max_targets > 0 and draws a goal near the robot (ignoring the list) when flag_max_targets < 1
     * (flag_max_target_dist > 0), :113-116; the constructor's either-or rule (:17-18) holds for manual envs too. */
    int32_t flag_manual_goals;
    /* manual_goal_creation: the longest list `env.goals = [...]` / hrl_set_goals() may hold, 1..HRL_MAX_GOALS (default 15; the record is
     * hrl_items_stride() floats long). */
    int32_t flag_goal_capacity;
    /* ABI v7: the class-level reward weights of AntFlagrunBulletEnv (ant_flagrun_env.py:157-160), read by step() as
     * r = ant_env_rew_weight * r_upstream + path_rew_weight * path_rew - dist_rew_weight * walk_target_dist (:169-178), + goal_reach_rew once per
     * goal (:184-186).  path_rew (:174-176) needs where the robot stood and how far the goal was when the goal was set (:100-103): kept in the
     * items record (HRL_FLAG_START_OFF, HRL_FLAG_SQDIST_OFF) by EVERY flagrun env whatever the weight is -- as the reference's set_target() does --, so a weight
     * switched on for a live env (hrl_update_config) finds them: a flagrun env must be given `items`. */
    float flag_ant_env_rew_weight, flag_path_rew_weight, flag_dist_rew_weight, flag_goal_reach_rew; /* 1, 0, 0, 5000 */
    /* ABI v7: the class-level cost weights of upstream WalkerBaseBulletEnv, which the reward of `super().step()` is made of in AntMazeBulletEnv
     * and AntFlagrunBulletEnv (alive + progress + electricity_cost * mean|a * joint_speed| + stall_torque_cost * mean(a^2)
     * + joints_at_limit_cost * joints_at_limit).  Upstream's values -2.0, -0.1, -0.1 are the default for AntMaze; AntFlagrunBulletEnv.reset()
     * sets all three to 0 ON THE UPSTREAM CLASS (ant_flagrun_env.py:133-135), so a flagrun env has 0, 0, 0 -- and, in the reference, so has every
     * other walker env of the process from then on; the Python classes mirror that through envs/upstream.py. */
    float walker_electricity_cost, walker_stall_torque_cost, walker_joints_at_limit_cost;
    hrl_model model;
} hrl_config;

/* Caller-owned buffers of one shard.  Unused pointers may be NULL (items for non-gather kinds).
 * INITIALISE IT: `hrl_buffers b; hrl_buffers_init(&b);` (or `hrl_buffers b = {sizeof b};`), then assign the pointers you have.  The library
 * reads the optional output pointers of its own ABI version only within `struct_size` bytes and treats the rest as NULL, and refuses a record
 * whose struct_size is not that of a known layout (HRL_ERR_BAD_ARG: what a record left uninitialised on the stack almost surely holds) --
 * a host rebuilt against a newer header never hands the kernels stack garbage as an output address. */
typedef struct hrl_buffers {
    uint64_t struct_size; /* sizeof(hrl_buffers) of the header the CALLER was compiled against (hrl_buffers_init sets it) */
    float *state;         /* [N][HRL_STATE_STRIDE]  in/out */
    float *items;         /* [N][hrl_items_stride()] in/out (gather kinds: the item positions; flagrun: set_target()'s bookkeeping, and the goals of the manual / near-the-robot modes); the other kinds keep nothing in it: pass NULL and the record is neither read nor written */
    int32_t *aux;         /* [N][HRL_AUX_STRIDE]    in/out */
    const float *actions; /* [N][act_dim]           in  (step only) */
    float *obs;           /* [N][obs_dim]           out */
    float *reward;        /* [N]                    out (step only) */
    uint8_t *done;        /* [N]                    out (step only) */
    float *info;          /* [N][HRL_INFO_STRIDE]   out (step only) */
    /* The observation of the step that ENDED an episode (ant_gather_env.py:96,118-119, ant_maze_bullet_env.py:82,97: step() returns the
     * state of the terminal step).  With auto_reset the env is reset inside hrl_step and `obs` then holds the first observation of the NEXT
     * episode; the terminal one -- what a trainer bootstraps the value of a truncated episode from -- is written here.  Rows of envs that
     * did not finish in this step are left untouched.  Written whenever done[i] != 0, with or without auto_reset.  May be NULL. */
    float *final_obs;     /* [N][obs_dim]           out (step only, optional) */
    /* gym TimeLimit (`max_episode_steps=2000`, hrl_pybullet_envs/__init__.py:15): 1 when the episode was ended by the step limit ALONE
     * (gym.wrappers.TimeLimit: info['TimeLimit.truncated'] = not done), else 0; written for every env in every step.  May be NULL. */
    uint8_t *truncated;   /* [N]                    out (step only, optional) */
    /* AntFlagrun (ABI v7): the goal being chased after this step, and whether this step switched to it -- the reference's
     * `info['target'] = self.goal`, set on the steps in which next_target() ran (ant_flagrun_env.py:191,199):
     * goal[i] = {x, y, 1.0 if retargeted in this step else 0.0, steps_since_goal_change}.  Written before an auto-reset.  May be NULL; the other
     * kinds never write it. */
    float *goal;          /* [N][HRL_GOAL_STRIDE]   out (step only, optional) */
    /* Diagnostic (ABI v7): hrl_step ADDS to solver_rows[i] the number of constraint rows env i's solver held in this step -- joint limits + 3 per
     * contact, summed over the step's substeps.  What a launch costs depends on it (the sweeps are serial in the rows), so a measurement reports
     * it next to the time (bench.py: solver_rows_per_env_step).  The caller zeroes it when it wants to.  May be NULL. */
    int32_t *solver_rows; /* [N]                    in/out (step only, optional) */
} hrl_buffers;
#define HRL_BUFFERS_SIZE_V7_BASE ((uint64_t)(sizeof(uint64_t) + 8 * sizeof(void *))) /* through `info`: the least a v7 caller hands over */

typedef struct hrl_handle hrl_handle;

/* Fill `cfg` with the reference's constructor defaults for `env_kind`
 * (ant_gather_env.py:16-29, point_gather_env.py:8-21, ant_maze_bullet_env.py:23-27, MjAnt.py:31-34). */
int hrl_default_config(int32_t env_kind, hrl_config *cfg);

/* obs/action widths implied by a config (ant_gather_env.py:53-55, gather_base.py:54-55,
 * ant_maze_bullet_env.py:54-57, MjAnt.py:15, point_bot.py:15-16). */
int hrl_obs_dim(const hrl_config *cfg);
int hrl_act_dim(const hrl_config *cfg);
/* floats per env of the `items` buffer: HRL_ITEMS_STRIDE (32) for up to 16 items -- every default config --, else the next multiple of 32 that
 * holds 2 * (n_food + n_poison) floats (gather_scene.py:33); a manual_goal_creation flagrun env: HRL_FLAG_PENDING_OFF + 2 * flag_goal_capacity
 * floats rounded up likewise (64 at the default capacity of 15); at most 128. */
int hrl_items_stride(const hrl_config *cfg);

/* Replaces env construction (gym.make / Env.__init__ + first BulletClient): validates and copies cfg. */
int hrl_create(const hrl_config *cfg, hrl_handle **out);
int hrl_destroy(hrl_handle *h);

/* Replaces the handle's config by `cfg`, on `stream` (in order with the launches queued there): what changes on a LIVE env in the reference --
 * `env.max_episode_steps` of gym's TimeLimit, AntFlagrunBulletEnv's class-level reward weights (ant_flagrun_env.py:157-160, read in every step),
 * tolerances, timeouts, the engine parameters of hrl_model ... -- without a new handle: state, items, aux and every buffer of the caller stay as
 * they are.  env_kind, num_envs, what the buffers' shapes depend on (observation / action width, items stride) and what the records' MEANING
 * depends on (n_food / n_poison: new slots would be uninitialised; a flagrun env's goal mode and list capacity) must be what they were, and cfg
 * must be a valid config (HRL_ERR_BAD_ARG, nothing changed; a NULL cfg included).  `seed` MAY change: that is `env.seed(s)` on a live env
 * (ant_gather_env.py:63-66, gather_scene.py:35-36 -- the RNG is reseeded, the simulation carries on): the random streams are counter-based
 * functions of (seed, global env id, counters in aux), so the draws that follow -- respawns, resets, goals -- come from the new seed's streams and
 * nothing else moves.  `env_id_offset` may change likewise (the shard is then taken to hold other global ids from the next draw on).
 * The constants are copied at the call (cfg may be freed on return). */
int hrl_update_config(hrl_handle *h, const hrl_config *cfg, void *stream);

/* Replaces Env.reset() (ant_gather_env.py:68-74, gather_base.py:67-72, ant_maze_bullet_env.py:104-121,
 * upstream WalkerBaseBulletEnv.reset): envs with mask[i] != 0 (all when mask == NULL) are put in the
 * reset distribution and their first observation is written to bufs->obs. */
int hrl_reset(hrl_handle *h, const hrl_buffers *bufs, const uint8_t *mask, void *stream);

/* Replaces Env.step(a): robot.apply_action + scene.global_step (pybullet stepSimulation) +
 * robot.calc_state + task logic (ant_gather_env.py:76-119, gather_base.py:74-109,
 * ant_maze_bullet_env.py:77-97, MjAnt.py:36-97).  One call steps all N envs once. */
int hrl_step(hrl_handle *h, const hrl_buffers *bufs, void *stream);

/* Zeroes `b` and sets b->struct_size. */
int hrl_buffers_init(hrl_buffers *b);

/* The observation of the state AS IT IS in bufs (state, items, aux) into bufs->obs, without stepping: what the reference does after a
 * teleport -- `resetBasePositionAndOrientation(...)`, `robot.calc_state()`, `_get_obs()` (ant_maze_bullet_env.py:117-121; upstream
 * calc_state + ant_gather_env.py:121-125 get_food_obs, sizeable_enclosed_scene.py:63-97 sense_walls, point_bot.py:48-67).  Nothing but `obs`
 * is written (no pickups, no counters, no reward; the feet-contact entries of AntMaze / AntFlagrun are what robot.feet_contact holds in the
 * reference at that point: the flags the LAST STEP left in bits 28..31 of aux[1], zeros after a reset -- a replay that wants zeros clears those
 * bits with the state it writes).  Envs with mask[i] == 0 keep their row
 * (mask == NULL: all).  hrl_set_state() + hrl_observe() is how identical-state parity tests replay the reference's fixtures on the device. */
int hrl_observe(hrl_handle *h, const hrl_buffers *bufs, const uint8_t *mask, void *stream);

/* State access for identical-state parity tests (replaces pybullet get/resetBasePositionAndOrientation,
 * get/resetJointState): copies between the packed state record and split qpos[N][15] / qvel[N][14]. */
int hrl_get_state(hrl_handle *h, const hrl_buffers *bufs, float *qpos, float *qvel, void *stream);
int hrl_set_state(hrl_handle *h, const hrl_buffers *bufs, const float *qpos, const float *qvel, void *stream);

/* AntFlagrunBulletEnv with manual_goal_creation: replaces `env.goals = [...]; env.next_target()`
 * (ant_flagrun_env.py:45,112-120): goals_xy[N][n_goals][2] (device) is every env's list.  As in the reference next_target()
 * is `self.goals.pop()`: the LAST goal of the list becomes the current target at once (set_target + calc_state: bufs->obs is
 * refreshed, the potential is left as it is, :119), the others follow from the back of the list to its front as goals are
 * reached or time out; the episode ends when they run out (IndexError, :193-194).  1 <= n_goals <= flag_goal_capacity.
 * Envs with mask[i] == 0 are left alone (mask == NULL: all).  Rejected when flag_max_targets < 1: next_target() then draws
 * goals near the robot and never reads the list (:113-114). */
int hrl_set_goals(hrl_handle *h, const hrl_buffers *bufs, const float *goals_xy, int32_t n_goals, const uint8_t *mask, void *stream);

/* `env.next_target()` alone (ant_flagrun_env.py:112-120): pops the last pending goal of a manual_goal_creation env, or the next goal of the
 * shared list reset() made (:150-153), or with flag_max_targets < 1 draws a goal near the robot (create_close_target, :80-89); clears
 * _rewarded, refreshes bufs->obs.
 * ok (device, [N], may be NULL): 1, or 0 for an env whose list is empty -- the reference raises IndexError there; such an
 * env is left unchanged.  (The pending list itself is plain data in the caller's `items` / `aux` tensors: `env.goals = [...]`
 * without next_target() is items[HRL_FLAG_PENDING_OFF + 2k..] = goals[k], aux[3] low 16 bits = len(goals) <= flag_goal_capacity.) */
int hrl_next_target(hrl_handle *h, const hrl_buffers *bufs, const uint8_t *mask, uint8_t *ok, void *stream);

/* Last error text of the calling thread ("" if none). */
const char *hrl_last_error(void);

/* Name of the backend that implements this library: "hip-gfx950" for the product. */
const char *hrl_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* HRL_ENVS_H */
