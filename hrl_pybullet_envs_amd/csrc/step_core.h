/*
 * step_core.h -- one-wavefront-per-environment step of the batched Ant / Point simulator (gfx950).
 *
 * This is the product's hot path: everything behind Env.step() of the reference
 *   robot.apply_action + scene.global_step + robot.calc_state     ant_gather_env.py:77-80 (pybullet, upstream)
 *   pickups / respawn / food sensor / alive / reward              ant_gather_env.py:84-119, gather_scene.py:52-114
 *   wall sensor / target obs / sparse reward                      ant_maze_bullet_env.py:63-97,
 *                                                                 sizeable_enclosed_scene.py:63-97, intersection_utils.py:74-104
 *   PointBot force + state                                        point_bot.py:28-31,48-67
 *   raw-state obs + locomotion reward                             MjAnt.py:17-25,36-97, ant_maze_mj_env.py:57-78
 *   goal chasing (reward, retarget, timeout)                      ant_flagrun_env.py:162-204
 * written as a sequence of wave-wide PHASES.  A phase is a function of (lane, LDS); lanes communicate only
 * through the per-wave LDS record `WaveLds` between phases (plus the wave primitives supplied by the executor:
 * a ballot/prefix compaction, a solver-row lane broadcast, a lane shuffle and per-lane persistent registers).  The executor `X`
 * is the HIP wave (hrl_hip.hip: one 64-thread workgroup = one wavefront = one env, phases separated by a
 * workgroup barrier); tests/emu provides a lock-step host executor so the same phases can be checked on a box
 * without a GPU -- that executor is test infrastructure and is never used by the product.
 *
 * Lane maps used by the phases:
 *   body map: lane >> 2 = rigid body (0-3 feet, 4-7 aux bodies, 8.. torso): phases K1, K2, B; the lanes of a group compute the same values
 *   leg map : leg = lane >> 4 (four 16-lane rows, one per leg)
 *   dof map : dof = lane & 15 (0-2 omega, 3-5 v, 6-13 joint rates, 14-15 zero padding)
 *   shape map: lane = contact shape (torso sphere, 12 leg capsules / their end spheres) x surface (ballot-compacted into contacts)
 *   row map : lane = constraint row (limits, contact normals, friction pairs; <= 44 rows), its solver state in registers
 *   item map: lane = food/poison slot (<= 16);  bin map: lane = sensor bin
 *
 * Numerics: fp32 throughout.  The algorithm is the build's own specification of the
 * rigid-body step; oracle/orc_impl.h restates it independently on the CPU and tests compare the two.
 */
#pragma once
#include <math.h>
#include <stdint.h>

#include <utility>

#ifdef HRL_EMU
#define HRL_DEV inline
#define HRL_PIN_VGPR(x) ((void)0)
#define HRL_PIN_INT(x) ((void)0)
#define HRL_PIN_SCALAR(x) ((void)0)
#else
#define HRL_DEV __device__ __forceinline__
/* Materialise a wave-uniform value in a register at this point (and make it opaque to the optimizer): keeps
 * scalar loads of constants out of the solver loop, and keeps run-time what must not be specialised at compile time. */
#define HRL_PIN_VGPR(x) asm volatile("" : "+v"(x))
#define HRL_PIN_INT(x) asm volatile("" : "+s"(x))
#define HRL_PIN_SCALAR(x) asm volatile("" : "+s"(x)) /* a float constant, loaded by a scalar load here */
#endif


namespace hrl {

constexpr int NJ = 8;     /* hinge joints: hip_1, ankle_1, ..., hip_4, ankle_4 (assets/ant.xml:18-54) */
constexpr int MAXC = 12;  /* contacts kept per substep */
constexpr int MAXR = 44;  /* constraint rows per substep: 8 limits + 12 * (normal + 2 friction) */
constexpr int MAXB = NJ + MAXC; /* bounded rows that come first in the sweep order: joint limits, then contact normals */
constexpr int MAXF = 2 * MAXC;  /* friction rows, a pair per contact, after all the normals */
static_assert(MAXB + MAXF == MAXR && MAXB % 4 == 0 && MAXF % 4 == 0, "row blocks are built in groups of four");

/* Everything the kernels need from hrl_config, plus constants derived from it on the host (host_cfg.h). */
struct DevCfg {
    int kind, n_envs, max_episode_steps, auto_reset;
    long long env_id_offset;
    unsigned seed_lo, seed_hi;
    int n_food, n_poison, n_bins, use_sensor, respawn;
    float world_sx, world_sy, sensor_range, sensor_span, coll_dist, spacing, dying_cost;
    int target_encoding, sense_target, sense_walls, done_at_target, max_steps, targ_dist_rew, n_targets;
    float tol, inner_rew_weight;
    float targets[HRL_MAX_TARGETS][2];
    float start_pos[3];
    int centroid_n_static;
    float centroid_sx, centroid_sy, walk_tx, walk_ty;
    int span_is_2pi;
    float h, inv_h, g, erp_c, erp_l, mu, cdist, lmargin, vmax, limp_max, ground_z, torque_scale, point_force, dt;
    int iters, nsub;
    float m0, a0, b0, m1, a1, b1, m2, a2, b2, L1, L2, r_torso, r_caps;
    float jdamp, armature; /* hrl_model.joint_damping / joint_armature (assets/ant.xml:8; 0, 0: left out of the specification): tau_j - jdamp * rate_j, D_j + armature */
    float jlo[NJ], jhi[NJ], jmid[NJ], jscale[NJ]; /* limits; mid-point and 2/(hi-lo) for the scaled joint observation */
    int n_planes;
    float plane_n[4][3], plane_d[4];
    int n_boxes;
    float box_lo[3], box_hi[3];
    float flag_size, flag_mtd; /* flag_mtd > 0: goals near the robot (ant_flagrun_env.py:80-89), kept in items[0..1] */
    int flag_max_targets, flag_timeout, flag_switch, flag_manual;
    int self_collision, item_collision;
    float mu_self; /* friction between two ant links */
    int obs_dim, act_dim;
    float flag_w_env, flag_w_path, flag_w_dist, flag_goal_rew; /* ant_flagrun_env.py:157-160 */
    float w_elec, w_stall, w_jal; /* upstream WalkerBaseBulletEnv.electricity_cost / stall_torque_cost / joints_at_limit_cost (AntMaze -2, -0.1, -0.1; AntFlagrun 0, 0, 0) */
    int flag_path_on;             /* the env keeps `_goal_start_pos` / `_sq_dist_goal` in its items record: every flagrun env (set_target() does, whatever the reward weights are) */
    int max_contacts;             /* contacts kept per substep, <= MAXC */
    int damping_on;               /* hrl_model.linear_damping / angular_damping (damp_lin / damp_ang) are not both zero: every body gets Bullet's damping wrench */
    float damp_lin, damp_ang;
    float restitution, rest_thr;  /* > 0: normal rows of fast approaches ask for a separating velocity */
    int items_stride; /* floats per env of the items buffer (hrl_items_stride): 32 for the default configs */
    int item_shift;   /* respawn key of a contact pickup: item | move << item_shift; 4 for up to 16 items (the streams of ABI <= 5), else 6 */
    int hot_rows;     /* solver rows per substep of an ant that STANDS under this config (host_cfg.h::standing_rows: its feet's contacts x 3 + its joints at their
                         stops); an env that holds more is the launch's straggler and takes the top issue priority for its block (ant_env_block).  Scheduling only. */
};

struct DevBufs {
    float *state, *items;
    int32_t *aux;
    const float *actions;
    float *obs, *reward;
    uint8_t *done;
    float *info;
    float *final_obs;   /* optional: the observation of the step that ended an episode (include/hrl_envs.h) */
    uint8_t *truncated; /* optional: ended by the step limit alone */
    float *goal;        /* optional, flagrun: goal x, y | switched to it in this step | steps since the goal changed */
    int32_t *rows;      /* optional, diagnostic: += solver rows of the step (limits + 3 x contacts over its substeps) */
    const uint8_t *mask;
    unsigned long long *stamps; /* diagnostic builds only (tools/stamp_profile.py): per-phase cycle sums */
};

/* Per-wave LDS record ("LDS-staged link/joint state"), 7.5 KB, 16-byte aligned (wide LDS accesses).  Three users with disjoint lifetimes share the first
 * block: the K1 -> K2 hand-off of a substep, the velocity responses B of the solver rows (written in phase R1, read
 * until the end of the substep) and the task scratch of the epilogue (observation packing, after the substeps). */
struct alignas(16) WaveLds {
    union {
        float Bt[MAXR + 2][16];  /* B[r][d] = (M^-1 J_r^T)[d]: velocity response of every solver row (+2: the last
                                    group of four friction columns may read two rows past the end, values unused) */
        struct {                 /* phase K1 -> K2 hand-off (dead before the rows are built) */
            float Iaf[4][24];    /* articulated inertia of the foot seen through the ankle (upper triangle) */
            float paf[4][8];     /* its bias force */
        };
        struct {
            float s28[32];       /* upstream 28-vector (WalkerBase.calc_state) */
            float obs[HRL_MAX_OBS];
            float ibin[HRL_MAX_ITEMS]; /* per item: sensor bin (as float, -1 = none) */
            float iint[HRL_MAX_ITEMS]; /* per item: intensity */
            float irew[HRL_MAX_ITEMS]; /* per item: pickup reward */
            float red[16];
            int flags[8];        /* 0: non-finite obs seen, 1: done, 2: ended by the step limit alone, 3: flagrun retarget, 4: flagrun packed goal state */
            float scal[8];       /* 0: reward, 1: food_rew, 2: dead_rew, 3: walk_target_dist, 4: yaw, 5: joints_at_limit, 6-7: parts centroid xy */
        };
    };
    float legI[4][28];   /* per leg, handed to the base: articulated inertia [0..20] (upper triangle), bias force [21..26] */
    float bsum[28];      /* their sum over the legs, (l0 + l1) + (l2 + l3), same layout */
    float lamf[MAXR];    /* final impulses, for the velocity reconstruction */
    float ustar[16];     /* unconstrained velocity (dof order) */
    float st[32];        /* packed state record as stored in HBM */
    float jlim[2][NJ];   /* joint ranges, copied from the constants when the env is loaded (phase L reads them per lane) */
    float items[2 * HRL_MAX_ITEMS]; /* the env's items record (the default configs use the first 32 floats) */
    float items0[32];    /* its first 32 floats as they were loaded: store_env writes back only the entries the step changed (a pickup, a new goal) */
    float act[8];
    int aux[4];
    float q[2][16];      /* ping-pong: substep s reads q[s&1], writes q[(s+1)&1] */
    float u[16], tau[8];
    float XYZ[12];       /* point bot: the columns of its rotation matrix */
    float ph[4][4], pa[4][4], tip[4][4];
    float S[NJ][8], U[NJ][8], cb[NJ][8];
    float invD[NJ], uterm[NJ];
    float Lb[16], idb[8]; /* base articulated inertia = L D L^T: strictly lower part of L (tl() order), 1/D */
    float a0[8];
    float cr[MAXC][4];       /* contact point relative to O */
    float cdir[3][MAXC][4];  /* contact frame: normal, tangent 1, tangent 2 (tangent_basis of the normal) */
    float cdist_[MAXC];
    float cmu[MAXC];     /* friction coefficient of the contact (ground/walls/cubes: mu, link against link: mu_self) */
    int clink[MAXC];     /* level | leg << 2 */
    int clink2[MAXC];    /* self contacts: level | leg << 2 of the second body (leg2 > leg), else -1 */
    int csurf[MAXC];     /* what the contact is with: SURF_* codes */
    float J2[MAXR][2];   /* self-contact rows: the (negated) hip / ankle entries of the second body's leg (substeps with a self contact only) */
    int ljoint[NJ];
    float lsign[NJ], ldist[NJ];
    int gtouch[16];
    int nC, nL, nS, on; /* contacts / limit rows / self contacts found by ant_contacts for this env's substep; on = the record holds an env */
    int ncnt;           /* ant_contacts_group: contacts a packed pass found for this env (added to nC by the next phase) */
    float planes[4][4];  /* lateral half-spaces (n, d), copied from the constants when the env is loaded: the collision passes index them per lane */
    /* (at the END of the record: in the middle, behind csurf, the same two arrays made every launch 0.4 us longer -- they shift the solver's arrays against the LDS banks) */
    float ct[MAXC];      /* contacts of a capsule with a box / cube: the parameter of the contact's point on the capsule's axis (read by second_support) */
    int csph[MAXC];      /* the shape that touches: 0 the torso sphere, 1..12 the capsule that ends in sphere s; -1: a capsule pair */
#ifdef HRL_WGTIME
    int dbg_rows; /* diagnostic build (tools/wg_times.py): solver rows | cube passes << 16 | self-contact substeps << 24, summed over the step */
#endif
};

/* Per-lane registers that live across phases: the solver's working set.  On the GPU these are VGPRs (the row-space
 * solver never touches LDS in its sweeps); every field is fully redefined in every substep. */
struct LaneRegs {
    float ud;                        /* dof map: the lane's velocity component */
    float rI[21], rp[6];             /* body map (phases K1 -> K2 -> B): the lane's rigid-body spatial inertia and bias force */
    float Jb[6], Jh, Ja;             /* row map: the row's Jacobian, sparse: torso twist part + the hip / ankle entries */
    int jslot;                       /* row map: dof slot of Jh (Ja is the next slot): 6 + 2 * leg, | (6 + 2 * leg2) << 8 for WaveLds::J2 */
    float mu;                        /* row map: friction rows: friction coefficient of their contact */
    float An[MAXB], Af[MAXF];        /* row map: the row's line of C = I - D^-1 A (A = J M^-1 J^T): limit/normal columns, friction columns */
    float c, lam, bias, lo, hi;      /* row map: unclamped impulse candidate lam - w / A_ii, impulse, bias, bounds */
    int fn;                          /* row map: friction rows: index of their normal row, else -1 */
};
struct F2b { float ln, dl; }; /* a row's candidate impulse and its change; the change of the row being solved is broadcast */

/* ------------------------------------------------------------------------------------------------ small math */
/* Fused multiply-adds are written out explicitly (and the sources are compiled with -ffp-contract=off) so that the
 * rounding of every operation is part of the algorithm's definition: DESIGN.md 3.7. */
HRL_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
HRL_DEV unsigned float_bits(float f) { unsigned u; __builtin_memcpy(&u, &f, sizeof u); return u; }
HRL_DEV void cross3(float *o, const float *a, const float *b) {
    float x = fma_(a[1], b[2], -(a[2] * b[1])), y = fma_(a[2], b[0], -(a[0] * b[2])), z = fma_(a[0], b[1], -(a[1] * b[0]));
    o[0] = x; o[1] = y; o[2] = z;
}
HRL_DEV float dot3(const float *a, const float *b) { return fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0])); }
HRL_DEV float dot6(const float *a, const float *b) {
    return fma_(a[5], b[5], fma_(a[4], b[4], fma_(a[3], b[3], fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0])))));
}
/* sin and cos for the dynamics (joint rotations, quaternion increment), specified operation by operation so that
 * every fp32 implementation of the step produces the same bits: quadrant k = rint(x * 2/pi), three-term
 * Cody-Waite reduction r = x - k*pi/2, then the classic single-precision minimax polynomials on [-pi/4, pi/4].
 * Accurate to ~1.5 ulp for |x| < 100. */
HRL_DEV void sincos_spec(float x, float *sn, float *cs) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = fma_(-k, 1.5703125f, x);
    r = fma_(-k, 4.837512969970703125e-4f, r);
    r = fma_(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = fma_(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fma_(ps, z, -1.6666654611e-1f);
    const float sr = fma_(ps * z, r, r);
    float pc = fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fma_(pc, z, 4.166664568298827e-2f);
    const float cr = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    /* quadrant; out of int range (exploded states, NaN) the conversion would be implementation-defined: quadrant 0 then */
    const int q = ((int)(fabsf(k) < 1e9f ? k : 0.f)) & 3;
    const float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}
HRL_DEV float sin_spec(float x) { float s, c; sincos_spec(x, &s, &c); return s; }
HRL_DEV float cos_spec(float x) { float s, c; sincos_spec(x, &s, &c); return c; }
/* atan2 and asin of the observation pipeline, specified like sincos_spec so that observations, too, are the same bits
 * on every fp32 implementation: a = min(|x|,|y|) / max(|x|,|y|) in [0, 1]; above tan(pi/8) reduced once more by
 * atan(a) = pi/4 + atan((a-1)/(a+1)); degree-9 odd minimax polynomial on [-tan(pi/8), tan(pi/8)]; then the octant,
 * half-plane and sign.  Max error 2.8e-7 rad.  atan2(0, 0) = 0. */
HRL_DEV float atan2_spec(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax < ay ? ax : ay;
    const float a = mx == 0.0f ? 0.0f : mn / mx;
    const bool big = a > 0.4142135679721832275390625f;
    const float t = big ? (a - 1.0f) / (a + 1.0f) : a;
    const float z = t * t;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fma_(p, z, 1.99777106478e-1f);
    p = fma_(p, z, -3.33329491539e-1f);
    float r = fma_(p * z, t, t);
    if (big) r = 0.785398185253143310546875f + r;
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (x < 0.0f) r = 3.1415927410125732421875f - r;
    return y < 0.0f ? -r : r;
}
HRL_DEV float asin_spec(float x) { return atan2_spec(x, sqrtf((1.0f - x) * (1.0f + x))); }
/* columns of the rotation matrix of the unit quaternion (x, y, z, w) */
HRL_DEV void quat_axes(float x, float y, float z, float w, float *X, float *Y, float *Z) {
    X[0] = fma_(-2.f, fma_(y, y, z * z), 1.f); X[1] = 2.f * fma_(x, y, w * z); X[2] = 2.f * fma_(x, z, -(w * y));
    Y[0] = 2.f * fma_(x, y, -(w * z)); Y[1] = fma_(-2.f, fma_(x, x, z * z), 1.f); Y[2] = 2.f * fma_(y, z, w * x);
    Z[0] = 2.f * fma_(x, z, w * y); Z[1] = 2.f * fma_(y, z, -(w * x)); Z[2] = fma_(-2.f, fma_(x, x, y * y), 1.f);
}
HRL_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* Solver clamp = median of three, specified with the semantics of gfx950's v_med3_f32 so that the device can use the
 * single instruction: any NaN operand -> minimum of the non-NaN operands; zeros ordered -0 < +0. */
HRL_DEV float med3_spec(float x, float lo, float hi) {
#ifdef HRL_EMU
    auto lt = [](float a, float b) { return a < b || (a == b && signbit(a) && !signbit(b)); };
    auto mn = [&](float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (lt(b, a) ? b : a)); };
    auto mx = [&](float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (lt(a, b) ? b : a)); };
    if (isnan(x) || isnan(lo) || isnan(hi)) return mn(mn(x, lo), hi);
    const float m3 = mx(mx(x, lo), hi);
    if (m3 == x && signbit(m3) == signbit(x)) return mx(lo, hi);
    if (m3 == lo && signbit(m3) == signbit(lo)) return mx(x, hi);
    return mx(x, lo);
#else
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#endif
}
/* symmetric 6x6 stored as the upper triangle, row-major (21 floats) */
HRL_DEV constexpr int si(int a, int b) { return a <= b ? a * 6 - (a * (a - 1)) / 2 + (b - a) : b * 6 - (b * (b - 1)) / 2 + (a - b); }
HRL_DEV void sym6_matvec(float *o, const float *A, const float *x) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
        o[i] = fma_(A[si(i, 5)], x[5], fma_(A[si(i, 4)], x[4], fma_(A[si(i, 3)], x[3], fma_(A[si(i, 2)], x[2], fma_(A[si(i, 1)], x[1], A[si(i, 0)] * x[0])))));
}
/* spatial inertia about O of a body with mass m, central inertia alpha*1 + beta*e e^T, COM offset c */
HRL_DEV void spatial_inertia(float *I, float m, float alpha, float beta, const float *e, const float *c) {
    float cc = dot3(c, c);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            float d = (i == j) ? 1.f : 0.f;
            I[si(i, j)] = fma_(m, fma_(-c[i], c[j], cc * d), fma_(beta * e[i], e[j], alpha * d));
            I[si(3 + i, 3 + j)] = m * d;
        }
    /* top-right block = m [c]x */
    I[si(0, 3)] = m * 0.f;   I[si(0, 4)] = m * -c[2]; I[si(0, 5)] = m * c[1];
    I[si(1, 3)] = m * c[2];  I[si(1, 4)] = m * 0.f;   I[si(1, 5)] = m * -c[0];
    I[si(2, 3)] = m * -c[1]; I[si(2, 4)] = m * c[0];  I[si(2, 5)] = m * 0.f;
}
HRL_DEV void crm(float *o, const float *v, const float *m) { /* spatial motion cross product */
    float a[3], b[3], c[3];
    cross3(a, v, m); cross3(b, v, m + 3); cross3(c, v + 3, m);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
HRL_DEV void crf(float *o, const float *v, const float *f) { /* spatial force cross product */
    float a[3], b[3], c[3];
    cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
HRL_DEV void tangent_basis(const float *n, float *t1, float *t2) {
    if (fabsf(n[2]) > 0.70710678118654752440f) {
        float a = fma_(n[1], n[1], n[2] * n[2]), k = 1.f / sqrtf(a);
        t1[0] = 0.f; t1[1] = -n[2] * k; t1[2] = n[1] * k;
        t2[0] = a * k; t2[1] = -n[0] * t1[2]; t2[2] = n[0] * t1[1];
    } else {
        float a = fma_(n[0], n[0], n[1] * n[1]), k = 1.f / sqrtf(a);
        t1[0] = -n[1] * k; t1[1] = n[0] * k; t1[2] = 0.f;
        t2[0] = -n[2] * t1[1]; t2[1] = n[2] * t1[0]; t2[2] = a * k;
    }
}
/* the three row directions of a contact, computed once by the lane that found it (phase C) for its three rows (R1).
 * up: the normal is exactly (0, 0, 1) (ground pass): tangent_basis then has a = 1 and k = 1 exactly, so its results are
 * the constants below (signed zeros as the formula produces them) without the square root and the division. */
HRL_DEV void store_contact_frame(WaveLds &L, int i, const float *n, bool up = false) {
    float t1[3], t2[3];
    if (up) { /* tangent_basis with a = 1, k = 1: (0, -1, 0) and (1, -0, -0), formed from n (as constants the compiler keeps them in
                 registers around the substep loop, and spills them) */
        t1[0] = 0.f; t1[1] = -n[2]; t1[2] = n[1];
        t2[0] = n[2]; t2[1] = -n[0] * t1[2]; t2[2] = n[0] * t1[1];
    }
    else tangent_basis(n, t1, t2);
#pragma unroll
    for (int k = 0; k < 3; ++k) L.cdir[0][i][k] = n[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) L.cdir[1][i][k] = t1[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) L.cdir[2][i][k] = t2[k];
}
/* Square-root-free Cholesky of an SPD 6x6 (si() layout): A = L D L^T with L unit lower triangular; the strictly lower
 * part of L in tl(i, j) order (i > j) and id = 1/D.  x = A^-1 b is then two triangular solves and a scaling. */
HRL_DEV constexpr int tl(int i, int j) { return i * (i - 1) / 2 + j; }
HRL_DEV void ldl6_factor(float *Lm /* [15] */, float *id /* [6] */, const float *A /* [21] */) {
    float d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float v[6], s = A[si(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) v[k] = Lm[tl(j, k)] * d[k];
#pragma unroll
        for (int k = 0; k < j; ++k) s = fma_(-Lm[tl(j, k)], v[k], s);
        d[j] = s; id[j] = 1.f / s;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            float t = A[si(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t = fma_(-Lm[tl(i, k)], v[k], t);
            Lm[tl(i, j)] = t * id[j];
        }
    }
}
HRL_DEV void ldl6_solve(float *x, const float *Lm, const float *id, const float *b) {
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t = fma_(-Lm[tl(i, k)], y[k], t);
        y[i] = t;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float t = y[i] * id[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) t = fma_(-Lm[tl(k, i)], x[k], t);
        x[i] = t;
    }
}
/* Philox4x32-10, keyed like the oracle: key = (seed_lo, seed_hi ^ env_hi), counter = (env_lo, index, w2, w3) */
HRL_DEV void philox4x32(const DevCfg &c, long long env, uint32_t index, uint32_t w2, uint32_t w3, uint32_t *out) {
    uint32_t k0 = c.seed_lo, k1 = c.seed_hi ^ (uint32_t)((unsigned long long)env >> 32);
    uint32_t c0 = (uint32_t)(unsigned long long)env, c1 = index, c2 = w2, c3 = w3;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
HRL_DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

/* ================================================================================================= ANT SUBSTEP */

/* Phase K1 (body map): lane group g = lane >> 2 owns one rigid body -- g 0..3 the foot of leg g, 4..7 the aux body of
 * leg g - 4, 8.. the torso.  Every lane computes the kinematics of its leg, then ONE instruction stream forms the
 * spatial inertia and bias force of all nine bodies at once (the lanes only differ in the parameters they select).
 * The foot lanes go on to the ankle joint (foot -> aux) and publish the leg's kinematic data; the aux and torso
 * lanes keep their rigid-body quantities in registers for phases K2 / B.
 * The two articulated-body joints of a leg are split over two phases so that only one 6x6 inertia is live in
 * registers at a time (<= 128 VGPRs without scratch). */
template <bool POS_ONLY = false> /* POS_ONLY: just the leg points ph / pa / tip of pose q (the parts centroid of the observation) */
HRL_DEV void phase_kin_ankle(const DevCfg &c, WaveLds &L, LaneRegs &g, const float *q, int lane) {
    const float is2 = 0.70710678118654752440f;
    const int grp = lane >> 2, type = grp >> 2, l = grp & 3;
    float x = q[3], y = q[4], z = q[5], w = q[6];
    float X[3], Y[3], Z[3];
    quat_axes(x, y, z, w, X, Y, Z);
    float qh = q[7 + 2 * l], qa = q[8 + 2 * l], qdh = L.u[6 + 2 * l], qda = L.u[7 + 2 * l];
    float ch, sh, ca, sa;
    sincos_spec(qh, &sh, &ch);
    sincos_spec(qa, &sa, &ca);
    /* leg direction signs, ankle axes and sigma = (ankle axis) x (leg dir) . z : assets/ant.xml:15-58 */
    float sx = (l == 0 || l == 3) ? 1.f : -1.f, sy = (l < 2) ? 1.f : -1.f;
    float ax = (l & 1) ? 1.f : -1.f, ay = 1.f, sg = (l == 1 || l == 2) ? 1.f : -1.f;
    float e1x = fma_(sx, ch, -(sy * sh)) * is2, e1y = fma_(sx, sh, sy * ch) * is2;
    float axx = fma_(ax, ch, -(ay * sh)) * is2, axy = fma_(ax, sh, ay * ch) * is2;
    float e1[3], axw[3], e2[3], ph[3], pa[3], tip[3], caux[3], cfoot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e1[k] = fma_(e1y, Y[k], e1x * X[k]);
        axw[k] = fma_(axy, Y[k], axx * X[k]);
        e2[k] = fma_(sg * sa, Z[k], ca * e1[k]);
        ph[k] = 0.2f * fma_(sy, Y[k], sx * X[k]);
        pa[k] = fma_(c.L1, e1[k], ph[k]);
        tip[k] = fma_(c.L2, e2[k], pa[k]);
        caux[k] = fma_(c.L1 * 0.5f, e1[k], ph[k]);
        cfoot[k] = fma_(c.L2 * 0.5f, e2[k], pa[k]);
    }
    if (POS_ONLY) {
        if (type == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { L.ph[l][k] = ph[k]; L.pa[l][k] = pa[k]; L.tip[l][k] = tip[k]; }
        }
        return;
    }
    const int jh = 2 * l, ja = jh + 1;
    float Sh[6], Sa[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { Sh[k] = Z[k]; Sa[k] = axw[k]; }
    cross3(Sh + 3, ph, Z);
    cross3(Sa + 3, pa, axw);
    float v0[6], vjh[6], vx[6], vja[6], vf[6], cbh[6], cba[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { v0[k] = L.u[k]; vjh[k] = Sh[k] * qdh; vx[k] = v0[k] + vjh[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { vja[k] = Sa[k] * qda; vf[k] = vx[k] + vja[k]; }
    crm(cbh, v0, vjh);
    crm(cba, vx, vja);
    /* the lane's rigid body: mass model, axis, COM (relative to O) and spatial velocity */
    /* the nine constants as scalar loads + lane selects (left to itself the compiler indexes the constant block by `type`,
     * i.e. three vector-memory loads on the leader's dependent path) */
    float m0 = c.m0, a0_ = c.a0, b0_ = c.b0, m1 = c.m1, a1_ = c.a1, b1_ = c.b1, m2 = c.m2, a2_ = c.a2, b2_ = c.b2;
    HRL_PIN_VGPR(m0); HRL_PIN_VGPR(a0_); HRL_PIN_VGPR(b0_); HRL_PIN_VGPR(m1); HRL_PIN_VGPR(a1_); HRL_PIN_VGPR(b1_);
    HRL_PIN_VGPR(m2); HRL_PIN_VGPR(a2_); HRL_PIN_VGPR(b2_);
    const float bm = type == 0 ? m2 : (type == 1 ? m1 : m0);
    const float bal = type == 0 ? a2_ : (type == 1 ? a1_ : a0_), bbe = type == 0 ? b2_ : (type == 1 ? b1_ : b0_);
    float be[3], bc[3], bv[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        be[k] = type == 0 ? e2[k] : (type == 1 ? e1[k] : Z[k]);
        bc[k] = type == 0 ? cfoot[k] : (type == 1 ? caux[k] : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) bv[k] = type == 0 ? vf[k] : (type == 1 ? vx[k] : v0[k]);
    float *If = g.rI, *pAf = g.rp; /* stay in the lane's registers: the aux and torso lanes use them in K2 / B */
    float Iv[6], f[6], ng[3];
    spatial_inertia(If, bm, bal, bbe, be, bc);
    { /* bias force of the body: v x* (I v) - gravity wrench */
        float fg2[3] = {0.f, 0.f, -bm * c.g};
        sym6_matvec(Iv, If, bv); crf(f, bv, Iv); cross3(ng, bc, fg2);
#pragma unroll
        for (int k = 0; k < 3; ++k) { pAf[k] = f[k] - ng[k]; pAf[3 + k] = f[3 + k] - fg2[k]; }
    }
    if (c.damping_on) { /* hrl_model.linear_damping / angular_damping (default 0 / 0: skipped, wave-uniform): the body's bias force gets Bullet's damping
                           wrench -- force m v_c k_l (1 + |v_c|) at the centre of mass, torque (I_c omega) k_a (1 + |omega|) -- about O */
        float wc[3], vc[3], fd[3], nd[3], cf[3];
        cross3(wc, bv, bc);
#pragma unroll
        for (int k = 0; k < 3; ++k) vc[k] = bv[3 + k] + wc[k];
        const float kl = c.damp_lin * (1.f + sqrtf(dot3(vc, vc))), ka = c.damp_ang * (1.f + sqrtf(dot3(bv, bv)));
        const float ew = dot3(be, bv);
#pragma unroll
        for (int k = 0; k < 3; ++k) { fd[k] = (bm * vc[k]) * kl; nd[k] = fma_(bbe * ew, be[k], bal * bv[k]) * ka; }
        cross3(cf, bc, fd);
#pragma unroll
        for (int k = 0; k < 3; ++k) { pAf[k] = pAf[k] + (nd[k] + cf[k]); pAf[3 + k] = pAf[3 + k] + fd[k]; }
    }
    if (type != 0) return; /* aux body: continues in K2; torso: in B */
    /* one destination at a time: stores to consecutive addresses that follow each other merge into wide LDS writes */
    /* (the leg points ph / pa / tip are not published here: their only readers, the collision passes and the parts centroid, take
     * them from a POS_ONLY pass of their own -- ant_contacts runs on another wave at the same time as this phase) */
#pragma unroll
    for (int k = 0; k < 6; ++k) L.S[jh][k] = Sh[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) L.S[ja][k] = Sa[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) L.cb[jh][k] = cbh[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) L.cb[ja][k] = cba[k];
    float Ua[6], Iac[6];
#ifdef HRL_VAR_LEAF_CLOSED_FORM /* A/B build only (profiles/EXPERIMENTS.md 9.a): the ankle's U = I S and D = S . U of the LEAF body in closed form -- the foot is rigid and its
                                   joint axis is fixed in it and perpendicular to its own axis, so U = [alpha w + c x (m d (w x e)); m d (w x e)] and D = alpha + m d^2 is a
                                   constant: what "joint axes with a zero linear part" buys for this joint without any shift of inertias.  Not the specification (other roundings). */
    {
        float tt[3], lin[3], cl[3];
        const float dd = c.L2 * 0.5f, md = m2 * dd;
        cross3(tt, axw, e2);
#pragma unroll
        for (int k = 0; k < 3; ++k) lin[k] = md * tt[k];
        cross3(cl, cfoot, lin);
#pragma unroll
        for (int k = 0; k < 3; ++k) { Ua[k] = fma_(a2_, axw[k], cl[k]); Ua[3 + k] = lin[k]; }
    }
    const float invDa = 1.f / (fma_(m2 * (c.L2 * 0.5f), c.L2 * 0.5f, a2_) + c.armature);
#else
    sym6_matvec(Ua, If, Sa);
    const float invDa = 1.f / (dot6(Sa, Ua) + c.armature);             /* + 0 at the default: the same bits */
#endif
    const float uta = fma_(-c.jdamp, qda, L.tau[ja]) - dot6(Sa, pAf);   /* - 0 * rate at the default: the same bits */
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) If[si(a, b)] = fma_(-(Ua[a] * invDa), Ua[b], If[si(a, b)]);
    sym6_matvec(Iac, If, cba);
    const float ud = uta * invDa;
#pragma unroll
    for (int a = 0; a < 6; ++a) L.paf[l][a] = fma_(Ua[a], ud, pAf[a] + Iac[a]);
#pragma unroll
    for (int a = 0; a < 6; ++a) L.U[ja][a] = Ua[a];
#pragma unroll
    for (int k = 0; k < 21; ++k) L.Iaf[l][k] = If[k];
    L.invD[ja] = invDa; L.uterm[ja] = uta;
}

/* Phase K2 (body map, same lanes as K1): the aux lanes add what the ankle handed over to their rigid part and process
 * the hip joint (aux -> torso).  Every lane runs the stream on its own registers; only the aux lanes publish. */
HRL_DEV void phase_hip(const DevCfg &c, WaveLds &L, const LaneRegs &g, int lane) {
    const int grp = lane >> 2, type = grp >> 2, l = grp & 3, jh = 2 * l;
    float Sh[6], cbh[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { Sh[k] = L.S[jh][k]; cbh[k] = L.cb[jh][k]; }
    float Ix[21], pAx[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) pAx[a] = g.rp[a] + L.paf[l][a];
#pragma unroll
    for (int k = 0; k < 21; ++k) Ix[k] = g.rI[k] + L.Iaf[l][k];
    float Uh[6], Iac[6];
    sym6_matvec(Uh, Ix, Sh);
    const float invDh = 1.f / (dot6(Sh, Uh) + c.armature);
    const float uth = fma_(-c.jdamp, L.u[6 + jh], L.tau[jh]) - dot6(Sh, pAx);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) Ix[si(a, b)] = fma_(-(Uh[a] * invDh), Uh[b], Ix[si(a, b)]);
    sym6_matvec(Iac, Ix, cbh);
    const float ud = uth * invDh;
    if (type != 1) return;
#pragma unroll
    for (int a = 0; a < 6; ++a) L.U[jh][a] = Uh[a];
#pragma unroll
    for (int k = 0; k < 21; ++k) L.legI[l][k] = Ix[k];
#pragma unroll
    for (int a = 0; a < 6; ++a) L.legI[l][21 + a] = fma_(Uh[a], ud, pAx[a] + Iac[a]);
    L.invD[jh] = invDh; L.uterm[jh] = uth;
}

/* Phase S (entry map): what the four legs hand to the base, summed entry by entry, (l0 + l1) + (l2 + l3) */
HRL_DEV void phase_leg_sum(WaveLds &L, int lane) {
    if (lane < 27) L.bsum[lane] = (L.legI[0][lane] + L.legI[1][lane]) + (L.legI[2][lane] + L.legI[3][lane]);
}

/* Phase B (body map; the torso lanes count): torso (rigid part from K1, in registers) + leg sums, factorization of the base articulated inertia, base acceleration. */
HRL_DEV void phase_base(WaveLds &L, const LaneRegs &g, int lane) {
    float I0[21], p0[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) I0[k] = g.rI[k] + L.bsum[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] = g.rp[k] + L.bsum[21 + k];
    float Lm[15], id[6], a0[6];
    ldl6_factor(Lm, id, I0);
    ldl6_solve(a0, Lm, id, p0);
    /* the torso lanes (32..63) hold the base's values; one group of them stores */
    if (lane >= 32 && lane < 48) {
#pragma unroll
        for (int a = 0; a < 6; ++a) L.a0[a] = -a0[a];
#pragma unroll
        for (int k = 0; k < 15; ++k) L.Lb[k] = Lm[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) L.idb[k] = id[k];
    }
}

/* Phase V (dof map): forward pass of the lane's leg, then the unconstrained velocity update into the lane register.
 * Written without branches: every lane runs the joint recursion of ITS leg (the lanes of the torso dofs and the padding lanes that of leg 0,
 * discarded) and picks its value at the end, so every LDS address is known at the top and the loads are one round trip -- as four nested
 * divergent paths (torso angular / torso linear / hip / ankle) the phase was 185 instructions and four dependent LDS round trips on the
 * leader's stream.  Per value the operations are unchanged. */
HRL_DEV float phase_forward_vel(const DevCfg &c, const WaveLds &L, int lane) {
    const int d = lane & 15, jd = d - 6;
    const int j = jd < 0 ? 0 : (jd > 7 ? 7 : jd), jh = j & ~1, ja = jh + 1;
    float a0[6], ap[6], ax_[6], wxv[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) a0[k] = L.a0[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) ap[k] = a0[k] + L.cb[jh][k];
    const float qddh = (L.uterm[jh] - dot6(L.U[jh], ap)) * L.invD[jh];
#pragma unroll
    for (int k = 0; k < 6; ++k) ax_[k] = fma_(L.S[jh][k], qddh, ap[k]) + L.cb[ja][k];
    const float qdda = (L.uterm[ja] - dot6(L.U[ja], ax_)) * L.invD[ja];
    cross3(wxv, L.u, L.u + 3);
    const float a0d = L.a0[d < 6 ? d : 0];
    const float lin = d == 4 ? wxv[1] : (d == 5 ? wxv[2] : wxv[0]);
    const float acc = d < 3 ? a0d : (d < 6 ? a0d + lin : ((jd & 1) ? qdda : qddh));
    const float r = fma_(c.h, acc, L.u[d]);
    return d >= 14 ? 0.f : r;
}

/* velocity response du = M^-1 (generalized impulse) through the articulated-body quantities in LDS */
HRL_DEV void response(const WaveLds &L, const float *phi, int level, int leg, float th, float ta, float *du) {
    const int jh = 2 * leg, ja = jh + 1;
    /* the lanes of a wave hold rows of all three levels: written with selects instead of three divergent branches (which ran one after the
       other, each with its loads inside), so that every LDS address is known at the top; per value the operations are the ones of the branches */
    float p[6], ua = ta, uh = th;
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = level == 2 ? -phi[k] : 0.f;
    {
        const float ua2 = ta - dot6(L.S[ja], p);
        ua = level == 2 ? ua2 : ta;
    }
    {
        float s = ua * L.invD[ja];
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = fma_(L.U[ja][k], s, p[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float pm = p[k] - phi[k]; p[k] = level == 1 ? pm : p[k]; }
    uh = th - dot6(L.S[jh], p);
    {
        float s = uh * L.invD[jh];
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = fma_(L.U[jh][k], s, p[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float pm = p[k] - phi[k]; p[k] = level == 0 ? pm : p[k]; }
    float dv0[6];
    ldl6_solve(dv0, L.Lb, L.idb, p);
#pragma unroll
    for (int a = 0; a < 6; ++a) dv0[a] = -dv0[a];
#pragma unroll
    for (int k = 0; k < 6; ++k) du[k] = dv0[k];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int h_ = 2 * l, a_ = h_ + 1;
        float uhl = (l == leg) ? uh : 0.f, ual = (l == leg) ? ua : 0.f, dvx[6];
        float dqh = (uhl - dot6(L.U[h_], dv0)) * L.invD[h_];
#pragma unroll
        for (int k = 0; k < 6; ++k) dvx[k] = fma_(L.S[h_][k], dqh, dv0[k]);
        float dqa = (ual - dot6(L.U[a_], dvx)) * L.invD[a_];
        du[6 + h_] = dqh; du[6 + a_] = dqa;
    }
    du[14] = 0.f; du[15] = 0.f;
}

/* what a contact is with (WaveLds::csurf; the oracle's ORC_SURF_* codes) */
constexpr int SURF_BOX = 8, SURF_ITEM = 16, SURF_SELF = 64;
/* code of item cube i: SURF_ITEM + i for the first 48 items (every config of ABI <= 5), the items beyond sit behind the 48 capsule-pair codes */
HRL_DEV int surf_item(int i) { return i + (i < 48 ? SURF_ITEM : SURF_SELF); }
constexpr float ITEM_HALF = 0.125f, ITEM_Z = 0.1f; /* assets/food.xml:12,19 (box size 0.25), gather_scene.py:62 */

/* r = contact point relative to O, n = normal towards the body `link`; link2 >= 0: against that ant body (self contact) */
struct Hit { bool ok; float dist, n[3], r[3], mu, t; int link, link2, surf, sph; }; /* t, sph: the axis parameter and the shape of a capsule's contact with a box (WaveLds::ct, csph) */

/* signed distance of a sphere (centre p, radius rad) to the axis-aligned box [lo, hi]; n = unit normal towards the sphere */
HRL_DEV float sphere_vs_box(const float *p, float rad, const float *lo, const float *hi, float *n) {
    float d[3], d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { float cp = clampf(p[k], lo[k], hi[k]); d[k] = p[k] - cp; d2 = fma_(d[k], d[k], d2); }
    if (d2 > 0.f) {
        const float len = sqrtf(d2), il = 1.f / len;
        n[0] = d[0] * il; n[1] = d[1] * il; n[2] = d[2] * il;
        return len - rad;
    }
    if (!(d2 == 0.f)) { n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; return 1e30f; } /* non-finite centre: no contact */
    /* centre inside the box: leave through the nearest face */
    int best = 0; float bd = 1e30f, sgn = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float dl = p[k] - lo[k], dh = hi[k] - p[k];
        if (dl < bd) { bd = dl; best = k; sgn = -1.f; }
        if (dh < bd) { bd = dh; best = k; sgn = 1.f; }
    }
    n[0] = best == 0 ? sgn : 0.f; n[1] = best == 1 ? sgn : 0.f; n[2] = best == 2 ? sgn : 0.f;
    return -bd - rad;
}

/* Parameter t in [0, 1] of the point of the segment P(t) = p + t d closest to the axis-aligned box [lo, hi]; where a whole stretch of the
 * segment is closest (it runs alongside a face, or through the box) the middle of that stretch.  g(t) = d . (P(t) - clamp(P(t), lo, hi)) is half
 * the derivative of the squared distance: nondecreasing, piecewise linear, with corners where a coordinate of P crosses a face.  Candidates: 0, 1
 * and the six crossing times clamped to [0, 1]; a = the largest candidate with g <= 0, b = the smallest with g >= 0; no candidate lies strictly
 * between them, so g is linear there: b <= a is the stretch g = 0 (its middle is taken), else the root of the chord; g(0) > 0: t = 0,
 * g(1) < 0: t = 1.  Written without branches (every lane of a collision pass holds another capsule); the operations and their order are the
 * oracle's (orc_impl.h: seg_box_t). */
HRL_DEV float seg_box_t(const float *p, const float *d, const float *lo, const float *hi) {
    float T[8], a = -1.f, ga = 0.f, b = 2.f, gb = 0.f;
    bool on[8]; /* candidate 2 + 2k / 3 + 2k IS the crossing of the low / high face of axis k (not clamped to an end of the segment) */
    T[0] = 0.f; T[1] = 1.f; on[0] = on[1] = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float inv = d[k] != 0.f ? 1.f / d[k] : 0.f;
        const float tl = (lo[k] - p[k]) * inv, th = (hi[k] - p[k]) * inv;
        T[2 + 2 * k] = clampf(tl, 0.f, 1.f); T[3 + 2 * k] = clampf(th, 0.f, 1.f);
        on[2 + 2 * k] = (d[k] != 0.f) & (T[2 + 2 * k] == tl); on[3 + 2 * k] = (d[k] != 0.f) & (T[3 + 2 * k] == th);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float e[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = fma_(d[k], T[i], p[k]);
            e[k] = x - clampf(x, lo[k], hi[k]);
            if (i >= 2 && k == (i - 2) / 2) e[k] = on[i] ? 0.f : e[k]; /* on the face by construction: exactly, not to rounding -- the stretch g = 0 is then found by comparisons with 0 */
        }
        const float g = dot3(d, e);
        const bool ta = (g <= 0.f) & (T[i] > a), tb = (g >= 0.f) & (T[i] < b);
        a = ta ? T[i] : a; ga = ta ? g : ga;
        b = tb ? T[i] : b; gb = tb ? g : gb;
    }
    const float mid = 0.5f * (a + b), root = fma_(b - a, ga / (ga - gb), a);
    float t = a < b ? root : mid;
    t = b > 1.f ? 1.f : t;
    t = a < 0.f ? 0.f : t;
    return t;
}

/* Broad phase of a cube pass, per lane: can shape `sph` (0 torso sphere, else the capsule that ends in sphere sph) of pose q come within the
 * contact distance of item cube `item` at all?  The capsule's bounding box, grown by radius + contact distance (+ 1 mm, so that the cull never
 * decides a case the distance computation would see differently by rounding), against the cube's box, axis by axis.  No lane of a pass says yes:
 * the pass finds nothing and is skipped -- the same contact list as running it.  A non-finite coordinate fails every comparison (such a
 * capsule touches nothing: sphere_vs_box). */
HRL_DEV bool shape_near_item(const DevCfg &c, const WaveLds &L, const float *q, int sph, int item) {
    if (sph < 0 || sph >= 13 || item < 0) return false;
    float r_torso = c.r_torso, r_caps = c.r_caps;
    HRL_PIN_SCALAR(r_torso);
    HRL_PIN_SCALAR(r_caps);
    const int leg = sph > 0 ? (sph - 1) / 3 : 0, level = sph > 0 ? (sph - 1) % 3 : 0;
    const float m = ((sph > 0 ? r_caps : r_torso) + c.cdist) + 1e-3f;
    const float *e1 = level == 0 ? L.ph[leg] : (level == 1 ? L.pa[leg] : L.tip[leg]), *e0 = level == 1 ? L.ph[leg] : L.pa[leg];
    const float ic[3] = {L.items[2 * item], L.items[2 * item + 1], ITEM_Z};
    bool near = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = sph > 0 ? e1[k] : 0.f, b = level == 0 ? 0.f : e0[k];
        const float pa = q[k] + a, pb = q[k] + b, mn = pa < pb ? pa : pb, mx = pa < pb ? pb : pa;
        near = near & (mx + m >= ic[k] - ITEM_HALF) & (mn - m <= ic[k] + ITEM_HALF);
    }
    return near;
}

/* Phase C helper: signed distance of contact shape `sph` (-1 = idle lane) to surface f: 0 ground, 1..n_planes lateral half-spaces,
 * n_planes+1.. world boxes; item >= 0: the item cube `item`.  Against the planes the shapes are the 13 spheres 0 torso, 1+3l hip point,
 * 2+3l ankle point, 3+3l foot tip (the deepest point of a capsule against a plane is one of its ends); against the convex boxes -- the maze
 * box, the food / poison cubes -- shape s > 0 is the whole CAPSULE that ends in sphere s (assets/ant.xml:16-55: O -> hip point, rigid with the
 * torso; hip -> ankle point, the aux body; ankle point -> tip, the foot): the point of its axis closest to the box stands in for the sphere
 * centre.  Shape and sphere share the owning body, so the lane map and the candidate order are the same in every pass. */
HRL_DEV Hit sphere_vs_surface(const DevCfg &c, const WaveLds &L, const float *q, int sph, int f, int item) {
    Hit h;
    float r_torso = c.r_torso, r_caps = c.r_caps, ctr[3] = {0.f, 0.f, 0.f};
    HRL_PIN_SCALAR(r_torso); /* two scalar loads and a select: left alone, the compiler selects the ADDRESS per lane and */
    HRL_PIN_SCALAR(r_caps);  /* fetches the radius with a vector memory load in the middle of the collision pass           */
    h.ok = false; h.dist = 0.f; h.link = 0; h.link2 = -1; h.surf = 0; h.mu = c.mu; h.t = 0.f; h.sph = sph;
    h.n[0] = h.n[1] = 0.f; h.n[2] = 1.f; h.r[0] = h.r[1] = h.r[2] = 0.f;
    if (sph < 0 || sph >= 13) return h;
    int level = 0, leg = 0;
    const float rad = sph > 0 ? r_caps : r_torso;
    if (sph > 0) {
        leg = (sph - 1) / 3; level = (sph - 1) % 3;
        const float *src = level == 0 ? L.ph[leg] : (level == 1 ? L.pa[leg] : L.tip[leg]);
        ctr[0] = src[0]; ctr[1] = src[1]; ctr[2] = src[2];
    }
    h.link = level | (leg << 2);
    float p[3] = {q[0] + ctr[0], q[1] + ctr[1], q[2] + ctr[2]};
    if (item >= 0 || f > c.n_planes) { /* a convex box: the capsule's axis point closest to it */
        float lo[3], hi[3];
        if (item >= 0) {
            const float ix = L.items[2 * item], iy = L.items[2 * item + 1];
            lo[0] = ix - ITEM_HALF; lo[1] = iy - ITEM_HALF; lo[2] = ITEM_Z - ITEM_HALF; hi[0] = ix + ITEM_HALF; hi[1] = iy + ITEM_HALF; hi[2] = ITEM_Z + ITEM_HALF;
            h.surf = surf_item(item);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = c.box_lo[k]; hi[k] = c.box_hi[k]; }
            h.surf = SURF_BOX + (f - 1 - c.n_planes);
        }
        /* start of the axis relative to O: the O -> hip capsule starts at O, and so does the torso sphere (a segment of length zero) */
        const float *s0 = level == 1 ? L.ph[leg] : L.pa[leg];
        float c0[3], pw[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { c0[k] = level == 0 ? 0.f : s0[k]; pw[k] = q[k] + c0[k]; d[k] = ctr[k] - c0[k]; }
        const float t = seg_box_t(pw, d, lo, hi);
        h.t = t;
#pragma unroll
        for (int k = 0; k < 3; ++k) { ctr[k] = fma_(d[k], t, c0[k]); p[k] = q[k] + ctr[k]; }
        h.dist = sphere_vs_box(p, rad, lo, hi, h.n);
    } else if (f == 0) h.dist = (p[2] - c.ground_z) - rad;
    else {
        h.n[0] = L.planes[f - 1][0]; h.n[1] = L.planes[f - 1][1]; h.n[2] = L.planes[f - 1][2];
        h.dist = (dot3(h.n, p) - L.planes[f - 1][3]) - rad; h.surf = f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) h.r[k] = fma_(-rad, h.n[k], ctr[k]);
    h.ok = h.dist < c.cdist;
    return h;
}

/* Is kept contact i of the record a capsule's (first) contact with a convex box -- the maze box or an item cube? */
HRL_DEV bool contact_is_capsule_on_box(const WaveLds &L, int i) {
    return (L.csurf[i] >= SURF_BOX) & (L.csph[i] > 0) & (L.clink2[i] < 0);
}
/* What is asked FIRST of every kept contact (a handful of loads and multiplies, in the ballot that decides whether second_support runs at all): is it a
 * capsule's contact with a box whose normal is a face normal to within 5.7 degrees, and can a point a radius or more along the capsule's axis still be
 * within the contact distance of that face -- it rises by at least |d_kf| rad / |d| against cdist - dist1 (+ 5 mm: just past the face's edge dist1 is the
 * distance to the edge, a hair more than the height over the face)?  A foot that stands on a cube at 30 degrees and more says no.  Implied by
 * second_support's own tests; the oracle rounds it alike (orc_impl.h: second_point). */
HRL_DEV bool second_worth_a_look(const DevCfg &c, const WaveLds &L, int i, int *kf_out) {
    const int sph = L.csph[i];
    const bool on_box = contact_is_capsule_on_box(L, i) & (sph < 13);
    const int s1 = on_box ? sph - 1 : 0, leg = s1 / 3, level = s1 % 3;
    const float a0 = fabsf(L.cdir[0][i][0]), a1 = fabsf(L.cdir[0][i][1]), a2 = fabsf(L.cdir[0][i][2]);
    const int kf = ((a0 >= a1) & (a0 >= a2)) ? 0 : (a1 >= a2 ? 1 : 2);
    *kf_out = kf;
    const bool face = (kf == 0 ? a0 : (kf == 1 ? a1 : a2)) >= 0.995f;
    const float *src = level == 0 ? L.ph[leg] : (level == 1 ? L.pa[leg] : L.tip[leg]), *s0 = level == 1 ? L.ph[leg] : L.pa[leg];
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = src[k] - (level == 0 ? 0.f : s0[k]);
    float r_caps = c.r_caps;
    HRL_PIN_SCALAR(r_caps);
    const float a = (kf == 0 ? d[0] : (kf == 1 ? d[1] : d[2])) * r_caps, b = (c.cdist - L.cdist_[i]) + 0.005f;
    return on_box & face & (a * a <= (b * b) * dot3(d, d));
}
/* Second support point of a capsule that rests (nearly) FLAT on a face of a box -- Bullet keeps a manifold of up to four points per pair there, one point
 * lets the capsule rock about it.  Kept contact i is the capsule's first contact: the axis point P(t1) closest to the box, normal n1.  When n1 is a FACE
 * normal to within 5.7 degrees (largest component >= 0.995: the box's closest point lies in a face's interior, P(t1) is inside the box, or -- a capsule longer
 * than the face -- P(t1) has just passed the face's edge and the normal leans by the capsule's own tilt), the part of the axis that projects into that face is [ta, tb] = [0, 1] clipped by the slabs of the two other axes; its end farther from t1 -- tb, towards the capsule's free end, unless ta
 * is farther by more than a thousandth of the axis (a tie at the exact middle of a stretch is not decided by rounding) -- is the second point P(t2) if it
 * is a radius or more away from P(t1) along the axis.  It keeps the first contact's normal and measures its distance to that face's plane (P(t2) sits on
 * the border of the face's region by construction, where the box's closest feature is a matter of rounding).  The second points follow the first contacts
 * of all boxes and cubes in the candidate list, in their order.  Branch-free; operations and order are the oracle's (orc_impl.h: second_point,
 * shape_vs_box_second). */
HRL_DEV Hit second_support(const DevCfg &c, const WaveLds &L, const float *q, int i) {
    Hit h;
    h.ok = false; h.dist = 0.f; h.link = 0; h.link2 = -1; h.surf = 0; h.mu = c.mu; h.t = 0.f; h.sph = -1;
    h.n[0] = h.n[1] = 0.f; h.n[2] = 1.f; h.r[0] = h.r[1] = h.r[2] = 0.f;
    if (i < 0 || i >= MAXC) return h;
    const int surf = L.csurf[i], sph = L.csph[i];
    if (!contact_is_capsule_on_box(L, i) || sph >= 13) return h;
    float r_caps = c.r_caps;
    HRL_PIN_SCALAR(r_caps);
    const float rad = r_caps;
    const int leg = (sph - 1) / 3, level = (sph - 1) % 3;
    float lo[3], hi[3];
    const int item = surf >= SURF_SELF ? surf - SURF_SELF : (surf >= SURF_ITEM ? surf - SURF_ITEM : -1); /* the inverse of surf_item(); -1: the maze box */
    if (item >= 0) {
        const float ix = L.items[2 * item], iy = L.items[2 * item + 1];
        lo[0] = ix - ITEM_HALF; lo[1] = iy - ITEM_HALF; lo[2] = ITEM_Z - ITEM_HALF; hi[0] = ix + ITEM_HALF; hi[1] = iy + ITEM_HALF; hi[2] = ITEM_Z + ITEM_HALF;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = c.box_lo[k]; hi[k] = c.box_hi[k]; }
    }
    const float *src = level == 0 ? L.ph[leg] : (level == 1 ? L.pa[leg] : L.tip[leg]), *s0 = level == 1 ? L.ph[leg] : L.pa[leg];
    float c0[3], pw[3], d[3], n1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { c0[k] = level == 0 ? 0.f : s0[k]; pw[k] = q[k] + c0[k]; d[k] = src[k] - c0[k]; n1[k] = L.cdir[0][i][k]; }
    const float t1 = L.ct[i];
    const float a0 = fabsf(n1[0]), a1 = fabsf(n1[1]), a2 = fabsf(n1[2]);
    const int kf = ((a0 >= a1) & (a0 >= a2)) ? 0 : (a1 >= a2 ? 1 : 2);
    const bool face = (kf == 0 ? a0 : (kf == 1 ? a1 : a2)) >= 0.995f;
    float ta = 0.f, tb = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool use = (k != kf) & (d[k] != 0.f);
        const float inv = 1.f / d[k];
        const float u = (lo[k] - pw[k]) * inv, v = (hi[k] - pw[k]) * inv;
        const float tl_ = u < v ? u : v, th_ = u < v ? v : u;
        ta = (use & (tl_ > ta)) ? tl_ : ta; tb = (use & (th_ < tb)) ? th_ : tb;
    }
    const float t2 = ((tb - t1) + 1e-3f >= t1 - ta) ? tb : ta, dt = t2 - t1;
    const bool far_enough = (dt * dt) * dot3(d, d) >= rad * rad;
    const float sa = (kf == 0 ? d[0] : (kf == 1 ? d[1] : d[2])) * rad, sb = (c.cdist - L.cdist_[i]) + 0.005f;
    const bool slope_ok = sa * sa <= (sb * sb) * dot3(d, d); /* second_worth_a_look's question, on the same operands */
    const float n1f = kf == 0 ? n1[0] : (kf == 1 ? n1[1] : n1[2]);
    const float sg = n1f > 0.f ? 1.f : -1.f;
    float ctr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ctr[k] = fma_(d[k], t2, c0[k]); h.n[k] = k == kf ? sg : 0.f; }
    const float pf = kf == 0 ? q[0] + ctr[0] : (kf == 1 ? q[1] + ctr[1] : q[2] + ctr[2]);
    const float lof = kf == 0 ? lo[0] : (kf == 1 ? lo[1] : lo[2]), hif = kf == 0 ? hi[0] : (kf == 1 ? hi[1] : hi[2]);
    h.dist = sg * (pf - (sg > 0.f ? hif : lof)) - rad;
#pragma unroll
    for (int k = 0; k < 3; ++k) h.r[k] = fma_(-rad, h.n[k], ctr[k]);
    h.link = L.clink[i]; h.surf = surf; h.t = t2; h.sph = sph;
    h.ok = face & slope_ok & far_enough & (h.dist < c.cdist);
    return h;
}

/* Phase C helper, self-collision: capsule pair `id` = 8 * legpair + 3 * segA + segB - 1 (legpair (0,1),(0,2),(0,3),(1,2),
 * (1,3),(2,3); seg 0 = the jointless leg capsule O -> hip point (torso body), 1 = aux, 2 = foot; (0,0) skipped).
 * Closest points of the two capsule axes (Ericson, Real-Time Collision Detection 5.1.9) with the fixed squared segment
 * lengths 0.08 / 0.32 and their exact reciprocals 12.5 / 3.125; contact point = midway between the two surface points. */
HRL_DEV Hit capsule_pair(const DevCfg &c, const WaveLds &L, int id) {
    Hit h;
    h.ok = false; h.dist = 0.f; h.link = 0; h.link2 = -1; h.surf = 0; h.mu = c.mu_self; h.t = 0.f; h.sph = -1;
    h.n[0] = h.n[1] = 0.f; h.n[2] = 1.f; h.r[0] = h.r[1] = h.r[2] = 0.f;
    if (id < 0 || id >= 48) return h;
    const int pp = id >> 3, k = (id & 7) + 1, a = k / 3, b = k - 3 * a;
    const int i = pp < 3 ? 0 : (pp < 5 ? 1 : 2), j = pp < 3 ? pp + 1 : (pp < 5 ? pp - 1 : 3);
    float p1[3], q1[3], p2[3], q2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        p1[t] = a == 0 ? 0.f : (a == 1 ? L.ph[i][t] : L.pa[i][t]); q1[t] = a == 0 ? L.ph[i][t] : (a == 1 ? L.pa[i][t] : L.tip[i][t]);
        p2[t] = b == 0 ? 0.f : (b == 1 ? L.ph[j][t] : L.pa[j][t]); q2[t] = b == 0 ? L.ph[j][t] : (b == 1 ? L.pa[j][t] : L.tip[j][t]);
    }
    const float aa = a == 2 ? 0.32f : 0.08f, ia = a == 2 ? 3.125f : 12.5f, ee = b == 2 ? 0.32f : 0.08f, ie = b == 2 ? 3.125f : 12.5f;
    float d1[3], d2[3], r[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { d1[t] = q1[t] - p1[t]; d2[t] = q2[t] - p2[t]; r[t] = p1[t] - p2[t]; }
    const float f = dot3(d2, r), cc = dot3(d1, r), bb = dot3(d1, d2);
    const float denom = fma_(aa, ee, -(bb * bb));
    float sp = 0.f, tp;
    if (denom > 1e-9f) sp = clampf(fma_(bb, f, -(cc * ee)) / denom, 0.f, 1.f);
    tp = fma_(bb, sp, f) * ie;
    if (tp < 0.f) { tp = 0.f; sp = clampf(-cc * ia, 0.f, 1.f); }
    else if (tp > 1.f) { tp = 1.f; sp = clampf((bb - cc) * ia, 0.f, 1.f); }
    float c1[3], c2[3], dv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { c1[t] = fma_(d1[t], sp, p1[t]); c2[t] = fma_(d2[t], tp, p2[t]); dv[t] = c1[t] - c2[t]; }
    const float d2n = dot3(dv, dv), thr = (c.r_caps + c.r_caps) + c.cdist;
    h.ok = d2n < thr * thr;
    h.link = a | (i << 2); h.link2 = b | (j << 2); h.surf = SURF_SELF + id;
    if (h.ok) {
        const float len = sqrtf(d2n);
        h.dist = len - (c.r_caps + c.r_caps);
        if (len > 0.f) { const float il = 1.f / len; h.n[0] = dv[0] * il; h.n[1] = dv[1] * il; h.n[2] = dv[2] * il; }
#pragma unroll
        for (int t = 0; t < 3; ++t) h.r[t] = 0.5f * (c1[t] + c2[t]);
    }
    return h;
}

struct LimitHit { bool ok; float sgn, dist; };

/* Phase R1 (row map): Jacobian row and its velocity response B = M^-1 J^T.  J, bias and bounds stay in the lane's
 * registers, B goes to LDS (every row needs every B to build its row of A).  Contacts between two ant bodies get the
 * first body's part here and the second body's in phase_self_rows. */
HRL_DEV void phase_build_row(const DevCfg &c, WaveLds &L, LaneRegs &g, int lane, int nL, int nC) {
    const int nR = nL + 3 * nC;
    /* Lanes beyond the last row recompute row 0 and discard it: every lane then (re)defines all of its solver
     * registers each substep, so none of them stays live around the substep loop (a partially written register
     * would, and the 60 of them would be spilled in the register-hungry leg phases). */
    const bool active = lane < nR;
    const int row_id = active ? lane : 0;
    float B[16], bias, hi = 0.f, Jh = 0.f, Ja = 0.f, mu = 0.f;
    int frn = -1;
    if (nR == 0) { /* no rows this substep (wave-uniform): still define every register */
#pragma unroll
        for (int k = 0; k < 6; ++k) g.Jb[k] = 0.f;
        g.Jh = 0.f; g.Ja = 0.f; g.jslot = 6 | (6 << 8); g.mu = 0.f;
        g.bias = 0.f; g.fn = -1; g.lam = 0.f; g.lo = 0.f; g.hi = 0.f;
        return;
    }
    /* limit rows and contact rows only differ in their impulse (phi / joint impulse) and parameters.  Both kinds sit in one wave, so the two
     * descriptions are read side by side with indices every lane can use (a limit lane reads contact 0, a contact lane limit 0: in bounds,
     * never used) and picked by selects: one LDS round trip for the lists, one for the joint axes of the row's leg -- as two divergent branches
     * the wave ran both one after the other, each waiting for its own loads.  Per value the operations are the ones of the branches. */
    float phi[6], th, ta;
    int level, leg;
    {
        const bool is_lim = row_id < nL;
        const int li = is_lim ? row_id : 0, row = is_lim ? 0 : row_id - nL;
        const int ci = row < nC ? row : (row - nC) >> 1, which = row < nC ? 0 : 1 + ((row - nC) & 1);
        const int j = L.ljoint[li] & 7, link = L.clink[ci];
        const float sgn = L.lsign[li], ldist = L.ldist[li], cdist = L.cdist_[ci], cmu = L.cmu[ci];
        const float r[3] = {L.cr[ci][0], L.cr[ci][1], L.cr[ci][2]};
        const float d[3] = {L.cdir[which][ci][0], L.cdir[which][ci][1], L.cdir[which][ci][2]}; /* the row's direction */
        float cphi[6];
        cross3(cphi, r, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) cphi[3 + k] = d[k];
        level = is_lim ? 0 : (link & 3);
        leg = (is_lim ? (j >> 1) : (link >> 2)) & 3;
        const float jh = dot6(cphi, L.S[2 * leg]), ja = dot6(cphi, L.S[2 * leg + 1]);
#pragma unroll
        for (int k = 0; k < 6; ++k) phi[k] = is_lim ? 0.f : cphi[k];
        th = is_lim ? ((j & 1) ? 0.f : sgn) : 0.f; ta = is_lim ? ((j & 1) ? sgn : 0.f) : 0.f;
        Jh = is_lim ? th : (level >= 1 ? jh : 0.f); /* limit: J = +-e_j; contact: J = [phi | phi.S on the joints between torso and body] */
        Ja = is_lim ? ta : (level >= 2 ? ja : 0.f);
        const float blim = (ldist > 0.f ? ldist : c.erp_l * ldist) * c.inv_h, bnor = (cdist > 0.f ? cdist : c.erp_c * cdist) * c.inv_h;
        const bool normal = !is_lim & (which == 0), fric = !is_lim & (which != 0);
        bias = is_lim ? blim : (normal ? bnor : 0.f);
        hi = is_lim ? c.limp_max : (normal ? 1e30f : 0.f);
        frn = fric ? nL + ci : -1;
        mu = fric ? cmu : 0.f;
    }
    response(L, phi, level, leg, th, ta, B);
    if (active) {
#pragma unroll
        for (int k = 0; k < 16; ++k) L.Bt[lane][k] = B[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) g.Jb[k] = phi[k]; /* zero for limit rows */
    g.Jh = Jh; g.Ja = Ja; g.jslot = (6 + 2 * leg) | ((6 + 2 * leg) << 8); g.mu = mu;
    g.bias = bias; g.fn = active ? frn : -1; g.lam = 0.f; g.lo = 0.f; g.hi = frn >= 0 ? 0.f : hi;
}

/* Phase R1b (row map; only substeps that kept a self contact run it): the rows of a contact between two ant bodies A (leg)
 * and B (leg2 > leg) are J = J_A - J_B -- both bodies move with the torso, so its part cancels exactly and what is left are
 * the hip / ankle entries of the two legs -- and B = B_A - B_B: phase R1 left the A parts, this phase subtracts the second
 * impulse response in the row's own LDS record, zeroes the torso part of J and leaves the second leg's entries in L.J2. */
HRL_DEV void phase_self_rows(const DevCfg &c, WaveLds &L, LaneRegs &g, int lane, int nL, int nC) {
    const int nR = nL + 3 * nC;
    if (lane >= MAXR) return;
    float z0 = 0.f;
    HRL_PIN_VGPR(z0); /* materialised here, not hoisted out of the substep loop as a live register */
    L.J2[lane][0] = z0; L.J2[lane][1] = z0;
    if (lane < nL || lane >= nR) return;
    const int row = lane - nL;
    const int ci = row < nC ? row : (row - nC) >> 1, which = row < nC ? 0 : 1 + ((row - nC) & 1);
    const int link2 = L.clink2[ci];
    if (link2 < 0) return;
    const int level2 = link2 & 3, leg2 = link2 >> 2;
    const float r[3] = {L.cr[ci][0], L.cr[ci][1], L.cr[ci][2]};
    const float d[3] = {L.cdir[which][ci][0], L.cdir[which][ci][1], L.cdir[which][ci][2]};
    float phi[6], B2[16];
    cross3(phi, r, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) phi[3 + k] = d[k];
    response(L, phi, level2, leg2, 0.f, 0.f, B2);
#pragma unroll
    for (int k = 0; k < 14; ++k) L.Bt[lane][k] = L.Bt[lane][k] - B2[k];
    const float jh2 = dot6(phi, L.S[2 * leg2]), ja2 = dot6(phi, L.S[2 * leg2 + 1]);
    L.J2[lane][0] = level2 >= 1 ? -jh2 : 0.f; L.J2[lane][1] = level2 >= 2 ? -ja2 : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) g.Jb[k] = 0.f;
    g.jslot = (g.jslot & 0xff) | ((6 + 2 * leg2) << 8);
    (void)c;
}

/* Phase R2 (row map): the row's line of C = I - D^-1 A with A = J M^-1 J^T (A[i][r] = J_i . B_r, D = diag A), and the
 * initial impulse candidate c_i = -(J_i . u* + bias_i) / A_ii.  Sequential fma chains over the 16 dof slots with their
 * exact zeros left out (the torso part, then the row's two joint slots in order); B_r and u* are LDS reads.
 * The line is kept in two register arrays, split where the sweep order changes kind: An[r] for the limit and normal
 * rows r < nB, Af[k] for the friction rows nB + k.  Both are indexed by template parameters (pack expansion) so that
 * they stay in registers: a runtime index would push them out into scratch memory.
 * `one(r)` is 1 on lane r and 0 elsewhere (the unit diagonal), supplied by the executor. */
struct J2pair { float h, a; }; /* self-contact rows: the second leg's entries of the lane's row (from WaveLds::J2) */
template <bool SELF, bool FREE = false> /* FREE: the row of a single free body (the point bot): no joint entries */
HRL_DEV float row_dot(const LaneRegs &g, const float *Brow, const J2pair &j2) {
    float a = g.Jb[0] * Brow[0];
#pragma unroll
    for (int d = 1; d < 6; ++d) a = fma_(g.Jb[d], Brow[d], a);
    /* the joint slots of a free body hold exact zeros on both sides: fma(+0, +0, a) is a + (+0) -- which turns a -0 into +0 and changes nothing
       else -- and a second one is the identity; written as that addition, the row needs neither the two loads nor Jh / Ja / jslot in registers */
    if (FREE) return a + 0.f;
    const int s1 = g.jslot & 0xff;
    a = fma_(g.Jh, Brow[s1], a);
    a = fma_(g.Ja, Brow[s1 + 1], a);
    if (SELF) { /* substeps with a self contact: the entries of the second body's leg, next in slot order (leg2 > leg) */
        const int s2 = g.jslot >> 8;
        a = fma_(j2.h, Brow[s2], a);
        a = fma_(j2.a, Brow[s2 + 1], a);
    }
    return a;
}
/* four rows' dot products at once, term by term: the same chains as four row_dot calls, written interleaved so that the four
 * independent fma chains issue back to back (column after column they ran as four dependent chains of 8 - 10 instructions) */
template <bool SELF>
HRL_DEV void row_dot4(const LaneRegs &g, const float *B0, const float *B1, const float *B2, const float *B3, const J2pair &j2, float *a) {
    const float *Br[4] = {B0, B1, B2, B3};
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = g.Jb[0] * Br[i][0];
#pragma unroll
    for (int d = 1; d < 6; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fma_(g.Jb[d], Br[i][d], a[i]);
    const int s1 = g.jslot & 0xff;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = fma_(g.Jh, Br[i][s1], a[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = fma_(g.Ja, Br[i][s1 + 1], a[i]);
    if (SELF) {
        const int s2 = g.jslot >> 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fma_(j2.h, Br[i][s2], a[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fma_(j2.a, Br[i][s2 + 1], a[i]);
    }
}
/* columns 4G..4G+3 of a block behind one wave-uniform test (flat sequence of groups, no nesting).  Columns past the
 * block's end inside its last group are computed from whatever LDS holds and never read by the sweeps. */
template <bool SELF, int G, bool WIDE, class One>
HRL_DEV void build_An_group(const WaveLds &L, LaneRegs &g, int nB, float ninvd, One one, const J2pair &j2) {
    constexpr bool FREE = !WIDE; /* the narrow build is the point bot's */
    if (4 * G < nB) {
        if constexpr (WIDE) {
            float a[4];
            row_dot4<SELF>(g, L.Bt[4 * G], L.Bt[4 * G + 1], L.Bt[4 * G + 2], L.Bt[4 * G + 3], j2, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) g.An[4 * G + i] = fma_(ninvd, a[i], one(4 * G + i));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) g.An[4 * G + i] = fma_(ninvd, row_dot<SELF, FREE>(g, L.Bt[4 * G + i], j2), one(4 * G + i));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.An[4 * G + i] = 0.f;
    }
}
template <bool SELF, int G, bool WIDE, class One>
HRL_DEV void build_Af_group(const WaveLds &L, LaneRegs &g, int nB, int nF, float ninvd, One one, const J2pair &j2) {
    constexpr bool FREE = !WIDE;
    if (4 * G < nF) {
        if constexpr (WIDE) {
            float a[4];
            row_dot4<SELF>(g, L.Bt[nB + 4 * G], L.Bt[nB + 4 * G + 1], L.Bt[nB + 4 * G + 2], L.Bt[nB + 4 * G + 3], j2, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) g.Af[4 * G + i] = fma_(ninvd, a[i], one(nB + 4 * G + i));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) g.Af[4 * G + i] = fma_(ninvd, row_dot<SELF, FREE>(g, L.Bt[nB + 4 * G + i], j2), one(nB + 4 * G + i));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.Af[4 * G + i] = 0.f;
    }
}
template <bool SELF, class One, int... Gn, int... Gf>
HRL_DEV void build_A_blocks(const WaveLds &L, LaneRegs &g, int nB, int nF, float ninvd, One one, const J2pair &j2, std::integer_sequence<int, Gn...>, std::integer_sequence<int, Gf...>) {
    constexpr bool WIDE = sizeof...(Gn) * 4 == MAXB; /* the ant kernels have the registers for four chains side by side; the point bot's (128, at its cap) has not */
    (build_An_group<SELF, Gn, WIDE>(L, g, nB, ninvd, one, j2), ...);
    (build_Af_group<SELF, Gf, WIDE>(L, g, nB, nF, ninvd, one, j2), ...);
}
template <bool SELF, int MB, class One> /* MB = most bounded rows the caller can have (a multiple of four): columns past it are not built */
HRL_DEV void phase_build_A(const DevCfg &c, const WaveLds &L, LaneRegs &g, int lane, int nL, int nB, int nF, One one) {
    J2pair j2{0.f, 0.f};
    if (SELF && lane < MAXR) { j2.h = L.J2[lane][0]; j2.a = L.J2[lane][1]; }
    constexpr bool FREE = MB != MAXB; /* the point bot's solver (no limit rows: at most MAXC bounded rows) */
    if (c.restitution > 0.f) { /* hrl_model.restitution (default 0: skipped, wave-uniform): a normal row whose bodies approach faster than the threshold
                                  (J . u at the start of the substep) asks for the separating velocity restitution * (approach speed) */
        const float vn = row_dot<SELF, FREE>(g, L.u, j2);
        const bool normal = (lane >= nL) & (lane < nB);
        g.bias = (normal & (vn < -c.rest_thr)) ? fma_(c.restitution, vn, g.bias) : g.bias;
    }
    const float invd = 1.f / row_dot<SELF, FREE>(g, L.Bt[lane < nB + nF ? lane : 0], j2); /* 1 / A_ii; idle lanes carry row 0's registers */
    build_A_blocks<SELF>(L, g, nB, nF, -invd, one, j2, std::make_integer_sequence<int, MB / 4>{}, std::make_integer_sequence<int, MAXF / 4>{});
    g.c = -(invd * (row_dot<SELF, FREE>(g, L.ustar, j2) + g.bias));
}

/* Phase I (dof map): integrate positions; the joint rates were clamped by the caller.  Lane k < 16 produces element k
 * of the new q: the quaternion (exponential map, every lane computes it) or one fma for a position / joint angle. */
HRL_DEV void phase_integrate(const DevCfg &c, WaveLds &L, const float *q, float *qn, int lane) {
    const float h = c.h, hh = 0.5f * h;
    const float u0 = L.u[0], u1 = L.u[1], u2 = L.u[2];
    /* quaternion increment exp(h omega / 2) = (omega (h/2) sinc(x), cos(x)), x = |omega| h / 2, through z = x^2: while x <= 0.5
     * (|omega| <= 242 rad/s at the default h) the Taylor polynomials of sinc and cos in z up to z^4 are exact to 3e-10 and need
     * neither the square root nor the division; beyond that the closed form is used (wave-uniform branch). */
    const float ww = fma_(u2, u2, fma_(u1, u1, u0 * u0)), z = ww * (hh * hh);
    const bool small = z <= 0.25f;
    float dq[4];
    if (small) {
        float ps = fma_(z, 2.75573192239858906e-6f, -1.98412698412698413e-4f);
        ps = fma_(ps, z, 8.33333333333333333e-3f); ps = fma_(ps, z, -1.66666666666666667e-1f); ps = fma_(ps, z, 1.0f);
        float pc = fma_(z, 2.48015873015873016e-5f, -1.38888888888888889e-3f);
        pc = fma_(pc, z, 4.16666666666666667e-2f); pc = fma_(pc, z, -0.5f); pc = fma_(pc, z, 1.0f);
        const float sc = hh * ps;
        dq[0] = u0 * sc; dq[1] = u1 * sc; dq[2] = u2 * sc; dq[3] = pc;
    } else {
        const float wn = sqrtf(ww);
        float sh_, ch_;
        sincos_spec(hh * wn, &sh_, &ch_);
        const float sc = sh_ / wn;
        dq[0] = u0 * sc; dq[1] = u1 * sc; dq[2] = u2 * sc; dq[3] = ch_;
    }
    float x = q[3], y = q[4], zq = q[5], w = q[6];
    float nx = fma_(-dq[2], y, fma_(dq[1], zq, fma_(dq[0], w, dq[3] * x)));
    float ny = fma_(dq[2], x, fma_(dq[1], w, fma_(-dq[0], zq, dq[3] * y)));
    float nz = fma_(dq[2], w, fma_(-dq[1], x, fma_(dq[0], y, dq[3] * zq)));
    float nw = fma_(-dq[2], zq, fma_(-dq[1], y, fma_(-dq[0], x, dq[3] * w)));
    /* renormalisation: the product of two unit quaternions is off unit length by rounding only, so one Newton step of
     * 1/sqrt at 1, (3 - n2) / 2, is exact to (n2 - 1)^2 ~ 1e-14; the closed-form branch keeps the exact 1/sqrt */
    const float n2 = fma_(nx, nx, ny * ny) + fma_(nz, nz, nw * nw);
    const float inv = small ? fma_(-0.5f, n2, 1.5f) : 1.f / sqrtf(n2);
    const int k = lane & 15;
    /* position k < 3 advances with the linear velocity u[3 + k], joint angle q[7 + j] with the joint rate u[6 + j] */
    const int ui = k < 3 ? k + 3 : (k >= 7 && k < 15 ? k - 1 : 0);
    float mine = fma_(h, L.u[ui], q[k]);
    mine = (k == 3) ? nx * inv : mine;
    mine = (k == 4) ? ny * inv : mine;
    mine = (k == 5) ? nz * inv : mine;
    mine = (k == 6) ? nw * inv : mine;
    mine = (k == 15) ? 0.f : mine;
    if (lane < 16) qn[lane] = mine;
}

/* Projected Gauss-Seidel in ROW SPACE: lane i owns solver row i -- its unclamped impulse candidate
 * c_i = lam_i - w_i / A_ii, impulse, bounds and its row of C = I - D^-1 A -- all in registers.  Solving row r: every lane
 * clamps the candidate of its own row, lane r keeps the result, its change is broadcast with a lane read and every lane
 * applies c_i += C[i][r] * dl (one executor primitive, each_row): four dependent instructions per row.
 * No LDS and no cross-lane reduction on the solver's dependent chain.  Rows run in order (limits, normals, then the
 * friction pairs); the friction rows take their bounds +-mu * (normal impulse) from their normal's lane once per
 * sweep, after the last normal row, which is exactly when a row-by-row update would have last changed them.
 * c.iters sweeps; the velocity is reconstructed once at the end from the impulses. */
HRL_DEV F2b pgs_candidate(const LaneRegs &g) {
    F2b o;
    o.ln = med3_spec(g.c, g.lo, g.hi);
    o.dl = o.ln - g.lam;
    return o;
}
template <int R, class X>
HRL_DEV bool pgs_row_bounded(X &x, int nB) { /* limit or normal row R (compile-time index: An[R] is a register) */
    if (R >= nB) return false; /* wave-uniform: ends the block (the fold below short-circuits) */
    x.each_row(R, [&](int lane) { return pgs_candidate(x.reg(lane)); },
               [&](int lane, float dl) { LaneRegs &g = x.reg(lane); g.c = fma_(g.An[R], dl, g.c); });
    return true;
}
template <int K, class X>
HRL_DEV bool pgs_row_friction(X &x, int nB, int nF) { /* friction rows nB + K, nB + K + 1 of one contact (K even): nF = 2 nC, so one test serves the pair */
    if (K >= nF) return false;
    x.each_row(nB + K, [&](int lane) { return pgs_candidate(x.reg(lane)); },
               [&](int lane, float dl) { LaneRegs &g = x.reg(lane); g.c = fma_(g.Af[K], dl, g.c); });
    x.each_row(nB + K + 1, [&](int lane) { return pgs_candidate(x.reg(lane)); },
               [&](int lane, float dl) { LaneRegs &g = x.reg(lane); g.c = fma_(g.Af[K + 1], dl, g.c); });
    return true;
}
template <class X, int... Rs, int... Ks>
HRL_DEV void pgs_sweep(X &x, int nB, int nF, std::integer_sequence<int, Rs...>, std::integer_sequence<int, Ks...>) {
    /* rows in order as flat sequences with one forward exit each (no nesting: nested wave-uniform ifs cost an SGPR pair each) */
    (void)(pgs_row_bounded<Rs>(x, nB) && ...);
    if (nF <= 0) return;
    x.each_shuffle([&](int lane) { return x.reg(lane).lam; },
                   [&](int lane) { const int fn = x.reg(lane).fn; return fn >= 0 ? fn : lane; },
                   [&](int lane, float ln) {
                       LaneRegs &g = x.reg(lane);
                       if (g.fn >= 0) { g.hi = g.mu * ln; g.lo = -g.hi; }
                   });
    (void)(pgs_row_friction<2 * Ks>(x, nB, nF) && ...);
}
template <int MB = MAXB, class X>
HRL_DEV void pgs_solve(X &x, const DevCfg &c, int nL, int nC, bool ant, bool self) {
    static_assert(MB % 4 == 0 && MB <= MAXB, "bounded rows are built in groups of four");
    WaveLds &L = x.lds();
    const int nB = ant ? nL + nC : nC, nF = 2 * nC, nR = nB + nF;
    if (nR <= 0) return;
    const int nLim = ant ? nL : 0;
    if (self) x.each([&](int lane) { phase_build_A<true, MB>(c, L, x.reg(lane), lane, nLim, nB, nF, [&](int r) { return x.lane_one(lane, r); }); });
    else x.each([&](int lane) { phase_build_A<false, MB>(c, L, x.reg(lane), lane, nLim, nB, nF, [&](int r) { return x.lane_one(lane, r); }); });
    if (MB != MAXB) /* the point bot's kernel sits at its register cap: lines spilled around the build are reloaded HERE, not inside the sweeps */
        x.each([&](int lane) {
            LaneRegs &g = x.reg(lane);
            (void)g;
#pragma unroll
            for (int k = 0; k < MB; ++k) HRL_PIN_VGPR(g.An[k]);
#pragma unroll
            for (int k = 0; k < MAXF; ++k) HRL_PIN_VGPR(g.Af[k]);
        });
    x.stamp(8);
    const int iters = c.iters;
    for (int it = 0; it < iters; ++it) {
        int nb = nB, nf = nF;
        x.refresh();
        x.refresh_uniform(nb); /* keeps the 44 row-count tests inside the sweep as scalar compares (hoisted out of */
        x.refresh_uniform(nf); /* the loop they become 44 live lane-mask pairs, most of them spilled)             */
        pgs_sweep(x, nb, nf, std::make_integer_sequence<int, MB>{}, std::make_integer_sequence<int, MAXF / 2>{});
    }
    x.stamp(9);
    x.each([&](int lane) { if (lane < nR) L.lamf[lane] = x.reg(lane).lam; });
    x.each([&](int lane) { /* dof map: u = u* + sum_r B_r * lambda_r, rows in order; blocks of 8, then 4, 2, 1 so that the loads of a block
                              are in flight together (a one-row remainder loop paid the LDS latency up to seven times) */
        float v = L.ustar[lane & 15];
        const int k = lane & 15;
        int r = 0;
        for (; r + 8 <= nR; r += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v = fma_(L.Bt[r + i][k], L.lamf[r + i], v);
        }
        if (nR & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v = fma_(L.Bt[r + i][k], L.lamf[r + i], v);
            r += 4;
        }
        if (nR & 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) v = fma_(L.Bt[r + i][k], L.lamf[r + i], v);
            r += 2;
        }
        if (nR & 1) v = fma_(L.Bt[r][k], L.lamf[r], v);
        x.reg(lane).ud = v;
    });
}

/* ------------------------------------------------------------------------------------------------ the ant substep
 * A substep is two blocks separated by workgroup barriers.
 * Block 1, two things side by side:
 *   - the GROUP block (ant_group_block) -- the lane-sparse dynamics K1, K2, S, B, V: nine bodies / 14 dofs of distinct work per
 *     env -- executed ONCE for all the envs of a workgroup by its leader wave, 16 lanes per env (lane >> 4 = env of the group,
 *     lane & 15 = body / dof): one instruction stream serves four envs instead of four streams serving one each;
 *   - contacts and limit rows (ant_contact_duty) -- functions of the pose alone -- on the waves that would otherwise wait for
 *     the leader.
 * Block 2 (ant_env_block), on every env's own wave: rows, A, sweeps, velocity reconstruction (up to 44 rows of distinct
 * work), then the position integration.
 * Inside a block the phases of a wave are separated by wave-level LDS synchronisation only.
 * With a group of one (executor G = 1) the only wave does all of it in order, the four 16-lane slices of the group block
 * computing the same values on the same record: the one-wave-per-env form.
 * qi = index of the position buffer the substep works on; its integration writes q[qi ^ 1]. */
template <class X>
HRL_DEV void ant_group_block(X &x, const DevCfg &c, int qi) {
    x.refresh(); /* keep lane-derived values local to the substep (see GpuExec::refresh) */
    x.stamp(0);
    /* the phases index bodies as lane >> 2 (four lanes per body in the one-env form): (lane & 15) << 2 gives body = lane & 15 */
    x.leader([&](int lane) { WaveLds &L = x.lds(lane >> 4); phase_kin_ankle(c, L, x.reg(lane), L.q[qi], (lane & 15) << 2); });
    x.stamp(1);
    x.leader([&](int lane) { WaveLds &L = x.lds(lane >> 4); phase_hip(c, L, x.reg(lane), (lane & 15) << 2); });
    x.stamp(2);
    x.leader([&](int lane) { WaveLds &L = x.lds(lane >> 4); phase_leg_sum(L, lane & 15); phase_leg_sum(L, (lane & 15) + 16); });
    x.leader([&](int lane) { WaveLds &L = x.lds(lane >> 4); phase_base(L, x.reg(lane), (lane & 15) << 2); });
    x.stamp(3);
    x.leader([&](int lane) { WaveLds &L = x.lds(lane >> 4); const float v = phase_forward_vel(c, L, lane & 15); L.ustar[lane & 15] = v; });
    x.stamp(4);
}

/* One kept contact into the record (contact index i < MAXC). */
HRL_DEV void store_contact(WaveLds &L, int i, const Hit &h, bool up, bool on_box = false) { /* on_box: a pass against the maze box / the cubes (its contacts' axis parameter and shape are read by second_support) */
    if (i < MAXC) {
#pragma unroll
        for (int k = 0; k < 3; ++k) L.cr[i][k] = h.r[k];
        store_contact_frame(L, i, h.n, up);
        L.cdist_[i] = h.dist; L.clink[i] = h.link; L.clink2[i] = h.link2; L.csurf[i] = h.surf; L.cmu[i] = h.mu;
        if (on_box) { L.ct[i] = h.t; L.csph[i] = h.sph; }
    }
}

/* Contacts and limit rows of ONE env's pose q[qi] into its record L -- executed by whichever wave has the time: it needs the pose
 * only (the leg points come from a POS_ONLY kinematics pass of its own), so it runs WHILE the leader wave works through the group
 * block, on the waves that would otherwise wait at the barrier.  Results: the contact / limit lists and L.nC / nL / nS. */
template <class X>
HRL_DEV void ant_contacts(X &x, const DevCfg &c, WaveLds &L, int qi, bool items_on) {
    const float *q = L.q[qi];
    const int cap = c.max_contacts; /* contacts kept per substep (hrl_model.max_contacts <= MAXC) */
    x.refresh();
    x.each([&](int lane) { phase_kin_ankle<true>(c, L, x.reg(lane), q, lane); });
    /* contacts in surface-major, sphere-minor order (ballot ranks follow lane order), at most MAXC kept:
     * pass 0 = ground (13 lanes), pass 1 = all lateral half-spaces (13 lanes each), pass 2 = world boxes, then the item
     * cubes near the robot (up to four cubes per pass, 13 lanes each), then the capsule pairs of different legs */
    int nC = 0, nS = 0;
#ifdef HRL_WGTIME
    int n_cube_passes = 0;
#endif
    /* Broad phase (wave-uniform): every contact sphere lies within 1.25 m of the torso centre (hip 0.283 + aux 0.283 +
     * foot 0.566 + radius 0.08 + contact_dist), so a lateral surface farther than that from the torso cannot produce a
     * contact and its pass is skipped.  Exactly the same contact list as testing every pair. */
    const float reach = 0.2f * 1.41421356f + c.L1 + c.L2 + c.r_caps + c.cdist + 0.02f;
    /* no short-circuit operators in these wave-uniform tests: `a && b` on LDS operands compiles to one load -> wait -> branch per
     * term, a chain of dependent round trips (8 of them in the joint-range test below cost 0.35 us per substep) */
    bool near_plane = false, near_box = false;
#pragma unroll
    for (int f = 0; f < 4; ++f)
        near_plane = near_plane | ((f < c.n_planes) & ((c.plane_n[f][0] * q[0] + c.plane_n[f][1] * q[1] + c.plane_n[f][2] * q[2]) - c.plane_d[f] < reach));
    if (c.n_boxes > 0) {
        float d2 = 0.f;
        for (int k = 0; k < 3; ++k) { const float cp = clampf(q[k], c.box_lo[k], c.box_hi[k]); d2 += (q[k] - cp) * (q[k] - cp); }
        near_box = d2 < reach * reach;
    }
    auto keep = [&](int base, bool up = false, bool on_box = false) {
        return [&L, base, up, on_box](int, int rank, const Hit &h) { store_contact(L, base + rank, h, up, on_box); };
    };
    bool any_box = false; /* a pass against the box / a cube kept a contact: only then is there anything for second_support to look at */
    for (int pass = 0; pass < 3; ++pass) {
        const int nsurf = pass == 0 ? 1 : (pass == 1 ? c.n_planes : c.n_boxes);
        if (nsurf == 0) continue;
        if ((pass == 1 && !x.uniform(near_plane)) || (pass == 2 && !x.uniform(near_box))) continue;
        const int f0 = pass == 0 ? 0 : (pass == 1 ? 1 : 1 + c.n_planes);
        int cnt = x.each_compact(
            [&](int lane) {
                const int fi = lane / 13, sph = lane - 13 * fi;
                return sphere_vs_surface(c, L, q, fi < nsurf ? sph : -1, f0 + fi, -1);
            },
            keep(nC, pass == 0, pass == 2),
            [&](int lane, const Hit &h) { if (pass == 0 && lane < 16) L.gtouch[lane] = h.ok ? 1 : 0; });
        nC += cnt;
        if (nC > cap) nC = cap;
        any_box = any_box | ((pass == 2) & (cnt > 0));
    }
    if (items_on) { /* food / poison cubes: lane = item decides whether its cube is within reach of any sphere (the cube's
                       half extent more than the planes' bound, per axis), then the near cubes are tested four at a time */
        const float R = reach + ITEM_HALF;
        unsigned long long near = x.each_ballot([&](int lane) { /* lane = item (at most 64 of them) */
            return (lane < c.n_food + c.n_poison) & (fabsf(q[0] - L.items[2 * lane]) < R) & (fabsf(q[1] - L.items[2 * lane + 1]) < R);
        });
        while (near) {
#ifdef HRL_WGTIME
            ++n_cube_passes;
#endif
            int it[4], n_it = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                it[k] = -1;
                if (near) { it[k] = (int)__builtin_ctzll(near); near &= near - 1; ++n_it; }
            }
            const unsigned long long maybe = x.each_ballot([&](int lane) { /* see ant_contacts_group */
                const int slot = lane / 13, sph = lane - 13 * slot;
                const int item = slot == 0 ? it[0] : (slot == 1 ? it[1] : (slot == 2 ? it[2] : (slot == 3 ? it[3] : -1)));
                return shape_near_item(c, L, q, sph, item);
            });
            if (!maybe) continue;
            int cnt = x.each_compact(
                [&](int lane) {
                    const int slot = lane / 13, sph = lane - 13 * slot;
                    const int item = slot == 0 ? it[0] : (slot == 1 ? it[1] : (slot == 2 ? it[2] : (slot == 3 ? it[3] : -1)));
                    return sphere_vs_surface(c, L, q, item >= 0 ? sph : -1, 0, item);
                },
                keep(nC, false, true), [&](int, const Hit &) {});
            nC += cnt;
            if (nC > cap) nC = cap;
            any_box = any_box | (cnt > 0);
        }
    }
#ifndef HRL_NO_SECOND /* (A/B builds of tools/variants.py define it) */
    if (x.uniform(any_box) && x.each_ballot([&](int lane) { int kf; return (lane < nC) & second_worth_a_look(c, L, lane < MAXC ? lane : 0, &kf); })) {
        /* second support points of the capsules that lie flat on a face of the box / a cube: one pass over the kept contacts, lane = contact */
        int cnt = x.each_compact([&](int lane) { return second_support(c, L, q, lane < nC ? lane : -1); }, keep(nC, false, true), [&](int, const Hit &) {});
        nC += cnt;
        if (nC > cap) nC = cap;
    }
#endif
    if (c.self_collision) { /* Seen from above in the torso frame, leg l is the jointless capsule O -> hip point followed by the aux
        and foot capsules, which both lie in the vertical plane through the hip point at 45 + 90 l degrees + hip angle: with
        |ankle angle| <= 2 rad the foot folds back by at most 0.566 cos(2) = 0.24 m < the aux length, so everything past the
        hip point projects onto the ray from it.  With every |hip angle| <= 0.75 rad those rays keep >= 0.2 m from the
        coordinate axes (hence from the other legs' jointless capsules) and >= 0.4 m from one another, so no two capsule
        axes of different legs come within 2 r + contact_dist < 0.2 m: the pair test is skipped -- the same contact list
        as testing all 48 pairs (the joints' own limits are +-0.698 and +-1.745 rad). */
        const bool thin = (c.r_caps + c.r_caps) + c.cdist < 0.2f;
        const bool spread = x.each_ballot([&](int lane) { /* lane = joint: outside the safe range? */
            return (lane < NJ) & !(thin & (fabsf(q[7 + (lane & 7)]) <= ((lane & 1) ? 2.0f : 0.75f)));
        }) == 0;
        if (!x.uniform(spread)) {
            int cnt = x.each_compact([&](int lane) { return capsule_pair(c, L, lane < 48 ? lane : -1); }, keep(nC), [&](int, const Hit &) {});
            nS = nC + cnt > cap ? cap - nC : cnt; /* self contacts among the kept ones: their rows take the two-body path */
            nC += nS;
        }
    }
    x.stamp(5);
    x.each([&](int lane) {
        if (lane == 0) {
            L.nC = nC; L.nS = nS;
#ifdef HRL_WGTIME
            L.dbg_rows += (3 * nC) | (n_cube_passes << 16) | ((nS > 0 ? 1 : 0) << 24);
#endif
        }
    });
}

/* Limit rows of ONE env's pose into its record (L.nL): like ant_contacts a function of the pose alone, run by whichever wave has
 * the time. */
template <class X>
HRL_DEV void ant_limits(X &x, const DevCfg &c, WaveLds &L, int qi) {
    const float *q = L.q[qi];
    x.refresh();
    /* joint limits (lane = joint) */
    int nL = x.each_compact(
        [&](int lane) {
            LimitHit r; r.ok = false; r.sgn = 0.f; r.dist = 0.f;
            if (lane < NJ) {
                float dlo = q[7 + lane] - L.jlim[0][lane], dhi = L.jlim[1][lane] - q[7 + lane];
                if (dlo < c.lmargin) { r.ok = true; r.sgn = 1.f; r.dist = dlo; }
                else if (dhi < c.lmargin) { r.ok = true; r.sgn = -1.f; r.dist = dhi; }
            }
            return r;
        },
        [&](int lane, int rank, const LimitHit &r) { L.ljoint[rank] = lane; L.lsign[rank] = r.sgn; L.ldist[rank] = r.dist; },
        [&](int, const LimitHit &) {});
    x.stamp(6);
    x.each([&](int lane) {
        if (lane == 0) {
            L.nL = nL;
#ifdef HRL_WGTIME
            L.dbg_rows += nL;
#endif
        }
    });
}

/* Contacts of ALL FOUR envs of a group by ONE wave, 16 lanes per env (lane >> 4 = env of the group, lane & 15 = sphere / item / joint):
 * the passes every substep runs -- leg points, the ground pass (13 spheres), the near-cube ballot and one cube per env and pass -- are
 * one instruction stream for the four envs instead of four streams (the group's vector-instruction count, which is what four workgroups per
 * CU contend for, falls by the contact phase of three envs); the passes that only envs near a wall / the maze box / with legs out of their
 * safe range need run env by env with all 64 lanes, as in ant_contacts.  Per env the candidate order is unchanged: ground, walls, box,
 * cubes in slot order -- each sphere-minor -- then the capsule pairs.  Results per record: the contact list, nC, nS, gtouch. */
template <class X>
HRL_DEV void ant_contacts_group(X &x, const DevCfg &c, int qi, bool items_on) {
    x.refresh();
    const int cap = c.max_contacts;
    bool any_box = false; /* a pass against the maze box / a cube ran (rare): only then does second_support have anything to look at */
    x.each([&](int lane) { WaveLds &L = x.lds(lane >> 4); phase_kin_ankle<true>(c, L, x.reg(lane), L.q[qi], (lane & 15) << 2); });
    const float reach = 0.2f * 1.41421356f + c.L1 + c.L2 + c.r_caps + c.cdist + 0.02f;
    /* ground pass of the four envs; its count starts the env's list */
    x.each_compact16(
        [&](int lane) { WaveLds &L = x.lds(lane >> 4); const int s = lane & 15; return sphere_vs_surface(c, L, L.q[qi], s < 13 ? s : -1, 0, -1); },
        [&](int lane, int rank, const Hit &h) { store_contact(x.lds(lane >> 4), rank, h, true); },
        [&](int lane, const Hit &h, int count) {
            WaveLds &L = x.lds(lane >> 4);
            L.gtouch[lane & 15] = h.ok ? 1 : 0;
            if ((lane & 15) == 0) { L.nC = count > cap ? cap : count; L.nS = 0; }
        });
    /* lateral surfaces and the maze box: the envs within reach of one, one after the other (wave-uniform loop, all 64 lanes per env) */
    if (c.n_planes > 0 || c.n_boxes > 0) {
        const unsigned long long need = x.each_ballot([&](int lane) {
            const float *q = x.lds(lane >> 4).q[qi];
            bool near = false;
#pragma unroll
            for (int f = 0; f < 4; ++f)
                near = near | ((f < c.n_planes) & ((c.plane_n[f][0] * q[0] + c.plane_n[f][1] * q[1] + c.plane_n[f][2] * q[2]) - c.plane_d[f] < reach));
            if (c.n_boxes > 0) {
                float d2 = 0.f;
                for (int k = 0; k < 3; ++k) { const float cp = clampf(q[k], c.box_lo[k], c.box_hi[k]); d2 += (q[k] - cp) * (q[k] - cp); }
                near = near | (d2 < reach * reach);
            }
            return ((lane & 15) == 0) & near;
        });
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            if (!((need >> (16 * e)) & 1ull)) continue;
            WaveLds &L = x.lds(e);
            const float *q = L.q[qi];
            int nC = x.uniform(L.nC);
            for (int pass = 1; pass < 3; ++pass) { /* the pass tests every surface of its kind: a surface out of reach yields no contact */
                const int nsurf = pass == 1 ? c.n_planes : c.n_boxes;
                if (nsurf == 0) continue;
                const int f0 = pass == 1 ? 1 : 1 + c.n_planes;
                int cnt = x.each_compact(
                    [&](int lane) { const int fi = lane / 13, sph = lane - 13 * fi; return sphere_vs_surface(c, L, q, fi < nsurf ? sph : -1, f0 + fi, -1); },
                    [&L, nC, pass](int, int rank, const Hit &h) { store_contact(L, nC + rank, h, false, pass == 2); },
                    [&](int, const Hit &) {});
                nC += cnt;
                if (nC > cap) nC = cap;
                any_box = any_box | ((pass == 2) & (cnt > 0));
            }
            x.each([&](int lane) { if (lane == 0) L.nC = nC; });
        }
    }
    if (items_on) { /* lane = (env, item): cubes within reach of any sphere of their env's ant; then one near cube per env and pass.
                       More than 16 items: slice after slice of 16 (one trip for the default configs), which keeps the slot order */
        const float R = reach + ITEM_HALF;
        const int n_items = c.n_food + c.n_poison;
#pragma unroll 1
        for (int ib = 0; ib < n_items; ib += 16) {
        unsigned long long near = x.each_ballot([&](int lane) {
            const WaveLds &L = x.lds(lane >> 4);
            const float *q = L.q[qi];
            const int it = ib + (lane & 15);
            return (it < n_items) & (fabsf(q[0] - L.items[2 * it]) < R) & (fabsf(q[1] - L.items[2 * it + 1]) < R);
        });
        while (near) {
            const unsigned long long cur = near;
            /* can any capsule of any env reach its env's cube of this pass?  (mostly not: a cube within arm's length is usually a metre from the feet) */
            const unsigned long long maybe = x.each_ballot([&](int lane) {
                const WaveLds &L = x.lds(lane >> 4);
                const unsigned sub = (unsigned)(cur >> (lane & 48)) & 0xffffu;
                return shape_near_item(c, L, L.q[qi], (lane & 15) < 13 ? (lane & 15) : -1, sub ? ib + __builtin_ctz(sub) : -1);
            });
            if (maybe) {
            x.each_compact16(
                [&](int lane) {
                    WaveLds &L = x.lds(lane >> 4);
                    const unsigned sub = (unsigned)(cur >> (lane & 48)) & 0xffffu;
                    const int s = lane & 15, item = sub ? ib + __builtin_ctz(sub) : -1;
                    return sphere_vs_surface(c, L, L.q[qi], (item >= 0 && s < 13) ? s : -1, 0, item);
                },
                [&](int lane, int rank, const Hit &h) { WaveLds &L = x.lds(lane >> 4); store_contact(L, L.nC + rank, h, false, true); },
                [&](int lane, const Hit &, int count) { if ((lane & 15) == 0) x.lds(lane >> 4).ncnt = count; });
            x.each([&](int lane) {
                if ((lane & 15) == 0) { WaveLds &L = x.lds(lane >> 4); const int n = L.nC + L.ncnt; L.nC = n > cap ? cap : n; }
            });
            any_box = true;
            }
            unsigned long long low = 0; /* every env's lowest set bit is done */
#pragma unroll
            for (int e = 0; e < 4; ++e) { const unsigned sub = (unsigned)(near >> (16 * e)) & 0xffffu; low |= (unsigned long long)(sub & (0u - sub)) << (16 * e); }
            near &= ~low;
        }
        }
    }
#ifndef HRL_NO_SECOND
    if (x.uniform(any_box)) { /* second support points (see ant_contacts): lane = (env, kept contact), the four envs at once */
        const unsigned long long any = x.each_ballot([&](int lane) {
            const WaveLds &L = x.lds(lane >> 4);
            const int i = lane & 15;
            int kf;
            return (i < L.nC) & second_worth_a_look(c, L, i < MAXC ? i : 0, &kf);
        });
        if (any) {
            x.each_compact16(
                [&](int lane) { const WaveLds &L = x.lds(lane >> 4); const int i = lane & 15; return second_support(c, L, L.q[qi], i < L.nC ? i : -1); },
                [&](int lane, int rank, const Hit &h) { WaveLds &L = x.lds(lane >> 4); store_contact(L, L.nC + rank, h, false, true); },
                [&](int lane, const Hit &, int count) { if ((lane & 15) == 0) x.lds(lane >> 4).ncnt = count; });
            x.each([&](int lane) {
                if ((lane & 15) == 0) { WaveLds &L = x.lds(lane >> 4); const int n = L.nC + L.ncnt; L.nC = n > cap ? cap : n; }
            });
        }
    }
#endif
    if (c.self_collision) { /* the envs with a joint outside the range in which no two legs can meet (see ant_contacts), one after the other */
        const bool thin = (c.r_caps + c.r_caps) + c.cdist < 0.2f;
        const unsigned long long unsafe = x.each_ballot([&](int lane) { /* lane = (env, joint) */
            const float *q = x.lds(lane >> 4).q[qi];
            const int j = lane & 15;
            return (j < NJ) & !(thin & (fabsf(q[7 + (j & 7)]) <= ((j & 1) ? 2.0f : 0.75f)));
        });
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            if (!((unsafe >> (16 * e)) & 0xffffull)) continue;
            WaveLds &L = x.lds(e);
            const int nC = x.uniform(L.nC);
            int cnt = x.each_compact([&](int lane) { return capsule_pair(c, L, lane < 48 ? lane : -1); },
                                     [&L, nC](int, int rank, const Hit &h) { store_contact(L, nC + rank, h, false); }, [&](int, const Hit &) {});
            const int nS = nC + cnt > cap ? cap - nC : cnt;
            x.each([&](int lane) { if (lane == 0) { L.nC = nC + nS; L.nS = nS; } });
        }
    }
    x.stamp(5);
#ifdef HRL_WGTIME
    x.each([&](int lane) { if ((lane & 15) == 0) { WaveLds &L = x.lds(lane >> 4); L.dbg_rows += 3 * L.nC | ((L.nS > 0 ? 1 : 0) << 24); } });
#endif
}

/* Limit rows of all four envs of a group by one wave: lane = (env, joint). */
template <class X>
HRL_DEV void ant_limits_group(X &x, const DevCfg &c, int qi) {
    x.refresh();
    x.each_compact16(
        [&](int lane) {
            const WaveLds &L = x.lds(lane >> 4);
            const float *q = L.q[qi];
            const int j = lane & 15;
            LimitHit r; r.ok = false; r.sgn = 0.f; r.dist = 0.f;
            if (j < NJ) {
                float dlo = q[7 + j] - L.jlim[0][j], dhi = L.jlim[1][j] - q[7 + j];
                if (dlo < c.lmargin) { r.ok = true; r.sgn = 1.f; r.dist = dlo; }
                else if (dhi < c.lmargin) { r.ok = true; r.sgn = -1.f; r.dist = dhi; }
            }
            return r;
        },
        [&](int lane, int rank, const LimitHit &r) { WaveLds &L = x.lds(lane >> 4); L.ljoint[rank] = lane & 15; L.lsign[rank] = r.sgn; L.ldist[rank] = r.dist; },
        [&](int lane, const LimitHit &, int count) {
            if ((lane & 15) == 0) {
                WaveLds &L = x.lds(lane >> 4);
                L.nL = count;
#ifdef HRL_WGTIME
                L.dbg_rows += count;
#endif
            }
        });
    x.stamp(6);
}

/* ENV block of a substep, on the env's own wave: on entry L.q[qi] / L.u / L.ustar, the articulated-body quantities of the group
 * block and the lists of ant_contacts are in the env's record; rows, sweeps, velocity reconstruction and clamp, then the
 * positions are integrated into L.q[qi ^ 1]. */
template <class X>
HRL_DEV int ant_env_block(X &x, const DevCfg &c, int qi) { /* returns the substep's solver rows (wave-uniform) */
    WaveLds &L = x.lds();
    x.refresh();
    const int nC = x.uniform(L.nC), nL = x.uniform(L.nL), nS = x.uniform(L.nS);
#ifndef HRL_NO_HOT_ROWS /* (A/B builds of tools/variants.py define it) */
    /* Longest job first: a launch lasts as long as its slowest env, and an env's chain grows with its rows.  An env with more rows than an ant
     * that stands under THIS config (DevCfg::hot_rows, derived from the model on the host: four feet x 3 + eight joints at their stops = 20 at the
     * defaults) takes the top issue priority for the rest of its block; the rotation at the head of the next substep takes it back.  Scheduling
     * only.  One such env among 4096 cost a launch 3 - 4 us; this gives 1.4 of them back (profiles/EXPERIMENTS.md 8). */
    if (nL + 3 * nC > c.hot_rows) x.priority(3);
#endif
    x.each([&](int lane) {
        x.reg(lane).ud = L.ustar[lane & 15]; /* the velocity if no row turns up */
        phase_build_row(c, L, x.reg(lane), lane, nL, nC);
    });
    if (nS > 0) x.each([&](int lane) { phase_self_rows(c, L, x.reg(lane), lane, nL, nC); });
    x.stamp(7);
    pgs_solve(x, c, nL, nC, true, nS > 0);
    x.each([&](int lane) {
        const int d = lane & 15;
        float v = x.reg(lane).ud;
        if (d >= 6 && d < 14) v = clampf(v, -c.vmax, c.vmax);
        if (lane < 16) L.u[lane] = v;
    });
    x.stamp(18);
    x.each([&](int lane) { phase_integrate(c, L, L.q[qi], L.q[qi ^ 1], lane & 15); });
    x.stamp(20);
    return nL + 3 * nC;
}

/* Who finds the contacts and limit rows while the leader (wave 0) runs the group block (7.5 k cycles): wave 1 the contacts of all four
 * envs, lane-packed (ant_contacts_group), wave 2 their limit rows (ant_limits_group); wave 3 waits at the barrier.  Records of a ragged
 * last group that hold no env are computed along (their results are never used).  A group of one does everything itself, in order. */
template <class X>
HRL_DEV void ant_contact_duty(X &x, const DevCfg &c, int qi, bool items_on) {
    const int w = x.wave_index();
    if (x.group_size() == 1) {
        if (x.uniform(x.lds().on)) { ant_contacts(x, c, x.lds(), qi, items_on); ant_limits(x, c, x.lds(), qi); }
        return;
    }
    if (w == 1) ant_contacts_group(x, c, qi, items_on);
    if (w == 2) ant_limits_group(x, c, qi);
}

/* ================================================================================================= POINT SUBSTEP
 * point_bot.py:10-74 + assets/player_cube.xml:8: free 10 kg cube (half extent 0.35).  Solid-cube inertia is
 * isotropic, so M^-1 = diag(1/I,1/I,1/I,1/m,1/m,1/m) and there is no gyroscopic term. */
struct CornerHit { bool ok; float dist, n[3], c[3]; int surf; };

template <class X>
HRL_DEV int point_substep(X &x, const DevCfg &c, int qi, bool items_on) {
    WaveLds &L = x.lds();
    const float *q = L.q[qi];
    float *qn = L.q[qi ^ 1];
    const float m = 10.f, he = 0.35f, I = m * (0.7f * 0.7f) / 6.f;
    const int cap = c.max_contacts;
    x.refresh();
    x.each([&](int lane) {
        const int d = lane & 15;
        float v = L.u[d];
        if (d == 3) v = fma_(c.h, L.tau[0] / m, L.u[3]);
        if (d == 4) v = fma_(c.h, L.tau[1] / m, L.u[4]);
        if (d == 5) v = fma_(c.h, L.tau[2] / m - c.g, L.u[5]);
        if (c.damping_on) { /* hrl_model.linear_damping / angular_damping (default: off): Bullet's damping of a free body with an isotropic inertia,
                               dv = -h v k_l (1 + |v|), domega = -h omega k_a (1 + |omega|), velocities of the start of the substep */
            const float nw = sqrtf(dot3(L.u, L.u)), nv = sqrtf(dot3(L.u + 3, L.u + 3));
            const float kk = d < 3 ? c.damp_ang * (1.f + nw) : c.damp_lin * (1.f + nv);
            v = fma_(-(c.h * kk), L.u[d < 6 ? d : 0], v);
        }
        x.reg(lane).ud = d < 6 ? v : 0.f;
        if (lane < 16) L.ustar[lane] = d < 6 ? v : 0.f;
        float ax_[3], ay_[3], az_[3];
        quat_axes(q[3], q[4], q[5], q[6], ax_, ay_, az_);
#pragma unroll
        for (int k = 0; k < 3; ++k) { L.XYZ[k] = ax_[k]; L.XYZ[3 + k] = ay_[k]; L.XYZ[6 + k] = az_[k]; }
    });
    int nC = 0;
    /* corner `lane & 7` against surface f (0 ground, 1.. lateral planes) or, item >= 0, against that item cube */
    auto corner = [&](int lane, bool on, int f, int item) {
        CornerHit h; h.ok = false; h.dist = 0.f; h.n[0] = h.n[1] = 0.f; h.n[2] = 1.f; h.c[0] = h.c[1] = h.c[2] = 0.f; h.surf = f;
        if (on) {
            float sx = (lane & 1) ? he : -he, sy = (lane & 2) ? he : -he, sz = (lane & 4) ? he : -he;
#pragma unroll
            for (int k = 0; k < 3; ++k) h.c[k] = fma_(sz, L.XYZ[6 + k], fma_(sy, L.XYZ[3 + k], sx * L.XYZ[k]));
            const float p[3] = {q[0] + h.c[0], q[1] + h.c[1], q[2] + h.c[2]};
            if (item >= 0) {
                const float ix = L.items[2 * item], iy = L.items[2 * item + 1];
                const float lo[3] = {ix - ITEM_HALF, iy - ITEM_HALF, ITEM_Z - ITEM_HALF}, hi[3] = {ix + ITEM_HALF, iy + ITEM_HALF, ITEM_Z + ITEM_HALF};
                h.dist = sphere_vs_box(p, 0.f, lo, hi, h.n); h.surf = surf_item(item);
            } else if (f == 0) h.dist = p[2] - c.ground_z;
            else {
                h.n[0] = L.planes[f - 1][0]; h.n[1] = L.planes[f - 1][1]; h.n[2] = L.planes[f - 1][2];
                h.dist = dot3(h.n, p) - L.planes[f - 1][3];
            }
            h.ok = h.dist < c.cdist;
        }
        return h;
    };
    auto keep = [&](int base) {
        return [&L, base](int, int rank, const CornerHit &h) {
            const int i = base + rank;
            if (i < MAXC) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { L.cr[i][k] = h.c[k]; L.cdir[0][i][k] = h.n[k]; }
                L.cdist_[i] = h.dist; L.csurf[i] = h.surf;
            }
        };
    };
    { /* ground and the (at most four) lateral planes in one pass: lane >> 3 = surface, lane & 7 = corner -- the surface-major,
         corner-minor candidate order of the specification is the lane order */
        int cnt = x.each_compact([&](int lane) { return corner(lane, (lane >> 3) < 1 + c.n_planes, lane >> 3, -1); }, keep(nC), [&](int, const CornerHit &) {});
        nC += cnt;
        if (nC > cap) nC = cap;
    }
    if (items_on) { /* cubes whose box comes within reach of a corner (half diagonal 0.35 sqrt 3 = 0.607), four per pass */
        const float R = 0.35f * 1.7320508f + ITEM_HALF + c.cdist + 0.02f;
        unsigned long long near = x.each_ballot([&](int lane) { /* lane = item (at most 64 of them) */
            return (lane < c.n_food + c.n_poison) & (fabsf(q[0] - L.items[2 * lane]) < R) & (fabsf(q[1] - L.items[2 * lane + 1]) < R);
        });
        const unsigned long long near0 = near;
        while (near) {
            int it[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                it[k] = -1;
                if (near) { it[k] = (int)__builtin_ctzll(near); near &= near - 1; }
            }
            int cnt = x.each_compact(
                [&](int lane) {
                    const int slot = lane >> 3;
                    const int item = slot == 0 ? it[0] : (slot == 1 ? it[1] : (slot == 2 ? it[2] : (slot == 3 ? it[3] : -1)));
                    return corner(lane, item >= 0, 0, item);
                },
                keep(nC), [&](int, const CornerHit &) {});
            nC += cnt;
            if (nC > cap) nC = cap;
        }
        /* then the near cubes' own 8 corners against the player's oriented box -- what catches a cube under the middle of a face --,
           eight cubes per pass: the corner in the box frame, its closest surface point, the normal turned back to the world and
           towards the player */
        x.refresh();
#pragma unroll 1
        for (int ib = 0; ib < 64; ib += 32) { /* items 0..31, then (configs with more than 32 items) 32..63: ascending item order either way */
        unsigned near32 = (unsigned)(near0 >> ib);
        while (near32) {
            const unsigned group = near32; /* slot `lane >> 3` takes the group's slot-th near cube */
#pragma unroll
            for (int k = 0; k < 8; ++k) near32 &= near32 - 1;
            int cnt = x.each_compact(
                [&](int lane) {
                    const int slot = lane >> 3;
                    unsigned m = group;
#pragma unroll
                    for (int k = 0; k < 7; ++k) m = k < slot ? (m & (m - 1)) : m;
                    const int item = m ? ib + (int)__builtin_ctz(m) : -1;
                    CornerHit h; h.ok = false; h.dist = 0.f; h.n[0] = h.n[1] = 0.f; h.n[2] = 1.f; h.c[0] = h.c[1] = h.c[2] = 0.f; h.surf = 0;
                    if (item >= 0) {
                        const float ix = L.items[2 * item], iy = L.items[2 * item + 1];
                        const float pc[3] = {(lane & 1) ? ix + ITEM_HALF : ix - ITEM_HALF, (lane & 2) ? iy + ITEM_HALF : iy - ITEM_HALF,
                                             (lane & 4) ? ITEM_Z + ITEM_HALF : ITEM_Z - ITEM_HALF};
                        const float d[3] = {pc[0] - q[0], pc[1] - q[1], pc[2] - q[2]};
                        const float l[3] = {dot3(d, &L.XYZ[0]), dot3(d, &L.XYZ[3]), dot3(d, &L.XYZ[6])}, blo[3] = {-he, -he, -he}, bhi[3] = {he, he, he};
                        float nl[3];
                        h.dist = sphere_vs_box(l, 0.f, blo, bhi, nl); h.surf = surf_item(item);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float nw = fma_(nl[2], L.XYZ[6 + k], fma_(nl[1], L.XYZ[3 + k], nl[0] * L.XYZ[k]));
                            h.n[k] = -nw; h.c[k] = fma_(-h.dist, nw, d[k]);
                        }
                        h.ok = h.dist < c.cdist;
                    }
                    return h;
                },
                keep(nC), [&](int, const CornerHit &) {});
            nC += cnt;
            if (nC > cap) nC = cap;
        }
        }
    }
    x.each([&](int lane) { /* contact map: tangents of the kept contacts */
        if (lane < nC) { const float n[3] = {L.cdir[0][lane][0], L.cdir[0][lane][1], L.cdir[0][lane][2]}; store_contact_frame(L, lane, n); }
    });
    x.each([&](int lane) { /* row map */
        LaneRegs &g = x.reg(lane);
        g.fn = -1;
        if (lane >= 3 * nC) return;
        const int ci = lane < nC ? lane : (lane - nC) >> 1, which = lane < nC ? 0 : 1 + ((lane - nC) & 1);
        const float r[3] = {L.cr[ci][0], L.cr[ci][1], L.cr[ci][2]};
        const float d[3] = {L.cdir[which][ci][0], L.cdir[which][ci][1], L.cdir[which][ci][2]};
        float J[6], B[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) B[k] = 0.f;
        cross3(J, r, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) { J[3 + k] = d[k]; B[k] = J[k] / I; B[3 + k] = d[k] / m; }
        const float dist = L.cdist_[ci];
#pragma unroll
        for (int k = 0; k < 6; ++k) g.Jb[k] = J[k];
        g.Jh = 0.f; g.Ja = 0.f; g.jslot = 6 | (6 << 8); g.mu = which == 0 ? 0.f : c.mu;
#pragma unroll
        for (int k = 0; k < 16; ++k) L.Bt[lane][k] = B[k];
        g.bias = which == 0 ? (dist > 0.f ? dist : c.erp_c * dist) * c.inv_h : 0.f;
        g.fn = which == 0 ? -1 : ci; g.lam = 0.f; g.lo = 0.f; g.hi = which == 0 ? 1e30f : 0.f;
    });
    pgs_solve<MAXC>(x, c, 0, nC, false, false); /* no limit rows: at most MAXC bounded rows */
    x.each([&](int lane) { if (lane < 16) L.u[lane] = x.reg(lane).ud; });
    x.stamp(10);
    x.each([&](int lane) { phase_integrate(c, L, q, qn, lane); });
    x.stamp(11);
    return nC;
}

/* ================================================================================================= OBSERVATIONS */

/* pybullet getEulerFromQuaternion (upstream, restated from memory) */
HRL_DEV void quat_to_rpy(const float *qq, float *rpy) {
    float x = qq[0], y = qq[1], z = qq[2], w = qq[3];
    float sarg = -2.f * (x * z - w * y);
    const float hp = 1.5707963267948966f;
    if (sarg <= -0.99999f) { rpy[0] = 0.f; rpy[1] = -hp; rpy[2] = 2.f * atan2_spec(x, -y); }
    else if (sarg >= 0.99999f) { rpy[0] = 0.f; rpy[1] = hp; rpy[2] = 2.f * atan2_spec(-x, y); }
    else {
        float sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
        rpy[0] = atan2_spec(2.f * (y * z + w * x), ((sqw - sqx) - sqy) + sqz);
        rpy[1] = asin_spec(sarg);
        rpy[2] = atan2_spec(2.f * (x * y + w * z), ((sqw + sqx) - sqy) - sqz);
    }
}
/* ant_gather_env.py:148-155: python `%` then fold to (-pi, pi] */
HRL_DEV float wrap_angle(float a) {
    const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
    if (!(fabsf(a) < two_pi)) a = fmodf(a, two_pi); /* fmod is exact: for |a| < 2 pi it returns a itself */
    if (a < 0.f) a += two_pi;
    if (a >= two_pi) a -= two_pi;
    if (a > pi) a = a - two_pi;
    if (a < -pi) a = a + two_pi;
    return a;
}
/* intersection_utils.py:93-104 */
HRL_DEV int quadrant(float x, float y) {
    if (x >= 0.f && y >= 0.f) return 1;
    if (x >= 0.f && y <= 0.f) return 4;
    if (x <= 0.f && y >= 0.f) return 2;
    if (x <= 0.f && y <= 0.f) return 3;
    return -1;
}
/* intersection_utils.py:84-90 */
HRL_DEV bool inf_intersection(float x1, float y1, float x2, float y2, float x3, float y3, float x4, float y4, float *px, float *py) {
    float d = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
    if (d == 0.f) return false;
    *px = ((x1 * y2 - y1 * x2) * (x3 - x4) - (x1 - x2) * (x3 * y4 - y3 * x4)) / d;
    *py = ((x1 * y2 - y1 * x2) * (y3 - y4) - (y1 - y2) * (x3 * y4 - y3 * x4)) / d;
    return true;
}
/* intersection_utils.py:14-71 */
HRL_DEV int orientation(float px, float py, float qx, float qy, float rx, float ry) {
    float val = ((qy - py) * (rx - qx)) - ((qx - px) * (ry - qy));
    return val > 0.f ? 1 : (val < 0.f ? 2 : 0);
}
HRL_DEV bool on_segment(float px, float py, float qx, float qy, float rx, float ry) {
    return (qx <= fmaxf(px, rx)) && (qx >= fminf(px, rx)) && (qy <= fmaxf(py, ry)) && (qy >= fminf(py, ry));
}
HRL_DEV bool segment_intersection(float p1x, float p1y, float q1x, float q1y, float p2x, float p2y, float q2x, float q2y) {
    int o1 = orientation(p1x, p1y, q1x, q1y, p2x, p2y), o2 = orientation(p1x, p1y, q1x, q1y, q2x, q2y);
    int o3 = orientation(p2x, p2y, q2x, q2y, p1x, p1y), o4 = orientation(p2x, p2y, q2x, q2y, q1x, q1y);
    if (o1 != o2 && o3 != o4) return true;
    if (o1 == 0 && on_segment(p1x, p1y, p2x, p2y, q1x, q1y)) return true;
    if (o2 == 0 && on_segment(p1x, p1y, q2x, q2y, q1x, q1y)) return true;
    if (o3 == 0 && on_segment(p2x, p2y, p1x, p1y, q2x, q2y)) return true;
    if (o4 == 0 && on_segment(p2x, p2y, q1x, q1y, q2x, q2y)) return true;
    return false;
}
/* MazeScene.bounds (maze_scene.py:15-21, sizeable_enclosed_scene.py:28-34): 4 world lines + 3 box lines */
HRL_DEV void maze_line(int l, float *a) {
    const float T[7][4] = {{5, 9, -5, 9}, {5, 9, 5, -9}, {-5, -9, -5, 9}, {-5, -9, 5, -9}, {1, 2, 1, -2}, {-5, -2, -5, 2}, {-5, -2, 1, -2}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = T[0][k];
#pragma unroll
        for (int i = 1; i < 7; ++i) v = (l == i) ? T[i][k] : v;
        a[k] = v;
    }
}

/* ant_flagrun_env.py:71-78 with counter-based draws: the k-th goal of episode `ep`; one stream shared by all envs
 * (the reference shares one RandomState between its parallel envs, :38-39); at most 64 attempts */
HRL_DEV void flag_goal(const DevCfg &c, uint32_t ep, uint32_t k, float *gx, float *gy) {
    for (uint32_t a = 0; a < 64; ++a) {
        uint32_t r[4];
        philox4x32(c, 0, ep, (4u << 16) | k, a, r);
        *gx = -c.flag_size / 2.f + c.flag_size * u01(r[0]); *gy = -c.flag_size / 2.f + c.flag_size * u01(r[1]);
        if (!(sqrtf(*gx * *gx + *gy * *gy) < 0.5f)) break;
    }
}

/* ant_flagrun_env.py:80-89 `create_close_target` with counter-based draws (max_target_dist mode): the k-th goal of episode
 * `ep` of env `env`, per axis +-U(tol, max_target_dist / 2) around the robot's xy, redrawn until strictly inside the
 * arena; one Philox block per attempt (two uniforms, two sign bits), at most 64 attempts, the last one kept */
HRL_DEV void flag_close_goal(const DevCfg &c, long long env, uint32_t ep, uint32_t k, float rx, float ry, float *gx, float *gy) {
    const float wb = c.flag_size / 2.f, half = c.flag_mtd / 2.f;
    for (uint32_t a = 0; a < 64; ++a) {
        uint32_t r[4];
        philox4x32(c, env, ep, (5u << 16) | (k & 0xffffu), a, r);
        *gx = (c.tol + (half - c.tol) * u01(r[0])) * ((r[2] & 1u) ? 1.f : -1.f) + rx;
        *gy = (c.tol + (half - c.tol) * u01(r[1])) * ((r[3] & 1u) ? 1.f : -1.f) + ry;
        if (-wb < *gx && *gx < wb && -wb < *gy && *gy < wb) break;
    }
}

/* ant_flagrun_env.py:98-103 `set_target`: besides the walk target, where the robot stands now (`robot_body.get_position()[:2]`) and
 * `np.linalg.norm(goal - pos) ** 2` -- the norm squared, not the sum of squares -- for the path reward of :174-176; into the items record */
HRL_DEV void flag_set_target_state(WaveLds &L, float gx, float gy, float px, float py) {
    const float dx = gx - px, dy = gy - py, nrm = sqrtf(dx * dx + dy * dy);
    L.items[HRL_FLAG_START_OFF] = px; L.items[HRL_FLAG_START_OFF + 1] = py; L.items[HRL_FLAG_SQDIST_OFF] = nrm * nrm;
}

/* sizeable_enclosed_scene.py:63-97 `sense_walls`, one bin: the ray and all 7 maze lines are INFINITE lines
 * (intersection_utils.py:74-90), filtered by range and quadrant */
HRL_DEV float wall_sensor_bin(const DevCfg &c, float rx, float ry, float yaw, int i, bool arena) {
    const float half_pi = 1.5707963267948966f;
    float phi;
    if (c.span_is_2pi) phi = half_pi + yaw + ((float)(i + 1) / (float)c.n_bins) * c.sensor_span;
    else phi = half_pi + yaw + ((float)i / (float)(c.n_bins - 1)) * c.sensor_span;
    const float svx = rx + c.sensor_range * cos_spec(phi), svy = ry + c.sensor_range * sin_spec(phi);
    const int sq = quadrant(svx - rx, svy - ry);
    float best = 0.f;
    for (int l = 0; l < (arena ? 4 : 7); ++l) {
        float a[4], px, py;
        if (arena) { /* the 4 world lines of a SizeableEnclosedScene (sizeable_enclosed_scene.py:28-34) */
            const float hx = c.world_sx / 2.f, hy = c.world_sy / 2.f;
            a[0] = l < 2 ? hx : -hx; a[1] = l < 2 ? hy : -hy;
            a[2] = (l == 0 || l == 2) ? -hx : hx; a[3] = (l == 0) ? hy : ((l == 1) ? -hy : ((l == 2) ? hy : -hy));
        } else maze_line(l, a);
        if (!inf_intersection(rx, ry, svx, svy, a[0], a[1], a[2], a[3], &px, &py)) continue;
        const float ddx = rx - px, ddy = ry - py, dist = sqrtf(ddx * ddx + ddy * ddy);
        if (dist > c.sensor_range) continue;
        if (sq != quadrant(px - rx, py - ry)) continue;
        const float val = 1.f - dist / c.sensor_range;
        if (val > best) best = val;
    }
    return best;
}

/* maze kinds: the episode's target, entry aux[3] of the constructor's `targets` (ant_maze_bullet_env.py:114-115): a wave-uniform index,
 * so the pair is a scalar load from the constants; an index outside the table (the caller owns `aux`) reads entry 0 */
template <class X>
HRL_DEV void maze_target(X &x, const DevCfg &c, WaveLds &L, float *tx, float *ty) {
    const unsigned ti = (unsigned)x.uniform(L.aux[3]);
    const int k = ti < (unsigned)HRL_MAX_TARGETS ? (int)ti : 0;
    *tx = c.targets[k][0]; *ty = c.targets[k][1];
}

/* Phase O1: upstream WalkerBase.calc_state (28-vector clipped to +-5) into L.s28, plus walk_target_dist, yaw and
 * joints_at_limit into L.scal.  `with_centroid` needs L.ph/pa/tip of the CURRENT qpos (phase_kin_ankle on L.st). */
template <int KIND>
HRL_DEV void phase_calc_state(const DevCfg &c, WaveLds &L, int lane, bool use_feet, bool with_centroid, float mtx, float mty) {
    const float *qp = L.st, *qv = L.st + 15;
    float rpy[3];
    quat_to_rpy(qp + 3, rpy);
    float tx = c.walk_tx, ty = c.walk_ty;
    if (KIND == 5) { /* the goal being chased: from the shared list, or (max_target_dist mode) the one kept in items[0..1] */
        if (c.flag_mtd > 0.f || c.flag_manual) { tx = L.items[0]; ty = L.items[1]; }
        else flag_goal(c, (uint32_t)L.aux[2], (uint32_t)L.aux[3] & 0xffffu, &tx, &ty);
    }
    if (KIND == 2 || KIND == 4) { tx = mtx; ty = mty; } /* maze kinds: the episode's target (maze_target) */
    if (KIND == 5) { L.red[4] = tx; L.red[5] = ty; } /* the goal this state looks at: path reward, `goal` output (every lane writes the same values) */
    float cx = qp[0], cy = qp[1];
    if (with_centroid) {
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            sx += (0.5f * L.ph[l][0] + 0.5f * (L.ph[l][0] + L.pa[l][0])) + 0.5f * (L.pa[l][0] + L.tip[l][0]);
            sy += (0.5f * L.ph[l][1] + 0.5f * (L.ph[l][1] + L.pa[l][1])) + 0.5f * (L.pa[l][1] + L.tip[l][1]);
        }
        const float np_ = (float)(13 + c.centroid_n_static);
        cx = ((13.f * qp[0] + sx) + c.centroid_sx) / np_;
        cy = ((13.f * qp[1] + sy) + c.centroid_sy) / np_;
    }
    const float dx = tx - cx, dy = ty - cy;
    /* AntGather / PointGather drop the two angle-to-target entries (ant_gather_env.py:81) and never use the target
     * distance: skip their transcendental work there */
    float sin_ang = 0.f, cos_ang = 0.f, wtd = 0.f;
    if (KIND != 1) {
        const float theta = atan2_spec(dy, dx), ang = theta - rpy[2];
        wtd = sqrtf(dy * dy + dx * dx); sin_ang = sin_spec(ang); cos_ang = cos_spec(ang);
    }
    const float cs = cos_spec(-rpy[2]), sn = sin_spec(-rpy[2]);
    const float vx = cs * qv[0] - sn * qv[1], vy = sn * qv[0] + cs * qv[1], vz = qv[2];
    int nlim = 0;
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float rel = (qp[7 + j] - c.jmid[j]) * c.jscale[j];
        if (fabsf(rel) > 0.99f) ++nlim;
        mine = (lane == 8 + 2 * j) ? rel : mine;
        mine = (lane == 9 + 2 * j) ? 0.1f * qv[6 + j] : mine;
    }
    mine = (lane == 0) ? qp[2] - L.st[30] : mine;
    mine = (lane == 1) ? sin_ang : mine;
    mine = (lane == 2) ? cos_ang : mine;
    mine = (lane == 3) ? 0.3f * vx : mine;
    mine = (lane == 4) ? 0.3f * vy : mine;
    mine = (lane == 5) ? 0.3f * vz : mine;
    mine = (lane == 6) ? rpy[0] : mine;
    mine = (lane == 7) ? rpy[1] : mine;
    if (lane >= 24 && lane < 28) { /* robot.feet_contact as the LAST step left it (aux[1] bits 28..31): upstream's step() calls calc_state() before it
                                      refreshes the flags from the step's contacts, so the observation shows the previous step's (step_entry) */
        const int l = lane - 24;
        mine = (use_feet && ((L.aux[1] >> (28 + l)) & 1)) ? 1.f : 0.f;
    }
    if (lane < 28) L.s28[lane] = clampf(mine, -5.f, 5.f);
    L.scal[3] = wtd; L.scal[4] = rpy[2]; L.scal[5] = (float)nlim; L.scal[6] = cx; L.scal[7] = cy;
}

/* gather_scene.py:52-62 with counter-based draws: at most 64 attempts, the last one is kept */
HRL_DEV void respawn_item(const DevCfg &c, long long env, uint32_t index, uint32_t purpose, int item, float ax, float ay, float *px, float *py) {
    const float sxw = c.world_sx - 1.f, syw = c.world_sy - 1.f;
    for (uint32_t a = 0; a < 64; ++a) {
        uint32_t r[4];
        philox4x32(c, env, index, (purpose << 16) | (uint32_t)item, a, r);
        *px = u01(r[0]) * sxw - sxw / 2.f; *py = u01(r[1]) * syw - syw / 2.f;
        const float dx = ax - *px, dy = ay - *py;
        if (!(sqrtf(dx * dx + dy * dy) < c.spacing)) break;
    }
}

/* Phase O2 (item map): pickup + respawn (ant_gather_env.py:84-92, gather_scene.py:95-114) and the item's sensor
 * contribution (ant_gather_env.py:145-162).  robot_coll_dist <= 0 (:113-116): +-1 per contact point between the robot and
 * the item's cube among the `n_contacts` contacts of the step's last collision pass (L.csurf), the observation is taken
 * BEFORE such an item is moved (:95-96 precede :113). */
HRL_DEV void phase_items(const DevCfg &c, WaveLds &L, int lane, long long env, bool pickups, int n_contacts) {
    const int n = c.n_food + c.n_poison; /* <= HRL_MAX_ITEMS = the lanes of the wave */
    float rew = 0.f, bin = -1.f, inten = 0.f;
    if (lane < n) {
        const float rx = L.st[0], ry = L.st[1], yaw = L.scal[4];
        float ix = L.items[2 * lane], iy = L.items[2 * lane + 1];
        float dx = ix - rx, dy = iy - ry, d2 = dx * dx + dy * dy;
        if (pickups && c.coll_dist > 0.f && d2 < c.coll_dist) {
            rew = lane < c.n_food ? 1.f : -1.f;
            if (c.respawn) respawn_item(c, env, (uint32_t)L.aux[1], 0u, lane, rx, ry, &ix, &iy);
            else { ix = 100.f; iy = 0.f; }
            L.items[2 * lane] = ix; L.items[2 * lane + 1] = iy;
            dx = ix - rx; dy = iy - ry; d2 = dx * dx + dy * dy;
        }
        if (!c.use_sensor) inten = d2; /* get_abs_pos (ant_gather_env.py:179-196) sorts by squared distance */
        else if (d2 != d2) bin = -2.f; /* a NaN distance (NaN robot or item coordinate): every reading of the item's type is NaN (phase_pack_obs) */
        else if (!(d2 > c.sensor_range)) {
            const float half_span = c.sensor_span * 0.5f, bin_res = c.sensor_span / (float)c.n_bins;
            const float angle = wrap_angle(atan2_spec(iy - ry, ix - rx) - yaw);
            if (fabsf(angle) <= half_span) {
                int b = (int)((angle + half_span) / bin_res);
                if (b >= c.n_bins) b = c.n_bins - 1;
                bin = (float)b; inten = 1.0f - d2 / c.sensor_range;
            }
        }
        if (pickups && !(c.coll_dist > 0.f)) { /* the item itself moves after the observation has been packed (phase_items_contact_move) */
            int hits = 0;
            for (int i = 0; i < n_contacts; ++i) hits += L.csurf[i] == surf_item(lane) ? 1 : 0;
            rew = (lane < c.n_food ? 1.f : -1.f) * (float)hits;
        }
    }
    L.irew[lane] = rew; L.ibin[lane] = bin; L.iint[lane] = inten;
}

/* robot_coll_dist <= 0: an item the robot touched moves once the observation is packed -- get_food_obs (ant_gather_env.py:95-96) reads the
 * item positions before reward_collision (:113-116) moves them, which shows where the observation holds positions (get_abs_pos).  The
 * reference moves the item once per contact point (gather_scene.py:95-114); the moves are independent draws, keyed item | move << 4 (<< 6 with
 * more than 16 items), so the last one is where it ends up.  The number of contact points is |L.irew|. */
HRL_DEV void phase_items_contact_move(const DevCfg &c, WaveLds &L, int lane, long long env) {
    if (lane >= c.n_food + c.n_poison) return;
    const int hits = (int)fabsf(L.irew[lane]);
    if (hits > 0) {
        float ix = 100.f, iy = 0.f;
        if (c.respawn) respawn_item(c, env, (uint32_t)L.aux[1], 0u, lane | ((hits - 1) << c.item_shift), L.st[0], L.st[1], &ix, &iy);
        L.items[2 * lane] = ix; L.items[2 * lane + 1] = iy;
    }
}

/* Phase O3: final observation vector into L.obs, non-finite flag into L.flags[0] */
template <int KIND>
HRL_DEV void phase_pack_obs(const DevCfg &c, WaveLds &L, int lane /* = observation element: lane + 64 * pass */, float mtx, float mty) {
    if (lane >= c.obs_dim) return;
    float v = 0.f;
    if (KIND == 0) v = L.st[lane];                       /* MjAnt.py:17-25: qpos | qvel */
    else if (KIND == 5) { /* ant_flagrun_env.py:122-130: the full 28-vector (+ wall sensor over the arena's 4 lines) */
        if (lane < 28) v = L.s28[lane];
        else v = wall_sensor_bin(c, L.st[0], L.st[1], L.scal[4], lane - 28, true);
    }
    else if (KIND == 4) { /* ant_maze_mj_env.py:57-64: state29 | walls | pit zeros | moveable zeros | t * 0.001 */
        if (lane < 29) v = L.st[lane];
        else if (lane < 29 + c.n_bins) v = wall_sensor_bin(c, L.st[0], L.st[1], L.scal[4], lane - 29, false);
        else if (lane == 29 + 3 * c.n_bins) v = (float)L.aux[0] * 0.001f;
    }
    else if (KIND == 3) {                                /* point: calc_state(8) | food | poison */
        if (lane < 8) v = L.s28[lane];
    } else if (lane < 26) v = L.s28[lane == 0 ? 0 : lane + 2]; /* ant_gather_env.py:81, ant_maze_bullet_env.py:75 */
    const int nb = (KIND == 3) ? 8 : 26;
    if ((KIND == 1 || KIND == 3) && lane >= nb && c.use_sensor) { /* ant_gather_env.py:128-177: nearest in-range item per bin and type */
        const int b = lane - nb, type = b / c.n_bins, bin = b - type * c.n_bins;
        const int k0 = type ? c.n_food : 0, k1 = type ? c.n_food + c.n_poison : c.n_food;
        float best = 0.f;
        /* all 16 slots, the type's range as a predicate: the 32 loads are in flight together (a loop over [k0, k1) paid the LDS latency
           twice per item); same order, same comparisons */
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool mine = (k >= k0) & (k < k1);
            const bool take = mine & (L.ibin[k] == (float)bin) & (L.iint[k] > best);
            best = take ? L.iint[k] : best;
            bad = bad | (mine & (L.ibin[k] == -2.f));
        }
#pragma unroll 1
        for (int k = 16; k < k1; ++k) { /* configs with more than 16 items: the rest of the slots, same order, same comparisons */
            const bool mine = k >= k0;
            const bool take = mine & (L.ibin[k] == (float)bin) & (L.iint[k] > best);
            best = take ? L.iint[k] : best;
            bad = bad | (mine & (L.ibin[k] == -2.f));
        }
        v = bad ? __builtin_nanf("") : best; /* a numerical failure, not a reading: the episode ends (ant_gather_env.py:101-103) */
    }
    if ((KIND == 1 || KIND == 3) && lane >= nb && !c.use_sensor) { /* ant_gather_env.py:179-196: xy of the nearest items, stable ascending by d2 */
        const int mf = c.n_food < c.n_bins ? c.n_food : c.n_bins;
        const int b = lane - nb, type = b >= 2 * mf, bb = type ? b - 2 * mf : b, want = bb >> 1, comp = bb & 1;
        const int k0 = type ? c.n_food : 0, k1 = type ? c.n_food + c.n_poison : c.n_food;
        bool bad = false;
        for (int i = k0; i < k1; ++i) {
            int rank = 0;
            for (int j = k0; j < k1; ++j) rank += (L.iint[j] < L.iint[i] || (L.iint[j] == L.iint[i] && j < i)) ? 1 : 0;
            if (rank == want) v = L.items[2 * i + comp];
            bad = bad | (L.iint[i] != L.iint[i]);
        }
        if (bad && want < (k1 - k0)) v = __builtin_nanf(""); /* a NaN distance has no place in the order: the type's outputs are NaN */
    }
    if (KIND == 2 && lane >= 26) {
        const float rx = L.st[0], ry = L.st[1], yaw = L.scal[4];
        const float tx = mtx, ty = mty;
        const int ntar = c.sense_target ? c.n_bins : 2;
        if (lane < 26 + ntar) {
            const int i = lane - 26;
            if (!c.sense_target) { /* ant_maze_bullet_env.py:123-133 */
                const float vx = tx - rx, vy = ty - ry;
                if (c.target_encoding == 0) { const float n = sqrtf(vx * vx + vy * vy); v = i == 0 ? vx / n : vy / n; }
                else { const float a = atan2_spec(vy, vx) - yaw; v = i == 0 ? sin_spec(a) : cos_spec(a); }
            } else { /* ant_maze_bullet_env.py:135-178 */
                const float wtd = L.scal[3];
                bool vis = !(wtd > c.sensor_range);
                for (int l = 4; l < 7 && vis; ++l) { float a[4]; maze_line(l, a); if (segment_intersection(rx, ry, tx, ty, a[0], a[1], a[2], a[3])) vis = false; }
                if (vis) {
                    const float angle = wrap_angle(atan2_spec(ty - ry, tx - rx) - yaw), half_span = c.sensor_span * 0.5f;
                    if (fabsf(angle) <= half_span) {
                        int b = (int)((angle + half_span) / (c.sensor_span / (float)c.n_bins));
                        if (b >= c.n_bins) b = c.n_bins - 1;
                        if (b == i) v = 1.0f - wtd / c.sensor_range;
                    }
                }
            }
        } else v = wall_sensor_bin(c, rx, ry, yaw, lane - 26 - ntar, false); /* lane = bin */
    }
    L.obs[lane] = v;
    if (!isfinite(v)) L.flags[0] = 1;
}

/* point_bot.py:48-67 into L.s28[0..7] (walk target (0,0), initial_z 1) */
HRL_DEV void phase_point_state(const DevCfg &c, WaveLds &L, int lane) {
    const float *qp = L.st, *qv = L.st + 15;
    float rpy[3];
    quat_to_rpy(qp + 3, rpy);
    const float theta = atan2_spec(0.f - qp[1], 0.f - qp[0]), a = theta - rpy[2];
    const float cs = cos_spec(-rpy[2]), sn = sin_spec(-rpy[2]);
    const float vx = cs * qv[0] - sn * qv[1], vy = sn * qv[0] + cs * qv[1], vz = qv[2];
    float mine = qp[2] - 1.f;
    mine = (lane == 1) ? sin_spec(a) : mine;
    mine = (lane == 2) ? cos_spec(a) : mine;
    mine = (lane == 3) ? 0.3f * vx : mine;
    mine = (lane == 4) ? 0.3f * vy : mine;
    mine = (lane == 5) ? 0.3f * vz : mine;
    mine = (lane == 6) ? rpy[0] : mine;
    mine = (lane == 7) ? rpy[1] : mine;
    if (lane < 8) L.s28[lane] = mine;
    L.scal[3] = 0.f; L.scal[4] = rpy[2]; L.scal[5] = 0.f;
    (void)c;
}

/* Observation of the state in L.st / L.items / L.aux into L.obs (and L.scal).  Used by step and by reset. */
template <int KIND, class X>
HRL_DEV void compute_obs(X &x, const DevCfg &c, long long env, bool step_mode, int n_contacts = 0) {
    WaveLds &L = x.lds();
    const bool centroid = (KIND == 0 || KIND == 2 || KIND == 4 || KIND == 5);
    if (centroid) { /* link positions of the final pose for the parts centroid; its LDS hand-off area overlays the
                       task scratch, so it runs before anything below is written */
        x.each([&](int lane) { if (lane < 16) L.q[0][lane] = lane < 15 ? L.st[lane] : 0.f; });
        x.each([&](int lane) { phase_kin_ankle<true>(c, L, x.reg(lane), L.q[0], lane); });
    }
    float mtx = 0.f, mty = 0.f;
    if (KIND == 2 || KIND == 4) maze_target(x, c, L, &mtx, &mty);
    x.each([&](int lane) { if (lane < 4) L.flags[lane] = 0; });
    if (KIND == 3) x.each([&](int lane) { phase_point_state(c, L, lane); });
    else {
        const bool feet = KIND == 2 || KIND == 5; /* ant_gather_env.py:105-111: feet flags stay 0 in AntGather; a reset clears them (upstream robot_specific_reset) */
        x.each([&](int lane) { phase_calc_state<KIND>(c, L, lane, feet, centroid, mtx, mty); });
    }
    x.stamp(21);
    if (KIND == 1 || KIND == 3) x.each([&](int lane) { phase_items(c, L, lane, env, step_mode, n_contacts); });
    x.stamp(22);
    /* lane = observation element; observations wider than the wave (more than 19 bins, say) take ceil(obs_dim / 64) passes (one copy of the
       packing code: a wave-uniform loop that the default configs run once) */
#pragma unroll 1
    for (int o = 0; o < c.obs_dim; o += 64) x.each([&](int lane) { phase_pack_obs<KIND>(c, L, lane + o, mtx, mty); });
    if ((KIND == 1 || KIND == 3) && step_mode && !(c.coll_dist > 0.f)) x.each([&](int lane) { phase_items_contact_move(c, L, lane, env); });
}

/* ================================================================================================= RESET / STEP */

/* Env.reset(): ant_gather_env.py:68-74, gather_scene.py:38-50, ant_maze_bullet_env.py:104-121, point_bot.py:12,25-26.
 * The potential a reset leaves behind is what upstream WalkerBaseBulletEnv.reset() computed BEFORE the in-tree reset code
 * moved the robot / switched the target (oracle: orc_reset_potential): flagrun -- the new pose against the PREVIOUS goal
 * (ant_flagrun_env.py:116 re-reads the walk_target_dist of :146); maze kinds -- the default pose (the start offset taken
 * out of the 13 robot parts of the centroid) against the PREVIOUS target (ant_maze_bullet_env.py:111 precedes :114-119);
 * before the first episode the walk target is upstream's default (1e3, 0). */
template <int KIND, class X>
HRL_DEV void reset_env(X &x, const DevCfg &c, long long env) {
    WaveLds &L = x.lds();
    float ptx = c.walk_tx, pty = c.walk_ty; /* what the robot was walking towards before this reset */
    if (KIND == 2 || KIND == 4 || KIND == 5) {
        if (x.uniform(L.aux[2]) > 0) {
            if (KIND == 5) {
                if (c.flag_mtd > 0.f || c.flag_manual) { ptx = L.items[0]; pty = L.items[1]; }
                else flag_goal(c, (uint32_t)L.aux[2], (uint32_t)L.aux[3] & 0xffffu, &ptx, &pty);
            } else maze_target(x, c, L, &ptx, &pty);
        }
        x.each([](int) {}); /* the reads above precede the writes below */
    }
    x.each([&](int lane) {
        const uint32_t ep = (uint32_t)L.aux[2];
        if (lane < 32) {
            float v = 0.f;
            if (lane == 6) v = 1.f;
            if (KIND == 3) { if (lane == 2) v = 0.5f; if (lane == 30) v = 1.f; }
            else {
                float z0 = 0.75f;
                if (KIND == 2 || KIND == 4 || KIND == 5) { z0 = c.start_pos[2]; if (lane == 0) v = c.start_pos[0]; if (lane == 1) v = c.start_pos[1]; }
                if (lane == 2 || lane == 30) v = z0;
                if (lane >= 7 && lane < 15) {
                    const int j = lane - 7;
                    uint32_t r[4];
                    philox4x32(c, env, ep, (2u << 16) | (uint32_t)(j >> 2), 0u, r);
                    const uint32_t rr = (j & 3) == 0 ? r[0] : ((j & 3) == 1 ? r[1] : ((j & 3) == 2 ? r[2] : r[3]));
                    v = -0.1f + 0.2f * u01(rr);
                }
            }
            L.st[lane] = v;
        } else if (KIND == 1 || KIND == 3) {
            const int i = lane - 32; /* lanes 32..47: item i */
            if (i < 16) {
                float px = 0.f, py = 0.f;
                if (i < c.n_food + c.n_poison) respawn_item(c, env, ep, 1u, i, 0.f, 0.f, &px, &py);
                L.items[2 * i] = px; L.items[2 * i + 1] = py;
            }
        } else if (KIND == 5 && c.flag_path_on) { /* lanes 32..63: the first 32 words of the items record (flagrun layout: include/hrl_envs.h, HRL_FLAG_*; an env
                                                     of the shared goal list keeps set_target()'s bookkeeping in it, not the goal) */
            const int w = lane - 32;
            if (c.flag_manual) { /* the walk target and the path-reward state survive the reset, the pending goals do not (ant_flagrun_env.py:149-152) */
                if (w == 0) L.items[0] = ptx;
                if (w == 1) L.items[1] = pty;
                if (w >= HRL_FLAG_PENDING_OFF) L.items[w] = 0.f;
            } else { /* reset -> next_target -> set_target(first goal of the new episode) with the robot at the start pose (:144-153) */
                float gx = 0.f, gy = 0.f;
                if (w == 0) {
                    if (c.flag_mtd > 0.f) flag_close_goal(c, env, ep + 1u, 1u, c.start_pos[0], c.start_pos[1], &gx, &gy); /* create_close_target around the start pose, new episode's stream */
                    else flag_goal(c, ep + 1u, 1u, &gx, &gy);
                    flag_set_target_state(L, gx, gy, c.start_pos[0], c.start_pos[1]);
                    L.items[0] = c.flag_mtd > 0.f ? gx : 0.f; L.items[1] = c.flag_mtd > 0.f ? gy : 0.f; /* the shared list's goals are functions of (seed, episode, k): not stored */
                    L.items[HRL_FLAG_SQDIST_OFF + 1] = 0.f;
                } else if (w >= HRL_FLAG_PENDING_OFF) L.items[w] = 0.f;
            }
        }
        if (lane < 16) L.u[lane] = 0.f;
        if (lane < 8) L.tau[lane] = 0.f;
        if (lane == 63 && KIND == 5) L.aux[3] = (int)((c.flag_manual ? 0u : 1u) | ((uint32_t)L.aux[3] & 0x7fff0000u)); /* first goal popped (manual: none pending), not rewarded; steps_since_goal_change survives a reset (ant_flagrun_env.py:132-155 never assigns it) */
        if (lane == 63 && (KIND == 2 || KIND == 4)) {
            uint32_t r[4];
            philox4x32(c, env, ep, (3u << 16), 0u, r);
            L.aux[3] = (int)(r[0] % (uint32_t)c.n_targets);
        }
    });
    if ((KIND == 1 || KIND == 3 || KIND == 5) && c.items_stride > 32) /* a longer items record: items 16.. of a gather env (gather_scene.py:38-50), zeros elsewhere (flagrun: the rest of the pending list) */
        x.each([&](int lane) {
            const uint32_t ep = (uint32_t)L.aux[2];
            const int i = 16 + lane;
            if (i < HRL_MAX_ITEMS) {
                float px = 0.f, py = 0.f;
                if ((KIND == 1 || KIND == 3) && i < c.n_food + c.n_poison) respawn_item(c, env, ep, 1u, i, 0.f, 0.f, &px, &py);
                L.items[2 * i] = px; L.items[2 * i + 1] = py;
            }
        });
    x.each([&](int lane) {
        if (lane == 0) L.aux[0] = 0;
        if (lane == 1 && (KIND == 2 || KIND == 5)) L.aux[1] = L.aux[1] & 0x0fffffff; /* feet_contact = 0 (upstream robot_specific_reset) */
        if (lane == 2) L.aux[2] = L.aux[2] + 1;
    });
    compute_obs<KIND>(x, c, env, false);
    x.each([&](int lane) {
        if (lane != 31) return;
        float pot = 0.f;
        if (KIND == 0) pot = -L.scal[3] / c.dt; /* upstream calc_potential at the reset pose */
        if (KIND == 2 || KIND == 4 || KIND == 5) {
            float cx = L.scal[6], cy = L.scal[7];
            if (KIND != 5) { const float np_ = (float)(13 + c.centroid_n_static); cx = cx - (13.f * c.start_pos[0]) / np_; cy = cy - (13.f * c.start_pos[1]) / np_; }
            const float dx = ptx - cx, dy = pty - cy;
            pot = -sqrtf(dy * dy + dx * dx) / c.dt;
        }
        L.st[31] = pot;
    });
}

template <class X>
HRL_DEV void load_env(X &x, const DevBufs &b, const DevCfg &c, int e, bool with_actions) {
    WaveLds &L = x.lds();
    x.each([&](int lane) {
        if (lane < 32) L.st[lane] = b.state[(size_t)e * 32 + lane];
        else { const float v = b.items ? b.items[(size_t)e * c.items_stride + (lane - 32)] : 0.f; L.items[lane - 32] = v; L.items0[lane - 32] = v; }
#pragma unroll 1
        for (int o = 32; o < c.items_stride; o += 64) /* a longer items record (more than 16 items / 15 manual goals): the rest of it */
            if (o + lane < c.items_stride) L.items[o + lane] = b.items ? b.items[(size_t)e * c.items_stride + (o + lane)] : 0.f;
        if (lane < 4) L.aux[lane] = b.aux[(size_t)e * 4 + lane];
        if (lane < 8) L.act[lane] = (with_actions && lane < c.act_dim) ? b.actions[(size_t)e * c.act_dim + lane] : 0.f;
        if (lane >= 8 && lane < 16) { L.jlim[0][lane - 8] = c.jlo[lane - 8]; L.jlim[1][lane - 8] = c.jhi[lane - 8]; }
        if (lane >= 16 && lane < 32) { const int f = (lane - 16) >> 2, k = lane & 3; L.planes[f][k] = k < 3 ? c.plane_n[f][k] : c.plane_d[f]; }
    });
}
template <class X>
HRL_DEV void store_env(X &x, const DevBufs &b, const DevCfg &c, int e) {
    WaveLds &L = x.lds();
    x.each([&](int lane) {
        if (lane < 32) b.state[(size_t)e * 32 + lane] = L.st[lane];
        else if (b.items) { /* most steps change nothing in the items record: 128 B per env that need not be written (bit compare: a NaN is a change too) */
            const float v = L.items[lane - 32];
            if (float_bits(v) != float_bits(L.items0[lane - 32])) b.items[(size_t)e * c.items_stride + (lane - 32)] = v;
        }
#pragma unroll 1
        for (int o = 32; o < c.items_stride; o += 64)
            if (b.items && o + lane < c.items_stride) b.items[(size_t)e * c.items_stride + (o + lane)] = L.items[o + lane];
        if (lane < 4) b.aux[(size_t)e * 4 + lane] = L.aux[lane];
        if (lane < c.obs_dim) b.obs[(size_t)e * c.obs_dim + lane] = L.obs[lane];
#pragma unroll 1
        for (int o = 64; o < c.obs_dim; o += 64) /* observations wider than the wave */
            if (o + lane < c.obs_dim) b.obs[(size_t)e * c.obs_dim + (o + lane)] = L.obs[o + lane];
    });
}

/* hrl_reset for one env (one wave) */
template <int KIND, class X>
HRL_DEV void reset_entry(X &x, const DevBufs &b, const DevCfg &c, int e) {
    if (b.mask && !b.mask[e]) return;
    load_env(x, b, c, e, false);
    reset_env<KIND>(x, c, c.env_id_offset + e);
    store_env(x, b, c, e);
}

/* hrl_observe for one env: the observation of the record as it stands (calc_state + the task's sensors, as after a teleport:
 * ant_maze_bullet_env.py:117-121); nothing but `obs` is written */
template <int KIND, class X>
HRL_DEV void observe_entry(X &x, const DevBufs &b, const DevCfg &c, int e) {
    if (b.mask && !b.mask[e]) return;
    WaveLds &L = x.lds();
    load_env(x, b, c, e, false);
    compute_obs<KIND>(x, c, c.env_id_offset + e, false);
    x.each([&](int lane) {
#pragma unroll 1
        for (int o = 0; o < c.obs_dim; o += 64)
            if (o + lane < c.obs_dim) b.obs[(size_t)e * c.obs_dim + (o + lane)] = L.obs[o + lane];
    });
}

/* hrl_set_goals / hrl_next_target for one env: `env.goals = [...]` (n_goals > 0) and `env.next_target()` of a
 * manual_goal_creation flagrun env (ant_flagrun_env.py:45,112-120).  The list is kept in list order behind the current goal and the
 * path-reward state (items[HRL_FLAG_PENDING_OFF + 2k..] = goals[k]); next_target() takes the LAST one (`self.goals.pop()`, :116), or with max_targets < 1 draws a
 * goal near the robot whatever the list holds (:113-114); _rewarded is cleared (:118), the potential is left alone (:119
 * re-reads the stale walk_target_dist), the observation is calc_state towards the new goal (:120).  ok[e] = 0 where the
 * reference raises IndexError (empty list): nothing but the observation is written then. */
template <class X>
HRL_DEV void set_goals_entry(X &x, const DevBufs &b, const DevCfg &c, int e, const float *goals_xy, int n_goals, uint8_t *ok) {
    if (b.mask && !b.mask[e]) return;
    WaveLds &L = x.lds();
    const long long env = c.env_id_offset + e;
    load_env(x, b, c, e, false);
    if (n_goals > 0) {
        x.each([&](int lane) {
#pragma unroll 1
            for (int w = lane; w < c.items_stride; w += 64)
                if (w >= HRL_FLAG_PENDING_OFF) { /* items word w: pending slot (w - HRL_FLAG_PENDING_OFF) >> 1 */
                    const int k = (w - HRL_FLAG_PENDING_OFF) >> 1, comp = w & 1;
                    L.items[w] = k < n_goals ? goals_xy[((size_t)e * n_goals + k) * 2 + comp] : 0.f;
                }
            if (lane == 63) L.aux[3] = (int)(((uint32_t)n_goals & 0xffffu) | ((uint32_t)L.aux[3] & 0xffff0000u));
        });
    }
    x.each([&](int lane) { /* next_target(): one lane reads and writes the goal words */
        if (lane != 0) return;
        const uint32_t a3 = (uint32_t)L.aux[3];
        uint32_t cur = a3 & 0xffffu;
        int good = 1;
        float gx = L.items[0], gy = L.items[1];
        if (c.flag_mtd > 0.f) {
            cur = (cur + 1u) & 0xffffu;
            flag_close_goal(c, env, (uint32_t)L.aux[2], cur, L.st[0], L.st[1], &gx, &gy);
        } else if (!c.flag_manual) { /* the shared list of a non-manual env (:91-96): goal k of the episode is a function of (seed, episode, k),
                                        `cur` of its flag_max_targets goals are used up; popping one more is cur + 1 */
            if (cur >= (uint32_t)c.flag_max_targets) good = 0; else cur += 1u;
        } else if (cur == 0u) good = 0;
        else { cur -= 1u; gx = L.items[HRL_FLAG_PENDING_OFF + 2 * cur]; gy = L.items[HRL_FLAG_PENDING_OFF + 1 + 2 * cur]; }
        if (good) {
            if (!c.flag_manual && !(c.flag_mtd > 0.f)) flag_goal(c, (uint32_t)L.aux[2], cur, &gx, &gy); /* the shared list's next goal, for set_target's bookkeeping */
            else { L.items[0] = gx; L.items[1] = gy; }
            if (c.flag_path_on) flag_set_target_state(L, gx, gy, L.st[0], L.st[1]); /* set_target (:98-103) with the robot where it is */
            L.aux[3] = (int)(cur | (a3 & 0x7fff0000u));
        }
        if (ok) ok[e] = (uint8_t)good;
    });
    compute_obs<5>(x, c, env, false);
    store_env(x, b, c, e);
}

/* hrl_step for one env (one wave; the waves of a workgroup form a group, see ant_group_block).  e >= n_envs: a wave of the
 * last group without an env of its own, which only takes part in the group's barriers. */
template <int KIND, class X>
HRL_DEV void step_entry(X &x, const DevBufs &b, const DevCfg &c, int e) {
    WaveLds &L = x.lds();
    const bool on = e < c.n_envs;
    if (on) {
        load_env(x, b, c, e, true);
        x.each([&](int lane) {
            if (lane < 16) {
                L.q[0][lane] = lane < 15 ? L.st[lane] : 0.f;
                /* qvel (v, omega, joint rates) -> u (omega, v, joint rates) */
                const int src = lane < 3 ? 15 + 3 + lane : (lane < 6 ? 15 + lane - 3 : 15 + lane);
                L.u[lane] = lane < 14 ? L.st[src] : 0.f;
            }
            if (lane >= 16 && lane < 24) {
                const int j = lane - 16;
                if (KIND == 3) { /* point_bot.py:28-31: a / |a| * 500 N in the world xy plane */
                    const float n = sqrtf(L.act[0] * L.act[0] + L.act[1] * L.act[1]);
                    L.tau[j] = j < 2 ? L.act[j] / n * c.point_force : 0.f;
                } else L.tau[j] = c.torque_scale * clampf(L.act[j], -1.f, 1.f);
            }
        });
    }
    x.stamp(12);
    int qi = 0, n_contacts = 0; /* n_contacts: the contacts of the step's last collision pass (contact-based pickup) */
    int rows_acc = 0;           /* solver rows over the step's substeps (wave-uniform; the optional diagnostic output `solver_rows`) */
    const bool items_on = (KIND == 1 || KIND == 3) && c.item_collision != 0;
    HRL_PIN_INT(qi);
    const int slot = x.slot();
    if constexpr (KIND == 3) {
#pragma unroll 1
        for (int s = 0; s < c.nsub; ++s) { /* one copy of the substep body: it is the kernel's instruction-cache footprint */
            x.priority(slot + s); /* scheduling only, no effect on results: see GpuExec::priority */
            n_contacts = point_substep(x, c, qi, items_on);
            rows_acc += 3 * n_contacts;
            qi ^= 1;
            HRL_PIN_INT(qi); /* keep the ping-pong index a run-time value so the body is not cloned per parity */
        }
    } else {
        x.each([&](int lane) { if (lane == 0) L.on = on ? 1 : 0; });
        if (!on) /* a record of a ragged last group that holds no env: the packed phases of the leader and of the contact waves compute it along
                    (results never used), so it is PARKED -- at rest high above the ground, far from every wall, box and cube, joints in the range
                    in which no legs can meet and away from their limits: no pass beyond the ones every record gets is ever run for it, and what
                    the group costs does not depend on what LDS held before */
            x.each([&](int lane) {
                if (lane < 16) {
                    const float v = lane == 2 ? 1e3f : (lane == 6 ? 1.f : 0.f);
                    L.q[0][lane] = v; L.q[1][lane] = v; L.u[lane] = 0.f;
                }
                if (lane < 8) { L.tau[lane] = 0.f; L.jlim[0][lane] = -1e3f; L.jlim[1][lane] = 1e3f; }
                L.items[lane] = 1e6f; L.items[64 + lane] = 1e6f;
                if (lane >= 16 && lane < 32) L.planes[(lane - 16) >> 2][lane & 3] = (lane & 3) == 2 ? 1.f : ((lane & 3) == 3 ? -1e6f : 0.f);
            });
        x.group_sync(); /* every record of the group is loaded before another wave reads it */
#pragma unroll 1
        for (int s = 0; s < c.nsub; ++s) { /* substep s: dynamics (leader) next to contacts (the other waves) | rows, solve, integrate */
            x.priority(slot + s); /* scheduling only, no effect on results: see GpuExec::priority */
            HRL_PIN_VGPR(qi); /* run-time value: one copy of the bodies for both parities (a vector register: the compiler does not
                                 know that the waves of a group agree on it) */
#ifndef HRL_ABLATE_GROUP /* timing-ablation builds only (tools/variants.py): never defined in the product */
            ant_group_block(x, c, qi);
#endif
#ifndef HRL_ABLATE_CONTACTS
            ant_contact_duty(x, c, qi, items_on);
#endif
            x.stamp(16);
            x.group_sync();
            x.stamp(17); /* waiting for the slower of the leader and the contact waves */
#ifndef HRL_ABLATE_ENV
            if (on) rows_acc += ant_env_block(x, c, qi);
#endif
            x.group_sync();
            x.stamp(19); /* waiting for the slowest env block of the group */
            qi ^= 1;
        }
        if (on) n_contacts = x.uniform(L.nC);
    }
    if (!on) return;
    int eo = e; /* the env index again, opaque: the output addresses and the 64-bit global id (the key of the epilogue's random draws) are
                   formed here, not kept in registers since the loads (the point kernel, at its register cap, spilled the id around the loop) */
    HRL_PIN_VGPR(eo);
    const long long env = c.env_id_offset + eo;
    x.each([&](int lane) { /* back to the packed record */
        if (lane < 15) L.st[lane] = L.q[qi][lane];
        if (lane >= 16 && lane < 30) {
            const int k = lane - 16; /* qvel index */
            L.st[15 + k] = k < 3 ? L.u[3 + k] : (k < 6 ? L.u[k - 3] : L.u[k]);
        }
    });
    x.stamp(13);
    compute_obs<KIND>(x, c, env, true, n_contacts);
    x.stamp(14);
    /* reward / done (uniform values, every lane computes them; lane-selected stores) */
    x.each([&](int lane) {
        float rew = 0.f, food = 0.f, dead = 0.f;
        int done = 0;
        if (KIND == 1 || KIND == 3) { /* ant_gather_env.py:99-119 */
            for (int k = 0; k < c.n_food + c.n_poison; ++k) food += L.irew[k];
            float alive = 1.f;
            if (KIND == 1) alive = (L.obs[0] + L.st[30] > 0.26f) ? 1.f : -1.f;
            done = (alive < 0.f) || L.flags[0];
            dead = alive < 0.f ? c.dying_cost : 0.f;
            rew = food + dead;
        } else if (KIND == 0) { /* MjAnt.py:36-97 */
            const float alive = L.st[2] > 0.26f ? 1.f : -1.f;
            done = (alive < 0.f) || L.flags[0];
            const float pot = -L.scal[3] / c.dt, progress = pot - L.st[31];
            rew = ((alive + progress) + -0.1f * L.scal[5]) + 0.f;
            food = alive; dead = progress; /* info[0..1] of the locomotion kinds: the first two entries of `self.rewards` (MjAnt.py:82-84) */
            L.red[0] = pot;
        } else if (KIND == 5) { /* upstream WalkerBaseBulletEnv.step with zeroed cost weights (ant_flagrun_env.py:133-135),
                                   then ant_flagrun_env.py:162-204: goal reward, retarget on reach / timeout, out of goals */
            const float alive = (L.s28[0] + L.st[30] > 0.26f) ? 1.f : -1.f;
            int idone = alive < 0.f;
            for (int i = 0; i < 28; ++i) if (!isfinite(L.s28[i])) idone = 1;
            const float wtd = L.scal[3], pot = -wtd / c.dt, progress = pot - L.st[31];
            L.red[0] = pot;
            int steps = ((L.aux[3] >> 16) & 0x7fff) + 1, rewarded = (L.aux[3] >> 31) & 1, cur = L.aux[3] & 0xffff, retarget = 0;
            /* goals left: the shared list has flag_max_targets of them, max_target_dist mode never runs out (:111-112), manual
             * list mode counts its pending goals in `cur` (downwards) */
            const bool close = c.flag_mtd > 0.f, listed = c.flag_manual && !close; /* max_targets < 1: goals near the robot whoever made the env (:113-114) */
            const int step_cur = listed ? -1 : 1;
            auto more = [&]() { return close ? true : (listed ? cur > 0 : cur < c.flag_max_targets); };
            float e1 = 0.f, e2 = 0.f; /* upstream WalkerBaseBulletEnv.step: the reward of super().step() with the cost weights reset() set (0, 0, 0 unless the caller set others) */
            for (int j = 0; j < NJ; ++j) { const float a = L.act[j]; e1 += fabsf(a * L.s28[9 + 2 * j]); e2 += a * a; }
            const float electricity = c.w_elec * (e1 / NJ) + c.w_stall * (e2 / NJ);
            rew = ((((alive + progress) + electricity) + c.w_jal * L.scal[5]) + 0.f) * c.flag_w_env; /* r *= ant_env_rew_weight (:169) */
            food = alive; dead = progress;
            if (c.flag_path_on) { /* :174-176: how far along the straight line from where the goal was received to the goal, over the squared distance then
                                     (the record holds both; 0 / 0 before a manual env got its first goal: NaN, as in the reference) */
                const float gsx = L.items[HRL_FLAG_START_OFF], gsy = L.items[HRL_FLAG_START_OFF + 1];
                const float path = ((L.st[0] - gsx) * (L.red[4] - gsx) + (L.st[1] - gsy) * (L.red[5] - gsy)) / L.items[HRL_FLAG_SQDIST_OFF];
                rew = rew + path * c.flag_w_path;
            }
            rew = rew + (-wtd) * c.flag_w_dist; /* :178 */
            done = idone;
            if (wtd < c.tol) {
                if (!rewarded) { rew += c.flag_goal_rew; rewarded = 1; }
                if (c.flag_switch) {
                    if (more()) { cur += step_cur; rewarded = 0; steps = 0; retarget = 1; } else done = 1;
                }
            }
            if (c.flag_timeout > 0 && c.flag_timeout <= steps) {
                if (more()) { cur += step_cur; rewarded = 0; steps = 0; retarget = 1; } else done = 1;
            }
            if (steps > 0x7fff) steps = 0x7fff; /* saturates in its 15-bit field (only reachable with the timeout off) */
            L.flags[4] = (int)(((uint32_t)cur & 0xffffu) | ((uint32_t)steps << 16) | ((uint32_t)rewarded << 31));
            L.flags[3] = retarget;
        } else if (KIND == 4) { /* MjAnt.py:36-97, then ant_maze_mj_env.py:66-78 */
            const float alive = L.st[2] > 0.26f ? 1.f : -1.f;
            int idone = alive < 0.f;
            for (int i = 0; i < 29; ++i) if (!isfinite(L.st[i])) idone = 1;
            const float wtd = L.scal[3], pot = -wtd / c.dt, progress = pot - L.st[31];
            const float inner = ((alive + progress) + -0.1f * L.scal[5]) + 0.f;
            food = alive; dead = progress;
            rew = inner * c.inner_rew_weight;
            done = idone;
            if (wtd < c.tol) { rew += 1.f; done = 1; }
            L.red[0] = pot;
        } else { /* upstream WalkerBaseBulletEnv.step, then ant_maze_bullet_env.py:84-95 */
            const float alive = (L.s28[0] + L.st[30] > 0.26f) ? 1.f : -1.f;
            int idone = alive < 0.f;
            for (int i = 0; i < 28; ++i) if (!isfinite(L.s28[i])) idone = 1;
            const float wtd = L.scal[3], pot = -wtd / c.dt, progress = pot - L.st[31];
            float e1 = 0.f, e2 = 0.f;
            for (int j = 0; j < NJ; ++j) { const float a = L.act[j]; e1 += fabsf(a * L.s28[9 + 2 * j]); e2 += a * a; }
            const float electricity = c.w_elec * (e1 / NJ) + c.w_stall * (e2 / NJ);
            const float inner = (((alive + progress) + electricity) + c.w_jal * L.scal[5]) + 0.f;
            food = alive; dead = progress;
            rew = inner * c.inner_rew_weight;
            done = idone;
            const int t = L.aux[0] + 1;
            if (wtd < c.tol) { if (c.done_at_target || (!c.done_at_target && t == c.max_steps - 1)) { rew += 1.f; done = 1; } }
            if (t == c.max_steps - 1) done = 1;
            if (c.targ_dist_rew && done) rew -= wtd;
            L.red[0] = pot;
        }
        const int t_ep = L.aux[0] + 1;
        int trunc = 0; /* gym TimeLimit (__init__.py:15): done, and info['TimeLimit.truncated'] = not done */
        if (c.max_episode_steps > 0 && t_ep >= c.max_episode_steps) { trunc = !done; done = 1; }
        L.scal[0] = rew; L.scal[1] = food; L.scal[2] = dead; L.flags[1] = done; L.flags[2] = trunc;
        L.red[1] = L.st[29] + rew; L.red[2] = (float)t_ep;
    });
    x.each([&](int lane) { /* each LDS word below is read and written by one lane only */
        if (lane == 0) {
            L.aux[0] = L.aux[0] + 1; b.reward[eo] = L.scal[0]; b.done[eo] = (uint8_t)L.flags[1];
            if (b.truncated) b.truncated[eo] = (uint8_t)L.flags[2];
            if (b.rows) b.rows[eo] += rows_acc;
        }
        if (lane == 1) {
            int a1 = L.aux[1] + 1;
            if (KIND == 2 || KIND == 5) { /* the kinds whose observation carries feet contacts: the step's flags (feet against the floor in its last collision pass,
                                             upstream WalkerBaseBulletEnv.step AFTER calc_state) ride in the top four bits for the next step's observation */
                int bits = 0;
#pragma unroll
                for (int l = 0; l < 4; ++l) bits |= ((L.gtouch[2 + 3 * l] | L.gtouch[3 + 3 * l]) & 1) << l;
                a1 = (a1 & 0x0fffffff) | (bits << 28);
            }
            L.aux[1] = a1;
        }
        if (lane == 2 && (KIND == 0 || KIND == 2 || KIND == 4 || KIND == 5)) L.st[31] = L.red[0];
        if (lane == 3) L.st[29] = L.red[1];
        if (lane == 8 && KIND == 5) L.aux[3] = L.flags[4];
        if (lane == 9 && KIND == 5 && L.flags[3]) { /* next_target() (ant_flagrun_env.py:110-118) -> set_target(new goal) with the robot where it is */
            float gx, gy;
            if (c.flag_mtd > 0.f) flag_close_goal(c, env, (uint32_t)L.aux[2], (uint32_t)L.flags[4] & 0xffffu, L.st[0], L.st[1], &gx, &gy); /* create_close_target around the robot's xy */
            else if (c.flag_manual) { /* goals.pop() (:116): the last goal of the pending list */
                const int top = L.flags[4] & 0xffff;
                gx = L.items[HRL_FLAG_PENDING_OFF + 2 * top]; gy = L.items[HRL_FLAG_PENDING_OFF + 1 + 2 * top];
            } else flag_goal(c, (uint32_t)L.aux[2], (uint32_t)L.flags[4] & 0xffffu, &gx, &gy); /* the shared list's next goal */
            if (c.flag_mtd > 0.f || c.flag_manual) { L.items[0] = gx; L.items[1] = gy; }
            if (c.flag_path_on) flag_set_target_state(L, gx, gy, L.st[0], L.st[1]);
        }
        if (lane >= 4 && lane < 8) {
            const int k = lane - 4;
            b.info[(size_t)eo * 4 + k] = k == 0 ? L.scal[1] : (k == 1 ? L.scal[2] : (k == 2 ? L.red[1] : L.red[2]));
        }
    });
    const int done_u = x.uniform(L.flags[1]);
    if constexpr (KIND == 5) { /* ant_flagrun_env.py:110-118: the returned state is calc_state() w.r.t. the NEW goal */
        const int retarget_u = x.uniform(L.flags[3]);
        if (retarget_u) compute_obs<KIND>(x, c, env, true);
        if (b.goal) /* `info['target'] = self.goal` of the steps that switched goals (:191,199): the goal this step's state looks at (phase_calc_state) */
            x.each([&](int lane) {
                if (lane < HRL_GOAL_STRIDE)
                    b.goal[(size_t)eo * HRL_GOAL_STRIDE + lane] = lane == 0 ? L.red[4] : (lane == 1 ? L.red[5] : (lane == 2 ? (retarget_u ? 1.f : 0.f) : (float)((L.aux[3] >> 16) & 0x7fff)));
            });
    }
    if (done_u && b.final_obs) /* the terminal observation (ant_gather_env.py:96,118-119): a reset below replaces L.obs by the next episode's first */
        x.each([&](int lane) {
#pragma unroll 1
            for (int o = 0; o < c.obs_dim; o += 64)
                if (o + lane < c.obs_dim) b.final_obs[(size_t)eo * c.obs_dim + (o + lane)] = L.obs[o + lane];
        });
    if (done_u && c.auto_reset) reset_env<KIND>(x, c, env);
    store_env(x, b, c, eo);
    x.stamp(15);
    x.flush_stamps(b);
}

/* run-time kind -> compile-time KIND (each instantiation only contains its own env's code) */
template <class X>
HRL_DEV void step_dispatch(X &x, const DevBufs &b, const DevCfg &c, int e) {
    switch (c.kind) {
        case 0: step_entry<0>(x, b, c, e); break;
        case 1: step_entry<1>(x, b, c, e); break;
        case 2: step_entry<2>(x, b, c, e); break;
        case 3: step_entry<3>(x, b, c, e); break;
        case 4: step_entry<4>(x, b, c, e); break;
        default: step_entry<5>(x, b, c, e); break;
    }
}
template <class X>
HRL_DEV void observe_dispatch(X &x, const DevBufs &b, const DevCfg &c, int e) {
    switch (c.kind) {
        case 0: observe_entry<0>(x, b, c, e); break;
        case 1: observe_entry<1>(x, b, c, e); break;
        case 2: observe_entry<2>(x, b, c, e); break;
        case 3: observe_entry<3>(x, b, c, e); break;
        case 4: observe_entry<4>(x, b, c, e); break;
        default: observe_entry<5>(x, b, c, e); break;
    }
}
template <class X>
HRL_DEV void reset_dispatch(X &x, const DevBufs &b, const DevCfg &c, int e) {
    switch (c.kind) {
        case 0: reset_entry<0>(x, b, c, e); break;
        case 1: reset_entry<1>(x, b, c, e); break;
        case 2: reset_entry<2>(x, b, c, e); break;
        case 3: reset_entry<3>(x, b, c, e); break;
        case 4: reset_entry<4>(x, b, c, e); break;
        default: reset_entry<5>(x, b, c, e); break;
    }
}

}  // namespace hrl
