/*
 * orc_impl.h -- CPU ORACLE (test infrastructure, NOT product code) for the hrl_pybullet_envs hot path.
 *
 * Included twice by hrl_oracle.c: REAL=double/SUF=_f64 and REAL=float/SUF=_f32.
 *
 * Two halves:
 *   (1) TASK LOGIC -- a line-by-line restatement of the reference's in-tree Python; each function cites the
 *       reference file:line it follows.  Pinned by tests/golden/NAME.json (generated from the reference itself).
 *   (2) RIGID-BODY STEP -- the reference delegates this to the third-party `pybullet` wheel (requirements.txt:1,
 *       `pybullet>=3.0.0`, unpinned; call sites ant_gather_env.py:77-80), whose source is NOT under
 *       /root/reference and which is not installed.  "PARITY UNPINNED": this half restates the published
 *       algorithm family Bullet uses (Featherstone articulated-body algorithm + sequential-impulse / projected
 *       Gauss-Seidel contact and joint-limit rows + semi-implicit Euler) as the build's own
 *       specification.  It is the spec the HIP kernels are checked against, and it is
 *       checked itself by self-consistency known-answer tests (tests/test_oracle_physics.py).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define R_(x) ((REAL)(x))
/* Fused multiply-adds are explicit (and the file is compiled with -ffp-contract=off): the rounding of every
 * operation of the rigid-body step is part of its specification. */
#if REAL_IS_FLOAT
#define FMA_(a, b, c) __builtin_fmaf(a, b, c)
#else
#define FMA_(a, b, c) __builtin_fma(a, b, c)
#endif

/* ---------------------------------------------------------------------------------------------- small vector helpers */
static inline void FN(v3set)(REAL *o, REAL x, REAL y, REAL z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void FN(v3cross)(REAL *o, const REAL *a, const REAL *b) {
    REAL x = FMA_(a[1], b[2], -(a[2] * b[1])), y = FMA_(a[2], b[0], -(a[0] * b[2])), z = FMA_(a[0], b[1], -(a[1] * b[0]));
    o[0] = x; o[1] = y; o[2] = z;
}
static inline REAL FN(v3dot)(const REAL *a, const REAL *b) { return FMA_(a[2], b[2], FMA_(a[1], b[1], a[0] * b[0])); }
static inline REAL FN(dot6)(const REAL *a, const REAL *b) {
    return FMA_(a[5], b[5], FMA_(a[4], b[4], FMA_(a[3], b[3], FMA_(a[2], b[2], FMA_(a[1], b[1], a[0] * b[0])))));
}
#if REAL_IS_FLOAT
#define RSQRT(x) sqrtf(x)
/* fp32 instantiation: sin / cos / atan2 / asin follow the operation-by-operation specification of DESIGN.md 3.7 (defined
 * below), so that the fp32 oracle and the device agree bit for bit on observations too; the fp64 instantiation, which the
 * golden vectors pin against the reference's numpy results, uses libm */
#define RSIN(x) orc_sin_spec_f32(x)
#define RCOS(x) orc_cos_spec_f32(x)
#define RATAN2(y, x) orc_atan2_spec_f32(y, x)
#define RASIN(x) orc_asin_spec_f32(x)
#define RFABS(x) fabsf(x)
#define RFMOD(x, y) fmodf(x, y)
#else
#define RSQRT(x) sqrt(x)
#define RSIN(x) sin(x)
#define RCOS(x) cos(x)
#define RATAN2(y, x) atan2(y, x)
#define RASIN(x) asin(x)
#define RFABS(x) fabs(x)
#define RFMOD(x, y) fmod(x, y)
#endif
/* sin/cos of the dynamics.  The fp32 instantiation follows the operation-by-operation specification of DESIGN.md 3.7
 * (so that two fp32 implementations agree bit for bit); the fp64 instantiation uses libm. */
static inline void FN(dyn_sincos)(REAL x, REAL *sn, REAL *cs) {
#if REAL_IS_FLOAT
    float k = rintf(x * 0.636619772367581343f);
    float r = FMA_(-k, 1.5703125f, x);
    r = FMA_(-k, 4.837512969970703125e-4f, r);
    r = FMA_(-k, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = FMA_(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = FMA_(ps, z, -1.6666654611e-1f);
    float sr = FMA_(ps * z, r, r);
    float pc = FMA_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = FMA_(pc, z, 4.166664568298827e-2f);
    float cr = FMA_(pc * z, z, FMA_(-0.5f, z, 1.0f));
    /* quadrant; out of int range (exploded states, NaN) the conversion would be implementation-defined: quadrant 0 then */
    int q = ((int)(fabsf(k) < 1e9f ? k : 0.f)) & 3;
    float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
#else
    *sn = sin(x); *cs = cos(x);
#endif
}
#if REAL_IS_FLOAT
static inline float orc_sin_spec_f32(float x) { float s, c; FN(dyn_sincos)(x, &s, &c); return s; }
static inline float orc_cos_spec_f32(float x) { float s, c; FN(dyn_sincos)(x, &s, &c); return c; }
/* atan2: a = min(|x|,|y|) / max(|x|,|y|) in [0, 1]; above tan(pi/8) reduced once more by atan(a) = pi/4 + atan((a-1)/(a+1));
 * degree-9 odd minimax polynomial on [-tan(pi/8), tan(pi/8)] (the classic single-precision coefficients); then the octant,
 * half-plane and sign.  Max error 2.8e-7 rad.  atan2(0, 0) = 0. */
static inline float orc_atan2_spec_f32(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax < ay ? ax : ay;
    float a = mx == 0.0f ? 0.0f : mn / mx;
    int big = a > 0.4142135679721832275390625f;
    float t = big ? (a - 1.0f) / (a + 1.0f) : a;
    float z = t * t;
    float p = FMA_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = FMA_(p, z, 1.99777106478e-1f);
    p = FMA_(p, z, -3.33329491539e-1f);
    float r = FMA_(p * z, t, t);
    if (big) r = 0.785398185253143310546875f + r;
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (x < 0.0f) r = 3.1415927410125732421875f - r;
    return y < 0.0f ? -r : r;
}
static inline float orc_asin_spec_f32(float x) { return orc_atan2_spec_f32(x, sqrtf((1.0f - x) * (1.0f + x))); }
#endif
static inline REAL FN(clampr)(REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* Solver clamp = median of three with the semantics of gfx950's v_med3_f32: any NaN operand -> minimum
 * of the non-NaN operands; zeros ordered -0 < +0. */
static inline int FN(zlt)(REAL a, REAL b) { return a < b || (a == b && signbit(a) && !signbit(b)); }
static inline REAL FN(zmin)(REAL a, REAL b) { return isnan(a) ? b : (isnan(b) ? a : (FN(zlt)(b, a) ? b : a)); }
static inline REAL FN(zmax)(REAL a, REAL b) { return isnan(a) ? b : (isnan(b) ? a : (FN(zlt)(a, b) ? b : a)); }
static inline REAL FN(med3)(REAL x, REAL lo, REAL hi) {
    if (isnan(x) || isnan(lo) || isnan(hi)) return FN(zmin)(FN(zmin)(x, lo), hi);
    REAL m3 = FN(zmax)(FN(zmax)(x, lo), hi);
    if (m3 == x && signbit(m3) == signbit(x)) return FN(zmax)(lo, hi);
    if (m3 == lo && signbit(m3) == signbit(lo)) return FN(zmax)(x, hi);
    return FN(zmax)(x, lo);
}
/* hrl_config stores angles as float; the reference kwargs are python floats (np.pi, 2*np.pi): promote exact matches */
static inline REAL FN(cfg_angle)(float v) {
    if (v == (float)3.14159265358979323846) return R_(3.14159265358979323846);
    if (v == (float)6.28318530717958647692) return R_(6.28318530717958647692);
    return R_(v);
}

/* =================================================================================================================
 * PART 1 -- TASK LOGIC (restates in-tree reference Python)
 * ================================================================================================================= */

/* envs/intersection_utils.py:84-90 `_find_intersection`: infinite line p1p2 x infinite line p3p4; d == 0 -> none. */
int FN(orc_inf_intersection)(const REAL *p, REAL *out) {
    REAL x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], x3 = p[4], y3 = p[5], x4 = p[6], y4 = p[7];
    REAL d = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
    if (d == 0) return 0;
    out[0] = ((x1 * y2 - y1 * x2) * (x3 - x4) - (x1 - x2) * (x3 * y4 - y3 * x4)) / d;
    out[1] = ((x1 * y2 - y1 * x2) * (y3 - y4) - (y1 - y2) * (x3 * y4 - y3 * x4)) / d;
    return 1;
}

/* envs/intersection_utils.py:93-104 `quadrant`: first match in the order 1,4,2,3 with >=/<= tests. */
int FN(orc_quadrant)(REAL x, REAL y) {
    if (x >= 0 && y >= 0) return 1;
    if (x >= 0 && y <= 0) return 4;
    if (x <= 0 && y >= 0) return 2;
    if (x <= 0 && y <= 0) return 3;
    return -1; /* NaN input: the reference raises (intersection_utils.py:104) */
}

/* envs/intersection_utils.py:14-36 `_on_segment`, `_orientation` */
static int FN(on_segment)(const REAL *p, const REAL *q, const REAL *r) {
    REAL mxx = p[0] > r[0] ? p[0] : r[0], mnx = p[0] < r[0] ? p[0] : r[0];
    REAL mxy = p[1] > r[1] ? p[1] : r[1], mny = p[1] < r[1] ? p[1] : r[1];
    return (q[0] <= mxx) && (q[0] >= mnx) && (q[1] <= mxy) && (q[1] >= mny);
}
static int FN(orientation)(const REAL *p, const REAL *q, const REAL *r) {
    REAL val = ((q[1] - p[1]) * (r[0] - q[0])) - ((q[0] - p[0]) * (r[1] - q[1]));
    return val > 0 ? 1 : (val < 0 ? 2 : 0);
}
/* envs/intersection_utils.py:39-71 `segment_intersection` */
int FN(orc_segment_intersection)(const REAL *p) {
    const REAL *p1 = p, *q1 = p + 2, *p2 = p + 4, *q2 = p + 6;
    int o1 = FN(orientation)(p1, q1, p2), o2 = FN(orientation)(p1, q1, q2);
    int o3 = FN(orientation)(p2, q2, p1), o4 = FN(orientation)(p2, q2, q1);
    if (o1 != o2 && o3 != o4) return 1;
    if (o1 == 0 && FN(on_segment)(p1, p2, q1)) return 1;
    if (o2 == 0 && FN(on_segment)(p1, q2, q1)) return 1;
    if (o3 == 0 && FN(on_segment)(p2, p1, q2)) return 1;
    if (o4 == 0 && FN(on_segment)(p2, q1, q2)) return 1;
    return 0;
}

/* envs/sizeable_enclosed_scene.py:63-97 `sense_walls` (+ intersection_utils.py:113-116 pol2cart).
 * lines: [n_lines][4] = (ax,ay,bx,by) in the order of Scene.bounds (world_bounds then box_bounds). */
void FN(orc_sense_walls)(int bins, REAL span, REAL range, const REAL *pos, REAL yaw, const REAL *lines, int n_lines,
                         int span_is_2pi, REAL *out) {
    const REAL half_pi = R_(1.5707963267948966);
    for (int i = 0; i < bins; ++i) {
        REAL phi; /* :68-71 -- the == 2*pi special case is decided by the caller on the exact kwarg value */
        if (span_is_2pi) phi = half_pi + yaw + (R_(i + 1) / R_(bins)) * span;
        else phi = half_pi + yaw + (R_(i) / R_(bins - 1)) * span;
        REAL sv[2] = {pos[0] + range * RCOS(phi), pos[1] + range * RSIN(phi)}; /* :72 */
        int sq = FN(orc_quadrant)(sv[0] - pos[0], sv[1] - pos[1]);              /* :74 */
        REAL best = 0;
        for (int l = 0; l < n_lines; ++l) { /* :79 */
            REAL p[8] = {pos[0], pos[1], sv[0], sv[1], lines[4 * l], lines[4 * l + 1], lines[4 * l + 2], lines[4 * l + 3]};
            REAL in[2];
            if (!FN(orc_inf_intersection)(p, in)) continue; /* :81-83 */
            REAL dx = pos[0] - in[0], dy = pos[1] - in[1];
            REAL dist = RSQRT(dx * dx + dy * dy);                                 /* :85 */
            if (dist > range) continue;                                           /* :86 */
            if (sq != FN(orc_quadrant)(in[0] - pos[0], in[1] - pos[1])) continue; /* :89 */
            REAL v = R_(1.) - dist / range;                                       /* :95 */
            if (v > best) best = v;
        }
        out[i] = best;
    }
}

/* ant_gather_env.py:148-155 angle wrap: python `%` (result in [0, 2pi)) then fold to (-pi, pi]. */
static REAL FN(wrap_angle)(REAL a) {
    const REAL two_pi = R_(6.283185307179586), pi = R_(3.141592653589793);
    a = RFMOD(a, two_pi);
    if (a < 0) a += two_pi;
    if (a >= two_pi) a -= two_pi; /* python: (-tiny) % 2pi rounds to 2pi in floating point only if fmod+2pi rounds up */
    if (a > pi) a = a - two_pi;
    if (a < -pi) a = a + two_pi;
    return a;
}

/* ant_gather_env.py:198-200 / gather_base.py:189-191 `sq_dist_robot`: SQUARED planar distance. */
REAL FN(orc_sq_dist)(const REAL *item_xy, const REAL *robot_xy) {
    REAL dx = item_xy[0] - robot_xy[0], dy = item_xy[1] - robot_xy[1];
    return dx * dx + dy * dy;
}

/* ant_gather_env.py:128-177 / gather_base.py:118-168 `get_sensor_readings`.
 * The reference sorts all items by d2 descending and lets nearer items overwrite farther ones per bin; since
 * intensity = 1 - d2/range is monotone in d2 that is "per bin and type, the nearest in-range in-span item wins",
 * implemented here literally (stable sort, reverse=True keeps input order among equal keys). */
void FN(orc_food_sensor)(int n_bins, REAL span, REAL range, const REAL *robot_xy, REAL yaw, const REAL *items_xy,
                         int n_food, int n_poison, const REAL *d2, REAL *food_out, REAL *poison_out) {
    int n = n_food + n_poison, m = 0, order[HRL_MAX_ITEMS];
    for (int i = 0; i < n_bins; ++i) food_out[i] = poison_out[i] = 0;
    /* An item whose squared distance is NaN (a NaN robot or item coordinate) is a numerical failure, not a reading: the reference raises at
     * int(nan) (:161).  Here it takes no part in the ordering (a NaN key would stop the insertion sort from ordering the others) and makes
     * every reading of its type NaN, which ends the episode (:101-103). */
    for (int i = 0; i < n; ++i) if (d2[i] == d2[i]) order[m++] = i;
    for (int i = 1; i < m; ++i) { /* stable insertion sort, descending d2 (:141) */
        int k = order[i], j = i - 1;
        while (j >= 0 && d2[order[j]] < d2[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    REAL bin_res = span / R_(n_bins); /* :142 */
    REAL half_span = span * R_(0.5);  /* :158 */
    for (int i = 0; i < n; ++i)
        if (d2[i] != d2[i]) { REAL *out = i < n_food ? food_out : poison_out; for (int b = 0; b < n_bins; ++b) out[b] = (REAL)NAN; }
    for (int s = 0; s < m; ++s) {
        int k = order[s];
        if (d2[k] > range) continue;                                                               /* :145 */
        REAL angle = RATAN2(items_xy[2 * k + 1] - robot_xy[1], items_xy[2 * k] - robot_xy[0]) - yaw; /* :148 */
        angle = FN(wrap_angle)(angle);                                                             /* :151-155 */
        if (!(RFABS(angle) <= half_span)) continue; /* :159; NaN-safe (the reference raises ValueError at int(nan), :161) */
        int bin = (int)((angle + half_span) / bin_res);                                            /* :161 */
        if (bin >= n_bins) bin = n_bins - 1; /* reference would raise IndexError at angle == +half_span (measure zero) */
        REAL intensity = R_(1.0) - d2[k] / range; /* :162 */
        REAL *out = k < n_food ? food_out : poison_out;
        if (out[bin] == out[bin]) out[bin] = intensity; /* a type poisoned by a NaN distance stays NaN */
    }
}

/* ant_gather_env.py:179-196 `get_abs_pos`: nearest-first xy of min(n, n_bins) food then poison items. */
void FN(orc_abs_pos)(int n_bins, const REAL *items_xy, int n_food, int n_poison, const REAL *d2, REAL *food_out,
                     REAL *poison_out) {
    for (int t = 0; t < 2; ++t) {
        int base = t ? n_food : 0, n = t ? n_poison : n_food, order[HRL_MAX_ITEMS];
        REAL *out = t ? poison_out : food_out;
        for (int i = 0; i < n; ++i) order[i] = base + i;
        for (int i = 1; i < n; ++i) { /* stable ascending (:183-184) */
            int k = order[i], j = i - 1;
            while (j >= 0 && d2[order[j]] > d2[k]) { order[j + 1] = order[j]; --j; }
            order[j + 1] = k;
        }
        int m = n < n_bins ? n : n_bins, bad = 0;
        for (int i = 0; i < n; ++i) bad |= d2[base + i] != d2[base + i]; /* a NaN distance has no place in the order: the type's outputs are NaN */
        for (int i = 0; i < m; ++i) { out[2 * i] = bad ? (REAL)NAN : items_xy[2 * order[i]]; out[2 * i + 1] = bad ? (REAL)NAN : items_xy[2 * order[i] + 1]; }
    }
}

/* Source of the uniform pairs `rs.rand(2)` consumes (gather_scene.py:58): the golden tests replay the logged draws of
 * the reference's RandomState in order; the env-level step draws counter-based Philox pairs keyed by (item, attempt). */
#ifndef ORC_DRAW_TYPES
#define ORC_DRAW_TYPES
typedef int (*orc_draw_fn)(void *ctx, int item, int attempt, double *pair); /* 0 = no more draws */
typedef struct orc_draw_array { const void *draws; int n, used, is_float; } orc_draw_array;
static int orc_draw_from_array(void *ctx, int item, int attempt, double *pair) {
    orc_draw_array *a = (orc_draw_array *)ctx;
    (void)item; (void)attempt;
    if (a->used >= a->n) return 0;
    if (a->is_float) { pair[0] = ((const float *)a->draws)[2 * a->used]; pair[1] = ((const float *)a->draws)[2 * a->used + 1]; }
    else { pair[0] = ((const double *)a->draws)[2 * a->used]; pair[1] = ((const double *)a->draws)[2 * a->used + 1]; }
    ++a->used;
    return 1;
}
#endif

/* gather_scene.py:52-62 `_random_on_plane`: pos = rand(2)*(size-1) - (size-1)/2, redrawn while |avoid-pos| < spacing.
 * Returns the number of pairs consumed, or -1 if the source ran out; at most `max_attempts` draws, the last one kept
 * (the reference loops until it succeeds). */
static int FN(random_on_plane_src)(const REAL *world_size, const REAL *avoid_xy, REAL spacing, orc_draw_fn draw, void *ctx,
                                   int item, int max_attempts, REAL *pos_out) {
    REAL sx = world_size[0] - 1, sy = world_size[1] - 1;
    for (int k = 0; k < max_attempts; ++k) {
        double pr[2];
        if (!draw(ctx, item, k, pr)) return -1;
        REAL px = R_(pr[0]) * sx - sx / 2, py = R_(pr[1]) * sy - sy / 2;
        REAL dx = avoid_xy[0] - px, dy = avoid_xy[1] - py;
        pos_out[0] = px; pos_out[1] = py;
        if (!(RSQRT(dx * dx + dy * dy) < spacing)) return k + 1;
    }
    return max_attempts;
}
int FN(orc_random_on_plane)(const REAL *world_size, const REAL *avoid_xy, REAL spacing, const REAL *draws, int n_draws,
                            REAL *pos_out) {
    orc_draw_array a = {draws, n_draws, 0, REAL_IS_FLOAT};
    REAL tmp[2];
    int k = FN(random_on_plane_src)(world_size, avoid_xy, spacing, orc_draw_from_array, &a, 0, 1 << 30, tmp);
    if (k > 0) { pos_out[0] = tmp[0]; pos_out[1] = tmp[1]; }
    return k;
}

/* ant_maze_bullet_env.py:123-133 `get_target_vec_obs` (encoding 0 normed vector, 1 sin/cos of relative angle). */
void FN(orc_target_vec_obs)(int encoding, const REAL *target, const REAL *robot_xy, REAL yaw, REAL *out) {
    REAL vx = target[0] - robot_xy[0], vy = target[1] - robot_xy[1];
    if (encoding == 0) {
        REAL n = RSQRT(vx * vx + vy * vy);
        out[0] = vx / n; out[1] = vy / n;
    } else {
        REAL a = RATAN2(vy, vx) - yaw;
        out[0] = RSIN(a); out[1] = RCOS(a);
    }
}

/* ant_maze_bullet_env.py:135-178 `get_target_sensor_obs`: box occlusion via segment_intersection, then one bin. */
void FN(orc_target_sensor_obs)(int n_bins, REAL span, REAL range, const REAL *target, const REAL *robot_xy, REAL yaw,
                               REAL walk_target_dist, const REAL *box_lines, int n_box_lines, REAL *out) {
    for (int i = 0; i < n_bins; ++i) out[i] = 0;
    if (n_bins <= 0) return;
    if (walk_target_dist > range) return; /* :145 */
    for (int l = 0; l < n_box_lines; ++l) { /* :148-150 */
        REAL p[8] = {robot_xy[0], robot_xy[1], target[0], target[1], box_lines[4 * l], box_lines[4 * l + 1],
                     box_lines[4 * l + 2], box_lines[4 * l + 3]};
        if (FN(orc_segment_intersection)(p)) return;
    }
    REAL angle = FN(wrap_angle)(RATAN2(target[1] - robot_xy[1], target[0] - robot_xy[0]) - yaw); /* :153-160 */
    REAL half_span = span * R_(0.5), bin_res = span / R_(n_bins);
    if (!(RFABS(angle) <= half_span)) return; /* :164, NaN-safe */
    int bin = (int)((angle + half_span) / bin_res);
    if (bin >= n_bins) bin = n_bins - 1;
    out[bin] = R_(1.0) - walk_target_dist / range; /* :167-168 */
}

/* point_bot.py:48-67 `PointBot.calc_state` (float32 array in the reference). */
void FN(orc_pointbot_state)(const REAL *xyz, const REAL *rpy, const REAL *speed, const REAL *target, REAL initial_z,
                            REAL *out) {
    REAL yaw = rpy[2];
    REAL theta = RATAN2(target[1] - xyz[1], target[0] - xyz[0]);
    REAL a = theta - yaw;
    REAL c = RCOS(-yaw), s = RSIN(-yaw);
    REAL vx = c * speed[0] - s * speed[1], vy = s * speed[0] + c * speed[1], vz = speed[2];
    out[0] = xyz[2] - initial_z; out[1] = RSIN(a); out[2] = RCOS(a);
    out[3] = R_(0.3) * vx; out[4] = R_(0.3) * vy; out[5] = R_(0.3) * vz; out[6] = rpy[0]; out[7] = rpy[1];
}

/* The task half of AntGatherBulletEnv.step / GatherBulletEnv.step given the post-physics robot state
 * (ant_gather_env.py:81-119, gather_base.py:80-109).
 *   base_state : robot.calc_state() (28 for the ant, 8 for the point bot), ant=1 drops elements 1:3 (:81)
 *   items_xy   : [n_food+n_poison][2] in/out (respawned in place); respawn draws from `draw`
 *   alive_z    : > 0 enables `alive = +1 if z > alive_z else -1` (ant); <= 0 means "can't die" (point_bot.py:73-74)
 *   contact_items : robot_coll_dist <= 0 branch (:113-116): cp[2] of every contact point the robot has, as an item
 *                   index (food slots then poison slots) or -1 for anything else (ground, walls: reward_collision
 *                   returns 0 for them, gather_scene.py:95-114); processed AFTER the observation was assembled (:95-96)
 * Returns number of uniform pairs consumed (-1: the draw source ran out). */
int FN(orc_gather_task_src)(const hrl_config *cfg, int ant, const REAL *base_state, int n_base, const REAL *torso_xyz,
                            REAL yaw, REAL initial_z, REAL alive_z, REAL *items_xy, orc_draw_fn draw, void *ctx, int max_attempts,
                            const int *contact_items, int n_contacts,
                            REAL *obs, REAL *rew, int *done, REAL *food_rew_out, REAL *dead_rew_out) {
    int nf = cfg->n_food, np_ = cfg->n_poison, n = nf + np_, no = 0, used = 0;
    REAL d2[HRL_MAX_ITEMS];
    if (ant) { obs[no++] = base_state[0]; for (int i = 3; i < n_base; ++i) obs[no++] = base_state[i]; } /* :81 */
    else for (int i = 0; i < n_base; ++i) obs[no++] = base_state[i];
    for (int i = 0; i < n; ++i) d2[i] = FN(orc_sq_dist)(items_xy + 2 * i, torso_xyz); /* :84 */
    REAL food_reward = 0;
    REAL ws[2] = {R_(cfg->world_size[0]), R_(cfg->world_size[1])};
    if (cfg->robot_coll_dist > 0) { /* :88 */
        for (int i = 0; i < n; ++i) {
            if (d2[i] < R_(cfg->robot_coll_dist)) { /* :90 -- squared distance vs linear threshold (SURVEY C-1) */
                food_reward += (i < nf) ? 1 : -1;   /* gather_scene.py:95-112 */
                if (cfg->respawn) {
                    int k = FN(random_on_plane_src)(ws, torso_xyz, R_(cfg->robot_object_spacing), draw, ctx, i, max_attempts, items_xy + 2 * i);
                    if (k < 0) return -1;
                    used += k;
                } else { items_xy[2 * i] = 100; items_xy[2 * i + 1] = 0; } /* gather_scene.py:13,100-102 */
                d2[i] = FN(orc_sq_dist)(items_xy + 2 * i, torso_xyz); /* :92 */
            }
        }
    }
    REAL fr[HRL_MAX_ITEMS * 2], pr[HRL_MAX_ITEMS * 2];
    int nfo, npo;
    if (cfg->use_sensor) { /* :121-125 */
        FN(orc_food_sensor)(cfg->n_bins, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), torso_xyz, yaw, items_xy, nf, np_,
                            d2, fr, pr);
        nfo = npo = cfg->n_bins;
    } else {
        FN(orc_abs_pos)(cfg->n_bins, items_xy, nf, np_, d2, fr, pr);
        nfo = 2 * (nf < cfg->n_bins ? nf : cfg->n_bins);
        npo = 2 * (np_ < cfg->n_bins ? np_ : cfg->n_bins);
    }
    for (int i = 0; i < nfo; ++i) obs[no++] = fr[i];
    for (int i = 0; i < npo; ++i) obs[no++] = pr[i]; /* :96 */
    REAL alive = 1;
    if (alive_z > 0) alive = (obs[0] + initial_z > alive_z) ? R_(1) : R_(-1); /* :99 + upstream Ant.alive_bonus */
    int d = alive < 0;                                                         /* :100 */
    for (int i = 0; i < no; ++i) if (!isfinite(obs[i])) d = 1;                 /* :101-103 */
    if (!(cfg->robot_coll_dist > 0)) { /* :113-116: one reward_collision() per contact point, in contact order: an item touched
                                         * by k contact points pays k times and is moved k times (gather_scene.py:95-114 does
                                         * not look at where the item is).  Move m of item i draws from the source under the
                                         * key i | m << 4 (m << 6 with more than 16 items); every move is independent of the previous one, so where the item ends
                                         * up is its LAST move (what the device computes directly). */
        int hits[HRL_MAX_ITEMS] = {0};
        for (int c = 0; c < n_contacts; ++c) {
            int i = contact_items[c];
            if (i < 0 || i >= n) continue; /* not an item: reward_collision returns 0 */
            food_reward += (i < nf) ? 1 : -1;
            if (cfg->respawn) {
                int k = FN(random_on_plane_src)(ws, torso_xyz, R_(cfg->robot_object_spacing), draw, ctx, i | (hits[i] << (n > 16 ? 6 : 4)), max_attempts, items_xy + 2 * i);
                if (k < 0) return -1;
                used += k;
            } else { items_xy[2 * i] = 100; items_xy[2 * i + 1] = 0; }
            ++hits[i];
        }
    }
    REAL dead_rew = alive < 0 ? R_(cfg->dying_cost) : 0;                       /* :118 */
    *rew = food_reward + dead_rew; *done = d; *food_rew_out = food_reward; *dead_rew_out = dead_rew; /* :119 */
    return used;
}
/* the same with the uniform pairs replayed from an array (golden tests) */
int FN(orc_gather_task)(const hrl_config *cfg, int ant, const REAL *base_state, int n_base, const REAL *torso_xyz,
                        REAL yaw, REAL initial_z, REAL alive_z, REAL *items_xy, const REAL *draws, int n_draws,
                        REAL *obs, REAL *rew, int *done, REAL *food_rew_out, REAL *dead_rew_out) {
    orc_draw_array a = {draws, n_draws, 0, REAL_IS_FLOAT};
    return FN(orc_gather_task_src)(cfg, ant, base_state, n_base, torso_xyz, yaw, initial_z, alive_z, items_xy, orc_draw_from_array, &a,
                                   1 << 30, 0, 0, obs, rew, done, food_rew_out, dead_rew_out);
}
int FN(orc_gather_task_contacts)(const hrl_config *cfg, int ant, const REAL *base_state, int n_base, const REAL *torso_xyz,
                                 REAL yaw, REAL initial_z, REAL alive_z, REAL *items_xy, const REAL *draws, int n_draws,
                                 const int *contact_items, int n_contacts,
                                 REAL *obs, REAL *rew, int *done, REAL *food_rew_out, REAL *dead_rew_out) {
    orc_draw_array a = {draws, n_draws, 0, REAL_IS_FLOAT};
    return FN(orc_gather_task_src)(cfg, ant, base_state, n_base, torso_xyz, yaw, initial_z, alive_z, items_xy, orc_draw_from_array, &a,
                                   1 << 30, contact_items, n_contacts, obs, rew, done, food_rew_out, dead_rew_out);
}

/* The task half of AntMazeBulletEnv.step given the upstream WalkerBaseBulletEnv.step result
 * (ant_maze_bullet_env.py:63-97).  lines = Scene.bounds (7 for the maze), box lines = last 3. */
void FN(orc_maze_task)(const hrl_config *cfg, const REAL *ant_obs28, REAL inner_rew, int inner_done, const REAL *torso_xy,
                       REAL yaw, const REAL *target, REAL walk_target_dist, int t_after_increment, const REAL *lines,
                       int n_lines, int n_box_lines, REAL *obs, REAL *rew, int *done) {
    int no = 0;
    obs[no++] = ant_obs28[0];
    for (int i = 3; i < 28; ++i) obs[no++] = ant_obs28[i]; /* :75 */
    if (cfg->sense_target) {
        FN(orc_target_sensor_obs)(cfg->n_bins, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), target, torso_xy, yaw,
                                  walk_target_dist, lines + 4 * (n_lines - n_box_lines), n_box_lines, obs + no);
        no += cfg->n_bins;
    } else {
        FN(orc_target_vec_obs)(cfg->target_encoding, target, torso_xy, yaw, obs + no);
        no += 2;
    }
    if (cfg->sense_walls) {
        const double two_pi = 6.283185307179586;
        FN(orc_sense_walls)(cfg->n_bins, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), torso_xy, yaw, lines, n_lines,
                            (double)cfg->sensor_span == (double)(float)two_pi, obs + no);
        no += cfg->n_bins;
    }
    REAL r = inner_rew * R_(cfg->inner_rew_weight); /* :84 */
    int d = inner_done;
    if (walk_target_dist < R_(cfg->tol)) { /* :86-89 */
        if (cfg->done_at_target || (!cfg->done_at_target && t_after_increment == cfg->max_steps - 1)) { r += 1; d = 1; }
    }
    if (t_after_increment == cfg->max_steps - 1) d = 1; /* :91-92 */
    if (cfg->targ_dist_rew && d) r -= walk_target_dist; /* :94-95 */
    *rew = r; *done = d;
}

/* MjAnt.py:36-97 `AntMjEnv.step` reward assembly: alive + progress + joints_at_limit_cost * n + 0. */
void FN(orc_antmj_reward)(const REAL *state29, REAL potential_old, REAL potential_new, int joints_at_limit,
                          REAL joints_at_limit_cost, REAL *rew, int *done) {
    REAL alive = state29[2] > R_(0.26) ? R_(1) : R_(-1); /* :27-28,44 (initial_z cancels) */
    int d = alive < 0;
    for (int i = 0; i < 29; ++i) if (!isfinite(state29[i])) d = 1; /* :46-48 */
    REAL progress = potential_new - potential_old;                   /* :50-52 */
    *rew = ((alive + progress) + joints_at_limit_cost * R_(joints_at_limit)) + 0; /* :68,82-88,97 */
    *done = d;
}

/* ant_maze_mj_env.py:57-78 `AntMazeMjEnv._get_obs` + `step` given the AntMjEnv.step result:
 * obs = [state29 | walls(n_bins) | pit zeros | moveable zeros | t*0.001] with t BEFORE its increment (:68-70);
 * rew = inner*weight (+1 and done when walk_target_dist < tol, :72-76). */
void FN(orc_maze_mj_task)(const hrl_config *cfg, const REAL *state29, REAL yaw, REAL inner_rew, int inner_done,
                          REAL walk_target_dist, int t_before, const REAL *lines, int n_lines, REAL *obs, REAL *rew, int *done) {
    int no = 0, nb = cfg->n_bins;
    for (int i = 0; i < 29; ++i) obs[no++] = state29[i];
    FN(orc_sense_walls)(nb, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), state29, yaw, lines, n_lines,
                        (double)cfg->sensor_span == (double)(float)6.283185307179586, obs + no); /* :58-59: pos = ant_obs[:2] */
    no += nb;
    for (int i = 0; i < 2 * nb; ++i) obs[no++] = 0; /* :61-62 */
    obs[no++] = R_(t_before) * R_(0.001);           /* :64 */
    REAL r = inner_rew * R_(cfg->inner_rew_weight);
    int d = inner_done;
    if (walk_target_dist < R_(cfg->tol)) { r += 1; d = 1; } /* :74-76 */
    *rew = r; *done = d;
}

/* ant_flagrun_env.py:71-78 `create_target`: g ~ U(-size/2, size/2)^2 redrawn while |g| < 0.5 (numpy uniform(lo, hi) =
 * lo + (hi - lo) * u).  Uniforms from `draws` (two per attempt); returns the number of uniforms consumed or -1. */
int FN(orc_flag_create_target)(REAL size, const REAL *draws, int n_draws, REAL *goal) {
    for (int k = 0; 2 * k + 1 < n_draws; ++k) {
        REAL gx = -size / 2 + size * draws[2 * k], gy = -size / 2 + size * draws[2 * k + 1];
        if (!(RSQRT(gx * gx + gy * gy) < R_(0.5))) { goal[0] = gx; goal[1] = gy; return 2 * k + 2; }
    }
    return -1;
}

/* ant_flagrun_env.py:80-89 `create_close_target`: per axis uniform(tol, max_target_dist / 2) * (randint(0, 2) * 2 - 1),
 * added to the robot's xy, redrawn until strictly inside (-size/2, size/2)^2.  `u` = the uniforms in [0, 1) behind the
 * uniform() calls and `b` = the randint results, two of each per attempt, in call order; returns the attempts used or -1. */
int FN(orc_flag_create_close_target)(REAL size, REAL tol, REAL mtd, const REAL *robot_xy, const REAL *u, const int *b, int n_attempts, REAL *goal) {
    REAL wb = size / 2;
    for (int k = 0; k < n_attempts; ++k) {
        REAL gx = (tol + (mtd / 2 - tol) * u[2 * k]) * R_(b[2 * k] * 2 - 1) + robot_xy[0];
        REAL gy = (tol + (mtd / 2 - tol) * u[2 * k + 1]) * R_(b[2 * k + 1] * 2 - 1) + robot_xy[1];
        if (-wb < gx && gx < wb && -wb < gy && gy < wb) { goal[0] = gx; goal[1] = gy; return k + 1; }
    }
    return -1;
}

/* ant_flagrun_env.py:174-176 `path_rew`: np.dot(body_real_xyz[:2] - _goal_start_pos, goal - _goal_start_pos) / _sq_dist_goal */
REAL FN(orc_flag_path_rew)(const REAL *robot_xy, const REAL *goal, const REAL *goal_start, REAL sq_dist_goal) {
    return ((robot_xy[0] - goal_start[0]) * (goal[0] - goal_start[0]) + (robot_xy[1] - goal_start[1]) * (goal[1] - goal_start[1])) / sq_dist_goal;
}
/* ant_flagrun_env.py:98-103 `set_target`: `_goal_start_pos` = where the robot stands, `_sq_dist_goal` = np.linalg.norm(goal - pos) ** 2 (the
 * norm squared, not the sum of squares); out3 = start x, start y, squared distance */
void FN(orc_flag_set_target_state)(const REAL *goal, const REAL *robot_xy, REAL *out3) {
    REAL dx = goal[0] - robot_xy[0], dy = goal[1] - robot_xy[1], nrm = RSQRT(dx * dx + dy * dy);
    out3[0] = robot_xy[0]; out3[1] = robot_xy[1]; out3[2] = nrm * nrm;
}
/* ant_flagrun_env.py:162-204 `step` after `super().step(a)`: the class-level reward weights (:157-160, hrl_config.flag_*: r *= ant_env_rew_weight,
 * += path_rew * path_rew_weight, += -walk_target_dist * dist_rew_weight; `with_path` = 0 leaves the path term out: envs that keep no
 * `_goal_start_pos`), goal reward, retargeting, timeout, running out of goals.
 * io: steps (steps_since_goal_change), rewarded, goals_left.  *retarget = 1 when next_target() succeeded. */
void FN(orc_flagrun_task_w)(const hrl_config *cfg, REAL inner_rew, int inner_done, REAL walk_target_dist, int with_path, REAL path_rew, int *steps,
                            int *rewarded, int *goals_left, REAL *rew, int *done, int *retarget) {
    REAL r = inner_rew * R_(cfg->flag_ant_env_rew_weight); /* :169 */
    int d = inner_done;
    *retarget = 0;
    *steps += 1; /* :171 */
    if (with_path) r = r + path_rew * R_(cfg->flag_path_rew_weight); /* :176 */
    r = r + (-walk_target_dist) * R_(cfg->flag_dist_rew_weight);     /* :178 */
    if (walk_target_dist < R_(cfg->tol)) { /* :183 */
        if (!*rewarded) { r += R_(cfg->flag_goal_reach_rew); *rewarded = 1; } /* :184-186, goal_reach_rew :160 */
        if (cfg->flag_switch_on_collision) {
            if (*goals_left > 0) { *goals_left -= 1; *rewarded = 0; *steps = 0; *retarget = 1; } /* next_target :110-118 */
            else d = 1; /* goals.pop() raises IndexError :193-194 */
        }
    }
    if (cfg->flag_timeout > 0 && cfg->flag_timeout <= *steps) { /* :196 */
        if (*goals_left > 0) { *goals_left -= 1; *rewarded = 0; *steps = 0; *retarget = 1; }
        else d = 1;
    }
    *rew = r; *done = d;
}
/* the same without a path term (the golden fixtures of the default weights call this one) */
void FN(orc_flagrun_task)(const hrl_config *cfg, REAL inner_rew, int inner_done, REAL walk_target_dist, int *steps,
                          int *rewarded, int *goals_left, REAL *rew, int *done, int *retarget) {
    FN(orc_flagrun_task_w)(cfg, inner_rew, inner_done, walk_target_dist, 0, 0, steps, rewarded, goals_left, rew, done, retarget);
}

/* =================================================================================================================
 * PART 2 -- RIGID-BODY STEP (build's own specification; PARITY UNPINNED against pybullet, see header)
 * Replaces robot.apply_action + scene.global_step at ant_gather_env.py:77-78 (upstream: setJointMotorControl2 +
 * stepSimulation with fixedTimeStep 0.0165, numSubSteps 4, 5 solver iterations -- SURVEY Appendix A.1/A.3).
 * ================================================================================================================= */

#define NJ 8
#define NBODY 9
#define NDOF 14
#define MAXC 12
#define MAXR 44

typedef struct FN(orc_consts) {
    REAL h, inv_h, g, erp_c, erp_l, mu, cdist, lmargin, vmax, limp_max, ground_z;
    REAL r_torso, r_caps, L1, L2;
    REAL m0, a0, b0;      /* composite torso: mass, inertia alpha*1 + beta*Z Z^T                         */
    REAL m1, a1, b1;      /* aux (short) capsule about its own COM: alpha*1 + beta*e e^T, e = capsule axis */
    REAL m2, a2, b2;      /* foot capsule                                                                */
    REAL lo[NJ], hi[NJ];  /* joint ranges, assets/ant.xml:18-54                                          */
    REAL mu_self;         /* friction between two ant links = friction_robot^2 (Bullet combines by product)       */
    int iters, nsub, self_collision, item_collision;
    int max_contacts, damping_on;   /* hrl_model.max_contacts; damping of the bodies switched on */
    REAL damp_lin, damp_ang;        /* hrl_model.linear_damping / angular_damping: k_l, k_a of Bullet's damping wrench on every body */
    REAL restitution, rest_thr;     /* hrl_model.restitution / restitution_threshold */
    REAL jdamp, armature;           /* hrl_model.joint_damping / joint_armature (assets/ant.xml:8; 0, 0 in the specification) */
} FN(orc_consts);

/* static collision world: ground plane + lateral half-spaces + axis-aligned boxes
 * (sizeable_enclosed_scene.py:39-61, maze_scene.py:33-38, assets/plane.xml, wall.xml, box.xml) */
typedef struct FN(orc_world) {
    int n_planes; REAL plane_n[4][3], plane_d[4]; /* inside: n.p - d > 0 */
    int n_boxes;  REAL box_lo[1][3], box_hi[1][3];
} FN(orc_world);

/* ant leg signs / ankle axes: assets/ant.xml:15-58 */
#ifndef ORC_LEG_TABLES
#define ORC_LEG_TABLES
static const int LEG_SX[4] = {1, -1, -1, 1}, LEG_SY[4] = {1, 1, -1, -1};
static const int ANK_AX[4] = {-1, 1, -1, 1}, ANK_AY[4] = {1, 1, 1, 1};
static const int LEG_SIGMA[4] = {-1, 1, 1, -1}; /* (ankle axis) x (leg direction) = sigma * z */
#endif

void FN(orc_consts_init)(const hrl_model *M, FN(orc_consts) * K) {
    const double pi = 3.14159265358979323846, s2 = 1.41421356237309504880;
    double rho = M->density, rt = 0.25, rc = 0.08, L1 = 0.2 * s2, L2 = 0.4 * s2; /* ant.xml:13,16,19,22 */
    double msph = rho * 4.0 / 3.0 * pi * rt * rt * rt, Isph = 0.4 * msph * rt * rt;
    double m[2], Ia[2], It[2], L[2] = {L1, L2};
    for (int k = 0; k < 2; ++k) { /* solid capsule = cylinder + two hemispheres */
        double mc = rho * pi * rc * rc * L[k], ms = rho * 4.0 / 3.0 * pi * rc * rc * rc;
        m[k] = mc + ms;
        Ia[k] = mc * rc * rc / 2 + ms * 0.4 * rc * rc;
        It[k] = mc * (L[k] * L[k] / 12 + rc * rc / 4) + ms * (0.4 * rc * rc + L[k] * L[k] / 4 + 3 * L[k] * rc / 8);
    }
    /* torso body = sphere + the four jointless "leg" capsules (ant.xml:15-16,26-27,37-38,48-49) merged rigidly */
    double c = L1 / 2, common = Isph + 4 * (It[0] + m[0] * c * c), dz = Ia[0] - It[0] - m[0] * c * c;
    K->m0 = R_(msph + 4 * m[0]); K->a0 = R_(common + 2 * dz); K->b0 = R_(common - (common + 2 * dz));
    K->m1 = R_(m[0]); K->a1 = R_(It[0]); K->b1 = R_(Ia[0] - It[0]);
    K->m2 = R_(m[1]); K->a2 = R_(It[1]); K->b2 = R_(Ia[1] - It[1]);
    K->r_torso = R_(rt); K->r_caps = R_(rc); K->L1 = R_(L1); K->L2 = R_(L2);
    K->h = R_(M->timestep); K->inv_h = R_(1) / K->h; K->g = R_(M->gravity); K->erp_c = R_(M->contact_erp); K->erp_l = R_(M->limit_erp);
    K->mu = R_(M->friction_ground * M->friction_robot); K->cdist = R_(M->contact_dist); K->lmargin = R_(M->limit_margin);
    K->vmax = R_(M->max_joint_vel); K->limp_max = R_(M->limit_max_impulse); K->ground_z = R_(M->ground_z);
    K->iters = M->solver_iters; K->nsub = M->frame_skip;
    K->mu_self = R_(M->friction_robot * M->friction_robot); K->self_collision = M->self_collision; K->item_collision = M->item_collision;
    K->max_contacts = M->max_contacts; K->damping_on = M->linear_damping != 0 || M->angular_damping != 0;
    K->damp_lin = R_(M->linear_damping); K->damp_ang = R_(M->angular_damping);
    K->restitution = R_(M->restitution); K->rest_thr = R_(M->restitution_threshold);
    K->jdamp = R_(M->joint_damping); K->armature = R_(M->joint_armature);
    const double d2r = pi / 180.0;
    const double lo[NJ] = {-40, 30, -40, -100, -40, -100, -40, 30}, hi[NJ] = {40, 100, 40, -30, 40, -30, 40, 100};
    for (int j = 0; j < NJ; ++j) { K->lo[j] = R_(lo[j] * d2r); K->hi[j] = R_(hi[j] * d2r); }
}

void FN(orc_world_init)(const hrl_config *cfg, FN(orc_world) * W) {
    memset(W, 0, sizeof(*W));
    REAL hx = 0, hy = 0;
    if (cfg->env_kind == HRL_ANT_GATHER || cfg->env_kind == HRL_POINT_GATHER) { hx = R_(cfg->world_size[0]) / 2; hy = R_(cfg->world_size[1]) / 2; }
    if (cfg->env_kind == HRL_ANT_FLAGRUN && (cfg->flag_enclosed || cfg->use_sensor)) { hx = R_(cfg->world_size[0]) / 2; hy = R_(cfg->world_size[1]) / 2; } /* ant_flagrun_env.py:59-61 */
    if (cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_MAZE_MJ) { hx = 5; hy = 9; } /* maze_scene.py:10 */
    if (hx > 0) { /* walls 0.1 thick centred on +-size/2 (sizeable_enclosed_scene.py:46-57, wall.xml:19) */
        REAL t = R_(0.05);
        REAL n[4][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}};
        REAL d[4] = {-(hx - t), -(hx - t), -(hy - t), -(hy - t)};
        W->n_planes = 4;
        for (int i = 0; i < 4; ++i) { for (int k = 0; k < 3; ++k) W->plane_n[i][k] = n[i][k]; W->plane_d[i] = d[i]; }
    }
    if (cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_MAZE_MJ) { /* box.xml:19 6x4x2 at (-2,0,1) (maze_scene.py:12-13,35) */
        W->n_boxes = 1;
        FN(v3set)(W->box_lo[0], -5, -2, 0); FN(v3set)(W->box_hi[0], 1, 2, 2);
    }
}

/* per-substep kinematic + articulated-body data */
typedef struct FN(orc_dyn) {
    REAL X[3], Y[3], Z[3];
    REAL ph[4][3], pa[4][3], tip[4][3]; /* hip anchor, ankle anchor, foot tip, relative to O = torso COM, world axes */
    REAL S[NJ][6], U[NJ][6], invD[NJ], uterm[NJ], cb[NJ][6];
    REAL Lb[6][6], idb[6]; /* base articulated inertia I0^A = L D L^T: unit lower L (strictly lower part used), 1/D */
    REAL a0[6], qdd[NJ];
} FN(orc_dyn);

/* columns of the rotation matrix of the unit quaternion (x, y, z, w) */
static void FN(quat_axes)(REAL x, REAL y, REAL z, REAL w, REAL *X, REAL *Y, REAL *Z) {
    X[0] = FMA_(R_(-2), FMA_(y, y, z * z), R_(1)); X[1] = R_(2) * FMA_(x, y, w * z); X[2] = R_(2) * FMA_(x, z, -(w * y));
    Y[0] = R_(2) * FMA_(x, y, -(w * z)); Y[1] = FMA_(R_(-2), FMA_(x, x, z * z), R_(1)); Y[2] = R_(2) * FMA_(y, z, w * x);
    Z[0] = R_(2) * FMA_(x, z, w * y); Z[1] = R_(2) * FMA_(y, z, -(w * x)); Z[2] = FMA_(R_(-2), FMA_(x, x, y * y), R_(1));
}
static void FN(spatial_inertia)(REAL I[6][6], REAL m, REAL alpha, REAL beta, const REAL *e, const REAL *c) {
    REAL cc = FN(v3dot)(c, c);
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) { /* upper triangle, mirrored: the matrix is exactly symmetric */
            REAL d = (i == j) ? R_(1) : R_(0);
            I[i][j] = I[j][i] = FMA_(m, FMA_(-c[i], c[j], cc * d), FMA_(beta * e[i], e[j], alpha * d));
            I[3 + i][3 + j] = I[3 + j][3 + i] = m * d;
        }
    /* top-right = m [c]x ; bottom-left = its transpose */
    REAL cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { I[i][3 + j] = m * cx[i][j]; I[3 + j][i] = m * cx[i][j]; }
}
static void FN(matvec6)(REAL *o, REAL A[6][6], const REAL *x) {
    for (int i = 0; i < 6; ++i) o[i] = FN(dot6)(A[i], x);
}
static void FN(crm)(REAL *o, const REAL *v, const REAL *m) { /* spatial motion cross product v x m */
    REAL a[3], b[3], c[3];
    FN(v3cross)(a, v, m); FN(v3cross)(b, v, m + 3); FN(v3cross)(c, v + 3, m);
    for (int i = 0; i < 3; ++i) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static void FN(crf)(REAL *o, const REAL *v, const REAL *f) { /* spatial force cross product v x* f */
    REAL a[3], b[3], c[3];
    FN(v3cross)(a, v, f); FN(v3cross)(b, v + 3, f + 3); FN(v3cross)(c, v, f + 3);
    for (int i = 0; i < 3; ++i) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}

/* Square-root-free Cholesky of a symmetric positive-definite 6x6: A = L D L^T with L unit lower triangular.
 * Only the strictly lower part of L and id = 1/D are produced; the lower triangle of A is read. */
static void FN(ldl6_factor)(REAL L[6][6], REAL *id, REAL A[6][6]) {
    REAL d[6];
    for (int j = 0; j < 6; ++j) {
        REAL v[6], s = A[j][j];
        for (int k = 0; k < j; ++k) v[k] = L[j][k] * d[k];
        for (int k = 0; k < j; ++k) s = FMA_(-L[j][k], v[k], s);
        d[j] = s; id[j] = R_(1) / s;
        for (int i = j + 1; i < 6; ++i) {
            REAL t = A[i][j];
            for (int k = 0; k < j; ++k) t = FMA_(-L[i][k], v[k], t);
            L[i][j] = t * id[j];
        }
    }
}
/* x = A^-1 b through the factors: L y = b, z = y / D, L^T x = z */
static void FN(ldl6_solve)(REAL *x, REAL L[6][6], const REAL *id, const REAL *b) {
    REAL y[6];
    for (int i = 0; i < 6; ++i) {
        REAL t = b[i];
        for (int k = 0; k < i; ++k) t = FMA_(-L[i][k], y[k], t);
        y[i] = t;
    }
    for (int i = 5; i >= 0; --i) {
        REAL t = y[i] * id[i];
        for (int k = i + 1; k < 6; ++k) t = FMA_(-L[k][i], x[k], t);
        x[i] = t;
    }
}

/* Kinematics at q, articulated inertias, and (if u/tau given) forward dynamics qdd, a0.
 * All spatial quantities are in world axes about the point O = current torso COM (an inertial frame that
 * instantaneously coincides with the torso), so parent<->child transforms are the identity. */
void FN(orc_dynamics)(const FN(orc_consts) * K, const REAL *q, const REAL *u, const REAL *tau, FN(orc_dyn) * D) {
    const REAL is2 = R_(0.70710678118654752440);
    REAL x = q[3], y = q[4], z = q[5], w = q[6];
    FN(quat_axes)(x, y, z, w, D->X, D->Y, D->Z);
    REAL IA[NBODY][6][6], pA[NBODY][6], v[NBODY][6];
    REAL zero3[3] = {0, 0, 0};
    FN(spatial_inertia)(IA[0], K->m0, K->a0, K->b0, D->Z, zero3);
    for (int i = 0; i < 6; ++i) v[0][i] = u[i];
    REAL com[NBODY][3], mass[NBODY], bal[NBODY], bbe[NBODY], bax[NBODY][3]; /* per body: COM, mass, central inertia alpha 1 + beta e e^T */
    FN(v3set)(com[0], 0, 0, 0); mass[0] = K->m0; bal[0] = K->a0; bbe[0] = K->b0;
    for (int k = 0; k < 3; ++k) bax[0][k] = D->Z[k];
    for (int l = 0; l < 4; ++l) {
        REAL qh = q[7 + 2 * l], qa = q[8 + 2 * l];
        REAL ch, sh, ca, sa;
        FN(dyn_sincos)(qh, &sh, &ch);
        FN(dyn_sincos)(qa, &sa, &ca);
        REAL sx = R_(LEG_SX[l]), sy = R_(LEG_SY[l]), ax = R_(ANK_AX[l]), ay = R_(ANK_AY[l]), sg = R_(LEG_SIGMA[l]);
        REAL e1x = FMA_(sx, ch, -(sy * sh)) * is2, e1y = FMA_(sx, sh, sy * ch) * is2; /* Rz(qh) * leg direction, torso frame */
        REAL axx = FMA_(ax, ch, -(ay * sh)) * is2, axy = FMA_(ax, sh, ay * ch) * is2; /* Rz(qh) * ankle axis            */
        REAL e1[3], axw[3], e2[3], caux[3], cfoot[3];
        for (int k = 0; k < 3; ++k) {
            e1[k] = FMA_(e1y, D->Y[k], e1x * D->X[k]);
            axw[k] = FMA_(axy, D->Y[k], axx * D->X[k]);
            e2[k] = FMA_(sg * sa, D->Z[k], ca * e1[k]);
            D->ph[l][k] = R_(0.2) * FMA_(sy, D->Y[k], sx * D->X[k]);
            D->pa[l][k] = FMA_(K->L1, e1[k], D->ph[l][k]);
            D->tip[l][k] = FMA_(K->L2, e2[k], D->pa[l][k]);
            caux[k] = FMA_(K->L1 * R_(0.5), e1[k], D->ph[l][k]);
            cfoot[k] = FMA_(K->L2 * R_(0.5), e2[k], D->pa[l][k]);
        }
        int jh = 2 * l, ja = 2 * l + 1, bx = 1 + 2 * l, bf = 2 + 2 * l;
        for (int k = 0; k < 3; ++k) { D->S[jh][k] = D->Z[k]; D->S[ja][k] = axw[k]; }
        FN(v3cross)(D->S[jh] + 3, D->ph[l], D->Z);
        FN(v3cross)(D->S[ja] + 3, D->pa[l], axw);
        FN(spatial_inertia)(IA[bx], K->m1, K->a1, K->b1, e1, caux);
        FN(spatial_inertia)(IA[bf], K->m2, K->a2, K->b2, e2, cfoot);
        for (int k = 0; k < 3; ++k) { com[bx][k] = caux[k]; com[bf][k] = cfoot[k]; bax[bx][k] = e1[k]; bax[bf][k] = e2[k]; }
        mass[bx] = K->m1; mass[bf] = K->m2; bal[bx] = K->a1; bbe[bx] = K->b1; bal[bf] = K->a2; bbe[bf] = K->b2;
        REAL qdh = u[6 + 2 * l], qda = u[7 + 2 * l], vjh[6], vja[6];
        for (int k = 0; k < 6; ++k) { vjh[k] = D->S[jh][k] * qdh; v[bx][k] = v[0][k] + vjh[k]; }
        for (int k = 0; k < 6; ++k) { vja[k] = D->S[ja][k] * qda; v[bf][k] = v[bx][k] + vja[k]; }
        FN(crm)(D->cb[jh], v[0], vjh);
        FN(crm)(D->cb[ja], v[bx], vja);
    }
    for (int b = 0; b < NBODY; ++b) { /* bias force: v x* (I v) - gravity wrench */
        REAL Iv[6], f[6], fg[3] = {0, 0, -mass[b] * K->g}, ng[3];
        FN(matvec6)(Iv, IA[b], v[b]);
        FN(crf)(f, v[b], Iv);
        FN(v3cross)(ng, com[b], fg);
        for (int k = 0; k < 3; ++k) { pA[b][k] = f[k] - ng[k]; pA[b][3 + k] = f[3 + k] - fg[k]; }
        if (K->damping_on) { /* hrl_model.linear_damping / angular_damping: Bullet's damping wrench of the body -- force m v_c k_l (1 + |v_c|) at its centre of
                                mass, torque (I_c omega) k_a (1 + |omega|) -- about O, from the velocities at the start of the substep */
            REAL wc[3], vc[3], fd[3], nd[3], cf[3];
            FN(v3cross)(wc, v[b], com[b]);
            for (int k = 0; k < 3; ++k) vc[k] = v[b][3 + k] + wc[k];
            const REAL kl = K->damp_lin * (R_(1) + RSQRT(FN(v3dot)(vc, vc))), ka = K->damp_ang * (R_(1) + RSQRT(FN(v3dot)(v[b], v[b])));
            const REAL ew = FN(v3dot)(bax[b], v[b]);
            for (int k = 0; k < 3; ++k) { fd[k] = (mass[b] * vc[k]) * kl; nd[k] = FMA_(bbe[b] * ew, bax[b][k], bal[b] * v[b][k]) * ka; }
            FN(v3cross)(cf, com[b], fd);
            for (int k = 0; k < 3; ++k) { pA[b][k] = pA[b][k] + (nd[k] + cf[k]); pA[b][3 + k] = pA[b][3 + k] + fd[k]; }
        }
    }
    /* backward pass: ankle then hip of every leg; leg contributions are summed (l0+l1)+(l2+l3) into the base */
    REAL Ileg[4][6][6], pleg[4][6];
    for (int l = 0; l < 4; ++l) {
        for (int s = 1; s >= 0; --s) {
            int j = 2 * l + s, child = 1 + j;
            FN(matvec6)(D->U[j], IA[child], D->S[j]);
            REAL Dj = FN(dot6)(D->S[j], D->U[j]) + K->armature; /* the rotor inertia of assets/ant.xml:8 where the model is told to have one */
            D->invD[j] = R_(1) / Dj;
            D->uterm[j] = FMA_(-K->jdamp, u[6 + j], (tau ? tau[j] : R_(0))) - FN(dot6)(D->S[j], pA[child]); /* viscous joint damping likewise */
            REAL Ia[6][6], pa_[6], Iac[6];
            for (int a = 0; a < 6; ++a) /* symmetric rank-1 downdate, upper triangle mirrored */
                for (int b = a; b < 6; ++b) { Ia[a][b] = FMA_(-(D->U[j][a] * D->invD[j]), D->U[j][b], IA[child][a][b]); Ia[b][a] = Ia[a][b]; }
            FN(matvec6)(Iac, Ia, D->cb[j]);
            REAL ud = D->uterm[j] * D->invD[j];
            for (int a = 0; a < 6; ++a) pa_[a] = FMA_(D->U[j][a], ud, pA[child][a] + Iac[a]);
            if (s == 1) { /* ankle -> accumulate into aux body */
                int par = 1 + 2 * l;
                for (int a = 0; a < 6; ++a) { for (int b = 0; b < 6; ++b) IA[par][a][b] += Ia[a][b]; pA[par][a] += pa_[a]; }
            } else {
                memcpy(Ileg[l], Ia, sizeof(Ia)); memcpy(pleg[l], pa_, sizeof(pa_));
            }
        }
    }
    REAL p0[6];
    for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) IA[0][a][b] += (Ileg[0][a][b] + Ileg[1][a][b]) + (Ileg[2][a][b] + Ileg[3][a][b]);
        p0[a] = pA[0][a] + ((pleg[0][a] + pleg[1][a]) + (pleg[2][a] + pleg[3][a]));
    }
    memset(D->Lb, 0, sizeof(D->Lb));
    FN(ldl6_factor)(D->Lb, D->idb, IA[0]);
    { REAL s0[6]; FN(ldl6_solve)(s0, D->Lb, D->idb, p0); for (int a = 0; a < 6; ++a) D->a0[a] = -s0[a]; }
    for (int l = 0; l < 4; ++l) { /* forward pass */
        REAL ap[6], ax_[6];
        int jh = 2 * l, ja = jh + 1;
        for (int k = 0; k < 6; ++k) ap[k] = D->a0[k] + D->cb[jh][k];
        D->qdd[jh] = (D->uterm[jh] - FN(dot6)(D->U[jh], ap)) * D->invD[jh];
        for (int k = 0; k < 6; ++k) ax_[k] = FMA_(D->S[jh][k], D->qdd[jh], ap[k]) + D->cb[ja][k];
        D->qdd[ja] = (D->uterm[ja] - FN(dot6)(D->U[ja], ax_)) * D->invD[ja];
    }
}

/* Velocity response du = M^-1 * (generalized impulse): spatial impulse `phi` (6, about O) on the body at
 * `level` (0 torso, 1 aux, 2 foot) of leg `leg`, plus direct joint impulses th (hip) / ta (ankle) of that leg. */
static void FN(orc_response)(const FN(orc_dyn) * D, const REAL *phi, int level, int leg, REAL th, REAL ta, REAL *du) {
    int jh = 2 * leg, ja = jh + 1;
    REAL p[6] = {0, 0, 0, 0, 0, 0}, ua = ta, uh = th;
    if (level == 2) {
        for (int k = 0; k < 6; ++k) p[k] = -phi[k];
        ua = ta - FN(dot6)(D->S[ja], p);
    }
    { REAL s = ua * D->invD[ja]; for (int k = 0; k < 6; ++k) p[k] = FMA_(D->U[ja][k], s, p[k]); }
    if (level == 1) for (int k = 0; k < 6; ++k) p[k] = p[k] - phi[k];
    uh = th - FN(dot6)(D->S[jh], p);
    { REAL s = uh * D->invD[jh]; for (int k = 0; k < 6; ++k) p[k] = FMA_(D->U[jh][k], s, p[k]); }
    if (level == 0) for (int k = 0; k < 6; ++k) p[k] = p[k] - phi[k];
    REAL dv0[6];
    { REAL s0[6]; FN(ldl6_solve)(s0, ((FN(orc_dyn) *)D)->Lb, D->idb, p); for (int a = 0; a < 6; ++a) dv0[a] = -s0[a]; }
    for (int k = 0; k < 6; ++k) du[k] = dv0[k];
    for (int l = 0; l < 4; ++l) {
        int h_ = 2 * l, a_ = h_ + 1;
        REAL uhl = (l == leg) ? uh : R_(0), ual = (l == leg) ? ua : R_(0), dvx[6];
        REAL dqh = (uhl - FN(dot6)(D->U[h_], dv0)) * D->invD[h_];
        for (int k = 0; k < 6; ++k) dvx[k] = FMA_(D->S[h_][k], dqh, dv0[k]);
        REAL dqa = (ual - FN(dot6)(D->U[a_], dvx)) * D->invD[a_];
        du[6 + h_] = dqh; du[6 + a_] = dqa;
    }
}

/* Orientation update q <- normalize(exp(h omega / 2) (x) q), omega = u[0..2] in world axes.
 * exp(h omega / 2) = (omega (h/2) sinc(x), cos(x)) with x = |omega| h / 2, evaluated through z = x^2: while x <= 0.5 the
 * Taylor polynomials of sinc and cos in z up to z^4 (exact to 3e-10 there) replace the square root, the division and the
 * sine / cosine; beyond that the closed form.  The product of two unit quaternions is off unit length by rounding only, so
 * the renormalisation on the polynomial branch is one Newton step of 1/sqrt at 1, (3 - n2) / 2, exact to (n2 - 1)^2;
 * the closed-form branch divides by the square root.  (The textbook reference uses the closed forms throughout.) */
static void FN(quat_integrate)(REAL h, REAL *q, const REAL *u) {
    const REAL hh = R_(0.5) * h;
    const REAL ww = FMA_(u[2], u[2], FMA_(u[1], u[1], u[0] * u[0])), z = ww * (hh * hh);
    const int small = z <= R_(0.25);
    REAL dq[4];
    if (small) {
        REAL ps = FMA_(z, R_(2.75573192239858906e-6), R_(-1.98412698412698413e-4));
        ps = FMA_(ps, z, R_(8.33333333333333333e-3)); ps = FMA_(ps, z, R_(-1.66666666666666667e-1)); ps = FMA_(ps, z, R_(1.0));
        REAL pc = FMA_(z, R_(2.48015873015873016e-5), R_(-1.38888888888888889e-3));
        pc = FMA_(pc, z, R_(4.16666666666666667e-2)); pc = FMA_(pc, z, R_(-0.5)); pc = FMA_(pc, z, R_(1.0));
        const REAL sc = hh * ps;
        dq[0] = u[0] * sc; dq[1] = u[1] * sc; dq[2] = u[2] * sc; dq[3] = pc;
    } else {
        const REAL wn = RSQRT(ww);
        REAL sh_, ch_;
        FN(dyn_sincos)(hh * wn, &sh_, &ch_);
        const REAL sc = sh_ / wn;
        dq[0] = u[0] * sc; dq[1] = u[1] * sc; dq[2] = u[2] * sc; dq[3] = ch_;
    }
    const REAL x = q[3], y = q[4], zq = q[5], w = q[6]; /* q <- dq (x) q */
    const REAL nx = FMA_(-dq[2], y, FMA_(dq[1], zq, FMA_(dq[0], w, dq[3] * x)));
    const REAL ny = FMA_(dq[2], x, FMA_(dq[1], w, FMA_(-dq[0], zq, dq[3] * y)));
    const REAL nz = FMA_(dq[2], w, FMA_(-dq[1], x, FMA_(dq[0], y, dq[3] * zq)));
    const REAL nw = FMA_(-dq[2], zq, FMA_(-dq[1], y, FMA_(-dq[0], x, dq[3] * w)));
    const REAL n2 = FMA_(nx, nx, ny * ny) + FMA_(nz, nz, nw * nw);
    const REAL inv = small ? FMA_(R_(-0.5), n2, R_(1.5)) : R_(1) / RSQRT(n2);
    q[3] = nx * inv; q[4] = ny * inv; q[5] = nz * inv; q[6] = nw * inv;
}

/* btPlaneSpace1-style tangent basis for a unit normal */
static void FN(tangent_basis)(const REAL *n, REAL *t1, REAL *t2) {
    if (RFABS(n[2]) > R_(0.70710678118654752440)) {
        REAL a = FMA_(n[1], n[1], n[2] * n[2]), k = R_(1) / RSQRT(a);
        FN(v3set)(t1, 0, -n[2] * k, n[1] * k);
        FN(v3set)(t2, a * k, -n[0] * t1[2], n[0] * t1[1]);
    } else {
        REAL a = FMA_(n[0], n[0], n[1] * n[1]), k = R_(1) / RSQRT(a);
        FN(v3set)(t1, -n[1] * k, n[0] * k, 0);
        FN(v3set)(t2, -n[2] * t1[1], n[2] * t1[0], a * k);
    }
}

/* Projected Gauss-Seidel in ROW SPACE.  With A = J M^-1 J^T (row i of A is computed from row i's own
 * Jacobian: A[i][r] = J_i . B_r, a sequential fma chain over the 16 dof slots) and w_i = J_i . u + b_i, every row keeps
 * its unclamped impulse candidate  c_i = lambda_i - w_i / A_ii  up to date: updating row r by dl changes w_i by
 * A[i][r] * dl and lambda_r by dl, i.e. c_i by C[i][r] * dl with C = I - D^-1 A (scaled, with the unit diagonal folded
 * in), so a row update is  ln = clamp(c_r), dl = ln - lambda_r, c_i += C[i][r] * dl for all i.  Rows are visited in
 * order (limits, normals, friction pairs) `iters` times; a friction row's bounds are +-mu * (current impulse of its
 * normal row).  The generalized velocity is reconstructed once at the end: u += sum_r B_r * lambda_r. */
#define ORC_MAXROWS 44
static void FN(orc_pgs)(int nr, REAL (*J)[16], REAL (*B)[16], const REAL *bias, const REAL *hic, const int *frn, const REAL *mu,
                        int iters, REAL *un, REAL *lam) {
    static _Thread_local REAL Cm[ORC_MAXROWS][ORC_MAXROWS];
    REAL c[ORC_MAXROWS], lo[ORC_MAXROWS], hi[ORC_MAXROWS];
    for (int i = 0; i < nr; ++i) {
        REAL Ai[ORC_MAXROWS];
        for (int r = 0; r < nr; ++r) {
            REAL a = J[i][0] * B[r][0];
            for (int d = 1; d < 16; ++d) a = FMA_(J[i][d], B[r][d], a);
            Ai[r] = a;
        }
        REAL invd = R_(1) / Ai[i];
        for (int r = 0; r < nr; ++r) Cm[i][r] = FMA_(-invd, Ai[r], r == i ? R_(1) : R_(0));
        REAL wi = J[i][0] * un[0];
        for (int d = 1; d < 16; ++d) wi = FMA_(J[i][d], un[d], wi);
        c[i] = -(invd * (wi + bias[i])); /* lambda_i = 0 */
        lam[i] = 0; lo[i] = 0; hi[i] = frn[i] >= 0 ? R_(0) : hic[i];
    }
    for (int it = 0; it < iters; ++it) {
#ifdef ORC_SWEEP_STATS /* diagnostic build (tools/sweep_stats.py): how often a whole sweep leaves every (c, lambda) bit unchanged -- the sweeps after it are then identical and could be skipped exactly */
        REAL c0[ORC_MAXROWS], l0[ORC_MAXROWS];
        memcpy(c0, c, sizeof(REAL) * nr); memcpy(l0, lam, sizeof(REAL) * nr);
#endif
        for (int r = 0; r < nr; ++r) {
            REAL ln = FN(med3)(c[r], lo[r], hi[r]);
            REAL dl = ln - lam[r];
            lam[r] = ln;
            for (int i = 0; i < nr; ++i) {
                c[i] = FMA_(Cm[i][r], dl, c[i]);
                if (frn[i] == r) { hi[i] = mu[i] * ln; lo[i] = -hi[i]; }
            }
        }
#ifdef ORC_SWEEP_STATS
        if (REAL_IS_FLOAT) {
            const int fixed = memcmp(c0, c, sizeof(REAL) * nr) == 0 && memcmp(l0, lam, sizeof(REAL) * nr) == 0;
            __atomic_fetch_add(&orc_sweep_stats[it][fixed], 1, __ATOMIC_RELAXED);
            __atomic_fetch_add(&orc_sweep_rows[it][fixed], nr, __ATOMIC_RELAXED);
            if (fixed) { for (int k = it + 1; k < iters; ++k) { __atomic_fetch_add(&orc_sweep_stats[k][2], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&orc_sweep_rows[k][2], nr, __ATOMIC_RELAXED); } break; }
        }
#endif
    }
    for (int d = 0; d < 16; ++d)
        for (int r = 0; r < nr; ++r) un[d] = FMA_(B[r][d], lam[r], un[d]);
}

/* A contact: body A = (level, leg) of the ant against the static world (level2 < 0) or against another ant body
 * B = (level2, leg2) (self-collision).  r = contact point relative to O, n = normal towards A, surface = what it is with:
 * 0 ground, 1..n_planes lateral walls, 8 + b box b of the world, 16 + k item cube k, 64 + pair id self. */
typedef struct FN(orc_contact) { int level, leg, level2, leg2, sphere, surface, second /* 1: a second support point */; REAL r[3], n[3], dist, mu; } FN(orc_contact);

#define ORC_SURF_BOX 8
#define ORC_SURF_ITEM 16
/* code of item cube i: ORC_SURF_ITEM + i for the first 48 items, the items beyond (configs with more than 48 items) behind the 48 capsule-pair codes */
#define ORC_SURF_OF_ITEM(i) ((i) + ((i) < 48 ? ORC_SURF_ITEM : ORC_SURF_SELF))
#define ORC_ITEM_OF_SURF(s) ((s) >= ORC_SURF_ITEM && (s) < ORC_SURF_SELF ? (s) - ORC_SURF_ITEM : ((s) >= ORC_SURF_SELF + 48 && (s) < ORC_SURF_SELF + HRL_MAX_ITEMS ? (s) - ORC_SURF_SELF : -1))
#define ORC_SURF_SELF 64
#define ORC_ITEM_HALF R_(0.125) /* assets/food.xml:12,19: box size 0.25 */
#define ORC_ITEM_Z R_(0.1)      /* gather_scene.py:62 */

/* sphere `s` (0 torso, 1+3l hip, 2+3l ankle, 3+3l tip): centre relative to O, radius, owning body */
static void FN(sphere_info)(const FN(orc_consts) * K, const FN(orc_dyn) * D, int s, REAL *c, REAL *rad, int *level, int *leg) {
    if (s == 0) { FN(v3set)(c, 0, 0, 0); *rad = K->r_torso; *level = 0; *leg = 0; return; }
    int l = (s - 1) / 3, w = (s - 1) % 3;
    const REAL *src = w == 0 ? D->ph[l] : (w == 1 ? D->pa[l] : D->tip[l]);
    for (int k = 0; k < 3; ++k) c[k] = src[k];
    *rad = K->r_caps; *level = w; *leg = l;
}

/* signed distance of a sphere (centre p, radius rad) to the axis-aligned box [lo, hi]; n = unit normal towards the sphere */
static REAL FN(sphere_vs_box)(const REAL *p, REAL rad, const REAL *lo, const REAL *hi, REAL *n) {
    REAL d[3], d2 = 0;
    for (int k = 0; k < 3; ++k) { REAL cp = FN(clampr)(p[k], lo[k], hi[k]); d[k] = p[k] - cp; d2 = FMA_(d[k], d[k], d2); }
    if (d2 > 0) { REAL len = RSQRT(d2), il = R_(1) / len; for (int k = 0; k < 3; ++k) n[k] = d[k] * il; return len - rad; } /* one reciprocal, three products */
    if (!(d2 == 0)) { FN(v3set)(n, 0, 0, 1); return R_(1e30); } /* non-finite centre: no contact */
    int best = 0; REAL bd = R_(1e30), sgn = 1; /* centre inside the box: exit through the nearest face */
    for (int k = 0; k < 3; ++k) {
        REAL dl = p[k] - lo[k], dh = hi[k] - p[k];
        if (dl < bd) { bd = dl; best = k; sgn = -1; }
        if (dh < bd) { bd = dh; best = k; sgn = 1; }
    }
    FN(v3set)(n, 0, 0, 0); n[best] = sgn;
    return -bd - rad;
}
/* The 12 leg capsules (assets/ant.xml:16-55: `fromto` capsules of radius 0.08) against the convex boxes of the world -- the maze box
 * (assets/box.xml:12,19) and the food / poison cubes (assets/food.xml:12,19) -- are tested along their whole axis: shape `s` of the
 * candidate list is the torso sphere (s = 0) or the capsule that ENDS in sphere s (1+3l: O -> hip point, rigid with the torso; 2+3l: hip
 * point -> ankle point, the aux body; 3+3l: ankle point -> foot tip, the foot), so shapes and spheres share the owning body.  Against
 * planes (ground, walls) the deepest point of a capsule is one of its end points, which is why those passes keep the end-point spheres.
 * Start point of shape s relative to O (its end point is the sphere centre of sphere_info). */
static void FN(capsule_start)(const FN(orc_dyn) * D, int s, REAL *c) {
    const int l = (s - 1) / 3, w = (s - 1) % 3;
    const REAL *src = (s == 0 || w == 0) ? 0 : (w == 1 ? D->ph[l] : D->pa[l]);
    for (int k = 0; k < 3; ++k) c[k] = src ? src[k] : R_(0);
}
/* Parameter t in [0, 1] of the point of the segment P(t) = p + t d closest to the axis-aligned box [lo, hi]; where a whole stretch of
 * the segment is closest (it runs alongside a face, or through the box) the middle of that stretch.  Exact up to rounding:
 * g(t) = d . (P(t) - clamp(P(t), lo, hi)) is half the derivative of the squared distance -- nondecreasing, piecewise linear, with corners
 * where a coordinate of P crosses a face of the box.  Candidates: 0, 1 and the six crossing times clamped to [0, 1]; a = the largest
 * candidate with g <= 0, b = the smallest with g >= 0.  No candidate lies strictly between them, so g is linear there: b <= a is the stretch
 * g = 0, else the root of the chord.  g(0) > 0: t = 0; g(1) < 0: t = 1.  A zero-length segment (the torso sphere) gives 0.5 (P = p for every t).
 * Every operation is pinned; non-finite input ends in t = 0 (no candidate passes a comparison). */
static REAL FN(seg_box_t)(const REAL *p, const REAL *d, const REAL *lo, const REAL *hi) {
    REAL T[8], a = -1, ga = 0, b = 2, gb = 0;
    int on[8]; /* candidate 2 + 2k / 3 + 2k IS the crossing of the low / high face of axis k (not clamped to an end of the segment) */
    T[0] = 0; T[1] = 1; on[0] = on[1] = 0;
    for (int k = 0; k < 3; ++k) {
        const REAL inv = d[k] != 0 ? R_(1) / d[k] : R_(0);
        const REAL tl = (lo[k] - p[k]) * inv, th = (hi[k] - p[k]) * inv;
        T[2 + 2 * k] = FN(clampr)(tl, 0, 1); T[3 + 2 * k] = FN(clampr)(th, 0, 1);
        on[2 + 2 * k] = d[k] != 0 && T[2 + 2 * k] == tl; on[3 + 2 * k] = d[k] != 0 && T[3 + 2 * k] == th;
    }
    for (int i = 0; i < 8; ++i) {
        REAL e[3];
        for (int k = 0; k < 3; ++k) {
            const REAL x = FMA_(d[k], T[i], p[k]);
            e[k] = x - FN(clampr)(x, lo[k], hi[k]);
            if (i >= 2 && k == (i - 2) / 2 && on[i]) e[k] = 0; /* on the face by construction: exactly, not to rounding -- the stretch g = 0 is then found by comparisons with 0 */
        }
        const REAL g = FN(v3dot)(d, e);
        if (g <= 0 && T[i] > a) { a = T[i]; ga = g; }
        if (g >= 0 && T[i] < b) { b = T[i]; gb = g; }
    }
    if (a < 0) return 0;
    if (b > 1) return 1;
    if (!(a < b)) return R_(0.5) * (a + b);
    return FMA_(b - a, ga / (ga - gb), a);
}
REAL FN(orc_seg_box_t)(const REAL *p, const REAL *d, const REAL *lo, const REAL *hi) { return FN(seg_box_t)(p, d, lo, hi); } /* tests */
/* shape s of pose `pos` against the box: the point of its axis closest to the box stands in for the sphere centre of sphere_vs_box
 * (an axis point inside the box leaves through the nearest face, as a sphere centre does).  c = that point relative to O. */
static REAL FN(shape_vs_box)(const FN(orc_dyn) * D, const REAL *pos, int s, const REAL *c_end, REAL rad, const REAL *lo, const REAL *hi, REAL *c, REAL *n, REAL *t_out) {
    REAL c0[3], pw[3], d[3], p[3];
    FN(capsule_start)(D, s, c0);
    for (int k = 0; k < 3; ++k) { pw[k] = pos[k] + c0[k]; d[k] = c_end[k] - c0[k]; }
    const REAL t = FN(seg_box_t)(pw, d, lo, hi);
    for (int k = 0; k < 3; ++k) { c[k] = FMA_(d[k], t, c0[k]); p[k] = pos[k] + c[k]; }
    if (t_out) *t_out = t;
    return FN(sphere_vs_box)(p, rad, lo, hi, n);
}
/* Second support point of a capsule that rests (nearly) flat on a face of a box (Bullet keeps a manifold of up to four points per pair there; one point
 * lets the capsule rock about it).  The first contact is the axis point P(t1) closest to the box, normal n1.  When n1 is a FACE normal to within 5.7 degrees -- its largest
 * component is 0.995 or more: the closest point of the box lies in a face's interior, P(t1) is inside the box, or (a capsule longer than the face) P(t1) has
 * just passed the face's edge on its way down and the normal leans by the capsule's own tilt -- the part of the axis that projects into that face is [ta, tb] = [0, 1] clipped by the slabs of the two other axes; its end farther from t1 -- tb, towards the capsule's free end (its start is where
 * the leg's neighbouring capsule ends), unless ta is farther by more than a thousandth of the axis -- is the candidate t2, if it is at least one radius away from P(t1) along the axis.  The caller tests
 * the sphere at P(t2) against the box like any other.  Every operation pinned; a NaN anywhere ends in "none". */
static int FN(second_point)(const REAL *p, const REAL *d, REAL t1, const REAL *n1, REAL dist1, REAL cdist, REAL rad, const REAL *lo, const REAL *hi, REAL *t2) {
    const REAL a0 = RFABS(n1[0]), a1 = RFABS(n1[1]), a2 = RFABS(n1[2]);
    const int kf = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
    if (!((kf == 0 ? a0 : (kf == 1 ? a1 : a2)) >= R_(0.995))) return 0;
    { /* what the wave form asks FIRST, for every kept box contact: can a point a radius or more along this axis still be within the contact distance of the
         face at all?  It rises by |d_kf| per unit of t, i.e. by at least |d_kf| rad / |d| -- against cdist - dist1 (+ 5 mm: just past the face's edge dist1 is
         the distance to the edge, a hair more than the height over the face).  Implied by the tests below, so no part of the textbook form; stated here
         because the wave form decides by it whether to run this function, and both must round alike. */
        const REAL a = d[kf] * rad, b = (cdist - dist1) + R_(0.005);
        if (!(a * a <= (b * b) * FN(v3dot)(d, d))) return 0;
    }
    REAL ta = 0, tb = 1;
    for (int k = 0; k < 3; ++k) {
        if (k == kf || d[k] == 0) continue;
        const REAL inv = R_(1) / d[k];
        const REAL u = (lo[k] - p[k]) * inv, v = (hi[k] - p[k]) * inv;
        const REAL tl = u < v ? u : v, th = u < v ? v : u;
        ta = tl > ta ? tl : ta; tb = th < tb ? th : tb;
    }
    const REAL tt = ((tb - t1) + R_(1e-3) >= t1 - ta) ? tb : ta, dt = tt - t1; /* the band: a tie at the exact middle of a stretch is not decided by rounding */
    *t2 = tt;
    return (dt * dt) * FN(v3dot)(d, d) >= rad * rad;
}
/* the second support point of shape s (first contact: parameter t1, normal n1): the sphere at P(t2) against the FACE of the first contact -- its normal, and the
 * distance to that face's plane (P(t2) sits on the border of the face's region by construction, where the box's closest feature is a matter of rounding);
 * returns 0 when the shape has none */
static int FN(shape_vs_box_second)(const FN(orc_dyn) * D, const REAL *pos, int s, const REAL *c_end, REAL rad, const REAL *lo, const REAL *hi, REAL t1, const REAL *n1,
                                   REAL dist1, REAL cdist, REAL *c, REAL *n, REAL *dist) {
    REAL c0[3], pw[3], d[3], t2;
    FN(capsule_start)(D, s, c0);
    for (int k = 0; k < 3; ++k) { pw[k] = pos[k] + c0[k]; d[k] = c_end[k] - c0[k]; }
    if (!FN(second_point)(pw, d, t1, n1, dist1, cdist, rad, lo, hi, &t2)) return 0;
    const REAL a0 = RFABS(n1[0]), a1 = RFABS(n1[1]), a2 = RFABS(n1[2]);
    const int kf = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
    const REAL sg = n1[kf] > 0 ? R_(1) : R_(-1);
    for (int k = 0; k < 3; ++k) { c[k] = FMA_(d[k], t2, c0[k]); n[k] = k == kf ? sg : R_(0); }
    *dist = sg * ((pos[kf] + c[kf]) - (sg > 0 ? hi[kf] : lo[kf])) - rad;
    return 1;
}
static void FN(item_box)(const REAL *item_xy, REAL *lo, REAL *hi) {
    lo[0] = item_xy[0] - ORC_ITEM_HALF; lo[1] = item_xy[1] - ORC_ITEM_HALF; lo[2] = ORC_ITEM_Z - ORC_ITEM_HALF;
    hi[0] = item_xy[0] + ORC_ITEM_HALF; hi[1] = item_xy[1] + ORC_ITEM_HALF; hi[2] = ORC_ITEM_Z + ORC_ITEM_HALF;
}

/* Closest points of the capsule axes P1Q1 and P2Q2 (Ericson, Real-Time Collision Detection 5.1.9), every operation
 * pinned.  The segments have the fixed squared lengths a, e of the model (ia = 1/a, ie = 1/e are exact binary
 * fractions: 12.5 and 3.125), so only the parallelism test divides. */
static void FN(seg_seg)(const REAL *p1, const REAL *q1, REAL a, REAL ia, const REAL *p2, const REAL *q2, REAL e, REAL ie,
                        REAL *c1, REAL *c2) {
    REAL d1[3], d2[3], r[3];
    for (int k = 0; k < 3; ++k) { d1[k] = q1[k] - p1[k]; d2[k] = q2[k] - p2[k]; r[k] = p1[k] - p2[k]; }
    REAL f = FN(v3dot)(d2, r), c = FN(v3dot)(d1, r), b = FN(v3dot)(d1, d2);
    REAL denom = FMA_(a, e, -(b * b)), sp = 0, t;
    if (denom > R_(1e-9)) sp = FN(clampr)(FMA_(b, f, -(c * e)) / denom, 0, 1);
    t = FMA_(b, sp, f) * ie;
    if (t < 0) { t = 0; sp = FN(clampr)(-c * ia, 0, 1); }
    else if (t > 1) { t = 1; sp = FN(clampr)((b - c) * ia, 0, 1); }
    for (int k = 0; k < 3; ++k) { c1[k] = FMA_(d1[k], sp, p1[k]); c2[k] = FMA_(d2[k], t, p2[k]); }
}

/* Candidate order: surface-major (ground, lateral planes, world boxes, item cubes), sphere-minor; then the capsule
 * pairs of different legs (pair id = 8 * legpair + 3 * segA + segB - 1; legpair (0,1),(0,2),(0,3),(1,2),(1,3),(2,3);
 * seg 0 = the jointless leg capsule O -> hip point (torso body), 1 = aux capsule, 2 = foot capsule; (0,0) skipped: both
 * rigid with the torso).  At most MAXC contacts are kept; *n_candidates counts them all. */
static int FN(orc_detect)(const FN(orc_consts) * K, const FN(orc_world) * W, const FN(orc_dyn) * D, const REAL *pos,
                          const REAL *items_xy, int n_items, FN(orc_contact) * C, int *ground_touch /* [13] */, int *n_candidates) {
    int nc = 0, ncand = 0;
    const int use_items = (K->item_collision && items_xy) ? n_items : 0;
    const int nsurf = 1 + W->n_planes + W->n_boxes + use_items;
    for (int s = 0; s < 13; ++s) ground_touch[s] = 0;
    /* the first contacts with boxes and cubes, in candidate order, for the second support points after them */
    struct { int s, surface; REAL t, dist, n[3], lo[3], hi[3]; } first[13 * (HRL_MAX_ITEMS + 4)];
    int n_first = 0;
    for (int f = 0; f < nsurf; ++f) {
        for (int s = 0; s < 13; ++s) {
            REAL c[3], rad, n[3], dist, p[3];
            int level, leg, surface;
            FN(sphere_info)(K, D, s, c, &rad, &level, &leg);
            for (int k = 0; k < 3; ++k) p[k] = pos[k] + c[k];
            if (f == 0) { FN(v3set)(n, 0, 0, 1); dist = (p[2] - K->ground_z) - rad; surface = 0; }
            else if (f <= W->n_planes) {
                const REAL *pn = W->plane_n[f - 1];
                FN(v3set)(n, pn[0], pn[1], pn[2]);
                dist = (FN(v3dot)(n, p) - W->plane_d[f - 1]) - rad; surface = f;
            } else if (f <= W->n_planes + W->n_boxes) {
                const int b = f - 1 - W->n_planes;
                REAL t1;
                dist = FN(shape_vs_box)(D, pos, s, c, rad, W->box_lo[b], W->box_hi[b], c, n, &t1); surface = ORC_SURF_BOX + b;
                if (dist < K->cdist && s > 0) {
                    first[n_first].s = s; first[n_first].surface = surface; first[n_first].t = t1; first[n_first].dist = dist;
                    for (int a = 0; a < 3; ++a) { first[n_first].n[a] = n[a]; first[n_first].lo[a] = W->box_lo[b][a]; first[n_first].hi[a] = W->box_hi[b][a]; }
                    ++n_first;
                }
            } else {
                const int k = f - 1 - W->n_planes - W->n_boxes;
                REAL lo[3], hi[3];
                if (items_xy[2 * k] != items_xy[2 * k] || items_xy[2 * k + 1] != items_xy[2 * k + 1]) continue; /* a cube at a NaN place is nowhere (a clamp between NaN bounds would put it everywhere) */
                { /* broad phase: every point of every shape lies within `reach` of the torso centre (hip 0.283 + aux 0.283 + foot 0.566 + radius + contact
                     distance), so a cube farther than that (+ its half extent) along x or y touches nothing -- the same list as testing it */
                    const REAL reach = R_(0.2) * R_(1.41421356237309504880) + K->L1 + K->L2 + K->r_caps + K->cdist + R_(0.02) + ORC_ITEM_HALF;
                    if (!(RFABS(pos[0] - items_xy[2 * k]) < reach) || !(RFABS(pos[1] - items_xy[2 * k + 1]) < reach)) continue;
                }
                FN(item_box)(items_xy + 2 * k, lo, hi);
                REAL t1;
                dist = FN(shape_vs_box)(D, pos, s, c, rad, lo, hi, c, n, &t1); surface = ORC_SURF_OF_ITEM(k);
                if (dist < K->cdist && s > 0) {
                    first[n_first].s = s; first[n_first].surface = surface; first[n_first].t = t1; first[n_first].dist = dist;
                    for (int a = 0; a < 3; ++a) { first[n_first].n[a] = n[a]; first[n_first].lo[a] = lo[a]; first[n_first].hi[a] = hi[a]; }
                    ++n_first;
                }
            }
            if (dist < K->cdist) {
                if (f == 0) ground_touch[s] = 1;
                ++ncand;
                if (nc < K->max_contacts) {
                    FN(orc_contact) *cc = &C[nc++];
                    cc->level = level; cc->leg = leg; cc->level2 = -1; cc->leg2 = 0; cc->sphere = s; cc->dist = dist; cc->surface = surface; cc->mu = K->mu; cc->second = 0;
                    for (int k = 0; k < 3; ++k) { cc->n[k] = n[k]; cc->r[k] = FMA_(-rad, n[k], c[k]); }
                }
            }
        }
    }
    /* after the first contacts of every box and cube: the second support points of the capsules that lie flat on a face, in the order of their first contacts */
    for (int i = 0; i < n_first; ++i) {
        REAL c[3], rad, n[3], dist;
        int level, leg;
        const int s = first[i].s;
        FN(sphere_info)(K, D, s, c, &rad, &level, &leg);
        if (!FN(shape_vs_box_second)(D, pos, s, c, rad, first[i].lo, first[i].hi, first[i].t, first[i].n, first[i].dist, K->cdist, c, n, &dist)) continue;
        if (dist < K->cdist) {
            ++ncand;
            if (nc < K->max_contacts) {
                FN(orc_contact) *cc = &C[nc++];
                cc->level = level; cc->leg = leg; cc->level2 = -1; cc->leg2 = 0; cc->sphere = s; cc->dist = dist; cc->surface = first[i].surface; cc->mu = K->mu; cc->second = 1;
                for (int k = 0; k < 3; ++k) { cc->n[k] = n[k]; cc->r[k] = FMA_(-rad, n[k], c[k]); }
            }
        }
    }
    if (K->self_collision) {
        const REAL zero[3] = {0, 0, 0};
        const REAL A0 = R_(0.08), A2 = R_(0.32), I0 = R_(12.5), I2 = R_(3.125); /* |(.2,.2)|^2, |(.4,.4)|^2 (ant.xml:16-22) and reciprocals */
        const REAL thr = (K->r_caps + K->r_caps) + K->cdist;
        int pair = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        if (a == 0 && b == 0) continue;
                        const REAL *p1 = a == 0 ? zero : (a == 1 ? D->ph[i] : D->pa[i]), *q1 = a == 0 ? D->ph[i] : (a == 1 ? D->pa[i] : D->tip[i]);
                        const REAL *p2 = b == 0 ? zero : (b == 1 ? D->ph[j] : D->pa[j]), *q2 = b == 0 ? D->ph[j] : (b == 1 ? D->pa[j] : D->tip[j]);
                        REAL c1[3], c2[3], dv[3];
                        FN(seg_seg)(p1, q1, a == 2 ? A2 : A0, a == 2 ? I2 : I0, p2, q2, b == 2 ? A2 : A0, b == 2 ? I2 : I0, c1, c2);
                        for (int k = 0; k < 3; ++k) dv[k] = c1[k] - c2[k];
                        const REAL d2n = FN(v3dot)(dv, dv);
                        const int id = pair++;
                        if (!(d2n < thr * thr)) continue;
                        ++ncand;
                        if (nc >= K->max_contacts) continue;
                        FN(orc_contact) *cc = &C[nc++];
                        const REAL len = RSQRT(d2n);
                        cc->level = a; cc->leg = i; cc->level2 = b; cc->leg2 = j; cc->sphere = -1; cc->surface = ORC_SURF_SELF + id; cc->mu = K->mu_self; cc->second = 0;
                        cc->dist = len - (K->r_caps + K->r_caps);
                        if (len > 0) { const REAL il = R_(1) / len; for (int k = 0; k < 3; ++k) cc->n[k] = dv[k] * il; } else FN(v3set)(cc->n, 0, 0, 1);
                        for (int k = 0; k < 3; ++k) cc->r[k] = R_(0.5) * (c1[k] + c2[k]); /* equal radii: midway between the two surface points */
                    }
    }
    if (n_candidates) *n_candidates = ncand;
    return nc;
}

typedef struct FN(orc_substep_dbg) { int n_rows, n_limits, n_contacts, n_candidates, surface[MAXC]; REAL lambda[MAXR]; } FN(orc_substep_dbg);

/* One physics substep on internal coordinates q[15] (x,y,z,qx,qy,qz,qw,joints) and u[14] (omega, v, joint rates). */
void FN(orc_ant_substep)(const FN(orc_consts) * K, const FN(orc_world) * W, REAL *q, REAL *u, const REAL *tau,
                         const REAL *items_xy, int n_items, int *ground_touch, FN(orc_substep_dbg) * dbg) {
    FN(orc_dyn) D;
    REAL h = K->h;
    FN(orc_dynamics)(K, q, u, tau, &D);
    /* (1) unconstrained velocity update; classical linear acceleration of the torso COM = a_O + omega x v */
    REAL wxv[3];
    FN(v3cross)(wxv, u, u + 3);
    REAL un[16];
    for (int k = 0; k < 3; ++k) { un[k] = FMA_(h, D.a0[k], u[k]); un[3 + k] = FMA_(h, D.a0[3 + k] + wxv[k], u[3 + k]); }
    for (int j = 0; j < NJ; ++j) un[6 + j] = FMA_(h, D.qdd[j], u[6 + j]);
    un[14] = un[15] = 0;
    REAL u16[16]; /* the velocity at the start of the substep, dof order: approach speeds of the restitution rows */
    for (int k = 0; k < NDOF; ++k) u16[k] = u[k];
    u16[14] = u16[15] = 0;
    /* (2) constraint rows: joint limits, contact normals, friction pairs */
    REAL J[MAXR][16], B[MAXR][16], bias[MAXR], hi[MAXR], lam[MAXR], mu_row[MAXR]; /* hi: bound of non-friction rows */
    int fr_normal[MAXR];
    int nr = 0, nl = 0;
    REAL zero6[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < NJ; ++j) {
        REAL dlo = q[7 + j] - K->lo[j], dhi = K->hi[j] - q[7 + j];
        int act = 0; REAL sgn = 0, dist = 0;
        if (dlo < K->lmargin) { act = 1; sgn = 1; dist = dlo; }
        else if (dhi < K->lmargin) { act = 1; sgn = -1; dist = dhi; }
        if (!act) continue;
        memset(J[nr], 0, sizeof(J[nr]));
        J[nr][6 + j] = sgn;
        FN(orc_response)(&D, zero6, 0, j / 2, (j & 1) ? R_(0) : sgn, (j & 1) ? sgn : R_(0), B[nr]);
        B[nr][14] = B[nr][15] = 0;
        bias[nr] = (dist > 0 ? dist : K->erp_l * dist) * K->inv_h;
        hi[nr] = K->limp_max; fr_normal[nr] = -1; mu_row[nr] = 0;
        ++nr; ++nl;
    }
    FN(orc_contact) C[MAXC];
    int ncand = 0;
    int nc = FN(orc_detect)(K, W, &D, q, items_xy, n_items, C, ground_touch, &ncand);
    /* rows nl..nl+nc-1: normals; then rows nl+nc+2c (t1), nl+nc+2c+1 (t2) */
    for (int row = 0; row < 3 * nc; ++row) {
        int c = row < nc ? row : (row - nc) / 2, which = row < nc ? 0 : 1 + ((row - nc) & 1);
        REAL t1[3], t2[3], phi[6];
        FN(tangent_basis)(C[c].n, t1, t2);
        const REAL *d = which == 0 ? C[c].n : (which == 1 ? t1 : t2);
        FN(v3cross)(phi, C[c].r, d);
        for (int k = 0; k < 3; ++k) phi[3 + k] = d[k];
        memset(J[nr], 0, sizeof(J[nr]));
        if (C[c].level >= 1) J[nr][6 + 2 * C[c].leg] = FN(dot6)(phi, D.S[2 * C[c].leg]);
        if (C[c].level >= 2) J[nr][7 + 2 * C[c].leg] = FN(dot6)(phi, D.S[2 * C[c].leg + 1]);
        FN(orc_response)(&D, phi, C[c].level, C[c].leg, 0, 0, B[nr]);
        B[nr][14] = B[nr][15] = 0;
        if (C[c].level2 < 0) for (int k = 0; k < 6; ++k) J[nr][k] = phi[k];
        else { /* self contact: J = J_A - J_B.  Both bodies move with the torso, so its part cancels exactly; what is left are
                * the joints between the torso and A (leg) and, negated, between the torso and B (leg2 > leg).  B = B_A - B_B. */
            REAL B2[16];
            if (C[c].level2 >= 1) J[nr][6 + 2 * C[c].leg2] = -FN(dot6)(phi, D.S[2 * C[c].leg2]);
            if (C[c].level2 >= 2) J[nr][7 + 2 * C[c].leg2] = -FN(dot6)(phi, D.S[2 * C[c].leg2 + 1]);
            FN(orc_response)(&D, phi, C[c].level2, C[c].leg2, 0, 0, B2);
            for (int k = 0; k < 14; ++k) B[nr][k] = B[nr][k] - B2[k];
        }
        if (which == 0) {
            bias[nr] = (C[c].dist > 0 ? C[c].dist : K->erp_c * C[c].dist) * K->inv_h;
            if (K->restitution > 0) { /* hrl_model.restitution: approaching faster than the threshold -> ask for restitution * (approach speed) of separation */
                REAL vn = J[nr][0] * u16[0];
                for (int d = 1; d < 16; ++d) vn = FMA_(J[nr][d], u16[d], vn);
                if (vn < -K->rest_thr) bias[nr] = FMA_(K->restitution, vn, bias[nr]);
            }
            hi[nr] = R_(1e30); fr_normal[nr] = -1; mu_row[nr] = 0;
        } else { bias[nr] = 0; hi[nr] = 0; fr_normal[nr] = nl + c; mu_row[nr] = C[c].mu; }
        ++nr;
    }
    /* (3) projected Gauss-Seidel in row space */
    FN(orc_pgs)(nr, J, B, bias, hi, fr_normal, mu_row, K->iters, un, lam);
    /* (4) joint-rate clamp and position integration (semi-implicit Euler, exponential map for the quaternion) */
    for (int j = 0; j < NJ; ++j) un[6 + j] = FN(clampr)(un[6 + j], -K->vmax, K->vmax);
    for (int k = 0; k < NDOF; ++k) u[k] = un[k];
    for (int k = 0; k < 3; ++k) q[k] = FMA_(h, u[3 + k], q[k]);
    FN(quat_integrate)(h, q, u);
    for (int j = 0; j < NJ; ++j) q[7 + j] = FMA_(h, u[6 + j], q[7 + j]);
    if (dbg) { dbg->n_rows = nr; dbg->n_limits = nl; dbg->n_contacts = nc; dbg->n_candidates = ncand; for (int r = 0; r < nr; ++r) dbg->lambda[r] = lam[r]; for (int c = 0; c < nc; ++c) dbg->surface[c] = C[c].surface; }
}

/* ---------------------------------------------------------------------------------------------- point bot body
 * point_bot.py:10-74 + assets/player_cube.xml:8: free 10 kg cube, half extent 0.35, friction 0.1.  A solid cube's
 * inertia is isotropic (m s^2/6), so M^-1 is diagonal and omega x I omega = 0. */
void FN(orc_point_substep)(const FN(orc_consts) * K, const FN(orc_world) * W, REAL *q, REAL *u, const REAL *force,
                           const REAL *items_xy, int n_items, FN(orc_substep_dbg) * dbg) {
    const REAL m = 10, he = R_(0.35), I = m * (R_(0.7) * R_(0.7)) / 6, h = K->h;
    REAL x = q[3], y = q[4], z = q[5], w = q[6], X[3], Y[3], Z[3];
    FN(quat_axes)(x, y, z, w, X, Y, Z);
    REAL un[6];
    for (int k = 0; k < 3; ++k) un[k] = u[k];
    un[3] = FMA_(h, force[0] / m, u[3]); un[4] = FMA_(h, force[1] / m, u[4]); un[5] = FMA_(h, force[2] / m - K->g, u[5]);
    if (K->damping_on) { /* Bullet's damping of a free body with an isotropic inertia, velocities of the start of the substep */
        const REAL nw = RSQRT(FN(v3dot)(u, u)), nv = RSQRT(FN(v3dot)(u + 3, u + 3));
        const REAL ka = K->damp_ang * (R_(1) + nw), kl = K->damp_lin * (R_(1) + nv);
        for (int k = 0; k < 3; ++k) { un[k] = FMA_(-(h * ka), u[k], un[k]); un[3 + k] = FMA_(-(h * kl), u[3 + k], un[3 + k]); }
    }
    /* contacts: 8 corners vs ground, lateral planes and item cubes in surface-major order, then every cube's 8 corners vs the
     * player's oriented box -- the half that catches a cube under the middle of a face --, at most MAXC in all */
    REAL Jr[3 * MAXC][16], Br[3 * MAXC][16], bias[3 * MAXC], hic[3 * MAXC], lam[3 * MAXC], mu_row[3 * MAXC];
    int frn[3 * MAXC], nc = 0, ncand = 0, csurf[MAXC];
    REAL cr[MAXC][3], cn[MAXC][3], cd[MAXC];
    const int use_items = (K->item_collision && items_xy) ? n_items : 0;
    for (int f = 0; f < 1 + W->n_planes + use_items; ++f)
        for (int s = 0; s < 8; ++s) {
            REAL c[3], n[3], dist;
            REAL sx = (s & 1) ? he : -he, sy = (s & 2) ? he : -he, sz = (s & 4) ? he : -he;
            int surface = f;
            for (int k = 0; k < 3; ++k) c[k] = FMA_(sz, Z[k], FMA_(sy, Y[k], sx * X[k]));
            REAL p[3] = {q[0] + c[0], q[1] + c[1], q[2] + c[2]};
            if (f == 0) { FN(v3set)(n, 0, 0, 1); dist = p[2] - K->ground_z; }
            else if (f <= W->n_planes) {
                const REAL *pn = W->plane_n[f - 1];
                FN(v3set)(n, pn[0], pn[1], pn[2]);
                dist = FN(v3dot)(n, p) - W->plane_d[f - 1];
            } else {
                REAL lo[3], hi[3];
                const REAL *ixy0 = items_xy + 2 * (f - 1 - W->n_planes);
                if (ixy0[0] != ixy0[0] || ixy0[1] != ixy0[1]) continue; /* a cube at a NaN place is nowhere */
                FN(item_box)(ixy0, lo, hi);
                dist = FN(sphere_vs_box)(p, 0, lo, hi, n); surface = ORC_SURF_OF_ITEM(f - 1 - W->n_planes);
            }
            if (dist < K->cdist) {
                ++ncand;
                if (nc < K->max_contacts) {
                    for (int k = 0; k < 3; ++k) { cr[nc][k] = c[k]; cn[nc][k] = n[k]; }
                    cd[nc] = dist; csurf[nc] = surface; ++nc;
                }
            }
        }
    /* then the cubes' own corners against the player's box: corner s of item f in the box frame, closest surface point, normal
     * turned back to the world and towards the player */
    for (int f = 0; f < use_items; ++f)
        for (int s = 0; s < 8; ++s) {
            const REAL *ixy = items_xy + 2 * f;
            if (ixy[0] != ixy[0] || ixy[1] != ixy[1]) continue; /* a cube at a NaN place is nowhere */
            REAL pc[3] = {(s & 1) ? ixy[0] + ORC_ITEM_HALF : ixy[0] - ORC_ITEM_HALF, (s & 2) ? ixy[1] + ORC_ITEM_HALF : ixy[1] - ORC_ITEM_HALF,
                          (s & 4) ? ORC_ITEM_Z + ORC_ITEM_HALF : ORC_ITEM_Z - ORC_ITEM_HALF};
            REAL d[3] = {pc[0] - q[0], pc[1] - q[1], pc[2] - q[2]};
            REAL l[3] = {FN(v3dot)(d, X), FN(v3dot)(d, Y), FN(v3dot)(d, Z)}, nl[3], c[3], n[3];
            const REAL blo[3] = {-he, -he, -he}, bhi[3] = {he, he, he};
            REAL dist = FN(sphere_vs_box)(l, 0, blo, bhi, nl);
            for (int k = 0; k < 3; ++k) {
                REAL nw = FMA_(nl[2], Z[k], FMA_(nl[1], Y[k], nl[0] * X[k]));
                n[k] = -nw; c[k] = FMA_(-dist, nw, d[k]);
            }
            if (dist < K->cdist) {
                ++ncand;
                if (nc < K->max_contacts) {
                    for (int k = 0; k < 3; ++k) { cr[nc][k] = c[k]; cn[nc][k] = n[k]; }
                    cd[nc] = dist; csurf[nc] = ORC_SURF_OF_ITEM(f); ++nc;
                }
            }
        }
    int nr = 0;
    for (int row = 0; row < 3 * nc; ++row) {
        int c = row < nc ? row : (row - nc) / 2, which = row < nc ? 0 : 1 + ((row - nc) & 1);
        REAL t1[3], t2[3];
        FN(tangent_basis)(cn[c], t1, t2);
        const REAL *d = which == 0 ? cn[c] : (which == 1 ? t1 : t2);
        memset(Jr[nr], 0, sizeof(Jr[nr])); memset(Br[nr], 0, sizeof(Br[nr]));
        FN(v3cross)(Jr[nr], cr[c], d);
        for (int k = 0; k < 3; ++k) { Jr[nr][3 + k] = d[k]; Br[nr][k] = Jr[nr][k] / I; Br[nr][3 + k] = d[k] / m; }
        bias[nr] = which == 0 ? (cd[c] > 0 ? cd[c] : K->erp_c * cd[c]) * K->inv_h : R_(0);
        if (which == 0 && K->restitution > 0) {
            REAL vn = Jr[nr][0] * u[0];
            for (int d = 1; d < 6; ++d) vn = FMA_(Jr[nr][d], u[d], vn);
            vn = vn + 0; /* the joint slots of a free body hold exact zeros (the device's row product ends in this addition) */
            if (vn < -K->rest_thr) bias[nr] = FMA_(K->restitution, vn, bias[nr]);
        }
        hic[nr] = which == 0 ? R_(1e30) : R_(0);
        frn[nr] = which == 0 ? -1 : c; mu_row[nr] = which == 0 ? R_(0) : K->mu;
        ++nr;
    }
    REAL un16[16] = {0};
    for (int k = 0; k < 6; ++k) un16[k] = un[k];
    FN(orc_pgs)(nr, Jr, Br, bias, hic, frn, mu_row, K->iters, un16, lam);
    if (dbg) { dbg->n_rows = nr; dbg->n_limits = 0; dbg->n_contacts = nc; dbg->n_candidates = ncand; for (int r = 0; r < nr; ++r) dbg->lambda[r] = lam[r]; for (int c = 0; c < nc; ++c) dbg->surface[c] = csurf[c]; }
    for (int k = 0; k < 6; ++k) un[k] = un16[k];
    for (int k = 0; k < 6; ++k) u[k] = un[k];
    for (int k = 0; k < 3; ++k) q[k] = FMA_(h, u[3 + k], q[k]);
    FN(quat_integrate)(h, q, u);
}

/* =================================================================================================================
 * PART 3 -- observation packing + env-level reset/step on the packed buffers of include/hrl_envs.h
 * ================================================================================================================= */

/* pybullet getEulerFromQuaternion (upstream, restated from memory): ZYX angles with a gimbal-lock guard */
void FN(orc_quat_to_rpy)(const REAL *qq, REAL *rpy) {
    REAL x = qq[0], y = qq[1], z = qq[2], w = qq[3];
    REAL sarg = R_(-2) * (x * z - w * y);
    const REAL hp = R_(1.5707963267948966);
    if (sarg <= R_(-0.99999)) { rpy[0] = 0; rpy[1] = -hp; rpy[2] = 2 * RATAN2(x, -y); }
    else if (sarg >= R_(0.99999)) { rpy[0] = 0; rpy[1] = hp; rpy[2] = 2 * RATAN2(-x, y); }
    else {
        REAL sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
        rpy[0] = RATAN2(2 * (y * z + w * x), ((sqw - sqx) - sqy) + sqz);
        rpy[1] = RASIN(sarg);
        rpy[2] = RATAN2(2 * (x * y + w * z), ((sqw + sqx) - sqy) - sqz);
    }
}

/* upstream WalkerBase.calc_state (restated from memory): 28-vector clipped to +-5.
 * Also returns joints_at_limit, walk_target_dist (centroid based) and yaw. */
void FN(orc_ant_calc_state)(const hrl_config *cfg, const FN(orc_consts) * K, const REAL *qpos, const REAL *qvel,
                            REAL initial_z, const REAL *target, const REAL *feet_contact, REAL *out28,
                            int *joints_at_limit, REAL *walk_target_dist, REAL *rpy_out, REAL *centroid_out) {
    REAL rpy[3];
    FN(orc_quat_to_rpy)(qpos + 3, rpy);
    /* centroid over robot.parts: 13 link COMs + the scene statics */
    FN(orc_dyn) D;
    REAL u0[14] = {0}, q[15];
    for (int i = 0; i < 15; ++i) q[i] = qpos[i];
    FN(orc_dynamics)(K, q, u0, 0, &D);
    REAL sx = 0, sy = 0;
    for (int l = 0; l < 4; ++l) {
        /* fixed leg COM = ph/2; aux COM = (ph+pa)/2; foot COM = (pa+tip)/2 -- all relative to the torso */
        sx += (R_(0.5) * D.ph[l][0] + R_(0.5) * (D.ph[l][0] + D.pa[l][0])) + R_(0.5) * (D.pa[l][0] + D.tip[l][0]);
        sy += (R_(0.5) * D.ph[l][1] + R_(0.5) * (D.ph[l][1] + D.pa[l][1])) + R_(0.5) * (D.pa[l][1] + D.tip[l][1]);
    }
    REAL np_ = R_(13 + cfg->centroid_n_static);
    REAL cx = ((R_(13) * qpos[0] + sx) + R_(cfg->centroid_static_sum[0])) / np_;
    REAL cy = ((R_(13) * qpos[1] + sy) + R_(cfg->centroid_static_sum[1])) / np_;
    REAL dx = target[0] - cx, dy = target[1] - cy;
    REAL theta = RATAN2(dy, dx);
    if (centroid_out) { centroid_out[0] = cx; centroid_out[1] = cy; }
    *walk_target_dist = RSQRT(dy * dy + dx * dx);
    REAL ang = theta - rpy[2];
    REAL c = RCOS(-rpy[2]), s = RSIN(-rpy[2]);
    REAL vx = c * qvel[0] - s * qvel[1], vy = s * qvel[0] + c * qvel[1], vz = qvel[2];
    out28[0] = qpos[2] - initial_z; out28[1] = RSIN(ang); out28[2] = RCOS(ang);
    out28[3] = R_(0.3) * vx; out28[4] = R_(0.3) * vy; out28[5] = R_(0.3) * vz; out28[6] = rpy[0]; out28[7] = rpy[1];
    int nlim = 0;
    for (int j = 0; j < NJ; ++j) {
        REAL mid = R_(0.5) * (K->lo[j] + K->hi[j]), scale = R_(2) / (K->hi[j] - K->lo[j]);
        REAL rel = (qpos[7 + j] - mid) * scale; /* upstream: 2 * (pos - mid) / (hi - lo) */
        out28[8 + 2 * j] = rel; out28[9 + 2 * j] = R_(0.1) * qvel[6 + j];
        if (RFABS(rel) > R_(0.99)) ++nlim;
    }
    for (int i = 0; i < 4; ++i) out28[24 + i] = feet_contact[i];
    for (int i = 0; i < 28; ++i) out28[i] = FN(clampr)(out28[i], -5, 5); /* NaN passes through like np.clip */
    *joints_at_limit = nlim;
    for (int i = 0; i < 3; ++i) rpy_out[i] = rpy[i];
}

static REAL FN(u01)(uint32_t x) { return R_(x >> 8) * R_(5.9604644775390625e-08); }

/* uniform pair for (env, index, purpose, sub, attempt) -- counter-based, no stored RNG state */
static void FN(draw_pair)(const hrl_config *cfg, int64_t env, uint32_t index, uint32_t purpose, uint32_t sub, uint32_t attempt, REAL *out) {
    uint32_t r[4];
    orc_philox4x32(cfg->seed, env, index, (purpose << 16) | sub, attempt, r);
    out[0] = FN(u01)(r[0]); out[1] = FN(u01)(r[1]);
}

/* counter-based source of respawn draws: Philox keyed by (env, index, purpose, item, attempt) */
typedef struct FN(orc_philox_src) { const hrl_config *cfg; int64_t env; uint32_t index, purpose; } FN(orc_philox_src);
static int FN(orc_draw_philox)(void *ctx, int item, int attempt, double *pair) {
    FN(orc_philox_src) *p = (FN(orc_philox_src) *)ctx;
    REAL d[2];
    FN(draw_pair)(p->cfg, p->env, p->index, p->purpose, (uint32_t)item, (uint32_t)attempt, d);
    pair[0] = d[0]; pair[1] = d[1];
    return 1;
}
/* gather_scene.py:52-62 with Philox draws; at most 64 attempts (the last is kept) */
static void FN(respawn_item)(const hrl_config *cfg, int64_t env, uint32_t index, uint32_t purpose, int item, const REAL *avoid, REAL *pos) {
    REAL ws[2] = {R_(cfg->world_size[0]), R_(cfg->world_size[1])};
    FN(orc_philox_src) src = {cfg, env, index, purpose};
    FN(random_on_plane_src)(ws, avoid, R_(cfg->robot_object_spacing), FN(orc_draw_philox), &src, item, 64, pos);
}

static void FN(gather_obs_tail)(const hrl_config *cfg, const REAL *xy, REAL yaw, const REAL *items, REAL *obs) {
    REAL d2[HRL_MAX_ITEMS];
    int n = cfg->n_food + cfg->n_poison;
    for (int i = 0; i < n; ++i) d2[i] = FN(orc_sq_dist)(items + 2 * i, xy);
    if (cfg->use_sensor) FN(orc_food_sensor)(cfg->n_bins, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), xy, yaw, items,
                                             cfg->n_food, cfg->n_poison, d2, obs, obs + cfg->n_bins);
    else {
        int nfo = 2 * (cfg->n_food < cfg->n_bins ? cfg->n_food : cfg->n_bins);
        FN(orc_abs_pos)(cfg->n_bins, items, cfg->n_food, cfg->n_poison, d2, obs, obs + nfo);
    }
}

/* the k-th goal of episode `ep` (k = 1, 2, ...): Philox stream shared by ALL envs, like the reference's common
 * RandomState (ant_flagrun_env.py:38-39), with create_target's rejection (:71-78); at most 64 attempts */
static void FN(flag_goal)(const hrl_config *cfg, uint32_t ep, uint32_t k, REAL *g) {
    REAL size = R_(cfg->flag_size);
    for (uint32_t a = 0; a < 64; ++a) {
        uint32_t r[4];
        orc_philox4x32(cfg->seed, 0, ep, (4u << 16) | k, a, r);
        g[0] = -size / 2 + size * FN(u01)(r[0]); g[1] = -size / 2 + size * FN(u01)(r[1]);
        if (!(RSQRT(g[0] * g[0] + g[1] * g[1]) < R_(0.5))) break;
    }
}
void FN(orc_flag_goal)(const hrl_config *cfg, int ep, int k, REAL *g) { FN(flag_goal)(cfg, (uint32_t)ep, (uint32_t)k, g); } /* tests */
/* max_target_dist mode: the k-th goal of episode `ep` of env `env`, drawn around the robot's xy (create_close_target,
 * ant_flagrun_env.py:80-89); one Philox block per attempt (two uniforms, two sign bits), at most 64 attempts, the last kept */
static void FN(flag_close_goal)(const hrl_config *cfg, int64_t env, uint32_t ep, uint32_t k, const REAL *robot_xy, REAL *g) {
    REAL wb = R_(cfg->flag_size) / 2, tol = R_(cfg->tol), half = R_(cfg->flag_max_target_dist) / 2;
    for (uint32_t a = 0; a < 64; ++a) {
        uint32_t r[4];
        orc_philox4x32(cfg->seed, env, ep, (5u << 16) | (k & 0xffffu), a, r);
        g[0] = (tol + (half - tol) * FN(u01)(r[0])) * ((r[2] & 1u) ? R_(1) : R_(-1)) + robot_xy[0];
        g[1] = (tol + (half - tol) * FN(u01)(r[1])) * ((r[3] & 1u) ? R_(1) : R_(-1)) + robot_xy[1];
        if (-wb < g[0] && g[0] < wb && -wb < g[1] && g[1] < wb) break;
    }
}
void FN(orc_flag_close_goal)(const hrl_config *cfg, int64_t env, int ep, int k, const REAL *robot_xy, REAL *g) { FN(flag_close_goal)(cfg, env, (uint32_t)ep, (uint32_t)k, robot_xy, g); } /* tests */
static const REAL FN(maze_lines)[7][4] = { /* MazeScene.bounds: maze_scene.py:15-21 + sizeable_enclosed_scene.py:28-34 */
    {5, 9, -5, 9}, {5, 9, 5, -9}, {-5, -9, -5, 9}, {-5, -9, 5, -9}, {1, 2, 1, -2}, {-5, -2, -5, 2}, {-5, -2, 1, -2}};

/* The potential a reset leaves behind.  upstream WalkerBaseBulletEnv.reset() sets `self.potential = robot.calc_potential()`
 * (= -walk_target_dist / dt) BEFORE the in-tree reset code moves the robot or switches the target:
 *   - AntFlagrunBulletEnv.reset (ant_flagrun_env.py:132-155): super().reset(), teleport (:144), calc_state (:146, the walk
 *     target is still the PREVIOUS goal), next_target(): set_target(new) then `self.potential = calc_potential()` (:116)
 *     re-reads that walk_target_dist: the distance of the NEW pose to the PREVIOUS goal;
 *   - AntMazeBulletEnv.reset (ant_maze_bullet_env.py:104-121), AntMazeMjEnv.reset (ant_maze_mj_env.py:85-104):
 *     super().reset() (:111) computes it at the default pose (0, 0, 0.75) against the PREVIOUS target; the new target
 *     (:114-115), the teleport (:117) and the calc_state (:119) do not touch it.
 * `centroid` = parts centroid of the FINAL reset pose; for the maze kinds the robot's 13 parts stood start_xy away when the
 * potential was taken (same joint angles here: the reference draws the joint noise twice, :111 and :118, this build once).
 * Before the first episode the walk target is upstream's default (1e3, 0) = cfg->walk_target. */
REAL FN(orc_reset_potential)(int maze_kind, const REAL *prev_target, const REAL *centroid, const REAL *start_xy, int n_parts, REAL dt) {
    REAL cx = centroid[0], cy = centroid[1];
    if (maze_kind) { cx = cx - (R_(13) * start_xy[0]) / R_(n_parts); cy = cy - (R_(13) * start_xy[1]) / R_(n_parts); }
    REAL dx = prev_target[0] - cx, dy = prev_target[1] - cy;
    return -RSQRT(dy * dy + dx * dx) / dt;
}

typedef struct FN(orc_env) {
    hrl_config cfg;
    FN(orc_consts) K;
    FN(orc_world) W;
} FN(orc_env);

void FN(orc_env_init)(const hrl_config *cfg, FN(orc_env) * E) {
    E->cfg = *cfg;
    FN(orc_consts_init)(&cfg->model, &E->K);
    FN(orc_world_init)(cfg, &E->W);
}

/* the env keeps `_goal_start_pos` / `_sq_dist_goal` in its items record (HRL_FLAG_START_OFF, HRL_FLAG_SQDIST_OFF) */
static int FN(flag_path_on)(const hrl_config *cfg) { (void)cfg; return 1; } /* set_target() keeps both whatever the weights are (ant_flagrun_env.py:98-103), so a weight switched on for a live env finds them */
/* the goal a flagrun env is chasing: kept in items[0..1] (max_target_dist and manual modes) or the k-th of the shared list */
static void FN(flag_current_goal)(const hrl_config *cfg, const REAL *items, const int32_t *aux, REAL *g) {
    if (cfg->flag_max_target_dist > 0 || cfg->flag_manual_goals) { g[0] = items[0]; g[1] = items[1]; }
    else FN(flag_goal)(cfg, (uint32_t)aux[2], (uint32_t)aux[3] & 0xffffu, g);
}

/* robot.feet_contact as the last step left it: bits 28..31 of aux[1] in the kinds whose observation carries the flags (AntMaze, AntFlagrun).
 * Upstream WalkerBaseBulletEnv.step calls calc_state() BEFORE it refreshes the flags from the step's contacts, so an observation shows the
 * flags of the step before; AntGather never refreshes them (ant_gather_env.py:105-111). */
static void FN(stored_feet)(const hrl_config *cfg, const int32_t *aux, REAL *feet) {
    const int on = cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_FLAGRUN;
    for (int l = 0; l < 4; ++l) feet[l] = (on && (((uint32_t)aux[1] >> (28 + l)) & 1u)) ? R_(1) : R_(0);
}
/* observation of the CURRENT state (used by reset and by step) for ant kinds that do not need step-only data */
static void FN(make_obs)(const FN(orc_env) * E, const REAL *st, const REAL *items, const int32_t *aux, const REAL *feet,
                         REAL *obs, REAL *wtd_out, int *nlim_out, REAL *s28_out, REAL *centroid_out) {
    const hrl_config *cfg = &E->cfg;
    if (cfg->env_kind == HRL_POINT_GATHER) {
        REAL rpy[3], tgt[2] = {0, 0};
        FN(orc_quat_to_rpy)(st + 3, rpy);
        FN(orc_pointbot_state)(st, rpy, st + HRL_QVEL_OFF, tgt, 1, obs);
        FN(gather_obs_tail)(cfg, st, rpy[2], items, obs + 8);
        return;
    }
    REAL s28[28], rpy[3], wtd, tgt[2] = {R_(cfg->walk_target[0]), R_(cfg->walk_target[1])};
    int nlim;
    if (cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_MAZE_MJ) { tgt[0] = R_(cfg->targets[aux[3]][0]); tgt[1] = R_(cfg->targets[aux[3]][1]); }
    if (cfg->env_kind == HRL_ANT_FLAGRUN) FN(flag_current_goal)(cfg, items, aux, tgt);
    FN(orc_ant_calc_state)(cfg, &E->K, st, st + HRL_QVEL_OFF, st[HRL_INITZ_OFF], tgt, feet, s28, &nlim, &wtd, rpy, centroid_out);
    if (wtd_out) *wtd_out = wtd;
    if (nlim_out) *nlim_out = nlim;
    if (s28_out) for (int i = 0; i < 28; ++i) s28_out[i] = s28[i];
    if (cfg->env_kind == HRL_ANT_FLAT) { for (int i = 0; i < 29; ++i) obs[i] = st[i]; return; } /* MjAnt.py:17-25 */
    if (cfg->env_kind == HRL_ANT_FLAGRUN) { /* ant_flagrun_env.py:122-130: the 28-vector (+ wall sensor over the 4 arena lines) */
        for (int i = 0; i < 28; ++i) obs[i] = s28[i];
        if (cfg->use_sensor) {
            REAL hx = R_(cfg->world_size[0]) / 2, hy = R_(cfg->world_size[1]) / 2;
            REAL ln[4][4] = {{hx, hy, -hx, hy}, {hx, hy, hx, -hy}, {-hx, -hy, -hx, hy}, {-hx, -hy, hx, -hy}}; /* sizeable_enclosed_scene.py:28-34 */
            FN(orc_sense_walls)(cfg->n_bins, FN(cfg_angle)(cfg->sensor_span), R_(cfg->sensor_range), st, rpy[2], &ln[0][0], 4,
                                (double)cfg->sensor_span == (double)(float)6.283185307179586, obs + 28);
        }
        return;
    }
    if (cfg->env_kind == HRL_ANT_MAZE_MJ) {
        REAL r_; int d_;
        FN(orc_maze_mj_task)(cfg, st, rpy[2], 0, 0, wtd, aux[0], &FN(maze_lines)[0][0], 7, obs, &r_, &d_);
        return;
    }
    if (cfg->env_kind == HRL_ANT_GATHER) {
        obs[0] = s28[0];
        for (int i = 3; i < 28; ++i) obs[i - 2] = s28[i];
        FN(gather_obs_tail)(cfg, st, rpy[2], items, obs + 26);
        return;
    }
    /* maze: ant_maze_bullet_env.py:63-75 (reward/done unused here) */
    REAL r; int d;
    FN(orc_maze_task)(cfg, s28, 0, 0, st, rpy[2], tgt, wtd, -1000000, &FN(maze_lines)[0][0], 7, 3, obs, &r, &d);
}

void FN(orc_env_reset_one)(const FN(orc_env) * E, int64_t env, REAL *st, REAL *items, int32_t *aux, REAL *obs) {
    const hrl_config *cfg = &E->cfg;
    uint32_t ep = (uint32_t)aux[2];
    const int maze_kind = cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_MAZE_MJ;
    /* what the robot was walking towards before this reset (see orc_reset_potential) */
    REAL prev_target[2] = {R_(cfg->walk_target[0]), R_(cfg->walk_target[1])};
    if (ep > 0) {
        if (maze_kind) { prev_target[0] = R_(cfg->targets[aux[3]][0]); prev_target[1] = R_(cfg->targets[aux[3]][1]); }
        if (cfg->env_kind == HRL_ANT_FLAGRUN) FN(flag_current_goal)(cfg, items, aux, prev_target);
    }
    for (int i = 0; i < HRL_STATE_STRIDE; ++i) st[i] = 0;
    st[6] = 1; /* identity quaternion (x,y,z,w) */
    if (cfg->env_kind == HRL_POINT_GATHER) { st[2] = R_(0.5); st[HRL_INITZ_OFF] = 1; } /* point_bot.py:12,18 */
    else {
        if (cfg->env_kind == HRL_ANT_FLAGRUN) { /* ant_flagrun_env.py:132-155: start (0,0,0.25), first goal popped */
            /* manual: goals.clear() (:150), nothing popped; else the first goal of the new list; _rewarded cleared (:137);
             * steps_since_goal_change is NOT reset by reset() (:132-155 never assigns it) */
            aux[3] = (int32_t)((cfg->flag_manual_goals ? 0u : 1u) | ((uint32_t)aux[3] & 0x7fff0000u));
            st[0] = R_(cfg->start_pos[0]); st[1] = R_(cfg->start_pos[1]); st[2] = R_(cfg->start_pos[2]);
        } else if (maze_kind) { /* ant_maze_bullet_env.py:108-118, ant_maze_mj_env.py:85-101 */
            uint32_t r[4];
            orc_philox4x32(cfg->seed, env, ep, (3u << 16), 0, r);
            aux[3] = (int32_t)(r[0] % (uint32_t)cfg->n_targets);
            st[0] = R_(cfg->start_pos[0]); st[1] = R_(cfg->start_pos[1]); st[2] = R_(cfg->start_pos[2]);
        } else st[2] = R_(0.75); /* assets/ant.xml:12 */
        for (int j = 0; j < NJ; ++j) { /* upstream robot_specific_reset: joints ~ U(-0.1, 0.1), zero rates */
            uint32_t r[4];
            orc_philox4x32(cfg->seed, env, ep, (2u << 16) | (uint32_t)(j / 4), 0, r);
            st[7 + j] = R_(-0.1) + R_(0.2) * FN(u01)(r[j % 4]);
        }
        st[HRL_INITZ_OFF] = st[2];
    }
    if (cfg->env_kind == HRL_ANT_GATHER || cfg->env_kind == HRL_POINT_GATHER) { /* gather_scene.py:38-50 */
        REAL origin[2] = {0, 0};
        for (int i = 0; i < orc_items_stride(cfg); ++i) items[i] = 0;
        for (int i = 0; i < cfg->n_food + cfg->n_poison; ++i) FN(respawn_item)(cfg, env, ep, 1, i, origin, items + 2 * i);
    }
    aux[0] = 0; aux[2] = (int32_t)(ep + 1); /* NB: the flagrun goal stream is keyed by the NEW episode index */
    if (cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_FLAGRUN) aux[1] &= 0x0fffffff; /* robot.feet_contact = 0 (upstream robot_specific_reset) */
    if (cfg->env_kind == HRL_ANT_FLAGRUN && items && FN(flag_path_on)(cfg)) { /* an env of the shared goal list keeps set_target()'s bookkeeping in the record, not the goal */
        if (cfg->flag_manual_goals) { /* the walk target and the path-reward state survive the reset (:149-152), pending goals do not */
            for (int i = HRL_FLAG_PENDING_OFF; i < orc_items_stride(cfg); ++i) items[i] = 0;
            items[0] = prev_target[0]; items[1] = prev_target[1];
        } else { /* reset -> next_target -> set_target(first goal of the new episode) at the start pose (:144-153): create_close_target (:111-112) or the shared list */
            REAL g2[2];
            for (int i = 0; i < orc_items_stride(cfg); ++i) items[i] = 0;
            if (cfg->flag_max_target_dist > 0) { FN(flag_close_goal)(cfg, env, ep + 1, 1, st, g2); items[0] = g2[0]; items[1] = g2[1]; }
            else FN(flag_goal)(cfg, ep + 1, 1, g2);
            FN(orc_flag_set_target_state)(g2, st, items + HRL_FLAG_START_OFF);
        }
    }
    REAL feet[4], wtd = 0, cen[2] = {0, 0};
    FN(stored_feet)(cfg, aux, feet); /* cleared above */
    FN(make_obs)(E, st, items, aux, feet, obs, &wtd, 0, 0, cen);
    const REAL dt = E->K.h * R_(E->K.nsub);
    if (cfg->env_kind == HRL_ANT_GATHER || cfg->env_kind == HRL_POINT_GATHER) st[HRL_POTENTIAL_OFF] = 0; /* never read */
    else if (cfg->env_kind == HRL_ANT_FLAT) st[HRL_POTENTIAL_OFF] = -wtd / dt; /* upstream reset: calc_potential at the reset pose */
    else {
        REAL sxy[2] = {R_(cfg->start_pos[0]), R_(cfg->start_pos[1])};
        st[HRL_POTENTIAL_OFF] = FN(orc_reset_potential)(maze_kind, prev_target, cen, sxy, 13 + cfg->centroid_n_static, dt);
    }
}

/* `env.goals = [...]` of a manual_goal_creation flagrun env (ant_flagrun_env.py:45,96,150): the pending list, in list order,
 * behind the current goal and the path-reward state in the items record (items[HRL_FLAG_PENDING_OFF + 2k..] = goals[k]); its length in the low 16 bits of aux[3]. */
void FN(orc_flag_goals_assign)(const hrl_config *cfg, REAL *items, int32_t *aux, const REAL *goals_xy, int n_goals) {
    for (int k = 0; k < n_goals; ++k) { items[HRL_FLAG_PENDING_OFF + 2 * k] = goals_xy[2 * k]; items[HRL_FLAG_PENDING_OFF + 1 + 2 * k] = goals_xy[2 * k + 1]; }
    for (int i = HRL_FLAG_PENDING_OFF + 2 * n_goals; i < orc_items_stride(cfg); ++i) items[i] = 0;
    aux[3] = (int32_t)(((uint32_t)n_goals & 0xffffu) | ((uint32_t)aux[3] & 0xffff0000u));
}

/* `env.next_target()` (ant_flagrun_env.py:112-120) called from outside step(): max_targets < 1 -> set_target(*create_close_target())
 * (:113-114, the list is ignored); else set_target(*self.goals.pop()) -- the LAST element of the list (:116) -- or IndexError
 * when it is empty (returns 0, nothing changed).  _rewarded is cleared (:118); steps_since_goal_change is not touched; the
 * potential is left alone (:119 re-reads the walk_target_dist of the last calc_state, which is what the stored potential was
 * computed from).  Returns 1 on success. */
int FN(orc_flag_next_target)(const hrl_config *cfg, int64_t env, const REAL *st, REAL *items, int32_t *aux) {
    uint32_t a3 = (uint32_t)aux[3], cur = a3 & 0xffffu;
    REAL g2[2];
    if (cfg->flag_max_target_dist > 0) { /* max_targets < 1 (:17-18): the goal counter of the episode keys the draw */
        cur = (cur + 1) & 0xffffu;
        FN(flag_close_goal)(cfg, env, (uint32_t)aux[2], cur, st, g2);
        items[0] = g2[0]; items[1] = g2[1];
    } else if (!cfg->flag_manual_goals) { /* the shared list reset() made (:91-96,150-153): `cur` of its max_targets goals are used up */
        if (cur >= (uint32_t)cfg->flag_max_targets) return 0; /* goals.pop() on an empty list */
        cur += 1;
        FN(flag_goal)(cfg, (uint32_t)aux[2], cur, g2);
    } else {
        if (cur == 0) return 0;
        cur -= 1;
        g2[0] = items[HRL_FLAG_PENDING_OFF + 2 * cur]; g2[1] = items[HRL_FLAG_PENDING_OFF + 1 + 2 * cur];
        items[0] = g2[0]; items[1] = g2[1];
    }
    if (items && FN(flag_path_on)(cfg)) FN(orc_flag_set_target_state)(g2, st, items + HRL_FLAG_START_OFF); /* set_target (:98-103) with the robot where it is */
    aux[3] = (int32_t)(cur | (a3 & 0x7fff0000u));
    return 1;
}

/* hrl_set_goals() of include/hrl_envs.h: `env.goals = [...]; env.next_target()`; the returned state is calc_state() towards
 * the new goal (:120). */
void FN(orc_env_set_goals_one)(const FN(orc_env) * E, int64_t env, REAL *st, REAL *items, int32_t *aux, const REAL *goals_xy, int n_goals, REAL *obs) {
    FN(orc_flag_goals_assign)(&E->cfg, items, aux, goals_xy, n_goals);
    FN(orc_flag_next_target)(&E->cfg, env, st, items, aux);
    REAL feet[4];
    FN(stored_feet)(&E->cfg, aux, feet); /* next_target() -> robot.calc_state() with the flags as the last step left them */
    FN(make_obs)(E, st, items, aux, feet, obs, 0, 0, 0, 0);
}
/* hrl_next_target() of include/hrl_envs.h: `env.next_target()` alone; ok = 0 where the reference raises IndexError */
void FN(orc_env_next_target_one)(const FN(orc_env) * E, int64_t env, REAL *st, REAL *items, int32_t *aux, REAL *obs, uint8_t *ok) {
    const int r = FN(orc_flag_next_target)(&E->cfg, env, st, items, aux);
    if (ok) *ok = (uint8_t)r;
    REAL feet[4];
    FN(stored_feet)(&E->cfg, aux, feet);
    FN(make_obs)(E, st, items, aux, feet, obs, 0, 0, 0, 0);
}

/* One env step on the packed record.  Mirrors hrl_step() of include/hrl_envs.h. */
void FN(orc_env_step_one)(const FN(orc_env) * E, int64_t env, REAL *st, REAL *items, int32_t *aux, const REAL *act,
                          REAL *obs, REAL *rew_out, uint8_t *done_out, REAL *info, REAL *final_obs, uint8_t *truncated, REAL *goal_out, int32_t *rows_out) {
    const hrl_config *cfg = &E->cfg;
    const FN(orc_consts) *K = &E->K;
    REAL q[15], u[14], feet[4], feet_new[4] = {0, 0, 0, 0};
    FN(stored_feet)(cfg, aux, feet); /* what calc_state() sees: the flags of the step BEFORE (upstream refreshes them after calc_state) */
    uint32_t t_life = (uint32_t)aux[1];
    const int gather = cfg->env_kind == HRL_ANT_GATHER || cfg->env_kind == HRL_POINT_GATHER;
    const int n_items = gather ? cfg->n_food + cfg->n_poison : 0;
    FN(orc_substep_dbg) dbg;
    int n_rows_step = 0; /* solver rows over the step's substeps (hrl_buffers.solver_rows) */
    memset(&dbg, 0, sizeof(dbg));
    for (int i = 0; i < 15; ++i) q[i] = st[i];
    for (int k = 0; k < 3; ++k) { u[k] = st[HRL_QVEL_OFF + 3 + k]; u[3 + k] = st[HRL_QVEL_OFF + k]; }
    for (int j = 0; j < NJ; ++j) u[6 + j] = st[HRL_QVEL_OFF + 6 + j];
    if (cfg->env_kind == HRL_POINT_GATHER) { /* point_bot.py:28-31: a/|a|*500 N in the world xy plane */
        REAL n = RSQRT(act[0] * act[0] + act[1] * act[1]), f[3] = {act[0] / n * R_(cfg->model.point_force), act[1] / n * R_(cfg->model.point_force), 0};
        for (int s = 0; s < K->nsub; ++s) { FN(orc_point_substep)(K, &E->W, q, u, f, items, n_items, &dbg); n_rows_step += dbg.n_rows; }
    } else {
        REAL tau[NJ];
        for (int j = 0; j < NJ; ++j) tau[j] = R_(cfg->model.torque_scale) * FN(clampr)(act[j], -1, 1);
        int gt[13];
        for (int s = 0; s < K->nsub; ++s) { FN(orc_ant_substep)(K, &E->W, q, u, tau, gather ? items : 0, n_items, gt, &dbg); n_rows_step += dbg.n_rows; }
        for (int l = 0; l < 4; ++l) feet_new[l] = (gt[2 + 3 * l] || gt[3 + 3 * l]) ? R_(1) : R_(0); /* this step's: feet against the floor in its last collision pass */
    }
    for (int i = 0; i < 15; ++i) st[i] = q[i];
    for (int k = 0; k < 3; ++k) { st[HRL_QVEL_OFF + 3 + k] = u[k]; st[HRL_QVEL_OFF + k] = u[3 + k]; }
    for (int j = 0; j < NJ; ++j) st[HRL_QVEL_OFF + 6 + j] = u[6 + j];

    REAL rew = 0, food_rew = 0, dead_rew = 0;
    int done = 0;
    if (gather) { /* the task half is the golden-pinned orc_gather_task (ant_gather_env.py:81-119, gather_base.py:80-109) */
        const int ant = cfg->env_kind == HRL_ANT_GATHER;
        REAL base[28], rpy[3], wtd, tgt[2] = {0, 0};
        int nlim, contact_items[MAXC];
        if (ant) { feet[0] = feet[1] = feet[2] = feet[3] = 0; /* ant_gather_env.py:105-111 */
                   FN(orc_ant_calc_state)(cfg, K, st, st + HRL_QVEL_OFF, st[HRL_INITZ_OFF], tgt, feet, base, &nlim, &wtd, rpy, 0); }
        else { FN(orc_quat_to_rpy)(st + 3, rpy); FN(orc_pointbot_state)(st, rpy, st + HRL_QVEL_OFF, tgt, 1, base); }
        /* getContactPoints after stepSimulation = the contacts of the step's last collision pass (:114) */
        for (int c = 0; c < dbg.n_contacts; ++c) contact_items[c] = ORC_ITEM_OF_SURF(dbg.surface[c]);
        FN(orc_philox_src) src = {cfg, env, t_life, 0};
        FN(orc_gather_task_src)(cfg, ant, base, ant ? 28 : 8, st, rpy[2], st[HRL_INITZ_OFF], ant ? R_(0.26) : R_(-1), items, FN(orc_draw_philox), &src, 64,
                                contact_items, dbg.n_contacts, obs, &rew, &done, &food_rew, &dead_rew);
    } else if (cfg->env_kind == HRL_ANT_FLAT) {
        REAL wtd; int nlim;
        FN(make_obs)(E, st, items, aux, feet, obs, &wtd, &nlim, 0, 0);
        REAL pot = -wtd / (K->h * R_(K->nsub));
        FN(orc_antmj_reward)(obs, st[HRL_POTENTIAL_OFF], pot, nlim, R_(-0.1), &rew, &done);
        food_rew = st[2] > R_(0.26) ? R_(1) : R_(-1); dead_rew = pot - st[HRL_POTENTIAL_OFF]; /* info[0..1] of the locomotion kinds: alive, progress (`self.rewards[0:2]`, MjAnt.py:82-84) */
        st[HRL_POTENTIAL_OFF] = pot;
    } else if (cfg->env_kind == HRL_ANT_FLAGRUN) { /* upstream WalkerBaseBulletEnv.step with its cost weights zeroed
                                                     (ant_flagrun_env.py:133-135), then ant_flagrun_env.py:162-204 */
        REAL wtd, s28[28], rpy[3], tgt[2];
        int nlim, steps = (aux[3] >> 16) & 0x7fff, rewarded = (aux[3] >> 31) & 1, cur = aux[3] & 0xffff, retarget;
        const int close_mode = cfg->flag_max_target_dist > 0, manual = cfg->flag_manual_goals && !close_mode; /* manual: the pending list is popped (:116); max_targets < 1: create_close_target whoever made the env (:113-114) */
        FN(flag_current_goal)(cfg, items, aux, tgt);
        FN(orc_ant_calc_state)(cfg, K, st, st + HRL_QVEL_OFF, st[HRL_INITZ_OFF], tgt, feet, s28, &nlim, &wtd, rpy, 0);
        REAL alive = (s28[0] + st[HRL_INITZ_OFF] > R_(0.26)) ? R_(1) : R_(-1);
        int idone = alive < 0;
        for (int i = 0; i < 28; ++i) if (!isfinite(s28[i])) idone = 1;
        REAL pot = -wtd / (K->h * R_(K->nsub)), progress = pot - st[HRL_POTENTIAL_OFF];
        food_rew = alive; dead_rew = progress; /* info[0..1]: the first two entries of upstream's `self.rewards` */
        st[HRL_POTENTIAL_OFF] = pot; /* next_target() re-reads the same stale potential (:116), i.e. leaves it unchanged */
        const int budget = close_mode ? (1 << 20) : cfg->flag_max_targets; /* max_target_dist mode never runs out (:113-114) */
        int goals_left = manual ? cur : budget - cur; /* manual: `cur` counts the pending goals */
        const int with_path = FN(flag_path_on)(cfg);
        const REAL path = with_path ? FN(orc_flag_path_rew)(st, tgt, items + HRL_FLAG_START_OFF, items[HRL_FLAG_SQDIST_OFF]) : 0;
        REAL fe1 = 0, fe2 = 0; /* upstream WalkerBaseBulletEnv.step with the cost weights AntFlagrunBulletEnv.reset() left on the class (hrl_config.walker_*: 0, 0, 0 unless set otherwise) */
        for (int j = 0; j < NJ; ++j) { REAL a = act[j]; fe1 += RFABS(a * s28[9 + 2 * j]); fe2 += a * a; }
        const REAL felec = R_(cfg->walker_electricity_cost) * (fe1 / NJ) + R_(cfg->walker_stall_torque_cost) * (fe2 / NJ);
        const REAL finner = (((alive + progress) + felec) + R_(cfg->walker_joints_at_limit_cost) * R_(nlim)) + 0;
        FN(orc_flagrun_task_w)(cfg, finner, idone, wtd, with_path, path, &steps, &rewarded, &goals_left, &rew, &done, &retarget);
        if (steps > 0x7fff) steps = 0x7fff; /* steps_since_goal_change saturates in its 15-bit field (only reachable with the timeout off) */
        if (retarget) { /* the next_target() of :190 / :198: goals.pop() (:116), create_close_target around the robot's xy (:113-114) or the shared list's next goal; set_target (:98-103) */
            FN(orc_flag_next_target)(cfg, env, st, items, aux);
            cur = aux[3] & 0xffff;
        } else cur = manual ? goals_left : budget - goals_left;
        aux[3] = (int32_t)(((uint32_t)cur & 0xffffu) | ((uint32_t)steps << 16) | ((uint32_t)rewarded << 31));
        FN(make_obs)(E, st, items, aux, retarget ? feet_new : feet, obs, 0, 0, 0, 0); /* the state of super().step(), or -- a switch -- next_target()'s calc_state()
                                                                                        w.r.t. the new goal, which runs after the flags were refreshed */
        if (goal_out) { /* info['target'] = self.goal on the steps that switched goals (:191,199) */
            REAL g2[2];
            FN(flag_current_goal)(cfg, items, aux, g2);
            goal_out[0] = g2[0]; goal_out[1] = g2[1]; goal_out[2] = retarget ? R_(1) : R_(0); goal_out[3] = R_(steps);
        }
    } else if (cfg->env_kind == HRL_ANT_MAZE_MJ) { /* MjAnt.py:36-97 then ant_maze_mj_env.py:66-78 */
        REAL wtd, s28[28], rpy[3], inner, tgt[2] = {R_(cfg->targets[aux[3]][0]), R_(cfg->targets[aux[3]][1])};
        int nlim, idone;
        FN(orc_ant_calc_state)(cfg, K, st, st + HRL_QVEL_OFF, st[HRL_INITZ_OFF], tgt, feet, s28, &nlim, &wtd, rpy, 0);
        REAL pot = -wtd / (K->h * R_(K->nsub));
        FN(orc_antmj_reward)(st, st[HRL_POTENTIAL_OFF], pot, nlim, R_(-0.1), &inner, &idone);
        food_rew = st[2] > R_(0.26) ? R_(1) : R_(-1); dead_rew = pot - st[HRL_POTENTIAL_OFF];
        st[HRL_POTENTIAL_OFF] = pot;
        FN(orc_maze_mj_task)(cfg, st, rpy[2], inner, idone, wtd, aux[0], &FN(maze_lines)[0][0], 7, obs, &rew, &done);
    } else { /* maze: upstream WalkerBaseBulletEnv.step then ant_maze_bullet_env.py:77-97 */
        REAL wtd, s28[28], rpy[3], tgt[2] = {R_(cfg->targets[aux[3]][0]), R_(cfg->targets[aux[3]][1])};
        int nlim;
        FN(orc_ant_calc_state)(cfg, K, st, st + HRL_QVEL_OFF, st[HRL_INITZ_OFF], tgt, feet, s28, &nlim, &wtd, rpy, 0);
        REAL alive = (s28[0] + st[HRL_INITZ_OFF] > R_(0.26)) ? R_(1) : R_(-1);
        int idone = alive < 0;
        for (int i = 0; i < 28; ++i) if (!isfinite(s28[i])) idone = 1;
        REAL pot = -wtd / (K->h * R_(K->nsub)), progress = pot - st[HRL_POTENTIAL_OFF];
        food_rew = alive; dead_rew = progress;
        st[HRL_POTENTIAL_OFF] = pot;
        REAL e1 = 0, e2 = 0;
        for (int j = 0; j < NJ; ++j) { REAL a = act[j]; e1 += RFABS(a * s28[9 + 2 * j]); e2 += a * a; }
        REAL electricity = R_(cfg->walker_electricity_cost) * (e1 / NJ) + R_(cfg->walker_stall_torque_cost) * (e2 / NJ);
        REAL inner = (((alive + progress) + electricity) + R_(cfg->walker_joints_at_limit_cost) * R_(nlim)) + 0;
        FN(orc_maze_task)(cfg, s28, inner, idone, st, rpy[2], tgt, wtd, aux[0] + 1, &FN(maze_lines)[0][0], 7, 3, obs, &rew, &done);
    }
    aux[0] += 1; aux[1] += 1;
    if (cfg->env_kind == HRL_ANT_MAZE || cfg->env_kind == HRL_ANT_FLAGRUN) { /* robot.feet_contact for the next step's observation */
        uint32_t bits = 0;
        for (int l = 0; l < 4; ++l) bits |= (feet_new[l] != 0 ? 1u : 0u) << l;
        aux[1] = (int32_t)(((uint32_t)aux[1] & 0x0fffffffu) | (bits << 28));
    }
    int trunc = 0; /* gym TimeLimit, __init__.py:15 (gym.wrappers.TimeLimit: info['TimeLimit.truncated'] = not done; done = True) */
    if (cfg->max_episode_steps > 0 && aux[0] >= cfg->max_episode_steps) { trunc = !done; done = 1; }
    st[HRL_EPRET_OFF] += rew;
    *rew_out = rew; *done_out = (uint8_t)done;
    if (truncated) *truncated = (uint8_t)trunc;
    if (rows_out) *rows_out += n_rows_step;
    info[0] = food_rew; info[1] = dead_rew; info[2] = st[HRL_EPRET_OFF]; info[3] = R_(aux[0]); /* (locomotion kinds: alive, progress in the first two) */
    /* what step() returns in the reference is the state of THIS step (ant_gather_env.py:96,118-119, ant_maze_bullet_env.py:82,97): kept in
     * final_obs when the episode ends, because the in-place reset below overwrites obs with the next episode's first observation */
    if (done && final_obs) { const int od = orc_obs_dim(cfg); for (int i = 0; i < od; ++i) final_obs[i] = obs[i]; }
    if (done && cfg->auto_reset) FN(orc_env_reset_one)(E, env, st, items, aux, obs);
}

/* batched drivers over host buffers laid out exactly like the device buffers */
void FN(orc_reset_batch)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const uint8_t *mask, REAL *obs) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    int od = orc_obs_dim(cfg);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < cfg->num_envs; ++i) {
        if (mask && !mask[i]) continue;
        FN(orc_env_reset_one)(&E, cfg->env_id_offset + i, state + (size_t)i * HRL_STATE_STRIDE,
                              items ? items + (size_t)i * orc_items_stride(cfg) : 0, aux + (size_t)i * HRL_AUX_STRIDE, obs + (size_t)i * od);
    }
}
/* hrl_observe of include/hrl_envs.h: the observation of the records as they stand, nothing else written (feet-contact entries: the flags the
 * last step left, as robot.feet_contact holds them in the reference; 0 after a reset) */
void FN(orc_observe_batch)(const hrl_config *cfg, const REAL *state, const REAL *items, const int32_t *aux, const uint8_t *mask, REAL *obs) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    int od = orc_obs_dim(cfg);
    REAL zero_items[2 * HRL_MAX_ITEMS] = {0};
    for (int i = 0; i < cfg->num_envs; ++i) {
        if (mask && !mask[i]) continue;
        REAL feet[4];
        FN(stored_feet)(cfg, aux + (size_t)i * HRL_AUX_STRIDE, feet);
        FN(make_obs)(&E, state + (size_t)i * HRL_STATE_STRIDE, items ? items + (size_t)i * orc_items_stride(cfg) : zero_items, aux + (size_t)i * HRL_AUX_STRIDE, feet, obs + (size_t)i * od, 0, 0, 0, 0);
    }
}
void FN(orc_step_batch_v7)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const REAL *actions, REAL *obs,
                           REAL *reward, uint8_t *done, REAL *info, REAL *final_obs, uint8_t *truncated, REAL *goal, int32_t *solver_rows) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    int od = orc_obs_dim(cfg), ad = orc_act_dim(cfg);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < cfg->num_envs; ++i)
        FN(orc_env_step_one)(&E, cfg->env_id_offset + i, state + (size_t)i * HRL_STATE_STRIDE,
                             items ? items + (size_t)i * orc_items_stride(cfg) : 0, aux + (size_t)i * HRL_AUX_STRIDE,
                             actions + (size_t)i * ad, obs + (size_t)i * od, reward + i, done + i, info + (size_t)i * HRL_INFO_STRIDE,
                             final_obs ? final_obs + (size_t)i * od : 0, truncated ? truncated + i : 0, goal ? goal + (size_t)i * HRL_GOAL_STRIDE : 0, solver_rows ? solver_rows + i : 0);
}
void FN(orc_step_batch_v6)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const REAL *actions, REAL *obs,
                           REAL *reward, uint8_t *done, REAL *info, REAL *final_obs, uint8_t *truncated) {
    FN(orc_step_batch_v7)(cfg, state, items, aux, actions, obs, reward, done, info, final_obs, truncated, 0, 0);
}
void FN(orc_step_batch)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const REAL *actions, REAL *obs,
                        REAL *reward, uint8_t *done, REAL *info) {
    FN(orc_step_batch_v6)(cfg, state, items, aux, actions, obs, reward, done, info, 0, 0);
}

void FN(orc_set_goals_batch)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const REAL *goals_xy, int n_goals, const uint8_t *mask, REAL *obs) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    int od = orc_obs_dim(cfg);
    for (int i = 0; i < cfg->num_envs; ++i) {
        if (mask && !mask[i]) continue;
        FN(orc_env_set_goals_one)(&E, cfg->env_id_offset + i, state + (size_t)i * HRL_STATE_STRIDE, items + (size_t)i * orc_items_stride(cfg), aux + (size_t)i * HRL_AUX_STRIDE,
                                  goals_xy + (size_t)i * n_goals * 2, n_goals, obs + (size_t)i * od);
    }
}
void FN(orc_next_target_batch)(const hrl_config *cfg, REAL *state, REAL *items, int32_t *aux, const uint8_t *mask, REAL *obs, uint8_t *ok) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    int od = orc_obs_dim(cfg);
    for (int i = 0; i < cfg->num_envs; ++i) {
        if (mask && !mask[i]) continue;
        FN(orc_env_next_target_one)(&E, cfg->env_id_offset + i, state + (size_t)i * HRL_STATE_STRIDE, items + (size_t)i * orc_items_stride(cfg), aux + (size_t)i * HRL_AUX_STRIDE,
                                    obs + (size_t)i * od, ok ? ok + i : 0);
    }
}

/* CPU baseline loop for bench.py: reset, then `steps` env steps of all cfg->num_envs envs with U(-1,1) actions from a
 * 64-bit LCG, on `threads` OpenMP threads.  Returns wall seconds of the stepping loop; *checksum guards against DCE. */
double FN(orc_bench)(const hrl_config *cfg, int steps, int threads, double *checksum) {
    int n = cfg->num_envs, od = orc_obs_dim(cfg), ad = orc_act_dim(cfg);
    REAL *state = calloc((size_t)n * HRL_STATE_STRIDE, sizeof(REAL)), *items = calloc((size_t)n * orc_items_stride(cfg), sizeof(REAL));
    REAL *obs = calloc((size_t)n * od, sizeof(REAL)), *rew = calloc(n, sizeof(REAL)), *info = calloc((size_t)n * HRL_INFO_STRIDE, sizeof(REAL));
    REAL *act = calloc((size_t)n * ad, sizeof(REAL));
    int32_t *aux = calloc((size_t)n * HRL_AUX_STRIDE, sizeof(int32_t));
    uint8_t *done = calloc(n, 1);
    omp_set_num_threads(threads > 0 ? threads : 1);
    FN(orc_reset_batch)(cfg, state, items, aux, 0, obs);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    double t0 = omp_get_wtime(), sum = 0;
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < n * ad; ++i) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; act[i] = R_((double)(lcg >> 11) * (2.0 / 9007199254740992.0) - 1.0); }
        FN(orc_step_batch)(cfg, state, items, aux, act, obs, rew, done, info);
        sum += rew[0];
    }
    double dt = omp_get_wtime() - t0;
    for (int i = 0; i < n; ++i) sum += state[(size_t)i * HRL_STATE_STRIDE + 2];
    if (checksum) *checksum = sum;
    free(state); free(items); free(obs); free(rew); free(info); free(act); free(aux); free(done);
    return dt;
}

/* ---------------------------------------------------------------------------------------------- KAT helpers (tests) */
/* kinetic energy + linear/angular momentum (about the world origin) of the ant by plain per-body sums */
void FN(orc_ant_energy_momentum)(const hrl_model *M, const REAL *q, const REAL *u, REAL *out /* T, V, P[3], L[3] */) {
    FN(orc_consts) K; FN(orc_dyn) D;
    FN(orc_consts_init)(M, &K);
    FN(orc_dynamics)(&K, q, u, 0, &D);
    REAL T = 0, V = 0, P[3] = {0, 0, 0}, L[3] = {0, 0, 0};
    for (int b = 0; b < NBODY; ++b) {
        REAL m, al, be, e[3], c[3], w[3], vO[3];
        for (int k = 0; k < 3; ++k) { w[k] = u[k]; vO[k] = u[3 + k]; }
        if (b == 0) { m = K.m0; al = K.a0; be = K.b0; for (int k = 0; k < 3; ++k) { e[k] = D.Z[k]; c[k] = 0; } }
        else {
            int l = (b - 1) / 2, foot = (b - 1) & 1;
            const REAL *p0 = foot ? D.pa[l] : D.ph[l], *p1 = foot ? D.tip[l] : D.pa[l];
            REAL len = foot ? K.L2 : K.L1;
            m = foot ? K.m2 : K.m1; al = foot ? K.a2 : K.a1; be = foot ? K.b2 : K.b1;
            for (int k = 0; k < 3; ++k) { e[k] = (p1[k] - p0[k]) / len; c[k] = R_(0.5) * (p0[k] + p1[k]); }
            for (int k = 0; k < 6; ++k) { REAL sv = D.S[2 * l][k] * u[6 + 2 * l]; if (k < 3) w[k] += sv; else vO[k - 3] += sv; }
            if (foot) for (int k = 0; k < 6; ++k) { REAL sv = D.S[2 * l + 1][k] * u[7 + 2 * l]; if (k < 3) w[k] += sv; else vO[k - 3] += sv; }
        }
        REAL wxc[3], vc[3], Iw[3], ew = FN(v3dot)(e, w), pw[3], cw[3], lw[3];
        FN(v3cross)(wxc, w, c);
        for (int k = 0; k < 3; ++k) { vc[k] = vO[k] + wxc[k]; Iw[k] = al * w[k] + be * e[k] * ew; pw[k] = q[k] + c[k]; }
        T += R_(0.5) * (m * FN(v3dot)(vc, vc) + FN(v3dot)(w, Iw));
        V += m * K.g * pw[2];
        for (int k = 0; k < 3; ++k) { cw[k] = m * vc[k]; P[k] += cw[k]; }
        FN(v3cross)(lw, pw, cw);
        for (int k = 0; k < 3; ++k) L[k] += lw[k] + Iw[k];
    }
    out[0] = T; out[1] = V; for (int k = 0; k < 3; ++k) { out[2 + k] = P[k]; out[5 + k] = L[k]; }
}
/* hip point, ankle point and foot tip of every leg in world coordinates, [4][3][3] (tests) */
void FN(orc_ant_leg_points)(const hrl_model *M, const REAL *q, REAL *out36) {
    FN(orc_consts) K; FN(orc_dyn) D;
    REAL u0[14] = {0};
    FN(orc_consts_init)(M, &K);
    FN(orc_dynamics)(&K, q, u0, 0, &D);
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < 3; ++k) { out36[9 * l + k] = q[k] + D.ph[l][k]; out36[9 * l + 3 + k] = q[k] + D.pa[l][k]; out36[9 * l + 6 + k] = q[k] + D.tip[l][k]; }
}
/* how many of the contacts the collision pass of pose q keeps are SECOND support points of capsules lying flat on a box face (tests: that a case exercises them) */
int FN(orc_ant_second_points)(const hrl_config *cfg, const REAL *q, const REAL *items_xy, int n_items) {
    FN(orc_consts) K; FN(orc_dyn) D; FN(orc_world) W; FN(orc_contact) C[MAXC];
    REAL u0[14] = {0};
    int gt[13], ncand = 0, n2 = 0;
    FN(orc_consts_init)(&cfg->model, &K);
    FN(orc_world_init)(cfg, &W);
    FN(orc_dynamics)(&K, q, u0, 0, &D);
    const int nc = FN(orc_detect)(&K, &W, &D, q, items_xy, n_items, C, gt, &ncand);
    for (int c = 0; c < nc; ++c) n2 += C[c].second;
    return n2;
}
/* the hard-wired ant model as numbers (tests/test_assets.py holds them against assets/ant.xml): radii, capsule lengths, masses and central
 * inertias (alpha, beta) of the three body types, joint ranges [rad] */
void FN(orc_model_constants)(const hrl_model *M, REAL *out29) {
    FN(orc_consts) K;
    FN(orc_consts_init)(M, &K);
    const REAL v[13] = {K.r_torso, K.r_caps, K.L1, K.L2, K.m0, K.a0, K.b0, K.m1, K.a1, K.b1, K.m2, K.a2, K.b2};
    for (int k = 0; k < 13; ++k) out29[k] = v[k];
    for (int j = 0; j < NJ; ++j) { out29[13 + j] = K.lo[j]; out29[21 + j] = K.hi[j]; }
}
/* accelerations [a0(6) | qdd(8)] for tests */
void FN(orc_ant_accel)(const hrl_model *M, const REAL *q, const REAL *u, const REAL *tau, REAL *out14) {
    FN(orc_consts) K; FN(orc_dyn) D;
    FN(orc_consts_init)(M, &K);
    FN(orc_dynamics)(&K, q, u, tau, &D);
    for (int k = 0; k < 6; ++k) out14[k] = D.a0[k];
    for (int j = 0; j < NJ; ++j) out14[6 + j] = D.qdd[j];
}
/* M^-1 (14x14, row-major) assembled column by column from unit generalized impulses via orc_response */
void FN(orc_ant_minv)(const hrl_model *M, const REAL *q, REAL *out196) {
    FN(orc_consts) K; FN(orc_dyn) D;
    REAL u0[14] = {0};
    FN(orc_consts_init)(M, &K);
    FN(orc_dynamics)(&K, q, u0, 0, &D);
    for (int c = 0; c < NDOF; ++c) {
        REAL phi[6] = {0, 0, 0, 0, 0, 0}, du[16];
        if (c < 6) { phi[c] = 1; FN(orc_response)(&D, phi, 0, 0, 0, 0, du); }
        else { int j = c - 6; FN(orc_response)(&D, phi, 0, j / 2, (j & 1) ? R_(0) : R_(1), (j & 1) ? R_(1) : R_(0), du); }
        for (int r = 0; r < NDOF; ++r) out196[r * NDOF + c] = du[r];
    }
}
void FN(orc_ant_substeps_items)(const hrl_config *cfg, REAL *q, REAL *u, const REAL *tau, int n, const REAL *items_xy, int n_items, int *info3, int *dbg_out /* [1 + MAXC]: candidates, surfaces */, REAL *lambda_out /* [MAXR] */) {
    FN(orc_env) E; FN(orc_substep_dbg) dbg; int gt[13];
    FN(orc_env_init)(cfg, &E);
    memset(&dbg, 0, sizeof(dbg));
    for (int s = 0; s < n; ++s) FN(orc_ant_substep)(&E.K, &E.W, q, u, tau, items_xy, n_items, gt, &dbg);
    if (info3) { info3[0] = dbg.n_rows; info3[1] = dbg.n_limits; info3[2] = dbg.n_contacts; }
    if (dbg_out) { dbg_out[0] = dbg.n_candidates; for (int c = 0; c < MAXC; ++c) dbg_out[1 + c] = c < dbg.n_contacts ? dbg.surface[c] : -1; }
    if (lambda_out) for (int r = 0; r < MAXR; ++r) lambda_out[r] = r < dbg.n_rows ? dbg.lambda[r] : 0;
}

void FN(orc_ant_substeps)(const hrl_config *cfg, REAL *q, REAL *u, const REAL *tau, int n, int *info3) {
    FN(orc_ant_substeps_items)(cfg, q, u, tau, n, 0, 0, info3, 0, 0);
}
/* n point-bot substeps on q[7] (x,y,z,quat xyzw) and u[6] (omega, v) under a constant world-frame force (tests) */
void FN(orc_point_substeps)(const hrl_config *cfg, REAL *q, REAL *u, const REAL *force, int n) {
    FN(orc_env) E;
    FN(orc_env_init)(cfg, &E);
    for (int s = 0; s < n; ++s) FN(orc_point_substep)(&E.K, &E.W, q, u, force, 0, 0, 0);
}
/* the same among static item cubes; info3 = rows, contacts with item cubes, contacts of the last substep (tests) */
void FN(orc_point_substeps_items)(const hrl_config *cfg, REAL *q, REAL *u, const REAL *force, int n, const REAL *items_xy, int n_items, int *info3) {
    FN(orc_env) E;
    FN(orc_substep_dbg) dbg;
    FN(orc_env_init)(cfg, &E);
    memset(&dbg, 0, sizeof(dbg));
    for (int s = 0; s < n; ++s) FN(orc_point_substep)(&E.K, &E.W, q, u, force, items_xy, n_items, &dbg);
    if (info3) {
        info3[0] = dbg.n_rows; info3[1] = 0; info3[2] = dbg.n_contacts;
        for (int c = 0; c < dbg.n_contacts; ++c) info3[1] += ORC_ITEM_OF_SURF(dbg.surface[c]) >= 0;
    }
}

#undef NJ
#undef NBODY
#undef NDOF
#undef MAXC
#undef MAXR
#undef RSQRT
#undef FMA_
#undef RSIN
#undef RCOS
#undef RATAN2
#undef RASIN
#undef RFABS
#undef RFMOD
