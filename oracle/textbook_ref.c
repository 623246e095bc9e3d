/*
 * textbook_ref.c -- FROZEN fp64 TEXTBOOK REFERENCE of the rigid-body substep.  TEST INFRASTRUCTURE, not product code.
 *
 * Purpose.  The reference delegates `scene.global_step()` (ant_gather_env.py:78, gather_base.py:76, MjAnt.py:38) to
 * the absent pybullet wheel, so the build had to publish its own specification of that step.
 * oracle/orc_impl.h states that specification in the OPTIMISED form the HIP kernels implement (articulated-body
 * recursion, L D L^T base factor, row-space Gauss-Seidel) and is co-edited with the kernels to keep the two bit-equal.
 * This file states the SAME model in the form found in a rigid-body dynamics textbook and is never touched by a
 * performance change:
 *
 *   - kinematics from rotation matrices (quaternion -> R, Rodrigues rotations about the MJCF joint axes,
 *     assets/ant.xml:12-58), inertia tensors of the solid shapes assembled part by part (sphere, cylinder,
 *     hemispheres, parallel-axis theorem);
 *   - equations of motion by the projected Newton-Euler (Kane) method with explicit dense body Jacobians:
 *       M(q) = sum_b m_b Jv_b^T Jv_b + Jw_b^T I_b Jw_b          (what CRBA computes)
 *       c(q,u) = sum_b Jv_b^T m_b (a0_b + g z) + Jw_b^T (I_b alpha0_b + w_b x I_b w_b)
 *     with a0/alpha0 the classical accelerations at zero generalized acceleration, then  udot = M^-1 (tau - c)  by a
 *     dense Cholesky factorisation;
 *   - contacts and joint limits as velocity-level rows  J u >= -bias  with J from the same body Jacobians,
 *     B = M^-1 J^T by dense solves;
 *   - the classical VELOCITY-SPACE sequential-impulse iteration (Catto / Bullet form): for every row in order
 *       dl = -(J_r . u + bias_r) / (J_r . B_r);  l' = clamp(l_r + dl);  u += B_r (l' - l_r)
 *     with the friction bounds +-mu * (normal impulse) refreshed row by row;
 *   - semi-implicit Euler with the exact quaternion exponential.
 *
 * tests/test_textbook_reference.py asserts that the optimised fp64 specification (orc_ant_substep_f64 /
 * orc_point_substep_f64) reproduces this file to <= 1e-9 per substep on random contact states, and holds the
 * quantitative contact known-answer tests (force balance, Coulomb cone, sliding deceleration, limit penetration,
 * LCP residual).  Parity with pybullet itself stays UNPINNED (no pybullet, no fixtures: SURVEY.md 8c).
 *
 * Conventions shared with the specification (they define WHAT is simulated, not how):
 *   q[15] = x,y,z, qx,qy,qz,qw, hip_1,ankle_1,...,hip_4,ankle_4      (MjAnt.py:19-20 order)
 *   u[14] = omega (world), v (world, torso COM), joint rates
 *   13 contact shapes: the torso sphere r .25 and the 12 leg capsules r .08 of assets/ant.xml:16-55 -- against planes through their end
 *   spheres (hip point, ankle point, foot tip: the deepest point of a capsule against a plane is an end), against the convex boxes (maze box,
 *   item cubes) through the point of their axis closest to the box; candidate contacts in the order ground, lateral planes, boxes, self
 *   pairs; at most 12 kept; rows = limits, normals, friction pairs.
 *
 * Frozen = never touched by a performance change.  It changes with the MODEL only; so far in round 5: the end-point spheres against boxes
 * became the capsules of the asset (a cube fits between the ankle and tip spheres of a foot capsule), and the parameters hrl_model gained
 * (Bullet's per-body damping, restitution, the contact cap, joint damping and armature; all off / unchanged at their defaults) were added.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define TB_NJ 8
#define TB_NB 9
#define TB_NV 14
#define TB_MAXC 12
#define TB_MAXR 44
#define TB_MAXBOX 20

typedef struct tb_params {
    double density, gravity, h, erp_c, erp_l, mu, mu_self, cdist, lmargin, vmax, limp_max, ground_z;
    int32_t iters, self_collision;
    int32_t n_planes, n_boxes;
    double plane_n[4][3], plane_d[4]; /* inside: n.p - d > 0 */
    double box_lo[TB_MAXBOX][3], box_hi[TB_MAXBOX][3];
    /* model parameters of hrl_model added in ABI v7 (defaults: 0, 0, 0, 0.2, 12) */
    double linear_damping, angular_damping; /* k_l, k_a: every body feels the force -m v k_l (1 + |v|) at its COM and the torque -(I omega) k_a (1 + |omega|) */
    double restitution, restitution_threshold; /* normal rows of approaches faster than the threshold ask for restitution * speed of separation */
    int32_t max_contacts; /* contacts kept per substep, <= TB_MAXC */
    double joint_damping, joint_armature; /* assets/ant.xml:8 `damping` / `armature` where the model is told to have them (defaults 0, 0) */
} tb_params;

typedef struct tb_out {
    int32_t n_limits, n_contacts, n_rows, n_candidates; /* n_candidates > n_contacts: contacts dropped by the cap */
    int32_t row_kind[TB_MAXR];                          /* 0 limit, 1 normal, 2 friction */
    int32_t row_normal[TB_MAXR];                        /* friction rows: index of their normal row */
    int32_t contact_surface[TB_MAXC];                   /* 0 ground, 1.. planes, 100+k box k, 200+pair self */
    double contact_dist[TB_MAXC];
    double lambda[TB_MAXR];
    double w_final[TB_MAXR]; /* J u + bias of every row at the end (LCP residual checks) */
    double total_mass;
} tb_out;

/* ------------------------------------------------------------------------------------------------ 3-vectors, 3x3 */
static void v3_set(double *o, double x, double y, double z) { o[0] = x; o[1] = y; o[2] = z; }
static void v3_cross(double *o, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double v3_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3_mulv(double *o, double M[3][3], const double *v) {
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = M[i][0] * v[0] + M[i][1] * v[1] + M[i][2] * v[2];
    for (int i = 0; i < 3; ++i) o[i] = t[i];
}
static void m3_mul(double O[3][3], double A[3][3], double B[3][3]) {
    double T[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
    memcpy(O, T, sizeof(T));
}
static void m3_from_quat(double R[3][3], const double *q) { /* (x, y, z, w), unit */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z);     R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);     R[2][1] = 2 * (y * z + w * x);     R[2][2] = 1 - 2 * (x * x + y * y);
}
static void m3_rodrigues(double R[3][3], const double *a, double th) { /* rotation by th about the unit axis a */
    double c = cos(th), s = sin(th);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = (i == j ? c : 0.0) + (1 - c) * a[i] * a[j];
    R[0][1] -= s * a[2]; R[0][2] += s * a[1];
    R[1][0] += s * a[2]; R[1][2] -= s * a[0];
    R[2][0] -= s * a[1]; R[2][1] += s * a[0];
}
/* world inertia tensor R I_local R^T */
static void m3_conj(double O[3][3], double R[3][3], double I[3][3]) {
    double T[3][3], Rt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[i][j] = R[j][i];
    m3_mul(T, R, I);
    m3_mul(O, T, Rt);
}

/* ------------------------------------------------------------------------------------------------ solid shapes */
static const double TB_PI = 3.14159265358979323846;
/* solid capsule (cylinder of length L + two hemispherical caps, radius r, axis = local x): mass and principal inertias
 * about its centre.  Hemisphere: mass m_h, COM 3r/8 from its flat face, central transverse inertia (83/320) m_h r^2. */
static void capsule_mass_props(double rho, double r, double L, double *m, double *I_axis, double *I_trans) {
    double mc = rho * TB_PI * r * r * L, mh = rho * (2.0 / 3.0) * TB_PI * r * r * r;
    double off = 0.5 * L + 3.0 * r / 8.0;
    *m = mc + 2 * mh;
    *I_axis = 0.5 * mc * r * r + 2 * (0.4 * mh * r * r);
    *I_trans = mc * (L * L / 12.0 + r * r / 4.0) + 2 * (mh * (83.0 / 320.0) * r * r + mh * off * off);
}

/* MJCF Ant (assets/ant.xml:12-58): leg directions, ankle axes */
static const double LEG_DIR[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};     /* :15-16, :26-27, :37-38, :48-49 */
static const double ANKLE_AXIS[4][2] = {{-1, 1}, {1, 1}, {-1, 1}, {1, 1}};    /* :21, :32, :43, :54 */
static const double JOINT_LO_DEG[TB_NJ] = {-40, 30, -40, -100, -40, -100, -40, 30};
static const double JOINT_HI_DEG[TB_NJ] = {40, 100, 40, -30, 40, -30, 40, 100};
static const double R_TORSO = 0.25, R_CAPS = 0.08;

typedef struct tb_body {
    double m, I[3][3]; /* world-axes inertia about the COM */
    double c[3];       /* COM relative to the torso COM O, world axes */
    double w[3], v[3]; /* angular velocity, COM velocity */
    double alpha0[3], a0[3];
    double Jw[3][TB_NV], Jv[3][TB_NV];
} tb_body;

typedef struct tb_kin {
    tb_body B[TB_NB];
    double axis[TB_NJ][3], anchor[TB_NJ][3]; /* joint axes / anchors (relative to O), world */
    double ph[4][3], pa[4][3], tip[4][3];    /* hip point, ankle point, foot tip, relative to O */
} tb_kin;

static void tb_kinematics(const tb_params *P, const double *q, const double *u, tb_kin *K) {
    double R0[3][3];
    m3_from_quat(R0, q + 3);
    memset(K, 0, sizeof(*K));
    double m1, Ia1, It1, m2, Ia2, It2;
    const double L1 = sqrt(0.2 * 0.2 + 0.2 * 0.2), L2 = sqrt(0.4 * 0.4 + 0.4 * 0.4);
    capsule_mass_props(P->density, R_CAPS, L1, &m1, &Ia1, &It1);
    capsule_mass_props(P->density, R_CAPS, L2, &m2, &Ia2, &It2);
    /* torso body = sphere + the four jointless leg capsules (fixed to it), assembled in the torso frame */
    double msph = P->density * (4.0 / 3.0) * TB_PI * R_TORSO * R_TORSO * R_TORSO;
    double It[3][3] = {{0}};
    for (int i = 0; i < 3; ++i) It[i][i] = 0.4 * msph * R_TORSO * R_TORSO;
    double mt = msph;
    for (int l = 0; l < 4; ++l) {
        double d[3] = {LEG_DIR[l][0] / sqrt(2.0), LEG_DIR[l][1] / sqrt(2.0), 0}, c[3] = {0.1 * LEG_DIR[l][0], 0.1 * LEG_DIR[l][1], 0};
        double cc = v3_dot(c, c);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                It[i][j] += (i == j ? It1 : 0.0) + (Ia1 - It1) * d[i] * d[j] + m1 * ((i == j ? cc : 0.0) - c[i] * c[j]);
        mt += m1;
    }
    tb_body *T = &K->B[0];
    T->m = mt;
    m3_conj(T->I, R0, It);
    for (int k = 0; k < 3; ++k) { T->w[k] = u[k]; T->v[k] = u[3 + k]; }
    for (int k = 0; k < 3; ++k) { T->Jw[k][k] = 1; T->Jv[k][3 + k] = 1; }
    for (int l = 0; l < 4; ++l) {
        const int jh = 2 * l, ja = jh + 1;
        tb_body *X = &K->B[1 + 2 * l], *F = &K->B[2 + 2 * l];
        double zloc[3] = {0, 0, 1}, Rz[3][3], Raux[3][3], Ra[3][3], Rfoot[3][3];
        m3_rodrigues(Rz, zloc, q[7 + jh]);
        m3_mul(Raux, R0, Rz);
        double aloc[3] = {ANKLE_AXIS[l][0] / sqrt(2.0), ANKLE_AXIS[l][1] / sqrt(2.0), 0};
        m3_rodrigues(Ra, aloc, q[7 + ja]);
        m3_mul(Rfoot, Raux, Ra);
        double hip_loc[3] = {0.2 * LEG_DIR[l][0], 0.2 * LEG_DIR[l][1], 0}, seg1[3] = {0.2 * LEG_DIR[l][0], 0.2 * LEG_DIR[l][1], 0};
        double seg2[3] = {0.4 * LEG_DIR[l][0], 0.4 * LEG_DIR[l][1], 0}, s1w[3], s2w[3];
        m3_mulv(K->ph[l], R0, hip_loc);
        m3_mulv(s1w, Raux, seg1);
        m3_mulv(s2w, Rfoot, seg2);
        for (int k = 0; k < 3; ++k) { K->pa[l][k] = K->ph[l][k] + s1w[k]; K->tip[l][k] = K->pa[l][k] + s2w[k]; }
        m3_mulv(K->axis[jh], R0, zloc);
        m3_mulv(K->axis[ja], Raux, aloc);
        for (int k = 0; k < 3; ++k) { K->anchor[jh][k] = K->ph[l][k]; K->anchor[ja][k] = K->pa[l][k]; }
        /* mass properties in world axes */
        double e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = s1w[k] / L1; e2[k] = s2w[k] / L2; X->c[k] = K->ph[l][k] + 0.5 * s1w[k]; F->c[k] = K->pa[l][k] + 0.5 * s2w[k]; }
        X->m = m1; F->m = m2;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                X->I[i][j] = (i == j ? It1 : 0.0) + (Ia1 - It1) * e1[i] * e1[j];
                F->I[i][j] = (i == j ? It2 : 0.0) + (Ia2 - It2) * e2[i] * e2[j];
            }
        /* velocities down the chain */
        double t[3], dX[3], dA[3], dF[3], v_hip[3], v_ank[3];
        for (int k = 0; k < 3; ++k) { dX[k] = X->c[k] - K->ph[l][k]; dA[k] = K->pa[l][k] - K->ph[l][k]; dF[k] = F->c[k] - K->pa[l][k]; }
        v3_cross(t, T->w, K->ph[l]);
        for (int k = 0; k < 3; ++k) { v_hip[k] = T->v[k] + t[k]; X->w[k] = T->w[k] + K->axis[jh][k] * u[6 + jh]; }
        v3_cross(t, X->w, dX);
        for (int k = 0; k < 3; ++k) X->v[k] = v_hip[k] + t[k];
        v3_cross(t, X->w, dA);
        for (int k = 0; k < 3; ++k) { v_ank[k] = v_hip[k] + t[k]; F->w[k] = X->w[k] + K->axis[ja][k] * u[6 + ja]; }
        v3_cross(t, F->w, dF);
        for (int k = 0; k < 3; ++k) F->v[k] = v_ank[k] + t[k];
        /* classical accelerations at zero generalized acceleration (the "J-dot u" terms) */
        double a_hip[3], a_ank[3], t2[3];
        v3_cross(t, T->w, K->ph[l]); v3_cross(a_hip, T->w, t);            /* w x (w x r) of a torso-fixed point */
        v3_cross(t, T->w, K->axis[jh]);
        for (int k = 0; k < 3; ++k) X->alpha0[k] = t[k] * u[6 + jh];      /* d/dt (axis) = w_parent x axis */
        v3_cross(t, X->alpha0, dX); v3_cross(t2, X->w, dX); v3_cross(t2, X->w, t2);
        for (int k = 0; k < 3; ++k) X->a0[k] = a_hip[k] + t[k] + t2[k];
        v3_cross(t, X->alpha0, dA); v3_cross(t2, X->w, dA); v3_cross(t2, X->w, t2);
        for (int k = 0; k < 3; ++k) a_ank[k] = a_hip[k] + t[k] + t2[k];
        v3_cross(t, X->w, K->axis[ja]);
        for (int k = 0; k < 3; ++k) F->alpha0[k] = X->alpha0[k] + t[k] * u[6 + ja];
        v3_cross(t, F->alpha0, dF); v3_cross(t2, F->w, dF); v3_cross(t2, F->w, t2);
        for (int k = 0; k < 3; ++k) F->a0[k] = a_ank[k] + t[k] + t2[k];
        /* body Jacobians: columns 0-2 omega, 3-5 v, 6+j joint rates */
        for (int b = 0; b < 2; ++b) {
            tb_body *Bd = b ? F : X;
            for (int col = 0; col < 3; ++col) {
                double ek[3] = {col == 0, col == 1, col == 2};
                v3_cross(t, ek, Bd->c); /* d(v_com)/d(omega_col) = e x r */
                for (int k = 0; k < 3; ++k) { Bd->Jw[k][col] = ek[k]; Bd->Jv[k][col] = t[k]; Bd->Jv[k][3 + col] = ek[k]; }
            }
            double rel[3];
            for (int k = 0; k < 3; ++k) rel[k] = Bd->c[k] - K->ph[l][k];
            v3_cross(t, K->axis[jh], rel);
            for (int k = 0; k < 3; ++k) { Bd->Jw[k][6 + jh] = K->axis[jh][k]; Bd->Jv[k][6 + jh] = t[k]; }
            if (b) {
                for (int k = 0; k < 3; ++k) rel[k] = Bd->c[k] - K->pa[l][k];
                v3_cross(t, K->axis[ja], rel);
                for (int k = 0; k < 3; ++k) { Bd->Jw[k][6 + ja] = K->axis[ja][k]; Bd->Jv[k][6 + ja] = t[k]; }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ dense algebra */
static int chol_factor(int n, double *A /* n x n, lower triangle overwritten by L */) {
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0)) return -1;
        A[j * n + j] = sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / A[j * n + j];
        }
    }
    return 0;
}
static void chol_solve(int n, const double *L, double *x /* in: b, out: A^-1 b */) {
    for (int i = 0; i < n; ++i) {
        double t = x[i];
        for (int k = 0; k < i; ++k) t -= L[i * n + k] * x[k];
        x[i] = t / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = x[i];
        for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * x[k];
        x[i] = t / L[i * n + i];
    }
}

static double v3_norm(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* mass matrix and bias force of the ant at (q, u) */
static void tb_mass_bias(const tb_params *P, const tb_kin *K, double *M /* 14x14 */, double *bias /* 14 */) {
    memset(M, 0, sizeof(double) * TB_NV * TB_NV);
    memset(bias, 0, sizeof(double) * TB_NV);
    for (int b = 0; b < TB_NB; ++b) {
        const tb_body *B = &K->B[b];
        double IJw[3][TB_NV];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < TB_NV; ++c) IJw[i][c] = B->I[i][0] * B->Jw[0][c] + B->I[i][1] * B->Jw[1][c] + B->I[i][2] * B->Jw[2][c];
        for (int r = 0; r < TB_NV; ++r)
            for (int c = 0; c < TB_NV; ++c) {
                double s = 0;
                for (int i = 0; i < 3; ++i) s += B->m * B->Jv[i][r] * B->Jv[i][c] + B->Jw[i][r] * IJw[i][c];
                M[r * TB_NV + c] += s;
            }
        double f[3], n[3], Iw[3], Ia[3], g[3];
        for (int i = 0; i < 3; ++i) {
            Iw[i] = B->I[i][0] * B->w[0] + B->I[i][1] * B->w[1] + B->I[i][2] * B->w[2];
            Ia[i] = B->I[i][0] * B->alpha0[0] + B->I[i][1] * B->alpha0[1] + B->I[i][2] * B->alpha0[2];
        }
        v3_cross(g, B->w, Iw);
        for (int i = 0; i < 3; ++i) { f[i] = B->m * (B->a0[i] + (i == 2 ? P->gravity : 0.0)); n[i] = Ia[i] + g[i]; }
        { /* damping of the body's motion, the form Bullet's multibodies use: an external wrench, so a bias force */
            const double kl = P->linear_damping * (1 + v3_norm(B->v)), ka = P->angular_damping * (1 + v3_norm(B->w));
            for (int i = 0; i < 3; ++i) { f[i] += B->m * B->v[i] * kl; n[i] += Iw[i] * ka; }
        }
        for (int r = 0; r < TB_NV; ++r)
            for (int i = 0; i < 3; ++i) bias[r] += B->Jv[i][r] * f[i] + B->Jw[i][r] * n[i];
    }
}

/* row of the Jacobian of  d . (velocity of the point p (relative to O) moving with body b) */
static void tb_point_row(const tb_kin *K, int b, const double *p, const double *d, double *row /* 14 */) {
    const tb_body *B = &K->B[b];
    double rel[3] = {p[0] - B->c[0], p[1] - B->c[1], p[2] - B->c[2]};
    for (int c = 0; c < TB_NV; ++c) {
        double jw[3] = {B->Jw[0][c], B->Jw[1][c], B->Jw[2][c]}, t[3];
        v3_cross(t, jw, rel);
        row[c] = d[0] * (B->Jv[0][c] + t[0]) + d[1] * (B->Jv[1][c] + t[1]) + d[2] * (B->Jv[2][c] + t[2]);
    }
}

/* tangent pair of a unit normal (the construction of Bullet's btPlaneSpace1, which the specification adopts) */
static void tb_tangents(const double *n, double *t1, double *t2) {
    if (fabs(n[2]) > 0.70710678118654752440) {
        double a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a);
        v3_set(t1, 0, -n[2] * k, n[1] * k);
        v3_set(t2, a * k, -n[0] * t1[2], n[0] * t1[1]);
    } else {
        double a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a);
        v3_set(t1, -n[1] * k, n[0] * k, 0);
        v3_set(t2, -n[2] * t1[1], n[2] * t1[0], a * k);
    }
}

/* signed distance of a sphere (centre p, world; radius rad) to an axis-aligned box; normal from the box to the sphere */
static double tb_sphere_box(const double *p, double rad, const double *lo, const double *hi, double *n) {
    double d[3], d2 = 0;
    for (int k = 0; k < 3; ++k) { double cp = p[k] < lo[k] ? lo[k] : (p[k] > hi[k] ? hi[k] : p[k]); d[k] = p[k] - cp; d2 += d[k] * d[k]; }
    if (d2 > 0) { double len = sqrt(d2); for (int k = 0; k < 3; ++k) n[k] = d[k] / len; return len - rad; }
    int best = 0; double bd = 1e30, sgn = 1; /* centre inside: leave through the nearest face (first minimum in the order x-,x+,y-,...) */
    for (int k = 0; k < 3; ++k) {
        double dl = p[k] - lo[k], dh = hi[k] - p[k];
        if (dl < bd) { bd = dl; best = k; sgn = -1; }
        if (dh < bd) { bd = dh; best = k; sgn = 1; }
    }
    v3_set(n, 0, 0, 0); n[best] = sgn;
    return -bd - rad;
}

/* Point of the segment p0 -> p1 closest to an axis-aligned box, by bisection: the squared distance D(t) of P(t) = p0 + t (p1 - p0) to the box
 * is convex in t, so its one-sided slope changes sign once.  The set of minimisers is the interval [t_first, t_last] with t_first = the first t
 * whose slope is >= 0 and t_last = the last t whose slope is <= 0 (one point, unless the segment runs alongside a face or through the box);
 * its middle is returned -- the convention of the model for a capsule lying flat on a box. */
static double tb_box_slope(const double *p0, const double *dd, double t, const double *lo, const double *hi) {
    double g = 0;
    for (int k = 0; k < 3; ++k) {
        double x = p0[k] + t * dd[k], cp = x < lo[k] ? lo[k] : (x > hi[k] ? hi[k] : x);
        g += dd[k] * (x - cp);
    }
    return g;
}
double tb_seg_box_param(const double *p0, const double *p1, const double *lo, const double *hi) {
    double dd[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double first, last;
    if (tb_box_slope(p0, dd, 0.0, lo, hi) >= 0) first = 0.0;
    else if (tb_box_slope(p0, dd, 1.0, lo, hi) < 0) first = 1.0;
    else { double a = 0, b = 1; for (int it = 0; it < 200; ++it) { double m = 0.5 * (a + b); if (tb_box_slope(p0, dd, m, lo, hi) >= 0) b = m; else a = m; } first = b; }
    if (tb_box_slope(p0, dd, 1.0, lo, hi) <= 0) last = 1.0;
    else if (tb_box_slope(p0, dd, 0.0, lo, hi) > 0) last = 0.0;
    else { double a = 0, b = 1; for (int it = 0; it < 200; ++it) { double m = 0.5 * (a + b); if (tb_box_slope(p0, dd, m, lo, hi) <= 0) a = m; else b = m; } last = a; }
    return 0.5 * (first + last);
}

/* Second support point of a capsule that rests (nearly) FLAT on a face of a box -- Bullet keeps a manifold of up to four points per pair there, a single
 * point lets the capsule rock about it.  The first contact is the axis point P(t1) closest to the box, with normal n1.  When n1 is a face normal to within
 * 5.7 degrees (its largest component is 0.995 or more: the closest point of the box lies in a face's interior, P(t1) is inside the box -- or, the usual
 * case of a capsule longer than the face, P(t1) has just passed the face's edge on its way down and the normal leans by the capsule's own tilt), the part
 * of the axis that projects into that face is
 * [ta, tb] = [0, 1] clipped by the two slabs of the other axes; its end FARTHER from t1 is the candidate -- tb, towards the capsule's free end (its
 * start is where the neighbouring capsule of the leg ends), unless ta is farther by more than a thousandth of the axis: where a whole stretch is closest
 * t1 is its exact middle, and a tie must not be decided by rounding --: it becomes a contact of its own if
 * it is at least one capsule radius away from P(t1) along the axis and itself closer to the FACE than the contact distance (judged by the caller: the
 * second contact keeps the first one's normal and measures its distance to that face's plane -- P(t2) sits on the border of the face's region by
 * construction, where the closest feature of the box is a matter of rounding).  Returns 1 and t2, or 0. */
static int tb_second_point(const double *p0, const double *p1, double t1, const double *n1, double rad, const double *lo, const double *hi, double *t2) {
    int kf = 0;
    for (int k = 1; k < 3; ++k) if (fabs(n1[k]) > fabs(n1[kf])) kf = k;
    if (!(fabs(n1[kf]) >= 0.995)) return 0; /* within 5.7 degrees of a face normal */
    double ta = 0, tb = 1, len2 = 0;
    for (int k = 0; k < 3; ++k) {
        double d = p1[k] - p0[k];
        len2 += d * d;
        if (k == kf || d == 0) continue;
        double u = (lo[k] - p0[k]) / d, v = (hi[k] - p0[k]) / d, tl = u < v ? u : v, th = u < v ? v : u;
        if (tl > ta) ta = tl;
        if (th < tb) tb = th;
    }
    *t2 = ((tb - t1) + 1e-3 >= t1 - ta) ? tb : ta;
    return (*t2 - t1) * (*t2 - t1) * len2 >= rad * rad;
}

/* closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9), both of positive length */
static void tb_seg_seg(const double *p1, const double *q1, const double *p2, const double *q2, double *c1, double *c2) {
    double d1[3], d2[3], r[3];
    for (int k = 0; k < 3; ++k) { d1[k] = q1[k] - p1[k]; d2[k] = q2[k] - p2[k]; r[k] = p1[k] - p2[k]; }
    double a = v3_dot(d1, d1), e = v3_dot(d2, d2), f = v3_dot(d2, r), c = v3_dot(d1, r), b = v3_dot(d1, d2);
    double denom = a * e - b * b, s = 0, t;
    if (denom > 1e-9) { s = (b * f - c * e) / denom; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    t = (b * s + f) / e;
    if (t < 0) { t = 0; s = -c / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    else if (t > 1) { t = 1; s = (b - c) / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    for (int k = 0; k < 3; ++k) { c1[k] = p1[k] + d1[k] * s; c2[k] = p2[k] + d2[k] * t; }
}

typedef struct tb_contact { int bodyA, bodyB; /* bodyB < 0: static world */ double p[3], n[3], dist, mu; int surface; } tb_contact;

/* classical velocity-space sequential impulses on `nr` rows over `nv` velocity coordinates */
static void tb_sequential_impulse(int nv, int nr, double J[][TB_NV], double B[][TB_NV], const double *bias, const double *lo0,
                                  const double *hi0, const int *normal_of, const double *mu_of, int iters, double *u, double *lam) {
    double Arr[TB_MAXR];
    for (int r = 0; r < nr; ++r) {
        double s = 0;
        for (int k = 0; k < nv; ++k) s += J[r][k] * B[r][k];
        Arr[r] = s; lam[r] = 0;
    }
    for (int it = 0; it < iters; ++it)
        for (int r = 0; r < nr; ++r) {
            double w = bias[r], lo = lo0[r], hi = hi0[r];
            for (int k = 0; k < nv; ++k) w += J[r][k] * u[k];
            if (normal_of[r] >= 0) { hi = mu_of[r] * lam[normal_of[r]]; lo = -hi; } /* Coulomb pyramid, refreshed row by row */
            double ln = lam[r] - w / Arr[r];
            ln = ln < lo ? lo : (ln > hi ? hi : ln);
            double dl = ln - lam[r];
            lam[r] = ln;
            for (int k = 0; k < nv; ++k) u[k] += B[r][k] * dl;
        }
}

static void tb_integrate_pose(double h, double *q, const double *w, const double *v) {
    for (int k = 0; k < 3; ++k) q[k] += h * v[k];
    double wn = sqrt(v3_dot(w, w)), th = wn * h, dq[4];
    if (th > 1e-12) { double s = sin(0.5 * th) / wn; dq[0] = w[0] * s; dq[1] = w[1] * s; dq[2] = w[2] * s; dq[3] = cos(0.5 * th); }
    else { dq[0] = 0.5 * h * w[0]; dq[1] = 0.5 * h * w[1]; dq[2] = 0.5 * h * w[2]; dq[3] = 1; }
    double x = q[3], y = q[4], z = q[5], s = q[6]; /* q <- dq (x) q, Hamilton product, (x, y, z, w) storage */
    double nx = dq[3] * x + dq[0] * s + dq[1] * z - dq[2] * y;
    double ny = dq[3] * y - dq[0] * z + dq[1] * s + dq[2] * x;
    double nz = dq[3] * z + dq[0] * y - dq[1] * x + dq[2] * s;
    double nw = dq[3] * s - dq[0] * x - dq[1] * y - dq[2] * z;
    double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    q[3] = nx * inv; q[4] = ny * inv; q[5] = nz * inv; q[6] = nw * inv;
}

/* ================================================================================================ ANT SUBSTEP */
void tb_ant_substep(const tb_params *P, double *q, double *u, const double *tau, tb_out *out) {
    static _Thread_local tb_kin K;
    static _Thread_local double M[TB_NV * TB_NV], Lc[TB_NV * TB_NV];
    double bias_f[TB_NV], rhs[TB_NV];
    const double h = P->h;
    tb_kinematics(P, q, u, &K);
    tb_mass_bias(P, &K, M, bias_f);
    for (int j = 0; j < TB_NJ; ++j) M[(6 + j) * TB_NV + (6 + j)] += P->joint_armature; /* rotor inertia on the joint's diagonal */
    memcpy(Lc, M, sizeof(M));
    chol_factor(TB_NV, Lc);
    for (int k = 0; k < TB_NV; ++k) rhs[k] = (k >= 6 ? tau[k - 6] - P->joint_damping * u[k] : 0.0) - bias_f[k];
    chol_solve(TB_NV, Lc, rhs);
    double un[TB_NV];
    for (int k = 0; k < TB_NV; ++k) un[k] = u[k] + h * rhs[k];
    const int cap = P->max_contacts > 0 && P->max_contacts < TB_MAXC ? P->max_contacts : TB_MAXC;

    /* ---- rows */
    static _Thread_local double J[TB_MAXR][TB_NV], B[TB_MAXR][TB_NV];
    double bias[TB_MAXR], lo[TB_MAXR], hi[TB_MAXR], mu_of[TB_MAXR], lam[TB_MAXR];
    int normal_of[TB_MAXR], kind[TB_MAXR], nr = 0, nl = 0;
    const double d2r = TB_PI / 180.0;
    for (int j = 0; j < TB_NJ; ++j) { /* joint limits within lmargin: speculative (dist > 0) or Baumgarte (dist < 0) */
        double dlo = q[7 + j] - JOINT_LO_DEG[j] * d2r, dhi = JOINT_HI_DEG[j] * d2r - q[7 + j], sgn, dist;
        if (dlo < P->lmargin) { sgn = 1; dist = dlo; }
        else if (dhi < P->lmargin) { sgn = -1; dist = dhi; }
        else continue;
        memset(J[nr], 0, sizeof(J[nr]));
        J[nr][6 + j] = sgn;
        bias[nr] = (dist > 0 ? dist : P->erp_l * dist) / h;
        lo[nr] = 0; hi[nr] = P->limp_max; normal_of[nr] = -1; mu_of[nr] = 0; kind[nr] = 0;
        ++nr; ++nl;
    }
    /* ---- contact candidates in order: ground, lateral planes, boxes, self pairs; first TB_MAXC kept */
    tb_contact C[TB_MAXC];
    int nc = 0, ncand = 0;
    double sc[13][3], s0[13][3], srad[13]; int sbody[13]; /* shape s: the torso sphere, or the capsule s0 -> sc of ant.xml:16-55 (three per leg) */
    v3_set(sc[0], 0, 0, 0); v3_set(s0[0], 0, 0, 0); srad[0] = R_TORSO; sbody[0] = 0;
    for (int l = 0; l < 4; ++l)
        for (int w = 0; w < 3; ++w) {
            const double *src = w == 0 ? K.ph[l] : (w == 1 ? K.pa[l] : K.tip[l]);
            const double zero[3] = {0, 0, 0}, *from = w == 0 ? zero : (w == 1 ? K.ph[l] : K.pa[l]);
            int s = 1 + 3 * l + w;
            for (int k = 0; k < 3; ++k) { sc[s][k] = src[k]; s0[s][k] = from[k]; }
            srad[s] = R_CAPS; sbody[s] = w == 0 ? 0 : (w == 1 ? 1 + 2 * l : 2 + 2 * l);
        }
    const int nsurf = 1 + P->n_planes + P->n_boxes;
    static _Thread_local double t_first_all[TB_MAXBOX][13], n_first_all[TB_MAXBOX][13][3]; /* box surfaces: the first contact of every shape, for the second support points below */
    static _Thread_local int ok_first_all[TB_MAXBOX][13];
    for (int f = 0; f < nsurf; ++f) {
        double (*n_first)[3] = f > P->n_planes ? n_first_all[f - 1 - P->n_planes] : 0, *t_first = f > P->n_planes ? t_first_all[f - 1 - P->n_planes] : 0;
        int *ok_first = f > P->n_planes ? ok_first_all[f - 1 - P->n_planes] : 0;
        for (int s = 0; s < 13; ++s) {
            double ctr[3] = {sc[s][0], sc[s][1], sc[s][2]}; /* centre of the sphere that touches, relative to O */
            double p[3] = {q[0] + ctr[0], q[1] + ctr[1], q[2] + ctr[2]}, n[3], dist;
            if (f == 0) { v3_set(n, 0, 0, 1); dist = p[2] - P->ground_z - srad[s]; }
            else if (f <= P->n_planes) { for (int k = 0; k < 3; ++k) n[k] = P->plane_n[f - 1][k]; dist = v3_dot(n, p) - P->plane_d[f - 1] - srad[s]; }
            else { /* a convex box: the whole capsule, through the point of its axis closest to the box (against a plane that point is an end) */
                const double *blo = P->box_lo[f - 1 - P->n_planes], *bhi = P->box_hi[f - 1 - P->n_planes];
                double w0[3] = {q[0] + s0[s][0], q[1] + s0[s][1], q[2] + s0[s][2]};
                double t = tb_seg_box_param(w0, p, blo, bhi);
                for (int k = 0; k < 3; ++k) { ctr[k] = s0[s][k] + t * (sc[s][k] - s0[s][k]); p[k] = q[k] + ctr[k]; }
                dist = tb_sphere_box(p, srad[s], blo, bhi, n);
                t_first[s] = t; ok_first[s] = dist < P->cdist;
                for (int k = 0; k < 3; ++k) n_first[s][k] = n[k];
            }
            if (!(dist < P->cdist)) continue;
            ++ncand;
            if (nc >= cap) continue;
            tb_contact *c = &C[nc++];
            c->bodyA = sbody[s]; c->bodyB = -1; c->dist = dist; c->mu = P->mu;
            c->surface = f <= P->n_planes ? f : 100 + (f - 1 - P->n_planes);
            for (int k = 0; k < 3; ++k) { c->n[k] = n[k]; c->p[k] = ctr[k] - srad[s] * n[k]; }
        }
    }
    /* after the first contacts of every box: the second support points of the capsules that lie flat on a face, in the order of their first contacts
       (box-major, shape-minor) */
    for (int f = 1 + P->n_planes; f < nsurf; ++f) {
        double (*n_first)[3] = n_first_all[f - 1 - P->n_planes], *t_first = t_first_all[f - 1 - P->n_planes];
        const int *ok_first = ok_first_all[f - 1 - P->n_planes];
            for (int s = 1; s < 13; ++s) {
                if (!ok_first[s]) continue;
                const double *blo = P->box_lo[f - 1 - P->n_planes], *bhi = P->box_hi[f - 1 - P->n_planes];
                double w0[3] = {q[0] + s0[s][0], q[1] + s0[s][1], q[2] + s0[s][2]}, w1[3] = {q[0] + sc[s][0], q[1] + sc[s][1], q[2] + sc[s][2]}, t2, ctr[3], p[3], n[3];
                if (!tb_second_point(w0, w1, t_first[s], n_first[s], srad[s], blo, bhi, &t2)) continue;
                for (int k = 0; k < 3; ++k) { ctr[k] = s0[s][k] + t2 * (sc[s][k] - s0[s][k]); p[k] = q[k] + ctr[k]; n[k] = 0; }
                int kf = 0;
                for (int k = 1; k < 3; ++k) if (fabs(n_first[s][k]) > fabs(n_first[s][kf])) kf = k;
                n[kf] = n_first[s][kf] > 0 ? 1.0 : -1.0; /* the face of the first contact */
                double dist = n[kf] * (p[kf] - (n[kf] > 0 ? bhi[kf] : blo[kf])) - srad[s];
                if (!(dist < P->cdist)) continue;
                ++ncand;
                if (nc >= cap) continue;
                tb_contact *c = &C[nc++];
                c->bodyA = sbody[s]; c->bodyB = -1; c->dist = dist; c->mu = P->mu; c->surface = 100 + (f - 1 - P->n_planes);
                for (int k = 0; k < 3; ++k) { c->n[k] = n[k]; c->p[k] = ctr[k] - srad[s] * n[k]; }
            }
    }
    if (P->self_collision) { /* capsules of different legs (links that are not ancestors of each other) */
        int pair = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        if (a == 0 && b == 0) continue; /* both fixed to the torso */
                        double zero[3] = {0, 0, 0}, c1[3], c2[3], dv[3];
                        const double *p1 = a == 0 ? zero : (a == 1 ? K.ph[i] : K.pa[i]), *q1 = a == 0 ? K.ph[i] : (a == 1 ? K.pa[i] : K.tip[i]);
                        const double *p2 = b == 0 ? zero : (b == 1 ? K.ph[j] : K.pa[j]), *q2 = b == 0 ? K.ph[j] : (b == 1 ? K.pa[j] : K.tip[j]);
                        tb_seg_seg(p1, q1, p2, q2, c1, c2);
                        for (int k = 0; k < 3; ++k) dv[k] = c1[k] - c2[k];
                        double len = sqrt(v3_dot(dv, dv)), dist = len - 2 * R_CAPS;
                        int id = pair++;
                        if (!(dist < P->cdist)) continue;
                        ++ncand;
                        if (nc >= cap) continue;
                        tb_contact *c = &C[nc++];
                        c->bodyA = a == 0 ? 0 : (a == 1 ? 1 + 2 * i : 2 + 2 * i);
                        c->bodyB = b == 0 ? 0 : (b == 1 ? 1 + 2 * j : 2 + 2 * j);
                        c->dist = dist; c->mu = P->mu_self; c->surface = 200 + id;
                        if (len > 0) for (int k = 0; k < 3; ++k) c->n[k] = dv[k] / len; else v3_set(c->n, 0, 0, 1);
                        for (int k = 0; k < 3; ++k) c->p[k] = 0.5 * (c1[k] + c2[k]); /* equal radii: the midpoint of the two surface points */
                    }
    }
    for (int row = 0; row < 3 * nc; ++row) { /* all normals, then (t1, t2) per contact */
        int ci = row < nc ? row : (row - nc) / 2, which = row < nc ? 0 : 1 + ((row - nc) & 1);
        double t1[3], t2[3];
        tb_tangents(C[ci].n, t1, t2);
        const double *d = which == 0 ? C[ci].n : (which == 1 ? t1 : t2);
        tb_point_row(&K, C[ci].bodyA, C[ci].p, d, J[nr]);
        if (C[ci].bodyB >= 0) {
            double jb[TB_NV];
            tb_point_row(&K, C[ci].bodyB, C[ci].p, d, jb);
            for (int k = 0; k < TB_NV; ++k) J[nr][k] -= jb[k];
        }
        if (which == 0) {
            bias[nr] = (C[ci].dist > 0 ? C[ci].dist : P->erp_c * C[ci].dist) / h; lo[nr] = 0; hi[nr] = 1e30; normal_of[nr] = -1; mu_of[nr] = 0; kind[nr] = 1;
            if (P->restitution > 0) { /* Newton restitution on the approach speed at the start of the substep */
                double vn = 0;
                for (int k = 0; k < TB_NV; ++k) vn += J[nr][k] * u[k];
                if (vn < -P->restitution_threshold) bias[nr] += P->restitution * vn;
            }
        }
        else { bias[nr] = 0; lo[nr] = 0; hi[nr] = 0; normal_of[nr] = nl + ci; mu_of[nr] = C[ci].mu; kind[nr] = 2; }
        ++nr;
    }
    for (int r = 0; r < nr; ++r) { memcpy(B[r], J[r], sizeof(B[r])); chol_solve(TB_NV, Lc, B[r]); }
    tb_sequential_impulse(TB_NV, nr, J, B, bias, lo, hi, normal_of, mu_of, P->iters, un, lam);
    if (out) {
        memset(out, 0, sizeof(*out));
        out->n_limits = nl; out->n_contacts = nc; out->n_rows = nr; out->n_candidates = ncand;
        for (int r = 0; r < nr; ++r) {
            double w = bias[r];
            for (int k = 0; k < TB_NV; ++k) w += J[r][k] * un[k];
            out->lambda[r] = lam[r]; out->w_final[r] = w; out->row_kind[r] = kind[r]; out->row_normal[r] = normal_of[r];
        }
        for (int c = 0; c < nc; ++c) { out->contact_surface[c] = C[c].surface; out->contact_dist[c] = C[c].dist; }
        for (int b = 0; b < TB_NB; ++b) out->total_mass += K.B[b].m;
    }
    for (int j = 0; j < TB_NJ; ++j) un[6 + j] = un[6 + j] < -P->vmax ? -P->vmax : (un[6 + j] > P->vmax ? P->vmax : un[6 + j]);
    memcpy(u, un, sizeof(un));
    tb_integrate_pose(h, q, u, u + 3);
    for (int j = 0; j < TB_NJ; ++j) q[7 + j] += h * u[6 + j];
}

/* hip point, ankle point and foot tip of every leg and the centres of mass of the nine bodies, world coordinates (tests/test_assets.py) */
void tb_ant_points(const tb_params *P, const double *q, double *legs36, double *coms27) {
    static _Thread_local tb_kin K;
    double u0[TB_NV] = {0};
    tb_kinematics(P, q, u0, &K);
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < 3; ++k) { legs36[9 * l + k] = q[k] + K.ph[l][k]; legs36[9 * l + 3 + k] = q[k] + K.pa[l][k]; legs36[9 * l + 6 + k] = q[k] + K.tip[l][k]; }
    for (int b = 0; b < TB_NB; ++b)
        for (int k = 0; k < 3; ++k) coms27[3 * b + k] = q[k] + K.B[b].c[k];
}

/* mass matrix, bias force, forward-dynamics acceleration and total momentum of the free ant (tests) */
void tb_ant_dynamics(const tb_params *P, const double *q, const double *u, const double *tau, double *M_out, double *bias_out, double *udot_out) {
    static _Thread_local tb_kin K;
    double M[TB_NV * TB_NV], Lc[TB_NV * TB_NV], b[TB_NV], rhs[TB_NV];
    tb_kinematics(P, q, u, &K);
    tb_mass_bias(P, &K, M, b);
    memcpy(Lc, M, sizeof(M));
    chol_factor(TB_NV, Lc);
    for (int k = 0; k < TB_NV; ++k) rhs[k] = (k >= 6 ? tau[k - 6] : 0.0) - b[k];
    chol_solve(TB_NV, Lc, rhs);
    if (M_out) memcpy(M_out, M, sizeof(M));
    if (bias_out) memcpy(bias_out, b, sizeof(b));
    if (udot_out) memcpy(udot_out, rhs, sizeof(rhs));
}

/* ================================================================================================ POINT SUBSTEP
 * point_bot.py:10-74 + assets/player_cube.xml:8: free 10 kg cube (half extent 0.35), contact points = its 8 corners;
 * u[6] = omega, v; force = world-frame force at the COM.  q[7] = x,y,z, qx,qy,qz,qw. */
void tb_point_substep(const tb_params *P, double *q, double *u, const double *force, tb_out *out) {
    const double m = 10.0, he = 0.35, I = m * (2 * he) * (2 * he) / 6.0, h = P->h;
    double R[3][3], un[TB_NV] = {0};
    m3_from_quat(R, q + 3);
    /* an isotropic inertia tensor has no gyroscopic torque: omega is unchanged by the free motion */
    for (int k = 0; k < 3; ++k) { un[k] = u[k]; un[3 + k] = u[3 + k] + h * (force[k] / m - (k == 2 ? P->gravity : 0.0)); }
    { /* damping: the same wrench on the one free body (isotropic inertia: the torque -I omega k_a (1 + |omega|) decelerates omega by omega k_a (1 + |omega|)) */
        const double kl = P->linear_damping * (1 + v3_norm(u + 3)), ka = P->angular_damping * (1 + v3_norm(u));
        for (int k = 0; k < 3; ++k) { un[k] -= h * u[k] * ka; un[3 + k] -= h * u[3 + k] * kl; }
    }
    const int cap = P->max_contacts > 0 && P->max_contacts < TB_MAXC ? P->max_contacts : TB_MAXC;
    static _Thread_local double J[TB_MAXR][TB_NV], B[TB_MAXR][TB_NV];
    double bias[TB_MAXR], lo[TB_MAXR], hi[TB_MAXR], mu_of[TB_MAXR], lam[TB_MAXR], cp[TB_MAXC][3], cn[TB_MAXC][3], cd[TB_MAXC];
    int normal_of[TB_MAXR], kind[TB_MAXR], csurf[TB_MAXC], nc = 0, ncand = 0;
    const int nsurf = 1 + P->n_planes + P->n_boxes;
    for (int f = 0; f < nsurf; ++f)
        for (int s = 0; s < 8; ++s) {
            double loc[3] = {(s & 1) ? he : -he, (s & 2) ? he : -he, (s & 4) ? he : -he}, c[3], p[3], n[3], dist;
            m3_mulv(c, R, loc);
            for (int k = 0; k < 3; ++k) p[k] = q[k] + c[k];
            if (f == 0) { v3_set(n, 0, 0, 1); dist = p[2] - P->ground_z; }
            else if (f <= P->n_planes) { for (int k = 0; k < 3; ++k) n[k] = P->plane_n[f - 1][k]; dist = v3_dot(n, p) - P->plane_d[f - 1]; }
            else dist = tb_sphere_box(p, 0.0, P->box_lo[f - 1 - P->n_planes], P->box_hi[f - 1 - P->n_planes], n);
            if (!(dist < P->cdist)) continue;
            ++ncand;
            if (nc >= cap) continue;
            for (int k = 0; k < 3; ++k) { cp[nc][k] = c[k]; cn[nc][k] = n[k]; }
            cd[nc] = dist; csurf[nc] = f <= P->n_planes ? f : 100 + (f - 1 - P->n_planes); ++nc;
        }
    int nr = 0;
    for (int row = 0; row < 3 * nc; ++row) {
        int ci = row < nc ? row : (row - nc) / 2, which = row < nc ? 0 : 1 + ((row - nc) & 1);
        double t1[3], t2[3], rxd[3];
        tb_tangents(cn[ci], t1, t2);
        const double *d = which == 0 ? cn[ci] : (which == 1 ? t1 : t2);
        v3_cross(rxd, cp[ci], d);
        memset(J[nr], 0, sizeof(J[nr])); memset(B[nr], 0, sizeof(B[nr]));
        for (int k = 0; k < 3; ++k) { J[nr][k] = rxd[k]; J[nr][3 + k] = d[k]; B[nr][k] = rxd[k] / I; B[nr][3 + k] = d[k] / m; }
        if (which == 0) {
            bias[nr] = (cd[ci] > 0 ? cd[ci] : P->erp_c * cd[ci]) / h; lo[nr] = 0; hi[nr] = 1e30; normal_of[nr] = -1; mu_of[nr] = 0; kind[nr] = 1;
            if (P->restitution > 0) {
                double vn = 0;
                for (int k = 0; k < 6; ++k) vn += J[nr][k] * u[k];
                if (vn < -P->restitution_threshold) bias[nr] += P->restitution * vn;
            }
        }
        else { bias[nr] = 0; lo[nr] = 0; hi[nr] = 0; normal_of[nr] = ci; mu_of[nr] = P->mu; kind[nr] = 2; }
        ++nr;
    }
    tb_sequential_impulse(6, nr, J, B, bias, lo, hi, normal_of, mu_of, P->iters, un, lam);
    if (out) {
        memset(out, 0, sizeof(*out));
        out->n_contacts = nc; out->n_rows = nr; out->n_candidates = ncand; out->total_mass = m;
        for (int r = 0; r < nr; ++r) {
            double w = bias[r];
            for (int k = 0; k < 6; ++k) w += J[r][k] * un[k];
            out->lambda[r] = lam[r]; out->w_final[r] = w; out->row_kind[r] = kind[r]; out->row_normal[r] = normal_of[r];
        }
        for (int c = 0; c < nc; ++c) { out->contact_surface[c] = csurf[c]; out->contact_dist[c] = cd[c]; }
    }
    for (int k = 0; k < 6; ++k) u[k] = un[k];
    tb_integrate_pose(h, q, u, u + 3);
}

/* total energy and momentum of the free ant by per-body sums (tests): T, V, P[3], L[3] about the world origin */
void tb_ant_energy_momentum(const tb_params *P, const double *q, const double *u, double *out8) {
    static _Thread_local tb_kin K;
    tb_kinematics(P, q, u, &K);
    double T = 0, V = 0, Pm[3] = {0, 0, 0}, L[3] = {0, 0, 0};
    for (int b = 0; b < TB_NB; ++b) {
        const tb_body *B = &K.B[b];
        double Iw[3], pw[3], mv[3], l[3];
        for (int i = 0; i < 3; ++i) { Iw[i] = B->I[i][0] * B->w[0] + B->I[i][1] * B->w[1] + B->I[i][2] * B->w[2]; pw[i] = q[i] + B->c[i]; mv[i] = B->m * B->v[i]; }
        T += 0.5 * (B->m * v3_dot(B->v, B->v) + v3_dot(B->w, Iw));
        V += B->m * P->gravity * pw[2];
        v3_cross(l, pw, mv);
        for (int i = 0; i < 3; ++i) { Pm[i] += mv[i]; L[i] += l[i] + Iw[i]; }
    }
    out8[0] = T; out8[1] = V;
    for (int i = 0; i < 3; ++i) { out8[2 + i] = Pm[i]; out8[5 + i] = L[i]; }
}
