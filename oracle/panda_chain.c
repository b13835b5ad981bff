/*
 * panda_chain.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * (1) Restatement of the reference's panda_env task costs -- PINNED by golden group G6b:
 *       get_panda_reach_cost   cost_functions.py:91-114
 *       get_panda_pick_cost    cost_functions.py:116-125
 *       get_panda_place_cost   cost_functions.py:127-136
 *       get_pick_tilt_cost     cost_functions.py:138-156
 *       get_motion_cost        cost_functions.py:158-169 (panda branch)
 * (2) Independent implementation of "Panda world spec v3.1" (DESIGN.md section 3): velocity-servoed
 *     9-dof chain with joint-space inertias derived from the collision meshes, forward kinematics from
 *     the URDF constants (assets/urdf/franka_description/robots/franka_panda.urdf:27-242), cubeA /
 *     cubeB as free rigid cubes and the dyn-obs plate as a free (non-rotating) body, CONTACT RESPONSE:
 *     the gripper's collision spheres and a held cube against table / shelf_stand / cubes / plate and
 *     the cubes' corners against table / shelf_stand / each other as velocity-level unilateral rows
 *     with Coulomb friction, solved together with the joint drives by projected Gauss-Seidel passes
 *     (solver settings: isaacgym_wrapper.py:26-31); position-level two-finger grasp (pad channel; v2.1:
 *     grasp region, pads' region, capture volume).
 *     It stands where the reference calls Isaac Gym / PhysX (isaacgym_wrapper.py:354-360);
 *     PARITY UNPINNED against PhysX.  Scene constants: config/panda_env/ yaml files.
 *     This file is written as a generic solver over dynamic row lists; the product's device code
 *     (csrc/panda_dyn.hpp) uses static slots and recomputes the geometry per pass -- same arithmetic.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "m3_oracle.h"

/* chain spec v1.2: mad(a, b, c) = a*b + c with ONE rounding (IEEE 754 fusedMultiplyAdd) where the spec writes it: the
 * sin/cos reduction and polynomials, the kinematic chain's rotations and offsets, the servo, the free cube's
 * integration and the penalty forces -- here and in the HIP kernel alike. */
static inline float mad(float a, float b, float c) { return fmaf(a, b, c); }

/* ---- spec sin/cos: Cody-Waite reduction to [-pi/4, pi/4] + minimax polynomials, plain f32
 * operations only (valid for |x| < 8), so every implementation agrees bit-for-bit ---- */
void m3o_sincos(float x, float* s, float* c) {
    const float k = rintf(x * 0.63661977236758134308f);
    float r = mad(-k, 1.5703125f, x);
    r = mad(-k, 4.837512969970703125e-4f, r);
    r = mad(-k, 7.54978995489188e-8f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = mad(ps, z, 8.3321608736e-3f);
    ps = mad(ps, z, -1.6666654611e-1f);
    const float sn = mad(r, z * ps, r);
    float pc = 2.443315711809948e-5f;
    pc = mad(pc, z, -1.388731625493765e-3f);
    pc = mad(pc, z, 4.166664568298827e-2f);
    const float cs = mad(z * z, pc, mad(-0.5f, z, 1.0f));
    const int q = ((int)k) & 3;
    if (q == 0) { *s = sn; *c = cs; }
    else if (q == 1) { *s = cs; *c = -sn; }
    else if (q == 2) { *s = -sn; *c = -cs; }
    else { *s = -cs; *c = sn; }
}

void m3o_panda_scene_default(m3o_panda_scene* sc) {
    memset(sc, 0, sizeof(*sc));
    sc->dt = 0.01f; sc->substeps = 2; sc->g = 9.8f;
    sc->base[0] = -0.45f; sc->base[1] = 0.0f; sc->base[2] = 1.125f; /* panda.yaml:7 */
    sc->drive_damping = 600.0f;                                      /* isaacgym_wrapper.py:344 */
    /* diagonal of the joint-space mass matrix at the initial pose, from the collision meshes at the default
     * density (tools/panda_inertia.py) */
    const float inertia[9] = {1.32f, 2.12f, 1.30f, 0.918f, 0.0271f, 0.0366f, 0.0030f, 0.022f, 0.022f};
    const float effort[9] = {87, 87, 87, 87, 12, 12, 12, 20, 20};   /* urdf :34..240 */
    const float vlim[9] = {2.175f, 2.175f, 2.175f, 2.175f, 2.61f, 2.61f, 2.61f, 0.2f, 0.2f};
    const float lo[9] = {-2.8973f, -1.7628f, -2.8973f, -3.0718f, -2.8973f, -0.0175f, -2.8973f, 0.0f, 0.0f};
    const float hi[9] = {2.8973f, 1.7628f, 2.8973f, -0.0698f, 2.8973f, 3.7525f, 2.8973f, 0.04f, 0.04f};
    for (int i = 0; i < 9; ++i) {
        sc->inertia[i] = inertia[i]; sc->effort[i] = effort[i]; sc->vlim[i] = vlim[i];
        sc->qlo[i] = lo[i]; sc->qhi[i] = hi[i];
    }
    sc->table[0] = 0.0f; sc->table[1] = 0.0f; sc->table[2] = 1.0f;      /* 1_table.yaml */
    sc->table[3] = 0.6f; sc->table[4] = 0.6f; sc->table[5] = 0.025f;
    sc->shelf[0] = 0.5f; sc->shelf[1] = 0.0f; sc->shelf[2] = 1.175f;    /* 3_shelf_stand.yaml */
    sc->shelf[3] = 0.1f; sc->shelf[4] = 0.1f; sc->shelf[5] = 0.15f;
    sc->cube_half = 0.025f;                                             /* 5_cubeA.yaml */
    sc->cube_m = 0.125f;           /* 0.05^3 at the default density 1000 */
    sc->cube_mu = 1.0f;
    sc->grasp_z = 0.1034f; sc->grasp_dx = 0.025f; sc->grasp_dz = 0.025f;
    sc->grasp_align = 0.95f; sc->grasp_tol = 0.002f;
    sc->k_contact = 5000.0f;
    sc->tip_z = 0.045f; sc->tip_r = 0.012f; sc->hand_z = 0.03f; sc->hand_r = 0.04f;
    sc->iters = 6;                                                      /* isaacgym_wrapper.py:28 */
    sc->contact_offset = 0.01f;                                         /* isaacgym_wrapper.py:30 */
    sc->slop = 0.001f; sc->baumgarte = 0.2f; sc->max_bias = 2.0f; sc->act_margin = 0.002f;
    sc->mu = 1.0f;
    sc->obs_half[0] = 0.1f; sc->obs_half[1] = 0.1f; sc->obs_half[2] = 0.01f; sc->obs_m = 0.8f;  /* 4_obs.yaml */
    sc->sleep_v = 0.02f; sc->sleep_w = 0.4f; sc->rest_gap = 0.002f;
}

void m3o_panda_world_init(m3o_panda_world* w, int cube_on_shelf) {
    memset(w, 0, sizeof(*w));
    const float q0[9] = {0, 0, 0, -2.0f, 0, 1.8675f, 0, 0.02f, 0.02f}; /* panda.yaml:10 */
    for (int i = 0; i < 9; ++i) w->q[i] = q0[i];
    if (cube_on_shelf) { w->cubeA[0] = 0.425f; w->cubeA[1] = 0.0f; w->cubeA[2] = 1.35f; }
    else { w->cubeA[0] = 0.2f; w->cubeA[1] = -0.2f; w->cubeA[2] = 1.06f; }
    w->cubeA[6] = 1.0f;
    w->cubeB[0] = 0.2f; w->cubeB[1] = 0.2f; w->cubeB[2] = 1.06f; w->cubeB[6] = 1.0f;
    w->obs[0] = 0.35f; w->obs[1] = 0.0f; w->obs[2] = 1.735f; w->obs[6] = 1.0f;   /* 4_obs.yaml */
    w->rel_q[3] = 1.0f;
    w->awake[0] = 1.0f; w->awake[1] = 1.0f;     /* the cubes start 1 cm above the table (5_cubeA.yaml, 6_cubeB.yaml) */
}

typedef struct { float x[3], y[3], z[3], p[3]; } frame_t;

static void rot_xp(frame_t* f) { /* R <- R * Rx(+90deg) */
    for (int i = 0; i < 3; ++i) { float y = f->y[i]; f->y[i] = f->z[i]; f->z[i] = -y; }
}
static void rot_xm(frame_t* f) { /* R <- R * Rx(-90deg) */
    for (int i = 0; i < 3; ++i) { float y = f->y[i]; f->y[i] = -f->z[i]; f->z[i] = y; }
}
static void rot_z(frame_t* f, float s, float c) { /* R <- R * Rz */
    for (int i = 0; i < 3; ++i) {
        float x = f->x[i], y = f->y[i];
        f->x[i] = mad(c, x, s * y);
        f->y[i] = mad(c, y, -(s * x));
    }
}
static void trans(frame_t* f, float tx, float ty, float tz) {
    for (int i = 0; i < 3; ++i) f->p[i] = mad(tz, f->z[i], mad(ty, f->y[i], mad(tx, f->x[i], f->p[i])));
}

/* rotation matrix (columns x,y,z) -> quaternion xyzw (Shepperd) */
static void mat2quat(const frame_t* f, float q[4]) {
    const float r00 = f->x[0], r10 = f->x[1], r20 = f->x[2];
    const float r01 = f->y[0], r11 = f->y[1], r21 = f->y[2];
    const float r02 = f->z[0], r12 = f->z[1], r22 = f->z[2];
    const float tr = (r00 + r11) + r22;
    if (tr > 0.0f) {
        float s = sqrtf(tr + 1.0f) * 2.0f;
        q[3] = 0.25f * s; q[0] = (r21 - r12) / s; q[1] = (r02 - r20) / s; q[2] = (r10 - r01) / s;
    } else if (r00 > r11 && r00 > r22) {
        float s = sqrtf(((1.0f + r00) - r11) - r22) * 2.0f;
        q[3] = (r21 - r12) / s; q[0] = 0.25f * s; q[1] = (r01 + r10) / s; q[2] = (r02 + r20) / s;
    } else if (r11 > r22) {
        float s = sqrtf(((1.0f + r11) - r00) - r22) * 2.0f;
        q[3] = (r02 - r20) / s; q[0] = (r01 + r10) / s; q[1] = 0.25f * s; q[2] = (r12 + r21) / s;
    } else {
        float s = sqrtf(((1.0f + r22) - r00) - r11) * 2.0f;
        q[3] = (r10 - r01) / s; q[0] = (r02 + r20) / s; q[1] = (r12 + r21) / s; q[2] = 0.25f * s;
    }
}

/* quaternion xyzw -> rotation matrix entries, the reference's formula (skill_utils.py:140-180) */
static void quat2mat(const float Q[4], float R[9]) {
    float q0 = Q[3], q1 = Q[0], q2 = Q[1], q3 = Q[2];
    R[0] = 2 * (q0 * q0 + q1 * q1) - 1; R[1] = 2 * (q1 * q2 - q0 * q3); R[2] = 2 * (q1 * q3 + q0 * q2);
    R[3] = 2 * (q1 * q2 + q0 * q3); R[4] = 2 * (q0 * q0 + q2 * q2) - 1; R[5] = 2 * (q2 * q3 - q0 * q1);
    R[6] = 2 * (q1 * q3 - q0 * q2); R[7] = 2 * (q2 * q3 + q0 * q1); R[8] = 2 * (q0 * q0 + q3 * q3) - 1;
}

/* forward kinematics: franka_panda.urdf joint origins (:29,50,71,92,116,137,160), hand
 * (:177-187), fingers (:226-242).  links[11] optional: link0..7, hand, left, right. */
void m3o_panda_fk(const m3o_panda_scene* sc, const float q[9], m3o_panda_links* L) {
    frame_t f;
    f.x[0] = 1; f.x[1] = 0; f.x[2] = 0; f.y[0] = 0; f.y[1] = 1; f.y[2] = 0;
    f.z[0] = 0; f.z[1] = 0; f.z[2] = 1;
    f.p[0] = sc->base[0]; f.p[1] = sc->base[1]; f.p[2] = sc->base[2];
    int li = 0;
#define STORE_LINK()                                                                  \
    do {                                                                              \
        for (int i_ = 0; i_ < 3; ++i_) {                                              \
            L->pos[li][i_] = f.p[i_]; L->ax[li][i_] = f.x[i_];                        \
            L->ay[li][i_] = f.y[i_]; L->az[li][i_] = f.z[i_];                         \
        }                                                                             \
        mat2quat(&f, L->quat[li]);                                                    \
        ++li;                                                                         \
    } while (0)
    STORE_LINK(); /* link0 */
    float s, c;
    trans(&f, 0, 0, 0.333f); m3o_sincos(q[0], &s, &c); rot_z(&f, s, c); STORE_LINK();
    rot_xm(&f); m3o_sincos(q[1], &s, &c); rot_z(&f, s, c); STORE_LINK();
    trans(&f, 0, -0.316f, 0); rot_xp(&f); m3o_sincos(q[2], &s, &c); rot_z(&f, s, c); STORE_LINK();
    trans(&f, 0.0825f, 0, 0); rot_xp(&f); m3o_sincos(q[3], &s, &c); rot_z(&f, s, c); STORE_LINK();
    trans(&f, -0.0825f, 0.384f, 0); rot_xm(&f); m3o_sincos(q[4], &s, &c); rot_z(&f, s, c); STORE_LINK();
    rot_xp(&f); m3o_sincos(q[5], &s, &c); rot_z(&f, s, c); STORE_LINK();
    trans(&f, 0.088f, 0, 0); rot_xp(&f); m3o_sincos(q[6], &s, &c); rot_z(&f, s, c); STORE_LINK();
    trans(&f, 0, 0, 0.107f); rot_z(&f, -0.70710678118654752f, 0.70710678118654752f); STORE_LINK(); /* hand */
    trans(&f, 0, 0, 0.0584f);
    frame_t l = f, r = f;
    for (int i = 0; i < 3; ++i) { l.p[i] = mad(q[7], f.y[i], f.p[i]); r.p[i] = mad(-q[8], f.y[i], f.p[i]); }
    f = l; STORE_LINK();
    f = r; STORE_LINK();
#undef STORE_LINK
}

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* =====================================================================================================
 * Panda world spec v2: contact response.
 * ===================================================================================================== */
/* spec: reciprocal square root = bit-trick seed + three Newton steps in binary32, in this order (as the planar spec) */
/* world spec v3.1: a contact row's effective mass is this reciprocal of its (positive) denominator, not an IEEE division -- the
 * planar spec's sequence (planar_world.c, spec v1.6): minimax bit-trick seed, three Newton steps in residual form */
static float spec_rcp(float x) {
    unsigned int i;
    float y, r;
    memcpy(&i, &x, 4);
    i = 0x7EF311C7u - i;
    memcpy(&y, &i, 4);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    return y;
}

static float spec_rsqrt(float a) {
    union { float f; unsigned u; } c;
    c.f = a;
    c.u = 0x5f3759dfu - (c.u >> 1);
    float y = c.f;
    const float hlf = 0.5f * a;
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    return y;
}
static void cross3(const float a[3], const float b[3], float c[3]) {
    c[0] = mad(a[1], b[2], -(a[2] * b[1]));
    c[1] = mad(a[2], b[0], -(a[0] * b[2]));
    c[2] = mad(a[0], b[1], -(a[1] * b[0]));
}
static float dotm(const float a[3], const float b[3]) { return mad(a[0], b[0], mad(a[1], b[1], a[2] * b[2])); }

/* a free body's rotation matrix (row-major) from its quaternion xyzw */
static void body_rot(const float q[4], float R[9]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = mad(-2.0f, yy + zz, 1.0f); R[1] = 2.0f * (xy - wz);          R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);          R[4] = mad(-2.0f, xx + zz, 1.0f); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);          R[7] = 2.0f * (yz + wx);          R[8] = mad(-2.0f, xx + yy, 1.0f);
}

enum { T_TABLE = 0, T_SHELF = 1, T_CUBEA = 2, T_CUBEB = 3, T_OBS = 4 };
typedef struct { float p[3], e[3], R[9]; int oriented; } box_t;

static void box_static(const float b[6], box_t* o) {
    for (int i = 0; i < 3; ++i) { o->p[i] = b[i]; o->e[i] = b[3 + i]; }
    o->oriented = 0;
}
static void box_body(const float body[13], const float e[3], int oriented, box_t* o) {
    for (int i = 0; i < 3; ++i) { o->p[i] = body[i]; o->e[i] = e[i]; }
    o->oriented = oriented;
    if (oriented) body_rot(body + 3, o->R);
}
static void box_local(const box_t* b, const float c[3], float l[3]) {
    const float dl[3] = {c[0] - b->p[0], c[1] - b->p[1], c[2] - b->p[2]};
    for (int i = 0; i < 3; ++i)
        l[i] = b->oriented ? mad(dl[0], b->R[0 * 3 + i], mad(dl[1], b->R[1 * 3 + i], dl[2] * b->R[2 * 3 + i])) : dl[i];
}
static void box_world(const box_t* b, const float v[3], float o[3]) {
    for (int j = 0; j < 3; ++j)
        o[j] = b->oriented ? mad(b->R[j * 3 + 0], v[0], mad(b->R[j * 3 + 1], v[1], b->R[j * 3 + 2] * v[2])) : v[j];
}

/* sphere (centre c, radius r; a cube's corner: r = 0) against a box: the gap, the unit normal from the box to the
 * sphere (world) and the contact point on the sphere's surface.  Centre outside: along the line to the closest point
 * of the box; centre inside (or on the surface): the face of least penetration, lowest axis first. */
static float pt_box(const box_t* b, const float c[3], float r, float n[3], float x[3]) {
    float l[3], d[3], nl[3], gap;
    box_local(b, c, l);
    for (int i = 0; i < 3; ++i) d[i] = l[i] - fminf(fmaxf(l[i], -b->e[i]), b->e[i]);
    const float d2 = mad(d[0], d[0], mad(d[1], d[1], d[2] * d[2]));
    if (d2 > 1.0e-12f) {
        const float rs = spec_rsqrt(d2);
        for (int i = 0; i < 3; ++i) nl[i] = d[i] * rs;
        gap = d2 * rs - r;
    } else {
        int f = 0;
        float pen = b->e[0] - fabsf(l[0]);
        for (int i = 1; i < 3; ++i) {
            const float pi = b->e[i] - fabsf(l[i]);
            if (pi < pen) { pen = pi; f = i; }
        }
        nl[0] = nl[1] = nl[2] = 0.0f;
        nl[f] = (l[f] >= 0.0f) ? 1.0f : -1.0f;
        gap = -pen - r;
    }
    box_world(b, nl, n);
    for (int j = 0; j < 3; ++j) x[j] = mad(-r, n[j], c[j]);
    return gap;
}

/* unit tangents: the coordinate axis least aligned with n, crossed with n and normalised; t2 = n x t1 */
static void tangents(const float n[3], float t1[3], float t2[3]) {
    const float ax = fabsf(n[0]), ay = fabsf(n[1]), az = fabsf(n[2]);
    float c[3];
    if (ax <= ay && ax <= az) { c[0] = 0.0f; c[1] = -n[2]; c[2] = n[1]; }
    else if (ay <= az) { c[0] = n[2]; c[1] = 0.0f; c[2] = -n[0]; }
    else { c[0] = -n[1]; c[1] = n[0]; c[2] = 0.0f; }
    const float rs = spec_rsqrt(mad(c[0], c[0], mad(c[1], c[1], c[2] * c[2])));
    for (int i = 0; i < 3; ++i) t1[i] = c[i] * rs;
    cross3(n, t1, t2);
}

/* ---- solver state of one substep ---- */
typedef struct {
    int robot;               /* the mover is the gripper (a collision sphere); else body `ma` (a cube's corner) */
    int ma, tb;              /* mover / target free body: 0 cubeA, 1 cubeB, 2 plate; -1 none (static target) */
    int target;              /* T_* */
    float d[3][3];           /* n, t1, t2 */
    float g[3][16];          /* robot rows in generalized coordinates (world spec v3, below) */
    float aa[3][3], ab[3][3];/* (mover arm) x d, (target arm) x d */
    float meff[3], bias, lam[3];
    int sphere;
} contact_t;

#define MAX_CONTACTS 40
typedef struct {
    const m3o_panda_scene* sc;
    float h, inv_h;
    float qd[9], p[9];       /* joint velocities and drive impulses of a world with robot rows */
    float* bv[3];            /* the free bodies' linear / angular velocities (into the world) */
    float* bw[3];
    float invm[3], invI[3];
    int rotates[3];
    float invIj[9];
    int n;
    contact_t c[MAX_CONTACTS];
} solver_t;

static __thread int g_last_robot_rows, g_last_body_rows;
int m3o_panda_last_rows(int* robot_rows, int* body_rows) {
    if (robot_rows) *robot_rows = g_last_robot_rows;
    if (body_rows) *body_rows = g_last_body_rows;
    return g_last_robot_rows + g_last_body_rows;
}

/* gripper geometry of one configuration: hand frame, the arm's Jacobian columns at the hand origin, the spheres */
typedef struct {
    float ph[3], hx[3], hy[3], hz[3];
    float Jv[7][3], Jw[7][3];
    float sc_c[4][3], sc_r[4];    /* tip left, tip right, hand, held cube */
    int n_spheres;
} gripper_t;

static void gripper_geometry(const m3o_panda_scene* sc, const m3o_panda_world* w, gripper_t* g) {
    m3o_panda_links L;
    m3o_panda_fk(sc, w->q, &L);
    for (int i = 0; i < 3; ++i) { g->ph[i] = L.pos[8][i]; g->hx[i] = L.ax[8][i]; g->hy[i] = L.ay[8][i]; g->hz[i] = L.az[8][i]; }
    for (int j = 0; j < 7; ++j) {
        float lever[3];
        for (int i = 0; i < 3; ++i) { g->Jw[j][i] = L.az[j + 1][i]; lever[i] = g->ph[i] - L.pos[j + 1][i]; }
        cross3(g->Jw[j], lever, g->Jv[j]);
    }
    for (int i = 0; i < 3; ++i) {
        g->sc_c[0][i] = mad(sc->tip_z, g->hz[i], L.pos[9][i]);
        g->sc_c[1][i] = mad(sc->tip_z, g->hz[i], L.pos[10][i]);
        g->sc_c[2][i] = mad(sc->hand_z, g->hz[i], g->ph[i]);
        g->sc_c[3][i] = w->cubeA[i];
    }
    g->sc_r[0] = sc->tip_r; g->sc_r[1] = sc->tip_r; g->sc_r[2] = sc->hand_r; g->sc_r[3] = sc->cube_half;
    g->n_spheres = (w->held != 0.0f) ? 4 : 3;
}

/* the joint-space row of direction d at the point x of the gripper (sphere s: the finger columns) */
static void robot_row(const gripper_t* g, int s, int held, const float x[3], const float d[3], float J[9]) {
    float rho[3], m[3];
    for (int i = 0; i < 3; ++i) rho[i] = x[i] - g->ph[i];
    cross3(rho, d, m);
    for (int j = 0; j < 7; ++j) J[j] = dotm(d, g->Jv[j]) + dotm(m, g->Jw[j]);
    const float dy = dotm(d, g->hy);
    J[7] = (s == 0 && !held) ? dy : 0.0f;
    J[8] = (s == 1 && !held) ? -dy : 0.0f;
}

/* ---- world spec v3: the gripper contacts' rows in GENERALIZED coordinates ---------------------------------------------
 * A sample's generalized velocity has 16 entries: the 9 joint velocities | the linear (3) and angular (3) velocity of the
 * free body the row's sphere touches | 0.  A gripper contact's row of direction d is g = (J_0..J_8 | -d | -(r_t x d) | 0):
 * the joint-space row as before, minus the direction and minus the target arm crossed with it (zeros for a static target;
 * the plate does not rotate: zeros in its angular entries); inverse masses m = (1/I_0..1/I_8 | 1/m_b x 3 | 1/I_b x 3 | 0).
 *   row velocity        v = SUM16(g_l u_l)
 *   effective mass      1 / SUM16((g_l m_l) g_l)
 *   impulse dl applied  u_l <- mad(g_l m_l, dl, u_l)
 * where SUM16 is the FIXED pairwise tree ((x0+x1)+(x2+x3)) + ((x4+x5)+(x6+x7)) ... of the sixteen rounded products (no fused
 * multiply-add inside the sum).  Why: the product's kernel holds one generalized coordinate per lane, sixteen lanes per
 * sample, and forms the sum with a four-step butterfly across them (csrc/panda_dyn.hpp, LPS = 16); v2's serial chain of
 * nine fused multiply-adds per row and visit was what C4's pick rollout spent its time in.  The entries of a row that has
 * no free target body are +0 and so are the velocities they multiply. */
static float sum16(const float x[16]) {
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = x[2 * i] + x[2 * i + 1];
    for (int i = 0; i < 4; ++i) b[i] = a[2 * i] + a[2 * i + 1];
    return (b[0] + b[1]) + (b[2] + b[3]);
}
/* (exported for tests/test_oracle_panda.py: the lane butterflies of the product's kernel, emulated, against these trees) */
float m3o_sum16(const float x[16]) { return sum16(x); }
static void gen_inv_mass(const solver_t* S, const contact_t* c, float m[16]) {
    const int b = (c->tb == 2) ? 2 : 0;       /* (both cubes have the same mass; a static target's entries multiply zeros) */
    for (int j = 0; j < 9; ++j) m[j] = S->invIj[j];
    for (int i = 0; i < 3; ++i) { m[9 + i] = S->invm[b]; m[12 + i] = S->invI[b]; }
    m[15] = 0.0f;
}
static void gen_get(const solver_t* S, const contact_t* c, const float* qd, float u[16]) {
    for (int j = 0; j < 9; ++j) u[j] = qd[j];
    for (int i = 0; i < 3; ++i) {
        u[9 + i] = (c->tb >= 0) ? S->bv[c->tb][i] : 0.0f;
        u[12 + i] = (c->tb >= 0) ? S->bw[c->tb][i] : 0.0f;
    }
    u[15] = 0.0f;
}
static void gen_put(solver_t* S, const contact_t* c, const float u[16]) {
    for (int j = 0; j < 9; ++j) S->qd[j] = u[j];
    if (c->tb >= 0) for (int i = 0; i < 3; ++i) { S->bv[c->tb][i] = u[9 + i]; S->bw[c->tb][i] = u[12 + i]; }
}
static float gen_vrel(const solver_t* S, const contact_t* c, int r, const float* qd) {
    float u[16], x[16];
    gen_get(S, c, qd, u);
    for (int l = 0; l < 16; ++l) x[l] = c->g[r][l] * u[l];
    return sum16(x);
}
static void gen_apply(solver_t* S, const contact_t* c, int r, float dl) {
    float u[16], m[16];
    gen_get(S, c, S->qd, u);
    gen_inv_mass(S, c, m);
    for (int l = 0; l < 16; ++l) u[l] = mad(c->g[r][l] * m[l], dl, u[l]);
    gen_put(S, c, u);
}

/* ---- world spec v3: the cubes' manifold rows.  A free cube has 8 generalized coordinates (v_x v_y v_z w_x w_y w_z 0 0); a
 * manifold row of direction d at the point X is (d | (X - p_M) x d | 0 0) on the mover and, for a cubeA-against-cubeB row,
 * (-d | -(X - p_T) x d | 0 0) on the target; inverse masses (1/m x 3 | 1/I x 3 | 0 0).
 *   row velocity   SUM8(g_M u_M) [+ SUM8(g_T u_T)]      SUM8 = ((x0+x1)+(x2+x3)) + ((x4+x5)+(x6+x7)), x6 = x7 = +0
 *   effective mass 1 / (SUM8((g_M m) g_M) [+ SUM8((g_T m) g_T)])
 *   impulse        u_l <- mad(g_l m_l, dl, u_l)
 * (the product's kernel: cubeA's coordinates on lanes 0-7 of a sample's sixteen, cubeB's on 8-15; SUM8 = a three-step
 * butterfly inside each half). */
static float sum8_6(float x0, float x1, float x2, float x3, float x4, float x5) {
    return ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + 0.0f);
}
float m3o_sum8_6(const float x[6]) { return sum8_6(x[0], x[1], x[2], x[3], x[4], x[5]); }
static float bodyrow_vel(const solver_t* S, int b, const float d[3], const float a[3], float sgn) {
    const float* v = S->bv[b];
    const float* w = S->bw[b];
    return sum8_6((sgn * d[0]) * v[0], (sgn * d[1]) * v[1], (sgn * d[2]) * v[2], (sgn * a[0]) * w[0], (sgn * a[1]) * w[1], (sgn * a[2]) * w[2]);
}
static float bodyrow_k(const solver_t* S, int b, const float d[3], const float a[3], float sgn) {
    const float im = S->invm[b], iI = S->invI[b];
    float g[6] = {sgn * d[0], sgn * d[1], sgn * d[2], sgn * a[0], sgn * a[1], sgn * a[2]};
    return sum8_6((g[0] * im) * g[0], (g[1] * im) * g[1], (g[2] * im) * g[2], (g[3] * iI) * g[3], (g[4] * iI) * g[4], (g[5] * iI) * g[5]);
}
static void bodyrow_apply(solver_t* S, int b, const float d[3], const float a[3], float sgn, float dl) {
    const float im = S->invm[b], iI = S->invI[b];
    for (int i = 0; i < 3; ++i) {
        S->bv[b][i] = mad((sgn * d[i]) * im, dl, S->bv[b][i]);
        S->bw[b][i] = mad((sgn * a[i]) * iI, dl, S->bw[b][i]);
    }
}

/* effective masses and the bias of a contact whose directions, rows and arms are set */
#define CAPTURE_DX 0.045f
#define CAPTURE_UP 0.08f
#define CAPTURE_DN 0.03f
#define CAPTURE_ALIGN 0.9f
#define PADS_DX 0.035f
#define PADS_DZ 0.03f
static void contact_prepare(solver_t* S, contact_t* c, float gap) {
    const m3o_panda_scene* sc = S->sc;
    for (int r = 0; r < 3; ++r) {
        float k;
        if (c->robot) {
            float m[16], x[16];
            gen_inv_mass(S, c, m);
            for (int l = 0; l < 16; ++l) x[l] = (c->g[r][l] * m[l]) * c->g[r][l];
            k = sum16(x);
        } else {
            k = bodyrow_k(S, c->ma, c->d[r], c->aa[r], 1.0f);
            if (c->tb >= 0) k = k + bodyrow_k(S, c->tb, c->d[r], c->ab[r], -1.0f);
        }
        c->meff[r] = spec_rcp(k); /* world spec v3.1 */
        c->lam[r] = 0.0f;
    }
    if (gap > 0.0f) {
        c->bias = gap * S->inv_h;
    } else {
        float pen = fmaxf(-gap - sc->slop, 0.0f);
        float push = fminf((sc->baumgarte * pen) * S->inv_h, sc->max_bias);
        c->bias = -push;
    }
}

static float contact_vrel(const solver_t* S, const contact_t* c, int r, const float* qd) {
    if (c->robot) return gen_vrel(S, c, r, qd);
    float v = bodyrow_vel(S, c->ma, c->d[r], c->aa[r], 1.0f);
    if (c->tb >= 0) v = v + bodyrow_vel(S, c->tb, c->d[r], c->ab[r], -1.0f);
    return v;
}

static void contact_solve(solver_t* S, contact_t* c) {
    /* the friction rows first (bounded by the normal impulse of the previous pass), the normal row last: what a
     * pass leaves exactly satisfied is non-penetration */
    for (int rr = 0; rr < 3; ++rr) {
        const int r = (rr + 1) % 3;
        const float v = contact_vrel(S, c, r, S->qd);
        float dl = -c->meff[r] * ((r == 0) ? v + c->bias : v);
        const float l0 = c->lam[r];
        float l1 = l0 + dl;
        if (r == 0) l1 = fmaxf(l1, 0.0f);
        else { const float mx = S->sc->mu * c->lam[0]; l1 = fminf(fmaxf(l1, -mx), mx); }
        c->lam[r] = l1;
        dl = l1 - l0;
        if (c->robot) gen_apply(S, c, r, dl);
        else {
            bodyrow_apply(S, c->ma, c->d[r], c->aa[r], 1.0f, dl);
            if (c->tb >= 0) bodyrow_apply(S, c->tb, c->d[r], c->ab[r], -1.0f, dl);
        }
    }
}

/* A cube against a box, face to face: the 4 corners of the cube's face that faces the box against the box's face
 * that faces the cube (its plane; the normal is that face's).  A corner that lies beyond the face's footprint is
 * moved onto the footprint's boundary and takes the height of the cube's face there -- for two aligned boxes the four
 * points are the corners of the overlap rectangle -- as long as the two faces oppose each other within 45 degrees;
 * otherwise it makes no contact.  x: the contact points (on the box's face plane), r = x - pb, gap: height above the
 * face.  Returns 0 when the cube is not near the box (or its centre is inside it). */
typedef struct { float r[4][3], x[4][3], n[3], gap[4]; int centre_over; } manifold_t;
static int cube_manifold(const m3o_panda_scene* sc, const float pb[3], const float Rb[9], const box_t* tgt, manifold_t* m) {
    float l[3], dd[3], dw[3], dlc[3];
    box_local(tgt, pb, l);
    for (int i = 0; i < 3; ++i) dd[i] = fminf(fmaxf(l[i], -tgt->e[i]), tgt->e[i]) - l[i];
    const float d2 = mad(dd[0], dd[0], mad(dd[1], dd[1], dd[2] * dd[2]));
    const float lim = 0.0434f + sc->contact_offset;     /* the cube's bounding radius, sqrt(3) * 0.025 rounded up */
    if (!(d2 > 1.0e-12f) || d2 > lim * lim) return 0;
    int pref = 0;
    if (fabsf(dd[1]) > fabsf(dd[pref])) pref = 1;
    if (fabsf(dd[2]) > fabsf(dd[pref])) pref = 2;
    const int a1 = (pref + 1) % 3, a2 = (pref + 2) % 3;
    const float side = (dd[pref] <= 0.0f) ? 1.0f : -1.0f;    /* the cube's centre lies beyond the box's +face / -face */
    m->centre_over = (dd[a1] == 0.0f && dd[a2] == 0.0f);       /* ... and projects into that face */
    box_world(tgt, dd, dw);
    for (int i = 0; i < 3; ++i) dlc[i] = mad(dw[0], Rb[0 * 3 + i], mad(dw[1], Rb[1 * 3 + i], dw[2] * Rb[2 * 3 + i]));
    int f = 0;
    if (fabsf(dlc[1]) > fabsf(dlc[f])) f = 1;
    if (fabsf(dlc[2]) > fabsf(dlc[f])) f = 2;
    const float e = sc->cube_half;
    const float sgn = (dlc[f] >= 0.0f) ? 1.0f : -1.0f;
    /* the cube's facing face normal in the box's frame; its component along the box's facing normal */
    float nmw[3], nml[3], nl[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 3; ++k) nmw[k] = sgn * Rb[k * 3 + f];
    if (tgt->oriented) for (int i = 0; i < 3; ++i) nml[i] = mad(nmw[0], tgt->R[0 * 3 + i], mad(nmw[1], tgt->R[1 * 3 + i], nmw[2] * tgt->R[2 * 3 + i]));
    else for (int i = 0; i < 3; ++i) nml[i] = nmw[i];
    const float mz = side * nml[pref];
    const int clip = (mz <= -0.7f);
    const float rz = clip ? 1.0f / mz : 0.0f;
    nl[pref] = side;
    box_world(tgt, nl, m->n);
    for (int j = 0; j < 4; ++j) {
        float cl[3], xw[3], lc[3], lq[3], qw[3];
        cl[f] = sgn * e;
        cl[(f + 1) % 3] = (j & 1) ? e : -e;
        cl[(f + 2) % 3] = (j & 2) ? e : -e;
        for (int k = 0; k < 3; ++k)
            xw[k] = pb[k] + mad(Rb[k * 3 + 0], cl[0], mad(Rb[k * 3 + 1], cl[1], Rb[k * 3 + 2] * cl[2]));
        box_local(tgt, xw, lc);
        const float hgt = side * lc[pref] - tgt->e[pref];
        const float q1 = fminf(fmaxf(lc[a1], -tgt->e[a1]), tgt->e[a1]);
        const float q2 = fminf(fmaxf(lc[a2], -tgt->e[a2]), tgt->e[a2]);
        const float s1 = q1 - lc[a1], s2 = q2 - lc[a2];
        const int moved = (s1 != 0.0f) || (s2 != 0.0f);
        if (moved && !clip) m->gap[j] = 1.0f;
        else if (moved) m->gap[j] = hgt - mad(nml[a1], s1, nml[a2] * s2) * rz;
        else m->gap[j] = hgt;
        lq[a1] = q1; lq[a2] = q2; lq[pref] = side * tgt->e[pref];
        box_world(tgt, lq, qw);
        for (int k = 0; k < 3; ++k) { m->x[j][k] = tgt->p[k] + qw[k]; m->r[j][k] = m->x[j][k] - pb[k]; }
    }
    return 1;
}

/* a cube makes contact with ONE of the two static boxes: the nearer to its centre (the table on a tie) */
static int nearer_static(const m3o_panda_scene* sc, const float pb[3]) {
    float d2[2];
    const float* stat[2] = {sc->table, sc->shelf};
    for (int s = 0; s < 2; ++s) {
        float dd[3];
        for (int i = 0; i < 3; ++i) {
            const float l = pb[i] - stat[s][i];
            dd[i] = fminf(fmaxf(l, -stat[s][3 + i]), stat[s][3 + i]) - l;
        }
        d2[s] = mad(dd[0], dd[0], mad(dd[1], dd[1], dd[2] * dd[2]));
    }
    return (d2[0] <= d2[1]) ? T_TABLE : T_SHELF;
}

/* a cube rests: all four contact points of its facing face within rest_gap of the top of its static box */
static int cube_on_static(const m3o_panda_scene* sc, const float cube[13]) {
    float R[9];
    body_rot(cube + 3, R);
    box_t b;
    manifold_t m;
    box_static(nearer_static(sc, cube) == T_TABLE ? sc->table : sc->shelf, &b);
    if (!cube_manifold(sc, cube, R, &b, &m)) return 0;
    for (int j = 0; j < 4; ++j) if (!(m.gap[j] < sc->rest_gap && m.n[2] >= 0.99f)) return 0;
    return 1;
}

/* counts[0]: contacts made, [1]: of them carrying the mover (n_z >= 0.99, gap below rest_gap), [2]: carried by it
 * (n_z <= -0.99), [3]: the mover's centre projects into the facing face */
static void add_cube_manifold(solver_t* S, int ma, const float* body, const float Rb[9], const box_t* tgt, int target,
                              int tb, const float* tbody, int counts[4]) {
    const m3o_panda_scene* sc = S->sc;
    manifold_t m;
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (!cube_manifold(sc, body, Rb, tgt, &m)) return;
    counts[3] = m.centre_over;
    for (int j = 0; j < 4; ++j) {
        if (!(m.gap[j] < sc->contact_offset)) continue;
        ++counts[0];
        if (m.n[2] >= 0.99f && m.gap[j] < sc->rest_gap) ++counts[1];
        if (m.n[2] <= -0.99f && m.gap[j] < sc->rest_gap) ++counts[2];
        contact_t* c = &S->c[S->n++];
        memset(c, 0, sizeof(*c));
        c->robot = 0; c->ma = ma; c->tb = tb; c->target = target;
        for (int i = 0; i < 3; ++i) c->d[0][i] = m.n[i];
        tangents(c->d[0], c->d[1], c->d[2]);
        float rt[3] = {0, 0, 0};
        if (tb >= 0) for (int i = 0; i < 3; ++i) rt[i] = m.x[j][i] - tbody[i];
        for (int r = 0; r < 3; ++r) {
            cross3(m.r[j], c->d[r], c->aa[r]);
            if (tb >= 0) cross3(rt, c->d[r], c->ab[r]);
        }
        contact_prepare(S, c, m.gap[j]);
    }
}

/* ... or on the other cube: its centre over that cube's top face and at least three of the face-to-face contact
 * points carrying it */
static int cube_on_cube(const m3o_panda_scene* sc, const float up[13], const float lo[13]) {
    const float e[3] = {sc->cube_half, sc->cube_half, sc->cube_half};
    float Ru[9];
    box_t bl;
    manifold_t m;
    body_rot(up + 3, Ru);
    box_body(lo, e, 1, &bl);
    int n = 0;
    if (!cube_manifold(sc, up, Ru, &bl, &m) || !m.centre_over || !(m.n[2] >= 0.99f)) return 0;
    for (int j = 0; j < 4; ++j) if (m.gap[j] < sc->rest_gap) ++n;
    return n >= 3;
}

static void integrate_quat(float q[4], const float w[3], float h) {
    if (w[0] == 0.0f && w[1] == 0.0f && w[2] == 0.0f) return;
    const float hh = 0.5f * h;
    const float tx = mad(w[0], q[3], mad(w[1], q[2], -(w[2] * q[1])));
    const float ty = mad(w[1], q[3], mad(w[2], q[0], -(w[0] * q[2])));
    const float tz = mad(w[2], q[3], mad(w[0], q[1], -(w[1] * q[0])));
    const float tw = -mad(w[0], q[0], mad(w[1], q[1], w[2] * q[2]));
    float n[4] = {mad(hh, tx, q[0]), mad(hh, ty, q[1]), mad(hh, tz, q[2]), mad(hh, tw, q[3])};
    const float rs = spec_rsqrt(mad(n[0], n[0], mad(n[1], n[1], mad(n[2], n[2], n[3] * n[3]))));
    for (int i = 0; i < 4; ++i) q[i] = n[i] * rs;
}

void m3o_panda_step(const m3o_panda_scene* sc, m3o_panda_world* w, const float u[9]) {
    const float h = sc->dt / (float)sc->substeps;
    const float inv_h = 1.0f / h;
    const float hD = h * sc->drive_damping;
    const float cube_e[3] = {sc->cube_half, sc->cube_half, sc->cube_half};
    float* bodies[3] = {w->cubeA, w->cubeB, w->obs};
    for (int sub = 0; sub < sc->substeps; ++sub) {
        /* 0. release: either finger commanded open lets a held cube go where it is, at rest */
        if (w->held != 0.0f && (u[7] >= 0.0f || u[8] >= 0.0f)) {
            w->held = 0.0f;
            for (int i = 7; i < 13; ++i) w->cubeA[i] = 0.0f;
            w->awake[0] = 1.0f;
        }
        const int held = (w->held != 0.0f);
        /* 1. the joint servos in closed form (implicit damper, torque + velocity limits): what a substep without
         * gripper contacts does, and the velocities the contact culling predicts with */
        float a_[9], rden[9], invI[9], pmax[9], qd1[9];
        for (int i = 0; i < 9; ++i) {
            a_[i] = hD / sc->inertia[i];
            rden[i] = 1.0f / (1.0f + a_[i]);
            invI[i] = 1.0f / sc->inertia[i];
            pmax[i] = h * sc->effort[i];
            if (held && i >= 7) { qd1[i] = 0.0f; continue; }      /* fingers locked on the cube */
            const float dv = pmax[i] * invI[i];
            float v = mad(a_[i], u[i], w->qd[i]) * rden[i];
            const float tau = sc->drive_damping * (u[i] - v);
            if (tau > sc->effort[i]) v = w->qd[i] + dv;
            if (tau < -sc->effort[i]) v = w->qd[i] - dv;
            qd1[i] = fminf(fmaxf(v, -sc->vlim[i]), sc->vlim[i]);
        }
        /* 2. the gripper's collision spheres against the boxes: one contact per sphere (the smallest gap, targets in
         * the order table, shelf_stand, cubeA, cubeB, plate), kept if it is predicted to close */
        solver_t S;
        memset(&S, 0, sizeof(S));
        S.sc = sc; S.h = h; S.inv_h = inv_h; S.n = 0;
        for (int b = 0; b < 3; ++b) { S.bv[b] = bodies[b] + 7; S.bw[b] = bodies[b] + 10; }
        S.invm[0] = S.invm[1] = 1.0f / sc->cube_m;
        S.invI[0] = S.invI[1] = 1.0f / ((sc->cube_m * ((2.0f * sc->cube_half) * (2.0f * sc->cube_half))) / 6.0f);
        S.invm[2] = 1.0f / sc->obs_m; S.invI[2] = 0.0f;
        S.rotates[0] = S.rotates[1] = 1; S.rotates[2] = 0;
        for (int i = 0; i < 9; ++i) S.invIj[i] = invI[i];
        gripper_t g;
        gripper_geometry(sc, w, &g);
        box_t tgt[5];
        box_static(sc->table, &tgt[T_TABLE]);
        box_static(sc->shelf, &tgt[T_SHELF]);
        box_body(w->cubeA, cube_e, 1, &tgt[T_CUBEA]);
        box_body(w->cubeB, cube_e, 1, &tgt[T_CUBEB]);
        box_body(w->obs, sc->obs_half, 0, &tgt[T_OBS]);
        /* the capture volume (spec v2.1): there the pads, not the tip spheres, act on cubeA */
        int in_channel = 0;
        if (!held) {
            const float dd[3] = {w->cubeA[0] - g.ph[0], w->cubeA[1] - g.ph[1], w->cubeA[2] - g.ph[2]};
            const float cx = dot3(dd, g.hx), cy = dot3(dd, g.hy), cz = dot3(dd, g.hz);
            float Rc[9];
            quat2mat(&w->cubeA[3], Rc);
            float ay = 0.0f, az = 0.0f;
            for (int j = 0; j < 3; ++j) {
                float col[3] = {Rc[j], Rc[3 + j], Rc[6 + j]};
                ay = fmaxf(ay, fabsf(dot3(g.hy, col)));
                az = fmaxf(az, fabsf(dot3(g.hz, col)));
            }
            /* spec v2.1, the CAPTURE volume: wider than the grasp rule's region -- the cube's centre between the pads' faces,
               within 4.5 cm of their centre line along the hand's x, from 3 cm above the grasp height (pads below the cube's
               middle) to 8 cm below it (the open gripper still coming down over the cube), the cube roughly upright; no yaw
               condition.  The small planner of the reference's shipped size (K = 200) arrives 2-3 cm off the centre line;
               with the exclusion limited to the grasp region a tip sphere then landed on the cube's edge on the way down
               and knocked it, and a cube drifting out of the region while the pads close met the tip spheres 12 mm inside
               it and was shot across the table: 39 of 60 picks (round 3's kinematic world: 60).  Inside the capture
               volume the pads' position-level model (sweep, latch) is the gripper's only action on cubeA, as in spec v1.1. */
            (void)ay;
            in_channel = fabsf(cx) <= CAPTURE_DX && (cz - sc->grasp_z) <= CAPTURE_UP && (cz - sc->grasp_z) >= -CAPTURE_DN &&
                         cy < w->q[7] && cy > -w->q[8] && az >= CAPTURE_ALIGN;
        }
        int robot_rows = 0;
        int touched[3] = {0, 0, 0};     /* a gripper row acts on this free body */
        for (int s = 0; s < g.n_spheres; ++s) {
            float best_gap = sc->contact_offset, bn[3] = {0, 0, 0}, bx[3] = {0, 0, 0};
            int best = -1;
            for (int t = 0; t < 5; ++t) {
                if (t == T_CUBEA && (held || (s < 2 && in_channel))) continue;
                float n[3], x[3];
                const float gap = pt_box(&tgt[t], g.sc_c[s], g.sc_r[s], n, x);
                if (gap < best_gap) { best_gap = gap; best = t; for (int i = 0; i < 3; ++i) { bn[i] = n[i]; bx[i] = x[i]; } }
            }
            if (best < 0) continue;
            contact_t c;
            memset(&c, 0, sizeof(c));
            c.robot = 1; c.ma = -1; c.target = best;
            c.tb = (best >= T_CUBEA) ? best - T_CUBEA : -1;
            for (int i = 0; i < 3; ++i) c.d[0][i] = bn[i];
            tangents(c.d[0], c.d[1], c.d[2]);
            float rt[3] = {0, 0, 0};
            if (c.tb >= 0) for (int i = 0; i < 3; ++i) rt[i] = bx[i] - bodies[c.tb][i];
            for (int r = 0; r < 3; ++r) {
                robot_row(&g, s, held, bx, c.d[r], c.g[r]);
                for (int l = 9; l < 16; ++l) c.g[r][l] = 0.0f;
                if (c.tb >= 0) {
                    cross3(rt, c.d[r], c.ab[r]);
                    for (int i = 0; i < 3; ++i) {
                        c.g[r][9 + i] = -c.d[r][i];
                        c.g[r][12 + i] = (c.tb < 2) ? -c.ab[r][i] : 0.0f;      /* the plate does not rotate */
                    }
                }
            }
            /* culling: the gap predicted for the end of the substep from the servo's velocities */
            const float vn0 = contact_vrel(&S, &c, 0, qd1);
            if (!(mad(h, vn0, best_gap) < sc->act_margin)) continue;
            contact_prepare(&S, &c, best_gap);
            c.sphere = s;
            if (w->warm_t[s] == (float)(best + 1)) c.lam[0] = w->warm_l[s];     /* warm start: last substep's normal impulse */
            S.c[S.n++] = c;
            ++robot_rows;
            if (c.tb >= 0) { touched[c.tb] = 1; if (c.tb < 2) w->awake[c.tb] = 1.0f; }
        }
        /* 3. an awake cube wakes the other one when they are close */
        const int freeA = !held;
        if (freeA && (w->awake[0] != 0.0f) != (w->awake[1] != 0.0f)) {
            const float dx = w->cubeA[0] - w->cubeB[0], dy = w->cubeA[1] - w->cubeB[1], dz = w->cubeA[2] - w->cubeB[2];
            const float lim = 2.0f * 0.0434f + sc->contact_offset;
            if (mad(dx, dx, mad(dy, dy, dz * dz)) < lim * lim) { w->awake[0] = 1.0f; w->awake[1] = 1.0f; }
        }
        const int act[2] = {freeA && w->awake[0] != 0.0f, w->awake[1] != 0.0f};
        /* 4. gravity, then the cubes' corner contacts */
        for (int b = 0; b < 2; ++b) if (act[b]) bodies[b][9] = mad(-sc->g, h, bodies[b][9]);
        int cAs[4] = {0, 0, 0, 0}, cAB[4] = {0, 0, 0, 0}, cBs[4] = {0, 0, 0, 0};
        float RA[9], RB[9];
        body_rot(w->cubeA + 3, RA);
        body_rot(w->cubeB + 3, RB);
        if (act[0]) { const int t = nearer_static(sc, w->cubeA); add_cube_manifold(&S, 0, w->cubeA, RA, &tgt[t], t, -1, NULL, cAs); }
        if (act[0] && act[1]) add_cube_manifold(&S, 0, w->cubeA, RA, &tgt[T_CUBEB], T_CUBEB, 1, w->cubeB, cAB);
        if (act[1]) { const int t = nearer_static(sc, w->cubeB); add_cube_manifold(&S, 1, w->cubeB, RB, &tgt[t], t, -1, NULL, cBs); }
        /* 5. velocity passes: the joint drives as rows (only in a world with gripper contacts), then the contacts */
        if (robot_rows) for (int i = 0; i < 9; ++i) { S.qd[i] = (held && i >= 7) ? 0.0f : w->qd[i]; S.p[i] = 0.0f; }
        for (int k = 0; k < robot_rows; ++k) {       /* the warm-start impulses act before the first pass */
            contact_t* c = &S.c[k];
            const float dl = c->lam[0];
            if (dl == 0.0f) continue;
            gen_apply(&S, c, 0, dl);
        }
        for (int pass = 0; pass < sc->iters; ++pass) {
            if (robot_rows) {
                for (int i = 0; i < 9; ++i) {
                    if (held && i >= 7) continue;
                    const float e = mad(hD, u[i] - S.qd[i], -S.p[i]);
                    float dp = e * rden[i];
                    const float p1 = fminf(fmaxf(S.p[i] + dp, -pmax[i]), pmax[i]);
                    dp = p1 - S.p[i];
                    S.p[i] = p1;
                    S.qd[i] = mad(invI[i], dp, S.qd[i]);
                }
            }
            for (int k = 0; k < S.n; ++k) contact_solve(&S, &S.c[k]);
        }
        /* ... and one more sweep over the contacts alone (isaacgym_wrapper.py:29, num_velocity_iterations = 1): the
         * drives pull against the contacts in every pass; what the substep ends on is the contacts' word */
        for (int k = 0; k < S.n; ++k) contact_solve(&S, &S.c[k]);
        for (int i = 0; i < 9; ++i) {
            float v = robot_rows ? fminf(fmaxf(S.qd[i], -sc->vlim[i]), sc->vlim[i]) : qd1[i];
            if (held && i >= 7) v = 0.0f;
            w->qd[i] = v;
        }
        /* net contact forces on table / shelf_stand / cubeB: this substep's impulses / h (a step reports its last) */
        float ft[3] = {0, 0, 0}, fs[3] = {0, 0, 0}, fb[3] = {0, 0, 0};
        for (int k = 0; k < S.n; ++k) {
            const contact_t* c = &S.c[k];
            for (int r = 0; r < 3; ++r) {
                const float sI = c->lam[r] * inv_h;
                float* dst = (c->target == T_TABLE) ? ft : (c->target == T_SHELF) ? fs : (c->target == T_CUBEB) ? fb : NULL;
                if (dst) for (int i = 0; i < 3; ++i) dst[i] = mad(-sI, c->d[r][i], dst[i]);
                if (!c->robot && c->ma == 1) for (int i = 0; i < 3; ++i) fb[i] = mad(sI, c->d[r][i], fb[i]);
            }
        }
        for (int i = 0; i < 3; ++i) { w->f_table[i] = ft[i]; w->f_shelf[i] = fs[i]; w->f_cubeB[i] = fb[i]; }
        for (int s4 = 0; s4 < 4; ++s4) { w->warm_t[s4] = 0.0f; w->warm_l[s4] = 0.0f; }
        for (int k = 0; k < robot_rows; ++k) { w->warm_t[S.c[k].sphere] = (float)(S.c[k].target + 1); w->warm_l[S.c[k].sphere] = S.c[k].lam[0]; }
        g_last_robot_rows = robot_rows; g_last_body_rows = S.n - robot_rows;
        if (getenv("M3O_DEBUG")) {
            for (int k = 0; k < S.n; ++k) {
                const contact_t* c = &S.c[k];
                fprintf(stderr, "  sub %d contact %d robot %d ma %d tb %d target %d n (%.3f %.3f %.3f) bias %.4f lam (%.5f %.5f %.5f) meff %.4f\n",
                        sub, k, c->robot, c->ma, c->tb, c->target, c->d[0][0], c->d[0][1], c->d[0][2], c->bias, c->lam[0], c->lam[1], c->lam[2], c->meff[0]);
            }
        }
        /* 6. sleep: slow cubes that rest -- the whole facing face on the table / shelf_stand, or stacked on the other
         * cube (centre over its face, three of the face-to-face contacts carrying it) which rests on a static box -- and
         * that no gripper contact touches; two cubes in contact go to sleep together or not at all */
        {
            int slow[2], on_static[2], rests[2];
            for (int b = 0; b < 2; ++b) {
                const float* v = bodies[b] + 7;
                const float* om = bodies[b] + 10;
                slow[b] = act[b] && !touched[b] && dotm(v, v) < sc->sleep_v * sc->sleep_v && dotm(om, om) < sc->sleep_w * sc->sleep_w;
            }
            on_static[0] = cAs[1] == 4;
            on_static[1] = cBs[1] == 4;
            rests[0] = on_static[0] || (on_static[1] && cAB[3] && cAB[1] >= 3);     /* A on B: the face pushes A up */
            rests[1] = on_static[1] || (on_static[0] && cAB[3] && cAB[2] >= 3);     /* B on A: it pushes A down */
            const int pair = cAB[0] > 0;
            for (int b = 0; b < 2; ++b) {
                const int ok = pair ? (slow[0] && slow[1] && rests[0] && rests[1]) : (slow[b] && on_static[b]);
                if (!ok) continue;
                for (int i = 7; i < 13; ++i) bodies[b][i] = 0.0f;
                w->awake[b] = 0.0f;
            }
        }
        /* 7. integration: joints (position limits: clamp and stop), awake cubes, the plate */
        for (int i = 0; i < 9; ++i) {
            float q1 = mad(h, w->qd[i], w->q[i]);
            if (q1 < sc->qlo[i]) { q1 = sc->qlo[i]; w->qd[i] = 0.0f; }
            if (q1 > sc->qhi[i]) { q1 = sc->qhi[i]; w->qd[i] = 0.0f; }
            w->q[i] = q1;
        }
        for (int b = 0; b < 2; ++b) {
            if (!(b == 0 ? (freeA && w->awake[0] != 0.0f) : (w->awake[1] != 0.0f))) continue;
            for (int i = 0; i < 3; ++i) bodies[b][i] = mad(h, bodies[b][7 + i], bodies[b][i]);
            integrate_quat(bodies[b] + 3, bodies[b] + 10, h);
        }
        for (int i = 0; i < 3; ++i) w->obs[i] = mad(h, w->obs[7 + i], w->obs[i]);
        /* 8. kinematics of the new configuration; the grasp rule (spec v1.1, position level) */
        m3o_panda_links L;
        m3o_panda_fk(sc, w->q, &L);
        const float* ph = L.pos[8];
        const float *hx = L.ax[8], *hy = L.ay[8], *hz = L.az[8];
        if (w->held != 0.0f) {
            for (int i = 0; i < 3; ++i)
                w->cubeA[i] = ph[i] + ((w->rel_p[0] * hx[i] + w->rel_p[1] * hy[i]) + w->rel_p[2] * hz[i]);
            /* R_c = R_h * R_rel */
            float Rr[9];
            quat2mat(w->rel_q, Rr);
            frame_t c;
            for (int i = 0; i < 3; ++i) {
                c.x[i] = (hx[i] * Rr[0] + hy[i] * Rr[3]) + hz[i] * Rr[6];
                c.y[i] = (hx[i] * Rr[1] + hy[i] * Rr[4]) + hz[i] * Rr[7];
                c.z[i] = (hx[i] * Rr[2] + hy[i] * Rr[5]) + hz[i] * Rr[8];
            }
            mat2quat(&c, &w->cubeA[3]);
            for (int i = 7; i < 13; ++i) w->cubeA[i] = 0.0f;
        } else {
            /* grasp test */
            float d[3] = {w->cubeA[0] - ph[0], w->cubeA[1] - ph[1], w->cubeA[2] - ph[2]};
            float cx = dot3(d, hx), cy = dot3(d, hy), cz = dot3(d, hz);
            float Rc[9];
            quat2mat(&w->cubeA[3], Rc);
            float ay = 0.0f, az = 0.0f;
            for (int j = 0; j < 3; ++j) {
                float col[3] = {Rc[j], Rc[3 + j], Rc[6 + j]};
                ay = fmaxf(ay, fabsf(dot3(hy, col)));
                az = fmaxf(az, fabsf(dot3(hz, col)));
            }
            /* spec v1.1, the pad channel: the cube is "between the pads" when its centre lies inside the pads' footprint
             * along the hand's x and z (|cx| <= grasp_dx, |cz - grasp_z| <= grasp_dz: the pad's centre is on the cube's
             * face), it is aligned with the pads, and its centre lies between the two pad faces (-q8 < cy < q7). */
            int in_region = fabsf(cx) <= sc->grasp_dx && fabsf(cz - sc->grasp_z) <= sc->grasp_dz &&
                            cy < w->q[7] && cy > -w->q[8] && ay >= sc->grasp_align && az >= sc->grasp_align;
            /* spec v2.1: the pads MEET the cube's side faces in a wider region than the one in which they can hold it -- the
             * cube's centre between the pad faces, within 3.5 cm of their centre line along x (pad half width 1 cm + the cube's
             * 2.5) and 3 cm along z, the cube roughly upright: there they cannot enter it and closing pads sweep it; the
             * latch needs the grasp rule's region */
            const int in_pads = fabsf(cx) <= PADS_DX && fabsf(cz - sc->grasp_z) <= PADS_DZ && cy < w->q[7] && cy > -w->q[8] &&
                                az >= CAPTURE_ALIGN;
            if (in_pads) {
                float gap = w->q[7] + w->q[8];
                float wdt = 2.0f * sc->cube_half;
                if (gap < wdt) { /* pads cannot enter the cube: keep the difference, open to the width */
                    float mid = 0.5f * (w->q[7] - w->q[8]);
                    w->q[7] = 0.5f * wdt + mid; w->q[8] = 0.5f * wdt - mid;
                    gap = wdt;
                }
                if (u[7] < 0.0f && u[8] < 0.0f) {
                    /* closing pads sweep the cube along the hand's y so that it stays between them (what the
                     * physical fingers do to a cube that is off their centre line); it slides on its support:
                     * only the horizontal part of the displacement is applied, the horizontal velocity is lost */
                    float lo = sc->cube_half - w->q[8], hi = w->q[7] - sc->cube_half;
                    float cyn = fminf(fmaxf(cy, lo), hi);
                    if (cyn != cy) {
                        float sh = cyn - cy;
                        w->cubeA[0] = w->cubeA[0] + sh * hy[0];
                        w->cubeA[1] = w->cubeA[1] + sh * hy[1];
                        w->cubeA[7] = 0.0f; w->cubeA[8] = 0.0f;
                        d[0] = w->cubeA[0] - ph[0]; d[1] = w->cubeA[1] - ph[1];
                        cx = dot3(d, hx); cz = dot3(d, hz);
                    }
                }
                if (in_region && gap <= wdt + sc->grasp_tol && u[7] < 0.0f && u[8] < 0.0f) {
                    w->held = 1.0f;
                    w->qd[7] = 0.0f; w->qd[8] = 0.0f;
                    w->rel_p[0] = cx; w->rel_p[1] = 0.5f * (w->q[7] - w->q[8]); w->rel_p[2] = cz;
                    /* R_rel = R_h^T R_c */
                    frame_t r;
                    for (int j = 0; j < 3; ++j) {
                        float col[3] = {Rc[j], Rc[3 + j], Rc[6 + j]};
                        float* dstc = (j == 0) ? r.x : (j == 1) ? r.y : r.z;
                        dstc[0] = dot3(hx, col); dstc[1] = dot3(hy, col); dstc[2] = dot3(hz, col);
                    }
                    mat2quat(&r, w->rel_q);
                    for (int i = 7; i < 13; ++i) w->cubeA[i] = 0.0f;
                }
            }
        }
    }
}

/* The wrapper's tensors carry neither a "held" bit nor the cubes' sleep state: when a world is loaded from them
 * (rollout start, set_*_state_tensor) both are inferred from geometry -- held: cube inside the grasp region, aligned
 * with the pads and the pads closed on it; asleep: all six velocities exactly zero and the whole facing face within
 * the contact offset of the top of the table / the shelf_stand. */
void m3o_panda_infer_held(const m3o_panda_scene* sc, m3o_panda_world* w) {
    m3o_panda_links L;
    m3o_panda_fk(sc, w->q, &L);
    const float* ph = L.pos[8];
    const float *hx = L.ax[8], *hy = L.ay[8], *hz = L.az[8];
    float d[3] = {w->cubeA[0] - ph[0], w->cubeA[1] - ph[1], w->cubeA[2] - ph[2]};
    float cx = dot3(d, hx), cy = dot3(d, hy), cz = dot3(d, hz);
    float Rc[9];
    quat2mat(&w->cubeA[3], Rc);
    float ay = 0.0f, az = 0.0f;
    for (int j = 0; j < 3; ++j) {
        float col[3] = {Rc[j], Rc[3 + j], Rc[6 + j]};
        ay = fmaxf(ay, fabsf(dot3(hy, col)));
        az = fmaxf(az, fabsf(dot3(hz, col)));
    }
    /* held = in the pad channel (spec v1.1), pads closed on the cube and the cube centred between them */
    float gap = w->q[7] + w->q[8];
    float mid = 0.5f * (w->q[7] - w->q[8]);
    int in_region = fabsf(cx) <= sc->grasp_dx && fabsf(cz - sc->grasp_z) <= sc->grasp_dz &&
                    fabsf(cy - mid) <= sc->grasp_tol && ay >= sc->grasp_align && az >= sc->grasp_align;
    w->held = 0.0f;
    if (in_region && gap <= 2.0f * sc->cube_half + sc->grasp_tol) {
        w->held = 1.0f;
        w->rel_p[0] = cx; w->rel_p[1] = cy; w->rel_p[2] = cz;
        frame_t r;
        for (int j = 0; j < 3; ++j) {
            float col[3] = {Rc[j], Rc[3 + j], Rc[6 + j]};
            float* dstc = (j == 0) ? r.x : (j == 1) ? r.y : r.z;
            dstc[0] = dot3(hx, col); dstc[1] = dot3(hy, col); dstc[2] = dot3(hz, col);
        }
        mat2quat(&r, w->rel_q);
    }
    /* nothing is carried over from before the load: no warm-start impulses, no reported forces */
    for (int i = 0; i < 4; ++i) { w->warm_t[i] = 0.0f; w->warm_l[i] = 0.0f; }
    for (int i = 0; i < 3; ++i) { w->f_table[i] = 0.0f; w->f_shelf[i] = 0.0f; w->f_cubeB[i] = 0.0f; }
    float* cubes[2] = {w->cubeA, w->cubeB};
    for (int b = 0; b < 2; ++b) {
        int still = 1;
        for (int i = 7; i < 13; ++i) if (cubes[b][i] != 0.0f) still = 0;
        w->awake[b] = still ? 0.0f : 1.0f;      /* (provisional: the candidates) */
    }
    const int stat[2] = {cube_on_static(sc, w->cubeA), cube_on_static(sc, w->cubeB)};
    const int freeA = (w->held == 0.0f);
    const int stackA = freeA && !stat[0] && stat[1] && w->awake[0] == 0.0f && w->awake[1] == 0.0f && cube_on_cube(sc, w->cubeA, w->cubeB);
    const int stackB = freeA && !stat[1] && stat[0] && w->awake[0] == 0.0f && w->awake[1] == 0.0f && cube_on_cube(sc, w->cubeB, w->cubeA);
    /* (cube_on_cube: cubeA's facing face against cubeB's / cubeB's against cubeA's -- the step itself forms only the first) */
    if (!(stat[0] || stackA)) w->awake[0] = 1.0f;
    if (!(stat[1] || stackB)) w->awake[1] = 1.0f;
}

/* ---- costs on observables (what the reference reads through the wrapper getters) ---- */
float m3o_panda_cost_obs(const m3o_cfg* cfg, const m3o_panda_obs* o, int k) {
    const int half = cfg->K / 2;
    const int task = cfg->task;
    if (task == M3O_TASK_REACH) {
        /* get_panda_reach_cost: cost_functions.py:91-114 */
        float ee[3], goal[3];
        for (int i = 0; i < 3; ++i) ee[i] = (o->left[i] + o->right[i]) / 2.0f;
        goal[0] = o->cube0[0]; goal[1] = o->cube0[1]; goal[2] = o->cube0[2];
        if (!cfg->multi_modal || k < half) {
            goal[2] = goal[2] + cfg->pre_height_diff;
        } else {
            goal[0] = goal[0] - cfg->pre_height_diff * cfg->tilt_cos_theta;
            goal[2] = goal[2] + cfg->pre_height_diff *
                                    sqrtf(1.0f - cfg->tilt_cos_theta * cfg->tilt_cos_theta);
        }
        float dx = ee[0] - goal[0], dy = ee[1] - goal[1], dz = ee[2] - goal[2];
        float reach = sqrtf((dx * dx + dy * dy) + dz * dz);
        /* get_pick_tilt_cost: cost_functions.py:138-156 */
        float tilt = (cfg->multi_modal && k >= half) ? cfg->tilt_cos_theta : 0.0f;
        float ori = m3o_ori_ee2cube(o->left_q, o->cube_q, tilt, o->cube_q_half0);
        return 10.0f * reach + 3.0f * ori;
    }
    if (task == M3O_TASK_PICK) {
        /* get_panda_pick_cost: cost_functions.py:116-125 + get_motion_cost :158-169 */
        float dx = cfg->goal[0] - o->cube[0], dy = cfg->goal[1] - o->cube[1], dz = cfg->goal[2] - o->cube[2];
        float gc = sqrtf((dx * dx + dy * dy) + dz * dz);
        float ori = m3o_ori_cube2goal(o->cube_q, &cfg->goal[3]);
        float fx = (o->f_table[0] + 4.0f * o->f_shelf[0]) + o->f_cubeB[0];
        float fy = (o->f_table[1] + 4.0f * o->f_shelf[1]) + o->f_cubeB[1];
        float coll = fabsf(fx) + fabsf(fy);
        return (10.0f * gc + 15.0f * ori) + ((coll > 0.1f) ? 1000.0f : 0.0f);
    }
    if (task == M3O_TASK_PLACE) {
        /* get_panda_place_cost: cost_functions.py:127-136 */
        float dx = o->left[0] - o->right[0], dy = o->left[1] - o->right[1], dz = o->left[2] - o->right[2];
        return 2.0f * (1.0f - sqrtf((dx * dx + dy * dy) + dz * dz));
    }
    return 0.0f;
}

void m3o_panda_observe(const m3o_panda_scene* sc, const m3o_panda_world* w, m3o_panda_obs* o) {
    m3o_panda_links L;
    m3o_panda_fk(sc, w->q, &L);
    for (int i = 0; i < 3; ++i) {
        o->left[i] = L.pos[9][i]; o->right[i] = L.pos[10][i];
        o->cube[i] = w->cubeA[i]; o->cube0[i] = w->cubeA[i];
    }
    for (int i = 0; i < 4; ++i) {
        o->left_q[i] = L.quat[9][i]; o->cube_q[i] = w->cubeA[3 + i]; o->cube_q_half0[i] = w->cubeA[3 + i];
    }
    o->f_table[0] = w->f_table[0]; o->f_table[1] = w->f_table[1];
    o->f_shelf[0] = w->f_shelf[0]; o->f_shelf[1] = w->f_shelf[1];
    o->f_cubeB[0] = w->f_cubeB[0]; o->f_cubeB[1] = w->f_cubeB[1];
}

/* rollout loop for the panda env: mppi.py:275-332 with dynamics = reactive_tamp.py:63-70 */
void m3o_panda_rollout(const m3o_cfg* cfg, const m3o_panda_scene* sc, const m3o_panda_world* w0,
                       const float* act, int k0, int k1, float* states, float* actions,
                       float* cost_h, float* J) {
    const int K = cfg->K, T = cfg->T, nu = 9;
    /* quirk Q8 (cost_functions.py:97, skill_utils.py:274): the reach cost reads environment 0's cube position and, in the
       tilted mode, the orientation of the first environment of the second half -- quantities of THOSE rollouts' simulations
       (a gripper can move its cube).  With all K samples in the call they are simulated first; a partial range (a rank of a
       sharded command, which does not hold sample 0's noise) uses each sample's own cube, as the device does. */
    const int env0 = (cfg->task == M3O_TASK_REACH && k0 == 0 && k1 == K && K >= 2);
    float (*cube0)[3] = NULL, (*q_first)[4] = NULL, (*q_second)[4] = NULL;
    if (env0) {
        cube0 = malloc(sizeof(float[3]) * T); q_first = malloc(sizeof(float[4]) * T); q_second = malloc(sizeof(float[4]) * T);
        for (int pass = 0; pass < (cfg->multi_modal ? 2 : 1); ++pass) {
            const int k = pass ? K / 2 : 0;
            m3o_panda_world w = *w0;
            m3o_panda_infer_held(sc, &w);
            for (int t = 0; t < T; ++t) {
                float u[9];
                for (int d = 0; d < nu; ++d) {
                    u[d] = cfg->u_scale * act[((size_t)k * T + t) * nu + d];
                    if (cfg->sample_null_action && k == K - 1) u[d] = 0.0f;
                }
                m3o_panda_step(sc, &w, u);
                if (!pass) for (int d = 0; d < 3; ++d) cube0[t][d] = w.cubeA[d];
                for (int d = 0; d < 4; ++d) (pass ? q_second : q_first)[t][d] = w.cubeA[3 + d];
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int k = k0; k < k1; ++k) {
        const int i = k - k0;
        m3o_panda_world w = *w0;
        m3o_panda_infer_held(sc, &w);
        float j = 0.0f, g = 1.0f;
        for (int t = 0; t < T; ++t) {
            float u[9];
            for (int d = 0; d < nu; ++d) {
                u[d] = cfg->u_scale * act[((size_t)i * T + t) * nu + d];
                if (cfg->sample_null_action && k == K - 1) u[d] = 0.0f;
            }
            m3o_panda_step(sc, &w, u);
            float* st = &states[((size_t)i * T + t) * 4];
            st[0] = w.q[0]; st[1] = w.qd[0]; st[2] = w.q[1]; st[3] = w.qd[1]; /* dof 0,1 */
            m3o_panda_obs o;
            m3o_panda_observe(sc, &w, &o);
            if (env0) {
                const float* qh = (cfg->multi_modal && k >= K / 2) ? q_second[t] : q_first[t];
                for (int d = 0; d < 3; ++d) o.cube0[d] = cube0[t][d];
                for (int d = 0; d < 4; ++d) o.cube_q_half0[d] = qh[d];
            }
            float c = m3o_panda_cost_obs(cfg, &o, k);
            cost_h[(size_t)i * T + t] = c;
            for (int d = 0; d < nu; ++d) actions[((size_t)i * T + t) * nu + d] = u[d];   /* mppi.py:313 (scaled, as the update sees it) */
            j = j + g * c;
            g = g * cfg->gamma;
        }
        J[i] = j;
    }
    free(cube0); free(q_first); free(q_second);
}

void m3o_panda_step_batch(const m3o_panda_scene* sc, float* worlds, int n, const float* u) {
    const int W = (int)(sizeof(m3o_panda_world) / sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        m3o_panda_step(sc, (m3o_panda_world*)(worlds + (size_t)i * W), u + (size_t)i * 9);
}

void m3o_panda_cost_obs_batch(const m3o_cfg* cfg, const m3o_panda_obs* obs, int n, int k0, float* c) {
    for (int i = 0; i < n; ++i) c[i] = m3o_panda_cost_obs(cfg, &obs[i], k0 + i);
}
