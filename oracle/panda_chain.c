/*
 * panda_chain.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * (1) Restatement of the reference's panda_env task costs -- PINNED by golden group G6b:
 *       get_panda_reach_cost   cost_functions.py:91-114
 *       get_panda_pick_cost    cost_functions.py:116-125
 *       get_panda_place_cost   cost_functions.py:127-136
 *       get_pick_tilt_cost     cost_functions.py:138-156
 *       get_motion_cost        cost_functions.py:158-169 (panda branch)
 * (2) Independent implementation of "Panda chain spec v1" (DESIGN.md): velocity-servoed
 *     9-dof chain, forward kinematics from the URDF constants
 *     (assets/urdf/franka_description/robots/franka_panda.urdf:27-242), cubeA as a free body
 *     with support contact and a position-level grasp model, penalty contact forces.  It
 *     stands where the reference calls Isaac Gym / PhysX (isaacgym_wrapper.py:354-360);
 *     PARITY UNPINNED against PhysX.  Scene constants: config/panda_env/ yaml files.
 */
#include <math.h>
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
    const float inertia[9] = {1.0f, 1.0f, 0.5f, 0.5f, 0.1f, 0.1f, 0.05f, 0.1f, 0.1f};
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
}

void m3o_panda_world_init(m3o_panda_world* w, int cube_on_shelf) {
    memset(w, 0, sizeof(*w));
    const float q0[9] = {0, 0, 0, -2.0f, 0, 1.8675f, 0, 0.02f, 0.02f}; /* panda.yaml:10 */
    for (int i = 0; i < 9; ++i) w->q[i] = q0[i];
    if (cube_on_shelf) { w->cubeA[0] = 0.425f; w->cubeA[1] = 0.0f; w->cubeA[2] = 1.35f; }
    else { w->cubeA[0] = 0.2f; w->cubeA[1] = -0.2f; w->cubeA[2] = 1.06f; }
    w->cubeA[6] = 1.0f;
    w->cubeB[0] = 0.2f; w->cubeB[1] = 0.2f; w->cubeB[2] = 1.06f; w->cubeB[6] = 1.0f;
    w->rel_q[3] = 1.0f;
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

/* sphere (centre c, radius r) against an axis-aligned box (centre/half in b[6]): penalty force
 * ON THE BOX accumulated into f[2] (xy only: get_motion_cost reads [:, :2]) */
static void sphere_box_force(const m3o_panda_scene* sc, const float c[3], float r, const float b[6],
                             float f[2]) {
    float d[3], n2 = 0.0f;
    int inside = 1;
    for (int i = 0; i < 3; ++i) {
        float l = c[i] - b[i];
        float cl = fminf(fmaxf(l, -b[3 + i]), b[3 + i]);
        d[i] = l - cl;
        if (d[i] != 0.0f) inside = 0;
        n2 = mad(d[i], d[i], n2);
    }
    if (inside) return; /* centre inside the box: no direction; ignored by the spec */
    float dist = sqrtf(n2);
    float pen = r - dist;
    if (!(pen > 0.0f)) return;
    float k = sc->k_contact * pen / dist;
    f[0] = mad(-k, d[0], f[0]);
    f[1] = mad(-k, d[1], f[1]);
}

void m3o_panda_step(const m3o_panda_scene* sc, m3o_panda_world* w, const float u[9]) {
    const float h = sc->dt / (float)sc->substeps;
    for (int sub = 0; sub < sc->substeps; ++sub) {
        /* 1. velocity servo per dof (implicit damper, torque + velocity + position limits) */
        for (int i = 0; i < 9; ++i) {
            if (w->held != 0.0f && i >= 7) { w->qd[i] = 0.0f; continue; } /* fingers locked on the cube */
            /* spec: per-dof constants a = hD/I, rden = 1/(1+a), dv = h*effort/I (f32) */
            float a = (h * sc->drive_damping) / sc->inertia[i];
            float rden = 1.0f / (1.0f + a);
            float dv = (h * sc->effort[i]) / sc->inertia[i];
            float qd1 = mad(a, u[i], w->qd[i]) * rden;
            float tau = sc->drive_damping * (u[i] - qd1);
            if (tau > sc->effort[i]) qd1 = w->qd[i] + dv;
            if (tau < -sc->effort[i]) qd1 = w->qd[i] - dv;
            qd1 = fminf(fmaxf(qd1, -sc->vlim[i]), sc->vlim[i]);
            float q1 = mad(h, qd1, w->q[i]);
            if (q1 < sc->qlo[i]) { q1 = sc->qlo[i]; qd1 = 0.0f; }
            if (q1 > sc->qhi[i]) { q1 = sc->qhi[i]; qd1 = 0.0f; }
            w->q[i] = q1; w->qd[i] = qd1;
        }
        /* 2. kinematics */
        m3o_panda_links L;
        m3o_panda_fk(sc, w->q, &L);
        const float* ph = L.pos[8];
        const float *hx = L.ax[8], *hy = L.ay[8], *hz = L.az[8];
        float ft[2] = {0, 0}, fs[2] = {0, 0}, fb[2] = {0, 0};
        float cubeB_box[6] = {w->cubeB[0], w->cubeB[1], w->cubeB[2], sc->cube_half, sc->cube_half, sc->cube_half};

        /* 3. cubeA */
        if (w->held != 0.0f && (u[7] >= 0.0f || u[8] >= 0.0f)) { /* release */
            w->held = 0.0f;
            for (int i = 7; i < 13; ++i) w->cubeA[i] = 0.0f;
        }
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
            /* free body: gravity, support planes, Coulomb friction on the support */
            w->cubeA[9] = mad(-sc->g, h, w->cubeA[9]);
            for (int i = 0; i < 3; ++i) w->cubeA[i] = mad(h, w->cubeA[7 + i], w->cubeA[i]);
            const float x = w->cubeA[0], y = w->cubeA[1];
            float sup = -1.0e30f;
            int which = 0; /* 1 table, 2 shelf, 3 cubeB */
            if (fabsf(x - sc->table[0]) <= sc->table[3] && fabsf(y - sc->table[1]) <= sc->table[4]) {
                sup = sc->table[2] + sc->table[5]; which = 1;
            }
            if (fabsf(x - sc->shelf[0]) <= sc->shelf[3] && fabsf(y - sc->shelf[1]) <= sc->shelf[4]) {
                float t = sc->shelf[2] + sc->shelf[5];
                if (t > sup) { sup = t; which = 2; }
            }
            if (fabsf(x - w->cubeB[0]) <= sc->cube_half && fabsf(y - w->cubeB[1]) <= sc->cube_half) {
                float t = w->cubeB[2] + sc->cube_half;
                if (t > sup) { sup = t; which = 3; }
            }
            if (which != 0 && w->cubeA[2] - sc->cube_half < sup) {
                w->cubeA[2] = sup + sc->cube_half;
                if (w->cubeA[9] < 0.0f) w->cubeA[9] = 0.0f;
                float vx = w->cubeA[7], vy = w->cubeA[8];
                float sp = sqrtf(vx * vx + vy * vy);
                if (sp > 0.0f) {
                    float dec = (sc->cube_mu * sc->g) * h;
                    float nvx, nvy;
                    if (sp <= dec) { nvx = 0.0f; nvy = 0.0f; }
                    else { float sc_ = 1.0f - dec / sp; nvx = vx * sc_; nvy = vy * sc_; }
                    float fx = sc->cube_m * (vx - nvx) / h, fy = sc->cube_m * (vy - nvy) / h;
                    float* dst = (which == 1) ? ft : (which == 2) ? fs : fb;
                    dst[0] = dst[0] + fx; dst[1] = dst[1] + fy;
                    w->cubeA[7] = nvx; w->cubeA[8] = nvy;
                }
            }
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
            if (in_region) {
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
                if (gap <= wdt + sc->grasp_tol && u[7] < 0.0f && u[8] < 0.0f) {
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
        /* 4. penalty contact forces on table / shelf_stand / cubeB (robot spheres + held cube) */
        {
            float tipl[3], tipr[3], hc[3];
            for (int i = 0; i < 3; ++i) {
                tipl[i] = mad(sc->tip_z, hz[i], L.pos[9][i]);
                tipr[i] = mad(sc->tip_z, hz[i], L.pos[10][i]);
                hc[i] = mad(sc->hand_z, hz[i], ph[i]);
            }
            const float* boxes[3] = {sc->table, sc->shelf, cubeB_box};
            float* fo[3] = {ft, fs, fb};
            for (int b = 0; b < 3; ++b) {
                sphere_box_force(sc, tipl, sc->tip_r, boxes[b], fo[b]);
                sphere_box_force(sc, tipr, sc->tip_r, boxes[b], fo[b]);
                sphere_box_force(sc, hc, sc->hand_r, boxes[b], fo[b]);
                if (w->held != 0.0f) sphere_box_force(sc, w->cubeA, sc->cube_half, boxes[b], fo[b]);
            }
        }
        w->f_table[0] = ft[0]; w->f_table[1] = ft[1];
        w->f_shelf[0] = fs[0]; w->f_shelf[1] = fs[1];
        w->f_cubeB[0] = fb[0]; w->f_cubeB[1] = fb[1];
    }
}

/* The wrapper's tensors carry no "held" bit: when a world is loaded from them (rollout start,
 * set_*_state_tensor) it is inferred from geometry -- cube inside the grasp region, aligned
 * with the pads and the pads closed on it. */
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
            float c = m3o_panda_cost_obs(cfg, &o, k);
            cost_h[(size_t)i * T + t] = c;
            for (int d = 0; d < nu; ++d) actions[((size_t)i * T + t) * nu + d] = u[d];   /* mppi.py:313 (scaled, as the update sees it) */
            j = j + g * c;
            g = g * cfg->gamma;
        }
        J[i] = j;
    }
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
