/*
 * planar_world.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Independent scalar implementation of the "Planar contact dynamics spec" (currently v1.4).
 * It stands where the reference calls Isaac Gym / PhysX (closed binary):
 *   IsaacGymWrapper.step()                      isaacgym_wrapper.py:354-360
 *   set_dof_velocity_target_tensor()            isaacgym_wrapper.py:196
 *   apply_rigid_body_force_tensors()            isaacgym_wrapper.py:202-203
 * Scene constants: config/point_env/ yaml files, assets/urdf/pointRobot.urdf,
 * solver constants isaacgym_wrapper.py:18-37, drive isaacgym_wrapper.py:341-344.
 * PARITY UNPINNED against PhysX (no reference fixture exists at this boundary).
 *
 * Written as a generic 4-body (robot, box, dyn-obs, static) sequential-impulse solver with
 * dynamic contact lists -- deliberately a different code shape from the static-slot HIP
 * kernel, while following the spec's expression order so both agree bit-for-bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "m3_oracle.h"
#include <stdint.h>
#include <string.h>

/* spec v1.3: reciprocal square root = bit-trick seed + three Newton steps, all in binary32 and
 * in exactly this order (relative error < 1e-9 before the final rounding).  It replaces
 * "sqrtf then divide" where only 1/sqrt or a normalisation is needed: on the GPU the correctly
 * rounded sqrtf + division are ~32 dependent instructions, this is 15. */
/* spec v1.4: mad(a, b, c) = a*b + c with ONE rounding (IEEE 754 fusedMultiplyAdd; fmaf is exact with or without
 * hardware FMA) -- every product-plus-sum of the dynamics is written with it, here and in the HIP kernel alike. */
static inline float mad(float a, float b, float c) { return fmaf(a, b, c); }

/* spec v1.6: a substep's reciprocals -- minimax bit-trick seed, three Newton steps in residual form (x positive, normal) */
static inline float spec_rcp(float x) {
    uint32_t i;
    float y, r;
    memcpy(&i, &x, 4);
    i = 0x7EF311C7u - i;
    memcpy(&y, &i, 4);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    return y;
}

float m3o_spec_rcp(float x) { return spec_rcp(x); }

static inline float spec_rsqrt(float a) {
    uint32_t i;
    float y;
    memcpy(&i, &a, 4);
    i = 0x5f3759dfu - (i >> 1);
    memcpy(&y, &i, 4);
    const float hlf = 0.5f * a;
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    return y;
}


enum { BR = 0, BB = 1, BD = 2, BS = 3, NBODY = 4 };

typedef struct {
    int a, b;
    float nx, ny;   /* unit normal from a to b */
    float rax, ray; /* arm from a's centre to the contact point */
    float rbx, rby;
    float sep;
    float mu;
    /* prepared */
    float rna, rnb, rta, rtb, mn, mt, bias;
    float ln, lt;
} contact_t;

#define MAXC 19

typedef struct {
    float invm[NBODY], invI[NBODY];
    float vx[NBODY], vy[NBODY], w[NBODY];
    contact_t c[MAXC];
    int nc;
} solver_t;

void m3o_point_scene_default(m3o_point_scene* sc) {
    sc->dt = 0.05f;
    sc->substeps = 2;
    sc->iters = 6;
    sc->g = 9.8f;
    sc->robot_r = 0.2f;
    sc->robot_m = 10.0f;
    sc->drive_damping = 600.0f;
    sc->drive_fmax = 1000.0f;
    /* boxes 0.4 x 0.4 x 0.1 at PhysX default density 1000 kg/m^3 (yaml mass is not
     * applied: isaacgym_wrapper.py:293-300) */
    sc->box_hx = 0.2f; sc->box_hy = 0.2f; sc->box_m = 16.0f;
    sc->box_I = 16.0f * (0.4f * 0.4f + 0.4f * 0.4f) / 12.0f;
    sc->box_mu_g = 0.75f;                /* average(box 0.5, ground 1.0) */
    sc->box_req = 0.3825978f * 0.4f;     /* mean lever arm of a square patch */
    sc->dyn_hx = 0.2f; sc->dyn_hy = 0.2f; sc->dyn_m = 16.0f;
    sc->dyn_I = 16.0f * (0.4f * 0.4f + 0.4f * 0.4f) / 12.0f;
    sc->dyn_mu_g = 1.0f;                 /* average(1.0, 1.0) */
    sc->dyn_req = 0.3825978f * 0.4f;
    sc->obs_x = 2.0f; sc->obs_y = 2.0f; sc->obs_hx = 0.15f; sc->obs_hy = 0.2f;
    sc->wall = 3.95f;
    sc->mu_rb = 0.275f;  /* average(0.05, 0.5) */
    sc->mu_rd = 0.525f;  /* average(0.05, 1.0) */
    sc->mu_ro = 0.525f;
    sc->mu_rw = 0.525f;
    sc->mu_bw = 0.75f;
    sc->mu_dw = 1.0f;
    sc->mu_bd = 0.75f;
    sc->mu_bo = 0.75f;
    sc->mu_do = 1.0f;
    sc->contact_offset = 0.01f;
    sc->baumgarte = 0.2f;
    sc->slop = 0.005f;
    sc->max_bias = 2.0f;
    sc->face_tol = 0.0005f;
    sc->friction_coupling = 1;           /* spec v1.5 */
    sc->fext_substeps = 1;               /* spec v1.7: a pending external force is consumed by the first substep */
}

void m3o_point_world_init(m3o_point_world* w) {
    memset(w, 0, sizeof(*w));
    w->R.c = 1.0f;
    w->B.x = 0.0f; w->B.y = 2.0f; w->B.c = 1.0f;   /* 7_box.yaml */
    w->D.x = -2.0f; w->D.y = 2.0f; w->D.c = 1.0f;  /* 6_dyn_obs.yaml */
}

static void add_contact(solver_t* s, int a, int b, float nx, float ny, float rax, float ray,
                        float rbx, float rby, float sep, float mu) {
    contact_t* c = &s->c[s->nc++];
    c->a = a; c->b = b; c->nx = nx; c->ny = ny;
    c->rax = rax; c->ray = ray; c->rbx = rbx; c->rby = rby;
    c->sep = sep; c->mu = mu; c->ln = 0.0f; c->lt = 0.0f;
}

/* disc A (robot) against box B (body id bid, possibly static) */
static void detect_disc_box(const m3o_point_scene* sc, solver_t* s, const m3o_body* R,
                            int bid, float qx, float qy, float c, float sn, float hx,
                            float hy, float mu) {
    const float r = sc->robot_r;
    float dx = R->x - qx, dy = R->y - qy;
    float lx = mad(c, dx, sn * dy);
    float ly = mad(c, dy, -(sn * dx));
    float cx = fminf(fmaxf(lx, -hx), hx);
    float cy = fminf(fmaxf(ly, -hy), hy);
    float ex = lx - cx, ey = ly - cy;
    float d2 = mad(ex, ex, ey * ey);
    float nlx, nly, sep;
    if (d2 > 0.0f) {
        float rd = spec_rsqrt(d2); /* spec v1.3 */
        float d = d2 * rd;
        nlx = ex * rd; nly = ey * rd;
        sep = d - r;
    } else {
        float px = hx - fabsf(lx), py = hy - fabsf(ly);
        if (px < py) {
            float sg = (lx >= 0.0f) ? 1.0f : -1.0f;
            nlx = sg; nly = 0.0f; cx = sg * hx; cy = ly; sep = -px - r;
        } else {
            float sg = (ly >= 0.0f) ? 1.0f : -1.0f;
            nlx = 0.0f; nly = sg; cx = lx; cy = sg * hy; sep = -py - r;
        }
    }
    if (!(sep < sc->contact_offset)) return;
    float wx = mad(c, nlx, -(sn * nly));
    float wy = mad(sn, nlx, c * nly);
    float rbx = mad(c, cx, -(sn * cy));
    float rby = mad(sn, cx, c * cy);
    add_contact(s, BR, bid, -wx, -wy, 0.0f, 0.0f, rbx, rby, sep, mu);
}

static void detect_disc_walls(const m3o_point_scene* sc, solver_t* s, const m3o_body* R) {
    float sg = (R->x >= 0.0f) ? 1.0f : -1.0f;
    float sep = mad(-sg, R->x, sc->wall) - sc->robot_r;
    if (sep < sc->contact_offset)
        add_contact(s, BR, BS, sg, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, sep, sc->mu_rw);
    sg = (R->y >= 0.0f) ? 1.0f : -1.0f;
    sep = mad(-sg, R->y, sc->wall) - sc->robot_r;
    if (sep < sc->contact_offset)
        add_contact(s, BR, BS, 0.0f, sg, 0.0f, 0.0f, 0.0f, 0.0f, sep, sc->mu_rw);
}

static void detect_box_walls(const m3o_point_scene* sc, solver_t* s, int bid,
                             const m3o_body* X, float hx, float hy, float mu) {
    float r0x = mad(X->c, hx, -(X->s * hy)), r0y = mad(X->s, hx, X->c * hy);   /* corner (+,+) */
    float r1x = mad(-X->c, hx, -(X->s * hy)), r1y = mad(-X->s, hx, X->c * hy); /* corner (-,+) */
    /* x walls */
    {
        float sg = (X->x >= 0.0f) ? 1.0f : -1.0f;
        float base = mad(-sg, X->x, sc->wall);
        float pa = sg * r0x, pb = sg * r1x;
        float sep1 = base - fabsf(pa), sep2 = base - fabsf(pb);
        if (sep1 < sc->contact_offset) {
            float f = (pa >= 0.0f) ? 1.0f : -1.0f;
            add_contact(s, bid, BS, sg, 0.0f, f * r0x, f * r0y, 0.0f, 0.0f, sep1, mu);
        }
        if (sep2 < sc->contact_offset) {
            float f = (pb >= 0.0f) ? 1.0f : -1.0f;
            add_contact(s, bid, BS, sg, 0.0f, f * r1x, f * r1y, 0.0f, 0.0f, sep2, mu);
        }
    }
    /* y walls */
    {
        float sg = (X->y >= 0.0f) ? 1.0f : -1.0f;
        float base = mad(-sg, X->y, sc->wall);
        float pa = sg * r0y, pb = sg * r1y;
        float sep1 = base - fabsf(pa), sep2 = base - fabsf(pb);
        if (sep1 < sc->contact_offset) {
            float f = (pa >= 0.0f) ? 1.0f : -1.0f;
            add_contact(s, bid, BS, 0.0f, sg, f * r0x, f * r0y, 0.0f, 0.0f, sep1, mu);
        }
        if (sep2 < sc->contact_offset) {
            float f = (pb >= 0.0f) ? 1.0f : -1.0f;
            add_contact(s, bid, BS, 0.0f, sg, f * r1x, f * r1y, 0.0f, 0.0f, sep2, mu);
        }
    }
}

/* box A (id ia) against box B (id ib): SAT, reference face + 2 clipped incident corners */
static void detect_box_box(const m3o_point_scene* sc, solver_t* s, int ia, float ax, float ay,
                           float ca, float sa, float hax, float hay, int ib, float bx,
                           float by, float cb, float sb, float hbx, float hby, float mu) {
    float dxw = bx - ax, dyw = by - ay;
    float dx = mad(ca, dxw, sa * dyw);
    float dy = mad(ca, dyw, -(sa * dxw));
    float cr = mad(ca, cb, sa * sb);
    float sr = mad(ca, sb, -(sa * cb));
    float acr = fabsf(cr), asr = fabsf(sr);
    float sAx = fabsf(dx) - (hax + mad(acr, hbx, asr * hby));
    float sAy = fabsf(dy) - (hay + mad(asr, hbx, acr * hby));
    float ex = -mad(cr, dx, sr * dy);
    float ey = -mad(cr, dy, -(sr * dx));
    float sBx = fabsf(ex) - (hbx + mad(acr, hax, asr * hay));
    float sBy = fabsf(ey) - (hby + mad(asr, hax, acr * hay));
    float best = sAx;
    int axis = 0;
    if (sAy > best + sc->face_tol) { best = sAy; axis = 1; }
    if (sBx > best + sc->face_tol) { best = sBx; axis = 2; }
    if (sBy > best + sc->face_tol) { best = sBy; axis = 3; }
    if (!(best < sc->contact_offset)) return;

    /* reference frame quantities */
    int refA = axis < 2;
    float drx = refA ? dx : ex, dry = refA ? dy : ey;  /* incident centre in ref frame */
    float crr = cr, srr = refA ? sr : -sr;             /* incident axes in ref frame */
    float hrx = refA ? hax : hbx, hry = refA ? hay : hby;
    float hix = refA ? hbx : hax, hiy = refA ? hby : hay;
    int xface = (axis & 1) == 0;
    float dn = xface ? drx : dry;
    float sg = (dn >= 0.0f) ? 1.0f : -1.0f;
    float hn = xface ? hrx : hry;
    float ht = xface ? hry : hrx;
    /* incident corners (+,+) and (-,+) in the ref frame */
    float r0x = mad(crr, hix, -(srr * hiy)), r0y = mad(srr, hix, crr * hiy);
    float r1x = mad(-crr, hix, -(srr * hiy)), r1y = mad(-srr, hix, crr * hiy);
    float pa = sg * (xface ? r0x : r0y);
    float pb = sg * (xface ? r1x : r1y);
    float f0 = (pa <= 0.0f) ? 1.0f : -1.0f;
    float f1 = (pb <= 0.0f) ? 1.0f : -1.0f;
    float p1x = mad(f0, r0x, drx), p1y = mad(f0, r0y, dry);
    float p2x = mad(f1, r1x, drx), p2y = mad(f1, r1y, dry);
    float s1 = mad(sg, xface ? p1x : p1y, -hn);
    float s2 = mad(sg, xface ? p2x : p2y, -hn);
    float t1 = xface ? p1y : p1x;
    float t2 = xface ? p2y : p2x;
    /* clip the incident edge against the side planes |t| <= ht (using the unclipped line) */
    float cs1 = s1, ct1 = t1, cs2 = s2, ct2 = t2;
    int ok = 1;
    if (t1 > ht) {
        if (t2 > ht) ok = 0;
        else { float lam = (ht - t2) / (t1 - t2); cs1 = mad(lam, s1 - s2, s2); ct1 = ht; }
    } else if (t1 < -ht) {
        if (t2 < -ht) ok = 0;
        else { float lam = (-ht - t2) / (t1 - t2); cs1 = mad(lam, s1 - s2, s2); ct1 = -ht; }
    }
    if (t2 > ht) {
        if (!(t1 > ht)) { float lam = (ht - t1) / (t2 - t1); cs2 = mad(lam, s2 - s1, s1); ct2 = ht; }
    } else if (t2 < -ht) {
        if (!(t1 < -ht)) { float lam = (-ht - t1) / (t2 - t1); cs2 = mad(lam, s2 - s1, s1); ct2 = -ht; }
    }
    if (!ok) return;
    /* reference box world frame */
    float qrx = refA ? ax : bx, qry = refA ? ay : by;
    float rc = refA ? ca : cb, rs = refA ? sa : sb;
    /* face normal (ref frame) -> world; n must point from A to B */
    float nrx = xface ? sg : 0.0f, nry = xface ? 0.0f : sg;
    float nwx = mad(rc, nrx, -(rs * nry)), nwy = mad(rs, nrx, rc * nry);
    if (!refA) { nwx = -nwx; nwy = -nwy; }
    float cs[2] = {cs1, cs2}, ct[2] = {ct1, ct2};
    for (int i = 0; i < 2; ++i) {
        if (!(cs[i] < sc->contact_offset)) continue;
        float pn = sg * (hn + cs[i]);
        float plx = xface ? pn : ct[i];
        float ply = xface ? ct[i] : pn;
        float pwx = qrx + mad(rc, plx, -(rs * ply));
        float pwy = qry + mad(rs, plx, rc * ply);
        add_contact(s, ia, ib, nwx, nwy, pwx - ax, pwy - ay, pwx - bx, pwy - by, cs[i], mu);
    }
}

static void prepare_contacts(const m3o_point_scene* sc, solver_t* s, float h) {
    const float inv_h = 1.0f / h; /* spec v1.2 */
    for (int i = 0; i < s->nc; ++i) {
        contact_t* c = &s->c[i];
        float tx = -c->ny, ty = c->nx;
        c->rna = mad(c->rax, c->ny, -(c->ray * c->nx));
        c->rnb = mad(c->rbx, c->ny, -(c->rby * c->nx));
        c->rta = mad(c->rax, ty, -(c->ray * tx));
        c->rtb = mad(c->rbx, ty, -(c->rby * tx));
        float kn = mad(s->invI[c->b] * c->rnb, c->rnb,
                       mad(s->invI[c->a] * c->rna, c->rna, s->invm[c->a] + s->invm[c->b]));
        float kt = mad(s->invI[c->b] * c->rtb, c->rtb,
                       mad(s->invI[c->a] * c->rta, c->rta, s->invm[c->a] + s->invm[c->b]));
        c->mn = spec_rcp(kn);
        c->mt = spec_rcp(kt);
        if (c->sep > 0.0f) {
            c->bias = c->sep * inv_h;
        } else {
            float pen = -c->sep - sc->slop;
            if (pen < 0.0f) pen = 0.0f;
            float push = (sc->baumgarte * pen) * inv_h;
            if (push > sc->max_bias) push = sc->max_bias;
            c->bias = -push;
        }
    }
}

static void solve_contact(solver_t* s, contact_t* c) {
    const int a = c->a, b = c->b;
    const float tx = -c->ny, ty = c->nx;
    /* normal */
    float dvx = s->vx[b] - s->vx[a], dvy = s->vy[b] - s->vy[a];
    float vn = mad(-s->w[a], c->rna, mad(s->w[b], c->rnb, mad(dvx, c->nx, dvy * c->ny)));
    float dl = -c->mn * (vn + c->bias);
    float l0 = c->ln;
    float l1 = l0 + dl;
    if (l1 < 0.0f) l1 = 0.0f;
    c->ln = l1;
    dl = l1 - l0;
    s->vx[a] = mad(-(s->invm[a] * dl), c->nx, s->vx[a]);
    s->vy[a] = mad(-(s->invm[a] * dl), c->ny, s->vy[a]);
    s->w[a] = mad(-(s->invI[a] * c->rna), dl, s->w[a]);
    s->vx[b] = mad(s->invm[b] * dl, c->nx, s->vx[b]);
    s->vy[b] = mad(s->invm[b] * dl, c->ny, s->vy[b]);
    s->w[b] = mad(s->invI[b] * c->rnb, dl, s->w[b]);
    /* friction */
    dvx = s->vx[b] - s->vx[a]; dvy = s->vy[b] - s->vy[a];
    float vt = mad(-s->w[a], c->rta, mad(s->w[b], c->rtb, mad(dvx, tx, dvy * ty)));
    dl = -c->mt * vt;
    float maxf = c->mu * c->ln;
    l0 = c->lt;
    l1 = l0 + dl;
    if (l1 > maxf) l1 = maxf;
    if (l1 < -maxf) l1 = -maxf;
    c->lt = l1;
    dl = l1 - l0;
    s->vx[a] = mad(-(s->invm[a] * dl), tx, s->vx[a]);
    s->vy[a] = mad(-(s->invm[a] * dl), ty, s->vy[a]);
    s->w[a] = mad(-(s->invI[a] * c->rta), dl, s->w[a]);
    s->vx[b] = mad(s->invm[b] * dl, tx, s->vx[b]);
    s->vy[b] = mad(s->invm[b] * dl, ty, s->vy[b]);
    s->w[b] = mad(s->invI[b] * c->rtb, dl, s->w[b]);
}

/* spec v1.4: in the rest test and the orientation update "zero" means below the smallest normal binary32 number (a
 * residual spin decays through the subnormals and can stay there: 1 / I is a rounded reciprocal) */
static inline int is_zero(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u & 0x7f800000u) == 0u;
}

typedef struct { float lx, ly, la; } fric_acc;

static void solve_ground_friction(solver_t* s, int b, float m, float I, float Llin, float Lang,
                                  fric_acc* f) {
    /* spec: a body at rest (v = 0 and w = 0) has no friction row in this pass */
    if (is_zero(s->vx[b]) && is_zero(s->vy[b]) && is_zero(s->w[b])) return;
    float nlx = mad(-m, s->vx[b], f->lx);
    float nly = mad(-m, s->vy[b], f->ly);
    float mag2 = mad(nlx, nlx, nly * nly);
    if (mag2 > Llin * Llin) {
        float sc = Llin * spec_rsqrt(mag2); /* spec v1.3 */
        nlx = nlx * sc; nly = nly * sc;
    }
    s->vx[b] = mad(s->invm[b], nlx - f->lx, s->vx[b]);
    s->vy[b] = mad(s->invm[b], nly - f->ly, s->vy[b]);
    f->lx = nlx; f->ly = nly;
    float nla = mad(-I, s->w[b], f->la);
    if (nla > Lang) nla = Lang;
    if (nla < -Lang) nla = -Lang;
    s->w[b] = mad(s->invI[b], nla - f->la, s->w[b]);
    f->la = nla;
}

/* spec v1.5: sliding-spinning coupling of a box's ground friction.  A contact patch that slides fast has almost no
 * resistance left against turning (every point of it moves along the slide), and one that spins fast little against
 * sliding: Contensou's law, in Zhuravlev's first-order Pade form for a disc of radius R under uniform pressure --
 *     F = F0 v / (v + 8/(3 pi) u),   M = M0 u / (u + 15 pi / 16 v),   u = R |w|,   M0 = 2/3 mu N R  (so R = 1.5 r_eq).
 * cf[0] scales the linear row's limit, cf[1] the torsion row's; a box that does not slide keeps its full linear
 * limit, one that does not spin its full torsion limit.  "Zero" = below the smallest normal number, as everywhere. */
static void friction_coupling(float vx, float vy, float w, float R, float cf[2]) {
    const float s2 = mad(vx, vx, vy * vy);
    const float v = is_zero(s2) ? 0.0f : s2 * spec_rsqrt(s2);
    const float u = is_zero(w) ? 0.0f : R * fabsf(w);
    cf[0] = 1.0f; cf[1] = 1.0f;
    if (!is_zero(v)) cf[0] = v * spec_rcp(mad(0.8488264f, u, v));
    if (!is_zero(u)) cf[1] = u * spec_rcp(mad(2.9452431f, v, u));
}

/* EXPERIMENT (tools/cpu_ab_default_size.py, not part of the spec): ground friction at the four corners of the box's
 * contact patch -- each corner carries m g / 4 and its Coulomb impulse opposes that corner's own velocity, so a box
 * that SLIDES fast has almost no resistance left against turning (the corners' velocities all point along the slide),
 * which is what a contact patch does and what the independent torsion row of the spec (limit mu m g r_eq whatever the
 * sliding speed) does not.  Selected by the scene field friction_coupling = 2 (0: spec v1.4, 1: spec v1.5). */
typedef struct { float lx[4], ly[4]; } patch_acc;
static void solve_ground_friction_patch4(solver_t* s, int b, const m3o_body* X, float hx, float hy, float Lpt,
                                         patch_acc* f, fric_acc* tot) {
    if (is_zero(s->vx[b]) && is_zero(s->vy[b]) && is_zero(s->w[b])) return;
    const float im = s->invm[b], iI = s->invI[b];
    for (int k = 0; k < 4; ++k) {
        const float lx = (k & 1) ? hx : -hx, ly = (k & 2) ? hy : -hy;
        const float rx = X->c * lx - X->s * ly, ry = X->s * lx + X->c * ly;
        const float vx = s->vx[b] - s->w[b] * ry, vy = s->vy[b] + s->w[b] * rx;
        /* K = im I + iI [[ry^2, -rx ry], [-rx ry, rx^2]];  p = -K^-1 v */
        const float a = im + iI * ry * ry, bb = -iI * rx * ry, d = im + iI * rx * rx;
        const float det = a * d - bb * bb;
        float px = -(d * vx - bb * vy) / det, py = -(-bb * vx + a * vy) / det;
        float nx = f->lx[k] + px, ny = f->ly[k] + py;
        const float mag2 = nx * nx + ny * ny;
        if (mag2 > Lpt * Lpt) { const float sc = Lpt / sqrtf(mag2); nx *= sc; ny *= sc; }
        px = nx - f->lx[k]; py = ny - f->ly[k];
        f->lx[k] = nx; f->ly[k] = ny;
        s->vx[b] += im * px; s->vy[b] += im * py;
        s->w[b] += iI * (rx * py - ry * px);
        tot->lx += px; tot->ly += py;
    }
}

static void integrate_body(m3o_body* X, float h, int rotate) {
    X->x = mad(h, X->vx, X->x);
    X->y = mad(h, X->vy, X->y);
    if (rotate && !is_zero(X->w)) { /* spec: orientation is only touched when w != 0 */
        float a = 0.5f * (h * X->w);
        float a2 = a * a;
        float den = 1.0f + a2;
        float rden = spec_rcp(den);
        float cd = (1.0f - a2) * rden;
        float sd = (2.0f * a) * rden;
        float c = mad(X->c, cd, -(X->s * sd));
        float s = mad(X->s, cd, X->c * sd);
        float rn = spec_rsqrt(mad(c, c, s * s)); /* spec v1.3 */
        X->c = c * rn;
        X->s = s * rn;
    }
}

void m3o_point_step(const m3o_point_scene* sc, m3o_point_world* w, const float u[2]) {
    const float h = sc->dt / (float)sc->substeps;
    const float inv_h = 1.0f / h;
    solver_t s;
    s.invm[BR] = 1.0f / sc->robot_m; s.invI[BR] = 0.0f;
    s.invm[BB] = 1.0f / sc->box_m;   s.invI[BB] = 1.0f / sc->box_I;
    s.invm[BD] = 1.0f / sc->dyn_m;   s.invI[BD] = 1.0f / sc->dyn_I;
    s.invm[BS] = 0.0f; s.invI[BS] = 0.0f;
    const float gam = 1.0f / (h * sc->drive_damping);
    const float md = 1.0f / (s.invm[BR] + gam);
    const float dmax = sc->drive_fmax * h;
    const float LlinB = ((sc->box_mu_g * sc->box_m) * sc->g) * h;
    const float LangB = LlinB * sc->box_req;
    const float LlinD = ((sc->dyn_mu_g * sc->dyn_m) * sc->g) * h;
    const float LangD = LlinD * sc->dyn_req;

    for (int sub = 0; sub < sc->substeps; ++sub) {
        /* 1. external forces (suction).  Spec v1.7: a pending force is CONSUMED by the first substep of the step -- the
         * reading of a one-shot `apply_rigid_body_force_tensors` under 2 substeps that the joint fit against the
         * reference's eight logged scenarios selects (tools/cpu_fit_physx.py, profiles/r06/fit_physx_*.json; up to v1.6
         * it acted in every substep: fext_substeps = 0, kept for that tool).  The later substeps still evaluate the same
         * expression, on a force of zero. */
        w->R.vx = mad(h * w->fext_R[0], s.invm[BR], w->R.vx);
        w->R.vy = mad(h * w->fext_R[1], s.invm[BR], w->R.vy);
        w->B.vx = mad(h * w->fext_B[0], s.invm[BB], w->B.vx);
        w->B.vy = mad(h * w->fext_B[1], s.invm[BB], w->B.vy);
        if (sc->fext_substeps > 0 && sub + 1 >= sc->fext_substeps) {
            w->fext_R[0] = w->fext_R[1] = 0.0f;
            w->fext_B[0] = w->fext_B[1] = 0.0f;
        }

        /* 2. contacts, fixed slot order */
        s.nc = 0;
        detect_disc_box(sc, &s, &w->R, BB, w->B.x, w->B.y, w->B.c, w->B.s, sc->box_hx,
                        sc->box_hy, sc->mu_rb);
        detect_disc_box(sc, &s, &w->R, BD, w->D.x, w->D.y, w->D.c, w->D.s, sc->dyn_hx,
                        sc->dyn_hy, sc->mu_rd);
        detect_disc_box(sc, &s, &w->R, BS, sc->obs_x, sc->obs_y, 1.0f, 0.0f, sc->obs_hx,
                        sc->obs_hy, sc->mu_ro);
        detect_disc_walls(sc, &s, &w->R);
        detect_box_walls(sc, &s, BB, &w->B, sc->box_hx, sc->box_hy, sc->mu_bw);
        detect_box_walls(sc, &s, BD, &w->D, sc->dyn_hx, sc->dyn_hy, sc->mu_dw);
        detect_box_box(sc, &s, BB, w->B.x, w->B.y, w->B.c, w->B.s, sc->box_hx, sc->box_hy, BD,
                       w->D.x, w->D.y, w->D.c, w->D.s, sc->dyn_hx, sc->dyn_hy, sc->mu_bd);
        detect_box_box(sc, &s, BB, w->B.x, w->B.y, w->B.c, w->B.s, sc->box_hx, sc->box_hy, BS,
                       sc->obs_x, sc->obs_y, 1.0f, 0.0f, sc->obs_hx, sc->obs_hy, sc->mu_bo);
        detect_box_box(sc, &s, BD, w->D.x, w->D.y, w->D.c, w->D.s, sc->dyn_hx, sc->dyn_hy, BS,
                       sc->obs_x, sc->obs_y, 1.0f, 0.0f, sc->obs_hx, sc->obs_hy, sc->mu_do);
        prepare_contacts(sc, &s, h);

        /* 3. velocity solve */
        s.vx[BR] = w->R.vx; s.vy[BR] = w->R.vy; s.w[BR] = 0.0f;
        s.vx[BB] = w->B.vx; s.vy[BB] = w->B.vy; s.w[BB] = w->B.w;
        s.vx[BD] = w->D.vx; s.vy[BD] = w->D.vy; s.w[BD] = w->D.w;
        s.vx[BS] = 0.0f; s.vy[BS] = 0.0f; s.w[BS] = 0.0f;
        float ldx = 0.0f, ldy = 0.0f;
        fric_acc fB = {0.0f, 0.0f, 0.0f}, fD = {0.0f, 0.0f, 0.0f};
        patch_acc pB, pD;
        memset(&pB, 0, sizeof pB); memset(&pD, 0, sizeof pD);
        const int patch4 = (sc->friction_coupling == 2);   /* (experiment of tools/cpu_ab_default_size.py only: a scene field, not an environment variable) */
        /* spec v1.5: the limits of a box's two ground-friction rows are coupled by the sliding-spinning law of a contact
         * patch, factors from the velocities this substep starts its passes with */
        float cfB[2] = {1.0f, 1.0f}, cfD[2] = {1.0f, 1.0f};
        if (sc->friction_coupling == 1) {
            friction_coupling(s.vx[BB], s.vy[BB], s.w[BB], 1.5f * sc->box_req, cfB);
            friction_coupling(s.vx[BD], s.vy[BD], s.w[BD], 1.5f * sc->dyn_req, cfD);
        }
        const float LlinBe = LlinB * cfB[0], LangBe = LangB * cfB[1], LlinDe = LlinD * cfD[0], LangDe = LangD * cfD[1];
        for (int it = 0; it < sc->iters; ++it) {
            /* velocity drive (soft constraint, implicit damper) */
            {
                float dl = -(mad(gam, ldx, s.vx[BR] - u[0]) * md);
                float l1 = ldx + dl;
                if (l1 > dmax) l1 = dmax;
                if (l1 < -dmax) l1 = -dmax;
                s.vx[BR] = mad(s.invm[BR], l1 - ldx, s.vx[BR]);
                ldx = l1;
                dl = -(mad(gam, ldy, s.vy[BR] - u[1]) * md);
                l1 = ldy + dl;
                if (l1 > dmax) l1 = dmax;
                if (l1 < -dmax) l1 = -dmax;
                s.vy[BR] = mad(s.invm[BR], l1 - ldy, s.vy[BR]);
                ldy = l1;
            }
            if (patch4) {
                solve_ground_friction_patch4(&s, BB, &w->B, sc->box_hx, sc->box_hy, 0.25f * LlinB, &pB, &fB);
                solve_ground_friction_patch4(&s, BD, &w->D, sc->dyn_hx, sc->dyn_hy, 0.25f * LlinD, &pD, &fD);
            } else {
                solve_ground_friction(&s, BB, sc->box_m, sc->box_I, LlinBe, LangBe, &fB);
                solve_ground_friction(&s, BD, sc->dyn_m, sc->dyn_I, LlinDe, LangDe, &fD);
            }
            for (int i = 0; i < s.nc; ++i) solve_contact(&s, &s.c[i]);
        }
        w->R.vx = s.vx[BR]; w->R.vy = s.vy[BR];
        w->B.vx = s.vx[BB]; w->B.vy = s.vy[BB]; w->B.w = s.w[BB];
        w->D.vx = s.vx[BD]; w->D.vy = s.vy[BD]; w->D.w = s.w[BD];

        /* net contact force of this substep (contacts + ground friction) */
        float fx[NBODY] = {0, 0, 0, 0}, fy[NBODY] = {0, 0, 0, 0};
        for (int i = 0; i < s.nc; ++i) {
            const contact_t* c = &s.c[i];
            float ix = mad(c->ln, c->nx, c->lt * (-c->ny));
            float iy = mad(c->ln, c->ny, c->lt * c->nx);
            fx[c->a] -= ix; fy[c->a] -= iy;
            fx[c->b] += ix; fy[c->b] += iy;
        }
        fx[BB] += fB.lx; fy[BB] += fB.ly;
        fx[BD] += fD.lx; fy[BD] += fD.ly;
        w->fc_R[0] = fx[BR] * inv_h; w->fc_R[1] = fy[BR] * inv_h;
        w->fc_B[0] = fx[BB] * inv_h; w->fc_B[1] = fy[BB] * inv_h;
        w->fc_D[0] = fx[BD] * inv_h; w->fc_D[1] = fy[BD] * inv_h;

        /* 4. integrate */
        integrate_body(&w->R, h, 0);
        integrate_body(&w->B, h, 1);
        integrate_body(&w->D, h, 1);
    }
    w->fext_R[0] = w->fext_R[1] = 0.0f;
    w->fext_B[0] = w->fext_B[1] = 0.0f;
}
