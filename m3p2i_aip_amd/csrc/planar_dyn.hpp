// planar_dyn.hpp -- device-side "Planar contact dynamics spec" (currently v1.5).
//
// Replaces, for the point_env scene, what the reference delegates to Isaac Gym / PhysX:
//   IsaacGymWrapper.step()                    isaacgym_wrapper.py:354-360
//   set_dof_velocity_target_tensor()          isaacgym_wrapper.py:196
//   apply_rigid_body_force_tensors()          isaacgym_wrapper.py:202-203
// Scene constants: config/point_env/*.yaml, assets/urdf/pointRobot.urdf; solver constants
// isaacgym_wrapper.py:18-37; velocity drive isaacgym_wrapper.py:341-344.
//
// Mapping: ONE LANE PER SAMPLE, the whole world (robot disc + 2 boxes, 18 floats) and all 19
// contact slots live in VGPRs.  Slots are STATIC (slot i always means the same body pair), so
// no dynamic register indexing and every `if (slot.on)` is a plain exec-mask branch that
// costs nothing when no lane of the wave is in that contact.  Only + - * / min max, the explicit
// fused multiply-add mad() and compares are used, in a fixed expression order (compiled with
// -ffp-contract=off), so the result is bit-identical to the scalar CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "spec_fma.hpp"

namespace m3 {

// Only what depends on dt / substeps / iterations travels as a kernel argument (10 scalars in
// SGPRs).  The scene the reference fixes in its yaml/urdf files is compile-time constant: it
// folds into the instruction stream instead of occupying ~40 more SGPRs (the first version
// passed everything at run time and spilled 120+ SGPRs to VGPR lanes).
struct PointScene {
    float h;  // substep = dt / substeps
    float inv_h;  // 1 / h (spec v1.2: divisions by h, d, den, nrm are one reciprocal + multiplies)
    int substeps;
    int iters;
    float gam, md, dmax;  // velocity drive: 1/(h*D), 1/(invm_r+gam), fmax*h
    float LlinB, LangB, LlinD, LangD;  // ground-friction impulse limits (mu m g h, * r_eq)
    float RcB, RcD;                    // spec v1.5: radius of the equivalent disc of the friction coupling, 1.5 r_eq
    // ---- constants of the scene (config/point_env/*.yaml, pointRobot.urdf) ----
    static constexpr float robot_r = 0.2f, invm_r = 1.0f / 10.0f;
    static constexpr float box_hx = 0.2f, box_hy = 0.2f, box_m = 16.0f;
    static constexpr float box_I = 16.0f * (0.4f * 0.4f + 0.4f * 0.4f) / 12.0f;
    static constexpr float invm_b = 1.0f / box_m, invI_b = 1.0f / box_I;
    static constexpr float dyn_hx = 0.2f, dyn_hy = 0.2f, dyn_m = 16.0f;
    static constexpr float dyn_I = 16.0f * (0.4f * 0.4f + 0.4f * 0.4f) / 12.0f;
    static constexpr float invm_d = 1.0f / dyn_m, invI_d = 1.0f / dyn_I;
    static constexpr float obs_x = 2.0f, obs_y = 2.0f, obs_hx = 0.15f, obs_hy = 0.2f;
    static constexpr float wall = 3.95f;
    static constexpr float mu_rb = 0.275f, mu_rd = 0.525f, mu_ro = 0.525f, mu_rw = 0.525f;
    static constexpr float mu_bw = 0.75f, mu_dw = 1.0f, mu_bd = 0.75f, mu_bo = 0.75f, mu_do = 1.0f;
    static constexpr float contact_offset = 0.01f, baumgarte = 0.2f, slop = 0.005f, max_bias = 2.0f;
    static constexpr float face_tol = 0.0005f;
    // bounding radii for the (conservative) broad phase: >= sqrt(hx^2 + hy^2)
    static constexpr float rad_b = 0.28285f, rad_d = 0.28285f, rad_o = 0.25f;
};

// the run-time part of the scene from the reference's solver settings (host side: m3_create, and the host build of this
// header in tests/native/): every product in binary32, in this order -- the oracle forms the same constants
constexpr PointScene point_scene_for(float dt, int substeps, int iters) {
    PointScene s{};
    const float h = dt / (float)substeps;
    s.h = h; s.inv_h = 1.0f / h; s.substeps = substeps; s.iters = iters;
    const float g = 9.8f;
    const float invm_r = 1.0f / 10.0f;
    s.gam = 1.0f / (h * 600.0f);
    s.md = 1.0f / (invm_r + s.gam);
    s.dmax = 1000.0f * h;
    const float req = 0.3825978f * 0.4f;
    s.LlinB = ((0.75f * 16.0f) * g) * h; s.LangB = s.LlinB * req;
    s.LlinD = ((1.0f * 16.0f) * g) * h; s.LangD = s.LlinD * req;
    s.RcB = 1.5f * req; s.RcD = 1.5f * req;
    return s;
}
inline void make_point_scene(PointScene& s, float dt, int substeps, int iters) { s = point_scene_for(dt, substeps, iters); }
// The reference's own solver settings (isaacgym/point.yaml:4 dt 0.05; isaacgym_wrapper.py:10,28: 2 substeps, 6 position
// iterations) as a COMPILE-TIME scene: the rollout kernel instance of a handle with exactly these values (the bits of the
// thirteen run-time fields compared, rollout_point_kernel.hpp) sees them as literals -- thirteen fewer live scalar registers in a
// kernel that spills scalars, constant trip counts of the substep and pass loops.  Same values, same bits.
constexpr PointScene POINT_SCENE_REFERENCE = point_scene_for(0.05f, 2, 6);
inline bool point_scene_is_reference(const PointScene& s) {
    const PointScene& r = POINT_SCENE_REFERENCE;
    auto same = [](float a, float b) { return __builtin_memcmp(&a, &b, sizeof(float)) == 0; };
    return s.substeps == r.substeps && s.iters == r.iters && same(s.h, r.h) && same(s.inv_h, r.inv_h) && same(s.gam, r.gam) &&
           same(s.md, r.md) && same(s.dmax, r.dmax) && same(s.LlinB, r.LlinB) && same(s.LangB, r.LangB) && same(s.LlinD, r.LlinD) &&
           same(s.LangD, r.LangD) && same(s.RcB, r.RcB) && same(s.RcD, r.RcD);
}

struct Box {
    float x, y, c, s, vx, vy, w;
};

struct PointWorld {
    float rx, ry, rvx, rvy;  // robot disc (2 prismatic dofs, no rotation)
    Box B, D;                // pushable box, dynamic obstacle
    float fRx, fRy, fBx, fBy;  // pending external (suction) force, consumed by next step
    float fcDx, fcDy;          // net contact force on dyn-obs during the last substep
    float fcBx, fcBy, fcRx, fcRy;  // (step mode only)
};

enum BodyId { ROBOT = 0, BOXB = 1, BOXD = 2, STATIC = 3 };

struct Slot {
    float nx, ny;              // unit normal from body a to body b
    float rna, rnb, rta, rtb;  // r x n, r x t for both bodies
    float mn, mt, bias;
    float ln, lt;
    bool on;
};

struct Vel {
    float rvx, rvy;
    float bvx, bvy, bw;
    float dvx, dvy, dw;
};

// Clamps as single VALU operations.  The spec (and the oracle) write them as compare-and-assign;
// for non-NaN operands v_max / v_med3 return the same value, and they take one dependent issue
// slot (~8 cycles for a lone wavefront) instead of v_cmp -> s_nop -> v_cndmask (~28 cycles,
// tools/ubench/valu_chain.hip): the solver's critical path is made of these.
// spec v1.3: reciprocal square root = bit-trick seed + three Newton steps in binary32, in
// exactly this order (the oracle runs the same sequence).  The correctly rounded sqrtf followed by
// an IEEE division is a ~32-instruction dependent chain here (3 of them compare->select pairs);
// this is 15, and it sits in the friction row of every solver pass.
// spec v1.4: mad(a, b, c) = a*b + c in ONE rounding (spec_fma.hpp) wherever the dynamics form a product plus a sum.
__device__ __forceinline__ float spec_rsqrt(float a) {
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(a) >> 1));
    const float hlf = 0.5f * a;
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    y = y * mad(-hlf, y * y, 1.5f);
    return y;
}

// spec v1.6: the reciprocals of a substep (a contact's effective masses, the coupling factors below, the orientation update's
// 1 / (1 + a^2)) are spec_rcp (spec_fma.hpp), not IEEE divisions.

// spec v1.5: sliding-spinning coupling of a box's ground friction (Contensou's law in Zhuravlev's first-order Pade
// form for a disc of radius R = 1.5 r_eq: F = F0 v / (v + 8/(3 pi) u), M = M0 u / (u + 15 pi/16 v), u = R |w|): the
// factors of the linear / torsion rows' limits from the velocities a substep starts its passes with.  A patch that
// slides fast hardly resists turning -- the independent torsion row of spec v1.4 resisted it in full, which is what
// wedged the box behind the goal at the reference's shipped planner size (profiles/r04/ab_default_size_push.json).
__device__ __forceinline__ void friction_coupling(float vx, float vy, float w, float R, float& cl, float& ca) {
    const float s2 = mad(vx, vx, vy * vy);
    const float v = ((__float_as_uint(s2) & 0x7f800000u) == 0u) ? 0.0f : s2 * spec_rsqrt(s2);
    auto zero = [](float x) { return (__float_as_uint(x) & 0x7f800000u) == 0u; };   // (below the smallest normal number)
    const float u = zero(w) ? 0.0f : R * fabsf(w);
    cl = 1.0f; ca = 1.0f;
    if (!zero(v)) cl = v * spec_rcp(mad(0.8488264f, u, v));
    if (!zero(u)) ca = u * spec_rcp(mad(2.9452431f, v, u));
}

__device__ __forceinline__ float clamp_lo0(float x) { return fmaxf(x, 0.0f); }
__device__ __forceinline__ float clamp_sym(float x, float lim) {  // lim >= 0
    return __builtin_amdgcn_fmed3f(x, -lim, lim);
}
__device__ __forceinline__ float clamp_hi(float x, float hi) { return fminf(x, hi); }

template <int ID> __device__ __forceinline__ float gvx(const Vel& v) {
    if constexpr (ID == ROBOT) return v.rvx;
    else if constexpr (ID == BOXB) return v.bvx;
    else if constexpr (ID == BOXD) return v.dvx;
    else return 0.0f;
}
template <int ID> __device__ __forceinline__ float gvy(const Vel& v) {
    if constexpr (ID == ROBOT) return v.rvy;
    else if constexpr (ID == BOXB) return v.bvy;
    else if constexpr (ID == BOXD) return v.dvy;
    else return 0.0f;
}
template <int ID> __device__ __forceinline__ float gw(const Vel& v) {
    if constexpr (ID == BOXB) return v.bw;
    else if constexpr (ID == BOXD) return v.dw;
    else return 0.0f;
}
template <int ID> __device__ __forceinline__ float invm(const PointScene& sc) {
    if constexpr (ID == ROBOT) return sc.invm_r;
    else if constexpr (ID == BOXB) return sc.invm_b;
    else if constexpr (ID == BOXD) return sc.invm_d;
    else return 0.0f;
}
template <int ID> __device__ __forceinline__ float invI(const PointScene& sc) {
    if constexpr (ID == BOXB) return sc.invI_b;
    else if constexpr (ID == BOXD) return sc.invI_d;
    else return 0.0f;
}
template <int ID> constexpr bool rotates() { return ID == BOXB || ID == BOXD; }
template <int ID> constexpr bool moves() { return ID != STATIC; }

// apply impulse dl along (dx,dy) with angular arm ra to body ID, sign sg (-1 for a, +1 for b)
template <int ID, int SG>
__device__ __forceinline__ void apply(const PointScene& sc, Vel& v, float dl, float dx, float dy,
                                      float arm) {
    if constexpr (!moves<ID>()) return;
    const float im = (SG < 0) ? -(invm<ID>(sc) * dl) : invm<ID>(sc) * dl;
    if constexpr (ID == ROBOT) {
        v.rvx = mad(im, dx, v.rvx); v.rvy = mad(im, dy, v.rvy);
    } else if constexpr (ID == BOXB) {
        const float ia = (SG < 0) ? -(invI<ID>(sc) * arm) : invI<ID>(sc) * arm;
        v.bvx = mad(im, dx, v.bvx); v.bvy = mad(im, dy, v.bvy); v.bw = mad(ia, dl, v.bw);
    } else if constexpr (ID == BOXD) {
        const float ia = (SG < 0) ? -(invI<ID>(sc) * arm) : invI<ID>(sc) * arm;
        v.dvx = mad(im, dx, v.dvx); v.dvy = mad(im, dy, v.dvy); v.dw = mad(ia, dl, v.dw);
    }
}

// spec: prepare (effective masses, bias) of one contact
template <int A, int B>
__device__ __forceinline__ void prepare(const PointScene& sc, Slot& c, float nx, float ny,
                                        float rax, float ray, float rbx, float rby, float sep) {
    const float tx = -ny, ty = nx;
    c.nx = nx; c.ny = ny;
    c.rna = rotates<A>() ? mad(rax, ny, -(ray * nx)) : 0.0f;
    c.rnb = rotates<B>() ? mad(rbx, ny, -(rby * nx)) : 0.0f;
    c.rta = rotates<A>() ? mad(rax, ty, -(ray * tx)) : 0.0f;
    c.rtb = rotates<B>() ? mad(rbx, ty, -(rby * tx)) : 0.0f;
    float kn = invm<A>(sc), kt = invm<A>(sc);
    if constexpr (moves<B>()) { kn = kn + invm<B>(sc); kt = kt + invm<B>(sc); }
    if constexpr (rotates<A>()) {
        kn = mad(invI<A>(sc) * c.rna, c.rna, kn);
        kt = mad(invI<A>(sc) * c.rta, c.rta, kt);
    }
    if constexpr (rotates<B>()) {
        kn = mad(invI<B>(sc) * c.rnb, c.rnb, kn);
        kt = mad(invI<B>(sc) * c.rtb, c.rtb, kt);
    }
    c.mn = spec_rcp(kn);
    c.mt = spec_rcp(kt);
    if (sep > 0.0f) {
        c.bias = sep * sc.inv_h;
    } else {
        float pen = -sep - sc.slop;
        pen = clamp_lo0(pen);
        float push = (sc.baumgarte * pen) * sc.inv_h;
        push = clamp_hi(push, sc.max_bias);
        c.bias = -push;
    }
    c.ln = 0.0f; c.lt = 0.0f;
    c.on = true;
}

template <int A, int B>
__device__ __forceinline__ void solve(const PointScene& sc, Vel& v, Slot& c, float mu) {
    const float tx = -c.ny, ty = c.nx;
    float dvx = gvx<B>(v) - gvx<A>(v), dvy = gvy<B>(v) - gvy<A>(v);
    float vn = mad(dvx, c.nx, dvy * c.ny);
    if constexpr (rotates<B>()) vn = mad(gw<B>(v), c.rnb, vn);
    if constexpr (rotates<A>()) vn = mad(-gw<A>(v), c.rna, vn);
    float dl = -c.mn * (vn + c.bias);
    float l0 = c.ln;
    float l1 = l0 + dl;
    l1 = clamp_lo0(l1);
    c.ln = l1;
    dl = l1 - l0;
    apply<A, -1>(sc, v, dl, c.nx, c.ny, c.rna);
    apply<B, +1>(sc, v, dl, c.nx, c.ny, c.rnb);
    dvx = gvx<B>(v) - gvx<A>(v); dvy = gvy<B>(v) - gvy<A>(v);
    float vt = mad(dvx, tx, dvy * ty);
    if constexpr (rotates<B>()) vt = mad(gw<B>(v), c.rtb, vt);
    if constexpr (rotates<A>()) vt = mad(-gw<A>(v), c.rta, vt);
    dl = -c.mt * vt;
    const float maxf = mu * c.ln;
    l0 = c.lt;
    l1 = l0 + dl;
    l1 = clamp_sym(l1, maxf);
    c.lt = l1;
    dl = l1 - l0;
    apply<A, -1>(sc, v, dl, tx, ty, c.rta);
    apply<B, +1>(sc, v, dl, tx, ty, c.rtb);
}

// ---- narrow phase ---------------------------------------------------------------------
// robot disc (A) vs box (B, possibly static)
template <int B>
__device__ __forceinline__ void detect_disc_box(const PointScene& sc, Slot& c, float px, float py,
                                                float qx, float qy, float bc, float bs, float hx,
                                                float hy, float rad) {
    c.on = false;
    const float r = sc.robot_r;
    const float dx = px - qx, dy = py - qy;
    {   // conservative broad phase (does not change results)
        const float lim = r + rad + sc.contact_offset + 1e-3f;
        if (mad(dx, dx, dy * dy) > lim * lim) return;
    }
    const float lx = mad(bc, dx, bs * dy);
    const float ly = mad(bc, dy, -(bs * dx));
    float cx = fminf(fmaxf(lx, -hx), hx);
    float cy = fminf(fmaxf(ly, -hy), hy);
    const float ex = lx - cx, ey = ly - cy;
    const float d2 = mad(ex, ex, ey * ey);
    float nlx, nly, sep;
    if (d2 > 0.0f) {
        const float rd = spec_rsqrt(d2);
        const float d = d2 * rd;
        nlx = ex * rd; nly = ey * rd;
        sep = d - r;
    } else {
        const float ppx = hx - fabsf(lx), ppy = hy - fabsf(ly);
        if (ppx < ppy) {
            const float sg = (lx >= 0.0f) ? 1.0f : -1.0f;
            nlx = sg; nly = 0.0f; cx = sg * hx; cy = ly; sep = -ppx - r;
        } else {
            const float sg = (ly >= 0.0f) ? 1.0f : -1.0f;
            nlx = 0.0f; nly = sg; cx = lx; cy = sg * hy; sep = -ppy - r;
        }
    }
    if (!(sep < sc.contact_offset)) return;
    const float wx = mad(bc, nlx, -(bs * nly));
    const float wy = mad(bs, nlx, bc * nly);
    const float rbx = mad(bc, cx, -(bs * cy));
    const float rby = mad(bs, cx, bc * cy);
    prepare<ROBOT, B>(sc, c, -wx, -wy, 0.0f, 0.0f, rbx, rby, sep);
}

__device__ __forceinline__ void detect_disc_walls(const PointScene& sc, Slot& cx_, Slot& cy_,
                                                  float px, float py) {
    cx_.on = false; cy_.on = false;
    float sg = (px >= 0.0f) ? 1.0f : -1.0f;
    float sep = mad(-sg, px, sc.wall) - sc.robot_r;      // (sg = +-1: an exact product)
    if (sep < sc.contact_offset) prepare<ROBOT, STATIC>(sc, cx_, sg, 0.0f, 0.f, 0.f, 0.f, 0.f, sep);
    sg = (py >= 0.0f) ? 1.0f : -1.0f;
    sep = mad(-sg, py, sc.wall) - sc.robot_r;
    if (sep < sc.contact_offset) prepare<ROBOT, STATIC>(sc, cy_, 0.0f, sg, 0.f, 0.f, 0.f, 0.f, sep);
}

template <int A>
__device__ __forceinline__ void detect_box_walls(const PointScene& sc, Slot& x1, Slot& x2,
                                                 Slot& y1, Slot& y2, const Box& X, float hx,
                                                 float hy, float rad) {
    x1.on = x2.on = y1.on = y2.on = false;
    const float lim = rad + sc.contact_offset + 1e-3f;
    const bool nearx = (sc.wall - fabsf(X.x)) <= lim;
    const bool neary = (sc.wall - fabsf(X.y)) <= lim;
    if (!(nearx || neary)) return;
    const float r0x = mad(X.c, hx, -(X.s * hy)), r0y = mad(X.s, hx, X.c * hy);
    const float r1x = mad(-X.c, hx, -(X.s * hy)), r1y = mad(-X.s, hx, X.c * hy);
    if (nearx) {
        const float sg = (X.x >= 0.0f) ? 1.0f : -1.0f;
        const float base = mad(-sg, X.x, sc.wall);
        const float pa = sg * r0x, pb = sg * r1x;
        const float sep1 = base - fabsf(pa), sep2 = base - fabsf(pb);
        if (sep1 < sc.contact_offset) {
            const float f = (pa >= 0.0f) ? 1.0f : -1.0f;
            prepare<A, STATIC>(sc, x1, sg, 0.0f, f * r0x, f * r0y, 0.f, 0.f, sep1);
        }
        if (sep2 < sc.contact_offset) {
            const float f = (pb >= 0.0f) ? 1.0f : -1.0f;
            prepare<A, STATIC>(sc, x2, sg, 0.0f, f * r1x, f * r1y, 0.f, 0.f, sep2);
        }
    }
    if (neary) {
        const float sg = (X.y >= 0.0f) ? 1.0f : -1.0f;
        const float base = mad(-sg, X.y, sc.wall);
        const float pa = sg * r0y, pb = sg * r1y;
        const float sep1 = base - fabsf(pa), sep2 = base - fabsf(pb);
        if (sep1 < sc.contact_offset) {
            const float f = (pa >= 0.0f) ? 1.0f : -1.0f;
            prepare<A, STATIC>(sc, y1, 0.0f, sg, f * r0x, f * r0y, 0.f, 0.f, sep1);
        }
        if (sep2 < sc.contact_offset) {
            const float f = (pb >= 0.0f) ? 1.0f : -1.0f;
            prepare<A, STATIC>(sc, y2, 0.0f, sg, f * r1x, f * r1y, 0.f, 0.f, sep2);
        }
    }
}

// box A vs box B: SAT over the 4 face axes, reference face + clipped incident edge
template <int A, int B>
__device__ __forceinline__ void detect_box_box(const PointScene& sc, Slot& c1, Slot& c2, float ax,
                                               float ay, float ca, float sa, float hax, float hay,
                                               float rada, float bx, float by, float cb, float sb,
                                               float hbx, float hby, float radb) {
    c1.on = false; c2.on = false;
    const float dxw = bx - ax, dyw = by - ay;
    {
        const float lim = rada + radb + sc.contact_offset + 1e-3f;
        if (mad(dxw, dxw, dyw * dyw) > lim * lim) return;
    }
    const float dx = mad(ca, dxw, sa * dyw);
    const float dy = mad(ca, dyw, -(sa * dxw));
    const float cr = mad(ca, cb, sa * sb);
    const float sr = mad(ca, sb, -(sa * cb));
    const float acr = fabsf(cr), asr = fabsf(sr);
    const float sAx = fabsf(dx) - (hax + mad(acr, hbx, asr * hby));
    const float sAy = fabsf(dy) - (hay + mad(asr, hbx, acr * hby));
    const float ex = -mad(cr, dx, sr * dy);
    const float ey = -mad(cr, dy, -(sr * dx));
    const float sBx = fabsf(ex) - (hbx + mad(acr, hax, asr * hay));
    const float sBy = fabsf(ey) - (hby + mad(asr, hax, acr * hay));
    float best = sAx;
    int axis = 0;
    if (sAy > best + sc.face_tol) { best = sAy; axis = 1; }
    if (sBx > best + sc.face_tol) { best = sBx; axis = 2; }
    if (sBy > best + sc.face_tol) { best = sBy; axis = 3; }
    if (!(best < sc.contact_offset)) return;

    const bool refA = axis < 2;
    const float drx = refA ? dx : ex, dry = refA ? dy : ey;
    const float crr = cr, srr = refA ? sr : -sr;
    const float hrx = refA ? hax : hbx, hry = refA ? hay : hby;
    const float hix = refA ? hbx : hax, hiy = refA ? hby : hay;
    const bool xface = (axis & 1) == 0;
    const float dn = xface ? drx : dry;
    const float sg = (dn >= 0.0f) ? 1.0f : -1.0f;
    const float hn = xface ? hrx : hry;
    const float ht = xface ? hry : hrx;
    const float r0x = mad(crr, hix, -(srr * hiy)), r0y = mad(srr, hix, crr * hiy);
    const float r1x = mad(-crr, hix, -(srr * hiy)), r1y = mad(-srr, hix, crr * hiy);
    const float pa = sg * (xface ? r0x : r0y);
    const float pb = sg * (xface ? r1x : r1y);
    const float f0 = (pa <= 0.0f) ? 1.0f : -1.0f;
    const float f1 = (pb <= 0.0f) ? 1.0f : -1.0f;
    const float p1x = mad(f0, r0x, drx), p1y = mad(f0, r0y, dry);     // (f, sg = +-1: exact products)
    const float p2x = mad(f1, r1x, drx), p2y = mad(f1, r1y, dry);
    const float s1 = mad(sg, xface ? p1x : p1y, -hn);
    const float s2 = mad(sg, xface ? p2x : p2y, -hn);
    const float t1 = xface ? p1y : p1x;
    const float t2 = xface ? p2y : p2x;
    float cs1 = s1, ct1 = t1, cs2 = s2, ct2 = t2;
    bool ok = true;
    if (t1 > ht) {
        if (t2 > ht) ok = false;
        else { const float lam = (ht - t2) / (t1 - t2); cs1 = mad(lam, s1 - s2, s2); ct1 = ht; }
    } else if (t1 < -ht) {
        if (t2 < -ht) ok = false;
        else { const float lam = (-ht - t2) / (t1 - t2); cs1 = mad(lam, s1 - s2, s2); ct1 = -ht; }
    }
    if (t2 > ht) {
        if (!(t1 > ht)) { const float lam = (ht - t1) / (t2 - t1); cs2 = mad(lam, s2 - s1, s1); ct2 = ht; }
    } else if (t2 < -ht) {
        if (!(t1 < -ht)) { const float lam = (-ht - t1) / (t2 - t1); cs2 = mad(lam, s2 - s1, s1); ct2 = -ht; }
    }
    if (!ok) return;
    const float qrx = refA ? ax : bx, qry = refA ? ay : by;
    const float rc = refA ? ca : cb, rs = refA ? sa : sb;
    const float nrx = xface ? sg : 0.0f, nry = xface ? 0.0f : sg;
    float nwx = mad(rc, nrx, -(rs * nry)), nwy = mad(rs, nrx, rc * nry);
    if (!refA) { nwx = -nwx; nwy = -nwy; }
    if (cs1 < sc.contact_offset) {
        const float pn = sg * (hn + cs1);
        const float plx = xface ? pn : ct1, ply = xface ? ct1 : pn;
        const float pwx = qrx + mad(rc, plx, -(rs * ply));
        const float pwy = qry + mad(rs, plx, rc * ply);
        prepare<A, B>(sc, c1, nwx, nwy, pwx - ax, pwy - ay, pwx - bx, pwy - by, cs1);
    }
    if (cs2 < sc.contact_offset) {
        const float pn = sg * (hn + cs2);
        const float plx = xface ? pn : ct2, ply = xface ? ct2 : pn;
        const float pwx = qrx + mad(rc, plx, -(rs * ply));
        const float pwy = qry + mad(rs, plx, rc * ply);
        prepare<A, B>(sc, c2, nwx, nwy, pwx - ax, pwy - ay, pwx - bx, pwy - by, cs2);
    }
}

struct Fric {
    float lx, ly, la;
};

// (A branch-free version of this row -- selects instead of the two exec-mask regions, so that the
// scheduler could overlap it with the drive rows and the other box's row -- was measured 9 %
// SLOWER: the disc clamp's IEEE sqrt + divide chain then sits on every pass's critical path, also
// for bodies at rest or inside the friction disc, and the compiler does not interleave two such
// expanded chains anyway: tools/ubench/valu_chain.hip, "2 independent div/sqrt".)
template <int ID>
__device__ __forceinline__ void solve_ground_friction(const PointScene& sc, Vel& v, Fric& f,
                                                      float m, float I, float Llin, float Lang) {
    // spec: a body at rest (v = 0 and w = 0) has no friction row in this pass.  v1.4: "zero" = below the smallest normal
    // number -- 1 / I is a rounded reciprocal, so an angular friction row leaves ~1e-8 of the spin, which the next
    // passes shrink by that factor each until it is subnormal, where it can stay (-I * 1e-45 rounds to 0): such a
    // body IS at rest.  (one v_or3 + v_and + compare on the bit patterns: the exponent fields of all three are zero
    // -- instead of three float compares joined through scalar registers)
    if (((__float_as_uint(gvx<ID>(v)) | __float_as_uint(gvy<ID>(v)) | __float_as_uint(gw<ID>(v))) & 0x7f800000u) == 0u) return;
    float nlx = mad(-m, gvx<ID>(v), f.lx);
    float nly = mad(-m, gvy<ID>(v), f.ly);
    const float mag2 = mad(nlx, nlx, nly * nly);
    {   // disc clamp as a select: a body that moves is almost always sliding (saturated), so the
        // sqrt + divide chain is on the path anyway and the exec-mask region around it only
        // added its ~40-cycle turnaround; unused lanes' inf / NaN are discarded by the select
        const float scl = Llin * spec_rsqrt(mag2);
        const bool sat = mag2 > Llin * Llin;
        nlx = sat ? nlx * scl : nlx;
        nly = sat ? nly * scl : nly;
    }
    float nla = mad(-I, gw<ID>(v), f.la);
    if constexpr (ID == BOXB) {
        v.bvx = mad(sc.invm_b, nlx - f.lx, v.bvx);
        v.bvy = mad(sc.invm_b, nly - f.ly, v.bvy);
    } else {
        v.dvx = mad(sc.invm_d, nlx - f.lx, v.dvx);
        v.dvy = mad(sc.invm_d, nly - f.ly, v.dvy);
    }
    f.lx = nlx; f.ly = nly;
    nla = clamp_sym(nla, Lang);
    if constexpr (ID == BOXB) v.bw = mad(sc.invI_b, nla - f.la, v.bw);
    else v.dw = mad(sc.invI_d, nla - f.la, v.dw);
    f.la = nla;
}

__device__ __forceinline__ void integrate_box(Box& X, float h) {
    X.x = mad(h, X.vx, X.x);
    X.y = mad(h, X.vy, X.y);
    if ((__float_as_uint(X.w) & 0x7f800000u) == 0u) return;  // spec: orientation is only touched when w != 0 (v1.4: subnormal = 0)
    const float a = 0.5f * (h * X.w);
    const float a2 = a * a;
    const float den = 1.0f + a2;
    const float rden = spec_rcp(den);
    const float cd = (1.0f - a2) * rden;
    const float sd = (2.0f * a) * rden;
    const float c = mad(X.c, cd, -(X.s * sd));
    const float s = mad(X.s, cd, X.c * sd);
    const float rn = spec_rsqrt(mad(c, c, s * s));
    X.c = c * rn;
    X.s = s * rn;
}

// impulse of slot c on its body b (+) / a (-), accumulated in slot order like the oracle
#define M3_ACC(fx, fy, c, sign)                                   \
    if ((c).on) {                                                 \
        const float ix_ = mad((c).ln, (c).nx, (c).lt * (-(c).ny)); \
        const float iy_ = mad((c).ln, (c).ny, (c).lt * (c).nx);    \
        if ((sign) > 0) { fx += ix_; fy += iy_; }                 \
        else { fx -= ix_; fy -= iy_; }                            \
    }

// Broad-phase predicates: the same expressions as the early-outs inside the detect functions, so
// "not near" here implies that the corresponding slots come out `on = false`.
__device__ __forceinline__ bool near_centres(const PointScene& sc, float ax, float ay, float bx, float by,
                                             float ra, float rb) {
    const float dx = bx - ax, dy = by - ay;
    const float lim = ra + rb + sc.contact_offset + 1e-3f;
    return !(mad(dx, dx, dy * dy) > lim * lim);
}
__device__ __forceinline__ bool near_walls_disc(const PointScene& sc, float px, float py) {
    return ((sc.wall - fabsf(px)) - sc.robot_r < sc.contact_offset) ||
           ((sc.wall - fabsf(py)) - sc.robot_r < sc.contact_offset);
}
__device__ __forceinline__ bool near_walls_box(const PointScene& sc, const Box& X, float rad) {
    const float lim = rad + sc.contact_offset + 1e-3f;
    return ((sc.wall - fabsf(X.x)) <= lim) || ((sc.wall - fabsf(X.y)) <= lim);
}

// One substep: forces, detect, solve, integrate.
// The mask M selects which pair groups' contact slots EXIST in the generated code.  The full
// version keeps 19 slots x 11 values live across the solver loop: 344 VGPRs (arch + acc, i.e. AGPR
// spill moves), 67 spilled SGPRs and 5.5 k instructions, and the common case -- nothing but (at
// most) the robot-box pair in range -- paid for carrying them: compiled WITHOUT the other slots
// the same substep needs 104 VGPRs and ran 1.75x faster.  point_step therefore forms, per substep
// and per WAVE, the mask of pair groups with any lane inside its broad-phase range and runs the
// leanest instance that covers it; results are identical because a pair outside its broad-phase
// range cannot produce a contact.  Which instances exist was chosen from the mask histogram of the
// bench workloads in steady state (tools/mask_count.py, -DM3_ABL_COUNT build; push / hybrid /
// north-star): no pair 36-41 %, robot-box 36-43 %, + robot-dyn-obs 16-19 %, + box-dyn-obs 1-9 %,
// everything else < 2 % together.  (History on C2: one general instance 0.227 ms; {none, robot-box,
// all} 0.213; + the robot's five slots 0.208; the set below 0.189.  Calling a heavy instance out of
// line (__noinline__, world through scratch) to shield the lean code's register allocation was
// slower, 0.263 ms; so was a verdict-only narrow phase to decide on real contacts instead of ranges,
// 0.237; keeping the 18 rare slots of the general instance in LDS instead of registers (164 VGPRs
// for the whole kernel) changed nothing at K <= 32000 and cost a quarter of the saturated
// throughput: 50 KB of LDS per wave allow three waves per CU instead of four.)
// pair groups of the 19 static slots (bit mask M of point_substep)
enum : unsigned {
    G_RB = 1u,    // robot - box                       1 slot
    G_RD = 2u,    // robot - dyn-obs                   1
    G_RO = 4u,    // robot - obstacle                  1
    G_RW = 8u,    // robot - walls                     2
    G_BW = 16u,   // box - walls                       4
    G_DW = 32u,   // dyn-obs - walls                   4
    G_BD = 64u,   // box - dyn-obs                     2
    G_BO = 128u,  // box - obstacle                    2
    G_DO = 256u,  // dyn-obs - obstacle                2
    G_ALL = 511u,
    G_NO_BOX_STATICS = G_RB | G_RD | G_RO | G_RW | G_BD,  // everything but box/dyn-obs vs walls and obstacle
    G_CORNER = G_RB | G_RW | G_BW   // the box being pushed along / into the walls: the scene of a goal in a corner
                                    // (config_point.yaml goal [-3.75, -3.75]) once the box gets there
};

// -DM3_ABL_PHASES (experiments, tools/phase_breakdown.py): shader-clock time of every wave per phase of
// the rollout -- 0 action assembly + prefetch, 1 broad-phase mask + dispatch, 2 forces + detection, 3 the
// solver passes, 4 net force + integration, 5 task cost, 6 stores + accumulation
#ifdef M3_ABL_PHASES
static __device__ unsigned long long g_phase[1024 * 8];
struct PhaseClock {
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    __device__ __forceinline__ void start() { last = __builtin_readcyclecounter(); }
    template <int N> __device__ __forceinline__ void mark() {
        const unsigned long long now = __builtin_readcyclecounter();
        acc[N] += now - last;
        last = now;
    }
};
#define M3_PH(n) if (pc_) pc_->template mark<n>()
#else
struct PhaseClock {};
#define M3_PH(n)
#endif

struct RowOn { static constexpr bool value = true; };     // compile-time flags of point_substep's pass versions
struct RowOff { static constexpr bool value = false; };
template <bool ALL_FORCES, unsigned M>
__device__ __forceinline__ void point_substep(const PointScene& sc, PointWorld& w, float ux, float uy,
                                              bool form_dyn_force, PhaseClock* pc_ = nullptr) {
    M3_PH(1);
    constexpr bool RB = M & G_RB, RD = M & G_RD, RO = M & G_RO, RW = M & G_RW, BW = M & G_BW,
                   DW = M & G_DW, BD = M & G_BD, BO = M & G_BO, DO = M & G_DO;
    constexpr bool ANY_RARE = (M & ~G_RB) != 0u, ANY_WALLS = RW || BW || DW, ANY_BOXES = BD || BO || DO;
    const float h = sc.h;
    // 1. external forces.  Spec v1.7: a pending force (the suction pair, Q5: staged during the previous step) is CONSUMED
    // by the first substep of the step -- cleared right here; the later substeps evaluate the same expression on zeros,
    // as the oracle does.  (Up to v1.6 it acted in every substep.  The joint fit against the reference's eight logged
    // scenarios, tools/cpu_fit_physx.py -> profiles/r06/fit_physx_*.json, selects this reading of a one-shot
    // apply_rigid_body_force_tensors under 2 substeps: DESIGN.md section 2.)
    w.rvx = mad(h * w.fRx, sc.invm_r, w.rvx);
    w.rvy = mad(h * w.fRy, sc.invm_r, w.rvy);
    w.B.vx = mad(h * w.fBx, sc.invm_b, w.B.vx);
    w.B.vy = mad(h * w.fBy, sc.invm_b, w.B.vy);
    w.fRx = 0.0f; w.fRy = 0.0f; w.fBx = 0.0f; w.fBy = 0.0f;

    // 2. contacts (static slots; a group that is not in M stays `on = false` and compiles away)
    Slot s_rb, s_rd, s_ro, s_rwx, s_rwy;
    Slot s_bx1, s_bx2, s_by1, s_by2, s_dx1, s_dx2, s_dy1, s_dy2;
    Slot s_bd1, s_bd2, s_bo1, s_bo2, s_do1, s_do2;
    s_rb.on = false;
    s_rd.on = s_ro.on = s_rwx.on = s_rwy.on = false;
    s_bx1.on = s_bx2.on = s_by1.on = s_by2.on = s_dx1.on = s_dx2.on = s_dy1.on = s_dy2.on = false;
    s_bd1.on = s_bd2.on = s_bo1.on = s_bo2.on = s_do1.on = s_do2.on = false;
    if constexpr (RB)
        detect_disc_box<BOXB>(sc, s_rb, w.rx, w.ry, w.B.x, w.B.y, w.B.c, w.B.s, sc.box_hx,
                              sc.box_hy, sc.rad_b);
    if constexpr (RD)
        detect_disc_box<BOXD>(sc, s_rd, w.rx, w.ry, w.D.x, w.D.y, w.D.c, w.D.s, sc.dyn_hx,
                              sc.dyn_hy, sc.rad_d);
    if constexpr (RO)
        detect_disc_box<STATIC>(sc, s_ro, w.rx, w.ry, sc.obs_x, sc.obs_y, 1.0f, 0.0f, sc.obs_hx,
                                sc.obs_hy, sc.rad_o);
    if constexpr (RW) detect_disc_walls(sc, s_rwx, s_rwy, w.rx, w.ry);
    if constexpr (BW)
        detect_box_walls<BOXB>(sc, s_bx1, s_bx2, s_by1, s_by2, w.B, sc.box_hx, sc.box_hy, sc.rad_b);
    if constexpr (DW)
        detect_box_walls<BOXD>(sc, s_dx1, s_dx2, s_dy1, s_dy2, w.D, sc.dyn_hx, sc.dyn_hy, sc.rad_d);
    if constexpr (BD)
        detect_box_box<BOXB, BOXD>(sc, s_bd1, s_bd2, w.B.x, w.B.y, w.B.c, w.B.s, sc.box_hx,
                                   sc.box_hy, sc.rad_b, w.D.x, w.D.y, w.D.c, w.D.s, sc.dyn_hx,
                                   sc.dyn_hy, sc.rad_d);
    if constexpr (BO)
        detect_box_box<BOXB, STATIC>(sc, s_bo1, s_bo2, w.B.x, w.B.y, w.B.c, w.B.s, sc.box_hx,
                                     sc.box_hy, sc.rad_b, sc.obs_x, sc.obs_y, 1.0f, 0.0f,
                                     sc.obs_hx, sc.obs_hy, sc.rad_o);
    if constexpr (DO)
        detect_box_box<BOXD, STATIC>(sc, s_do1, s_do2, w.D.x, w.D.y, w.D.c, w.D.s, sc.dyn_hx,
                                     sc.dyn_hy, sc.rad_d, sc.obs_x, sc.obs_y, 1.0f, 0.0f,
                                     sc.obs_hx, sc.obs_hy, sc.rad_o);
    // (flags of absent groups are compile-time false)
    const bool on_walls = s_rwx.on | s_rwy.on | s_bx1.on | s_bx2.on | s_by1.on | s_by2.on |
                          s_dx1.on | s_dx2.on | s_dy1.on | s_dy2.on;
    const bool on_boxes = s_bd1.on | s_bd2.on | s_bo1.on | s_bo2.on | s_do1.on | s_do2.on;
    const bool rare = s_rd.on | s_ro.on | on_walls | on_boxes;

    M3_PH(2);
    // 3. velocity solve
    Vel v = {w.rvx, w.rvy, w.B.vx, w.B.vy, w.B.w, w.D.vx, w.D.vy, w.D.w};
    float ldx = 0.0f, ldy = 0.0f;
    Fric fB = {0.f, 0.f, 0.f}, fD = {0.f, 0.f, 0.f};
    // spec v1.5: the friction rows' limits of this substep (a wave in which no lane's box moves keeps the plain ones:
    // the factors of a body at rest are 1)
    float LlinBe = sc.LlinB, LangBe = sc.LangB, LlinDe = sc.LlinD, LangDe = sc.LangD;
    if (__builtin_amdgcn_ballot_w64(((__float_as_uint(v.bvx) | __float_as_uint(v.bvy) | __float_as_uint(v.bw)) & 0x7f800000u) != 0u) != 0ull) {
        float cl, ca;
        friction_coupling(v.bvx, v.bvy, v.bw, sc.RcB, cl, ca);
        LlinBe = sc.LlinB * cl; LangBe = sc.LangB * ca;
    }
    if (__builtin_amdgcn_ballot_w64(((__float_as_uint(v.dvx) | __float_as_uint(v.dvy) | __float_as_uint(v.dw)) & 0x7f800000u) != 0u) != 0ull) {
        float cl, ca;
        friction_coupling(v.dvx, v.dvy, v.dw, sc.RcD, cl, ca);
        LlinDe = sc.LlinD * cl; LangDe = sc.LangD * ca;
    }
    // A body that no slot of this instance touches and that is at rest now stays at rest for the
    // whole substep (its friction row is a no-op at rest), so its per-pass rest test -- an exec-mask
    // region of ~36 cycles, twelve of them per substep -- is decided once per wave instead.
    constexpr bool B_FREE = (M & (G_RB | G_BW | G_BD | G_BO)) == 0u, D_FREE = (M & (G_RD | G_DW | G_BD | G_DO)) == 0u;
    bool skipB = false, skipD = false;
    if constexpr (B_FREE && !ALL_FORCES)
        skipB = __builtin_amdgcn_ballot_w64(((__float_as_uint(v.bvx) | __float_as_uint(v.bvy) | __float_as_uint(v.bw)) & 0x7f800000u) != 0u) == 0ull;
    if constexpr (D_FREE && !ALL_FORCES)
        skipD = __builtin_amdgcn_ballot_w64(((__float_as_uint(v.dvx) | __float_as_uint(v.dvy) | __float_as_uint(v.dw)) & 0x7f800000u) != 0u) == 0ull;
    // The three lean instances (no pair at all / robot-box / + robot-dyn-obs: nearly every substep of the planned
    // rollouts) run their passes in VERSIONS chosen once per substep by wave-uniform flags -- with / without the
    // box's and the dyn-obs' friction rows and the robot-dyn-obs row -- and, for the reference's six iterations,
    // fully unrolled: inside the generic loop below every pass paid two or three TAKEN scalar branches (rows of
    // bodies at rest and of contacts no lane has are skipped) and the loop's own -- a taken branch refills the lone
    // wavefront's instruction buffer, ~30 cycles against a drive-row pass of ~120 -- and the straight-line form lets
    // the scheduler interleave the independent rows.  Same operations in the same order: identical bits.
    // (C2 0.144 -> 0.136 ms, C3 0.182 -> 0.170, north-star 0.173 -> 0.163.)
    constexpr bool LEAN = !ALL_FORCES && (M == 0u || M == G_RB || M == (G_RB | G_RD));
    if constexpr (LEAN) {
        auto pass = [&](auto with_b, auto with_d, auto with_rd) {
            {
                float dl = -(mad(sc.gam, ldx, v.rvx - ux) * sc.md);
                float l1 = ldx + dl;
                l1 = clamp_sym(l1, sc.dmax);
                v.rvx = mad(sc.invm_r, l1 - ldx, v.rvx);
                ldx = l1;
                dl = -(mad(sc.gam, ldy, v.rvy - uy) * sc.md);
                l1 = ldy + dl;
                l1 = clamp_sym(l1, sc.dmax);
                v.rvy = mad(sc.invm_r, l1 - ldy, v.rvy);
                ldy = l1;
            }
            if constexpr (decltype(with_b)::value) solve_ground_friction<BOXB>(sc, v, fB, sc.box_m, sc.box_I, LlinBe, LangBe);
            if constexpr (RB) {
                if (s_rb.on) solve<ROBOT, BOXB>(sc, v, s_rb, sc.mu_rb);
            }
            if constexpr (decltype(with_d)::value) solve_ground_friction<BOXD>(sc, v, fD, sc.dyn_m, sc.dyn_I, LlinDe, LangDe);
            if constexpr (RD && decltype(with_rd)::value) {
                if (s_rd.on) solve<ROBOT, BOXD>(sc, v, s_rd, sc.mu_rd);
            }
        };
        // (a version runs to the end of the substep: the integration of a body whose rows it does not have is skipped
        // with them -- x + h * 0 == x -- without a branch of its own)
        auto passes = [&](auto with_b, auto with_d, auto with_rd) {
            if (sc.iters == 6) {
                pass(with_b, with_d, with_rd); pass(with_b, with_d, with_rd); pass(with_b, with_d, with_rd);
                pass(with_b, with_d, with_rd); pass(with_b, with_d, with_rd); pass(with_b, with_d, with_rd);
            } else {
                for (int it = 0; it < sc.iters; ++it) pass(with_b, with_d, with_rd);
            }
            w.rvx = v.rvx; w.rvy = v.rvy;
            w.B.vx = v.bvx; w.B.vy = v.bvy; w.B.w = v.bw;
            w.D.vx = v.dvx; w.D.vy = v.dvy; w.D.w = v.dw;
            M3_PH(3);
            if (form_dyn_force) {   // (see below: the navigation cost's input)
                float fx = 0.0f, fy = 0.0f;
                if constexpr (RD) { M3_ACC(fx, fy, s_rd, +1) }
                fx += fD.lx; fy += fD.ly;
                w.fcDx = fx * sc.inv_h; w.fcDy = fy * sc.inv_h;
            }
            w.rx = mad(h, w.rvx, w.rx);
            w.ry = mad(h, w.rvy, w.ry);
            if constexpr (decltype(with_b)::value) integrate_box(w.B, h);
            if constexpr (decltype(with_d)::value) integrate_box(w.D, h);
            M3_PH(4);
        };
        if constexpr (RD) {
            // (the dyn-obs is in reach of the robot: usually no lane touches it, and then a dyn-obs that rests in
            // every lane stays at rest)
            const bool any_rd = __builtin_amdgcn_ballot_w64(s_rd.on) != 0ull;
            const bool d_rests = __builtin_amdgcn_ballot_w64(((__float_as_uint(v.dvx) | __float_as_uint(v.dvy) | __float_as_uint(v.dw)) & 0x7f800000u) != 0u) == 0ull;
            if (any_rd) passes(RowOn{}, RowOn{}, RowOn{});
            else if (d_rests) passes(RowOn{}, RowOff{}, RowOff{});
            else passes(RowOn{}, RowOn{}, RowOff{});
        } else {
            if (skipB) { if (skipD) passes(RowOff{}, RowOff{}, RowOff{}); else passes(RowOff{}, RowOn{}, RowOff{}); }
            else { if (skipD) passes(RowOn{}, RowOff{}, RowOff{}); else passes(RowOn{}, RowOn{}, RowOff{}); }
        }
        return;
    }
    // (generic pass; the corner instance -- the whole closed loop once the box is pushed along the walls -- picks a
    // version without the dyn-obs friction row / the robot-wall rows when no lane needs them: rows every lane would
    // skip cost a taken branch each, see LEAN above; not unrolled: six copies of these passes measured +7 %)
    auto gen_pass = [&](auto with_d, auto with_rw) {
        {
            float dl = -(mad(sc.gam, ldx, v.rvx - ux) * sc.md);
            float l1 = ldx + dl;
            l1 = clamp_sym(l1, sc.dmax);
            v.rvx = mad(sc.invm_r, l1 - ldx, v.rvx);
            ldx = l1;
            dl = -(mad(sc.gam, ldy, v.rvy - uy) * sc.md);
            l1 = ldy + dl;
            l1 = clamp_sym(l1, sc.dmax);
            v.rvy = mad(sc.invm_r, l1 - ldy, v.rvy);
            ldy = l1;
        }
        if (!skipB) solve_ground_friction<BOXB>(sc, v, fB, sc.box_m, sc.box_I, LlinBe, LangBe);
        // spec order: friction(dyn-obs) then robot-box.  The two rows share no body, so they
        // commute exactly; solving robot-box first lets the dyn-obs row (usually at rest) join
        // the rarely-taken group below: one skipped branch per pass instead of two.
        if constexpr (RB) {
            if (s_rb.on) solve<ROBOT, BOXB>(sc, v, s_rb, sc.mu_rb);
        }
        const bool d_moving = decltype(with_d)::value && ((__float_as_uint(v.dvx) | __float_as_uint(v.dvy) | __float_as_uint(v.dw)) & 0x7f800000u) != 0u;
        // (skipD: the dyn-obs rests in every lane and no slot of this instance touches it -- its friction
        // row is a no-op; the rarely-active slots below do not depend on it)
        if ((!skipD && d_moving) | rare) {
            if constexpr (decltype(with_d)::value) { if (!skipD) solve_ground_friction<BOXD>(sc, v, fD, sc.dyn_m, sc.dyn_I, LlinDe, LangDe); }
            if constexpr (ANY_RARE) {
                // one outer flag + two group flags: a pass in which no lane has any of these pays
                // one skipped exec-mask branch (same solve order as the spec)
                if (rare) {
                    if constexpr (RD) if (s_rd.on) solve<ROBOT, BOXD>(sc, v, s_rd, sc.mu_rd);
                    if constexpr (RO) if (s_ro.on) solve<ROBOT, STATIC>(sc, v, s_ro, sc.mu_ro);
                    if constexpr (ANY_WALLS) if (on_walls) {
                        if constexpr (RW && decltype(with_rw)::value) {
                            if (s_rwx.on) solve<ROBOT, STATIC>(sc, v, s_rwx, sc.mu_rw);
                            if (s_rwy.on) solve<ROBOT, STATIC>(sc, v, s_rwy, sc.mu_rw);
                        }
                        if constexpr (BW) {
                            if (s_bx1.on) solve<BOXB, STATIC>(sc, v, s_bx1, sc.mu_bw);
                            if (s_bx2.on) solve<BOXB, STATIC>(sc, v, s_bx2, sc.mu_bw);
                            if (s_by1.on) solve<BOXB, STATIC>(sc, v, s_by1, sc.mu_bw);
                            if (s_by2.on) solve<BOXB, STATIC>(sc, v, s_by2, sc.mu_bw);
                        }
                        if constexpr (DW) {
                            if (s_dx1.on) solve<BOXD, STATIC>(sc, v, s_dx1, sc.mu_dw);
                            if (s_dx2.on) solve<BOXD, STATIC>(sc, v, s_dx2, sc.mu_dw);
                            if (s_dy1.on) solve<BOXD, STATIC>(sc, v, s_dy1, sc.mu_dw);
                            if (s_dy2.on) solve<BOXD, STATIC>(sc, v, s_dy2, sc.mu_dw);
                        }
                    }
                    if constexpr (ANY_BOXES) if (on_boxes) {
                        if constexpr (BD) {
                            if (s_bd1.on) solve<BOXB, BOXD>(sc, v, s_bd1, sc.mu_bd);
                            if (s_bd2.on) solve<BOXB, BOXD>(sc, v, s_bd2, sc.mu_bd);
                        }
                        if constexpr (BO) {
                            if (s_bo1.on) solve<BOXB, STATIC>(sc, v, s_bo1, sc.mu_bo);
                            if (s_bo2.on) solve<BOXB, STATIC>(sc, v, s_bo2, sc.mu_bo);
                        }
                        if constexpr (DO) {
                            if (s_do1.on) solve<BOXD, STATIC>(sc, v, s_do1, sc.mu_do);
                            if (s_do2.on) solve<BOXD, STATIC>(sc, v, s_do2, sc.mu_do);
                        }
                    }
                }
            }
        }
        };
    auto gen_passes = [&](auto with_d, auto with_rw) {
#pragma nounroll      // (six copies of these passes measured +7 %; with the reference scene compiled in the trip count is a constant)
        for (int it = 0; it < sc.iters; ++it) gen_pass(with_d, with_rw);
    };
    if constexpr (M == G_CORNER && !ALL_FORCES) {
        const bool any_rw = __builtin_amdgcn_ballot_w64(s_rwx.on | s_rwy.on) != 0ull;
        if (skipD) { if (any_rw) gen_passes(RowOff{}, RowOn{}); else gen_passes(RowOff{}, RowOff{}); }
        else { if (any_rw) gen_passes(RowOn{}, RowOn{}); else gen_passes(RowOn{}, RowOff{}); }
    } else {
        gen_passes(RowOn{}, RowOn{});
    }
    w.rvx = v.rvx; w.rvy = v.rvy;
    w.B.vx = v.bvx; w.B.vy = v.bvy; w.B.w = v.bw;
    w.D.vx = v.dvx; w.D.vy = v.dvy; w.D.w = v.dw;
    M3_PH(3);

    // net contact force on the dyn-obs (get_motion_cost reads it), slot order.  The cost sees only
    // the LAST substep's value (spec), and only the navigation cost reads it at all
    // (cost_functions.py:38,158-169), so the rollout forms it just then (the step-mode wrapper
    // exposes all bodies' forces and keeps the general path).
    if (ALL_FORCES || form_dyn_force) {
        float fx = 0.0f, fy = 0.0f;
        if constexpr (RD) { M3_ACC(fx, fy, s_rd, +1) }
        if constexpr (DW) if (on_walls) {
            M3_ACC(fx, fy, s_dx1, -1) M3_ACC(fx, fy, s_dx2, -1)
            M3_ACC(fx, fy, s_dy1, -1) M3_ACC(fx, fy, s_dy2, -1)
        }
        if constexpr (BD || DO) if (on_boxes) {
            M3_ACC(fx, fy, s_bd1, +1) M3_ACC(fx, fy, s_bd2, +1)
            M3_ACC(fx, fy, s_do1, -1) M3_ACC(fx, fy, s_do2, -1)
        }
        fx += fD.lx; fy += fD.ly;
        w.fcDx = fx * sc.inv_h; w.fcDy = fy * sc.inv_h;
    }
    if constexpr (ALL_FORCES) {
        float fx = 0.0f, fy = 0.0f;
        M3_ACC(fx, fy, s_rb, +1)
        M3_ACC(fx, fy, s_bx1, -1) M3_ACC(fx, fy, s_bx2, -1)
        M3_ACC(fx, fy, s_by1, -1) M3_ACC(fx, fy, s_by2, -1)
        M3_ACC(fx, fy, s_bd1, -1) M3_ACC(fx, fy, s_bd2, -1)
        M3_ACC(fx, fy, s_bo1, -1) M3_ACC(fx, fy, s_bo2, -1)
        fx += fB.lx; fy += fB.ly;
        w.fcBx = fx * sc.inv_h; w.fcBy = fy * sc.inv_h;
        fx = 0.0f; fy = 0.0f;
        M3_ACC(fx, fy, s_rb, -1) M3_ACC(fx, fy, s_rd, -1) M3_ACC(fx, fy, s_ro, -1)
        M3_ACC(fx, fy, s_rwx, -1) M3_ACC(fx, fy, s_rwy, -1)
        w.fcRx = fx * sc.inv_h; w.fcRy = fy * sc.inv_h;
    }

    // 4. integrate
    w.rx = mad(h, w.rvx, w.rx);
    w.ry = mad(h, w.rvy, w.ry);
    if (!skipB) integrate_box(w.B, h);   // (a wave whose boxes all rest: x + h * 0 == x)
    if (!skipD) integrate_box(w.D, h);
    M3_PH(4);
}

#ifdef M3_ABL_COUNT
static __device__ unsigned int g_lvl[512];
static __device__ unsigned int g_cyc[64 * 16];
#endif

// one sim.step(): substeps x (forces, detect, solve, integrate)
template <bool ALL_FORCES>
__device__ __forceinline__ void point_step(const PointScene& sc, PointWorld& w, float ux, float uy,
                                           bool need_dyn_force = true, PhaseClock* pc_ = nullptr) {
    for (int sub = 0; sub < sc.substeps; ++sub) {
        const bool form = need_dyn_force && sub == sc.substeps - 1;
        if constexpr (ALL_FORCES) {
            point_substep<true, G_ALL>(sc, w, ux, uy, form);   // step mode: one environment per lane, general path
        } else {
            // wave-uniform mask of the pair groups with any lane inside its broad-phase range
            unsigned m = 0u;
#define M3_NEAR(bit, expr) if (__builtin_amdgcn_ballot_w64(expr) != 0ull) m |= (bit);
            M3_NEAR(G_RB, near_centres(sc, w.B.x, w.B.y, w.rx, w.ry, sc.robot_r, sc.rad_b))
            M3_NEAR(G_RD, near_centres(sc, w.D.x, w.D.y, w.rx, w.ry, sc.robot_r, sc.rad_d))
            M3_NEAR(G_RO, near_centres(sc, sc.obs_x, sc.obs_y, w.rx, w.ry, sc.robot_r, sc.rad_o))
            M3_NEAR(G_RW, near_walls_disc(sc, w.rx, w.ry))
            M3_NEAR(G_BW, near_walls_box(sc, w.B, sc.rad_b))
            M3_NEAR(G_DW, near_walls_box(sc, w.D, sc.rad_d))
            M3_NEAR(G_BD, near_centres(sc, w.B.x, w.B.y, w.D.x, w.D.y, sc.rad_b, sc.rad_d))
            M3_NEAR(G_BO, near_centres(sc, w.B.x, w.B.y, sc.obs_x, sc.obs_y, sc.rad_b, sc.rad_o))
            M3_NEAR(G_DO, near_centres(sc, w.D.x, w.D.y, sc.obs_x, sc.obs_y, sc.rad_d, sc.rad_o))
#undef M3_NEAR
#ifdef M3_ABL_COUNT
            atomicAdd(&g_lvl[m], 1u);   // (every active lane: the tools normalise the histogram)
#ifdef M3_ABL_COUNT_CYCLES   // (instance timings as well: this part is NOT safe with the pass versions of point_substep
            // -- the build passes none of the bit-exactness tests, trajectory costs come out wrong; a compiler problem
            // around the clock reads / atomics in this much code that was not chased.  The histogram alone is fine:
            // tests/test_hip_parity_point.py passes on a -DM3_ABL_COUNT build.)
            const unsigned long long t0_ = wall_clock64();
#endif
#endif
            // the leanest instance that covers the mask
#define M3_COVERS(set) ((m & ~(set)) == 0u)
            if (m == 0u) point_substep<false, 0u>(sc, w, ux, uy, form, pc_);
            else if (M3_COVERS(G_RB)) point_substep<false, G_RB>(sc, w, ux, uy, form, pc_);
            else if (M3_COVERS(G_RB | G_RD)) point_substep<false, G_RB | G_RD>(sc, w, ux, uy, form, pc_);
            else if (M3_COVERS(G_RB | G_RD | G_BD)) point_substep<false, G_RB | G_RD | G_BD>(sc, w, ux, uy, form, pc_);
#ifndef M3_ABL_NO_CORNER
            else if (M3_COVERS(G_CORNER)) point_substep<false, G_CORNER>(sc, w, ux, uy, form, pc_);
#endif
            else if (M3_COVERS(G_NO_BOX_STATICS)) point_substep<false, G_NO_BOX_STATICS>(sc, w, ux, uy, form, pc_);
            else point_substep<false, G_ALL>(sc, w, ux, uy, form, pc_);
#if defined(M3_ABL_COUNT) && defined(M3_ABL_COUNT_CYCLES)
            {
                const unsigned long long dt_ = wall_clock64() - t0_;
                const int cls_ = m == 0u ? 0 : M3_COVERS(G_RB) ? 1 : M3_COVERS(G_RB | G_RD) ? 2 : M3_COVERS(G_RB | G_RD | G_BD) ? 3 : M3_COVERS(G_CORNER) ? 6 : M3_COVERS(G_NO_BOX_STATICS) ? 4 : 5;
                // (every active lane adds -- sums and counts alike, the tools take their ratio)
                if (blockIdx.x < 64) {
                    atomicAdd(&g_cyc[blockIdx.x * 16 + cls_], (unsigned int)dt_);
                    atomicAdd(&g_cyc[blockIdx.x * 16 + 8 + cls_], 1u);
                }
            }
#endif
#undef M3_COVERS
        }
    }
}

}  // namespace m3
