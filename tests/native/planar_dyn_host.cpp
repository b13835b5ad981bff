// Host build of the product's csrc/planar_dyn.hpp for the CPU tests (tests/test_device_dynamics_on_host.py): the device
// code of the point_env dynamics, lane by lane, against the oracle -- without a GPU.
//   g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off -Itests/native/shim planar_dyn_host.cpp -o libplanar_dyn_host.so
#include "../../m3p2i_aip_amd/csrc/planar_dyn.hpp"

namespace {
void scene(m3::PointScene& s, float dt, int substeps, int iters) { m3::make_point_scene(s, dt, substeps, iters); }
void load(const float* w, m3::PointWorld& p) {   // oracle row (31 floats): 3 bodies x (x y c s vx vy w) | fext R, B | fc R, B, D
    p.rx = w[0]; p.ry = w[1]; p.rvx = w[4]; p.rvy = w[5];
    p.B = {w[7], w[8], w[9], w[10], w[11], w[12], w[13]};
    p.D = {w[14], w[15], w[16], w[17], w[18], w[19], w[20]};
    p.fRx = w[21]; p.fRy = w[22]; p.fBx = w[23]; p.fBy = w[24];
    p.fcRx = w[25]; p.fcRy = w[26]; p.fcBx = w[27]; p.fcBy = w[28]; p.fcDx = w[29]; p.fcDy = w[30];
}
void store(const m3::PointWorld& p, float* w, bool all_forces) {
    w[0] = p.rx; w[1] = p.ry; w[4] = p.rvx; w[5] = p.rvy;
    const m3::Box* b[2] = {&p.B, &p.D};
    for (int i = 0; i < 2; ++i) {
        float* o = w + 7 + 7 * i;
        o[0] = b[i]->x; o[1] = b[i]->y; o[2] = b[i]->c; o[3] = b[i]->s; o[4] = b[i]->vx; o[5] = b[i]->vy; o[6] = b[i]->w;
    }
    w[21] = p.fRx; w[22] = p.fRy; w[23] = p.fBx; w[24] = p.fBy;
    w[29] = p.fcDx; w[30] = p.fcDy;
    if (all_forces) { w[25] = p.fcRx; w[26] = p.fcRy; w[27] = p.fcBx; w[28] = p.fcBy; }
}
}  // namespace

// n worlds (rows of 31 floats, the oracle's layout), one step each with controls u[n][2].
// mode 0: point_step<true> (the step-mode kernel's path: general instance, all bodies' contact forces);
// mode 1: point_step<false> (the rollout's path: instance dispatch by the broad-phase mask, dyn-obs force only).
extern "C" void pdh_step(float dt, int substeps, int iters, float* worlds, int n, const float* u, int mode) {
    m3::PointScene sc;
    scene(sc, dt, substeps, iters);
    for (int i = 0; i < n; ++i) {
        m3::PointWorld p;
        load(worlds + 31 * (long long)i, p);
        if (mode == 0) m3::point_step<true>(sc, p, u[2 * i], u[2 * i + 1]);
        else m3::point_step<false>(sc, p, u[2 * i], u[2 * i + 1], true);
        store(p, worlds + 31 * (long long)i, mode == 0);
    }
}
