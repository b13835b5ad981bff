// Host build of the product's csrc/panda_dyn.hpp for the CPU tests (tests/test_device_dynamics_on_host.py): the device
// code of the panda_env chain, lane by lane, against the oracle -- without a GPU.
//   g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off -Itests/native/shim panda_dyn_host.cpp -o libpanda_dyn_host.so
#include "../../m3p2i_aip_amd/csrc/panda_dyn.hpp"

namespace {
void scene(m3::PandaScene& s, float dt, int substeps) { m3::make_panda_scene(s, dt, substeps); }
// oracle row (58 floats): q9 qd9 | cubeA pos3 quat4 vel3 angvel3 | cubeB (13) | held | rel_p3 rel_q4 | f_table2 f_shelf2 f_cubeB2
void load(const float* w, m3::PandaWorld& p) {
    for (int i = 0; i < 9; ++i) { p.q[i] = w[i]; p.qd[i] = w[9 + i]; }
    for (int i = 0; i < 3; ++i) { p.cube[i] = w[18 + i]; p.cube_v[i] = w[25 + i]; p.cubeB[i] = w[31 + i]; p.rel_p[i] = w[45 + i]; }
    for (int i = 0; i < 4; ++i) { p.cube_q[i] = w[21 + i]; p.rel_q[i] = w[48 + i]; }
    p.held = w[44];
    for (int i = 0; i < 2; ++i) { p.f_table[i] = w[52 + i]; p.f_shelf[i] = w[54 + i]; p.f_cubeB[i] = w[56 + i]; }
}
void store(const m3::PandaWorld& p, float* w) {
    for (int i = 0; i < 9; ++i) { w[i] = p.q[i]; w[9 + i] = p.qd[i]; }
    for (int i = 0; i < 3; ++i) { w[18 + i] = p.cube[i]; w[25 + i] = p.cube_v[i]; w[45 + i] = p.rel_p[i]; }
    for (int i = 0; i < 4; ++i) { w[21 + i] = p.cube_q[i]; w[48 + i] = p.rel_q[i]; }
    w[44] = p.held;
    for (int i = 0; i < 2; ++i) { w[52 + i] = p.f_table[i]; w[54 + i] = p.f_shelf[i]; w[56 + i] = p.f_cubeB[i]; }
}
}  // namespace

// n worlds, one step each with controls u[n][9]; obs[n][10] = left pos3, left quat4, right pos3 (what the costs read).
// mode 0: panda_step<FORCES, !LAZY_FK> (step mode); mode 1: <FORCES, LAZY_FK> (the pick rollout); mode 2: <!FORCES,
// LAZY_FK> (reach / place rollouts: no contact forces formed).  hp / trav: the lazy kinematics' state per world,
// carried by the caller over the steps of a rollout (hp[n][3], trav[n]).
extern "C" void pnh_step(float dt, int substeps, float* worlds, int n, const float* u, float* obs, int mode, float* hp,
                         float* trav) {
    m3::PandaScene sc;
    scene(sc, dt, substeps);
    for (int i = 0; i < n; ++i) {
        m3::PandaWorld p;
        load(worlds + 58 * (long long)i, p);
        m3::PandaObs o;
        if (mode == 0) m3::panda_step<true, false>(sc, p, u + 9 * i, o);
        else if (mode == 1) m3::panda_step<true, true>(sc, p, u + 9 * i, o, hp + 3 * i, trav + i);
        else m3::panda_step<false, true>(sc, p, u + 9 * i, o, hp + 3 * i, trav + i);
        store(p, worlds + 58 * (long long)i);
        for (int j = 0; j < 3; ++j) { obs[10 * i + j] = o.left[j]; obs[10 * i + 7 + j] = o.right[j]; }
        for (int j = 0; j < 4; ++j) obs[10 * i + 3 + j] = o.left_q[j];
    }
}
// world load: the grasp state inferred from the geometry + the hand origin for the lazy kinematics
extern "C" void pnh_infer_held(float dt, int substeps, float* worlds, int n, float* hp) {
    m3::PandaScene sc;
    scene(sc, dt, substeps);
    for (int i = 0; i < n; ++i) {
        m3::PandaWorld p;
        load(worlds + 58 * (long long)i, p);
        m3::panda_infer_held(sc, p, hp + 3 * i);
        store(p, worlds + 58 * (long long)i);
    }
}
