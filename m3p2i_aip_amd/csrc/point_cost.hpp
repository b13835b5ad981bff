// point_cost.hpp -- device-side task costs of the point_env, evaluated in registers right
// after each physics step (the reference launches 20-150 tiny torch kernels per step here).
//
//   Objective.compute_cost      cost_functions.py:19-36
//   get_navigation_cost         cost_functions.py:38
//   calculate_dist              cost_functions.py:41-50
//   get_push_cost               cost_functions.py:52-60
//   get_pull_cost               cost_functions.py:62-89
//   get_motion_cost             cost_functions.py:158-169 (point_env branch)
//   calculate_suction           skill_utils.py:59-94
#pragma once
#include "planar_dyn.hpp"

namespace m3 {

struct CostParams {
    int task;
    int multi_modal;
    int half_K;        // GLOBAL K / 2
    float goal[7];
    float kp_suction;
    float suction_thresh;  // 1.8 (K > 1) or 1.5 (K == 1), skill_utils.py:75-82
    int avoid_dyn_obs;     // EXTENSION (m3_set_avoid_dyn_obs, default 0 = the reference): push / pull add get_motion_cost
};

__device__ __forceinline__ float clamp500(float v) { return fminf(fmaxf(v, -500.0f), 500.0f); }

// k = GLOBAL sample index.  Writes the pending suction force (acts during the NEXT step,
// cost_functions.py:76) into w.
__device__ __forceinline__ float point_cost(const CostParams& cp, PointWorld& w, int k) {
    const int task = cp.task;
    if (task == 0) {  // navigation
        const float dx = w.rx - cp.goal[0], dy = w.ry - cp.goal[1];
        const float coll = fabsf(w.fcDx) + fabsf(w.fcDy);
        return sqrtf(dx * dx + dy * dy) + ((coll > 0.1f) ? 1000.0f : 0.0f);
    }
    // calculate_dist
    const float r2bx = w.rx - w.B.x, r2by = w.ry - w.B.y;
    const float b2gx = cp.goal[0] - w.B.x, b2gy = cp.goal[1] - w.B.y;
    const float d1 = sqrtf(r2bx * r2bx + r2by * r2by);
    const float d2 = sqrtf(b2gx * b2gx + b2gy * b2gy);
    const float dist_cost = d1 + d2 * 10.0f;
    const float cos_theta = (r2bx * b2gx + r2by * b2gy) / (d1 * d2);
    float push = 0.0f, pull = 0.0f;
    if (task == 1 || task == 3) {
        const float align = (cos_theta > 0.0f) ? cos_theta : 0.0f;
        push = 3.0f * dist_cost + 1.0f * align;
    }
    if (task == 2 || task == 3) {
        const float pdx = w.B.x - w.rx, pdy = w.B.y - w.ry;
        const float rdist = sqrtf(pdx * pdx + pdy * pdy);
        const bool toward = (w.rvx * pdx + w.rvy * pdy) > 0.0f;
        const float mag = 1.0f / rdist;
        const float ux = pdx * mag, uy = pdy * mag;
        const bool mask = mag > cp.suction_thresh;
        float fbx = 0.f, fby = 0.f, frx = 0.f, fry = 0.f;
        if (mask) {
            fbx = clamp500(-cp.kp_suction * ux);
            fby = clamp500(-cp.kp_suction * uy);
            frx = clamp500(cp.kp_suction * ux);
            fry = clamp500(cp.kp_suction * uy);
        }
        if (toward || (cp.multi_modal && k < cp.half_K)) { fbx = fby = frx = fry = 0.0f; }
        w.fBx = fbx; w.fBy = fby; w.fRx = frx; w.fRy = fry;
        const float align = (cos_theta < 0.0f) ? -cos_theta : 0.0f;
        const float vel_cost = (toward && rdist <= 0.5f) ? 0.6f : 0.0f;
        pull = 3.0f * dist_cost + 3.0f * vel_cost + 7.0f * align;
    }
    float c = 0.0f;
    if (task == 1) c = push;
    else if (task == 2) c = pull;
    else if (task == 3) c = (k < cp.half_K) ? push : pull;
    if (cp.avoid_dyn_obs) {   // (general rollout instance / step mode only: the per-task instances compile it away)
        const float coll = fabsf(w.fcDx) + fabsf(w.fcDy);
        c = c + ((coll > 0.1f) ? 1000.0f : 0.0f);
    }
    return c;
}

}  // namespace m3
