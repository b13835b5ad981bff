// Microbenchmark for the north-star sentence "one wavefront per sample ... lanes cooperating on one
// sample" (VERDICT r1, item 5): the velocity solve of the COMMON contact situation of the push task --
// robot drive (2 rows), ground friction of the box (linear + angular), one robot-box contact (normal +
// tangent) -- in three mappings, timed as cycles per solver pass of ONE resident wavefront (what bounds
// C2: 32 waves on 1024 SIMDs, each a dependent chain) and as kernel time for K = 2000 samples:
//
//   A  lane per sample, Gauss-Seidel over the rows in spec order: the product's own code
//      (planar_dyn.hpp: drive rows, solve_ground_friction<BOXB>, solve<ROBOT, BOXB>)            64 samples / wave
//   B  8 lanes per sample, one GENERIC row per lane (J[8], effective mass, bias, softness, clamp kind),
//      block-Jacobi: every row reads the same velocities, the 8 impulses are applied together through
//      DPP reductions over the 8 lanes.  Needs a different spec: pyramid instead of disc friction (the
//      disc clamp is not a generic row), Jacobi instead of Gauss-Seidel (different numbers, slower
//      convergence per pass)                                                                       8 samples / wave
//   C  the same rows, coloured Gauss-Seidel: {drive x, drive y, friction x, y, angular} | {contact normal}
//      | {contact tangent} -- three dependent phases per pass, each one generic-row evaluation + reduction
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I m3p2i_aip_amd/csrc tools/ubench/coop_rows.hip -o coop_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "planar_dyn.hpp"

using namespace m3;

__device__ __forceinline__ PointScene make_scene() {
    PointScene s;
    const float h = 0.025f;
    s.h = h; s.inv_h = 1.0f / h; s.substeps = 2; s.iters = 6;
    s.gam = 1.0f / (h * 600.0f); s.md = 1.0f / (0.1f + s.gam); s.dmax = 1000.0f * h;
    s.LlinB = ((0.75f * 16.0f) * 9.8f) * h; s.LangB = s.LlinB * (0.3825978f * 0.4f);
    s.LlinD = ((1.0f * 16.0f) * 9.8f) * h; s.LangD = s.LlinD * (0.3825978f * 0.4f);
    return s;
}

// ---- A: the product's rows, lane per sample --------------------------------------------------
__global__ __launch_bounds__(64) void k_lane_gs(float* out, int passes, float ux, float uy) {
    const PointScene sc = make_scene();
    const int i = blockIdx.x * 64 + threadIdx.x;
    Vel v = {0.3f + 1e-3f * i, 0.1f, 0.05f, 0.02f, 0.01f, 0.f, 0.f, 0.f};
    Slot c;
    prepare<ROBOT, BOXB>(sc, c, 0.0f, 1.0f, 0.f, 0.f, 0.05f + 1e-4f * i, -0.2f, -0.002f);
    float ldx = 0.f, ldy = 0.f;
    Fric fB = {0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
        float dl = -(((v.rvx - ux) + sc.gam * ldx) * sc.md);
        float l1 = clamp_sym(ldx + dl, sc.dmax);
        v.rvx += sc.invm_r * (l1 - ldx); ldx = l1;
        dl = -(((v.rvy - uy) + sc.gam * ldy) * sc.md);
        l1 = clamp_sym(ldy + dl, sc.dmax);
        v.rvy += sc.invm_r * (l1 - ldy); ldy = l1;
        solve_ground_friction<BOXB>(sc, v, fB, sc.box_m, sc.box_I, sc.LlinB, sc.LangB);
        if (c.on) solve<ROBOT, BOXB>(sc, v, c, sc.mu_rb);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[i] = v.rvx + v.rvy + v.bvx + v.bvy + v.bw + c.ln + c.lt + fB.lx;
    if (i == 0) out[gridDim.x * 64] = (float)(t1 - t0);
}

// ---- A2: the same rows as ONE straight-line block per pass (no exec-mask regions: the rest test of the
// friction row becomes a select, the contact row is unconditional), so that the scheduler may interleave the
// independent drive / friction chains.  Same arithmetic per row => same numbers as A.
__device__ __forceinline__ void friction_branch_free(const PointScene& sc, Vel& v, Fric& f, float m, float I, float Llin, float Lang) {
    const bool moving = ((__float_as_uint(v.bvx) | __float_as_uint(v.bvy) | __float_as_uint(v.bw)) << 1) != 0u;
    float nlx = f.lx + (-m * v.bvx), nly = f.ly + (-m * v.bvy);
    const float mag2 = nlx * nlx + nly * nly;
    const float scl = Llin * spec_rsqrt(mag2);
    const bool sat = mag2 > Llin * Llin;
    nlx = sat ? nlx * scl : nlx;
    nly = sat ? nly * scl : nly;
    float nla = clamp_sym(f.la + (-I * v.bw), Lang);
    const float dvx = sc.invm_b * (nlx - f.lx), dvy = sc.invm_b * (nly - f.ly), dw = sc.invI_b * (nla - f.la);
    v.bvx = moving ? v.bvx + dvx : v.bvx;
    v.bvy = moving ? v.bvy + dvy : v.bvy;
    v.bw = moving ? v.bw + dw : v.bw;
    f.lx = moving ? nlx : f.lx; f.ly = moving ? nly : f.ly; f.la = moving ? nla : f.la;
}
template <bool INTERLEAVE>
__global__ __launch_bounds__(64) void k_lane_gs_flat(float* out, int passes, float ux, float uy) {
    const PointScene sc = make_scene();
    const int i = blockIdx.x * 64 + threadIdx.x;
    Vel v = {0.3f + 1e-3f * i, 0.1f, 0.05f, 0.02f, 0.01f, 0.f, 0.f, 0.f};
    Slot c;
    prepare<ROBOT, BOXB>(sc, c, 0.0f, 1.0f, 0.f, 0.f, 0.05f + 1e-4f * i, -0.2f, -0.002f);
    float ldx = 0.f, ldy = 0.f;
    Fric fB = {0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
        if (INTERLEAVE) {
            // friction first in program order, drive rows after it: they share no variable, so the order is
            // free; the scheduler sees both chains in one block
            friction_branch_free(sc, v, fB, sc.box_m, sc.box_I, sc.LlinB, sc.LangB);
        }
        float dl = -(((v.rvx - ux) + sc.gam * ldx) * sc.md);
        float l1 = clamp_sym(ldx + dl, sc.dmax);
        v.rvx += sc.invm_r * (l1 - ldx); ldx = l1;
        dl = -(((v.rvy - uy) + sc.gam * ldy) * sc.md);
        l1 = clamp_sym(ldy + dl, sc.dmax);
        v.rvy += sc.invm_r * (l1 - ldy); ldy = l1;
        if (!INTERLEAVE) friction_branch_free(sc, v, fB, sc.box_m, sc.box_I, sc.LlinB, sc.LangB);
        solve<ROBOT, BOXB>(sc, v, c, sc.mu_rb);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[i] = v.rvx + v.rvy + v.bvx + v.bvy + v.bw + c.ln + c.lt + fB.lx;
    if (i == 0) out[gridDim.x * 64] = (float)(t1 - t0);
}

// ---- B / C: 8 lanes per sample, one generic row per lane ----------------------------------------
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum8(float v) {   // sum over the 8 lanes of a sample, result in all 8
    v += dppf<DPP_XOR1>(v);
    v += dppf<DPP_XOR2>(v);
    v += dppf<DPP_HALF_MIRROR>(v);
    return v;
}
enum { CL_SYM = 0, CL_LOW0 = 1, CL_CONE = 2 };

template <bool JACOBI>
__global__ __launch_bounds__(64) void k_coop(float* out, int passes, float ux, float uy) {
    const PointScene sc = make_scene();
    const int lane = threadIdx.x, row = lane & 7, sample = (blockIdx.x * 64 + lane) >> 3;
    // velocities of the sample, replicated in its 8 lanes: rvx rvy bvx bvy bw (dvx dvy dw unused here)
    float v[5] = {0.3f + 1e-3f * sample, 0.1f, 0.05f, 0.02f, 0.01f};
    const float minv[5] = {sc.invm_r, sc.invm_r, sc.invm_b, sc.invm_b, sc.invI_b};
    // row tables (what detect/prepare would leave per lane): rows 0,1 drive x,y; 2,3,4 box friction x, y,
    // angular (pyramid); 5 contact normal; 6 contact tangent; 7 idle
    const float nx = 0.0f, ny = 1.0f, rbx = 0.05f + 1e-4f * sample, rby = -0.2f;
    const float rnb = rbx * ny - rby * nx, rtb = rbx * nx + rby * ny;
    float J[5] = {0, 0, 0, 0, 0}, meff = 0.f, bias = 0.f, soft = 0.f, lim = 0.f, lam = 0.f;
    int kind = CL_SYM, colour = 0;
    if (row == 0) { J[0] = 1.f; meff = sc.md; soft = sc.gam; lim = sc.dmax; bias = -ux; }
    if (row == 1) { J[1] = 1.f; meff = sc.md; soft = sc.gam; lim = sc.dmax; bias = -uy; }
    if (row == 2) { J[2] = 1.f; meff = sc.box_m; lim = sc.LlinB; }
    if (row == 3) { J[3] = 1.f; meff = sc.box_m; lim = sc.LlinB; }
    if (row == 4) { J[4] = 1.f; meff = sc.box_I; lim = sc.LangB; }
    if (row == 5) { J[0] = -nx; J[1] = -ny; J[2] = nx; J[3] = ny; J[4] = rnb; kind = CL_LOW0; colour = 1; bias = -0.002f * sc.inv_h;
                    meff = 1.0f / (sc.invm_r + sc.invm_b + sc.invI_b * rnb * rnb); }
    if (row == 6) { J[0] = ny; J[1] = -nx; J[2] = -ny; J[3] = nx; J[4] = rtb; kind = CL_CONE; colour = 2;
                    meff = 1.0f / (sc.invm_r + sc.invm_b + sc.invI_b * rtb * rtb); }
    if (row == 7) colour = 3;
    float Jm[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) Jm[q] = minv[q] * J[q];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
#pragma unroll
        for (int ph = 0; ph < (JACOBI ? 1 : 3); ++ph) {
            float r = J[0] * v[0];
#pragma unroll
            for (int q = 1; q < 5; ++q) r = r + J[q] * v[q];
            float nl = lam - (r + bias + soft * lam) * meff;
            // the tangent row's limit is mu * lambda of the normal row (one lane to the left)
            const float lnorm = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(lam), 0x111 /* row_shr:1 */, 0xF, 0xF, false));
            const float hi = (kind == CL_CONE) ? sc.mu_rb * lnorm : lim;
            nl = (kind == CL_LOW0) ? fmaxf(nl, 0.0f) : __builtin_amdgcn_fmed3f(nl, -hi, hi);
            float dlam = nl - lam;
            if (!JACOBI && colour != ph) dlam = 0.0f;     // coloured Gauss-Seidel: only this phase's rows act
            lam = lam + dlam;
#pragma unroll
            for (int q = 0; q < 5; ++q) v[q] += sum8(Jm[q] * dlam);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (row == 0) out[sample] = v[0] + v[1] + v[2] + v[3] + v[4] + lam;
    if (blockIdx.x == 0 && lane == 0) out[gridDim.x * 8] = (float)(t1 - t0);
}

static float ms_of(hipEvent_t a, hipEvent_t b) { float m; hipEventElapsedTime(&m, a, b); return m; }

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    const int passes = 6 * 2 * 30;   // the solver passes of one C2 rollout (T = 30 steps x 2 substeps x 6)
    float* d;
    hipMalloc(&d, (size_t)(K + 4096) * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; int lanes_per_sample; } cfg[5] = {{"A lane/sample, Gauss-Seidel (product rows)", 1},
                                                                {"B 8 lanes/sample, block-Jacobi generic rows", 8},
                                                                {"C 8 lanes/sample, coloured Gauss-Seidel", 8},
                                                                {"A2 lane/sample, rows as one straight-line block (selects instead of exec-mask regions)", 1},
                                                                {"A3 = A2 with the friction row ahead of the drive rows", 1}};
    printf("{\"K\": %d, \"passes\": %d, \"results\": [\n", K, passes);
    for (int m = 0; m < 5; ++m) {
        const int waves = (K * cfg[m].lanes_per_sample + 63) / 64;
        float cyc = 0, ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (m == 0) hipLaunchKernelGGL(k_lane_gs, dim3(waves), dim3(64), 0, 0, d, passes, 1.0f, -0.5f);
            if (m == 1) hipLaunchKernelGGL(k_coop<true>, dim3(waves), dim3(64), 0, 0, d, passes, 1.0f, -0.5f);
            if (m == 2) hipLaunchKernelGGL(k_coop<false>, dim3(waves), dim3(64), 0, 0, d, passes, 1.0f, -0.5f);
            if (m == 3) hipLaunchKernelGGL(k_lane_gs_flat<false>, dim3(waves), dim3(64), 0, 0, d, passes, 1.0f, -0.5f);
            if (m == 4) hipLaunchKernelGGL(k_lane_gs_flat<true>, dim3(waves), dim3(64), 0, 0, d, passes, 1.0f, -0.5f);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            ms = ms_of(e0, e1);
            hipMemcpy(&cyc, d + (cfg[m].lanes_per_sample == 1 ? waves * 64 : waves * 8), sizeof(float), hipMemcpyDeviceToHost);
        }
        printf("  {\"mapping\": \"%s\", \"waves\": %d, \"cycles_per_pass_one_wave\": %.1f, \"kernel_us_for_K\": %.1f}%s\n",
               cfg[m].name, waves, cyc / passes, ms * 1e3, m < 4 ? "," : "");
    }
    printf("], \"note\": \"clocks = __builtin_readcyclecounter ticks (shader clock: 360 passes x 378 ticks = 60 us); B and C use pyramid friction (and B is Jacobi): other numbers than A; A2 / A3 compute the same numbers as A\"}\n");
    return 0;
}
