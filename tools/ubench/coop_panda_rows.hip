// Microbenchmark for VERDICT r4 item 1: the Panda world's velocity passes (csrc/panda_dyn.hpp, step 5) with a sample's
// joint-space rows run ACROSS LANES instead of serially inside one lane.
//
// Workload = what k_rollout_panda<FORCES = true> does in C4's pick scene per substep: 9 drive rows + 4 gripper contacts
// x 3 rows (friction, friction, normal) in joint space, 6 + 1 Gauss-Seidel passes; 40 substeps = one T = 20 rollout.
//
//   A    lane per sample; a row's 9 Jacobian entries live in a lane-strided LDS store and are re-read on every visit;
//        row velocity = 9 dependent FMAs, impulse application = 9 mul (J * invI, recomputed) + 9 FMA   [the product, round 4]
//   A2   the same with J * invI formed once per substep and kept in a second LDS store (VERDICT's "hoist")
//   B16  16 lanes per sample, lane l owns generalized coordinate l (joints 0-8; lanes 9-14 are where a free target body's
//        6 velocity components go; lane 15 idle): a row is ONE register (its column entry), the row velocity is 1 mul +
//        a 4-step symmetric DPP butterfly (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane of the row ends
//        with the same bits), the impulse application 1 FMA; the 9 drive rows are ONE lane-parallel row
//   B16r the same with J * invM recomputed on every visit (12 fewer registers, one more multiply per visit)
//   B16g B16 + the selects that let lanes 9-14 address one of three free bodies per row (per-sample data)
//   B8   8 lanes per sample (3-step butterfly), 8 coordinates: what a 7 + 1 split of the arm would cost per row
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/coop_panda_rows.hip -o coop_panda_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int L> __device__ __forceinline__ float tree(float v) {   // sum over the L lanes of a sample, same bits in all of them
    v += dppf<DPP_XOR1>(v);
    v += dppf<DPP_XOR2>(v);
    v += dppf<DPP_HALF_MIRROR>(v);
    if (L == 16) v += dppf<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct Sc { float invI[9], rden[9], pmax[9], hD; };
static Sc make_sc() {
    Sc s;
    const float h = 0.005f, inertia[9] = {1.32f, 2.12f, 1.30f, 0.918f, 0.0271f, 0.0366f, 0.0030f, 0.022f, 0.022f};
    const float effort[9] = {87, 87, 87, 87, 12, 12, 12, 20, 20};
    s.hD = h * 600.0f;
    for (int i = 0; i < 9; ++i) { s.invI[i] = 1.0f / inertia[i]; s.rden[i] = 1.0f / (1.0f + s.hD / inertia[i]); s.pmax[i] = h * effort[i]; }
    return s;
}
__device__ __forceinline__ float rnd(unsigned a) {      // [-0.3, 0.3)
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return ((float)(a >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.6f;
}
__device__ __forceinline__ float Jentry(int sample, int s, int r, int j) {
    if (j == 7) return (s == 0) ? rnd(sample * 977u + s * 31u + r * 7u + 100u) : 0.0f;
    if (j == 8) return (s == 1) ? rnd(sample * 977u + s * 31u + r * 7u + 200u) : 0.0f;
    if (j > 8) return 0.0f;
    return rnd(sample * 977u + s * 131u + r * 17u + j);
}

// ---- A / A2: lane per sample -----------------------------------------------------------------------------------------
template <bool HOIST>
__global__ __launch_bounds__(64) void k_lane(const Sc sc, float* out, int nsub, int K) {
    __shared__ float rows[12 * 9 * 64];
    __shared__ float rowsI[HOIST ? 12 * 9 * 64 : 64];
    const int lane = threadIdx.x, sample = blockIdx.x * 64 + lane;
    float* R = rows + lane;
    float* RI = rowsI + lane;
    float meff[12], bias[4], u[9], qd0[9];
    for (int j = 0; j < 9; ++j) { u[j] = rnd(sample * 13u + j + 5000u); qd0[j] = rnd(sample * 29u + j + 7000u); }
    for (int s = 0; s < 4; ++s) {
        bias[s] = -0.05f;
        for (int r = 0; r < 3; ++r) {
            float k = 0.0f;
            for (int j = 0; j < 9; ++j) {
                const float J = Jentry(sample, s, r, j);
                R[((s * 3 + r) * 9 + j) * 64] = J;
                if (HOIST) RI[((s * 3 + r) * 9 + j) * 64] = J * sc.invI[j];
                k = mad(J * sc.invI[j], J, k);
            }
            meff[s * 3 + r] = 1.0f / k;
        }
    }
    float chk = 0.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int sub = 0; sub < nsub; ++sub) {
        float qds[9], pdrv[9], lam[12];
#pragma unroll
        for (int j = 0; j < 9; ++j) { qds[j] = qd0[j] + chk * 1e-6f; pdrv[j] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 12; ++i) lam[i] = 0.0f;
        for (int pass = 0; pass <= 6; ++pass) {
            asm volatile("" ::: "memory");     // the product re-reads its rows on every visit (450 registers hold the world): no caching of the store here either
            if (pass < 6) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const float e = mad(sc.hD, u[i] - qds[i], -pdrv[i]);
                    float dp = e * sc.rden[i];
                    const float p1 = fminf(fmaxf(pdrv[i] + dp, -sc.pmax[i]), sc.pmax[i]);
                    dp = p1 - pdrv[i];
                    pdrv[i] = p1;
                    qds[i] = mad(sc.invI[i], dp, qds[i]);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const int r3 = (rr + 1) % 3;
                    float J[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) J[j] = R[((s * 3 + r3) * 9 + j) * 64];
                    float v = 0.0f;
#pragma unroll
                    for (int j = 0; j < 9; ++j) v = mad(J[j], qds[j], v);
                    float dl = -meff[s * 3 + r3] * (v + ((r3 == 0) ? bias[s] : 0.0f));
                    const float l0 = lam[s * 3 + r3];
                    float l1 = l0 + dl;
                    if (r3 == 0) l1 = fmaxf(l1, 0.0f);
                    else { const float mx = lam[s * 3]; l1 = fminf(fmaxf(l1, -mx), mx); }
                    lam[s * 3 + r3] = l1;
                    dl = l1 - l0;
                    if (HOIST) {
#pragma unroll
                        for (int j = 0; j < 9; ++j) qds[j] = mad(RI[((s * 3 + r3) * 9 + j) * 64], dl, qds[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 9; ++j) qds[j] = mad(J[j] * sc.invI[j], dl, qds[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) chk += qds[j];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (sample < K) out[sample] = chk;
    if (blockIdx.x == 0 && lane == 0) out[K] = (float)(t1 - t0);
}

// ---- B: L lanes per sample ---------------------------------------------------------------------------------------------
template <int L, bool RECOMP, bool GENERIC>
__global__ __launch_bounds__(64) void k_coop(const Sc sc, float* out, int nsub, int K) {
    const int lane = threadIdx.x, l = lane & (L - 1), sample = (blockIdx.x * 64 + lane) / L;
    // lane constants of coordinate l (a table lookup in the product)
    float invM = 0.0f, rden = 0.0f, pmax = 0.0f, u = 0.0f, qd0 = 0.0f;
#pragma unroll
    for (int j = 0; j < 9 && j < L; ++j) if (l == j) { invM = sc.invI[j]; rden = sc.rden[j]; pmax = sc.pmax[j]; }
    if (l < 9) { u = rnd(sample * 13u + l + 5000u); qd0 = rnd(sample * 29u + l + 7000u); }
    float J[12], JI[12], meff[12], bias[4];
    int tb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bias[s] = -0.05f;
        tb[s] = GENERIC ? (int)((sample + s) & 3) - 1 : -1;     // (per-sample data: which free body lanes 9-14 address)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float Jl = Jentry(sample, s, r, l);
            J[s * 3 + r] = Jl;
            JI[s * 3 + r] = Jl * invM;
            meff[s * 3 + r] = 1.0f / tree<L>((Jl * invM) * Jl);
        }
    }
    float chk = 0.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int sub = 0; sub < nsub; ++sub) {
        float V0 = qd0 + chk * 1e-6f, V1 = 0.0f, V2 = 0.0f, pdrv = 0.0f, lam[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) lam[i] = 0.0f;
        for (int pass = 0; pass <= 6; ++pass) {
            if (pass < 6) {     // the 9 drive rows: one lane-parallel row (lanes >= 9: invM = 0, pmax = 0: no-ops)
                const float e = mad(sc.hD, u - V0, -pdrv);
                float dp = e * rden;
                const float p1 = fminf(fmaxf(pdrv + dp, -pmax), pmax);
                dp = p1 - pdrv;
                pdrv = p1;
                V0 = mad(invM, dp, V0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool mB = GENERIC && (l >= 9) && tb[s] == 1, mO = GENERIC && (l >= 9) && tb[s] == 2;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const int r3 = (rr + 1) % 3, i = s * 3 + r3;
                    float Ve = V0;
                    if (GENERIC) { Ve = mB ? V1 : Ve; Ve = mO ? V2 : Ve; }
                    const float v = tree<L>(J[i] * Ve);
                    float dl = -meff[i] * (v + ((r3 == 0) ? bias[s] : 0.0f));
                    const float l0 = lam[i];
                    float l1 = l0 + dl;
                    if (r3 == 0) l1 = fmaxf(l1, 0.0f);
                    else { const float mx = lam[s * 3]; l1 = fminf(fmaxf(l1, -mx), mx); }
                    lam[i] = l1;
                    dl = l1 - l0;
                    const float Vn = RECOMP ? mad(J[i] * invM, dl, Ve) : mad(JI[i], dl, Ve);
                    if (GENERIC) { V0 = (mB || mO) ? V0 : Vn; V1 = mB ? Vn : V1; V2 = mO ? Vn : V2; }
                    else V0 = Vn;
                }
            }
        }
        chk += tree<L>(V0 + V1 + V2);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0 && sample < K) out[sample] = chk;
    if (blockIdx.x == 0 && lane == 0) out[K] = (float)(t1 - t0);
}

static float ms_of(hipEvent_t a, hipEvent_t b) { float m; hipEventElapsedTime(&m, a, b); return m; }

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4000;
    const int nsub = 40;                      // T = 20 steps x 2 substeps
    const Sc sc = make_sc();
    float* d;
    hipMalloc(&d, (size_t)(K + 64) * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; int L; } cfg[6] = {
        {"A lane/sample, rows re-read from LDS, J*invI recomputed (round-4 product)", 1},
        {"A2 lane/sample, J*invI hoisted into a second LDS store", 1},
        {"B16 16 lanes/sample, row = 1 register, DPP butterfly", 16},
        {"B16r = B16 with J*invM recomputed per visit", 16},
        {"B16g = B16 + per-row selects of the free body lanes 9-14 address", 16},
        {"B8 8 lanes/sample (8 coordinates)", 8}};
    printf("{\"K\": %d, \"substeps\": %d, \"passes_per_substep\": 7, \"rows_per_pass\": \"9 drive + 12 contact\", \"results\": [\n", K, nsub);
    for (int m = 0; m < 6; ++m) {
        const int waves = (K * cfg[m].L + 63) / 64;
        float cyc = 0, ms = 0, c0 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (m == 0) hipLaunchKernelGGL((k_lane<false>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            if (m == 1) hipLaunchKernelGGL((k_lane<true>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            if (m == 2) hipLaunchKernelGGL((k_coop<16, false, false>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            if (m == 3) hipLaunchKernelGGL((k_coop<16, true, false>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            if (m == 4) hipLaunchKernelGGL((k_coop<16, false, true>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            if (m == 5) hipLaunchKernelGGL((k_coop<8, false, false>), dim3(waves), dim3(64), 0, 0, sc, d, nsub, K);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            ms = ms_of(e0, e1);
            hipMemcpy(&cyc, d + K, sizeof(float), hipMemcpyDeviceToHost);
            hipMemcpy(&c0, d + 5, sizeof(float), hipMemcpyDeviceToHost);
        }
        printf("  {\"mapping\": \"%s\", \"waves\": %d, \"cycles_per_pass_one_wave\": %.1f, \"kernel_us_for_K\": %.1f, \"checksum_sample5\": %.6g}%s\n",
               cfg[m].name, waves, cyc / (nsub * 7.0f), ms * 1e3, c0, m < 5 ? "," : "");
    }
    printf("], \"note\": \"cycles = __builtin_readcyclecounter ticks of wave 0 over the 40 x 7 passes; kernel time for K samples by HIP events (third launch); "
           "A / A2 compute identical numbers, the B variants the same rows in another summation order (checksums agree to rounding; B8 has 8 coordinates)\"}\n");
    return 0;
}
