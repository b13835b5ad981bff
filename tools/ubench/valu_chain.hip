// Microbenchmark: issue rate of ONE wavefront for dependent vs independent VALU chains
// (fma, mul+add pairs, v_rcp, v_sqrt, v_cndmask after v_cmp) -- decides whether restructuring the
// rollout kernel for instruction-level parallelism can pay.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 dependent fma
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = __builtin_fmaf(x0, a, b);
        } else if (MODE == 1) {  // 8 independent fma
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 2) {  // 8 dependent rcp
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = __builtin_amdgcn_rcpf(x0) + a;
        } else if (MODE == 3) {  // 8 independent rcp
            x0 = __builtin_amdgcn_rcpf(x0) + a; x1 = __builtin_amdgcn_rcpf(x1) + a; x2 = __builtin_amdgcn_rcpf(x2) + a; x3 = __builtin_amdgcn_rcpf(x3) + a;
            x4 = __builtin_amdgcn_rcpf(x4) + a; x5 = __builtin_amdgcn_rcpf(x5) + a; x6 = __builtin_amdgcn_rcpf(x6) + a; x7 = __builtin_amdgcn_rcpf(x7) + a;
        } else if (MODE == 4) {  // 8 dependent cmp+select
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = (x0 > b) ? x0 * a : x0 + a;
        } else if (MODE == 5) {  // dependent IEEE divide + sqrt (the friction clamp)
#pragma unroll
            for (int u = 0; u < 2; ++u) x0 = a / __builtin_sqrtf(x0 * x0 + b);
        } else if (MODE == 6) {  // two independent IEEE divide + sqrt chains
            x0 = a / __builtin_sqrtf(x0 * x0 + b); x1 = a / __builtin_sqrtf(x1 * x1 + b);
        } else if (MODE == 7) {  // divergent-if region (half the lanes) around 4 fma
            if (x0 > b) {
#pragma unroll
                for (int u = 0; u < 4; ++u) x0 = __builtin_fmaf(x0, a, b);
            }
            x0 = x0 * a;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0);
}

int main() {
    float* d; hipMalloc(&d, 65 * sizeof(float));
    const int iters = 100000;
    const char* names[] = {"8 dependent fma", "8 independent fma", "8 dependent rcp+add", "8 independent rcp+add",
                           "8 dependent cmp+select", "2 dependent div/sqrt", "2 independent div/sqrt", "if-region + 4 fma"};
    for (int m = 0; m < 8; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, iters, 0.999f, 0.001f); break;
                case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, iters, 0.999f, 0.001f); break;
                case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, iters, 0.5f, 0.001f); break;
                case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, iters, 0.5f, 0.001f); break;
                case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, iters, 0.999f, 0.03f); break;
                case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, d, iters, 0.999f, 0.001f); break;
                case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, d, iters, 0.999f, 0.001f); break;
                case 7: hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, d, iters, 0.999f, 0.03f); break;
            }
            hipDeviceSynchronize();
        }
        float h[65]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s %8.1f cycles/iteration (s_memtime units)\n", names[m], h[64] / iters);
    }
    return 0;
}
