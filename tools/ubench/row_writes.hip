// Micro-benchmark: HBM write traffic of per-lane row writes.  Each lane owns a row of T x 16 B and
// writes one 16 B piece per "time step" with ~1 us of dependent ALU work in between (as the rollout
// does).  mode 0: rows contiguous per lane (sample-major, [K][T][4]); mode 1: time-major [T][K][4]
// with a scattered lane -> sample permutation; mode 2: time-major, identity (coalesced).
// Build: hipcc --offload-arch=gfx950 -O3 -o row_writes row_writes.hip ; run under
//   rocprofv3 --pmc WRITE_SIZE -- ./row_writes <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

__global__ void k_rows(float4* out, const int* perm, int K, int T, int mode, int spin) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= K) return;
    const int i = mode == 1 ? perm[slot] : slot;
    float x = (float)i * 1e-3f;
    for (int t = 0; t < T; ++t) {
        for (int s = 0; s < spin; ++s) x = x * 1.0001f + 0.5f;   // dependent chain
        const size_t o = mode == 0 ? (size_t)i * T + t : (size_t)t * K + i;
        out[o] = make_float4(x, x + 1.f, x + 2.f, x + 3.f);
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, K = argc > 2 ? atoi(argv[2]) : 2000, T = 30;
    float4* out; int* perm;
    hipMalloc(&out, sizeof(float4) * K * T);
    hipMalloc(&perm, sizeof(int) * K);
    std::vector<int> p(K);
    std::iota(p.begin(), p.end(), 0);
    std::shuffle(p.begin(), p.end(), std::mt19937(1));
    hipMemcpy(perm, p.data(), sizeof(int) * K, hipMemcpyHostToDevice);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_rows, dim3((K + 63) / 64), dim3(64), 0, 0, out, perm, K, T, mode, 300);
    hipDeviceSynchronize();
    printf("mode %d K %d useful bytes per launch %zu\n", mode, K, sizeof(float4) * (size_t)K * T);
    return 0;
}
