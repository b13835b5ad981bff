// p2p.hip -- device-side exchange of the per-rank records over peer-mapped memory (xGMI between the GPUs of
// one node; gfx950).  SURVEY.md section 5 / 8(e): "every rank writes its slot into every peer's buffer ... 1 hop".
//
// The sharded update needs ONE all-gather of small records per command (m3_internal.hpp: record_length /
// regen_record_length; at most ~33 KB per rank at K = 64000).  As an RCCL collective that is a host-issued call
// (8-14 us of host time, 15-17 us on the stream at world_size 1 -- profiles/r02/collective_overhead_c5.json).  Here
// it is two tiny kernels on the handle's own stream, no library call in between:
//
//   k_p2p_put   workgroup p copies this rank's record into slot [parity][rank] of PEER p's exchange block (plain
//               stores through the peer mapping: one hop over xGMI, or a local copy for p == rank), makes them
//               visible at system scope and then releases flag [parity][rank] of that peer with the exchange's
//               sequence number;
//   k_p2p_wait  lane p of one wavefront acquires flag [parity][p] of the OWN block until it shows the sequence
//               number (bounded: the handle's time-out on the 100 MHz wall clock -- 0.5 s, 30 s for the first
//               exchange --, then the error word is set, the missing rank's slot is filled with NaN -- the plan of
//               that command is NaN, not a silently different one -- and the stream moves on: a peer that died must
//               not hang the GPU; the planner polls the error word, distributed.attach_p2p).
//
// The kernels that follow on the stream (k_mix / k_regen_part / ...) read the records from the own block.  The block
// is allocated uncached (hipDeviceMallocUncached; fine-grained or plain device memory as fallbacks), so neither the
// peers' stores nor the owner's loads can be served from a stale L2 line.  Two parities: a rank can run at most one
// exchange ahead of the slowest one (it needs every peer's flag of exchange n before it can finish it, and a peer
// raises that flag only after its own finalize of exchange n - 1), so slot n & 1 is never overwritten while read.
#include <hip/hip_runtime.h>

#include "m3_internal.hpp"

namespace m3 {

__device__ __forceinline__ void p2p_put_body(const P2PArgs& a, int p) {
    // (slots are rec_stride = rec_len rounded up to 4 floats apart: 16-byte stores -- uncached stores are not
    // combined, a dword per lane was 41 us per put of 38 KB records to 8 peers)
    float* dst = a.peer_data[p] + ((size_t)a.slot * a.n_ranks + a.rank) * a.rec_stride;
    const int n4 = a.rec_len >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(a.rec);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int o = threadIdx.x; o < n4; o += blockDim.x) d4[o] = s4[o];
    for (int o = (n4 << 2) + threadIdx.x; o < a.rec_len; o += blockDim.x) dst[o] = a.rec[o];
    // Order: every data store of the workgroup acknowledged, THEN the flag.  The block is uncached (fine-grained)
    // memory, so the stores are not held in this GPU's L2 and `s_waitcnt vmcnt(0)` is all the release needs; a
    // system-scope fence would also write back the WHOLE L2 (`buffer_wbl2 sc0 sc1`: the rollout's outputs are in
    // there -- measured 110 us per exchange).  Only a block in plain device memory (the last fallback) needs it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.plain_memory) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(a.peer_flags[p] + a.slot * MIX_MAX_RANKS + a.rank, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(1024) void k_p2p_put(const P2PArgs a) { p2p_put_body(a, blockIdx.x); }

__device__ __forceinline__ void p2p_wait_body(const P2PArgs& a) {
    const int p = threadIdx.x;
    if (p < a.n_ranks) {
        const int* f = a.peer_flags[a.rank] + a.slot * MIX_MAX_RANKS + p;
        const unsigned long long t0 = wall_clock64();
        // (relaxed system-scope loads: they bypass the caches; the acquire is the end of this kernel -- the readers
        // of the records are the NEXT kernels on the stream)
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            if (wall_clock64() - t0 > a.timeout_ticks) {   // the peer never arrived: report, do not hang
                atomicExch(a.err, 1 + p);
                // ... and make the miss unmistakable downstream: the slot this rank would have read stale or zeroed
                // data from is filled with NaN, so the plan of THIS command comes out NaN instead of silently
                // diverging from the other ranks' (the sticky error word alone is only seen by m3_p2p_status)
                float* miss = a.peer_data[a.rank] + ((size_t)a.slot * a.n_ranks + p) * a.rec_stride;
                for (int o = 0; o < a.rec_len; ++o) miss[o] = __builtin_nanf("");
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (a.plain_memory) __threadfence_system();
}
__global__ __launch_bounds__(64) void k_p2p_wait(const P2PArgs a) { p2p_wait_body(a); }

// put + wait in ONE launch (m3_p2p_exchange, one process per GPU): workgroups 0 .. n_ranks - 1 put, workgroup n_ranks
// waits -- the wait concerns the PEERS' stores into the own block, not this rank's puts, so it needs no ordering with
// them.  (A process that drives several handles on one stream must use the two separate launches, every put before any
// wait: a wait in front of another handle's put would spin until its time-out.)
__global__ __launch_bounds__(1024) void k_p2p_exchange(const P2PArgs a) {
    if ((int)blockIdx.x < a.n_ranks) p2p_put_body(a, blockIdx.x);
    else if (threadIdx.x < 64) p2p_wait_body(a);
}

void launch_p2p_put(const P2PArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_p2p_put, dim3(a.n_ranks), dim3(1024), 0, s, a);
}
void launch_p2p_wait(const P2PArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, s, a);
}
void launch_p2p_exchange(const P2PArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_p2p_exchange, dim3(a.n_ranks + 1), dim3(1024), 0, s, a);
}

}  // namespace m3
