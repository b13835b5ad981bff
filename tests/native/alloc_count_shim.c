/* LD_PRELOAD interposer for tests/test_no_alloc_in_command_gpu.py: counts every call of the HIP runtime's allocation entry
 * points made by ANY library of the process (libm3p2i_hip.so and torch alike) and forwards it.  The test reads the counter
 * through m3shim_alloc_calls() before and after a run of commands: SURVEY.md 8(b) "no allocation in m3_command".
 * Test infrastructure; not part of the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <stddef.h>
#include <string.h>

static volatile long g_calls = 0;
long m3shim_alloc_calls(void) { return g_calls; }

/* The HIP runtime is loaded by python's dlopen of torch's libraries with RTLD_LOCAL: its symbols are not in the global
 * scope, so dlsym(RTLD_NEXT, ...) does not see them.  Find the loaded libamdhip64 by name and take its handle. */
static void* g_hip = 0;
static int find_hip(struct dl_phdr_info* info, size_t size, void* data) {
    (void)size; (void)data;
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) {
        g_hip = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_LAZY);
        return g_hip != 0;
    }
    return 0;
}
static void* real_sym(const char* name) {
    void* p = dlsym(RTLD_NEXT, name);
    if (p) return p;
    if (!g_hip) dl_iterate_phdr(find_hip, 0);
    return g_hip ? dlsym(g_hip, name) : 0;
}

#define FORWARD(name, proto, args)                                  \
    int name proto {                                                \
        static int (*real) proto = 0;                               \
        if (!real) real = (int (*) proto)real_sym(#name);           \
        __sync_fetch_and_add(&g_calls, 1);                          \
        return real ? real args : 2 /* hipErrorOutOfMemory */;      \
    }

FORWARD(hipMalloc, (void** p, size_t n), (p, n))
FORWARD(hipExtMallocWithFlags, (void** p, size_t n, unsigned f), (p, n, f))
FORWARD(hipHostMalloc, (void** p, size_t n, unsigned f), (p, n, f))
FORWARD(hipHostAlloc, (void** p, size_t n, unsigned f), (p, n, f))
FORWARD(hipMallocManaged, (void** p, size_t n, unsigned f), (p, n, f))
FORWARD(hipMallocAsync, (void** p, size_t n, void* s), (p, n, s))
FORWARD(hipMallocPitch, (void** p, size_t* pitch, size_t w, size_t h), (p, pitch, w, h))
FORWARD(hipFree, (void* p), (p))
FORWARD(hipHostFree, (void* p), (p))
