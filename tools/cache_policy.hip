// Developer micro-benchmark: which cache-policy bits (sc0 / sc1 / nt of global_load on gfx950) suit a
// matrix stream that is read once per launch?  A bare streaming read (8 loads of 16 B per lane in flight,
// 8 KB per wave, like the product kernel) of a C2-sized buffer (54.5 MB: lives in the Infinity Cache
// between launches) and of a 1 GB buffer, timed with HIP events over back-to-back launches.
// build: hipcc --offload-arch=gfx950 -O3 tools/cache_policy.hip -o tools/bin/cache_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));

#define DEFINE_KERNEL(NAME, MODS)                                                                          \
    __global__ __launch_bounds__(256) void NAME(const v2d *__restrict__ src, double *__restrict__ out,     \
                                                long long total16) {                                       \
        const int lane = threadIdx.x & 63;                                                                 \
        const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);                             \
        const long long p = wave * 512 + lane;                                                             \
        v2d v[8];                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                    \
            const v2d *q = src + (p + 64 * k < total16 ? p + 64 * k : 0);                                  \
            asm volatile("global_load_dwordx4 %0, %1, off " MODS : "=v"(v[k]) : "v"(q) : "memory");        \
        }                                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)"                                                                  \
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), \
                       "+v"(v[7])                                                                          \
                     :                                                                                     \
                     : "memory");                                                                          \
        double acc = 0.0;                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 8; k++) acc += v[k].x * 1.0000001 + v[k].y;                  \
        if (acc == 123456.789) out[wave] = acc;                                                            \
    }

DEFINE_KERNEL(k_plain, "")
DEFINE_KERNEL(k_nt, "nt")
DEFINE_KERNEL(k_sc0, "sc0")
DEFINE_KERNEL(k_sc1, "sc1")
DEFINE_KERNEL(k_sc0sc1, "sc0 sc1")
DEFINE_KERNEL(k_sc0nt, "sc0 nt")
DEFINE_KERNEL(k_sc1nt, "sc1 nt")
DEFINE_KERNEL(k_sc0sc1nt, "sc0 sc1 nt")

typedef void (*kern_t)(const v2d *, double *, long long);

static void run(const char *name, kern_t k, const v2d *src, double *out, long long bytes, int reps) {
    const long long total16 = bytes / 16;
    const unsigned grid = (unsigned)((total16 + 2047) / 2048);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, total16);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, total16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    printf("%-12s %8.1f MB: %8.2f us  %6.0f GB/s\n", name, bytes / 1e6, us, bytes / us / 1e3);
    fflush(stdout);
}

int main() {
    const long long big = 1LL << 30;
    v2d *src;
    double *out;
    hipMalloc(&src, big);
    hipMalloc(&out, 1 << 24);
    hipMemset(src, 0, big);
    const struct { const char *n; kern_t k; } ks[] = {{"plain", k_plain}, {"nt", k_nt}, {"sc0", k_sc0}, {"sc1", k_sc1},
                                                      {"sc0 sc1", k_sc0sc1}, {"sc0 nt", k_sc0nt}, {"sc1 nt", k_sc1nt},
                                                      {"sc0 sc1 nt", k_sc0sc1nt}};
    for (int pass = 0; pass < 2; pass++)
        for (auto &e : ks) run(e.n, e.k, src, out, 54553920, 2000);
    for (auto &e : ks) run(e.n, e.k, src, out, big, 100);
    return 0;
}
