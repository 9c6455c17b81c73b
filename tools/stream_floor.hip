// Developer micro-benchmark: the practical floor of a WARM streaming read of a C2-sized operator
// (54.5 MB, resident in the Infinity Cache between launches) as a function of the grid shape, the
// bytes each wave owns and the number of 16-byte loads a lane keeps in flight.  Times come from HIP
// events around 200 back-to-back launches (the same protocol as bench.py's eager mode).
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_floor.hip -o /tmp/stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));

// each wave streams `per_wave` consecutive bytes, L loads of 16 B per lane per iteration
template <int L>
__global__ __launch_bounds__(256) void stream_kernel(const v2d *__restrict__ src, double *__restrict__ out,
                                                     long long per_wave16, long long total16) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    long long lo = wave * per_wave16, hi = lo + per_wave16;
    if (hi > total16) hi = total16;
    double acc = 0.0;
    for (long long p = lo + lane; p < hi; p += 64 * L) {
        v2d v[L];
#pragma unroll
        for (int k = 0; k < L; k++) {
            long long q = p + 64 * k;
            v[k] = q < hi ? __builtin_nontemporal_load(&src[q]) : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < L; k++) acc += v[k].x * 1.0000001 + v[k].y;
    }
    if (acc == 123456.789) out[wave] = acc;
}

// "panel-shaped" variant: only M of the 64 lanes load (a 36-row panel: 576-byte wave-loads at a
// 576-byte stride, cache-line-misaligned), optionally after a dependent descriptor fetch (the
// wave's byte offset comes from a table), optionally followed by an LDS round trip per load
template <int M, bool DESC, bool LDSRT, bool NT = false>
__global__ __launch_bounds__(256) void panel_like_kernel(const v2d *__restrict__ src, double *__restrict__ out,
                                                          const long long *__restrict__ table, long long loads_per_wave,
                                                          long long total16) {
    __shared__ double stage[4][64];
    const int lane = threadIdx.x & 63;
    const int w4 = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + w4;
    long long base = wave * loads_per_wave * M;  // 16-byte units
    if (DESC) base = table[wave];
    double acc = 0.0;
    for (long long l0 = 0; l0 < loads_per_wave; l0 += 8) {
        v2d v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long q = base + (l0 + k) * M + lane;
            v[k] = (lane < M && l0 + k < loads_per_wave && q < total16) ? (NT ? __builtin_nontemporal_load(&src[q]) : src[q]) : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            double t = v[k].x * 1.0000001 + v[k].y;
            if (LDSRT) {
                stage[w4][lane] = t;
                t = stage[w4][lane ^ 1];
            }
            acc += t;
        }
    }
    if (acc == 123456.789) out[wave] = acc;
}

template <int M, bool DESC, bool LDSRT, bool NT = false>
void run_panel_like(const v2d *src, double *out, long long *table, long long bytes, long long loads_per_wave) {
    const long long total16 = bytes / 16;
    const long long per_wave16 = loads_per_wave * M;
    const long long nwaves = (total16 + per_wave16 - 1) / per_wave16;
    const unsigned grid = (unsigned)((nwaves + 3) / 4);
    if (DESC) {
        long long *h = (long long *)malloc(sizeof(long long) * grid * 4);
        for (long long w = 0; w < (long long)grid * 4; w++) h[w] = w * per_wave16;
        hipMemcpy(table, h, sizeof(long long) * grid * 4, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL((panel_like_kernel<M, DESC, LDSRT, NT>), dim3(grid), dim3(256), 0, 0, src, out, table, loads_per_wave, total16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 2000; i++)
        hipLaunchKernelGGL((panel_like_kernel<M, DESC, LDSRT, NT>), dim3(grid), dim3(256), 0, 0, src, out, table, loads_per_wave, total16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / 2000;
    printf("panel-like nt=%d M=%2d desc=%d ldsrt=%d loads/wave=%3lld grid=%5u: %7.2f us  %6.0f GB/s\n", (int)NT, M, (int)DESC, (int)LDSRT,
           loads_per_wave, grid, us, bytes / us / 1e3);
    fflush(stdout);
}

// progressively closer to the product kernel: descriptor hop, M-lane wave-loads, then
//   XG : gather ncols = 2*loads x entries into LDS before the matrix loads (dependent on the descriptor)
//   FM : read x back from LDS as 16-byte broadcasts and do the 2 FMAs per load
//   ST : cross-wave combine through LDS (+ barrier) and a y store of M rows by the lead wave
template <int M, bool XG, bool FM, bool ST>
__global__ __launch_bounds__(256) void product_like_kernel(const v2d *__restrict__ src, double *__restrict__ y,
                                                            const double *__restrict__ x, const long long *__restrict__ table,
                                                            long long loads_per_wave, long long total16) {
    __shared__ __attribute__((aligned(16))) double xs[4][512];
    const int lane = threadIdx.x & 63;
    const int w4 = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + w4;
    const long long base = table[wave];
    const int ncols = (int)(2 * loads_per_wave);
    if (XG) {
        const long long col0 = (base / M) % 90000;  // "column start" derived from the descriptor
        for (int c = lane; c < ncols + 16; c += 64) xs[w4][c] = c < ncols ? x[col0 + c] : 0.0;
    }
    double acc0 = 0.0, acc1 = 0.0;
    for (long long l0 = 0; l0 < loads_per_wave; l0 += 8) {
        v2d v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long q = base + (l0 + k) * M + lane;
            v[k] = (lane < M && l0 + k < loads_per_wave && q < total16) ? __builtin_nontemporal_load(&src[q]) : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (FM) {
                const v2d xv = *reinterpret_cast<const v2d *>(&xs[w4][2 * (l0 + k)]);
                acc0 = __builtin_fma(v[k].x, xv.x, acc0);
                acc1 = __builtin_fma(v[k].y, xv.y, acc1);
            } else {
                acc0 += v[k].x;
                acc1 += v[k].y;
            }
        }
    }
    double u = acc0 + acc1;
    if (ST) {
        xs[w4][lane] = u;
        __syncthreads();
        if (w4 == 0) {
            u += xs[1][lane] + xs[2][lane] + xs[3][lane];
            if (lane < M) y[(wave / 4) * 64 % 90000 + lane] = u;
        }
    } else if (u == 123456.789) {
        y[wave] = u;
    }
}

template <int M, bool XG, bool FM, bool ST>
void run_product_like(const v2d *src, double *y, const double *x, long long *table, long long bytes, long long loads_per_wave) {
    const long long total16 = bytes / 16;
    const long long per_wave16 = loads_per_wave * M;
    const long long nwaves = (total16 + per_wave16 - 1) / per_wave16;
    const unsigned grid = (unsigned)((nwaves + 3) / 4);
    long long *h = (long long *)malloc(sizeof(long long) * grid * 4);
    for (long long w = 0; w < (long long)grid * 4; w++) h[w] = w * per_wave16;
    hipMemcpy(table, h, sizeof(long long) * grid * 4, hipMemcpyHostToDevice);
    free(h);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL((product_like_kernel<M, XG, FM, ST>), dim3(grid), dim3(256), 0, 0, src, y, x, table, loads_per_wave, total16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 2000; i++)
        hipLaunchKernelGGL((product_like_kernel<M, XG, FM, ST>), dim3(grid), dim3(256), 0, 0, src, y, x, table, loads_per_wave, total16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / 2000;
    printf("product-like M=%2d xgather=%d fma=%d store=%d loads/wave=%3lld grid=%5u: %7.2f us  %6.0f GB/s\n", M, (int)XG, (int)FM,
           (int)ST, loads_per_wave, grid, us, bytes / us / 1e3);
    fflush(stdout);
}

// grid-stride variant: a fixed number of waves, wave w takes chunks w, w + nwaves, ... of `chunk16`
template <int L>
__global__ __launch_bounds__(256) void stride_kernel(const v2d *__restrict__ src, double *__restrict__ out,
                                                     long long chunk16, long long total16) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    double acc = 0.0;
    for (long long lo = wave * chunk16; lo < total16; lo += nwaves * chunk16) {
        long long hi = lo + chunk16;
        if (hi > total16) hi = total16;
        for (long long p = lo + lane; p < hi; p += 64 * L) {
            v2d v[L];
#pragma unroll
            for (int k = 0; k < L; k++) {
                long long q = p + 64 * k;
                v[k] = q < hi ? __builtin_nontemporal_load(&src[q]) : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int k = 0; k < L; k++) acc += v[k].x * 1.0000001 + v[k].y;
        }
    }
    if (acc == 123456.789) out[wave] = acc;
}

template <int L>
float run_stride(const v2d *src, double *out, long long bytes, long long chunk_bytes, unsigned grid) {
    const long long total16 = bytes / 16, chunk16 = chunk_bytes / 16;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((stride_kernel<L>), dim3(grid), dim3(256), 0, 0, src, out, chunk16, total16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((stride_kernel<L>), dim3(grid), dim3(256), 0, 0, src, out, chunk16, total16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / 200;
    printf("stride L=%d chunk=%6lld B grid=%5u : %7.2f us  %6.0f GB/s\n", L, chunk_bytes, grid, us, bytes / us / 1e3);
    fflush(stdout);
    return (float)us;
}

template <int L>
float run(const v2d *src, double *out, long long bytes, long long per_wave_bytes, int block, int lds = 0) {
    const long long total16 = bytes / 16, per_wave16 = per_wave_bytes / 16;
    const long long nwaves = (total16 + per_wave16 - 1) / per_wave16;
    const int wpb = block / 64;
    const unsigned grid = (unsigned)((nwaves + wpb - 1) / wpb);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((stream_kernel<L>), dim3(grid), dim3(block), lds, 0, src, out, per_wave16, total16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((stream_kernel<L>), dim3(grid), dim3(block), lds, 0, src, out, per_wave16, total16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / 200;
    printf("L=%d block=%4d lds=%6d per_wave=%7lld B grid=%6u waves=%7lld : %7.2f us  %6.0f GB/s\n", L, block, lds,
           per_wave_bytes, grid, nwaves, us, bytes / us / 1e3);
    fflush(stdout);
    return (float)us;
}

int main(int argc, char **argv) {
    const long long maxbytes = 1000000000LL;
    v2d *src;
    double *out;
    hipMalloc(&src, maxbytes + 4096);
    hipMalloc(&out, 1 << 22);
    hipMemset(src, 0, maxbytes + 4096);
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const long long bytes = 54553920LL;
    if (mode == 0) {
        for (long long pw : {4096LL, 8192LL, 16384LL, 32768LL, 65536LL, 131072LL}) {
            run<4>(src, out, bytes, pw, 256);
            run<8>(src, out, bytes, pw, 256);
            run<16>(src, out, bytes, pw, 256);
        }
        for (unsigned grid : {256u, 512u, 1024u, 1536u, 2048u})
            for (long long ch : {1024LL, 4096LL, 8192LL}) {
                run_stride<1>(src, out, bytes, ch, grid);
                run_stride<4>(src, out, bytes, ch, grid);
            }
    } else if (mode == 6) {
        // HBM-streaming ceiling: operators far larger than the Infinity Cache
        long long *table;
        hipMalloc(&table, 64 << 20);
        for (long long b : {950000000LL, 500000000LL}) {
            run<8>(src, out, b, 8192, 256);
            run<8>(src, out, b, 32768, 256);
            run<8>(src, out, b, 131072, 256);
            run_panel_like<64, false, false, false>(src, out, table, b, 8);
            run_panel_like<64, false, false, true>(src, out, table, b, 8);
            run_panel_like<64, false, false, true>(src, out, table, b, 64);
        }
    } else if (mode == 5) {
        long long *table;
        double *xv, *yv;
        hipMalloc(&table, 8 << 20);
        hipMalloc(&xv, 1 << 20);
        hipMalloc(&yv, 1 << 20);
        hipMemset(xv, 0, 1 << 20);
        for (int rep = 0; rep < 2; rep++) {
            run_product_like<64, false, false, false>(src, yv, xv, table, bytes, 8);
            run_product_like<64, true, false, false>(src, yv, xv, table, bytes, 8);
            run_product_like<64, true, true, false>(src, yv, xv, table, bytes, 8);
            run_product_like<64, true, true, true>(src, yv, xv, table, bytes, 8);
            run_product_like<36, false, false, false>(src, yv, xv, table, bytes, 8);
            run_product_like<36, true, true, true>(src, yv, xv, table, bytes, 8);
            run_product_like<36, true, true, true>(src, yv, xv, table, bytes, 16);
            run_product_like<36, true, true, true>(src, yv, xv, table, bytes, 24);
            run_product_like<64, true, true, true>(src, yv, xv, table, bytes, 16);
        }
    } else if (mode == 4) {
        long long *table;
        hipMalloc(&table, 8 << 20);
        // steady state first (the first ~100 ms of a process run faster)
        for (int i = 0; i < 3; i++) run<8>(src, out, bytes, 8192, 256);
        run_panel_like<64, false, false, false>(src, out, table, bytes, 8);
        run_panel_like<64, false, false, true>(src, out, table, bytes, 8);
        run_panel_like<64, true, false, false>(src, out, table, bytes, 8);
        run_panel_like<64, true, false, true>(src, out, table, bytes, 8);
        run_panel_like<36, false, false, false>(src, out, table, bytes, 8);
        run_panel_like<36, false, false, true>(src, out, table, bytes, 8);
        run_panel_like<36, true, false, true>(src, out, table, bytes, 8);
        run_panel_like<36, true, false, true>(src, out, table, bytes, 16);
        run_panel_like<20, true, false, true>(src, out, table, bytes, 8);
        run_panel_like<64, false, false, false>(src, out, table, bytes, 8);
        run_panel_like<64, false, false, true>(src, out, table, bytes, 8);
        run<8>(src, out, bytes, 8192, 256);
    } else if (mode == 3) {
        // cold: a 512 MiB write between launches evicts the operator from L2 / Infinity Cache
        char *flush;
        hipMalloc(&flush, 512LL << 20);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (long long pw : {4096LL, 8192LL, 16384LL, 32768LL}) {
            const long long total16 = bytes / 16, per_wave16 = pw / 16;
            const unsigned grid = (unsigned)(((total16 + per_wave16 - 1) / per_wave16 + 3) / 4);
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 12; rep++) {
                hipMemsetAsync(flush, rep, 512LL << 20, 0);
                hipEventRecord(a, 0);
                hipLaunchKernelGGL((stream_kernel<8>), dim3(grid), dim3(256), 0, 0, src, out, per_wave16, total16);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            printf("cold L=8 per_wave=%6lld: mean %.2f us, min %.2f us (%.0f GB/s)\n", pw, sum * 100.0, best * 1e3,
                   bytes / (sum * 100.0) / 1e3);
        }
    } else if (mode == 2) {
        // time series: does the duration of the SAME launch drift with time since the process began?
        const long long total16 = bytes / 16, per_wave16 = 8192 / 16;
        const unsigned grid = (unsigned)(((total16 + per_wave16 - 1) / per_wave16 + 3) / 4);
        hipEvent_t ev[64];
        for (auto &e : ev) hipEventCreate(&e);
        for (int rep = 0; rep < 3; rep++) {
            for (int b = 0; b < 60; b++) {
                hipEventRecord(ev[b], 0);
                for (int i = 0; i < 100; i++)
                    hipLaunchKernelGGL((stream_kernel<8>), dim3(grid), dim3(256), 0, 0, src, out, per_wave16, total16);
            }
            hipEventRecord(ev[60], 0);
            hipEventSynchronize(ev[60]);
            printf("rep %d (100 launches per sample, us per launch):", rep);
            for (int b = 0; b < 60; b++) {
                float ms;
                hipEventElapsedTime(&ms, ev[b], ev[b + 1]);
                printf(" %.2f", ms * 10.0);
            }
            printf("\n");
            if (rep == 0) hipDeviceSynchronize();
            if (rep == 1) { hipDeviceSynchronize(); system("sleep 1"); }
        }
    } else {
        printf("-- occupancy-limited (dynamic LDS per workgroup)\n");
        for (int lds : {0, 20480, 40960, 65536}) {
            run<8>(src, out, bytes, 8192, 256, lds);
            run<4>(src, out, bytes, 4096, 256, lds);
        }
        printf("-- workgroup size\n");
        for (int block : {64, 128, 256, 512, 1024}) {
            run<4>(src, out, bytes, 4096, block);
            run<8>(src, out, bytes, 8192, block);
        }
        printf("-- bytes per wave, one shot\n");
        run<2>(src, out, bytes, 2048, 256);
        run<3>(src, out, bytes, 3072, 256);
        run<4>(src, out, bytes, 4096, 256);
        run<5>(src, out, bytes, 5120, 256);
        run<6>(src, out, bytes, 6144, 256);
        run<8>(src, out, bytes, 8192, 256);
        run<12>(src, out, bytes, 12288, 256);
        printf("-- operator size\n");
        for (long long b : {13638480LL, 27276960LL, 54553920LL, 109107840LL, 218215680LL}) {
            run<4>(src, out, b, 4096, 256);
            run<8>(src, out, b, 8192, 256);
            run<8>(src, out, b, 16384, 256);
        }
    }
    return 0;
}
