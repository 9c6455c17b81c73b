// Developer micro-benchmark: what a cross-stream dependency costs on this runtime -- hipEventRecord + hipStreamWaitEvent
// against hipStreamWriteValue64 + hipStreamWaitValue64 (stream memory operations on a flag in signal / pinned host memory).
// Two streams play ping-pong with a tiny kernel each; time per hop = total / (2 N) - kernel time.
// build: hipcc --offload-arch=gfx950 -O3 tools/hop_latency.hip -o tools/bin/hop_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(int *p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int *d;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));
    const int N = 2000;
    // same-stream baseline: 2 N dependent tiny kernels
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2 * N; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
    CK(hipStreamSynchronize(a));
    double base = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * N);
    printf("same stream, dependent tiny kernels      : %6.2f us per kernel\n", base);
    // events
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
        CK(hipEventRecord(ea, a));
        CK(hipStreamWaitEvent(b, ea, 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
        CK(hipEventRecord(eb, b));
        CK(hipStreamWaitEvent(a, eb, 0));
    }
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    double ev = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * N);
    printf("two streams, event record + wait per hop  : %6.2f us per hop (incl. the kernel)\n", ev);
    if (can) {
        for (int mode = 0; mode < 2; mode++) {
            uint64_t *fa = nullptr, *fb = nullptr;
            if (mode == 0) {
                CK(hipExtMallocWithFlags((void **)&fa, 8, hipMallocSignalMemory));
                CK(hipExtMallocWithFlags((void **)&fb, 8, hipMallocSignalMemory));
            } else {
                CK(hipHostMalloc((void **)&fa, 8, hipHostMallocCoherent));
                CK(hipHostMalloc((void **)&fb, 8, hipHostMallocCoherent));
            }
            CK(hipMemset(fa, 0, 8));
            CK(hipMemset(fb, 0, 8));
            CK(hipDeviceSynchronize());
            t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= N; i++) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
                CK(hipStreamWriteValue64(a, fa, (uint64_t)i, 0));
                CK(hipStreamWaitValue64(b, fa, (uint64_t)i, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
                CK(hipStreamWriteValue64(b, fb, (uint64_t)i, 0));
                CK(hipStreamWaitValue64(a, fb, (uint64_t)i, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
            }
            CK(hipStreamSynchronize(a));
            CK(hipStreamSynchronize(b));
            double sv = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * N);
            printf("two streams, write / wait value (%s): %6.2f us per hop (incl. the kernel)\n",
                   mode == 0 ? "signal memory" : "pinned host  ", sv);
        }
    }
    int h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", h);
    return 0;
}
