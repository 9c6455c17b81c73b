// Developer micro-benchmark: what the matrix pipe of gfx950 gives the element types of this library.
// north_star asks for "MFMA tiles for blocks large enough to be a real dense GEMV" and the multi right-hand-side
// product (bsm_mul_multi) is the only place where a block x X is a contraction an MFMA tile could hold.  This
// measures the instructions such a tile would use against the vector FMAs the kernels issue today:
//   v_mfma_f64_16x16x4_f64 (2048 FLOP), v_mfma_f32_16x16x4_f32 (2048 FLOP), v_fma_f64, v_pk_fma_f32
// (independent accumulators, operands in registers, every SIMD of the chip busy).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_mfma_f64(double *out, int iters) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}
__global__ __launch_bounds__(256) void k_mfma_f32(float *out, int iters) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}
__global__ __launch_bounds__(256) void k_fma_f64(double *out, int iters) {
    double acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_fma(acc[k], a, b);
    }
    double s = 0;
    for (int k = 0; k < 8; k++) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkfma_f32(float *out, int iters) {
    f2 acc[8];
    for (int k = 0; k < 8; k++) acc[k] = f2{(float)threadIdx.x, (float)k};
    const f2 a = {1.0f + threadIdx.x * 1e-6f, 1.0f}, b = {1e-6f, 1e-6f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_elementwise_fma(acc[k], a, b);
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += acc[k].x + acc[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K, typename T> static double run(K kern, T *out, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3;
}

int main() {
    void *out;
    (void)hipMalloc(&out, 256 * 4 * 256 * 8);
    const int iters = 20000;
    const double waves = 256.0 * 4 * 4;  // 1024 workgroups of 4 waves: 4 waves per SIMD
    double t = run(k_mfma_f64, (double *)out, iters);
    printf("v_mfma_f64_16x16x4_f64 : %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", waves * iters * 4 * 2048.0 / t / 1e12,
           t * 2.4e9 / (iters * 4.0 * 4));
    t = run(k_mfma_f32, (float *)out, iters);
    printf("v_mfma_f32_16x16x4_f32 : %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", waves * iters * 4 * 2048.0 / t / 1e12,
           t * 2.4e9 / (iters * 4.0 * 4));
    t = run(k_fma_f64, (double *)out, iters);
    printf("v_fma_f64              : %7.1f TFLOP/s\n", waves * iters * 8 * 64 * 2.0 / t / 1e12);
    t = run(k_pkfma_f32, (float *)out, iters);
    printf("v_pk_fma_f32           : %7.1f TFLOP/s\n", waves * iters * 8 * 64 * 4.0 / t / 1e12);
    return 0;
}
