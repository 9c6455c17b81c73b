/* parts_demo.cpp -- bsm_mul_parts from compiled code with raw HIP memory: what a caller that keeps its vectors on the
 * GPUs (AMDGPU.jl ROCArrays behind a ccall, a C++ solver) does with a matrix spread over several devices.
 *
 *   hipcc -O1 -I include examples/parts_demo.cpp -L blocksparsematrices.jl_amd -lbsmrocm \
 *       -Wl,-rpath,$PWD/blocksparsematrices.jl_amd -o /tmp/parts_demo && /tmp/parts_demo
 *
 * The SymmetricBlockMatrix known answer of SURVEY.md 8c (reference src/symmetricblockmatrix.jl:394-432: D = [7] on index 2,
 * B = [1 2] on rows [1], columns [3,4], x = [1,2,3,4] => y = [11,14,1,2]) on a context of two (virtual) devices: the
 * diagonal block and the off-diagonal block land on different parts, x and y are PARTITIONED over the parts like
 * the rows, the transposed contribution y[3:4] crosses the partition inside the library.  Then y of the first
 * product is x of a second one without leaving the devices (A * (A * x) = [5, 98, 11, 22]). */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "bsm_rocm.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bsm_last_error());      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)
#define HCHECK(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));           \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main() {
    int32_t devs[2] = {0, 0};
    bsm_ctx_t ctx;
    CHECK(bsm_ctx_create(devs, 2, &ctx));
    bsm_options o;
    bsm_options_default(&o);
    o.ctx = ctx;
    double D[1] = {7}, O[2] = {1, 2};
    const void *dd[1] = {D}, *oo[1] = {O};
    int64_t ds[1] = {1}, dl[1] = {1}, om[1] = {1}, on[1] = {2}, old_[1] = {1};
    int64_t di0[1] = {2}, or0[1] = {1}, oc0[2] = {3, 4};
    const int64_t *di[1] = {di0}, *orr[1] = {or0}, *occ[1] = {oc0};
    bsm_matrix_t S;
    CHECK(bsm_symmetric_create(BSM_F64, 4, 4, 1, dd, ds, dl, di, 1, oo, om, on, old_, orr, occ, &o, &S));
    const double x[4] = {1, 2, 3, 4};
    double *xp[2], *yp[2], *zp[2];
    bsm_part_info_t pi[2];
    for (int p = 0; p < 2; p++) {
        CHECK(bsm_part_info(S, p, &pi[p]));
        HCHECK(hipSetDevice(pi[p].device));
        const size_t n = (size_t)(pi[p].own_hi - pi[p].own_lo + 1);
        HCHECK(hipMalloc((void **)&xp[p], n * 8 + 8));
        HCHECK(hipMalloc((void **)&yp[p], n * 8 + 8));
        HCHECK(hipMalloc((void **)&zp[p], n * 8 + 8));
        HCHECK(hipMemcpy(xp[p], x + pi[p].col_lo - 1, n * 8, hipMemcpyHostToDevice));  /* square: col range == row range */
        printf("part %d on device %d owns rows %lld..%lld\n", p, pi[p].device, (long long)pi[p].own_lo, (long long)pi[p].own_hi);
    }
    const void *xin[2] = {xp[0], xp[1]};
    void *yout[2] = {yp[0], yp[1]}, *zout[2] = {zp[0], zp[1]};
    CHECK(bsm_mul_parts(S, BSM_OP_N, xin, yout, NULL, NULL, 1, NULL));
    const void *yin[2] = {yp[0], yp[1]};
    CHECK(bsm_mul_parts(S, BSM_OP_N, yin, zout, NULL, NULL, 1, NULL));  /* y parts of one product = x parts of the next */
    double y[4], z[4];
    for (int p = 0; p < 2; p++) {
        const size_t n = (size_t)(pi[p].own_hi - pi[p].own_lo + 1);
        HCHECK(hipSetDevice(pi[p].device));
        HCHECK(hipDeviceSynchronize());
        HCHECK(hipMemcpy(y + pi[p].own_lo - 1, yp[p], n * 8, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(z + pi[p].own_lo - 1, zp[p], n * 8, hipMemcpyDeviceToHost));
    }
    printf("partitioned SymmetricBlockMatrix KAT y = [%g %g %g %g], A*y = [%g %g %g %g]\n", y[0], y[1], y[2], y[3], z[0], z[1], z[2], z[3]);
    if (y[0] != 11 || y[1] != 14 || y[2] != 1 || y[3] != 2) return 1;
    if (z[0] != 5 || z[1] != 98 || z[2] != 11 || z[3] != 22) return 1;
    for (int p = 0; p < 2; p++) {
        (void)hipFree(xp[p]);
        (void)hipFree(yp[p]);
        (void)hipFree(zp[p]);
    }
    CHECK(bsm_destroy(S));
    CHECK(bsm_ctx_destroy(ctx));
    {
        /* the two helpers of a process-per-GPU layer above this ABI: a compute stream whose CU mask leaves CUs to the
         * collective layer's kernels, and the one-launch delivery of received partial-y segments; an ordinary handle's
         * product on that stream, then y[0..1] += {1, 2}, y[3] += 40 */
        void *st = NULL;
        CHECK(bsm_stream_create_reserved(0, 8, &st));
        bsm_options o1;
        bsm_options_default(&o1);
        bsm_matrix_t S1 = NULL;
        CHECK(bsm_symmetric_create(BSM_F64, 4, 4, 1, dd, ds, dl, di, 1, oo, om, on, old_, orr, occ, &o1, &S1));
        double *dx = NULL, *dy = NULL, *seg = NULL;
        const double xh[4] = {1, 2, 3, 4}, add[3] = {1, 2, 40};
        HCHECK(hipSetDevice(0));
        HCHECK(hipMalloc((void **)&dx, 32));
        HCHECK(hipMalloc((void **)&dy, 32));
        HCHECK(hipMalloc((void **)&seg, 24));
        HCHECK(hipMemcpy(dx, xh, 32, hipMemcpyHostToDevice));
        HCHECK(hipMemcpy(seg, add, 24, hipMemcpyHostToDevice));
        CHECK(bsm_mul(S1, BSM_OP_N, dx, dy, NULL, NULL, 1, BSM_MEM_DEVICE, st));
        const int64_t off[2] = {0, 3}, len[2] = {2, 1};
        const void *src[2] = {seg, seg + 2};
        CHECK(bsm_vec_add_segments(BSM_F64, dy, 2, off, src, len, st));
        HCHECK(hipStreamSynchronize((hipStream_t)st));
        double yh[4];
        HCHECK(hipMemcpy(yh, dy, 32, hipMemcpyDeviceToHost));
        printf("product on the CU-reserved stream + segment add: [%g %g %g %g]\n", yh[0], yh[1], yh[2], yh[3]);
        if (yh[0] != 12 || yh[1] != 16 || yh[2] != 1 || yh[3] != 42) return 1;
        const int64_t bad_off[2] = {0, 1};
        if (bsm_vec_add_segments(BSM_F64, dy, 2, bad_off, src, len, st) == BSM_OK) return 1; /* overlapping segments are refused */
        (void)hipFree(dx);
        (void)hipFree(dy);
        (void)hipFree(seg);
        CHECK(bsm_destroy(S1));
        CHECK(bsm_stream_destroy(st));
    }
    printf("OK\n");
    return 0;
}
