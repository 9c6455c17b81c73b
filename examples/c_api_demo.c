/* c_api_demo.c -- the C ABI of include/bsm_rocm.h used from plain C (what a Julia `ccall`,
 * a cgo stub or any other FFI sees).  Builds the hand-derived known answers of SURVEY.md 8c.
 *
 *   gcc -O1 -I include examples/c_api_demo.c -L blocksparsematrices.jl_amd -lbsmrocm \
 *       -Wl,-rpath,$PWD/blocksparsematrices.jl_amd -o /tmp/c_api_demo
 *   /tmp/c_api_demo            # analysis only (no GPU needed): bookkeeping
 *   /tmp/c_api_demo gpu        # also runs the products on the current HIP device
 */
#include <stdio.h>
#include <string.h>

#include "bsm_rocm.h"

#define CHECK(call)                                                      \
    do {                                                                 \
        int rc_ = (call);                                                \
        if (rc_ != 0) {                                                  \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bsm_last_error()); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char **argv) {
    const int use_gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    bsm_options o;
    bsm_options_default(&o);
    o.device = use_gpu ? BSM_DEVICE_CURRENT : BSM_DEVICE_NONE;
    printf("%s\n", bsm_version());

    /* VBCRS constructor KAT (reference src/vbcrs.jl:84-117):
     * (row0, col0) = (5,1), (1,7), (1,3), (5,9)  =>  perm = [3,2,1,4], rowptr = [1,3,5],
     * rowindices = [1,5], colindices = [3,7,1,9] */
    double b0[2 * 3] = {1, 2, 3, 4, 5, 6}, b1[4 * 2] = {1, 1, 1, 1, 2, 2, 2, 2};
    double b2[4 * 4] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, b3[2 * 1] = {7, 8};
    const void *blocks[4] = {b0, b1, b2, b3};
    int64_t m[4] = {2, 4, 4, 2}, n[4] = {3, 2, 4, 1}, ld[4] = {2, 4, 4, 2};
    int64_t r0[4] = {5, 1, 1, 5}, c0[4] = {1, 7, 3, 9};
    bsm_matrix_t V;
    CHECK(bsm_vbcrs_create(BSM_F64, 6, 9, 4, blocks, m, n, ld, r0, c0, &o, &V));
    int64_t buf[16], len = 16;
    CHECK(bsm_get_bookkeeping(V, BSM_BK_VBCRS_PERM, buf, &len));
    printf("perm       =");
    for (int64_t i = 0; i < len; i++) printf(" %lld", (long long)buf[i]);
    const int perm_ok = len == 4 && buf[0] == 3 && buf[1] == 2 && buf[2] == 1 && buf[3] == 4;
    len = 16;
    CHECK(bsm_get_bookkeeping(V, BSM_BK_VBCRS_ROWPTR, buf, &len));
    printf("\nrowptr     =");
    for (int64_t i = 0; i < len; i++) printf(" %lld", (long long)buf[i]);
    const int rowptr_ok = len == 3 && buf[0] == 1 && buf[1] == 3 && buf[2] == 5;
    bsm_stats_t st;
    CHECK(bsm_stats(V, &st));
    printf("\nnnz = %lld, exclusive = %lld\n", (long long)st.nnz, (long long)st.exclusive);
    if (!perm_ok || !rowptr_ok || st.nnz != 32) {
        fprintf(stderr, "bookkeeping mismatch\n");
        return 1;
    }

    if (use_gpu) {
        /* BlockSparseMatrix KAT (src/blockmatrix.jl:231-244): y = [10, 0, 27, 0] */
        double B1[4] = {1, 3, 2, 4}, B2[1] = {5}; /* column-major [1 2; 3 4] */
        const void *bb[2] = {B1, B2};
        int64_t bm[2] = {2, 1}, bn[2] = {2, 1}, bld[2] = {2, 1};
        int64_t ri0[2] = {1, 3}, ri1[1] = {3}, ci0[2] = {2, 4}, ci1[1] = {1};
        const int64_t *ri[2] = {ri0, ri1}, *ci[2] = {ci0, ci1};
        bsm_matrix_t A;
        CHECK(bsm_blocksparse_create(BSM_F64, 4, 4, 2, bb, bm, bn, bld, ri, ci, &o, &A));
        double x[4] = {1, 2, 3, 4}, y[4] = {-1, -1, -1, -1};
        CHECK(bsm_mul(A, BSM_OP_N, x, y, NULL, NULL, 1, BSM_MEM_HOST, NULL));
        printf("BlockSparseMatrix KAT y = [%g %g %g %g]\n", y[0], y[1], y[2], y[3]);
        if (y[0] != 10 || y[1] != 0 || y[2] != 27 || y[3] != 0) return 1;
        /* SymmetricBlockMatrix KAT (src/symmetricblockmatrix.jl:394-432): y = [11, 14, 1, 2] */
        double D[1] = {7}, O[2] = {1, 2};
        const void *dd[1] = {D}, *oo[1] = {O};
        int64_t ds[1] = {1}, dl[1] = {1}, om[1] = {1}, on[1] = {2}, old_[1] = {1};
        int64_t di0[1] = {2}, or0[1] = {1}, oc0[2] = {3, 4};
        const int64_t *di[1] = {di0}, *orr[1] = {or0}, *occ[1] = {oc0};
        bsm_matrix_t S;
        CHECK(bsm_symmetric_create(BSM_F64, 4, 4, 1, dd, ds, dl, di, 1, oo, om, on, old_, orr, occ, &o, &S));
        CHECK(bsm_mul(S, BSM_OP_N, x, y, NULL, NULL, 1, BSM_MEM_HOST, NULL));
        printf("SymmetricBlockMatrix KAT y = [%g %g %g %g]\n", y[0], y[1], y[2], y[3]);
        if (y[0] != 11 || y[1] != 14 || y[2] != 1 || y[3] != 2) return 1;
        CHECK(bsm_destroy(A));
        CHECK(bsm_destroy(S));
    }
    CHECK(bsm_destroy(V));
    printf("OK\n");
    return 0;
}
