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
    /* VariableBlockCompressedRowStorage(bsm) (src/vbcrs.jl:150-160): same four blocks given with index
     * lists; only first(rowindices), first(colindices) are used => identical bookkeeping */
    {
        int64_t l5[2] = {5, 6}, l1[4] = {1, 2, 3, 4}, c1[3] = {1, 2, 3}, c7[2] = {7, 8}, c3[4] = {3, 4, 5, 6}, c9[1] = {9};
        const int64_t *ri[4] = {l5, l1, l1, l5}, *ci[4] = {c1, c7, c3, c9};
        bsm_matrix_t W;
        CHECK(bsm_vbcrs_create_from_blocksparse(BSM_F64, 6, 9, 4, blocks, m, n, ld, ri, ci, &o, &W));
        int64_t b2[16], l2 = 16;
        CHECK(bsm_get_bookkeeping(W, BSM_BK_VBCRS_PERM, b2, &l2));
        if (l2 != 4 || b2[0] != 3 || b2[1] != 2 || b2[2] != 1 || b2[3] != 4) {
            fprintf(stderr, "converter bookkeeping mismatch\n");
            return 1;
        }
        CHECK(bsm_destroy(W));
    }
    /* the row partition both multi-GPU layers use: 4 blocks, keys 5 1 1 5, weights = stored entries */
    {
        int64_t w[4] = {6, 8, 16, 2}, lo[2], hi[2];
        int32_t part[4];
        CHECK(bsm_partition_rows(6, 4, r0, w, 2, part, lo, hi));
        printf("partition  = %d %d %d %d, own = [%lld,%lld] [%lld,%lld]\n", part[0], part[1], part[2], part[3],
               (long long)lo[0], (long long)hi[0], (long long)lo[1], (long long)hi[1]);
        if (part[1] != 0 || part[2] != 0 || part[0] != 1 || part[3] != 1 || lo[0] != 1 || hi[0] != 4 || lo[1] != 5 || hi[1] != 6)
            return 1;
    }
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
        /* A * X with 5 right-hand sides (bsm_mul_multi: ONE padded 8-column pass over A instead of LinearMaps' five
         * sweeps): column j = (j + 1) * x, so Y[:, j] = (j + 1) * [11, 14, 1, 2]; column-major, leading dimension 6 */
        double X5[5 * 6], Y5[5 * 6];
        for (int j = 0; j < 5; j++)
            for (int i = 0; i < 6; i++) {
                X5[j * 6 + i] = i < 4 ? (j + 1) * x[i] : -99;  /* rows 4, 5: padding of the leading dimension */
                Y5[j * 6 + i] = -7;
            }
        CHECK(bsm_mul_multi(S, BSM_OP_N, 5, X5, 6, Y5, 6, NULL, NULL, 1, BSM_MEM_HOST, NULL));
        for (int j = 0; j < 5; j++)
            if (Y5[j * 6] != 11 * (j + 1) || Y5[j * 6 + 1] != 14 * (j + 1) || Y5[j * 6 + 2] != 1 * (j + 1) ||
                Y5[j * 6 + 3] != 2 * (j + 1) || Y5[j * 6 + 4] != -7 || Y5[j * 6 + 5] != -7) {
                fprintf(stderr, "bsm_mul_multi mismatch in column %d\n", j);
                return 1;
            }
        printf("SymmetricBlockMatrix * X (5 columns): Y[:, 4] = [%g %g %g %g]\n", Y5[24], Y5[25], Y5[26], Y5[27]);
        CHECK(bsm_destroy(A));
        CHECK(bsm_destroy(S));
        /* the same symmetric KAT with the handle spread over a context of TWO (virtual) devices: the
         * diagonal block and the off-diagonal block land on different parts, the transposed
         * contribution y[3:4] crosses the partition (bsm_ctx_create, bsm_options.ctx) */
        int32_t devs[2] = {0, 0};
        bsm_ctx_t ctx;
        CHECK(bsm_ctx_create(devs, 2, &ctx));
        bsm_options oc = o;
        oc.ctx = ctx;
        bsm_matrix_t S2;
        CHECK(bsm_symmetric_create(BSM_F64, 4, 4, 1, dd, ds, dl, di, 1, oo, om, on, old_, orr, occ, &oc, &S2));
        bsm_part_info_t pi;
        CHECK(bsm_part_info(S2, 1, &pi));
        double y2[4] = {-1, -1, -1, -1};
        CHECK(bsm_mul(S2, BSM_OP_N, x, y2, NULL, NULL, 1, BSM_MEM_HOST, NULL));
        printf("two-device SymmetricBlockMatrix KAT y = [%g %g %g %g] (part 1 owns rows %lld..%lld)\n", y2[0], y2[1],
               y2[2], y2[3], (long long)pi.own_lo, (long long)pi.own_hi);
        if (y2[0] != 11 || y2[1] != 14 || y2[2] != 1 || y2[3] != 2) return 1;
        /* rowcolvals: nnz = 2*2 + 1 = 5 triples (src/symmetricblockmatrix.jl:377-382) */
        int64_t rr[8], cc[8], cnt = 8;
        double vv[8];
        CHECK(bsm_rowcolvals(S2, rr, cc, vv, &cnt, BSM_MEM_HOST, NULL));
        double dense[16] = {0};
        for (int64_t k = 0; k < cnt; k++) dense[(rr[k] - 1) * 4 + (cc[k] - 1)] += vv[k];
        printf("rowcolvals: %lld triples, A[1,3] = %g, A[3,1] = %g, A[2,2] = %g\n", (long long)cnt, dense[2], dense[8], dense[5]);
        if (cnt != 5 || dense[2] != 1 || dense[3] != 2 || dense[8] != 1 || dense[12] != 2 || dense[5] != 7) return 1;
        CHECK(bsm_destroy(S2));
        CHECK(bsm_ctx_destroy(ctx));
    }
    CHECK(bsm_destroy(V));
    printf("OK\n");
    return 0;
}
