/* bsm_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Included once per element type by bsm_oracle.c with
 *   T        element type            (float, double, float _Complex, double _Complex)
 *   SFX      symbol suffix           (f32, f64, c64, c128)
 *   CONJ(z)  complex conjugate of a T (identity for real T)
 *
 * Every function restates one piece of the reference, loop for loop, and cites it.
 * Index arrays are int64 and 1-BASED exactly as the Julia reference stores them.
 * Nothing here is product code: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this file.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* `y .*= beta` -- reference src/blockmatrix.jl:231, src/symmetricblockmatrix.jl:392,
 * src/vbcrs.jl:273,313.  When beta arrives as the Bool `false` (3-arg mul!, reference
 * src/abstractblockmatrix.jl:27-34) Julia's false is a STRONG zero: false*NaN == 0.
 * A numeric 0.0 multiplies (NaN/Inf propagate). */
static void FN(scale_y)(int64_t len, T *y, T beta, int beta_strong_zero) {
    if (beta_strong_zero) {
        for (int64_t i = 0; i < len; i++) y[i] = (T)0;
    } else {
        for (int64_t i = 0; i < len; i++) y[i] = y[i] * beta;
    }
}

/* One per-block product `mul!(view(y, ri), op(B), view(x, ci), alpha, true)`.
 * Restates Julia's LinearAlgebra.generic_matvecmul! (the path taken for the
 * non-strided gather views of reference src/blockmatrix.jl:236-242):
 *   'N': column sweep, b = x[k]*alpha; y[i] += B[i,k]*b
 *   'T'/'C': per output a dot product s = sum_k op(B[k,i])*x[k]; y[i] += s*alpha
 * B is column-major m x n with leading dimension ld (a Julia Matrix).
 * ri/ci: 1-based index lists, or NULL meaning the contiguous range starting at
 * r0/c0 (1-based) -- the VBCRS views of reference src/vbcrs.jl:279-283. */
static void FN(block_gemv)(int op, int64_t m, int64_t n, const T *B, int64_t ld,
                           const int64_t *ri, int64_t r0, const int64_t *ci, int64_t c0,
                           const T *x, T *y, T alpha) {
    if (op == 0) { /* y[rows] += alpha * B * x[cols] */
        for (int64_t k = 0; k < n; k++) {
            int64_t xc = ci ? ci[k] - 1 : c0 - 1 + k;
            T b = x[xc] * alpha;
            const T *col = B + k * ld;
            for (int64_t i = 0; i < m; i++) {
                int64_t yr = ri ? ri[i] - 1 : r0 - 1 + i;
                y[yr] += col[i] * b;
            }
        }
    } else { /* y[cols] += alpha * op(B) * x[rows], op = transpose (1) or adjoint (2) */
        for (int64_t k = 0; k < n; k++) {
            const T *col = B + k * ld;
            T s = (T)0;
            for (int64_t i = 0; i < m; i++) {
                int64_t xr = ri ? ri[i] - 1 : r0 - 1 + i;
                T a = (op == 2) ? CONJ(col[i]) : col[i];
                s += a * x[xr];
            }
            int64_t yc = ci ? ci[k] - 1 : c0 - 1 + k;
            y[yc] += s * alpha;
        }
    }
}

/* BlockSparseMatrix 5-arg mul -- reference src/blockmatrix.jl:225-247.
 * op: 0 = A, 1 = transpose(A), 2 = A'.  For the wrappers the reference swaps the
 * row/col lists (src/symmetricblockmatrix.jl:345-365), wraps each block in
 * transpose/adjoint (src/blockmatrix.jl:154-160) and walks transposecolors
 * (src/blockmatrix.jl:200-206).  colors: CSR-style (colorptr[ncolors+1] 0-based
 * offsets into colorblk, entries 1-based block ids). */
void FN(orc_bsm_mul)(int op, int64_t nrows_y, int64_t nblocks, const T *const *blocks,
                     const int64_t *m, const int64_t *n, const int64_t *ld,
                     const int64_t *const *rowidx, const int64_t *const *colidx,
                     int64_t ncolors, const int64_t *colorptr, const int64_t *colorblk,
                     const T *x, T *y, T alpha, T beta, int beta_strong_zero) {
    (void)nblocks;
    FN(scale_y)(nrows_y, y, beta, beta_strong_zero);
    for (int64_t c = 0; c < ncolors; c++) {
        for (int64_t q = colorptr[c]; q < colorptr[c + 1]; q++) {
            int64_t b = colorblk[q] - 1;
            FN(block_gemv)(op, m[b], n[b], blocks[b], ld[b], rowidx[b], 0, colidx[b], 0, x, y,
                           alpha);
        }
    }
}

/* SymmetricBlockMatrix 5-arg mul -- reference src/symmetricblockmatrix.jl:386-435.
 * Three coloured sweeps.  For S (op 0):
 *   phase 1 (:394-405)  y[rows] += a * B   * x[cols]
 *   phase 2 (:407-418)  y[cols] += a * B^T * x[rows]
 *   phase 3 (:420-432)  y[idx]  += a * D   * x[idx]
 * For transpose(S)/S' the accessors swap lists and colour sets and wrap the blocks
 * (:219-237, :307-325, :345-365): phase 1 walks the *transpose* colours computing
 * y[cols] += a * op(B) * x[rows]; phase 2 computes transpose(op(B)): for op = T that is
 * B itself, for op = C it is conj(B), y[rows] += a * (B|conj B) * x[cols]; phase 3 uses
 * op(D).  Colour sets are CSR-style like orc_bsm_mul; set k: 0 = offdiagonalcolors,
 * 1 = transposeoffdiagonalcolors, 2 = diagonalcolors of the UNWRAPPED matrix. */
void FN(orc_sym_mul)(int op, int64_t nrows_y, int64_t ndiag, const T *const *diag,
                     const int64_t *dsize, const int64_t *dld, const int64_t *const *didx,
                     int64_t noff, const T *const *off, const int64_t *m, const int64_t *n,
                     const int64_t *ld, const int64_t *const *rowidx,
                     const int64_t *const *colidx, const int64_t *ncolors,
                     const int64_t *const *colorptr, const int64_t *const *colorblk, const T *x,
                     T *y, T alpha, T beta, int beta_strong_zero) {
    (void)ndiag;
    (void)noff;
    FN(scale_y)(nrows_y, y, beta, beta_strong_zero);
    /* phase 1: offdiagonalcolors(A) -- for wrappers that is transposeoffdiagonalcolors(S) */
    int s1 = (op == 0) ? 0 : 1;
    for (int64_t c = 0; c < ncolors[s1]; c++)
        for (int64_t q = colorptr[s1][c]; q < colorptr[s1][c + 1]; q++) {
            int64_t b = colorblk[s1][q] - 1;
            /* op 0: y[rows] += B x[cols]; op 1/2: y[cols] += op(B) x[rows] */
            FN(block_gemv)(op, m[b], n[b], off[b], ld[b], rowidx[b], 0, colidx[b], 0, x, y, alpha);
        }
    /* phase 2: transposeoffdiagonalcolors(A), block = transpose(offdiagonal(A, i)) */
    int s2 = (op == 0) ? 1 : 0;
    for (int64_t c = 0; c < ncolors[s2]; c++)
        for (int64_t q = colorptr[s2][c]; q < colorptr[s2][c + 1]; q++) {
            int64_t b = colorblk[s2][q] - 1;
            if (op == 0) {
                FN(block_gemv)(1, m[b], n[b], off[b], ld[b], rowidx[b], 0, colidx[b], 0, x, y,
                               alpha);
            } else if (op == 1) { /* transpose(transpose(B)) = B */
                FN(block_gemv)(0, m[b], n[b], off[b], ld[b], rowidx[b], 0, colidx[b], 0, x, y,
                               alpha);
            } else { /* transpose(adjoint(B)) = conj(B): y[rows] += a * conj(B) * x[cols] */
                for (int64_t k = 0; k < n[b]; k++) {
                    T bb = x[colidx[b][k] - 1] * alpha;
                    const T *col = off[b] + k * ld[b];
                    for (int64_t i = 0; i < m[b]; i++) y[rowidx[b][i] - 1] += CONJ(col[i]) * bb;
                }
            }
        }
    /* phase 3: diagonal blocks, op(D) on the same index list both sides */
    for (int64_t c = 0; c < ncolors[2]; c++)
        for (int64_t q = colorptr[2][c]; q < colorptr[2][c + 1]; q++) {
            int64_t d = colorblk[2][q] - 1;
            FN(block_gemv)(op, dsize[d], dsize[d], diag[d], dld[d], didx[d], 0, didx[d], 0, x, y,
                           alpha);
        }
}

/* VBCRS forward 5-arg mul -- reference src/vbcrs.jl:266-288.  Inputs are the
 * constructor's outputs (rowptr 1-based, length nblockrows+1; colindices per block;
 * rowindices per block ROW); blocks already permuted into sorted order.  Heights
 * are taken per block (size(block,1), :281). */
void FN(orc_vbcrs_mul)(int64_t nrows_y, int64_t nblockrows, const int64_t *rowptr,
                       const int64_t *colindices, const int64_t *rowindices,
                       const T *const *blocks, const int64_t *m, const int64_t *n,
                       const int64_t *ld, const T *x, T *y, T alpha, T beta,
                       int beta_strong_zero) {
    FN(scale_y)(nrows_y, y, beta, beta_strong_zero);
    for (int64_t br = 0; br < nblockrows; br++)
        for (int64_t bi = rowptr[br]; bi < rowptr[br + 1]; bi++) {
            int64_t b = bi - 1;
            FN(block_gemv)(0, m[b], n[b], blocks[b], ld[b], NULL, rowindices[br], NULL,
                           colindices[b], x, y, alpha);
        }
}

/* The same forward product with the reference's task structure made explicit: `@tasks for
 * browidx` (src/vbcrs.jl:275-276) == one OpenMP task per block row (dynamic schedule, like
 * DynamicScheduler()).  Used only for the all-cores CPU baseline of bench.py; built with
 * -fopenmp (without it the pragma is ignored and this equals orc_vbcrs_mul). */
void FN(orc_vbcrs_mul_par)(int64_t nrows_y, int64_t nblockrows, const int64_t *rowptr,
                           const int64_t *colindices, const int64_t *rowindices,
                           const T *const *blocks, const int64_t *m, const int64_t *n,
                           const int64_t *ld, const T *x, T *y, T alpha, T beta,
                           int beta_strong_zero) {
    FN(scale_y)(nrows_y, y, beta, beta_strong_zero);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t br = 0; br < nblockrows; br++)
        for (int64_t bi = rowptr[br]; bi < rowptr[br + 1]; bi++) {
            int64_t b = bi - 1;
            FN(block_gemv)(0, m[b], n[b], blocks[b], ld[b], NULL, rowindices[br], NULL,
                           colindices[b], x, y, alpha);
        }
}

/* VBCRS adjoint/transpose -- reference src/vbcrs.jl:303-329 (serial double loop,
 * y[col range] += a * op(block) * x[row range]); op: 1 = transpose, 2 = adjoint.
 * The 3-arg form (:331-341) zero-fills y then calls this with (true, true): the
 * caller expresses that as beta_strong_zero = 1. */
void FN(orc_vbcrs_mul_t)(int op, int64_t ncols_y, int64_t nblockrows, const int64_t *rowptr,
                         const int64_t *colindices, const int64_t *rowindices,
                         const T *const *blocks, const int64_t *m, const int64_t *n,
                         const int64_t *ld, const T *x, T *y, T alpha, T beta,
                         int beta_strong_zero) {
    FN(scale_y)(ncols_y, y, beta, beta_strong_zero);
    for (int64_t br = 0; br < nblockrows; br++)
        for (int64_t bi = rowptr[br]; bi < rowptr[br + 1]; bi++) {
            int64_t b = bi - 1;
            FN(block_gemv)(op, m[b], n[b], blocks[b], ld[b], NULL, rowindices[br], NULL,
                           colindices[b], x, y, alpha);
        }
}

/* Independent check path -- the reference's own test oracle: sparse(A) * x
 * (reference src/sparse.jl:127-129; duplicates are summed by `sparse`, which is
 * what `+=` in mul does).  COO triples (1-based), y = beta*y + alpha*A*x with the
 * same strong-zero rule.  Accumulates per output row in long-hand, no blocking. */
void FN(orc_coo_mul)(int64_t nrows_y, int64_t nnz, const int64_t *rows, const int64_t *cols,
                     const T *vals, const T *x, T *y, T alpha, T beta, int beta_strong_zero) {
    FN(scale_y)(nrows_y, y, beta, beta_strong_zero);
    for (int64_t k = 0; k < nnz; k++) y[rows[k] - 1] += alpha * (vals[k] * x[cols[k] - 1]);
}

/* Timed loop for bench.py's CPU figures of the symmetric legs: `reps` products S * x (the 3-argument form, alpha = 1,
 * strong-zero beta: the reference's three coloured sweeps, src/symmetricblockmatrix.jl:386-435) on PRE-MARSHALLED
 * arguments, the loop and the clock in C.  Returns the elapsed seconds. */
double FN(orc_sym_bench)(int64_t reps, int64_t nrows_y, int64_t ndiag, const T *const *diag, const int64_t *dsize,
                         const int64_t *dld, const int64_t *const *didx, int64_t noff, const T *const *off,
                         const int64_t *m, const int64_t *n, const int64_t *ld, const int64_t *const *rowidx,
                         const int64_t *const *colidx, const int64_t *ncolors, const int64_t *const *colorptr,
                         const int64_t *const *colorblk, const T *x, T *y) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int64_t r = 0; r < reps; r++)
        FN(orc_sym_mul)(0, nrows_y, ndiag, diag, dsize, dld, didx, noff, off, m, n, ld, rowidx, colidx, ncolors, colorptr,
                        colorblk, x, y, (T)1, (T)0, 1);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

#undef FN
#undef CAT
#undef CAT_
