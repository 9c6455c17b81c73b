/* bsm_oracle.c -- CPU restatement of BlockSparseMatrices.jl's mul! hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP library in
 * blocksparsematrices.jl_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it; the product path never calls it and fails loudly if
 * the HIP library is missing.
 *
 * Pinning: the reference is pure Julia and no Julia toolchain exists in the build
 * container, so the reference cannot be executed.  The oracle is pinned by
 *   (1) the reference's own fixture test/assets/symmetricblockexamples.jld2 (inputs;
 *       decoded to tests/golden/symmetric_{cuboid,sphere}.bin), checked the way the reference's tests check it:
 *       against an independent sparse (COO) product (orc_coo_mul here, scipy.sparse in
 *       tests/) -- reference test/test_symmetricblockmatrix.jl:48-97;
 *   (2) the hand-derived known answers of SURVEY.md section 8c (tests/test_oracle.py).
 * No reference-generated OUTPUT vectors exist (the reference's tests store none).
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define T float
#define SFX f32
#define CONJ(z) (z)
#include "bsm_oracle_impl.h"
#undef T
#undef SFX
#undef CONJ

#define T double
#define SFX f64
#define CONJ(z) (z)
#include "bsm_oracle_impl.h"
#undef T
#undef SFX
#undef CONJ

#define T float _Complex
#define SFX c64
#define CONJ(z) conjf(z)
#include "bsm_oracle_impl.h"
#undef T
#undef SFX
#undef CONJ

#define T double _Complex
#define SFX c128
#define CONJ(z) conj(z)
#include "bsm_oracle_impl.h"
#undef T
#undef SFX
#undef CONJ

/* ------------------------------------------------------------------------------------
 * VBCRS constructor bookkeeping -- reference src/vbcrs.jl:78-122.
 *   perm = sortperm(1:n; by = i -> (rowindices[i], colindices[i]))          (:84)
 *   sortperm is stable, so ties keep input order.
 *   rowptr (1-based, length nblockrows+1, last = n+1)                       (:97,103,110,117)
 *   colindices permuted per block (:115); rowindices one entry per block ROW (:100,104,111)
 * All outputs 1-based like the reference.  Returns nblockrows.
 * Implemented as a plain stable insertion/merge sort on (row, col, input position).
 * ---------------------------------------------------------------------------------- */
static int64_t *g_rs, *g_cs;
static int cmp_rowcol(const void *a, const void *b) {
    int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    if (g_rs[i] != g_rs[j]) return g_rs[i] < g_rs[j] ? -1 : 1;
    if (g_cs[i] != g_cs[j]) return g_cs[i] < g_cs[j] ? -1 : 1;
    return i < j ? -1 : (i > j ? 1 : 0); /* stability: ties keep input order */
}

int64_t orc_vbcrs_build(int64_t n, const int64_t *rowstart, const int64_t *colstart,
                        int64_t *perm, int64_t *rowptr, int64_t *colindices,
                        int64_t *rowindices) {
    if (n <= 0) return 0; /* the reference throws on matrices[1] for n = 0 (:81) */
    g_rs = (int64_t *)rowstart;
    g_cs = (int64_t *)colstart;
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    qsort(perm, (size_t)n, sizeof(int64_t), cmp_rowcol);
    int64_t rowidx = 0;
    rowptr[0] = 1;
    rowindices[0] = rowstart[perm[0]];
    for (int64_t out = 0; out < n; out++) {
        int64_t in = perm[out];
        if (rowstart[in] != rowindices[rowidx]) {
            rowidx++;
            rowptr[rowidx] = out + 1;
            rowindices[rowidx] = rowstart[in];
        }
        colindices[out] = colstart[in];
    }
    rowptr[rowidx + 1] = n + 1;
    for (int64_t i = 0; i < n; i++) perm[i] += 1; /* 1-based */
    return rowidx + 1;
}

/* ------------------------------------------------------------------------------------
 * Colouring.  The reference hands (1:nblocks, block -> index list, 1:maxindex) to
 * GraphsColoring.jl (reference src/coloring.jl:45-61); two blocks conflict iff their
 * index lists intersect.  GraphsColoring (WorkstreamDSATUR, compat 0.2.0, no Manifest)
 * is NOT in the reference tree, and no reference test inspects colours, so the exact
 * assignment is unpinnable.  The contract restated here is
 *   (a) serial scheduler: exactly one colour [1..nblocks]  (reference src/blockmatrix.jl:91-92)
 *   (b) otherwise: a VALID colouring (classes partition 1..nblocks, no two blocks of a
 *       class share an index) that is DETERMINISTIC, by this specification:
 *       DSATUR -- repeatedly take the uncoloured block with the largest saturation
 *       (number of distinct colours among its neighbours), ties by larger degree
 *       (distinct neighbours), then by smaller block id; give it the smallest colour
 *       not used by a neighbour.  Classes are listed by colour id, members ascending.
 * This O(V^2 + E) form is for small test cases only.
 * lists: idx[b] = 1-based index list of block b with length len[b]; maxindex = largest index.
 * color_out[b] = 0-based colour of block b.  Returns number of colours.
 * ---------------------------------------------------------------------------------- */
int64_t orc_color_dsatur(int64_t nblocks, const int64_t *const *idx, const int64_t *len,
                         int64_t maxindex, int64_t *color_out) {
    if (nblocks <= 0) return 0;
    /* incidence: for every index, the blocks touching it */
    int64_t *cnt = calloc((size_t)maxindex + 2, sizeof(int64_t));
    for (int64_t b = 0; b < nblocks; b++)
        for (int64_t k = 0; k < len[b]; k++) cnt[idx[b][k]]++;
    int64_t *ptr = malloc(((size_t)maxindex + 2) * sizeof(int64_t));
    ptr[0] = 0;
    for (int64_t i = 0; i <= maxindex; i++) ptr[i + 1] = ptr[i] + cnt[i];
    int64_t *inc = malloc((size_t)(ptr[maxindex + 1] + 1) * sizeof(int64_t));
    memset(cnt, 0, ((size_t)maxindex + 2) * sizeof(int64_t));
    for (int64_t b = 0; b < nblocks; b++)
        for (int64_t k = 0; k < len[b]; k++) {
            int64_t i = idx[b][k];
            inc[ptr[i] + cnt[i]++] = b;
        }
    /* dense adjacency (small cases only) */
    unsigned char *adj = calloc((size_t)nblocks * nblocks, 1);
    for (int64_t i = 0; i <= maxindex; i++)
        for (int64_t p = ptr[i]; p < ptr[i + 1]; p++)
            for (int64_t q = ptr[i]; q < ptr[i + 1]; q++)
                if (inc[p] != inc[q]) adj[inc[p] * nblocks + inc[q]] = 1;
    int64_t *deg = calloc((size_t)nblocks, sizeof(int64_t));
    for (int64_t a = 0; a < nblocks; a++)
        for (int64_t b = 0; b < nblocks; b++) deg[a] += adj[a * nblocks + b];
    for (int64_t b = 0; b < nblocks; b++) color_out[b] = -1;
    unsigned char *used = malloc((size_t)nblocks + 1);
    int64_t ncolors = 0;
    for (int64_t step = 0; step < nblocks; step++) {
        int64_t best = -1, bestsat = -1;
        for (int64_t v = 0; v < nblocks; v++) {
            if (color_out[v] >= 0) continue;
            memset(used, 0, (size_t)nblocks + 1);
            int64_t sat = 0;
            for (int64_t w = 0; w < nblocks; w++)
                if (adj[v * nblocks + w] && color_out[w] >= 0 && !used[color_out[w]]) {
                    used[color_out[w]] = 1;
                    sat++;
                }
            if (best < 0 || sat > bestsat || (sat == bestsat && deg[v] > deg[best])) {
                best = v;
                bestsat = sat;
            }
        }
        memset(used, 0, (size_t)nblocks + 1);
        for (int64_t w = 0; w < nblocks; w++)
            if (adj[best * nblocks + w] && color_out[w] >= 0) used[color_out[w]] = 1;
        int64_t c = 0;
        while (used[c]) c++;
        color_out[best] = c;
        if (c + 1 > ncolors) ncolors = c + 1;
    }
    free(used);
    free(deg);
    free(adj);
    free(inc);
    free(ptr);
    free(cnt);
    return ncolors;
}

/* ------------------------------------------------------------------------------------
 * WorkstreamDSATUR -- the reference's default `coloringalgorithm` (src/BlockSparseMatrices.jl:10),
 * which lives in GraphsColoring.jl (not in the reference tree).  Restated from the PUBLISHED
 * algorithm: Turcksin, Kronbichler, Bangerth, "WorkStream -- a design pattern for multicore-enabled
 * finite element computations", ACM TOMS 43(1), 2016, section 3.2, with every tie-break fixed here
 * (this is the specification the HIP library's host analysis is compared against bit for bit; it is
 * NOT claimed to equal GraphsColoring's output, whose source is not available):
 *   1. zones: seed = smallest block id not yet in a zone; the next zone = every block not yet in a
 *      zone that conflicts with a block of the current zone; repeat until it is empty, then the
 *      next seed opens the next zone (zone numbers keep counting).  Zone k conflicts only with
 *      zones k-1 and k+1.
 *   2. every zone is coloured on its own with DSATUR (rule of orc_color_dsatur; saturation, degree
 *      and forbidden colours count neighbours INSIDE the zone only).
 *   3. gather, separately for the even and for the odd zones: the zone with the most colours (the
 *      first such) founds the global classes of its parity, colour c -> class c; every other zone,
 *      in zone order, sorts its colours by size (largest first, ties: smaller colour id) and hands
 *      each to the global class with the fewest members among those it has not used yet (ties:
 *      smaller class id).  Final colour ids: even classes first, then odd classes.
 * O(V^2) dense form for small test cases.  Returns the number of colours.
 * ---------------------------------------------------------------------------------- */
int64_t orc_color_workstream(int64_t nblocks, const int64_t *const *idx, const int64_t *len,
                             int64_t maxindex, int64_t *color_out) {
    if (nblocks <= 0) return 0;
    const int64_t n = nblocks;
    unsigned char *adj = calloc((size_t)n * n, 1);
    {
        int64_t *cnt = calloc((size_t)maxindex + 2, sizeof(int64_t));
        for (int64_t b = 0; b < n; b++)
            for (int64_t k = 0; k < len[b]; k++) cnt[idx[b][k]]++;
        int64_t *ptr = malloc(((size_t)maxindex + 2) * sizeof(int64_t));
        ptr[0] = 0;
        for (int64_t i = 0; i <= maxindex; i++) ptr[i + 1] = ptr[i] + cnt[i];
        int64_t *inc = malloc((size_t)(ptr[maxindex + 1] + 1) * sizeof(int64_t));
        memset(cnt, 0, ((size_t)maxindex + 2) * sizeof(int64_t));
        for (int64_t b = 0; b < n; b++)
            for (int64_t k = 0; k < len[b]; k++) inc[ptr[idx[b][k]] + cnt[idx[b][k]]++] = b;
        for (int64_t i = 0; i <= maxindex; i++)
            for (int64_t p = ptr[i]; p < ptr[i + 1]; p++)
                for (int64_t q = ptr[i]; q < ptr[i + 1]; q++)
                    if (inc[p] != inc[q]) adj[inc[p] * n + inc[q]] = 1;
        free(inc);
        free(ptr);
        free(cnt);
    }
    /* 1. zones */
    int64_t *zone = malloc((size_t)n * sizeof(int64_t));
    for (int64_t v = 0; v < n; v++) zone[v] = -1;
    int64_t nzones = 0;
    for (int64_t seed = 0; seed < n; seed++) {
        if (zone[seed] >= 0) continue;
        zone[seed] = nzones;
        for (;;) {
            int64_t grew = 0;
            for (int64_t v = 0; v < n; v++) {
                if (zone[v] >= 0) continue;
                for (int64_t w = 0; w < n; w++)
                    if (zone[w] == nzones && adj[v * n + w]) {
                        zone[v] = nzones + 1;
                        grew = 1;
                        break;
                    }
            }
            nzones++;
            if (!grew) break;
        }
    }
    /* 2. DSATUR inside every zone */
    int64_t *col = malloc((size_t)n * sizeof(int64_t));
    for (int64_t v = 0; v < n; v++) col[v] = -1;
    int64_t *zcolors = calloc((size_t)nzones, sizeof(int64_t));
    unsigned char *used = malloc((size_t)n + 1);
    for (int64_t z = 0; z < nzones; z++) {
        for (;;) {
            int64_t best = -1, bestsat = -1, bestdeg = -1;
            for (int64_t v = 0; v < n; v++) {
                if (zone[v] != z || col[v] >= 0) continue;
                memset(used, 0, (size_t)n + 1);
                int64_t sat = 0, deg = 0;
                for (int64_t w = 0; w < n; w++)
                    if (zone[w] == z && adj[v * n + w]) {
                        deg++;
                        if (col[w] >= 0 && !used[col[w]]) {
                            used[col[w]] = 1;
                            sat++;
                        }
                    }
                if (best < 0 || sat > bestsat || (sat == bestsat && deg > bestdeg)) {
                    best = v;
                    bestsat = sat;
                    bestdeg = deg;
                }
            }
            if (best < 0) break;
            memset(used, 0, (size_t)n + 1);
            for (int64_t w = 0; w < n; w++)
                if (zone[w] == z && adj[best * n + w] && col[w] >= 0) used[col[w]] = 1;
            int64_t c = 0;
            while (used[c]) c++;
            col[best] = c;
            if (c + 1 > zcolors[z]) zcolors[z] = c + 1;
        }
    }
    /* 3. gather */
    int64_t base = 0;
    int64_t *gsize = malloc((size_t)n * sizeof(int64_t)), *csize = malloc((size_t)n * sizeof(int64_t));
    int64_t *target = malloc((size_t)n * sizeof(int64_t));
    for (int parity = 0; parity < 2; parity++) {
        int64_t zmax = -1;
        for (int64_t z = parity; z < nzones; z += 2)
            if (zmax < 0 || zcolors[z] > zcolors[zmax]) zmax = z;
        if (zmax < 0) continue;
        const int64_t K = zcolors[zmax];
        for (int64_t g = 0; g < K; g++) gsize[g] = 0;
        for (int64_t v = 0; v < n; v++)
            if (zone[v] == zmax) {
                color_out[v] = base + col[v];
                gsize[col[v]]++;
            }
        for (int64_t z = parity; z < nzones; z += 2) {
            if (z == zmax) continue;
            for (int64_t c = 0; c < zcolors[z]; c++) csize[c] = 0, target[c] = -1;
            for (int64_t v = 0; v < n; v++)
                if (zone[v] == z) csize[col[v]]++;
            memset(used, 0, (size_t)n + 1);
            for (int64_t step = 0; step < zcolors[z]; step++) {
                int64_t c = -1; /* largest not yet placed colour, ties: smaller id */
                for (int64_t k = 0; k < zcolors[z]; k++)
                    if (target[k] < 0 && (c < 0 || csize[k] > csize[c])) c = k;
                int64_t g = -1; /* smallest unused global class, ties: smaller id */
                for (int64_t k = 0; k < K; k++)
                    if (!used[k] && (g < 0 || gsize[k] < gsize[g])) g = k;
                used[g] = 1;
                gsize[g] += csize[c];
                target[c] = g;
            }
            for (int64_t v = 0; v < n; v++)
                if (zone[v] == z) color_out[v] = base + target[col[v]];
        }
        base += K;
    }
    free(target);
    free(csize);
    free(gsize);
    free(used);
    free(zcolors);
    free(col);
    free(zone);
    free(adj);
    return base;
}

/* Validity check of any colouring against the conflict definition of reference
 * src/coloring.jl:58-60.  Returns 0 when valid, else 1 + the first offending index. */
int64_t orc_color_check(int64_t nblocks, const int64_t *const *idx, const int64_t *len,
                        int64_t maxindex, const int64_t *color) {
    /* for each index, colours seen so far must be distinct */
    int64_t bad = 0;
    int64_t *cnt = calloc((size_t)maxindex + 2, sizeof(int64_t));
    for (int64_t b = 0; b < nblocks; b++)
        for (int64_t k = 0; k < len[b]; k++) cnt[idx[b][k]]++;
    int64_t *ptr = malloc(((size_t)maxindex + 2) * sizeof(int64_t));
    ptr[0] = 0;
    for (int64_t i = 0; i <= maxindex; i++) ptr[i + 1] = ptr[i] + cnt[i];
    int64_t *inc = malloc((size_t)(ptr[maxindex + 1] + 1) * sizeof(int64_t));
    memset(cnt, 0, ((size_t)maxindex + 2) * sizeof(int64_t));
    for (int64_t b = 0; b < nblocks; b++)
        for (int64_t k = 0; k < len[b]; k++) {
            int64_t i = idx[b][k];
            inc[ptr[i] + cnt[i]++] = b;
        }
    for (int64_t i = 0; i <= maxindex && !bad; i++)
        for (int64_t p = ptr[i]; p < ptr[i + 1] && !bad; p++)
            for (int64_t q = p + 1; q < ptr[i + 1]; q++)
                if (inc[p] != inc[q] && color[inc[p]] == color[inc[q]]) {
                    bad = 1 + i;
                    break;
                }
    free(inc);
    free(ptr);
    free(cnt);
    return bad;
}

/* ------------------------------------------------------------------------------------
 * Timed loop for bench.py's `cpu_baseline` leg: `reps` forward VBCRS products (the 3-argument
 * form, alpha = 1, strong-zero beta) on PRE-MARSHALLED arguments, timed here in C so that no
 * Python / ctypes marshalling is inside the sample.  parallel != 0: the OpenMP variant (one task
 * per block row == the reference's `@tasks for browidx`, src/vbcrs.jl:275-276).
 * Returns the elapsed seconds.
 * ---------------------------------------------------------------------------------- */
#include <time.h>
double orc_vbcrs_bench_f64(int64_t reps, int parallel, int64_t nrows_y, int64_t nblockrows,
                           const int64_t *rowptr, const int64_t *colindices, const int64_t *rowindices,
                           const double *const *blocks, const int64_t *m, const int64_t *n,
                           const int64_t *ld, const double *x, double *y) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int64_t r = 0; r < reps; r++) {
        if (parallel)
            orc_vbcrs_mul_par_f64(nrows_y, nblockrows, rowptr, colindices, rowindices, blocks, m, n, ld, x, y, 1.0, 0.0, 1);
        else
            orc_vbcrs_mul_f64(nrows_y, nblockrows, rowptr, colindices, rowindices, blocks, m, n, ld, x, y, 1.0, 0.0, 1);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
