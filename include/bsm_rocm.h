/* bsm_rocm.h -- C ABI of libbsmrocm.so: the MI355X (gfx950) block-sparse mat-vec engine.
 *
 * Drop-in boundary for ONE hot path of djukic14/BlockSparseMatrices.jl: the
 * `LinearMaps._unsafe_mul!(y, A, x[, alpha, beta])` methods of its three storage types
 * (and their Adjoint/Transpose wrappers) plus the constructors that feed them.
 * The reference is pure Julia and has no FFI; a Julia maintainer binds these entry
 * points with `ccall` from methods of the same names (INTEGRATION.md shows the stub).
 *
 * Conventions (chosen so a Julia caller passes its data untouched):
 *   - every index is 1-BASED int64 (Julia Int);
 *   - a block is a column-major m x n array with leading dimension ld (a Julia Matrix);
 *   - `blocks` is an array of nblocks pointers (pointer.(blocks) of a Vector{Matrix});
 *   - *_create COPIES everything into library-owned device memory (repacked for
 *     coalesced 16-byte lane loads); the caller may free its arrays afterwards;
 *   - all functions return 0 on success, a negative bsm_status otherwise, never throw;
 *     bsm_last_error() returns a thread-local message for the last failure;
 *   - a handle is immutable after creation: concurrent bsm_mul calls with distinct y
 *     (and distinct streams) are legal.
 */
#ifndef BSM_ROCM_H
#define BSM_ROCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsm_matrix_s *bsm_matrix_t;

typedef enum {
    BSM_OK = 0,
    BSM_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, index out of range) */
    BSM_ERR_UNSUPPORTED = -2,
    BSM_ERR_DEVICE = -3,    /* HIP runtime error, or no device image (analysis-only handle) */
    BSM_ERR_ALLOC = -4
} bsm_status;

/* element type T of blocks, x, y, alpha, beta */
typedef enum { BSM_F32 = 0, BSM_F64 = 1, BSM_C64 = 2, BSM_C128 = 3 } bsm_dtype;

/* which operator of A is applied: A, transpose(A), A' -- the reference's
 * LinearMaps.TransposeMap / AdjointMap wrappers (src/blockmatrix.jl:154-160,200-206,
 * src/symmetricblockmatrix.jl:345-365, src/vbcrs.jl:298-354) */
typedef enum { BSM_OP_N = 0, BSM_OP_T = 1, BSM_OP_C = 2 } bsm_op;

/* where x and y live */
typedef enum { BSM_MEM_HOST = 0, BSM_MEM_DEVICE = 1 } bsm_memspace;

/* reference `scheduler=` keyword: SerialScheduler() gives the single colour
 * [1:nblocks] (src/blockmatrix.jl:91-92), anything else colours the blocks
 * (src/blockmatrix.jl:94-98, src/symmetricblockmatrix.jl:104-110) */
typedef enum { BSM_SCHED_SERIAL = 0, BSM_SCHED_DYNAMIC = 1 } bsm_scheduler;

/* how contributions of different blocks to the same y entries are combined on the GPU */
typedef enum {
    BSM_ACC_AUTO = 0,    /* exclusive direct stores when provably conflict-free, else atomics.  Large
                            conflict-free operators made of deep row groups (most bytes in row groups
                            above 128 KiB, e.g. 128x128 blocks, 16 per block row) are ALSO scheduled as
                            32 KiB work items combined with atomics: 8 % faster, but the last bits then
                            depend on the order of the adds -- BSM_ACC_DIRECT keeps them exclusive */
    BSM_ACC_ATOMIC = 1,  /* hardware fp atomics into y, blocks ordered by colour class */
    BSM_ACC_COLORED = 2, /* one launch per colour class, plain read-modify-write: bitwise
                            reproducible run to run (the reference's own scheme) */
    BSM_ACC_GATHER = 3,  /* no atomics at all: every block contribution is stored once in a
                            workspace owned by the handle and a second launch sums, per y entry,
                            its contributions in a fixed order (and applies alpha, beta): bitwise
                            reproducible, two launches.  The workspace makes products on ONE handle
                            stream-ordered: a call that finds another product of the same handle
                            still in flight on a DIFFERENT stream (or being enqueued by another
                            thread) uses the atomic path for that call.  Single right-hand side
                            only. */
    BSM_ACC_DIRECT = 4   /* like AUTO, but a conflict-free operator always takes the exclusive direct
                            stores (one launch, no atomics, bitwise reproducible) */
} bsm_accumulate;

/* colouring algorithms of bsm_color / bsm_options.coloring (GraphsColoring.jl's names) */
typedef enum { BSM_COLOR_WORKSTREAM_DSATUR = 0, BSM_COLOR_DSATUR = 1 } bsm_coloring;

#define BSM_DEVICE_CURRENT (-1)
#define BSM_DEVICE_NONE (-2) /* analysis only: bookkeeping queries work, bsm_mul fails */

typedef struct {
    int32_t struct_size; /* = sizeof(bsm_options) */
    int32_t device;      /* HIP ordinal, BSM_DEVICE_CURRENT or BSM_DEVICE_NONE */
    int32_t scheduler;   /* bsm_scheduler */
    int32_t accumulate;  /* bsm_accumulate */
    int32_t validate;    /* kept for ABI stability: every index is ALWAYS range-checked at create time
                            (an out-of-range index must never reach a kernel) */
    /* 1: VBCRS / BlockSparseMatrix handles also keep a SECOND, transposed ordering of the blocks
     * (the reference's own TODO, src/vbcrs.jl:124): transpose(A)*x and A'*x then run as a forward
     * product on it -- one launch, no atomics, bitwise reproducible -- at twice the device
     * memory.  0 (default): the transposed products run on the single image with atomics.
     * 2: build the second ordering when it is cheap -- packed values of at most 1/16 of the device's
     * free memory -- else behave like 0 (the default of the Julia binding's ROCmScheduler: Krylov
     * solvers that apply A' every iteration get the one-launch transposed product). */
    int32_t transpose_image;
    /* rows of y this handle is responsible for scaling by beta (1-based, inclusive);
     * 0,0 = all rows.  Used when block rows are partitioned over several GPUs. */
    int64_t own_lo, own_hi;
    /* non-NULL: a bsm_ctx_t (below).  The handle is then spread over the context's devices -- the
     * block rows are partitioned among them, bsm_mul fans out to one stream per device and exchanges
     * the overlapping y segments over xGMI -- exactly where the reference has its `@tasks` fan-out
     * (src/vbcrs.jl:275-276, src/blockmatrix.jl:233-245, src/symmetricblockmatrix.jl:395-432).
     * `device`, own_lo/own_hi and transpose_image are ignored for such a handle. */
    void *ctx;
    /* where the BLOCK arrays passed to *_create live: BSM_MEM_HOST (0, default) or BSM_MEM_DEVICE:
     * device pointers valid on the handle's device (e.g. AMDGPU.jl ROCArrays); the repacking then
     * runs as a kernel on that device and no matrix byte crosses PCIe.  Index lists, sizes and the
     * pointer arrays themselves are always host memory. */
    int64_t blocks_memspace;
    /* the reference's `coloringalgorithm=` keyword (src/blockmatrix.jl:67,86): which algorithm produces
     * the colour classes reported through bsm_get_bookkeeping -- bsm_coloring, default
     * BSM_COLOR_WORKSTREAM_DSATUR like the reference (src/BlockSparseMatrices.jl:10) */
    int64_t coloring;
    int64_t reserved[1];
} bsm_options;

/* The layout the bindings mirror field by field (julia/BlockSparseMatricesROCm.jl: BsmOptions, BsmPartInfo;
 * blocksparsematrices.jl_amd/_lib.py: BsmOptions, BsmPartInfo, BsmStats; tests/test_c_abi_from_c.py asserts the
 * same numbers against the ctypes mirror): a field added or moved here fails the build instead of drifting silently
 * away from a binding that cannot be compiled against this header. */
#if defined(__cplusplus) && __cplusplus >= 201103L
#define BSM_LAYOUT_ASSERT(cond, msg) static_assert(cond, msg)
#elif !defined(__cplusplus) && defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
#define BSM_LAYOUT_ASSERT(cond, msg) _Static_assert(cond, msg)
#else /* C99 / pre-C++11 consumers of the C ABI: the header still compiles, the checks are the library build's */
#define BSM_LAYOUT_ASSERT(cond, msg) struct bsm_layout_assert_unused_
#endif
BSM_LAYOUT_ASSERT(sizeof(bsm_options) == 72, "bsm_options is 72 bytes");
BSM_LAYOUT_ASSERT(offsetof(bsm_options, device) == 4 && offsetof(bsm_options, scheduler) == 8 &&
                      offsetof(bsm_options, accumulate) == 12 && offsetof(bsm_options, validate) == 16 &&
                      offsetof(bsm_options, transpose_image) == 20,
                  "bsm_options: six int32 fields first");
BSM_LAYOUT_ASSERT(offsetof(bsm_options, own_lo) == 24 && offsetof(bsm_options, own_hi) == 32 &&
                      offsetof(bsm_options, ctx) == 40 && offsetof(bsm_options, blocks_memspace) == 48 &&
                      offsetof(bsm_options, coloring) == 56 && offsetof(bsm_options, reserved) == 64,
                  "bsm_options: 8-byte fields from offset 24");

/* ---- several GPUs of one node behind ONE handle -------------------------------------------------
 * The reference runs its block rows / colour classes as tasks of one process (OhMyThreads `@tasks`
 * with the scheduler stored in the matrix).  The MI355X counterpart of that fan-out is a context of
 * devices: a handle created with bsm_options.ctx set keeps one packed image per device (contiguous
 * ranges of block rows, balanced by stored bytes) and bsm_mul / bsm_mul_multi
 *   1. make x available on every device (host memory: one H2D copy per device over its own PCIe
 *      link; device memory: peer copies from the device that holds x),
 *   2. run the local products concurrently, one stream per device,
 *   3. exchange ONLY the y segments a device produced for rows another device owns (symmetric /
 *      index-list operators: the halo; transposed products of a row partition: a reduce-scatter
 *      onto equal column chunks) as direct peer-to-peer copies over the xGMI links + a local add,
 *   4. deliver the owned y ranges to the caller's y (host memory or the device that holds y).
 * VBCRS forward products need no step 3 (block rows own disjoint y ranges, src/vbcrs.jl:275-283).
 * The same device may be listed several times (virtual devices: how the exchange is tested on a
 * one-GPU machine).  One process per GPU with RCCL collectives is the OTHER way to use several GPUs
 * (blocksparsematrices.jl_amd/distributed.py: every rank creates an ordinary single-device handle
 * with own_lo/own_hi from bsm_partition_rows). */
typedef struct bsm_ctx_s *bsm_ctx_t;
int bsm_ctx_create(const int32_t *device_ids, int32_t ndevices, bsm_ctx_t *out);
int bsm_ctx_destroy(bsm_ctx_t ctx); /* after every handle created with it has been destroyed */
/* devices of the context (device_ids may be NULL); returns the count through *ndevices */
int bsm_ctx_devices(bsm_ctx_t ctx, int32_t *ndevices, int32_t *device_ids, int32_t capacity);

/* The row partition both multi-GPU layers use.  Block b has row key rowkey[b] (its smallest row
 * index, 1-based) and weight[b] (stored entries).  The distinct keys, ascending, are cut into
 * nparts contiguous ranges of about equal weight; part_of_block[b] receives the part of block b,
 * own_lo[p] / own_hi[p] (1-based, inclusive; own_hi = own_lo - 1 for an empty part) the rows part p
 * owns: from its first key up to the next part's first key - 1 (part 0 from row 1, the last one to
 * nrows).  Deterministic; blocks with equal keys always land in the same part. */
int bsm_partition_rows(int64_t nrows, int64_t nblocks, const int64_t *rowkey, const int64_t *weight,
                       int32_t nparts, int32_t *part_of_block, int64_t *own_lo, int64_t *own_hi);

/* Per-device view of a multi-device handle (tests, reports): rows owned / touched by part p
 * (1-based inclusive), its device ordinal and the bytes of its packed image. */
typedef struct {
    int32_t device, reserved32;
    int64_t own_lo, own_hi, touched_lo, touched_hi, device_bytes, nblocks;
    /* the part's share of the COLUMN partition (1-based inclusive): what it holds of a vector of length
     * size(A,2) in bsm_mul_parts.  Square operators: the row partition itself (own_lo..own_hi), so that the
     * y parts of one product are the x parts of the next; otherwise equal chunks of the columns. */
    int64_t col_lo, col_hi;
    int64_t reserved[2];
} bsm_part_info_t;
BSM_LAYOUT_ASSERT(sizeof(bsm_part_info_t) == 88 && offsetof(bsm_part_info_t, own_lo) == 8 &&
                      offsetof(bsm_part_info_t, col_lo) == 56 && offsetof(bsm_part_info_t, reserved) == 72,
                  "bsm_part_info_t layout");
int bsm_part_info(bsm_matrix_t A, int32_t part, bsm_part_info_t *out);

/* fills *o with defaults (device = current, serial scheduler, auto accumulate, validate) */
void bsm_options_default(bsm_options *o);

/* VariableBlockCompressedRowStorage(matrices, rowindices, colindices, size; scheduler)
 * -- reference src/vbcrs.jl:78-122.  rowstart/colstart: first row/column of each block
 * (1-based), blocks in any order; the library sorts them exactly as the reference does
 * (stable by (rowstart, colstart)) and builds rowptr.  nblocks >= 1. */
int bsm_vbcrs_create(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                     const void *const *blocks, const int64_t *m, const int64_t *n,
                     const int64_t *ld, const int64_t *rowstart, const int64_t *colstart,
                     const bsm_options *opts, bsm_matrix_t *out);

/* VariableBlockCompressedRowStorage(sbm::SymmetricBlockMatrix; scheduler) -- reference
 * src/vbcrs.jl:189-264.  The reference expands the matrix into [diagonals..., offdiagonals...,
 * transpose(offdiagonals)...] and MATERIALISES the transposes (twice the off-diagonal storage).
 * Here the bookkeeping (perm, rowptr, colindices, rowindices over those ndiag + 2*noff virtual
 * blocks) is identical, but the device image keeps every off-diagonal block once and the product
 * applies it and its transpose from one read.  Like the reference's converter only the FIRST
 * index of every list is used (contiguous ranges are assumed, src/vbcrs.jl:183-184,230-240). */
int bsm_vbcrs_create_from_symmetric(int dtype, int64_t nrows, int64_t ncols, int64_t ndiag,
                                    const void *const *diag, const int64_t *dsize,
                                    const int64_t *dld, const int64_t *diagstart, int64_t noff,
                                    const void *const *off, const int64_t *m, const int64_t *n,
                                    const int64_t *ld, const int64_t *rowstart,
                                    const int64_t *colstart, const bsm_options *opts,
                                    bsm_matrix_t *out);

/* VariableBlockCompressedRowStorage(bsm::BlockSparseMatrix; scheduler) -- reference
 * src/vbcrs.jl:150-160 with the functors of :201-215: block i is placed at
 * (first(rowindices(bsm, i)), first(colindices(bsm, i))), i.e. only the FIRST entry of every index
 * list is used (contiguous ranges are assumed, like the reference's "no sanity checks", :146).
 * Arguments as bsm_blocksparse_create; blocks with an empty list are rejected. */
int bsm_vbcrs_create_from_blocksparse(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                                      const void *const *blocks, const int64_t *m, const int64_t *n,
                                      const int64_t *ld, const int64_t *const *rowidx,
                                      const int64_t *const *colidx, const bsm_options *opts,
                                      bsm_matrix_t *out);

/* BlockSparseMatrix(blocks, rowindices, colindices, size; scheduler, coloringalgorithm)
 * -- reference src/blockmatrix.jl:62-109.  rowidx[b] has m[b] entries, colidx[b] n[b]. */
int bsm_blocksparse_create(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                           const void *const *blocks, const int64_t *m, const int64_t *n,
                           const int64_t *ld, const int64_t *const *rowidx,
                           const int64_t *const *colidx, const bsm_options *opts,
                           bsm_matrix_t *out);

/* SymmetricBlockMatrix(diagonals, diagonalindices, offdiagonals, rowindices, colindices,
 * size; scheduler) -- reference src/symmetricblockmatrix.jl:73-126.  Diagonal block d is
 * dsize[d] x dsize[d] on index list diagidx[d]; only one triangle of the off-diagonal
 * blocks is passed, the product applies each of them and its transpose. */
int bsm_symmetric_create(int dtype, int64_t nrows, int64_t ncols, int64_t ndiag,
                         const void *const *diag, const int64_t *dsize, const int64_t *dld,
                         const int64_t *const *diagidx, int64_t noff, const void *const *off,
                         const int64_t *m, const int64_t *n, const int64_t *ld,
                         const int64_t *const *rowidx, const int64_t *const *colidx,
                         const bsm_options *opts, bsm_matrix_t *out);

/* y = alpha * op(A) * x + beta * y  -- LinearMaps._unsafe_mul!(y, A, x, alpha, beta):
 * reference src/blockmatrix.jl:225-247, src/symmetricblockmatrix.jl:386-435,
 * src/vbcrs.jl:266-288 (forward), :303-354 (adjoint/transpose).
 *   alpha, beta : pointers to one T each (host memory); NULL = 1 resp. 0.
 *   beta_strong_zero != 0 : beta is Julia's Bool `false` (the 3-arg form,
 *     src/abstractblockmatrix.jl:27-34): y is OVERWRITTEN, NaN/Inf in the incoming y do
 *     not propagate.  With beta_strong_zero == 0 a numeric beta = 0 multiplies.
 *   memspace BSM_MEM_DEVICE: x, y are device pointers valid on the handle's device; the
 *     call only enqueues work on `stream` (a hipStream_t, NULL = default stream) and
 *     returns; no allocation or synchronisation happens, so it can be graph-captured (single-device handles
 *     only: a multi-device handle issues on several streams and devices and must not be captured; a
 *     BSM_ACC_GATHER handle takes its atomic path while the stream is capturing -- the workspace of the
 *     gather path admits one product in flight, which a replayed graph could not promise).
 *   memspace BSM_MEM_HOST: x, y are host arrays; the library stages them through device
 *     buffers and returns when y is complete.
 * x has size(op(A),2) entries, y size(op(A),1); they must not alias. */
int bsm_mul(bsm_matrix_t A, int op, const void *x, void *y, const void *alpha,
            const void *beta, int beta_strong_zero, int memspace, void *stream);

/* bsm_mul for a multi-device handle (bsm_options.ctx) with x and y PARTITIONED over its devices -- what an
 * iterative solver on N GPUs holds: block rows own disjoint y ranges (reference src/vbcrs.jl:275-283), so
 * nobody ever needs the whole of x or y in one place.  Part p (bsm_part_info) passes
 *   x_parts[p]: device pointer ON ITS DEVICE to the x entries it holds -- op N: columns col_lo..col_hi,
 *               op T / C: rows own_lo..own_hi (first entry of the range at x_parts[p][0]);
 *   y_parts[p]: device pointer on its device to the y entries it receives -- op N: rows own_lo..own_hi,
 *               op T / C: columns col_lo..col_hi.  Empty ranges may pass NULL.
 * Only what a device's blocks read beyond its own part travels (the x halo), and only the y segments it
 * produced for rows of another device (symmetric / index-list operators; a reduce-scatter for products across
 * the partition): both as ONE fused kernel per device that reads its peers' memory over xGMI -- no copy of the
 * full x to anybody, no staging.  streams[p] (may be NULL = the device's default stream; the array itself may be
 * NULL): the stream of part p's device on which x_parts[p] / y_parts[p] are produced and consumed; the call
 * orders its work behind them and them behind its result, and returns without synchronising.
 * Needs peer access between all devices of the context (BSM_ERR_UNSUPPORTED otherwise).  alpha, beta,
 * beta_strong_zero as in bsm_mul. */
int bsm_mul_parts(bsm_matrix_t A, int op, const void *const *x_parts, void *const *y_parts, const void *alpha,
                  const void *beta, int beta_strong_zero, void *const *streams);

/* A HIP stream for the products of a rank that exchanges vector segments with its neighbours WHILE it multiplies (one
 * process per GPU, RCCL): its CU mask leaves `reserved_cus` compute units -- rounded up to a multiple of 8, one per XCD
 * -- to the collective layer's kernels.  Beside a product launch that fills every CU those kernels otherwise wait for
 * the launch to drain (measured: DESIGN.md section 5b).  0 = an ordinary non-blocking stream.  Pass the stream to
 * bsm_mul / bsm_mul_multi like any other (`stream` argument); the reference has no counterpart (shared-memory tasks,
 * src/symmetricblockmatrix.jl:395-432). */
int bsm_stream_create_reserved(int device, int reserved_cus, void **stream);
int bsm_stream_destroy(void *stream);

/* y[offset[s] + i] += src[s][i], i < len[s], for nseg DISJOINT segments of a device vector y, in ONE launch on
 * `stream` (dtype as in the *_create calls; offsets 0-based, in elements).  The delivery step of a row-partitioned
 * product in a process-per-GPU layer above this ABI (blocksparsematrices.jl_amd/distributed.py: the own rows of the
 * boundary blocks' sums + every partial-y segment received from a neighbour -- the y segments other tasks of the
 * reference's fan-out would have added in shared memory, src/symmetricblockmatrix.jl:407-418): one launch behind
 * the join of the exchange instead of one per segment.  Overlapping segments are refused (BSM_ERR_INVALID). */
int bsm_vec_add_segments(int dtype, void *y, int32_t nseg, const int64_t *offset, const void *const *src,
                         const int64_t *len, void *stream);

/* Page-locks a HOST vector the caller keeps using as x or y of BSM_MEM_HOST products (a Julia
 * Vector{T} that lives through a solver loop): bsm_mul then moves it by DMA straight from / to the
 * caller's memory instead of copying it through the library's pinned mirrors (C2-sized product:
 * measured in DESIGN.md section 6).  The memory must stay allocated until bsm_host_unregister; the
 * Julia binding registers in the constructor of its vector wrapper and unregisters in the finalizer.
 * (hipHostRegister / hipHostUnregister, exported so that the host language needs no HIP binding.) */
int bsm_host_register(void *ptr, int64_t bytes);
int bsm_host_unregister(void *ptr);

/* Y = alpha * op(A) * X + beta * Y for nrhs right-hand sides -- `A * X` / `mul!(Y, A, X, a, b)`
 * with matrices.  LinearMaps loops the columns of X through _unsafe_mul! (nrhs full sweeps of A);
 * here A is streamed ONCE per batch of up to 8 columns (16 for Float32 / Float64 matrices from 9 columns on; a
 * remainder is one padded pass; nothing outside the nrhs columns of X and Y is read or written).  The 8-column
 * passes of the complex types and the 16-column passes of the real ones run on the matrix pipe (8 complex
 * columns = 16 real ones = N of v_mfma_{f64,f32}_16x16x4).  X is size(op(A),2) x nrhs
 * and Y is size(op(A),1) x nrhs, both column-major with leading dimensions ldx / ldy (elements).
 * Every other argument as in bsm_mul; each column gives what nrhs = 1 semantics prescribe (same
 * alpha, beta, strong zero). */
int bsm_mul_multi(bsm_matrix_t A, int op, int64_t nrhs, const void *X, int64_t ldx, void *Y,
                  int64_t ldy, const void *alpha, const void *beta, int beta_strong_zero, int memspace,
                  void *stream);

/* Bookkeeping queries (bit-exact contract; every value 1-based int64 like the reference).
 * Call with out == NULL to obtain the required length in *len. */
typedef enum {
    BSM_BK_VBCRS_PERM = 0,        /* sortperm of src/vbcrs.jl:84 */
    BSM_BK_VBCRS_ROWPTR = 1,      /* src/vbcrs.jl:97-117, length nblockrows+1 */
    BSM_BK_VBCRS_COLINDICES = 2,  /* per block, sorted order */
    BSM_BK_VBCRS_ROWINDICES = 3,  /* per block ROW */
    /* colour sets, flattened as [ncolors, len_1, ids_1..., len_2, ids_2..., ...] */
    BSM_BK_COLORS = 4,            /* BlockSparseMatrix.colors / Symmetric offdiagonalcolors */
    BSM_BK_TRANSPOSECOLORS = 5,   /* transposecolors / transposeoffdiagonalcolors */
    BSM_BK_DIAGONALCOLORS = 6     /* Symmetric diagonalcolors */
} bsm_bookkeeping;
int bsm_get_bookkeeping(bsm_matrix_t A, int which, int64_t *out, int64_t *len);

/* rowcolvals(A) -- reference src/sparse.jl:17-123, the COO triples behind `sparse(A)`
 * (src/sparse.jl:125-129): written by a kernel straight from the packed DEVICE image -- every stored
 * entry once, the off-diagonal blocks of a SymmetricBlockMatrix a second time transposed -- so that a
 * CSR / CSC matrix can be assembled on the GPU without the blocks ever returning to the host.
 * rows / cols: 1-based int64, vals: the handle's element type, all with room for *count entries
 * (call with NULL arrays to obtain the count = nnz(A) as the reference defines it).  The order of the
 * triples is fixed but unspecified (`sparse` sums duplicates, like mul!'s +=).  memspace: where the
 * three arrays live (BSM_MEM_DEVICE: on the handle's device).  Synchronous. */
int bsm_rowcolvals(bsm_matrix_t A, int64_t *rows, int64_t *cols, void *vals, int64_t *count, int memspace,
                   void *stream);

/* Statistics of a handle. */
typedef struct {
    int64_t nnz;            /* SparseArrays.nnz as the reference defines it (off-diagonal
                               blocks of a SymmetricBlockMatrix count twice,
                               src/symmetricblockmatrix.jl:367-384) */
    int64_t stored_entries; /* matrix entries held on the device (each stored once) */
    int64_t alg_bytes;      /* algorithmic bytes of one mul with beta = 0 */
    int64_t device_bytes;   /* bytes of the packed device image (values + metadata) */
    int64_t npanels, ntasks, nworkgroups;
    int64_t exclusive;      /* 1: forward product needs no atomics and no pre-scale pass */
    /* SymmetricBlockMatrix, op N: y contributions the fused launch produces (forward rows + transposed
     * columns of every wave), how many of them are added up in a workgroup's LDS window first, and how
     * many entries those windows then add to y -- global atomics per product =
     * win_emissions - win_inside + win_flushed */
    int64_t win_emissions, win_inside, win_flushed;
    int64_t reserved[5];
} bsm_stats_t;
BSM_LAYOUT_ASSERT(sizeof(bsm_stats_t) == 128 && offsetof(bsm_stats_t, win_emissions) == 64, "bsm_stats_t layout");
int bsm_stats(bsm_matrix_t A, bsm_stats_t *out);

/* Debug / test hook: copies one array of the packed device image (host copy) out.
 * which: 0 values (bytes), 1 rows (int32), 2 cols (int32), 3 waves (64-byte records;
 * layout in blocksparsematrices.jl_amd/csrc/bsm_layout.h); gather handles also 4 / 5 = row
 * pointers (int64) / workspace slots (int32) of the op-N inverted index and 6 / 7 for op T;
 * 8 = the coarser wave records bsm_mul_multi walks (empty when the records of 3 serve both).
 * Add 16 to `which` for the arrays of the transposed image (bsm_options.transpose_image).
 * Only available on analysis-only handles (BSM_DEVICE_NONE), which keep the host copy.
 * Call with out == NULL to obtain the size in bytes. */
int bsm_get_image(bsm_matrix_t A, int which, void *out, int64_t *nbytes);

/* color(conflictgraph(ColorInfo(lists)); algorithm).colors -- reference src/coloring.jl:15-61 +
 * GraphsColoring.jl (compat 0.2.0, NOT in the reference tree; the reference's default algorithm is its
 * WorkstreamDSATUR, src/BlockSparseMatrices.jl:10).  Two lists conflict iff they share an index.
 * algorithm: BSM_COLOR_WORKSTREAM_DSATUR -- the published WorkStream colouring (Turcksin, Kronbichler,
 * Bangerth, ACM TOMS 43, 2016, section 3.2: zones = breadth-first layers of the conflict graph, DSATUR
 * inside every zone, classes of the even and of the odd zones gathered largest-to-smallest) -- or
 * BSM_COLOR_DSATUR (plain DSATUR on the whole graph).
 * CONTRACT: the classes returned here (and by BSM_BK_COLORS / _TRANSPOSECOLORS / _DIAGONALCOLORS) are
 * VALID -- they partition 1..nlists and no two lists of a class share an index, which is all the
 * reference's mul! relies on (src/blockmatrix.jl:233-245) -- and DETERMINISTIC (every tie-break is
 * specified in oracle/bsm_oracle.c and compared bit-exactly in tests/test_host_logic.py).  They follow
 * the published algorithm, but are NOT claimed to be identical to GraphsColoring's output: its source
 * and version are not available here and no reference test inspects colour classes.  The serial
 * scheduler's single class [1:nblocks] (src/blockmatrix.jl:91-92) IS reproduced exactly.  The GPU
 * product does not depend on the classes (BSM_ACC_COLORED colours row GROUPS itself, with DSATUR).
 * lists[b] has lens[b] 1-based entries; color_out[b] receives the 0-based colour of list b;
 * *ncolors the number of colours. */
int bsm_color(int64_t nlists, const int64_t *const *lists, const int64_t *lens, int algorithm,
              int64_t *color_out, int64_t *ncolors);

int bsm_destroy(bsm_matrix_t A);

/* thread-local message of the last failing call in this thread ("" if none) */
const char *bsm_last_error(void);

/* library / build identification, e.g. "bsmrocm 0.3 gfx950 build 1a2b3c4d5e6f": the build id is a hash of the
 * kernel and schedule sources, so that a measurement can be tied to what actually ran */
const char *bsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BSM_ROCM_H */
