// bsm_kernels.h -- interface between the C ABI glue and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

namespace bsm {

struct DeviceImage {
    int dtype = 1;
    int device = 0;
    long long nrows = 0, ncols = 0;
    long long own_lo = 0, own_hi = 0;  // 0-based [lo, hi) rows of y scaled by this handle
    void *d_values = nullptr, *d_rows = nullptr, *d_cols = nullptr;
    void *d_waves = nullptr;
    long long nwg_main = 0, nwg_total = 0;
    void *d_waves_multi = nullptr;  // coarser split of the same panels for the multi-RHS kernels (may be null)
    long long nwg_multi = 0;
    float mean_rows = 64.f;  // Analysis::mean_rows: fused fp32 products of short panels keep 4 loads per lane in flight
    float lane_fill = 1.f;  // Analysis::lane_fill: two right-hand sides go through a padded 4-column pass above 0.85
    int max_rows = 64;      // tallest row group of the image (the interleaved multi-RHS kernels: row blocks per panel)
    bool exclusive_fwd = false;
    bool has_off = false;  // SymmetricBlockMatrix off-diagonal pieces present
    long long device_bytes = 0;
    long long value_bytes = 0;  // packed matrix bytes (decides the cache policy of the matrix loads)
    std::vector<long long> color_wg_ptr;  // non-empty: coloured launches, plain read-modify-write
    // gather mode (BSM_ACC_GATHER): workspace + inverted indices for op N [0] and op T / C [1]
    void *d_ws = nullptr;
    void *d_inv_ptr[2] = {nullptr, nullptr}, *d_inv_idx[2] = {nullptr, nullptr};
    long long ws_fbase = 0;
};

// Enqueues y = alpha*op(A)*x + beta*y on `stream`.  x, y device pointers.  No allocation,
// no synchronisation (graph-capturable).
// opT: apply the transposed operator of the image; conj: conjugate every stored entry.
// use_gather: take the two-launch gather path (only if the image has a workspace).
hipError_t launch_mul(const DeviceImage &img, bool opT, bool conj, const void *x, void *y,
                      const void *alpha, const void *beta, int strong_zero, hipStream_t stream,
                      bool use_gather = false, const long long *zrange = nullptr);
// zrange = {lo, hi} (0-based, exclusive): the y entries the `y .*= beta` pass of the accumulate path
// covers instead of the image's own range (multi-device fan-out; ignored by exclusive forward images,
// whose coverage is part of the image)

// dst[0..n) += src[0..n);   y[0..n) = beta * y + r   (element type `dtype`, beta: pointer to one T)
hipError_t launch_vec_add(int dtype, void *dst, const void *src, long long n, hipStream_t stream);
hipError_t launch_vec_axpby(int dtype, void *y, const void *r, long long n, const void *beta, hipStream_t stream);

// Fused vector traffic of multi-device handles (peer-accessible devices): up to kMaxVecPieces sources per launch,
// each a virtual base pointer (element i of the global vector at base + i; column k of a multi-RHS batch
// `ld` elements further when strided) valid for indices [lo, hi).
constexpr int kMaxVecPieces = 8;
struct VecPieces {
    const void *base[kMaxVecPieces];
    long long lo[kMaxVecPieces], hi[kMaxVecPieces];
    int strided[kMaxVecPieces];
};
// dst[i + k * ld_dst] = piece(i)[k * ld_src] for every piece's range, K columns
hipError_t launch_vec_fetch(int dtype, void *dst, long long ld_dst, const VecPieces &pc, int npieces, long long ld_src, int K,
                            hipStream_t stream);
// y[i] = beta * y[i] + w[i] + sum of the pieces covering i, i in [lo, hi), K columns (w and the pieces: column stride
// ldw); accumulate_only: the sum goes back to w instead (more than kMaxVecPieces contributions)
hipError_t launch_vec_finish(int dtype, void *y, long long ldy, void *w, long long ldw, const VecPieces &pc, int npieces,
                             long long lo, long long hi, const void *beta, int strong_zero, int accumulate_only, int rezero, int K,
                             hipStream_t stream);

// y[lo_c + i] += base_c[i] for npieces <= kMaxVecPieces disjoint segments [lo_c, hi_c) of y, one launch
hipError_t launch_vec_add_segments(int dtype, void *y, const VecPieces &pc, int npieces, hipStream_t stream);

// Work arrays of the INTERLEAVED multi-RHS pass (bsm_kernels.hip: panel_kernel_il_*): X and the accumulated Y row-major,
// one 128-byte line per vector index.  Owned by the handle (bsm_capi.cpp: ILClaim), one product in flight.
struct ILWork {
    void *xr = nullptr;   // rows x 128 bytes: alpha * X, K-interleaved
    void *w = nullptr;    // rows x 128 bytes: the sums; zero between products (the finish pass zeroes behind its read)
    long long rows = 0;   // capacity of both, in vector entries
    bool w_clean = false; // w is known to be zero
};
// whether launch_mul_multi would take the interleaved pass for this image / op / batch (so that the caller only claims
// -- and allocates -- the work arrays when they will be used)
bool il_applies(const DeviceImage &img, bool opT, long long nrhs);

// nrhs right-hand sides: X (ldx) and Y (ldy) column-major; A is streamed once per batch of <= 8.
hipError_t launch_mul_multi(const DeviceImage &img, bool opT, bool conj, long long nrhs, const void *x,
                            long long ldx, void *y, long long ldy, const void *alpha, const void *beta,
                            int strong_zero, hipStream_t stream, const long long *zrange = nullptr, ILWork *il = nullptr);

// rowcolvals(A): COO triples (1-based int64 rows / cols, values of the image's element type) written from
// the packed device image; d_out_off[w] = first output slot of wave descriptor w (host prefix sum of
// m * ncols + m * #KIND_OFF columns)
hipError_t launch_export_coo(int dtype, const void *d_waves, long long nwaves, const void *d_out_off,
                             const void *d_values, const void *d_rows, const void *d_cols, void *orow, void *ocol,
                             void *oval, hipStream_t stream);

// executes Analysis::pack_plan on the device (blocks already in HBM): d_plan = PackChunk[nchunks],
// d_colpos = int32 placements (may be null when no chunk is scattered), es = element bytes
hipError_t launch_pack(int es, const void *d_plan, long long nchunks, const void *d_colpos, void *d_values,
                       hipStream_t stream);

// synthetic operators generated in HBM (include/bsm_synth.h); d_desc = SynthBlock[nblocks]
struct SynthBlockDesc {
    uint64_t dst, stream;
    int32_t m, n, symmetrise, pad;
};
// bare streaming read of `bytes` at src (bsm_bench_stream): sink >= 8 KB of scratch, hop = optional table of one
// 64-byte record per wave whose first int64 is the wave's position (a dependent scalar load per wave)
hipError_t launch_stream_floor(const void *src, long long bytes, void *sink, const void *hop, hipStream_t stream);
hipError_t launch_synth_blocks(int dtype, const void *d_desc, long long nblocks, int tiles, hipStream_t stream);
hipError_t launch_synth_vector(int dtype, void *dst, long long n, unsigned long long stream_seed, hipStream_t stream);

}  // namespace bsm
