/* bsm_synth.h -- synthetic block-sparse operators generated IN HBM (bench / test utility of
 * libbsmrocm.so; not part of the reference's interface).
 *
 * The five BASELINE.json configurations are defined on portable counter-based SplitMix64 streams
 * (blocksparsematrices.jl_amd/synthetic.py):
 *     mix(z)      = SplitMix64 finaliser
 *     u(s, k)     = mix(s + 0x9E3779B97F4A7C15 * (k + 1))
 *     value       = (u >> 11) * 2^-53 * 2 - 1  in [-1, 1), computed in fp64, then cast to T
 *     block b     : column-major m x n, entry k drawn from stream mix(seed ^ mix(b + 1))
 *     x           : stream mix(seed ^ 0x5851F42D4C957F2D)
 * These two entry points fill DEVICE memory with exactly those values (bit-identical to the numpy
 * generator), so that a 16-29 GB operator (C4 / C5) exists in HBM in milliseconds and is handed to
 * bsm_*_create with bsm_options.blocks_memspace = BSM_MEM_DEVICE -- no matrix byte crosses PCIe.
 * dtype: BSM_F32 or BSM_F64.  All pointer ARRAYS are host memory; dst[...] are device pointers.
 */
#ifndef BSM_SYNTH_H
#define BSM_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* block ids[k] (the `b` of the recipe) of size m[k] x n[k] is written to dst[k] (leading dimension
 * m[k]).  symmetrise (may be NULL): != 0 stores (D + D^T) / 2 of the m x m draw instead (the diagonal
 * blocks of a SymmetricBlockMatrix, reference docs/src/symmetric.md:49-50).  Enqueued on `stream`
 * (a hipStream_t, NULL = default stream) of the current device; returns after the descriptors have
 * been uploaded, the fill itself is asynchronous. */
int bsm_synth_blocks(int dtype, uint64_t seed, int64_t nblocks, const int64_t *ids, const int64_t *m,
                     const int64_t *n, const int32_t *symmetrise, void *const *dst, void *stream);

/* the right-hand side x of configuration `seed`: entries [first, first + count) into dst */
int bsm_synth_vector(int dtype, uint64_t seed, int64_t first, int64_t count, void *dst, void *stream);

/* Measurement utility (bench.py: roofline.stream_floor_us, extra.cold_floor_us): ONE launch of a bare
 * streaming read of buf[0, bytes) with the request shape of the product kernels -- 8 independent 16-byte
 * non-temporal loads per lane, 8 KB per wave, nothing else to do.  hop != 0: every wave first reads its
 * position from a 64-byte record through a scalar load (the one dependent round trip a product wave
 * starts with: its descriptor).  scratch: device memory, 8 KB + (hop ? 64 bytes per 8 KB of `bytes` : 0).
 * Enqueued on `stream`; the first call with a given scratch buffer and hop != 0 uploads the record
 * table synchronously. */
int bsm_bench_stream(const void *buf, int64_t bytes, void *scratch, int64_t scratch_bytes, int hop, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BSM_SYNTH_H */
