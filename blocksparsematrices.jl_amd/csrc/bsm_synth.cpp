// bsm_synth.cpp -- include/bsm_synth.h: synthetic operators of BASELINE.json generated in HBM
// (bench / test utility; kernels in bsm_kernels.hip).
#include <cstring>
#include <vector>

#include "../../include/bsm_synth.h"
#include "bsm_internal.h"

using namespace bsm;

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" int bsm_synth_blocks(int dtype, uint64_t seed, int64_t nblocks, const int64_t *ids, const int64_t *m,
                                const int64_t *n, const int32_t *symmetrise, void *const *dst, void *stream) {
    try {
        if (dtype != BSM_F32 && dtype != BSM_F64) return fail(BSM_ERR_UNSUPPORTED, "bsm_synth: fp32 / fp64 only");
        if (nblocks < 0 || (nblocks > 0 && (!ids || !m || !n || !dst))) return fail(BSM_ERR_INVALID, "null argument");
        if (nblocks == 0) return BSM_OK;
        std::vector<SynthBlockDesc> desc((size_t)nblocks);
        int64_t biggest = 0;
        for (int64_t k = 0; k < nblocks; k++) {
            if (m[k] < 0 || n[k] < 0 || m[k] > INT32_MAX || n[k] > INT32_MAX || (m[k] * n[k] > 0 && !dst[k]))
                return fail(BSM_ERR_INVALID, "bad block " + std::to_string(k + 1));
            const bool sym = symmetrise && symmetrise[k];
            if (sym && m[k] != n[k]) return fail(BSM_ERR_INVALID, "only square blocks can be symmetrised");
            desc[k].dst = (uint64_t)(uintptr_t)dst[k];
            desc[k].stream = mix64(seed ^ mix64((uint64_t)ids[k] + 1));
            desc[k].m = (int32_t)m[k];
            desc[k].n = (int32_t)n[k];
            desc[k].symmetrise = sym ? 1 : 0;
            desc[k].pad = 0;
            biggest = std::max(biggest, m[k] * n[k]);
        }
        void *d = nullptr;
        hipStream_t st = (hipStream_t)stream;
        hipError_t e = hipMalloc(&d, desc.size() * sizeof(SynthBlockDesc));
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        e = hipMemcpyAsync(d, desc.data(), desc.size() * sizeof(SynthBlockDesc), hipMemcpyHostToDevice, st);
        // one workgroup per 16 K entries of the largest block (grid.y), every block in grid.x
        const int tiles = (int)std::min<int64_t>(64, std::max<int64_t>(1, (biggest + 16383) / 16384));
        // grid.x is limited to 2^31 - 1 blocks; launch in slabs to be safe
        const int64_t slab = 1 << 24;
        for (int64_t b0 = 0; b0 < nblocks && e == hipSuccess; b0 += slab)
            e = launch_synth_blocks(dtype, (const char *)d + (size_t)b0 * sizeof(SynthBlockDesc),
                                    std::min(slab, nblocks - b0), tiles, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host descriptor array goes away
        (void)hipFree(d);
        if (e != hipSuccess) return hip_fail(e, "bsm_synth_blocks");
        return BSM_OK;
    } catch (const std::bad_alloc &) {
        return fail(BSM_ERR_ALLOC, "out of host memory");
    }
}

extern "C" int bsm_synth_vector(int dtype, uint64_t seed, int64_t first, int64_t count, void *dst, void *stream) {
    if (dtype != BSM_F32 && dtype != BSM_F64) return fail(BSM_ERR_UNSUPPORTED, "bsm_synth: fp32 / fp64 only");
    if (first < 0 || count < 0 || (count > 0 && !dst)) return fail(BSM_ERR_INVALID, "bad argument");
    // entry k of the stream is u(s, k): a sub-range starts at counter `first`
    const uint64_t s = mix64(seed ^ 0x5851F42D4C957F2Dull) + 0x9E3779B97F4A7C15ull * (uint64_t)first;
    hipError_t e = launch_synth_vector(dtype, dst, count, s, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "bsm_synth_vector");
    return BSM_OK;
}

// One launch of the bare streaming read (stream_floor_kernel) over buf[0, bytes).  `scratch` must hold 8 KB +
// (hop ? 64 bytes per 8 KB of `bytes` : 0): with hop != 0 its tail is filled (once per call signature: every
// call, it is cheap) with the identity table the waves read their position from.
extern "C" int bsm_bench_stream(const void *buf, int64_t bytes, void *scratch, int64_t scratch_bytes, int hop,
                                void *stream) {
    if (!buf || bytes < 16 || !scratch) return fail(BSM_ERR_INVALID, "bad argument");
    const int64_t nwaves = ((bytes / 16 + 2047) / 2048) * 4;
    const int64_t need = 8192 + (hop ? nwaves * 64 : 0);
    if (scratch_bytes < need) return fail(BSM_ERR_INVALID, "scratch too small: " + std::to_string(need) + " bytes needed");
    hipStream_t st = (hipStream_t)stream;
    const void *table = nullptr;
    if (hop) {
        // the table is written by the host once per scratch buffer: its first word tells whether it is there
        static thread_local const void *filled = nullptr;
        static thread_local int64_t filled_waves = 0;
        char *t = (char *)scratch + 8192;
        if (filled != scratch || filled_waves < nwaves) {
            std::vector<int64_t> h((size_t)nwaves * 8, 0);
            for (int64_t w = 0; w < nwaves; w++) h[(size_t)w * 8] = w;
            hipError_t e = hipMemcpy(t, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            if (e != hipSuccess) return hip_fail(e, "hop table upload");
            filled = scratch;
            filled_waves = nwaves;
        }
        table = t;
    }
    hipError_t e = launch_stream_floor(buf, bytes, scratch, table, st);
    if (e != hipSuccess) return hip_fail(e, "bsm_bench_stream");
    return BSM_OK;
}
