// bsm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the block-sparse
// mat-vec engine.  One kernel family streams STRIP-packed pieces (bsm_layout.h):
//
//   forward    u[i]  = sum_w B[i,w] * x[col(w)]     lane owns a row, 16-byte lane loads,
//                                                    x slice staged through LDS per wave
//   transposed v[w]  = sum_i B[i,w] * x[row(i)]     same bytes, halving butterfly across the
//                                                    P lanes of a strip, then y[col(w)] += v
//
// Both directions are taken from ONE read of the piece (SymmetricBlockMatrix off-diagonal
// blocks: reference src/symmetricblockmatrix.jl:394-418 reads them twice).
// The kernels are HBM-bound (0.25-0.5 FLOP/B): no MFMA, everything is about keeping
// >= 8 KB per wave in flight with perfectly coalesced 16-byte loads.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "bsm_analysis.h"
#include "bsm_kernels.h"
#include "bsm_layout.h"

namespace bsm {

// ----------------------------------------------------------------------------------------
// element types
// ----------------------------------------------------------------------------------------
struct c64 {
    float re, im;
};
struct c128 {
    double re, im;
};

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int E = 4;
};
template <> struct TT<double> {
    static constexpr int E = 2;
};
template <> struct TT<c64> {
    static constexpr int E = 2;
};
template <> struct TT<c128> {
    static constexpr int E = 1;
};

template <typename T> struct alignas(16) Vec16 {
    T v[TT<T>::E];
};

// 16-byte matrix load with the non-temporal hint (global_load_dwordx4 ... nt): every matrix byte
// is used exactly once per launch.  Measured on a bare streaming read of a C2-sized operator out of
// the Infinity Cache (tools/stream_floor.hip, mode 4): 6.65 us with the hint, 8.45 us without.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ Vec16<T> load_stream16(const Vec16<T> *p) {
    const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    Vec16<T> out;
    __builtin_memcpy(&out, &raw, 16);
    return out;
}

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ c64 zero_of(c64) { return c64{0.f, 0.f}; }
__device__ __forceinline__ c128 zero_of(c128) { return c128{0.0, 0.0}; }

__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ double add(double a, double b) { return a + b; }
__device__ __forceinline__ c64 add(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c128 add(c128 a, c128 b) { return c128{a.re + b.re, a.im + b.im}; }

__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ double mul(double a, double b) { return a * b; }
__device__ __forceinline__ c64 mul(c64 a, c64 b) {
    return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ c128 mul(c128 a, c128 b) {
    return c128{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// acc + a*b
__device__ __forceinline__ float madd(float acc, float a, float b) { return fmaf(a, b, acc); }
__device__ __forceinline__ double madd(double acc, double a, double b) { return fma(a, b, acc); }
__device__ __forceinline__ c64 madd(c64 acc, c64 a, c64 b) {
    acc.re = fmaf(a.re, b.re, acc.re);
    acc.re = fmaf(-a.im, b.im, acc.re);
    acc.im = fmaf(a.re, b.im, acc.im);
    acc.im = fmaf(a.im, b.re, acc.im);
    return acc;
}
__device__ __forceinline__ c128 madd(c128 acc, c128 a, c128 b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
    return acc;
}

__device__ __forceinline__ float cj(float a, bool) { return a; }
__device__ __forceinline__ double cj(double a, bool) { return a; }
__device__ __forceinline__ c64 cj(c64 a, bool c) { return c64{a.re, c ? -a.im : a.im}; }
__device__ __forceinline__ c128 cj(c128 a, bool c) { return c128{a.re, c ? -a.im : a.im}; }

__device__ __forceinline__ float shx(float a, int d) { return __shfl_xor(a, d, 64); }
__device__ __forceinline__ double shx(double a, int d) { return __shfl_xor(a, d, 64); }
__device__ __forceinline__ c64 shx(c64 a, int d) {
    return c64{__shfl_xor(a.re, d, 64), __shfl_xor(a.im, d, 64)};
}
__device__ __forceinline__ c128 shx(c128 a, int d) {
    return c128{__shfl_xor(a.re, d, 64), __shfl_xor(a.im, d, 64)};
}

// hardware floating-point atomics (global_atomic_add_f32 / _f64; built with
// -munsafe-fp-atomics so no compare-and-swap loop is emitted)
__device__ __forceinline__ void atomic_acc(float *p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_acc(double *p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_acc(c64 *p, c64 v) {
    atomicAdd(&p->re, v.re);
    atomicAdd(&p->im, v.im);
}
__device__ __forceinline__ void atomic_acc(c128 *p, c128 v) {
    atomicAdd(&p->re, v.re);
    atomicAdd(&p->im, v.im);
}

// LDS accumulation (ds_add_f32 / ds_add_f64): the workgroup's y window
__device__ __forceinline__ void lds_acc(float *p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_acc(double *p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_acc(c64 *p, c64 v) {
    atomicAdd(&p->re, v.re);
    atomicAdd(&p->im, v.im);
}
__device__ __forceinline__ void lds_acc(c128 *p, c128 v) {
    atomicAdd(&p->re, v.re);
    atomicAdd(&p->im, v.im);
}
__device__ __forceinline__ bool is_zero(float a) { return a == 0.f; }
__device__ __forceinline__ bool is_zero(double a) { return a == 0.0; }
__device__ __forceinline__ bool is_zero(c64 a) { return a.re == 0.f && a.im == 0.f; }
__device__ __forceinline__ bool is_zero(c128 a) { return a.re == 0.0 && a.im == 0.0; }

// ----------------------------------------------------------------------------------------
// halving butterfly: every lane of a P-lane group holds V partial values; afterwards the group's
// sums are spread over its lanes: lane i keeps max(1, V/P) of them, starting at value index `pos`;
// lanes with (i & dup) != 0 hold duplicates and must not emit.
// The four exchanges inside a 16-lane row are DPP moves (row_mirror = lane^15, row_half_mirror =
// lane^7, quad_perm = lane^3, lane^1: plain VALU, no LDS traffic); only the 16- and 32-lane
// exchanges, which carry the fewest values, go through ds_bpermute.  Each exchange pairs lanes
// that agree on every bit decided so far, so both hold the same value subset.
// ----------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ int dpp32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float dppx(float a) {
    return __int_as_float(dpp32<CTRL>(__float_as_int(a)));
}
template <int CTRL> __device__ __forceinline__ double dppx(double a) {
    const long long v = __double_as_longlong(a);
    const int lo = dpp32<CTRL>((int)(v & 0xffffffffll));
    const int hi = dpp32<CTRL>((int)(v >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL> __device__ __forceinline__ c64 dppx(c64 a) { return c64{dppx<CTRL>(a.re), dppx<CTRL>(a.im)}; }
template <int CTRL> __device__ __forceinline__ c128 dppx(c128 a) { return c128{dppx<CTRL>(a.re), dppx<CTRL>(a.im)}; }

constexpr int DPP_ROW_MIRROR = 0x140;       // lane ^ 15
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane ^ 7
constexpr int DPP_QUAD_XOR3 = 0x1B;         // quad_perm [3,2,1,0]
constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm [1,0,3,2]

// one exchange: BIT decides who keeps which half; XCH(v) returns the partner's value
template <typename T, int CUR, int BIT, typename XCH>
__device__ __forceinline__ void bfly_step(T *v, int i, int &pos, int &dup, XCH xch) {
    const bool hi = (i & BIT) != 0;
    if constexpr (CUR >= 2) {
        constexpr int H = CUR / 2;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const T keep = hi ? v[H + j] : v[j];
            const T send = hi ? v[j] : v[H + j];
            v[j] = add(keep, xch(send));
        }
        if (hi) pos += H;
    } else {
        v[0] = add(v[0], xch(v[0]));
        dup |= BIT;
    }
}

template <typename T, int V, int P> struct Butterfly {
    static constexpr int half(int cur) { return cur >= 2 ? cur / 2 : 1; }
    static __device__ __forceinline__ void run(T *v, int i, int &pos, int &dup) {
        constexpr int C0 = V;
        constexpr int C1 = (P >= 16) ? half(C0) : C0;  // after lane^15 (bit 3)
        constexpr int C2 = half(C1);                   // after lane^7  (bit 2)   (P >= 8 always)
        constexpr int C3 = half(C2);                   // after lane^3  (bit 1)
        constexpr int C4 = half(C3);                   // after lane^1  (bit 0)
        constexpr int C5 = (P >= 32) ? half(C4) : C4;  // after lane^16 (bit 4)
        if constexpr (P >= 16) bfly_step<T, C0, 8>(v, i, pos, dup, [](T a) { return dppx<DPP_ROW_MIRROR>(a); });
        bfly_step<T, C1, 4>(v, i, pos, dup, [](T a) { return dppx<DPP_ROW_HALF_MIRROR>(a); });
        bfly_step<T, C2, 2>(v, i, pos, dup, [](T a) { return dppx<DPP_QUAD_XOR3>(a); });
        bfly_step<T, C3, 1>(v, i, pos, dup, [](T a) { return dppx<DPP_QUAD_XOR1>(a); });
        if constexpr (P >= 32) bfly_step<T, C4, 16>(v, i, pos, dup, [](T a) { return shx(a, 16); });
        if constexpr (P >= 64) bfly_step<T, C5, 32>(v, i, pos, dup, [](T a) { return shx(a, 32); });
    }
};

// halving reduction over the lane bits D, 2D, ... 32 (the lanes that differ only in those bits hold partial
// sums of the same CUR values); afterwards as for Butterfly: lane keeps max(1, CUR * D / 64) values from `pos`
template <typename T, int CUR, int D> struct ReduceAbove {
    static __device__ __forceinline__ void run(T *v, int lane, int &pos, int &dup) {
        if constexpr (D < 64) {
            bfly_step<T, CUR, D>(v, lane, pos, dup, [](T a) { return shx(a, D); });
            ReduceAbove<T, (CUR >= 2 ? CUR / 2 : 1), D * 2>::run(v, lane, pos, dup);
        }
    }
};

#ifdef BSM_TRACE
// developer build (make trace): per-wave phase timestamps for tools/wavetrace.py.  The stamps
// (s_memtime) are parked in LDS and leave the wave once, at its end: a global store per stamp would
// sit in the in-order vmcnt queue in front of the matrix loads and triple the kernel time.
__device__ unsigned long long *g_trace = nullptr;
__shared__ unsigned long long t_trace[kWavesPerWg][16];
#define BSM_TSTAMP(slot)                                                   \
    do {                                                                   \
        if (lane == 0) t_trace[threadIdx.x >> 6][(slot)] = clock64();      \
    } while (0)
#else
#define BSM_TSTAMP(slot) \
    do {                 \
    } while (0)
#endif

#ifndef BSM_C64_FUSED_WAVES
#define BSM_C64_FUSED_WAVES 5
#endif
// multi-RHS register path: y indices of a chunk kept in LDS (0: read from the column list in every iteration)
#ifndef BSM_MULTI_IX
#define BSM_MULTI_IX 1
#endif
#ifndef BSM_C128_FUSED_WAVES
#define BSM_C128_FUSED_WAVES 6
#endif
#ifndef BSM_C64_L
#define BSM_C64_L 4
#endif
// (developer builds: loads per lane of the fused kernels of the other element types)
#ifndef BSM_F32_L
#define BSM_F32_L 4
#endif
#ifndef BSM_F64_L
#define BSM_F64_L 8
#endif
#ifndef BSM_C128_L
#define BSM_C128_L 8
#endif
constexpr int FLAG_STRONG_ZERO = 1;
constexpr int FLAG_DIRECT = 2;
constexpr int FLAG_CONJ = 4;
constexpr int FLAG_OPT = 8;
constexpr int FLAG_RMW = 16;     // coloured launch: conflict-free by construction, plain read-modify-write
constexpr int FLAG_GATHER = 32;  // contributions are stored in the workspace, gather_kernel sums them
// multi-RHS kernels: bits 8-11 = number of ACTIVE right-hand sides of a padded batch (0: all K).  Columns past it
// read the last active column of X (valid memory, arithmetic wasted) and are never written.
constexpr int FLAG_KACT_SHIFT = 8;
#ifdef BSM_EXPERIMENT
// developer build (make exp): timing-only ablations of the fused kernel, selected by BSM_DEBUG_FLAGS
// (results are WRONG with any bit set; tools/ablate.py)
constexpr int DBG_NO_GLOBAL_ATOMICS = 1 << 16;  // transposed emission: sums outside the window are dropped
constexpr int DBG_NO_WINDOW_ADD = 1 << 17;      // ... sums inside the window are dropped
constexpr int DBG_NO_BUTTERFLY = 1 << 18;       // lane-local values are parked instead of the group sums
constexpr int DBG_NO_EMISSION = 1 << 19;        // the emission loop is skipped altogether
constexpr int DBG_NO_XGATHER = 1 << 20;         // the x slice is a constant (no column-list / x loads)
constexpr int DBG_NO_FWD_OUT = 1 << 21;         // forward sums are not written
constexpr int DBG_NO_MATRIX = 1 << 22;          // multi-RHS tile pipeline: the matrix loads are not issued
constexpr int DBG_NO_FWD_HALF = 1 << 23;        // ... the forward half of an iteration is skipped
constexpr int DBG_NO_TRN_HALF = 1 << 24;        // ... the transposed half of an iteration is skipped
#define BSM_DBG(bit) ((flags & (bit)) != 0)
#else
#define BSM_DBG(bit) false
#endif

// ----------------------------------------------------------------------------------------
// descriptors: fetched as whole 16-byte words through a wave-uniform address (scalar loads),
// so a wave reaches its matrix bytes after ONE dependent memory round trip.
// ----------------------------------------------------------------------------------------
struct PieceD {
    uint32_t val_lo, val_hi;
    int xbase, col_off, nstrips, ncols, kind, seg2_x;
};
struct WaveD {
    int npieces, row_off, rbase, m, work, grp, lead, wg_sync, seg1_w, seg1_x, seg2_w, win_base, win_n;
    PieceD first;
};

__device__ __forceinline__ PieceD decode_piece(const uint4 a, const uint4 b) {
    PieceD p;
    p.val_lo = a.x;
    p.val_hi = a.y;
    p.xbase = (int)a.z;
    p.col_off = (int)a.w;
    p.nstrips = (int)b.x;
    p.ncols = (int)b.y;
    p.kind = (int)b.z;
    p.seg2_x = (int)b.w;
    return p;
}

__device__ __forceinline__ WaveD load_wave(const WaveWork *__restrict__ wp) {
    const uint4 *__restrict__ q = reinterpret_cast<const uint4 *>(wp);
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    WaveD w;
    w.seg1_w = (int)q0.x;
    w.win_base = (int)q0.y;
    w.row_off = (int)q0.z;
    w.rbase = (int)q0.w;
    w.m = (int)(q1.x & 0xffffu);
    w.work = (int)((q1.x >> 16) & 0xffu);
    w.grp = (int)(q1.x >> 24);
    w.lead = (int)(q1.y & 0xffu);
    w.wg_sync = (int)((q1.y >> 8) & 0xffu);
    w.npieces = (int)((q1.y >> 16) & 0xffu);
    w.win_n = (int)(q1.y >> 24) * 8;
    w.seg1_x = (int)q1.z;
    w.seg2_w = (int)q1.w;
    w.first = decode_piece(q2, q3);
    return w;
}

// ----------------------------------------------------------------------------------------
// one wave streams its pieces; returns the forward partial sum of row (lane % P)
// ----------------------------------------------------------------------------------------
// x slice staged per wave in LDS.  Forward-only kernels: 2 KB (512 fp32 / 256 fp64, complex64 / 128
// complex128 columns): the batched gather keeps one x entry per 64 columns in registers.
// Fused kernels: 2 KB too (512 fp32 ... 128 complex128 columns) -- with the y window and the emission
// staging their occupancy is bounded by LDS and by the registers of the gather.
template <typename T, bool TRN = false> constexpr int x_chunk_cols() {
    return TRN ? 2048 / (int)sizeof(T) : (sizeof(T) >= 16 ? 128 : (sizeof(T) == 8 ? 256 : 512));
}

template <typename T, int L, int P, bool FWD, bool TRN, bool NT>
__device__ __forceinline__ T run_panel(const WaveD &wd, const uint4 *__restrict__ values,
                                       const int *__restrict__ rows, const int *__restrict__ cols,
                                       const T *__restrict__ x, T *__restrict__ y, T alpha,
                                       int flags, int lane, T *xs, T *vs, T *win, int win_n,
                                       T *__restrict__ ws) {
    constexpr int E = TT<T>::E;
    constexpr int G = 64 / P;
    constexpr int V = L * E;
    constexpr int NC = G * L * E;                        // columns covered per iteration
    constexpr int XCH = x_chunk_cols<T, TRN>();            // columns staged per x chunk
    // iterations per transposed emission: the column sums of a whole staged chunk leave the wave
    // together.  Atomics (and the plain stores of the gather mode) sit in the same in-order vmcnt
    // queue as the loads and take 2-3x as long under load (MI355X_MICROARCH.md: ~3000 cycles with
    // every CU issuing): emitted every iteration, each one is waited for by the NEXT iteration's
    // matrix loads; emitted at the chunk end of a small panel, nothing ever waits for them.
    constexpr int BF = XCH / NC;
    static_assert(XCH % NC == 0, "x chunk must hold whole iterations");
    constexpr bool INPLACE = FWD && TRN;  // column sums parked in the x slice, y indices of the chunk kept in `vs`
    int *ix = reinterpret_cast<int *>(vs);
    const bool opT = (flags & FLAG_OPT) != 0;
    const bool cjf = (flags & FLAG_CONJ) != 0;
    const int m = wd.m;
    const int i = lane & (P - 1);
    const int g = lane / P;
    const bool row_ok = i < m;

    T acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = zero_of(T{});
    T xr = zero_of(T{});
    if (TRN && row_ok) {
        // (issuing this second round trip together with the x gather of the first chunk's columns, so that
        // the rows -> x and columns -> x chains overlap, changed nothing on the tiled BEM fixture and costs
        // the fp64 / fp32 fused instances a register they do not have)
        const int ri = (wd.rbase >= 0) ? wd.rbase + i : rows[wd.row_off + i];
        xr = x[ri];
    }

    const PieceD pc = wd.first;
    if (wd.npieces > 0) {
        const int xbase = pc.xbase;
        const int col_off = pc.col_off;
        const int nstrips = pc.nstrips;
        const int ncols = pc.ncols;
        // per-column kinds (a symmetric row group holds its diagonal block and its off-diagonal
        // blocks in one panel): forward uses a column unless (op T/C and it is not KIND_OFF),
        // transposed uses it iff (op T/C or KIND_OFF)
        const int kinds = pc.kind;
        const bool has_off = (kinds & kKindHasOff) != 0;
        const bool fwd_en = FWD && (!opT || has_off);
        const bool trn_en = TRN && (opT || has_off);
        const Vec16<T> *__restrict__ vb = reinterpret_cast<const Vec16<T> *>(
            values + (((uint64_t)pc.val_hi << 32) | pc.val_lo));
        // piece column -> x / y index: up to three inline contiguous runs, else the cols pool
        const int s1w = wd.seg1_w, s1x = wd.seg1_x - wd.seg1_w;
        const int s2w = wd.seg2_w, s2x = pc.seg2_x - wd.seg2_w;
        // -> x / y index of piece column w; `off` tells whether the column is KIND_OFF
        auto col_lookup = [&](int w, bool &off) -> int {
            if (xbase < 0) {
                const int raw = cols[col_off + w];
                off = raw >= 0 && (kinds & 3) == KIND_OFF;
                return raw & 0x7fffffff;
            }
            const int sh = w < s1w ? 0 : (w < s2w ? 2 : 4);
            off = ((kinds >> sh) & 3) == KIND_OFF;
            return w + (w < s1w ? xbase : (w < s2w ? s1x : s2x));
        };

        // L independent 16-byte loads per lane: 8 KB of the matrix per wave in flight
        auto load_b = [&](Vec16<T>(&b)[L], int s0) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int s = s0 + l * G + g;
                if (row_ok && s < nstrips) {
                    b[l] = NT ? load_stream16(&vb[(uint32_t)(s * m + i)]) : vb[(uint32_t)(s * m + i)];
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) b[l].v[e] = zero_of(T{});
                }
            }
        };
        // x slice of a chunk: gathered ONCE per wave into LDS (contiguous runs or through the merged
        // column list), then read back as 16-byte broadcasts by every iteration of the chunk.  The
        // gather is branch-free and batched -- all column-list loads, then all x loads, then the LDS
        // stores: ONE memory round trip (two through the list) instead of a dependent load / wait /
        // store per 64 columns, which tools/wavetrace.py showed as 0.9 us (median) to 3.4 us (p90) of
        // a 9.7 us C2 launch.  A lane past the last column loads the last column's entry and stores
        // zero (the tail one iteration past the end must read as zero).
        auto stage_x = [&](int c0, auto kx_tag) {
            constexpr int KX = decltype(kx_tag)::value;
            const bool pool = xbase < 0;  // wave-uniform
            int raw[KX];                  // column-list entries (pool) -- the only state kept per column
            T xv[KX];
            if (pool) {
#pragma unroll
                for (int k = 0; k < KX; ++k) raw[k] = cols[col_off + min(c0 + k * 64 + lane, ncols - 1)];
#pragma unroll
                for (int k = 0; k < KX; ++k) xv[k] = x[raw[k] & 0x7fffffff];
            } else {
#pragma unroll
                for (int k = 0; k < KX; ++k) {
                    bool off;
                    raw[k] = 0;
                    xv[k] = x[col_lookup(min(c0 + k * 64 + lane, ncols - 1), off)];
                }
            }
#pragma unroll
            for (int k = 0; k < KX; ++k) {
                const int w = c0 + k * 64 + lane;
                bool off;
                int yi = raw[k] & 0x7fffffff;
                if (pool)
                    off = raw[k] >= 0 && (kinds & 3) == KIND_OFF;
                else
                    yi = col_lookup(min(w, ncols - 1), off);
                xs[k * 64 + lane] = (w < ncols && (!opT || off)) ? xv[k] : zero_of(T{});
                // fused kernels: the transposed emission of this chunk finds its y index here (sign bit: the column
                // takes no part in it) instead of reading the column list a second time, a dependent round trip per
                // 64 columns in front of the atomics
                if (INPLACE) ix[k * 64 + lane] = (w < ncols && (opT || off)) ? yi : -1;
            }
        };

        // the x slice of the chunk that starts at column c0
        auto stage_chunk = [&](int c0) {
            // (a fused wave whose piece has no forward half in this op still needs the chunk's y indices)
            if (!fwd_en && !(INPLACE && trn_en)) return;
            if (BSM_DBG(DBG_NO_XGATHER)) {
#pragma unroll
                for (int k = 0; k < XCH / 64; ++k) {
                    xs[k * 64 + lane] = alpha;
                    if (INPLACE) {
                        const int w = c0 + k * 64 + lane;
                        bool off = false;
                        const int yi = col_lookup(min(w, ncols - 1), off);
                        ix[k * 64 + lane] = (w < ncols && (opT || off)) ? yi : -1;
                    }
                }
                return;
            }
            constexpr int KXM = XCH / 64;
            const int need = min(ncols - c0, XCH) + NC;  // columns the chunk's iterations read
            if (KXM >= 4 && need <= (KXM / 4) * 64)
                stage_x(c0, std::integral_constant<int, (KXM >= 4 ? KXM / 4 : 1)>{});
            else if (KXM >= 2 && need <= (KXM / 2) * 64)
                stage_x(c0, std::integral_constant<int, (KXM >= 2 ? KXM / 2 : 1)>{});
            else
                stage_x(c0, std::integral_constant<int, KXM>{});
        };
        // one iteration on the L loaded strips-per-group starting at strip s0 of the chunk [c0, c0 + XCH)
        auto iteration = [&](Vec16<T>(&b)[L], int s0, int c0, int s_end) {
            if (fwd_en) {
                const int cb = (s0 - c0 / E) * E;  // first column of this iteration inside the chunk
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const Vec16<T> xv = *reinterpret_cast<const Vec16<T> *>(&xs[cb + (l * G + g) * E]);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = madd(acc[e], cj(b[l].v[e], cjf), xv.v[e]);
                }
            }
#ifdef BSM_TRACE
            if (s0 == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BSM_TSTAMP(3);  // first iteration's matrix bytes have arrived
            }
#endif
            if (trn_en) {
                T vals[V];
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int e = 0; e < E; ++e) vals[l * E + e] = mul(cj(b[l].v[e], cjf), xr);
                int pos = 0, dup = 0;
                if (!BSM_DBG(DBG_NO_BUTTERFLY)) Butterfly<T, V, P>::run(vals, i, pos, dup);
                constexpr int CF = (V / P) > 1 ? (V / P) : 1;
                // the column sums of the chunk's iterations are parked in LDS and leave the wave
                // together (64 busy lanes per atomic wave-instruction instead of NC)
                const int slot = (s0 - c0 / E) / (G * L);
                // fused kernels park IN PLACE: slot c of the x slice holds x of chunk column c until the forward half of
                // its iteration has read it (above: the same wave, LDS in program order), then the column's sum
                T *park = INPLACE ? xs : vs;
                if ((i & dup) == 0) {
#pragma unroll
                    for (int j = 0; j < CF; ++j) {
                        const int q = pos + j;  // original value index l*E + e
                        const int l = q / E, e = q % E;
                        park[slot * NC + (l * G + g) * E + e] = vals[j];
                    }
                }
                const bool last_it = (s0 + G * L >= s_end);
                if ((slot == BF - 1 || last_it) && !BSM_DBG(DBG_NO_EMISSION)) {
                    const int sb = s0 - slot * (G * L);  // first strip of the batch
                    const int nbatch = min((slot + 1) * NC, ncols - sb * E);  // columns of the batch
#pragma unroll 1
                    for (int k = 0; k * 64 < nbatch; ++k) {
                        const int c = k * 64 + lane;
                        const int w = sb * E + c;
                        if (c < nbatch) {
                            int yi;
                            if (INPLACE) {
                                yi = ix[c];
                                if (yi < 0) continue;  // a diagonal column in op N: forward only
                            } else {
                                bool off;
                                yi = col_lookup(w, off);
                                if (!(opT || off)) continue;
                            }
                            if (flags & FLAG_GATHER) {  // one plain, coalesced store per column sum
                                ws[col_off + w] = park[c];
                                continue;
                            }
                            const T val = mul(alpha, park[c]);
                            const unsigned wi = (unsigned)(yi - wd.win_base);
                            if (wi < (unsigned)win_n) {
                                if (!BSM_DBG(DBG_NO_WINDOW_ADD)) lds_acc(&win[wi], val);  // leaves the CU once, with the window
                            } else if (flags & FLAG_RMW) {
                                y[yi] = add(y[yi], val);
                            } else if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS)) {
                                atomic_acc(&y[yi], val);
                            }
                        }
                    }
                }
            }
        };

        for (int c0 = 0; c0 < ncols; c0 += XCH) {
            stage_chunk(c0);
#ifdef BSM_TRACE
            if (c0 == 0) BSM_TSTAMP(2);  // x slice staged (loads issued and stored to LDS)
#endif
            const int s_end = min(nstrips, (c0 + XCH) / E);
            for (int s0 = c0 / E; s0 < s_end; s0 += G * L) {
                // (Software-pipelined variants -- two register buffers of L / 2 loads with the next
                // iteration's loads issued before the butterfly / emission of the current one and the
                // first ones before the x gather; every next iteration's loads already in flight; only a
                // chunk's first loads issued inside the x gather -- were measured on forward AND fused
                // kernels and lost every time (tiled BEM fixture, fused: ComplexF64 153 -> 160 us, fp64
                // 82 -> 117 us with 36 B of scratch): the memory system serves requests first come, first
                // served, occupancy provides the parallelism of a long launch, and the extra iterations
                // cost issue slots.)  Round 5, once more for the forward-only kernels with BRANCH-FREE loads (strip and
                // row clamped, so that hipcc counts the loads in flight exactly -- docs/experiments_r05.md): 108 VGPRs = 4
                // waves instead of 6; C2 9.5 -> 12.1 us, C4 slice 301 -> 358, 1 GB VBCRS 166-175 -> 173-191: a wave of one
                // iteration issues a second, redundant batch, and the waves lost cost more than the overlap gains.
                Vec16<T> b[L];
                // a wave about to request matrix bytes is served before its SIMD's other waves (which are in
                // their butterfly / FMA phases): +0.3-2 % on every operator, nothing it costs
                __builtin_amdgcn_s_setprio(3);
                load_b(b, s0);
                __builtin_amdgcn_s_setprio(0);
                iteration(b, s0, c0, s_end);
            }
        }
    }
    T a = acc[0];
    if (FWD) {
#pragma unroll
        for (int e = 1; e < E; ++e) a = add(a, acc[e]);
#pragma unroll
        for (int d = P; d < 64; d <<= 1) a = add(a, shx(a, d));
    }
    return a;
}

// Occupancy is what the small-panel (BEM-shaped) products live on: a small panel is a chain of
// dependent memory round trips, hidden only by other resident waves.
//   fp64 forward-only: capped at 80 VGPRs (>= 6 waves per SIMD = 1536 resident workgroups: every
//     workgroup of a C2-sized launch is resident at once); compiles to 78 (a cap of 72 = 7 waves
//     compiles to 70 without scratch and measures the same: C2, 1 GB VBCRS, BEM forward).
//   fp64 fused: capped at 64 VGPRs = 8 waves per SIMD, no scratch; with 20 KB of LDS per workgroup
//     exactly 8 workgroups fit a CU (+11-13 % on 3-28-row fp64 panels over 6 waves).
//   complex128: capped at 80 (the fused instance compiles to 71: 7 waves).
//   fp32 / complex64: capped at 96 = 5 waves (fp32 fused compiles to 80: 6), no scratch anywhere.
template <typename T, int L, bool FWD, bool TRN, bool NT>
__global__ void __launch_bounds__(64 * kWavesPerWg) __attribute__((amdgpu_waves_per_eu(
    (FWD && TRN && (std::is_same<T, double>::value || std::is_same<T, float>::value)) ? 8 :
    (FWD && TRN && std::is_same<T, c128>::value) ? (L == 4 ? 8 : BSM_C128_FUSED_WAVES) :
    (FWD && TRN && std::is_same<T, c64>::value) ? (L == 4 ? 8 : BSM_C64_FUSED_WAVES) :
    (((!TRN && std::is_same<T, double>::value) || std::is_same<T, c128>::value) ? 6 : 5))))
    // <= 96 SGPRs: a CU admits 7 workgroups of 256 threads (the ComplexF64 fused instance compiled to 106 =
    // 6 workgroups; tiled BEM fixture 147.7 -> 143.9 us with the cap, nothing else changes)
    __attribute__((amdgpu_num_sgpr(96)))
    panel_kernel(const WaveWork *__restrict__ waves, const uint4 *__restrict__ values, const int *__restrict__ rows,
                 const int *__restrict__ cols, const T *__restrict__ x, T *__restrict__ y, T alpha,
                 T beta, int flags, unsigned wg_base, T *__restrict__ ws, long long ws_fbase) {
    constexpr int XS = x_chunk_cols<T, TRN>();         // staged x slice per wave
    constexpr int VS = XS;                             // transposed column sums of one staged chunk
    __shared__ __attribute__((aligned(16))) T xs[kWavesPerWg][FWD ? XS : 1];
    __shared__ __attribute__((aligned(16))) T vs[kWavesPerWg][TRN ? VS : 1];
    // (the cross-wave combine slab of split groups aliases xs: a wave's x slice is dead by then)
    // y window of workgroups that pack neighbouring small row groups of a symmetric operator
    constexpr bool WIN = FWD && TRN;
    __shared__ T win[WIN ? window_entries((int)sizeof(T)) : 1];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    BSM_TSTAMP(0);  // wave started
    const WaveD wd = load_wave(waves + ((size_t)(blockIdx.x + wg_base) * kWavesPerWg + wave));
    const int work = wd.work;
    const int m = wd.m;
#ifdef BSM_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BSM_TSTAMP(1);  // descriptor arrived
    if (lane == 0) {
        t_trace[threadIdx.x >> 6][6] =
            (unsigned long long)(wd.work == WORK_PANEL && wd.npieces ? (long long)wd.first.nstrips * wd.m * 16 : 0);
        t_trace[threadIdx.x >> 6][7] = wall_clock64();
    }
#endif
    // workgroup-uniform (all 4 descriptors carry the same window; coloured launches keep plain RMW)
    const int win_n = (WIN && !(flags & FLAG_RMW)) ? wd.win_n : 0;
    if (WIN && win_n > 0) {
        for (int e = threadIdx.x; e < win_n; e += 64 * kWavesPerWg) win[e] = zero_of(T{});
        __syncthreads();
    }

    T u = zero_of(T{});
    if (work == WORK_PANEL) {
        if (m <= 8)
            u = run_panel<T, L, 8, FWD, TRN, NT>(wd, values, rows, cols, x, y, alpha, flags, lane, xs[wave], vs[wave], win, win_n, ws);
        else if (m <= 16)
            u = run_panel<T, L, 16, FWD, TRN, NT>(wd, values, rows, cols, x, y, alpha, flags, lane, xs[wave], vs[wave], win, win_n, ws);
        else if (m <= 32)
            u = run_panel<T, L, 32, FWD, TRN, NT>(wd, values, rows, cols, x, y, alpha, flags, lane, xs[wave], vs[wave], win, win_n, ws);
        else
            u = run_panel<T, L, 64, FWD, TRN, NT>(wd, values, rows, cols, x, y, alpha, flags, lane, xs[wave], vs[wave], win, win_n, ws);
    }
    BSM_TSTAMP(4);  // the wave's piece is streamed
    const bool direct = (flags & FLAG_DIRECT) != 0;
    const bool sz = (flags & FLAG_STRONG_ZERO) != 0;
    if (FWD) {
        if (wd.wg_sync) {  // workgroup-uniform: only groups split over several waves meet in LDS
            xs[wave][lane] = u;
            __syncthreads();
        }
        if (work == WORK_PANEL && wd.lead && !BSM_DBG(DBG_NO_FWD_OUT)) {
            for (int k = 1; k < wd.grp; ++k) u = add(u, xs[wave + k][lane]);
            if (flags & FLAG_GATHER) {
                // forward partial sums of this workgroup item: slots ws_fbase + win_base + row
                const bool fwd_on = !(flags & FLAG_OPT) || (wd.first.kind & kKindGroupHasOff);
                if (lane < m && fwd_on) ws[ws_fbase + wd.win_base + lane] = u;
            } else if (lane < m) {
                const int yi = (wd.rbase >= 0) ? wd.rbase + lane : rows[wd.row_off + lane];
                const T val = mul(alpha, u);
                const unsigned wi = (unsigned)(yi - wd.win_base);
                if (direct) {
                    y[yi] = sz ? val : madd(val, beta, y[yi]);
                } else if (wi < (unsigned)win_n) {
                    lds_acc(&win[wi], val);
                } else if (flags & FLAG_RMW) {
                    y[yi] = add(y[yi], val);
                } else {
                    atomic_acc(&y[yi], val);
                }
            }
        }
    }
    if (WIN && win_n > 0) {
        // every wave has parked its sums: the window leaves the CU once, 64 contiguous entries
        // per atomic wave-instruction
        __syncthreads();
        for (int e = threadIdx.x; e < win_n; e += 64 * kWavesPerWg) {
            const T v = win[e];
            if (!is_zero(v)) atomic_acc(&y[wd.win_base + e], v);
        }
    }
    if (work == WORK_SCALE && direct) {
        const int cnt = wd.first.ncols;
        for (int r = lane; r < cnt; r += 64)
            y[wd.rbase + r] = sz ? zero_of(T{}) : mul(beta, y[wd.rbase + r]);
    }
#ifdef BSM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSM_TSTAMP(5);  // everything stored
    if (lane == 0) t_trace[threadIdx.x >> 6][8] = wall_clock64();
    if (g_trace && lane < 16)
        g_trace[((size_t)blockIdx.x * kWavesPerWg + (threadIdx.x >> 6)) * 16 + lane] = t_trace[threadIdx.x >> 6][lane];
#endif
}

#ifdef BSM_TRACE
extern "C" int bsm_debug_set_trace(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf));
}
#endif

// ========================================================================================
// multi right-hand-side variant: Y = alpha*op(A)*X + beta*Y for K columns per pass.  A is
// streamed ONCE for the K columns (LinearMaps loops the columns through _unsafe_mul!, i.e. K
// full sweeps of A).  Same work distribution and layout as panel_kernel; every lane keeps K
// accumulators, the staged x slice is [column][k] in LDS.
// ========================================================================================
template <typename T, int K> constexpr int x_chunk_cols_multi_vec() {
    return (x_chunk_cols<T>() / K) > 64 * TT<T>::E ? (x_chunk_cols<T>() / K) : 64 * TT<T>::E;
}
// the tile-pipelined kernels (below) stage shorter slices: their LDS goes to the two matrix tiles.  A chunk
// must hold whole iterations of every strip height (8 * L strips of E columns) and the 64 * K combine slab.
template <typename T> constexpr bool kRealType = false;
template <> constexpr bool kRealType<float> = true;
template <> constexpr bool kRealType<double> = true;
// which multi-RHS kernels run the tile pipeline: the transposed / fused 8- and 4-column ones in real arithmetic
// with 4 loads per lane (the complex 8-column ones are at the register limit as they are: c64 fused 242 -> 256
// VGPRs + scratch with it; the ComplexF64 4-column one gains nothing over its register path: 487 vs 490 us)
template <typename T, int L, bool TRN, int K> constexpr bool kTilePipe = TRN && L == 4 && kRealType<T> && K >= 4 && K <= 8;
template <typename T, int L> constexpr int x_chunk_cols_pipe() {
    return 8 * L * TT<T>::E > 64 ? 8 * L * TT<T>::E : 64;
}

// XOR swizzle of the LDS matrix tile (16-byte units; strip sI of an iteration, row r of the strip -> unit
// sI * P + (r ^ tile_swz(sI))): a global -> LDS load writes 64 consecutive units per wave-instruction, so the
// image cannot be padded; instead every lane FETCHES row (lane ^ swz) of its strip.  The masks make both ways
// the tile is read -- by row (lane = row, one strip per load) and by column (lane = strip, L rows per lane) --
// free of bank conflicts under ds_read_b128's four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (found by exhaustive search over the linear maps strip bits -> row bits; value b = mask of strip bit b).
template <int P, int L> struct TileSwz;
template <> struct TileSwz<8, 4> { static constexpr int col[6] = {0, 1, 0, 2, 4, 0}; };
template <> struct TileSwz<16, 4> { static constexpr int col[6] = {1, 2, 0, 8, 0, 0}; };
template <> struct TileSwz<32, 4> { static constexpr int col[6] = {1, 2, 0, 0, 0, 0}; };
template <> struct TileSwz<64, 4> { static constexpr int col[6] = {1, 2, 0, 0, 0, 0}; };
template <int P, int L> __device__ __forceinline__ constexpr int tile_swz(int s) {
    int h = 0;
    for (int b = 0; b < 6; ++b)
        if ((s >> b) & 1) h ^= TileSwz<P, L>::col[b];
    return h;
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4 ... nt): lane l's bytes land
// at lds_dst + 16 * l, lds_dst wave-uniform.  No VGPR destination and hipcc does not count it: the caller waits
// with vm_wait(n) (loads, atomics and these complete in issue order).
__device__ __forceinline__ void glds16_nt(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their maxima), as the builtin: hipcc's own wait insertion
// sees it, so loads it still tracks as pending are retired in its model too and it does not add a vmcnt(0) of its
// own further down (which would drain the prefetched tile)
__device__ __forceinline__ void vm_wait(int younger) {  // wave-uniform: all but the `younger` newest are done
    asm volatile("" ::: "memory");  // (the builtin is IntrNoMem: loads and LDS reads may not cross it either way)
    switch (younger) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F74); break;
    }
    asm volatile("" ::: "memory");
}
// a value whose load must be complete -- for hipcc too -- from here on
template <typename T> __device__ __forceinline__ void settle(T &v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two registers");
    asm volatile("" : "+v"(v));
}

// ----------------------------------------------------------------------------------------
// ComplexF64, 8 right-hand sides: the matrix pipe.  A complex product with 8 columns is a REAL product with 16:
//     Y = B X,  X' = X as 16 real columns (Re, Im interleaved),  X'' = i X likewise   =>   Y (interleaved) = Re(B) X' + Im(B) X''
// -- exactly the N = 16 of v_mfma_f64_16x16x4_f64, and both halves land in ONE accumulator.  (conj(B): X'' negated.)
// The instruction runs at the vector FMA rate (tools/mfma_rate.hip: 47 vs 56 TFLOP/s), so in real arithmetic -- 8
// of the 16 columns idle -- it buys nothing; here it replaces 64 v_fma_f64 wave-instructions per 16 bytes of
// matrix and lane by 4 MFMAs per 64 lanes, and the K accumulators / x rows per lane (241 VGPRs = 2 waves per SIMD,
// BEM x 8 at 6.1 single products) by 8 VGPRs per 16 x 16 output tile.
//   lane = (ln = lane % 16, lk = lane / 16);  A operand: lane holds A[ln][lk], B operand: B[lk][ln],
//   C / D: column ln, rows lk + 4 r (r = 0..3)                      (guide: cdna_hip_programming.md, f64 layout)
// Forward half, per tile of 16 rows x 16 columns (4 loads of 16 bytes per lane: lane = row ln of the row block,
// column 4 j + lk): A = Re / Im of the loaded element, B = X' / X'' of the staged x slice (LDS, [column][k] complex
// = 16 doubles per column; X'' is X' with neighbouring lanes swapped and a sign: one DPP move), accumulator = the row block's
// 16 x 16 sums for the whole panel.
// Transposed half: the tile goes through LDS once ([column][row], stride 17 units) and comes back with lane =
// (column ln, row 4 q + lk) -- as the B operand; A = X' / X'' of the panel's x ROWS (registers, loaded once per
// panel), so the 16 x 16 sums of a column tile come out with the COLUMN on the lane (consecutive lanes = consecutive
// y entries); they are complete after the panel's row blocks and leave as scalar atomics, 4 per lane.
// ----------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
#ifndef BSM_MFMA_C128
#define BSM_MFMA_C128 1
#endif
#ifndef BSM_MFMA_C128_WGS  // resident workgroups per CU the ComplexF64 instance is compiled for (3: 168 VGPRs)
#define BSM_MFMA_C128_WGS 3
#endif
template <typename T, int K> constexpr bool kMfmaPath = BSM_MFMA_C128 && std::is_same<T, c128>::value && K == 8;
// ComplexF32 likewise on v_mfma_f32_16x16x4_f32 (C / D: column ln, rows 4 lk + r -- the f32 map, not the f64 one).  A
// 16-byte load holds TWO columns of a row (strip = 2 columns): one load feeds 4 MFMAs (2 columns x Re / Im), the k
// index of an MFMA runs over the 4 strips of the load.  For the transposed sums to leave as contiguous runs (one
// wave-instruction = Re and Im of 16 consecutive y entries for two k) the x rows enter the A operand with their 16
// components in transposed order: lane ln holds component 4 (ln % 4) + ln / 4, so accumulator row 4 lk + r is
// component 4 r + lk = (k = 2 r + lk / 2, Re / Im = lk % 2).
#ifndef BSM_MFMA_C64
#define BSM_MFMA_C64 1
#endif
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <typename T, int K> constexpr bool kMfmaPath32 = BSM_MFMA_C64 && std::is_same<T, c64>::value && K == 8;
// Real arithmetic: N = 16 is 16 right-hand sides.  The K = 16 instances (bsm_mul_multi: batches of 16, remainders of 9-15
// padded) run the same loop with ONE MFMA per operand (no X''): Float64 = two columns per 16-byte load and the f64
// accumulator map, Float32 = four columns per load and the f32 map with the components in transposed order.
#ifndef BSM_MFMA_REAL
#define BSM_MFMA_REAL 1
#endif
template <typename T, int K> constexpr bool kMfmaReal = BSM_MFMA_REAL && kRealType<T> && K == 16;
template <typename T, int K> constexpr bool kMfmaAny = kMfmaPath<T, K> || kMfmaPath32<T, K> || kMfmaReal<T, K>;
__device__ __forceinline__ v4f64 mfma16(double a, double b, v4f64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f32 mfma16(float a, float b, v4f32 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// columns per staged chunk: the matrix-pipe kernels take 64 (whole 16-column tiles; 64 x K elements is also the combine
// slab of coloured / exclusive launches) -- ComplexF32: 4 KB per wave instead of 8, a fourth workgroup per CU
template <typename T, int K> constexpr int x_chunk_cols_multi() {
    return kMfmaAny<T, K> ? 64 : x_chunk_cols_multi_vec<T, K>();
}

template <typename T, int L, int P, bool FWD, bool TRN, int K>
__device__ __forceinline__ void run_panel_multi(const WaveD &wd, const uint4 *__restrict__ values,
                                                const int *__restrict__ rows,
                                                const int *__restrict__ cols, const T *__restrict__ x,
                                                long long ldx, T *__restrict__ y, long long ldy, T alpha,
                                                int flags, int lane, T *xs, Vec16<T> *tile, int *ixm, T (&out)[K],
                                                bool &fwd_done) {
    constexpr int E = TT<T>::E;
    constexpr int G = 64 / P;
    constexpr int NC = G * L * E;
    constexpr bool PIPE = kTilePipe<T, L, TRN, K>;  // matrix tiles prefetched into LDS (below)
    constexpr int XCH = PIPE ? x_chunk_cols_pipe<T, L>() : x_chunk_cols_multi<T, K>();
    static_assert(kMfmaAny<T, K> || XCH % NC == 0, "x chunk must hold whole iterations");
    const bool opT = (flags & FLAG_OPT) != 0;
    const bool cjf = (flags & FLAG_CONJ) != 0;
    const int kact = ((flags >> FLAG_KACT_SHIFT) & 15) ? ((flags >> FLAG_KACT_SHIFT) & 15) : K;
    auto kc = [&](int k) { return k < kact ? k : kact - 1; };  // the column of X a (possibly padded) slot reads
    const int m = wd.m;
    const int i = lane & (P - 1);
    const int g = lane / P;
    const bool row_ok = i < m;

    T acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = zero_of(T{});
    // Transposed half, V[w][k] = sum_i B[i][w] * X[row(i)][k].  With the lane on the row (the layout the
    // matrix arrives in) every one of the K right-hand sides would need its own cross-lane reduction per
    // iteration (K halving butterflies: the multi-RHS fused products were bound by exactly that: C3 x 8 at
    // 2.6 products, the BEM fixture at 6.4).  Instead the loaded tile (L * G strips of P rows, 16 bytes per
    // lane and load) goes through LDS once and comes back in a COLUMN-role layout: lane (sp = lane % NS,
    // rg = lane / NS) holds strip sp (E columns) for the L rows rg * L .. rg * L + L - 1 -- the same 16
    // bytes per lane and load, transposed.  The x entries of those L rows stay in registers for the whole
    // piece, the transposed product becomes L * E * K in-lane FMAs like the forward one, and only E * K
    // partial sums per lane are reduced over the RG = 64 / NS lanes of a strip.
    constexpr int NS = G * L;
    constexpr int RG = 64 / NS;
    constexpr int PM = P + 1;  // strip stride in the LDS tile (16-byte units): conflict-free for both layouts
    static_assert(NS <= 64, "an iteration's strips must fit the wave");
    const int sp = lane % NS, rg = lane / NS;
    // matrix-pipe path (above): row blocks of 16, the x rows of the panel as B operands, one accumulator per row block
    constexpr bool MF = kMfmaPath<T, K>;
    constexpr int MR = (P + 15) / 16;
    const int ln = lane & 15, lk = lane >> 4;
    double xr1[(MF && TRN) ? 4 * MR : 1];  // (X'' of a row is X' with neighbouring lanes swapped, and a sign)
    constexpr bool MFR = kMfmaReal<T, K>;
    constexpr bool MFR64 = MFR && sizeof(T) == 8, MFR32 = MFR && sizeof(T) == 4;
    v4f64 facc[(MF || MFR64) ? MR : 1];
    if constexpr (MF) {
#pragma unroll
        for (int rb = 0; rb < MR; ++rb) facc[rb] = v4f64{0.0, 0.0, 0.0, 0.0};
        if (TRN) {
#pragma unroll
            for (int q = 0; q < 4 * MR; ++q) {
                const int r = 4 * q + lk;
                double re = 0.0, im = 0.0;
                if (r < m && !BSM_DBG(DBG_NO_XGATHER)) {
                    const int ri = (wd.rbase >= 0) ? wd.rbase + r : rows[wd.row_off + r];
                    const double *px = reinterpret_cast<const double *>(&x[ri + kc(ln >> 1) * ldx]);
                    re = px[0];
                    im = px[1];
                }
                // (alpha goes in here: the transposed sums leave the lanes as they come out of the accumulator)
                xr1[q] = (ln & 1) ? alpha.re * im + alpha.im * re : alpha.re * re - alpha.im * im;
            }
        }
    }
    constexpr bool MF32 = kMfmaPath32<T, K>;
    constexpr bool MFA = MF || MF32 || MFR;
    float fr1[(MF32 && TRN) ? 4 * MR : 1], fr2[(MF32 && TRN) ? 4 * MR : 1];
    v4f32 facc32[(MF32 || MFR32) ? MR : 1];
    // real types, K = 16: the panel's x rows (alpha folded in) as A operands of the transposed half; the lane carries
    // component (= right-hand side) compA: ln for the f64 accumulator map, the transposed order for the f32 one
    const int compA = MFR32 ? 4 * (ln & 3) + (ln >> 2) : ln;
    T rr[(MFR && TRN) ? 4 * MR : 1];
    if constexpr (MFR) {
#pragma unroll
        for (int rb = 0; rb < MR; ++rb) {
            if constexpr (MFR64) facc[rb] = v4f64{0.0, 0.0, 0.0, 0.0};
            if constexpr (MFR32) facc32[rb] = v4f32{0.f, 0.f, 0.f, 0.f};
        }
        if (TRN) {
#pragma unroll
            for (int q = 0; q < 4 * MR; ++q) {
                const int r = 4 * q + lk;
                T v = zero_of(T{});
                if (r < m && !BSM_DBG(DBG_NO_XGATHER)) {
                    const int ri = (wd.rbase >= 0) ? wd.rbase + r : rows[wd.row_off + r];
                    v = mul(alpha, x[ri + kc(compA) * ldx]);
                }
                rr[q] = v;
            }
        }
    }
    if constexpr (MF32) {
#pragma unroll
        for (int rb = 0; rb < MR; ++rb) facc32[rb] = v4f32{0.f, 0.f, 0.f, 0.f};
        if (TRN) {
            const int comp = 4 * (ln & 3) + (ln >> 2);  // the component this lane carries in the A operand (above)
#pragma unroll
            for (int q = 0; q < 4 * MR; ++q) {
                const int r = 4 * q + lk;
                float re = 0.f, im = 0.f;
                if (r < m && !BSM_DBG(DBG_NO_XGATHER)) {
                    const int ri = (wd.rbase >= 0) ? wd.rbase + r : rows[wd.row_off + r];
                    const c64 xv = x[ri + kc(comp >> 1) * ldx];
                    re = alpha.re * xv.re - alpha.im * xv.im;  // (alpha goes in here)
                    im = alpha.re * xv.im + alpha.im * xv.re;
                }
                fr1[q] = (comp & 1) ? im : re;                    // X'
                fr2[q] = (comp & 1) ? re : -im;                   // X'' = i X
                if (cjf) fr2[q] = -fr2[q];
            }
        }
    }
    T xrr[(TRN && !MFA) ? L : 1][K];
    if (TRN && !MFA) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int r = rg * L + j;
            const bool ok = r < m;
            int ri = 0;
            if (ok) ri = (wd.rbase >= 0) ? wd.rbase + r : rows[wd.row_off + r];
#pragma unroll
            for (int k = 0; k < K; ++k) xrr[j][k] = ok ? x[ri + kc(k) * ldx] : zero_of(T{});
        }
    }

    const PieceD pc = wd.first;
    if (wd.npieces > 0) {
        const int xbase = pc.xbase;
        const int col_off = pc.col_off;
        const int nstrips = pc.nstrips;
        const int ncols = pc.ncols;
        // per-column kinds (a symmetric row group holds its diagonal block and its off-diagonal
        // blocks in one panel): forward uses a column unless (op T/C and it is not KIND_OFF),
        // transposed uses it iff (op T/C or KIND_OFF)
        const int kinds = pc.kind;
        const bool has_off = (kinds & kKindHasOff) != 0;
        const bool fwd_en = FWD && (!opT || has_off);
        const bool trn_en = TRN && (opT || has_off);
        const Vec16<T> *__restrict__ vb = reinterpret_cast<const Vec16<T> *>(
            values + (((uint64_t)pc.val_hi << 32) | pc.val_lo));
        const int s1w = wd.seg1_w, s1x = wd.seg1_x - wd.seg1_w;
        const int s2w = wd.seg2_w, s2x = pc.seg2_x - wd.seg2_w;
        // -> x / y index of piece column w; `off` tells whether the column is KIND_OFF
        auto col_lookup = [&](int w, bool &off) -> int {
            if (xbase < 0) {
                const int raw = cols[col_off + w];
                off = raw >= 0 && (kinds & 3) == KIND_OFF;
                return raw & 0x7fffffff;
            }
            const int sh = w < s1w ? 0 : (w < s2w ? 2 : 4);
            off = ((kinds >> sh) & 3) == KIND_OFF;
            return w + (w < s1w ? xbase : (w < s2w ? s1x : s2x));
        };

        if constexpr (PIPE) {
            // Fused / transposed multi-RHS products at 2-3 waves per SIMD were bound by the bytes in flight (one
            // 4 KB tile per wave, and only while the wave was not computing: C3 x 8 ran at 2.5 TB/s with the VALU
            // 39 % and the LDS 43 % busy), and a second register tile costs a resident wave.  The tile goes
            // through LDS anyway (transposition above), so it is loaded THERE directly, one iteration ahead, with
            // no register landing: two LDS tiles per wave, global_load_lds into the one while the other is read
            // by row (forward half) and by column (transposed half).  In-order completion of vector memory
            // operations does the bookkeeping: the wait for tile n is `all but tile n+1's loads`, which also
            // covers every older atomic -- so iteration n's y contributions are issued AFTER that wait in
            // iteration n+1 (held in CF registers meanwhile), and nothing else in the loop may load from global
            // memory: a slice's gathered column indices are fetched with its x values and kept (in registers).
            {
                constexpr int NSI = G * L;
                constexpr int CF = (E * K / RG) > 1 ? (E * K / RG) : 1;
                constexpr int NE = CF > K ? CF / K : 1;  // columns (of E) a lane's CF sums belong to
                const int nit = (nstrips + NSI - 1) / NSI;
                const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)tile);
                const int hg = tile_swz<P, L>(g);
                const int cbase = sp * P + ((rg * L) ^ tile_swz<P, L>(sp));
                auto issue = [&](int it, int buf) {
                    const int s0 = it * NSI;
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        if (s0 + l * G < nstrips) {  // wave-uniform: the load is issued, and counted
                            const int s = s0 + l * G + g;
                            const int rho = i ^ tile_swz<P, L>(l * G) ^ hg;
                            if (BSM_DBG(DBG_NO_MATRIX)) {
                            } else if (rho < m && s < nstrips)
                                glds16_nt(&vb[(uint32_t)(s * m + rho)], lds0 + (unsigned)((buf * L + l) * 1024));
                            else
                                tile[(buf * L + l) * 64 + lane] = Vec16<T>{};
                        } else {
                            tile[(buf * L + l) * 64 + lane] = Vec16<T>{};
                        }
                    }
                };
                auto loads_of = [&](int it) {
                    const int left = nstrips - it * NSI;
                    const int nl = (left + G - 1) / G;
                    return nl < L ? nl : L;
                };
                int dq_yi[NE];
                T dq_val[CF];
#pragma unroll
                for (int q = 0; q < NE; ++q) dq_yi[q] = -1;
                int pos = 0, dup = 0;
                auto emit = [&]() {
#pragma unroll
                    for (int jj = 0; jj < CF; ++jj) {
                        const int yi = dq_yi[jj / K];
                        const int k = (pos + jj) % K;
                        // (atomics in coloured launches too: a read-modify-write's load would be waited for with
                        // vmcnt(0) by the compiler, i.e. drain the prefetch every iteration)
                        if (yi >= 0 && k < kact) atomic_acc(&y[yi + k * ldy], mul(alpha, dq_val[jj]));
                    }
                };
                // a slice of columns: x values (forward half) and the gathered indices.  The indices stay in
                // REGISTERS, lane c of cir[q] holding column q * 64 + c of the slice, and are fetched across lanes
                // (ds_bpermute) when a lane emits: an LDS array of them was the 256 bytes per wave that kept a fourth
                // workgroup of the 4-column fp64 kernel off the CU.
                int cir[XCH / 64];
#pragma unroll
                for (int q = 0; q < XCH / 64; ++q) cir[q] = 0;
                auto stage_slice = [&](int c0) {
#pragma unroll
                    for (int q = 0; q < XCH / 64; ++q) {
                        const int c = q * 64 + lane;
                        const int w = c0 + c;
                        if (xbase < 0) {
                            // (waited for HERE: a loaded register whose first use lies in the loop body makes hipcc
                            // wait vmcnt(0) there in every iteration, which drains the prefetched tile)
                            cir[q] = cols[col_off + min(w, ncols - 1)];
                            settle(cir[q]);
                        }
                        if (w < ncols + NC) {
                            bool ok = w < ncols, off = false;
                            if (fwd_en) {
                                const int xi = ok ? col_lookup(w, off) : 0;
                                ok = ok && (!opT || off);
#pragma unroll
                                for (int k = 0; k < K; ++k) xs[c * K + k] = ok ? x[xi + kc(k) * ldx] : zero_of(T{});
                            }
                        }
                    }
                };
                // A wave's start is a chain of dependent round trips (descriptor -> row list -> x rows, column list
                // -> x slice -> first tile), and a panel of the BEM fixture is 3-4 iterations long: the first tile's
                // loads go out first (they need the descriptor only), the first slice is staged while the x rows
                // above are still in flight, and only then everything is waited for -- three round trips, not six.
                if (nit > 0) issue(0, 0);
                if (nit > 0) stage_slice(0);
                // the x rows: no load hipcc knows of may be pending inside the loop, or its wait for it (a
                // vmcnt(0) at the first use, executed every iteration) would drain the prefetched tile
                if (TRN) {
#pragma unroll
                    for (int j = 0; j < L; ++j)
#pragma unroll
                        for (int k = 0; k < K; ++k) settle(xrr[j][k]);
                }
                for (int it = 0; it < nit; ++it) {
                    const int buf = it & 1;
                    const int s0 = it * NSI;
                    const int c0 = (s0 * E) / XCH * XCH;
                    if (s0 * E == c0 && it > 0) stage_slice(c0);
                    int younger = 0;
                    if (it + 1 < nit) {
                        issue(it + 1, buf ^ 1);
                        younger = BSM_DBG(DBG_NO_MATRIX) ? 0 : loads_of(it + 1);
                    }
                    vm_wait(younger);
                    if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS)) emit();  // iteration it-1's sums
                    if (fwd_en && !BSM_DBG(DBG_NO_FWD_HALF)) {
                        const int cb = s0 * E - c0;
#pragma unroll
                        for (int l = 0; l < L; ++l) {
                            const Vec16<T> bl = tile[(buf * L + l) * 64 + (lane ^ tile_swz<P, L>(l * G) ^ hg)];
                            const T *xp = &xs[(cb + (l * G + g) * E) * K];
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const T bv = cj(bl.v[e], cjf);
#pragma unroll
                                for (int k = 0; k < K; ++k) acc[k] = madd(acc[k], bv, xp[e * K + k]);
                            }
                        }
                    }
                    if (!trn_en || BSM_DBG(DBG_NO_TRN_HALF)) continue;
                    T tv[E * K];
#pragma unroll
                    for (int q = 0; q < E * K; ++q) tv[q] = zero_of(T{});
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        const Vec16<T> u = tile[buf * L * 64 + (cbase ^ j)];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const T bv = cj(u.v[e], cjf);
#pragma unroll
                            for (int k = 0; k < K; ++k) tv[e * K + k] = madd(tv[e * K + k], bv, xrr[j][k]);
                        }
                    }
                    pos = 0, dup = 0;
                    ReduceAbove<T, E * K, NS>::run(tv, lane, pos, dup);
                    const int s = s0 + sp;
                    const bool mine = (lane & dup) == 0 && s < nstrips;
#pragma unroll
                    for (int q = 0; q < NE; ++q) {
                        const int w = s * E + pos / K + q;
                        int raw = 0;
                        if (xbase < 0) {  // (every lane takes part in the exchange)
                            const int c = (w - c0) & (XCH - 1);
#pragma unroll
                            for (int h = 0; h < XCH / 64; ++h) {
                                const int v = __shfl(cir[h], c & 63, 64);
                                if ((c >> 6) == h) raw = v;
                            }
                        }
                        int yi = -1;
                        if (mine && w < ncols) {
                            bool off;
                            if (xbase < 0) {
                                off = raw >= 0 && (kinds & 3) == KIND_OFF;
                                yi = raw & 0x7fffffff;
                            } else {
                                yi = col_lookup(w, off);
                            }
                            if (!(opT || off)) yi = -1;
                        }
                        dq_yi[q] = yi;
                    }
#pragma unroll
                    for (int jj = 0; jj < CF; ++jj) dq_val[jj] = tv[jj];
                }
                emit();
            }
        }
        // the x slice (forward half) and the y indices (transposed half) of the chunk of columns at c0
        // (matrix-pipe path in accumulate mode: alpha goes into the slice, the forward sums leave from the accumulators)
        const bool fold = MFA && !(flags & (FLAG_DIRECT | FLAG_RMW));
        // (the lane that stages a column writes its K entries 64 / 128 bytes apart from its neighbours': 16- / 32-way bank
        // conflicts per store -- an XOR swizzle of the slot (k ^ column index within the bank row) was measured: +-0, the
        // staging is bound by the K x 64 scattered line requests of the gather, not by the LDS)
        auto stage_columns = [&](int c0) {
            if (fwd_en || (BSM_MULTI_IX && trn_en)) {
#pragma unroll
                for (int q = 0; q < XCH / 64; ++q) {
                    const int c = q * 64 + lane;
                    const int w = c0 + c;
                    if (BSM_DBG(DBG_NO_XGATHER)) {  // (timing probe: no column list, no x loads)
                        if (BSM_MULTI_IX && TRN) ixm[c] = w < ncols ? w : -1;
                        if (fwd_en) {
#pragma unroll
                            for (int k = 0; k < K; ++k) xs[c * K + k] = zero_of(T{});
                        }
                    } else if (MFA || w < ncols + NC) {  // (the matrix-pipe tiles read whole 16-column tiles of the slice)
                        bool ok = w < ncols, off = false;
                        const int xi = ok ? col_lookup(w, off) : 0;
                        // the chunk's y indices stay in LDS for the emission of its iterations (-1: the column takes no
                        // part): read from the column list there, every iteration waited for a dependent load in front
                        // of its atomics
                        if (BSM_MULTI_IX && TRN) ixm[c] = (ok && (opT || off)) ? xi : -1;
                        ok = ok && (!opT || off);
                        if (fwd_en) {
#pragma unroll
                            for (int k = 0; k < K; ++k)
                                xs[c * K + k] = ok ? (fold ? mul(alpha, x[xi + kc(k) * ldx]) : x[xi + kc(k) * ldx]) : zero_of(T{});
                        }
                    }
                }
            }
        };
        if constexpr (MF) {
            // one step = one row block (16 rows) of one column tile (16 columns): 4 loads of 16 bytes per lane, issued
            // one step ahead of their use
            const double *xsd = reinterpret_cast<const double *>(xs);
            const int nrb = (m + 15) >> 4;
            auto fetch = [&](c128(&b)[4], int t0, int rb) {
                const int row = rb * 16 + ln;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int w = t0 + 4 * j + lk;
                    if (row < m && w < ncols && !BSM_DBG(DBG_NO_MATRIX)) {
                        const Vec16<T> q = load_stream16(&vb[(uint32_t)(w * m + row)]);
                        b[j] = q.v[0];
                    } else {
                        b[j] = c128{0.0, 0.0};
                    }
                }
            };
            // Vector-memory operations retire in issue order and an atomic's round trip to the memory side is long: sums
            // emitted right behind a tile would stand between the NEXT loads and their wait.  They are parked (4 sums, 4
            // indices per lane) and leave one step later, right BEHIND the following step's loads -- whose wait then
            // only has to let the 4 younger atomics pass.
            // The transposed sums come out TRANSPOSED (operands swapped: A = the x rows, B = the tile), lane = (column
            // ln, component n = lk + 4 r = Re / Im of k = n / 2): one wave-instruction adds Re and Im of 16 consecutive
            // columns for two k -- two runs of 256 contiguous bytes, 8-10 cache lines.  What the memory-side atomics
            // cost is the number of LINES a wave-instruction touches (measured on the BEM fixture, atomics alone:
            // 12-16 lines 390 us, 16-20 lines 520 us).
            double pd[4];
            int pyi = -1;
            bool pending = false;
            auto emit = [&]() {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = lk + 4 * r;  // k = n / 2, Re / Im = n % 2 (alpha is in the x rows already)
                    const double val = pd[r];
                    if (pyi >= 0 && (n >> 1) < kact) {
                        double *yp = reinterpret_cast<double *>(&y[pyi + (n >> 1) * ldy]) + (n & 1);
                        if (flags & FLAG_RMW)
                            *yp += val;
                        else if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS))
                            atomicAdd(yp, val);
                    }
                }
                pending = false;
            };
            c128 nxt[4];
            fetch(nxt, 0, 0);
            for (int t0 = 0; t0 < ncols; t0 += 16) {
                const int c0 = t0 & ~(XCH - 1);
                if (t0 == c0) stage_columns(c0);
                const int t_end = min(ncols, c0 + XCH);
                v4f64 dt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int rb = 0; rb < MR; ++rb) {
                    if (rb >= nrb) break;  // (wave-uniform)
                    c128 b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = nxt[j];
                    if (rb + 1 < nrb)
                        fetch(nxt, t0, rb + 1);
                    else if (t0 + 16 < ncols)
                        fetch(nxt, t0 + 16, 0);
                    if (pending) emit();
                    if (fwd_en && !BSM_DBG(DBG_NO_FWD_HALF)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int wl = t0 - c0 + 4 * j + lk;  // column of the staged slice ([column][k] complex)
                            const double x1 = xsd[wl * 16 + ln];
                            double x2 = dppx<DPP_QUAD_XOR1>(x1);  // the other component of the same k: the neighbouring lane
                            x2 = (((ln & 1) == 0) != cjf) ? -x2 : x2;
                            facc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b[j].re, facc[rb], 0, 0, 0);
                            facc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b[j].im, facc[rb], 0, 0, 0);
                        }
                    }
                    if (trn_en && !BSM_DBG(DBG_NO_TRN_HALF)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) tile[(4 * j + lk) * 17 + ln].v[0] = b[j];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const c128 u = tile[ln * 17 + 4 * q + lk].v[0];
                            const double r1 = xr1[rb * 4 + q];
                            double r2 = dppx<DPP_QUAD_XOR1>(r1);
                            r2 = (((ln & 1) == 0) != cjf) ? -r2 : r2;
                            dt = __builtin_amdgcn_mfma_f64_16x16x4f64(r1, u.re, dt, 0, 0, 0);
                            dt = __builtin_amdgcn_mfma_f64_16x16x4f64(r2, u.im, dt, 0, 0, 0);
                        }
                    }
                }
                if (trn_en) {
                    // lane (ln, lk) holds components lk + 4 r of column t0 + ln
#pragma unroll
                    for (int r = 0; r < 4; ++r) pd[r] = dt[r];
                    pyi = (t0 + ln < t_end) ? ixm[t0 + ln - c0] : -1;  // (read now: the next chunk's staging overwrites the list)
                    pending = true;
                }
            }
            if (pending) emit();
        }
        if constexpr (MF32) {
            // one step = one row block (16 rows) of one column tile (16 columns = 8 strips): 2 loads of 16 bytes per
            // lane (lane = row ln, strip 4 j + lk), issued one step ahead; everything else as in the ComplexF64 loop
            const float *xsf = reinterpret_cast<const float *>(xs);
            c64 *tile8 = reinterpret_cast<c64 *>(tile);
            const int comp = 4 * (ln & 3) + (ln >> 2);
            const int nrb = (m + 15) >> 4;
            auto fetch = [&](Vec16<T>(&b)[2], int t0, int rb) {
                const int row = rb * 16 + ln;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int sidx = (t0 >> 1) + 4 * j + lk;
                    if (row < m && sidx < nstrips && !BSM_DBG(DBG_NO_MATRIX)) {
                        b[j] = load_stream16(&vb[(uint32_t)(sidx * m + row)]);
                    } else {
                        b[j].v[0] = c64{0.f, 0.f};
                        b[j].v[1] = c64{0.f, 0.f};
                    }
                }
            };
            float pd[4];
            int pyi = -1;
            bool pending = false;
            auto emit = [&]() {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kq = 2 * r + (lk >> 1);  // accumulator row 4 lk + r = component 4 r + lk
                    if (pyi >= 0 && kq < kact) {
                        float *yp = reinterpret_cast<float *>(&y[pyi + kq * ldy]) + (lk & 1);
                        if (flags & FLAG_RMW)
                            *yp += pd[r];
                        else if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS))
                            atomicAdd(yp, pd[r]);
                    }
                }
                pending = false;
            };
            Vec16<T> nxt[2];
            fetch(nxt, 0, 0);
            for (int t0 = 0; t0 < ncols; t0 += 16) {
                const int c0 = t0 & ~(XCH - 1);
                if (t0 == c0) stage_columns(c0);
                const int t_end = min(ncols, c0 + XCH);
                v4f32 dt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rb = 0; rb < MR; ++rb) {
                    if (rb >= nrb) break;  // (wave-uniform)
                    Vec16<T> b[2];
                    b[0] = nxt[0];
                    b[1] = nxt[1];
                    if (rb + 1 < nrb)
                        fetch(nxt, t0, rb + 1);
                    else if (t0 + 16 < ncols)
                        fetch(nxt, t0 + 16, 0);
                    if (pending) emit();
                    if (fwd_en && !BSM_DBG(DBG_NO_FWD_HALF)) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int wl = t0 - c0 + 2 * (4 * j + lk) + e;  // column of the staged slice
                                const float x1 = xsf[wl * 16 + comp];  // (components in the transposed order, as the x rows)
                                float x2 = xsf[wl * 16 + (comp ^ 1)];
                                x2 = (((comp & 1) == 0) != cjf) ? -x2 : x2;
                                facc32[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b[j].v[e].re, facc32[rb], 0, 0, 0);
                                facc32[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, b[j].v[e].im, facc32[rb], 0, 0, 0);
                            }
                    }
                    if (trn_en && !BSM_DBG(DBG_NO_TRN_HALF)) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 2; ++e) tile8[(2 * (4 * j + lk) + e) * 17 + ln] = b[j].v[e];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const c64 u = tile8[ln * 17 + 4 * q + lk];
                            dt = __builtin_amdgcn_mfma_f32_16x16x4f32(fr1[rb * 4 + q], u.re, dt, 0, 0, 0);
                            dt = __builtin_amdgcn_mfma_f32_16x16x4f32(fr2[rb * 4 + q], u.im, dt, 0, 0, 0);
                        }
                    }
                }
                if (trn_en) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pd[r] = dt[r];
                    pyi = (t0 + ln < t_end) ? ixm[t0 + ln - c0] : -1;
                    pending = true;
                }
            }
            if (pending) emit();
        }
        if constexpr (MFR) {
            // one step = one row block (16 rows) of one column tile (16 columns = 16 / E strips): 16 / (4 E) loads of
            // 16 bytes per lane (lane = row ln, strip 4 j + lk), issued one step ahead; E MFMAs per load and half
            using V4 = typename std::conditional<MFR64, v4f64, v4f32>::type;
            constexpr int NLD = 4 / E;  // loads per lane and step: 2 (Float64), 1 (Float32)
            T *tileT = reinterpret_cast<T *>(tile);
            const int nrb = (m + 15) >> 4;
            auto fetch = [&](Vec16<T>(&b)[NLD], int t0, int rb) {
                const int row = rb * 16 + ln;
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int sidx = t0 / E + 4 * j + lk;
                    if (row < m && sidx < nstrips && !BSM_DBG(DBG_NO_MATRIX)) {
                        b[j] = load_stream16(&vb[(uint32_t)(sidx * m + row)]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) b[j].v[e] = zero_of(T{});
                    }
                }
            };
            auto comp_of = [&](int r) { return MFR64 ? lk + 4 * r : 4 * r + lk; };  // accumulator row -> right-hand side
            T pd[4];
            int pyi = -1;
            bool pending = false;
            auto emit = [&]() {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kq = comp_of(r);
                    if (pyi >= 0 && kq < kact) {
                        T *yp = &y[pyi + kq * ldy];
                        if (flags & FLAG_RMW)
                            *yp += pd[r];
                        else if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS))
                            atomicAdd(yp, pd[r]);
                    }
                }
                pending = false;
            };
            Vec16<T> nxt[NLD];
            fetch(nxt, 0, 0);
            for (int t0 = 0; t0 < ncols; t0 += 16) {
                const int c0 = t0 & ~(XCH - 1);
                if (t0 == c0) stage_columns(c0);
                const int t_end = min(ncols, c0 + XCH);
                V4 dt = {0, 0, 0, 0};
#pragma unroll
                for (int rb = 0; rb < MR; ++rb) {
                    if (rb >= nrb) break;  // (wave-uniform)
                    Vec16<T> b[NLD];
#pragma unroll
                    for (int j = 0; j < NLD; ++j) b[j] = nxt[j];
                    if (rb + 1 < nrb)
                        fetch(nxt, t0, rb + 1);
                    else if (t0 + 16 < ncols)
                        fetch(nxt, t0 + 16, 0);
                    if (pending) emit();
                    if (fwd_en && !BSM_DBG(DBG_NO_FWD_HALF)) {
#pragma unroll
                        for (int j = 0; j < NLD; ++j)
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const int wl = t0 - c0 + E * (4 * j + lk) + e;  // column of the staged slice
                                const T x1 = xs[wl * 16 + compA];
                                if constexpr (MFR64) facc[rb] = mfma16(x1, b[j].v[e], facc[rb]);
                                if constexpr (MFR32) facc32[rb] = mfma16(x1, b[j].v[e], facc32[rb]);
                            }
                    }
                    if (trn_en && !BSM_DBG(DBG_NO_TRN_HALF)) {
#pragma unroll
                        for (int j = 0; j < NLD; ++j)
#pragma unroll
                            for (int e = 0; e < E; ++e) tileT[(E * (4 * j + lk) + e) * 17 + ln] = b[j].v[e];
#pragma unroll
                        for (int q = 0; q < 4; ++q) dt = mfma16(rr[rb * 4 + q], tileT[ln * 17 + 4 * q + lk], dt);
                    }
                }
                if (trn_en) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pd[r] = dt[r];
                    pyi = (t0 + ln < t_end) ? ixm[t0 + ln - c0] : -1;
                    pending = true;
                }
            }
            if (pending) emit();
        }
        for (int c0 = 0; !PIPE && !MFA && c0 < ncols; c0 += XCH) {
            stage_columns(c0);
            const int s_end = min(nstrips, (c0 + XCH) / E);
            for (int s0 = c0 / E; s0 < s_end; s0 += G * L) {
                // (issuing the next iteration's matrix loads before this iteration's arithmetic -- two register
                // buffers -- was measured here too: C3 x 8 379 -> 462 us, C4 slice x 8 406 -> 570 us)
                Vec16<T> b[L];
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int s = s0 + l * G + g;
                    if (row_ok && s < nstrips) {
                        b[l] = load_stream16(&vb[(uint32_t)(s * m + i)]);  // multi-RHS: always with the hint
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) b[l].v[e] = zero_of(T{});
                    }
                }
                if (fwd_en) {
                    const int cb = (s0 - c0 / E) * E;
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const T *xp = &xs[(cb + (l * G + g) * E) * K];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const T bv = cj(b[l].v[e], cjf);
#pragma unroll
                            for (int k = 0; k < K; ++k) acc[k] = madd(acc[k], bv, xp[e * K + k]);
                        }
                    }
                }
                if (trn_en) {
                    // registers (row role) -> LDS -> registers (column role); one wave, LDS runs in order
#pragma unroll
                    for (int l = 0; l < L; ++l) tile[(l * G + g) * PM + i] = b[l];
                    T tv[E * K];
#pragma unroll
                    for (int q = 0; q < E * K; ++q) tv[q] = zero_of(T{});
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        const Vec16<T> u = tile[sp * PM + rg * L + j];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const T bv = cj(u.v[e], cjf);
#pragma unroll
                            for (int k = 0; k < K; ++k) tv[e * K + k] = madd(tv[e * K + k], bv, xrr[j][k]);
                        }
                    }
                    int pos = 0, dup = 0;
                    ReduceAbove<T, E * K, NS>::run(tv, lane, pos, dup);
                    constexpr int CF = (E * K / RG) > 1 ? (E * K / RG) : 1;
                    const int s = s0 + sp;  // the lane's strip of the piece
                    if ((lane & dup) == 0 && s < nstrips) {
#pragma unroll
                        for (int jj = 0; jj < CF; ++jj) {
                            const int q = pos + jj;
                            const int e = q / K, k = q % K;
                            const int w = s * E + e;
                            if (w < ncols) {
                                bool off = false;
                                int yi;
                                if (BSM_MULTI_IX) {
                                    yi = ixm[w - c0];
                                    off = yi >= 0;
                                } else {
                                    yi = col_lookup(w, off);
                                    off = opT || off;
                                }
                                if (off && k < kact) {
                                    T *yp = &y[yi + k * ldy];
                                    const T val = mul(alpha, tv[jj]);
                                    if (flags & FLAG_RMW)
                                        *yp = add(*yp, val);
                                    else if (!BSM_DBG(DBG_NO_GLOBAL_ATOMICS))
                                        atomic_acc(yp, val);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if constexpr (MFA) {
        // The forward sums sit in the accumulators TRANSPOSED as well (A = the x slice, B = the tile): lane = (row ln of
        // the row block, lk), register r = component  lk + 4 r (ComplexF64) / 4 r + lk (ComplexF32: the slice enters
        // in transposed component order), i.e. Re and Im of two k for 16 consecutive rows per wave-instruction.
        // (real types, K = 16: a component is a right-hand side)
        using R = typename std::conditional<MF || MFR64, double, float>::type;
        constexpr bool M64 = MF || MFR64;  // the f64 accumulator map
        constexpr int CS = MFR ? 0 : 1;    // component -> k: comp >> CS; Re / Im: comp & CS
        auto comp_of = [&](int r) { return M64 ? lk + 4 * r : 4 * r + lk; };
        if (FWD && !(flags & (FLAG_DIRECT | FLAG_RMW))) {
            // atomic mode: every wave adds its own partial sums (alpha is in the slice already) -- contiguous runs
            // again instead of Re and Im of one k per instruction, no slab, no combine (a group's waves add separately;
            // coloured launches keep the combine: their plain read-modify-write is race-free between groups only)
            if (wd.npieces > 0 && (!(flags & FLAG_OPT) || (wd.first.kind & kKindHasOff)) && !BSM_DBG(DBG_NO_FWD_OUT)) {
#pragma unroll
                for (int rb = 0; rb < MR; ++rb) {
                    const int row = rb * 16 + ln;
                    if (rb * 16 >= m) break;
                    int yi = -1;
                    if (row < m) yi = (wd.rbase >= 0) ? wd.rbase + row : rows[wd.row_off + row];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cq = comp_of(r);
                        R val;
                        if constexpr (M64) val = facc[rb][r]; else val = facc32[rb][r];
                        if (yi >= 0 && (cq >> CS) < kact)
                            atomicAdd(reinterpret_cast<R *>(&y[yi + (cq >> CS) * ldy]) + (cq & CS), val);
                    }
                }
            }
            fwd_done = true;
            return;
        }
        if (FWD) {
            // exclusive launches (plain stores, beta fused, groups combined in LDS by the caller): lane = row, K complex
            // sums -- through the dead x slice, 64 rows x 16 components, component n of row i at i * 16 + (n ^ s(i))
            R *sl = reinterpret_cast<R *>(xs);
            auto swz = [&](int row) { return M64 ? ((row >> 1) & 15) : (row & 15); };
#pragma unroll
            for (int rb = 0; rb < MR; ++rb) {
                const int row = rb * 16 + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    R val;
                    if constexpr (M64) val = facc[rb][r]; else val = facc32[rb][r];
                    sl[row * 16 + (comp_of(r) ^ swz(row))] = val;
                }
            }
            const int sw = swz(lane);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                T a = zero_of(T{});
                if (lane < 16 * MR) {
                    if constexpr (MFR) {
                        a = sl[lane * 16 + (k ^ sw)];
                    } else {
                        a.re = sl[lane * 16 + ((2 * k) ^ sw)];
                        a.im = sl[lane * 16 + ((2 * k + 1) ^ sw)];
                    }
                }
                out[k] = a;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T a = acc[k];
        if (FWD) {
#pragma unroll
            for (int d = P; d < 64; d <<= 1) a = add(a, shx(a, d));
        }
        out[k] = a;
    }
}

// resident workgroups per CU the register allocation must leave room for: 3 for the pipelined fp32 kernels (their
// LDS admits 3; fused: 189 VGPRs = 2 per CU without the bound, 168 + 16 spilled dwords with it: C3 in fp32 x 8
// 176 -> 159 us), 2 otherwise (the fp64 ones fit 3 by themselves)
template <typename T, int L, bool FWD, bool TRN, int K>
__global__ void __launch_bounds__(64 * kWavesPerWg, ((kMfmaPath32<T, K> || (kMfmaReal<T, K> && sizeof(T) == 4)) ? 4 : ((kTilePipe<T, L, TRN, K> && sizeof(T) == 4) || (kMfmaAny<T, K> && BSM_MFMA_C128_WGS == 3) ? 3 : 2)))
    panel_kernel_multi(const WaveWork *__restrict__ waves, const uint4 *__restrict__ values,
                       const int *__restrict__ rows, const int *__restrict__ cols,
                       const T *__restrict__ x, long long ldx, T *__restrict__ y, long long ldy, T alpha,
                       T beta, int flags, unsigned wg_base) {
    constexpr bool PIPE = kTilePipe<T, L, TRN, K>;
    constexpr int XCH = PIPE ? x_chunk_cols_pipe<T, L>() : x_chunk_cols_multi<T, K>();
    constexpr int XS = XCH * K;  // >= 64*K: also holds the combine slab
    // 16-byte units: max over P of (64 / P) * L strips of P + 1 units; the pipelined kernels hold two tiles
    // (matrix-pipe kernels: one 16 x 16 tile of elements, column stride 17)
    constexpr int TILE = PIPE ? 2 * L * 64 : (kMfmaAny<T, K> ? (16 * 17 * (int)sizeof(T) + 15) / 16 : L * 72);
    __shared__ __attribute__((aligned(16))) T xs[kWavesPerWg][FWD ? XS : 1];
    __shared__ Vec16<T> tl[kWavesPerWg][TRN ? TILE : 1];
    __shared__ int ixm[kWavesPerWg][(TRN && !PIPE && BSM_MULTI_IX) ? XCH : 1];  // y indices of the staged chunk (register path)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const WaveD wd = load_wave(waves + ((size_t)(blockIdx.x + wg_base) * kWavesPerWg + wave));
    const int work = wd.work;
    const int m = wd.m;

    T u[K];
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] = zero_of(T{});
    bool fwd_done = false;  // the wave has added its forward sums to y itself (matrix-pipe path, accumulate mode)
    if (work == WORK_PANEL) {
        if (m <= 8)
            run_panel_multi<T, L, 8, FWD, TRN, K>(wd, values, rows, cols, x, ldx, y, ldy, alpha, flags, lane, xs[wave], tl[wave], ixm[wave], u, fwd_done);
        else if (m <= 16)
            run_panel_multi<T, L, 16, FWD, TRN, K>(wd, values, rows, cols, x, ldx, y, ldy, alpha, flags, lane, xs[wave], tl[wave], ixm[wave], u, fwd_done);
        else if (m <= 32)
            run_panel_multi<T, L, 32, FWD, TRN, K>(wd, values, rows, cols, x, ldx, y, ldy, alpha, flags, lane, xs[wave], tl[wave], ixm[wave], u, fwd_done);
        else
            run_panel_multi<T, L, 64, FWD, TRN, K>(wd, values, rows, cols, x, ldx, y, ldy, alpha, flags, lane, xs[wave], tl[wave], ixm[wave], u, fwd_done);
    }
    const bool direct = (flags & FLAG_DIRECT) != 0;
    const bool sz = (flags & FLAG_STRONG_ZERO) != 0;
    const int kact = ((flags >> FLAG_KACT_SHIFT) & 15) ? ((flags >> FLAG_KACT_SHIFT) & 15) : K;
    if (FWD) {
        if (wd.wg_sync) {
            // a wave's staged x slice is dead once it has left its loop: reuse it as this wave's
            // part of the combine slab [wave][lane][k]
#pragma unroll
            for (int k = 0; k < K; ++k) xs[wave][lane * K + k] = u[k];
            __syncthreads();
        }
        if (work == WORK_PANEL && wd.lead && !fwd_done && !BSM_DBG(DBG_NO_FWD_OUT)) {
            for (int w2 = 1; w2 < wd.grp; ++w2)
#pragma unroll
                for (int k = 0; k < K; ++k) u[k] = add(u[k], xs[wave + w2][lane * K + k]);
            if (lane < m) {
                const int yi = (wd.rbase >= 0) ? wd.rbase + lane : rows[wd.row_off + lane];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k >= kact) continue;
                    T *yp = &y[yi + k * ldy];
                    const T val = mul(alpha, u[k]);
                    if (direct) {
                        *yp = sz ? val : madd(val, beta, *yp);
                    } else if (flags & FLAG_RMW) {
                        *yp = add(*yp, val);
                    } else {
                        atomic_acc(yp, val);
                    }
                }
            }
        }
    }
    if (work == WORK_SCALE && direct) {
        const int cnt = wd.first.ncols;
        for (int r = lane; r < cnt; r += 64)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k >= kact) continue;
                T *yp = &y[wd.rbase + r + k * ldy];
                *yp = sz ? zero_of(T{}) : mul(beta, *yp);
            }
    }
}

// ----------------------------------------------------------------------------------------
// INTERLEAVED multi-RHS pass (round 5): 8 ComplexF64 right-hand sides with X and the accumulated Y held ROW-major
// ("K-interleaved") in two work arrays of the handle:
//     Xr[i][c], W[i][c],  c = 2 k + (0: Re, 1: Im),  16 doubles = ONE 128-byte line per vector index
// Xr = alpha * X is written by il_pack_kernel in front of the pass, Y = beta * Y + W is read back (and W zeroed behind)
// by il_finish_kernel.  What it changes against panel_kernel_multi's matrix-pipe path (counters of round 4: 9.7 M atomic
// line requests and 11.6 M read requests per BEM launch, the vector L1 stalled on pending requests 87 % of the time):
//   * the x operand of an MFMA (component on lane % 16) is ONE coalesced 128-byte line per column / row index, loaded
//     straight from Xr into the operand register -- no K-fold gather of column-major X, no x slice in LDS (32 KB per
//     workgroup: 3 workgroups per CU), no dependent "column list -> x gather -> LDS" round trips per 64 columns;
//   * both halves are computed with the COMPONENT on the lane (A = the matrix tile, B = the x lines), so a sum leaves as
//     16 consecutive doubles of one line of W: 4 lines per atomic wave-instruction whatever the column list looks like
//     (column-major Y: one line per (index, k) -- 16-32 per instruction for the scattered lists of a BEM panel);
//   * the whole column list of a panel (<= 256 columns per refill) is staged once, so a wave's start is descriptor ->
//     {row list -> x rows, column list, first tile}: three round trips for the whole panel.
// One step = one row block (16 rows) of one column tile (16 columns), operands one step ahead, as in the path above.
// ----------------------------------------------------------------------------------------
// resident workgroups per CU the interleaved kernels are compiled for (their natural register need, fused instances --
// panels of at most 32 rows, two row blocks, one step ahead: Float32 / ComplexF32 84-92 VGPRs, Float64 110, ComplexF64 144;
// tall panels, four row blocks of the next 16 columns in flight: Float32 128, ComplexF32 145, Float64 184, ComplexF64 231)
#ifndef BSM_IL_C128_WGS
#define BSM_IL_C128_WGS 3
#endif
template <typename T, int MRMAX> constexpr int il_wgs() {
    constexpr bool f64 = std::is_same<T, double>::value;
    if (MRMAX > 2) return sizeof(T) == 4 ? 4 : (sizeof(T) == 16 || f64 ? 2 : 3);
    return sizeof(T) == 16 ? BSM_IL_C128_WGS : (f64 ? 4 : 5);
}
constexpr int kIlCols = 256;            // columns of a panel staged per refill of the index list
constexpr int IL_NOFWD = 1 << 30;       // staged column entry: takes no part in the forward half
constexpr int IL_NOTRN = (int)(1u << 31);  // ... in the transposed half
constexpr int IL_MASK = (1 << 30) - 1;

// The loop is written BRANCH-FREE on purpose.  hipcc places its own s_waitcnt in front of the first use of every loaded
// register, and wherever control flow (a lane-masked `if` around a load or an atomic, a scratch reload, paths with different
// numbers of memory operations) keeps it from counting the operations in flight exactly it waits for ALL of them:
// the first version of this kernel -- loads and atomics under `if (row < m && w < ncols)` -- compiled to a
// `s_waitcnt vmcnt(0)` in front of every step's MFMAs, i.e. the operands requested one step ahead were drained at once
// and every step cost a full memory round trip (tools/il_trace.py: 3.5 us per step, 35 us per 19 KB panel).  Here every
// load and every atomic of the loop is issued unconditionally, with indices clamped into the panel (rows >= m read row
// m - 1, columns >= ncols the last column) and the VALUES masked instead: rows beyond m meet x rows that are zero and
// their forward sums are never delivered, columns beyond the panel get a zero x operand and deliver +0.0.  One body per
// number of row blocks (NRB), so that a step is the same instruction sequence every time.
// atomic add of the lanes with `ok`, WITHOUT control flow: the other lanes are switched off for the one instruction
// (EXEC), not branched around -- a lane-masked `if` around an atomic becomes a branch, and hipcc then no longer knows how
// many operations are in flight behind it (above).  Masked lanes must not be routed to a dummy target instead: lanes of
// one instruction that add to the SAME address are serialised on the memory side (measured with +0.0 deliveries to a
// clamped index: the atomics of the BEM pass went from 30 to 390 us).  hipcc does not count the instruction either; it is
// always issued IN FRONT of the step's loads, so every wait it computes for those is still sufficient.
__device__ __forceinline__ void il_atomic_add(double *p, double v, bool ok) {
    unsigned long long save;
    const int flag = ok ? 1 : 0;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_cmpx_ne_u32_e32 0, %1\n\t"
        "global_atomic_add_f64 %2, %3, off\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(save)
        : "v"(flag), "v"(p), "v"(v)
        : "vcc", "memory");
}
__device__ __forceinline__ void il_atomic_add(float *p, float v, bool ok) {
    unsigned long long save;
    const int flag = ok ? 1 : 0;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_cmpx_ne_u32_e32 0, %1\n\t"
        "global_atomic_add_f32 %2, %3, off\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(save)
        : "v"(flag), "v"(p), "v"(v)
        : "vcc", "memory");
}

// The four element types of the interleaved pass: 16 real COMPONENTS per vector index -- 8 complex right-hand sides
// (component 2 k + Re / Im) or 16 real ones -- of type R, one N = 16 of v_mfma_{f64,f32}_16x16x4.  A 16-byte load holds E
// columns of one row; a step (16 rows x 16 columns) is NLD = 4 / E loads per lane (lane = row ln, strip 4 j + lk).
template <typename T> struct ILT;
template <> struct ILT<c128> { using R = double; using V4 = v4f64; static constexpr bool CPLX = true; static constexpr int KK = 8; };
template <> struct ILT<double> { using R = double; using V4 = v4f64; static constexpr bool CPLX = false; static constexpr int KK = 16; };
template <> struct ILT<c64> { using R = float; using V4 = v4f32; static constexpr bool CPLX = true; static constexpr int KK = 8; };
template <> struct ILT<float> { using R = float; using V4 = v4f32; static constexpr bool CPLX = false; static constexpr int KK = 16; };
__device__ __forceinline__ double il_re(const c128 &a) { return a.re; }
__device__ __forceinline__ double il_im(const c128 &a) { return a.im; }
__device__ __forceinline__ float il_re(const c64 &a) { return a.re; }
__device__ __forceinline__ float il_im(const c64 &a) { return a.im; }
__device__ __forceinline__ double il_re(double a) { return a; }
__device__ __forceinline__ double il_im(double) { return 0.0; }
__device__ __forceinline__ float il_re(float a) { return a; }
__device__ __forceinline__ float il_im(float) { return 0.f; }

// CS = components stored per vector index: 16, or 8 for real types with at most 8 right-hand sides (half a tile of the
// MFMA stays empty -- lanes ln >= 8 carry a zero x operand and deliver nothing -- but a vector index is 64 bytes of Xr and
// of W instead of 128: what bounds this pass over short panels is its vector-side traffic, not the matrix pipe)
// DEEP (instances for tall panels, MRMAX = 4): the tiles of ALL NRB row blocks of the next 16 columns are requested
// while the current ones are consumed (each buffer re-requested in place right behind its last use) instead of one step
// ahead -- with one 16 x 16 tile per wave in flight a pass over 64-row panels is bound by tile latency x resident waves
// (fp64: 12 waves per CU x 2 KB per 1.8 us = 3.5 TB/s, matrix pipe half idle).
template <typename T, int NRB, bool FWD, bool TRN, int CS, bool DEEP>
__device__ __forceinline__ void il_panel(const WaveD &wd, const uint4 *__restrict__ values, const int *__restrict__ rows,
                                         const int *__restrict__ cols, const typename ILT<T>::R *__restrict__ xr,
                                         typename ILT<T>::R *__restrict__ wacc, int flags, int lane, T *tile, int *cix) {
    using R = typename ILT<T>::R;
    using V4 = typename ILT<T>::V4;
    constexpr bool CPLX = ILT<T>::CPLX;
    constexpr bool F64MAP = sizeof(R) == 8;  // accumulator rows: lk + 4 r (f64) / 4 lk + r (f32)
    constexpr int E = TT<T>::E;
    constexpr int NLD = 4 / E;
    const bool opT = (flags & FLAG_OPT) != 0;
    const bool cjf = (flags & FLAG_CONJ) != 0;
    const int m = wd.m;
    const int ln = lane & 15, lk = lane >> 4;
    const int lc = CS == 16 ? ln : min(ln, CS - 1);  // the component this lane addresses
    const bool live = CS == 16 || ln < CS;           // ... and whether it carries one at all
    const PieceD pc = wd.first;
    const int xbase = pc.xbase, col_off = pc.col_off, ncols = pc.ncols, nstrips = pc.nstrips, kinds = pc.kind;
    const bool has_off = (kinds & kKindHasOff) != 0;
    const bool fwd_en = FWD && (!opT || has_off);
    const bool trn_en = TRN && (opT || has_off);
    const Vec16<T> *__restrict__ vb = reinterpret_cast<const Vec16<T> *>(values + (((uint64_t)pc.val_hi << 32) | pc.val_lo));
    const int s1w = wd.seg1_w, s1x = wd.seg1_x - wd.seg1_w;
    const int s2w = wd.seg2_w, s2x = pc.seg2_x - wd.seg2_w;
    // complex: the sign of X'' = i X (conj(B): -i X) on this lane: component 2 k takes -Im, component 2 k + 1 takes +Re
    const bool neg2 = ((ln & 1) == 0) != cjf;
    auto second = [&](R v1) {
        const R v2 = dppx<DPP_QUAD_XOR1>(v1);
        return neg2 ? -v2 : v2;
    };
    auto mfma = [&](R a, R b, V4 c) { return mfma16(a, b, c); };
    auto accrow = [&](int r) { return F64MAP ? lk + 4 * r : 4 * lk + r; };  // accumulator register r of this lane -> row of D
    // matrix operand of step (t0, rb): lane = (row rb * 16 + ln, strip t0 / E + 4 j + lk), indices clamped into the panel
    // (the last strip of a panel is zero-padded to E columns)
    auto mat = [&](int t0, int rb, int j) -> Vec16<T> {
        const int sidx = min(t0 / E + 4 * j + lk, nstrips - 1);
        const int row = min(rb * 16 + ln, m - 1);
        if (BSM_DBG(DBG_NO_MATRIX)) return Vec16<T>{};
        return load_stream16(&vb[(uint32_t)(sidx * m + row)]);
    };
    // ---- first batch of requests: the column list of the first block, the row list, the first tile -- all need the
    // descriptor only
    int craw[kIlCols / 64];
    const int nq0 = (min(ncols, kIlCols) + 63) >> 6;  // (wave-uniform)
#pragma unroll
    for (int q = 0; q < kIlCols / 64; ++q) {
        craw[q] = 0;
        if (xbase < 0 && q < nq0) craw[q] = cols[col_off + min(q * 64 + lane, ncols - 1)];
    }
    // rows 4 q + lk (the k index of the transposed half's MFMA q) and, where the accumulator map differs, the rows the
    // forward sums of this lane belong to
    int ri[4 * NRB], ro[F64MAP ? 1 : 4 * NRB];
#pragma unroll
    for (int q = 0; q < 4 * NRB; ++q) {
        ri[q] = wd.rbase + min(4 * q + lk, m - 1);
        if (!F64MAP) ro[q] = wd.rbase + min((q >> 2) * 16 + accrow(q & 3), m - 1);
    }
    if (wd.rbase < 0) {  // (wave-uniform)
#pragma unroll
        for (int q = 0; q < 4 * NRB; ++q) {
            ri[q] = rows[wd.row_off + min(4 * q + lk, m - 1)];
            if (!F64MAP) ro[q] = rows[wd.row_off + min((q >> 2) * 16 + accrow(q & 3), m - 1)];
        }
    }
    constexpr int NBUF = DEEP ? NRB : 1;
    Vec16<T> nb[NBUF][NLD];
#pragma unroll
    for (int rb = 0; rb < NBUF; ++rb)
#pragma unroll
        for (int j = 0; j < NLD; ++j) nb[rb][j] = mat(0, rb, j);
    // x / y index of panel column w with its roles (IL_NOFWD / IL_NOTRN)
    auto entry = [&](int w, int raw) -> int {
        bool off;
        int xi;
        if (xbase < 0) {
            off = raw >= 0 && (kinds & 3) == KIND_OFF;
            xi = raw & 0x7fffffff;
        } else {
            const int sh = w < s1w ? 0 : (w < s2w ? 2 : 4);
            off = ((kinds >> sh) & 3) == KIND_OFF;
            xi = w + (w < s1w ? xbase : (w < s2w ? s1x : s2x));
        }
        return xi | ((!opT || off) ? 0 : IL_NOFWD) | ((opT || off) ? 0 : IL_NOTRN);
    };
    // ---- second batch: the x rows of the panel (operand of the transposed half), one line of Xr per row
    R rr[4 * NRB];
#pragma unroll
    for (int q = 0; q < 4 * NRB; ++q) rr[q] = (TRN && !BSM_DBG(DBG_NO_XGATHER)) ? xr[(size_t)ri[q] * CS + lc] : R(0);
    V4 facc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) facc[rb] = V4{0, 0, 0, 0};
    R pd[4] = {0, 0, 0, 0};
    int pe[4] = {0, 0, 0, 0};
    bool pok[4] = {false, false, false, false};
    auto emit = [&]() {
        if (BSM_DBG(DBG_NO_GLOBAL_ATOMICS)) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) il_atomic_add(&wacc[(size_t)pe[r] * CS + lc], pd[r], pok[r]);
    };
    for (int cb = 0; cb < ncols; cb += kIlCols) {
        const int c_end = min(ncols, cb + kIlCols);
        // (re)fill the staged index list of [cb, c_end)
        const int nq = (c_end - cb + 63) >> 6;
        if (cb > 0) {
#pragma unroll
            for (int q = 0; q < kIlCols / 64; ++q)
                if (xbase < 0 && q < nq) craw[q] = cols[col_off + min(cb + q * 64 + lane, ncols - 1)];
        }
#pragma unroll
        for (int q = 0; q < kIlCols / 64; ++q)
            if (q < nq) cix[q * 64 + lane] = entry(min(cb + q * 64 + lane, ncols - 1), craw[q]);
        // entry of column w of this block (clamped into it)
        auto ent = [&](int w) { return cix[min(w, c_end - 1) - cb]; };
        // x operand of load j, column e of its strip: lane (component ln, column t0 + E (4 j + lk) + e)
        auto xop = [&](int t0, int j, int e) -> R {
            if (!FWD || BSM_DBG(DBG_NO_XGATHER)) return R(0);
            return xr[(size_t)(ent(t0 + E * (4 * j + lk) + e) & IL_MASK) * CS + lc];
        };
        if (cb == 0) {
#ifdef BSM_TRACE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            BSM_TSTAMP(2);  // lists, x rows and the first tile are there
#endif
#pragma unroll
            for (int q = 0; q < 4 * NRB; ++q) rr[q] = (trn_en && live && 4 * q + lk < m) ? rr[q] : R(0);
        }
        R xn[4];
#pragma unroll
        for (int j = 0; j < NLD; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) xn[j * E + e] = xop(cb, j, e);
#ifdef BSM_TRACE
        if (cb == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BSM_TSTAMP(3);  // first x operands arrived
        }
#endif
        for (int t0 = cb; t0 < c_end; t0 += 16) {
            // this tile's x operand (requested one tile ago), masked: columns beyond the block / without a forward role
            R xq[4];
#pragma unroll
            for (int je = 0; je < 4; ++je) {
                const int w = t0 + E * (4 * (je / E) + lk) + (je % E);
                const int en = ent(w);  // (read unconditionally: a short-circuit around an LDS read is a branch)
                const bool ok = fwd_en & live & (w < c_end) & ((en & IL_NOFWD) == 0);
                xq[je] = ok ? xn[je] : R(0);
            }
            // the PREVIOUS tile's sums first (vector-memory operations retire in order: they have the whole step, and the
            // latency of the requests behind them, to complete), then the next tile's operands
            emit();
            // (DEEP: everything this tile needs was requested a tile ago; hipcc does not count the atomics above and
            // drains what is in flight at the first use behind them -- so the new requests go out behind that use)
            if (!DEEP) {
#pragma unroll
                for (int j = 0; j < NLD; ++j)
#pragma unroll
                    for (int e = 0; e < E; ++e) xn[j * E + e] = xop(t0 + 16, j, e);
            }
            V4 dt = {0, 0, 0, 0};
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                Vec16<T> b[NLD];
#pragma unroll
                for (int j = 0; j < NLD; ++j) b[j] = nb[DEEP ? rb : 0][j];
                // the next step's tile: the next row block of these columns, or the first one of the next 16 columns
                if (!DEEP) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j) nb[0][j] = (rb + 1 < NRB) ? mat(t0, rb + 1, j) : mat(t0 + 16, 0, j);
                }
                if (FWD && !BSM_DBG(DBG_NO_FWD_HALF)) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j)
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const R x1 = xq[j * E + e];
                            facc[rb] = mfma(il_re(b[j].v[e]), x1, facc[rb]);
                            if (CPLX) facc[rb] = mfma(il_im(b[j].v[e]), second(x1), facc[rb]);
                        }
                }
                if (TRN && !BSM_DBG(DBG_NO_TRN_HALF)) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j)
#pragma unroll
                        for (int e = 0; e < E; ++e) tile[(E * (4 * j + lk) + e) * 17 + ln] = b[j].v[e];
                }
                if (DEEP) {  // this row block's tile of the next 16 columns, into the registers just consumed
                    if (rb == 0) {
#pragma unroll
                        for (int j = 0; j < NLD; ++j)
#pragma unroll
                            for (int e = 0; e < E; ++e) xn[j * E + e] = xop(t0 + 16, j, e);
                    }
#pragma unroll
                    for (int j = 0; j < NLD; ++j) nb[rb][j] = mat(t0 + 16, rb, j);
                }
                if (TRN && !BSM_DBG(DBG_NO_TRN_HALF)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const T u = tile[ln * 17 + 4 * q + lk];
                        const R r1 = rr[rb * 4 + q];
                        dt = mfma(il_re(u), r1, dt);
                        if (CPLX) dt = mfma(il_im(u), second(r1), dt);
                    }
                }
            }
            // lane (component ln, lk), register r: the sums of column t0 + accrow(r): parked until the next step
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int w = t0 + accrow(r);
                const int en = ent(w);
                pe[r] = en & IL_MASK;
                pd[r] = dt[r];
                pok[r] = trn_en & live & (w < c_end) & ((en & IL_NOTRN) == 0);
            }
        }
    }
#ifdef BSM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSM_TSTAMP(4);  // every tile done
#endif
    emit();
    if (fwd_en && !BSM_DBG(DBG_NO_FWD_OUT)) {
        // lane (component ln, lk), register r of row block rb: row rb * 16 + accrow(r) -- every wave adds its own partial sums
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rb * 16 + accrow(r);
                const int yi = F64MAP ? ri[4 * rb + r] : ro[4 * rb + r];  // (f64 map: row = 4 q + lk for q = 4 rb + r)
                il_atomic_add(&wacc[(size_t)yi * CS + lc], facc[rb][r], live & (row < m));
            }
    }
}

// (one wave per workgroup -- the waves of this pass share nothing, and a workgroup's slot is only recycled when its
// SLOWEST wave is done: tools/il_trace.py showed 66 % of the wave slots occupied -- was measured at +-0 and removed)
template <typename T, int MRMAX, bool FWD, bool TRN, int CS>
__global__ void __launch_bounds__(64 * kWavesPerWg, (il_wgs<T, MRMAX>()))
    panel_kernel_il(const WaveWork *__restrict__ waves, const uint4 *__restrict__ values, const int *__restrict__ rows,
                    const int *__restrict__ cols, const typename ILT<T>::R *__restrict__ xr, typename ILT<T>::R *__restrict__ wacc,
                    int flags, unsigned wg_base, unsigned xcd_run) {
    constexpr int WPW = kWavesPerWg;
    constexpr bool DEEP = MRMAX > 2;  // tall panels: all row blocks of the next 16 columns in flight (il_panel)
    __shared__ T tl[WPW][TRN ? 16 * 17 : 1];
    __shared__ int cixs[WPW][kIlCols];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    BSM_TSTAMP(0);  // wave started
    // XCD-aware order (xcd_run > 0): workgroups are dealt to the 8 XCDs round robin, so with the plain order eight
    // NEIGHBOURING panels -- which read mostly the same lines of Xr -- land in eight different L2s.  Here every XCD takes
    // RUNS of xcd_run consecutive workgroups of the list (8 * xcd_run workgroups = one run per XCD), so neighbours share
    // an L2 while the list is still consumed front to back on all XCDs (its heavy items come first: a contiguous eighth
    // per XCD, the first form of this, left XCD 0 with all of them -- C5 slice x 8 967 -> 1252 us).
    unsigned bid = blockIdx.x;
    if (xcd_run) {  // (the launcher pads the grid to a multiple of 8 * xcd_run; surplus blocks leave at once)
        const unsigned span = 8u * xcd_run, in = bid % span;
        bid = bid - in + (in & 7u) * xcd_run + (in >> 3);
    }
    if (bid >= wg_base) return;  // wg_base: number of workgroups of the record list (re-used argument)
    const WaveD wd = load_wave(waves + ((size_t)bid * WPW + wave));
    if (wd.work != WORK_PANEL || wd.npieces <= 0 || wd.first.ncols <= 0 || wd.m <= 0) return;
#ifdef BSM_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BSM_TSTAMP(1);  // descriptor arrived
    if (lane == 0) {
        t_trace[threadIdx.x >> 6][6] = (unsigned long long)((long long)wd.first.ncols * 65536 + wd.m);
        t_trace[threadIdx.x >> 6][7] = wall_clock64();
    }
#endif
    const int nrb = (wd.m + 15) >> 4;  // (wave-uniform)
    if (nrb == 1)
        il_panel<T, 1, FWD, TRN, CS, DEEP>(wd, values, rows, cols, xr, wacc, flags, lane, tl[wave], cixs[wave]);
    else if (nrb == 2 || MRMAX <= 2)
        il_panel<T, 2, FWD, TRN, CS, DEEP>(wd, values, rows, cols, xr, wacc, flags, lane, tl[wave], cixs[wave]);
    else if (nrb == 3)
        il_panel<T, (MRMAX > 2 ? 3 : 2), FWD, TRN, CS, DEEP>(wd, values, rows, cols, xr, wacc, flags, lane, tl[wave], cixs[wave]);
    else
        il_panel<T, (MRMAX > 2 ? 4 : 2), FWD, TRN, CS, DEEP>(wd, values, rows, cols, xr, wacc, flags, lane, tl[wave], cixs[wave]);
#ifdef BSM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSM_TSTAMP(5);  // everything stored
    if (lane == 0) t_trace[threadIdx.x >> 6][8] = wall_clock64();
    if (g_trace && lane < 16)
        g_trace[((size_t)blockIdx.x * WPW + (threadIdx.x >> 6)) * 16 + lane] = t_trace[threadIdx.x >> 6][lane];
#endif
}

// Xr[i][k] = alpha * X[i + kc(k) * ldx], k < KK (k >= kact: the last active column again, as the padded passes read it):
// 256 rows per workgroup, read down the columns of X, written along the lines of Xr (LDS transposition, row stride KK + 1)
template <typename T, int KK>
__global__ void __launch_bounds__(256) il_pack_kernel(const T *__restrict__ x, long long ldx, long long n, T alpha, int kact,
                                                      T *__restrict__ xr) {
    __shared__ T s[256 * (KK + 1)];
    const long long r0 = (long long)blockIdx.x * 256;
    const int t = threadIdx.x;
    if (r0 + t < n) {
#pragma unroll
        for (int k = 0; k < KK; ++k) s[t * (KK + 1) + k] = mul(alpha, x[r0 + t + (long long)(k < kact ? k : kact - 1) * ldx]);
    }
    __syncthreads();
    const long long cnt = (n - r0 < 256 ? n - r0 : 256) * KK;
#pragma unroll
    for (int j = 0; j < KK; ++j) {
        const int idx = j * 256 + t;
        if (idx < cnt) xr[r0 * KK + idx] = s[(idx / KK) * (KK + 1) + (idx % KK)];
    }
}
// Y[i + k * ldy] = (strong zero ? 0 : beta * Y) + W[i][k] for i in [lo, hi), k < kact;  W[i][:] = 0 behind the read
template <typename T, int KK>
__global__ void __launch_bounds__(256) il_finish_kernel(T *__restrict__ y, long long ldy, long long lo, long long hi, T beta,
                                                        int strong_zero, int kact, T *__restrict__ wacc) {
    __shared__ T s[256 * (KK + 1)];
    const long long r0 = lo + (long long)blockIdx.x * 256;
    const int t = threadIdx.x;
    const long long cnt = (hi - r0 < 256 ? hi - r0 : 256) * KK;
#pragma unroll
    for (int j = 0; j < KK; ++j) {
        const int idx = j * 256 + t;
        if (idx < cnt) {
            s[(idx / KK) * (KK + 1) + (idx % KK)] = wacc[r0 * KK + idx];
            wacc[r0 * KK + idx] = zero_of(T{});
        }
    }
    __syncthreads();
    if (r0 + t < hi) {
        for (int k = 0; k < kact; ++k) {
            T *yp = &y[r0 + t + (long long)k * ldy];
            const T v = s[t * (KK + 1) + k];
            *yp = strong_zero ? v : madd(v, beta, *yp);
        }
    }
}

// y[lo .. hi) = beta * y  (or 0 for the strong zero) -- `y .*= beta`,
// reference src/blockmatrix.jl:231, src/symmetricblockmatrix.jl:392, src/vbcrs.jl:273,313.
// A streaming pass: one 16-byte unit per lane and step (the element-per-thread form took 5.6 us for the 1.6 MB
// of a C3-sized y under rocprofv3 -- 0.3 TB/s -- and is a launch of its own in front of every accumulate-mode
// product), the unaligned head and tail of the range element by element.
template <typename T>
__global__ void __launch_bounds__(256) scale_kernel(T *__restrict__ y, long long ldy, long long lo,
                                                    long long hi, T beta, int strong_zero) {
    constexpr int E = TT<T>::E;
    T *__restrict__ yc = y + (long long)blockIdx.y * ldy + lo;  // one grid row per right-hand side
    const long long n = hi - lo;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // elements in front of the first 16-byte boundary
    const unsigned gap = (unsigned)((16 - ((uintptr_t)yc & 15)) & 15);
    if (gap % sizeof(T)) {  // (a complex vector on an 8-byte boundary only: no element count reaches a 16-byte one)
        for (long long i = tid; i < n; i += stride) yc[i] = strong_zero ? zero_of(T{}) : mul(beta, yc[i]);
        return;
    }
    long long head = (long long)(gap / sizeof(T));
    if (head > n) head = n;
    const long long nvec = (n - head) / E;
    Vec16<T> *__restrict__ yv = reinterpret_cast<Vec16<T> *>(yc + head);
    for (long long u = tid; u < nvec; u += stride) {
        Vec16<T> v;
        if (strong_zero) {
#pragma unroll
            for (int e = 0; e < E; ++e) v.v[e] = zero_of(T{});
        } else {
            v = yv[u];
#pragma unroll
            for (int e = 0; e < E; ++e) v.v[e] = mul(beta, v.v[e]);
        }
        yv[u] = v;
    }
    const long long tail0 = head + nvec * E;
    if (tid < head) yc[tid] = strong_zero ? zero_of(T{}) : mul(beta, yc[tid]);
    if (tid < n - tail0) yc[tail0 + tid] = strong_zero ? zero_of(T{}) : mul(beta, yc[tail0 + tid]);
}
// grid of the pass over n elements: one 16-byte unit per thread, at most 2048 workgroups
template <typename T> static unsigned scale_blocks(long long n) {
    long long nblk = (n / TT<T>::E + 255) / 256 + 1;
    return (unsigned)(nblk > 2048 ? 2048 : nblk);
}

// second launch of the gather mode: y[j] = beta*y[j] + alpha * (sum of the workspace slots that
// contribute to j, in their fixed ascending order).  Outside the owned range only rows that
// receive contributions are touched (and not scaled), like the atomic path.
template <typename T>
__global__ void __launch_bounds__(256)
    gather_kernel(T *__restrict__ y, long long ylen, long long own_lo, long long own_hi,
                  const long long *__restrict__ ptr, const int *__restrict__ idx, const T *__restrict__ ws,
                  T alpha, T beta, int strong_zero) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ylen) return;
    // ELL per 64-row tile: line l holds the l-th contribution of each row of the tile (-1: none)
    const long long t = j >> 6;
    const int lane = (int)(j & 63);
    const long long a = ptr[t], b = ptr[t + 1];
    T s = zero_of(T{});
    bool any = false;
    for (long long l = a; l < b; ++l) {
        const int slot = idx[l * 64 + lane];
        if (slot >= 0) {
            s = add(s, ws[slot]);
            any = true;
        }
    }
    const T val = mul(alpha, s);
    if (j >= own_lo && j < own_hi)
        y[j] = strong_zero ? val : madd(val, beta, y[j]);
    else if (any)
        y[j] = add(y[j], val);
}

// ----------------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------------
template <typename T> static T make_scalar(double v);
template <> float make_scalar<float>(double v) { return (float)v; }
template <> double make_scalar<double>(double v) { return v; }
template <> c64 make_scalar<c64>(double v) { return c64{(float)v, 0.f}; }
template <> c128 make_scalar<c128>(double v) { return c128{v, 0.0}; }

template <typename T> static T load_scalar(const void *p, double dflt) {
    return p ? *reinterpret_cast<const T *>(p) : make_scalar<T>(dflt);
}

template <typename T> static bool is_one(T v);
template <> bool is_one(float v) { return v == 1.f; }
template <> bool is_one(double v) { return v == 1.0; }
template <> bool is_one(c64 v) { return v.re == 1.f && v.im == 0.f; }
template <> bool is_one(c128 v) { return v.re == 1.0 && v.im == 0.0; }

// Non-temporal matrix loads: every matrix byte is used once per launch, and allocating it in the
// L2 / Infinity Cache like ordinary data costs bandwidth (a bare streaming read of a C2-sized operator
// runs 6.65 us with the hint and 8.45 us without; operators larger than the 256 MiB Infinity Cache gain
// 5-12 %, and a launch that finds the caches full of someone else's dirty lines 40 %).  The exception
// are operators that just fit the Infinity Cache: streamed with the hint they are not retained as
// well between launches (136-298 MB: 2-9 % slower), so they keep ordinary loads; so do tiny ones.
static bool stream_policy(const DeviceImage &img) {
    static const int forced = [] {
        const char *v = std::getenv("BSM_NT");
        return v ? std::atoi(v) : -1;
    }();
    if (forced >= 0) return forced != 0;
    // (operators of a few tens of MB are a single round of resident workgroups bound by one
    // workgroup's dependency chain, where the hint costs ~5 %: 27 MB 5.8 vs 6.3 us)
    const long long mb = img.value_bytes >> 20;
    // One rule for exclusive AND accumulate-mode launches.  (Round 2 kept the hint for every fused operator from 40 MB
    // on: "a 242 MB fused symmetric product runs the same warm either way".  The tiled BEM fixture does not --
    // profiles/r04_nt_sweep.txt, hint / plain in us: fp32 109 MB 30.9 / 29.0, 163 MB 44.2 / 39.8, 218 MB 61.4 / 53.5;
    // fp64 203 MB 50.1 / 42.7, 254 MB 60.7 / 51.1, 305 MB 71.4 / 68.4, 407 MB 85.7 / 85.9; ComplexF64 98 MB 24.1 / 25.8,
    // 196 MB 45.9 / 43.9, 392 MB 81.8 / 85.0, 783 MB 145 / 151: between ~100 and ~320 MB of values the operator stays in
    // the Infinity Cache between two products only when it is loaded like ordinary data.)
    return (mb >= 40 && mb < 100) || mb > 320;  // measured crossovers: ~105 MB and ~310-330 MB of values
}

// one launch of panel_kernel<T, L, FWD, TRN, NT> with NT taken from the run-time policy `nt`
#define BSM_LAUNCH_PANEL(FWD_, TRN_, ...)                                                \
    do {                                                                                  \
        if (nt)                                                                           \
            hipLaunchKernelGGL((panel_kernel<T, L, FWD_, TRN_, true>), __VA_ARGS__);      \
        else                                                                              \
            hipLaunchKernelGGL((panel_kernel<T, L, FWD_, TRN_, false>), __VA_ARGS__);     \
    } while (0)

template <typename T, int L>
static hipError_t launch_typed(const DeviceImage &img, bool opT, bool conj, const void *x, void *y,
                               const void *alpha_p, const void *beta_p, int strong_zero,
                               hipStream_t stream, bool use_gather = false, const long long *zrange = nullptr) {
    const T alpha = load_scalar<T>(alpha_p, 1.0);
    const T beta = load_scalar<T>(beta_p, 0.0);
    int flags = 0;
    if (strong_zero) flags |= FLAG_STRONG_ZERO;
    if (conj) flags |= FLAG_CONJ;
    if (opT) flags |= FLAG_OPT;
#ifdef BSM_EXPERIMENT
    if (const char *v = std::getenv("BSM_DEBUG_FLAGS")) flags |= std::atoi(v) << 16;
#endif
    const WaveWork *waves = (const WaveWork *)img.d_waves;
    const uint4 *values = (const uint4 *)img.d_values;
    const int *rows = (const int *)img.d_rows;
    const int *cols = (const int *)img.d_cols;
    const T *xd = (const T *)x;
    T *yd = (T *)y;
    const dim3 block(64 * kWavesPerWg);
    const bool nt = stream_policy(img);

    if (!opT && img.exclusive_fwd) {
        // one launch: every y row has exactly one producer; beta is fused into its store and
        // the rows no block covers are scaled by WORK_SCALE waves of the same grid.
        flags |= FLAG_DIRECT;
        if (img.nwg_total > 0)
            BSM_LAUNCH_PANEL(true, false, dim3((unsigned)img.nwg_total), block, 0,
                               stream, waves, values, rows, cols, xd, yd, alpha, beta, flags, 0u, (T *)nullptr, 0LL);
        return hipGetLastError();
    }
    // accumulate mode: y .*= beta over the owned range, then hardware atomics
    const long long ylen = opT ? img.ncols : img.nrows;
    long long lo = 0, hi = ylen;
    if (!opT) {
        lo = img.own_lo;
        hi = img.own_hi;
    }
    if (zrange) {  // the caller (multi-device fan-out) knows which y entries this image must define
        lo = zrange[0];
        hi = zrange[1];
    }
    const bool gather = use_gather && img.d_ws != nullptr;
    T *ws = gather ? (T *)img.d_ws : (T *)nullptr;
    if (gather) flags |= FLAG_GATHER;
    if (!gather && hi > lo && (strong_zero || !is_one(beta)))
        hipLaunchKernelGGL((scale_kernel<T>), dim3(scale_blocks<T>(hi - lo)), dim3(256), 0, stream, yd, 0LL, lo, hi, beta,
                           strong_zero);
    // one launch over every workgroup (atomics), or one launch per colour class (plain RMW:
    // the classes touch pairwise disjoint y entries, so the result is bitwise reproducible)
    const bool colored = !img.color_wg_ptr.empty();
    if (colored) flags |= FLAG_RMW;
    const size_t nlaunch = colored ? img.color_wg_ptr.size() - 1 : 1;
    for (size_t c = 0; c < nlaunch; ++c) {
        const long long wg0 = colored ? img.color_wg_ptr[c] : 0;
        const long long wg1 = colored ? img.color_wg_ptr[c + 1] : img.nwg_main;
        if (wg1 <= wg0) continue;
        const dim3 grid((unsigned)(wg1 - wg0));
        const unsigned wg_base = (unsigned)wg0;
        if (!opT) {
            if (img.has_off)
                BSM_LAUNCH_PANEL(true, true, grid, block, 0, stream, waves,
                                   values, rows, cols, xd, yd, alpha, beta, flags, wg_base, ws, img.ws_fbase);
            else
                BSM_LAUNCH_PANEL(true, false, grid, block, 0, stream, waves,
                                   values, rows, cols, xd, yd, alpha, beta, flags, wg_base, ws, img.ws_fbase);
        } else {
            if (img.has_off)
                BSM_LAUNCH_PANEL(true, true, grid, block, 0, stream, waves,
                                   values, rows, cols, xd, yd, alpha, beta, flags, wg_base, ws, img.ws_fbase);
            else
                BSM_LAUNCH_PANEL(false, true, grid, block, 0, stream, waves,
                                   values, rows, cols, xd, yd, alpha, beta, flags, wg_base, ws, img.ws_fbase);
        }
    }
    if (gather && ylen > 0) {
        const int k = opT ? 1 : 0;
        const long long nblk = (ylen + 255) / 256;
        hipLaunchKernelGGL((gather_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, stream, yd, ylen, lo, hi,
                           (const long long *)img.d_inv_ptr[k], (const int *)img.d_inv_idx[k], (const T *)ws,
                           alpha, beta, strong_zero);
    }
    return hipGetLastError();
}

// K right-hand sides per pass
template <typename T, int L, int K>
static hipError_t launch_typed_multi(const DeviceImage &img, bool opT, bool conj, const T *xd, long long ldx,
                                     T *yd, long long ldy, T alpha, T beta, int strong_zero,
                                     hipStream_t stream, const long long *zrange, int kact = K) {
    int flags = 0;
    if (strong_zero) flags |= FLAG_STRONG_ZERO;
    if (conj) flags |= FLAG_CONJ;
    if (opT) flags |= FLAG_OPT;
    if (kact < K) flags |= kact << FLAG_KACT_SHIFT;  // a padded batch: kact of the K slots carry columns
#ifdef BSM_EXPERIMENT
    if (const char *v = std::getenv("BSM_DEBUG_FLAGS")) flags |= std::atoi(v) << 16;
#endif
    const WaveWork *waves = (const WaveWork *)img.d_waves;
    const uint4 *values = (const uint4 *)img.d_values;
    const int *rows = (const int *)img.d_rows;
    const int *cols = (const int *)img.d_cols;
    const dim3 block(64 * kWavesPerWg);
    if (!opT && img.exclusive_fwd) {
        flags |= FLAG_DIRECT;
        if (img.nwg_total > 0)
            hipLaunchKernelGGL((panel_kernel_multi<T, L, true, false, K>), dim3((unsigned)img.nwg_total), block,
                               0, stream, waves, values, rows, cols, xd, ldx, yd, ldy, alpha, beta, flags, 0u);
        return hipGetLastError();
    }
    const long long ylen = opT ? img.ncols : img.nrows;
    long long lo = 0, hi = ylen;
    if (!opT) {
        lo = img.own_lo;
        hi = img.own_hi;
    }
    if (zrange) {  // multi-device fan-out: see launch_typed
        lo = zrange[0];
        hi = zrange[1];
    }
    if (hi > lo && (strong_zero || !is_one(beta)))
        hipLaunchKernelGGL((scale_kernel<T>), dim3(scale_blocks<T>(hi - lo), (unsigned)kact), dim3(256), 0, stream, yd, ldy, lo,
                           hi, beta, strong_zero);
    const bool colored = !img.color_wg_ptr.empty();
    if (colored) flags |= FLAG_RMW;
    // the coarser split of the panels (bsm_analysis.h: Tunables::multi_wave_bytes), where the image has one
    long long nwg_main = img.nwg_main;
    if (img.d_waves_multi && !colored) {
        waves = (const WaveWork *)img.d_waves_multi;
        nwg_main = img.nwg_multi;
    }
    const size_t nlaunch = colored ? img.color_wg_ptr.size() - 1 : 1;
    for (size_t c = 0; c < nlaunch; ++c) {
        const long long wg0 = colored ? img.color_wg_ptr[c] : 0;
        const long long wg1 = colored ? img.color_wg_ptr[c + 1] : nwg_main;
        if (wg1 <= wg0) continue;
        const dim3 grid((unsigned)(wg1 - wg0));
        const unsigned wg_base = (unsigned)wg0;
        if (!opT && !img.has_off)
            hipLaunchKernelGGL((panel_kernel_multi<T, L, true, false, K>), grid, block, 0, stream, waves, values,
                               rows, cols, xd, ldx, yd, ldy, alpha, beta, flags, wg_base);
        else if (img.has_off)
            hipLaunchKernelGGL((panel_kernel_multi<T, L, true, true, K>), grid, block, 0, stream, waves, values,
                               rows, cols, xd, ldx, yd, ldy, alpha, beta, flags, wg_base);
        else
            hipLaunchKernelGGL((panel_kernel_multi<T, L, false, true, K>), grid, block, 0, stream, waves, values,
                               rows, cols, xd, ldx, yd, ldy, alpha, beta, flags, wg_base);
    }
    return hipGetLastError();
}

// ---- the interleaved pass (panel_kernel_il): policy and launch ---------------------------------------------
// BSM_MULTI_IL: 0 = never, 1 = automatic (default: images that accumulate with atomics -- short scattered panels, mean
// group height below 32, in every element type; ComplexF64 / ComplexF32 from BSM_MFMA_MIN_COLS columns on, real types
// from BSM_IL_REAL_MIN_COLS), 2 = every image that accumulates with atomics (A / B)
static int il_mode() {
    static const int v = [] {
        const char *e = std::getenv("BSM_MULTI_IL");
        return e ? std::atoi(e) : 1;
    }();
    return v;
}
static int mfma_min_cols() {
    static const int v = [] {
        const char *e = std::getenv("BSM_MFMA_MIN_COLS");
        return e ? std::atoi(e) : 3;  // (BEM fixture x 4: 362 us padded against 486 us through the 4-column kernel)
    }();
    return v;
}
static int il_real_min_cols() {
    static const int v = [] {
        const char *e = std::getenv("BSM_IL_REAL_MIN_COLS");
        // (8 components per index from 5 columns on: BEM fp64 x 8 318 -> 202 us, C3 x 8 256 -> 217 us; 4 columns stay on the
        // vector kernels: BEM 194 us, C3 195 us against ~ 200 / 217)
        return e ? std::atoi(e) : 5;
    }();
    return v;
}
bool il_applies(const DeviceImage &img, bool opT, long long nrhs) {
    const bool cplx = img.dtype >= 2;
    if (il_mode() == 0 || nrhs < (cplx ? mfma_min_cols() : il_real_min_cols())) return false;
    if (!opT && img.exclusive_fwd) return false;   // plain stores with beta fused: nothing to gain
    if (!img.color_wg_ptr.empty()) return false;   // coloured launches keep their bitwise reproducible read-modify-write
    if (std::max(img.nrows, img.ncols) >= (1ll << 30)) return false;  // (staged entries carry two role bits)
    // automatic: short scattered panels, and tall panels where the product is FUSED (symmetric operators: both halves, the
    // transposed one all atomics -- C3 x 16 377 -> 305 us, x 8 264 -> 220, C5 slice x 16 1607 -> 1131; the C3 structure with
    // complex entries, tools/c3_complex.py: ComplexF64 x 8 175 -> 138 us, ComplexF32 76 -> 63).  Forward-only products of
    // tall panels keep their kernels: on the C4 slice (128 x 128 fp32 blocks of one GPU of eight, vectors of the full 2 M
    // entries) the pass's two vector sweeps cost more than it saves (x 8 363 -> 457 us, x 16 453 -> 562).
    (void)cplx;
    return il_mode() == 2 || img.mean_rows < 32.f || img.has_off;
}
template <typename T, int KK>
static hipError_t launch_il(const DeviceImage &img, bool opT, bool conj, const T *xd, long long ldx, T *yd, long long ldy, T alpha,
                            T beta, int strong_zero, hipStream_t stream, int kact, ILWork &il, const long long *zrange) {
    using R = typename ILT<T>::R;
    constexpr int CS = ILT<T>::CPLX ? 2 * KK : KK;  // components per vector index (8 or 16)
    const long long xlen = opT ? img.nrows : img.ncols, ylen = opT ? img.ncols : img.nrows;
    if (xlen > il.rows || ylen > il.rows) return hipErrorInvalidValue;
    int flags = 0;
    if (opT) flags |= FLAG_OPT;
    if (conj) flags |= FLAG_CONJ;
#ifdef BSM_EXPERIMENT
    if (const char *v = std::getenv("BSM_DEBUG_FLAGS")) flags |= std::atoi(v) << 16;
#endif
    hipError_t e = hipSuccess;
    if (!il.w_clean) e = hipMemsetAsync(il.w, 0, (size_t)il.rows * 128, stream);
    il.w_clean = false;  // (until the finish pass has been enqueued)
    if (e != hipSuccess) return e;
    if (xlen > 0)
        hipLaunchKernelGGL((il_pack_kernel<T, KK>), dim3((unsigned)((xlen + 255) / 256)), dim3(256), 0, stream, xd, ldx, xlen, alpha,
                           kact, (T *)il.xr);
    const WaveWork *waves = (const WaveWork *)(img.d_waves_multi ? img.d_waves_multi : img.d_waves);
    const long long nwg = img.d_waves_multi ? img.nwg_multi : img.nwg_main;
    const uint4 *values = (const uint4 *)img.d_values;
    const int *rows = (const int *)img.d_rows, *cols = (const int *)img.d_cols;
    const R *xr = (const R *)il.xr;
    R *w = (R *)il.w;
    if (nwg > 0) {
        // BSM_IL_XCD = R: XCD-aware workgroup order, runs of R consecutive workgroups per XCD (0: plain order).  Default:
        // 16 for operators with tall panels (C3 x 16 312 -> 262 us -- neighbouring 64-row panels read the same 9 x 64
        // lines of Xr, L2 hits 0.5 M -> of 9.5 M read requests with the plain order; R = 4 ... 64 alike), plain for
        // short scattered panels (the tiled BEM fixture: +-0, 367 / 204 / 359 us against 374 / 208 / 352)
        static const int xcd_env = [] {
            const char *v = std::getenv("BSM_IL_XCD");
            return v ? std::atoi(v) : -1;
        }();
        const bool small = img.max_rows <= 32;
        const int xcd = xcd_env >= 0 ? xcd_env : (small ? 0 : 16);
        const unsigned nblk = (unsigned)nwg;
        const unsigned xcd_run = xcd > 0 ? (unsigned)xcd : 0u, span = 8u * xcd_run;
        const dim3 grid(xcd_run ? (nblk + span - 1) / span * span : nblk), block(64 * kWavesPerWg);
#define BSM_IL_LAUNCH(MR)                                                                                                              \
    do {                                                                                                                           \
        if (!opT && !img.has_off)                                                                                                  \
            hipLaunchKernelGGL((panel_kernel_il<T, MR, true, false, CS>), grid, block, 0, stream, waves, values, rows, cols, xr, w,  \
                               flags, nblk, xcd_run);                                                                                        \
        else if (img.has_off)                                                                                                      \
            hipLaunchKernelGGL((panel_kernel_il<T, MR, true, true, CS>), grid, block, 0, stream, waves, values, rows, cols, xr, w,   \
                               flags, nblk, xcd_run);                                                                                        \
        else                                                                                                                       \
            hipLaunchKernelGGL((panel_kernel_il<T, MR, false, true, CS>), grid, block, 0, stream, waves, values, rows, cols, xr, w,  \
                               flags, nblk, xcd_run);                                                                                        \
    } while (0)
        // (tall panels: all row blocks of the next 16 columns in flight -- il_panel's DEEP form, worth 1-6 % over one step
        // ahead with the XCD-aware order, profiles/r05_il_tall_panels.txt)
        if (small)
            BSM_IL_LAUNCH(2);
        else
            BSM_IL_LAUNCH(4);
#undef BSM_IL_LAUNCH
    }
    // Y = beta * Y + W over the rows this handle scales (all of them for op T / C), Y += W elsewhere; W = 0 behind
    long long lo = 0, hi = ylen;
    if (!opT) {
        lo = img.own_lo;
        hi = img.own_hi;
    }
    if (zrange) {  // multi-device fan-out: the y entries this part's `y .*= beta` covers (see launch_typed)
        lo = zrange[0];
        hi = zrange[1];
    }
    const T one = make_scalar<T>(1.0);
    auto finish = [&](long long a, long long b, T bt, int sz) {
        if (b > a)
            hipLaunchKernelGGL((il_finish_kernel<T, KK>), dim3((unsigned)((b - a + 255) / 256)), dim3(256), 0, stream, yd, ldy, a, b, bt,
                               sz, kact, (T *)il.w);
    };
    finish(0, lo, one, 0);
    finish(lo, hi, beta, strong_zero);
    finish(hi, ylen, one, 0);
    e = hipGetLastError();
    if (e == hipSuccess) il.w_clean = true;
    return e;
}

template <typename T>
static hipError_t launch_multi_typed(const DeviceImage &img, bool opT, bool conj, long long nrhs,
                                     const void *x, long long ldx, void *y, long long ldy,
                                     const void *alpha_p, const void *beta_p, int strong_zero,
                                     hipStream_t stream, const long long *zrange, ILWork *il) {
    const T alpha = load_scalar<T>(alpha_p, 1.0);
    const T beta = load_scalar<T>(beta_p, 0.0);
    const T *xd = (const T *)x;
    T *yd = (T *)y;
    long long k = 0;
    hipError_t e = hipSuccess;
    // batches of 8, then 4, then single columns: A is streamed once per batch.  The 8-column
    // kernels keep L = 4 loads per lane in flight instead of 8: with 8 accumulators and 8 x values
    // per lane the registers, i.e. the resident waves, are worth more than the deeper load queue
    // (fp64 fused: 143 -> 103 VGPRs; C3 3.0x -> 3.3x, 8-28-row blocks 2.0x -> 2.4x over 8 products).
    // A remainder of 5-7 columns is one PADDED 8-column pass and 3 columns one padded 4-column pass (the idle
    // slots repeat the last column and are never written): a pass costs 1.2-1.8 (8) / 1.1-1.4 (4) single products
    // on the large operators, 3.9 / 3.0 on the BEM fixture -- never more than the 4 + singles it replaces; two
    // columns stay two single products (a 4-column pass over 3-28-row panels costs three).
    // real types, 9 columns and more: batches of 16 on the matrix pipe (N = 16 of v_mfma_*_16x16x4: kMfmaReal), a
    // remainder of 9-15 as one padded pass (C3 x 16: 374 us against 2 x 264, C4 slice 442 against 2 x 364; x 9: one
    // padded pass against an 8-column pass + a single product).  Short scattered panels (the BEM fixture: mean group
    // height below 32) gain nothing below 15 columns: their passes are bound by the x gather and the atomics, which
    // grow with the padded width (fp64 x 16: 615 us against 2 x 320).  BSM_MFMA_REAL_MIN_COLS overrides (17: off).
    // short scattered panels: the interleaved pass (above) -- complex types in batches of 8 columns, real types of 16,
    // then one padded remainder
    if (il && il_applies(img, opT, nrhs)) {
        constexpr int KK = ILT<T>::KK;
        const int least = ILT<T>::CPLX ? mfma_min_cols() : il_real_min_cols();
        while (e == hipSuccess && nrhs - k >= least) {
            const int kact = (int)std::min<long long>(KK, nrhs - k);
            if constexpr (!ILT<T>::CPLX) {
                if (kact <= 8)  // real types, at most 8 columns left: 8 components per index (64-byte lines)
                    e = launch_il<T, 8>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, kact, *il, zrange);
                else
                    e = launch_il<T, 16>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, kact, *il, zrange);
            } else {
                if (kact <= 4)  // complex types, at most 4 columns left: 8 components per index likewise
                    e = launch_il<T, 4>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, kact, *il, zrange);
                else
                    e = launch_il<T, 8>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, kact, *il, zrange);
            }
            k += kact;
        }
    }
    if constexpr (kMfmaReal<T, 16>) {
        static const int mr_env = [] {
            const char *v = std::getenv("BSM_MFMA_REAL_MIN_COLS");
            return v ? std::atoi(v) : 0;
        }();
        const int mr_min = mr_env ? mr_env : (img.mean_rows < 32.f ? 15 : 9);
        while (e == hipSuccess && nrhs - k >= 16 && mr_min <= 16) {
            e = launch_typed_multi<T, 4, 16>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, zrange);
            k += 16;
        }
        if (e == hipSuccess && nrhs - k >= mr_min && nrhs - k < 16) {
            const int rem = (int)(nrhs - k);
            e = launch_typed_multi<T, 4, 16>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta, strong_zero, stream, zrange, rem);
            k += rem;
        }
    }
    while (e == hipSuccess && nrhs - k >= 8) {
        e = launch_typed_multi<T, 4, 8>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta,
                                        strong_zero, stream, zrange);
        k += 8;
    }
    // (ComplexF64: the 8-column pass runs on the matrix pipe -- a padded pass beats the 4-column register kernel
    // from 3 columns on: BSM_MFMA_MIN_COLS)
    const int mf_min = mfma_min_cols();
    if (e == hipSuccess && nrhs - k >= (kMfmaAny<T, 8> ? mf_min : 5)) {
        const int rem = (int)(nrhs - k);
        e = launch_typed_multi<T, 4, 8>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta,
                                        strong_zero, stream, zrange, rem);
        k += rem;
    }
    // (two columns: a padded 4-column pass where the row groups fill their lanes -- 1.1-1.3 single products on C3 /
    // C4 -- and two single products over short panels, where the pass would cost 2.6)
    if (e == hipSuccess && (nrhs - k >= 3 || (nrhs - k == 2 && img.lane_fill >= 0.85f))) {
        const int rem = (int)(nrhs - k);  // 2, 3 or 4
        // (real arithmetic and a transposed half in the product: L = 4, i.e. the tile-pipelined kernels -- BEM fp64
        // x 4 249 -> 211 us, C3 / C5 +-0; forward-only launches and complex: 8 loads per lane on the register path,
        // C4 slice x 4 1.10 vs 1.16 single products with L = 4)
        const bool fwd_only = !opT && (img.exclusive_fwd || !img.has_off);
        if ((kRealType<T> || sizeof(T) == 16) && !fwd_only)  // (ComplexF64: L = 4 on the register path, BEM x 4 518 -> 490 us)
            e = launch_typed_multi<T, 4, 4>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta,
                                            strong_zero, stream, zrange, rem);
        else
            e = launch_typed_multi<T, 8, 4>(img, opT, conj, xd + k * ldx, ldx, yd + k * ldy, ldy, alpha, beta,
                                            strong_zero, stream, zrange, rem);
        k += rem;
    }
    for (; e == hipSuccess && k < nrhs; ++k)
        e = launch_typed<T, 8>(img, opT, conj, xd + k * ldx, yd + k * ldy, alpha_p, beta_p, strong_zero, stream, false, zrange);
    return e;
}

hipError_t launch_mul_multi(const DeviceImage &img, bool opT, bool conj, long long nrhs, const void *x,
                            long long ldx, void *y, long long ldy, const void *alpha, const void *beta,
                            int strong_zero, hipStream_t stream, const long long *zrange, ILWork *il) {
    switch (img.dtype) {
        case 0: return launch_multi_typed<float>(img, opT, conj, nrhs, x, ldx, y, ldy, alpha, beta, strong_zero, stream, zrange, il);
        case 1: return launch_multi_typed<double>(img, opT, conj, nrhs, x, ldx, y, ldy, alpha, beta, strong_zero, stream, zrange, il);
        case 2: return launch_multi_typed<c64>(img, opT, conj, nrhs, x, ldx, y, ldy, alpha, beta, strong_zero, stream, zrange, il);
        case 3: return launch_multi_typed<c128>(img, opT, conj, nrhs, x, ldx, y, ldy, alpha, beta, strong_zero, stream, zrange, il);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mul(const DeviceImage &img, bool opT, bool conj, const void *x, void *y,
                      const void *alpha, const void *beta, int strong_zero, hipStream_t stream,
                      bool use_gather, const long long *zrange) {
    switch (img.dtype) {
        case 0:
            // fp32 fused products of SHORT panels: 4 loads per lane (tiled BEM fixture 48.6 -> 46.5 us; 16-256-row
            // operators lose 3-5 % with it and keep 8: profiles/r04_fused_loads_per_lane.txt)
            if (BSM_F32_L != 8 && img.has_off && !img.exclusive_fwd && img.mean_rows < 32.f)
                return launch_typed<float, BSM_F32_L>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
            return launch_typed<float, 8>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
        case 1:
            if (BSM_F64_L != 8 && img.has_off && !img.exclusive_fwd)
                return launch_typed<double, BSM_F64_L>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
            return launch_typed<double, 8>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
        // complex64 fused products: 4 loads per lane in flight -- 61 VGPRs, 8 waves per SIMD; with 8 the fused instance
        // needs 93-95 (5 waves): tiled BEM fixture 105.9 -> 95.1 us (profiles/r04_c64_l4.txt)
        case 2:
            if (img.has_off && !img.exclusive_fwd)
                return launch_typed<c64, BSM_C64_L>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
            return launch_typed<c64, 8>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
        case 3:
            if (BSM_C128_L != 8 && img.has_off && !img.exclusive_fwd)
                return launch_typed<c128, BSM_C128_L>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
            return launch_typed<c128, 8>(img, opT, conj, x, y, alpha, beta, strong_zero, stream, use_gather, zrange);
    }
    return hipErrorInvalidValue;
}

// ---- vector helpers of the multi-device fan-out (bsm_dist.cpp) ---------------------------------
// dst[i] += src[i]: a halo segment received from a peer is added to the local result
template <typename T>
__global__ void __launch_bounds__(256) vec_add_kernel(T *__restrict__ dst, const T *__restrict__ src, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = add(dst[i], src[i]);
}
// y[i] = beta * y[i] + r[i]: the delivered segment meets the caller's y (numeric beta)
template <typename T>
__global__ void __launch_bounds__(256) vec_axpby_kernel(T *__restrict__ y, const T *__restrict__ r, long long n, T beta) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = madd(r[i], beta, y[i]);
}
// ---- fused fan-out kernels of multi-device handles (bsm_dist.cpp) ------------------------------------
// The devices of a context can read each other's memory over xGMI (peer access), so the vector traffic of
// a product needs no copy engine and no staging buffer: ONE launch per device gathers the x pieces the
// device's blocks read (from the caller's x or from the x parts of its peers), ONE launch adds the y
// segments the peers produced for its rows to its own and writes the result to the caller's y (beta fused).
// Every pointer is a "virtual base": element i of the global vector lives at base + i.
template <typename T>
__global__ void __launch_bounds__(256) vec_fetch_kernel(T *__restrict__ dst, long long ld_dst, VecPieces pc, long long ld_src) {
    const int c = blockIdx.y;
    const T *__restrict__ src = reinterpret_cast<const T *>(pc.base[c]) + (long long)blockIdx.z * (pc.strided[c] ? ld_src : 0);
    T *__restrict__ d = dst + (long long)blockIdx.z * ld_dst;
    const long long lo = pc.lo[c], hi = pc.hi[c];
    for (long long i = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (long long)gridDim.x * blockDim.x)
        d[i] = src[i];
}

// y[i] = (strong ? 0 : beta * y[i]) + w[i] + sum over the pieces that cover i;  accumulate_only: w[i] += ... (no y).
// rezero: zeros are written behind everything that is read (own work vector and the peers' segments, over xGMI where
// they are remote), so the work vectors are zero again when the launch is over -- the next product accumulates into
// them without a `w = 0` launch in front (bsm_dist.cpp: DistState::w_clean)
template <typename T>
__global__ void __launch_bounds__(256) vec_finish_kernel(T *__restrict__ y, long long ldy, T *__restrict__ w, long long ldw,
                                                         VecPieces pc, int npieces, long long lo, long long hi, T beta,
                                                         int strong_zero, int accumulate_only, int rezero) {
    const long long k = blockIdx.y;
    T *__restrict__ wk = w + k * ldw;
    for (long long i = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (long long)gridDim.x * blockDim.x) {
        T v = wk[i];
        for (int c = 0; c < npieces; ++c)
            if (i >= pc.lo[c] && i < pc.hi[c]) {
                T *pe = const_cast<T *>(reinterpret_cast<const T *>(pc.base[c])) + k * ldw + i;
                v = add(v, *pe);
                if (rezero) *pe = zero_of(T{});
            }
        if (accumulate_only) {
            wk[i] = v;
        } else {
            if (rezero) wk[i] = zero_of(T{});
            T *__restrict__ yk = y + k * ldy;
            yk[i] = strong_zero ? v : madd(v, beta, yk[i]);
        }
    }
}

template <typename T>
static hipError_t fetch_typed(void *dst, long long ld_dst, const VecPieces &pc, int npieces, long long ld_src, int K,
                              hipStream_t stream) {
    long long longest = 0;
    for (int c = 0; c < npieces; ++c) longest = longest > pc.hi[c] - pc.lo[c] ? longest : pc.hi[c] - pc.lo[c];
    if (npieces <= 0 || longest <= 0) return hipSuccess;
    long long nblk = (longest + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL((vec_fetch_kernel<T>), dim3((unsigned)nblk, (unsigned)npieces, (unsigned)K), dim3(256), 0, stream, (T *)dst,
                       ld_dst, pc, ld_src);
    return hipGetLastError();
}
hipError_t launch_vec_fetch(int dtype, void *dst, long long ld_dst, const VecPieces &pc, int npieces, long long ld_src, int K,
                            hipStream_t stream) {
    switch (dtype) {
        case 0: return fetch_typed<float>(dst, ld_dst, pc, npieces, ld_src, K, stream);
        case 1: return fetch_typed<double>(dst, ld_dst, pc, npieces, ld_src, K, stream);
        case 2: return fetch_typed<c64>(dst, ld_dst, pc, npieces, ld_src, K, stream);
        case 3: return fetch_typed<c128>(dst, ld_dst, pc, npieces, ld_src, K, stream);
    }
    return hipErrorInvalidValue;
}
template <typename T>
static hipError_t finish_typed(void *y, long long ldy, void *w, long long ldw, const VecPieces &pc, int npieces, long long lo,
                               long long hi, const void *beta_p, int strong_zero, int accumulate_only, int rezero, int K,
                               hipStream_t stream) {
    if (hi <= lo) return hipSuccess;
    long long nblk = (hi - lo + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL((vec_finish_kernel<T>), dim3((unsigned)nblk, (unsigned)K), dim3(256), 0, stream, (T *)y, ldy, (T *)w, ldw, pc,
                       npieces, lo, hi, load_scalar<T>(beta_p, 0.0), strong_zero, accumulate_only, rezero);
    return hipGetLastError();
}
hipError_t launch_vec_finish(int dtype, void *y, long long ldy, void *w, long long ldw, const VecPieces &pc, int npieces,
                             long long lo, long long hi, const void *beta, int strong_zero, int accumulate_only, int rezero, int K,
                             hipStream_t stream) {
    switch (dtype) {
        case 0: return finish_typed<float>(y, ldy, w, ldw, pc, npieces, lo, hi, beta, strong_zero, accumulate_only, rezero, K, stream);
        case 1: return finish_typed<double>(y, ldy, w, ldw, pc, npieces, lo, hi, beta, strong_zero, accumulate_only, rezero, K, stream);
        case 2: return finish_typed<c64>(y, ldy, w, ldw, pc, npieces, lo, hi, beta, strong_zero, accumulate_only, rezero, K, stream);
        case 3: return finish_typed<c128>(y, ldy, w, ldw, pc, npieces, lo, hi, beta, strong_zero, accumulate_only, rezero, K, stream);
    }
    return hipErrorInvalidValue;
}

// y[lo_c + i] += src_c[i], i < hi_c - lo_c, for up to kMaxVecPieces DISJOINT segments in one launch (blockIdx.y = segment):
// the delivery of a row-partitioned product in the process-per-GPU layer (distributed.py: own rows of the boundary
// blocks' sums + every received partial-y segment) -- one launch behind the join instead of one per segment.
// Here pc.base[c] is the segment's own first element (not a virtual base), lo / hi its range in y.
template <typename T>
__global__ void __launch_bounds__(256) vec_add_segments_kernel(T *__restrict__ y, VecPieces pc) {
    const int c = blockIdx.y;
    const T *__restrict__ src = reinterpret_cast<const T *>(pc.base[c]);
    const long long lo = pc.lo[c], n = pc.hi[c] - lo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[lo + i] = add(y[lo + i], src[i]);
}
template <typename T>
static hipError_t add_segments_typed(void *y, const VecPieces &pc, int npieces, hipStream_t stream) {
    long long longest = 0;
    for (int c = 0; c < npieces; ++c) longest = longest > pc.hi[c] - pc.lo[c] ? longest : pc.hi[c] - pc.lo[c];
    if (npieces <= 0 || longest <= 0) return hipSuccess;
    long long nblk = (longest + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL((vec_add_segments_kernel<T>), dim3((unsigned)nblk, (unsigned)npieces), dim3(256), 0, stream, (T *)y, pc);
    return hipGetLastError();
}
hipError_t launch_vec_add_segments(int dtype, void *y, const VecPieces &pc, int npieces, hipStream_t stream) {
    switch (dtype) {
        case 0: return add_segments_typed<float>(y, pc, npieces, stream);
        case 1: return add_segments_typed<double>(y, pc, npieces, stream);
        case 2: return add_segments_typed<c64>(y, pc, npieces, stream);
        case 3: return add_segments_typed<c128>(y, pc, npieces, stream);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t vec_launch(int which, void *dst, const void *src, long long n, const void *beta_p, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long long nblk = (n + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    if (which == 0)
        hipLaunchKernelGGL((vec_add_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, stream, (T *)dst, (const T *)src, n);
    else
        hipLaunchKernelGGL((vec_axpby_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, stream, (T *)dst, (const T *)src, n,
                           load_scalar<T>(beta_p, 0.0));
    return hipGetLastError();
}
hipError_t launch_vec_add(int dtype, void *dst, const void *src, long long n, hipStream_t stream) {
    switch (dtype) {
        case 0: return vec_launch<float>(0, dst, src, n, nullptr, stream);
        case 1: return vec_launch<double>(0, dst, src, n, nullptr, stream);
        case 2: return vec_launch<c64>(0, dst, src, n, nullptr, stream);
        case 3: return vec_launch<c128>(0, dst, src, n, nullptr, stream);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_vec_axpby(int dtype, void *y, const void *r, long long n, const void *beta, hipStream_t stream) {
    switch (dtype) {
        case 0: return vec_launch<float>(1, y, r, n, beta, stream);
        case 1: return vec_launch<double>(1, y, r, n, beta, stream);
        case 2: return vec_launch<c64>(1, y, r, n, beta, stream);
        case 3: return vec_launch<c128>(1, y, r, n, beta, stream);
    }
    return hipErrorInvalidValue;
}

// ========================================================================================
// rowcolvals(A) from the packed image (reference src/sparse.jl:17-123): every stored entry leaves as
// a COO triple (1-based), the off-diagonal columns of a symmetric operator a second time transposed.
// One wave per WaveWork descriptor; its output offset was summed up on the host (no atomics, the
// order of the triples is fixed).  HBM-bound, one-off.
// ========================================================================================
template <typename T>
__global__ void __launch_bounds__(64 * kWavesPerWg) export_coo_kernel(const WaveWork *__restrict__ waves, long long nwaves,
                                                         const long long *__restrict__ out_off,
                                                         const uint4 *__restrict__ values, const int *__restrict__ rows,
                                                         const int *__restrict__ cols, long long *__restrict__ orow,
                                                         long long *__restrict__ ocol, T *__restrict__ oval) {
    constexpr int E = TT<T>::E;
    const long long wv = (long long)blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
    if (wv >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const WaveD wd = load_wave(waves + wv);
    if (wd.work != WORK_PANEL || wd.npieces == 0) return;
    const PieceD pc = wd.first;
    const int m = wd.m, ncols = pc.ncols, kinds = pc.kind;
    const T *__restrict__ vb = reinterpret_cast<const T *>(values + (((uint64_t)pc.val_hi << 32) | pc.val_lo));
    const int s1w = wd.seg1_w, s1x = wd.seg1_x - wd.seg1_w;
    const int s2w = wd.seg2_w, s2x = pc.seg2_x - wd.seg2_w;
    const long long base = out_off[wv];
    const long long tbase = base + (long long)m * ncols;  // transposed copies follow the forward triples
    // t-th KIND_OFF column of the piece gets the t-th transposed slot: count them in order per lane
    // group is not needed -- the host laid the transposed region out per column index w as well, with
    // holes squeezed out by its own prefix; here the prefix over columns is recomputed by lane 0..63
    // cooperatively in chunks of 64 columns
    int toff = 0;  // number of KIND_OFF columns in front of the current chunk
    for (int w0 = 0; w0 < ncols; w0 += 64) {
        const int w = w0 + lane;
        bool off = false;
        int ci = 0;
        if (w < ncols) {
            if (pc.xbase < 0) {
                const int raw = cols[pc.col_off + w];
                off = raw >= 0 && (kinds & 3) == KIND_OFF;
                ci = raw & 0x7fffffff;
            } else {
                const int sh = w < s1w ? 0 : (w < s2w ? 2 : 4);
                off = ((kinds >> sh) & 3) == KIND_OFF;
                ci = w + (w < s1w ? pc.xbase : (w < s2w ? s1x : s2x));
            }
        }
        const unsigned long long mask = __ballot(off);
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (w < ncols) {
            const int s = w / E, e = w % E;
            for (int i = 0; i < m; ++i) {
                const int ri = (wd.rbase >= 0) ? wd.rbase + i : rows[wd.row_off + i];
                const T v = vb[((long long)s * m + i) * E + e];
                const long long o = base + (long long)w * m + i;
                orow[o] = ri + 1;
                ocol[o] = ci + 1;
                oval[o] = v;
                if (off) {
                    const long long t = tbase + (long long)(toff + rank) * m + i;
                    orow[t] = ci + 1;
                    ocol[t] = ri + 1;
                    oval[t] = v;
                }
            }
        }
        toff += __popcll(mask);
    }
}

hipError_t launch_export_coo(int dtype, const void *d_waves, long long nwaves, const void *d_out_off,
                             const void *d_values, const void *d_rows, const void *d_cols, void *orow, void *ocol,
                             void *oval, hipStream_t stream) {
    if (nwaves <= 0) return hipSuccess;
    const dim3 grid((unsigned)((nwaves + kWavesPerWg - 1) / kWavesPerWg)), block(64 * kWavesPerWg);
#define BSM_EXPORT(T)                                                                                             \
    hipLaunchKernelGGL((export_coo_kernel<T>), grid, block, 0, stream, (const WaveWork *)d_waves, nwaves,         \
                       (const long long *)d_out_off, (const uint4 *)d_values, (const int *)d_rows,                \
                       (const int *)d_cols, (long long *)orow, (long long *)ocol, (T *)oval)
    switch (dtype) {
        case 0: BSM_EXPORT(float); break;
        case 1: BSM_EXPORT(double); break;
        case 2: BSM_EXPORT(c64); break;
        case 3: BSM_EXPORT(c128); break;
        default: return hipErrorInvalidValue;
    }
#undef BSM_EXPORT
    return hipGetLastError();
}

// ========================================================================================
// device-side repacking (bsm_options.blocks_memspace = BSM_MEM_DEVICE): the caller's blocks already
// live in HBM (e.g. ROCArrays), so the strip layout is written by a kernel instead of the host
// packer -- no matrix byte crosses PCIe.  One workgroup per chunk (<= 64 rows of one block);
// consecutive lanes read consecutive rows of a column (coalesced) and write the same slot of
// consecutive 16-byte units.  HBM-bound, runs once per operator.
// ========================================================================================
template <typename U>
__global__ void __launch_bounds__(256) pack_kernel(const PackChunk *__restrict__ plan, const int *__restrict__ colpos,
                                                   U *__restrict__ values, int E) {
    const PackChunk c = plan[blockIdx.x];
    const U *__restrict__ src = reinterpret_cast<const U *>(c.src);
    U *__restrict__ dst = values + c.dst_unit * (uint64_t)E;
    const int mc = c.mc;
    // lanes run over the rows of the chunk, rounded up to a power of two <= 64 so that a wave covers
    // whole columns
    int rp = 1;
    while (rp < mc) rp <<= 1;
    const int i = threadIdx.x & (rp - 1);
    const int cpw = 256 / rp;  // columns per pass
    if (i >= mc) return;
    for (int w = threadIdx.x / rp; w < c.n; w += cpw) {
        const int q = c.perm_off < 0 ? c.woff + w : colpos[c.perm_off + w];
        const U v = c.trans ? src[(int64_t)w + (int64_t)(c.ra + i) * c.ld] : src[(int64_t)(c.ra + i) + (int64_t)w * c.ld];
        dst[((int64_t)(q / E) * mc + i) * E + (q % E)] = v;
    }
}

hipError_t launch_pack(int es, const void *d_plan, long long nchunks, const void *d_colpos, void *d_values,
                       hipStream_t stream) {
    if (nchunks <= 0) return hipSuccess;
    const PackChunk *plan = (const PackChunk *)d_plan;
    const int *cp = (const int *)d_colpos;
    const dim3 grid((unsigned)nchunks), block(256);
    if (es == 4)
        hipLaunchKernelGGL((pack_kernel<uint32_t>), grid, block, 0, stream, plan, cp, (uint32_t *)d_values, 4);
    else if (es == 8)
        hipLaunchKernelGGL((pack_kernel<uint64_t>), grid, block, 0, stream, plan, cp, (uint64_t *)d_values, 2);
    else
        hipLaunchKernelGGL((pack_kernel<uint4>), grid, block, 0, stream, plan, cp, (uint4 *)d_values, 1);
    return hipGetLastError();
}

// ========================================================================================
// synthetic operators of BASELINE.json generated IN HBM (include/bsm_synth.h): the counter-based
// SplitMix64 streams of blocksparsematrices.jl_amd/synthetic.py, bit-identical to the numpy code.
//   u(s, k) = mix(s + GOLDEN * (k + 1)),  value = (u >> 11) * 2^-53 * 2 - 1  (fp64, then cast)
// ========================================================================================
__host__ __device__ __forceinline__ uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double synth_unit(uint64_t stream, uint64_t k) {
    const uint64_t u = synth_mix(stream + 0x9E3779B97F4A7C15ull * (k + 1));
    return (double)(u >> 11) * 0x1.0p-53 * 2.0 - 1.0;
}
__device__ __forceinline__ void synth_store(float *p, double v) { *p = (float)v; }
__device__ __forceinline__ void synth_store(double *p, double v) { *p = v; }

struct SynthBlock {
    uint64_t dst;     // device address, column-major m x n, leading dimension m
    uint64_t stream;  // mix(seed ^ mix(b + 1))
    int32_t m, n;
    int32_t symmetrise, pad;  // 1: (D + D^T) / 2 of the m x m draw (diagonal blocks, docs/src/symmetric.md:49-50)
};

template <typename T>
__global__ void __launch_bounds__(256) synth_blocks_kernel(const SynthBlock *__restrict__ blocks) {
    const SynthBlock b = blocks[blockIdx.x];
    T *__restrict__ dst = reinterpret_cast<T *>(b.dst);
    const long long cnt = (long long)b.m * b.n;
    for (long long k = (long long)blockIdx.y * 256 + threadIdx.x; k < cnt; k += (long long)gridDim.y * 256) {
        double v = synth_unit(b.stream, (uint64_t)k);
        if (b.symmetrise) {
            const long long i = k % b.m, j = k / b.m;
            // the reference recipe rounds the draw to T first, then averages in T
            const T a = (T)v, c = (T)synth_unit(b.stream, (uint64_t)(j + i * b.m));
            dst[k] = (a + c) / (T)2;
        } else {
            synth_store(&dst[k], v);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) synth_vector_kernel(T *__restrict__ dst, long long n, uint64_t stream) {
    long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; k < n; k += stride) synth_store(&dst[k], synth_unit(stream, (uint64_t)k));
}

// ========================================================================================
// Bare streaming read (include/bsm_synth.h: bsm_bench_stream): what the memory system delivers for a
// buffer of a given size with the product kernel's request shape -- 8 independent 16-byte non-temporal
// loads per lane, 8 KB per wave, 4 waves per workgroup -- and nothing else to do.  `hop` adds the one
// dependent scalar load every product wave starts with (its 64-byte descriptor): the wave's offset comes
// out of a table instead of blockIdx.  The floor bench.py prints beside the product's time.
// ========================================================================================
__global__ void __launch_bounds__(256) stream_floor_kernel(const u32x4 *__restrict__ src, double *__restrict__ sink,
                                                           long long total16, const long long *__restrict__ hop) {
    const int lane = threadIdx.x & 63;
    long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (hop) wave = __builtin_amdgcn_readfirstlane((int)hop[__builtin_amdgcn_readfirstlane((int)wave) * 8]);  // one 64-byte record per wave
    const long long p = wave * 512 + lane;
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + (p + 64 * k < total16 ? p + 64 * k : 0));
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (acc == 0x9E3779B9u) sink[wave & 1023] = (double)acc;  // keeps the loads alive; practically never taken
}

hipError_t launch_stream_floor(const void *src, long long bytes, void *sink, const void *hop, hipStream_t stream) {
    const long long total16 = bytes / 16;
    if (total16 <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((total16 + 2047) / 2048);
    hipLaunchKernelGGL(stream_floor_kernel, dim3(grid), dim3(256), 0, stream, (const u32x4 *)src, (double *)sink, total16,
                       (const long long *)hop);
    return hipGetLastError();
}

hipError_t launch_synth_blocks(int dtype, const void *d_desc, long long nblocks, int tiles, hipStream_t stream) {
    if (nblocks <= 0) return hipSuccess;
    const dim3 grid((unsigned)nblocks, (unsigned)tiles), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL((synth_blocks_kernel<float>), grid, block, 0, stream, (const SynthBlock *)d_desc);
    else if (dtype == 1)
        hipLaunchKernelGGL((synth_blocks_kernel<double>), grid, block, 0, stream, (const SynthBlock *)d_desc);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_synth_vector(int dtype, void *dst, long long n, unsigned long long stream_seed, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    long long nblk = (n + 255) / 256;
    if (nblk > 8192) nblk = 8192;
    if (dtype == 0)
        hipLaunchKernelGGL((synth_vector_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, stream, (float *)dst, n, (uint64_t)stream_seed);
    else if (dtype == 1)
        hipLaunchKernelGGL((synth_vector_kernel<double>), dim3((unsigned)nblk), dim3(256), 0, stream, (double *)dst, n, (uint64_t)stream_seed);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace bsm
