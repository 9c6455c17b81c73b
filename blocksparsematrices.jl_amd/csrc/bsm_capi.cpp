// bsm_capi.cpp -- the extern "C" surface of libbsmrocm.so (include/bsm_rocm.h).
// Plain pointers and sizes only; converts arguments, runs the host analysis, uploads the
// packed image and forwards bsm_mul to the HIP launchers.  Never throws across the ABI.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "bsm_internal.h"

using namespace bsm;

namespace {
// RAII staging buffers: cached in the handle when uncontended, temporary otherwise
struct Staging {
    bsm_matrix_s *A;
    bool locked = false;
    void *dx = nullptr, *dy = nullptr;
    bool own = false;
    hipError_t acquire(bsm_matrix_s *a, size_t xbytes, size_t ybytes) {
        A = a;
        locked = A->host_mu.try_lock();
        hipError_t e = hipSuccess;
        if (locked) {
            if (A->stage_x_bytes < xbytes) {
                if (A->stage_x) (void)hipFree(A->stage_x);
                A->stage_x = nullptr;
                A->stage_x_bytes = 0;
                e = hipMalloc(&A->stage_x, xbytes + 16);
                if (e != hipSuccess) return e;
                A->stage_x_bytes = xbytes;
            }
            if (A->stage_y_bytes < ybytes) {
                if (A->stage_y) (void)hipFree(A->stage_y);
                A->stage_y = nullptr;
                A->stage_y_bytes = 0;
                e = hipMalloc(&A->stage_y, ybytes + 16);
                if (e != hipSuccess) return e;
                A->stage_y_bytes = ybytes;
            }
            dx = A->stage_x;
            dy = A->stage_y;
        } else {
            own = true;
            e = hipMalloc(&dx, xbytes + 16);
            if (e == hipSuccess) e = hipMalloc(&dy, ybytes + 16);
        }
        return e;
    }
    ~Staging() {
        if (own) {
            if (dx) (void)hipFree(dx);
            if (dy) (void)hipFree(dy);
        }
        if (locked) A->host_mu.unlock();
    }
};
}  // namespace

static thread_local std::string g_err;

namespace bsm {
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(BSM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

hipError_t DeviceGuard::enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev != dev) {
        e = hipSetDevice(dev);
        if (e != hipSuccess) return e;
        active = true;
    }
    return hipSuccess;
}
DeviceGuard::~DeviceGuard() {
    if (active) (void)hipSetDevice(prev);
}
}  // namespace bsm

extern "C" const char *bsm_last_error(void) { return g_err.c_str(); }

#ifndef BSM_BUILD_ID
#define BSM_BUILD_ID "unknown"
#endif
extern "C" const char *bsm_version(void) { return "bsmrocm 0.3 gfx950 build " BSM_BUILD_ID; }

extern "C" void bsm_options_default(bsm_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->struct_size = (int32_t)sizeof(bsm_options);
    o->device = BSM_DEVICE_CURRENT;
    o->scheduler = BSM_SCHED_SERIAL;
    o->accumulate = BSM_ACC_AUTO;
    o->validate = 1;
}

namespace bsm {

int read_options(const bsm_options *opts, bsm_options &o) {
    bsm_options_default(&o);
    if (opts) {
        if (opts->struct_size != (int32_t)sizeof(bsm_options))
            return fail(BSM_ERR_INVALID, "bsm_options.struct_size mismatch (call bsm_options_default)");
        o = *opts;
    }
    if (o.accumulate != BSM_ACC_AUTO && o.accumulate != BSM_ACC_ATOMIC && o.accumulate != BSM_ACC_COLORED &&
        o.accumulate != BSM_ACC_GATHER && o.accumulate != BSM_ACC_DIRECT)
        return fail(BSM_ERR_INVALID, "unknown accumulate mode");
    if (o.own_lo < 0 || o.own_hi < 0 || (o.own_hi > 0 && o.own_hi < o.own_lo))
        return fail(BSM_ERR_INVALID, "bad own_lo/own_hi");
    if (o.transpose_image < 0 || o.transpose_image > 2) return fail(BSM_ERR_INVALID, "bad transpose_image");
    if (o.coloring != BSM_COLOR_WORKSTREAM_DSATUR && o.coloring != BSM_COLOR_DSATUR)
        return fail(BSM_ERR_INVALID, "unknown colouring algorithm");
    if (o.blocks_memspace != BSM_MEM_HOST && o.blocks_memspace != BSM_MEM_DEVICE)
        return fail(BSM_ERR_INVALID, "bad blocks_memspace");
    return BSM_OK;
}

template <typename V> hipError_t upload(const V &v, void **dptr, long long &total) {
    const size_t bytes = v.size() * sizeof(decltype(v[0]));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    total += (long long)bytes;
    if (bytes) e = hipMemcpy(*dptr, v.data(), bytes, hipMemcpyHostToDevice);
    return e;
}

// Streams the packed values to the device while they are being packed: two pinned staging windows
// (kept for the life of the process: pinning 2 x 64 MiB costs more than packing a C2-sized operator)
// and asynchronous copies on a private stream.  Small operators decline and take the one-shot path.
struct PinnedPool {
    std::mutex mu;  // one streamed create at a time per process
    char *buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
};
PinnedPool g_pool;
size_t stream_min_bytes() {  // BSM_STREAM_MIN_BYTES: smaller operators take the one-shot upload
    const char *e = std::getenv("BSM_STREAM_MIN_BYTES");
    return (e && *e) ? (size_t)std::strtoull(e, nullptr, 10) : ((size_t)128 << 20);
}

struct DeviceSink : ValueSink {
    void **dptr;
    std::unique_lock<std::mutex> lock;
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    int k = 0;
    bool active = false;
    explicit DeviceSink(void **d) : dptr(d) {}
    static std::string msg(hipError_t e, const char *what) { return std::string(what) + ": " + hipGetErrorString(e); }
    std::string begin(size_t total, bool *use) override {
        *use = false;
        if (total == 0 || total < stream_min_bytes()) return "";
        lock = std::unique_lock<std::mutex>(g_pool.mu);
        hipError_t e = hipMalloc(dptr, total);
        if (e != hipSuccess) return msg(e, "hipMalloc(values)");
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return msg(e, "upload stream");
        active = *use = true;
        return "";
    }
    char *window(size_t bytes) override {
        const int i = k & 1;
        if (pending[i]) {
            if (hipEventSynchronize(ev[i]) != hipSuccess) return nullptr;
            pending[i] = false;
        }
        if (g_pool.cap[i] < bytes) {
            if (g_pool.buf[i]) (void)hipHostFree(g_pool.buf[i]);
            g_pool.buf[i] = nullptr;
            g_pool.cap[i] = 0;
            const size_t want = std::max<size_t>(bytes, 64u << 20);
            if (hipHostMalloc((void **)&g_pool.buf[i], want, hipHostMallocDefault) != hipSuccess) return nullptr;
            g_pool.cap[i] = want;
        }
        return g_pool.buf[i];
    }
    std::string commit(size_t offset, size_t bytes) override {
        const int i = k & 1;
        hipError_t e = hipMemcpyAsync((char *)*dptr + offset, g_pool.buf[i], bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ev[i], st);
        if (e != hipSuccess) return msg(e, "streamed upload");
        pending[i] = true;
        k++;
        return "";
    }
    std::string end() override {
        hipError_t e = hipStreamSynchronize(st);
        pending[0] = pending[1] = false;
        if (lock.owns_lock()) lock.unlock();  // the staging windows are free for the next create
        return e == hipSuccess ? "" : msg(e, "streamed upload");
    }
    ~DeviceSink() override {
        if (st) {
            (void)hipStreamSynchronize(st);  // the staging buffers go back to the pool idle
            (void)hipStreamDestroy(st);
        }
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

std::unique_ptr<ValueSink> make_device_sink(void **d_values) { return std::unique_ptr<ValueSink>(new DeviceSink(d_values)); }

void free_image(DeviceImage &img) {
    for (void **p : {&img.d_values, &img.d_rows, &img.d_cols, &img.d_waves, &img.d_waves_multi, &img.d_ws, &img.d_inv_ptr[0],
                     &img.d_inv_ptr[1], &img.d_inv_idx[0], &img.d_inv_idx[1]}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

void fill_image(const Analysis &an, const bsm_options &o, bool use_own, DeviceImage &img) {
    img.dtype = an.dtype;
    img.nrows = an.nrows;
    img.ncols = an.ncols;
    img.own_lo = (use_own && o.own_lo > 0) ? o.own_lo - 1 : 0;
    img.own_hi = (use_own && o.own_hi > 0) ? std::min<long long>(o.own_hi, an.nrows) : an.nrows;
    img.value_bytes = an.value_bytes;
    img.nwg_main = an.nwg_main;
    img.nwg_total = an.nwg_total;
    img.nwg_multi = an.nwg_multi;
    img.lane_fill = (float)an.lane_fill;
    img.mean_rows = (float)an.mean_rows;
    img.exclusive_fwd = an.exclusive_fwd && (o.accumulate == BSM_ACC_AUTO || o.accumulate == BSM_ACC_DIRECT);
    img.has_off = false;
    img.max_rows = 1;
    for (const WaveWork &w : an.waves)
        if (w.work == WORK_PANEL && w.npieces > 0) {
            if (w.first.kind & kKindHasOff) img.has_off = true;
            img.max_rows = std::max(img.max_rows, (int)w.m);
        }
    if (!img.exclusive_fwd) img.nwg_total = img.nwg_main;
    img.color_wg_ptr.assign(an.color_wg_ptr.begin(), an.color_wg_ptr.end());
    img.device_bytes = (long long)((size_t)an.value_bytes + an.rows.size() * 4 + an.cols.size() * 4 +
                                   (an.waves.size() + an.waves_multi.size()) * sizeof(WaveWork));
    if (an.gather)
        img.device_bytes += (long long)((an.ws_slots + 8) * an.es + (an.inv_ptr[0].size() + an.inv_ptr[1].size()) * 8 +
                                        (an.inv_idx[0].size() + an.inv_idx[1].size()) * 4);
}

hipError_t upload_image(Analysis &an, DeviceImage &img, int dev) {
    img.device = dev;
    long long total = 0;
    hipError_t e = hipSuccess;
    if (img.d_values)  // the packer streamed them (DeviceSink)
        total += an.value_bytes;
    else
        e = upload(an.values, &img.d_values, total);
    if (e == hipSuccess) e = upload(an.rows, &img.d_rows, total);
    if (e == hipSuccess) e = upload(an.cols, &img.d_cols, total);
    if (e == hipSuccess) e = upload(an.waves, &img.d_waves, total);
    if (e == hipSuccess && !an.waves_multi.empty()) e = upload(an.waves_multi, &img.d_waves_multi, total);
    if (e == hipSuccess && an.gather) {
        img.ws_fbase = an.ws_fbase;
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = upload(an.inv_ptr[k], &img.d_inv_ptr[k], total);
            if (e == hipSuccess) e = upload(an.inv_idx[k], &img.d_inv_idx[k], total);
        }
        if (e == hipSuccess) {
            const size_t wsb = (size_t)(an.ws_slots + 8) * (size_t)an.es;
            e = hipMalloc(&img.d_ws, wsb);
            if (e == hipSuccess) e = hipMemset(img.d_ws, 0, wsb);
            total += (long long)wsb;
        }
    }
    if (e == hipSuccess) an.values.release();  // packed host copy no longer needed
    if (std::getenv("BSM_PLACEMENT_DEBUG") && e == hipSuccess && an.value_bytes >= (64 << 20)) {
        // developer probe (tools/placement_which.py): where the image landed, and what a BARE streaming read of its
        // value stream takes there
        void *sink = nullptr;
        hipEvent_t a, b;
        float ms = 0.f;
        if (hipMalloc(&sink, 8192) == hipSuccess && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
            for (int r = 0; r < 3; r++) (void)launch_stream_floor(img.d_values, an.value_bytes / 16 * 16, sink, nullptr, nullptr);
            (void)hipEventRecord(a, nullptr);
            for (int r = 0; r < 10; r++) (void)launch_stream_floor(img.d_values, an.value_bytes / 16 * 16, sink, nullptr, nullptr);
            (void)hipEventRecord(b, nullptr);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
            (void)hipFree(sink);
        }
        std::fprintf(stderr, "[bsm image] values %p (%lld B) bare stream %.1f us | rows %p cols %p waves %p (%zu)\n", img.d_values,
                     (long long)an.value_bytes, ms * 100.f, img.d_rows, img.d_cols, img.d_waves, an.waves.size());
    }
    return e;
}

// Device side of a *_create call: the target device is made current BEFORE the analysis runs, so
// that the packer can stream the values to it (DeviceSink) instead of building a host copy first.
struct CreateCtx {
    bsm_matrix_s *A;  // owned until release(): any early return or exception frees it, device side included
    int dev = BSM_DEVICE_NONE;
    DeviceGuard guard;
    DeviceSink sink, sink_t;
    CreateCtx() : A(new bsm_matrix_s()), sink(&A->img.d_values), sink_t(&A->img_t.d_values) {}
    CreateCtx(const CreateCtx &) = delete;
    ~CreateCtx() {
        if (!A) return;
        free_image(A->img);  // the handle's device is still current (guard outlives this body)
        free_image(A->img_t);
        dist_destroy(A);
        delete A;
    }
    bsm_matrix_s *release() {
        bsm_matrix_s *p = A;
        A = nullptr;
        return p;
    }
    int open(const bsm_options &o) {
        if (o.device == BSM_DEVICE_NONE) return BSM_OK;
        dev = o.device;
        if (dev == BSM_DEVICE_CURRENT) {
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
        }
        hipError_t e = guard.enter(dev);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
        return BSM_OK;
    }
    ValueSink *values() { return dev == BSM_DEVICE_NONE ? nullptr : &sink; }
    ValueSink *values_t() { return dev == BSM_DEVICE_NONE ? nullptr : &sink_t; }
};

int finish_create(CreateCtx &cx, const bsm_options &o, bsm_matrix_t *out) {
    bsm_matrix_s *A = cx.A;
    fill_image(A->an, o, true, A->img);
    if (A->has_t) fill_image(A->an_t, o, false, A->img_t);
    if (cx.dev != BSM_DEVICE_NONE) {
        hipError_t e = upload_image(A->an, A->img, cx.dev);
        if (e == hipSuccess && A->has_t) e = upload_image(A->an_t, A->img_t, cx.dev);
        if (e != hipSuccess) return hip_fail(e, "device upload");
        A->on_device = true;
    }
    *out = cx.release();
    return BSM_OK;
}

// analysis errors are the caller's (bad arguments); the value sink reports device failures
int build_error(const std::string &err) {
    const bool device = err.compare(0, 3, "hip") == 0 || err.find("upload") != std::string::npos ||
                        err.find("value sink") != std::string::npos;
    return fail(device ? BSM_ERR_DEVICE : BSM_ERR_INVALID, err);
}

// Second ordering: the transposed operator as a forward image (rows <-> columns, blocks read
// transposed by the packer).  Built from the same caller arrays, before they are released.
std::string build_transpose_image(bsm_matrix_s *A, const std::vector<BlockIn> &in, const bsm_options &o,
                                  ValueSink *sink) {
    std::vector<BlockIn> t(in.size());
    for (size_t b = 0; b < in.size(); b++) {
        const BlockIn &B = in[b];
        BlockIn &Tb = t[b];
        Tb.data = B.data;
        Tb.m = B.n;
        Tb.n = B.m;
        Tb.ld = B.ld;
        Tb.ridx = B.cidx;
        Tb.cidx = B.ridx;
        Tb.r0 = B.c0;
        Tb.c0 = B.r0;
        Tb.kind = KIND_PLAIN;
        Tb.trans = !B.trans;
    }
    AnalysisOptions a;
    a.scheduler = 0;
    a.accumulate = o.accumulate;
    a.sink = sink;
    a.blocks_on_device = (o.blocks_memspace == BSM_MEM_DEVICE);
    std::string err = A->an_t.build(MT_BLOCKSPARSE, A->an.dtype, A->an.ncols, A->an.nrows, t, a);
    if (err.empty()) A->has_t = true;
    return err;
}

// Executes the pack plan of an analysis built with blocks_on_device on the CURRENT device: the
// strip-packed value stream is written by a kernel straight from the caller's device blocks.
hipError_t device_pack(Analysis &an, void **d_values) {
    hipError_t e = hipMalloc(d_values, (size_t)std::max<int64_t>(an.value_bytes, 16));
    if (e == hipSuccess && std::getenv("BSM_TIMING"))
        std::fprintf(stderr, "[bsm] values at %p (%lld bytes)\n", *d_values, (long long)an.value_bytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(*d_values, 0, (size_t)std::max<int64_t>(an.value_bytes, 16), nullptr);  // strip tails
    void *d_plan = nullptr, *d_cp = nullptr;
    if (e == hipSuccess && !an.pack_plan.empty()) {
        e = hipMalloc(&d_plan, an.pack_plan.size() * sizeof(PackChunk));
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_plan, an.pack_plan.data(), an.pack_plan.size() * sizeof(PackChunk),
                               hipMemcpyHostToDevice, nullptr);
        if (e == hipSuccess && !an.pack_colpos.empty()) {
            e = hipMalloc(&d_cp, an.pack_colpos.size() * 4);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_cp, an.pack_colpos.data(), an.pack_colpos.size() * 4, hipMemcpyHostToDevice, nullptr);
        }
        if (e == hipSuccess) e = launch_pack(an.es, d_plan, (long long)an.pack_plan.size(), d_cp, *d_values, nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (d_plan) (void)hipFree(d_plan);
    if (d_cp) (void)hipFree(d_cp);
    an.pack_plan.clear();
    an.pack_plan.shrink_to_fit();
    an.pack_colpos.clear();
    an.pack_colpos.shrink_to_fit();
    return e;
}

AnalysisOptions to_aopt(const bsm_options &o, ValueSink *sink) {
    AnalysisOptions a;
    a.sink = sink;
    a.blocks_on_device = (o.blocks_memspace == BSM_MEM_DEVICE);
    a.coloring = (int)o.coloring;
    a.scheduler = o.scheduler;
    a.validate = 1;  // indices are always range-checked: a bad index must never reach a kernel
    a.accumulate = o.accumulate;
    a.own_lo = o.own_lo;
    a.own_hi = o.own_hi;
    return a;
}

}  // namespace bsm

namespace {

// Common tail of every *_create: `in` is the block list in its final order (VBCRS: sorted, with
// A->an's perm / rowptr / ... already filled).  Single device: analysis + packing + upload; with
// bsm_options.ctx: whole-operator bookkeeping, then one image per device of the context.
int create_handle(int mtype, int dtype, int64_t nrows, int64_t ncols, const std::vector<BlockIn> &in,
                  const bsm_options &o, CreateCtx &cx, bsm_matrix_t *out) {
    bsm_matrix_s *A = cx.A;
    if (o.ctx) {
        AnalysisOptions ao = to_aopt(o, nullptr);
        ao.meta_only = true;
        ao.own_lo = ao.own_hi = 0;
        std::string err = A->an.build(mtype, dtype, nrows, ncols, in, ao);
        if (!err.empty()) return build_error(err);
        int rc = dist_create(A, (bsm_ctx_s *)o.ctx, mtype, dtype, nrows, ncols, in, o);
        if (rc != BSM_OK) return rc;  // ~CreateCtx releases whatever the parts already hold
        A->on_device = true;
        *out = cx.release();
        return BSM_OK;
    }
    int rc = cx.open(o);
    if (rc != BSM_OK) return rc;
    const bool devblocks = (o.blocks_memspace == BSM_MEM_DEVICE);
    if (devblocks && cx.dev == BSM_DEVICE_NONE)
        return fail(BSM_ERR_INVALID, "device-resident blocks need a device handle");
    std::string err = A->an.build(mtype, dtype, nrows, ncols, in, to_aopt(o, devblocks ? nullptr : cx.values()));
    bool want_t = o.transpose_image == 1;
    if (o.transpose_image == 2 && cx.dev != BSM_DEVICE_NONE) {  // "when it is cheap"
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            want_t = (size_t)A->an.value_bytes <= free_b / 16;
        else
            (void)hipGetLastError();
    }
    if (err.empty() && want_t && mtype != MT_SYMMETRIC) {
        bool plain = true;
        for (const BlockIn &B : in) plain &= (B.kind == KIND_PLAIN);
        if (plain) err = build_transpose_image(A, in, o, devblocks ? nullptr : cx.values_t());
    }
    if (!err.empty()) return build_error(err);
    if (devblocks) {
        hipError_t e = device_pack(A->an, &A->img.d_values);
        if (e == hipSuccess && A->has_t) e = device_pack(A->an_t, &A->img_t.d_values);
        if (e != hipSuccess) return hip_fail(e, "device-side packing");
    }
    return finish_create(cx, o, out);
}

// VBCRS front end (reference src/vbcrs.jl:84-117): stable sort by (rowstart, colstart), rowptr, ...
int create_vbcrs(int dtype, int64_t nrows, int64_t ncols, const std::vector<BlockIn> &unsorted,
                 const bsm_options &o, bsm_matrix_t *out) {
    const int64_t nb = (int64_t)unsorted.size();
    std::vector<int64_t> rs(nb), cs(nb);
    for (int64_t b = 0; b < nb; b++) {
        rs[b] = unsorted[b].r0;
        cs[b] = unsorted[b].c0;
    }
    CreateCtx cx;
    const std::vector<int64_t> p = cx.A->an.vbcrs_bookkeeping(nb, rs.data(), cs.data());
    std::vector<BlockIn> in(nb);
    for (int64_t k = 0; k < nb; k++) in[k] = unsorted[p[k]];
    return create_handle(MT_VBCRS, dtype, nrows, ncols, in, o, cx, out);
}

}  // namespace

#define BSM_GUARDED(...)                                         \
    try {                                                        \
        __VA_ARGS__                                              \
    } catch (const std::bad_alloc &) {                           \
        return fail(BSM_ERR_ALLOC, "out of host memory");        \
    } catch (const std::exception &e) {                          \
        return fail(BSM_ERR_INVALID, e.what());                  \
    }

extern "C" int bsm_vbcrs_create(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                                const void *const *blocks, const int64_t *m, const int64_t *n,
                                const int64_t *ld, const int64_t *rowstart, const int64_t *colstart,
                                const bsm_options *opts, bsm_matrix_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (nblocks < 1) return fail(BSM_ERR_INVALID, "VBCRS needs at least one block (reference src/vbcrs.jl:81)");
        if (!blocks || !m || !n || !ld || !rowstart || !colstart) return fail(BSM_ERR_INVALID, "null argument");
        bsm_options o;
        int rc = read_options(opts, o);
        if (rc) return rc;
        std::vector<BlockIn> in((size_t)nblocks);
        for (int64_t b = 0; b < nblocks; b++) {
            BlockIn &B = in[b];
            B.data = (const char *)blocks[b];
            B.m = m[b];
            B.n = n[b];
            B.ld = ld[b];
            B.ridx = B.cidx = nullptr;
            B.r0 = rowstart[b];
            B.c0 = colstart[b];
            B.kind = KIND_PLAIN;
        }
        return create_vbcrs(dtype, nrows, ncols, in, o, out);)
}

extern "C" int bsm_vbcrs_create_from_blocksparse(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                                                 const void *const *blocks, const int64_t *m,
                                                 const int64_t *n, const int64_t *ld,
                                                 const int64_t *const *rowidx, const int64_t *const *colidx,
                                                 const bsm_options *opts, bsm_matrix_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (nblocks < 1) return fail(BSM_ERR_INVALID, "VBCRS needs at least one block (reference src/vbcrs.jl:81)");
        if (!blocks || !m || !n || !ld || !rowidx || !colidx) return fail(BSM_ERR_INVALID, "null argument");
        bsm_options o;
        int rc = read_options(opts, o);
        if (rc) return rc;
        std::vector<BlockIn> in((size_t)nblocks);
        for (int64_t b = 0; b < nblocks; b++) {
            BlockIn &B = in[b];
            // first(rowindices(bsm, i)), first(colindices(bsm, i)): reference src/vbcrs.jl:201-215
            if (m[b] < 1 || n[b] < 1 || !rowidx[b] || !colidx[b])
                return fail(BSM_ERR_INVALID, "block " + std::to_string(b + 1) + ": empty index list (first() of it is undefined)");
            B.data = (const char *)blocks[b];
            B.m = m[b];
            B.n = n[b];
            B.ld = ld[b];
            B.ridx = B.cidx = nullptr;
            B.r0 = rowidx[b][0];
            B.c0 = colidx[b][0];
            B.kind = KIND_PLAIN;
        }
        return create_vbcrs(dtype, nrows, ncols, in, o, out);)
}

extern "C" int bsm_vbcrs_create_from_symmetric(int dtype, int64_t nrows, int64_t ncols, int64_t ndiag,
                                               const void *const *diag, const int64_t *dsize,
                                               const int64_t *dld, const int64_t *diagstart,
                                               int64_t noff, const void *const *off, const int64_t *m,
                                               const int64_t *n, const int64_t *ld,
                                               const int64_t *rowstart, const int64_t *colstart,
                                               const bsm_options *opts, bsm_matrix_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (ndiag < 0 || noff < 0 || ndiag + noff < 1) return fail(BSM_ERR_INVALID, "VBCRS needs at least one block");
        if (ndiag > 0 && (!diag || !dsize || !dld || !diagstart)) return fail(BSM_ERR_INVALID, "null argument");
        if (noff > 0 && (!off || !m || !n || !ld || !rowstart || !colstart)) return fail(BSM_ERR_INVALID, "null argument");
        bsm_options o;
        int rc = read_options(opts, o);
        if (rc) return rc;
        // bookkeeping over the virtual block list of the reference's functors (src/vbcrs.jl:222-262):
        // [diagonals..., offdiagonals..., transpose(offdiagonals)...]
        const int64_t nv = ndiag + 2 * noff;
        std::vector<int64_t> rs(nv), cs(nv);
        for (int64_t d = 0; d < ndiag; d++) rs[d] = cs[d] = diagstart[d];
        for (int64_t b = 0; b < noff; b++) {
            rs[ndiag + b] = rowstart[b];
            cs[ndiag + b] = colstart[b];
            rs[ndiag + noff + b] = colstart[b];
            cs[ndiag + noff + b] = rowstart[b];
        }
        CreateCtx cx;
        cx.A->an.vbcrs_bookkeeping(nv, rs.data(), cs.data());
        // ... the image keeps every off-diagonal block once (the symmetric one)
        std::vector<BlockIn> in;
        in.reserve((size_t)(ndiag + noff));
        for (int64_t d = 0; d < ndiag; d++) {
            BlockIn B;
            B.data = (const char *)diag[d];
            B.m = B.n = dsize[d];
            B.ld = dld[d];
            B.ridx = B.cidx = nullptr;
            B.r0 = B.c0 = diagstart[d];
            B.kind = KIND_DIAG;
            in.push_back(B);
        }
        for (int64_t b = 0; b < noff; b++) {
            BlockIn B;
            B.data = (const char *)off[b];
            B.m = m[b];
            B.n = n[b];
            B.ld = ld[b];
            B.ridx = B.cidx = nullptr;
            B.r0 = rowstart[b];
            B.c0 = colstart[b];
            B.kind = KIND_OFF;
            in.push_back(B);
        }
        return create_handle(MT_VBCRS, dtype, nrows, ncols, in, o, cx, out);)
}

extern "C" int bsm_blocksparse_create(int dtype, int64_t nrows, int64_t ncols, int64_t nblocks,
                                      const void *const *blocks, const int64_t *m, const int64_t *n,
                                      const int64_t *ld, const int64_t *const *rowidx,
                                      const int64_t *const *colidx, const bsm_options *opts,
                                      bsm_matrix_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (nblocks < 0) return fail(BSM_ERR_INVALID, "negative block count");
        if (nblocks > 0 && (!blocks || !m || !n || !ld || !rowidx || !colidx)) return fail(BSM_ERR_INVALID, "null argument");
        bsm_options o;
        int rc = read_options(opts, o);
        if (rc) return rc;
        std::vector<BlockIn> in((size_t)nblocks);
        for (int64_t b = 0; b < nblocks; b++) {
            BlockIn &B = in[b];
            B.data = (const char *)blocks[b];
            B.m = m[b];
            B.n = n[b];
            B.ld = ld[b];
            B.ridx = rowidx[b];
            B.cidx = colidx[b];
            B.r0 = B.c0 = 0;
            B.kind = KIND_PLAIN;
            if ((B.m > 0 && !B.ridx) || (B.n > 0 && !B.cidx))
                return fail(BSM_ERR_INVALID, "block " + std::to_string(b + 1) + ": null index list");
        }
        CreateCtx cx;
        return create_handle(MT_BLOCKSPARSE, dtype, nrows, ncols, in, o, cx, out);)
}

extern "C" int bsm_symmetric_create(int dtype, int64_t nrows, int64_t ncols, int64_t ndiag,
                                    const void *const *diag, const int64_t *dsize, const int64_t *dld,
                                    const int64_t *const *diagidx, int64_t noff,
                                    const void *const *off, const int64_t *m, const int64_t *n,
                                    const int64_t *ld, const int64_t *const *rowidx,
                                    const int64_t *const *colidx, const bsm_options *opts,
                                    bsm_matrix_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (ndiag < 0 || noff < 0) return fail(BSM_ERR_INVALID, "negative block count");
        if (ndiag > 0 && (!diag || !dsize || !dld || !diagidx)) return fail(BSM_ERR_INVALID, "null argument");
        if (noff > 0 && (!off || !m || !n || !ld || !rowidx || !colidx)) return fail(BSM_ERR_INVALID, "null argument");
        bsm_options o;
        int rc = read_options(opts, o);
        if (rc) return rc;
        std::vector<BlockIn> in;
        in.reserve((size_t)(ndiag + noff));
        for (int64_t d = 0; d < ndiag; d++) {
            BlockIn B;
            B.data = (const char *)diag[d];
            B.m = B.n = dsize[d];
            B.ld = dld[d];
            B.ridx = B.cidx = diagidx[d];
            B.r0 = B.c0 = 0;
            B.kind = KIND_DIAG;
            if (B.m > 0 && !B.ridx)
                return fail(BSM_ERR_INVALID, "diagonal block " + std::to_string(d + 1) + ": null index list");
            in.push_back(B);
        }
        for (int64_t b = 0; b < noff; b++) {
            BlockIn B;
            B.data = (const char *)off[b];
            B.m = m[b];
            B.n = n[b];
            B.ld = ld[b];
            B.ridx = rowidx[b];
            B.cidx = colidx[b];
            B.r0 = B.c0 = 0;
            B.kind = KIND_OFF;
            if ((B.m > 0 && !B.ridx) || (B.n > 0 && !B.cidx))
                return fail(BSM_ERR_INVALID, "off-diagonal block " + std::to_string(b + 1) + ": null index list");
            in.push_back(B);
        }
        CreateCtx cx;
        return create_handle(MT_SYMMETRIC, dtype, nrows, ncols, in, o, cx, out);)
}

// ---- contexts of devices (multi-GPU handles) -----------------------------------------------------
extern "C" int bsm_ctx_create(const int32_t *device_ids, int32_t ndevices, bsm_ctx_t *out) {
    BSM_GUARDED(
        if (!out) return fail(BSM_ERR_INVALID, "out is null");
        *out = nullptr;
        if (ndevices < 1 || ndevices > 64 || !device_ids) return fail(BSM_ERR_INVALID, "a context needs 1..64 devices");
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
        std::unique_ptr<bsm_ctx_s> ctx(new bsm_ctx_s());
        for (int32_t i = 0; i < ndevices; i++) {
            if (device_ids[i] < 0 || device_ids[i] >= count)
                return fail(BSM_ERR_INVALID, "device ordinal " + std::to_string(device_ids[i]) + " does not exist");
            ctx->devices.push_back(device_ids[i]);
        }
        // direct xGMI access between every pair of distinct devices (the halo copies then run device
        // to device; without it the runtime stages them through host memory, still correct)
        // (what the fused fan-out kernels of bsm_dist.cpp rely on is that the access was ENABLED, not that it is
        // possible: only hipSuccess / "already enabled" count, anything else leaves the context on the copy path)
        ctx->peer_ok = true;
        for (int a : ctx->devices)
            for (int b : ctx->devices) {
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                    (void)hipGetLastError();
                    ctx->peer_ok = false;
                    continue;
                }
                DeviceGuard g;
                if (g.enter(a) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->peer_ok = false;
                    continue;
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(b, 0);
                if (pe != hipSuccess) {
                    (void)hipGetLastError();
                    if (pe != hipErrorPeerAccessAlreadyEnabled) ctx->peer_ok = false;
                }
            }
        *out = ctx.release();
        return BSM_OK;)
}

extern "C" int bsm_ctx_destroy(bsm_ctx_t ctx) {
    delete ctx;
    return BSM_OK;
}

extern "C" int bsm_ctx_devices(bsm_ctx_t ctx, int32_t *ndevices, int32_t *device_ids, int32_t capacity) {
    if (!ctx || !ndevices) return fail(BSM_ERR_INVALID, "null argument");
    *ndevices = (int32_t)ctx->devices.size();
    if (device_ids) {
        if (capacity < *ndevices) return fail(BSM_ERR_INVALID, "output buffer too small");
        for (size_t i = 0; i < ctx->devices.size(); i++) device_ids[i] = ctx->devices[i];
    }
    return BSM_OK;
}

extern "C" int bsm_partition_rows(int64_t nrows, int64_t nblocks, const int64_t *rowkey, const int64_t *weight,
                                  int32_t nparts, int32_t *part_of_block, int64_t *own_lo, int64_t *own_hi) {
    BSM_GUARDED(
        if (nparts < 1 || nblocks < 0 || nrows < 0 || !own_lo || !own_hi || (nblocks > 0 && (!rowkey || !weight || !part_of_block)))
            return fail(BSM_ERR_INVALID, "bad argument");
        std::vector<int64_t> key(rowkey, rowkey + nblocks), w(weight, weight + nblocks), lo, hi;
        for (int64_t b = 0; b < nblocks; b++)
            if (key[b] < 1 || key[b] > std::max<int64_t>(nrows, 1)) return fail(BSM_ERR_INVALID, "row key outside the matrix");
        std::vector<int32_t> part;
        partition_rows(nrows, key, w, nparts, part, lo, hi);
        for (int64_t b = 0; b < nblocks; b++) part_of_block[b] = part[b];
        for (int32_t p = 0; p < nparts; p++) {
            own_lo[p] = lo[p];
            own_hi[p] = hi[p];
        }
        return BSM_OK;)
}

namespace {
// triples of one packed image -> device buffers of its device (orow / ocol int64, oval element type)
int export_image(const Analysis &an, const DeviceImage &img, void *orow, void *ocol, void *oval, hipStream_t st) {
    const long long nw = (long long)an.waves.size();
    std::vector<long long> off((size_t)nw + 1, 0);
    for (long long w = 0; w < nw; w++) {
        const WaveWork &W = an.waves[w];
        long long cnt = 0;
        if (W.work == WORK_PANEL && W.npieces > 0) {
            const Piece &P = W.first;
            long long noff = 0;
            if (P.xbase < 0) {
                if ((P.kind & 3) == KIND_OFF)
                    for (int32_t k = 0; k < P.ncols; k++) noff += an.cols[(size_t)P.col_off + k] >= 0;
            } else {
                const long long w1 = std::min<long long>(W.seg1_w, P.ncols), w2 = std::min<long long>(W.seg2_w, P.ncols);
                if ((P.kind & 3) == KIND_OFF) noff += w1;
                if (((P.kind >> 2) & 3) == KIND_OFF) noff += std::max<long long>(0, w2 - w1);
                if (((P.kind >> 4) & 3) == KIND_OFF) noff += std::max<long long>(0, P.ncols - w2);
            }
            cnt = (long long)W.m * (P.ncols + noff);
        }
        off[w + 1] = off[w] + cnt;
    }
    if (off[nw] != an.nnz) return fail(BSM_ERR_DEVICE, "rowcolvals: image / nnz mismatch");
    if (nw == 0 || an.nnz == 0) return BSM_OK;
    void *d_off = nullptr;
    hipError_t e = hipMalloc(&d_off, off.size() * sizeof(long long));
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = launch_export_coo(an.dtype, img.d_waves, nw, d_off, img.d_values, img.d_rows, img.d_cols, orow, ocol, oval, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_off) (void)hipFree(d_off);
    if (e != hipSuccess) return hip_fail(e, "rowcolvals");
    return BSM_OK;
}
}  // namespace

extern "C" int bsm_rowcolvals(bsm_matrix_t A, int64_t *rows, int64_t *cols, void *vals, int64_t *count,
                              int memspace, void *stream) {
    BSM_GUARDED(
        if (!A || !count) return fail(BSM_ERR_INVALID, "null argument");
        if (!rows || !cols || !vals) {
            *count = A->an.nnz;
            return BSM_OK;
        }
        if (*count < A->an.nnz) return fail(BSM_ERR_INVALID, "output buffers too small");
        if (!A->on_device) return fail(BSM_ERR_DEVICE, "handle has no device image (created with BSM_DEVICE_NONE)");
        if (memspace != BSM_MEM_HOST && memspace != BSM_MEM_DEVICE) return fail(BSM_ERR_INVALID, "bad memspace");
        const size_t es = (size_t)A->an.es;
        // one image per device part (a single one for ordinary handles); parts are written one after another
        std::vector<std::pair<const Analysis *, const DeviceImage *>> imgs;
        if (A->dist)
            dist_images(A, imgs);
        else
            imgs.emplace_back(&A->an, &A->img);
        int64_t done = 0;
        for (auto &pi : imgs) {
            const Analysis &an = *pi.first;
            const DeviceImage &img = *pi.second;
            const int64_t n = an.nnz;
            if (n == 0) continue;
            DeviceGuard g;
            hipError_t e = g.enter(img.device);
            if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
            // staged through buffers on the image's device unless the caller's arrays already live there
            bool direct = false;
            if (memspace == BSM_MEM_DEVICE && !A->dist) direct = true;
            void *r = nullptr; void *c = nullptr; void *v = nullptr;
            if (direct) {
                r = rows + done;
                c = cols + done;
                v = (char *)vals + (size_t)done * es;
            } else {
                e = hipMalloc(&r, (size_t)n * 8);
                if (e == hipSuccess) e = hipMalloc(&c, (size_t)n * 8);
                if (e == hipSuccess) e = hipMalloc(&v, (size_t)n * es);
                if (e != hipSuccess) {
                    for (void *q : {r, c, v}) if (q) (void)hipFree(q);
                    return hip_fail(e, "rowcolvals staging");
                }
            }
            int rc = export_image(an, img, r, c, v, direct ? (hipStream_t)stream : nullptr);
            if (rc == BSM_OK && !direct) {
                e = hipMemcpy(rows + done, r, (size_t)n * 8, hipMemcpyDefault);
                if (e == hipSuccess) e = hipMemcpy(cols + done, c, (size_t)n * 8, hipMemcpyDefault);
                if (e == hipSuccess) e = hipMemcpy((char *)vals + (size_t)done * es, v, (size_t)n * es, hipMemcpyDefault);
                if (e != hipSuccess) rc = hip_fail(e, "rowcolvals copy");
            }
            if (!direct) for (void *q : {r, c, v}) (void)hipFree(q);
            if (rc != BSM_OK) return rc;
            done += n;
        }
        *count = done;
        return BSM_OK;)
}

extern "C" int bsm_host_register(void *ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(BSM_ERR_INVALID, "bad argument");
    hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return hip_fail(e, "hipHostRegister");
    return BSM_OK;
}

extern "C" int bsm_host_unregister(void *ptr) {
    if (!ptr) return BSM_OK;
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) return hip_fail(e, "hipHostUnregister");
    return BSM_OK;
}

// A compute stream that leaves CUs to the collective layer.  Measured with the one-rank RCCL loopback of the C5 step
// (tools/loopback_trace.py, profiles/r05_loopback_*.txt): beside a product launch that fills every CU, RCCL's send /
// recv kernel (one large workgroup per channel) found no CU with enough free registers and LDS at once -- 8 us alone,
// 470 us beside the interior launch, i.e. the "overlapped" exchange finished when the product did.  With the product
// on a stream whose CU mask leaves one CU per XCD free the exchange runs beside it (87-137 us, hidden) and the step
// shrinks from 698 to 656 us at 1 % cost for the product.  Mask bit i is CU i / 8 of XCD i % 8 on this part: clearing
// bits in groups of 8 keeps the XCDs equal -- masks that do not (4 CUs of one XCD: 917 instead of 615 us) slow every
// launch, which is why `reserved_cus` is rounded up to a multiple of 8.
extern "C" int bsm_stream_create_reserved(int device, int reserved_cus, void **stream) {
    if (!stream || reserved_cus < 0) return fail(BSM_ERR_INVALID, "bad argument");
    DeviceGuard g;
    hipError_t e = g.enter(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int ncu = 0;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute");
    const int r = std::min((reserved_cus + 7) / 8 * 8, std::max(ncu - 8, 0));
    hipStream_t st = nullptr;
    if (r == 0) {
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    } else {
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int i = r; i < ncu; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
        e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    }
    if (e != hipSuccess) return hip_fail(e, "stream creation");
    *stream = (void *)st;
    return BSM_OK;
}

extern "C" int bsm_stream_destroy(void *stream) {
    if (!stream) return BSM_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamDestroy");
    return BSM_OK;
}

extern "C" int bsm_vec_add_segments(int dtype, void *y, int32_t nseg, const int64_t *offset, const void *const *src,
                                    const int64_t *len, void *stream) {
    if (dtype < 0 || dtype > 3) return fail(BSM_ERR_INVALID, "bad dtype");
    if (nseg < 0 || (nseg > 0 && (!y || !offset || !src || !len))) return fail(BSM_ERR_INVALID, "null argument");
    // disjoint segments only: the launch adds without atomics
    for (int32_t a = 0; a < nseg; a++) {
        if (offset[a] < 0 || len[a] < 0 || (len[a] > 0 && !src[a])) return fail(BSM_ERR_INVALID, "bad segment");
        for (int32_t b = a + 1; b < nseg; b++)
            if (len[a] > 0 && len[b] > 0 && offset[a] < offset[b] + len[b] && offset[b] < offset[a] + len[a])
                return fail(BSM_ERR_INVALID, "segments overlap");
    }
    for (int32_t c = 0; c < nseg;) {  // kMaxVecPieces non-empty segments per launch
        VecPieces pc;
        int np = 0;
        for (; c < nseg && np < kMaxVecPieces; c++) {
            if (len[c] == 0) continue;
            pc.base[np] = src[c];
            pc.lo[np] = offset[c];
            pc.hi[np] = offset[c] + len[c];
            pc.strided[np] = 0;
            np++;
        }
        if (np == 0) break;
        hipError_t e = launch_vec_add_segments(dtype, y, pc, np, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "segment add launch");
    }
    return BSM_OK;
}

extern "C" int bsm_part_info(bsm_matrix_t A, int32_t part, bsm_part_info_t *out) {
    if (!A || !out) return fail(BSM_ERR_INVALID, "null argument");
    if (!A->dist) return fail(BSM_ERR_INVALID, "not a multi-device handle");
    return dist_part_info(A, part, out);
}

// The gather workspace of an image (column sums + inverted indices) belongs to the handle and admits
// ONE product in flight.  The enqueue is serialised; a caller that races on the same handle from
// another thread, or whose predecessor may still be running on ANOTHER stream (same stream: stream
// order protects it), does not get the claim and its product takes the atomic path.  Nothing is
// enqueued for the bookkeeping (an event recorded per product costs 3 us between two 9 us launches):
// the previous stream is queried only when the stream changes.  A product enqueued while the stream is
// being captured into a graph never gets the claim (atomic path): a replay could meet an eager product.
struct WorkspaceClaim {
    bsm_matrix_s *A;
    hipStream_t st;
    std::unique_lock<std::mutex> lock;
    bool held = false, track = false;
    WorkspaceClaim(bsm_matrix_s *A_, const DeviceImage &img, hipStream_t st_)
        : A(A_), st(st_), lock(A_->gather_mu, std::defer_lock) {
        held = img.d_ws != nullptr && lock.try_lock();
        if (!held) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
            (void)hipGetLastError();
            cs = hipStreamCaptureStatusNone;
        }
        if (cs != hipStreamCaptureStatusNone) {
            // a captured product would use the workspace at every replay, on whatever stream, beside eager
            // products nobody can order against: captured products take the atomic path
            held = track = false;
            return;
        }
        track = true;
        if (A->ws_pending && A->ws_stream == st) return;  // the common case: one stream, nothing to ask
        if (A->ws_pending) {  // the stream changed: is the previous one idle?
            const hipError_t q = hipStreamQuery(A->ws_stream);
            if (q != hipSuccess) {
                (void)hipGetLastError();  // hipErrorNotReady is not a failure
                // (any other answer -- the stream may be gone -- is treated the same way once, then
                // forgotten: work of a destroyed stream does not outlive a whole product by much)
                if (q != hipErrorNotReady) A->ws_pending = false;
                held = track = false;
            }
        }
    }
    void mark() {  // after the product has been enqueued
        if (!held || !track) return;
        A->ws_stream = st;
        A->ws_pending = true;
    }
};

// The work arrays of the interleaved multi-RHS pass (Xr, W: 128 bytes per vector entry each) belong to the handle like the
// gather workspace, with the same rules: one product in flight -- a racing thread, a predecessor that may still run on
// ANOTHER stream, or a stream under graph capture do not get the claim and their product takes the ordinary kernels.
// Allocated (and grown) here, at the first product that uses them.
struct ILClaim {
    bsm_matrix_s *A;
    hipStream_t st;
    std::unique_lock<std::mutex> lock;
    bool held = false;
    ILClaim(bsm_matrix_s *A_, const DeviceImage &img, bool opT, long long nrhs, hipStream_t st_)
        : A(A_), st(st_), lock(A_->il_mu, std::defer_lock) {
        if (!il_applies(img, opT, nrhs) || !lock.try_lock()) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
            (void)hipGetLastError();
            cs = hipStreamCaptureStatusNone;
        }
        if (cs != hipStreamCaptureStatusNone) return;
        if (A->il_pending && A->il_stream != st) {  // the stream changed: is the previous one idle?
            const hipError_t q = hipStreamQuery(A->il_stream);
            if (q != hipSuccess) {
                (void)hipGetLastError();
                if (q != hipErrorNotReady) A->il_pending = false;
                return;
            }
            A->il_pending = false;
        }
        const long long need = std::max(img.nrows, img.ncols);
        if (A->il.rows < need) {
            if (A->il_pending && hipStreamSynchronize(A->il_stream) != hipSuccess) (void)hipGetLastError();
            A->il_pending = false;
            if (A->il.xr) (void)hipFree(A->il.xr);
            if (A->il.w) (void)hipFree(A->il.w);
            A->il = ILWork{};
            void *xr = nullptr, *w = nullptr;
            if (hipMalloc(&xr, (size_t)need * 128) != hipSuccess || hipMalloc(&w, (size_t)need * 128) != hipSuccess) {
                (void)hipGetLastError();
                if (xr) (void)hipFree(xr);
                return;  // no memory for the work arrays: the ordinary kernels need none
            }
            A->il.xr = xr;
            A->il.w = w;
            A->il.rows = need;
            A->il.w_clean = false;
        }
        held = true;
    }
    ILWork *work() { return held ? &A->il : nullptr; }
    void mark() {
        if (!held) return;
        A->il_stream = st;
        A->il_pending = true;
    }
};

extern "C" int bsm_mul(bsm_matrix_t A, int op, const void *x, void *y, const void *alpha,
                       const void *beta, int beta_strong_zero, int memspace, void *stream) {
    if (!A) return fail(BSM_ERR_INVALID, "null handle");
    if (op < 0 || op > 2) return fail(BSM_ERR_INVALID, "bad op");
    if (!x || !y) return fail(BSM_ERR_INVALID, "null vector");
    if (!A->on_device)
        return fail(BSM_ERR_DEVICE, "handle has no device image (created with BSM_DEVICE_NONE)");
    if (A->dist) return dist_mul(A, op, x, y, alpha, beta, beta_strong_zero, memspace, (hipStream_t)stream);
    // transposed products run forward on the second ordering when the handle has one
    const bool use_t = (op != BSM_OP_N) && A->has_t;
    const DeviceImage &img = use_t ? A->img_t : A->img;
    const bool opT = (op != BSM_OP_N) && !use_t;
    const bool conj = (op == BSM_OP_C);
    DeviceGuard guard;
    hipError_t e = guard.enter(img.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipStream_t st = (hipStream_t)stream;
    WorkspaceClaim claim(A, img, st);
    const bool use_gather = claim.held;
    auto mark_gather = [&]() { claim.mark(); };
    if (memspace == BSM_MEM_DEVICE) {
        e = launch_mul(img, opT, conj, x, y, alpha, beta, beta_strong_zero, st, use_gather);
        if (e != hipSuccess) return hip_fail(e, "kernel launch");
        mark_gather();
        return BSM_OK;
    }
    if (memspace != BSM_MEM_HOST) return fail(BSM_ERR_INVALID, "bad memspace");
    // host vectors: stage through device buffers (PCIe), synchronous
    const size_t es = (size_t)A->an.es;
    const size_t xlen = (size_t)(op == 0 ? A->img.ncols : A->img.nrows);
    const size_t ylen = (size_t)(op == 0 ? A->img.nrows : A->img.ncols);
    Staging sg;
    e = sg.acquire(A, xlen * es, ylen * es);
    void *dx = sg.dx, *dy = sg.dy;
    // the incoming y travels when beta uses it -- and whenever the handle owns only a row range: rows
    // outside it that no block reaches are left untouched by the product and must come back unchanged
    const bool partial = (op == BSM_OP_N) && (img.own_lo > 0 || img.own_hi < img.nrows);
    const bool y_in = !beta_strong_zero || partial;
    // (page-locked vectors -- bsm_host_register -- make both copies true DMA; pageable ones are staged
    // by the runtime: 74 vs 118 us per C2-sized product)
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, xlen * es, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && y_in) e = hipMemcpyAsync(dy, y, ylen * es, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_mul(img, opT, conj, dx, dy, alpha, beta, beta_strong_zero, st, use_gather);
    if (e == hipSuccess) e = hipMemcpyAsync(y, dy, ylen * es, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "host-staged mul");
    return BSM_OK;
}

extern "C" int bsm_mul_parts(bsm_matrix_t A, int op, const void *const *x_parts, void *const *y_parts,
                             const void *alpha, const void *beta, int beta_strong_zero, void *const *streams) {
    if (!A) return fail(BSM_ERR_INVALID, "null handle");
    if (op < 0 || op > 2) return fail(BSM_ERR_INVALID, "bad op");
    if (!x_parts || !y_parts) return fail(BSM_ERR_INVALID, "null vector parts");
    if (!A->dist) return fail(BSM_ERR_INVALID, "bsm_mul_parts needs a multi-device handle (bsm_options.ctx)");
    return dist_mul_parts(A, op, x_parts, y_parts, alpha, beta, beta_strong_zero, streams);
}

extern "C" int bsm_mul_multi(bsm_matrix_t A, int op, int64_t nrhs, const void *X, int64_t ldx, void *Y,
                             int64_t ldy, const void *alpha, const void *beta, int beta_strong_zero,
                             int memspace, void *stream) {
    if (!A) return fail(BSM_ERR_INVALID, "null handle");
    if (op < 0 || op > 2) return fail(BSM_ERR_INVALID, "bad op");
    if (nrhs < 0) return fail(BSM_ERR_INVALID, "negative nrhs");
    if (nrhs == 0) return BSM_OK;
    if (!X || !Y) return fail(BSM_ERR_INVALID, "null matrix");
    if (!A->on_device)
        return fail(BSM_ERR_DEVICE, "handle has no device image (created with BSM_DEVICE_NONE)");
    // (the whole operator's size: a multi-device handle has no image of its own)
    const long long xlen = (op == 0 ? A->an.ncols : A->an.nrows);
    const long long ylen = (op == 0 ? A->an.nrows : A->an.ncols);
    if (ldx < std::max<long long>(xlen, 1) || ldy < std::max<long long>(ylen, 1))
        return fail(BSM_ERR_INVALID, "leading dimension smaller than the vector length");
    if (A->dist)  // multi-device handles: every device streams its part once per batch of <= 8 columns
        return dist_mul_multi(A, op, nrhs, X, ldx, Y, ldy, alpha, beta, beta_strong_zero, memspace, (hipStream_t)stream);
    const bool use_t = (op != BSM_OP_N) && A->has_t;
    const DeviceImage &img = use_t ? A->img_t : A->img;
    const bool opT = (op != BSM_OP_N) && !use_t;
    const bool conj = (op == BSM_OP_C);
    DeviceGuard guard;
    hipError_t e = guard.enter(img.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipStream_t st = (hipStream_t)stream;
    ILClaim il(A, img, opT, nrhs, st);
    if (memspace == BSM_MEM_DEVICE) {
        e = launch_mul_multi(img, opT, conj, nrhs, X, ldx, Y, ldy, alpha, beta, beta_strong_zero, st, nullptr, il.work());
        if (e != hipSuccess) return hip_fail(e, "kernel launch");
        il.mark();
        return BSM_OK;
    }
    if (memspace != BSM_MEM_HOST) return fail(BSM_ERR_INVALID, "bad memspace");
    const size_t es = (size_t)A->an.es;
    Staging sg;
    e = sg.acquire(A, (size_t)xlen * nrhs * es, (size_t)ylen * nrhs * es);
    void *dx = sg.dx, *dy = sg.dy;
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(dx, (size_t)xlen * es, X, (size_t)ldx * es, (size_t)xlen * es, (size_t)nrhs,
                             hipMemcpyHostToDevice, st);
    const bool partial = (op == BSM_OP_N) && (img.own_lo > 0 || img.own_hi < img.nrows);  // see bsm_mul
    if (e == hipSuccess && (!beta_strong_zero || partial))
        e = hipMemcpy2DAsync(dy, (size_t)ylen * es, Y, (size_t)ldy * es, (size_t)ylen * es, (size_t)nrhs,
                             hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = launch_mul_multi(img, opT, conj, nrhs, dx, xlen, dy, ylen, alpha, beta, beta_strong_zero, st, nullptr, il.work());
    il.mark();
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(Y, (size_t)ldy * es, dy, (size_t)ylen * es, (size_t)ylen * es, (size_t)nrhs,
                             hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "host-staged multi mul");
    return BSM_OK;
}

static int copy_out(const std::vector<int64_t> &v, int64_t *out, int64_t *len) {
    if (!len) return fail(BSM_ERR_INVALID, "len is null");
    if (out) {
        if (*len < (int64_t)v.size()) return fail(BSM_ERR_INVALID, "output buffer too small");
        std::memcpy(out, v.data(), v.size() * sizeof(int64_t));
    }
    *len = (int64_t)v.size();
    return BSM_OK;
}

extern "C" int bsm_get_bookkeeping(bsm_matrix_t A, int which, int64_t *out, int64_t *len) {
    if (!A) return fail(BSM_ERR_INVALID, "null handle");
    const Analysis &an = A->an;
    switch (which) {
        case BSM_BK_VBCRS_PERM:
        case BSM_BK_VBCRS_ROWPTR:
        case BSM_BK_VBCRS_COLINDICES:
        case BSM_BK_VBCRS_ROWINDICES: {
            if (an.mtype != MT_VBCRS) return fail(BSM_ERR_INVALID, "not a VBCRS handle");
            const std::vector<int64_t> *v = which == BSM_BK_VBCRS_PERM         ? &an.perm
                                            : which == BSM_BK_VBCRS_ROWPTR     ? &an.rowptr
                                            : which == BSM_BK_VBCRS_COLINDICES ? &an.colindices
                                                                               : &an.rowindices;
            return copy_out(*v, out, len);
        }
        case BSM_BK_COLORS:
        case BSM_BK_TRANSPOSECOLORS:
        case BSM_BK_DIAGONALCOLORS: {
            if (an.mtype == MT_VBCRS) return fail(BSM_ERR_INVALID, "VBCRS has no colours");
            if (which == BSM_BK_DIAGONALCOLORS && an.mtype != MT_SYMMETRIC)
                return fail(BSM_ERR_INVALID, "not a symmetric handle");
            const auto &cs = an.colors[which - BSM_BK_COLORS];
            std::vector<int64_t> flat;
            flat.push_back((int64_t)cs.size());
            for (const auto &c : cs) {
                flat.push_back((int64_t)c.size());
                flat.insert(flat.end(), c.begin(), c.end());
            }
            return copy_out(flat, out, len);
        }
    }
    return fail(BSM_ERR_INVALID, "unknown bookkeeping id");
}

extern "C" int bsm_get_image(bsm_matrix_t A, int which, void *out, int64_t *nbytes) {
    if (!A || !nbytes) return fail(BSM_ERR_INVALID, "null argument");
    if (A->on_device || A->dist) return fail(BSM_ERR_UNSUPPORTED, "image dump needs an analysis-only handle");
    if (which >= 16 && !A->has_t) return fail(BSM_ERR_INVALID, "handle has no transposed image");
    const Analysis &an = (which >= 16) ? A->an_t : A->an;
    which &= 15;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = an.values.data(); bytes = an.values.size(); break;
        case 1: src = an.rows.data(); bytes = an.rows.size() * 4; break;
        case 2: src = an.cols.data(); bytes = an.cols.size() * 4; break;
        case 3: src = an.waves.data(); bytes = an.waves.size() * sizeof(WaveWork); break;
        case 4: src = an.inv_ptr[0].data(); bytes = an.inv_ptr[0].size() * 8; break;
        case 5: src = an.inv_idx[0].data(); bytes = an.inv_idx[0].size() * 4; break;
        case 6: src = an.inv_ptr[1].data(); bytes = an.inv_ptr[1].size() * 8; break;
        case 7: src = an.inv_idx[1].data(); bytes = an.inv_idx[1].size() * 4; break;
        case 8: src = an.waves_multi.data(); bytes = an.waves_multi.size() * sizeof(WaveWork); break;
        default: return fail(BSM_ERR_INVALID, "unknown image array");
    }
    if (out) {
        if (*nbytes < (int64_t)bytes) return fail(BSM_ERR_INVALID, "output buffer too small");
        std::memcpy(out, src, bytes);
    }
    *nbytes = (int64_t)bytes;
    return BSM_OK;
}

// Developer probe (tools/placement_move.py; not declared in the public header): moves one array of a device image
// to a fresh allocation -- which = 0 values, 1 rows, 2 cols, 3 wave records -- to find out which one a handle's
// "placement level" depends on.
extern "C" int bsm_debug_move_image_array(bsm_matrix_t A, int which) {
    if (!A || !A->on_device || A->dist) return fail(BSM_ERR_INVALID, "needs a single-device handle");
    DeviceGuard guard;
    hipError_t e = guard.enter(A->img.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    void **slot = which == 0 ? &A->img.d_values : which == 1 ? &A->img.d_rows : which == 2 ? &A->img.d_cols : &A->img.d_waves;
    const size_t bytes = which == 0 ? (size_t)A->an.value_bytes
                         : which == 1 ? A->an.rows.size() * 4
                         : which == 2 ? A->an.cols.size() * 4
                                      : A->an.waves.size() * sizeof(WaveWork);
    if (!*slot || bytes == 0) return BSM_OK;
    void *fresh = nullptr;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "sync");
    if ((e = hipMalloc(&fresh, bytes)) != hipSuccess) return hip_fail(e, "hipMalloc");
    if ((e = hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice)) != hipSuccess) return hip_fail(e, "copy");
    (void)hipFree(*slot);
    *slot = fresh;
    if (std::getenv("BSM_PLACEMENT_DEBUG")) std::fprintf(stderr, "[bsm image] array %d now at %p (%zu B)\n", which, fresh, bytes);
    return BSM_OK;
}

extern "C" int bsm_stats(bsm_matrix_t A, bsm_stats_t *out) {
    if (!A || !out) return fail(BSM_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    out->nnz = A->an.nnz;
    out->stored_entries = A->an.stored_entries;
    out->alg_bytes = A->an.alg_bytes;
    out->device_bytes = A->dist ? dist_device_bytes(A) : A->img.device_bytes + (A->has_t ? A->img_t.device_bytes : 0);
    out->npanels = A->an.ngroups;
    out->ntasks = (int64_t)A->an.waves.size();
    out->nworkgroups = A->img.nwg_total ? A->img.nwg_total : A->an.nwg_total;
    out->exclusive = A->img.exclusive_fwd ? 1 : 0;
    out->win_emissions = A->an.win_emissions;
    out->win_inside = A->an.win_inside;
    out->win_flushed = A->an.win_flushed;
    return BSM_OK;
}

extern "C" int bsm_color(int64_t nlists, const int64_t *const *lists, const int64_t *lens, int algorithm,
                         int64_t *color_out, int64_t *ncolors) {
    try {
        if (nlists < 0 || (nlists > 0 && (!lists || !lens || !color_out)) || !ncolors)
            return fail(BSM_ERR_INVALID, "null argument");
        std::vector<const int64_t *> lp((size_t)nlists);
        std::vector<int64_t> ln((size_t)nlists);
        for (int64_t b = 0; b < nlists; b++) {
            if (lens[b] < 0 || (lens[b] > 0 && !lists[b])) return fail(BSM_ERR_INVALID, "bad list");
            for (int64_t k = 0; k < lens[b]; k++)
                if (lists[b][k] < 1) return fail(BSM_ERR_INVALID, "indices are 1-based");
            lp[b] = lists[b];
            ln[b] = lens[b];
        }
        if (algorithm != BSM_COLOR_WORKSTREAM_DSATUR && algorithm != BSM_COLOR_DSATUR)
            return fail(BSM_ERR_INVALID, "unknown colouring algorithm");
        auto classes = algorithm == BSM_COLOR_DSATUR ? color_dsatur(lp, ln) : color_workstream_dsatur(lp, ln);
        for (size_t c = 0; c < classes.size(); c++)
            for (int64_t id : classes[c]) color_out[id - 1] = (int64_t)c;
        *ncolors = (int64_t)classes.size();
        return BSM_OK;
    } catch (const std::bad_alloc &) {
        return fail(BSM_ERR_ALLOC, "out of host memory");
    }
}

extern "C" int bsm_destroy(bsm_matrix_t A) {
    if (!A) return BSM_OK;
    if (A->dist) {
        dist_destroy(A);
    } else if (A->on_device) {
        DeviceGuard guard;
        (void)guard.enter(A->img.device);
        free_image(A->img);
        free_image(A->img_t);
        if (A->stage_x) (void)hipFree(A->stage_x);
        if (A->stage_y) (void)hipFree(A->stage_y);
        if (A->il.xr) (void)hipFree(A->il.xr);
        if (A->il.w) (void)hipFree(A->il.w);
    }
    delete A;
    return BSM_OK;
}
