// bsm_dist.cpp -- ONE handle spread over the devices of a bsm_ctx_t (include/bsm_rocm.h).
//
// The reference fans its block rows / colour classes out as tasks of one process
// (`@tasks for ...` with the scheduler stored in the matrix: src/vbcrs.jl:275-276,
// src/blockmatrix.jl:233-245, src/symmetricblockmatrix.jl:395-432).  Here the same call fans out
// over the GPUs of one node: contiguous ranges of block rows per device (balanced by stored
// entries), one stream per device, and only the y segments a device produced for rows of ANOTHER
// device travel -- as direct peer-to-peer copies over the xGMI links followed by a local add.
// One host thread issues everything; every step is asynchronous and ordered by events.
#include <algorithm>
#include <atomic>
#include <complex>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <numeric>
#include <thread>

#include "bsm_internal.h"

namespace bsm {

namespace {

struct Range {  // 0-based [lo, hi)
    long long lo = 0, hi = 0;
    bool empty() const { return hi <= lo; }
    long long len() const { return hi > lo ? hi - lo : 0; }
};
Range isect(Range a, Range b) { return Range{std::max(a.lo, b.lo), std::min(a.hi, b.hi)}; }
Range hull(Range a, Range b) {
    if (a.empty()) return b;
    if (b.empty()) return a;
    return Range{std::min(a.lo, b.lo), std::max(a.hi, b.hi)};
}

struct Transfer {  // part `from` produced y[range] for rows part `to` owns
    int from, to;
    Range range;
    size_t recv_off;  // byte offset in the receiver's staging buffer
};

// everything that depends on the direction of the product relative to the partition
struct Plan {
    std::vector<Range> xr;    // x entries part p reads
    std::vector<Range> in;    // x entries part p HOLDS when the vectors are partitioned (bsm_mul_parts): a tiling of [0, xlen)
    std::vector<Range> out;   // y entries part p delivers (a tiling of [0, ylen))
    std::vector<Range> zr;    // y entries part p must define in its work vector (touched + out)
    std::vector<Transfer> transfers;
    std::vector<size_t> recv_bytes;
};

}  // namespace

struct Part {
    int device = 0;
    Analysis an;
    DeviceImage img;
    bool has_image = false;
    int64_t nblocks = 0;
    Range own;         // rows it owns
    Range rows, cols;  // hull of the row / column indices of its blocks
    hipStream_t stream = nullptr;
    hipEvent_t ev_prod = nullptr, ev_done = nullptr, ev_in = nullptr;
    hipStream_t done_stream = nullptr;  // where ev_done (or its flag) was last recorded, on `device`
    bool w_clean = false;               // d_w is zero everywhere (DistState::rezero)
    void *d_x = nullptr, *d_w = nullptr, *d_recv = nullptr;
    Range colpart;  // the part's share of the COLUMN partition (vectors of length ncols held in parts)
    // work arrays of the interleaved multi-RHS pass (bsm_kernels.h: ILWork) of this part's image, allocated at the first
    // multi-RHS product that takes it.  A part's successive products are ordered by the fan-out itself (a product waits for
    // the part's own previous delivery, or everything runs on one stream), which is all the arrays need.
    ILWork il;
    bool il_failed = false;  // no memory for them: the ordinary kernels from then on
};

// the part's work arrays if its image / this product take the interleaved pass (and they can be had), else null
static ILWork *part_il(Part &pt, bool opT, int K) {
    if (pt.il_failed || !il_applies(pt.img, opT, K)) return nullptr;
    const long long need = std::max(pt.img.nrows, pt.img.ncols);
    if (pt.il.rows < need) {
        void *xr = nullptr, *w = nullptr;
        if (hipMalloc(&xr, (size_t)need * 128) != hipSuccess || hipMalloc(&w, (size_t)need * 128) != hipSuccess) {
            (void)hipGetLastError();
            if (xr) (void)hipFree(xr);
            pt.il_failed = true;
            return nullptr;
        }
        pt.il.xr = xr;
        pt.il.w = w;
        pt.il.rows = need;
        pt.il.w_clean = false;
    }
    return &pt.il;
}

// One persistent thread per part, bound to the part's device once.  run(f) executes f(p) on every
// worker and returns when all are done (first failure wins; its message is handed to the caller's
// thread-local error slot).
class Workers {
  public:
    explicit Workers(const std::vector<int> &devices) : n_((int)devices.size()), rc_(devices.size(), BSM_OK), msg_(devices.size()) {
        for (int p = 0; p < n_; p++) th_.emplace_back([this, p, dev = devices[p]] { loop(p, dev); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
            gen_++;
        }
        go_.notify_all();
        for (auto &t : th_) t.join();
    }
    int run(const std::function<int(int)> &f) {
        std::unique_lock<std::mutex> l(mu_);
        fn_ = &f;
        pending_ = n_;
        gen_++;
        go_.notify_all();
        done_.wait(l, [&] { return pending_ == 0; });
        fn_ = nullptr;
        for (int p = 0; p < n_; p++)
            if (rc_[p] != BSM_OK) return fail(rc_[p], msg_[p]);
        return BSM_OK;
    }

  private:
    void loop(int p, int dev) {
        (void)hipSetDevice(dev);
        uint64_t seen = 0;
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            go_.wait(l, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            const std::function<int(int)> *f = fn_;
            l.unlock();
            int rc = BSM_ERR_DEVICE;
            std::string msg;
            try {
                rc = (*f)(p);
                if (rc != BSM_OK) msg = bsm_last_error();  // this thread's slot
            } catch (const std::exception &e) {
                msg = e.what();
            }
            l.lock();
            rc_[p] = rc;
            msg_[p] = msg;
            if (--pending_ == 0) done_.notify_all();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable go_, done_;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
    const std::function<int(int)> *fn_ = nullptr;
    std::vector<int> rc_;
    std::vector<std::string> msg_;
};

struct DistState {
    std::unique_ptr<Workers> workers;  // more than two parts: one issuing thread per device
    bsm_ctx_s *ctx = nullptr;
    int dtype = 1, es = 8;
    long long nrows = 0, ncols = 0;
    bool symmetric = false;  // op(A) has A's structure: every op runs ALONG the partition
    std::vector<std::unique_ptr<Part>> parts;
    Plan plan_n, plan_t;
    size_t vlen = 0;  // elements per column of the parts' x / work buffers
    int kcap = 1;     // columns those buffers hold (grows with bsm_mul_multi)
    std::mutex mu;  // one product in flight per handle: the work vectors belong to the handle
    std::map<int, hipEvent_t> ev_x;  // "x and y are ready" on the caller's stream, one event per device
    void *d_res = nullptr;  // numeric beta, y on a device: segments of remote parts wait here
    int res_dev = -1;
    size_t res_bytes = 0;
    std::vector<char> h_res;  // numeric beta, y on the host
    // every pair of the context's devices can address each other's memory: the vector traffic of a product
    // with device-resident vectors runs as two fused kernels per device (fetch / finish) instead of copies
    bool all_peer = false;
    bool produced = false;       // some product has been issued: ev_done / ev_tail carry a record
    hipEvent_t ev_tail = nullptr;  // end of the last copy-path product on its caller's stream
    int tail_dev = -1;
    // Cross-stream ordering of the fused path by FLAGS instead of events (BSM_DIST_FLAGS): a product has a sequence
    // number, "recorded" = hipStreamWriteValue64(stream, flag, seq), "waited for" = hipStreamWaitValue64(stream, flag,
    // seq, >=) on 64-bit counters in coherent pinned host memory (every device's command processor can poll them).
    // tools/hop_latency.hip: a dependency between two streams costs ~6 us this way against ~12.4 us through
    // hipEventRecord + hipStreamWaitEvent, and a product has two to three of them on its critical path.
    // flag[k * P + p], k = 0 "inputs of part p ready", 1 "product of part p done", 2 "delivery of part p done";
    // flag[3 P] "x / y of a full-vector call ready".
    bool use_flags = false;
    uint64_t *flags = nullptr;
    uint64_t seq = 0;
    // 0 events, 1 flags, 2 nothing at all (every part AND the caller's vectors on one stream of one device: stream order
    // is the order): a change drains every device first (the forms do not see each other)
    int last_mode = -1;
    hipStream_t last_single_stream = nullptr;  // mode 2: the stream (a different one next time drains as well)
    // The work vectors of the fused path are ZERO between two products: the finish kernels write zeros behind what they
    // read (every entry a product writes is read exactly once -- by its owner's finish kernel or by the peer it is
    // for), so a product accumulates into its work vector without the `w = 0` launch in front (one dependent launch
    // per part off the critical path).  Part::w_clean = false: unknown contents (fresh buffers, a copy-path product,
    // a failed product) -- the next fused product that uses the vector clears the whole buffer once.
    bool rezero = true;  // BSM_DIST_REZERO=0: the zeroing launch in front of every product, as before (A/B, diagnosis)
    bool one_stream = true;  // BSM_DIST_ONE_STREAM=0: parts that share the caller's device still get streams of their own
};

}  // namespace bsm

// (defined here: DistState must be complete where the handle's unique_ptr is destroyed)
bsm_matrix_s::bsm_matrix_s() = default;
bsm_matrix_s::~bsm_matrix_s() = default;

namespace bsm {

void block_row_keys(const std::vector<BlockIn> &in, std::vector<int64_t> &key, std::vector<int64_t> &weight) {
    key.resize(in.size());
    weight.resize(in.size());
    for (size_t b = 0; b < in.size(); b++) {
        const BlockIn &B = in[b];
        int64_t k = B.r0 > 0 ? B.r0 : 1;
        if (B.ridx && B.m > 0) {
            k = B.ridx[0];
            for (int64_t i = 1; i < B.m; i++) k = std::min(k, B.ridx[i]);
        }
        key[b] = k;
        weight[b] = B.m * B.n;
    }
}

void partition_rows(int64_t nrows, const std::vector<int64_t> &key, const std::vector<int64_t> &weight,
                    int nparts, std::vector<int32_t> &part_of_block, std::vector<int64_t> &own_lo,
                    std::vector<int64_t> &own_hi) {
    const size_t nb = key.size();
    std::vector<int64_t> uk(key);
    std::sort(uk.begin(), uk.end());
    uk.erase(std::unique(uk.begin(), uk.end()), uk.end());
    const size_t nk = uk.size();
    std::vector<double> csum(nk + 1, 0.0);
    std::vector<size_t> kidx(nb);
    for (size_t b = 0; b < nb; b++) {
        kidx[b] = (size_t)(std::lower_bound(uk.begin(), uk.end(), key[b]) - uk.begin());
        csum[kidx[b] + 1] += (double)std::max<int64_t>(weight[b], 0);
    }
    for (size_t k = 0; k < nk; k++) csum[k + 1] += csum[k];
    const double total = csum[nk];
    std::vector<size_t> cut((size_t)nparts + 1, 0);
    for (int p = 1; p < nparts; p++) {
        const double target = total * p / nparts;
        size_t k = (size_t)(std::lower_bound(csum.begin(), csum.end(), target) - csum.begin());
        cut[p] = std::min(std::max(k, cut[p - 1]), nk);
    }
    cut[nparts] = nk;
    part_of_block.assign(nb, 0);
    {
        std::vector<int32_t> part_of_key(nk, 0);
        for (int p = 0; p < nparts; p++)
            for (size_t k = cut[p]; k < cut[p + 1]; k++) part_of_key[k] = p;
        for (size_t b = 0; b < nb; b++) part_of_block[b] = part_of_key[kidx[b]];
    }
    own_lo.assign(nparts, 1);
    own_hi.assign(nparts, 0);
    int first = -1, last = -1;
    for (int p = 0; p < nparts; p++) {
        if (cut[p] == cut[p + 1]) {  // no key: an empty range just below the next part's first row
            own_lo[p] = cut[p] < nk ? uk[cut[p]] : nrows + 1;
            own_hi[p] = own_lo[p] - 1;
            continue;
        }
        if (first < 0) first = p;
        last = p;
        own_lo[p] = uk[cut[p]];
        own_hi[p] = cut[p + 1] < nk ? uk[cut[p + 1]] - 1 : nrows;
    }
    if (first >= 0) {  // rows in front of the first key / behind the last block row belong to somebody
        own_lo[first] = 1;
        own_hi[last] = nrows;
    } else if (nparts > 0) {  // no block at all: part 0 owns (and scales) every row
        own_lo[0] = 1;
        own_hi[0] = nrows;
    }
}

namespace {

void block_hulls(const std::vector<BlockIn> &sub, Range &rows, Range &cols) {
    rows = cols = Range{};
    for (const BlockIn &B : sub) {
        if (B.m <= 0 || B.n <= 0) continue;
        int64_t rl, rh, cl, ch;
        if (B.ridx) {
            rl = rh = B.ridx[0];
            for (int64_t i = 1; i < B.m; i++) rl = std::min(rl, B.ridx[i]), rh = std::max(rh, B.ridx[i]);
        } else {
            rl = B.r0;
            rh = B.r0 + B.m - 1;
        }
        if (B.cidx) {
            cl = ch = B.cidx[0];
            for (int64_t i = 1; i < B.n; i++) cl = std::min(cl, B.cidx[i]), ch = std::max(ch, B.cidx[i]);
        } else {
            cl = B.c0;
            ch = B.c0 + B.n - 1;
        }
        rows = hull(rows, Range{rl - 1, rh});
        cols = hull(cols, Range{cl - 1, ch});
    }
}

// transfers + receive-buffer layout of a plan whose xr / out / touched ranges are known
void finish_plan(Plan &pl, const std::vector<Range> &touched, int es) {
    const int P = (int)pl.out.size();
    pl.zr.resize(P);
    pl.recv_bytes.assign(P, 0);
    for (int p = 0; p < P; p++) pl.zr[p] = hull(touched[p], pl.out[p]);
    for (int q = 0; q < P; q++)
        for (int p = 0; p < P; p++) {
            if (p == q) continue;
            const Range o = isect(touched[p], pl.out[q]);
            if (o.empty()) continue;
            pl.transfers.push_back(Transfer{p, q, o, pl.recv_bytes[q]});
            pl.recv_bytes[q] += ((size_t)o.len() * es + 255) & ~(size_t)255;
        }
}

hipError_t copy_between(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (ddev == sdev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st);  // xGMI, device to device
}

template <typename T> void host_axpby(T *y, const T *r, long long n, T beta) {
    for (long long i = 0; i < n; i++) y[i] = beta * y[i] + r[i];
}

int pointer_device(const void *p, int fallback) {
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return fallback;
    }
    return at.device;
}

}  // namespace

void dist_destroy(bsm_matrix_s *A) {
    if (!A->dist) return;
    DistState &D = *A->dist;
    D.workers.reset();
    for (auto &pp : D.parts) {
        Part &p = *pp;
        DeviceGuard g;
        (void)g.enter(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        free_image(p.img);
        for (void *q : {p.d_x, p.d_w, p.d_recv, p.il.xr, p.il.w})
            if (q) (void)hipFree(q);
        if (p.ev_prod) (void)hipEventDestroy(p.ev_prod);
        if (p.ev_done) (void)hipEventDestroy(p.ev_done);
        if (p.ev_in) (void)hipEventDestroy(p.ev_in);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    if (D.ev_tail) {
        DeviceGuard g;
        (void)g.enter(D.tail_dev);
        (void)hipEventDestroy(D.ev_tail);
    }
    for (auto &kv : D.ev_x) {
        DeviceGuard g;
        (void)g.enter(kv.first);
        (void)hipEventDestroy(kv.second);
    }
    if (D.d_res) {
        DeviceGuard g;
        (void)g.enter(D.res_dev);
        (void)hipFree(D.d_res);
    }
    if (D.flags) (void)hipHostFree(D.flags);
    A->dist.reset();
}

int dist_create(bsm_matrix_s *A, bsm_ctx_s *ctx, int mtype, int dtype, int64_t nrows, int64_t ncols,
                const std::vector<BlockIn> &in, const bsm_options &o) {
    const int P = (int)ctx->devices.size();
    if (P < 1) return fail(BSM_ERR_INVALID, "context has no device");
    A->dist.reset(new DistState());
    DistState &D = *A->dist;
    D.ctx = ctx;
    D.dtype = dtype;
    D.es = A->an.es;
    D.nrows = nrows;
    D.ncols = ncols;
    D.symmetric = (mtype == MT_SYMMETRIC);
    for (const BlockIn &B : in) D.symmetric |= (B.kind != KIND_PLAIN);  // symmetric view of a VBCRS

    std::vector<int64_t> key, weight, own_lo, own_hi;
    std::vector<int32_t> part_of;
    block_row_keys(in, key, weight);
    partition_rows(nrows, key, weight, P, part_of, own_lo, own_hi);

    std::vector<Range> touched_n(P), touched_t(P);
    D.plan_n.xr.resize(P);
    D.plan_n.out.resize(P);
    D.plan_t.xr.resize(P);
    D.plan_t.out.resize(P);
    const long long chunk_t = (ncols + P - 1) / P;
    // the COLUMN partition (what a part holds of a vector of length ncols when the vectors are partitioned,
    // and what it delivers of a product across the row partition): the row partition itself for square
    // operators -- y of one product is x of the next, part by part -- else equal chunks
    const bool square = (nrows == ncols);
    D.plan_n.in.resize(P);
    D.plan_t.in.resize(P);
    std::vector<std::vector<BlockIn>> subs(P);
    for (size_t b = 0; b < in.size(); b++) subs[part_of[b]].push_back(in[b]);
    for (int p = 0; p < P; p++) {
        D.parts.emplace_back(new Part());
        Part &pt = *D.parts.back();
        pt.device = ctx->devices[p];
        pt.nblocks = (int64_t)subs[p].size();
        pt.own = Range{own_lo[p] - 1, own_hi[p]};
        block_hulls(subs[p], pt.rows, pt.cols);
        if (D.symmetric) pt.rows = pt.cols = hull(pt.rows, pt.cols);
        pt.has_image = !pt.rows.empty();
        // products ALONG the partition (op N; every op of a symmetric operator)
        D.plan_n.xr[p] = pt.cols;
        D.plan_n.out[p] = pt.own;
        touched_n[p] = pt.rows;
        // products ACROSS it (transpose / adjoint of a row-partitioned VBCRS / BlockSparseMatrix): every
        // part holds a partial result over the columns of its blocks; reduce-scatter onto equal chunks
        pt.colpart = square ? pt.own : Range{std::min<long long>(p * chunk_t, ncols), std::min<long long>((p + 1) * chunk_t, ncols)};
        D.plan_t.xr[p] = pt.rows;
        D.plan_t.out[p] = pt.colpart;
        touched_t[p] = pt.cols;
        D.plan_n.in[p] = pt.colpart;
        D.plan_t.in[p] = pt.own;
    }
    finish_plan(D.plan_n, touched_n, D.es);
    finish_plan(D.plan_t, touched_t, D.es);

    D.vlen = (size_t)std::max<long long>(std::max(nrows, ncols), 1) + 2;
    const size_t vec_bytes = D.vlen * D.es;
    for (int p = 0; p < P; p++) {
        Part &pt = *D.parts[p];
        DeviceGuard g;
        hipError_t e = g.enter(pt.device);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
        if (pt.has_image) {
            bsm_options po = o;
            po.ctx = nullptr;
            po.device = pt.device;
            po.transpose_image = 0;
            // the image defines (scales / overwrites) every row it touches plus the rows it owns
            po.own_lo = D.plan_n.zr[p].lo + 1;
            po.own_hi = D.plan_n.zr[p].hi;
            const bool devblocks = (o.blocks_memspace == BSM_MEM_DEVICE);
            std::unique_ptr<ValueSink> sink;
            if (!devblocks) sink = make_device_sink(&pt.img.d_values);
            AnalysisOptions ao = to_aopt(po, sink.get());
            ao.skip_colors = true;
            std::string err = pt.an.build(mtype, dtype, nrows, ncols, subs[p], ao);
            if (!err.empty()) return build_error("device part " + std::to_string(p) + ": " + err);
            if (devblocks) {  // the blocks may live on another device of the context: read over xGMI
                e = device_pack(pt.an, &pt.img.d_values);
                if (e != hipSuccess) return hip_fail(e, "device-side packing");
            }
            fill_image(pt.an, po, true, pt.img);
            e = upload_image(pt.an, pt.img, pt.device);
            if (e != hipSuccess) return hip_fail(e, "device upload");
        }
        e = hipStreamCreateWithFlags(&pt.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pt.ev_prod, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pt.ev_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pt.ev_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&pt.d_x, vec_bytes);
        if (e == hipSuccess) e = hipMalloc(&pt.d_w, vec_bytes);
        const size_t rb = std::max(D.plan_n.recv_bytes[p], D.plan_t.recv_bytes[p]);
        if (e == hipSuccess && rb) e = hipMalloc(&pt.d_recv, rb);
        if (e != hipSuccess) return hip_fail(e, "multi-device buffers");
    }
    D.all_peer = ctx->peer_ok;  // enabled for every pair when the context was created (bsm_ctx_create)
    if (const char *v = std::getenv("BSM_DIST_COPIES")) D.all_peer = D.all_peer && std::atoi(v) == 0;  // tests: force the copy path
    if (const char *v = std::getenv("BSM_DIST_ONE_STREAM")) D.one_stream = std::atoi(v) != 0;
    {
        // flags and zero-keeping: on by default where they have been exercised -- every part on ONE physical device
        // (virtual devices).  On distinct devices a flag is polled by another GPU's command processor and the finish
        // kernels write their zeros into a PEER's work vector over xGMI (visible to the owner's next atomic
        // accumulation only if those remote writes have reached its L2 at the event boundary): neither has ever run
        // on this pool, so both are opt-in there (BSM_DIST_FLAGS=1 / BSM_DIST_REZERO=1) until one 2-GPU parity run of
        // test_work_vectors_stay_zero_* / test_flag_and_event_ordering_* has passed on real devices.
        bool one_device = true;
        for (int d : ctx->devices) one_device = one_device && d == ctx->devices[0];
        D.rezero = one_device;
        if (const char *v = std::getenv("BSM_DIST_REZERO")) D.rezero = std::atoi(v) != 0;
        int can = 1;
        for (int d : ctx->devices) {
            int c = 0;
            if (hipDeviceGetAttribute(&c, hipDeviceAttributeCanUseStreamWaitValue, d) != hipSuccess) {
                (void)hipGetLastError();
                c = 0;
            }
            can = can && c;  // EVERY device of the context must be able to run the wait packets
        }
        D.use_flags = one_device && can;
        if (const char *v = std::getenv("BSM_DIST_FLAGS")) D.use_flags = std::atoi(v) != 0 && can;
        if (D.use_flags) {
            hipError_t e = hipHostMalloc((void **)&D.flags, (size_t)(3 * P + 1) * sizeof(uint64_t), hipHostMallocCoherent | hipHostMallocPortable);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                D.flags = nullptr;
                D.use_flags = false;
            } else {
                std::memset(D.flags, 0, (size_t)(3 * P + 1) * sizeof(uint64_t));
            }
        }
    }
    // one persistent issuing thread per device from five parts on (BSM_DIST_WORKERS = smallest part count that
    // gets them): below, the calling thread issues everything faster than the threads can be woken twice
    int wmin = 5;
    if (const char *v = std::getenv("BSM_DIST_WORKERS")) wmin = std::atoi(v);
    if (P >= wmin) D.workers.reset(new Workers(ctx->devices));
    return BSM_OK;
}

// part buffers for K columns (grow-only).  EVERY part's stream is drained before ANY buffer goes: a peer may
// still be reading this part's work vector.
static hipError_t grow_buffers(DistState &D, int K) {
    if (K <= D.kcap) return hipSuccess;
    const int P = (int)D.parts.size();
    const size_t es = (size_t)D.es;
    hipError_t e = hipSuccess;
    for (int p = 0; p < P && e == hipSuccess; p++) {
        DeviceGuard g;
        e = g.enter(D.parts[p]->device);
        if (e == hipSuccess) e = hipStreamSynchronize(D.parts[p]->stream);
    }
    if (e == hipSuccess && D.ev_tail) e = hipEventSynchronize(D.ev_tail);
    // the fused path works on the CALLERS' streams (run[p] = the part's caller stream), and a peer's finish kernel
    // reads this part's work vector over xGMI: ev_done is recorded on run[p] after the part's last use of the
    // buffers, so every part's ev_done has to be reached too before the first buffer is freed
    if (D.produced && D.last_mode == 0)
        for (int p = 0; p < P && e == hipSuccess; p++) e = hipEventSynchronize(D.parts[p]->ev_done);
    if (D.produced && D.last_mode >= 1) {  // ordered by flags / by one stream: no event carries the last use -- drain the devices
        for (int p = 0; p < P && e == hipSuccess; p++) {
            DeviceGuard g;
            e = g.enter(D.parts[p]->device);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
    }
    for (int p = 0; p < P && e == hipSuccess; p++) {
        Part &pt = *D.parts[p];
        DeviceGuard g;
        e = g.enter(pt.device);
        if (e != hipSuccess) break;
        for (void **q : {&pt.d_x, &pt.d_w, &pt.d_recv}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        e = hipMalloc(&pt.d_x, D.vlen * es * K);
        if (e == hipSuccess) e = hipMalloc(&pt.d_w, D.vlen * es * K);
        const size_t rb = std::max(D.plan_n.recv_bytes[p], D.plan_t.recv_bytes[p]);
        if (e == hipSuccess && rb) e = hipMalloc(&pt.d_recv, rb * K);
    }
    if (e == hipSuccess) D.kcap = K;
    for (auto &pt : D.parts) pt->w_clean = false;
    return e;
}

// ---- the fused path: device-resident vectors, every device of the context peer-accessible ----------------
// Where a part finds the x entries it reads, and where it delivers its y entries.  bsm_mul: ONE source (the
// caller's x) and one destination (the caller's y), on the caller's stream.  bsm_mul_parts: part q HOLDS the
// x entries plan.in[q] and receives the y entries plan.out[q], on its own stream.
struct VecSource {
    const char *base;   // virtual base: entry i of the global vector at base + i * es
    Range valid;        // entries this source holds
    int device;         // where the MEMORY lives (decides reading in place)
    int sdev;           // where the STREAM lives (a NULL stream is a different stream on every device): events of
                        // this source are recorded there, and "same stream" means the same stream on that device
    hipStream_t stream; // where the entries are produced
    hipEvent_t ready;   // recorded on `stream` at the start of the call
    int strided;        // column k of a multi-RHS batch at + k * ldx (else the source holds one column)
    int ready_flag = -1;  // flags mode: the counter that stands for `ready`
};
struct VecDest {
    char *base;  // virtual base of the y entries
    int device;  // where the memory lives (decides multiplying straight into y)
    int sdev;    // where the stream lives
    hipStream_t stream;
    hipEvent_t ready;
    int ready_flag = -1;
};

static int dist_mul_fused_issue(DistState &D, int op, int K, const std::vector<VecSource> &src, long long ldx,
                                const std::vector<VecDest> &dst, long long ldy, const void *alpha, const void *beta,
                                int beta_strong_zero);

// test hook (unexported in the header, like bsm_debug_move_image_array): the n-th fused fan-out from now fails between
// its two phases, the way a launch error would (tests/test_gpu_multidevice.py: the product after it must not hang)
static std::atomic<int> g_fail_countdown{-1};
extern "C" void bsm_debug_dist_fail_after(int n) { g_fail_countdown.store(n); }
static bool injected_failure() {
    int c = g_fail_countdown.load();
    if (c < 0) return false;
    return g_fail_countdown.fetch_sub(1) == 0;
}

// The fan-out, and what is left behind when it fails half-way.  Events degrade by themselves (an event that was never
// recorded counts as complete); sequence counters do not: the failed product has consumed a number, wait packets for it
// may already be queued, and the write packets that would release them were never issued -- the next product's
// previous() would wait for flag >= seq forever and the drain of a mode change would block the host.  So on any error
// the host writes the product's number into every counter (coherent pinned memory: the queued waits fall through),
// drains the devices and leaves the handle as after create: no ordering form remembered, no delivery to wait for, work
// vectors "unknown" (the next fused product clears them once).
static int dist_mul_fused(DistState &D, int op, int K, const std::vector<VecSource> &src, long long ldx,
                          const std::vector<VecDest> &dst, long long ldy, const void *alpha, const void *beta,
                          int beta_strong_zero) {
    const int rc = dist_mul_fused_issue(D, op, K, src, ldx, dst, ldy, alpha, beta, beta_strong_zero);
    if (rc == BSM_OK) return rc;
    const int P = (int)D.parts.size();
    if (D.flags) {
        volatile uint64_t *F = D.flags;
        for (int i = 0; i < 3 * P + 1; i++)
            if (F[i] < D.seq) F[i] = D.seq;
        __sync_synchronize();
    }
    for (int p = 0; p < P; p++) {
        DeviceGuard g;
        if (g.enter(D.parts[p]->device) == hipSuccess) (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        D.parts[p]->w_clean = false;
    }
    D.last_mode = -1;
    D.last_single_stream = nullptr;
    D.produced = false;
    return rc;
}

static int dist_mul_fused_issue(DistState &D, int op, int K, const std::vector<VecSource> &src, long long ldx,
                                const std::vector<VecDest> &dst, long long ldy, const void *alpha, const void *beta,
                                int beta_strong_zero) {
    const int P = (int)D.parts.size();
    const bool along = (op == BSM_OP_N) || D.symmetric;
    const Plan &pl = along ? D.plan_n : D.plan_t;
    const bool opT = (op != BSM_OP_N);
    const bool conj = (op == BSM_OP_C);
    const size_t es = (size_t)D.es;
    const size_t vlen = D.vlen;
    hipError_t e = hipSuccess;
#define DCHECK(call, what)                             \
    do {                                               \
        e = (call);                                    \
        if (e != hipSuccess) return hip_fail(e, what); \
    } while (0)
    DCHECK(grow_buffers(D, K), "multi-device buffers");
    // The stream a part's work is issued on.  bsm_mul_parts: the caller's stream of that part -- local work needs
    // no cross-stream hop at all, only what depends on a PEER waits for an event.  bsm_mul: the first part that
    // lives on the caller's device works on the caller's stream for the same reason (the hops of the other
    // parts then overlap with it); the others use their own streams.
    std::vector<hipStream_t> run((size_t)P);
    {
        bool taken = false;
        for (int p = 0; p < P; p++) {
            const VecDest &yd = dst[dst.size() == 1 ? 0 : (size_t)p];
            // (bsm_mul: EVERY part on the caller's device works on the caller's stream -- parts that share a device gain
            // nothing from streams of their own, and each one costs two cross-stream hops per product)
            const bool mine = yd.sdev == D.parts[p]->device && (dst.size() > 1 || !taken || D.one_stream);
            run[p] = mine ? yd.stream : D.parts[p]->stream;
            if (mine) taken = true;
        }
    }
    // flags pay where streams wait for each other; a product whose parts all work on ONE stream orders nothing at all
    // (below)
    bool several_streams = false;
    for (int p = 1; p < P; p++) several_streams = several_streams || run[p] != run[0] || D.parts[p]->device != D.parts[0]->device;
    // everything on ONE stream of one device -- the parts and the caller's vectors (one part, or partitioned vectors
    // driven from one stream of one virtual device): stream order is all the ordering there is to do, no event is
    // recorded and none waited for (every record is a barrier packet between two launches: one part through a
    // context cost +15 us over the plain handle, tools/distbench.py)
    bool single = !several_streams;
    for (const VecSource &s : src) single = single && s.stream == run[0] && s.sdev == D.parts[0]->device;
    for (const VecDest &d : dst) single = single && d.stream == run[0] && d.sdev == D.parts[0]->device;
    const bool flags = D.use_flags && several_streams;
    const int mode = single ? 2 : (flags ? 1 : 0);
    if (D.last_mode >= 0 && (D.last_mode != mode || (single && D.last_single_stream != run[0]))) {
        // the previous product of the handle was ordered the other way (events / flags, e.g. the copy path of a host
        // vector in between): the two forms do not see each other, so everything is drained once
        for (int p = 0; p < P; p++) {
            DeviceGuard g;
            DCHECK(g.enter(D.parts[p]->device), "hipSetDevice");
            DCHECK(hipDeviceSynchronize(), "hipDeviceSynchronize");
        }
    }
    D.last_mode = mode;
    D.last_single_stream = single ? run[0] : nullptr;
    // (the counters only ever hold numbers of products ordered by flags: a product ordered by events in between must not
    // consume one, or the next one would wait for a value nobody writes)
    const uint64_t prev_seq = D.seq;
    const uint64_t seq = flags ? ++D.seq : D.seq;
    uint64_t *const F = D.flags;
    const uint64_t kAll = ~(uint64_t)0;
    // "recorded" / "waited for": an event, or (flags) the product's sequence number in a counter
    auto signal = [&](hipEvent_t ev, int flag, hipStream_t st) -> hipError_t {
        if (single) return hipSuccess;
        return flags ? hipStreamWriteValue64(st, F + flag, seq, 0) : hipEventRecord(ev, st);
    };
    auto await = [&](hipStream_t st, hipEvent_t ev, int flag, uint64_t value) -> hipError_t {
        if (single) return hipSuccess;
        return flags ? hipStreamWaitValue64(st, F + flag, value, hipStreamWaitValueGte, kAll) : hipStreamWaitEvent(st, ev, 0);
    };
    // "x (and the incoming y) are ready" on every stream that produces them
    {
        std::vector<std::pair<hipEvent_t, int>> done;
        auto record = [&](hipEvent_t ev, int flag, hipStream_t st, int dev) -> hipError_t {
            for (const auto &d : done)
                if (d.first == ev && d.second == flag) return hipSuccess;
            done.emplace_back(ev, flag);
            DeviceGuard g;
            hipError_t e2 = g.enter(dev);
            return e2 == hipSuccess ? signal(ev, flag, st) : e2;
        };
        for (const VecSource &s : src) DCHECK(record(s.ready, s.ready_flag, s.stream, s.sdev), "record: inputs ready");
        for (const VecDest &d : dst) DCHECK(record(d.ready, d.ready_flag, d.stream, d.sdev), "record: inputs ready");
    }
    const bool was_produced = D.produced;
    const bool rezero = D.rezero;  // this product leaves the work vectors it uses zero again
    static const float kOneF[2] = {1.f, 0.f};
    static const double kOneD[2] = {1.0, 0.0};
    const void *one = (D.dtype == 0 || D.dtype == 2) ? (const void *)kOneF : (const void *)kOneD;
    // A part that neither sends nor receives a y segment (VBCRS forward: block rows own disjoint y ranges,
    // reference src/vbcrs.jl:275-283) multiplies straight into the caller's y -- beta fused, no work vector,
    // no delivery launch.
    std::vector<char> direct((size_t)P, 0);
    for (int p = 0; p < P; p++) {
        // (only into y on the part's OWN device: accumulate-mode images add with hardware fp atomics, which are
        // not relied upon across xGMI; a remote y receives plain stores from the finish kernel instead)
        const VecDest &yd = dst[dst.size() == 1 ? 0 : (size_t)p];
        bool alone = D.parts[p]->has_image && yd.device == D.parts[p]->device && pl.zr[p].lo == pl.out[p].lo &&
                     pl.zr[p].hi == pl.out[p].hi;
        for (const Transfer &t : pl.transfers) alone = alone && t.from != p && t.to != p;
        direct[p] = alone;
    }
    // (a NULL stream is a different stream on every device: "same stream" needs the same device too)
    auto wait_for = [&](int p, hipEvent_t ev, int flag, hipStream_t recorded_on, int recorded_dev) -> hipError_t {
        if (recorded_on == run[p] && recorded_dev == D.parts[p]->device) return hipSuccess;  // same stream: already ordered
        return await(run[p], ev, flag, seq);
    };
    // phase 0 (part p): wait for its inputs, gather the x pieces it reads (one launch), local product
    // phase 1 (part q): add the y segments its peers produced for its rows, deliver (one launch)
    auto phase = [&](int ph, int p, bool bind) -> int {
        hipError_t e = hipSuccess;
        Part &pt = *D.parts[p];
        DeviceGuard g;
        if (bind) DCHECK(g.enter(pt.device), "hipSetDevice");
        const VecDest &yd = dst[dst.size() == 1 ? 0 : (size_t)p];
        hipStream_t st = run[(size_t)p];
        if (ph == 0) {
            // peers that read this part's work vector in the previous product have finished (their ev_done
            // carries that product's record until phase 1 of THIS product re-records it, after a host barrier)
            if (was_produced && !single) {
                // (flags: the previous product's number; a first fused product after copy-path ones was drained above.
                // Every wait is a packet in front of the product: each delivery once, and none for a delivery that was
                // recorded on this very stream)
                std::vector<int> waited;
                auto previous = [&](int q) -> hipError_t {
                    const Part &pq = *D.parts[q];
                    if (pq.done_stream == st && pq.device == pt.device) return hipSuccess;
                    for (int w : waited)
                        if (w == q) return hipSuccess;
                    waited.push_back(q);
                    return await(st, pq.ev_done, 2 * P + q, prev_seq);
                };
                for (const Transfer &t : D.plan_n.transfers)
                    if (t.from == p) DCHECK(previous(t.to), "wait: previous delivery");
                for (const Transfer &t : D.plan_t.transfers)
                    if (t.from == p) DCHECK(previous(t.to), "wait: previous delivery");
                DCHECK(previous(p), "wait: previous delivery");  // its own previous delivery (another stream, perhaps)
                if (!flags && D.ev_tail) DCHECK(hipStreamWaitEvent(st, D.ev_tail, 0), "hipStreamWaitEvent");
            }
            DCHECK(wait_for(p, yd.ready, yd.ready_flag, yd.stream, yd.sdev), "wait: y ready");  // the incoming y (numeric beta) / its buffer
            const Range zr = pl.zr[p];
            if (pt.has_image) {
                const Range xr = pl.xr[p];
                const void *xp = pt.d_x;
                long long xld = (long long)vlen;
                VecPieces pc;
                int np = 0;
                bool in_place = false;
                for (const VecSource &s : src) {
                    const Range o = isect(s.valid, xr);
                    if (o.empty()) continue;
                    DCHECK(wait_for(p, s.ready, s.ready_flag, s.stream, s.sdev), "wait: x ready");
                    if (o.lo == xr.lo && o.hi == xr.hi && s.device == pt.device) {  // everything it reads lies on its own device
                        xp = s.base;
                        xld = s.strided ? ldx : 0;
                        in_place = true;
                        break;
                    }
                    if (np == kMaxVecPieces) {
                        DCHECK(launch_vec_fetch(D.dtype, pt.d_x, (long long)vlen, pc, np, ldx, K, st), "x fetch");
                        np = 0;
                    }
                    pc.base[np] = s.base;
                    pc.lo[np] = o.lo;
                    pc.hi[np] = o.hi;
                    pc.strided[np] = s.strided;
                    np++;
                }
                if (!in_place && np) DCHECK(launch_vec_fetch(D.dtype, pt.d_x, (long long)vlen, pc, np, ldx, K, st), "x fetch");
                const long long z[2] = {zr.lo, zr.hi};
                void *target = direct[p] ? (void *)yd.base : pt.d_w;
                const long long tld = direct[p] ? ldy : (long long)vlen;
                // into the work vector: `w = 0` first -- or, when the finish kernels keep it zero, plain accumulation
                // (beta = 1: no launch in front; unknown contents are cleared once, whole buffer)
                if (!direct[p] && rezero && !pt.w_clean)
                    DCHECK(hipMemsetAsync(pt.d_w, 0, (size_t)D.kcap * vlen * es, st), "memset");
                if (!direct[p]) pt.w_clean = false;  // (until the whole product has been issued)
                const void *b = direct[p] ? beta : (rezero ? one : nullptr);
                const int sz = direct[p] ? beta_strong_zero : (rezero ? 0 : 1);
                if (K == 1)
                    DCHECK(launch_mul(pt.img, opT, conj, xp, target, alpha, b, sz, st, false, z), "kernel launch");
                else
                    DCHECK(launch_mul_multi(pt.img, opT, conj, K, xp, xld, target, tld, alpha, b, sz, st, z, part_il(pt, opT, K)), "kernel launch");
            } else if (rezero) {
                if (!pt.w_clean) DCHECK(hipMemsetAsync(pt.d_w, 0, (size_t)D.kcap * vlen * es, st), "memset");
                pt.w_clean = false;
            } else if (!zr.empty()) {
                for (int k = 0; k < K; k++)
                    DCHECK(hipMemsetAsync((char *)pt.d_w + ((size_t)k * vlen + zr.lo) * es, 0, (size_t)zr.len() * es, st), "memset");
            }
            DCHECK(signal(pt.ev_prod, P + p, st), "record: product done");
            return BSM_OK;
        }
        const Range o = pl.out[p];
        VecPieces pc;
        int np = 0;
        for (const Transfer &t : pl.transfers) {
            if (t.to != p) continue;
            Part &from = *D.parts[t.from];
            DCHECK(wait_for(p, from.ev_prod, P + t.from, run[(size_t)t.from], from.device), "wait: peer's product");
            if (np == kMaxVecPieces) {  // more peers than one launch takes: fold these into the work vector first
                DCHECK(launch_vec_finish(D.dtype, nullptr, 0, pt.d_w, (long long)vlen, pc, np, o.lo, o.hi, nullptr, 1, 1, rezero, K, st), "halo add");
                np = 0;
            }
            pc.base[np] = from.d_w;
            pc.lo[np] = t.range.lo;
            pc.hi[np] = t.range.hi;
            pc.strided[np] = 1;
            np++;
        }
        if (!o.empty() && !direct[p])
            DCHECK(launch_vec_finish(D.dtype, yd.base, ldy, pt.d_w, (long long)vlen, pc, np, o.lo, o.hi, beta, beta_strong_zero, 0, rezero,
                                     K, st), "y delivery");
        DCHECK(signal(pt.ev_done, 2 * P + p, st), "record: delivery done");
        pt.done_stream = st;
        return BSM_OK;
    };
    for (int ph = 0; ph < 2; ph++) {
        int rc = BSM_OK;
        if (ph == 1 && injected_failure()) return hip_fail(hipErrorUnknown, "injected failure between the phases of a fan-out");
        if (D.workers)
            rc = D.workers->run([&](int p) { return phase(ph, p, false); });
        else
            for (int p = 0; p < P && rc == BSM_OK; p++) rc = phase(ph, p, true);
        if (rc != BSM_OK) return rc;
    }
    D.produced = true;
    for (int p = 0; p < P; p++)
        if (!direct[p]) D.parts[p]->w_clean = rezero;
    // the consumers of y continue when the parts that deliver to them are done
    for (int q = 0; q < P; q++) {
        const VecDest &yd = dst[dst.size() == 1 ? 0 : (size_t)q];
        if (yd.stream == run[(size_t)q] && yd.sdev == D.parts[q]->device) continue;  // delivered on the consumer's own stream
        DeviceGuard g;
        DCHECK(g.enter(yd.sdev), "hipSetDevice");
        DCHECK(await(yd.stream, D.parts[q]->ev_done, 2 * P + q, seq), "wait: delivery");
    }
#undef DCHECK
    return BSM_OK;
}

// K <= 16 right-hand sides in one fan-out (K = 1: bsm_mul).  X / Y: column k at x + k * ldx / y + k * ldy
// elements.  The part buffers hold column k at k * vlen elements.
static int dist_mul_copies(bsm_matrix_s *A, int op, int K, const void *x, long long ldx, void *y, long long ldy,
                           const void *alpha, const void *beta, int beta_strong_zero, int memspace, hipStream_t stream) {
    DistState &D = *A->dist;
    const int P = (int)D.parts.size();
    const bool along = (op == BSM_OP_N) || D.symmetric;
    const Plan &pl = along ? D.plan_n : D.plan_t;
    const bool opT = (op != BSM_OP_N);
    const bool conj = (op == BSM_OP_C);
    const size_t es = (size_t)D.es;
    const long long ylen = (op == BSM_OP_N) ? D.nrows : D.ncols;
    static const double kZero[2] = {0.0, 0.0};
    if (!beta) beta = kZero;  // "NULL = 0" of include/bsm_rocm.h, also where this file reads beta itself
    const size_t vlen = D.vlen;  // elements per column of the part buffers
    const char *xb = (const char *)x;
    char *yb = (char *)y;
    const bool host = (memspace == BSM_MEM_HOST);
    if (!host && memspace != BSM_MEM_DEVICE) return fail(BSM_ERR_INVALID, "bad memspace");
    hipError_t e = hipSuccess;
#define DCHECK(call, what)                        \
    do {                                          \
        e = (call);                               \
        if (e != hipSuccess) return hip_fail(e, what); \
    } while (0)

    DCHECK(grow_buffers(D, K), "multi-device buffers");
    if (D.last_mode >= 1) {  // the previous product was ordered by flags (or by one stream), this path orders by events: drain once
        for (int p = 0; p < P; p++) {
            DeviceGuard g;
            DCHECK(g.enter(D.parts[p]->device), "hipSetDevice");
            DCHECK(hipDeviceSynchronize(), "hipDeviceSynchronize");
        }
    }
    D.last_mode = 0;
    for (auto &pt : D.parts) pt->w_clean = false;  // (this path leaves its partial sums in the work vectors)
    int cur = 0;
    DCHECK(hipGetDevice(&cur), "hipGetDevice");
    int xdev = -1, ydev = -1, sdev = cur;
    hipEvent_t ev_ready = nullptr;
    if (!host) {
        xdev = pointer_device(x, cur);
        ydev = pointer_device(y, cur);
        if (stream) {
            int sd = cur;
            if (hipStreamGetDevice(stream, &sd) == hipSuccess) sdev = sd;
            else (void)hipGetLastError();
        }
        auto it = D.ev_x.find(sdev);
        if (it == D.ev_x.end()) {
            DeviceGuard g;
            DCHECK(g.enter(sdev), "hipSetDevice");
            hipEvent_t ev;
            DCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            it = D.ev_x.emplace(sdev, ev).first;
        }
        ev_ready = it->second;
        DeviceGuard g;
        DCHECK(g.enter(sdev), "hipSetDevice");
        DCHECK(hipEventRecord(ev_ready, stream), "hipEventRecord");  // x (and the incoming y) are ready
    }
    const bool numeric_beta = !beta_strong_zero;
    const size_t res_need = (size_t)ylen * es * K;  // staging of numeric-beta results: column k at k * ylen
    if (numeric_beta && host && D.h_res.size() < res_need) D.h_res.resize(res_need);
    if (numeric_beta && !host) {
        bool remote = false;
        for (int q = 0; q < P; q++) remote |= (!pl.out[q].empty() && D.parts[q]->device != ydev);
        if (remote && (D.res_dev != ydev || D.res_bytes < res_need)) {
            if (D.d_res) {
                DeviceGuard g;
                (void)g.enter(D.res_dev);
                (void)hipFree(D.d_res);
                D.d_res = nullptr;
            }
            DeviceGuard g;
            DCHECK(g.enter(ydev), "hipSetDevice");
            DCHECK(hipMalloc(&D.d_res, res_need + 16), "hipMalloc(result staging)");
            D.res_dev = ydev;
            D.res_bytes = res_need;
        }
    }

    // The issue of every part's work -- one stream per device -- is two phases with a host barrier in
    // between (a stream can only wait for an event that HAS been recorded):
    //   phase 0 (part p): x to its device, local product, record ev_prod
    //   phase 1 (part q): the y segments other parts produced for q's rows (peer copy over xGMI + add),
    //                     then q's owned range to the caller's y, record ev_done
    // Run inline by the calling thread for up to two parts, by one persistent worker thread per device
    // beyond that (each bound to its device once: the host-side issue cost no longer grows with the
    // number of GPUs).
    std::vector<char> late_flag((size_t)P, 0);  // numeric beta, remote part, y on a device: combined below
    auto phase = [&](int ph, int p, bool bind) -> int {
        hipError_t e = hipSuccess;  // (per invocation: the phases of different parts run concurrently)
        Part &pt = *D.parts[p];
        DeviceGuard g;
        if (bind) DCHECK(g.enter(pt.device), "hipSetDevice");
        if (ph == 0) {
            // the previous product of this handle -- peers still reading this part's work vector, the late
            // combine on the caller's stream reading the result staging -- is complete before this one starts
            if (D.produced) {
                for (int q = 0; q < P; q++) DCHECK(hipStreamWaitEvent(pt.stream, D.parts[q]->ev_done, 0), "hipStreamWaitEvent");
                if (D.ev_tail) DCHECK(hipStreamWaitEvent(pt.stream, D.ev_tail, 0), "hipStreamWaitEvent");
            }
            if (ev_ready) DCHECK(hipStreamWaitEvent(pt.stream, ev_ready, 0), "hipStreamWaitEvent");
            const Range zr = pl.zr[p];
            if (pt.has_image) {
                const Range xr = pl.xr[p];
                const void *xp = pt.d_x;
                long long xld = (long long)vlen;
                if (!host && xdev == pt.device) {
                    xp = x;  // same device: the local product reads the caller's x directly
                    xld = ldx;
                } else {
                    for (int k = 0; k < K; k++) {
                        char *dst = (char *)pt.d_x + ((size_t)k * vlen + xr.lo) * es;
                        const char *src = xb + ((size_t)k * ldx + xr.lo) * es;
                        if (host)
                            DCHECK(hipMemcpyAsync(dst, src, (size_t)xr.len() * es, hipMemcpyHostToDevice, pt.stream), "x upload");
                        else
                            DCHECK(copy_between(dst, pt.device, src, xdev, (size_t)xr.len() * es, pt.stream), "x peer copy");
                    }
                }
                const long long z[2] = {zr.lo, zr.hi};
                if (K == 1)
                    DCHECK(launch_mul(pt.img, opT, conj, xp, pt.d_w, alpha, nullptr, 1, pt.stream, pt.img.d_ws != nullptr, z),
                           "kernel launch");
                else
                    DCHECK(launch_mul_multi(pt.img, opT, conj, K, xp, xld, pt.d_w, (long long)vlen, alpha, nullptr, 1,
                                            pt.stream, z, part_il(pt, opT, K)), "kernel launch");
            } else if (!zr.empty()) {
                for (int k = 0; k < K; k++)
                    DCHECK(hipMemsetAsync((char *)pt.d_w + ((size_t)k * vlen + zr.lo) * es, 0, (size_t)zr.len() * es, pt.stream),
                           "memset");
            }
            DCHECK(hipEventRecord(pt.ev_prod, pt.stream), "hipEventRecord");
            return BSM_OK;
        }
        // 3: y segments produced for rows of THIS part: peer copy over xGMI + local add
        const size_t rstride = std::max(D.plan_n.recv_bytes[p], D.plan_t.recv_bytes[p]);
        for (const Transfer &t : pl.transfers) {
            if (t.to != p) continue;
            Part &src = *D.parts[t.from];
            DCHECK(hipStreamWaitEvent(pt.stream, src.ev_prod, 0), "hipStreamWaitEvent");
            for (int k = 0; k < K; k++) {
                char *rb = (char *)pt.d_recv + (size_t)k * rstride + t.recv_off;
                const size_t off = ((size_t)k * vlen + t.range.lo) * es;
                DCHECK(copy_between(rb, pt.device, (char *)src.d_w + off, src.device, (size_t)t.range.len() * es, pt.stream),
                       "halo peer copy");
                DCHECK(launch_vec_add(D.dtype, (char *)pt.d_w + off, rb, t.range.len(), pt.stream), "halo add");
            }
        }
        // 4: deliver the owned range
        const Range o = pl.out[p];
        if (!o.empty()) {
            const size_t bytes = (size_t)o.len() * es;
            for (int k = 0; k < K; k++) {
                const char *w = (const char *)pt.d_w + ((size_t)k * vlen + o.lo) * es;
                char *yk = yb + ((size_t)k * ldy + o.lo) * es;
                const size_t roff = ((size_t)k * ylen + o.lo) * es;
                if (host) {
                    char *dst = numeric_beta ? D.h_res.data() + roff : yk;
                    DCHECK(hipMemcpyAsync(dst, w, bytes, hipMemcpyDeviceToHost, pt.stream), "y download");
                } else if (!numeric_beta) {
                    DCHECK(copy_between(yk, ydev, w, pt.device, bytes, pt.stream), "y peer copy");
                } else if (pt.device == ydev) {
                    DCHECK(launch_vec_axpby(D.dtype, yk, w, o.len(), beta, pt.stream), "y combine");
                } else {
                    DCHECK(copy_between((char *)D.d_res + roff, ydev, w, pt.device, bytes, pt.stream), "y peer copy");
                    late_flag[p] = 1;
                }
            }
        }
        DCHECK(hipEventRecord(pt.ev_done, pt.stream), "hipEventRecord");
        return BSM_OK;
    };
    for (int ph = 0; ph < 2; ph++) {
        int rc = BSM_OK;
        if (ph == 1 && injected_failure()) return hip_fail(hipErrorUnknown, "injected failure between the phases of a fan-out");
        if (D.workers)
            rc = D.workers->run([&](int p) { return phase(ph, p, false); });
        else
            for (int p = 0; p < P && rc == BSM_OK; p++) rc = phase(ph, p, true);
        if (rc != BSM_OK) return rc;
    }
    D.produced = true;
    if (host) {
        for (int q = 0; q < P; q++) {
            Part &pt = *D.parts[q];
            DeviceGuard g;
            DCHECK(g.enter(pt.device), "hipSetDevice");
            DCHECK(hipStreamSynchronize(pt.stream), "multi-device mul");
        }
        if (numeric_beta) {
            for (int q = 0; q < P; q++) {
                const Range o = pl.out[q];
                if (o.empty()) continue;
                for (int k = 0; k < K; k++) {
                    char *yk = yb + ((size_t)k * ldy + o.lo) * es;
                    const char *rk = D.h_res.data() + ((size_t)k * ylen + o.lo) * es;
                    switch (D.dtype) {
                        case BSM_F32: host_axpby((float *)yk, (const float *)rk, o.len(), *(const float *)beta); break;
                        case BSM_F64: host_axpby((double *)yk, (const double *)rk, o.len(), *(const double *)beta); break;
                        case BSM_C64: host_axpby((std::complex<float> *)yk, (const std::complex<float> *)rk, o.len(), *(const std::complex<float> *)beta); break;
                        default: host_axpby((std::complex<double> *)yk, (const std::complex<double> *)rk, o.len(), *(const std::complex<double> *)beta); break;
                    }
                }
            }
        }
        return BSM_OK;
    }
    // y on a device: the caller's stream continues when every part has delivered
    {
        DeviceGuard g;
        DCHECK(g.enter(sdev), "hipSetDevice");
        for (int q = 0; q < P; q++) DCHECK(hipStreamWaitEvent(stream, D.parts[q]->ev_done, 0), "hipStreamWaitEvent");
    }
    bool any_late = false;
    for (int q = 0; q < P; q++) any_late |= late_flag[q] != 0;
    if (any_late) {
        DeviceGuard g;
        DCHECK(g.enter(ydev), "hipSetDevice");
        for (int q = 0; q < P; q++) {
            if (!late_flag[q]) continue;
            const Range o = pl.out[q];
            for (int k = 0; k < K; k++)
                DCHECK(launch_vec_axpby(D.dtype, yb + ((size_t)k * ldy + o.lo) * es,
                                        (char *)D.d_res + ((size_t)k * ylen + o.lo) * es, o.len(), beta, stream), "y combine");
        }
    }
    {  // the end of this product on the caller's stream: what the next product of the handle waits for
        DeviceGuard g;
        DCHECK(g.enter(sdev), "hipSetDevice");
        if (D.ev_tail && D.tail_dev != sdev) {
            (void)hipEventDestroy(D.ev_tail);
            D.ev_tail = nullptr;
        }
        if (!D.ev_tail) {
            DCHECK(hipEventCreateWithFlags(&D.ev_tail, hipEventDisableTiming), "hipEventCreate");
            D.tail_dev = sdev;
        }
        DCHECK(hipEventRecord(D.ev_tail, stream), "hipEventRecord");
    }
#undef DCHECK
    return BSM_OK;
}

static int dist_mul_k(bsm_matrix_s *A, int op, int K, const void *x, long long ldx, void *y, long long ldy,
                      const void *alpha, const void *beta, int beta_strong_zero, int memspace, hipStream_t stream) {
    DistState &D = *A->dist;
    std::lock_guard<std::mutex> lock(D.mu);  // one product of a handle is ISSUED at a time (the work vectors are the handle's)
    if (memspace == BSM_MEM_DEVICE && stream) {
        // a multi-device product issues on several streams and devices and waits for events of earlier products:
        // it cannot be recorded into the caller's graph (include/bsm_rocm.h) -- refuse instead of corrupting the capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
        if (cs != hipStreamCaptureStatusNone)
            return fail(BSM_ERR_UNSUPPORTED, "a multi-device handle cannot be captured into a graph");
    }
    if (memspace == BSM_MEM_DEVICE && D.all_peer) {
        int cur = 0;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
        const int xdev = pointer_device(x, cur), ydev = pointer_device(y, cur);
        int sdev = cur;
        if (stream && hipStreamGetDevice(stream, &sdev) != hipSuccess) {
            (void)hipGetLastError();
            sdev = cur;
        }
        bool inside = false, xin = false, yin = false;
        for (const auto &pp : D.parts) {
            inside |= pp->device == sdev;
            xin |= pp->device == xdev;
            yin |= pp->device == ydev;
        }
        if (inside && xin && yin) {  // x, y and the stream live on devices of the context: everybody can address them
            auto it = D.ev_x.find(sdev);
            if (it == D.ev_x.end()) {
                DeviceGuard g;
                e = g.enter(sdev);
                hipEvent_t ev = nullptr;
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
                it = D.ev_x.emplace(sdev, ev).first;
            }
            const long long xlen = (op == BSM_OP_N) ? D.ncols : D.nrows;
            const int P = (int)D.parts.size();
            std::vector<VecSource> src{VecSource{(const char *)x, Range{0, xlen}, xdev, sdev, stream, it->second, 1, 3 * P}};
            std::vector<VecDest> dst{VecDest{(char *)y, ydev, sdev, stream, it->second, 3 * P}};
            return dist_mul_fused(D, op, K, src, ldx, dst, ldy, alpha, beta, beta_strong_zero);
        }
    }
    return dist_mul_copies(A, op, K, x, ldx, y, ldy, alpha, beta, beta_strong_zero, memspace, stream);
}

// bsm_mul_parts: x and y PARTITIONED over the devices of the handle (include/bsm_rocm.h)
int dist_mul_parts(bsm_matrix_s *A, int op, const void *const *x_parts, void *const *y_parts, const void *alpha,
                   const void *beta, int beta_strong_zero, void *const *streams) {
    DistState &D = *A->dist;
    std::lock_guard<std::mutex> lock(D.mu);
    if (!D.all_peer)
        return fail(BSM_ERR_UNSUPPORTED, "bsm_mul_parts needs peer access between all devices of the context");
    const int P = (int)D.parts.size();
    const bool along = (op == BSM_OP_N) || D.symmetric;
    const Plan &pl = along ? D.plan_n : D.plan_t;
    const size_t es = (size_t)D.es;
    std::vector<VecSource> src;
    std::vector<VecDest> dst;
    for (int p = 0; p < P; p++) {
        const Part &pt = *D.parts[p];
        const Range in = pl.in[p], out = pl.out[p];
        if ((!in.empty() && !x_parts[p]) || (!out.empty() && !y_parts[p]))
            return fail(BSM_ERR_INVALID, "part " + std::to_string(p) + ": null vector part");
        hipStream_t st = streams ? (hipStream_t)streams[p] : nullptr;
        if (st) {
            // like bsm_mul on a multi-device handle (include/bsm_rocm.h): the product waits for events of other streams
            // and devices, which must not be recorded into a caller's graph
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
            if (cs != hipStreamCaptureStatusNone)
                return fail(BSM_ERR_UNSUPPORTED, "a multi-device handle cannot be captured into a graph");
        }
        src.push_back(VecSource{(const char *)x_parts[p] - (size_t)in.lo * es, in, pt.device, pt.device, st, pt.ev_in, 0, p});
        dst.push_back(VecDest{(char *)y_parts[p] - (size_t)out.lo * es, pt.device, pt.device, st, pt.ev_in, p});
    }
    return dist_mul_fused(D, op, 1, src, 0, dst, 0, alpha, beta, beta_strong_zero);
}

int dist_mul(bsm_matrix_s *A, int op, const void *x, void *y, const void *alpha, const void *beta,
             int beta_strong_zero, int memspace, hipStream_t stream) {
    return dist_mul_k(A, op, 1, x, 0, y, 0, alpha, beta, beta_strong_zero, memspace, stream);
}

// Y = alpha op(A) X + beta Y: every device streams its part of A ONCE per batch of up to 8 columns
// (bsm_mul_multi's own batching), the exchange and the delivery move the batch's columns together
int dist_mul_multi(bsm_matrix_s *A, int op, long long nrhs, const void *X, long long ldx, void *Y, long long ldy,
                   const void *alpha, const void *beta, int beta_strong_zero, int memspace, hipStream_t stream) {
    const size_t es = (size_t)A->dist->es;
    // batches of 8 columns per fan-out; Float32 / Float64 operators take 9 and more columns 16 at a time (the parts'
    // 16-column matrix-pipe passes, bsm_kernels.hip: kMfmaReal)
    const bool real = A->dist->dtype == 0 || A->dist->dtype == 1;
    for (long long k = 0; k < nrhs;) {
        const long long left = nrhs - k;
        const int kb = (int)((real && left >= 9) ? std::min<long long>(16, left) : std::min<long long>(8, left));
        int rc = dist_mul_k(A, op, kb, (const char *)X + (size_t)k * ldx * es, ldx, (char *)Y + (size_t)k * ldy * es, ldy,
                            alpha, beta, beta_strong_zero, memspace, stream);
        if (rc != BSM_OK) return rc;
        k += kb;
    }
    return BSM_OK;
}

int dist_part_info(bsm_matrix_s *A, int32_t part, bsm_part_info_t *out) {
    DistState &D = *A->dist;
    if (part < 0 || part >= (int32_t)D.parts.size()) return fail(BSM_ERR_INVALID, "part index out of range");
    const Part &p = *D.parts[part];
    std::memset(out, 0, sizeof *out);
    out->device = p.device;
    out->own_lo = p.own.lo + 1;
    out->own_hi = p.own.hi;
    const Range z = D.plan_n.zr[part];
    out->touched_lo = z.lo + 1;
    out->touched_hi = z.hi;
    out->device_bytes = p.img.device_bytes;
    out->nblocks = p.nblocks;
    out->col_lo = p.colpart.lo + 1;
    out->col_hi = p.colpart.hi;
    return BSM_OK;
}

void dist_images(bsm_matrix_s *A, std::vector<std::pair<const Analysis *, const DeviceImage *>> &out) {
    for (const auto &p : A->dist->parts)
        if (p->has_image) out.emplace_back(&p->an, &p->img);
}

int64_t dist_device_bytes(const bsm_matrix_s *A) {
    int64_t s = 0;
    for (const auto &p : A->dist->parts) s += p->img.device_bytes;
    return s;
}

}  // namespace bsm
