// bsm_analysis.cpp -- host analysis: reference bookkeeping + GPU schedule.  See bsm_analysis.h.
#include "bsm_analysis.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <thread>
#include <tuple>
#include <unordered_map>

namespace bsm {

Tunables Tunables::from_env() {
    Tunables t;
    auto geti = [](const char *name, int64_t dflt) -> int64_t {
        const char *s = std::getenv(name);
        if (!s || !*s) return dflt;
        return std::strtoll(s, nullptr, 10);
    };
    t.split2_bytes = geti("BSM_SPLIT2_BYTES", t.split2_bytes);
    t.split4_bytes = geti("BSM_SPLIT4_BYTES", t.split4_bytes);
    t.wgitem_max_bytes = geti("BSM_WGITEM_MAX_BYTES", t.wgitem_max_bytes);
    t.wave_bytes = geti("BSM_WAVE_BYTES", t.wave_bytes);
    t.multi_wave_bytes = geti("BSM_MULTI_WAVE_BYTES", t.multi_wave_bytes);
    t.target_waves = std::max<int64_t>(1, geti("BSM_TARGET_WAVES", t.target_waves));
    if (const char *f = std::getenv("BSM_FAT_FILL_BELOW")) t.fat_fill_below = std::atof(f);
    t.pack_threads = (int)geti("BSM_PACK_THREADS", t.pack_threads);
    t.wg_order = (int)geti("BSM_ORDER", t.wg_order);
    t.deep_group_bytes = geti("BSM_DEEP_GROUP_BYTES", t.deep_group_bytes);
    t.deep_total_bytes = geti("BSM_DEEP_TOTAL_BYTES", t.deep_total_bytes);
    t.window_bytes = (size_t)geti("BSM_UPLOAD_WINDOW_BYTES", (int64_t)t.window_bytes);
    if (t.pack_threads < 1) t.pack_threads = 1;
    return t;
}

// ---------------------------------------------------------------------------------------
// Colouring (specifications: bsm_analysis.h / oracle/bsm_oracle.c).
// ---------------------------------------------------------------------------------------
namespace {

// distinct neighbours of every list (two lists are adjacent iff they share an index)
std::vector<std::vector<int32_t>> conflict_adjacency(const std::vector<const int64_t *> &lists,
                                                     const std::vector<int64_t> &lens) {
    const int64_t nb = (int64_t)lists.size();
    int64_t maxindex = 0;
    for (int64_t b = 0; b < nb; b++)
        for (int64_t k = 0; k < lens[b]; k++) maxindex = std::max(maxindex, lists[b][k]);
    // incidence index -> blocks
    std::vector<int64_t> ptr(maxindex + 2, 0);
    for (int64_t b = 0; b < nb; b++)
        for (int64_t k = 0; k < lens[b]; k++) ptr[lists[b][k] + 1]++;
    for (int64_t i = 0; i <= maxindex; i++) ptr[i + 1] += ptr[i];
    std::vector<int64_t> inc(ptr[maxindex + 1]);
    {
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        for (int64_t b = 0; b < nb; b++)
            for (int64_t k = 0; k < lens[b]; k++) inc[fill[lists[b][k]]++] = b;
    }
    std::vector<std::vector<int32_t>> adj(nb);
    std::vector<int64_t> stamp(nb, -1);
    for (int64_t v = 0; v < nb; v++) {
        stamp[v] = v;
        for (int64_t k = 0; k < lens[v]; k++) {
            int64_t i = lists[v][k];
            for (int64_t p = ptr[i]; p < ptr[i + 1]; p++) {
                int64_t w = inc[p];
                if (stamp[w] != v) {
                    stamp[w] = v;
                    adj[v].push_back((int32_t)w);
                }
            }
        }
    }
    return adj;
}

// DSATUR on the subgraph induced by `members` (all vertices when zone_of == nullptr): repeatedly the
// uncoloured vertex with the largest saturation, ties by larger degree INSIDE the subgraph, then by
// smaller id; smallest colour no neighbour inside the subgraph has.  color[v] (0-based) is written for
// the members only; returns the number of colours.  Ordered-set formulation, keyed by
// (-saturation, -degree, id).
int32_t dsatur_subgraph(const std::vector<std::vector<int32_t>> &adj, const std::vector<int64_t> &members,
                        const std::vector<int32_t> *zone_of, int32_t zone, std::vector<int32_t> &color) {
    auto inside = [&](int32_t w) { return !zone_of || (*zone_of)[w] == zone; };
    std::unordered_map<int64_t, int64_t> local;  // vertex -> position in members
    local.reserve(members.size() * 2);
    for (size_t k = 0; k < members.size(); k++) local[members[k]] = (int64_t)k;
    const size_t n = members.size();
    std::vector<int64_t> deg(n, 0);
    std::vector<int32_t> sat(n, 0);
    std::vector<std::vector<int32_t>> seen(n);  // sorted colours among neighbours
    for (size_t k = 0; k < n; k++)
        for (int32_t w : adj[members[k]]) deg[k] += inside(w);
    using Key = std::tuple<int32_t, int64_t, int64_t>;  // (-sat, -deg, id)
    std::set<Key> queue;
    for (size_t k = 0; k < n; k++) queue.insert(Key(0, -deg[k], members[k]));
    int32_t ncolors = 0;
    while (!queue.empty()) {
        auto it = queue.begin();
        const int64_t v = std::get<2>(*it);
        queue.erase(it);
        const int64_t kv = local[v];
        int32_t c = 0;
        for (int32_t sc : seen[kv]) {  // sorted: first gap
            if (sc == c)
                c++;
            else if (sc > c)
                break;
        }
        color[v] = c;
        ncolors = std::max(ncolors, c + 1);
        for (int32_t w : adj[v]) {
            if (!inside(w) || color[w] >= 0) continue;
            const int64_t kw = local[w];
            auto pos = std::lower_bound(seen[kw].begin(), seen[kw].end(), c);
            if (pos != seen[kw].end() && *pos == c) continue;
            queue.erase(Key(-sat[kw], -deg[kw], w));
            seen[kw].insert(pos, c);
            sat[kw]++;
            queue.insert(Key(-sat[kw], -deg[kw], w));
        }
    }
    return ncolors;
}

}  // namespace

std::vector<std::vector<int64_t>> color_dsatur(const std::vector<const int64_t *> &lists,
                                               const std::vector<int64_t> &lens) {
    const int64_t nb = (int64_t)lists.size();
    std::vector<std::vector<int64_t>> classes;
    if (nb == 0) return classes;
    const auto adj = conflict_adjacency(lists, lens);
    std::vector<int64_t> all(nb);
    std::iota(all.begin(), all.end(), (int64_t)0);
    std::vector<int32_t> color(nb, -1);
    const int32_t ncolors = dsatur_subgraph(adj, all, nullptr, 0, color);
    classes.assign(ncolors, {});
    for (int64_t v = 0; v < nb; v++) classes[color[v]].push_back(v + 1);
    return classes;
}

// WorkstreamDSATUR -- the reference's default (src/BlockSparseMatrices.jl:10) as published: Turcksin,
// Kronbichler, Bangerth, "WorkStream -- a design pattern for multicore-enabled finite element
// computations", ACM TOMS 43 (2016), section 3.2: (1) PARTITION the conflict graph into zones -- a seed,
// then repeatedly every not yet assigned neighbour of the previous zone, so that zone k only
// conflicts with zones k-1 and k+1; (2) COLOUR every zone on its own (DSATUR); (3) GATHER: zones of
// equal parity are mutually conflict-free, so their colour classes are merged -- the zone with the
// most colours founds the global classes of its parity, every other zone hands its classes, largest
// first, to the currently smallest global class it has not used yet.  Tie-breaking is fixed here
// (seed = smallest unassigned id; see oracle/bsm_oracle.c:orc_color_workstream for the full rules).
std::vector<std::vector<int64_t>> color_workstream_dsatur(const std::vector<const int64_t *> &lists,
                                                          const std::vector<int64_t> &lens) {
    const int64_t nb = (int64_t)lists.size();
    std::vector<std::vector<int64_t>> classes;
    if (nb == 0) return classes;
    const auto adj = conflict_adjacency(lists, lens);
    // (1) zones
    std::vector<int32_t> zone_of(nb, -1);
    std::vector<std::vector<int64_t>> zones;
    int64_t next_seed = 0;
    while (true) {
        while (next_seed < nb && zone_of[next_seed] >= 0) next_seed++;
        if (next_seed >= nb) break;
        std::vector<int64_t> cur{next_seed};
        zone_of[next_seed] = (int32_t)zones.size();
        while (!cur.empty()) {
            const int32_t z = (int32_t)zones.size();
            std::vector<int64_t> nxt;
            for (int64_t v : cur)
                for (int32_t w : adj[v])
                    if (zone_of[w] < 0) {
                        zone_of[w] = z + 1;
                        nxt.push_back(w);
                    }
            std::sort(nxt.begin(), nxt.end());
            zones.push_back(std::move(cur));
            cur = std::move(nxt);
        }
    }
    // (2) DSATUR inside every zone
    std::vector<int32_t> color(nb, -1);
    std::vector<int32_t> zcolors(zones.size());
    for (size_t z = 0; z < zones.size(); z++) zcolors[z] = dsatur_subgraph(adj, zones[z], &zone_of, (int32_t)z, color);
    // (3) gather, parity by parity
    std::vector<int32_t> global(nb, -1);
    int32_t base = 0;
    for (int parity = 0; parity < 2; parity++) {
        size_t zmax = zones.size();
        for (size_t z = parity; z < zones.size(); z += 2)
            if (zmax == zones.size() || zcolors[z] > zcolors[zmax]) zmax = z;
        if (zmax == zones.size()) continue;
        const int32_t K = zcolors[zmax];
        std::vector<int64_t> gsize(K, 0);
        for (int64_t v : zones[zmax]) {
            global[v] = base + color[v];
            gsize[color[v]]++;
        }
        for (size_t z = parity; z < zones.size(); z += 2) {
            if (z == zmax) continue;
            std::vector<int64_t> csize(zcolors[z], 0);
            for (int64_t v : zones[z]) csize[color[v]]++;
            std::vector<int32_t> order(zcolors[z]);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return csize[a] > csize[b]; });
            std::vector<int32_t> target(zcolors[z], -1);
            std::vector<char> used(K, 0);
            for (int32_t c : order) {
                int32_t best = -1;
                for (int32_t g = 0; g < K; g++)
                    if (!used[g] && (best < 0 || gsize[g] < gsize[best])) best = g;
                used[best] = 1;
                gsize[best] += csize[c];
                target[c] = best;
            }
            for (int64_t v : zones[z]) global[v] = base + target[color[v]];
        }
        base += K;
    }
    classes.assign(base, {});
    for (int64_t v = 0; v < nb; v++) classes[global[v]].push_back(v + 1);
    return classes;
}

// ---------------------------------------------------------------------------------------
namespace {

struct Chunk {
    int32_t blk;    // input block
    int32_t ra;     // first row of the block in this chunk
    int32_t mc;     // rows (<= 64)
    int64_t group;  // row group id
    int64_t woff;   // first column of this chunk inside the group's merged panel
};

// A row group = every chunk living on the same y rows (for a symmetric operator the diagonal block
// of a row set and its off-diagonal blocks share one group: the kind is kept per COLUMN).  Its
// chunks are concatenated column-wise into ONE merged panel, so a wave always streams a single piece.
struct Group {
    int32_t mc = 0;
    int32_t kind = 0;      // kind of its non-diagonal columns (KIND_OFF if it has any, else the chunks' kind)
    bool has_off = false, has_diag = false;
    int32_t rbase = -1;    // 0-based first row when contiguous
    int32_t row_off = -1;  // rows pool offset when indexed
    std::vector<int32_t> chunks;
    int64_t width = 0;     // merged columns
    int64_t strips = 0;    // ceil(width / E)
    int64_t col_off = 0;   // cols pool offset of the merged column list
    uint64_t val_off = 0;  // 16-byte units
};

struct Item {
    int64_t group;
    int64_t s_begin, s_end;  // strips of the group's merged panel
    int64_t bytes;
    int nw;
    int32_t color;
};

// scattered variant: block column w lands at merged panel column dstpos[w] (the merged column
// list of a row group with index lists is kept SORTED by x index, see build())
template <typename U>
void pack_chunk_perm(const U *src, int64_t ld, int ra, int mc, int64_t n, const int32_t *dstpos, int E,
                     bool trans, U *dst) {
    for (int64_t w = 0; w < n; w++) {
        const int64_t mp = dstpos[w];
        U *d = dst + ((mp / E) * mc) * E + (mp % E);
        if (!trans) {
            const U *col = src + ra + w * ld;
            for (int i = 0; i < mc; i++) d[(int64_t)i * E] = col[i];
        } else {
            const U *row = src + w + (int64_t)ra * ld;
            for (int i = 0; i < mc; i++) d[(int64_t)i * E] = row[(int64_t)i * ld];
        }
    }
}

template <typename U>
void pack_chunk(const U *src, int64_t ld, int ra, int mc, int64_t n, int64_t woff, int E, bool trans,
                U *dst) {
    // dst[(s*mc + i)*E + e] = B[ra+i, w],  s*E + e = woff + w;
    // B[r, w] = src[r + w*ld] (stored as is) or src[w + r*ld] (logical block = transpose of storage)
    // Strip-major: every 16-byte lane unit of the destination is written once, sequentially.
    const int64_t s_first = woff / E, s_last = (woff + n - 1) / E;
    for (int64_t s = s_first; s <= s_last; s++) {
        const int64_t w0 = s * E - woff;  // block column of e = 0 (may be < 0 / >= n at the ends)
        const int e_lo = (int)std::max<int64_t>(0, -w0);
        const int e_hi = (int)std::min<int64_t>(E, n - w0);
        U *d = dst + (s * mc) * E;
        if (!trans) {
            for (int e = e_lo; e < e_hi; e++) {
                const U *col = src + ra + (w0 + e) * ld;
                for (int i = 0; i < mc; i++) d[(int64_t)i * E + e] = col[i];
            }
        } else {
            for (int i = 0; i < mc; i++) {
                const U *row = src + w0 + (int64_t)(ra + i) * ld;
                for (int e = e_lo; e < e_hi; e++) d[(int64_t)i * E + e] = row[e];
            }
        }
    }
}

struct U16 {
    uint64_t a, b;
};

uint64_t hash_list(const int64_t *p, int64_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    for (int64_t i = 0; i < n; i++) {
        h ^= (uint64_t)p[i];
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    }
    return h;
}

}  // namespace

// Working state shared by the stages of Analysis::build (each stage below is one step of the host
// analysis; the CPU image-interpreter tests of tests/test_host_logic.py check their combined result).
struct Analysis::BuildState {
    bool sym = false;      // symmetric operator (or the symmetric view of a VBCRS)
    bool colored = false;  // BSM_ACC_COLORED
    int64_t nb = 0;
    int64_t own_lo = 0, own_hi = 0;  // 0-based [lo, hi): rows this handle scales by beta
    std::vector<Chunk> chunks;
    std::vector<Group> groups;
    std::vector<const int64_t *> glist;  // representative index list of indexed groups
    std::vector<int32_t> colpos;         // merged panel column of every column in chunk order
    std::vector<uint8_t> ckind;          // kind of every merged panel column (parallel to cols)
    std::vector<uint8_t> group_perm;     // 1: the group's merged columns were re-ordered (sorted by x index)
    std::vector<int64_t> layout;         // row groups in the order their panels lie in the value stream
    uint64_t val_units = 0;              // 16-byte units of the whole value stream
    std::vector<uint8_t> cover;          // rows some group produces
    std::vector<int32_t> group_color;
    int32_t ncolors_fused = 1;
    std::vector<Item> items;
    bool use_window = false;
};

std::string Analysis::build(int mtype_, int dtype_, int64_t nrows_, int64_t ncols_,
                            const std::vector<BlockIn> &blocks, const AnalysisOptions &opt_) {
    mtype = mtype_;
    dtype = dtype_;
    nrows = nrows_;
    ncols = ncols_;
    opt = opt_;
    tun = Tunables::from_env();
    const bool timing = std::getenv("BSM_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bsm analysis] %-28s %8.1f ms\n", what,
                     std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    BuildState st;
    std::string err = stage_validate(blocks, st);
    if (!err.empty()) return err;
    lap("validate");
    // the reference colourings are independent of everything the GPU path needs (they only read the
    // caller's index lists and fill `colors`): they run on their own thread while the values are packed
    // and uploaded
    bool colour_oom = false;
    if (opt.meta_only) {  // whole-operator bookkeeping of a multi-device handle: no image
        reference_colourings(blocks, colour_oom);
        return colour_oom ? "out of host memory (colouring)" : "";
    }
    if (!(err = stage_row_groups(blocks, st)).empty()) return err;
    if (!(err = stage_merge_columns(blocks, st)).empty()) return err;
    lap("groups + column lists");
    if (!(err = stage_accumulation(blocks, st)).empty()) return err;
    lap("exclusivity / fused colours");
    std::thread colour_thread([&] { reference_colourings(blocks, colour_oom); });
    struct Joiner {
        std::thread &t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } colour_join{colour_thread};  // also on the error returns below
    stage_work_items(st);
    stage_place_values(st);
    lap("work items + value placement");
    if (!(err = stage_pack_values(blocks, st)).empty()) return err;
    lap("pack values");
    stage_waves(st);
    stage_windows(st);
    if (!(err = stage_gather_index(st)).empty()) return err;
    lap("schedule");
    colour_thread.join();
    lap("reference colourings (tail not hidden by packing)");
    if (colour_oom) return "out of host memory (colouring)";
    return "";
}

// ---- validation, statistics ----------------------------------------------------------------------
std::string Analysis::stage_validate(const std::vector<BlockIn> &blocks, BuildState &st) {
    static const int kEs[4] = {4, 8, 8, 16};
    if (dtype < 0 || dtype > 3) return "unknown dtype";
    es = kEs[dtype];
    E = 16 / es;
    if (nrows < 0 || ncols < 0) return "negative matrix size";
    if (nrows > INT32_MAX - 64 || ncols > INT32_MAX - 64) return "matrix dimension exceeds int32 range";
    const int64_t nb = st.nb = (int64_t)blocks.size();
    bool sym = (mtype == MT_SYMMETRIC);
    for (const BlockIn &B : blocks) sym |= (B.kind != KIND_PLAIN);  // symmetric view of a VBCRS
    st.sym = sym;
    st.own_lo = (opt.own_lo > 0) ? opt.own_lo - 1 : 0;
    st.own_hi = (opt.own_hi > 0) ? std::min(opt.own_hi, nrows) : nrows;  // exclusive
    // rows index y for op N and x for op T; for a symmetric matrix every list indexes both
    const int64_t rlim = sym ? std::min(nrows, ncols) : nrows;
    const int64_t clim = sym ? std::min(nrows, ncols) : ncols;

    nnz = 0;
    stored_entries = 0;
    int64_t idx_meta = 0;
    for (int64_t b = 0; b < nb; b++) {
        const BlockIn &B = blocks[b];
        if (B.m < 0 || B.n < 0) return "block " + std::to_string(b + 1) + ": negative size";
        if (B.m > 0 && B.n > 0 && !B.data) return "block " + std::to_string(b + 1) + ": null data";
        if (B.ld < std::max<int64_t>(B.trans ? B.n : B.m, 1))
            return "block " + std::to_string(b + 1) + ": ld < m";
        if (B.m > 0 && !B.ridx && (B.r0 < 1 || B.r0 + B.m - 1 > rlim))
            return "block " + std::to_string(b + 1) + ": row range outside matrix";
        if (B.n > 0 && !B.cidx && (B.c0 < 1 || B.c0 + B.n - 1 > clim))
            return "block " + std::to_string(b + 1) + ": column range outside matrix";
        if (opt.validate) {
            if (B.ridx)
                for (int64_t i = 0; i < B.m; i++)
                    if (B.ridx[i] < 1 || B.ridx[i] > rlim)
                        return "block " + std::to_string(b + 1) + ": row index out of range";
            if (B.cidx)
                for (int64_t i = 0; i < B.n; i++)
                    if (B.cidx[i] < 1 || B.cidx[i] > clim)
                        return "block " + std::to_string(b + 1) + ": column index out of range";
        }
        stored_entries += B.m * B.n;
        nnz += (B.kind == KIND_OFF ? 2 : 1) * B.m * B.n;
        idx_meta += (B.kind == KIND_DIAG) ? B.m : (B.m + B.n);
    }
    // algorithmic bytes of one mul with beta = 0
    {
        int64_t meta;
        if (mtype == MT_VBCRS)
            meta = 8 * ((int64_t)rowptr.size() + (int64_t)rowindices.size() + 4 * nb);
        else
            meta = 8 * idx_meta;
        alg_bytes = stored_entries * es + meta + ncols * es + nrows * es;
    }

    return "";
}

// ---- colouring (reference bookkeeping): `colors` as the reference's constructors compute them ---------
void Analysis::reference_colourings(const std::vector<BlockIn> &blocks, bool &colour_oom) {
    const int64_t nb = (int64_t)blocks.size();
    {
        try {
            for (auto &c : colors) c.clear();
            if (opt.skip_colors) return;
            // the reference's `coloringalgorithm` keyword: WorkstreamDSATUR unless told otherwise
            auto refcolor = [&](const std::vector<const int64_t *> &l, const std::vector<int64_t> &n) {
                return opt.coloring == 1 ? color_dsatur(l, n) : color_workstream_dsatur(l, n);
            };
            auto single = [](int64_t n) {
                std::vector<std::vector<int64_t>> out(1);
                out[0].resize(n);
                std::iota(out[0].begin(), out[0].end(), (int64_t)1);
                return out;
            };
            if (mtype == MT_BLOCKSPARSE) {
                if (opt.scheduler == 0) {  // reference src/blockmatrix.jl:91-92
                    colors[0] = single(nb);
                    colors[1] = single(nb);
                } else {  // src/blockmatrix.jl:94-98
                    std::vector<const int64_t *> rl(nb), cl(nb);
                    std::vector<int64_t> rn(nb), cn(nb);
                    for (int64_t b = 0; b < nb; b++) {
                        rl[b] = blocks[b].ridx;
                        rn[b] = blocks[b].ridx ? blocks[b].m : 0;
                        cl[b] = blocks[b].cidx;
                        cn[b] = blocks[b].cidx ? blocks[b].n : 0;
                    }
                    colors[0] = refcolor(rl, rn);
                    colors[1] = refcolor(cl, cn);
                }
            } else if (mtype == MT_SYMMETRIC) {
                // always three colourings, also for the serial scheduler: src/symmetricblockmatrix.jl:104-110
                std::vector<const int64_t *> dl, rl, cl;
                std::vector<int64_t> dn, rn, cn;
                for (const BlockIn &B : blocks) {
                    if (B.kind == KIND_DIAG) {
                        dl.push_back(B.ridx);
                        dn.push_back(B.ridx ? B.m : 0);
                    } else {
                        rl.push_back(B.ridx);
                        rn.push_back(B.ridx ? B.m : 0);
                        cl.push_back(B.cidx);
                        cn.push_back(B.cidx ? B.n : 0);
                    }
                }
                colors[0] = refcolor(rl, rn);
                colors[1] = refcolor(cl, cn);
                colors[2] = refcolor(dl, dn);
            }
        } catch (const std::bad_alloc &) {
            colour_oom = true;
        }
    }
}

// ---- chunks (<= 64 rows) and row groups ------------------------------------------------------------
std::string Analysis::stage_row_groups(const std::vector<BlockIn> &blocks, BuildState &st) {
    const int64_t nb = st.nb;
    auto &chunks = st.chunks;
    auto &groups = st.groups;
    auto &glist = st.glist;
    const int chunk_rows = tun.chunk_rows;
    std::unordered_map<uint64_t, std::vector<int64_t>> gmap;  // hash -> candidate groups

    rows.clear();
    cols.clear();
    for (int64_t b = 0; b < nb; b++) {
        const BlockIn &B = blocks[b];
        if (B.m == 0 || B.n == 0) continue;
        for (int64_t ra = 0; ra < B.m; ra += chunk_rows) {
            const int mc = (int)std::min<int64_t>(chunk_rows, B.m - ra);
            bool rcontig = true;
            if (B.ridx)
                for (int i = 1; i < mc; i++)
                    if (B.ridx[ra + i] != B.ridx[ra] + i) {
                        rcontig = false;
                        break;
                    }
            const int64_t rbase = rcontig ? (B.ridx ? B.ridx[ra] : B.r0 + ra) - 1 : -1;
            uint64_t h;
            if (rcontig)
                h = ((uint64_t)rbase * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)mc << 56) ^ 0x51ull;
            else
                h = hash_list(B.ridx + ra, mc);
            // plain blocks never share a group with symmetric ones; DIAG and OFF chunks of the same
            // rows do (per-column kinds)
            const int gkind = (B.kind == KIND_PLAIN) ? KIND_PLAIN : KIND_OFF;
            h ^= (uint64_t)gkind * 0xD6E8FEB86659FD93ull;
            int64_t gid = -1;
            auto &cand = gmap[h];
            for (int64_t g : cand) {
                const Group &G = groups[g];
                if (G.mc != mc || ((G.kind == KIND_PLAIN) != (B.kind == KIND_PLAIN))) continue;
                if (rcontig) {
                    if (G.rbase == rbase) gid = g;
                } else if (G.rbase < 0 &&
                           std::memcmp(glist[g], B.ridx + ra, sizeof(int64_t) * mc) == 0) {
                    gid = g;
                }
                if (gid >= 0) break;
            }
            if (gid < 0) {
                gid = (int64_t)groups.size();
                Group G;
                G.mc = mc;
                G.kind = B.kind;
                G.rbase = (int32_t)rbase;
                if (!rcontig) {
                    if ((int64_t)rows.size() + mc + 8 > INT32_MAX) return "row index pool exceeds int32";
                    G.row_off = (int32_t)rows.size();
                    for (int i = 0; i < mc; i++) rows.push_back((int32_t)(B.ridx[ra + i] - 1));
                }
                groups.push_back(G);
                glist.push_back(rcontig ? nullptr : B.ridx + ra);
                cand.push_back(gid);
            }
            Chunk c;
            c.blk = (int32_t)b;
            c.ra = (int32_t)ra;
            c.mc = mc;
            c.group = gid;
            c.woff = groups[gid].width;
            groups[gid].width += B.n;
            if (B.kind == KIND_OFF) {
                groups[gid].has_off = true;
                groups[gid].kind = KIND_OFF;
            } else if (B.kind == KIND_DIAG) {
                groups[gid].has_diag = true;
            }
            groups[gid].chunks.push_back((int32_t)chunks.size());
            chunks.push_back(c);
        }
    }
    ngroups = (int64_t)groups.size();
    return "";
}

// ---- merged column list of every row group -----------------------------------------------------------
std::string Analysis::stage_merge_columns(const std::vector<BlockIn> &blocks, BuildState &st) {
    auto &chunks = st.chunks;
    auto &groups = st.groups;
    auto &colpos = st.colpos;
    auto &ckind = st.ckind;
    auto &group_perm = st.group_perm;
    uint64_t val_units = 0;

    {
        int64_t total_cols = 0;
        for (const Group &G : groups) total_cols += G.width + E;
        if (total_cols + 8 > INT32_MAX) return "column index pool exceeds int32";
        cols.reserve((size_t)total_cols);
    }
    // colpos[G.col_off + q]: merged panel column of the q-th column in chunk order.  The order of
    // the columns inside a panel is free (a sum), so the merged column list is kept SORTED by x
    // index: neighbouring lanes then gather neighbouring x entries and emit neighbouring y
    // entries, and scattered index lists (BEM near-field panels) collapse into contiguous runs.
    group_perm.assign(groups.size(), 0);
    colpos.reserve(cols.capacity());
    ckind.reserve(cols.capacity());
    {
        std::vector<int32_t> ord;
        size_t gi = 0;
        for (Group &G : groups) {
            G.strips = (G.width + E - 1) / E;
            if (G.strips * (int64_t)G.mc * 16 >= ((int64_t)1 << 31)) return "row group panel exceeds 2 GiB";
            G.col_off = (int64_t)cols.size();
            for (int32_t ci : G.chunks) {
                const BlockIn &B = blocks[chunks[ci].blk];
                for (int64_t k = 0; k < B.n; k++) {
                    cols.push_back((int32_t)((B.cidx ? B.cidx[k] : B.c0 + k) - 1));
                    ckind.push_back((uint8_t)B.kind);
                }
            }
            int32_t *gc = cols.data() + G.col_off;
            uint8_t *gk = ckind.data() + G.col_off;
            colpos.resize(cols.size());
            int32_t *gp = colpos.data() + G.col_off;
            if (std::is_sorted(gc, gc + G.width)) {
                for (int64_t q = 0; q < G.width; q++) gp[q] = (int32_t)q;
            } else {
                group_perm[gi] = 1;
                ord.resize((size_t)G.width);
                std::iota(ord.begin(), ord.end(), 0);
                std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return gc[a] < gc[b]; });
                std::vector<int32_t> sorted((size_t)G.width);
                std::vector<uint8_t> sortedk((size_t)G.width);
                for (int64_t r = 0; r < G.width; r++) {
                    sorted[r] = gc[ord[r]];
                    sortedk[r] = gk[ord[r]];
                    gp[ord[r]] = (int32_t)r;
                }
                std::copy(sorted.begin(), sorted.end(), gc);
                std::copy(sortedk.begin(), sortedk.end(), gk);
            }
            val_units += (uint64_t)G.mc * (uint64_t)G.strips;
            gi++;
        }
    }
    st.val_units = val_units;
    value_bytes = (int64_t)val_units * 16;
    return "";
}

// ---- forward exclusivity, coverage, accumulation mode ---------------------------------------------------
std::string Analysis::stage_accumulation(const std::vector<BlockIn> &blocks, BuildState &st) {
    auto &groups = st.groups;
    auto &cover = st.cover;
    auto &group_color = st.group_color;
    const bool sym = st.sym;
    const uint64_t val_units = st.val_units;
    cover.assign((size_t)nrows, 0);

    bool excl = true;
    for (const Group &G : groups) {
        for (int i = 0; i < G.mc; i++) {
            int64_t r = (G.rbase >= 0) ? (int64_t)G.rbase + i : rows[G.row_off + i];
            if (cover[r]) excl = false;
            cover[r] = 1;
        }
    }
    // a symmetric matrix adds transposed contributions into the same y: never exclusive
    if (sym) {
        bool any_off = false;
        for (const BlockIn &B : blocks) any_off |= (B.kind == KIND_OFF && B.m > 0 && B.n > 0);
        if (any_off) excl = false;
    }
    exclusive_fwd = excl;
    // forced atomics: schedule like a non-exclusive matrix (large row groups are cut into several
    // workgroup items; their partial sums meet in y through atomics)
    if (opt.accumulate == 1 || opt.accumulate == 3) exclusive_fwd = false;
    if (opt.accumulate == 0 && exclusive_fwd) {
        // auto: a large conflict-free operator made of DEEP row groups (a group has to stay in one
        // workgroup of 4 waves to keep its rows exclusive: 128 KB per wave for C4's 512 KB groups)
        // streams 8 % faster as 32 KB work items combined with atomics -- shallow waves, many more
        // of them; the extra `y .*= beta` launch is noise at this size (C4 slice 328 -> 303 us)
        int64_t deep_units = 0;
        for (const Group &G : groups) {
            const int64_t u = (int64_t)G.mc * G.strips;
            if (u * 16 > tun.deep_group_bytes) deep_units += u;
        }
        if ((int64_t)val_units * 16 >= tun.deep_total_bytes && 2 * deep_units > (int64_t)val_units) exclusive_fwd = false;
    }
    const bool colored = st.colored = (opt.accumulate == 2);
    gather = (opt.accumulate == 3);
    group_color.assign(groups.size(), 0);
    int32_t &ncolors_fused = st.ncolors_fused;
    ncolors_fused = 1;
    if (colored) {
        exclusive_fwd = false;
        // fused conflict lists: a row group writes its rows (forward) and its columns (transposed)
        std::vector<std::vector<int64_t>> lists(groups.size());
        std::vector<const int64_t *> lp(groups.size());
        std::vector<int64_t> ln(groups.size());
        for (size_t g = 0; g < groups.size(); g++) {
            const Group &G = groups[g];
            auto &l = lists[g];
            for (int i = 0; i < G.mc; i++)
                l.push_back(1 + ((G.rbase >= 0) ? (int64_t)G.rbase + i : rows[G.row_off + i]));
            const size_t nrow = l.size();
            {
                std::vector<int64_t> r(l.begin(), l.end());
                std::sort(r.begin(), r.end());
                if (std::adjacent_find(r.begin(), r.end()) != r.end())
                    return "coloured accumulation: a block repeats a row index (use atomic mode)";
            }
            std::vector<int64_t> c;
            for (int64_t k = 0; k < G.width; k++) c.push_back(1 + (int64_t)cols[G.col_off + k]);
            std::sort(c.begin(), c.end());
            if (std::adjacent_find(c.begin(), c.end()) != c.end())
                return "coloured accumulation: blocks sharing rows repeat a column index (use atomic mode)";
            // columns live in x-space for op N and in y-space for op T; rows the other way round.
            // One colouring must serve every op, so rows and columns share one index space only
            // when the matrix is square-indexed (symmetric); otherwise offset the columns.
            const int64_t off = sym ? 0 : std::max(nrows, ncols) + 1;
            for (int64_t v : c) l.push_back(v + off);
            (void)nrow;
            lp[g] = l.data();
            ln[g] = (int64_t)l.size();
        }
        auto classes = color_dsatur(lp, ln);
        ncolors_fused = (int32_t)classes.size();
        for (size_t c = 0; c < classes.size(); c++)
            for (int64_t id : classes[c]) group_color[id - 1] = (int32_t)c;
    }
    return "";
}

// ---- work items: how many waves stream which strips of which row group, in dispatch order ---------------
void Analysis::stage_work_items(BuildState &st) {
    auto &groups = st.groups;
    auto &group_color = st.group_color;
    auto &items = st.items;
    const bool colored = st.colored, sym = st.sym;
    const uint64_t val_units = st.val_units;
    items.clear();

    {  // bytes per wave for this operator (bsm_analysis.h: Tunables::wave_bytes)
        int64_t W = tun.wave_bytes;
        // lane fill: rows of a group over the lanes its strips occupy (8, 16, 32 or 64 per strip)
        double rows_b = 0, lanes_b = 0;
        for (const Group &G : groups) {
            rows_b += (double)G.mc * (double)G.strips;
            lanes_b += (double)lanes_per_strip(G.mc) * (double)G.strips;
        }
        lane_fill = lanes_b > 0 ? rows_b / lanes_b : 1.0;
        {  // byte-weighted mean of the rows of a row group (what a strip of a typical streamed byte is tall)
            double w = 0, wr = 0;
            for (const Group &G : groups) {
                const double b = (double)G.mc * (double)G.strips;
                w += b;
                wr += b * (double)G.mc;
            }
            mean_rows = w > 0 ? wr / w : 64.0;
        }
        if (W <= 0) {
            const bool low_fill = lanes_b > 0 && rows_b < tun.fat_fill_below * lanes_b;
            W = tun.wave_bytes_min;
            if (low_fill) {
                W = std::min(tun.wave_bytes_max, (int64_t)val_units * 16 / tun.target_waves);
                if (4 * W < 5 * tun.wave_bytes_min) W = tun.wave_bytes_min;  // not long enough to pay
            }
            if (W > tun.wave_bytes_min) {
                // fat waves: a row group gets a second wave only from 2 W on and never four -- every
                // extra wave of a group is another fixed chain plus a workgroup barrier (tiled BEM
                // fixture: 4.84 TB/s with the W / 3 W rule, 5.0 with this one)
                if (tun.split2_bytes <= 0) tun.split2_bytes = 2 * W;
                if (tun.split4_bytes <= 0) tun.split4_bytes = INT64_MAX / 4;
            }
        }
        if (tun.split2_bytes <= 0) tun.split2_bytes = W;
        if (tun.split4_bytes <= 0) tun.split4_bytes = 3 * W;
        if (tun.wgitem_max_bytes <= 0) tun.wgitem_max_bytes = 4 * W;
    }
    for (int64_t g = 0; g < ngroups; g++) {
        const Group &G = groups[g];
        const int64_t strip_bytes = (int64_t)G.mc * 16;
        int64_t per_item = G.strips;
        if (!exclusive_fwd && !colored) {
            int64_t maxs = std::max<int64_t>(1, tun.wgitem_max_bytes / strip_bytes);
            int64_t nitem = (G.strips + maxs - 1) / maxs;
            per_item = (G.strips + nitem - 1) / nitem;
        }
        for (int64_t s = 0; s < G.strips; s += per_item) {
            Item it;
            it.group = g;
            it.s_begin = s;
            it.s_end = std::min(G.strips, s + per_item);
            it.bytes = (it.s_end - it.s_begin) * strip_bytes;
            it.nw = it.bytes >= tun.split4_bytes ? 4 : (it.bytes >= tun.split2_bytes ? 2 : 1);
            it.color = group_color[g];
            items.push_back(it);
        }
    }
    // symmetric operators: small items (1 or 2 waves) are ordered by LOCALITY (first row of their
    // group) so that the waves of one workgroup touch neighbouring y entries and can share an LDS
    // accumulation window; everything else largest-first
    const bool use_window = st.use_window = sym && !colored && !gather && !exclusive_fwd && tun.lds_window;
    auto locality = [&](const Item &it) -> int64_t {
        const Group &G = groups[it.group];
        return (G.rbase >= 0) ? G.rbase : rows[G.row_off];
    };
    std::stable_sort(items.begin(), items.end(), [&](const Item &a, const Item &b) {
        if (a.color != b.color) return a.color < b.color;
        if (a.nw != b.nw) return a.nw > b.nw;
        if (use_window && a.nw < 4) return locality(a) < locality(b);
        return a.bytes > b.bytes;
    });
}

// ---- value stream layout: the panels lie in the order in which the launch reaches them --------------------
// Workgroups are dispatched in index order, so the waves in flight at any moment stream NEIGHBOURING
// addresses when the value stream follows the dispatch order (a stream laid out in creation order and
// walked in locality or size order lost a third of its rate on the tiled BEM fixture).
void Analysis::stage_place_values(BuildState &st) {
    auto &groups = st.groups;
    std::vector<uint8_t> placed(groups.size(), 0);
    st.layout.clear();
    st.layout.reserve(groups.size());
    for (const Item &it : st.items)
        if (!placed[it.group]) {
            placed[it.group] = 1;
            st.layout.push_back(it.group);
        }
    for (size_t g = 0; g < groups.size(); g++)
        if (!placed[g]) st.layout.push_back((int64_t)g);
    uint64_t off = 0;
    for (int64_t g : st.layout) {
        groups[g].val_off = off;
        off += (uint64_t)groups[g].mc * (uint64_t)groups[g].strips;
    }
}

// ---- pack values: every stored entry once, strip order (bsm_layout.h) ---------------------------------------
std::string Analysis::stage_pack_values(const std::vector<BlockIn> &blocks, BuildState &st) {
    auto &chunks = st.chunks;
    auto &groups = st.groups;
    auto &colpos = st.colpos;
    auto &group_perm = st.group_perm;
    const uint64_t val_units = st.val_units;

    auto pack_one = [&](const Chunk &c, char *dst) {  // dst: first byte of the chunk's row group panel
        const BlockIn &B = blocks[c.blk];
        if (group_perm[c.group]) {
            const int32_t *dp = colpos.data() + groups[c.group].col_off + c.woff;
            if (es == 4)
                pack_chunk_perm<uint32_t>((const uint32_t *)B.data, B.ld, c.ra, c.mc, B.n, dp, E, B.trans, (uint32_t *)dst);
            else if (es == 8)
                pack_chunk_perm<uint64_t>((const uint64_t *)B.data, B.ld, c.ra, c.mc, B.n, dp, E, B.trans, (uint64_t *)dst);
            else
                pack_chunk_perm<U16>((const U16 *)B.data, B.ld, c.ra, c.mc, B.n, dp, E, B.trans, (U16 *)dst);
            return;
        }
        if (es == 4)
            pack_chunk<uint32_t>((const uint32_t *)B.data, B.ld, c.ra, c.mc, B.n, c.woff, E, B.trans, (uint32_t *)dst);
        else if (es == 8)
            pack_chunk<uint64_t>((const uint64_t *)B.data, B.ld, c.ra, c.mc, B.n, c.woff, E, B.trans, (uint64_t *)dst);
        else
            pack_chunk<U16>((const U16 *)B.data, B.ld, c.ra, c.mc, B.n, c.woff, E, B.trans, (U16 *)dst);
    };
    // packs the chunks `ids` (all of them when null); base = address of stream byte `wbase`
    auto pack_many = [&](const std::vector<int32_t> *ids, size_t nchunks, char *base, size_t wbase) {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(tun.pack_threads, (int64_t)nchunks / 64 + 1));
        auto worker = [&](int t) {
            for (size_t k = t; k < nchunks; k += nt) {
                const Chunk &c = chunks[ids ? (size_t)(*ids)[k] : k];
                pack_one(c, base + ((size_t)groups[c.group].val_off * 16 - wbase));
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto &x : th) x.join();
    };
    auto zero_tail = [&](const Group &G, char *base, size_t wbase) {  // padded tail of a panel's last strip
        if (G.strips * E - G.width > 0)
            std::memset(base + (((size_t)G.val_off + (size_t)(G.strips - 1) * G.mc) * 16 - wbase), 0, (size_t)G.mc * 16);
    };
    bool streamed = false;
    pack_plan.clear();
    pack_colpos.clear();
    if (opt.blocks_on_device) {
        // the blocks live on the device: leave a plan, the caller runs the pack kernel
        pack_plan.reserve(chunks.size());
        bool any_perm = false;
        for (const Chunk &c : chunks) {
            const BlockIn &B = blocks[c.blk];
            PackChunk pc;
            pc.src = (uint64_t)(uintptr_t)B.data;
            pc.dst_unit = groups[c.group].val_off;
            pc.ld = B.ld;
            pc.ra = c.ra;
            pc.mc = c.mc;
            pc.n = (int32_t)B.n;
            pc.woff = (int32_t)c.woff;
            pc.perm_off = group_perm[c.group] ? (int32_t)(groups[c.group].col_off + c.woff) : -1;
            pc.trans = B.trans ? 1 : 0;
            any_perm |= pc.perm_off >= 0;
            pack_plan.push_back(pc);
        }
        if (any_perm) pack_colpos = colpos;
    } else if (opt.sink) {
        std::string err = opt.sink->begin((size_t)value_bytes, &streamed);
        if (!err.empty()) return err;
    }
    if (opt.blocks_on_device) {
        // nothing to pack here
    } else if (streamed) {
        // windows of consecutive row groups of the stream (layout order), packed into the sink's staging
        // buffer by all threads, shipped while the next window is being packed
        const std::vector<int64_t> &lay = st.layout;
        std::vector<int32_t> ids;
        size_t g0 = 0;
        while (g0 < lay.size()) {
            const size_t wbase = (size_t)groups[lay[g0]].val_off * 16;
            size_t g1 = g0, wbytes = 0;
            while (g1 < lay.size() && (g1 == g0 || wbytes + (size_t)groups[lay[g1]].mc * groups[lay[g1]].strips * 16 <= tun.window_bytes)) {
                wbytes += (size_t)groups[lay[g1]].mc * (size_t)groups[lay[g1]].strips * 16;
                g1++;
            }
            char *buf = opt.sink->window(wbytes);
            if (!buf) return "value sink: no staging buffer";
            ids.clear();
            for (size_t g = g0; g < g1; g++) {
                zero_tail(groups[lay[g]], buf, wbase);
                ids.insert(ids.end(), groups[lay[g]].chunks.begin(), groups[lay[g]].chunks.end());
            }
            pack_many(&ids, ids.size(), buf, wbase);
            std::string err = opt.sink->commit(wbase, wbytes);
            if (!err.empty()) return err;
            g0 = g1;
        }
        std::string err = opt.sink->end();
        if (!err.empty()) return err;
    } else {
        values.allocate((size_t)val_units * 16);
        for (const Group &G : groups) zero_tail(G, values.data(), 0);
        pack_many(nullptr, chunks.size(), values.data(), 0);
    }
    return "";
}

// ---- waves: every wave streams ONE piece (a strip range of a merged panel) --------------------------------
void Analysis::stage_waves(BuildState &st) {
    auto &groups = st.groups;
    auto &ckind = st.ckind;
    auto &cover = st.cover;
    auto &items = st.items;
    const bool colored = st.colored;
    const int32_t ncolors_fused = st.ncolors_fused;
    const int64_t own_lo = st.own_lo, own_hi = st.own_hi;

    waves.clear();
    std::vector<WaveWork> *nop_out = &waves;
    auto emit_nop = [&]() {
        WaveWork w;
        std::memset(&w, 0, sizeof w);
        w.work = WORK_NOP;
        w.grp = 1;
        w.rbase = -1;
        nop_out->push_back(w);
    };
    color_wg_ptr.clear();
    int32_t cur_color = -1;
    std::vector<WaveWork> *out = &waves;  // the list item_waves() appends to
    auto item_waves = [&](const Item &it) {
        std::vector<WaveWork> &waves = *out;
        // align the start of a multi-wave item to its group size inside the workgroup
        while ((int)(waves.size() % kWavesPerWg) % it.nw != 0) emit_nop();
        const Group &G = groups[it.group];
        const int64_t total = it.s_end - it.s_begin;
        const int64_t per_wave = (total + it.nw - 1) / it.nw;
        for (int w = 0; w < it.nw; w++) {
            const int64_t wa = it.s_begin + std::min<int64_t>(total, (int64_t)w * per_wave);
            const int64_t wb = it.s_begin + std::min<int64_t>(total, (int64_t)(w + 1) * per_wave);
            WaveWork W;
            std::memset(&W, 0, sizeof W);
            W.work = WORK_PANEL;
            W.m = (uint16_t)G.mc;
            W.grp = (uint8_t)it.nw;
            W.lead = (w == 0) ? 1 : 0;
            W.rbase = G.rbase;
            W.row_off = G.row_off;
            W.npieces = 0;
            W.first.kind = G.kind | (G.has_off ? kKindGroupHasOff : 0);
            if (wb > wa) {
                Piece &P = W.first;
                const int64_t c0 = wa * E;
                const int64_t c1 = std::min<int64_t>(G.width, wb * E);
                P.val_off = G.val_off + (uint64_t)wa * (uint64_t)G.mc;
                P.nstrips = (int32_t)(wb - wa);
                P.ncols = (int32_t)(c1 - c0);
                P.col_off = (int32_t)(G.col_off + c0);
                // contiguous runs of x (of one kind) inside [c0, c1): up to three are described inline
                int nseg = 1;
                int32_t segw[3] = {0, P.ncols, P.ncols};
                int32_t segx[3] = {cols[G.col_off + c0], 0, 0};
                int32_t segk[3] = {ckind[G.col_off + c0], 0, 0};
                bool piece_off = ckind[G.col_off + c0] == KIND_OFF;
                for (int64_t k = c0 + 1; k < c1; k++) {
                    piece_off |= ckind[G.col_off + k] == KIND_OFF;
                    if (nseg <= 3 && (cols[G.col_off + k] != cols[G.col_off + k - 1] + 1 ||
                                      ckind[G.col_off + k] != ckind[G.col_off + k - 1])) {
                        if (nseg < 3) {
                            segw[nseg] = (int32_t)(k - c0);
                            segx[nseg] = cols[G.col_off + k];
                            segk[nseg] = ckind[G.col_off + k];
                        }
                        nseg++;
                    }
                }
                if (nseg <= 3) {
                    P.xbase = segx[0];
                    W.seg1_w = segw[1];
                    W.seg1_x = segx[1];
                    W.seg2_w = segw[2];
                    P.seg2_x = segx[2];
                    P.kind = segk[0] | (segk[1] << 2) | (segk[2] << 4);
                } else {
                    P.xbase = -1;
                    W.seg1_w = W.seg2_w = P.ncols;
                    P.kind = G.kind;  // kind of the columns without the diagonal flag in the cols pool
                }
                if (piece_off) P.kind |= kKindHasOff;
                if (G.has_off) P.kind |= kKindGroupHasOff;
                W.npieces = 1;
            }
            waves.push_back(W);
        }
    };
    for (const Item &it : items) {
        if (colored && it.color != cur_color) {  // a colour class starts on a workgroup boundary
            while (waves.size() % kWavesPerWg) emit_nop();
            while ((int32_t)color_wg_ptr.size() <= it.color)
                color_wg_ptr.push_back((int64_t)waves.size() / kWavesPerWg);
            cur_color = it.color;
        }
        item_waves(it);
    }
    while (waves.size() % kWavesPerWg) emit_nop();
    nwg_main = (int64_t)waves.size() / kWavesPerWg;
    // dispatch order: largest first; for exclusive forward images every other block of 256 workgroups
    // (one per CU) is reversed, so that the same CUs do not receive the larger workgroup of every
    // layer (C2-sized operators: +2.5 %; neutral on long launches).  BSM_ORDER=0 keeps plain
    // largest-first.
    const bool snake = tun.wg_order != 0 && exclusive_fwd;
    if (!colored && snake && nwg_main > 2) {
        std::vector<WaveWork> re(waves.size());
        const int64_t blk = 256;
        int64_t k = 0;
        for (int64_t a0 = 0; a0 < nwg_main; a0 += blk) {
            const int64_t a1 = std::min(nwg_main, a0 + blk);
            for (int64_t j = 0; j < a1 - a0; j++, k++) {
                const int64_t src = ((a0 / blk) & 1) ? a1 - 1 - j : a0 + j;
                for (int w = 0; w < kWavesPerWg; w++) re[k * kWavesPerWg + w] = waves[src * kWavesPerWg + w];
            }
        }
        waves.swap(re);
    }
    if (colored) {
        while ((int32_t)color_wg_ptr.size() <= ncolors_fused) color_wg_ptr.push_back(nwg_main);
    }

    // ---- scale work for rows no group covers (exclusive forward launch) -------------------
    if (exclusive_fwd) {
        int64_t r = own_lo;
        while (r < own_hi) {
            if (cover[r]) {
                r++;
                continue;
            }
            int64_t e = r;
            while (e < own_hi && !cover[e] && e - r < kScaleRowsPerWave) e++;
            WaveWork W;
            std::memset(&W, 0, sizeof W);
            W.work = WORK_SCALE;
            W.grp = 1;
            W.rbase = (int32_t)r;
            W.first.ncols = (int32_t)(e - r);
            waves.push_back(W);
            r = e;
        }
        while (waves.size() % kWavesPerWg) emit_nop();
    }
    nwg_total = (int64_t)waves.size() / kWavesPerWg;
    auto mark_sync = [](std::vector<WaveWork> &ws) {
        for (size_t wg = 0; wg + kWavesPerWg <= ws.size(); wg += kWavesPerWg) {
            uint8_t sync = 0;
            for (int w = 0; w < kWavesPerWg; w++) sync |= (ws[wg + w].work == WORK_PANEL && ws[wg + w].grp > 1);
            for (int w = 0; w < kWavesPerWg; w++) ws[wg + w].wg_sync = sync;
        }
    };
    mark_sync(waves);

    // ---- the coarser split for multi-RHS products (Tunables::multi_wave_bytes) -------------
    // Same panels, same value stream, same order (the row groups in the order their panels lie in the
    // stream), cut with W = multi_wave_bytes: items of at most 4 W, 2 waves from W on, 4 from 3 W on.
    // Only for launches that accumulate with atomics (no colour classes, no gather slots, no scale work).
    waves_multi.clear();
    nwg_multi = 0;
    if (tun.multi_wave_bytes > 0 && !exclusive_fwd && !colored && !gather) {
        const int64_t Wm = tun.multi_wave_bytes;
        std::vector<Item> mi;
        for (int64_t g : st.layout) {
            const Group &G = groups[g];
            if (G.strips <= 0) continue;
            const int64_t strip_bytes = (int64_t)G.mc * 16;
            const int64_t maxs = std::max<int64_t>(1, 4 * Wm / strip_bytes);
            const int64_t nitem = (G.strips + maxs - 1) / maxs;
            const int64_t per_item = (G.strips + nitem - 1) / nitem;
            for (int64_t s = 0; s < G.strips; s += per_item) {
                Item it;
                it.group = g;
                it.s_begin = s;
                it.s_end = std::min(G.strips, s + per_item);
                it.bytes = (it.s_end - it.s_begin) * strip_bytes;
                it.nw = it.bytes >= 3 * Wm ? 4 : (it.bytes >= Wm ? 2 : 1);
                it.color = 0;
                mi.push_back(it);
            }
        }
        std::stable_sort(mi.begin(), mi.end(), [](const Item &a, const Item &b) { return a.nw > b.nw; });
        out = nop_out = &waves_multi;
        for (const Item &it : mi) item_waves(it);
        while (waves_multi.size() % kWavesPerWg) emit_nop();
        out = nop_out = &waves;
        size_t panel_main = 0, panel_multi = 0;
        for (const WaveWork &w : waves) panel_main += w.work == WORK_PANEL;
        for (const WaveWork &w : waves_multi) panel_multi += w.work == WORK_PANEL;
        if (4 * panel_multi > 3 * panel_main) {
            waves_multi.clear();  // (nearly) the same split: one list serves both
        } else {
            mark_sync(waves_multi);
            nwg_multi = (int64_t)waves_multi.size() / kWavesPerWg;
        }
    }
    if (rows.empty()) rows.push_back(0);
    if (cols.empty()) cols.push_back(0);
    if (values.empty()) {
        values.allocate(16);
        std::memset(values.data(), 0, 16);
    }
}

// ---- LDS y windows of workgroups that pack neighbouring small row groups of a symmetric operator ------------
void Analysis::stage_windows(BuildState &st) {
    auto &ckind = st.ckind;
    const bool use_window = st.use_window;
    win_emissions = win_inside = win_flushed = 0;

    if (use_window) {
        // The waves of such a workgroup stream NEIGHBOURING row groups: their forward rows and their
        // transposed (KIND_OFF) columns largely coincide (BEM near-field panels: 4 neighbouring leaves
        // name every y entry twice on average), so sums that fall into one dense index range are added
        // up in LDS and leave the CU once, as contiguous atomics.  The range need not hold EVERYTHING
        // the workgroup emits -- a panel usually has a few far columns -- it is the best-filled range
        // of at most window_entries(es) indices (sliding window over the sorted emissions); what falls
        // outside goes to y directly.
        const int64_t cap = window_entries(es);
        std::vector<int64_t> em;
        win_emissions = win_inside = win_flushed = 0;
        for (size_t wg = 0; wg + kWavesPerWg <= (size_t)nwg_main * kWavesPerWg; wg += kWavesPerWg) {
            int npanel = 0;
            int32_t first_rbase = INT32_MIN, first_rowoff = INT32_MIN;
            bool distinct = false;
            em.clear();
            for (int w = 0; w < kWavesPerWg; w++) {
                const WaveWork &W = waves[wg + w];
                if (W.work != WORK_PANEL || W.npieces == 0) continue;
                if (npanel == 0) {
                    first_rbase = W.rbase;
                    first_rowoff = W.row_off;
                } else if (W.rbase != first_rbase || W.row_off != first_rowoff) {
                    distinct = true;
                }
                npanel++;
                if (W.lead)
                    for (int i = 0; i < W.m; i++) em.push_back((W.rbase >= 0) ? (int64_t)W.rbase + i : rows[W.row_off + i]);
                for (int32_t k = 0; k < W.first.ncols; k++)
                    if (ckind[(size_t)W.first.col_off + k] == KIND_OFF) em.push_back(cols[W.first.col_off + k]);
            }
            win_emissions += (int64_t)em.size();
            // worth it only when different groups meet
            if (!distinct || em.empty()) continue;
            std::sort(em.begin(), em.end());
            size_t best_j = 0, best_k = 0, j = 0;
            for (size_t k = 0; k < em.size(); k++) {
                while (em[k] - em[j] >= cap) j++;
                if (k - j > best_k - best_j || k == 0) best_j = j, best_k = k;
            }
            const int64_t inside = (int64_t)(best_k - best_j + 1);
            int64_t uniq = 1;
            for (size_t k = best_j + 1; k <= best_k; k++) uniq += em[k] != em[k - 1];
            // the flush costs a barrier and ceil(span / 64) atomic wave-instructions: take the window
            // when it merges at least a fifth of what passes through it
            if (inside < 32 || 5 * uniq > 4 * inside) continue;
            const int64_t lo = em[best_j], span = em[best_k] - lo + 1;
            for (int w = 0; w < kWavesPerWg; w++) {
                waves[wg + w].win_base = (int32_t)lo;
                waves[wg + w].win_span8 = (uint8_t)((span + 7) / 8);
            }
            win_inside += inside;
            win_flushed += uniq;
        }
    }
}

// ---- gather mode: workspace slots + inverted indices; kind flags of the cols pool ---------------------------
std::string Analysis::stage_gather_index(BuildState &st) {
    auto &groups = st.groups;
    auto &ckind = st.ckind;

    inv_ptr[0].clear();
    inv_ptr[1].clear();
    inv_idx[0].clear();
    inv_idx[1].clear();
    ws_fbase = ws_slots = 0;
    if (gather) {
        ws_fbase = (int64_t)cols.size();
        int64_t nf = 0;
        for (WaveWork &W : waves) {
            if (W.work != WORK_PANEL || !W.lead) continue;
            W.win_base = (int32_t)nf;  // forward slots of this workgroup item
            nf += W.m;
        }
        ws_slots = ws_fbase + nf;
        if (ws_slots + 8 > INT32_MAX) return "gather workspace exceeds int32 slots";
        const int64_t ylen[2] = {nrows, ncols};
        for (int k = 0; k < 2; k++) {  // k = 0: op N, k = 1: op T / C
            // per-row contribution lists first (CSR), then re-laid out as ELL per 64-row tile:
            // inv_ptr[t] = first 64-slot line of tile t, line l of a tile holds the l-th
            // contribution of each of its 64 rows (-1 = none) -> coalesced index reads
            std::vector<int64_t> ptr((size_t)ylen[k] + 1, 0);
            auto visit = [&](auto &&emit) {
                // transposed column sums: slot = position in the cols pool, ascending
                for (const Group &G : groups)
                    for (int64_t q = 0; q < G.width; q++)
                        if (k == 1 || ckind[G.col_off + q] == KIND_OFF) emit(cols[G.col_off + q], G.col_off + q);
                // forward partial sums of every workgroup item
                for (const WaveWork &W : waves) {
                    if (W.work != WORK_PANEL || !W.lead) continue;
                    const bool fwd = (k == 0) || (W.first.kind & kKindGroupHasOff);
                    if (!fwd) continue;
                    for (int i = 0; i < W.m; i++) {
                        const int64_t r = (W.rbase >= 0) ? (int64_t)W.rbase + i : rows[W.row_off + i];
                        emit(r, ws_fbase + W.win_base + i);
                    }
                }
            };
            visit([&](int64_t yi, int64_t) {
                if (yi < ylen[k]) ptr[yi + 1]++;
            });
            for (int64_t j = 0; j < ylen[k]; j++) ptr[j + 1] += ptr[j];
            std::vector<int32_t> csr((size_t)ptr[ylen[k]] + 1, 0);
            {
                std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
                visit([&](int64_t yi, int64_t slot) {
                    if (yi < ylen[k]) csr[fill[yi]++] = (int32_t)slot;
                });
            }
            const int64_t ntiles = (ylen[k] + 63) / 64;
            std::vector<int64_t> &tp = inv_ptr[k];
            tp.assign((size_t)ntiles + 1, 0);
            for (int64_t t = 0; t < ntiles; t++) {
                int64_t kmax = 0;
                for (int64_t j = t * 64; j < std::min(ylen[k], (t + 1) * 64); j++)
                    kmax = std::max(kmax, ptr[j + 1] - ptr[j]);
                tp[t + 1] = tp[t] + kmax;
            }
            if (tp[ntiles] * 64 + 64 > INT32_MAX) return "gather index exceeds int32";
            inv_idx[k].assign((size_t)tp[ntiles] * 64 + 64, -1);
            for (int64_t t = 0; t < ntiles; t++)
                for (int64_t j = t * 64; j < std::min(ylen[k], (t + 1) * 64); j++)
                    for (int64_t c = ptr[j]; c < ptr[j + 1]; c++)
                        inv_idx[k][(size_t)(tp[t] + (c - ptr[j])) * 64 + (size_t)(j - t * 64)] = csr[c];
        }
    }
    // the kernels' indexed path learns the kind of a column from the cols pool: flag the diagonal
    // columns of panels that also hold off-diagonal ones (done last: everything above reads cols)
    for (const Group &G : groups)
        if (G.has_off && G.has_diag)
            for (int64_t q = 0; q < G.width; q++)
                if (ckind[G.col_off + q] == KIND_DIAG) cols[G.col_off + q] = (int32_t)((uint32_t)cols[G.col_off + q] | kColDiagBit);
    return "";
}

std::vector<int64_t> Analysis::vbcrs_bookkeeping(int64_t nblocks, const int64_t *rowstart,
                                                 const int64_t *colstart) {
    // perm = sortperm(1:n; by = i -> (rowindices[i], colindices[i]))   (src/vbcrs.jl:84), stable
    std::vector<int64_t> p(nblocks);
    std::iota(p.begin(), p.end(), (int64_t)0);
    std::stable_sort(p.begin(), p.end(), [&](int64_t a, int64_t b) {
        if (rowstart[a] != rowstart[b]) return rowstart[a] < rowstart[b];
        return colstart[a] < colstart[b];
    });
    perm.resize(nblocks);
    colindices.resize(nblocks);
    rowptr.clear();
    rowindices.clear();
    // src/vbcrs.jl:103-117
    rowptr.push_back(1);
    rowindices.push_back(rowstart[p[0]]);
    for (int64_t out = 0; out < nblocks; out++) {
        const int64_t in = p[out];
        if (rowstart[in] != rowindices.back()) {
            rowptr.push_back(out + 1);
            rowindices.push_back(rowstart[in]);
        }
        perm[out] = in + 1;
        colindices[out] = colstart[in];
    }
    rowptr.push_back(nblocks + 1);
    return p;
}

}  // namespace bsm
