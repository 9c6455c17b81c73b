// bsm_analysis.h -- host-side analysis stage: everything the reference does at
// construction time (VBCRS sort + rowptr, colouring) plus what the GPU path needs
// (row grouping, strip packing, wave/workgroup schedule).  Pure C++, no HIP calls.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bsm_layout.h"

namespace bsm {

// byte buffer whose storage is NOT zero-filled on allocation (the packer writes every byte it
// needs and zeroes the strip tails itself; a 16 GB memset per create would be pure waste)
class RawBuffer {
  public:
    RawBuffer() = default;
    RawBuffer(const RawBuffer &) = delete;
    RawBuffer &operator=(const RawBuffer &) = delete;
    ~RawBuffer() { release(); }
    void allocate(size_t n) {
        release();
        p_ = n ? static_cast<char *>(::operator new(n)) : nullptr;
        n_ = n;
    }
    void release() {
        if (p_) ::operator delete(p_);
        p_ = nullptr;
        n_ = 0;
    }
    char *data() { return p_; }
    const char *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    const char &operator[](size_t i) const { return p_[i]; }

  private:
    char *p_ = nullptr;
    size_t n_ = 0;
};

enum MatType { MT_VBCRS = 0, MT_BLOCKSPARSE = 1, MT_SYMMETRIC = 2 };

struct BlockIn {
    const char *data;     // column-major m x n, leading dimension ld (elements)
    int64_t m, n, ld;
    const int64_t *ridx;  // 1-based row index list (m entries) or nullptr -> contiguous at r0
    const int64_t *cidx;  // 1-based col index list (n entries) or nullptr -> contiguous at c0
    int64_t r0, c0;       // 1-based first row / column when the lists are null
    int kind;             // KIND_*
    bool trans = false;   // the logical m x n block is the TRANSPOSE of the stored n x m array
};

struct Tunables {
    // Bytes one wave streams (W).  A wave's life is a chain of memory round trips -- descriptor, x
    // slice, one per 8 KB iteration, emission (tools/wavetrace.py: 9 us for the 8 KB waves of a
    // BEM-shaped fused launch, of which 1.7 us stream the matrix) -- and a workgroup keeps its slot
    // until its slowest wave is done.  W = 8 KB, one iteration per wave, puts every request of a
    // launch in flight at once: right for short launches and for row groups that fill their lanes
    // (64-row blocks stream at 6.5-6.9 TB/s that way; fatter waves cost them 2-3 %).  Row groups
    // that do NOT fill their lanes (3-28-row BEM panels, mixed block sizes) move fewer bytes per
    // iteration and pay the fixed part of the chain more often: a LONG launch of them is bound by
    // wave slots x chain, and fewer, fatter waves amortise it (BEM fixture tiled to 0.7 GB: fused
    // 4.4 -> 4.9 TB/s, forward-only 5.7 -> 6.2 TB/s).  So: W = 8 KB, unless the byte-weighted lane
    // fill is below fat_fill_below and the operator is long enough (operator bytes / target_waves >=
    // 10 KB, i.e. >= 0.33 GB), then W = min(operator bytes / target_waves, 24 KB), a row group gets its
    // second wave only from 2 W on and never four (target_waves = 4 rounds of the ~8 K resident
    // waves; measured on 0.3-2.2 GB operators: waves of 20-45 KB are the optimum, 64 KB lose again).
    // Otherwise row groups of >= W get 2 waves, >= 3 W get 4.  Non-exclusive groups are cut into
    // workgroup items of 4 W.  BSM_WAVE_BYTES fixes W.
    int64_t wave_bytes = 0;              // 0: automatic
    int64_t target_waves = 32768;        // BSM_TARGET_WAVES
    double fat_fill_below = 0.9;         // BSM_FAT_FILL_BELOW
    int64_t wave_bytes_min = 8 << 10, wave_bytes_max = 24 << 10;
    // Multi-RHS products (bsm_mul_multi) walk a SECOND, coarser split of the same panels (Analysis::waves_multi):
    // with K right-hand sides a wave's fixed part -- K x rows kept in registers, K-wide x slices, K-wide output
    // and combine -- is K times that of the single product, and at 8 KB per wave it outweighed the matrix bytes
    // (C3 x 8: 216 of 383 us were left with the loads, the arithmetic and the atomics all switched off; waves
    // of 64 KB: 383 -> 266 us).  BSM_MULTI_WAVE_BYTES, 0 = one split for both.
    int64_t multi_wave_bytes = 64 << 10;
    int64_t split2_bytes = 0;     // row groups at least this big get 2 waves (0: W)
    int64_t split4_bytes = 0;     // ... and 4 waves (0: 3 W)
    int64_t wgitem_max_bytes = 0;   // non-exclusive groups are cut into items this big (0: 4 W)
    int chunk_rows = kMaxRowsPerChunk;  // blocks taller than this are cut into chunks
    int64_t deep_group_bytes = 128 << 10;  // auto mode: row groups above this are "deep" (BSM_DEEP_GROUP_BYTES)
    int64_t deep_total_bytes = 64 << 20;  // ... and only operators at least this big are split (BSM_DEEP_TOTAL_BYTES)
    int pack_threads = 8;
    size_t window_bytes = 64u << 20;  // staging window of a streamed upload (BSM_UPLOAD_WINDOW_BYTES)
    int lds_window = 1;  // LDS y window for locality-packed small symmetric row groups
    int wg_order = -1;  // workgroup dispatch order (BSM_ORDER): -1 auto (snake for exclusive images), 0 plain largest first
    static Tunables from_env();
};

// Receives the packed value stream window by window in ascending offset order, instead of one host
// buffer holding the whole operator (device handles: pinned staging + asynchronous upload, see
// bsm_capi.cpp).  Every method returns "" on success, else an error message.
struct ValueSink {
    virtual ~ValueSink() = default;
    // total size of the stream; *use = false declines (the analysis then packs into Analysis::values)
    virtual std::string begin(size_t total_bytes, bool *use) = 0;
    virtual char *window(size_t bytes) = 0;  // writable staging for the next window, nullptr on failure
    virtual std::string commit(size_t offset, size_t bytes) = 0;  // the window is complete: ship it
    virtual std::string end() = 0;                                // every window has arrived
};

struct AnalysisOptions {
    int scheduler = 0;   // 0 serial, 1 dynamic (colour like the reference)
    int validate = 1;
    int accumulate = 0;  // 0 auto, 1 atomic, 2 coloured launches, 3 gather (both bitwise reproducible), 4 direct
    int64_t own_lo = 0, own_hi = 0;  // 1-based inclusive, 0,0 = all rows
    ValueSink *sink = nullptr;       // not owned; nullptr: pack into Analysis::values
    // the block arrays are DEVICE memory (bsm_options.blocks_memspace): the analysis never reads them;
    // instead of packed values it leaves a pack plan (Analysis::pack_plan / pack_colpos) that a kernel
    // executes on the device
    bool blocks_on_device = false;
    int coloring = 0;  // reference colourings: 0 WorkstreamDSATUR (the reference's default), 1 plain DSATUR
    bool meta_only = false;    // validation, statistics and the reference colourings only (no image)
    bool skip_colors = false;  // leave `colors` empty (the parts of a multi-device handle)
};

// Deterministic DSATUR colouring of blocks by index-list conflicts (two blocks conflict
// iff their lists share an index -- reference src/coloring.jl:45-61).  Specification
// identical to oracle/bsm_oracle.c:orc_color_dsatur.  Returns classes of 1-based ids.
std::vector<std::vector<int64_t>> color_dsatur(const std::vector<const int64_t *> &lists,
                                               const std::vector<int64_t> &lens);
// WorkstreamDSATUR (the reference's default, src/BlockSparseMatrices.jl:10): zones / DSATUR per zone /
// gather, as published by Turcksin, Kronbichler & Bangerth (ACM TOMS 2016).  Specification identical
// to oracle/bsm_oracle.c:orc_color_workstream.  Classes of 1-based ids.
std::vector<std::vector<int64_t>> color_workstream_dsatur(const std::vector<const int64_t *> &lists,
                                                          const std::vector<int64_t> &lens);

// One chunk (<= 64 rows of one input block) of the device-side repacking: element (i, w) of the
// chunk goes to merged panel column q = perm_off < 0 ? woff + w : colpos[perm_off + w] of the row
// group whose panel starts at 16-byte unit dst_unit:  unit (q / E) * mc + i, slot q % E.
struct PackChunk {
    uint64_t src;       // device address of the block (column-major, leading dimension ld)
    uint64_t dst_unit;  // first 16-byte unit of the row group's panel in the value stream
    int64_t ld;
    int32_t ra, mc;     // first row of the chunk inside the block, rows
    int32_t n, woff;    // columns of the block, first merged panel column (identity placement)
    int32_t perm_off;   // >= 0: offset into pack_colpos (scattered placement)
    int32_t trans;      // 1: the logical block is the transpose of the stored array
};
static_assert(sizeof(PackChunk) == 48, "PackChunk must be 48 bytes");

class Analysis {
  public:
    int mtype = 0, dtype = 0;
    int es = 8;  // element size in bytes
    int E = 2;   // columns per strip
    int64_t nrows = 0, ncols = 0;
    AnalysisOptions opt;
    Tunables tun;

    // ---- reference bookkeeping (1-based) ----
    std::vector<int64_t> perm, rowptr, colindices, rowindices;  // VBCRS, src/vbcrs.jl:84-117
    std::vector<std::vector<int64_t>> colors[3];  // 0 colors/offdiag, 1 transpose, 2 diagonal
    int64_t nnz = 0, stored_entries = 0, alg_bytes = 0;

    // ---- device image (host copy) ----
    RawBuffer values;          // empty when the stream went to AnalysisOptions::sink
    std::vector<PackChunk> pack_plan;   // blocks_on_device: what the pack kernel executes
    std::vector<int32_t> pack_colpos;   // ... and the scattered column placements it refers to
    int64_t value_bytes = 0;   // size of the packed value stream
    std::vector<int32_t> rows, cols;
    std::vector<WaveWork> waves;
    std::vector<WaveWork> waves_multi;  // the coarser split for multi-RHS products (empty: `waves` serves both)
    int64_t nwg_multi = 0;
    double lane_fill = 1.0;  // byte-weighted share of the lanes the row groups' strips fill (stage_work_items)
    double mean_rows = 64.0;  // byte-weighted mean height of the row groups
    int64_t nwg_main = 0;   // workgroups holding panel work
    int64_t nwg_total = 0;  // + workgroups of scale work (exclusive forward launch only)
    int64_t ngroups = 0;
    bool exclusive_fwd = false;  // every y row is produced by at most one row group
    // coloured mode: workgroups [color_wg_ptr[c], color_wg_ptr[c+1]) form launch c; the row groups
    // of one launch touch pairwise disjoint y entries (rows and columns), for every op
    std::vector<int64_t> color_wg_ptr;
    // gather mode: workspace slots.  Transposed column sum of merged panel column q of a group:
    // slot col_off + q (its position in the cols pool); forward partial sums of a workgroup item:
    // slots ws_fbase + WaveWork::win_base + row.  inv_ptr/inv_idx[k]: CSR over the y entries of
    // op N (k = 0) and op T / C (k = 1) listing the slots that contribute, in ascending order.
    // LDS y window (symmetric operators): y contributions of op N in all (forward rows + transposed
    // columns), those that pass through a workgroup's window, and what the windows flush to y
    int64_t win_emissions = 0, win_inside = 0, win_flushed = 0;
    bool gather = false;
    int64_t ws_fbase = 0, ws_slots = 0;
    std::vector<int64_t> inv_ptr[2];
    std::vector<int32_t> inv_idx[2];

    // Builds everything.  Returns "" on success, else an error message.
    std::string build(int mtype, int dtype, int64_t nrows, int64_t ncols,
                      const std::vector<BlockIn> &blocks, const AnalysisOptions &opt);

    // Fills perm / rowptr / colindices / rowindices exactly as src/vbcrs.jl:84-117 and returns the
    // sorted order (0-based input positions).
    std::vector<int64_t> vbcrs_bookkeeping(int64_t nblocks, const int64_t *rowstart, const int64_t *colstart);

  private:
    // the stages of build(), in the order they run (bsm_analysis.cpp)
    struct BuildState;
    std::string stage_validate(const std::vector<BlockIn> &blocks, BuildState &st);
    void reference_colourings(const std::vector<BlockIn> &blocks, bool &colour_oom);
    std::string stage_row_groups(const std::vector<BlockIn> &blocks, BuildState &st);
    std::string stage_merge_columns(const std::vector<BlockIn> &blocks, BuildState &st);
    std::string stage_accumulation(const std::vector<BlockIn> &blocks, BuildState &st);
    void stage_work_items(BuildState &st);
    void stage_place_values(BuildState &st);
    std::string stage_pack_values(const std::vector<BlockIn> &blocks, BuildState &st);
    void stage_waves(BuildState &st);
    void stage_windows(BuildState &st);
    std::string stage_gather_index(BuildState &st);
};

}  // namespace bsm
