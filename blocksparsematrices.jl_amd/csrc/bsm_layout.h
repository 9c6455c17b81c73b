// bsm_layout.h -- device image layout shared by the host analysis and the HIP kernels.
//
// HBM layout (one allocation each, all owned by the handle):
//   values : every stored matrix entry exactly once, in STRIP order.  All blocks (of one
//            kind) that live on the same <= 64 y rows form a ROW GROUP; they are
//            concatenated column-wise into one merged mc x W panel (a VBCRS block row is one
//            such panel).  Its entries are stored as [strip s][row i][e], E = 16 / sizeof(T)
//            columns per strip, so that one lane load is 16 bytes = E consecutive columns of
//            one row, and one wave-instruction reads G = 64/P consecutive strips =
//            G*mc*16 contiguous bytes (P = lanes per strip = max(8, nextpow2(mc))).  The
//            last strip of a panel is zero-padded to E columns.
//   rows   : int32 0-based y/x index lists of row groups whose rows are not a contiguous
//            range.
//   cols   : int32 0-based merged column list of every row group (x index per panel column).
//   waves  : WaveWork descriptors (64 B), 4 per workgroup.  A wave streams ONE piece = a
//            strip range of one merged panel, described inline, so it reaches its matrix
//            bytes after ONE dependent (scalar) load.
#pragma once
#include <cstdint>

namespace bsm {

enum : uint8_t {
    KIND_PLAIN = 0,  // BlockSparseMatrix / VBCRS block: op N -> forward, op T/C -> transposed
    KIND_DIAG = 1,   // SymmetricBlockMatrix diagonal block: like PLAIN
    KIND_OFF = 2,    // SymmetricBlockMatrix off-diagonal block: forward AND transposed
};

enum : uint8_t {
    WORK_NOP = 0,    // padding wave of a workgroup
    WORK_PANEL = 1,  // stream pieces of one row group
    WORK_SCALE = 2,  // y[rbase .. rbase+count) *= beta (rows no block covers); count in piece0.ncols
};

struct Piece {
    uint64_t val_off;  // offset into values, in 16-byte units
    int32_t xbase;     // first x index of column segment 0 (0-based); -1: use the cols pool
    int32_t col_off;   // offset into cols pool of the first column
    int32_t nstrips;   // strips in this piece
    int32_t ncols;     // valid columns (<= nstrips * E)
    int32_t kind;      // kinds word: bits 0-1 kind of column segment 0 (of every non-flagged column for
                       // pieces that use the cols pool), 2-3 segment 1, 4-5 segment 2; bit 8: the
                       // piece has KIND_OFF columns; bit 9: its ROW GROUP has (other waves of the
                       // workgroup item may hold them: decides whether forward sums exist in op T).  In the cols pool a set sign bit marks a
                       // KIND_DIAG column of a panel that also holds KIND_OFF columns.
    int32_t seg2_x;    // first x index of column segment 2
};
static_assert(sizeof(Piece) == 32, "Piece must be 32 bytes");

// The columns of a piece are usually a few contiguous runs of x (one per block of a VBCRS block
// row).  Up to three runs are described inline -- segment k covers piece columns
// [seg_k_w, seg_{k+1}_w) and maps column w to x index seg_k_x + (w - seg_k_w), seg_0_w = 0 --
// so the wave computes its x addresses without a dependent load of the cols pool.
// Workgroups whose waves work on DIFFERENT small row groups of a symmetric operator are packed by
// locality; their y contributions (forward rows and transposed columns) largely coincide, so they
// are accumulated in an LDS window [win_base, win_base + 8*win_span8) and leave the CU once.
// 4 KB of LDS per workgroup whatever the element type (512 fp64 entries; win_span8 is a byte, <= 2040):
// with the 8 KB x slices and 8 KB emission staging a fused workgroup stays at 20 KB = 8 per CU
constexpr int window_entries(int elem_bytes) { return 4096 / elem_bytes; }

struct WaveWork {
    int32_t seg1_w;       // first piece column of segment 1 (>= ncols when unused)
    int32_t win_base;     // first y index of the workgroup's LDS accumulation window
    int32_t row_off;      // rows pool offset (indexed row groups)
    int32_t rbase;        // >= 0: rows are the contiguous range starting here (0-based); -1: indexed
    uint16_t m;           // rows of the group (1..64)
    uint8_t work;         // WORK_*
    uint8_t grp;          // waves of this workgroup sharing the row group (1, 2 or 4)
    uint8_t lead;         // 1: this wave combines the group's partial sums and writes y
    uint8_t wg_sync;      // 1: some wave of this workgroup has grp > 1 (all 4 waves carry the same value)
    uint8_t npieces;      // 0 (nothing to stream) or 1
    uint8_t win_span8;    // window length / 8 (0: no window); same value in all 4 waves
    int32_t seg1_x;       // first x index of segment 1
    int32_t seg2_w;       // first piece column of segment 2 (>= ncols when unused)
    Piece first;
};
static_assert(sizeof(WaveWork) == 64, "WaveWork must be 64 bytes");

constexpr int kKindHasOff = 1 << 8;
constexpr int kKindGroupHasOff = 1 << 9;
constexpr uint32_t kColDiagBit = 0x80000000u;

constexpr int kWavesPerWg = 4;
constexpr int kMaxRowsPerChunk = 64;
constexpr int kScaleRowsPerWave = 1024;

inline int lanes_per_strip(int m) {
    int p = 8;
    while (p < m) p <<= 1;
    return p;
}

}  // namespace bsm
