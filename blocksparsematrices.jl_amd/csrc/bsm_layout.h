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
    int32_t xbase;     // >= 0: columns are the contiguous range starting here (0-based); -1: indexed
    int32_t col_off;   // offset into cols pool of the first column
    int32_t nstrips;   // strips in this piece
    int32_t ncols;     // valid columns (<= nstrips * E)
    int32_t kind;      // KIND_*
    int32_t pad;
};
static_assert(sizeof(Piece) == 32, "Piece must be 32 bytes");

struct WaveWork {
    int32_t reserved0;
    int32_t npieces;      // 0 (nothing to stream) or 1
    int32_t row_off;      // rows pool offset (indexed row groups)
    int32_t rbase;        // >= 0: rows are the contiguous range starting here (0-based); -1: indexed
    uint16_t m;           // rows of the group (1..64)
    uint8_t work;         // WORK_*
    uint8_t grp;          // waves of this workgroup sharing the row group (1, 2 or 4)
    uint8_t lead;         // 1: this wave combines the group's partial sums and writes y
    uint8_t wg_sync;      // 1: some wave of this workgroup has grp > 1 (all 4 waves carry the same value)
    uint8_t pad0[2];
    int32_t pad1[2];
    Piece first;
};
static_assert(sizeof(WaveWork) == 64, "WaveWork must be 64 bytes");

constexpr int kWavesPerWg = 4;
constexpr int kMaxRowsPerChunk = 64;
constexpr int kScaleRowsPerWave = 1024;

inline int lanes_per_strip(int m) {
    int p = 8;
    while (p < m) p <<= 1;
    return p;
}

}  // namespace bsm
