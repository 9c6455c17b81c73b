// bsm_internal.h -- what bsm_capi.cpp (single-device handles) and bsm_dist.cpp (handles spread
// over the devices of a bsm_ctx_t) share.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsm_rocm.h"
#include "bsm_analysis.h"
#include "bsm_kernels.h"

struct bsm_ctx_s {
    std::vector<int> devices;  // HIP ordinals; the same ordinal may appear several times (virtual devices)
    // peer access was ENABLED (hipSuccess or hipErrorPeerAccessAlreadyEnabled) for every ordered pair of distinct
    // devices: the condition of the fused fan-out kernels, which dereference peer pointers directly
    bool peer_ok = false;
};

namespace bsm {
struct DistState;  // bsm_dist.cpp
}

struct bsm_matrix_s {
    bsm::Analysis an;  // multi-device handles: bookkeeping / statistics of the WHOLE operator, no image
    bsm::DeviceImage img;
    bool on_device = false;
    // optional second ordering (bsm_options.transpose_image): the transposed operator as its own
    // forward image
    bool has_t = false;
    bsm::Analysis an_t;
    bsm::DeviceImage img_t;
    std::mutex gather_mu;  // the gather workspace admits one product in flight per handle
    hipStream_t ws_stream = nullptr;  // stream of the last gather-mode product
    bool ws_pending = false;          // ... which may still be running (see WorkspaceClaim)
    // work arrays of the interleaved multi-RHS pass (bsm_kernels.h: ILWork): allocated at the first product that takes
    // it, one product in flight (ILClaim in bsm_capi.cpp: same rules as the gather workspace)
    std::mutex il_mu;
    bsm::ILWork il;
    hipStream_t il_stream = nullptr;
    bool il_pending = false;
    // device staging buffers of the BSM_MEM_HOST path, kept between calls (grow-only); a second
    // concurrent host call on the same handle falls back to temporary buffers
    std::mutex host_mu;
    void *stage_x = nullptr, *stage_y = nullptr;
    size_t stage_x_bytes = 0, stage_y_bytes = 0;
    // handle spread over the devices of a context (bsm_options.ctx)
    std::unique_ptr<bsm::DistState> dist;
    bsm_matrix_s();
    ~bsm_matrix_s();
};

namespace bsm {

int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
int build_error(const std::string &err);  // analysis error -> BSM_ERR_INVALID, value sink error -> BSM_ERR_DEVICE

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    hipError_t enter(int dev);
    ~DeviceGuard();
};

AnalysisOptions to_aopt(const bsm_options &o, ValueSink *sink);
void fill_image(const Analysis &an, const bsm_options &o, bool use_own, DeviceImage &img);
hipError_t upload_image(Analysis &an, DeviceImage &img, int dev);
void free_image(DeviceImage &img);
hipError_t device_pack(Analysis &an, void **d_values);  // blocks_on_device: run the pack plan on the current device
// the value sink of a device handle (pinned staging windows + asynchronous upload); nullptr for none
std::unique_ptr<ValueSink> make_device_sink(void **d_values);

// ---- bsm_dist.cpp -------------------------------------------------------------------------------
// Row partition of `in` (already in its final order) over the context's devices; fills A->dist.
// A->an must already hold the whole operator's bookkeeping (meta-only analysis).
int dist_create(bsm_matrix_s *A, bsm_ctx_s *ctx, int mtype, int dtype, int64_t nrows, int64_t ncols,
                const std::vector<BlockIn> &in, const bsm_options &o);
int dist_mul(bsm_matrix_s *A, int op, const void *x, void *y, const void *alpha, const void *beta,
             int beta_strong_zero, int memspace, hipStream_t stream);
int dist_mul_multi(bsm_matrix_s *A, int op, long long nrhs, const void *X, long long ldx, void *Y, long long ldy,
                   const void *alpha, const void *beta, int beta_strong_zero, int memspace, hipStream_t stream);
int dist_mul_parts(bsm_matrix_s *A, int op, const void *const *x_parts, void *const *y_parts, const void *alpha,
                   const void *beta, int beta_strong_zero, void *const *streams);
void dist_destroy(bsm_matrix_s *A);
int dist_part_info(bsm_matrix_s *A, int32_t part, bsm_part_info_t *out);
int64_t dist_device_bytes(const bsm_matrix_s *A);
// (analysis, image) of every part that holds blocks
void dist_images(bsm_matrix_s *A, std::vector<std::pair<const Analysis *, const DeviceImage *>> &out);

// smallest row index of every block (its partition key) and its weight (stored entries)
void block_row_keys(const std::vector<BlockIn> &in, std::vector<int64_t> &key, std::vector<int64_t> &weight);
// see bsm_partition_rows in include/bsm_rocm.h
void partition_rows(int64_t nrows, const std::vector<int64_t> &key, const std::vector<int64_t> &weight,
                    int nparts, std::vector<int32_t> &part_of_block, std::vector<int64_t> &own_lo,
                    std::vector<int64_t> &own_hi);

}  // namespace bsm
