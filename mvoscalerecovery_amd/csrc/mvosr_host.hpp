// mvosr_host.hpp — host-side plumbing shared by the C-ABI translation units: context object,
// thread-local error message, HIP error mapping.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <unordered_map>

#include "../../include/mvosr.h"

// A block of the context's caching allocators (device memory / pinned host memory): blocks are handed back to a
// size-ordered free list instead of hipFree / hipHostFree, so that a steady-state caller allocates nothing.  `ev` is
// recorded on the compute stream when the block is released and waited for when it is handed out again: work that was
// still queued on the old contents cannot race with the next user (what hipFree's implicit synchronisation gave).
struct mvosr_block {
    void *ptr;
    size_t bytes;
    hipEvent_t ev;
    bool pending;
    bool marked;       // `ev` already marks the block's last use (mvosr_block_mark): releasing it records nothing later
    bool idle;         // marked MVOSR_MARK_IDLE: nothing queued uses the block any more
};
struct mvosr_block_cache {
    std::multimap<size_t, mvosr_block> free_blocks;
    std::unordered_map<void *, mvosr_block> live;
    size_t cached_bytes = 0;
};

struct mvosr_ctx {
    int device;
    hipStream_t own_stream;
    hipStream_t stream;        // own_stream, or an adopted external stream
    int n_cu;
    int max_lds_per_block;
    char name[128];
    // grow-only workspace between the scale kernel and the road-model kernel
    double *ws_ysel;
    size_t ws_ysel_len;
    int32_t *ws_nsel;
    size_t ws_nsel_len;
    // dense-frame variant: remapped planes {v|x,z'} x2 and y' x2 (48 B per feature), grow-only
    void *ws_dense[2];            // {x, z'} (16 B) and y' (8 B) per surviving feature of a dense batch
    size_t ws_dense_len;
    void *ws_bytes;               // generic grow-only scratch (the Delaunay kernel's per-frame arrays beyond the LDS capacity)
    size_t ws_bytes_len;
    size_t ws_bytes_limit;        // mvosr_ctx_workspace_limit: 0 = none
    // delaunay_kernel's PARTS launches (a few frames, several workgroups each): per frame an arrival counter and the parts'
    // decline flags (16 words a frame; zero between launches: the last part to arrive resets them), allocated on first use
    unsigned int *dt_parts_head;
    // optional per-call kernel timing (mvosr_ctx_profile): start / between the two kernels / end
    int prof_on;
    int prof_calls;
    hipEvent_t prof_ev[64][3];
    // host <-> device plumbing: a second stream for uploads (so that the next chunk's inputs travel under the current
    // chunk's kernels), the caching allocators, and their counters (mvosr_ctx_alloc_stats)
    hipStream_t upload_stream;
    hipEvent_t upload_ev;
    mvosr_block_cache dev_cache, host_cache;
    int64_t n_hip_malloc, n_hip_free, n_host_malloc, n_host_free, n_cache_hits;
};

constexpr int kProfRing = 64;

namespace mvosr {

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int set_hip_error(const char *what, hipError_t e);
int check_launch(const char *kernel);
int ctx_activate(mvosr_ctx *ctx);                 // hipSetDevice(ctx->device)
inline hipStream_t ctx_stream(mvosr_ctx *ctx) { return ctx->stream; }
void set_max_dynamic_lds(int bytes);
// Make the context's workspace at least (n_frames, total_feat) large; hipMalloc only when it grows.
int ctx_workspace(mvosr_ctx *ctx, int64_t n_frames, int64_t total_feat, double **ysel, int32_t **nsel);
int ctx_workspace_dense(mvosr_ctx *ctx, int64_t total_feat, void *planes[2]);   // P2 (16 B/feature), Y2 (8 B/feature)
int ctx_workspace_bytes(mvosr_ctx *ctx, size_t bytes, void **ptr);               // generic grow-only scratch

}  // namespace mvosr

// mvosr_qhull.hip: SciPy/Qhull's rows for the frames of a device-side list (list[0] = how many), written in place
int qh_rows_for_list(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt, const double *u, const double *v,
                     const int32_t *keep, int max_pts, const int64_t *tri_off, int32_t *tri, int32_t *tri_cnt, int32_t *status,
                     const int32_t *list);
