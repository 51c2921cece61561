/*
 * mvosr.h — C ABI of libmvosr.so: MI355X (gfx950) kernels for the per-frame scale-recovery hot
 * path of TimingSpace/MVOScaleRecovery.
 *
 * The reference is pure Python and has no FFI/plugin registry: the drop-in boundary is the
 * Python class `ScaleEstimator` chosen by an import line (/root/reference/src/main.py:18-20,
 * /root/reference/src/main_offline.py:18-20).  mvoscalerecovery_amd/scale_calculator.py keeps
 * that class surface and binds the entry points below with ctypes (see INTEGRATION.md for the
 * stub).  Each entry point cites the reference routine whose per-frame work it replaces; all of
 * them process a BATCH of independent frames, one workgroup (or one wavefront for small
 * frames) per frame.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer inside mvosr_batch / mvosr_outputs is a DEVICE
 *     pointer (hipMalloc'ed by mvosr_malloc, or e.g. torch.Tensor.data_ptr()); the structs
 *     themselves live in host memory and are read during the call;
 *   - every function returns 0 on success, a negative mvosr_err on failure, never throws;
 *     mvosr_last_error() returns a thread-local message;
 *   - launches are asynchronous on the context's HIP stream; call mvosr_ctx_sync() (or
 *     synchronise the adopted stream yourself) before reading outputs;
 *   - all floating-point data is IEEE binary64 (the reference computes in NumPy float64 and
 *     its result is a quantised histogram mode, so threshold decisions must match bit-for-bit:
 *     SURVEY.md fact 5); triangle vertex ids are int32 exactly as scipy.spatial.Delaunay
 *     emits `simplices` (row-major (T,3); the order of vertices inside a row matters,
 *     /root/reference/src/scale_calculator.py:113-115).
 */
#ifndef MVOSR_H
#define MVOSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVOSR_ABI_VERSION 12

/* error codes (function return values) */
enum mvosr_err {
    MVOSR_OK = 0,
    MVOSR_ERR_HIP = -1,          /* a HIP runtime call failed (message has hipGetErrorString) */
    MVOSR_ERR_ARG = -2,          /* bad argument (null pointer, negative size, ...) */
    MVOSR_ERR_TOO_LARGE = -3,    /* a frame does not fit the requested kernel variant */
    MVOSR_ERR_NO_DEVICE = -4,    /* no gfx950 device visible */
    MVOSR_ERR_ALLOC = -5         /* device memory for a grow-only workspace could not be allocated (the triangulation kernels'
                                    per-frame arrays): nothing was launched; the caller may retry with fewer frames or take its host path */
};

/* per-frame status codes written to mvosr_outputs.status.  0-3 are the reference's normal
 * return branches, 4 its "no enough flat feature" fallback, 5-7 the places where the reference
 * raises; the Python shim maps them back to return-or-raise. */
enum mvosr_status {
    MVOSR_ST_MODE = 0,          /* height = mode/10               scale_calculator.py:354     */
    MVOSR_ST_RIGHT = 1,         /* skew > 0.3: right minimum edge scale_calculator.py:348-352 */
    MVOSR_ST_MEDIAN = 2,        /* no modes: median(y)            scale_calculator.py:331-333 */
    MVOSR_ST_LEVEL = 3,         /* no modes, no points: height_level          :334-335        */
    MVOSR_ST_NO_FLAT = 4,       /* selection empty: scale = ref/height_level  :277-279,:420-422 */
    MVOSR_ST_ERR_LEFT = 5,      /* IndexError at scale_calculator.py:343 */
    MVOSR_ST_ERR_RIGHT = 6,     /* IndexError at scale_calculator.py:344 */
    MVOSR_ST_ERR_SINGULAR = 7,  /* LinAlgError at scale_calculator.py:229 */
    MVOSR_ST_ERR_MASK = 8,      /* tri2 inconsistent with the vote computed on the GPU, or a
                                   vertex id out of range (build-side check, no reference analogue) */
    MVOSR_ST_ERR_EMPTY = 9,     /* frame without triangles, or with fewer than 3 features below the vanishing row (where the
                                   reference's first Delaunay call raises QhullError, :257: the caller must raise) */
    MVOSR_ST_TOO_FEW = 10,      /* exactly 3 features below the vanishing row: the reference skips the second
                                   triangulation (:263-270) and divides by the PREVIOUS frame's height_level
                                   (:420-422, std 100); raw_scale/height_level are NaN here and the host's
                                   cross-frame step supplies that level */
    MVOSR_ST_RS_FEW = 11        /* rescale variant: fewer than 12 selected points — no RANSAC, the previous scale is pushed
                                   again (/root/reference/src/rescale.py:152,175); raw_scale is NaN */
};

/* number of int32 per frame in mvosr_outputs.counts */
#define MVOSR_N_COUNTS 8
enum mvosr_count_slot {
    MVOSR_CNT_VALID = 0,        /* features with vote counter >= 0            (:164-166) */
    MVOSR_CNT_TRI_PITCH = 1,    /* triangles with pitch_deg < -80             (:235)     */
    MVOSR_CNT_TRI_VALID = 2,    /* ... and mean height > height_level         (:243-244); -1 where the tiled
                                   dense variant ran (it keeps per-vertex maxima, not per-triangle flags) */
    MVOSR_CNT_SELECTED = 3,     /* unique vertices of those triangles         (:247)     */
    MVOSR_CNT_KEPT = 4,         /* selected points left after remove_single   (:284-293) */
    MVOSR_CNT_MODES = 5,        /* number of mode clusters                    (:468-481) */
    MVOSR_CNT_MODE_LEFT = 6,    /* mode_left  (:339), -1 if none */
    MVOSR_CNT_MODE_RIGHT = 7    /* mode_right (:338), -1 if none */
};

#define MVOSR_HIST_BINS 169     /* np.histogram(y, bins=arange(170)*0.1)      (:326) */

typedef struct mvosr_ctx mvosr_ctx;

/* Scalars of the path.  cos/sin are passed (not the angle) so that the kernels use the very
 * doubles NumPy produced on the host (scale_calculator.py:391-392). */
typedef struct mvosr_params {
    double cos_pitch;            /* np.cos(camera_pitch), camera_pitch = -0.5*pi/180 (:24) */
    double sin_pitch;            /* np.sin(camera_pitch) */
    double absolute_reference;   /* real camera height, param.camera_h (main.py:55) */
    double pitch_threshold_deg;  /* -80 (:235,:239) */
    double skew_threshold;       /* 0.3 (:348) */
    double mode_rel;             /* 0.33 (:461) */
    int32_t mode_min;            /* 2 (:451,:462) */
    int32_t vote_mode;           /* MVOSR_VOTE_REFERENCE (0): check_triangle's flag pattern exactly as the reference has it —
                                    `b > 0` marks vertices 0 and 1 (:113-115), so the vote depends on the order of the
                                    vertices inside a row and the rows must be SciPy's, verbatim;
                                    MVOSR_VOTE_FIXED (1): `b > 0` marks vertices 0 and 2 (the evident intent of those lines) —
                                    a DECLARED DEVIATION from the reference: the vote is then invariant
                                    under the order of a row's vertices, so any row form of the same triangle set gives
                                    the same counters (what makes the device triangulation's rows usable) */
} mvosr_params;
#define MVOSR_VOTE_REFERENCE 0
#define MVOSR_VOTE_FIXED 1

/* A packed batch of F frames, resident in HBM.  Features are those that passed the
 * vanishing-row filter (:252-254), stored as planes (structure of arrays) with the frames'
 * segments back to back; segment starts are even (16-byte aligned doubles).  x/y/z are the
 * caller's raw feature3d columns — feature_remap (:390-394) is applied by the kernels at load. */
#define MVOSR_TRI2_SURVIVORS 0
#define MVOSR_TRI2_FEATURES 1

typedef struct mvosr_batch {
    int64_t n_frames;
    const int64_t *feat_off;     /* [F]   start of frame f in x/y/z/v (multiple of 2)          */
    const int32_t *feat_cnt;     /* [F]   number of features of frame f                        */
    const double *x, *y, *z;     /* feature3d[:,0..2] before the remap                         */
    const double *v;             /* feature2d[:,1]  (pixel row; u is never read by the path)   */
    const int64_t *tri1_off;     /* [F+1] triangle offsets of the first triangulation          */
    const int32_t *tri1;         /* [tri1_off[F]*3] Delaunay(feature2d).simplices  (:257-258)  */
    const int64_t *tri2_off;     /* [F+1] triangle offsets of the second triangulation         */
    const int32_t *tri2;         /* [tri2_off[F]*3] Delaunay(feature2d[valid]).simplices (:266-267);
                                    ids index the features that survive the vote, in order     */
    const int32_t *n2_expected;  /* [F] or NULL: number of points tri2 was built on; a frame whose
                                    vote keeps a different number gets MVOSR_ST_ERR_MASK        */
    int32_t max_feat;            /* max(feat_cnt) — sizes the LDS request; a frame with more
                                    features than this gets MVOSR_ST_ERR_MASK                  */
    int32_t tri2_ids;            /* MVOSR_TRI2_SURVIVORS (0): tri2 as SciPy returns it, ids index the survivors;
                                    MVOSR_TRI2_FEATURES (1): the same rows relabelled to index the frame's features
                                    (every vertex a survivor, else MVOSR_ST_ERR_MASK) — the frames then run the
                                    gather variant without compacting, whatever their size (meant for dense frames:
                                    a quarter less HBM traffic; results identical)                              */
    int64_t total_feat;          /* length (elements) of the x/y/z/v planes: feat_off[F-1] + padded
                                    count of the last frame; sizes the context's workspace      */
    /* Optional tile index of dense frames (all NULL / 0: none).  With feature-numbered rows (tri2_ids ==
     * MVOSR_TRI2_FEATURES), features sorted so that a triangle's vertices lie close together in memory, and the rows
     * of both triangulations sorted by smallest vertex, the gather variant reads every input byte once: it walks the
     * frame in tiles of MVOSR_TILE_W features with two tiles resident in LDS.  Per frame f the index holds
     * ntiles(f) + 1 = ceil(feat_cnt[f] / MVOSR_TILE_W) + 1 entries starting at tile_base[f]: entry k (k < ntiles) =
     * index, within the frame's rows, of the first row that is walked with tile k — its vertices all lie in tiles k
     * and k+1; entry ntiles = where the "far" rows start (rows whose vertices are further apart: they come last, a
     * few per thousand in a Delaunay triangulation laid out this way, and are gathered from global memory).  The
     * kernel checks the index (starts at 0, monotone, in range) and every walked row against its tile window; an
     * inconsistent index gives MVOSR_ST_ERR_MASK, never a wrong result. */
    int32_t tile_w;              /* MVOSR_TILE_W, or 0 */
    int32_t min_feat;            /* min(feat_cnt), or 0 = not stated.  With waves_per_frame == 0 a batch whose smallest
                                    and largest frames fall into different variants' ranges (below) is launched per
                                    size class — each class with the variant and the LDS request of its own largest
                                    frame; stating min_feat lets a uniform batch skip the classification launch    */
    const int64_t *tile_base;    /* [F+1] */
    const int32_t *tile1_off;    /* [tile_base[F]] index into the frame's tri1 rows */
    const int32_t *tile2_off;    /* [tile_base[F]] index into the frame's tri2 rows */
    int32_t size_hint[4];        /* opaque, all zero = none.  mvosr_batch_size_hint() fills it from the host's copy of
                                    feat_cnt (how many frames each size class holds), so that the per-class launches
                                    of a ragged batch are exactly as long as their lists; without it every class is
                                    launched over n_launch workgroups, most of which leave at once                  */
    /* Optional companion of the tile index (NULL: the kernel gathers instead): the far rows' vertices, copied out of
     * the planes by the packer so that the kernel reads them as one contiguous block per frame instead of a 128-byte
     * line per 8-byte element.  Per frame f, tile_far_off[f+1] - tile_far_off[f] = 9 * (far rows of tri1 + far rows of
     * tri2) doubles starting at tile_far[tile_far_off[f]]: first every far row of tri1 in row order as (y, z, v) of
     * its three vertices, then every far row of tri2 as (x, y, z) of its three vertices — the values of the planes,
     * verbatim (raw, before the remap).  A block of the wrong length gives MVOSR_ST_ERR_MASK; the VALUES cannot be
     * checked by the kernel: they are the caller's copy of its own planes. */
    const double *tile_far;
    const int64_t *tile_far_off; /* [F+1] */
    /* Optional explicit row counts (NULL: frame f has tri*_off[f+1] - tri*_off[f] rows).  With counts, frame f's rows are
     * the first tri*_cnt[f] rows at tri*_off[f] and the offsets only say where a frame's rows start — the form
     * mvosr_delaunay_batch writes (a frame's capacity is 2 * points rows; how many it holds is known on the device
     * only), so that a device-built triangulation goes into the scale kernel without a trip through the host. */
    const int32_t *tri1_cnt;     /* [F] or NULL */
    const int32_t *tri2_cnt;     /* [F] or NULL */
    /* Optional, for layouts that PERMUTE the rows of tri2 (the dense tile layout sorts them by smallest vertex): laid out
     * like tri2's rows, tri2_order[tri2_off[f] + k] = index, within frame f's rows as stored, of the k-th row of the
     * caller's original order (SciPy's).  height_level is np.mean over the steep triangles' heights IN ROW ORDER
     * (/root/reference/src/scale_calculator.py:239-240): with the table the exact pass sums in the original order and
     * the level is the reference's double to the last bit; without it (NULL) in the stored order (equal to rounding). */
    const int32_t *tri2_order;
    /* Optional [F] (NULL: none): frames whose byte is non-zero are finished in the EXACT mode — height_level summed in
     * NumPy's own order — by the product launch itself (they join the few frames its exact pass redoes anyway).  For the
     * frames whose level a LATER step reads: the frame before one that takes the reference's "no enough feature for
     * triangulation" branch and divides by the previous level (/root/reference/src/scale_calculator.py:263-270,:420-422),
     * the last frame of a chunk, the frame the estimator's height_level is left at when the next one raises. */
    const uint8_t *exact_mask;
    /* Optional (all NULL: none) — tri2 is a STAND-IN: the triangle set of the second triangulation with other rows than SciPy's
     * (mvosr_delaunay_batch's canonical form: ~1 us per frame instead of the ~8 us of mvosr_delaunay_qhull_batch).  The rows of
     * tri2 reach the result only through rounding — the LU solve sees the vertices in row order (:229), height_level is summed in
     * row order (:239-240) —, and every frame in which rounding can decide is on the exact pass's list already (a flat triangle
     * within the guard band of the level, a pitch inside the band where the reference's own formulation is evaluated, a level
     * that is itself the result, the exact mask): for THOSE frames mvosr_scale_batch builds SciPy's own rows in place
     * (mvosr_delaunay_qhull_batch's kernel over the list) before the exact pass reads them.  HOT mode only (no stage outputs),
     * survivor-numbered rows, frames that fit the LDS-resident kernels.  standin_u: pixel columns laid out like v; standin_keep:
     * the vote the second triangulation was built on (>= 0: kept), laid out like v; standin_rows / standin_cnt: tri2 / tri2_cnt
     * again, writable; standin_status [F]: receives MVOSR_DT_DEGENERATE | reason << 8 for a listed frame whose rows Qhull's
     * replay declines (the caller then takes that frame through the host's SciPy). */
    const double *standin_u;
    const int32_t *standin_keep;
    int32_t *standin_rows;
    int32_t *standin_cnt;
    int32_t *standin_status;
} mvosr_batch;

#define MVOSR_TILE_W 512

/* Outputs (device pointers; any optional pointer may be NULL). */
typedef struct mvosr_outputs {
    double *raw_scale;           /* [F] absolute_reference/height, before the window median (:419,:421) */
    double *height;              /* [F] camera height over the road in VO units (:418)          */
    double *height_level;        /* [F] mean height of the non-flat triangles (:239-241)        */
    int32_t *status;             /* [F] enum mvosr_status                                      */
    int32_t *counts;             /* [F*MVOSR_N_COUNTS] or NULL                                 */
    /* stage-level outputs for parity tests / the per-frame drop-in call; all optional */
    int32_t *vote_counters;      /* [sum feat_cnt, laid out like x] per-feature counters (:153-163) */
    uint8_t *selected;           /* [like x] 1 where the k-th SURVIVING feature of the frame is selected (:247) */
    double *tri_normals;         /* [tri2_off[F]*3] n = A^-1 . 1  (:229-230)                   */
    double *tri_pitch_deg;       /* [tri2_off[F]]   asin(-n_y/|n|)*180/pi (:233)               */
    double *tri_heights;         /* [tri2_off[F]]   mean y of the 3 vertices (:238)            */
    int32_t *hist;               /* [F*2*MVOSR_HIST_BINS] histogram before / after zeroing single bins (:326,:328) */
    double *stats;               /* [F*4] mean, std, skew, median-or-nan of the kept points (:346,:496) */
} mvosr_outputs;

/* ---- library / context ------------------------------------------------------------------ */
int mvosr_abi_version(void);
const char *mvosr_last_error(void);
int mvosr_device_count(void);
/* The NUMA node the device hangs off (its PCI function's numa_node in sysfs), or -1 where the system does not say.  A host
 * that packs per-frame arrays into page-locked memory and uploads them (mvosr_pack_fill + mvosr_memcpy_h2d_async) is 8-10 %
 * faster, and steadier, with its threads on that node's CPUs: staging memory is then local to the copy engine's root port
 * (measured on a two-socket host: 502-505 k frames/s on the device's node, 460-488 k on the other, 441-468 k unpinned). */
int mvosr_device_numa_node(int device);

/* One context = one device + one HIP stream + a small workspace. */
int mvosr_ctx_create(int device, mvosr_ctx **out);
int mvosr_ctx_destroy(mvosr_ctx *ctx);
/* Adopt an external stream (e.g. torch.cuda.current_stream().cuda_stream); NULL restores the
 * context's own stream. */
int mvosr_ctx_set_stream(mvosr_ctx *ctx, void *hip_stream);
void *mvosr_ctx_stream(mvosr_ctx *ctx);
int mvosr_ctx_sync(mvosr_ctx *ctx);
/* Pre-size the context's workspace (the dense lists of selected y' the scale kernel hands to the
 * road-model kernel: total_feat doubles + n_frames int32; dense batches with survivor-numbered rows
 * need 24 more bytes per feature, allocated at their first launch).  mvosr_scale_batch grows it on demand
 * with hipMalloc; call this first if the launches must not allocate (e.g. under graph capture). */
int mvosr_ctx_reserve(mvosr_ctx *ctx, int64_t n_frames, int64_t total_feat);
/* Cap, in bytes, on the grow-only workspace of the triangulation kernels (mvosr_delaunay_*_batch: per-frame state beyond LDS —
 * ~0.55 KB per point of frames x largest frame for the Qhull replay): a launch that would need more returns MVOSR_ERR_ALLOC
 * without launching anything, exactly as when the device cannot satisfy the request, and the drop-in sends that chunk through
 * the host's triangulations.  For hosts that share the device with another tenant's allocations.  0 (default): no cap. */
int mvosr_ctx_workspace_limit(mvosr_ctx *ctx, int64_t bytes);
/* Per-call kernel timing of mvosr_scale_batch with HIP events on the launch stream: after
 * mvosr_ctx_profile(ctx, 1) every call (up to 64) records an event before the scale kernel, between
 * the two kernels and after the road-model kernel; mvosr_ctx_profile_read returns the two
 * durations of call `call_index` (0-based since the enable; synchronises on that call's end). */
int mvosr_ctx_profile(mvosr_ctx *ctx, int enable);
int mvosr_ctx_profile_read(mvosr_ctx *ctx, int call_index, float *scale_kernel_ms, float *road_kernel_ms);
/* Device facts for the host (name, CU count, LDS per workgroup). */
int mvosr_ctx_device_info(mvosr_ctx *ctx, char *name, int name_len, int *n_cu, int *lds_per_block);

/* ---- device memory / events (so that a ctypes-only host needs no other GPU library) ------
 * mvosr_malloc / mvosr_host_alloc are CACHING allocators: a block that is freed goes to a size-ordered free list of the
 * context instead of hipFree / hipHostFree, and the next request of a similar size takes it from there — a caller in
 * steady state (a chunk loop, the per-frame drop-in call at /root/reference/src/main.py:110-113) allocates nothing.  A
 * freed block may still be in use by work queued on the context's streams: its next user waits for that work
 * (what hipFree's implicit synchronisation gave).  mvosr_ctx_trim returns the cached blocks to the runtime;
 * mvosr_ctx_alloc_stats reports {hipMalloc calls, hipFree calls, hipHostMalloc calls, hipHostFree calls, cache hits,
 * cached device bytes, cached host bytes, live blocks} (the first n_out of them). */
int mvosr_malloc(mvosr_ctx *ctx, size_t bytes, void **dptr);
int mvosr_free(mvosr_ctx *ctx, void *dptr);
int mvosr_host_alloc(mvosr_ctx *ctx, size_t bytes, void **hptr);                   /* page-locked host memory (staging) */
int mvosr_host_free(mvosr_ctx *ctx, void *hptr);
int mvosr_ctx_trim(mvosr_ctx *ctx);
/* A cached block's release normally marks "everything queued so far" as its last use — in a chunk loop that includes the
 * NEXT chunk's kernels, and the block's next user would wait for them.  mvosr_block_mark(ctx, ptr, 1) records the last use
 * NOW (call it right after the last launch that touches the block) and a later mvosr_free keeps that mark;
 * MVOSR_MARK_UPLOAD: the block (a staging buffer) is used by the upload stream only — its last use is what that stream
 * has queued so far; MVOSR_MARK_IDLE: the caller knows that nothing queued uses the block (a download target whose copy
 * it has waited for); 0 withdraws a mark (the block is used again). */
#define MVOSR_MARK_NOW 1
#define MVOSR_MARK_IDLE 2
#define MVOSR_MARK_UPLOAD 3
int mvosr_block_mark(mvosr_ctx *ctx, void *ptr, int marked);
int mvosr_ctx_alloc_stats(mvosr_ctx *ctx, int64_t *out, int n_out);
int mvosr_memcpy_h2d(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes);   /* stream-ordered, returns after the copy */
int mvosr_memcpy_d2h(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes);   /* stream-ordered, returns after the copy */
/* Asynchronous forms.  Uploads run on the context's UPLOAD stream (so that the next chunk's inputs travel under the
 * current chunk's kernels; `src` should be page-locked — mvosr_host_alloc — or the copy is staged by the runtime) and
 * mvosr_upload_fence makes the compute stream wait for everything uploaded so far (no host wait).  Downloads are
 * queued on the compute stream; the data is there after mvosr_ctx_sync. */
int mvosr_memcpy_h2d_async(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes);
int mvosr_upload_fence(mvosr_ctx *ctx);
int mvosr_memcpy_d2h_async(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes);
/* Device -> PAGE-LOCKED host memory (mvosr_host_alloc) by a KERNEL on the context's compute stream, asynchronous: no copy engine is
 * involved.  For the results of a streamed chunk: a hipMemcpyAsync queued on the compute stream behind long kernels parks the SDMA engine
 * HIP assigns it until those kernels end, and an upload of ANOTHER stream that lands on that engine waits with it (whether it does is
 * decided per process); a copy issued after the kernels, on any stream, waits the other way round behind an upload in flight.  Record
 * an event behind this call and wait for it (mvosr_event_sync / _query) before reading `dst`. */
int mvosr_memcpy_d2h_kernel(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes);
int mvosr_memset(mvosr_ctx *ctx, void *dst, int value, size_t bytes);
int mvosr_event_create(mvosr_ctx *ctx, void **event);
int mvosr_event_record(mvosr_ctx *ctx, void *event);          /* on the context's current stream */
int mvosr_event_elapsed_ms(mvosr_ctx *ctx, void *start, void *stop, float *ms);  /* synchronises on `stop` */
int mvosr_event_sync(mvosr_ctx *ctx, void *event);             /* host waits for the event (not for later work of the stream) */
int mvosr_event_query(mvosr_ctx *ctx, void *event, int *done);  /* *done = 1 when the work recorded before the event has finished, else 0; never waits */
int mvosr_event_destroy(mvosr_ctx *ctx, void *event);

/* ---- host-side packing (no GPU work) ---------------------------------------------------------
 * The per-frame arrays of the reference's call surface — feature3d (N,3) and feature2d (N,2), C-contiguous float64, as
 * /root/reference/src/main.py:102-113 hands them to scale_calculation — laid out as the planes of mvosr_batch by
 * `threads` host threads (<= 0: one per hardware thread, at most 16).  mvosr_pack_count applies the vanishing-row filter
 * (/root/reference/src/scale_calculator.py:252-254: feature2d[:,1] > vanish) and returns the survivors per frame; the
 * caller derives feat_off (even, ascending) and sizes the planes — e.g. page-locked staging memory, so that the packed
 * batch is uploaded without another copy —; mvosr_pack_fill writes x|y|z|u|v at feat_off[f] in the caller's order and,
 * when feat_cnt_out is given, the number of features it kept per frame — so a caller that lays the frames out by their
 * UNFILTERED sizes (feat_off from n_points: a few per cent of slack) needs no counting pass at all.
 * remap_in_place != 0 also applies feature_remap (:390-394) to EVERY row of the caller's feature3d arrays, as the
 * reference does at :414 (the planes keep the raw values: the kernels remap at load). */
int mvosr_pack_count(int64_t n_frames, const double *const *feature2d, const int32_t *n_points, double vanish, int32_t *feat_cnt,
                     int threads);
int mvosr_pack_fill(int64_t n_frames, double *const *feature3d, const double *const *feature2d, const int32_t *n_points,
                    double vanish, const int64_t *feat_off, double *x, double *y, double *z, double *u, double *v,
                    int remap_in_place, double cos_pitch, double sin_pitch, int threads, int32_t *feat_cnt_out);

/* ---- the hot path ------------------------------------------------------------------------- */
void mvosr_default_params(mvosr_params *p, double absolute_reference);

/*
 * Fused per-frame scale recovery: replaces, for every frame of the batch, the body of
 * ScaleEstimator.scale_calculation (scale_calculator.py:411-422) between the two host
 * Delaunay calls and the window median:
 *   feature_remap (:390-394) -> find_outliers/check_triangle on tri1 (:151-167,:105-119)
 *   -> compaction of the survivors (:264-265) -> feature_selection_by_tri on tri2 (:225-248)
 *   -> road_model_calculation_static (:324-354) -> raw scale (:419/:421).
 * Two launches on the context's stream: the scale kernel (remap, vote, compaction, triangle
 * sweeps; one workgroup per frame, its features resident in LDS) leaves every frame's selected
 * y' as a dense list in the context's workspace, and the road-model kernel (one WAVEFRONT per
 * frame, no LDS-resident frame, full occupancy) turns each list into height / scale / status.
 * `waves_per_frame` selects the scale kernel's variant: 0 = choose from max_feat (measured crossovers:
 * 1 up to 320 features, 4 up to 1152, 8 while two workgroups fit a CU's LDS — about 3000 —, 16 above),
 * 1 = one wavefront per frame (<= 512 features), 4/8/16 = one workgroup of that many wavefronts.  With 0, a batch
 * of at least 2048 frames whose sizes span more than one of those ranges (b->min_feat) is split on the device into
 * its size classes, one launch per class over the class's frame list (results do not depend on the variant).
 * Frames that do not fit LDS in fp64 (max_feat > mvosr_max_lds_features(), about 6200) and batches with
 * feature-numbered second triangulations (b->tri2_ids) run the gather variant, which keeps only the vote
 * counters in LDS; with survivor-numbered rows it uses 24 bytes of context workspace per feature.
 * `first_frame`/`n_launch` restrict the launch to a sub-range of the batch (n_launch <= 0: all).
 * MVOSR_WAVES_EXACT or-ed into `waves_per_frame` runs the frames in the exact mode — height_level summed in NumPy's
 * order for every frame, what the stage outputs select implicitly — without asking for stage outputs (the host re-runs
 * single frames this way when a later step will read their level).
 */
#define MVOSR_WAVES_EXACT 0x100
/* or-ed into waves_per_frame: ONLY the frames of the range with a non-zero exact_mask byte are (re)done, in the exact mode;
 * every other frame's outputs stay as they are (the host asks for the neighbours of a frame that raised, once it knows). */
#define MVOSR_WAVES_EXACT_MASKED 0x200
/* or-ed into waves_per_frame: the product (HOT) kernels and the road model ONLY, whatever outputs are asked for (selected,
 * vote_counters: written; per-triangle outputs: refused) — no exact pass, no exact_mask.  A frame the exact pass would have
 * redone (its level may decide its result in the last bit, or IS its result) is left with status MVOSR_ST_REDO and no result:
 * the caller redoes it.  Every other frame's raw_scale, status, counts and selected are final; its height_level is the kernel's
 * own fixed-order sum (equal to NumPy's to ~1e-16 relative, not bit for bit).  The per-frame call of the reference-exact
 * estimator runs this way on stand-in rows and asks SciPy for the second triangulation only when a frame comes back
 * MVOSR_ST_REDO or its exact level is read (mvoscalerecovery_amd/scale_calculator.py). */
#define MVOSR_WAVES_HOT_ONLY 0x400
#define MVOSR_ST_REDO (-2)
int mvosr_scale_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b,
                      const mvosr_outputs *o, int waves_per_frame,
                      int64_t first_frame, int64_t n_launch);

/* Host helper (no GPU work): from the HOST copy of the batch's feat_cnt set b->max_feat, b->min_feat and
 * b->size_hint.  Optional — see mvosr_batch.size_hint. */
int mvosr_batch_size_hint(const int32_t *feat_cnt_host, int64_t n_frames, mvosr_batch *b);

/*
 * Stage K1 alone: find_outliers (scale_calculator.py:151-167) for every frame; writes
 * o->vote_counters (required) and counts[MVOSR_CNT_VALID] (optional).  Used by the per-frame
 * drop-in call and by the batch path when the second triangulation still has to be built on
 * the host from the surviving features.  Only b->feat_*, y, z, v, tri1* are read.
 */
int mvosr_outlier_vote_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b,
                             const mvosr_outputs *o, int waves_per_frame);

/*
 * Stage K3 alone: road_model_calculation_static (scale_calculator.py:324-354) on packed
 * lists of already-remapped y values (feat_off/feat_cnt/y of `b`; nothing else is read);
 * `height_level_in` [F] supplies the fallback level (:335).  Writes height, status and the
 * optional counts/hist/stats; raw_scale = absolute_reference/height.  Same kernel as the second
 * launch of mvosr_scale_batch (one wavefront per list); `waves_per_frame` is ignored.
 */
int mvosr_road_model_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b,
                           const double *height_level_in, const mvosr_outputs *o,
                           int waves_per_frame);

/*
 * K4: scale_filtering (scale_calculator.py:396-400) over a whole sequence: out[i] = median of
 * the last `window` pushed raw scales, the deque pre-loaded with `queue_in[0..n_queue)`.
 * raw/out are device pointers of n doubles; queue_in is a HOST pointer (n_queue <= window).
 */
int mvosr_window_median(mvosr_ctx *ctx, const double *raw, int64_t n, int window,
                        const double *queue_in, int n_queue, double *out);

/*
 * K4 on a sequence that was all-gathered from `n_blocks` ranks and is still in its gathered form:
 * block r starts at blocks + r*block_stride (in doubles) and holds rank r's contiguous share of the
 * n frames (shares as in a contiguous partition: the first n % n_blocks ranks one frame more than
 * n / n_blocks).  Same result as mvosr_window_median on the concatenated sequence; the gathered
 * buffer is read in place, so the multi-GPU step is ONE collective and no repacking kernel.
 */
int mvosr_window_median_blocked(mvosr_ctx *ctx, const double *blocks, int64_t n, int n_blocks, int64_t block_stride,
                                int window, const double *queue_in, int n_queue, double *out);

/* ---- the `rescale` variant (the estimator /root/reference/src/main.py:20 imports) ------------- */

/*
 * GraphChecker.find_inliers (/root/reference/src/graph.py:18-36): per feature the number of incident
 * triangles of tri1 (`total`) and how many of them give the vertex a marginal > 0.6 (`good`); the host
 * keeps a feature when good/total > 0.5 (graph.py:35; 0/0 -> dropped).  `good_bits`: bit 3*code+k is
 * set when vertex k of a triangle with edge-order code `code` (graph.py:124-129) has marginal > 0.6
 * under the 8x8 triangle potential (graph.py:6-17,134-145) — 24 bits computed once on the host.
 * Reads feat_off/feat_cnt/z/v/tri1_off/tri1 of `b` (no remap: rescale.py:25 sets camera_pitch = 0).
 * total/good: int32 device arrays laid out like z.  status [F] (optional): 0 or MVOSR_ST_ERR_MASK.
 */
int mvosr_graph_inliers_batch(mvosr_ctx *ctx, const mvosr_batch *b, uint32_t good_bits, int32_t *total, int32_t *good,
                              int32_t *status);

/*
 * ScaleEstimator.flat_selection (/root/reference/src/rescale.py:75-102) on the features that
 * survived (x/y/z of `b`, already compacted by the host) and tri2: per triangle heights = 1/|n| with
 * n = A^-1.1, flags bit0 pitch < loose_deg (-80), bit1 pitch < tight_deg (-85), bit2 kept = bit1 and
 * heights > height_factor (0.9) * median(heights[bit0]).  tri_height/tri_flags: per triangle of tri2;
 * height_level/n_kept/status: per frame (status 0, MVOSR_ST_ERR_SINGULAR, _MASK or _EMPTY).
 * max_tri: largest triangle count of a frame (sizes LDS; <= 0: 2*max_feat).
 */
int mvosr_flat_selection_batch(mvosr_ctx *ctx, const mvosr_batch *b, double loose_deg, double tight_deg, double height_factor,
                               double *tri_height, uint8_t *tri_flags, double *height_level, int32_t *n_kept, int32_t *status,
                               int64_t max_tri);

/*
 * run_ransac (/root/reference/src/thirdparty/Ransac/ransac.py:3-23) with estimate / is_inlier of
 * /root/reference/src/estimate_road_norm.py:8-18 for every frame, made deterministic by taking the
 * sample sequence as input: triples[f][h][0..2] are row indices into frame f's points (planes
 * px/py/pz, pts_off/pts_cnt per frame), consumed in order h = 0..n_hyp-1.  A hypothesis replaces the
 * best when its inlier count (|m.[p,1]| < threshold over ALL points) is strictly larger; the scan
 * stops at the first improvement whose count exceeds goal_fraction * n_points.  Outputs: counts
 * [F][n_hyp] (optional), model [F][4] = unit (n, d) of the best plane with n_y >= 0
 * (/root/reference/src/rescale.py:159-161), best_ic [F], used [F] (hypotheses consumed).
 */
int mvosr_ransac_plane_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                             const double *px, const double *py, const double *pz, const int32_t *triples, int n_hyp,
                             double threshold, double goal_fraction, int32_t *counts, double *model, int32_t *best_ic,
                             int32_t *used);

/*
 * The 2-D line variant, get_pitch_line_ransac (/root/reference/src/estimate_road_norm.py:60-64) with estimate_line /
 * is_inlier_line (:39-49): same kernel and replay rule; samples are pairs, stored like triples
 * (pairs[f][h][0..1] used, [2] ignored), points are (px, py); model [F][4] = unit (a, b, 0, c) of the best line
 * a x + b y + c = 0 with b >= 0 (the reference's SVD null vector has an arbitrary sign).
 */
int mvosr_ransac_line_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                            const double *px, const double *py, const int32_t *pairs, int n_hyp,
                            double threshold, double goal_fraction, int32_t *counts, double *model, int32_t *best_ic,
                            int32_t *used);

/* ---- the `rescale` variant, device-resident (triangulations built by mvosr_delaunay_batch, nothing visits the host) ---- */

typedef struct mvosr_rescale_params {
    uint32_t good_bits;          /* as mvosr_graph_inliers_batch */
    int32_t min_valid;           /* 10: the vote's survivors are re-triangulated only when more than this many are left
                                    (/root/reference/src/rescale.py:133); otherwise every feature stays               */
    double loose_deg, tight_deg; /* -80, -85 (rescale.py:85-86) */
    double height_factor;        /* 0.9 (rescale.py:91) */
    int32_t ransac_min_points;   /* 12 (rescale.py:152) */
    int32_t n_hyp;               /* 100 (rescale.py:155); <= 512 */
    double threshold;            /* 0.005 (rescale.py:155) */
    double goal_fraction;        /* 0.8 (/root/reference/src/estimate_road_norm.py:68) */
    double absolute_reference;   /* real camera height (rescale.py:167) */
    uint64_t seed;               /* key of the sample sequence (below) */
    int64_t frame_base;          /* counter of frame 0 of the batch in the sample sequence (frame f: frame_base + f) */
} mvosr_rescale_params;

/*
 * GraphChecker.find_inliers (/root/reference/src/graph.py:18-36) with the decision made on the device:
 * keep[i] (laid out like z) = 1 where good/total > 0.5 (graph.py:34-35: 2*good > total in integers, 0/0 dropped),
 * -1 where not — unless min_valid or fewer features pass, in which case the reference keeps the frame as it is
 * (/root/reference/src/rescale.py:133-137) and the failed ones get 0: `keep >= 0` is the set the second triangulation is
 * built on (what mvosr_delaunay_batch's `keep` wants), `keep > 0` the vote itself.  n_valid[f] (optional) = features that
 * passed the vote.  Rows: b->tri1 at b->tri1_off[f], b->tri1_cnt[f] of them when tri1_cnt is given (the form
 * mvosr_delaunay_batch writes), else tri1_off[f+1] - tri1_off[f].  A frame whose first triangulation was declined
 * (dt_status[f] != 0, optional) is skipped.
 */
int mvosr_graph_keep_batch(mvosr_ctx *ctx, const mvosr_batch *b, uint32_t good_bits, int32_t min_valid, const int32_t *dt_status,
                           int32_t *keep, int32_t *n_valid, int32_t *status);

/* per-frame outputs of mvosr_flat_ransac_batch (device pointers; the optional ones may be NULL) */
typedef struct mvosr_rescale_outputs {
    double *raw_scale;           /* [F] absolute_reference / camera height of the RANSAC plane (rescale.py:156-167); NaN when
                                    status != 0 */
    double *height_level;        /* [F] 0.9 * median(heights[pitch < -80]) (rescale.py:91-92) */
    double *model;               /* [F][4] unit (n, d) of the best plane, n_y >= 0 (rescale.py:159-161) */
    int32_t *best_ic, *used;     /* [F] inlier count of the best hypothesis, hypotheses consumed (ransac.py:9-22) */
    int32_t *n_kept;             /* [F] kept triangles (rescale.py:96); the point list has 3 * n_kept entries (:101) */
    int32_t *status;             /* [F] 0, MVOSR_ST_RS_FEW, MVOSR_ST_ERR_SINGULAR (rescale.py:79 raises), _MASK, _EMPTY */
    double *tri_height;          /* optional [rows of tri2, laid out like tri2] 1/|n| (rescale.py:89) */
    uint8_t *tri_flags;          /* optional [like tri_height] bit0 pitch < loose, bit1 pitch < tight, bit2 kept */
    int32_t *hyp_counts;         /* optional [F][n_hyp] inlier count of every hypothesis */
} mvosr_rescale_outputs;

/*
 * flat_selection (rescale.py:75-102) + scale_calculation_ransac's plane fit (rescale.py:151-167) in ONE kernel per
 * frame: the features with keep[i] >= 0 (keep == NULL: all) are compacted, in order, into LDS at load; b->tri2 (rows
 * numbered over those survivors; b->tri2_cnt[f] rows at b->tri2_off[f], or the offsets' difference) gives heights,
 * flags, the median level and the kept rows as mvosr_flat_selection_batch does; the kept rows' vertices, in row order
 * with repeats (rescale.py:101), are the RANSAC's point list, which never leaves LDS.  The reference draws its sample
 * triples from OS entropy (/root/reference/src/thirdparty/Ransac/ransac.py:6,10), so any uniform draw of three distinct
 * list positions per hypothesis is a realisation of it; here hypothesis h of frame f takes
 *     key = mix(seed ^ (frame_base + f) * 0xD1B54A32D192ED03),  hk = mix(key + h),  r_k = mix(hk + 3 a + k),  k = 0, 1, 2
 *     i0 = mulhi(r_0, M), i1 = mulhi(r_1, M - 1) skipping i0, i2 = mulhi(r_2, M - 2) skipping both      (mix = splitmix64's finaliser)
 * for attempt a = 0, 1, ... until the three positions name three DIFFERENT vertices (at most 16 attempts): the list repeats
 * every vertex once per kept triangle, and a sample with a repeated vertex is rank-deficient — the reference's SVD then
 * returns whatever plane of the pencil through two points rounding noise selects; the product's sequence leaves those out
 * — a counter-based sequence that depends on (seed, frame counter, hypothesis) only, so a batch, its chunks and
 * per-frame calls draw the same triples (oracle/rescale_oracle.py restates it).  `id_triples` (optional, [F][n_hyp][3]
 * survivor-numbered VERTEX ids) replaces the draw: a recorded sample sequence of the reference mapped to point ids
 * replays its run whatever the row order.  `frame_ids` (optional, [F]) replaces frame_base + f (re-runs of single frames).
 * The replay rule is mvosr_ransac_plane_batch's.  max_tri: largest row count of a frame (<= 0: 2 * b->max_feat).
 */
int mvosr_flat_ransac_batch(mvosr_ctx *ctx, const mvosr_batch *b, const int32_t *keep, const mvosr_rescale_params *rp,
                            const int32_t *id_triples, const int64_t *frame_ids, const int32_t *dt_status,
                            const mvosr_rescale_outputs *o, int64_t max_tri);

/*
 * The cross-frame tail of scale_calculation_ransac (rescale.py:169-178) over a run of frames, on the device: the slew
 * limiter — a frame with apply[i] != 0 moves the running scale towards raw[i] by at most `slew` (0.3), any other frame
 * leaves it — followed by the window median of the pushed values (np.median(self.scale_queue)).  raw/apply/pushed/
 * filtered are device arrays of n; scale_in and queue_in[0..n_queue) (HOST pointer) are the estimator's state before the
 * run.  apply: int32, non-zero where the frame has a RANSAC model (status 0).  One wavefront walks the sequence (the
 * limiter is a sequential recurrence of rounded additions), the median is mvosr_window_median's kernel.
 */
int mvosr_slew_median(mvosr_ctx *ctx, const double *raw, const int32_t *apply, int64_t n, double slew, double scale_in,
                      int window, const double *queue_in, int n_queue, double *pushed, double *filtered);

/* The same on HOST arrays, without the GPU: the recurrence is sequential, and when the per-frame results are on the host
 * anyway (the estimator needs the statuses there to raise where the reference does) one core walks a million frames in a
 * few milliseconds — the device kernel's single wavefront needs 55 ns per frame.  scale_out (optional): the running scale
 * after the run. */
int mvosr_slew_median_host(const double *raw, const int32_t *apply, int64_t n, double slew, double scale_in, int window,
                           const double *queue_in, int n_queue, double *pushed, double *filtered, double *scale_out);

/*
 * The legacy per-triangle batch of /root/reference/src/triangle_batch.py:14-68: features are
 * [u, v, depth] (b->x = u, b->v = v, b->z = depth; b->tri1 = Delaunay over (u,v), :23-25).  Per
 * triangle: back-projection with (focus, cx, cy) (:32-33), n = A^-1.1 (:36-37), s = n_y/|n| (:43),
 * h = mean y (:40); keep s > s_min (0.98) and h > 0 (:54-55); mean/std (:57-58); drop values outside
 * mean +- n_sigma (3) std (:60-61); height[f] = mean of the rest (:62).  counts [F][2] = kept / kept
 * after the clip; status [F] = 0, MVOSR_ST_ERR_SINGULAR, _MASK or _EMPTY.
 */
int mvosr_triangle_batch(mvosr_ctx *ctx, const mvosr_batch *b, double focus, double cx, double cy, double s_min,
                         double n_sigma, double *height, int32_t *counts, int32_t *status);

/* get_inliers, /root/reference/src/estimate_road_norm.py:71-78: mask[i] = |n.p_i + d| < threshold for a
 * plane model4 = (n, d) given on the HOST; px/py/pz/mask are device arrays of n elements. */
int mvosr_plane_inliers(mvosr_ctx *ctx, int64_t n, const double *px, const double *py, const double *pz, const double *model4,
                        double threshold, uint8_t *mask);

/* ---- optional device stage for the triangulations themselves -------------------- */

enum mvosr_dt_status {
    MVOSR_DT_OK = 0,
    MVOSR_DT_DEGENERATE = 1      /* duplicate / collinear / cocircular points within the guard bands, or a row count that
                                    is not Euler's 2n - 2 - h: rows are not to be used — triangulate this frame on the host */
};

/*
 * Batched 2-D Delaunay triangulation: what scipy.spatial.Delaunay(points).simplices computes at
 * /root/reference/src/scale_calculator.py:257-258 and :266-267, as a device stage.  For points in general position the
 * triangle SET is the one Qhull returns; the rows come in a canonical form — vertex ids ascending inside a row, rows in
 * lexicographic order — a function of the set alone.  Qhull's own rotation of each row, which the reference's vote
 * depends on (:113-115), is not reproducible from the geometry: these rows are meant for MVOSR_VOTE_FIXED, under which
 * they and SciPy's rows give bit-identical results (the host selects both with triangulation="gpu",
 * check_triangle="fixed"); with MVOSR_VOTE_REFERENCE they are a measured deviation.
 * Frame f's points are (u, v)[pts_off[f] .. +pts_cnt[f]); with `keep` (laid out like u; e.g. the vote counters of
 * mvosr_outlier_vote_batch) only the points with keep[i] >= 0 take part and the ids are their ranks among those, in
 * order — the second triangulation over the survivors of the vote (:264-266) without a compaction pass.  n_used[f]
 * (optional) = the number of points triangulated (what mvosr_batch.n2_expected wants).  Rows are written at
 * tri + 3*tri_off[f] (room for 2 * points rows), tri_cnt[f] says how many; status[f] is an mvosr_dt_status (a declined
 * frame has tri_cnt 0 and the reason in the status' bits 8..).  max_pts = max(pts_cnt) <= mvosr_delaunay_max_points();
 * The context's workspace (grow-only, hipMalloc when it grows) holds 76 bytes per point of the LAUNCH — n_frames * max_pts
 * points: the stars' hint caches and the order the points are taken in — and ~33 more above mvosr_delaunay_lds_points(): callers with very many
 * frames launch in chunks (the host class uses up to 8192 frames or 10 M points).
 */
int mvosr_delaunay_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                         const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                         int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status);
/* The same with SEEDS: rows (seed_tri + 3*seed_off[f], seed_cnt[f] of them) of a triangulation of ALL the frame's points,
 * ids = positions in (u, v) — what a call without `keep` wrote for the same frames.  A triangle of that triangulation whose
 * three vertices are kept is a triangle of this one (its circumcircle was empty among more points), so it is not searched
 * for again: the second triangulation of a frame (:264-266, over the ~85 % of the points the vote keeps) starts from the
 * ~60 % of the first one's triangles that survive.  Same rows as without seeds.  Seeds that are not a Delaunay triangulation
 * of the frame's points give undefined rows. */
int mvosr_delaunay_batch_seeded(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                                int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status,
                                const int64_t *seed_off, const int32_t *seed_tri, const int32_t *seed_cnt);
/* The same with the per-point facts that let the SECOND triangulation skip the stars the vote did not touch.  info_out
 * (optional; laid out like u, one word per point; only for a call without `keep`): rows the point owns | star degree << 6 |
 * hull flag << 15, and in the high half the index of its first row within the frame's rows.  seed_info (optional, with
 * seeds): what the call that built the seeds wrote.  A kept point none of whose seed triangles lost a vertex has the same
 * star among the survivors (its triangles keep their empty circles and still close the fan): its rows are copied with the
 * ids mapped to ranks and its star is not walked — at 95 % kept points three stars in four.  Same rows as without. */
int mvosr_delaunay_batch_ex(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                            const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                            int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status,
                            const int64_t *seed_off, const int32_t *seed_tri, const int32_t *seed_cnt,
                            const uint32_t *seed_info, uint32_t *info_out);
/* Largest frame mvosr_delaunay_batch takes (32 000 points: ids and row indices are 16-bit), and the largest frame whose
 * points, grid and rows fit one workgroup's LDS (about 4 700): a launch whose max_pts is larger runs the kernel's
 * global-memory variant — the same algorithm with the per-frame arrays in a slice of the context's workspace, read
 * through L1/L2 (dense frames, BASELINE configs[4]); slower per point, same rows. */
int mvosr_delaunay_max_points(void);
int mvosr_delaunay_lds_points(void);
/* Frames of a batch whose largest frame has max_pts points that ONE compute unit works on at a time in a launch of 512 frames
 * and more (the launcher's choice of wavefronts per frame and of the arena's home: 8, 4, 3, 2 or 1; 1 for the global-memory
 * variant).  A host that cuts a long batch into chunks makes a chunk a multiple of this x the device's CU count: the kernel
 * then has no partly filled last round. */
int mvosr_delaunay_frames_per_cu(int max_pts);

/*
 * SciPy/Qhull's rows themselves: the triangle set, the ORDER of the rows and the ROTATION of every row of
 * scipy.spatial.Delaunay(points).simplices (/root/reference/src/scale_calculator.py:257-258, :266-267) — what the reference's
 * check_triangle (:105-119) reads.  The kernel replays the beneath-beyond of the Qhull SciPy bundles (qhull_r 7.3.2,
 * options `d Qbb Qc Qz Q12 Qt`) decision by decision, one wavefront per frame, for points in general position; a frame in
 * which a decision falls inside a roundoff guard band (Qhull would merge facets) is declined like mvosr_delaunay_batch
 * declines (status = MVOSR_DT_DEGENERATE | reason << 8, tri_cnt 0): the host triangulates it with SciPy.  Rows are meant for
 * MVOSR_VOTE_REFERENCE: with them triangulation="gpu" IS the reference's result.  Arguments as mvosr_delaunay_batch;
 * order_out (optional, laid out like u): the insertion step at which a point became a vertex (0: initial simplex), indexed
 * by the point's rank among the kept points.  max_pts <= mvosr_delaunay_qhull_max_points() (60 000; launches whose max_pts is above
 * 8 000 use 32-bit facet ids and 80-byte facet records).  Workspace: ~0.55 KB (0.66 KB) per point of the launch
 * (n_frames * (max_pts + 1)), grow-only, in the context.
 */
int mvosr_delaunay_qhull_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                               const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                               int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status, int32_t *order_out);
int mvosr_delaunay_qhull_max_points(void);
/*
 * The same replay for ONE point set on the HOST (plain C, no device, no context; thread-safe: per-thread workspace) — replaces the
 * reference's scipy.spatial.Delaunay call for the first triangulation of a per-frame scale_calculation
 * (/root/reference/src/scale_calculator.py:257 from src/main.py:110-113): one frame on the device is a chain of ~n dependent
 * insertions (20 ms at 2000 points); SciPy is 2.6 ms; this loop, which skips what Qhull does for inputs that are not in general
 * position and what SciPy builds around the rows, is several times faster.  points: n_points rows of (x, y), stride_doubles apart
 * (2 for a C-contiguous (n, 2) array); rows: room for rows_cap (>= 2 n_points) int32 triples; *n_rows: rows written; order
 * (optional, n_points): the step at which a site became a vertex.  Returns 0 (rows = SciPy's simplices, bit for bit), > 0: declined
 * — a decision inside a roundoff guard band, fewer than 3 points; the reason code as in the batch kernel's status >> 8; the
 * caller asks SciPy —, < 0: MVOSR_ERR_ARG / MVOSR_ERR_ALLOC.
 */
int mvosr_qhull_rows_host(const double *points, int64_t n_points, int64_t stride_doubles, int32_t *rows, int64_t rows_cap,
                          int32_t *n_rows, int32_t *order);

/* LDS bytes the fused kernel requests for a frame of n features (host-side planning). */
size_t mvosr_lds_bytes(int n_features);
/* Largest frame the LDS-resident variant accepts on this build. */
int mvosr_max_lds_features(void);

#ifdef __cplusplus
}
#endif
#endif /* MVOSR_H */
