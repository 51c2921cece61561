// mvosr_rescale.hip — kernels for the `rescale` variant of the scale estimator (the one
// /root/reference/src/main.py:20 imports): GraphChecker vote, flat_selection, and the RANSAC plane
// fit made deterministic by taking its sample triples as input.  SURVEY.md §8 rows f2, f4, a11, a13.
//
//   graph_inliers_kernel   GraphChecker.find_inliers        /root/reference/src/graph.py:18-36,124-145
//   flat_selection_kernel  ScaleEstimator.flat_selection    /root/reference/src/rescale.py:75-102
//   ransac_plane_kernel    run_ransac + estimate/is_inlier  /root/reference/src/thirdparty/Ransac/ransac.py:3-23,
//                                                           /root/reference/src/estimate_road_norm.py:8-18,66-70
//
// Same conventions as mvosr_kernels.hip: one frame per workgroup, fp64, planes staged in LDS,
// compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mvosr.h"
#include "mvosr_device.hpp"
#include "mvosr_host.hpp"

namespace mvosr {

constexpr int kRsWaves = 8;
constexpr int kRsBlock = kRsWaves * kWave;

// ---------------------------------------------------------------------------------------------
// GraphChecker.find_inliers: per triangle the edge-order code a*4+b*2+c with
// a=(v0-v1)(d0-d1)<0, b=(v1-v2)(d1-d2)<0, c=(v0-v2)(d0-d2)<0 selects a column of the 8x8 triangle
// potential; the three vertex marginals of that column are compared with 0.6 on the HOST once (they
// depend on the code only), so the kernel receives 8x3 flag bits.  Each vertex tallies
// (incident triangles, triangles that vouch for it) in one 32-bit LDS word (low/high half).
// ---------------------------------------------------------------------------------------------
struct GraphArgs {
    int64_t n_frames;
    const int64_t *feat_off; const int32_t *feat_cnt;
    const double *z, *v;
    const int64_t *tri_off; const int32_t *tri;
    uint32_t good_bits;            // bit 3*code+k: vertex k of a triangle with this code has marginal > 0.6
    int32_t *total, *good;         // [like z] outputs
    int32_t *status;               // [F] 0 ok / MVOSR_ST_ERR_MASK on a bad vertex id
    // the device-resident form (graph_inliers_kernel<true>, mvosr_graph_keep_batch): the decision itself
    const int32_t *tri_cnt;        // [F] rows of the frame (null: tri_off[f+1] - tri_off[f])
    const int32_t *dt_status;      // [F] non-zero: the frame's triangulation was declined — nothing to do (null: none)
    int32_t min_valid;             // rescale.py:133
    int32_t *keep;                 // [like z] 1 passed / 0 failed but the frame stays whole / -1 dropped
    int32_t *n_valid;              // [F] features that passed (null: not wanted)
};

template <bool KEEP>
__global__ __launch_bounds__(kRsBlock) void graph_inliers_kernel(const GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n = a.feat_cnt[f];
    if (n <= 0) { if (threadIdx.x == 0) { if (a.status) a.status[f] = MVOSR_ST_ERR_EMPTY; if (KEEP && a.n_valid) a.n_valid[f] = 0; } return; }
    if (KEEP && a.dt_status && a.dt_status[f] != 0) {
        // a declined first triangulation: the frame is redone on the host; its keep flags say "dropped" so that the second
        // triangulation and flat_selection of this chunk see an empty frame instead of whatever the recycled block held
        const int64_t off0 = a.feat_off[f];
        for (int i = threadIdx.x; i < n; i += kRsBlock) a.keep[off0 + i] = -1;
        if (threadIdx.x == 0) { if (a.status) a.status[f] = MVOSR_ST_ERR_EMPTY; if (a.n_valid) a.n_valid[f] = 0; }
        return;
    }
    const int64_t off = a.feat_off[f];
    const int64_t tb = a.tri_off[f];
    const int tn = a.tri_cnt ? a.tri_cnt[f] : (int)(a.tri_off[f + 1] - tb);
    double2 *P = reinterpret_cast<double2 *>(smem);                       // {v, z}
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + 16u * (uint32_t)((n + 1) & ~1));
    int *flag = reinterpret_cast<int *>(cnt + n + 4);                     // [0] bad vertex id, [1] features that passed
    const int tid = threadIdx.x;
    if (tid < 2) flag[tid] = 0;
    for (int i = tid; i < n; i += kRsBlock) {
        double2 p; p.x = a.v[off + i]; p.y = a.z[off + i];               // rescale.py:25: camera_pitch = 0, no remap
        P[i] = p;
        cnt[i] = 0u;
    }
    __syncthreads();
    int bad = 0;
    for (int t = tid; t < tn; t += kRsBlock) {
        const TriIds q = load_tri(a.tri + 3 * tb, t);
        if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) { bad = 1; continue; }
        const double2 p0 = P[q.a], p1 = P[q.b], p2 = P[q.c];
        const int ca = (p0.x - p1.x) * (p0.y - p1.y) < 0.0;              // graph.py:125
        const int cb = (p1.x - p2.x) * (p1.y - p2.y) < 0.0;              // graph.py:126
        const int cc = (p0.x - p2.x) * (p0.y - p2.y) < 0.0;              // graph.py:127
        const uint32_t g = a.good_bits >> (3 * (ca * 4 + cb * 2 + cc));  // graph.py:128,140-145
        atomicAdd(&cnt[q.a], 1u + ((g & 1u) << 16));                     // graph.py:28-30
        atomicAdd(&cnt[q.b], 1u + (((g >> 1) & 1u) << 16));
        atomicAdd(&cnt[q.c], 1u + (((g >> 2) & 1u) << 16));
    }
    if (bad) flag[0] = 1;
    __syncthreads();
    if constexpr (!KEEP) {
        for (int i = tid; i < n; i += kRsBlock) {
            const uint32_t c = cnt[i];
            a.total[off + i] = (int32_t)(c & 0xFFFFu);
            a.good[off + i] = (int32_t)(c >> 16);
        }
    } else {
        // good/total > 0.5 (graph.py:34-35,131-132): the quotient of two integers below 2^16 is above one half exactly when
        // 2*good > total (the nearest quotient below is 1/2 - 1/(2 total), far from a rounding); 0/0 is nan: dropped
        int mine = 0;
        for (int i = tid; i < n; i += kRsBlock) { const uint32_t c = cnt[i]; mine += (2u * (c >> 16) > (c & 0xFFFFu)) ? 1 : 0; }
        mine = wave_sum(mine);
        if (lane_id() == 0 && mine) atomicAdd(&flag[1], mine);
        __syncthreads();
        const int nv = flag[1];
        const int fail = nv > a.min_valid ? -1 : 0;                       // rescale.py:133-137: ten or fewer left — the frame stays whole
        for (int i = tid; i < n; i += kRsBlock) { const uint32_t c = cnt[i]; a.keep[off + i] = (2u * (c >> 16) > (c & 0xFFFFu)) ? 1 : fail; }
        if (tid == 0 && a.n_valid) a.n_valid[f] = nv;
    }
    if (tid == 0 && a.status) a.status[f] = flag[0] ? MVOSR_ST_ERR_MASK : 0;
}

// ---------------------------------------------------------------------------------------------
// flat_selection: per triangle n = A^-1 . 1 (LU, like np.matrix.I), heights = 1/|n|,
// pitch = asin(-n_y/|n|) deg; level = 0.9 * median(heights[pitch < -80]); a triangle is kept when
// pitch < -85 and heights > level.  The median is the mean of the two middle order statistics; every triangle's height
// and flags stay in LDS by row (no compacted list, no append counter), and the order statistic is found on the heights'
// bit patterns by a 2048-bin histogram over [smallest, largest] loose height — narrowed to the bin that holds the rank
// and repeated if needed — with wavefront 0 ranking the last <= 64 candidates directly.
// ---------------------------------------------------------------------------------------------
struct FlatArgs {
    int64_t n_frames;
    const int64_t *feat_off; const int32_t *feat_cnt;
    const double *x, *y, *z;
    const int64_t *tri_off; const int32_t *tri;
    double loose_deg, tight_deg, height_factor;
    double *tri_height;            // [T] 1/|n|
    uint8_t *tri_flags;            // [T] bit0: pitch < loose, bit1: pitch < tight, bit2: kept
    double *height_level;          // [F]
    int32_t *status;               // [F] 0 / MVOSR_ST_ERR_SINGULAR / MVOSR_ST_ERR_MASK / MVOSR_ST_ERR_EMPTY
    int32_t *n_kept;               // [F]
    // the device-resident form (flat_selection_kernel<true>, mvosr_flat_ransac_batch): survivors compacted at load, explicit
    // row counts, and the RANSAC plane fit over the kept rows' vertices as the kernel's tail (tri_height / tri_flags optional)
    const int32_t *tri_cnt;        // [F] rows of the frame (null: tri_off[f+1] - tri_off[f])
    const int32_t *keep;           // [like x] a feature takes part iff keep[i] >= 0 (null: all)
    const int32_t *dt_status;      // [F] non-zero: declined triangulation, frame skipped (null: none)
    const int32_t *id_triples;     // [F][H][3] survivor-numbered vertex ids replacing the draw (null: draw)
    const int64_t *frame_ids;      // [F] sample-sequence counter of the frame (null: frame_base + f)
    int32_t n_hyp, ransac_min_points, max_tri;
    double threshold, goal_fraction, absolute_reference;
    uint64_t seed; int64_t frame_base;
    double *raw_scale, *model;     // [F], [F][4]
    int32_t *best_ic, *used, *hyp_counts;
};

constexpr int kFlatBins = 2048;         // one histogram pass resolves 11 bits of the candidates' range
constexpr int kFlatRows = 4;            // triangle rows a thread keeps in flight
constexpr int kFlatDirect = 64;         // that few candidates left: wavefront 0 ranks them directly
constexpr int kMaxHyp = 512;
constexpr int kRansacPPT = 8;           // points per thread per chunk (chunks of 4096 points)
// misc[] slots of flat_selection_kernel (slots below FM_WSUM are zeroed at the start)
enum { FM_K = 0, FM_SINGULAR = 1, FM_BADID = 2, FM_KEPT = 3, FM_BIN = 4, FM_RANK = 5, FM_BINCNT = 6, FM_LE = 7, FM_LIST = 8, FM_ND = 9,
       FM_WSUM = 16 /* [16] per-wave bin totals */, FM_CW = 32 /* [16] per-wave counts of the ordered compactions */, FM_N = 48 };

// The sample sequence of the device-resident RANSAC (include/mvosr.h, mvosr_flat_ransac_batch): splitmix64's finaliser as a
// counter-based generator.  oracle/rescale_oracle.py restates it.
__host__ __device__ __forceinline__ uint64_t rs_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// Hypothesis h of a frame: three distinct list positions, uniform (ransac.py:10, random.sample over the list).  The list repeats
// every vertex once per kept triangle (rescale.py:101), so 0.5-2 % of the samples name one VERTEX twice.  The reference spends
// the iteration on such a sample (ransac.py:8-21): its SVD of the rank-2 matrix returns a plane that rounding noise picks from
// the pencil through two points.  Here the iteration is spent as well — the sample is NOT drawn again (rounds 4-5 did, a
// declared deviation that inflated the iteration budget) —: the cross product of a repeated vertex is exactly zero, the model
// NaN, the hypothesis counts zero inliers and can never be the best — what the id_triples path has always done with such a
// triple (pinned against the reference's own run on such triples: tests/golden/rescale.npz frame 26).
__device__ __forceinline__ void rs_draw3(uint64_t key, int h, int M, const uint16_t *L, int &v0, int &v1, int &v2) {
    const uint64_t hk = rs_mix64(key + (uint64_t)h);
    const uint64_t r0 = rs_mix64(hk), r1 = rs_mix64(hk + 1ull), r2 = rs_mix64(hk + 2ull);
    const int i0 = (int)__umul64hi(r0, (uint64_t)M);                          // uniform on [0, M) up to M / 2^64
    int i1 = (int)__umul64hi(r1, (uint64_t)(M - 1)); if (i1 >= i0) ++i1;       // ... on the M - 1 other positions
    int i2 = (int)__umul64hi(r2, (uint64_t)(M - 2));
    const int lo = min(i0, i1), hi = max(i0, i1);
    if (i2 >= lo) ++i2;
    if (i2 >= hi) ++i2;
    v0 = L[i0]; v1 = L[i1]; v2 = L[i2];
}

#ifndef MVOSR_FLAT_DEV_WAVES
#define MVOSR_FLAT_DEV_WAVES 16
#endif
constexpr int kFlatDevWaves = MVOSR_FLAT_DEV_WAVES;   // the device-resident form holds 96 KB of LDS — one workgroup per CU —, so it brings its own occupancy: 16 wavefronts

template <bool DEV, int WAVES = kRsWaves>
__global__ __launch_bounds__(WAVES *kWave) void flat_selection_kernel(const FlatArgs a) {
    constexpr int BLK = WAVES * kWave;
    static_assert(WAVES <= 16 && kFlatBins % BLK == 0, "flat_selection_kernel: misc slots / bins per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n_all = a.feat_cnt[f];
    const int64_t off = a.feat_off[f];
    const int64_t tb = a.tri_off[f];
    const int tn = a.tri_cnt ? a.tri_cnt[f] : (int)(a.tri_off[f + 1] - tb);
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const bool skip = DEV && a.dt_status && a.dt_status[f] != 0;
    if (n_all <= 0 || tn <= 0 || skip || (DEV && tn > a.max_tri)) {
        if (tid == 0) {
            a.status[f] = (DEV && !skip && tn > a.max_tri && n_all > 0) ? MVOSR_ST_ERR_MASK : MVOSR_ST_ERR_EMPTY;
            a.height_level[f] = nan(""); a.n_kept[f] = 0;
            if constexpr (DEV) {
                a.raw_scale[f] = nan(""); a.best_ic[f] = 0; a.used[f] = 0;
                for (int k = 0; k < 4; ++k) a.model[4 * f + k] = nan("");
            }
        }
        return;
    }
    // LDS: heights and flags of every triangle by row, the scalars, then the vertex planes.  Stage form: the planes are dead
    // once the normals are done and the histogram of the median search takes their place.  Device-resident form: the planes
    // live on for the RANSAC, the histogram has its own room, and the heights' room is reused by the point list.
    const uint32_t npad = (uint32_t)((n_all + 1) & ~1);
    double *Hh = reinterpret_cast<double *>(smem);               // every triangle's height, by row
    unsigned long long *U = reinterpret_cast<unsigned long long *>(Hh);   // (heights are >= 0: the bit patterns order like the values)
    unsigned long long *ext = U + tn;                            // [2] smallest / largest loose height (bits), [2] scratch
    int *misc = reinterpret_cast<int *>(ext + 4);                // FM_N scalars
    double *X = reinterpret_cast<double *>(misc + FM_N);
    double *Y = X + npad;
    double *Z = Y + npad;
    int *hist = DEV ? reinterpret_cast<int *>(Z + npad) : reinterpret_cast<int *>(X);   // kFlatBins bins; later the short candidate list (64-bit)
    const uint32_t planes = DEV ? 24u * npad + 4u * kFlatBins : (24u * npad > 4u * kFlatBins ? 24u * npad : 4u * kFlatBins);
    uint8_t *Fl = reinterpret_cast<uint8_t *>(X) + planes;       // every triangle's flags, by row
#ifdef MVOSR_FS_STAMPS
    unsigned long long st[10];
#define FS_STAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define FS_STAMP(i) do {} while (0)
#endif
    FS_STAMP(0);
    if (tid < FM_WSUM) misc[tid] = 0;
    if (tid == 0) { ext[0] = ~0ull; ext[1] = 0ull; ext[2] = ~0ull; }
    // a thread's next kFlatRows triangle rows are in flight while it works on the current ones (and the first ones while
    // the vertex planes stream in): a row is a dependent global load in front of nine LDS gathers and the LU chain
    TriIds rows[kFlatRows];
    auto load_rows = [&](int t0) {
#pragma unroll
        for (int j = 0; j < kFlatRows; ++j) rows[j] = load_tri(a.tri + 3 * tb, min(t0 + j * BLK, tn - 1));
    };
    load_rows(tid);
    int n = n_all;                                               // features in LDS (the survivors)
    if (DEV && a.keep) {
        // ordered compaction at load (rescale.py:134-135: feature3d[valid_id]): every wavefront owns a contiguous segment of
        // the frame, counts its survivors, and after one barrier knows where its segment starts in LDS
        const int seg = ((n_all + BLK - 1) / BLK) * kWave;
        const int s0 = wave * seg, s1 = min(n_all, s0 + seg);
        // (a lane's features — up to four: 4 096 per frame on sixteen wavefronts — with their keep flags in one batch of loads,
        // before the count: flag, wait, coordinates, wait, twice over, was a tenth of this kernel's time with nothing else on the CU)
        constexpr int kOwn = 4;
        const bool own = seg <= kOwn * kWave;
        double ox[kOwn], oy[kOwn], oz[kOwn];
        unsigned okeep = 0u;
        int c = 0;
        if (own) {
#pragma unroll
            for (int r = 0; r < kOwn; ++r) {
                const int i = s0 + r * kWave + lane;
                const int ic = min(i, n_all - 1);
                const bool k = i < s1 && a.keep[off + ic] >= 0;
                okeep |= k ? (1u << r) : 0u;
                ox[r] = a.x[off + ic]; oy[r] = a.y[off + ic]; oz[r] = a.z[off + ic];
            }
#pragma unroll
            for (int r = 0; r < kOwn; ++r) c += __popcll(__ballot((okeep >> r) & 1u));
        } else
        for (int i0 = s0; i0 < s1; i0 += kWave) {
            const int i = i0 + lane;
            c += __popcll(__ballot(i < s1 && a.keep[off + i] >= 0));
        }
        if (lane == 0) misc[FM_CW + wave] = c;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { const int cw = misc[FM_CW + w]; total += cw; if (w < wave) base += cw; }
        n = total;
        if (own) {
#pragma unroll
            for (int r = 0; r < kOwn; ++r) {
                const bool k = (okeep >> r) & 1u;
                const unsigned long long m = __ballot(k);
                if (k) {
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    X[pos] = ox[r]; Y[pos] = oy[r]; Z[pos] = oz[r];
                }
                base += __popcll(m);
            }
        } else
        for (int i0 = s0; i0 < s1; i0 += kWave) {
            const int i = i0 + lane;
            const bool k = i < s1 && a.keep[off + i] >= 0;
            const unsigned long long m = __ballot(k);
            if (k) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                X[pos] = a.x[off + i]; Y[pos] = a.y[off + i]; Z[pos] = a.z[off + i];
            }
            base += __popcll(m);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < n_all; i += BLK) { X[i] = a.x[off + i]; Y[i] = a.y[off + i]; Z[i] = a.z[off + i]; }
    }
    const double s_loose = sin(a.loose_deg * 3.141592653589793 / 180.0), s_tight = sin(a.tight_deg * 3.141592653589793 / 180.0);
    __syncthreads();
    FS_STAMP(1);
    // phase 1: every triangle's height and flags into LDS (and the height to the output), the loose ones counted and bounded
    {
        int k_mine = 0;
        unsigned long long umin = ~0ull, umax = 0ull;
        auto one_row = [&](const TriIds q, const int t) {
            if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) {
                misc[FM_BADID] = 1; Fl[t] = 0; Hh[t] = nan(""); if (!DEV || a.tri_height) a.tri_height[tb + t] = nan(""); return;
            }
            double nx, ny, nz;
            if (!plane_normal(X[q.a], Y[q.a], Z[q.a], X[q.b], Y[q.b], Z[q.b], X[q.c], Y[q.c], Z[q.c], nx, ny, nz)) misc[FM_SINGULAR] = 1;   // rescale.py:79-80
            const double len = sqrt((nx * nx + ny * ny) + nz * nz);                          // :81
            const double mu = -(ny / len);                                                   // :82
            const double h = 1.0 / len;                                                      // :89
            // pitch = asin(mu) * 180/pi (:83) is increasing in mu: away from the two thresholds the comparison is
            // made on mu itself, within 1e-12 of one (or for NaN) on the reference's own expression
            bool loose, tight;
            if (fabs(mu - s_loose) > 1e-12 && fabs(mu - s_tight) > 1e-12) { loose = mu < s_loose; tight = mu < s_tight; }
            else {
                const double pitch = asin(mu) * 180.0 / 3.141592653589793;
                loose = pitch < a.loose_deg; tight = pitch < a.tight_deg;
            }
            Hh[t] = h;
            Fl[t] = (uint8_t)((loose ? 1 : 0) | (tight ? 2 : 0));                            // :85-86
            if (!DEV || a.tri_height) a.tri_height[tb + t] = h;
            if (loose) {
                const unsigned long long u = (unsigned long long)__double_as_longlong(h);
                ++k_mine; umin = u < umin ? u : umin; umax = u > umax ? u : umax;
            }
        };
        for (int t0 = tid; t0 < tn; t0 += kFlatRows * BLK) {
            TriIds cur[kFlatRows];
#pragma unroll
            for (int j = 0; j < kFlatRows; ++j) cur[j] = rows[j];
            if (t0 + kFlatRows * BLK < tn) load_rows(t0 + kFlatRows * BLK);
#pragma unroll
            for (int j = 0; j < kFlatRows; ++j) if (t0 + j * BLK < tn) one_row(cur[j], t0 + j * BLK);
        }
        k_mine = wave_sum(k_mine);
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned long long lo = __shfl_xor(umin, o), hi = __shfl_xor(umax, o);
            umin = lo < umin ? lo : umin; umax = hi > umax ? hi : umax;
        }
        if (lane == 0 && k_mine) { atomicAdd(&misc[FM_K], k_mine); atomicMin(&ext[0], umin); atomicMax(&ext[1], umax); }
    }
    __syncthreads();
    FS_STAMP(2);
    const int k = misc[FM_K];
    double level = nan("");                                      // median of an empty set is nan (nothing passes)
    if (k > 0) {                                                                         // np.median, :91
        // The lower middle order statistic (rank klo): histogram the candidates' bit patterns over [lo, hi] with bins of
        // 2^shift patterns (<= 2048 bins), find the bin that holds the rank, narrow [lo, hi] to it, repeat — until a bin is a
        // single pattern, or holds few enough candidates for wavefront 0 to rank them directly.  Real frames take one
        // pass: their heights spread over a few thousand distinct patterns per bin at most.
        const int klo = (k - 1) >> 1, khi = k >> 1;
        unsigned long long lo = ext[0], hi = ext[1];
        int rank = klo;
        unsigned long long vlo_bits = lo;
        for (;;) {
            const unsigned long long range = hi - lo;
            if (range == 0ull) { vlo_bits = lo; break; }
            const int bits = 64 - __clzll((long long)range);                             // range < 2^bits
            const int shift = bits > 11 ? bits - 11 : 0;
            for (int b = tid; b < kFlatBins; b += BLK) hist[b] = 0;
            __syncthreads();
            #pragma unroll 4
            for (int t = tid; t < tn; t += BLK) {
                if (!(Fl[t] & 1)) continue;
                const unsigned long long u = U[t];
                if (u >= lo && u <= hi) atomicAdd(&hist[(int)((u - lo) >> shift)], 1);
            }
            __syncthreads();
            {   // the bin whose cumulative count passes the rank: four bins per thread, wave scan, wave totals through LDS
                constexpr int BPT = kFlatBins / BLK;                // bins per thread (4 with 512 threads, 2 with 1024)
                const int b0 = hist[BPT * tid], b1 = BPT > 1 ? hist[BPT * tid + 1] : 0, b2 = BPT > 2 ? hist[BPT * tid + 2] : 0, b3 = BPT > 3 ? hist[BPT * tid + 3] : 0;
                const int mine = (b0 + b1) + (b2 + b3);
                int incl = mine;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
                if (lane == kWave - 1) misc[FM_WSUM + wave] = incl;
                __syncthreads();
                int before = 0;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) if (w < wave) before += misc[FM_WSUM + w];
                incl += before;
                const int excl = incl - mine;
                if (rank >= excl && rank < incl) {               // exactly one thread
                    int r = rank - excl, bin = BPT * tid, c = b0;
                    if (r >= b0) { r -= b0; ++bin; c = b1; if (r >= b1) { r -= b1; ++bin; c = b2; if (r >= b2) { r -= b2; ++bin; c = b3; } } }
                    misc[FM_BIN] = bin; misc[FM_RANK] = r; misc[FM_BINCNT] = c;
                }
                __syncthreads();
            }
            const int bin = misc[FM_BIN], c = misc[FM_BINCNT];
            rank = misc[FM_RANK];
            lo += (unsigned long long)bin << shift;
            { const unsigned long long top = lo + ((1ull << shift) - 1ull); hi = top < hi ? top : hi; }
            if (shift == 0) { vlo_bits = lo; break; }            // the bin is one pattern
            if (c <= kFlatDirect) {
                unsigned long long *list = reinterpret_cast<unsigned long long *>(hist);
                __syncthreads();                                 // every thread has read misc / hist
                #pragma unroll 4
                for (int t = tid; t < tn; t += BLK) {
                    if (!(Fl[t] & 1)) continue;
                    const unsigned long long u = U[t];
                    if (u >= lo && u <= hi) list[atomicAdd(&misc[FM_LIST], 1)] = u;
                }
                __syncthreads();
                if (wave == 0) {
                    const unsigned long long mine = lane < c ? list[lane] : ~0ull;
                    int below = 0;
                    for (int j = 0; j < c; ++j) { const unsigned long long v = list[j]; below += (v < mine || (v == mine && j < lane)) ? 1 : 0; }
                    if (lane < c && below == rank) ext[3] = mine;
                }
                __syncthreads();
                vlo_bits = ext[3];
                break;
            }
            __syncthreads();                                     // hist is zeroed again at the top
        }
        const double vlo = __longlong_as_double((long long)vlo_bits);
        double vhi = vlo;
        if (khi != klo) {
            // the next order statistic: vlo again if it occurs often enough, else the smallest value above it
            int le = 0;
            unsigned long long above = ~0ull;
            #pragma unroll 4
            for (int t = tid; t < tn; t += BLK) {
                if (!(Fl[t] & 1)) continue;
                const unsigned long long u = U[t];
                if (u <= vlo_bits) ++le; else above = u < above ? u : above;
            }
            le = wave_sum(le);
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) { const unsigned long long other = __shfl_xor(above, o); above = other < above ? other : above; }
            if (lane == 0) { atomicAdd(&misc[FM_LE], le); atomicMin(&ext[2], above); }
            __syncthreads();
            if (misc[FM_LE] < khi + 1) vhi = __longlong_as_double((long long)ext[2]);
        }
        level = a.height_factor * ((klo == khi) ? vlo : (vlo + vhi) / 2.0);
    }
    FS_STAMP(3);
    if constexpr (!DEV) {
        int kept = 0;
        #pragma unroll 4
        for (int t = tid; t < tn; t += BLK) {
            uint8_t fl = Fl[t];
            if ((fl & 2) && Hh[t] > level) { fl |= 4; ++kept; }                                 // :94-96
            a.tri_flags[tb + t] = fl;
        }
        kept = wave_sum(kept);
        if (lane == 0 && kept) atomicAdd(&misc[FM_KEPT], kept);
        __syncthreads();
        if (tid == 0) {
            a.height_level[f] = level;
            a.n_kept[f] = misc[FM_KEPT];
            a.status[f] = misc[FM_BADID] ? MVOSR_ST_ERR_MASK : (misc[FM_SINGULAR] ? MVOSR_ST_ERR_SINGULAR : 0);
        }
#ifdef MVOSR_FS_STAMPS
        FS_STAMP(4);                                                 // (diagnostic build: the stamps overwrite the frame's first heights)
        if (tid == 0 && tn >= 5) for (int i = 0; i < 5; ++i) a.tri_height[tb + i] = (double)(st[i] - st[0]);
#endif
    } else {
        // ---- the kept rows (:94-96), wavefront by wavefront over contiguous row segments so that the point list —
        // triangle_ids[valid_id].reshape(-1), :101 — comes out in row order without a sort
        const int seg = ((tn + BLK - 1) / BLK) * kWave;
        const int s0 = wave * seg, s1 = min(tn, s0 + seg);
        int c = 0;
        for (int t0 = s0; t0 < s1; t0 += kWave) {
            const int t = t0 + lane;
            bool kp = false;
            if (t < s1) {
                uint8_t fl = Fl[t];
                kp = (fl & 2) && Hh[t] > level;
                if (kp) { fl |= 4; Fl[t] = fl; }
                if (a.tri_flags) a.tri_flags[tb + t] = fl;
            }
            c += __popcll(__ballot(kp));
        }
        if (lane == 0) misc[FM_CW + wave] = c;
        __syncthreads();                                         // (every height has been compared: their room is the list's now)
        int base = 0, K = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { const int cw = misc[FM_CW + w]; K += cw; if (w < wave) base += cw; }
        const int M = 3 * K;                                     // len(point_selected), :140
        uint16_t *L = reinterpret_cast<uint16_t *>(Hh);          // the point list as vertex ids
        const int H = a.n_hyp;
        double2 *mods = reinterpret_cast<double2 *>(smem + (((uint32_t)(Fl - reinterpret_cast<uint8_t *>(smem)) + (uint32_t)tn + 15u) & ~15u));   // [H][2] unit (n, d) as (nx, ny), (nz, d)
        int *cnts = reinterpret_cast<int *>(mods + 2 * H);       // [H] inlier counts
        const bool bad = misc[FM_BADID] || misc[FM_SINGULAR];
        const bool fit = !bad && M >= a.ransac_min_points;       // :152
        if (fit) {
            for (int t0 = s0; t0 < s1; t0 += kWave) {
                const int t = t0 + lane;
                const bool kp = t < s1 && (Fl[t] & 4);
                const unsigned long long m = __ballot(kp);
                if (kp) {
                    const TriIds q = load_tri(a.tri + 3 * tb, t);
                    const int pos = 3 * (base + __popcll(m & ((1ull << lane) - 1ull)));
                    L[pos] = (uint16_t)q.a; L[pos + 1] = (uint16_t)q.b; L[pos + 2] = (uint16_t)q.c;
                }
                base += __popcll(m);
            }
            __syncthreads();
            FS_STAMP(4);
            // the hypotheses' planes, one thread each (ransac.py:10-11, estimate_road_norm.py:13-15)
            const uint64_t fc = (uint64_t)(a.frame_ids ? a.frame_ids[f] : a.frame_base + f);
            const uint64_t key = rs_mix64(a.seed ^ (fc * 0xD1B54A32D192ED03ull));
            for (int h = tid; h < H; h += BLK) {
                int v0, v1, v2;
                if (a.id_triples) {
                    const int32_t *t = a.id_triples + ((int64_t)f * H + h) * 3;
                    v0 = min(max(t[0], 0), n - 1); v1 = min(max(t[1], 0), n - 1); v2 = min(max(t[2], 0), n - 1);
                } else {
                    rs_draw3(key, h, M, L, v0, v1, v2);
                }
                const double x0 = X[v0], y0 = Y[v0], z0 = Z[v0];
                const double e1x = X[v1] - x0, e1y = Y[v1] - y0, e1z = Z[v1] - z0;
                const double e2x = X[v2] - x0, e2y = Y[v2] - y0, e2z = Z[v2] - z0;
                const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
                const double d = -((nx * x0 + ny * y0) + nz * z0);
                const double inv = 1.0 / sqrt(((nx * nx + ny * ny) + nz * nz) + d * d);
                double2 m0, m1; m0.x = nx * inv; m0.y = ny * inv; m1.x = nz * inv; m1.y = d * inv;
                mods[2 * h] = m0; mods[2 * h + 1] = m1;
                cnts[h] = 0;
            }
            __syncthreads();
            // Inlier counts (estimate_road_norm.py:17-18 over every list entry, repeats included).  The list names each of its
            // ~600 distinct vertices five or six times (once per kept triangle): a vertex is tested ONCE per hypothesis and
            // counts with its multiplicity.  Multiplicities by LDS atomics into 16-bit halves (the histogram's room is free by
            // now), the distinct vertices as (id | multiplicity << 16) behind the list, then one wavefront per hypothesis: ten
            // trips over the vertices, an integer sum per lane, one reduction.  (3 300 list entries against 100 hypotheses
            // with ballots and popcounts — the points in registers — was two thirds of this kernel's instructions; the first
            // attempt at the multiplicities kept the ballots, bit-sliced, and was slower than the list.)
            FS_STAMP(5);
            uint32_t *W2 = reinterpret_cast<uint32_t *>(hist);                   // [n / 2 + 1] two 16-bit counts per word
            uint16_t *Dv = L + ((M + 1) & ~1);                                   // the distinct vertices, behind the list
            // (the list and the vertices share the heights' 8 bytes per row: 6 per KEPT row + 2 per vertex — they fit unless
            // nearly every row is kept in a frame of a few points; then the list itself is counted, entry by entry)
            const bool dedup = 2 * ((M + 1) & ~1) + 2 * n <= 8 * tn && n <= 2 * kFlatBins - 2;
            int n_items = M;
            if (dedup) {
                for (int v = tid; v < (n + 1) / 2 + 1; v += BLK) W2[v] = 0u;
                if (tid == 0) misc[FM_ND] = 0;
                __syncthreads();
                for (int j = tid; j < M; j += BLK) { const uint32_t id = L[j]; atomicAdd(&W2[id >> 1], 1u << (16u * (id & 1u))); }
                __syncthreads();
                for (int v = tid; v < n; v += BLK)
                    if ((W2[v >> 1] >> (16u * ((uint32_t)v & 1u))) & 0xFFFFu) Dv[atomicAdd(&misc[FM_ND], 1)] = (uint16_t)v;
                __syncthreads();
                n_items = misc[FM_ND];
            }
            // ... and, where they fit the heights' room (28 bytes per distinct vertex: they do unless more than ~1 000 of a frame's
            // vertices lie on kept rows), the vertices' coordinates and multiplicities side by side, in the order of the list of
            // distinct vertices: the counting loop then reads four contiguous arrays instead of gathering by id
            const bool packed = dedup && n_items <= 2 * BLK && 28 * n_items + 8 <= 8 * tn;
            double *PX = Hh, *PY = PX + n_items, *PZ = PY + n_items;
            int *PW = reinterpret_cast<int *>(PZ + n_items);
            if (packed) {
                double gx[2], gy[2], gz[2];
                int gw[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int j = tid + r * BLK;
                    const uint32_t id = Dv[min(j, n_items - 1)];
                    gx[r] = X[id]; gy[r] = Y[id]; gz[r] = Z[id]; gw[r] = (int)((W2[id >> 1] >> (16u * (id & 1u))) & 0xFFFFu);
                }
                __syncthreads();                                                 // (the list and the ids have been read: their room is the arrays')
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int j = tid + r * BLK;
                    if (j < n_items) { PX[j] = gx[r]; PY[j] = gy[r]; PZ[j] = gz[r]; PW[j] = gw[r]; }
                }
                __syncthreads();
            }
            FS_STAMP(6);
            const uint16_t *items = dedup ? Dv : L;
            // a wavefront's hypotheses seven at a time (all of them, with 100 hypotheses on 16 wavefronts), their planes in registers: a vertex is gathered (id -> multiplicity, x, y,
            // z: four LDS gathers with bank conflicts) once per pass and tested against all four (one hypothesis per pass spent
            // 41 % of the kernel's time on those gathers)
            constexpr int kHypPass = 7;
            for (int k0 = 0; wave + WAVES * k0 < H; k0 += kHypPass) {
                double2 ma[kHypPass], mb[kHypPass];
                int acc[kHypPass];
#pragma unroll
                for (int q = 0; q < kHypPass; ++q) {
                    const int h = min(wave + WAVES * (k0 + q), H - 1);
                    ma[q] = mods[2 * h]; mb[q] = mods[2 * h + 1]; acc[q] = 0;
                }
                if (packed) {
                    for (int j = lane; j < n_items; j += kWave) {
                        const double px = PX[j], py = PY[j], pz = PZ[j];
                        const int wgt = PW[j];
#pragma unroll
                        for (int q = 0; q < kHypPass; ++q)
                            acc[q] += (fabs(((px * ma[q].x + py * ma[q].y) + pz * mb[q].x) + mb[q].y) < a.threshold) ? wgt : 0;   // estimate_road_norm.py:18
                    }
                } else {
                    for (int j = lane; j < n_items; j += kWave) {
                        const uint32_t id = items[j];
                        const int wgt = dedup ? (int)((W2[id >> 1] >> (16u * (id & 1u))) & 0xFFFFu) : 1;
                        const double px = X[id], py = Y[id], pz = Z[id];
#pragma unroll
                        for (int q = 0; q < kHypPass; ++q)
                            acc[q] += (fabs(((px * ma[q].x + py * ma[q].y) + pz * mb[q].x) + mb[q].y) < a.threshold) ? wgt : 0;
                    }
                }
#pragma unroll
                for (int q = 0; q < kHypPass; ++q) {
                    const int h = wave + WAVES * (k0 + q);
                    const int sum = wave_sum(acc[q]);
                    if (lane == 0 && h < H) cnts[h] = sum;
                }
            }
            __syncthreads();
            FS_STAMP(7);
            if (a.hyp_counts) for (int h = tid; h < H; h += BLK) a.hyp_counts[(int64_t)f * H + h] = cnts[h];
        }
        if (wave == 0) {
            int status = misc[FM_BADID] ? MVOSR_ST_ERR_MASK : (misc[FM_SINGULAR] ? MVOSR_ST_ERR_SINGULAR : (fit ? 0 : MVOSR_ST_RS_FEW));
            int best = -1, best_ic = 0, used = 0;
            double m[4] = {nan(""), nan(""), nan(""), nan("")};
            double raw = nan("");
            if (fit) {
                // ransac.py:9-22 — a hypothesis is the new best when it counts MORE than the best so far, and the loop stops
                // at a new best above the goal — by the wavefront, 64 hypotheses at a time: the loop stops at the first count
                // above the goal (the best before it was not, so it is a new best), and the best is the first occurrence of
                // the largest count up to there.  (One thread walking the hundred counts was 16 % of the kernel's time.)
                const double goal = (double)M * a.goal_fraction;                  // estimate_road_norm.py:68
                used = H;
                for (int h0 = 0; h0 < H; h0 += kWave) {
                    const int h = h0 + lane;
                    const int c = h < H ? cnts[h] : -1;
                    const unsigned long long over = __ballot(h < H && (double)c > goal);
                    const int limit = over ? (int)__ffsll((long long)over) - 1 : kWave - 1;
                    const bool in = h < H && lane <= limit;
                    const int mx = wave_max(in ? c : -1);
                    if (mx > best_ic) {
                        const unsigned long long who = __ballot(in && c == mx);
                        best = h0 + (int)__ffsll((long long)who) - 1; best_ic = mx;
                    }
                    if (over) { used = h0 + limit + 1; break; }
                }
                if (best >= 0) {
                    const double2 b0 = mods[2 * best], b1 = mods[2 * best + 1];
                    const double sgn = (b0.y < 0.0) ? -1.0 : 1.0;                 // rescale.py:159-161
                    m[0] = sgn * b0.x; m[1] = sgn * b0.y; m[2] = sgn * b1.x; m[3] = sgn * b1.y;
                    const double h_bar = -m[3];                                    // :158
                    const double norm_norm = sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]) / h_bar;   // :162-163
                    const double cam_h = 1.0 / norm_norm;                          // :165
                    raw = a.absolute_reference / cam_h;                            // :167
                } else status = MVOSR_ST_RS_FEW;                                   // (no hypothesis with an inlier: NaN planes only)
            }
            if (lane == 0) {
                a.height_level[f] = level;
                a.n_kept[f] = K;
                a.status[f] = status;
                a.raw_scale[f] = raw;
                a.best_ic[f] = best_ic; a.used[f] = used;
                for (int kk = 0; kk < 4; ++kk) a.model[4 * f + kk] = m[kk];
            }
#ifdef MVOSR_FS_STAMPS
            // diagnostic build: the results are overwritten by the phase durations (profiles/stamps_flat_dev.py)
            FS_STAMP(8);
            if (fit && lane == 0) {
                a.model[4 * f] = (double)(st[1] - st[0]); a.model[4 * f + 1] = (double)(st[2] - st[1]); a.model[4 * f + 2] = (double)(st[3] - st[2]);
                a.model[4 * f + 3] = (double)(st[4] - st[3]); a.raw_scale[f] = (double)(st[5] - st[4]); a.height_level[f] = (double)(st[6] - st[5]);
                a.best_ic[f] = (int)(st[7] - st[6]); a.used[f] = (int)(st[8] - st[7]);
            }
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// RANSAC plane fit with the sample triples given (ransac.py:3-23).  One workgroup per frame; every
// hypothesis is the plane through its three sample points as the unit 4-vector (n, d)/|(n, d)| — the
// null vector the reference gets from the SVD of [x y z 1] (estimate_road_norm.py:13-15), up to sign
// — and |m.[p,1]| < threshold is counted over the frame's points (estimate_road_norm.py:17-18), the
// points held in registers, the hypotheses streamed from LDS.  One lane then replays the reference's sequential
// rule: a hypothesis replaces the best when its count is strictly larger, and the loop stops at
// the first such improvement that exceeds the goal.
// ---------------------------------------------------------------------------------------------
struct RansacArgs {
    int64_t n_frames;
    const int64_t *pts_off; const int32_t *pts_cnt;     // [F]
    const double *px, *py, *pz;                          // planes of the selected points (with repeats, :101)
    const int32_t *triples;                              // [F][H][3] row indices into the frame's points
    int32_t n_hyp;                                       // H (<= kMaxHyp)
    double threshold, goal_fraction;
    int32_t *counts;                                     // [F][H] inlier counts (or NULL)
    double *model;                                       // [F][4] best model, sign fixed so that n_y >= 0
    int32_t *best_ic, *used;                             // [F]
    int32_t line;                                        // 1: 2-D line fit (estimate_line / is_inlier_line, estimate_road_norm.py:39-49):
                                                         // samples are PAIRS (still 3 ints apart), pz is not read, model = (a, b, 0, c)
};
__global__ __launch_bounds__(kRsBlock) void ransac_plane_kernel(const RansacArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int M = a.pts_cnt[f];
    const int64_t off = a.pts_off[f];
    const int H = a.n_hyp;
    const int tid = threadIdx.x, lane = lane_id();
    double4 *mods = reinterpret_cast<double4 *>(smem);                    // [H] unit (n, d)
    int *cnts = reinterpret_cast<int *>(mods + H);                        // [H] inlier counts
    if (M <= 0) {
        if (tid == 0) { a.best_ic[f] = 0; a.used[f] = 0; for (int k = 0; k < 4; ++k) a.model[4 * f + k] = nan(""); }
        return;
    }
    const double *px = a.px + off, *py = a.py + off, *pz = a.pz + off;
    // the hypotheses' planes, one thread each
    for (int h = tid; h < H; h += kRsBlock) {
        const int32_t *t = a.triples + ((int64_t)f * H + h) * 3;
        const int i0 = t[0], i1 = t[1], i2 = t[2];
        double nx, ny, nz, d;
        if (a.line) {
            // the line a x + b y + c = 0 through two points: the null vector of [x y 1] (estimate_road_norm.py:44-46)
            const double x0 = px[i0], y0 = py[i0];
            nx = py[i1] - y0; ny = -(px[i1] - x0); nz = 0.0;
            d = -(nx * x0 + ny * y0);
        } else {
            const double x0 = px[i0], y0 = py[i0], z0 = pz[i0];
            const double e1x = px[i1] - x0, e1y = py[i1] - y0, e1z = pz[i1] - z0;
            const double e2x = px[i2] - x0, e2y = py[i2] - y0, e2z = pz[i2] - z0;
            nx = e1y * e2z - e1z * e2y; ny = e1z * e2x - e1x * e2z; nz = e1x * e2y - e1y * e2x;
            d = -((nx * x0 + ny * y0) + nz * z0);
        }
        const double inv = 1.0 / sqrt(((nx * nx + ny * ny) + nz * nz) + d * d);
        double4 m; m.x = nx * inv; m.y = ny * inv; m.z = nz * inv; m.w = d * inv;
        mods[h] = m;
        cnts[h] = 0;
    }
    __syncthreads();
    // Every thread keeps up to kRansacPPT points in registers (read from HBM once, coalesced) and the
    // hypotheses stream past them from LDS (wave-uniform reads); a hypothesis' inliers among a
    // wavefront's points are counted on the scalar unit (ballot + popcount), one LDS add per wave.
    for (int c0 = 0; c0 < M; c0 += kRsBlock * kRansacPPT) {
        double qx[kRansacPPT], qy[kRansacPPT], qz[kRansacPPT];
#pragma unroll
        for (int k = 0; k < kRansacPPT; ++k) {
            const int j = c0 + k * kRsBlock + tid;
            const int jc = min(j, M - 1);
            qx[k] = px[jc]; qy[k] = py[jc]; qz[k] = a.line ? 0.0 : pz[jc];
            if (j >= M) qx[k] = nan("");                         // never an inlier: no masks or branches in the loop below
        }
        const int rows = min(kRansacPPT, (M - c0 + kRsBlock - 1) / kRsBlock);      // workgroup-uniform: rows that hold any point
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const double4 m = mods[h];
            int ic = 0;
#pragma unroll
            for (int k = 0; k < kRansacPPT; ++k)
                if (k < rows) ic += __popcll(__ballot(fabs(((qx[k] * m.x + qy[k] * m.y) + qz[k] * m.z) + m.w) < a.threshold));   // estimate_road_norm.py:18
            if (lane == 0 && ic) atomicAdd(&cnts[h], ic);
        }
    }
    __syncthreads();
    if (a.counts) for (int h = tid; h < H; h += kRsBlock) a.counts[(int64_t)f * H + h] = cnts[h];
    if (tid == 0) {
        const double goal = (double)M * a.goal_fraction;                  // estimate_road_norm.py:68
        int best = -1, best_ic = 0, used = 0;
        for (int h = 0; h < H; ++h) {                                     // ransac.py:9-22
            used = h + 1;
            if (cnts[h] > best_ic) {
                best_ic = cnts[h]; best = h;
                if ((double)best_ic > goal) break;
            }
        }
        a.best_ic[f] = best_ic; a.used[f] = used;
        double m[4] = {nan(""), nan(""), nan(""), nan("")};
        if (best >= 0) {
            const double4 bm = mods[best];
            const double sgn = (bm.y < 0.0) ? -1.0 : 1.0;                 // rescale.py:159-161
            m[0] = sgn * bm.x; m[1] = sgn * bm.y; m[2] = sgn * bm.z; m[3] = sgn * bm.w;
        }
        for (int k = 0; k < 4; ++k) a.model[4 * f + k] = m[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Legacy per-triangle batch, /root/reference/src/triangle_batch.py:14-68: features
// are [u, v, depth]; every Delaunay triangle is back-projected (:32-33), n = A^-1 . 1 (:36-37),
// s = n_y/|n| (:43), h = mean y (:40); triangles with s > 0.98 and h > 0 (:54-55) give mean and std
// (:57-58), values outside mean +- 3 std are clipped (:60-61) and the mean of the rest is the camera
// height (:62).  Three sweeps over the triangles (sum / squared deviations / clipped sum), each
// thread visiting its triangles in a fixed order, so the result is run-to-run identical.
// ---------------------------------------------------------------------------------------------
struct TriBatchArgs {
    int64_t n_frames;
    const int64_t *feat_off; const int32_t *feat_cnt;
    const double *u, *v, *depth;
    const int64_t *tri_off; const int32_t *tri;
    double focus, cx, cy, s_min, n_sigma;
    double *height;                // [F]
    int32_t *counts;               // [F][2]: kept, kept after the clip
    int32_t *status;               // [F]
};

__global__ __launch_bounds__(kRsBlock) void triangle_batch_kernel(const TriBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n = a.feat_cnt[f];
    const int64_t off = a.feat_off[f];
    const int64_t tb = a.tri_off[f];
    const int tn = (int)(a.tri_off[f + 1] - tb);
    const int tid = threadIdx.x;
    if (n <= 0 || tn <= 0) {
        if (tid == 0) { a.status[f] = MVOSR_ST_ERR_EMPTY; a.height[f] = nan(""); a.counts[2 * f] = a.counts[2 * f + 1] = 0; }
        return;
    }
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    double *X = reinterpret_cast<double *>(smem);
    double *Y = X + npad;
    double *Z = Y + npad;
    double *red = Z + npad;                                     // 3 slots x 2*kRsWaves doubles
    int *flag = reinterpret_cast<int *>(red + 3 * 2 * kRsWaves);
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    for (int i = tid; i < n; i += kRsBlock) {
        const double d = a.depth[off + i];
        X[i] = d * (a.u[off + i] - a.cx) / a.focus;                                  // :32
        Y[i] = d * (a.v[off + i] - a.cy) / a.focus;                                  // :33
        Z[i] = d;
    }
    __syncthreads();
    // height of triangle t if it is kept, NaN otherwise
    auto kept_height = [&](int t) -> double {
        const TriIds q = load_tri(a.tri + 3 * tb, t);
        if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) { flag[1] = 1; return nan(""); }
        double nx, ny, nz;
        if (!plane_normal(X[q.a], Y[q.a], Z[q.a], X[q.b], Y[q.b], Z[q.b], X[q.c], Y[q.c], Z[q.c], nx, ny, nz)) flag[0] = 1;   // :36
        const double s = ny / sqrt((nx * nx + ny * ny) + nz * nz);                   // :38-39,:43
        const double h = div3((Y[q.a] + Y[q.b]) + Y[q.c]);                           // :40
        return (s > a.s_min && h > 0.0) ? h : nan("");                               // :54-55
    };
    // the three sweeps need every kept height three times: a thread's first kTbKeep triangles keep theirs
    // in registers (all of them for frames of up to 4096 triangles), the rest are recomputed
    constexpr int kTbKeep = 8;
    double hk[kTbKeep];
    double sum = 0.0, cnt = 0.0;
#pragma unroll
    for (int k = 0; k < kTbKeep; ++k) {
        const int t = tid + k * kRsBlock;
        hk[k] = (t < tn) ? kept_height(t) : nan("");
        if (hk[k] == hk[k]) { sum += hk[k]; cnt += 1.0; }
    }
    for (int t = tid + kTbKeep * kRsBlock; t < tn; t += kRsBlock) { const double h = kept_height(t); if (h == h) { sum += h; cnt += 1.0; } }
    block_sum2<kRsWaves>(sum, cnt, red);
    const double mean = sum / cnt;                                                   // :57
    double ss = 0.0, dummy = 0.0;
#pragma unroll
    for (int k = 0; k < kTbKeep; ++k) if (hk[k] == hk[k]) { const double d = hk[k] - mean; ss += d * d; }
    for (int t = tid + kTbKeep * kRsBlock; t < tn; t += kRsBlock) { const double h = kept_height(t); if (h == h) { const double d = h - mean; ss += d * d; } }
    block_sum2<kRsWaves>(ss, dummy, red + 2 * kRsWaves);
    const double sd = sqrt(ss / cnt);                                                // :58
    const double lo = mean - a.n_sigma * sd, hi = mean + a.n_sigma * sd;             // :60-61
    double sum2 = 0.0, cnt2 = 0.0;
#pragma unroll
    for (int k = 0; k < kTbKeep; ++k) if (hk[k] == hk[k] && hk[k] > lo && hk[k] < hi) { sum2 += hk[k]; cnt2 += 1.0; }
    for (int t = tid + kTbKeep * kRsBlock; t < tn; t += kRsBlock) { const double h = kept_height(t); if (h == h && h > lo && h < hi) { sum2 += h; cnt2 += 1.0; } }
    block_sum2<kRsWaves>(sum2, cnt2, red + 4 * kRsWaves);
    __syncthreads();
    if (tid == 0) {
        a.height[f] = sum2 / cnt2;                                                   // :62
        a.counts[2 * f] = (int)cnt; a.counts[2 * f + 1] = (int)cnt2;
        a.status[f] = flag[1] ? MVOSR_ST_ERR_MASK : (flag[0] ? MVOSR_ST_ERR_SINGULAR : 0);
    }
}

// The slew limiter of scale_calculation_ransac (/root/reference/src/rescale.py:169-177) over a run of frames: a sequential
// recurrence of rounded additions, walked by ONE wavefront — 64 frames are loaded at a time, their values broadcast lane by
// lane (v_readlane), the running scale identical in every lane; lane j keeps the value pushed at its frame.
template <int J>
__device__ __forceinline__ void slew_step(double r, unsigned long long am, double slew, int lane, double &s, double &out) {
    if constexpr (J < kWave) {
        // branch-free: the three candidates depend on s alone, the selects on one subtraction (the recurrence's only chain)
        const double rj = readlane_d(r, J);                                  // (constant lane: no M0 set-up)
        const double d = rj - s;
        const double up = s + slew, dn = s - slew;
        const double moved = d > slew ? up : (d < -slew ? dn : rj);          // rescale.py:169-174
        s = ((am >> J) & 1ull) ? moved : s;                                  // :152 — only a frame with a RANSAC plane moves the scale
        out = lane == J ? s : out;                                           // :175 scale_queue.append(self.scale)
        slew_step<J + 1>(r, am, slew, lane, s, out);
    }
}

__global__ __launch_bounds__(kWave) void slew_kernel(const double *raw, const int32_t *apply, int64_t n, double slew, double s_in,
                                                     double *pushed) {
    const int lane = lane_id();
    double s = s_in;
    double r = lane < n ? raw[lane] : 0.0;
    int ap = lane < n ? apply[lane] : 0;
    for (int64_t i0 = 0; i0 < n; i0 += kWave) {
        const int64_t i = i0 + lane;
        const double r_cur = r;
        const unsigned long long am = __ballot(ap != 0);                     // (lanes beyond n: no plane, the scale stays)
        if (i + kWave < n) { r = raw[i + kWave]; ap = apply[i + kWave]; } else ap = 0;   // the next 64 frames load under this block's chain
        double out = 0.0;
        slew_step<0>(r_cur, am, slew, lane, s, out);
        if (i < n) pushed[i] = out;
    }
}

// get_inliers, /root/reference/src/estimate_road_norm.py:71-78: |n.p + d| < threshold per point
__global__ __launch_bounds__(256) void plane_inliers_kernel(int64_t n, const double *px, const double *py, const double *pz,
                                                            double m0, double m1, double m2, double m3, double threshold,
                                                            uint8_t *mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = fabs(((px[i] * m0 + py[i] * m1) + pz[i] * m2) + m3) < threshold;
}

static int g_rs_max_lds = 160 * 1024;

template <typename K>
static int rs_prepare(K kernel, size_t lds) {
    if ((int64_t)lds > (int64_t)g_rs_max_lds) return set_error(MVOSR_ERR_TOO_LARGE, "frame needs %zu B of LDS (> %d) in the rescale-variant kernels", lds, g_rs_max_lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_hip_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return MVOSR_OK;
}

}  // namespace mvosr

using namespace mvosr;

extern "C" {

static int launch_graph(mvosr_ctx *ctx, const mvosr_batch *b, GraphArgs &a, bool keep) {
    if (!b->feat_off || !b->feat_cnt || !b->z || !b->v || !b->tri1_off || !b->tri1) return set_error(MVOSR_ERR_ARG, "graph_inliers: missing z/v/tri1");
    if (b->n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    g_rs_max_lds = ctx->max_lds_per_block;
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.z = b->z; a.v = b->v;
    a.tri_off = b->tri1_off; a.tri = b->tri1; a.tri_cnt = b->tri1_cnt;
    const size_t lds = 16u * (size_t)((b->max_feat + 1) & ~1) + 4u * ((size_t)b->max_feat + 4) + 16;
    if (keep) {
        if ((rc = rs_prepare(graph_inliers_kernel<true>, lds))) return rc;
        hipLaunchKernelGGL(graph_inliers_kernel<true>, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    } else {
        if ((rc = rs_prepare(graph_inliers_kernel<false>, lds))) return rc;
        hipLaunchKernelGGL(graph_inliers_kernel<false>, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    }
    return check_launch("graph_inliers_kernel");
}

int mvosr_graph_inliers_batch(mvosr_ctx *ctx, const mvosr_batch *b, uint32_t good_bits, int32_t *total, int32_t *good,
                              int32_t *status) {
    if (!ctx || !b || !total || !good) return set_error(MVOSR_ERR_ARG, "graph_inliers: null argument");
    GraphArgs a = {};
    a.good_bits = good_bits; a.total = total; a.good = good; a.status = status;
    return launch_graph(ctx, b, a, false);
}

int mvosr_graph_keep_batch(mvosr_ctx *ctx, const mvosr_batch *b, uint32_t good_bits, int32_t min_valid, const int32_t *dt_status,
                           int32_t *keep, int32_t *n_valid, int32_t *status) {
    if (!ctx || !b || !keep) return set_error(MVOSR_ERR_ARG, "graph_keep: null argument");
    GraphArgs a = {};
    a.good_bits = good_bits; a.status = status; a.dt_status = dt_status; a.min_valid = min_valid; a.keep = keep; a.n_valid = n_valid;
    return launch_graph(ctx, b, a, true);
}

int mvosr_flat_selection_batch(mvosr_ctx *ctx, const mvosr_batch *b, double loose_deg, double tight_deg, double height_factor,
                               double *tri_height, uint8_t *tri_flags, double *height_level, int32_t *n_kept, int32_t *status,
                               int64_t max_tri) {
    if (!ctx || !b || !tri_height || !tri_flags || !height_level || !n_kept || !status) return set_error(MVOSR_ERR_ARG, "flat_selection: null argument");
    if (!b->feat_off || !b->feat_cnt || !b->x || !b->y || !b->z || !b->tri2_off || !b->tri2) return set_error(MVOSR_ERR_ARG, "flat_selection: missing x/y/z/tri2");
    if (b->n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    g_rs_max_lds = ctx->max_lds_per_block;
    FlatArgs a = {};
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.x = b->x; a.y = b->y; a.z = b->z;
    a.tri_off = b->tri2_off; a.tri = b->tri2; a.tri_cnt = b->tri2_cnt; a.loose_deg = loose_deg; a.tight_deg = tight_deg; a.height_factor = height_factor;
    a.tri_height = tri_height; a.tri_flags = tri_flags; a.height_level = height_level; a.status = status; a.n_kept = n_kept;
    if (max_tri <= 0) max_tri = 2 * (int64_t)b->max_feat;
    size_t lds = 24u * (size_t)((b->max_feat + 1) & ~1);
    if (lds < 4u * 2048) lds = 4u * 2048;                        // (the histogram of the median search reuses the vertex planes)
    lds += 9u * (size_t)max_tri + 32 + 4u * 48 + 16;
    if ((rc = rs_prepare(flat_selection_kernel<false>, lds))) return rc;
    hipLaunchKernelGGL(flat_selection_kernel<false>, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    return check_launch("flat_selection_kernel");
}

int mvosr_flat_ransac_batch(mvosr_ctx *ctx, const mvosr_batch *b, const int32_t *keep, const mvosr_rescale_params *rp,
                            const int32_t *id_triples, const int64_t *frame_ids, const int32_t *dt_status,
                            const mvosr_rescale_outputs *o, int64_t max_tri) {
    if (!ctx || !b || !rp || !o) return set_error(MVOSR_ERR_ARG, "flat_ransac: null argument");
    if (!o->raw_scale || !o->height_level || !o->model || !o->best_ic || !o->used || !o->n_kept || !o->status)
        return set_error(MVOSR_ERR_ARG, "flat_ransac: a required output is null");
    if (!b->feat_off || !b->feat_cnt || !b->x || !b->y || !b->z || !b->tri2_off || !b->tri2) return set_error(MVOSR_ERR_ARG, "flat_ransac: missing x/y/z/tri2");
    if (rp->n_hyp < 1 || rp->n_hyp > kMaxHyp) return set_error(MVOSR_ERR_ARG, "flat_ransac: n_hyp must be in 1..%d", kMaxHyp);
    if (rp->ransac_min_points < 3) return set_error(MVOSR_ERR_ARG, "flat_ransac: ransac_min_points < 3");
    if (b->max_feat > 65535) return set_error(MVOSR_ERR_TOO_LARGE, "flat_ransac: vertex ids are 16-bit in the point list");
    if (b->n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    g_rs_max_lds = ctx->max_lds_per_block;
    if (max_tri <= 0) max_tri = 2 * (int64_t)b->max_feat;
    FlatArgs a = {};
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.x = b->x; a.y = b->y; a.z = b->z;
    a.tri_off = b->tri2_off; a.tri = b->tri2; a.tri_cnt = b->tri2_cnt;
    a.loose_deg = rp->loose_deg; a.tight_deg = rp->tight_deg; a.height_factor = rp->height_factor;
    a.tri_height = o->tri_height; a.tri_flags = o->tri_flags; a.height_level = o->height_level; a.status = o->status; a.n_kept = o->n_kept;
    a.keep = keep; a.dt_status = dt_status; a.id_triples = id_triples; a.frame_ids = frame_ids;
    a.n_hyp = rp->n_hyp; a.ransac_min_points = rp->ransac_min_points; a.max_tri = (int32_t)max_tri;
    a.threshold = rp->threshold; a.goal_fraction = rp->goal_fraction; a.absolute_reference = rp->absolute_reference;
    a.seed = rp->seed; a.frame_base = rp->frame_base;
    a.raw_scale = o->raw_scale; a.model = o->model; a.best_ic = o->best_ic; a.used = o->used; a.hyp_counts = o->hyp_counts;
    // heights (reused by the 16-bit point list: 6 B per row <= 8), scalars, planes, histogram, flags, hypotheses
    const size_t lds = 8u * (size_t)max_tri + 32 + 4u * 48 + 24u * (size_t)((b->max_feat + 1) & ~1) + 4u * 2048 + (size_t)max_tri + 32
                       + 36u * (size_t)rp->n_hyp + 16;
    if ((rc = rs_prepare(flat_selection_kernel<true, kFlatDevWaves>, lds))) return rc;
    hipLaunchKernelGGL((flat_selection_kernel<true, kFlatDevWaves>), dim3((unsigned)b->n_frames), dim3(kFlatDevWaves * kWave), lds, ctx_stream(ctx), a);
    return check_launch("flat_selection_kernel<device>");
}

int mvosr_slew_median(mvosr_ctx *ctx, const double *raw, const int32_t *apply, int64_t n, double slew, double scale_in,
                      int window, const double *queue_in, int n_queue, double *pushed, double *filtered) {
    if (!ctx || (n > 0 && (!raw || !apply || !pushed || !filtered))) return set_error(MVOSR_ERR_ARG, "slew_median: null argument");
    if (n <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(slew_kernel, dim3(1), dim3(kWave), 0, ctx_stream(ctx), raw, apply, n, slew, scale_in, pushed);
    if ((rc = check_launch("slew_kernel"))) return rc;
    return mvosr_window_median(ctx, pushed, n, window, queue_in, n_queue, filtered);    // np.median(self.scale_queue), rescale.py:178
}

static int launch_ransac(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                         const double *px, const double *py, const double *pz, const int32_t *samples, int n_hyp,
                         double threshold, double goal_fraction, int32_t *counts, double *model, int32_t *best_ic,
                         int32_t *used, int line) {
    if (!ctx || !pts_off || !pts_cnt || !px || !py || (!pz && !line) || !samples || !model || !best_ic || !used)
        return set_error(MVOSR_ERR_ARG, "ransac: null argument");
    if (n_hyp < 1 || n_hyp > kMaxHyp) return set_error(MVOSR_ERR_ARG, "ransac: n_hyp must be in 1..%d", kMaxHyp);
    if (n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    RansacArgs a;
    a.n_frames = n_frames; a.pts_off = pts_off; a.pts_cnt = pts_cnt; a.px = px; a.py = py; a.pz = pz; a.triples = samples;
    a.n_hyp = n_hyp; a.threshold = threshold; a.goal_fraction = goal_fraction; a.counts = counts; a.model = model;
    a.best_ic = best_ic; a.used = used; a.line = line;
    const size_t lds = 36u * (size_t)n_hyp + 16;
    hipLaunchKernelGGL(ransac_plane_kernel, dim3((unsigned)n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    return check_launch("ransac_plane_kernel");
}

int mvosr_ransac_plane_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                             const double *px, const double *py, const double *pz, const int32_t *triples, int n_hyp,
                             double threshold, double goal_fraction, int32_t *counts, double *model, int32_t *best_ic,
                             int32_t *used) {
    return launch_ransac(ctx, n_frames, pts_off, pts_cnt, px, py, pz, triples, n_hyp, threshold, goal_fraction, counts, model,
                         best_ic, used, 0);
}

int mvosr_ransac_line_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                            const double *px, const double *py, const int32_t *pairs, int n_hyp,
                            double threshold, double goal_fraction, int32_t *counts, double *model, int32_t *best_ic,
                            int32_t *used) {
    return launch_ransac(ctx, n_frames, pts_off, pts_cnt, px, py, nullptr, pairs, n_hyp, threshold, goal_fraction, counts, model,
                         best_ic, used, 1);
}

int mvosr_triangle_batch(mvosr_ctx *ctx, const mvosr_batch *b, double focus, double cx, double cy, double s_min,
                         double n_sigma, double *height, int32_t *counts, int32_t *status) {
    if (!ctx || !b || !height || !counts || !status) return set_error(MVOSR_ERR_ARG, "triangle_batch: null argument");
    if (!b->feat_off || !b->feat_cnt || !b->x || !b->v || !b->z || !b->tri1_off || !b->tri1)
        return set_error(MVOSR_ERR_ARG, "triangle_batch: missing u (x) / v / depth (z) / tri1");
    if (b->n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    g_rs_max_lds = ctx->max_lds_per_block;
    TriBatchArgs a;
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.u = b->x; a.v = b->v; a.depth = b->z;
    a.tri_off = b->tri1_off; a.tri = b->tri1; a.focus = focus; a.cx = cx; a.cy = cy; a.s_min = s_min; a.n_sigma = n_sigma;
    a.height = height; a.counts = counts; a.status = status;
    const size_t lds = 24u * (size_t)((b->max_feat + 1) & ~1) + 8u * 3 * 2 * kRsWaves + 32;
    if ((rc = rs_prepare(triangle_batch_kernel, lds))) return rc;
    hipLaunchKernelGGL(triangle_batch_kernel, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    return check_launch("triangle_batch_kernel");
}

int mvosr_plane_inliers(mvosr_ctx *ctx, int64_t n, const double *px, const double *py, const double *pz, const double *model4,
                        double threshold, uint8_t *mask) {
    if (!ctx || !model4 || (n > 0 && (!px || !py || !pz || !mask))) return set_error(MVOSR_ERR_ARG, "plane_inliers: null argument");
    if (n <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(plane_inliers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx_stream(ctx), n, px, py, pz,
                       model4[0], model4[1], model4[2], model4[3], threshold, mask);
    return check_launch("plane_inliers_kernel");
}

}  // extern "C"
