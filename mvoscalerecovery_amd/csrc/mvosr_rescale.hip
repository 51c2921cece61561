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
};

__global__ __launch_bounds__(kRsBlock) void graph_inliers_kernel(const GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n = a.feat_cnt[f];
    if (n <= 0) { if (threadIdx.x == 0 && a.status) a.status[f] = MVOSR_ST_ERR_EMPTY; return; }
    const int64_t off = a.feat_off[f];
    const int64_t tb = a.tri_off[f];
    const int tn = (int)(a.tri_off[f + 1] - tb);
    double2 *P = reinterpret_cast<double2 *>(smem);                       // {v, z}
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + 16u * (uint32_t)((n + 1) & ~1));
    int *flag = reinterpret_cast<int *>(cnt + n + 4);
    const int tid = threadIdx.x;
    if (tid == 0) *flag = 0;
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
    if (bad) *flag = 1;
    __syncthreads();
    for (int i = tid; i < n; i += kRsBlock) {
        const uint32_t c = cnt[i];
        a.total[off + i] = (int32_t)(c & 0xFFFFu);
        a.good[off + i] = (int32_t)(c >> 16);
    }
    if (tid == 0 && a.status) a.status[f] = *flag ? MVOSR_ST_ERR_MASK : 0;
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
};

constexpr int kFlatBins = 2048;         // one histogram pass resolves 11 bits of the candidates' range
constexpr int kFlatRows = 4;            // triangle rows a thread keeps in flight
constexpr int kFlatDirect = 64;         // that few candidates left: wavefront 0 ranks them directly
// misc[] slots of flat_selection_kernel
enum { FM_K = 0, FM_SINGULAR = 1, FM_BADID = 2, FM_KEPT = 3, FM_BIN = 4, FM_RANK = 5, FM_BINCNT = 6, FM_LE = 7, FM_LIST = 8,
       FM_WSUM = 16 /* [8] per-wave bin totals */, FM_N = 32 };

__global__ __launch_bounds__(kRsBlock) void flat_selection_kernel(const FlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n = a.feat_cnt[f];
    const int64_t off = a.feat_off[f];
    const int64_t tb = a.tri_off[f];
    const int tn = (int)(a.tri_off[f + 1] - tb);
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    if (n <= 0 || tn <= 0) {
        if (tid == 0) { a.status[f] = MVOSR_ST_ERR_EMPTY; a.height_level[f] = nan(""); a.n_kept[f] = 0; }
        return;
    }
    // LDS: heights and flags of every triangle by row, the scalars, then the vertex planes — which are dead once the
    // normals are done: the histogram of the median search takes their place
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    double *Hh = reinterpret_cast<double *>(smem);               // every triangle's height, by row
    unsigned long long *U = reinterpret_cast<unsigned long long *>(Hh);   // (heights are >= 0: the bit patterns order like the values)
    unsigned long long *ext = U + tn;                            // [2] smallest / largest loose height (bits), [2] scratch
    int *misc = reinterpret_cast<int *>(ext + 4);                // FM_N scalars
    double *X = reinterpret_cast<double *>(misc + FM_N);
    double *Y = X + npad;
    double *Z = Y + npad;
    int *hist = reinterpret_cast<int *>(X);                      // kFlatBins bins; later the short candidate list (64-bit)
    const uint32_t planes = 24u * npad > 4u * kFlatBins ? 24u * npad : 4u * kFlatBins;
    uint8_t *Fl = reinterpret_cast<uint8_t *>(X) + planes;       // every triangle's flags, by row
#ifdef MVOSR_FS_STAMPS
    unsigned long long st[6];
#define FS_STAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define FS_STAMP(i) do {} while (0)
#endif
    FS_STAMP(0);
    if (tid < FM_N) misc[tid] = 0;
    if (tid == 0) { ext[0] = ~0ull; ext[1] = 0ull; ext[2] = ~0ull; }
    // a thread's next kFlatRows triangle rows are in flight while it works on the current ones (and the first ones while
    // the vertex planes stream in): a row is a dependent global load in front of nine LDS gathers and the LU chain
    TriIds rows[kFlatRows];
    auto load_rows = [&](int t0) {
#pragma unroll
        for (int j = 0; j < kFlatRows; ++j) rows[j] = load_tri(a.tri + 3 * tb, min(t0 + j * kRsBlock, tn - 1));
    };
    load_rows(tid);
#pragma unroll 4
    for (int i = tid; i < n; i += kRsBlock) { X[i] = a.x[off + i]; Y[i] = a.y[off + i]; Z[i] = a.z[off + i]; }
    const double s_loose = sin(a.loose_deg * 3.141592653589793 / 180.0), s_tight = sin(a.tight_deg * 3.141592653589793 / 180.0);
    __syncthreads();
    FS_STAMP(1);
    // phase 1: every triangle's height and flags into LDS (and the height to the output), the loose ones counted and bounded
    {
        int k_mine = 0;
        unsigned long long umin = ~0ull, umax = 0ull;
        auto one_row = [&](const TriIds q, const int t) {
            if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) {
                misc[FM_BADID] = 1; Fl[t] = 0; Hh[t] = nan(""); a.tri_height[tb + t] = nan(""); return;
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
            a.tri_height[tb + t] = h;
            if (loose) {
                const unsigned long long u = (unsigned long long)__double_as_longlong(h);
                ++k_mine; umin = u < umin ? u : umin; umax = u > umax ? u : umax;
            }
        };
        for (int t0 = tid; t0 < tn; t0 += kFlatRows * kRsBlock) {
            TriIds cur[kFlatRows];
#pragma unroll
            for (int j = 0; j < kFlatRows; ++j) cur[j] = rows[j];
            if (t0 + kFlatRows * kRsBlock < tn) load_rows(t0 + kFlatRows * kRsBlock);
#pragma unroll
            for (int j = 0; j < kFlatRows; ++j) if (t0 + j * kRsBlock < tn) one_row(cur[j], t0 + j * kRsBlock);
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
            for (int b = tid; b < kFlatBins; b += kRsBlock) hist[b] = 0;
            __syncthreads();
            #pragma unroll 4
            for (int t = tid; t < tn; t += kRsBlock) {
                if (!(Fl[t] & 1)) continue;
                const unsigned long long u = U[t];
                if (u >= lo && u <= hi) atomicAdd(&hist[(int)((u - lo) >> shift)], 1);
            }
            __syncthreads();
            {   // the bin whose cumulative count passes the rank: four bins per thread, wave scan, wave totals through LDS
                const int b0 = hist[4 * tid], b1 = hist[4 * tid + 1], b2 = hist[4 * tid + 2], b3 = hist[4 * tid + 3];
                const int mine = (b0 + b1) + (b2 + b3);
                int incl = mine;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
                if (lane == kWave - 1) misc[FM_WSUM + wave] = incl;
                __syncthreads();
                int before = 0;
#pragma unroll
                for (int w = 0; w < kRsWaves; ++w) if (w < wave) before += misc[FM_WSUM + w];
                incl += before;
                const int excl = incl - mine;
                if (rank >= excl && rank < incl) {               // exactly one thread
                    int r = rank - excl, bin = 4 * tid, c = b0;
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
                for (int t = tid; t < tn; t += kRsBlock) {
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
            for (int t = tid; t < tn; t += kRsBlock) {
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
    int kept = 0;
    #pragma unroll 4
    for (int t = tid; t < tn; t += kRsBlock) {
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
constexpr int kMaxHyp = 512;

constexpr int kRansacPPT = 8;           // points per thread per chunk (chunks of 4096 points)

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

int mvosr_graph_inliers_batch(mvosr_ctx *ctx, const mvosr_batch *b, uint32_t good_bits, int32_t *total, int32_t *good,
                              int32_t *status) {
    if (!ctx || !b || !total || !good) return set_error(MVOSR_ERR_ARG, "graph_inliers: null argument");
    if (!b->feat_off || !b->feat_cnt || !b->z || !b->v || !b->tri1_off || !b->tri1) return set_error(MVOSR_ERR_ARG, "graph_inliers: missing z/v/tri1");
    if (b->n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    g_rs_max_lds = ctx->max_lds_per_block;
    GraphArgs a;
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.z = b->z; a.v = b->v;
    a.tri_off = b->tri1_off; a.tri = b->tri1; a.good_bits = good_bits; a.total = total; a.good = good; a.status = status;
    const size_t lds = 16u * (size_t)((b->max_feat + 1) & ~1) + 4u * ((size_t)b->max_feat + 4) + 16;
    if ((rc = rs_prepare(graph_inliers_kernel, lds))) return rc;
    hipLaunchKernelGGL(graph_inliers_kernel, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    return check_launch("graph_inliers_kernel");
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
    FlatArgs a;
    a.n_frames = b->n_frames; a.feat_off = b->feat_off; a.feat_cnt = b->feat_cnt; a.x = b->x; a.y = b->y; a.z = b->z;
    a.tri_off = b->tri2_off; a.tri = b->tri2; a.loose_deg = loose_deg; a.tight_deg = tight_deg; a.height_factor = height_factor;
    a.tri_height = tri_height; a.tri_flags = tri_flags; a.height_level = height_level; a.status = status; a.n_kept = n_kept;
    if (max_tri <= 0) max_tri = 2 * (int64_t)b->max_feat;
    size_t lds = 24u * (size_t)((b->max_feat + 1) & ~1);
    if (lds < 4u * 2048) lds = 4u * 2048;                        // (the histogram of the median search reuses the vertex planes)
    lds += 9u * (size_t)max_tri + 32 + 4u * 32 + 16;
    if ((rc = rs_prepare(flat_selection_kernel, lds))) return rc;
    hipLaunchKernelGGL(flat_selection_kernel, dim3((unsigned)b->n_frames), dim3(kRsBlock), lds, ctx_stream(ctx), a);
    return check_launch("flat_selection_kernel");
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
