// mvosr_delaunay.hip — batched 2-D Delaunay triangulation on the GPU: the two
// scipy.spatial.Delaunay calls of /root/reference/src/scale_calculator.py:257-258,266-267, which cost 3-3.5 ms each
// on a host core and bound the end-to-end rate of the drop-in path, as an optional device stage.
//
// This is a DELIBERATE DEVIATION, selected explicitly (triangulation="gpu"), never the default: the reference's
// depth-order vote depends on the order of the vertices INSIDE a row of Qhull's output (scale_calculator.py:113-115),
// and that order is a by-product of Qhull's processing order, not a function of the geometry.  For points in general
// position the triangle SET is unique and this kernel returns exactly that set (tested against SciPy); what it cannot
// return is Qhull's rotation of each row.  Rows come out positively oriented (like SciPy's) with the smallest vertex
// first, sorted by that vertex and then counter-clockwise around it — a canonical, documented form.  DESIGN.md reports
// how often the quantised scales still equal the reference's.
//
// Algorithm: one workgroup per frame, the frame's points in LDS in fp64; every point builds its own Delaunay star,
// independently of all others (no shared mutable structure, no ordering between stars): one scan of the frame collects
// the points within a radius R of p (a few average spacings), the nearest of them is a Delaunay neighbour, and from it
// the star is wrapped counter-clockwise (then clockwise, if p is on the hull): the third vertex of the triangle on the
// left of the directed edge (p, q) is the point c on that side that sees the edge under the largest angle, i.e. with
// the smallest cot = (c-p).(c-q) / cross(q-p, c-p).  A completion is final when its circumcircle lies inside the
// candidates' radius (2r <= R); otherwise, and on the hull, it is redone over all points.  A triangle is written by its
// smallest vertex, so every triangle appears once.  Points are processed in chunks; a chunk's rows are staged in LDS
// and written in point order (block prefix sum), so the output does not depend on scheduling.
//
// Robustness: plain fp64 predicates with guard bands.  A frame in which a decision is within the guard band — two
// candidates with (nearly) the same cot: four cocircular points; a point (nearly) on the line through an edge;
// duplicate points — or whose row count is not Euler's 2n - 2 - h is flagged MVOSR_DT_DEGENERATE and left to the
// host's Qhull (SciPy resolves such inputs by its own joggling rules, which are not reproducible here).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mvosr.h"
#include "mvosr_device.hpp"
#include "mvosr_host.hpp"

namespace mvosr {

constexpr int kDtWaves = 8;
constexpr int kDtBlock = kDtWaves * kWave;
constexpr int kDtChunk = 256;            // points per chunk (rows staged in LDS per chunk)
constexpr int kDtMaxOwn = 32;            // rows a point may own (it owns the triangles in which it is the smallest vertex)
constexpr int kDtMaxCand = 256;          // candidates within the radius, per point
constexpr int kDtMaxDeg = 64;
constexpr double kDtTieTol = 1e-9;       // relative guard band on cot differences / collinearity

struct DtArgs {
    int64_t n_frames;
    const int64_t *pts_off; const int32_t *pts_cnt;      // [F] the frame's points in u/v
    const double *u, *v;
    const int64_t *tri_off;                              // [F] start of the frame's rows in `tri` (capacity 2*n rows)
    int32_t *tri;                                        // rows (a, b, c)
    int32_t *tri_cnt;                                    // [F] rows written
    int32_t *status;                                     // [F] MVOSR_DT_*
};

struct DtBest { double t; int id; int tie; };

// among the wavefront's lanes: the smallest t, its id, and whether another lane's DIFFERENT point comes within the guard band
__device__ __forceinline__ DtBest dt_wave_best(double t, int id, double t2nd) {
    // per-lane (t, id) is that lane's best; t2nd its runner-up (of another point)
    double m = t;
    m = fmin(m, dpp_mov<kDppXor1>(m));
    m = fmin(m, dpp_mov<kDppXor2>(m));
    m = fmin(m, dpp_mov<kDppHalfMirror>(m));
    m = fmin(m, dpp_mov<kDppMirror>(m));
    m = fmin(fmin(readlane_d(m, 0), readlane_d(m, 16)), fmin(readlane_d(m, 32), readlane_d(m, 48)));
    DtBest r;
    r.t = m; r.id = -1; r.tie = 0;
    if (!(m < INFINITY)) return r;
    const unsigned long long who = __ballot(t == m);
    const int src = (int)__ffsll((long long)who) - 1;
    r.id = __builtin_amdgcn_readlane(id, src);
    const double band = kDtTieTol * (fabs(m) + 1.0);
    // a tie: another lane's best (a different point) or any lane's runner-up within the band
    const bool close = (t - m <= band && id != r.id && id >= 0) || (t2nd - m <= band);
    r.tie = __ballot(close) != 0ull;
    return r;
}

__global__ __launch_bounds__(kDtBlock) void delaunay_kernel(const DtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = blockIdx.x;
    const int n = a.pts_cnt[f];
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (n < 3) {
        if (tid == 0) { a.tri_cnt[f] = 0; a.status[f] = MVOSR_DT_DEGENERATE; }
        return;
    }
    const int64_t off = a.pts_off[f];
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    double2 *P = reinterpret_cast<double2 *>(smem);                                  // {u, v}
    int *stage = reinterpret_cast<int *>(smem + 16u * npad);                         // [kDtChunk][kDtMaxOwn] packed rows (b | c << 16)
    uint8_t *own = reinterpret_cast<uint8_t *>(stage + kDtChunk * kDtMaxOwn);        // [kDtChunk] rows owned
    uint16_t *cand_all = reinterpret_cast<uint16_t *>(own + kDtChunk);               // [kDtWaves][kDtMaxCand]
    double *red = reinterpret_cast<double *>(smem + ((16u * npad + 4u * kDtChunk * kDtMaxOwn + kDtChunk + 2u * kDtWaves * kDtMaxCand + 15u) & ~15u));
    int *misc = reinterpret_cast<int *>(red + 8 * kDtWaves);
    uint16_t *cand = cand_all + w * kDtMaxCand;

    // points -> LDS, bounding box
    double lo_u = INFINITY, hi_u = -INFINITY, lo_v = INFINITY, hi_v = -INFINITY;
    for (int i = tid; i < n; i += kDtBlock) {
        double2 p; p.x = a.u[off + i]; p.y = a.v[off + i];
        P[i] = p;
        lo_u = fmin(lo_u, p.x); hi_u = fmax(hi_u, p.x); lo_v = fmin(lo_v, p.y); hi_v = fmax(hi_v, p.y);
    }
    {
        // block min/max through LDS (one slot per wave and quantity)
        auto wave_min_d = [&](double x) {
            x = fmin(x, dpp_mov<kDppXor1>(x)); x = fmin(x, dpp_mov<kDppXor2>(x)); x = fmin(x, dpp_mov<kDppHalfMirror>(x)); x = fmin(x, dpp_mov<kDppMirror>(x));
            return fmin(fmin(readlane_d(x, 0), readlane_d(x, 16)), fmin(readlane_d(x, 32), readlane_d(x, 48)));
        };
        const double a0 = wave_min_d(lo_u), a1 = wave_min_d(-hi_u), a2 = wave_min_d(lo_v), a3 = wave_min_d(-hi_v);
        if (lane == 0) { red[4 * w] = a0; red[4 * w + 1] = a1; red[4 * w + 2] = a2; red[4 * w + 3] = a3; }
        if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0; }      // [0] degenerate flag, [1] rows written so far, [2] hull edges
        __syncthreads();
        lo_u = INFINITY; hi_u = INFINITY; lo_v = INFINITY; hi_v = INFINITY;
        for (int i = 0; i < kDtWaves; ++i) { lo_u = fmin(lo_u, red[4 * i]); hi_u = fmin(hi_u, red[4 * i + 1]); lo_v = fmin(lo_v, red[4 * i + 2]); hi_v = fmin(hi_v, red[4 * i + 3]); }
        hi_u = -hi_u; hi_v = -hi_v;
    }
    const double area = fmax((hi_u - lo_u) * (hi_v - lo_v), 1e-300);
    const double R0 = 4.0 * sqrt(area / (double)n);              // a few average spacings
    int degenerate = 0, hull_edges = 0;
    int32_t *rows = a.tri + 3 * a.tri_off[f];

    // one completion: the point c strictly on the side `sgn` of the directed edge p -> q (relative coordinates
    // a = q - p) that minimises cot(angle pcq); `list` != nullptr: among the candidates, else among all points
    auto complete = [&](const double2 p, int ip, int iq, double sgn, const uint16_t *list, int nlist) -> DtBest {
        const double2 qa = P[iq];
        const double ax = qa.x - p.x, ay = qa.y - p.y;
        const double la = sqrt(ax * ax + ay * ay);
        double bt = INFINITY, bt2 = INFINITY;
        int bid = -1;
        int flag = 0;
        const int count = list ? nlist : n;
        for (int j = lane; j < count; j += kWave) {
            const int ic = list ? (int)list[j] : j;
            if (ic == ip || ic == iq) continue;
            const double2 c = P[ic];
            const double bx = c.x - p.x, by = c.y - p.y;
            const double cr = sgn * (ax * by - ay * bx);
            const double lb = sqrt(bx * bx + by * by);
            if (fabs(cr) <= kDtTieTol * 1e-3 * la * lb) { if (bx * ax + by * ay > 0.0 || lb == 0.0) flag = 1; continue; }   // (nearly) on the line, ahead of p
            if (cr <= 0.0) continue;
            const double t = (bx * (bx - ax) + by * (by - ay)) / cr;
            if (t < bt) { bt2 = bt; bt = t; bid = ic; } else if (t < bt2) bt2 = t;
        }
        DtBest r = dt_wave_best(bt, bid, bt2);
        r.tie = (r.tie ? 2 : 0) | (__ballot(flag) != 0ull ? 4 : 0);
        return r;
    };

    for (int c0 = 0; c0 < n; c0 += kDtChunk) {
        const int cn = min(kDtChunk, n - c0);
        for (int i = tid; i < cn; i += kDtBlock) own[i] = 0;
        __syncthreads();
        // ---- stars of the chunk's points, one wavefront per point
        for (int ip = c0 + w; ip < c0 + cn; ip += kDtWaves) {
            const double2 p = P[ip];
            double R = R0, Rlist = R0;           // Rlist: the radius the candidate list was actually collected with
            int ncand = 0, inear = -1;
            // candidates within R (grown until there are at least 8), nearest neighbour
            for (int attempt = 0; attempt < 6; ++attempt) {
                ncand = 0;
                Rlist = R;
                double dn = INFINITY; int in_ = -1;
                const double R2 = R * R;
                for (int j0 = 0; j0 < n; j0 += kWave) {
                    const int j = j0 + lane;
                    bool in = false;
                    if (j < n && j != ip) {
                        const double2 c = P[j];
                        const double dx = c.x - p.x, dy = c.y - p.y, d2 = dx * dx + dy * dy;
                        in = d2 <= R2;
                        if (d2 < dn) { dn = d2; in_ = j; }
                    }
                    const unsigned long long m = __ballot(in);
                    const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
                    if (in && pos < kDtMaxCand) cand[pos] = (uint16_t)j;
                    ncand += __popcll(m);
                }
                // wave argmin of the nearest neighbour
                double m = dn;
                m = fmin(m, dpp_mov<kDppXor1>(m)); m = fmin(m, dpp_mov<kDppXor2>(m)); m = fmin(m, dpp_mov<kDppHalfMirror>(m)); m = fmin(m, dpp_mov<kDppMirror>(m));
                m = fmin(fmin(readlane_d(m, 0), readlane_d(m, 16)), fmin(readlane_d(m, 32), readlane_d(m, 48)));
                const unsigned long long who = __ballot(dn == m);
                inear = __builtin_amdgcn_readlane(in_, (int)__ffsll((long long)who) - 1);
                if (m == 0.0 || __popcll(who) > 1) degenerate |= 1;      // duplicate point / two equally near neighbours
                if (ncand > kDtMaxCand) { R *= 0.5; continue; }
                if (ncand >= 8 || ncand >= n - 1) break;
                R *= 2.0;
            }
            if (ncand > kDtMaxCand) ncand = 0;                           // (give up on the list: every completion scans all points)
            // wrap the star: counter-clockwise from the nearest neighbour, then clockwise if the star is open
            int nown = 0, deg = 0;
            for (int dir = 0; dir < 2; ++dir) {
                const double sgn = dir == 0 ? 1.0 : -1.0;
                int iq = inear;
                bool open = false;
                while (true) {
                    DtBest b = complete(p, ip, iq, sgn, cand, ncand);
                    bool redo = b.id < 0;
                    if (!redo) {
                        // the circumcircle of (p, q, c) must lie inside the candidates' radius for the answer to be final
                        const double2 q = P[iq], c = P[b.id];
                        const double ax = q.x - p.x, ay = q.y - p.y, bx = c.x - p.x, by = c.y - p.y;
                        const double cr = ax * by - ay * bx, a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
                        const double ox = (by * a2 - ay * b2) / (2.0 * cr), oy = (ax * b2 - bx * a2) / (2.0 * cr);
                        if (!(4.0 * (ox * ox + oy * oy) <= Rlist * Rlist)) redo = true;
                    }
                    if (redo) b = complete(p, ip, iq, sgn, nullptr, 0);
                    if (b.tie) degenerate |= b.tie;
                    if (b.id < 0) { open = true; ++hull_edges; break; }          // nothing on that side: a hull edge
                    if (++deg > kDtMaxDeg) { degenerate |= 8; break; }
                    // the triangle (p, q, c) [dir 0] / (p, c, q) [dir 1] is positively oriented; p writes it when p is its smallest vertex
                    if (ip < iq && ip < b.id) {
                        if (nown < kDtMaxOwn) {
                            if (lane == 0) stage[(ip - c0) * kDtMaxOwn + nown] = dir == 0 ? (iq | (b.id << 16)) : (b.id | (iq << 16));
                            ++nown;
                        } else degenerate |= 16;
                    }
                    iq = b.id;
                    if (iq == inear) break;                                       // closed
                }
                if (!open) break;                                                 // interior point: one direction closes the star
                if (dir == 0 && deg == 0) { /* the first edge is a hull edge on its left: go clockwise from it */ }
            }
            if (lane == 0) own[ip - c0] = (uint8_t)nown;
        }
        __syncthreads();
        // ---- the chunk's rows in point order: block prefix over own[], then (smallest, b, c)
        {
            int mine = 0;
            const int per = (cn + kDtBlock - 1) / kDtBlock;
            const int i0 = tid * per, i1 = min(cn, i0 + per);
            for (int i = i0; i < i1; ++i) mine += own[i];
            // exclusive scan of `mine` over the block
            int incl = mine;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            int *wsum = misc + 8;
            if (lane == kWave - 1) wsum[w] = incl;
            __syncthreads();
            int base = misc[1];
            for (int i = 0; i < w; ++i) base += wsum[i];
            int at = base + incl - mine;
            for (int i = i0; i < i1; ++i) {
                const int k = own[i];
                for (int j = 0; j < k; ++j) {
                    const int pk = stage[i * kDtMaxOwn + j];
                    if (at < 2 * n) { rows[3 * at] = c0 + i; rows[3 * at + 1] = pk & 0xFFFF; rows[3 * at + 2] = (pk >> 16) & 0xFFFF; }
                    ++at;
                }
            }
            __syncthreads();
            if (tid == kDtBlock - 1) misc[1] = at;                       // (the last thread's end = the chunk's end)
            __syncthreads();
        }
    }
    // flags / hull count of all wavefronts, Euler's relation
    if (lane == 0) { if (degenerate) atomicOr(&misc[0], degenerate); atomicAdd(&misc[2], hull_edges); }
    __syncthreads();
    if (tid == 0) {
        const int total = misc[1];
        const int h = misc[2] / 2;                      // every hull edge is met from both of its end points
        int st = MVOSR_DT_OK;
        // (bits 8.. say why, for diagnostics: 1 duplicate / equidistant nearest points, 2 cocircular, 4 collinear, 8 degree cap,
        //  16 rows-per-point cap, 32 Euler's relation)
        int why = misc[0];
        if (total != 2 * n - 2 - h || (misc[2] & 1) || total > 2 * n) why |= 32;
        if (why) st = MVOSR_DT_DEGENERATE | (why << 8);
        a.tri_cnt[f] = total > 2 * n ? 0 : total;
        a.status[f] = st;
    }
}

}  // namespace mvosr

using namespace mvosr;

extern "C" int mvosr_delaunay_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                    const double *u, const double *v, int max_pts, const int64_t *tri_off, int32_t *tri,
                                    int32_t *tri_cnt, int32_t *status) {
    if (!ctx || !pts_off || !pts_cnt || !u || !v || !tri_off || !tri || !tri_cnt || !status)
        return set_error(MVOSR_ERR_ARG, "delaunay_batch: null argument");
    if (max_pts < 0 || max_pts > 65535) return set_error(MVOSR_ERR_TOO_LARGE, "delaunay_batch: at most 65535 points per frame");
    if (n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    const uint32_t npad = (uint32_t)((max_pts + 1) & ~1);
    const size_t lds = ((16u * npad + 4u * kDtChunk * kDtMaxOwn + kDtChunk + 2u * kDtWaves * kDtMaxCand + 15u) & ~15u) + 8u * 8u * kDtWaves + 4u * 32u;
    if (lds > 160u * 1024u) return set_error(MVOSR_ERR_TOO_LARGE, "delaunay_batch: %d points need %zu B of LDS", max_pts, lds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(delaunay_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_hip_error("hipFuncSetAttribute(delaunay_kernel)", e);
    DtArgs a;
    a.n_frames = n_frames; a.pts_off = pts_off; a.pts_cnt = pts_cnt; a.u = u; a.v = v; a.tri_off = tri_off; a.tri = tri;
    a.tri_cnt = tri_cnt; a.status = status;
    hipLaunchKernelGGL(delaunay_kernel, dim3((unsigned)n_frames), dim3(kDtBlock), lds, ctx_stream(ctx), a);
    return check_launch("delaunay_kernel");
}
