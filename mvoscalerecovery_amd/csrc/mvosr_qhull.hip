// mvosr_qhull.hip — SciPy/Qhull's Delaunay ROWS on the device: the triangle set AND the order of the rows AND the rotation of
// every row, i.e. exactly `scipy.spatial.Delaunay(points2d).simplices` of /root/reference/src/scale_calculator.py:257-258 and
// :266-267.  The reference's check_triangle (:105-119) reads the rotation of a row, and that rotation is Qhull's insertion
// order (a row is the facet's vertices by decreasing vertex id, first two swapped for orientation): this kernel replays
// Qhull's beneath-beyond (qhull_r 7.3.2 / 2019.1.r inside SciPy 1.15.3, options `d Qbb Qc Qz Q12 Qt`) decision for decision,
// for sites in general position.  oracle/qhull_rows.py is the CPU restatement the tests compare it with; it carries the
// step-by-step description.  A frame that leaves the general-position regime (a decision inside a roundoff guard band, a
// merge, a narrow initial simplex) is DECLINED (status != 0): the host triangulates it with SciPy itself.
//
// One wavefront per frame; the algorithm is a chain of ~n dependent insertions, each a handful of dependent memory round
// trips, so the GPU is filled with FRAMES (thousands of wavefronts), not with lanes: a frame's state (64 B per facet, ~5.6
// facets per point over the run) lives in a slice of the context's workspace and is read through L2; the lanes of a
// wavefront share the visibility search (three neighbours per visible facet at a time), build the cone's facets (one lane
// each, planes in LDS) and partition the visible facets' points among them (one lane per point, a directed walk over the
// cone in LDS).  What is order-dependent in Qhull is kept order-exact: the breadth-first visible list, the creation order of
// the cone's facets, the arrival order inside every outside set (the furthest point rides last; a displaced furthest point
// takes the arrival slot of its displacer), the point from which a 'sharp' cone switches the partition to a linear scan.
#include "mvosr_device.hpp"
#include "mvosr_host.hpp"

using namespace mvosr;

namespace {

constexpr int kQhMaxPoints = 8000;        // facet ids are 16-bit: 7 * (n + 1) + 64 facets per run
constexpr int kQhMaxPointsWide = 60000;   // 32-bit facet ids (80-byte records); point ids stay 16-bit
// table entries per insertion: 64 (one frame per wavefront) or 32 (packed kernels) — cone facets 5.6 on average, 35 the most seen
// in 512 frames of 2000 points; visible facets 3.6 on average, 33 the most seen; horizon facets about as many
constexpr int kQhPickWindow = 16;
constexpr double kQhEps = 2.220446049250313e-16;
constexpr double kQhHuge = 1.797e308;
constexpr uint16_t kQhNone = 0xFFFFu;

enum QhWhy : int {
    QH_OK = 0, QH_FEW_POINTS = 1, QH_ZERO_WIDTH = 2, QH_SIMPLEX_SEARCH = 3, QH_FLAT_SIMPLEX = 4, QH_NARROW = 5, QH_INSIDE_SIMPLEX = 6,
    QH_BAND = 7, QH_COPLANAR_HORIZON = 8, QH_TOO_MANY_VISIBLE = 9, QH_CONE_TOO_LARGE = 10, QH_OPEN_CONE = 11, QH_NOT_CONVEX = 12,
    QH_GAUSS = 13, QH_NOT_SHARP = 14, QH_ABOVE_NONE = 15, QH_FACETS_FULL = 16, QH_ARENA_FULL = 17
};

// A facet record, FID = the type of a facet id: 16 bits up to 8 000 points (7 n + 64 facets per run fit), 32 bits beyond
// (the dense frames of BASELINE configs[4]: 20 000 points, 140 000 facets).
template <typename FID> struct QhFacetT;
template <> struct __attribute__((aligned(16))) QhFacetT<uint16_t> {
    uint16_t p[3];      // vertices as point ids, by decreasing vertex id (reverse insertion order)
    uint16_t flags;     // 1 top-oriented, 2 upper Delaunay, 4 dead
    uint16_t nb[3];     // neighbour k is opposite vertex k
    uint16_t bestp;     // the outside set's furthest point (kQhNone: empty set)
    uint32_t off;       // the rest of the outside set: arena[off .. off + cnt), in Qhull's list order
    uint16_t cnt;
    uint16_t mark;      // (spare)
    double bestd;
    double n0, n1, n2, d;
};
template <> struct __attribute__((aligned(16))) QhFacetT<uint32_t> {
    uint16_t p[3];
    uint16_t flags;
    uint32_t nb[3];
    uint16_t bestp;
    uint16_t cnt;
    uint32_t off;
    uint32_t mark;
    double bestd;
    double n0, n1, n2, d;
    double pad;
};
static_assert(sizeof(QhFacetT<uint16_t>) == 64, "one facet per 64-byte line");
static_assert(sizeof(QhFacetT<uint32_t>) == 80, "five 16-byte words");

// the first 32 bytes of a record (what the pick reads for a window of facets): flags, furthest point, neighbours, list
template <typename FID> struct QhHead { uint32_t flags, bestp, nb0, nb1, nb2, off, cnt; };
__device__ __forceinline__ QhHead<uint16_t> qh_head(const QhFacetT<uint16_t> *, uint4 r0, uint4 r1) {
    QhHead<uint16_t> h;
    h.flags = r0.y >> 16; h.bestp = r0.w >> 16; h.nb0 = r0.z & 0xFFFFu; h.nb1 = r0.z >> 16; h.nb2 = r0.w & 0xFFFFu; h.off = r1.x; h.cnt = r1.y & 0xFFFFu;
    return h;
}
__device__ __forceinline__ QhHead<uint32_t> qh_head(const QhFacetT<uint32_t> *, uint4 r0, uint4 r1) {
    QhHead<uint32_t> h;
    h.flags = r0.y >> 16; h.nb0 = r0.z; h.nb1 = r0.w; h.nb2 = r1.x; h.bestp = r1.y & 0xFFFFu; h.cnt = r1.y >> 16; h.off = r1.z;
    return h;
}

struct QhArgs {
    int64_t n_frames;
    const int64_t *pts_off; const int32_t *pts_cnt; const double *u; const double *v; const int32_t *keep;
    const int64_t *tri_off; int32_t *tri; int32_t *tri_cnt; int32_t *n_used; int32_t *status; int32_t *order_out;
    unsigned long long *stamps;
    int32_t *redo;               // packed kernels: frames that overflowed a table ([0] = how many), for the list kernel
    const int32_t *list;         // null, or: list[0] frames list[1..] (a redo list of the scale kernels), walked by a persistent grid
    const int32_t *launch_order; // null, or: workgroup b runs frame launch_order[b] — the largest frames first (qh_order_kernel)
    char *ws; size_t ws_stride; int cap_pts;      // per-frame slice, laid out by QhPlan for cap_pts = max_pts + 1 points
};

#ifdef MVOSR_QH_STAMPS
static unsigned long long *g_qh_stamps = nullptr;
#define QH_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define QH_STAMP(k) do { } while (0)
#endif

struct QhPlan { size_t pts, fac, arena, tt, dd, total; uint32_t fcap, acap; };
__host__ __device__ inline QhPlan qh_plan(int cap_pts, bool wide) {
    QhPlan P;
    const size_t n = (size_t)cap_pts;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    P.pts = take(32 * n);            // (x, y, lifted z, pad): one 32-byte sector per point — three planes were three sectors per access
    P.fcap = (uint32_t)(7 * n + 64); if (!wide && P.fcap > 65534u) P.fcap = 65534u;
    P.fac = take((wide ? sizeof(QhFacetT<uint32_t>) : sizeof(QhFacetT<uint16_t>)) * (size_t)(P.fcap + 1));
    P.acap = (uint32_t)(28 * n + 256);
    P.arena = take(2 * (size_t)P.acap);
    P.tt = take(2 * n); P.dd = take(8 * n);
    P.total = o;
    return P;
}

// a coordinate of the per-point records, indexed like an array
struct QhCoord {
    double4 *p; int k;
    __device__ __forceinline__ double &operator[](int i) const { return (&p[i].x)[k]; }
};

// The tables of one insertion, TAB entries each (a frame whose insertion needs more is declined — or, from the packed kernel,
// redone by the 64-entry one).
template <typename FID, int TAB> struct QhLdsT {
    double npl[TAB][4];        // the cone's planes (n0, n1, n2, offset)
    double nxy[TAB][6];        // coordinates of a cone facet's two horizon vertices (convexity test between cone facets)
    uint16_t nva[TAB], nvb[TAB];     // a cone facet's horizon vertices (point ids)
    uint8_t nnb[TAB][4];       // cone neighbours 1, 2 as cone-local indices; [0] = top-oriented, [3] = upper flag
    FID visq[TAB];             // visible facets in Qhull's breadth-first order
    FID visnb[TAB][4];         // (a stride of four: the entry/neighbour pair of a lane is a shift and a mask)
    uint16_t visrep[TAB];      // cone-local index of the visible facet's replacement (kQhNone: the first cone facet)
    uint32_t visoff[TAB];
    uint16_t viscnt[TAB];      // points of its outside set without the furthest
    uint16_t visbest[TAB];
    uint32_t viscum[TAB + 1];
    FID nhz[TAB];              // a cone facet's horizon neighbour (facet id)
    FID hzq[TAB];              // horizon facets tested in this insertion, and their records (p0 p1 p2 flags nb0 nb1 nb2)
    FID hzr[TAB][8];
    uint32_t t_off[TAB]; uint32_t t_total[TAB]; uint32_t t_cnt[TAB]; uint16_t t_bestp[TAB]; double t_bestd[TAB];
};
struct QhArrival { int tgt, p; double d; };      // (packed kernels: a chunk of arrivals staged over `nxy`, which placement no longer needs)

// A frame's lanes: the whole wavefront (G = 64), or an aligned group of G = 32 / 16 lanes — then 64 / G frames share a wavefront
// and every vector instruction serves all of them (one insertion is ~1 200 vector instructions whether it feeds one frame or
// four).  The run is written for "the G lanes of a frame": ballots, shuffles and reductions stay inside the group; a loop whose
// trip count differs between the groups of a wavefront diverges and reconverges like any SIMT loop; the lanes of a declined
// frame drop out.  LDS instructions of a wavefront execute in order, so a group's writes are seen by its later reads.
template <int G> struct Sg {
    static_assert(G == 64 || G == 32 || G == 16, "lanes per frame");
    static constexpr uint64_t kMask = ~0ull >> (64 - G);
    static __device__ __forceinline__ int sl() { return lane_id() & (G - 1); }
    static __device__ __forceinline__ int sub() { return lane_id() / G; }
    static __device__ __forceinline__ int first() { return lane_id() & ~(G - 1); }
    static __device__ __forceinline__ uint64_t ballot(bool p) {
        const uint64_t b = __ballot(p);
        if constexpr (G == 64) return b; else return (b >> first()) & kMask;
    }
    static __device__ __forceinline__ bool any(bool p) { if constexpr (G == 64) return __any(p); else return ballot(p) != 0ull; }
    static __device__ __forceinline__ uint64_t below() { return (1ull << sl()) - 1ull; }
    template <typename T> static __device__ __forceinline__ T shfl(T v, int i) {
        if constexpr (G == 64) return __shfl(v, i); else return __shfl(v, first() + i);
    }
    static __device__ __forceinline__ int uni(int v) { if constexpr (G == 64) return __builtin_amdgcn_readfirstlane(v); else return v; }
    // (i: uniform over the wavefront when G = 64)
    static __device__ __forceinline__ double bcast_d(double v, int i) { if constexpr (G == 64) return readlane_d(v, i); else return __shfl(v, first() + i); }
    static __device__ __forceinline__ double max_d(double v) {
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o); v = w > v ? w : v; }
        return v;
    }
    static __device__ __forceinline__ double min_d(double v) {
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; }
        return v;
    }
};

__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int ffs64(uint64_t m) { return __ffsll((long long)m) - 1; }

// One wavefront per workgroup: LDS instructions of a wavefront execute in order, so a write by one lane is seen by a later read
// of another lane without a barrier — only the COMPILER must keep the order.  (__syncthreads() also waits for every global
// load and store in flight: in the insertion loop that was a memory round trip per call.)
__device__ __forceinline__ void qh_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// ... and the same for global memory: the wavefront's earlier stores have landed before its later loads are issued
__device__ __forceinline__ void qh_mem_sync() {
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

struct QhPlane { double n0, n1, n2, d; bool gauss; bool upper; };

// qh_sethyperplane_det for three vertices (rows in the facet's vertex order), qh_normalize2, the offset; where a vertex is
// further than DISTround from that plane (thin facets), qh_sethyperplane_gauss as Qhull does: elimination with partial
// pivoting on the two edge vectors, back substitution from normal[2] = -+1, positive normalisation.  `gauss` (declined): a
// pivot within Qhull's NEARzero, where it would go on to qh_orientoutside.
__device__ __forceinline__ QhPlane qh_plane(double x0, double y0, double z0, double x1, double y1, double z1, double x2, double y2, double z2,
                                            bool top, double distround, double anground, double nz0, double nz1) {
    QhPlane P;
    const double dx1 = x1 - x0, dy1 = y1 - y0, dz1 = z1 - z0;
    const double dx2 = x2 - x0, dy2 = y2 - y0, dz2 = z2 - z0;
    double n0 = dy2 * dz1 - dz2 * dy1;
    double n1 = dx1 * dz2 - dz1 * dx2;
    double n2 = dx2 * dy1 - dy2 * dx1;
    double norm = __builtin_sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    P.gauss = !(norm > 1e-290);
    if (!top) norm = -norm;
    n0 = n0 / norm; n1 = n1 / norm; n2 = n2 / norm;
    double d = -(x0 * n0 + y0 * n1 + z0 * n2);
    const double e2 = d + (x2 * n0 + y2 * n1 + z2 * n2);
    const double e1 = d + (x1 * n0 + y1 * n1 + z1 * n2);
    if (e2 > distround || e2 < -distround || e1 > distround || e1 < -distround) {
        double a0 = dx1, a1 = dy1, a2 = dz1, b0 = dx2, b1 = dy2, b2 = dz2;
        bool sign = top;
        if (__builtin_fabs(b0) > __builtin_fabs(a0)) {
            double t;
            t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; t = a2; a2 = b2; b2 = t;
            sign = !sign;
        }
        if (__builtin_fabs(a0) <= nz0) P.gauss = true;
        const double q = b0 / a0;
        b1 -= q * a1;
        b2 -= q * a2;
        if (__builtin_fabs(b1) <= nz1) P.gauss = true;
        if (b1 < 0) sign = !sign;
        if (a0 < 0) sign = !sign;
        n2 = sign ? -1.0 : 1.0;
        n1 = 0.0;
        n1 -= b2 * n2;
        n1 /= b1;
        n0 = 0.0;
        n0 -= a1 * n1;
        n0 -= a2 * n2;
        n0 /= a0;
        norm = __builtin_sqrt(n0 * n0 + n1 * n1 + n2 * n2);
        n0 = n0 / norm; n1 = n1 / norm; n2 = n2 / norm;
        d = -(x0 * n0);
        d -= y0 * n1;
        d -= z0 * n2;
    }
    P.n0 = n0; P.n1 = n1; P.n2 = n2; P.d = d;
    P.upper = n2 > -anground * 2.0;
    return P;
}

__device__ __forceinline__ double qh_dist(double x, double y, double z, double n0, double n1, double n2, double d) {
    return d + x * n0 + y * n1 + z * n2;
}

struct QhConst { double distround, minvisible, minoutside, distoutside, guard, anground, nz0, nz1; };

struct QhWalk { int tgt; double d; int state; int best; };     // state 0 placed, 1 ended below its best, 2 no best, 3 bad (band / above none)

// qh_findbestnew over the cone in LDS: list order from `start`, wrapping; the first facet 2 MINoutside above wins
template <typename L_t> __device__ __forceinline__ QhWalk qh_scan_cone(const L_t &L, int m, int start, double x, double y, double z, const QhConst &K) {
    QhWalk R; R.state = 3; R.tgt = 0; R.d = 0.0; R.best = -1;
    double bestd = -kQhHuge;
    bool bad = false;
    for (int i = 0; i < m; ++i) {
        int j = start + i; if (j >= m) j -= m;
        const double d = qh_dist(x, y, z, L.npl[j][0], L.npl[j][1], L.npl[j][2], L.npl[j][3]);
        if (d > -K.guard && d < K.guard) bad = true;
        if (d > bestd && (!L.nnb[j][3] || d >= K.minoutside)) {
            if (d >= K.distoutside) { R.tgt = j; R.d = d; R.state = bad ? 3 : 0; return R; }
            bestd = d;
        }
    }
    return R;         // above no cone facet by 2 MINoutside: Qhull goes on to qh_findbesthorizon — declined
}

// qh_findbest(isnewfacets): the directed walk; visited cone facets as a bit mask
template <typename L_t> __device__ __forceinline__ QhWalk qh_walk_cone(const L_t &L, int start, double x, double y, double z, const QhConst &K) {
    QhWalk R; R.state = 0; R.best = -1;
    bool bad = false;
    double d = qh_dist(x, y, z, L.npl[start][0], L.npl[start][1], L.npl[start][2], L.npl[start][3]);
    if (d > -K.guard && d < K.guard) bad = true;
    if (d >= K.minoutside) { R.tgt = start; R.d = d; R.state = bad ? 3 : 0; return R; }
    double bestd = d;
    int best = L.nnb[start][3] ? -1 : start;
    uint64_t seen = 1ull << start;
    int f = start;
    while (f >= 0) {
        int nxt = -1;
        for (int k = 1; k <= 2; ++k) {                      // neighbour 0 is the horizon facet (not new)
            const int g = L.nnb[f][k];
            if (seen >> g & 1ull) continue;
            seen |= 1ull << g;
            d = qh_dist(x, y, z, L.npl[g][0], L.npl[g][1], L.npl[g][2], L.npl[g][3]);
            if (d > -K.guard && d < K.guard) bad = true;
            if (d > bestd) {
                if (d >= K.minoutside) { R.tgt = g; R.d = d; R.state = bad ? 3 : 0; return R; }
                if (!L.nnb[g][3]) { best = g; bestd = d; nxt = g; break; }
                else if (best < 0) { bestd = d; nxt = g; break; }
            }
        }
        f = nxt;
    }
    R.tgt = 0; R.d = bestd; R.best = best;
    R.state = bad ? 3 : (best < 0 ? 2 : (bestd < -K.distround ? 1 : 3));
    return R;
}

// One chunk (<= G arrivals, lanes 0 .. count-1 of the frame's group in Qhull's processing order) appended to the targets' outside
// sets: qh_partitionpoint's list rule.  A target's list is arena[t_off .. t_off + t_cnt) + [t_bestp]; the furthest point stays
// last, a point that arrives further than it takes over and the old one enters the list in the arrival's place.  Lane j keeps
// target j's state in registers and the arrivals are broadcast one by one (v_readlane; from LDS in the packed kernels):
// ~14 instructions per arrival.
template <int G, typename L_t> __device__ __forceinline__ void qh_place_chunk(L_t &L, uint16_t *arena, int count, int m, int tgt, int p, double d) {
    const int sl = Sg<G>::sl();
    if constexpr (G == 64) {
        uint32_t base = 0, cnt = 0; int bestp = kQhNone; double bestd = -kQhHuge;
        if (sl < m) { base = L.t_off[sl]; cnt = L.t_cnt[sl]; bestp = L.t_bestp[sl]; bestd = L.t_bestd[sl]; }
        for (int i = 0; i < count; ++i) {
            const int ti = __builtin_amdgcn_readlane(tgt, i);
            const int pi = __builtin_amdgcn_readlane(p, i);
            const double di = readlane_d(d, i);
            if (sl == ti) {
                if (bestp == kQhNone) { bestp = pi; bestd = di; }
                else {
                    const bool further = di > bestd;
                    arena[base + cnt] = (uint16_t)(further ? bestp : pi);
                    ++cnt;
                    if (further) { bestp = pi; bestd = di; }
                }
            }
        }
        if (sl < m) { L.t_cnt[sl] = cnt; L.t_bestp[sl] = (uint16_t)bestp; L.t_bestd[sl] = bestd; }
    } else {
        QhArrival *A = reinterpret_cast<QhArrival *>(&L.nxy[0][0]);
        { QhArrival a; a.tgt = tgt; a.p = p; a.d = d; A[sl] = a; }
        qh_lds_sync();
        for (int jb = 0; jb < m; jb += G) {
            const int j = jb + sl;
            uint32_t base = 0, cnt = 0; int bestp = kQhNone; double bestd = -kQhHuge;
            if (j < m) { base = L.t_off[j]; cnt = L.t_cnt[j]; bestp = L.t_bestp[j]; bestd = L.t_bestd[j]; }
            for (int i = 0; i < count; ++i) {
                const QhArrival a = A[i];
                if (j == a.tgt) {
                    if (bestp == kQhNone) { bestp = a.p; bestd = a.d; }
                    else {
                        const bool further = a.d > bestd;
                        arena[base + cnt] = (uint16_t)(further ? bestp : a.p);
                        ++cnt;
                        if (further) { bestp = a.p; bestd = a.d; }
                    }
                }
            }
            if (j < m) { L.t_cnt[j] = cnt; L.t_bestp[j] = (uint16_t)bestp; L.t_bestd[j] = bestd; }
        }
        qh_lds_sync();
    }
}

// One frame, G lanes.  Returns the reason the frame was declined (QH_OK: rows written, `nrows` of them).
template <typename FID, int G, int TAB> __device__ __forceinline__ int qh_run(const QhArgs &a, QhLdsT<FID, TAB> &L, const int64_t f, const int64_t slot, int &nrows) {
    typedef QhFacetT<FID> QhFacet;
    typedef Sg<G> S_;
    const int lane = S_::sl();
    const QhPlan P = qh_plan(a.cap_pts, sizeof(FID) == 4);
    char *ws = a.ws + (size_t)slot * a.ws_stride;
    double4 *PT = reinterpret_cast<double4 *>(ws + P.pts);
    const QhCoord X{PT, 0}, Y{PT, 1}, Z{PT, 2};
    QhFacet *fac = reinterpret_cast<QhFacet *>(ws + P.fac);
    uint16_t *arena = reinterpret_cast<uint16_t *>(ws + P.arena);
    uint16_t *TT = reinterpret_cast<uint16_t *>(ws + P.tt);
    double *DD = reinterpret_cast<double *>(ws + P.dd);
    const int64_t off = a.pts_off[f];
    const int cnt = a.pts_cnt[f];
    int why = QH_OK, n = 0;
    nrows = 0;
    QhConst K;
    int nfac = 0;                 // facets created (ids 1 .. nfac)
    uint32_t atop = 0;            // arena bump pointer
    int perm0 = 1;                // the initial facet moved to the head of the list (qh_furthestnext)

    // ---- 0. the sites: kept points compacted, lifted; the point 'at infinity'; extremes ----
    for (int base = 0; base < cnt; base += G) {
        const int i = base + lane;
        bool k = i < cnt;
        if (k && a.keep) k = a.keep[off + i] >= 0;
        const uint64_t m = S_::ballot(k);
        if (k) {
            const int r = n + popc64(m & S_::below());
            if (r < a.cap_pts - 1) {
                const double x = a.u[off + i], y = a.v[off + i];
                X[r] = x; Y[r] = y; Z[r] = x * x + y * y;
            }
        }
        n += popc64(m);
    }
    n = S_::uni(n);
    if (a.n_used) { if (lane == 0) a.n_used[f] = n; }
    if (n < 3 || n > a.cap_pts - 1 || n > (sizeof(FID) == 4 ? kQhMaxPointsWide : kQhMaxPoints)) return QH_FEW_POINTS;
    qh_mem_sync();
    {
        // sums in input order (the point at infinity sits over the mean), the largest lifted height
        double sx = 0.0, sy = 0.0, mz = -kQhHuge;
        for (int base = 0; base < n; base += G) {
            const int i = base + lane;
            const double x = i < n ? X[i] : 0.0, y = i < n ? Y[i] : 0.0, z = i < n ? Z[i] : -kQhHuge;
            const int c = min(G, n - base);
            for (int j = 0; j < c; ++j) { sx += S_::bcast_d(x, j); sy += S_::bcast_d(y, j); }
            mz = fmax(mz, S_::max_d(z));
        }
        if (lane == 0) { X[n] = sx / (double)n; Y[n] = sy / (double)n; Z[n] = mz * 1.1; }
    }
    qh_mem_sync();
    int ext[6];
    double maxabs = 0.0, maxwidth = 0.0, maxsum = 0.0, zlow = 0.0, zhigh = 0.0, nz0 = 0.0, nz1 = 0.0, nz2 = 0.0;
    {
        const int m1 = n + 1;
        for (int k = 0; k < 3; ++k) {
            const QhCoord C{PT, k};
            double hi = -kQhHuge, lo = kQhHuge; int hii = 0, loi = 0;
            for (int base = 0; base < m1; base += G) {
                const int i = base + lane;
                const double c = i < m1 ? C[i] : 0.0;
                const double cm = S_::max_d(i < m1 ? c : -kQhHuge), cn = S_::min_d(i < m1 ? c : kQhHuge);
                if (cm > hi) { hi = cm; hii = base + ffs64(S_::ballot(i < m1 && c == cm)); }
                if (cn < lo) { lo = cn; loi = base + ffs64(S_::ballot(i < m1 && c == cn)); }
            }
            double maxcoord;
            if (k == 2) { zlow = lo; zhigh = hi; maxcoord = maxabs; }
            else { maxcoord = fmax(hi, -lo); maxwidth = fmax(maxwidth, hi - lo); }
            maxabs = fmax(maxabs, maxcoord);
            maxsum += maxcoord;
            ext[2 * k] = loi; ext[2 * k + 1] = hii;
            if (k == 0) nz0 = 80 * maxsum * kQhEps;
            if (k == 1) nz1 = 80 * maxsum * kQhEps;
            if (k == 2) nz2 = 80 * maxsum * kQhEps;
        }
    }
    if (!(maxwidth > 0.0) || !(zhigh > zlow)) return QH_ZERO_WIDTH;
    {
        // 'Qbb': the lifted coordinate scaled to [0, maxabs]
        const double scale = maxabs / (zhigh - zlow);
        const double shift = 0.0 - zlow * scale;
        for (int i = lane; i <= n; i += G) Z[i] = Z[i] * scale + shift;
        const double maxdistsum = fmin(__builtin_sqrt(3.0) * maxabs, maxsum);
        K.distround = kQhEps * (3 * maxdistsum * 1.01 + maxabs);
        K.anground = 1.01 * 3 * kQhEps;
        K.minvisible = 2 * K.distround;
        K.minoutside = 2 * K.minvisible;
        K.distoutside = 2 * K.minoutside;
        K.guard = 64 * K.distround;
        K.nz0 = nz0; K.nz1 = nz1;
    }
    qh_mem_sync();

    // ---- 1. the initial simplex (qh_maxsimplex over the six extreme points) and its four facets ----
    int simplex[4];
    {
        double maxc = -kQhHuge, minc = kQhHuge; int maxx = -1, minx = -1;
        for (int k = 0; k < 6; ++k) {
            const double x = X[ext[k]];
            if (maxc < x) { maxc = x; maxx = ext[k]; }
            if (minc > x) { minc = x; minx = ext[k]; }
        }
        if (maxx == minx) return QH_ZERO_WIDTH;
        simplex[0] = minx; simplex[1] = maxx;
        double maxdet = maxc - minc;
        for (int i = 2; i < 4; ++i) {
            const double prevdet = maxdet;
            int maxpoint = -1; bool maxnear = false;
            maxdet = -1.0;
            for (int k = 0; k < 6; ++k) {
                const int p = ext[k];
                bool in = false;
                for (int j = 0; j < i; ++j) in = in || simplex[j] == p;
                if (in) continue;
                double det; bool near;
                const double ax = X[p], ay = Y[p], az = Z[p];
                if (i == 2) {
                    const double r00 = X[simplex[0]] - ax, r01 = Y[simplex[0]] - ay, r10 = X[simplex[1]] - ax, r11 = Y[simplex[1]] - ay;
                    det = r00 * r11 - r01 * r10;
                    near = __builtin_fabs(det) < 10 * nz1;
                } else {
                    const double a1 = X[simplex[0]] - ax, a2 = Y[simplex[0]] - ay, a3 = Z[simplex[0]] - az;
                    const double b1 = X[simplex[1]] - ax, b2 = Y[simplex[1]] - ay, b3 = Z[simplex[1]] - az;
                    const double c1 = X[simplex[2]] - ax, c2 = Y[simplex[2]] - ay, c3 = Z[simplex[2]] - az;
                    det = a1 * (b2 * c3 - b3 * c2) - b1 * (a2 * c3 - a3 * c2) + c1 * (a2 * b3 - a3 * b2);
                    near = __builtin_fabs(det) < 10 * nz2;
                }
                det = __builtin_fabs(det);
                if (det > maxdet) { maxdet = det; maxpoint = p; maxnear = near; }
            }
            const double targetdet = prevdet * maxwidth;
            if (maxpoint < 0 || maxnear || (maxdet > 0.0 && maxdet / targetdet < 0.02)) return QH_SIMPLEX_SEARCH;
            simplex[i] = maxpoint;
        }
    }
    {
        // vertices v1..v4 = simplex[0..3]; facet i omits the i-th of (v4, v3, v2, v1); orientation alternates from 'top'
        const int vs[4] = {simplex[3], simplex[2], simplex[1], simplex[0]};
        double cx = 0.0, cy = 0.0, cz = 0.0;
        for (int k = 0; k < 4; ++k) { cx += X[vs[k]]; cy += Y[vs[k]]; cz += Z[vs[k]]; }
        cx = cx / 4; cy = cy / 4; cz = cz / 4;
        bool flip = false;
        {
            const QhPlane F0 = qh_plane(X[vs[1]], Y[vs[1]], Z[vs[1]], X[vs[2]], Y[vs[2]], Z[vs[2]], X[vs[3]], Y[vs[3]], Z[vs[3]], true, K.distround, K.anground, K.nz0, K.nz1);
            if (F0.gauss) return QH_GAUSS;
            const double d = F0.d + cx * F0.n0 + cy * F0.n1 + cz * F0.n2;
            if (d > K.distround) flip = true;
            else if (d > -K.distround) return QH_FLAT_SIMPLEX;
        }
        bool gauss = false;
        if (lane < 4) {
            int q[3], k = 0;
            for (int j = 0; j < 4; ++j) if (j != lane) q[k++] = vs[j];
            const bool top = ((lane & 1) == 0) != flip;
            const QhPlane F = qh_plane(X[q[0]], Y[q[0]], Z[q[0]], X[q[1]], Y[q[1]], Z[q[1]], X[q[2]], Y[q[2]], Z[q[2]], top, K.distround, K.anground, K.nz0, K.nz1);
            gauss = F.gauss;
            L.npl[lane][0] = F.n0; L.npl[lane][1] = F.n1; L.npl[lane][2] = F.n2; L.npl[lane][3] = F.d;
            L.nnb[lane][3] = F.upper ? 1 : 0;
            QhFacet Gf;
            Gf.p[0] = (uint16_t)q[0]; Gf.p[1] = (uint16_t)q[1]; Gf.p[2] = (uint16_t)q[2];
            Gf.flags = (uint16_t)((top ? 1 : 0) | (F.upper ? 2 : 0));
            k = 0;
            for (int j = 0; j < 4; ++j) if (j != lane) Gf.nb[k++] = (FID)(j + 1);
            Gf.bestp = kQhNone; Gf.off = 0; Gf.cnt = 0; Gf.mark = 0; Gf.bestd = 0.0;
            Gf.n0 = F.n0; Gf.n1 = F.n1; Gf.n2 = F.n2; Gf.d = F.d;
            fac[lane + 1] = Gf;
            L.t_total[lane] = 0;
        }
        if (S_::any(gauss)) return QH_GAUSS;
        nfac = 4;
        qh_mem_sync();
        // narrow initial simplex: Qhull switches to another furthest-point rule
        if (lane < 4) {
            double mina = 2.0;
            for (int j = 0; j < 4; ++j) if (j != lane)
                mina = fmin(mina, L.npl[lane][0] * L.npl[j][0] + L.npl[lane][1] * L.npl[j][1] + L.npl[lane][2] * L.npl[j][2]);
            gauss = mina < -0.99999999;
        }
        if (S_::any(gauss)) return QH_NARROW;
    }
    qh_mem_sync();

    // ---- 2. qh_partitionall: every other point to the first initial facet it is 2 MINoutside above ----
    {
        const int total = n + 1;
        bool bad = false, inside = false;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                if (lane < 4) {
                    uint32_t o = 0;
                    for (int j = 0; j < lane; ++j) o += L.t_total[j];
                    L.t_off[lane] = o; L.t_cnt[lane] = 0; L.t_bestp[lane] = kQhNone; L.t_bestd[lane] = -kQhHuge;
                }
                atop = L.t_total[0] + L.t_total[1] + L.t_total[2] + L.t_total[3];
                qh_mem_sync();
            }
            for (int base = 0; base < total; base += G) {
                const int p = base + lane;
                bool valid = p < total && p != simplex[0] && p != simplex[1] && p != simplex[2] && p != simplex[3];
                int tgt = 0; double d = 0.0;
                if (valid) {
                    if (pass == 0) {
                        const double x = X[p], y = Y[p], z = Z[p];
                        tgt = -1;
                        for (int j = 0; j < 4 && tgt < 0; ++j) {
                            d = qh_dist(x, y, z, L.npl[j][0], L.npl[j][1], L.npl[j][2], L.npl[j][3]);
                            if (d >= K.distoutside) tgt = j;
                            else if (d > -K.guard) bad = true;
                        }
                        if (d < K.guard && tgt >= 0) bad = true;
                        if (tgt < 0) { inside = true; tgt = 0; }
                        TT[p] = (uint16_t)tgt; DD[p] = d;
                        atomicAdd(&L.t_total[tgt], 1u);
                    } else { tgt = TT[p]; d = DD[p]; }
                }
                if (pass == 1) qh_place_chunk<G>(L, arena, min(G, total - base), 4, valid ? tgt : -1, p, d);
            }
            qh_mem_sync();
            if (S_::any(inside)) return QH_INSIDE_SIMPLEX;
            if (S_::any(bad)) return QH_BAND;
        }
        if (lane < 4) {
            fac[lane + 1].off = L.t_off[lane]; fac[lane + 1].cnt = (uint16_t)L.t_cnt[lane];
            fac[lane + 1].bestp = L.t_bestp[lane]; fac[lane + 1].bestd = L.t_bestd[lane];
        }
        // qh_furthestnext: the facet with the furthest point of all goes to the head of the list
        double bd = -kQhHuge; perm0 = 0;
        for (int j = 0; j < 4; ++j) if (L.t_bestp[j] != kQhNone && L.t_bestd[j] > bd) { bd = L.t_bestd[j]; perm0 = j + 1; }
        qh_mem_sync();
    }

    // ---- 3. the insertions ----
    {
        // list position -> facet id: perm0 first, then the other initial facets in order, then creation order
        auto pos2id = [&](int pos) { return pos >= 4 || perm0 == 0 ? pos + 1 : (pos == 0 ? perm0 : (pos < perm0 ? pos : pos + 1)); };
        int pos = 0, step = 0;
#ifdef MVOSR_QH_STAMPS
        unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // 0..6 sections; 8..10 / 11..13: partition + placement clocks / insertions with S = 0, 1..64, more
        unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
        while (true) {
            // (a) the first facet in list order with an outside set (its record's first half comes along)
            int cur = -1;
            QhHead<FID> H;
            H.flags = 0; H.bestp = kQhNone; H.nb0 = H.nb1 = H.nb2 = H.off = H.cnt = 0;
            while (pos < nfac) {
                // (the next facet with an outside set is 3 positions on in the median, within 16 in 94 % of the insertions: a window of
                // 16 records — 512 bytes — instead of 64)
                const int id = (lane < kQhPickWindow && pos + lane < nfac) ? pos2id(pos + lane) : 0;
                uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
                if (id) { const uint4 *Gp = reinterpret_cast<const uint4 *>(&fac[id]); r0 = Gp[0]; r1 = Gp[1]; }
                const QhHead<FID> h = qh_head(static_cast<const QhFacet *>(nullptr), r0, r1);
                const bool has = id && !(h.flags & 4u) && h.bestp != kQhNone;
                const uint64_t hm = S_::ballot(has);
                if (hm) {
                    const int l = ffs64(hm);
                    pos += l; cur = pos2id(pos);
                    H.bestp = S_::shfl(h.bestp, l); H.nb0 = S_::shfl(h.nb0, l); H.nb1 = S_::shfl(h.nb1, l); H.nb2 = S_::shfl(h.nb2, l);
                    H.off = S_::shfl(h.off, l); H.cnt = S_::shfl(h.cnt, l);
                    break;
                }
                pos += kQhPickWindow;
            }
            if (cur < 0) break;
            cur = S_::uni(cur);
            ++step;
            const int p = (int)H.bestp;
            if (a.order_out) { if (lane == 0 && p < n) a.order_out[off + p] = step; }     // (compacted ids when `keep` is given)
            QH_STAMP(0);
            // (b) qh_findhorizon: visible facets, breadth first, neighbours in order.  Facets tested in this insertion are
            // remembered in LDS (visible ones with their outside sets, horizon ones with their vertices and neighbours)
            int nvis = 1, head = 0, nhz = 0;
            if (lane == 0) {
                L.visq[0] = (FID)cur; L.visnb[0][0] = (FID)H.nb0; L.visnb[0][1] = (FID)H.nb1; L.visnb[0][2] = (FID)H.nb2;
                L.visoff[0] = H.off; L.viscnt[0] = (uint16_t)H.cnt; L.visbest[0] = kQhNone;
            }
            qh_lds_sync();
            const double px = X[p], py = Y[p], pz = Z[p];        // (in flight together with the first round's facet records)
            bool bad = false, copl = false;
            while (head < nvis) {
                const int ne = min(nvis - head, G / 4);            // four lanes per visible facet: neighbours 0..2 (the fourth lane idles)
                const int e = head + (lane >> 2), k = lane & 3;
                const bool act = (lane >> 2) < ne && k < 3;
                int g = 0;
                QhFacet Gf;
                bool cand = false;
                if (act) {
                    g = L.visnb[e][k];
                    cand = true;
                    for (int i = 0; i < nvis; ++i) if (L.visq[i] == g) cand = false;
                    for (int i = 0; i < nhz; ++i) if (L.hzq[i] == g) cand = false;
                    if (cand) Gf = fac[g];
                }
                // the same facet reached twice in this round: the earlier (entry, neighbour) pair tests it
                const uint64_t cm = S_::ballot(cand);
                bool dup = false;
                for (uint64_t r = cm; r; r &= r - 1) {
                    const int j = ffs64(r);
                    const int gj = S_::shfl(g, j);
                    if (cand && lane > j && gj == g) dup = true;
                }
                bool vis = false, hzn = false;
                if (cand && !dup) {
                    const double d = qh_dist(px, py, pz, Gf.n0, Gf.n1, Gf.n2, Gf.d);
                    if (d > K.minvisible) { vis = true; if (d < K.guard) bad = true; }
                    else { hzn = true; if (d >= -K.guard) copl = true; }
                }
                const uint64_t vm = S_::ballot(vis), zm = S_::ballot(hzn);
                const int add = popc64(vm), addz = popc64(zm);
                if (nvis + add > TAB || nhz + addz > TAB) { why = QH_TOO_MANY_VISIBLE; break; }
                if (vis) {
                    const int q = nvis + popc64(vm & S_::below());
                    L.visq[q] = (FID)g; L.visnb[q][0] = Gf.nb[0]; L.visnb[q][1] = Gf.nb[1]; L.visnb[q][2] = Gf.nb[2];
                    L.visoff[q] = Gf.off; L.viscnt[q] = Gf.cnt; L.visbest[q] = Gf.bestp;
                }
                if (hzn) {
                    const int q = nhz + popc64(zm & S_::below());
                    L.hzq[q] = (FID)g;
                    L.hzr[q][0] = Gf.p[0]; L.hzr[q][1] = Gf.p[1]; L.hzr[q][2] = Gf.p[2]; L.hzr[q][3] = Gf.flags;
                    L.hzr[q][4] = Gf.nb[0]; L.hzr[q][5] = Gf.nb[1]; L.hzr[q][6] = Gf.nb[2];
                }
                head += ne; nvis += add; nhz += addz;
                qh_lds_sync();
            }
            if (why) return why;
            if (S_::any(copl)) return QH_COPLANAR_HORIZON;
            if (S_::any(bad)) return QH_BAND;
            QH_STAMP(1);
            // The visible facets' points (what (e) partitions) are known as soon as the visible list is: their ids (a load from the
            // arena) and then their coordinates (a dependent load) are two of an insertion's ~8 memory round trips.  The first chunk's
            // ids can be asked for HERE, under the cone's own loads, and their coordinates before the matching, which is LDS work only:
            // (e) then starts with its operands in registers.  MEASURED (round 6, same box, -DMVOSR_QH_PREFETCH against without): 34.75 /
            // 34.94 against 34.86 / 34.84 ms per launch of 4 096 sets, 126.41 / 125.74 against 126.56 / 126.04 for 16 384 — nothing, like
            // round 5's variant that only touched the sectors (LABNOTES §9.15): at four wavefronts per SIMD a wavefront's own round trips
            // are hidden behind the others' instructions already.  Off by default; the prefix sum alone stays here.
            int S = 0;
            {
                if (lane == 0) {
                    uint32_t c = 0;
                    for (int e = 0; e < nvis; ++e) { L.viscum[e] = c; c += L.viscnt[e] + (L.visbest[e] != kQhNone ? 1u : 0u); }
                    L.viscum[nvis] = c;
                }
                qh_lds_sync();
                S = (int)L.viscum[nvis];
            }
            if (atop + (uint32_t)S > P.acap) return QH_ARENA_FULL;
#ifdef MVOSR_QH_PREFETCH
            int q_pre = 0, e_pre = 0;
            if (lane < S) {
                while (e_pre + 1 < nvis && (int)L.viscum[e_pre + 1] <= lane) ++e_pre;
                const int r = lane - (int)L.viscum[e_pre];
                q_pre = r < (int)L.viscnt[e_pre] ? arena[L.visoff[e_pre] + r] : L.visbest[e_pre];
            }
#endif
            // (c) qh_makenewfacets: for each visible facet in order, for each horizon neighbour in order, a facet (apex first)
            int m = 0;
            for (int e = lane; e < nvis; e += G) L.visrep[e] = kQhNone;
            qh_lds_sync();
            bool gauss = false, notconv = false;
            for (int base = 0; base < nvis; base += G / 4) {
                const int ne = min(nvis - base, G / 4);
                const int e = base + (lane >> 2), k = lane & 3;
                const bool act = (lane >> 2) < ne && k < 3;
                int g = 0, hi = -1;
                if (act) {
                    g = L.visnb[e][k];
                    for (int i = 0; i < nhz; ++i) if (L.hzq[i] == g) hi = i;
                }
                const bool hz = hi >= 0;
                const uint64_t hm = S_::ballot(hz);
                const int j = m + popc64(hm & S_::below());
                const int add = popc64(hm);
                if (m + add > TAB) { why = QH_CONE_TOO_LARGE; break; }
                if (nfac + m + add > (int)P.fcap) { why = QH_FACETS_FULL; break; }
                if (hz) {
                    const int vid = L.visq[e];
                    const int g0 = L.hzr[hi][0], g1 = L.hzr[hi][1], g2 = L.hzr[hi][2], gf = L.hzr[hi][3];
                    const int skip = L.hzr[hi][4] == vid ? 0 : (L.hzr[hi][5] == vid ? 1 : 2);
                    const int va = skip == 0 ? g1 : g0, vb = skip == 2 ? g1 : g2, vo = skip == 0 ? g0 : (skip == 1 ? g1 : g2);
                    const bool gtop = gf & 1;
                    const bool top = gtop ? (skip & 1) : !(skip & 1);
                    const double ax = X[va], ay = Y[va], az = Z[va], bx = X[vb], by = Y[vb], bz = Z[vb];
                    const double ox = X[vo], oy = Y[vo], oz = Z[vo];
                    const QhPlane F = qh_plane(px, py, pz, ax, ay, az, bx, by, bz, top, K.distround, K.anground, K.nz0, K.nz1);
                    gauss = gauss || F.gauss;
                    // the horizon facet's vertex opposite the shared ridge must lie below the cone facet (else Qhull merges)
                    if (qh_dist(ox, oy, oz, F.n0, F.n1, F.n2, F.d) > -K.guard) notconv = true;
                    L.npl[j][0] = F.n0; L.npl[j][1] = F.n1; L.npl[j][2] = F.n2; L.npl[j][3] = F.d;
                    L.nxy[j][0] = ax; L.nxy[j][1] = ay; L.nxy[j][2] = az; L.nxy[j][3] = bx; L.nxy[j][4] = by; L.nxy[j][5] = bz;
                    L.nva[j] = (uint16_t)va; L.nvb[j] = (uint16_t)vb;
                    L.nnb[j][0] = (uint8_t)(top ? 1 : 0); L.nnb[j][3] = F.upper ? 1 : 0;
                    L.t_total[j] = 0;
                    L.nhz[j] = (FID)g;
                    fac[g].nb[skip] = (FID)(nfac + 1 + j);
                    // a visible facet's replacement: the last cone facet made from it
                    const uint64_t mine = hm & (15ull << (lane & ~3));
                    if ((63 - __clzll((long long)mine)) == lane) L.visrep[e] = (uint16_t)j;
                }
                m += add;
            }
            if (why) return why;
            qh_lds_sync();
            if (S_::any(gauss)) return QH_GAUSS;
            if (S_::any(notconv)) return QH_NOT_CONVEX;
            if (m < 3) return QH_OPEN_CONE;
#ifdef MVOSR_QH_PREFETCH
            double x_pre = 0.0, y_pre = 0.0, z_pre = 0.0;
            if (lane < S) { x_pre = X[q_pre]; y_pre = Y[q_pre]; z_pre = Z[q_pre]; }
#endif
            QH_STAMP(2);
            // (d) qh_matchnewfacets: neighbour 1 shares the ridge {apex, b}, neighbour 2 the ridge {apex, a};
            //     qh_sharpnewfacets: the cone's normals in more than one orthant
            bool sharp;
            {
                bool open = false, diff = false; notconv = false;
                const int q0 = (L.npl[0][0] > 0 ? 1 : 0) | (L.npl[0][1] > 0 ? 2 : 0) | (L.npl[0][2] > 0 ? 4 : 0);
                for (int jb = 0; jb < m; jb += G) {
                    const int j = jb + lane;
                    if (j < m) {
                        const int va = L.nva[j], vb = L.nvb[j];
                        int n1 = -1, n2 = -1, c1 = 0, c2 = 0;
                        for (int i = 0; i < m; ++i) if (i != j) {
                            const int ia = L.nva[i], ib = L.nvb[i];
                            if (ia == vb || ib == vb) { n1 = i; ++c1; }
                            if (ia == va || ib == va) { n2 = i; ++c2; }
                        }
                        if (c1 != 1 || c2 != 1) open = true;
                        else {
                            // convexity between cone facets: the neighbour's other horizon vertex lies below this facet
                            const int o1 = L.nva[n1] == vb ? 3 : 0, o2 = L.nva[n2] == va ? 3 : 0;
                            const double d1 = qh_dist(L.nxy[n1][o1], L.nxy[n1][o1 + 1], L.nxy[n1][o1 + 2], L.npl[j][0], L.npl[j][1], L.npl[j][2], L.npl[j][3]);
                            const double d2 = qh_dist(L.nxy[n2][o2], L.nxy[n2][o2 + 1], L.nxy[n2][o2 + 2], L.npl[j][0], L.npl[j][1], L.npl[j][2], L.npl[j][3]);
                            if (d1 > -K.guard || d2 > -K.guard) notconv = true;
                            L.nnb[j][1] = (uint8_t)n1; L.nnb[j][2] = (uint8_t)n2;
                        }
                        if (((L.npl[j][0] > 0 ? 1 : 0) | (L.npl[j][1] > 0 ? 2 : 0) | (L.npl[j][2] > 0 ? 4 : 0)) != q0) diff = true;
                    }
                }
                if (S_::any(open)) return QH_OPEN_CONE;
                if (S_::any(notconv)) return QH_NOT_CONVEX;
                sharp = S_::any(diff);
                qh_lds_sync();
            }
            QH_STAMP(3);
#ifdef MVOSR_QH_STAMPS
            const unsigned long long t_part0 = t_last;
#endif
            // (e) qh_partitionvisible: the visible facets' points, in list order, to the cone
            {
                // targets: a directed walk per point; from the first point that ends below its best facet on a sharp cone,
                // a linear scan (qh.findbestnew stays set for the rest of this insertion)
                bool mode = false, fail_band = false, fail_sharp = false, fail_none = false;
                int q0 = 0, tgt0 = 0; double d0 = 0.0;
                for (int base = 0; base < S; base += G) {
                    const int i = base + lane;
                    const bool valid = i < S;
                    int q = 0, start = 0;
                    double x = 0.0, y = 0.0, z = 0.0;
#ifdef MVOSR_QH_PREFETCH
                    if (valid && base == 0) {
                        q = q_pre; x = x_pre; y = y_pre; z = z_pre;
                        start = L.visrep[e_pre] == kQhNone ? 0 : L.visrep[e_pre];
                    } else
#endif
                    if (valid) {
                        int e = 0;
                        while (e + 1 < nvis && (int)L.viscum[e + 1] <= i) ++e;
                        const int r = i - (int)L.viscum[e];
                        q = r < (int)L.viscnt[e] ? arena[L.visoff[e] + r] : L.visbest[e];
                        start = L.visrep[e] == kQhNone ? 0 : L.visrep[e];
                        x = X[q]; y = Y[q]; z = Z[q];
                    }
                    QhWalk R; R.state = 0; R.tgt = 0; R.d = 0.0; R.best = -1;
                    if (valid) R = mode ? qh_scan_cone(L, m, start, x, y, z, K) : qh_walk_cone(L, start, x, y, z, K);
                    if (!mode) {
                        if (valid && R.state == 2) R = qh_scan_cone(L, m, 0, x, y, z, K);       // no best: all cone facets from the first
                        const uint64_t trig = S_::ballot(valid && R.state == 1);
                        if (trig) {
                            if (!sharp) fail_sharp = true;
                            const int t = ffs64(trig);
                            if (valid && lane >= t) R = qh_scan_cone(L, m, lane == t ? R.best : start, x, y, z, K);
                            mode = true;
                        }
                    }
                    if (valid && R.state != 0) { if (R.state == 3) fail_band = true; else fail_none = true; }
                    if (valid) {
                        atomicAdd(&L.t_total[R.tgt], 1u);
                        if (S > G) { TT[i] = (uint16_t)R.tgt; DD[i] = R.d; }
                    }
                    if (base == 0) { q0 = q; tgt0 = R.tgt; d0 = R.d; }
                }
                if (S_::any(fail_band)) return QH_BAND;
                if (fail_sharp) return QH_NOT_SHARP;
                if (S_::any(fail_none)) return QH_ABOVE_NONE;
                qh_lds_sync();
                if (S > G) __threadfence_block();
                QH_STAMP(4);
                // room for the cone's outside sets, then the placement in arrival order
                for (int jb = 0; jb < m; jb += G) {
                    const int j = jb + lane;
                    if (j < m) {
                        uint32_t o = atop;
                        for (int i = 0; i < j; ++i) o += L.t_total[i];
                        L.t_off[j] = o; L.t_cnt[j] = 0; L.t_bestp[j] = kQhNone; L.t_bestd[j] = -kQhHuge;
                    }
                }
                atop += (uint32_t)S;
                qh_lds_sync();
                if (S > 0 && S <= G) qh_place_chunk<G>(L, arena, S, m, tgt0, q0, d0);
                else for (int base = 0; base < S; base += G) {
                    const int i = base + lane;
                    const bool valid = i < S;
                    int q = 0, tgt = 0; double d = 0.0;
                    if (valid) {
                        int e = 0;
                        while (e + 1 < nvis && (int)L.viscum[e + 1] <= i) ++e;
                        const int r = i - (int)L.viscum[e];
                        q = r < (int)L.viscnt[e] ? arena[L.visoff[e] + r] : L.visbest[e];
                        tgt = TT[i]; d = DD[i];
                    }
                    qh_place_chunk<G>(L, arena, min(G, S - base), m, tgt, q, d);
                    qh_lds_sync();
                }
                qh_lds_sync();
            }
            QH_STAMP(5);
#ifdef MVOSR_QH_STAMPS
            { const int c_ = S == 0 ? 0 : (S <= 64 ? 1 : 2); acc[8 + c_] += t_last - t_part0; acc[11 + c_] += 1; }
#endif
            // (f) the cone's facet records; the visible facets die
            for (int jb = 0; jb < m; jb += G) {
                const int j = jb + lane;
                if (j < m) {
                    QhFacet Gf;
                    Gf.p[0] = (uint16_t)p; Gf.p[1] = L.nva[j]; Gf.p[2] = L.nvb[j];
                    Gf.flags = (uint16_t)((L.nnb[j][0] ? 1 : 0) | (L.nnb[j][3] ? 2 : 0));
                    Gf.nb[0] = L.nhz[j];
                    Gf.nb[1] = (FID)(nfac + 1 + L.nnb[j][1]); Gf.nb[2] = (FID)(nfac + 1 + L.nnb[j][2]);
                    Gf.bestp = L.t_bestp[j]; Gf.off = L.t_off[j]; Gf.cnt = (uint16_t)L.t_cnt[j]; Gf.mark = 0; Gf.bestd = L.t_bestd[j];
                    Gf.n0 = L.npl[j][0]; Gf.n1 = L.npl[j][1]; Gf.n2 = L.npl[j][2]; Gf.d = L.npl[j][3];
                    fac[nfac + 1 + j] = Gf;
                }
            }
            for (int e = lane; e < nvis; e += G) fac[L.visq[e]].flags |= 4;
            nfac += m;
            qh_mem_sync();
            QH_STAMP(6);
        }
#ifdef MVOSR_QH_STAMPS
        if (a.stamps && f == 0 && lane == 0) { for (int k = 0; k < 8; ++k) a.stamps[k] = acc[k]; a.stamps[8] = (unsigned long long)step; for (int k = 8; k < 14; ++k) a.stamps[k + 8] = acc[k]; }
#endif
        // ---- 4. SciPy's rows: lower facets in list order; vertices by decreasing vertex id, first two swapped unless top ----
        const int64_t toff = a.tri_off[f];
        for (int base = 0; base < nfac; base += G) {
            const int id = base + lane < nfac ? pos2id(base + lane) : 0;
            bool row = false; QhFacet Gf;
            if (id) { Gf = fac[id]; row = !(Gf.flags & 4) && !(Gf.flags & 2); }
            const uint64_t rm = S_::ballot(row);
            if (row) {
                int32_t *t = a.tri + 3 * (toff + nrows + popc64(rm & S_::below()));
                const bool top = Gf.flags & 1;
                t[0] = top ? Gf.p[0] : Gf.p[1]; t[1] = top ? Gf.p[1] : Gf.p[0]; t[2] = Gf.p[2];
            }
            nrows += popc64(rm);
        }
    }
    return why;
}

// G = 64 (one frame per wavefront): 119 registers and 9.7 KB of LDS, four wavefronts per SIMD.  (Compiled for five, six and eight
// — 96 / 80 / 64 registers with spills, the LDS tables halved — the same launch of 4096 frames took 41 / 50 / 66 ms instead of
// 34: LABNOTES §9.)  G = 32 / 16 (two / four frames per wavefront, 32-entry tables): a frame whose insertion overflows a table goes
// to the `redo` list, which the G = 64 list kernel walks next.
template <typename FID, int G, int TAB, bool LIST>
__global__ __launch_bounds__(64, (G == 16 ? 2 : 4)) void qhull_rows_kernel(const QhArgs a) {
    __shared__ QhLdsT<FID, TAB> Ls[64 / G];
    QhLdsT<FID, TAB> &L = Ls[Sg<G>::sub()];
    if constexpr (LIST) {
        static_assert(G == 64, "the list walk is one frame per wavefront");
        // the frames of a list whose length is known on the device only: a persistent grid, one workspace slice per workgroup
        const int64_t todo = (int64_t)a.list[0];
        for (int64_t it = blockIdx.x; it < todo; it += gridDim.x) {
            const int64_t f = (int64_t)a.list[1 + it];
            if (f < 0 || f >= a.n_frames) continue;
            int nrows = 0;
            __syncthreads();
            const int why = qh_run<FID, G, TAB>(a, L, f, (int64_t)blockIdx.x, nrows);
            if (lane_id() == 0) {
                if (why) { a.tri_cnt[f] = 0; a.status[f] = MVOSR_DT_DEGENERATE | (why << 8); }
                else a.tri_cnt[f] = nrows;           // (status stays what the first attempt left: 0)
            }
            __threadfence_block();
        }
        return;
    } else {
        int64_t f = (int64_t)blockIdx.x * (64 / G) + Sg<G>::sub();
        if (f >= a.n_frames) return;
        if (G == 64 && a.launch_order) f = (int64_t)a.launch_order[f];
        int nrows = 0;
        const int why = qh_run<FID, G, TAB>(a, L, f, f, nrows);
        if (Sg<G>::sl() == 0) {
            if (G < 64 && a.redo && (why == QH_TOO_MANY_VISIBLE || why == QH_CONE_TOO_LARGE)) {
                const int k = atomicAdd(&a.redo[0], 1);
                a.redo[1 + k] = (int32_t)f;
                a.tri_cnt[f] = 0; a.status[f] = MVOSR_DT_OK;
            } else {
                a.tri_cnt[f] = why ? 0 : nrows;
                a.status[f] = why ? (MVOSR_DT_DEGENERATE | (why << 8)) : MVOSR_DT_OK;
            }
        }
    }
}

// Launch order of a ragged batch: a replay is a chain of ~n insertions, so a wavefront's life is proportional to its frame's points,
// and a launch of more frames than the machine holds wavefronts (16 per CU) ends when its LAST-started long frame ends.  Workgroups are
// dispatched in index order: frames sorted by size, largest first (64 size classes, one counting sort by one workgroup; the order inside
// a class is whatever the atomics give — every frame's rows are its own), leave the short frames for the tail.
constexpr int kQhOrderThreads = 1024, kQhOrderClasses = 64;
__global__ __launch_bounds__(kQhOrderThreads) void qh_order_kernel(const int32_t *pts_cnt, int64_t n_frames, int max_pts, int32_t *order) {
    __shared__ int cnt[kQhOrderClasses], base[kQhOrderClasses];
    const int tid = threadIdx.x;
    if (tid < kQhOrderClasses) cnt[tid] = 0;
    __syncthreads();
    auto cls = [&](int n) { const int c = (int)(((int64_t)max(n, 0) * kQhOrderClasses) / ((int64_t)max_pts + 1)); return kQhOrderClasses - 1 - min(c, kQhOrderClasses - 1); };
    for (int64_t f = tid; f < n_frames; f += kQhOrderThreads) atomicAdd(&cnt[cls(pts_cnt[f])], 1);
    __syncthreads();
    if (tid == 0) { int b = 0; for (int c = 0; c < kQhOrderClasses; ++c) { base[c] = b; b += cnt[c]; } }
    __syncthreads();
    for (int64_t f = tid; f < n_frames; f += kQhOrderThreads) order[atomicAdd(&base[cls(pts_cnt[f])], 1)] = (int32_t)f;
}

// Lanes per frame of the product launch: 64.  The packed instantiations (two / four frames per wavefront) are measured A/B
// variants of builds with -DMVOSR_ABLATE (env MVOSR_QH_GROUP = 32 | 16): rows identical, but a packed wavefront's insertion is
// the LONGEST of its frames' steps (52.7 k clocks for four frames against 27.5 k for one, alone on a SIMD) and 20 KB of LDS
// leave two wavefronts per SIMD: 113 k sets/s (G = 16), 103 k (G = 32) against 129 k (LABNOTES §9.8).
// MVOSR_QH_NO_ORDER=1 (diagnostic builds, -DMVOSR_ABLATE): frames in index order, as before round 6
bool qh_no_order() {
#ifdef MVOSR_ABLATE
    static const bool v = [] { const char *e = getenv("MVOSR_QH_NO_ORDER"); return e && e[0] == '1'; }();
    return v;
#else
    return false;
#endif
}

int qh_group() {
#ifdef MVOSR_ABLATE
    static const int g = [] {
        const char *e = getenv("MVOSR_QH_GROUP");
        const int v = e ? atoi(e) : 0;
        return v == 32 || v == 16 ? v : 64;
    }();
    return g;
#else
    return 64;
#endif
}

}  // namespace

#ifdef MVOSR_QH_STAMPS
extern "C" void mvosr_debug_qh_stamps(void *dptr) { g_qh_stamps = reinterpret_cast<unsigned long long *>(dptr); }
#endif

extern "C" int mvosr_delaunay_qhull_max_points(void) { return kQhMaxPointsWide; }

static int qh_launch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt, const double *u, const double *v,
                     const int32_t *keep, int max_pts, const int64_t *tri_off, int32_t *tri, int32_t *tri_cnt, int32_t *n_used,
                     int32_t *status, int32_t *order_out, const int32_t *list, int list_blocks) {
    if (!ctx || !pts_off || !pts_cnt || !u || !v || !tri_off || !tri || !tri_cnt || !status)
        return set_error(MVOSR_ERR_ARG, "delaunay_qhull_batch: null argument");
    if (max_pts < 0) return set_error(MVOSR_ERR_ARG, "delaunay_qhull_batch: max_pts < 0");
    if (n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    if (max_pts < 3) max_pts = 3;
    if (max_pts > kQhMaxPointsWide)
        return set_error(MVOSR_ERR_TOO_LARGE, "delaunay_qhull_batch: %d points per frame (limit: %d)", max_pts, kQhMaxPointsWide);
    const bool wide = max_pts > kQhMaxPoints;
    QhArgs a;
    a.n_frames = n_frames; a.pts_off = pts_off; a.pts_cnt = pts_cnt; a.u = u; a.v = v; a.keep = keep; a.tri_off = tri_off; a.tri = tri;
    a.tri_cnt = tri_cnt; a.n_used = n_used; a.status = status; a.order_out = order_out; a.list = list; a.redo = nullptr;
    a.launch_order = nullptr;
    a.cap_pts = max_pts + 1;
#ifdef MVOSR_QH_STAMPS
    a.stamps = g_qh_stamps;
#else
    a.stamps = nullptr;
#endif
    const QhPlan P = qh_plan(a.cap_pts, wide);
    a.ws_stride = P.total;
    const int64_t slices = list ? (int64_t)(list_blocks < n_frames ? list_blocks : n_frames) : n_frames;
    const int group = (list || wide) ? 64 : qh_group();
    const size_t redo_bytes = group < 64 ? (((size_t)n_frames + 1) * sizeof(int32_t) + 255) & ~(size_t)255 : 0;
    // (a launch that holds more frames than the machine holds wavefronts — 16 per CU — and is not a list walk: largest frames first)
    const bool ordered = !list && group == 64 && n_frames > (int64_t)16 * ctx->n_cu && n_frames < 0x7fffffff && !qh_no_order();
    const size_t order_bytes = ordered ? ((size_t)n_frames * sizeof(int32_t) + 255) & ~(size_t)255 : 0;
    void *ws = nullptr;
    if ((rc = ctx_workspace_bytes(ctx, (size_t)slices * P.total + redo_bytes + order_bytes, &ws))) return rc;
    a.ws = reinterpret_cast<char *>(ws);
    hipStream_t st = ctx_stream(ctx);
    if (ordered) {
        int32_t *ord = reinterpret_cast<int32_t *>(a.ws + (size_t)slices * P.total + redo_bytes);
        hipLaunchKernelGGL(qh_order_kernel, dim3(1), dim3(kQhOrderThreads), 0, st, pts_cnt, n_frames, max_pts, ord);
        if ((rc = check_launch("qh_order_kernel"))) return rc;
        a.launch_order = ord;
    }
    // (the list walk is an instantiation of its own: the loop around the run cost the product kernel registers — spills in its hot loop)
    if (list) {
        if (wide) hipLaunchKernelGGL((qhull_rows_kernel<uint32_t, 64, 64, true>), dim3((unsigned)slices), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((qhull_rows_kernel<uint16_t, 64, 64, true>), dim3((unsigned)slices), dim3(64), 0, st, a);
    } else if (wide) hipLaunchKernelGGL((qhull_rows_kernel<uint32_t, 64, 64, false>), dim3((unsigned)slices), dim3(64), 0, st, a);
    else if (group == 64) hipLaunchKernelGGL((qhull_rows_kernel<uint16_t, 64, 64, false>), dim3((unsigned)slices), dim3(64), 0, st, a);
#ifdef MVOSR_ABLATE
    else {
        // several frames per wavefront; the frames that overflowed a 32-entry table, next, one per wavefront
        a.redo = reinterpret_cast<int32_t *>(a.ws + (size_t)slices * P.total);
        if (hipMemsetAsync(a.redo, 0, sizeof(int32_t), st) != hipSuccess) return set_error(MVOSR_ERR_HIP, "delaunay_qhull_batch: memset");
        const int per = 64 / group;
        const unsigned blocks = (unsigned)((n_frames + per - 1) / per);
        if (group == 32) hipLaunchKernelGGL((qhull_rows_kernel<uint16_t, 32, 32, false>), dim3(blocks), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((qhull_rows_kernel<uint16_t, 16, 32, false>), dim3(blocks), dim3(64), 0, st, a);
        if ((rc = check_launch("qhull_rows_kernel"))) return rc;
        QhArgs b = a;
        b.list = a.redo; b.redo = nullptr;
        const int64_t rb = n_frames < 256 ? n_frames : 256;
        hipLaunchKernelGGL((qhull_rows_kernel<uint16_t, 64, 64, true>), dim3((unsigned)rb), dim3(64), 0, st, b);
    }
#endif
    return check_launch("qhull_rows_kernel");
}

extern "C" int mvosr_delaunay_qhull_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                          const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                                          int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status, int32_t *order_out) {
    return qh_launch(ctx, n_frames, pts_off, pts_cnt, u, v, keep, max_pts, tri_off, tri, tri_cnt, n_used, status, order_out, nullptr, 0);
}

// The frames of a device-side list (list[0] = how many, list[1..] = frame indices): the exact pass of mvosr_scale_batch asks for
// SciPy's own rows of the frames it redoes when the batch's second triangulation is a stand-in (mvosr_batch.standin_*).
int qh_rows_for_list(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt, const double *u, const double *v,
                     const int32_t *keep, int max_pts, const int64_t *tri_off, int32_t *tri, int32_t *tri_cnt, int32_t *status,
                     const int32_t *list) {
    return qh_launch(ctx, n_frames, pts_off, pts_cnt, u, v, keep, max_pts, tri_off, tri, tri_cnt, nullptr, status, nullptr, list, 1024);
}
