// mvosr_kernels.hip — HIP kernels (CDNA4 / gfx950) for the per-frame scale-recovery hot path and
// the C-ABI launchers declared in include/mvosr.h.
//
// One frame = one workgroup of WAVES wavefronts (WAVES = 1 for small frames: one frame per
// wavefront).  A frame's features are staged once from HBM into LDS in fp64 and every
// per-triangle gather of the three stages is served from LDS:
//
//   phase A  load v, remapped y', z'            (feature_remap, scale_calculator.py:390-394)
//            vote over tri1 with LDS atomics    (find_outliers/check_triangle, :151-167,:105-119)
//            survivors are moved down in place  (feature2d[valid], :264-265), x replacing v
//   phase B  per-triangle plane normal / pitch test / mean height over tri2; block reduction
//            -> height_level; second sweep marks the selected vertices in an LDS bit-set
//                                                (feature_selection_by_tri, :225-248)
//   phase C  169-bin LDS histogram of the selected y', remove_single, mode / local-minimum
//            logic on 64-bit ballots, mean/std/skew, median fallback
//                                                (road_model_calculation_static, :324-354)
//
// LDS per frame: 26 B per feature — P[i] = {v|x, z'} (16 B, one ds_read_b128 per vertex), Y[i] =
// y' (8 B), a 16-bit vote counter (whose space later holds the selected bit-set) — plus ~1.3 KB:
// 53.4 KB at N=2000, three workgroups per CU.  HBM traffic per frame = the inputs once
// (x,y,z,v + tri1 + tri2) and 28 B of results.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mvosr.h"
#include "mvosr_device.hpp"
#include "mvosr_host.hpp"

namespace mvosr {

// Diagnostic build only (-DMVOSR_STAMPS): thread 0 of every workgroup records s_memtime at the
// phase boundaries and dumps the stamps over the frame's `hist` debug output.  Never defined in
// the shipped library.
#ifdef MVOSR_STAMPS
#define MVOSR_STAMP(i) do { if (threadIdx.x == 0 && stamps) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MVOSR_STAMP_DECL unsigned long long stamps[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define MVOSR_STAMP_ARG , unsigned long long *stamps = nullptr
#define MVOSR_STAMP_PASS , stamps
#define MVOSR_RSTAMP(i) do { if (stamps) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MVOSR_RSTAMP(i) do {} while (0)
#define MVOSR_STAMP(i) do {} while (0)
#define MVOSR_STAMP_DECL
#define MVOSR_STAMP_ARG
#define MVOSR_STAMP_PASS
#endif

// ---------------------------------------------------------------------------------------------
// LDS carve-up (byte offsets, all multiples of 16)
// ---------------------------------------------------------------------------------------------
// Experiment (review item 4 ii): an XOR swizzle of the vertex records' slots.  -DMVOSR_PSWZ=k permutes the slots inside
// aligned groups of 8 by bits k.. of the index.  The gathers address the records through triangle rows, i.e. at random
// with respect to the banks, so no fixed permutation changes how many of a ds_read_b128's 16 lanes collide
// (measured: profiles/r03_ab_swizzle.txt); the product build has no swizzle.
// -DMVOSR_PSPLIT keeps the records as two 8-byte planes an odd number of doubles apart instead (LDS-resident variants
// only: an experiment build, not a product one).
#if defined(MVOSR_PSWZ)
#define PSW(i) ((i) ^ ((((unsigned)(i)) >> MVOSR_PSWZ) & 7u))
#define MVOSR_NPAD(n) ((uint32_t)(((n) + 7) & ~7))
#elif defined(MVOSR_PSPLIT)
#define PSW(i) (i)
#define MVOSR_NPAD(n) ((uint32_t)(((n) + 3) & ~1))
#else
#define PSW(i) (i)
#define MVOSR_NPAD(n) ((uint32_t)(((n) + 1) & ~1))
#endif
#ifdef MVOSR_PSPLIT
__device__ __forceinline__ int p_stride(const double2 *P, const double *Y) {
    return (int)((Y - reinterpret_cast<const double *>(P)) >> 1) - 1;      // npad - 1: odd
}
__device__ __forceinline__ double2 p_ld(const double2 *P, const double *Y, int i) {
    const double *d = reinterpret_cast<const double *>(P);
    double2 r; r.x = d[i]; r.y = d[p_stride(P, Y) + i]; return r;
}
__device__ __forceinline__ double p_ldy(const double2 *P, const double *Y, int i) {
    return reinterpret_cast<const double *>(P)[p_stride(P, Y) + i];
}
__device__ __forceinline__ void p_st(double2 *P, const double *Y, int i, double2 v) {
    double *d = reinterpret_cast<double *>(P);
    d[i] = v.x; d[p_stride(P, Y) + i] = v.y;
}
#else
__device__ __forceinline__ double2 p_ld(const double2 *P, const double *, int i) { return P[PSW(i)]; }
__device__ __forceinline__ double p_ldy(const double2 *P, const double *, int i) { return P[PSW(i)].y; }
__device__ __forceinline__ void p_st(double2 *P, const double *, int i, double2 v) { P[PSW(i)] = v; }
#endif

struct LdsPlan {
    uint32_t p, y, c, hist, red, misc, total;
};
__host__ __device__ inline uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }
constexpr int kRedSlots = 6;                       // one scratch slot per block reduction site
enum { R_SEL_H = 0, R_SEL_CNT = 1, R_SEL_ABS = 2, R_ROAD_SUM = 3, R_ROAD_SS = 4, R_MISC = 5 };
__host__ __device__ inline LdsPlan lds_plan(int n, int waves) {
    LdsPlan p;
    const uint32_t npad = MVOSR_NPAD(n);
    p.p = 0;                                      // double2 {v|x, z'} per feature
    p.y = p.p + 16u * npad;                       // y' per feature
    p.c = p.y + 8u * npad;                        // 16-bit vote counter per feature; later the selected bit-set
    p.hist = align16(p.c + 2u * npad + 16u);
    p.red = p.hist + 4u * 176u;                   // 169 bins (+pad)
    p.misc = p.red + 8u * (uint32_t)(kRedSlots * 2 * waves);   // reduction scratch: 2*waves doubles per site
    p.total = p.misc + 4u * 32u;                  // 32 ints of per-frame scalars
    return p;
}

struct Smem {
    double2 *P;
    double *Y;
    uint16_t *c16;
    uint32_t *c32;
    uint32_t *sel;        // aliases the counters once they are consumed
    int *hist;
    double *red;
    int *misc;
};
__device__ __forceinline__ Smem carve(char *base, int n, int waves) {
    const LdsPlan p = lds_plan(n, waves);
    Smem s;
    s.P = reinterpret_cast<double2 *>(base + p.p);
    s.Y = reinterpret_cast<double *>(base + p.y);
    s.c16 = reinterpret_cast<uint16_t *>(base + p.c);
    s.c32 = reinterpret_cast<uint32_t *>(base + p.c);
    s.sel = reinterpret_cast<uint32_t *>(base + p.c);
    s.hist = reinterpret_cast<int *>(base + p.hist);
    s.red = reinterpret_cast<double *>(base + p.red);
    s.misc = reinterpret_cast<int *>(base + p.misc);
    return s;
}

// misc[] slots
enum { M_WCNT = 0 /* [0..15] per-wave survivor counts */, M_LIST = 16, M_MEDLO = 18, M_MEDHI = 20 /* doubles at 18..21 */ };


// ---------------------------------------------------------------------------------------------
// Triangle ids are streamed in chunks of kTC triangles per thread: all loads of a chunk are in
// flight together, and the first chunk of each sweep is issued a phase early (tri1 with the
// feature loads, tri2 before the compaction barriers) so that its latency is off the sweep.
// ---------------------------------------------------------------------------------------------
#ifndef MVOSR_TC
#define MVOSR_TC 2
#endif
constexpr int kTC = MVOSR_TC;   // triangles per thread per chunk (3 VGPRs each)


// Inputs that are read exactly once are loaded non-temporally: measured, the L2-miss traffic of the
// fused kernel drops from 1.23x to 0.99x the algorithmic bytes (the second read of tri2 then hits L2).
typedef double dvec2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 stream_load2(const double2 *p) {
    const dvec2 v = __builtin_nontemporal_load(reinterpret_cast<const dvec2 *>(p));
    double2 r; r.x = v.x; r.y = v.y;
    return r;
}
__device__ __forceinline__ double stream_load(const double *p) {
    return __builtin_nontemporal_load(p);
}

template <int B>
struct TriChunk {
    TriIds q[kTC];
    // STREAM: the rows are not read again (non-temporal loads, so that they do not push rows that are
    // read twice — the second triangulation — out of L2)
    template <bool STREAM = false>
    __device__ __forceinline__ void load(const int32_t *tri, int64_t begin, int count, int base, int tid) {
#pragma unroll
        for (int k = 0; k < kTC; ++k) {
            const int t = base + k * B + tid;
            if (t < count) {
                if constexpr (STREAM) {
                    const int32_t *p = tri + 3 * begin + 3 * t;
                    q[k].a = __builtin_nontemporal_load(p); q[k].b = __builtin_nontemporal_load(p + 1); q[k].c = __builtin_nontemporal_load(p + 2);
                } else {
                    q[k] = load_tri(tri + 3 * begin, t);
                }
            }
        }
    }
};

// The three vertices' flags from check_triangle's three pair tests (:110-118).  fixed == false: the reference's own
// pattern — the (0,2) test marks vertices 0 and 1 (:113-115); true (mvosr_params.vote_mode == MVOSR_VOTE_FIXED): it marks
// 0 and 2, so that a vertex is flagged iff one of its two pair tests fails, whatever the order of the row's vertices.
// `fixed` is uniform: the two extra terms are scalar mask operations.
struct VoteFlags { bool f0, f1, f2; };
__device__ __forceinline__ VoteFlags vote_flags(bool pa, bool pb, bool pc, bool fixed) {
    const bool pb1 = pb & !fixed, pb2 = pb & fixed;
    VoteFlags r;
    r.f0 = pa | pb; r.f1 = pa | pb1 | pc; r.f2 = pb2 | pc;
    return r;
}
// rows of frame f in a triangulation given by offsets (+ optional explicit counts: device-built triangulations)
__device__ __forceinline__ int tri_rows(const int64_t *off, const int32_t *cnt, int64_t f) {
    return cnt ? cnt[f] : (int)(off[f + 1] - off[f]);
}

// ---------------------------------------------------------------------------------------------
// Phase A: stage {v, z'}, y' and run the depth-order vote over the first triangulation; then the
// survivors are moved down in place (order kept) with x taking v's slot, so that the second
// triangulation's vertex ids address LDS directly.  Every wave owns a contiguous slice of SC*64
// features for the compaction.  Returns the number of survivors; `bad` is set when a vertex id
// is out of range.  When tri2 is given, its first chunk is put in flight before the compaction
// barriers and handed back in `next`.
// ---------------------------------------------------------------------------------------------
template <int WAVES, int SC>
__device__ __forceinline__ int phase_vote(const Smem &s, int n, const double *gx, const double *gy, const double *gz,
                                          const double *gv, const int32_t *tri1, int64_t t1_begin, int t1_count,
                                          double cp, double sp, int32_t *g_counters, int &bad,
                                          const int32_t *tri2, int64_t t2_begin, int t2_count, TriChunk<WAVES * kWave> &next,
                                          bool fixed, int dbg = 0 MVOSR_STAMP_ARG) {
    constexpr int B = WAVES * kWave;
    const int tid = threadIdx.x;
    const int npad2 = (n + 1) >> 1;
    TriChunk<B> tc;
#ifdef MVOSR_PRIO_LOAD
    __builtin_amdgcn_s_setprio(MVOSR_PRIO_LOAD);                       // (experiment: the streaming phase ahead of other workgroups' sweeps)
#endif
    tc.template load<true>(tri1, t1_begin, t1_count, 0, tid);          // in flight while the features stream in
    // planes are 16-byte aligned per frame: two features per lane and load, two loads per plane in flight
    const double2 *gy2 = reinterpret_cast<const double2 *>(gy);
    const double2 *gz2 = reinterpret_cast<const double2 *>(gz);
    const double2 *gv2 = reinterpret_cast<const double2 *>(gv);
    double2 *sY2 = reinterpret_cast<double2 *>(s.Y);
    const uint32_t ones = ((uint32_t)(kCounterBias + 1) << 16) | (uint32_t)(kCounterBias + 1);     // np.ones, :153
    for (int i0 = tid; i0 < npad2; i0 += 2 * B) {
        const int i1 = i0 + B;
        const bool two = i1 < npad2;
        const double2 ya = stream_load2(gy2 + i0), za = stream_load2(gz2 + i0), va = stream_load2(gv2 + i0);
        double2 yb = ya, zb = za, vb = va;
        if (two) { yb = stream_load2(gy2 + i1); zb = stream_load2(gz2 + i1); vb = stream_load2(gv2 + i1); }
        double2 yr, p0, p1;
        yr.x = ya.x * cp - za.x * sp;  yr.y = ya.y * cp - za.y * sp;      // :391
        p0.x = va.x; p0.y = ya.x * sp + za.x * cp;                        // :392
        p1.x = va.y; p1.y = ya.y * sp + za.y * cp;
        sY2[i0] = yr; p_st(s.P, s.Y, 2 * i0, p0); p_st(s.P, s.Y, 2 * i0 + 1, p1); s.c32[i0] = ones;
        if (two) {
            yr.x = yb.x * cp - zb.x * sp;  yr.y = yb.y * cp - zb.y * sp;
            p0.x = vb.x; p0.y = yb.x * sp + zb.x * cp;
            p1.x = vb.y; p1.y = yb.y * sp + zb.y * cp;
            sY2[i1] = yr; p_st(s.P, s.Y, 2 * i1, p0); p_st(s.P, s.Y, 2 * i1 + 1, p1); s.c32[i1] = ones;
        }
    }
#ifdef MVOSR_PRIO_LOAD
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    MVOSR_STAMP(1);

    // x is needed only after the vote: fetch it now in the compaction's slice layout, park it in registers
    const int w = wave_id(), lane = lane_id();
    const int begin = w * (SC * kWave);
    double xs[SC];
#pragma unroll
    for (int k = 0; k < SC; ++k) {
        const int i = begin + k * kWave + lane;
        xs[k] = (gx && i < n) ? stream_load(gx + i) : 0.0;
    }

    // the vote: +1 on a vertex the triangle does not flag, -1 on one it flags (:160-163)
    const int t1c = (dbg & 1) ? 0 : t1_count;
    TriChunk<B> tn;                                   // the next chunk streams in while this one is processed
    for (int base = 0; base < t1c; base += kTC * B) {
        const bool more = base + kTC * B < t1c;
        if (more) tn.template load<true>(tri1, t1_begin, t1_count, base + kTC * B, tid);
        // all vertex reads of the chunk first: the reads of one triangle cannot be moved across the
        // LDS atomics of another by the compiler, and issued together their latencies overlap
        double2 vp[kTC][3];
        bool vok[kTC];
#pragma unroll
        for (int k = 0; k < kTC; ++k) {
            const TriIds q = tc.q[k];
            vok[k] = base + k * B + tid < t1_count;
            if (vok[k] && ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n)) { bad = 1; vok[k] = false; }
            if (vok[k]) { vp[k][0] = p_ld(s.P, s.Y, q.a); vp[k][1] = p_ld(s.P, s.Y, q.b); vp[k][2] = p_ld(s.P, s.Y, q.c); }      // {v, z'}
        }
#pragma unroll
        for (int k = 0; k < kTC; ++k) {
            if (!vok[k]) continue;
            const TriIds q = tc.q[k];
            const double2 p0 = vp[k][0], p1 = vp[k][1], p2 = vp[k][2];
            const bool pa = (p0.x - p1.x) * (p0.y - p1.y) > 0.0;       // :107,:110
            const bool pb = (p0.x - p2.x) * (p0.y - p2.y) > 0.0;       // :108,:113  (marks vertices 0 and 1, as the reference does — unless `fixed`)
            const bool pc = (p1.x - p2.x) * (p1.y - p2.y) > 0.0;       // :109,:116
            const VoteFlags vf = vote_flags(pa, pb, pc, fixed);
            const bool f0 = vf.f0, f1 = vf.f1, f2 = vf.f2;
            // one wrap-around add per vertex: +-1 in the vertex's 16-bit half (the halves stay in
            // [1, 0xFFFE], so a -1 never borrows across them)
            // (-1 << 16 = 0xFFFF0000 = -(1 << 16): the sign is chosen first, then shifted into the vertex's half)
            atomicAdd(&s.c32[q.a >> 1], (f0 ? 0xFFFFFFFFu : 1u) << ((q.a & 1) * 16));
            atomicAdd(&s.c32[q.b >> 1], (f1 ? 0xFFFFFFFFu : 1u) << ((q.b & 1) * 16));
            atomicAdd(&s.c32[q.c >> 1], (f2 ? 0xFFFFFFFFu : 1u) << ((q.c & 1) * 16));
        }
        if (more) tc = tn;
    }
    if (tri2) next.load(tri2, t2_begin, t2_count, 0, tid);   // lands while the survivors are compacted
    __syncthreads();
    MVOSR_STAMP(2);

    // compaction, step 1: every wave reads its slice (flags, y', z') into registers and counts
    unsigned keepmask = 0u;                      // bit k: my feature of sub-chunk k survives
    double yk[SC], zk[SC];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < SC; ++k) {
        const int i = begin + k * kWave + lane;
        bool keep = false;
        yk[k] = 0.0; zk[k] = 0.0;
        if (i < n) {
            const int c = (int)s.c16[i] - kCounterBias;
            keep = c >= 0;                                                    // :166
            if (g_counters) g_counters[i] = c;
            yk[k] = s.Y[i];
            zk[k] = p_ldy(s.P, s.Y, i);
        }
        if (keep) keepmask |= 1u << k;
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) s.misc[M_WCNT + w] = cnt;
    __syncthreads();                             // every slice is in registers: LDS may be overwritten
    // step 2: survivors go to their compacted positions; the counters' space becomes the selected bit-set
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) { const int c = s.misc[M_WCNT + i]; if (i < w) base += c; total += c; }
#pragma unroll
    for (int k = 0; k < SC; ++k) {
        const bool keep = (keepmask >> k) & 1u;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            double2 pv; pv.x = xs[k]; pv.y = zk[k];
            p_st(s.P, s.Y, pos, pv);
            s.Y[pos] = yk[k];
        }
        base += __popcll(m);
    }
    for (int i = tid; i < (n + 31) / 32; i += B) s.sel[i] = 0u;
    __syncthreads();
    MVOSR_STAMP(3);
    return total;
}

// np.add.reduce's summation order for a 1-D float64 array (numpy/core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum: below 8 values a plain loop; up to 128 eight strided accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the remainder added one by one; above that the halves —
// the first rounded down to a multiple of 8 — summed recursively).  With sq the terms are
// (a[i]-shift)^2, as in np.std's  x = arr - mean; x = x*x; sum(x).  Every lane of the wavefront runs it
// redundantly on the same packed list; it is the cold path behind the skewness decision (road_wave).
__device__ __forceinline__ double np_term(const double *a, int i, double shift, bool sq) {
    const double v = a[i];
    if (!sq) return v;
    const double d = v - shift;
    return d * d;
}
__device__ double np_leaf_sum(const double *a, int n, double shift, bool sq) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += np_term(a, i, shift, sq);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = np_term(a, j, shift, sq);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += np_term(a, i + j, shift, sq);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += np_term(a, i, shift, sq);
    return res;
}
// The recursion's stack lives in LDS (kNpStackInts ints per wavefront, 8-byte aligned; every lane of the wavefront walks
// the same list and writes the same values): as private arrays it was 688 B of scratch memory per lane of every wavefront
// of road_model_kernel, for a branch one list in thousands takes.
constexpr int kNpDepth = 8;                                  // a chunk has <= 8192 elements, a leaf <= 128: at most 7 levels
constexpr int kNpStackInts = 3 * kNpDepth + 2 * kNpDepth;
__device__ __attribute__((noinline)) double np_pairwise_chunk_cold(const double *a, int n, double shift, int sq, int *stk) {
    int *lo_s = stk, *n_s = stk + kNpDepth, *stage_s = stk + 2 * kNpDepth;
    double *val = reinterpret_cast<double *>(stk + 3 * kNpDepth);
    int sp = 0, vp = 0;
    lo_s[0] = 0; n_s[0] = n; stage_s[0] = 0; sp = 1;
    while (sp > 0) {
        const int lo = lo_s[sp - 1], m = n_s[sp - 1], stage = stage_s[sp - 1];
        if (m <= 128) { val[vp++] = np_leaf_sum(a + lo, m, shift, sq != 0); --sp; continue; }
        int m2 = m / 2;
        m2 -= m2 % 8;
        if (stage == 0) { stage_s[sp - 1] = 1; lo_s[sp] = lo; n_s[sp] = m2; stage_s[sp] = 0; ++sp; }
        else if (stage == 1) { stage_s[sp - 1] = 2; lo_s[sp] = lo + m2; n_s[sp] = m - m2; stage_s[sp] = 0; ++sp; }
        else { const double r = val[--vp], l = val[--vp]; val[vp++] = l + r; --sp; }
    }
    return val[0];
}

// np.add.reduce hands its inner loop at most `bufsize` (8192, np.getbufsize()) elements at a time and adds the
// chunks' pairwise sums up from left to right (checked against NumPy 2.2 for lists of 10^4 - 2*10^5 elements: a single
// pairwise recursion over the whole list differs in the last bits from 10291 elements on).
constexpr int kNpBufSize = 8192;
__device__ __forceinline__ double np_pairwise_sum_cold(const double *a, int n, double shift, int sq, int *stk) {
    double res = np_pairwise_chunk_cold(a, min(n, kNpBufSize), shift, sq, stk);
    for (int lo = kNpBufSize; lo < n; lo += kNpBufSize) res += np_pairwise_chunk_cold(a + lo, min(kNpBufSize, n - lo), shift, sq, stk);
    return res;
}
// The same summation over a STREAM of values (next() hands out the list's elements in order; all lanes of
// the wavefront run it redundantly with wave-uniform values): the cold path that makes height_level the very
// double np.mean(heights[pitch_deg >= -80]) is (scale_calculator.py:239-240).
template <class S>
__device__ __forceinline__ double np_leaf_stream(S &st, int n) {
    if (n < 8) {
        double res = 0.0;
#pragma unroll 1
        for (int i = 0; i < n; ++i) res += st.next();
        return res;
    }
    double r[8];
#pragma unroll 1
    for (int j = 0; j < 8; ++j) r[j] = st.next();
    int i = 8;
#pragma unroll 1
    for (; i < n - (n % 8); ++i) r[i & 7] = r[i & 7] + st.next();
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll 1
    for (; i < n; ++i) res += st.next();
    return res;
}
template <class S>
__device__ __forceinline__ double np_pairwise_chunk_stream(S &st, int n) {
    constexpr int kDepth = 32;
    int n_s[kDepth], stage_s[kDepth];
    double val[kDepth];
    int sp = 1, vp = 0;
    n_s[0] = n; stage_s[0] = 0;
    while (sp > 0) {
        const int m = n_s[sp - 1], stage = stage_s[sp - 1];
        if (m <= 128) { val[vp++] = np_leaf_stream(st, m); --sp; continue; }
        int m2 = m / 2;
        m2 -= m2 % 8;
        if (stage == 0) { stage_s[sp - 1] = 1; n_s[sp] = m2; stage_s[sp] = 0; ++sp; }
        else if (stage == 1) { stage_s[sp - 1] = 2; n_s[sp] = m - m2; stage_s[sp] = 0; ++sp; }
        else { const double r = val[--vp], l = val[--vp]; val[vp++] = l + r; --sp; }
    }
    return val[0];
}
template <class S>
__device__ __forceinline__ double np_pairwise_stream(S &st, int n) {
    double res = np_pairwise_chunk_stream(st, min(n, kNpBufSize));
    for (int lo = kNpBufSize; lo < n; lo += kNpBufSize) res += np_pairwise_chunk_stream(st, min(kNpBufSize, n - lo));
    return res;
}

// ---------------------------------------------------------------------------------------------
// Phase B: plane normals / pitch / height over the second triangulation, height_level, and the
// bit-set of selected (surviving-feature) indices.
// ---------------------------------------------------------------------------------------------
// The reference's own formulation of the per-triangle test (scale_calculator.py:229-239): LU solve
// for the plane normal, normalise, asin, degrees, compare.  Bit 0: pitch < thr (flat), bit 1:
// pitch >= thr, bit 2: exactly singular.  Not inlined in the product kernel, where it is the
// cold path taken only inside the 1e-9 band around the threshold.
template <bool INLINE>
__device__ int pitch_reference_impl(double x0, double y0, double z0, double x1, double y1, double z1,
                                    double x2, double y2, double z2, double thr_deg,
                                    double &nx, double &ny, double &nz, double &pitch) {
    int r = 0;
    if (!plane_normal(x0, y0, z0, x1, y1, z1, x2, y2, z2, nx, ny, nz)) r |= 4;        // :229-230
    const double len2 = (nx * nx + ny * ny) + nz * nz;                               // :231
    const double len = sqrt(len2);
    const double uy = ny / len;                                                      // :232
    pitch = asin(-uy) * 180.0 / 3.141592653589793;                                   // :233
    if (pitch < thr_deg) r |= 1;                                                     // :235
    if (pitch >= thr_deg) r |= 2;                                                    // :239  (NaN: neither)
    return r;
}
__device__ __attribute__((noinline)) int pitch_reference_cold(double x0, double y0, double z0, double x1, double y1, double z1,
                                                              double x2, double y2, double z2, double thr_deg) {
    double nx, ny, nz, pitch;
    return pitch_reference_impl<false>(x0, y0, z0, x1, y1, z1, x2, y2, z2, thr_deg, nx, ny, nz, pitch);
}
template <bool FULL>
__device__ __forceinline__ int pitch_reference(double x0, double y0, double z0, double x1, double y1, double z1,
                                               double x2, double y2, double z2, double thr_deg,
                                               double &nx, double &ny, double &nz, double &pitch) {
    if constexpr (FULL) return pitch_reference_impl<true>(x0, y0, z0, x1, y1, z1, x2, y2, z2, thr_deg, nx, ny, nz, pitch);
    else { nx = ny = nz = pitch = 0.0; return pitch_reference_cold(x0, y0, z0, x1, y1, z1, x2, y2, z2, thr_deg); }
}

struct SelectResult {
    double height_level;
    int n_pitch, n_tri_valid;
    int singular, bad;
    int n_steep;          // triangles with pitch_deg >= thr (what height_level averages over)
    int near;             // a flat triangle's height is within rounding of height_level (hot mode only)
};

// Kernel modes.  HOT: the product path — height_level from the sweep's own fixed-order sum; a frame whose result could
// depend on the last bits of that sum (a flat triangle within kLevelGuard of the level, or so few selected points that
// the level itself may become the height) is not finished but put on the context's redo list.  EXACT: height_level in
// NumPy's summation order before it is used — the redo pass over that list, and every frame when stage outputs are
// requested.  FULL: EXACT plus the reference's literal per-triangle formulation and the per-triangle debug outputs.
enum { MODE_HOT = 0, MODE_EXACT = 1, MODE_FULL = 2 };

struct PitchTest {
    double thr_deg;      // -80
    double s2_hi, s2_lo; // sin^2(|thr|) * (1 +- 1e-9): outside this band the decision needs no asin
};

// One triangle of the first sweep (:229-240): bit 0 = pitch_deg < thr (flat), bit 1 = pitch_deg >= thr,
// bit 2 = exactly singular.  `h` is the triangle's mean y' (only written out in FULL mode).
template <bool FULL>
__device__ __forceinline__ int classify_triangle(double x0, double y0, double z0, double x1, double y1, double z1,
                                                 double x2, double y2, double z2, double h, const PitchTest &pt,
                                                 double *g_normals, double *g_pitch, double *g_heights, int64_t tg) {
    bool is_flat = false, is_steep = false, is_singular = false;
    bool decided = false;
    if constexpr (!FULL) {
        // The plane n.p = 1 through the vertices has n = c / det with c = (p1-p0) x (p2-p0) and
        // det = p0 . c, so  pitch_deg < thr  <=>  n_y/|n| > sin(|thr|)  <=>  c_y det > 0 and
        // c_y^2 > sin^2(|thr|) |c|^2 : no division, square root or asin.  Only inside a 1e-9
        // band around the threshold (where the outcome depends on how the LU solve and asin
        // round), for collinear vertices (c = 0 fails both comparisons) and when det is lost to
        // cancellation (possible exact singularity) is the reference's own formulation evaluated below.
        const double e1x = x1 - x0, e1y = y1 - y0, e1z = z1 - z0;
        const double e2x = x2 - x0, e2y = y2 - y0, e2z = z2 - z0;
        const double cx = __builtin_fma(e1y, e2z, -(e1z * e2y));
        const double cy = __builtin_fma(e1z, e2x, -(e1x * e2z));
        const double cz = __builtin_fma(e1x, e2y, -(e1y * e2x));
        const double tx = x0 * cx, ty = y0 * cy, tz = z0 * cz;
        const double det = (tx + ty) + tz;
        const double mag = (fabs(tx) + fabs(ty)) + fabs(tz);
        const double c2 = __builtin_fma(cz, cz, __builtin_fma(cy, cy, cx * cx));
        const double q2 = cy * cy;
        const double sy = cy * det;
        const bool safe = fabs(det) > 1e-9 * mag;
        // (bitwise, not short-circuit: no control flow for five comparisons)
        is_flat = safe & (sy > 0.0) & (q2 > pt.s2_hi * c2);
        is_steep = safe & ((sy <= 0.0) | (q2 < pt.s2_lo * c2));       // exclusive with is_flat: s2_lo < s2_hi
        decided = is_flat | is_steep;
    }
    if (!decided) {
        double nx, ny, nz, pitch;
        const int r = pitch_reference<FULL>(x0, y0, z0, x1, y1, z1, x2, y2, z2, pt.thr_deg, nx, ny, nz, pitch);
        is_singular = r & 4;
        is_flat = r & 1;
        is_steep = r & 2;
        if constexpr (FULL) {
            if (g_normals) { double *o = g_normals + 3 * tg; o[0] = nx; o[1] = ny; o[2] = nz; }
            if (g_pitch) g_pitch[tg] = pitch;
            if (g_heights) g_heights[tg] = h;
        }
    }
    // bit 3: decided by the reference's own formulation — the one place where the ROTATION of the row (the order LAPACK's LU sees the
    // vertices in) can move the outcome: a caller whose rows are stand-ins for SciPy's (same triangles, other rotation) redoes the frame
    return (is_flat ? 1 : 0) | (is_steep ? 2 : 0) | (is_singular ? 4 : 0) | (decided ? 0 : 8);
}

// The heights of the frame's STEEP triangles (pitch_deg >= thr, :239) in the row order of tri2, one at a time and
// wave-uniform: 64 rows are classified per refill (same decisions as the sweep), the ballot of the steep ones is
// consumed lowest lane first.  `fetch(q, x0..z2)` supplies a row's vertices (LDS or global) and says whether its
// ids are legal.
template <bool FULL, class Fetch>
struct SteepStream {
    const int32_t *tri;           // the frame's first row
    const int32_t *order;         // null, or: the k-th row of the caller's ORIGINAL order is row order[k] here (mvosr_batch.tri2_order)
    int t2n;
    const PitchTest &pt;
    Fetch fetch;
    int t_next;
    unsigned long long mask;
    double h;
    __device__ __forceinline__ void refill() {
        const int t = t_next + lane_id();
        bool steep = false;
        h = 0.0;
        if (t < t2n) {
            const int row = order ? min(max(order[t], 0), t2n - 1) : t;      // (clamped: a bad table cannot index outside the frame's rows)
            const TriIds q = load_tri(tri, row);
            double x0, y0, z0, x1, y1, z1, x2, y2, z2;
            if (fetch(q, x0, y0, z0, x1, y1, z1, x2, y2, z2)) {
                h = div3((y0 + y1) + y2);
                steep = classify_triangle<FULL>(x0, y0, z0, x1, y1, z1, x2, y2, z2, h, pt, nullptr, nullptr, nullptr, 0) & 2;
            }
        }
        mask = __ballot(steep);
        t_next += kWave;
    }
    __device__ __forceinline__ double next() {
        while (mask == 0ull) {
            if (t_next >= t2n) return nan("");                 // (cannot happen: the caller asks for exactly the steep count)
            refill();
        }
        const int l = (int)__ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        return readlane_d(h, l);
    }
};

// height_level exactly as NumPy computes it (:239-240): np.mean = pairwise add.reduce over heights[pitch_deg >= -80]
// in row order, divided by the count.  `n_steep` is the count the sweep found.  Every wavefront that calls it
// computes the same value (no barrier, nothing written).  Only the EXACT / FULL kernel variants contain it.
template <bool FULL, class Fetch>
__device__ __forceinline__ double exact_height_level(const int32_t *tri, int t2n, int n_steep, const PitchTest &pt, Fetch fetch,
                                                     const int32_t *order = nullptr) {
    if (n_steep <= 0) return nan("");                           // np.mean of an empty slice
    SteepStream<FULL, Fetch> st{tri, order, t2n, pt, fetch, 0, 0ull, 0.0};
    return (0.0 + np_pairwise_stream(st, n_steep)) / (double)n_steep;
}

// The same double, computed by the whole workgroup.  The streamed version above is one dependent chain per steep triangle
// (≈0.3 ms for a 2000-feature frame: a quarter of the per-frame call's GPU time); here the steep heights are first packed
// into `hs` in row order (ballots + a running count: two barriers per WAVES*64 rows), then NumPy's pairwise tree is
// evaluated with one thread per LEAF (<= 128 elements, the eight-accumulator loop of np_leaf_sum) and the leaves' sums
// are added up in the recursion's own order by one thread — the same additions in the same order, hence the same bits.
// `hs`: global scratch of the frame (capacity `cap` doubles; the caller falls back to the streamed version when
// n_steep + kNpMaxLeaves does not fit).  `wcnt`: WAVES ints of LDS.  `slot`: one double of LDS.  Workgroup-uniform call.
constexpr int kNpMaxLeaves = kNpBufSize / 64;      // a leaf has more than 64 elements unless the chunk itself is one
template <class Visit>
__device__ __forceinline__ void np_pairwise_leaves(int n, Visit visit) {      // visit(lo, len, is_leaf) in post-order; false = an inner node
    int lo_s[kNpDepth], n_s[kNpDepth], stage_s[kNpDepth];
    int sp = 1;
    lo_s[0] = 0; n_s[0] = n; stage_s[0] = 0;
    while (sp > 0) {
        const int lo = lo_s[sp - 1], m = n_s[sp - 1], stage = stage_s[sp - 1];
        if (m <= 128) { visit(lo, m, true); --sp; continue; }
        int m2 = m / 2;
        m2 -= m2 % 8;
        if (stage == 0) { stage_s[sp - 1] = 1; lo_s[sp] = lo; n_s[sp] = m2; stage_s[sp] = 0; ++sp; }
        else if (stage == 1) { stage_s[sp - 1] = 2; lo_s[sp] = lo + m2; n_s[sp] = m - m2; stage_s[sp] = 0; ++sp; }
        else { visit(lo, m, false); --sp; }
    }
}
template <int WAVES, bool FULL, class Fetch>
__device__ __forceinline__ double exact_height_level_block(const int32_t *tri, int t2n, int n_steep, const PitchTest &pt, Fetch fetch,
                                                           const int32_t *order, double *hs, int *wcnt, double *slot) {
    constexpr int B = WAVES * kWave;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    if (n_steep <= 0) return nan("");                           // np.mean of an empty slice
    int packed = 0;
    for (int r0 = 0; r0 < t2n; r0 += B) {
        const int t = r0 + tid;
        bool steep = false;
        double h = 0.0;
        if (t < t2n) {
            const int row = order ? min(max(order[t], 0), t2n - 1) : t;
            const TriIds q = load_tri(tri, row);
            double x0, y0, z0, x1, y1, z1, x2, y2, z2;
            if (fetch(q, x0, y0, z0, x1, y1, z1, x2, y2, z2)) {
                h = div3((y0 + y1) + y2);
                steep = classify_triangle<FULL>(x0, y0, z0, x1, y1, z1, x2, y2, z2, h, pt, nullptr, nullptr, nullptr, 0) & 2;
            }
        }
        const unsigned long long m = __ballot(steep);
        if (lane == 0) wcnt[w] = __popcll(m);
        __syncthreads();
        int before = packed, total = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) { const int c = wcnt[i]; if (i < w) before += c; total += c; }
        if (steep) { const int pos = before + __popcll(m & ((1ull << lane) - 1ull)); if (pos < n_steep) hs[pos] = h; }
        packed += total;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    double *leaf = hs + n_steep;                                 // the leaves' sums of the current chunk
    double res = 0.0;
    for (int c0 = 0; c0 < n_steep; c0 += kNpBufSize) {
        const int m = min(kNpBufSize, n_steep - c0);
        int k = 0;
        np_pairwise_leaves(m, [&](int lo, int len, bool is_leaf) {
            if (is_leaf) { if (k % B == tid) leaf[k] = np_leaf_sum(hs + c0 + lo, len, 0.0, false); ++k; }
        });
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            double val[kNpDepth + 1];
            int vp = 0, kk = 0;
            np_pairwise_leaves(m, [&](int, int, bool is_leaf) {
                if (is_leaf) val[vp++] = leaf[kk++];
                else { const double r = val[--vp], l = val[--vp]; val[vp++] = l + r; }
            });
            *slot = c0 == 0 ? val[0] : *slot + val[0];
        }
        __syncthreads();
    }
    res = *slot;
    __syncthreads();
    return (0.0 + res) / (double)n_steep;
}

struct LdsFetch {            // rows of tri2 index the compacted survivors in LDS
    const double2 *P; const double *Y; int n_valid;
    __device__ __forceinline__ bool operator()(const TriIds q, double &x0, double &y0, double &z0, double &x1, double &y1, double &z1,
                                               double &x2, double &y2, double &z2) const {
        if ((unsigned)q.a >= (unsigned)n_valid || (unsigned)q.b >= (unsigned)n_valid || (unsigned)q.c >= (unsigned)n_valid) return false;
        const double2 p0 = p_ld(P, Y, q.a), p1 = p_ld(P, Y, q.b), p2 = p_ld(P, Y, q.c);
        x0 = p0.x; z0 = p0.y; x1 = p1.x; z1 = p1.y; x2 = p2.x; z2 = p2.y;
        y0 = Y[q.a]; y1 = Y[q.b]; y2 = Y[q.c];
        return true;
    }
};

// |h - level| within this distance, RELATIVE TO THE MEAN MAGNITUDE OF WHAT WAS SUMMED (sum |h| / count, not |level|: the
// rounding error of a sum scales with the magnitudes of its terms, and steep triangles above and below the camera can
// cancel to a level near zero): the comparison h > level (:243) may depend on the summation order behind the level, so
// the level is recomputed in NumPy's order before it is trusted (the sweep's own sum agrees with NumPy's to ~1e-15 of
// that magnitude)
constexpr double kLevelGuard = 1e-12;

template <int WAVES, int MODE, int FW = 1>
__device__ __forceinline__ SelectResult phase_select(const Smem &s, int n_valid, const int32_t *tri2, int64_t t2_begin,
                                                     int t2_count, TriChunk<WAVES * kWave> &tc, PitchTest pt,
                                                     double *g_normals, double *g_pitch, double *g_heights, int bad_in,
                                                     double *scratch, int scratch_cap, int dbg = 0 MVOSR_STAMP_ARG) {
    constexpr int B = WAVES * kWave;
    constexpr bool FULL = MODE == MODE_FULL;
    const int tid = threadIdx.x;
    unsigned long long flat = 0ull;          // bit kk: my kk-th triangle has pitch_deg < thr
    unsigned long long flat_hi = 0ull;       // bits 64..127 (FW == 2: dense frames, up to 128 triangles per thread)
    double hsum = 0.0, hcnt = 0.0, habs = 0.0;
    int npitch = 0, singular = 0, bad = 0;
    int undecided = 0;       // HOT: a triangle inside the pitch test's band (classified by the reference's formulation): the frame joins the redo list
    // more rows than the per-thread flag words can name (not a triangulation of this frame's points): refuse
    if (t2_count > 64 * FW * B) { bad = 1; t2_count = 0; }
    // One triangle of the first sweep (:229-240).
    auto test_triangle = [&](int t, int kk, const TriIds q) {
    if ((unsigned)q.a >= (unsigned)n_valid || (unsigned)q.b >= (unsigned)n_valid || (unsigned)q.c >= (unsigned)n_valid) { bad = 1; return; }
    const double2 p0 = p_ld(s.P, s.Y, q.a), p1 = p_ld(s.P, s.Y, q.b), p2 = p_ld(s.P, s.Y, q.c);      // {x, z'}
    const double y0 = s.Y[q.a], y1 = s.Y[q.b], y2 = s.Y[q.c];
    const double x0 = p0.x, z0 = p0.y, x1 = p1.x, z1 = p1.y, x2 = p2.x, z2 = p2.y;
    // :238.  The product path works on 3h = (y0+y1)+y2: its level is only trusted outside the guard band anyway (a frame
    // with a flat triangle inside it is redone in the EXACT variant), which is ~10^4 roundings wide — so the division
    // by 3 per triangle, and its rounding, are left out of both sweeps.
    const double h = MODE == MODE_HOT ? (y0 + y1) + y2 : div3((y0 + y1) + y2);
    const int r = classify_triangle<FULL>(x0, y0, z0, x1, y1, z1, x2, y2, z2, h, pt, g_normals, g_pitch, g_heights, t2_begin + t);
    const bool is_flat = r & 1, is_steep = r & 2;
    if (r & 4) singular = 1;
    if constexpr (MODE == MODE_HOT) { if (r & 8) undecided = 1; }
    if (is_steep) { hsum += h; hcnt += 1.0; if constexpr (MODE == MODE_HOT) habs += fabs(h); }     // :240
    if (is_flat) {
        if (FW == 1 || kk < 64) flat |= 1ull << (kk & 63); else flat_hi |= 1ull << (kk & 63);
        ++npitch;
    }
    };
    // `tc` arrives with the first chunk of tri2 already loaded (phase_vote); the next chunk streams
    // in while the current one is processed.
    const int t2c = (dbg & 2) ? 0 : t2_count;
    TriChunk<B> tn;
    for (int base = 0; base < t2c; base += kTC * B) {
        const bool more = base + kTC * B < t2c;
        if (more) tn.load(tri2, t2_begin, t2_count, base + kTC * B, tid);
#pragma unroll
        for (int k = 0; k < kTC; ++k) {
            const int t = base + k * B + tid;
            if (t < t2c) test_triangle(t, base / B + k, tc.q[k]);
        }
        if (more) tc = tn;
    }
    // second sweep: its first chunk is re-read now, under the reduction's barrier (unless the whole
    // triangulation was one chunk and is still in registers)
    const int t2d = (dbg & 4) ? 0 : t2_count;
    const bool in_regs = t2_count <= kTC * B;
    if (!in_regs && t2d > 0) tc.template load<true>(tri2, t2_begin, t2_count, 0, tid);
    if constexpr (MODE == MODE_HOT) block_sum3<WAVES>(hsum, hcnt, habs, s.red + R_SEL_H * 2 * WAVES, s.red + R_SEL_ABS * 2 * WAVES);
    else block_sum2<WAVES>(hsum, hcnt, s.red + R_SEL_H * 2 * WAVES);
    MVOSR_STAMP(4);
    SelectResult r;
    double hl = hsum / hcnt;                      // np.mean of an empty set -> 0/0 = NaN, like :240.  (HOT: the mean of 3h)
    const double guard = kLevelGuard * (habs / hcnt);
    if constexpr (MODE != MODE_HOT) {
        // (workgroup-uniform condition: hcnt is the block sum)
        if (scratch && (int)hcnt + kNpMaxLeaves <= scratch_cap)
            hl = exact_height_level_block<WAVES, FULL>(tri2 + 3 * t2_begin, t2_count, (int)hcnt, pt, LdsFetch{s.P, s.Y, n_valid}, nullptr,
                                                       scratch, s.misc + M_WCNT, s.red + R_MISC * 2 * WAVES);
        else
            hl = exact_height_level<FULL>(tri2 + 3 * t2_begin, t2_count, (int)hcnt, pt, LdsFetch{s.P, s.Y, n_valid});
    }
    int ntv = 0, near = undecided;
    auto mark_triangle = [&](int kk, int qa, int qb, int qc) {
        const unsigned long long fw = (FW == 1 || kk < 64) ? flat : flat_hi;
        if (!((fw >> (kk & 63)) & 1ull)) return;
        const double y0 = s.Y[qa], y1 = s.Y[qb], y2 = s.Y[qc];
        const double h = MODE == MODE_HOT ? (y0 + y1) + y2 : div3((y0 + y1) + y2);           // HOT: 3h against 3 * level
        if constexpr (MODE == MODE_HOT) { if (fabs(h - hl) <= guard) near = 1; }
        if (h > hl) {                                                                        // :243-244
            ++ntv;
            atomicOr(&s.sel[qa >> 5], 1u << (qa & 31));                                      // :247
            atomicOr(&s.sel[qb >> 5], 1u << (qb & 31));
            atomicOr(&s.sel[qc >> 5], 1u << (qc & 31));
        }
    };
    for (int base = 0; base < t2d; base += kTC * B) {
        const bool more = base + kTC * B < t2d;
        if (more) tn.template load<true>(tri2, t2_begin, t2_count, base + kTC * B, tid);
#pragma unroll
        for (int k = 0; k < kTC; ++k) {
            if (base + k * B + tid < t2d) mark_triangle(base / B + k, tc.q[k].a, tc.q[k].b, tc.q[k].c);
        }
        if (more) tc = tn;
    }
    bad |= bad_in;
    int sn = singular | (near << 16);
    block_sum4i<WAVES>(npitch, ntv, sn, bad, s.red + R_SEL_CNT * 2 * WAVES);   // also orders the atomicOr's
    singular = sn & 0xFFFF;
    r.near = sn >> 16;
    MVOSR_STAMP(5);
    r.height_level = MODE == MODE_HOT ? hl / 3.0 : hl;
    r.n_steep = (int)hcnt;
    r.n_pitch = npitch; r.n_tri_valid = ntv; r.singular = singular; r.bad = bad;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Phase C (its own kernel, one frame per WAVEFRONT): road model on a dense list of y' values —
// the selected points the scale kernel wrote to the workspace, or caller-supplied lists.
// No barriers: a wave keeps up to 16 values per lane in registers, builds the 169-bin histogram
// with wave-local LDS atomics, evaluates remove_single / modes / local minima on 64-bit ballots
// and reduces mean / std with DPP.  Runs at full occupancy, so its latency chains are hidden by
// other waves instead of holding a 53 KB LDS allocation idle.
//                                                (road_model_calculation_static, :324-354)
// ---------------------------------------------------------------------------------------------
struct RoadResult {
    double height;
    int status;
    int n_sel, n_kept, n_modes, mode_left, mode_right;
    double mean, std, skew, median;
};

// is y (histogram bin `bin`) inside the interval remove_single deletes for a single-count bin?
// (:284-293)  `single` = bins whose raw count is exactly 1; an interval is built from the bin's
// RIGHT edge r as [r-0.1, r] for the first single bin and (r-0.1, r] for the others — r-0.1 is
// not always the bin's own left edge in floating point, so neighbours are checked too.
__device__ __forceinline__ bool dropped_by_single(double y, int bin, const Bits192 &single, int first_single) {
    bool d = false;
#pragma unroll
    for (int dk = -1; dk <= 1; ++dk) {
        const int k = bin + dk;
        const bool is_single = (k >= 0) && (k < kBins) && single.test_lane(max(k, 0));
        const double r = bin_edge(k + 1);
        const double lo = r - 0.1;
        const bool in = (k == first_single) ? (y >= lo && y <= r) : (y > lo && y <= r);
        d |= is_single && in;
    }
    return d;
}

#ifndef MVOSR_ROAD_RC
#define MVOSR_ROAD_RC 20
#endif
constexpr int kRoadRC = MVOSR_ROAD_RC;   // deepest register cache: values per lane kept in registers (lists up to 64*RC values; longer ones
                                         // re-read).  20 keeps the kernel at 4 wavefronts per SIMD (121 VGPRs); 32 was measured: 2 per SIMD, slower
constexpr int kDropStride = 32;          // verdict bytes per lane in LDS (two 16-byte words)
constexpr int kRoadWaves = 4;          // frames (wavefronts) per workgroup
constexpr int kTrash = 175;            // histogram slot for values that are not binned (bins are 0..168)
constexpr int kStPending = -1;         // scale kernel -> road kernel: "road model still to run"
constexpr int kStRedo = -2;            // HOT scale kernel -> EXACT pass: "needs height_level in NumPy's summation order"
static_assert(kStRedo == MVOSR_ST_REDO, "include/mvosr.h");
constexpr int kMaxVoteRows = 32765;    // a 16-bit biased vote counter stays in [1, 0xFFFE] whatever the rows say while a vertex has at most this many

struct RoadArgs {
    mvosr_params P;
    const int64_t *off;          // [F] start of frame f's list in `y`
    const int32_t *cnt;          // [F] list length
    const double *y;             // the values
    double *scratch;             // same layout, writable: kept values for the median fallback (may alias y)
    const double *height_level;  // [F] fallback level (:335), may be NULL
    mvosr_outputs o;
    int64_t first_frame, n_frames;
    int pending_only;            // 1: fused path, only frames the scale kernel left pending
    int32_t *level_redo;         // fused HOT path: frames that end on the fallback level (:334-335) are not finished but appended
                                 // here ([0] = count): height_level becomes their result, so it must be NumPy's own double first
    const int32_t *list;         // process the frames list[1 .. list[0]] (grid-strided) instead of first_frame + index
    int wide;                    // one WORKGROUP per frame (dense batches: long lists, few frames)
    const uint8_t *listed_mask = nullptr;   // with level_redo: frames whose byte is set are ON that list already (append_mask_kernel put
                                 // them there behind the HOT kernel) — marked for the exact pass, not appended a second time
};

// bin of y against the workgroup's table edges[k] = {edge k, edge k+1} (same doubles as bin_edge)
__device__ __forceinline__ int bin_of_table(double y, const double2 *edges) {
    if (!(y >= 0.0 && y <= bin_edge(kBins))) return -1;
    int k = min((int)(y * 10.0), kBins - 1);
    const double2 e = edges[k];
    k += (k < kBins - 1 && y >= e.y) ? 1 : 0;
    k -= (y < e.x) ? 1 : 0;
    return k;
}

// WW = 1: the whole list in one wavefront.  WW = kRoadWaves (dense frames, lists of thousands of values, too few
// frames to fill the GPU with one wavefront each): the workgroup's wavefronts share the frame — each takes a
// contiguous part `[lo, hi)` of the list through the per-value passes (histogram into the SHARED `hist`, suspects,
// sums; partial sums meet in `part`, added in wavefront order), every wavefront evaluates the histogram logic
// redundantly (same inputs, same result), and wavefront 0 alone finishes the frame (modes, the decision, the cold
// exact branches over the whole list).  `Mall` is the whole list's length; `yv` its first value.
template <int RC, int WW = 1>
__device__ __forceinline__ RoadResult road_wave(int *hist, int *nearflag, uint16_t *slots, uint8_t *dropb, const double2 *edges,
                                                const double *yv_all, double *scratch,
                                                int Mall, double height_level, const mvosr_params &P, int32_t *g_hist, bool exact_stats,
                                                int part_lo, int part_hi, double *part, int *np_stack MVOSR_STAMP_ARG) {
    const int lane = lane_id();
    RoadResult R;
    R.height = nan(""); R.status = MVOSR_ST_MODE; R.n_sel = Mall; R.n_kept = 0; R.n_modes = 0; R.mode_left = -1; R.mode_right = -1;
    R.mean = R.std = R.skew = R.median = nan("");
    if (Mall == 0) { R.status = MVOSR_ST_NO_FLAT; return R; }
    const double *yv = yv_all + (part_hi > part_lo ? part_lo : 0);     // my part of the list (WW == 1: all of it; an empty part: any legal address)
    const int M = part_hi - part_lo;                    // (may be 0 for the last wavefronts of a short list)
    if (WW == 1 || wave_id() == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int b = lane + 64 * c; if (b < 176) hist[b] = 0; }
    }
    if constexpr (WW > 1) __syncthreads();

    // histogram (np.histogram, :326); the first RC values of every lane stay in registers
    double yc[RC];
    int binc[RC];
    // Branch-free on purpose: with no control flow between them the 16 loads, table reads and
    // atomics of a lane are scheduled in batches instead of one dependent round trip per value.
    // Values that take no part (beyond the list, or outside [0,16.9]) go to the trash bin kTrash.
    const int nfull = M / kWave;
    const unsigned valid = (nfull >= RC) ? (unsigned)((1ull << RC) - 1ull)     // bit k: value k*64+lane exists
                                              : (((1u << nfull) - 1u) | ((lane < M - nfull * kWave ? 1u : 0u) << nfull));
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        const int i = max(min(k * kWave + lane, M - 1), 0);                               // clamped: always a legal address
        yc[k] = yv[i];
    }
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        const double y = yc[k];
        const bool inr = ((valid >> k) & 1u) && (y >= 0.0) && (y <= bin_edge(kBins));
        const int g = inr ? min((int)(y * 10.0), kBins - 1) : 0;
        const double2 e = edges[g];
        int bin = g + ((g < kBins - 1 && y >= e.y) ? 1 : 0) - ((y < e.x) ? 1 : 0);
        bin = inr ? bin : kTrash;
        atomicAdd(&hist[bin], 1);
        binc[k] = bin;
    }
    // (lists longer than the register cache — dense frames: four loads in flight per trip instead of one dependent
    // round trip per value; the same in the two passes further down)
    constexpr int kLongUnroll = 8;
    for (int i0 = RC * kWave + lane; i0 < M; i0 += kLongUnroll * kWave) {
        double yl[kLongUnroll];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) yl[j] = yv[max(min(i0 + j * kWave, M - 1), 0)];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) {
            const int bin = (i0 + j * kWave < M) ? bin_of_table(yl[j], edges) : -1;
            if (bin >= 0) atomicAdd(&hist[bin], 1);
        }
    }
    MVOSR_RSTAMP(2);
    if constexpr (WW > 1) __syncthreads();              // every part's values are in the shared histogram
    // (one wave: its LDS operations execute in order, the reads below see the atomics above)
    int hraw[3], hz[3];
    Bits192 single, modes, mins;
    int mx = 0, mn = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int b = lane + 64 * c;
        hraw[c] = (b < kBins) ? hist[b] : 0;
        hz[c] = (hraw[c] == 1) ? 0 : hraw[c];                                  // dis[dis==1]=0, :328
        single.w[c] = __ballot(b < kBins && hraw[c] == 1);
        mx = max(mx, hz[c]);
        if (b < kBins) mn = min(mn, hz[c]);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    if (g_hist && (WW == 1 || wave_id() == 0)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int b = lane + 64 * c; if (b < kBins) { g_hist[b] = hraw[c]; g_hist[kBins + b] = hz[c]; } }
    }
    MVOSR_RSTAMP(3);
    const int first_single = single.lowest_from(0);
    // a value can only be dropped if its own bin or a neighbouring one has count 1
    Bits192 near;
    near.w[0] = single.w[0] | (single.w[0] << 1) | (single.w[0] >> 1) | (single.w[1] << 63);
    near.w[1] = single.w[1] | (single.w[1] << 1) | (single.w[1] >> 1) | (single.w[0] >> 63) | (single.w[2] << 63);
    near.w[2] = single.w[2] | (single.w[2] << 1) | (single.w[2] >> 1) | (single.w[1] >> 63);
#pragma unroll
    for (int c = 0; c < 3; ++c) { const int b = lane + 64 * c; if (b < 176) nearflag[b] = (b < kBins) ? (int)((near.w[c] >> lane) & 1ull) : 0; }
    // check_mode returns no modes iff max <= 2 (:451-452); otherwise the maximum bin itself is one
    const bool have_modes = mx > P.mode_min;

    // second pass: drop the points inside a single bin's interval (:284-293), accumulate the mean.
    // Only values whose own or neighbouring bin has count 1 can be dropped: they are flagged here
    // (one LDS read each) and examined in a rolled loop below, which most iterations skip.
    // The few suspects of the whole list are packed into one dense list (slot numbers, ballot prefix)
    // and examined with full lanes — row by row almost every row would run the whole interval test
    // for one or two lanes.  Verdicts return through a byte per slot, 16 contiguous bytes per lane.
    unsigned kept = valid;
    {
        uint4 zero; zero.x = zero.y = zero.z = zero.w = 0u;
        reinterpret_cast<uint4 *>(dropb)[2 * lane] = zero;
        if (RC > 16) reinterpret_cast<uint4 *>(dropb)[2 * lane + 1] = zero;
        int ns = 0;
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const bool sus = nearflag[binc[k]] != 0;                          // nearflag[kTrash] == 0
            const unsigned long long m = __ballot(sus);
            if (sus) slots[ns + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(k * kWave + lane);
            ns += __popcll(m);
        }
        for (int i = lane; i < ns; i += kWave) {
            const int e = slots[i];
            const double y = yv[e];                                           // (a cache hit; avoids indexing the register array)
            if (dropped_by_single(y, bin_of_table(y, edges), single, first_single)) dropb[(e & (kWave - 1)) * kDropStride + (e >> 6)] = 1;
        }
        if (ns > 0) {
#pragma unroll
            for (int j = 0; j < (RC + 15) / 16; ++j) {
                const uint4 d = reinterpret_cast<const uint4 *>(dropb)[2 * lane + j];
#pragma unroll
                for (int k = 16 * j; k < RC && k < 16 * j + 16; ++k) {
                    const int kk = k & 15;
                    const unsigned w = (kk >> 2) == 0 ? d.x : ((kk >> 2) == 1 ? d.y : ((kk >> 2) == 2 ? d.z : d.w));
                    if ((w >> (8 * (kk & 3))) & 0xFFu) kept &= ~(1u << k);
                }
            }
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < RC; ++k) sum += ((kept >> k) & 1u) ? yc[k] : 0.0;
    double cntd = (double)__popc(kept);
    for (int i0 = RC * kWave + lane; i0 < M; i0 += kLongUnroll * kWave) {   // lists longer than the register cache
        double yl[kLongUnroll];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) yl[j] = yv[max(min(i0 + j * kWave, M - 1), 0)];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) {
            if (i0 + j * kWave >= M) continue;
            const double y = yl[j];
            const int bin = bin_of_table(y, edges);
            if (bin >= 0 && nearflag[bin] && dropped_by_single(y, bin, single, first_single)) continue;
            sum += y; cntd += 1.0;
        }
    }
    sum = wave_sum(sum);
    cntd = wave_sum(cntd);
    if constexpr (WW > 1) {
        if (lane == 0) { part[2 * wave_id()] = sum; part[2 * wave_id() + 1] = cntd; }
        __syncthreads();
        sum = 0.0; cntd = 0.0;
#pragma unroll
        for (int i = 0; i < WW; ++i) { sum += part[2 * i]; cntd += part[2 * i + 1]; }
        __syncthreads();                                // (the slots are used again for the squares)
    }
    MVOSR_RSTAMP(4);
    const int nkept = (int)cntd;
    R.n_kept = nkept;

    // the kept values in list order, packed into `scratch` (for the median fallback and the exact sums)
    auto pack_kept = [&]() {
        int nlist = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < Mall; i0 += kWave) {
            const int i = i0 + lane;
            bool keep = false;
            double y = 0.0;
            if (i < Mall) {
                y = yv_all[i];
                const int bin = bin_of_table(y, edges);
                keep = !(bin >= 0 && nearflag[bin] && dropped_by_single(y, bin, single, first_single));
            }
            const unsigned long long m = __ballot(keep);
            if (keep) scratch[nlist + __popcll(m & ((1ull << lane) - 1ull))] = y;
            nlist += __popcll(m);
        }
        return nlist;
    };

    if (!have_modes) {
        if constexpr (WW > 1) { if (wave_id() != 0) return R; }                             // (no barrier lies ahead on this branch)
        if (nkept == 0) { R.height = height_level; R.status = MVOSR_ST_LEVEL; return R; }   // :334-335
        // np.median (:333): pack the kept values into `scratch` (index <= source index, so in place
        // is safe when scratch == yv), then rank counting: the two middle order statistics
        pack_kept();
        __threadfence_block();                   // the wave's own stores, visible to all its lanes
        const int klo = (nkept - 1) >> 1, khi = nkept >> 1;
        double mlo = 0.0, mhi = 0.0;
        int flo = 0, fhi = 0;
        for (int i = lane; i < nkept; i += kWave) {
            const double yi = scratch[i];
            int rank = 0;
            for (int j = 0; j < nkept; ++j) {
                const double yj = scratch[j];
                rank += (yj < yi) || (yj == yi && j < i);
            }
            if (rank == klo) { mlo = yi; flo = 1; }
            if (rank == khi) { mhi = yi; fhi = 1; }
        }
        // exactly one lane holds each of them
        const unsigned long long blo = __ballot(flo), bhi = __ballot(fhi);
        mlo = readlane_d(mlo, (int)__ffsll((long long)blo) - 1);
        mhi = readlane_d(mhi, (int)__ffsll((long long)bhi) - 1);
        R.median = (klo == khi) ? mlo : (mlo + mhi) / 2.0;
        R.height = R.median; R.status = MVOSR_ST_MEDIAN;
        return R;
    }

    // third pass: standard deviation around the mean (np.std, :496)
    const double mean = sum / cntd;                                             // np.mean
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        const double d = ((kept >> k) & 1u) ? yc[k] - mean : 0.0;
        ss += d * d;
    }
    for (int i0 = RC * kWave + lane; i0 < M; i0 += kLongUnroll * kWave) {
        double yl[kLongUnroll];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) yl[j] = yv[max(min(i0 + j * kWave, M - 1), 0)];
#pragma unroll
        for (int j = 0; j < kLongUnroll; ++j) {
            if (i0 + j * kWave >= M) continue;
            const double y = yl[j];
            const int bin = bin_of_table(y, edges);
            if (bin >= 0 && nearflag[bin] && dropped_by_single(y, bin, single, first_single)) continue;
            const double d = y - mean;
            ss += d * d;
        }
    }
    ss = wave_sum(ss);
    if constexpr (WW > 1) {
        if (lane == 0) part[wave_id()] = ss;
        __syncthreads();
        if (wave_id() != 0) return R;                   // wavefront 0 finishes the frame
        ss = 0.0;
#pragma unroll
        for (int i = 0; i < WW; ++i) ss += part[i];
    }
    MVOSR_RSTAMP(5);

#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int b = lane + 64 * c;
        bool is_mode = false, is_min = false;
        if (b < kBins) {
            const int h = hz[c];
            if (b == 0 || b == kBins - 1) {
                is_mode = (h == mx);                                            // :454-458
                is_min = (h == mn);                                             // :433-437
            } else {
                const int lraw = hist[b - 1], rraw = hist[b + 1];
                const int hl_ = (lraw == 1) ? 0 : lraw, hr_ = (rraw == 1) ? 0 : rraw;
                is_mode = (h >= hl_) && (h >= hr_) && ((double)h >= P.mode_rel * (double)mx) && (h >= P.mode_min);   // :459-463
                is_min = (h <= hl_) && (h <= hr_) && !((h == hr_) && (h == hl_));                                   // :438-442
            }
        }
        modes.w[c] = __ballot(is_mode);
        mins.w[c] = __ballot(is_min);
    }
    R.n_modes = modes.runs();                                                   // :468-481 (edges 0.1 apart cluster, gap < 0.11)
    // last cluster = last run of consecutive mode bins (:338-340); int(edge*10) == bin+1
    const int i_last = modes.highest_below(kBins);
    int i_first = i_last;
    while (i_first > 0 && modes.test(i_first - 1)) --i_first;
    const int ml = i_first + 1, mr = i_last + 1;
    R.mode_left = ml; R.mode_right = mr;
    const double mode = (double)(ml + mr) / 2.0;                                // :340
    const int il = mins.highest_below(ml);                                      // :341,:343  bins 0..ml-1
    if (il < 0) { R.status = MVOSR_ST_ERR_LEFT; return R; }
    const int ir = mins.lowest_from(mr);                                        // :342,:344  bins mr..168
    if (ir < 0) { R.status = MVOSR_ST_ERR_RIGHT; return R; }
    const double right = bin_edge(ir + 1);
    double mean_o = mean;
    double sd = sqrt(ss / cntd);                                                // np.std
    // The sums above run in the wavefront's order, NumPy's in its pairwise order: mean and std agree to
    // ~1e-15 relative, which decides `skew > 0.3` the same way unless the two sides are that close — in
    // practice only for (nearly) constant lists, where std is rounding noise.  Then, and when exact
    // statistics are asked for, both sums are redone in NumPy's own order on the packed list.
    {
        const double var = ss / cntd, rms = sqrt(mean * mean + var);
        const double margin = fabs((mean - mode / 10.0) - P.skew_threshold * sd);
        const double bound = 1e-11 * (rms + sd) + 1e-22 * rms * rms / sd;
        if (!(margin > bound) || exact_stats) {
            const int nl = pack_kept();
            __threadfence_block();               // the wave's own stores, visible to all its lanes
            mean_o = np_pairwise_sum_cold(scratch, nl, 0.0, 0, np_stack) / (double)nl;    // np.mean
            sd = sqrt(np_pairwise_sum_cold(scratch, nl, mean_o, 1, np_stack) / (double)nl);
        }
    }
    const double skew = (mean_o - mode / 10.0) / sd;                            // :496
    R.mean = mean_o; R.std = sd; R.skew = skew;
    if (skew > P.skew_threshold) { R.height = right; R.status = MVOSR_ST_RIGHT; }   // :348-352
    else { R.height = mode / 10.0; R.status = MVOSR_ST_MODE; }                      // :354
    return R;
}

#ifndef MVOSR_ROAD_MINW
#define MVOSR_ROAD_MINW 1
#endif
// LIST: the frames of a.list, grid-strided (the rare second pass over the frames that ended on the fallback level); otherwise
// frame first_frame + wavefront index, no loop (a loop around the body costs the product variant 57 VGPRs, i.e. half its occupancy)
template <bool LIST, int WW = 1>
__global__ __launch_bounds__(kRoadWaves *kWave, (LIST ? 1 : 4)) void road_model_kernel(const RoadArgs a) {
    static_assert(kRoadRC >= 16 && kRoadRC <= kDropStride, "verdict bytes per lane; the keep masks are 32 bits");
    static_assert(WW == 1 || (WW == kRoadWaves && !LIST), "wide variant: the whole workgroup on one frame");
    __shared__ int hist_all[kRoadWaves][2][176];
    __shared__ double2 edges[kBins + 1];
    __shared__ uint16_t slots_all[kRoadWaves][kRoadRC * kWave];
    __shared__ __attribute__((aligned(16))) uint8_t drop_all[kRoadWaves][kDropStride * kWave];
    __shared__ double part[2 * kRoadWaves];
    __shared__ __attribute__((aligned(8))) int np_stack_all[kRoadWaves][kNpStackInts];
    for (int k = threadIdx.x; k < kBins; k += kRoadWaves * kWave) { double2 e; e.x = bin_edge(k); e.y = bin_edge(k + 1); edges[k] = e; }
    __syncthreads();
    const int64_t slot = WW > 1 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * kRoadWaves + wave_id();
    auto one_frame = [&](const int64_t f) {
    if (a.pending_only && a.o.status[f] != kStPending) return;          // (wide: the same answer in every wavefront)
    MVOSR_STAMP_DECL
    MVOSR_RSTAMP(0);
    const int M = a.cnt[f];
    const int64_t off = a.off[f];
    const double hl = a.height_level ? a.height_level[f] : nan("");
    MVOSR_RSTAMP(1);
    // values per lane kept in registers: as few as the list (wide: this wavefront's part of it) needs (the passes over
    // them are branch-free, so a short list would otherwise pay for sixteen rows of padding)
    int *h0 = hist_all[WW > 1 ? 0 : wave_id()][0], *h1 = hist_all[wave_id()][1];
    int32_t *gh = a.o.hist ? a.o.hist + f * 2 * kBins : nullptr;
    const bool ex = a.o.stats != nullptr;
    int lo = 0, hi = M;
    if constexpr (WW > 1) {
        const int q = ((M + WW * kWave - 1) / (WW * kWave)) * kWave;     // values per wavefront, whole rows of 64
        lo = min(M, wave_id() * q);
        hi = min(M, lo + q);
    }
    const int Mp = hi - lo;
#define MVOSR_ROAD_CALL(RC_) road_wave<RC_, WW>(h0, h1, slots_all[wave_id()], drop_all[wave_id()], edges, a.y + off, a.scratch + off, M, hl, a.P, gh, ex, lo, hi, part, np_stack_all[wave_id()] MVOSR_STAMP_PASS)
    const RoadResult R = (Mp <= 4 * kWave) ? MVOSR_ROAD_CALL(4)
                       : (Mp <= 8 * kWave) ? MVOSR_ROAD_CALL(8)
                       : (Mp <= 12 * kWave) ? MVOSR_ROAD_CALL(12)
                       : (kRoadRC == 16 || Mp <= 16 * kWave) ? MVOSR_ROAD_CALL(16)
                                          : MVOSR_ROAD_CALL(kRoadRC);
#undef MVOSR_ROAD_CALL
    MVOSR_RSTAMP(6);
#ifdef MVOSR_STAMPS
    if (lane_id() == 0 && a.o.hist) { unsigned long long *d = reinterpret_cast<unsigned long long *>(a.o.hist + f * 2 * kBins) + 16; for (int i = 0; i < 8; ++i) d[i] = stamps[i]; }
#endif
    if (WW > 1 && wave_id() != 0) return;                               // wavefront 0 holds the frame's result
    if (R.status == MVOSR_ST_LEVEL && a.level_redo) {
        // the frame's height IS height_level, which the HOT scale kernel summed in its own order: the EXACT pass redoes it
        // (a frame of the exact mask is on the list already: a second entry would run two workgroups of the exact pass on one frame —
        // its list is compacted in place — and could push the list beyond its n_frames + 1 slots)
        if (lane_id() == 0) {
            if (!(a.listed_mask && a.listed_mask[f])) a.level_redo[1 + atomicAdd(a.level_redo, 1)] = (int32_t)f;
            a.o.status[f] = kStRedo;
        }
        return;
    }
    if (lane_id() == 0) {
        double height = nan(""), raw = nan("");
        if (R.status == MVOSR_ST_NO_FLAT) raw = a.P.absolute_reference / hl;                        // :421
        else if (R.status <= MVOSR_ST_LEVEL) { height = R.height; raw = a.P.absolute_reference / height; }   // :419
        a.o.raw_scale[f] = raw; a.o.height[f] = height; a.o.status[f] = R.status;
        if (!a.pending_only && a.o.height_level) a.o.height_level[f] = hl;
        if (a.o.counts) {
            int32_t *c = a.o.counts + f * MVOSR_N_COUNTS;
            if (!a.pending_only) { c[MVOSR_CNT_VALID] = M; c[MVOSR_CNT_TRI_PITCH] = 0; c[MVOSR_CNT_TRI_VALID] = 0; }
            c[MVOSR_CNT_SELECTED] = R.n_sel; c[MVOSR_CNT_KEPT] = R.n_kept; c[MVOSR_CNT_MODES] = R.n_modes;
            c[MVOSR_CNT_MODE_LEFT] = R.mode_left; c[MVOSR_CNT_MODE_RIGHT] = R.mode_right;
        }
        if (a.o.stats) { double *st = a.o.stats + 4 * f; st[0] = R.mean; st[1] = R.std; st[2] = R.skew; st[3] = R.median; }
    }
    };
    if constexpr (LIST) {
        const int64_t n_todo = (int64_t)a.list[0];
        for (int64_t it = slot; it < n_todo; it += (int64_t)gridDim.x * kRoadWaves) one_frame((int64_t)a.list[1 + it]);
    } else {
        if (slot < a.n_frames) one_frame(a.first_frame + slot);
    }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct KArgs {
    mvosr_params P;
    mvosr_batch b;
    mvosr_outputs o;
    PitchTest pt;
    int64_t first_frame;
    const double *height_level_in;
    int debug_skip;          // ablation bits for profiling runs (env MVOSR_DEBUG_SKIP); 0 in production
    int32_t *redo;           // workspace: redo[0] = number of frames the HOT kernel left for the EXACT pass, redo[1..] their indices
    int redo_pass;           // EXACT kernels: 1 = process the redo list (grid-strided), 0 = frame first_frame + blockIdx.x
    double *ysel;            // workspace plane (laid out like x): the selected y' of every frame, dense
    int32_t *nsel;           // workspace [F]: how many
    // size-class launches of a ragged batch (LIST kernels): the class's frame list and its length
    const int32_t *cls_list;
    const int32_t *cls_cnt;
};

__device__ __forceinline__ void write_counts(const KArgs &a, int64_t f, int nvalid, int npitch, int ntv, const RoadResult &R) {
    if (!a.o.counts) return;
    int32_t *c = a.o.counts + f * MVOSR_N_COUNTS;
    c[MVOSR_CNT_VALID] = nvalid; c[MVOSR_CNT_TRI_PITCH] = npitch; c[MVOSR_CNT_TRI_VALID] = ntv;
    c[MVOSR_CNT_SELECTED] = R.n_sel; c[MVOSR_CNT_KEPT] = R.n_kept; c[MVOSR_CNT_MODES] = R.n_modes;
    c[MVOSR_CNT_MODE_LEFT] = R.mode_left; c[MVOSR_CNT_MODE_RIGHT] = R.mode_right;
}

// Frames the sweeps never see: nothing below the vanishing row / no second triangulation (build-side
// MVOSR_ST_ERR_EMPTY), and the reference's "no enough feature for triangulation" branch
// (scale_calculator.py:263-270): with at most 3 features below the vanishing row it skips the second
// triangulation and the scale comes from the previous frame's height_level (:420-422) — a cross-frame
// quantity, resolved by the host's push step from the MVOSR_ST_TOO_FEW status.
__device__ __forceinline__ bool early_frame_exit(const KArgs &a, int64_t f, int n, int t2n) {
    // (exactly 3: with 1 or 2 points the reference's first Delaunay call raises QhullError at :257 — the host path reports
    // that from its own call; a C-ABI caller gets MVOSR_ST_ERR_EMPTY, never a silent scale)
    const bool too_few = n == 3;
    // a frame larger than the batch header says (max_feat sized this launch's LDS and variant): refused, not processed
    const bool oversize = n > a.b.max_feat;
    if (!(n <= 2 || too_few || t2n <= 0 || oversize)) return false;
    if (threadIdx.x == 0) {
        RoadResult R;
        R.n_sel = R.n_kept = R.n_modes = 0; R.mode_left = R.mode_right = -1;
        a.o.raw_scale[f] = nan(""); a.o.height[f] = nan(""); a.o.height_level[f] = nan("");
        a.o.status[f] = oversize ? MVOSR_ST_ERR_MASK : (too_few ? MVOSR_ST_TOO_FEW : MVOSR_ST_ERR_EMPTY);
        a.nsel[f] = 0;
        write_counts(a, f, too_few ? n : 0, 0, 0, R);
    }
    return true;
}

// Tail shared by the scale kernels: status, the dense list of selected y' for the road-model kernel
// (count per wave slice -> barrier -> ordered store), per-frame outputs.  HOT mode hands a frame to the EXACT pass
// when its result could depend on the last bits of height_level: a flat triangle within rounding of the level
// (S.near), or so few selected points that the level itself may become the height — nothing selected
// (:277-279,:421) or every point alone in its bin, at most one per bin (:334-335).
template <int BW, int MODE>
__device__ __forceinline__ void frame_tail(const KArgs &a, const Smem &s, int64_t f, int64_t off, int nvalid, bool refused,
                                           SelectResult &S, RoadResult &R) {
    constexpr int B = BW * kWave;
    const int tid = threadIdx.x;
    int status = kStPending;
    double raw = nan("");
    int nsel = 0;
    if (refused || S.bad) {
        status = MVOSR_ST_ERR_MASK;
    } else if (S.singular) {
        status = MVOSR_ST_ERR_SINGULAR;
    } else {
        const int w = wave_id(), lane = lane_id();
        const int per = ((nvalid + B - 1) / B) * kWave;
        const int begin = w * per, end = min(nvalid, begin + per);
        int cnt = 0;
        for (int j0 = begin; j0 < end; j0 += kWave) {
            const int j = j0 + lane;
            const bool sel = (j < end) && ((s.sel[j >> 5] >> (j & 31)) & 1u);
            if (a.o.selected && j < end) a.o.selected[off + j] = (uint8_t)sel;                   // :247
            cnt += __popcll(__ballot(sel));
        }
        if (lane == 0) s.misc[M_WCNT + w] = cnt;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int i = 0; i < BW; ++i) { const int c = s.misc[M_WCNT + i]; if (i < w) base += c; nsel += c; }
        if constexpr (MODE == MODE_HOT) {
            if (S.near || nsel == 0) {
                if (tid == 0) { a.redo[1 + atomicAdd(a.redo, 1)] = (int32_t)f; a.o.status[f] = kStRedo; a.nsel[f] = 0; }
                return;
            }
        }
        double *dst = a.ysel + off;
        for (int j0 = begin; j0 < end; j0 += kWave) {
            const int j = j0 + lane;
            const bool sel = (j < end) && ((s.sel[j >> 5] >> (j & 31)) & 1u);
            const unsigned long long m = __ballot(sel);
            if (sel) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = s.Y[j];
            base += __popcll(m);
        }
        if (nsel == 0) { status = MVOSR_ST_NO_FLAT; raw = a.P.absolute_reference / S.height_level; }   // :277-279,:421
    }
    if (tid == 0) {
        a.o.raw_scale[f] = raw;
        a.o.height[f] = nan("");
        a.o.height_level[f] = S.height_level;
        a.o.status[f] = status;                  // kStPending: the road-model kernel finishes the frame
        a.nsel[f] = nsel;
        R.n_sel = nsel;
        write_counts(a, f, nvalid, S.n_pitch, S.n_tri_valid, R);
        if (a.o.stats) { double *st = a.o.stats + 4 * f; st[0] = st[1] = st[2] = st[3] = nan(""); }
    }
}

// The frames of a launch: frame first_frame + blockIdx.x — or, in the EXACT pass over the redo list, the list's entries
// strided over the grid (the list is short or empty; its length is only known on the device) — or, in a size-class
// launch (LIST), entry blockIdx.x of the class's list (the grid covers the longest possible list; workgroups beyond the
// list's end leave at once.  A grid of resident workgroups looping over the list was tried first: the loop makes the
// compiler keep the kernel arguments live in registers — 69 -> 109 VGPRs for the 4-wavefront variant).
template <int MODE, bool LIST, class Body>
__device__ __forceinline__ void for_frames(const KArgs &a, Body body) {
    if constexpr (MODE == MODE_EXACT) {
        if (a.redo_pass) {
            const int count = a.redo[0];
            for (int i = blockIdx.x; i < count; i += gridDim.x) {
                body((int64_t)a.redo[1 + i]);
                __syncthreads();                  // the next frame reuses the workgroup's LDS
            }
            return;
        }
    }
    if constexpr (LIST) {
        if ((int)blockIdx.x < *a.cls_cnt) body((int64_t)a.cls_list[blockIdx.x]);
        return;
    }
    body(a.first_frame + blockIdx.x);
}

#ifndef MVOSR_MINW
#define MVOSR_MINW 1
#endif
#ifndef MVOSR_HOTW
#define MVOSR_HOTW 6
#endif
// (the 8-wavefront product variants must stay within 80 VGPRs: three workgroups per CU are six wavefronts per SIMD)
template <int WAVES, int SC, int MODE, bool LIST = false>
__global__ __launch_bounds__(WAVES *kWave, (WAVES == 8 && MODE == MODE_HOT ? MVOSR_HOTW : MVOSR_MINW)) void scale_frames_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for_frames<MODE, LIST>(a, [&](const int64_t f) {
    constexpr int B = WAVES * kWave;
    MVOSR_STAMP_DECL
    MVOSR_STAMP(11);                               // (before the frame's offsets and counts are asked for)
    const int n = a.b.feat_cnt[f];
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f], t2b = a.b.tri2_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f), t2n = tri_rows(a.b.tri2_off, a.b.tri2_cnt, f);
    const Smem s = carve(smem, n, WAVES);
#ifdef MVOSR_STAGGER
    // Experiment (round 6, LABNOTES 10.10): the launch's first generation of workgroups starts spread over MVOSR_STAGGER
    // ticks of s_memtime (100 MHz) instead of all at once, so that equal-length frames do not march through their phases in step.
    if (MODE == MODE_HOT && !LIST && blockIdx.x < 768u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = ((blockIdx.x * 2654435761u) >> 12) % (unsigned)(MVOSR_STAGGER);
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
#ifdef MVOSR_STAMPS
    if (n < 0) return;                             // (never: makes the stamp below wait for the frame's counts)
#endif
    MVOSR_STAMP(0);
#ifdef MVOSR_STAMPS
    if (threadIdx.x == 0) {                        // where the workgroup ran: HW_ID (CU, SE) and XCC_ID
        stamps[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif

    RoadResult R;
    R.height = nan(""); R.n_sel = R.n_kept = R.n_modes = 0; R.mode_left = R.mode_right = -1;
    R.mean = R.std = R.skew = R.median = nan("");
    if (early_frame_exit(a, f, n, t2n)) return;
    int bad = 0;
    TriChunk<B> tc2;
    const int nvalid = phase_vote<WAVES, SC>(s, n, a.b.x + off, a.b.y + off, a.b.z + off, a.b.v + off, a.b.tri1, t1b, t1n,
                                             a.P.cos_pitch, a.P.sin_pitch,
                                             a.o.vote_counters ? a.o.vote_counters + off : nullptr, bad,
                                             a.b.tri2, t2b, t2n, tc2, a.P.vote_mode == MVOSR_VOTE_FIXED, a.debug_skip MVOSR_STAMP_PASS);
    const bool mask_mismatch = a.b.n2_expected && a.b.n2_expected[f] != nvalid;

    SelectResult S;
    S.height_level = nan(""); S.n_pitch = S.n_tri_valid = 0; S.singular = 0; S.bad = 1; S.n_steep = 0; S.near = 0;
    // more rows than a vertex's 16-bit vote counter can absorb (|votes| < 32766): not a triangulation of this frame
    const bool too_many_rows = t1n > kMaxVoteRows;
    if (!mask_mismatch && !too_many_rows)
        S = phase_select<WAVES, MODE>(s, nvalid, a.b.tri2, t2b, t2n, tc2, a.pt, a.o.tri_normals, a.o.tri_pitch_deg,
                                      a.o.tri_heights, bad, a.ysel ? a.ysel + off : nullptr, n, a.debug_skip MVOSR_STAMP_PASS);
    if (a.debug_skip & 8) S.bad = 1;
    frame_tail<WAVES, MODE>(a, s, f, off, nvalid, mask_mismatch || too_many_rows, S, R);
    MVOSR_STAMP(9);
#ifdef MVOSR_STAMPS
    if (threadIdx.x == 0 && a.o.hist) { unsigned long long *d = reinterpret_cast<unsigned long long *>(a.o.hist + f * 2 * kBins); for (int i = 0; i < 12; ++i) d[i] = stamps[i]; }
#endif
    });
}

// ---------------------------------------------------------------------------------------------
// Dense frames (more features than fit LDS in fp64, e.g. N = 20000): same algorithm, but the
// triangle sweeps gather through L1/L2 ("L2-gather variant"): the vote from the caller's planes
// (remap on the fly), the selection from the compacted survivors' planes in a per-frame workspace.  LDS keeps only what is hit by atomics: the 16-bit
// vote counters and the selected bit-set.  One workgroup of 16 wavefronts per frame (128 flag bits per
// thread: up to 131072 triangles, i.e. the 65535-feature limit of the 16-bit counters' index space).  Survivors are
// compacted into a second workspace copy (no in-place hazard, no registers held across barriers).
// ---------------------------------------------------------------------------------------------
struct DenseWs { double2 *P2; double *Y2; };      // per-batch planes of the survivors, laid out like x (feat_off)

__host__ __device__ inline uint32_t dense_lds_bytes(int n, int waves) {
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    return align16(2u * npad + 16u) + align16(4u * ((uint32_t)(n + 31) / 32u)) + 8u * (uint32_t)(kRedSlots * 2 * waves) + 4u * 32u;
}

template <int DW, bool FUSED>
__device__ __forceinline__ int phase_vote_dense(uint32_t *c32, int *misc, int n, const double *gx, const double *gy, const double *gz,
                                                const double *gv, const int32_t *tri1, int64_t t1_begin, int t1_count,
                                                double cp, double sp, int32_t *g_counters, int &bad,
                                                double2 *P2, double *Y2, bool fixed) {
    constexpr int B = DW * kWave;
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const uint16_t *c16 = reinterpret_cast<const uint16_t *>(c32);
    const uint32_t ones = ((uint32_t)(kCounterBias + 1) << 16) | (uint32_t)(kCounterBias + 1);     // np.ones, :153
    for (int i = tid; i < ((n + 1) >> 1); i += B) c32[i] = ones;
    __threadfence_block();
    __syncthreads();
    const bool checked = t1_count > kMaxVoteRows;
    for (int t = tid; t < t1_count; t += B) {
        const TriIds q = load_tri(tri1 + 3 * t1_begin, t);
        if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) { bad = 1; continue; }
        // {v, z'} gathered from the caller's planes through L1/L2, remap (:392) on the fly: a staged copy
        // would cost a write and two more reads of every feature in HBM traffic
        double2 p0, p1, p2;
        p0.x = gv[q.a]; p0.y = gy[q.a] * sp + gz[q.a] * cp;
        p1.x = gv[q.b]; p1.y = gy[q.b] * sp + gz[q.b] * cp;
        p2.x = gv[q.c]; p2.y = gy[q.c] * sp + gz[q.c] * cp;
        const bool pa = (p0.x - p1.x) * (p0.y - p1.y) > 0.0;       // :107,:110
        const bool pb = (p0.x - p2.x) * (p0.y - p2.y) > 0.0;       // :108,:113  (marks vertices 0 and 1, as the reference does)
        const bool pc = (p1.x - p2.x) * (p1.y - p2.y) > 0.0;       // :109,:116
        const VoteFlags vf = vote_flags(pa, pb, pc, fixed);
        const bool f0 = vf.f0, f1 = vf.f1, f2 = vf.f2;
        const int s0 = (q.a & 1) * 16, s1 = (q.b & 1) * 16, s2 = (q.c & 1) * 16;
        const uint32_t u0 = 1u << s0, u1 = 1u << s1, u2 = 1u << s2;
        if (!checked) {
            atomicAdd(&c32[q.a >> 1], f0 ? 0u - u0 : u0);
            atomicAdd(&c32[q.b >> 1], f1 ? 0u - u1 : u1);
            atomicAdd(&c32[q.c >> 1], f2 ? 0u - u2 : u2);
        } else {
            // the frame has enough rows for one vertex to push its 16-bit half over an end: look at the value each
            // update found (a -1 on 0 borrows from, a +1 on 0xFFFF carries into, the neighbouring feature's half)
            const uint32_t o0 = (atomicAdd(&c32[q.a >> 1], f0 ? 0u - u0 : u0) >> s0) & 0xFFFFu;
            const uint32_t o1 = (atomicAdd(&c32[q.b >> 1], f1 ? 0u - u1 : u1) >> s1) & 0xFFFFu;
            const uint32_t o2 = (atomicAdd(&c32[q.c >> 1], f2 ? 0u - u2 : u2) >> s2) & 0xFFFFu;
            if (o0 == (f0 ? 0u : 0xFFFFu) || o1 == (f1 ? 0u : 0xFFFFu) || o2 == (f2 ? 0u : 0xFFFFu)) bad = 1;
        }
    }
    __syncthreads();
    const int per = ((n + B - 1) / B) * kWave;
    const int begin = w * per, end = min(n, begin + per);
    int cnt = 0;
    for (int i0 = begin; i0 < end; i0 += kWave) {
        const int i = i0 + lane;
        bool keep = false;
        if (i < end) {
            const int c = (int)c16[i] - kCounterBias;
            keep = c >= 0;                                                    // :166
            if (g_counters) g_counters[i] = c;
        }
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) misc[M_WCNT + w] = cnt;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < DW; ++i) { const int c = misc[M_WCNT + i]; if (i < w) base += c; total += c; }
    if constexpr (FUSED) {
        for (int i0 = begin; i0 < end; i0 += kWave) {
            const int i = i0 + lane;
            const bool keep = (i < end) && ((int)c16[i] - kCounterBias >= 0);
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                const double yy = gy[i], zz = gz[i];
                double2 pv; pv.x = gx[i]; pv.y = yy * sp + zz * cp;           // {x, z'} of the survivor (:264-265, :392)
                P2[pos] = pv;
                Y2[pos] = yy * cp - zz * sp;                                  // :391
            }
            base += __popcll(m);
        }
        __threadfence_block();
    }
    __syncthreads();
    return total;
}

struct DenseArgs { KArgs k; DenseWs ws; };

template <int DW, int MODE>
__global__ __launch_bounds__(DW *kWave) void scale_frames_dense_kernel(const DenseArgs da) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs &a = da.k;
    for_frames<MODE, false>(a, [&](const int64_t f) {
    constexpr int B = DW * kWave;
    const int tid = threadIdx.x;
    const int n = a.b.feat_cnt[f];
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f], t2b = a.b.tri2_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f), t2n = tri_rows(a.b.tri2_off, a.b.tri2_cnt, f);
    RoadResult R;
    R.height = nan(""); R.n_sel = R.n_kept = R.n_modes = 0; R.mode_left = R.mode_right = -1;
    R.mean = R.std = R.skew = R.median = nan("");
    if (early_frame_exit(a, f, n, t2n)) return;
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    Smem s;
    s.c32 = reinterpret_cast<uint32_t *>(smem);
    s.c16 = reinterpret_cast<uint16_t *>(smem);
    s.sel = reinterpret_cast<uint32_t *>(smem + align16(2u * npad + 16u));
    s.red = reinterpret_cast<double *>(smem + align16(2u * npad + 16u) + align16(4u * ((uint32_t)(n + 31) / 32u)));
    s.misc = reinterpret_cast<int *>(s.red + kRedSlots * 2 * DW);
    s.hist = nullptr;
    s.P = da.ws.P2 + off;              // what the second triangulation indexes: the compacted copy
    s.Y = da.ws.Y2 + off;
    for (int i = tid; i < (n + 31) / 32; i += B) s.sel[i] = 0u;
    int bad = 0;
    const int nvalid = phase_vote_dense<DW, true>(s.c32, s.misc, n, a.b.x + off, a.b.y + off, a.b.z + off, a.b.v + off, a.b.tri1, t1b, t1n,
                                              a.P.cos_pitch, a.P.sin_pitch, a.o.vote_counters ? a.o.vote_counters + off : nullptr, bad,
                                              da.ws.P2 + off, da.ws.Y2 + off, a.P.vote_mode == MVOSR_VOTE_FIXED);
    const bool mask_mismatch = a.b.n2_expected && a.b.n2_expected[f] != nvalid;
    SelectResult S;
    S.height_level = nan(""); S.n_pitch = S.n_tri_valid = 0; S.singular = 0; S.bad = 1; S.n_steep = 0; S.near = 0;
    if (!mask_mismatch) {
        TriChunk<B> tc2;
        tc2.load(a.b.tri2, t2b, t2n, 0, tid);
        S = phase_select<DW, MODE, 2>(s, nvalid, a.b.tri2, t2b, t2n, tc2, a.pt, a.o.tri_normals, a.o.tri_pitch_deg,
                                      a.o.tri_heights, bad, nullptr, 0, 0);
    }
    frame_tail<DW, MODE>(a, s, f, off, nvalid, mask_mismatch, S, R);
    });
}

// ---------------------------------------------------------------------------------------------
// Dense frames whose second triangulation is numbered over the frame's FEATURES
// (mvosr_batch.tri2_ids == MVOSR_TRI2_FEATURES; the host relabels SciPy's rows with the vote mask
// it already holds).  Nothing is compacted and no workspace is touched: both selection sweeps
// gather x, y, z from the caller's planes and remap on the fly, the vote counters stay in LDS to
// check that every vertex is a survivor, and the selected bit-set is indexed by feature.  HBM
// traffic: the planes twice (two gather sweeps) and the triangle rows — about a quarter less than
// the variant that writes and re-reads the survivors' planes.
// ---------------------------------------------------------------------------------------------
template <int DW, int MODE>
__global__ __launch_bounds__(DW *kWave) void scale_frames_dense_feat_kernel(const DenseArgs da) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs &a = da.k;
    for_frames<MODE, false>(a, [&](const int64_t f) {
    constexpr int B = DW * kWave;
    constexpr bool FULL = MODE == MODE_FULL;
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const int n = a.b.feat_cnt[f];
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f], t2b = a.b.tri2_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f), t2n = tri_rows(a.b.tri2_off, a.b.tri2_cnt, f);
    RoadResult R;
    R.height = nan(""); R.n_sel = R.n_kept = R.n_modes = 0; R.mode_left = R.mode_right = -1;
    R.mean = R.std = R.skew = R.median = nan("");
    if (early_frame_exit(a, f, n, t2n)) return;
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    uint32_t *c32 = reinterpret_cast<uint32_t *>(smem);
    const uint16_t *c16 = reinterpret_cast<const uint16_t *>(smem);
    uint32_t *sel = reinterpret_cast<uint32_t *>(smem + align16(2u * npad + 16u));
    double *red = reinterpret_cast<double *>(smem + align16(2u * npad + 16u) + align16(4u * ((uint32_t)(n + 31) / 32u)));
    int *misc = reinterpret_cast<int *>(red + kRedSlots * 2 * DW);
    const double *gx = a.b.x + off, *gy = a.b.y + off, *gz = a.b.z + off;
    const double cp = a.P.cos_pitch, sp = a.P.sin_pitch;
    for (int i = tid; i < (n + 31) / 32; i += B) sel[i] = 0u;
    int bad = 0;
    const int nvalid = phase_vote_dense<DW, false>(c32, misc, n, nullptr, gy, gz, a.b.v + off, a.b.tri1, t1b, t1n, cp, sp,
                                                   a.o.vote_counters ? a.o.vote_counters + off : nullptr, bad, nullptr, nullptr,
                                                   a.P.vote_mode == MVOSR_VOTE_FIXED);
    const bool mask_mismatch = a.b.n2_expected && a.b.n2_expected[f] != nvalid;
    SelectResult S;
    S.height_level = nan(""); S.n_pitch = S.n_tri_valid = 0; S.singular = 0; S.bad = 1; S.n_steep = 0; S.near = 0;
    const int32_t *tri = a.b.tri2 + 3 * t2b;
    // a row's vertices from the caller's planes (remap on the fly); legal only if every vertex survived the vote
    auto fetch = [=](const TriIds q, double &x0, double &y0, double &z0, double &x1, double &y1, double &z1,
                     double &x2, double &y2, double &z2) -> bool {
        const bool ok = (unsigned)q.a < (unsigned)n && (unsigned)q.b < (unsigned)n && (unsigned)q.c < (unsigned)n &&
                        c16[q.a] >= kCounterBias && c16[q.b] >= kCounterBias && c16[q.c] >= kCounterBias;      // survivors only (:166)
        if (!ok) return false;
        const double ya = gy[q.a], za = gz[q.a], yb = gy[q.b], zb = gz[q.b], yc = gy[q.c], zc = gz[q.c];
        x0 = gx[q.a]; x1 = gx[q.b]; x2 = gx[q.c];
        y0 = ya * cp - za * sp; y1 = yb * cp - zb * sp; y2 = yc * cp - zc * sp;       // :391
        z0 = ya * sp + za * cp; z1 = yb * sp + zb * cp; z2 = yc * sp + zc * cp;       // :392
        return true;
    };
    if (!mask_mismatch) {
        unsigned long long flat = 0ull, flat_hi = 0ull;      // bit kk: my kk-th triangle has pitch_deg < thr
        double hsum = 0.0, hcnt = 0.0, habs = 0.0;
        int npitch = 0, singular = 0, ntv = 0, near = 0;
        if (t2n > 128 * B) bad = 1;                          // more rows than the per-thread flag words can name
        const int t2s = bad ? 0 : t2n;
        TriIds cur = {0, 0, 0};
        if (tid < t2s) cur = load_tri(tri, tid);
        for (int base = 0, kk = 0; base < t2s; base += B, ++kk) {
            const TriIds q = cur;
            if (base + B + tid < t2s) cur = load_tri(tri, base + B + tid);
            if (base + tid >= t2s) continue;
            double x0, y0, z0, x1, y1, z1, x2, y2, z2;
            if (!fetch(q, x0, y0, z0, x1, y1, z1, x2, y2, z2)) { bad = 1; continue; }
            const double h = div3((y0 + y1) + y2);                                               // :238
            const int r = classify_triangle<FULL>(x0, y0, z0, x1, y1, z1, x2, y2, z2, h, a.pt, a.o.tri_normals, a.o.tri_pitch_deg,
                                                  a.o.tri_heights, t2b + base + tid);
            if (r & 4) singular = 1;
            if (r & 2) { hsum += h; hcnt += 1.0; if constexpr (MODE == MODE_HOT) habs += fabs(h); }     // :240
            if (r & 1) {
                if (kk < 64) flat |= 1ull << (kk & 63); else flat_hi |= 1ull << (kk & 63);
                ++npitch;
            }
        }
        cur = {0, 0, 0};
        if (tid < t2s) cur = load_tri(tri, tid);
        if constexpr (MODE == MODE_HOT) block_sum3<DW>(hsum, hcnt, habs, red + R_SEL_H * 2 * DW, red + R_SEL_ABS * 2 * DW);
        else block_sum2<DW>(hsum, hcnt, red + R_SEL_H * 2 * DW);
        double hl = hsum / hcnt;                      // np.mean of an empty set -> 0/0 = NaN, like :240
        const double guard = kLevelGuard * (habs / hcnt);
        if constexpr (MODE != MODE_HOT)
            hl = exact_height_level<FULL>(tri, t2s, (int)hcnt, a.pt, fetch, a.b.tri2_order ? a.b.tri2_order + t2b : nullptr);
        for (int base = 0, kk = 0; base < t2s; base += B, ++kk) {
            const TriIds q = cur;
            if (base + B + tid < t2s) cur = load_tri(tri, base + B + tid);
            const unsigned long long fw = kk < 64 ? flat : flat_hi;
            if (base + tid < t2s && ((fw >> (kk & 63)) & 1ull)) {
                const double y0 = gy[q.a] * cp - gz[q.a] * sp, y1 = gy[q.b] * cp - gz[q.b] * sp, y2 = gy[q.c] * cp - gz[q.c] * sp;
                const double h = div3((y0 + y1) + y2);
                if constexpr (MODE == MODE_HOT) { if (fabs(h - hl) <= guard) near = 1; }
                if (h > hl) {                                                                    // :243-244
                    ++ntv;
                    atomicOr(&sel[q.a >> 5], 1u << (q.a & 31));                                  // :247
                    atomicOr(&sel[q.b >> 5], 1u << (q.b & 31));
                    atomicOr(&sel[q.c >> 5], 1u << (q.c & 31));
                }
            }
        }
        int sn = singular | (near << 16);
        block_sum4i<DW>(npitch, ntv, sn, bad, red + R_SEL_CNT * 2 * DW);   // also orders the atomicOr's
        S.height_level = hl; S.n_steep = (int)hcnt; S.near = sn >> 16;
        S.n_pitch = npitch; S.n_tri_valid = ntv; S.singular = sn & 0xFFFF; S.bad = bad;
    }
    int status = kStPending;
    double raw = nan("");
    int nsel = 0;
    if (mask_mismatch || S.bad) {
        status = MVOSR_ST_ERR_MASK;
    } else if (S.singular) {
        status = MVOSR_ST_ERR_SINGULAR;
    } else {
        // per wave slice of the features: survivors before it (for the `selected` output, which is indexed
        // by survivor) and selected ones before it (for the dense list of y' handed to the road-model kernel)
        const int per = ((n + B - 1) / B) * kWave;
        const int begin = w * per, end = min(n, begin + per);
        int nv = 0, ns = 0;
        for (int i0 = begin; i0 < end; i0 += kWave) {
            const int i = i0 + lane;
            const bool valid = (i < end) && c16[i] >= kCounterBias;
            const bool on = valid && ((sel[i >> 5] >> (i & 31)) & 1u);
            nv += __popcll(__ballot(valid));
            ns += __popcll(__ballot(on));
        }
        if (lane == 0) { misc[M_WCNT + w] = ns; misc[M_WCNT + DW + w] = nv; }
        __syncthreads();
        int base_s = 0, base_v = 0;
#pragma unroll
        for (int i = 0; i < DW; ++i) { const int c = misc[M_WCNT + i]; if (i < w) { base_s += c; base_v += misc[M_WCNT + DW + i]; } nsel += c; }
        if constexpr (MODE == MODE_HOT) {
            if (S.near || nsel == 0) {           // the level's last bits may matter: leave the frame to the EXACT pass
                if (tid == 0) { a.redo[1 + atomicAdd(a.redo, 1)] = (int32_t)f; a.o.status[f] = kStRedo; a.nsel[f] = 0; }
                return;
            }
        }
        double *dst = a.ysel + off;
        for (int i0 = begin; i0 < end; i0 += kWave) {
            const int i = i0 + lane;
            const bool valid = (i < end) && c16[i] >= kCounterBias;
            const bool on = valid && ((sel[i >> 5] >> (i & 31)) & 1u);
            const unsigned long long mv = __ballot(valid), ms = __ballot(on);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (valid && a.o.selected) a.o.selected[off + base_v + __popcll(mv & below)] = (uint8_t)on;      // :247
            if (on) dst[base_s + __popcll(ms & below)] = gy[i] * cp - gz[i] * sp;
            base_v += __popcll(mv);
            base_s += __popcll(ms);
        }
        if (nsel == 0) { status = MVOSR_ST_NO_FLAT; raw = a.P.absolute_reference / S.height_level; }   // :277-279,:421
    }
    if (tid == 0) {
        a.o.raw_scale[f] = raw;
        a.o.height[f] = nan("");
        a.o.height_level[f] = S.height_level;
        a.o.status[f] = status;
        a.nsel[f] = nsel;
        R.n_sel = nsel;
        write_counts(a, f, nvalid, S.n_pitch, S.n_tri_valid, R);
        if (a.o.stats) { double *st = a.o.stats + 4 * f; st[0] = st[1] = st[2] = st[3] = nan(""); }
    }
    });
}

// ---------------------------------------------------------------------------------------------
// Dense frames, traffic-lean ("tiled") variant: every input byte is read from HBM ONCE, coalesced.
//
// The host lays a dense frame out for this kernel (packing.apply_tile_order): features sorted along the
// image axis of larger extent, both triangulations numbered over the frame's features, their rows sorted by
// smallest vertex, and a tile index — for tiles of kTileW consecutive features the first row whose smallest
// vertex lies in the tile.  A Delaunay triangle's vertices are then a few dozen positions apart (median 58,
// 99th percentile 180 at N = 20000), so the rows of tile k find their vertices in tiles k and k+1 — except in
// ~0.2 % of the rows (hull slivers), whose far vertices are fetched from global memory and whose contributions
// to those vertices wait in a short pending list until their tile arrives.
//
// One workgroup walks the tiles with a ring of kRing (three) tiles in LDS: per vertex {v, z'}, {x, y'} (fed by coalesced
// loads of the caller's planes, remap fused, the tile after next in flight in registers), a 32-bit vote counter,
// the largest height of its flat triangles as an order-preserving 64-bit key (LDS atomic max) and a "named by
// tri2" flag.  The vote and the selection sweep of a tile's rows run in the same step; nothing is compacted or
// staged.  When a tile leaves the ring its vertices are final: survivors are counted, tri2's claim that it names
// survivors only is checked, and the vertices with a flat triangle ("candidates", about a third) park {y', largest
// flat height} in a scratch plane.  "Some flat triangle at this vertex is higher than the level" (:243-247) — the
// reference's second pass over the triangles — is then one comparison per candidate once height_level is known.
// LDS: 80 KB whatever the frame size, two workgroups per CU.  With three slots a step has ONE barrier: tile k-1 retires and tile
// k+2 is stored into its slot while the rows of tile k (vertices in tiles k and k+1) are walked.
//
// HBM traffic per frame: planes 32 B x N + rows 12 B x (T1 + T2), each once, + 32 B per candidate: about 1.2 x the
// algorithmic bytes (the two-sweep gather variant: 2.2 x).
//
// HOT-mode only: a frame in which a candidate's height is within kLevelGuard of the level, whose level may become
// the result, or whose pending lists overflow goes on the redo list and the EXACT pass runs
// scale_frames_dense_feat_kernel on it; stage outputs select that kernel too.
// ---------------------------------------------------------------------------------------------
constexpr int kTileW = 512;                 // features per tile (MVOSR_TILE_W in the header; the host's index uses the same)
#ifndef MVOSR_TILED_WAVES
#define MVOSR_TILED_WAVES 8
#endif
constexpr int kTiledWaves = MVOSR_TILED_WAVES;
#ifndef MVOSR_TILED_RING
#define MVOSR_TILED_RING 3
#endif
constexpr int kRing = MVOSR_TILED_RING;     // tiles in the LDS ring: 2 = two barriers per tile (a tile retires and its successor is stored between them),
                                            // 3 = one (tile k-1 retires and tile k+2 takes its slot while the rows of tile k are still being walked)
constexpr int kPendCap = 1024;              // pending votes for vertices whose tile has not arrived yet (4 B each) ...
constexpr int kPendCapH = kRing == 3 ? 576 : 1024;      // ... and pending heights (12 B each): what fits two workgroups per CU
#ifndef MVOSR_TILE_ROWS
#define MVOSR_TILE_ROWS 2
#endif
constexpr int kTileRows = MVOSR_TILE_ROWS;  // rows per thread and triangulation prefetched for the coming step
#ifndef MVOSR_TILED_SUBC
#define MVOSR_TILED_SUBC 1
#endif
constexpr int kSubC = MVOSR_TILED_SUBC;     // vote counters per ring vertex (a lane adds to counter lane % kSubC): rows are sorted by smallest vertex, so the
                                            // lanes of one LDS atomic name the same vertices again and again, and updates of one word are serialised

struct TiledPlan { uint32_t ringA, ringB, ringH, ringC, used, pendV, pendH, toff, red, misc, total; };
__host__ __device__ inline TiledPlan tiled_plan(int n, int waves) {
    TiledPlan p;
    const uint32_t ntiles = (uint32_t)((n + kTileW - 1) / kTileW);
    constexpr uint32_t RW = (uint32_t)kRing * kTileW;            // vertices in the ring
    p.ringA = 0;                                                 // double2 {v, z'}
    p.ringB = p.ringA + 16u * RW;                                // double2 {x, y'}
    p.ringH = p.ringB + 16u * RW;                                // uint64 key of the largest flat height
    p.ringC = p.ringH + 8u * RW;                                 // int32 vote counters (kSubC per vertex)
    p.used = p.ringC + 4u * RW * kSubC;                          // uint8 "a tri2 row names this vertex"
    p.pendV = align16(p.used + RW);                              // vertex << 1 | (vote is -1)
    p.pendH = p.pendV + 4u * kPendCap;                           // key64 x kPendCapH, then vertex x kPendCapH
    p.toff = align16(p.pendH + 12u * kPendCapH);                 // the frame's tile index: 2 x (ntiles + 1) ints
    p.red = align16(p.toff + 8u * (ntiles + 2u));
    p.misc = p.red + 8u * (uint32_t)(kRedSlots * 2 * waves);
    p.total = p.misc + 4u * 64u;
    return p;
}

// order-preserving map of a double onto uint64 (0 is never produced)
__device__ __forceinline__ unsigned long long height_key64(double h) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(h);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double height_of_key64(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

// (two workgroups per CU: with 8 wavefronts each that is 4 per SIMD, i.e. at most 128 VGPRs)
template <int DW>
__global__ __launch_bounds__(DW *kWave, (DW == 8 ? 4 : 1)) void scale_frames_tiled_kernel(const DenseArgs da) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs &a = da.k;
    const bool fixed = a.P.vote_mode == MVOSR_VOTE_FIXED;
    constexpr int B = DW * kWave, W = kTileW, M = 2 * kTileW - 1, VPT = W / B > 0 ? W / B : 1;   // VPT: a tile's vertices per thread
    static_assert(W % B == 0 || B > W, "tile width / block size");
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const int64_t f = a.first_frame + blockIdx.x;
    const int n = a.b.feat_cnt[f];
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f], t2b = a.b.tri2_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f), t2n = tri_rows(a.b.tri2_off, a.b.tri2_cnt, f);
    RoadResult R;
    R.height = nan(""); R.n_sel = R.n_kept = R.n_modes = 0; R.mode_left = R.mode_right = -1;
    R.mean = R.std = R.skew = R.median = nan("");
    if (early_frame_exit(a, f, n, t2n)) return;
    const TiledPlan pl = tiled_plan(a.b.max_feat, DW);
    double2 *ringA = reinterpret_cast<double2 *>(smem + pl.ringA);
    double2 *ringB = reinterpret_cast<double2 *>(smem + pl.ringB);
    unsigned long long *ringH = reinterpret_cast<unsigned long long *>(smem + pl.ringH);
    int *ringC = reinterpret_cast<int *>(smem + pl.ringC);
    uint8_t *used = reinterpret_cast<uint8_t *>(smem + pl.used);
    int *pendV = reinterpret_cast<int *>(smem + pl.pendV);
    unsigned long long *pendH = reinterpret_cast<unsigned long long *>(smem + pl.pendH);
    int *pendHv = reinterpret_cast<int *>(pendH + kPendCapH);
    int *toff1 = reinterpret_cast<int *>(smem + pl.toff);
    double *red = reinterpret_cast<double *>(smem + pl.red);
    int *misc = reinterpret_cast<int *>(smem + pl.misc);
    int *n_pendV = misc + 40, *n_pendH = misc + 41;
    const int ntiles = (n + W - 1) / W;
    int *toff2 = toff1 + ntiles + 1;
    const double *gx = a.b.x + off, *gy = a.b.y + off, *gz = a.b.z + off, *gv = a.b.v + off;
    const double cp = a.P.cos_pitch, sp = a.P.sin_pitch;
    const int32_t *rows1 = a.b.tri1 + 3 * t1b, *rows2 = a.b.tri2 + 3 * t2b;
    int bad = 0;
#ifdef MVOSR_STAMPS
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define MVOSR_TSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define MVOSR_TSTAMP(i) do {} while (0)
#endif

    // ---- the frame's tile index (validated: the walk below visits every row exactly once whatever it says)
    {
        const int64_t tb = a.b.tile_base[f];
        if ((int64_t)(a.b.tile_base[f + 1] - tb) != (int64_t)(ntiles + 1)) bad = 1;
        for (int k = tid; k <= ntiles; k += B) {
            const int o1 = bad ? 0 : a.b.tile1_off[tb + k], o2 = bad ? 0 : a.b.tile2_off[tb + k];
            toff1[k] = o1; toff2[k] = o2;
            if (k == 0 && (o1 != 0 || o2 != 0)) bad = 1;
            if (k == ntiles && (o1 > t1n || o2 > t2n)) bad = 1;
            if (k > 0 && !bad && (o1 < a.b.tile1_off[tb + k - 1] || o2 < a.b.tile2_off[tb + k - 1])) bad = 1;
        }
    }
    if (tid == 0) { n_pendV[0] = 0; n_pendH[0] = 0; }
    // a tile's vertices: coalesced loads of the caller's planes, remap (:391-392) fused
    double tv[VPT], tx[VPT], ty[VPT], tz[VPT];
    // (loads that are consumed a step later are issued UNCONDITIONALLY, on a clamped index: a load under a divergent
    // branch is waited for at the end of that branch, which would put the whole memory latency back on the step)
    auto tile_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int i = min(t * W + j * B + (tid & (W - 1)), n - 1);
            tv[j] = stream_load(gv + i); tx[j] = stream_load(gx + i); ty[j] = stream_load(gy + i); tz[j] = stream_load(gz + i);
        }
    };
    // ring slots.  Two tiles: feature i sits at i mod 2W.  Three: at i mod 3W — for a feature at distance d <= 2W - 1 from the
    // start of a tile whose slot begins at sb that is sb + d, wrapped once.
    auto tile_slot = [&](int t) { return kRing == 2 ? (t & 1) * W : (t % 3) * W; };
    auto slot_at = [&](int sb, int d) {
        if constexpr (kRing == 2) return (sb + d) & M;
        else { const int s_ = sb + d; return s_ >= 3 * W ? s_ - 3 * W : s_; }
    };
    auto tile_store = [&](int t) {
        const int sb = tile_slot(t);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            if (j * B + tid >= W) continue;
            const int i = t * W + j * B + tid, slot = sb + j * B + tid;
            double2 pa, pb;
            pa.x = tv[j]; pa.y = ty[j] * sp + tz[j] * cp;          // {v, z'}
            pb.x = tx[j]; pb.y = ty[j] * cp - tz[j] * sp;          // {x, y'}
            if (i < n) { ringA[slot] = pa; ringB[slot] = pb; }
#pragma unroll
            for (int c = 0; c < kSubC; ++c) ringC[slot * kSubC + c] = c == 0 ? 1 : 0;      // np.ones, :153 (the sub-counters add up)
            ringH[slot] = 0ull;
            used[slot] = 0;
        }
    };
    tile_load(0); tile_store(0);
    tile_load(1); tile_store(1);
    tile_load(2);       // (stored at the end of step 0: into tile 0's slot when that tile has retired, or into the third slot)
    // the block agrees on the index before anyone walks it
    {
        int b0 = bad, b1 = 0, b2 = 0, b3 = 0;
        block_sum4i<DW>(b0, b1, b2, b3, red + R_MISC * 2 * DW);
        bad = b0;
    }
    if (bad) {
        if (tid == 0) {
            a.o.raw_scale[f] = nan(""); a.o.height[f] = nan(""); a.o.height_level[f] = nan("");
            a.o.status[f] = MVOSR_ST_ERR_MASK; a.nsel[f] = 0;
            write_counts(a, f, 0, 0, 0, R);
        }
        return;
    }
    MVOSR_TSTAMP(0);

    double hsum = 0.0, hcnt = 0.0;
    bool sneg = false, spos = false, mixed_signs = false;      // steep heights below / above zero
    int npitch = 0, singular = 0, nvalid = 0, overflow = 0;
    const int rid = tid;       // (lane l of wavefront w taking row l*DW + w — neighbouring rows, which share vertices, then meet in
                               // different LDS atomic instructions — was measured 16 % slower: the row loads lose their coalescing)

    // ---- far rows first: the few rows (about 2 per thousand) whose vertices are not within two tiles of each other.
    // All three vertices come from the caller's planes (one batch of gathers per frame instead of a stall in every
    // step), and what the row contributes to them waits in the pending lists until their tiles are in the ring.
    // With the packer's far table (mvosr_batch.tile_far) the vertices are one contiguous block per frame, 72 bytes per
    // row; without it they are gathered from the planes (a 128-byte line per element).
    const int far1 = toff1[ntiles], far2 = toff2[ntiles];
    const double *fart = nullptr;
    if (a.b.tile_far && a.b.tile_far_off) {
        const int64_t fb = a.b.tile_far_off[f];
        if (a.b.tile_far_off[f + 1] - fb != 9 * ((int64_t)(t1n - far1) + (int64_t)(t2n - far2))) bad = 1;     // (uniform: all threads see it)
        else fart = a.b.tile_far + fb;
    }
    for (int t = far1 + tid; t < t1n; t += B) {
        const TriIds q = load_tri(rows1, t);
        if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) { bad = 1; continue; }
        double2 p0, p1, p2;
        if (fart) {
            const double *r = fart + 9 * (t - far1);                 // (y, z, v) x 3
            p0.x = r[2]; p0.y = r[0] * sp + r[1] * cp;
            p1.x = r[5]; p1.y = r[3] * sp + r[4] * cp;
            p2.x = r[8]; p2.y = r[6] * sp + r[7] * cp;
        } else {
            p0.x = gv[q.a]; p0.y = gy[q.a] * sp + gz[q.a] * cp;
            p1.x = gv[q.b]; p1.y = gy[q.b] * sp + gz[q.b] * cp;
            p2.x = gv[q.c]; p2.y = gy[q.c] * sp + gz[q.c] * cp;
        }
        const bool pa = (p0.x - p1.x) * (p0.y - p1.y) > 0.0;       // :107,:110
        const bool pb = (p0.x - p2.x) * (p0.y - p2.y) > 0.0;       // :108,:113  (marks vertices 0 and 1, as the reference does)
        const bool pc = (p1.x - p2.x) * (p1.y - p2.y) > 0.0;       // :109,:116
        const int e = atomicAdd(n_pendV, 3);
        if (e + 3 <= kPendCap) {
            const VoteFlags vf = vote_flags(pa, pb, pc, fixed);
            pendV[e] = q.a << 1 | (vf.f0 ? 1 : 0);
            pendV[e + 1] = q.b << 1 | (vf.f1 ? 1 : 0);
            pendV[e + 2] = q.c << 1 | (vf.f2 ? 1 : 0);
        } else overflow = 1;
    }
    for (int t = far2 + tid; t < t2n; t += B) {
        const TriIds q = load_tri(rows2, t);
        if ((unsigned)q.a >= (unsigned)n || (unsigned)q.b >= (unsigned)n || (unsigned)q.c >= (unsigned)n) { bad = 1; continue; }
        double xa, ya, za, xb, yb, zb, xc, yc, zc;
        if (fart) {
            const double *r = fart + 9 * ((t1n - far1) + (t - far2));    // (x, y, z) x 3
            xa = r[0]; ya = r[1]; za = r[2]; xb = r[3]; yb = r[4]; zb = r[5]; xc = r[6]; yc = r[7]; zc = r[8];
        } else {
            xa = gx[q.a]; ya = gy[q.a]; za = gz[q.a]; xb = gx[q.b]; yb = gy[q.b]; zb = gz[q.b]; xc = gx[q.c]; yc = gy[q.c]; zc = gz[q.c];
        }
        const double y0 = ya * cp - za * sp, y1 = yb * cp - zb * sp, y2 = yc * cp - zc * sp;
        const double h = (y0 + y1) + y2;                                                     // :238 — 3h, like every height of this kernel
        const int r = classify_triangle<false>(xa, y0, ya * sp + za * cp, xb, y1, yb * sp + zb * cp,
                                               xc, y2, yc * sp + zc * cp, h, a.pt, nullptr, nullptr, nullptr, 0);
        if (r & 4) singular = 1;
        if (r & 2) { hsum += h; hcnt += 1.0; sneg |= h < 0.0; spos |= h > 0.0; }             // :240
        if (r & 1) ++npitch;
        const unsigned long long key = ((r & 1) && h == h) ? height_key64(h) : 0ull;
        const int e = atomicAdd(n_pendH, 3);
        if (e + 3 <= kPendCapH) {
            pendH[e] = key; pendH[e + 1] = key; pendH[e + 2] = key;
            pendHv[e] = q.a; pendHv[e + 1] = q.b; pendHv[e + 2] = q.c;
        } else overflow = 1;
    }
    __syncthreads();
    // contributions of the far rows to the vertices of a tile (applied once the tile is in the ring)
    auto apply_pending = [&](int tile) {
        const int nv = min(n_pendV[0], kPendCap), nh = min(n_pendH[0], kPendCapH), sb = tile_slot(tile);
        for (int e = tid; e < nv; e += B) {
            const int p = pendV[e], vtx = p >> 1;
            if (vtx / W == tile) atomicAdd(&ringC[(sb + vtx - tile * W) * kSubC], (p & 1) ? -1 : 1);
        }
        for (int e = tid; e < nh; e += B) {
            const int vtx = pendHv[e];
            if (vtx / W == tile) { const unsigned long long key = pendH[e]; if (key) atomicMax(&ringH[sb + vtx - tile * W], key); used[sb + vtx - tile * W] = 1; }
        }
    };
    apply_pending(0);
    MVOSR_TSTAMP(1);

    const int32_t *rows1c = t1n > 0 ? rows1 : rows2;   // (an empty first triangulation: any readable row will do for the clamped loads)
    TriIds n1[kTileRows], n2[kTileRows];              // this thread's first rows of the coming step, in flight
    // (clamped to the tile's OWN last row: a slot beyond it re-reads that row's line instead of pulling in rows of the
    // next tile, which the next step would fetch a second time — tri2 has ~830 rows per tile against 1024 slots)
    auto prefetch_rows = [&](int k) {
        const int last1 = max(min(toff1[k + 1], max(t1n, 1)) - 1, 0), last2 = max(min(toff2[k + 1], t2n) - 1, 0);
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) {
            n1[j] = load_tri(rows1c, min(toff1[k] + j * B + rid, last1));
            n2[j] = load_tri(rows2, min(toff2[k] + j * B + rid, last2));
        }
    };
    prefetch_rows(0);
    // Candidates' largest flat height / y': two planes of n doubles in the workspace.  Wavefront w retires the 64-feature
    // groups g = w (mod DW) of every tile and APPENDS their candidates to its own dense list (capacity = the features it
    // retires, so the lists tile the planes exactly): the tail then reads candidates only, not group slots — the planes
    // were read in full before, 320 KB per 20000-feature frame for the third of the slots that held something.
    double *cand_h = reinterpret_cast<double *>(da.ws.P2 + off), *cand_y = cand_h + ((n + 1) & ~1);
    int cand_base = 0, cand_cnt = 0;                   // my wavefront's list: where it starts, how many entries (wave-uniform)
    {
        const int nfull = n >> 6, rem = n & 63;
        for (int ww = 0; ww < w; ++ww)
            cand_base += 64 * (nfull > ww ? (nfull - 1 - ww) / DW + 1 : 0) + ((rem && (nfull % DW) == ww) ? rem : 0);
    }

    // a tile leaves the ring: final vote of its vertices (:166), candidates parked
    auto retire = [&](int t) {
        const int sb = tile_slot(t);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            if (j * B + tid >= W) continue;
            const int i = t * W + j * B + tid, slot = sb + j * B + tid;
            bool survivor = false, is_cand = false;
            double cy = 0.0, ch = 0.0;
            if (i < n) {
                int csum = 0;
#pragma unroll
                for (int c = 0; c < kSubC; ++c) csum += ringC[slot * kSubC + c];
                survivor = csum >= 0;                                                  // :166
                if (used[slot] && !survivor) bad = 1;                                  // tri2 names a feature the vote dropped
                const unsigned long long hk = ringH[slot];
                is_cand = hk != 0ull;
                cy = ringB[slot].y; ch = height_of_key64(hk);                          // y', largest flat height
            }
            nvalid += __popcll(__ballot(survivor));
            const unsigned long long mc = __ballot(is_cand);
            if (is_cand) { const int at = cand_base + cand_cnt + __popcll(mc & ((1ull << lane) - 1ull)); cand_h[at] = ch; cand_y[at] = cy; }
            cand_cnt += __popcll(mc);
        }
    };
    // ---- the walk over the tiles
    for (int k = 0; k < ntiles; ++k) {
        const int lo = k * W;
        const int b1 = toff1[k], e1 = toff1[k + 1], b2 = toff2[k], e2 = toff2[k + 1];
        const int sbk = tile_slot(k);
        auto S = [&](int v) { return slot_at(sbk, v - lo); };            // (a checked row: 0 <= v - lo <= 2W - 1)
        TriIds c1[kTileRows], c2[kTileRows];
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) { c1[j] = n1[j]; c2[j] = n2[j]; }
        prefetch_rows(min(k + 1, ntiles - 1));
        apply_pending(k + 1);                          // (tile k+1 entered the ring at the end of the last step)
        // the vote over this tile's rows of tri1 (:151-167): vertices from the ring, 32-bit counters in the ring.
        // Every row of the walk has its vertices in tiles k and k+1 (checked: anything else is not the promised layout).
        auto row_ok = [&](const TriIds q) {
            const unsigned far = max(max((unsigned)(q.a - lo), (unsigned)(q.b - lo)), (unsigned)(q.c - lo));
            return far <= (unsigned)M && max(max(q.a, q.b), q.c) < n;
        };
        auto vote_row = [&](const TriIds q) {
            if (!row_ok(q)) { bad = 1; return; }
            const double2 p0 = ringA[S(q.a)], p1 = ringA[S(q.b)], p2 = ringA[S(q.c)];      // {v, z'}
            const bool pa = (p0.x - p1.x) * (p0.y - p1.y) > 0.0;       // :107,:110
            const bool pb = (p0.x - p2.x) * (p0.y - p2.y) > 0.0;       // :108,:113  (marks vertices 0 and 1, as the reference does)
            const bool pc = (p1.x - p2.x) * (p1.y - p2.y) > 0.0;       // :109,:116
            const VoteFlags vf = vote_flags(pa, pb, pc, fixed);
            atomicAdd(&ringC[(S(q.a)) * kSubC], vf.f0 ? -1 : 1);
            atomicAdd(&ringC[(S(q.b)) * kSubC], vf.f1 ? -1 : 1);
            atomicAdd(&ringC[(S(q.c)) * kSubC], vf.f2 ? -1 : 1);
        };
        {
            // the prefetched rows: every ring read first (the compiler cannot move the reads of one row across the LDS
            // atomics of another, and issued together their latencies overlap), then the tests and the atomics
            double2 vp[kTileRows][3];
            bool ok[kTileRows];
#pragma unroll
            for (int j = 0; j < kTileRows; ++j) {
                const TriIds q = c1[j];
                const bool live = b1 + j * B + rid < e1;
                ok[j] = live && row_ok(q);
                if (live && !ok[j]) bad = 1;
                if (ok[j]) { vp[j][0] = ringA[S(q.a)]; vp[j][1] = ringA[S(q.b)]; vp[j][2] = ringA[S(q.c)]; }      // {v, z'}
            }
#pragma unroll
            for (int j = 0; j < kTileRows; ++j) {
                if (!ok[j]) continue;
                const TriIds q = c1[j];
                const double2 p0 = vp[j][0], p1 = vp[j][1], p2 = vp[j][2];
                const bool pa = (p0.x - p1.x) * (p0.y - p1.y) > 0.0;       // :107,:110
                const bool pb = (p0.x - p2.x) * (p0.y - p2.y) > 0.0;       // :108,:113  (marks vertices 0 and 1, as the reference does)
                const bool pc = (p1.x - p2.x) * (p1.y - p2.y) > 0.0;       // :109,:116
                const VoteFlags vf = vote_flags(pa, pb, pc, fixed);
                atomicAdd(&ringC[(S(q.a)) * kSubC], vf.f0 ? -1 : 1);
                atomicAdd(&ringC[(S(q.b)) * kSubC], vf.f1 ? -1 : 1);
                atomicAdd(&ringC[(S(q.c)) * kSubC], vf.f2 ? -1 : 1);
            }
        }
        for (int t = b1 + kTileRows * B + rid; t < e1; t += B) vote_row(load_tri(rows1, t));
        MVOSR_TSTAMP(2);
        // the selection sweep over this tile's rows of tri2 (:225-240)
        auto select_row = [&](const TriIds q) {
            if (!row_ok(q)) { bad = 1; return; }
            const double2 a0 = ringA[S(q.a)], a1 = ringA[S(q.b)], a2 = ringA[S(q.c)];
            const double2 g0_ = ringB[S(q.a)], g1_ = ringB[S(q.b)], g2_ = ringB[S(q.c)];
            used[S(q.a)] = 1; used[S(q.b)] = 1; used[S(q.c)] = 1;        // (every writer stores the same value)
            // :238.  3h instead of h: the keys, their maxima and the level all scale by three, the comparisons of the tail are
            // the same outside the guard band (as in the LDS-resident product kernel), and no triangle pays the division
            const double h = (g0_.y + g1_.y) + g2_.y;
            const int r = classify_triangle<false>(g0_.x, g0_.y, a0.y, g1_.x, g1_.y, a1.y, g2_.x, g2_.y, a2.y, h, a.pt,
                                                   nullptr, nullptr, nullptr, 0);
            if (r & 4) singular = 1;
            if (r & 2) { hsum += h; hcnt += 1.0; sneg |= h < 0.0; spos |= h > 0.0; }             // :240
            if (r & 1) {
                ++npitch;
                if (h == h) {
                    const unsigned long long key = height_key64(h);
                    atomicMax(&ringH[S(q.a)], key); atomicMax(&ringH[S(q.b)], key); atomicMax(&ringH[S(q.c)], key);
                }
            }
        };
#pragma unroll
        for (int j = 0; j < kTileRows; ++j) { if (b2 + j * B + rid < e2) select_row(c2[j]); }
        for (int t = b2 + kTileRows * B + rid; t < e2; t += B) select_row(load_tri(rows2, t));
        MVOSR_TSTAMP(3);
        if constexpr (kRing == 2) {
            __syncthreads();                           // every row that names a vertex of tile k has been processed
            MVOSR_TSTAMP(4);
            retire(k);                                 // tile k+2 takes the slot
        } else if (k > 0) retire(k - 1);               // (final since the barrier that ended the last step; tile k+2 takes ITS slot)
        tile_store(k + 2);
        tile_load(k + 3);
        MVOSR_TSTAMP(5);
        __syncthreads();
        MVOSR_TSTAMP(6);
    }
    if constexpr (kRing == 3) retire(ntiles - 1);

    block_sum2<DW>(hsum, hcnt, red + R_SEL_H * 2 * DW);
    {
        // flags: one vote per wavefront (<= 16 of them: the three 8-bit fields cannot carry into each other)
        const int wf = (__ballot(singular != 0) ? 1 : 0) | (__ballot(bad != 0) ? 1 << 8 : 0) | (__ballot(overflow != 0) ? 1 << 16 : 0);
        int sb = (lane == 0) ? wf : 0;
        int nv = (lane == 0) ? nvalid : 0;              // (nvalid is wave-uniform: count it once per wave)
        // steep heights of both signs: the level's rounding error is then relative to sum |h| / count, not to |level| (they
        // may cancel), and the guard band below, which is relative to |level|, is not wide enough — the EXACT pass takes the frame
        int z0 = (lane == 0) ? ((__ballot(sneg) ? 1 : 0) | (__ballot(spos) ? 1 << 8 : 0)) : 0;
        block_sum4i<DW>(npitch, nv, sb, z0, red + R_SEL_CNT * 2 * DW);
        nvalid = nv; singular = sb & 0xFF; bad = (sb >> 8) & 0xFF; overflow = sb >> 16;
        mixed_signs = (z0 & 0xFF) && (z0 >> 8);
    }
    MVOSR_TSTAMP(7);
    const double hl = hsum / hcnt;                    // np.mean of an empty set -> 0/0 = NaN, like :240  (three times the level)
    const bool mask_mismatch = a.b.n2_expected && a.b.n2_expected[f] != nvalid;
    int status = kStPending;
    double raw = nan("");
    int nsel = 0;
    if (overflow && !bad) {
        // more far rows than the pending lists hold: not what the tiled layout promises — the two-sweep kernel takes the frame
        if (tid == 0) { a.redo[1 + atomicAdd(a.redo, 1)] = (int32_t)f; a.o.status[f] = kStRedo; a.nsel[f] = 0; }
        return;
    }
    if (mask_mismatch || bad) {
        status = MVOSR_ST_ERR_MASK;
    } else if (singular) {
        status = MVOSR_ST_ERR_SINGULAR;
    } else {
        // a vertex is selected when its largest flat height exceeds the level (:243-247): one comparison per candidate.
        // A wavefront owns a contiguous range of 64-feature groups.  The candidates' heights and y' were parked in two
        // planes, so the count reads one and the ordered store the other — every byte once — in batches of kTailBatch
        // loads in flight (one dependent load per group would put the memory latency on every group).
        constexpr int kTailBatch = 8;
        const int nchunks = (cand_cnt + 63) >> 6;      // 64 entries of my list per chunk
        const double *my_h = cand_h + cand_base, *my_y = cand_y + cand_base;
        unsigned long long selbits = 0ull;             // bit c: my entry of chunk c is selected (lists of up to 64 chunks)
        int cnt = 0, near = (hl == hl && !mixed_signs) ? 0 : 1;
        for (int cb = 0; cb < nchunks; cb += kTailBatch) {
            double hm[kTailBatch];
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) hm[j] = my_h[min(((cb + j) << 6) + lane, max(cand_cnt, 1) - 1)];
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                const bool have = ((cb + j) << 6) + lane < cand_cnt;
                // (|hl| IS sum |h| / count here — the form of the LDS-resident kernels' guard — because frames whose steep heights
                // have both signs never get this far: `near` starts at 1 for them, above; tests: fuzz kind 10)
                if (have && fabs(hm[j] - hl) <= kLevelGuard * fabs(hl)) near = 1;
                const bool sel = have && hm[j] > hl;                                            // :243-244
                if (sel && cb + j < 64) selbits |= 1ull << (cb + j);
                cnt += __popcll(__ballot(sel));
            }
        }
        const int wave_near = __ballot(near != 0) != 0ull;
        if (lane == 0) { misc[M_WCNT + w] = cnt; misc[M_WCNT + DW + w] = wave_near; }
        __syncthreads();
        int base = 0, any_near = 0;
#pragma unroll
        for (int i = 0; i < DW; ++i) { const int c = misc[M_WCNT + i]; if (i < w) base += c; nsel += c; any_near |= misc[M_WCNT + DW + i]; }
        MVOSR_TSTAMP(8);
        if (any_near || nsel == 0) {                  // the level's last bits could matter, or the level is the result: EXACT pass
            if (tid == 0) { a.redo[1 + atomicAdd(a.redo, 1)] = (int32_t)f; a.o.status[f] = kStRedo; a.nsel[f] = 0; }
            return;
        }
        double *dst = a.ysel + off;
        for (int cb = 0; cb < nchunks; cb += kTailBatch) {
            double yv[kTailBatch], hm[kTailBatch];
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                const int at = min(((cb + j) << 6) + lane, max(cand_cnt, 1) - 1);
                yv[j] = my_y[at];
                hm[j] = nchunks > 64 ? my_h[at] : 0.0;        // (more than 64 chunks in my list: the bits above do not reach)
            }
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                bool sel;
                if (nchunks > 64) sel = ((cb + j) << 6) + lane < cand_cnt && hm[j] > hl;
                else sel = cb + j < 64 && ((selbits >> (cb + j)) & 1ull) && ((cb + j) << 6) + lane < cand_cnt;
                const unsigned long long ms = __ballot(sel);
                if (sel) dst[base + __popcll(ms & ((1ull << lane) - 1ull))] = yv[j];
                base += __popcll(ms);
            }
        }
    }
    MVOSR_TSTAMP(10);
#ifdef MVOSR_STAMPS
    if (tid == 0 && a.o.hist) { unsigned long long *d = reinterpret_cast<unsigned long long *>(a.o.hist + f * 2 * kBins); for (int i = 0; i < 12; ++i) d[i] = tacc[i]; }
#endif
    if (tid == 0) {
        a.o.raw_scale[f] = raw;
        a.o.height[f] = nan("");
        a.o.height_level[f] = hl / 3.0;
        a.o.status[f] = status;
        a.nsel[f] = nsel;
        R.n_sel = nsel;
        write_counts(a, f, nvalid, npitch, -1, R);       // (the number of flat triangles above the level is not formed here)
        if (a.o.stats) { double *st = a.o.stats + 4 * f; st[0] = st[1] = st[2] = st[3] = nan(""); }
    }
}

template <int DW>
__global__ __launch_bounds__(DW *kWave) void outlier_vote_dense_kernel(const DenseArgs da) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KArgs &a = da.k;
    const int64_t f = a.first_frame + blockIdx.x;
    const int n = a.b.feat_cnt[f];
    if (n <= 0) { if (threadIdx.x == 0 && a.o.counts) a.o.counts[f * MVOSR_N_COUNTS + MVOSR_CNT_VALID] = 0; return; }
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f);
    const uint32_t npad = (uint32_t)((n + 1) & ~1);
    uint32_t *c32 = reinterpret_cast<uint32_t *>(smem);
    double *red = reinterpret_cast<double *>(smem + align16(2u * npad + 16u) + align16(4u * ((uint32_t)(n + 31) / 32u)));
    int *misc = reinterpret_cast<int *>(red + kRedSlots * 2 * DW);
    int bad = 0;
    const int nvalid = phase_vote_dense<DW, false>(c32, misc, n, nullptr, a.b.y + off, a.b.z + off, a.b.v + off, a.b.tri1, t1b, t1n,
                                               a.P.cos_pitch, a.P.sin_pitch, a.o.vote_counters + off, bad,
                                               nullptr, nullptr, a.P.vote_mode == MVOSR_VOTE_FIXED);
    int b0 = bad, b1 = 0, b2 = 0, b3 = 0;
    block_sum4i<DW>(b0, b1, b2, b3, red + R_MISC * 2 * DW);
    if (threadIdx.x == 0) {
        if (a.o.counts) a.o.counts[f * MVOSR_N_COUNTS + MVOSR_CNT_VALID] = nvalid;
        if (a.o.status) a.o.status[f] = b0 ? MVOSR_ST_ERR_MASK : MVOSR_ST_MODE;
    }
}

// K1 alone
template <int WAVES, int SC>
__global__ __launch_bounds__(WAVES *kWave) void outlier_vote_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = a.first_frame + blockIdx.x;
    const int n = a.b.feat_cnt[f];
    if (n <= 0) { if (threadIdx.x == 0 && a.o.counts) a.o.counts[f * MVOSR_N_COUNTS + MVOSR_CNT_VALID] = 0; return; }
    const int64_t off = a.b.feat_off[f];
    const int64_t t1b = a.b.tri1_off[f];
    const int t1n = tri_rows(a.b.tri1_off, a.b.tri1_cnt, f);
    const Smem s = carve(smem, n, WAVES);
    int bad = 0;
    MVOSR_STAMP_DECL
    TriChunk<WAVES * kWave> unused;
    const int nvalid = phase_vote<WAVES, SC>(s, n, nullptr, a.b.y + off, a.b.z + off, a.b.v + off, a.b.tri1, t1b, t1n,
                                             a.P.cos_pitch, a.P.sin_pitch, a.o.vote_counters + off, bad, nullptr, 0, 0, unused,
                                             a.P.vote_mode == MVOSR_VOTE_FIXED, 0 MVOSR_STAMP_PASS);
    int b0 = bad | (t1n > kMaxVoteRows ? 1 : 0), b1 = 0, b2 = 0, b3 = 0;
    block_sum4i<WAVES>(b0, b1, b2, b3, s.red + R_MISC * 2 * WAVES);
    if (threadIdx.x == 0) {
        if (a.o.counts) a.o.counts[f * MVOSR_N_COUNTS + MVOSR_CNT_VALID] = nvalid;
        if (a.o.status) a.o.status[f] = b0 ? MVOSR_ST_ERR_MASK : MVOSR_ST_MODE;
    }
}

// K4: sliding-window median of the raw scale sequence (scale_filtering, :396-400).  The sequence is
// either one contiguous array or, after the all-gather of the per-rank records, `n_blocks` blocks
// `stride` doubles apart holding the ranks' contiguous shares (the first `extra` blocks one frame more
// than `base_len`): element i is read in place, the gathered buffer is never repacked.
constexpr int kMaxWindow = 64;
struct MedianArgs {
    const double *raw; double *out; int64_t n; int window; int n_queue;
    int n_blocks; int64_t base_len, extra, stride;
    double queue[kMaxWindow];
};
__device__ __forceinline__ double median_seq_at(const MedianArgs &a, int64_t i) {
    if (a.n_blocks <= 1) return a.raw[i];
    const int64_t head = a.extra * (a.base_len + 1);
    int64_t r, j;
    if (i < head) { r = i / (a.base_len + 1); j = i - r * (a.base_len + 1); }
    else { const int64_t k = i - head; r = a.extra + k / a.base_len; j = k - (r - a.extra) * a.base_len; }
    return a.raw[r * a.stride + j];
}
__global__ __launch_bounds__(256) void window_median_kernel(const MedianArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    // position in the concatenated push sequence [queue..., raw...]
    const int64_t last = a.n_queue + i;
    int64_t first = last - a.window + 1;
    if (first < 0) first = 0;
    const int m = (int)(last - first + 1);
    double w[kMaxWindow];
    bool has_nan = false;
    for (int k = 0; k < m; ++k) {
        const int64_t p = first + k;
        const double x = (p < a.n_queue) ? a.queue[p] : median_seq_at(a, p - a.n_queue);
        has_nan |= (x != x);
        int j = k;                                   // insertion sort
        while (j > 0 && w[j - 1] > x) { w[j] = w[j - 1]; --j; }
        w[j] = x;
    }
    double r;
    if (has_nan) r = nan("");                        // np.median propagates NaN
    else if (m & 1) r = w[m >> 1];
    else r = (w[(m >> 1) - 1] + w[m >> 1]) / 2.0;     // np.mean of the two middle values
    a.out[i] = r;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static int g_max_dyn_lds = 160 * 1024;       // refined from the device in ctx_create

template <typename K>
static int prepare_kernel(K kernel, size_t lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_hip_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    return MVOSR_OK;
}

static PitchTest make_pitch_test(double thr_deg) {
    PitchTest pt;
    pt.thr_deg = thr_deg;
    if (thr_deg < -1.0 && thr_deg > -89.9) {
        const double sn = sin(-thr_deg * 3.141592653589793 / 180.0);
        pt.s2_hi = sn * sn * (1.0 + 1e-9);
        pt.s2_lo = sn * sn * (1.0 - 1e-9);
    } else {                                   // no safe band: always evaluate asin
        pt.s2_hi = INFINITY;
        pt.s2_lo = -INFINITY;
    }
    return pt;
}

// Ablation / A-B switches of profiling runs: ONLY in builds with -DMVOSR_ABLATE (profiles/ab_build.sh); the shipped library
// ignores the variable (tested).  Env MVOSR_DEBUG_SKIP, a bit mask — results with any bit of 1..16 set are NOT the path's
// results: 1 / 2 / 4 skip the vote sweep / first / second selection sweep, 8 forces the refused-frame tail, 16 skips the
// road-model launches, 32 ignores a batch's tile index (two-sweep dense kernel), 64 launches a ragged batch with one
// variant instead of per size class, 256 keeps the one-wavefront road model for dense batches.
static int debug_skip_env() {
#ifdef MVOSR_ABLATE
    static int v = -1;
    if (v < 0) { const char *e = getenv("MVOSR_DEBUG_SKIP"); v = e ? atoi(e) : 0; }
    return v;
#else
    return 0;
#endif
}

// MVOSR_LDS_PAD (same builds only): bytes added to the LDS request of scale_frames_kernel so that fewer workgroups fit a CU
// — the kernel lays its arrays out from the frame size, the padding is never touched (profiles/ab_occupancy.sh).
static size_t lds_pad_env() {
#ifdef MVOSR_ABLATE
    static long v = -1;
    if (v < 0) { const char *e = getenv("MVOSR_LDS_PAD"); v = e ? atol(e) : 0; }
    return (size_t)v;
#else
    return 0;
#endif
}

// Size classes of a ragged batch (the crossovers of pick_waves): the frames of a launch are split into up to three
// lists, and every list is launched with the variant and the LDS request of its own largest possible frame.
constexpr int kClassHeader = 4;                // cnt[3] | pad
constexpr int kWaves1Max = 320, kWaves4Max = 1152;      // pick_waves' crossovers
constexpr int kClassThr0 = kWaves1Max, kClassThr1 = 1024;   // (the 4-wavefront class stops where its SC = 4 instantiation does)
constexpr int64_t kClassMinFrames = 2048;      // below: one launch (three short grids would cost more than they save)
constexpr int32_t kClassHintMagic = 0x4d56;    // mvosr_batch.size_hint[3] when mvosr_batch_size_hint filled it
constexpr int kClassifyWaves = 16;
struct ClassArgs {
    const int32_t *feat_cnt;
    int64_t first_frame, n_frames;
    int32_t *hdr;                              // kClassHeader ints, zeroed before the launch
    int32_t *lists;                            // 3 x stride entries
    int64_t stride;
    int32_t *redo;                             // the (zeroed) redo list
    int grid[3];                               // workgroups of each class's launch
};

__host__ __device__ inline int size_class_of(int n) { return n <= kClassThr0 ? 0 : (n <= kClassThr1 ? 1 : 2); }

// One atomic per workgroup and class; the order inside a list is free.  A frame that does not fit its class's launch
// (a size hint that understates the class) goes to the redo list: the EXACT pass walks that list whatever its length.
__global__ __launch_bounds__(kClassifyWaves *kWave) void classify_frames_kernel(const ClassArgs c) {
    __shared__ int wcnt[kClassifyWaves][3];
    __shared__ int bbase[3];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    int cls = -1;
    int32_t f = 0;
    if (i < c.n_frames) {
        f = (int32_t)(c.first_frame + i);
        cls = size_class_of(c.feat_cnt[f]);
    }
    const uint64_t m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
    if (lane == 0) { wcnt[w][0] = __popcll(m0); wcnt[w][1] = __popcll(m1); wcnt[w][2] = __popcll(m2); }
    __syncthreads();
    if (threadIdx.x < 3) {
        int tot = 0;
        for (int ww = 0; ww < kClassifyWaves; ++ww) tot += wcnt[ww][threadIdx.x];
        bbase[threadIdx.x] = tot ? atomicAdd(c.hdr + threadIdx.x, tot) : 0;
    }
    __syncthreads();
    if (cls < 0) return;
    const uint64_t m = cls == 0 ? m0 : (cls == 1 ? m1 : m2);
    int pos = bbase[cls] + __popcll(m & ((1ull << lane) - 1ull));
    for (int ww = 0; ww < w; ++ww) pos += wcnt[ww][cls];
    if (pos < c.grid[cls]) c.lists[cls * c.stride + pos] = f;
    else c.redo[1 + atomicAdd(c.redo, 1)] = f;
}

// Variants: (wavefronts per frame, 64-feature sub-chunks each wave compacts).  Capacity of a variant
// is WAVES*SC*64 features; 16-bit vote counters and the 64-bit per-thread triangle flags are wider
// than any frame that fits LDS.
static int pick_waves(int requested, int max_feat) {
    if (requested == 1 || requested == 4 || requested == 8 || requested == 16) return requested;
    // measured crossovers (frames/s at 256 ... 4096 features per frame, re-measured after the round-2 kernel changes):
    // 1 wave up to 320 (256: 0.51 of the HBM peak against 0.44 with 4; 384: 0.46 against 0.50), 4 up to 1152 (1024: 0.57
    // against 0.51 with 8; 1280: 0.50 against 0.52), 8 as long as two workgroups fit a CU's LDS (about 3000 features:
    // 0.49 against 0.37 with 16), 16 above
    if (max_feat <= kWaves1Max) return 1;
    if (max_feat <= kWaves4Max) return 4;
    if (max_feat <= 2048 || 2 * (int64_t)lds_plan(max_feat, 8).total <= (int64_t)g_max_dyn_lds) return 8;
    return 16;
}
static int variant_capacity(int waves, int sc) { return waves * sc * kWave; }
// largest frame the LDS-resident variant takes (16 wavefronts): above it the dense variant runs
static int lds_capacity_features() {
    int lo = 1, hi = variant_capacity(16, 8);
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if ((int64_t)lds_plan(mid, 16).total <= (int64_t)g_max_dyn_lds) lo = mid; else hi = mid - 1;
    }
    return lo;
}

static int check_common(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b, const mvosr_outputs *o) {
    if (!ctx || !p || !b || !o) return set_error(MVOSR_ERR_ARG, "null argument");
    if (b->n_frames < 0 || !b->feat_off || !b->feat_cnt) return set_error(MVOSR_ERR_ARG, "batch without frames/offsets");
    if (b->max_feat < 0) return set_error(MVOSR_ERR_ARG, "max_feat < 0");
    return MVOSR_OK;
}

static int check_fit(const mvosr_batch *b, int waves, int sc, size_t lds) {
    if ((int64_t)lds > (int64_t)g_max_dyn_lds)
        return set_error(MVOSR_ERR_TOO_LARGE, "frame of %d features needs %zu B of LDS (> %d)", b->max_feat, lds, g_max_dyn_lds);
    if (b->max_feat > variant_capacity(waves, sc))
        return set_error(MVOSR_ERR_TOO_LARGE, "%d features exceed what %d wavefronts per frame handle (%d)", b->max_feat, waves,
                         variant_capacity(waves, sc));
    // a triangle sweep keeps one flag bit per iteration in a 64-bit register: T2 <= 64 * block
    if ((int64_t)2 * b->max_feat > (int64_t)64 * kWave * waves)
        return set_error(MVOSR_ERR_TOO_LARGE, "%d features give more triangles than %d wavefronts sweep", b->max_feat, waves);
    return MVOSR_OK;
}

// mvosr_batch.exact_mask: frames the caller wants finished in the exact mode (their height_level is read by a later step)
// are appended to the redo list behind the HOT kernel — unless that kernel put them there itself or refused them.  The HOT
// kernels are untouched: a masked frame is simply done twice, and a chunk has one or two of them.
__global__ __launch_bounds__(256) void append_mask_kernel(const uint8_t *mask, const int32_t *status, int64_t first, int64_t n, int32_t *redo) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t f = first + k;
    // (behind the HOT kernel: only frames it left for the road model — not the ones already on the list, refused, or in error)
    if (mask[f] && (!status || status[f] == kStPending)) redo[1 + atomicAdd(redo, 1)] = (int32_t)f;
}
static int launch_append_mask(mvosr_ctx *ctx, const KArgs &ka, int64_t nl, bool check_status) {
    if (!ka.b.exact_mask) return MVOSR_OK;
    hipLaunchKernelGGL(append_mask_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, ctx_stream(ctx), ka.b.exact_mask,
                       check_status ? ka.o.status : (const int32_t *)nullptr, ka.first_frame, nl, ka.redo);
    return check_launch("append_mask_kernel");
}

// One step of a scale-kernel family (its HOT / EXACT / FULL instantiations): FULL and EXACT run every frame of the
// range in that mode; HOT runs the product variant and then the EXACT variant over the redo list the HOT kernel
// filled (a short persistent grid: the list's length is only known on the device, and is almost always zero).
constexpr int kRedoGrid = 512;
constexpr int kModeExactList = 3;
static inline KArgs &kargs_of(KArgs &a) { return a; }
static inline KArgs &kargs_of(DenseArgs &a) { return a.k; }

// set by mvosr_scale_batch around its HOT dispatch: the exact pass over the redo list is left to the caller
static thread_local bool g_defer_exact = false;
static thread_local bool g_hot_only = false;      // MVOSR_WAVES_HOT_ONLY: no exact_mask frames on the redo list either

template <class Args>
static int launch_modes(mvosr_ctx *ctx, void (*k_hot)(const Args), void (*k_exact)(const Args), void (*k_full)(const Args),
                        Args args, int64_t nl, int threads, size_t lds, int mode, const char *name, size_t lds_hot = 0,
                        int threads_hot = 0) {
    int rc;
    if (lds_hot == 0) lds_hot = lds;
    if (threads_hot == 0) threads_hot = threads;
    kargs_of(args).redo_pass = 0;
    if (mode == MODE_FULL) {
        if ((rc = prepare_kernel(k_full, lds))) return rc;
        hipLaunchKernelGGL(k_full, dim3((unsigned)nl), dim3(threads), lds, ctx_stream(ctx), args);
        return check_launch(name);
    }
    if ((rc = prepare_kernel(k_exact, lds))) return rc;
    if (mode == kModeExactList) {                 // the EXACT variant over the list in args.redo (filled by the road-model kernel)
        kargs_of(args).redo_pass = 1;
        const unsigned grid = (unsigned)(nl < (int64_t)kRedoGrid ? nl : (int64_t)kRedoGrid);
        hipLaunchKernelGGL(k_exact, dim3(grid), dim3(threads), lds, ctx_stream(ctx), args);
        return check_launch(name);
    }
    if (mode == MODE_EXACT) {
        hipLaunchKernelGGL(k_exact, dim3((unsigned)nl), dim3(threads), lds, ctx_stream(ctx), args);
        return check_launch(name);
    }
    if ((rc = prepare_kernel(k_hot, lds_hot))) return rc;
    const hipError_t e = hipMemsetAsync(kargs_of(args).redo, 0, sizeof(int32_t), ctx_stream(ctx));
    if (e != hipSuccess) return set_hip_error("hipMemsetAsync(redo list)", e);
    hipLaunchKernelGGL(k_hot, dim3((unsigned)nl), dim3(threads_hot), lds_hot, ctx_stream(ctx), args);
    if ((rc = check_launch(name))) return rc;
    if (!g_hot_only && (rc = launch_append_mask(ctx, kargs_of(args), nl, true))) return rc;
    if (g_defer_exact) return MVOSR_OK;           // mvosr_scale_batch runs ONE exact pass, after the road model has added its frames to the list
    kargs_of(args).redo_pass = 1;
    const unsigned grid = (unsigned)(nl < (int64_t)kRedoGrid ? nl : (int64_t)kRedoGrid);
    hipLaunchKernelGGL(k_exact, dim3(grid), dim3(threads), lds, ctx_stream(ctx), args);
    return check_launch(name);
}

template <int WAVES, int SC>
static int launch_scale(mvosr_ctx *ctx, const KArgs &ka, int64_t nl, int mode) {
    const size_t lds = lds_plan(ka.b.max_feat, WAVES).total + lds_pad_env();
    int rc = check_fit(&ka.b, WAVES, SC, lds);
    if (rc) return rc;
    return launch_modes<KArgs>(ctx, scale_frames_kernel<WAVES, SC, MODE_HOT>, scale_frames_kernel<WAVES, SC, MODE_EXACT>,
                               scale_frames_kernel<WAVES, SC, MODE_FULL>, ka, nl, WAVES * kWave, lds, mode, "scale_frames_kernel");
}

template <int WAVES, int SC>
static int launch_vote(mvosr_ctx *ctx, const KArgs &ka, int64_t nl) {
    const size_t lds = lds_plan(ka.b.max_feat, WAVES).total;
    int rc = check_fit(&ka.b, WAVES, SC, lds);
    if (rc) return rc;
    if ((rc = prepare_kernel(outlier_vote_kernel<WAVES, SC>, lds))) return rc;
    hipLaunchKernelGGL((outlier_vote_kernel<WAVES, SC>), dim3((unsigned)nl), dim3(WAVES * kWave), lds, ctx_stream(ctx), ka);
    return check_launch("outlier_vote_kernel");
}

// dense frames: survivors' planes in a global workspace (24 B per feature), 16 wavefronts per frame.  (8 wavefronts
// per frame, three workgroups per CU, measured the same: the sweeps are bound by L2 gather traffic, not occupancy.)
constexpr int kDenseWaves = 16;

template <int DW>
static int launch_scale_dense_w(mvosr_ctx *ctx, const KArgs &ka, int64_t nl, int mode, bool vote_only) {
    if ((int64_t)2 * ka.b.max_feat > (int64_t)128 * kWave * DW)
        return set_error(MVOSR_ERR_TOO_LARGE, "%d features give more triangles than %d wavefronts sweep", ka.b.max_feat, DW);
    const size_t lds = dense_lds_bytes(ka.b.max_feat, DW);
    if ((int64_t)lds > (int64_t)g_max_dyn_lds)
        return set_error(MVOSR_ERR_TOO_LARGE, "frame of %d features needs %zu B of LDS (> %d)", ka.b.max_feat, lds, g_max_dyn_lds);
    DenseArgs da;
    da.k = ka;
    da.ws.P2 = nullptr; da.ws.Y2 = nullptr;
    int rc = MVOSR_OK;
    // the tiled variant: feature-numbered rows with the host's tile index, frames small enough for its LDS plan
    const bool tiled = !vote_only && ka.b.tri2_ids == MVOSR_TRI2_FEATURES && ka.b.tile_w == kTileW && ka.b.tile_base &&
                       ka.b.tile1_off && ka.b.tile2_off && !(debug_skip_env() & 32) &&
                       (int64_t)tiled_plan(ka.b.max_feat, kTiledWaves).total <= (int64_t)g_max_dyn_lds;
    if (!vote_only && (ka.b.tri2_ids != MVOSR_TRI2_FEATURES || tiled)) {    // (the vote alone and the two-sweep feature-numbered variant need no workspace)
        void *p[2];
        if ((rc = ctx_workspace_dense(ctx, ka.b.total_feat, p))) return rc;
        da.ws.P2 = reinterpret_cast<double2 *>(p[0]); da.ws.Y2 = reinterpret_cast<double *>(p[1]);
    }
    if (vote_only) {
        if ((rc = prepare_kernel(outlier_vote_dense_kernel<DW>, lds))) return rc;
        hipLaunchKernelGGL((outlier_vote_dense_kernel<DW>), dim3((unsigned)nl), dim3(DW * kWave), lds, ctx_stream(ctx), da);
        return check_launch("outlier_vote_dense_kernel");
    }
    if (tiled)
        return launch_modes<DenseArgs>(ctx, scale_frames_tiled_kernel<kTiledWaves>, scale_frames_dense_feat_kernel<DW, MODE_EXACT>,
                                       scale_frames_dense_feat_kernel<DW, MODE_FULL>, da, nl, DW * kWave, lds, mode,
                                       "scale_frames_tiled_kernel", tiled_plan(ka.b.max_feat, kTiledWaves).total, kTiledWaves * kWave);
    if (ka.b.tri2_ids == MVOSR_TRI2_FEATURES)
        return launch_modes<DenseArgs>(ctx, scale_frames_dense_feat_kernel<DW, MODE_HOT>, scale_frames_dense_feat_kernel<DW, MODE_EXACT>,
                                       scale_frames_dense_feat_kernel<DW, MODE_FULL>, da, nl, DW * kWave, lds, mode,
                                       "scale_frames_dense_feat_kernel");
    return launch_modes<DenseArgs>(ctx, scale_frames_dense_kernel<DW, MODE_HOT>, scale_frames_dense_kernel<DW, MODE_EXACT>,
                                   scale_frames_dense_kernel<DW, MODE_FULL>, da, nl, DW * kWave, lds, mode, "scale_frames_dense_kernel");
}

static int launch_scale_dense(mvosr_ctx *ctx, const KArgs &ka, int64_t nl, int mode, bool vote_only) {
    if (ka.b.max_feat > 65535) return set_error(MVOSR_ERR_TOO_LARGE, "more than 65535 features per frame");
    return launch_scale_dense_w<kDenseWaves>(ctx, ka, nl, mode, vote_only);
}

// one wavefront per frame, kRoadWaves frames per workgroup
static int launch_road(mvosr_ctx *ctx, const RoadArgs &ra, hipStream_t stream) {
    if (ra.n_frames <= 0) return MVOSR_OK;
    if (ra.wide && !ra.list) {
        hipLaunchKernelGGL((road_model_kernel<false, kRoadWaves>), dim3((unsigned)ra.n_frames), dim3(kRoadWaves * kWave), 0, stream, ra);
        return check_launch("road_model_kernel (wide)");
    }
    const unsigned blocks = ra.list ? 64u : (unsigned)((ra.n_frames + kRoadWaves - 1) / kRoadWaves);
    if (ra.list) hipLaunchKernelGGL(road_model_kernel<true>, dim3(blocks), dim3(kRoadWaves * kWave), 0, stream, ra);
    else hipLaunchKernelGGL(road_model_kernel<false>, dim3(blocks), dim3(kRoadWaves * kWave), 0, stream, ra);
    return check_launch("road_model_kernel");
}

// dispatch on (waves, sub-chunks): the smaller SC keeps fewer registers live across the compaction
#define MVOSR_DISPATCH(FN, ...)                                                             \
    do {                                                                                    \
        const int n_ = ka.b.max_feat;                                                       \
        switch (waves) {                                                                    \
            case 1: return FN<1, 8>(__VA_ARGS__);                                           \
            case 4: return (n_ <= variant_capacity(4, 4)) ? FN<4, 4>(__VA_ARGS__) : FN<4, 8>(__VA_ARGS__);     \
            case 16: return (n_ <= variant_capacity(16, 4)) ? FN<16, 4>(__VA_ARGS__) : FN<16, 8>(__VA_ARGS__); \
            default: return (n_ <= variant_capacity(8, 4)) ? FN<8, 4>(__VA_ARGS__) : FN<8, 8>(__VA_ARGS__);    \
        }                                                                                   \
    } while (0)

static int dispatch_scale(mvosr_ctx *ctx, const KArgs &ka, int waves, int64_t nl, int mode) { MVOSR_DISPATCH(launch_scale, ctx, ka, nl, mode); }
static int dispatch_vote(mvosr_ctx *ctx, const KArgs &ka, int waves, int64_t nl) { MVOSR_DISPATCH(launch_vote, ctx, ka, nl); }

// the HOT list variant of (waves, sub-chunks)
typedef void (*scale_kernel_fn)(const KArgs);
static scale_kernel_fn hot_list_kernel(int waves, int n) {
    switch (waves) {
        case 1: return scale_frames_kernel<1, 8, MODE_HOT, true>;
        case 4: return (n <= variant_capacity(4, 4)) ? scale_frames_kernel<4, 4, MODE_HOT, true> : scale_frames_kernel<4, 8, MODE_HOT, true>;
        case 16: return (n <= variant_capacity(16, 4)) ? scale_frames_kernel<16, 4, MODE_HOT, true> : scale_frames_kernel<16, 8, MODE_HOT, true>;
        default: return (n <= variant_capacity(8, 4)) ? scale_frames_kernel<8, 4, MODE_HOT, true> : scale_frames_kernel<8, 8, MODE_HOT, true>;
    }
}

// HOT pass of a ragged batch: classify, then one launch per size class over the class's list.  The frames every class leaves for the EXACT pass land on the one redo
// list; the caller runs that pass (kModeExactList) with the batch-wide variant.
static int launch_scale_classes(mvosr_ctx *ctx, const KArgs &ka, int64_t nl) {
    int rc;
    hipError_t e;
    int32_t *hdr = ka.nsel + 3 * ka.b.n_frames + 4;
    ClassArgs ca;
    ca.feat_cnt = ka.b.feat_cnt; ca.first_frame = ka.first_frame; ca.n_frames = nl;
    ca.hdr = hdr; ca.lists = hdr + kClassHeader; ca.stride = ka.b.n_frames; ca.redo = ka.redo;
    const bool hinted = ka.b.size_hint[3] == kClassHintMagic;
    const int thr[3] = {kClassThr0, kClassThr1, ka.b.max_feat};
    scale_kernel_fn fn[3];
    int waves[3];
    size_t lds[3];
    KArgs kc[3];
    for (int c = 0; c < 3; ++c) {
        kc[c] = ka;
        kc[c].b.max_feat = thr[c] < ka.b.max_feat ? thr[c] : ka.b.max_feat;
        const int n = kc[c].b.max_feat;
        waves[c] = pick_waves(0, n);
        lds[c] = lds_plan(n, waves[c]).total;
        fn[c] = hot_list_kernel(waves[c], n);
        const int sc = (n <= variant_capacity(waves[c], 4) && waves[c] != 1) ? 4 : 8;
        if ((rc = check_fit(&kc[c].b, waves[c], sc, lds[c]))) return rc;
        if ((rc = prepare_kernel(fn[c], lds[c]))) return rc;
        int64_t bound = hinted && (int64_t)ka.b.size_hint[c] < nl ? (int64_t)ka.b.size_hint[c] : nl;
        // a class the batch header rules out (no frame that large / that small) is not launched: its grid is 0, so a
        // frame that lands in it after all — a header that understates the batch — goes to the redo list like any overflow
        if ((c > 0 && thr[c - 1] >= ka.b.max_feat) || ka.b.min_feat > thr[c]) bound = 0;
        ca.grid[c] = (int)(bound < 0 ? 0 : bound);
        kc[c].cls_list = ca.lists + c * ca.stride;
        kc[c].cls_cnt = hdr + c;
        kc[c].redo_pass = 0;
    }
    if ((e = hipMemsetAsync(hdr, 0, kClassHeader * sizeof(int32_t), ctx_stream(ctx))) != hipSuccess)
        return set_hip_error("hipMemsetAsync(class lists)", e);
    if ((e = hipMemsetAsync(ka.redo, 0, sizeof(int32_t), ctx_stream(ctx))) != hipSuccess)
        return set_hip_error("hipMemsetAsync(redo list)", e);
    const int cb = kClassifyWaves * kWave;
    hipLaunchKernelGGL(classify_frames_kernel, dim3((unsigned)((nl + cb - 1) / cb)), dim3(cb), 0, ctx_stream(ctx), ca);
    if ((rc = check_launch("classify_frames_kernel"))) return rc;
    for (int c = 2; c >= 0; --c) {              // largest frames first
        if (ca.grid[c] == 0) continue;
        hipLaunchKernelGGL(fn[c], dim3((unsigned)ca.grid[c]), dim3(waves[c] * kWave), lds[c], ctx_stream(ctx), kc[c]);
        if ((rc = check_launch("scale_frames_kernel (size class)"))) return rc;
    }
    return MVOSR_OK;
}

void set_max_dynamic_lds(int bytes) { g_max_dyn_lds = bytes; }

}  // namespace mvosr

using namespace mvosr;

extern "C" {

void mvosr_default_params(mvosr_params *p, double absolute_reference) {
    if (!p) return;
    const double pitch = -0.5 * 3.141592653589793 / 180.0;   // scale_calculator.py:24 (the Python shim overwrites cos/sin with NumPy's)
    p->cos_pitch = cos(pitch);
    p->sin_pitch = sin(pitch);
    p->absolute_reference = absolute_reference;
    p->pitch_threshold_deg = -80.0;
    p->skew_threshold = 0.3;
    p->mode_rel = 0.33;
    p->mode_min = 2;
    p->vote_mode = MVOSR_VOTE_REFERENCE;
}

size_t mvosr_lds_bytes(int n_features) {
    const int n = n_features < 1 ? 1 : n_features;
    return lds_plan(n, pick_waves(0, n)).total;
}

int mvosr_max_lds_features(void) { return lds_capacity_features(); }

int mvosr_scale_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b, const mvosr_outputs *o,
                      int waves_per_frame, int64_t first_frame, int64_t n_launch) {
    int rc = check_common(ctx, p, b, o);
    if (rc) return rc;
    const bool want_exact = (waves_per_frame & MVOSR_WAVES_EXACT) != 0;
    const bool want_masked = (waves_per_frame & MVOSR_WAVES_EXACT_MASKED) != 0;
    const bool hot_only = (waves_per_frame & MVOSR_WAVES_HOT_ONLY) != 0;
    waves_per_frame &= ~(MVOSR_WAVES_EXACT | MVOSR_WAVES_EXACT_MASKED | MVOSR_WAVES_HOT_ONLY);
    if (hot_only && (want_exact || want_masked)) return set_error(MVOSR_ERR_ARG, "scale_batch: MVOSR_WAVES_HOT_ONLY excludes the exact flags");
    if (hot_only && (o->tri_normals || o->tri_pitch_deg || o->tri_heights)) return set_error(MVOSR_ERR_ARG, "scale_batch: MVOSR_WAVES_HOT_ONLY has no per-triangle outputs");
    if (!b->x || !b->y || !b->z || !b->v || !b->tri1_off || !b->tri2_off || !b->tri2)
        return set_error(MVOSR_ERR_ARG, "scale_batch: missing input plane / triangulation");
    if (!o->raw_scale || !o->height || !o->height_level || !o->status)
        return set_error(MVOSR_ERR_ARG, "scale_batch: raw_scale/height/height_level/status are required outputs");
    if (n_launch <= 0) { first_frame = 0; n_launch = b->n_frames; }
    if (first_frame < 0 || first_frame + n_launch > b->n_frames) return set_error(MVOSR_ERR_ARG, "frame range outside the batch");
    if (n_launch == 0) return MVOSR_OK;
    if ((rc = ctx_activate(ctx))) return rc;
    KArgs ka;
    ka.P = *p; ka.b = *b; ka.o = *o; ka.pt = make_pitch_test(p->pitch_threshold_deg);
    ka.first_frame = first_frame; ka.height_level_in = nullptr; ka.debug_skip = debug_skip_env();
    ka.redo_pass = 0; ka.cls_list = nullptr; ka.cls_cnt = nullptr;
    if (b->total_feat <= 0) return set_error(MVOSR_ERR_ARG, "scale_batch: batch.total_feat (length of the feature planes) not set");
    if (b->n_frames >= ((int64_t)1 << 31) - 1) return set_error(MVOSR_ERR_TOO_LARGE, "scale_batch: more than 2^31-2 frames in one batch");
    if ((rc = ctx_workspace(ctx, b->n_frames, b->total_feat, &ka.ysel, &ka.nsel))) return rc;
    ka.redo = ka.nsel + b->n_frames;            // [1 + n_frames] ints behind the nsel array
    int32_t *redo2 = ka.redo + b->n_frames + 1; // a second list: frames the road model ends on the fallback level
    // FULL: per-triangle debug outputs; EXACT: stage outputs requested (height_level bit-equal to NumPy's for every
    // frame); HOT: the product path + its exact pass over the frames that need it
    const int mode = hot_only ? MODE_HOT : (o->tri_normals || o->tri_pitch_deg || o->tri_heights) ? MODE_FULL
                     : ((o->selected || o->vote_counters || want_exact) ? MODE_EXACT : MODE_HOT);
    const int waves = pick_waves(waves_per_frame, b->max_feat);
    RoadArgs ra;
    ra.P = *p; ra.off = b->feat_off; ra.cnt = ka.nsel; ra.y = ka.ysel; ra.scratch = ka.ysel;
    ra.height_level = o->height_level; ra.o = *o; ra.pending_only = 1; ra.level_redo = nullptr; ra.list = nullptr; ra.wide = 0;
    // The step is two launches on the context's stream: the scale kernel, then the road model (one
    // wavefront per frame) on the dense lists it left in the workspace.  (Splitting the batch into
    // chunks to run the road model of one chunk under the scale kernel of the next was measured
    // and lost 10%: smaller grids pay more tail than the overlap returns.)
    ka.first_frame = first_frame;
    if (b->tri2_ids != MVOSR_TRI2_SURVIVORS && b->tri2_ids != MVOSR_TRI2_FEATURES)
        return set_error(MVOSR_ERR_ARG, "scale_batch: batch.tri2_ids must be MVOSR_TRI2_SURVIVORS or MVOSR_TRI2_FEATURES");
    if (b->tri2_ids == MVOSR_TRI2_FEATURES && !(waves_per_frame == 0 || waves_per_frame == 16))
        return set_error(MVOSR_ERR_ARG, "scale_batch: feature-numbered tri2 runs the gather variant (waves_per_frame 0 or 16)");
    const bool dense = (waves_per_frame == 0 || waves_per_frame == 16) &&
                       (b->max_feat > lds_capacity_features() || b->tri2_ids == MVOSR_TRI2_FEATURES);
    hipEvent_t *pev = (ctx->prof_on && ctx->prof_calls < kProfRing) ? ctx->prof_ev[ctx->prof_calls] : nullptr;
    hipError_t ee = hipSuccess;
    if (pev && (ee = hipEventRecord(pev[0], ctx_stream(ctx))) != hipSuccess) return set_hip_error("hipEventRecord(profile)", ee);
    // ragged batch: the HOT pass per size class (each class with its own variant and LDS request), then the EXACT pass
    // over the redo list as usual
    const bool by_class = mode == MODE_HOT && !dense && waves_per_frame == 0 && n_launch >= kClassMinFrames &&
                          b->max_feat > kClassThr0 && !(debug_skip_env() & 64) &&
                          (b->min_feat <= 0 || pick_waves(0, b->min_feat) != waves);
    // stand-in rows (mvosr_batch.standin_*): the frames of the exact pass's list get SciPy's own rows before the pass reads them
    const bool standin = b->standin_u != nullptr;
    if (standin) {
        if (!b->standin_keep || !b->standin_rows || !b->standin_cnt || !b->standin_status || !b->tri2_cnt || b->tri2_ids != MVOSR_TRI2_SURVIVORS)
            return set_error(MVOSR_ERR_ARG, "scale_batch: stand-in rows need standin_keep/_rows/_cnt/_status, tri2_cnt and survivor-numbered rows");
        if (dense) return set_error(MVOSR_ERR_TOO_LARGE, "scale_batch: stand-in rows are for frames that fit the LDS-resident kernels");
        if (!(want_masked || hot_only || (mode == MODE_HOT && !(debug_skip_env() & (16 | 512)))))
            return set_error(MVOSR_ERR_ARG, "scale_batch: stand-in rows need the HOT mode (no stage outputs, no EXACT-for-all)");
    }
    auto standin_rows = [&](const int32_t *list) {
        return qh_rows_for_list(ctx, b->n_frames, b->feat_off, b->feat_cnt, b->standin_u, b->v, b->standin_keep, b->max_feat, b->tri2_off,
                                b->standin_rows, b->standin_cnt, b->standin_status, list);
    };
    if (want_masked) {
        // the EXACT variant over the frames of the range whose exact_mask byte is set, the road model over the same list;
        // every other frame's outputs stay as they are
        if (!b->exact_mask) return set_error(MVOSR_ERR_ARG, "scale_batch: MVOSR_WAVES_EXACT_MASKED without batch.exact_mask");
        const hipError_t e0 = hipMemsetAsync(ka.redo, 0, sizeof(int32_t), ctx_stream(ctx));
        if (e0 != hipSuccess) return set_hip_error("hipMemsetAsync(redo list)", e0);
        if ((rc = launch_append_mask(ctx, ka, n_launch, false))) return rc;
        if (standin && (rc = standin_rows(ka.redo))) return rc;
        if ((rc = dense ? launch_scale_dense(ctx, ka, n_launch, kModeExactList, false) : dispatch_scale(ctx, ka, waves, n_launch, kModeExactList))) return rc;
        ra.first_frame = first_frame; ra.n_frames = n_launch; ra.list = ka.redo;
        ra.wide = (dense && !(debug_skip_env() & 256)) ? 1 : 0;
        return launch_road(ctx, ra, ctx_stream(ctx));
    }
    // HOT: ONE exact pass per call (round 5; two before: one behind the HOT kernel, one behind the road model).  The HOT kernel
    // and the exact mask put their frames on the redo list; the road model runs over every other frame and APPENDS the frames that
    // end on the fallback level (:334-335; rare) to the same list; then the EXACT variant over the list, then the road model over it.
    const bool fold = mode == MODE_HOT && (hot_only || (!(debug_skip_env() & 16) && !(debug_skip_env() & 512)));
    if (by_class) {
        if ((rc = launch_scale_classes(ctx, ka, n_launch))) return rc;
        if (!hot_only && (rc = launch_append_mask(ctx, ka, n_launch, true))) return rc;
        if (!fold && (rc = dispatch_scale(ctx, ka, waves, n_launch, kModeExactList))) return rc;
    } else {
        g_defer_exact = fold;
        g_hot_only = hot_only;
        rc = dense ? launch_scale_dense(ctx, ka, n_launch, mode, false) : dispatch_scale(ctx, ka, waves, n_launch, mode);
        g_defer_exact = false;
        g_hot_only = false;
        if (rc) return rc;
    }
    if (pev && (ee = hipEventRecord(pev[1], ctx_stream(ctx))) != hipSuccess) return set_hip_error("hipEventRecord(profile)", ee);
    ra.first_frame = first_frame; ra.n_frames = n_launch;
    ra.wide = (dense && !(debug_skip_env() & 256)) ? 1 : 0;     // dense batches: thousands of values per list, few frames
    if (!(debug_skip_env() & 16)) {
        if (fold) {
            ra.level_redo = ka.redo;
            ra.listed_mask = hot_only ? nullptr : b->exact_mask;    // (hot_only: append_mask_kernel did not run)
            if ((rc = launch_road(ctx, ra, ctx_stream(ctx)))) return rc;
            ra.listed_mask = nullptr;
            if (hot_only) return MVOSR_OK;            // (the list's frames stay MVOSR_ST_REDO: the caller's)
            if (standin && (rc = standin_rows(ka.redo))) return rc;          // SciPy's own rows for the frames the exact pass redoes
            if ((rc = dense ? launch_scale_dense(ctx, ka, n_launch, kModeExactList, false) : dispatch_scale(ctx, ka, waves, n_launch, kModeExactList))) return rc;
            ra.level_redo = nullptr; ra.list = ka.redo;
            if ((rc = launch_road(ctx, ra, ctx_stream(ctx)))) return rc;
        } else {
            if (mode == MODE_HOT) {
                // frames whose road model ends on the fallback level come back on a second list: the EXACT variant redoes them
                // (height_level in NumPy's order), then the road model runs on that list alone
                const hipError_t e2 = hipMemsetAsync(redo2, 0, sizeof(int32_t), ctx_stream(ctx));
                if (e2 != hipSuccess) return set_hip_error("hipMemsetAsync(redo list 2)", e2);
                ra.level_redo = redo2;
            }
            if ((rc = launch_road(ctx, ra, ctx_stream(ctx)))) return rc;
            if (mode == MODE_HOT) {
                KArgs k2 = ka;
                k2.redo = redo2;
                if ((rc = dense ? launch_scale_dense(ctx, k2, n_launch, kModeExactList, false) : dispatch_scale(ctx, k2, waves, n_launch, kModeExactList))) return rc;
                ra.level_redo = nullptr; ra.list = redo2;
                if ((rc = launch_road(ctx, ra, ctx_stream(ctx)))) return rc;
            }
        }
    }
    if (pev) {
        if ((ee = hipEventRecord(pev[2], ctx_stream(ctx))) != hipSuccess) return set_hip_error("hipEventRecord(profile)", ee);
        ctx->prof_calls++;
    }
    return MVOSR_OK;
}

int mvosr_batch_size_hint(const int32_t *feat_cnt_host, int64_t n_frames, mvosr_batch *b) {
    if (!b || n_frames < 0 || (n_frames > 0 && !feat_cnt_host)) return set_error(MVOSR_ERR_ARG, "batch_size_hint: null argument");
    int64_t cnt[3] = {0, 0, 0};
    int32_t lo = 0, hi = 0;
    for (int64_t f = 0; f < n_frames; ++f) {
        const int32_t n = feat_cnt_host[f];
        if (n < 0) return set_error(MVOSR_ERR_ARG, "batch_size_hint: negative feature count");
        if (f == 0 || n < lo) lo = n;
        if (f == 0 || n > hi) hi = n;
        ++cnt[size_class_of(n)];
    }
    b->max_feat = hi;
    b->min_feat = lo;            // (0 reads as "not stated": a batch with an empty frame is treated as ragged)
    for (int c = 0; c < 3; ++c) b->size_hint[c] = (int32_t)(cnt[c] > 0x7fffffff ? 0x7fffffff : cnt[c]);
    b->size_hint[3] = kClassHintMagic;
    return MVOSR_OK;
}

int mvosr_outlier_vote_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b, const mvosr_outputs *o,
                             int waves_per_frame) {
    int rc = check_common(ctx, p, b, o);
    if (rc) return rc;
    if (!b->y || !b->z || !b->v || !b->tri1_off || !b->tri1) return set_error(MVOSR_ERR_ARG, "outlier_vote: missing y/z/v/tri1");
    if (!o->vote_counters) return set_error(MVOSR_ERR_ARG, "outlier_vote: vote_counters output required");
    if (b->n_frames == 0) return MVOSR_OK;
    if ((rc = ctx_activate(ctx))) return rc;
    KArgs ka;
    ka.P = *p; ka.b = *b; ka.o = *o; ka.pt = make_pitch_test(p->pitch_threshold_deg);
    ka.first_frame = 0; ka.height_level_in = nullptr; ka.debug_skip = 0; ka.ysel = nullptr; ka.nsel = nullptr; ka.redo = nullptr; ka.redo_pass = 0;
    ka.cls_list = nullptr; ka.cls_cnt = nullptr;
    if ((waves_per_frame == 0 || waves_per_frame == 16) && b->max_feat > lds_capacity_features()) {
        if (b->total_feat <= 0) return set_error(MVOSR_ERR_ARG, "outlier_vote: batch.total_feat not set");
        return launch_scale_dense(ctx, ka, b->n_frames, MODE_HOT, true);
    }
    return dispatch_vote(ctx, ka, pick_waves(waves_per_frame, b->max_feat), b->n_frames);
}

int mvosr_road_model_batch(mvosr_ctx *ctx, const mvosr_params *p, const mvosr_batch *b, const double *height_level_in,
                           const mvosr_outputs *o, int waves_per_frame) {
    (void)waves_per_frame;                      // the road model always runs one wavefront per frame
    int rc = check_common(ctx, p, b, o);
    if (rc) return rc;
    if (!b->y) return set_error(MVOSR_ERR_ARG, "road_model: y plane required");
    if (!o->raw_scale || !o->height || !o->status) return set_error(MVOSR_ERR_ARG, "road_model: raw_scale/height/status required");
    if (b->n_frames == 0) return MVOSR_OK;
    if (b->total_feat <= 0) return set_error(MVOSR_ERR_ARG, "road_model: batch.total_feat (length of the y plane) not set");
    if ((rc = ctx_activate(ctx))) return rc;
    RoadArgs ra;
    double *scratch = nullptr;
    int32_t *unused = nullptr;
    if ((rc = ctx_workspace(ctx, b->n_frames, b->total_feat, &scratch, &unused))) return rc;
    ra.P = *p; ra.off = b->feat_off; ra.cnt = b->feat_cnt; ra.y = b->y; ra.scratch = scratch;
    ra.height_level = height_level_in; ra.o = *o;
    ra.first_frame = 0; ra.n_frames = b->n_frames; ra.pending_only = 0; ra.level_redo = nullptr; ra.list = nullptr;
    ra.wide = b->max_feat > 4 * kRoadRC * kWave ? 1 : 0;        // lists beyond four wavefronts' register caches
    return launch_road(ctx, ra, ctx_stream(ctx));
}

static int launch_window_median(mvosr_ctx *ctx, const double *raw, int64_t n, int n_blocks, int64_t stride, int window,
                                const double *queue_in, int n_queue, double *out) {
    if (!ctx || (n > 0 && (!raw || !out))) return set_error(MVOSR_ERR_ARG, "window_median: null argument");
    if (window < 1 || window > kMaxWindow) return set_error(MVOSR_ERR_ARG, "window_median: window must be in 1..%d", kMaxWindow);
    if (n_queue < 0 || n_queue > window || (n_queue > 0 && !queue_in)) return set_error(MVOSR_ERR_ARG, "window_median: bad carried-in queue");
    if (n_blocks < 1) return set_error(MVOSR_ERR_ARG, "window_median: n_blocks < 1");
    if (n <= 0) return MVOSR_OK;
    MedianArgs ma;
    ma.raw = raw; ma.out = out; ma.n = n; ma.window = window; ma.n_queue = n_queue;
    ma.n_blocks = n_blocks; ma.base_len = n / n_blocks; ma.extra = n % n_blocks; ma.stride = stride;
    if (n_blocks > 1 && stride < ma.base_len + (ma.extra ? 1 : 0)) return set_error(MVOSR_ERR_ARG, "window_median: block stride shorter than a block");
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    for (int i = 0; i < kMaxWindow; ++i) ma.queue[i] = (i < n_queue) ? queue_in[i] : 0.0;
    hipLaunchKernelGGL(window_median_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx_stream(ctx), ma);
    return check_launch("window_median_kernel");
}

int mvosr_window_median(mvosr_ctx *ctx, const double *raw, int64_t n, int window, const double *queue_in, int n_queue,
                        double *out) {
    return launch_window_median(ctx, raw, n, 1, 0, window, queue_in, n_queue, out);
}

int mvosr_window_median_blocked(mvosr_ctx *ctx, const double *blocks, int64_t n, int n_blocks, int64_t block_stride, int window,
                                const double *queue_in, int n_queue, double *out) {
    return launch_window_median(ctx, blocks, n, n_blocks, block_stride, window, queue_in, n_queue, out);
}

}  // extern "C"

#ifdef MVOSR_ABLATE
// diagnostic builds only (profiles/ab_build.sh ... -DMVOSR_ABLATE): the exact pass's list of the context's last mvosr_scale_batch —
// out[0] = how many frames, out[1..] = which (profiles/redo_list_census.py)
extern "C" int mvosr_debug_redo_list(mvosr_ctx *ctx, int64_t n_frames, int32_t *out, int cap) {
    if (!ctx || !out || cap < 1 || !ctx->ws_nsel) return MVOSR_ERR_ARG;
    (void)hipStreamSynchronize(ctx_stream(ctx));
    const hipError_t e = hipMemcpy(out, ctx->ws_nsel + n_frames, sizeof(int32_t) * (size_t)cap, hipMemcpyDeviceToHost);
    return e == hipSuccess ? MVOSR_OK : MVOSR_ERR_HIP;
}
#endif
