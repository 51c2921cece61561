/* mvosr_qhull_rows_host — `scipy.spatial.Delaunay(points).simplices`, row for row, on the HOST: the replay of Qhull's run that
 * qhull_rows_kernel (mvosr_qhull.hip) runs on the device, as one lean C loop for ONE point set.
 *
 * Why it exists: the reference calls SciPy's Delaunay on the host once per image (/root/reference/src/scale_calculator.py:257 from
 * src/main.py:110-113), and its vote reads the ROTATION of every row (:113-115) — Qhull's insertion order.  A single frame on the
 * device is a chain of ~n dependent insertions at ~10 us each (20 ms); SciPy itself is 2.6 ms at 2000 points, most of it Qhull's
 * general machinery (merge tests after every cone, vertex neighbourhoods, ridge hashing, the output structures SciPy builds:
 * neighbours, coplanar lists, the paraboloid).  The replay needs none of that for sites in general position: ~2000 insertions on
 * state that lives in the L1/L2 of one core.  It serves the per-frame call of the default estimator (first triangulation), and the
 * batch path's handful-of-frames case.
 *
 * What is replayed (qhull_r 7.3.2 = 2019.1.r as bundled with SciPy 1.15.3, options "Qbb Qc Qz Q12 Qt"; third-party, not under
 * /root/reference — the published algorithm: Barber, Dobkin, Huhdanpaa, "The Quickhull algorithm for convex hulls", ACM TOMS 1996):
 * lift to the paraboloid, the point at infinity of 'Qz', 'Qbb' scaling, the initial simplex from the extreme points, the first
 * partition, then per insertion: the first facet in list order with an outside set, its furthest point, the visible facets breadth
 * first in neighbour order, one cone facet per horizon ridge appended to the list, the directed walk (qh_findbest) / linear scan
 * (qh_findbestnew, after a sharp cone) that assigns the visible facets' points, outside-set order (furthest last).  Arithmetic in
 * Qhull's order of operations, no contraction (-ffp-contract=off).  A decision within 64 DISTround of a threshold — where Qhull
 * would merge facets or treat a point as coplanar — DECLINES the set (return > 0): never a guessed row.  The caller then asks SciPy.
 *
 * The same contract, decision for decision, as oracle/qhull_rows.py (test infrastructure; tests/test_qhull_host.py compares the two
 * and both with SciPy) — this file does not use it.  selfcheck.py compares this replay with the INSTALLED SciPy at first use.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QH_EPS 2.220446049250313e-16
#define QH_BIG 1.797e308
#define PX(p) pt[4 * (size_t)(p)]
#define PY(p) pt[4 * (size_t)(p) + 1]
#define PZ(p) pt[4 * (size_t)(p) + 2]

enum {  /* decline reasons (return value; 0 = rows written) — the device kernel's list, mvosr_qhull.hip */
    QHH_OK = 0, QHH_FEW_POINTS = 1, QHH_ZERO_WIDTH = 2, QHH_FLAT_SIMPLEX = 3, QHH_NARROW_SIMPLEX = 4, QHH_ONE_EXTREME = 5,
    QHH_SIMPLEX_SEARCH = 6, QHH_DEGENERATE_FACET = 7, QHH_NEAR_ZERO_PIVOT = 8, QHH_INITIAL_ROUNDOFF = 9, QHH_INSIDE_SIMPLEX = 10,
    QHH_PARTITION_ROUNDOFF = 11, QHH_ABOVE_NO_FACET = 12, QHH_VISIBILITY_ROUNDOFF = 13, QHH_COPLANAR_HORIZON = 14, QHH_OPEN_CONE = 15,
    QHH_CONE_NOT_CONVEX = 16, QHH_ROWS_OVERFLOW = 17
};
#define QHH_ERR_ALLOC (-5)   /* MVOSR_ERR_ALLOC */
#define QHH_ERR_ARG (-2)     /* MVOSR_ERR_ARG */

/* A facet in two halves: what a distance test and a walk touch — the plane, the neighbours, the visit stamp, the flags, the vertices —
 * is ONE 64-byte cache line; list links and the outside set live apart (a run is ~100 000 distance tests against ~30 000 list edits) */
typedef struct __attribute__((aligned(64))) {
    double n0, n1, n2, off;
    int32_t nb[3];       /* neighbour i is opposite vertex i */
    uint32_t visit;
    int32_t v[3];        /* vertex ids, newest first */
    uint8_t top, upper, visible, isnew;
} qh_facet;
typedef struct {
    double fdist;
    int32_t prev, next;  /* the facet list: order is part of the algorithm */
    int32_t out_first, out_last;   /* outside set: a doubly linked list through the points (a point is in one set at most) */
    int32_t replace;
    uint8_t has_out, pad[3];
} qh_cold;

typedef struct {
    int n, m;                       /* sites; points including the one at infinity */
    double *pt;                     /* one 32-byte record per point: x, y, lifted z, pad (a distance test touches one line of it) */
    int32_t *onext, *oprev;         /* outside-set links per point */
    int32_t *vpoint;                /* vertex id (from 1) -> point */
    int nvert;
    qh_facet *F;
    qh_cold *C;
    int ncap, nused, free_head;     /* facet slots: [0] is the list's tail sentinel; dead facets are reused */
    int head, facet_next;
    uint32_t visit_id;
    double nearzero[3], distround, anground, minvisible, minoutside, distoutside, maxwidth, guard;
    int32_t *visible, *newf, *stack;
    int listcap;
    int32_t *ridge_nf, *ridge_gen;  /* per vertex id: the cone facet waiting for its partner on the ridge {apex, vertex} */
    uint8_t *ridge_k;
    int newlist;
    int findbestnew, notsharp;
    int why;
} qh_state;

/* ---- per-thread workspace, grow-only (a call is ~0.5 ms: no malloc / page faults in steady state) ---- */
typedef struct { void *p; size_t cap; } qh_buf;
static __thread qh_buf g_bufs[12];

static void *ws_get(int slot, size_t bytes) {
    qh_buf *b = &g_bufs[slot];
    if (b->cap < bytes) {
        free(b->p);
        b->cap = bytes + bytes / 4 + 256;
        b->p = malloc(b->cap);
        if (!b->p) { b->cap = 0; return NULL; }
    }
    return b->p;
}

/* ---- list plumbing ---- */
static inline void fl_append(qh_state *S, int f) {
    qh_cold *C = S->C;
    const int t = 0;
    C[f].prev = C[t].prev;
    C[f].next = t;
    if (C[t].prev >= 0) C[C[t].prev].next = f; else S->head = f;
    C[t].prev = f;
    if (S->facet_next == t) S->facet_next = f;
}
static inline void fl_remove(qh_state *S, int f) {
    qh_cold *C = S->C;
    if (f == S->facet_next) S->facet_next = C[f].next;
    if (C[f].prev >= 0) C[C[f].prev].next = C[f].next; else S->head = C[f].next;
    C[C[f].next].prev = C[f].prev;
    C[f].prev = C[f].next = -1;
}
static inline void fl_prepend(qh_state *S, int f, int before) {
    qh_cold *C = S->C;
    C[f].prev = C[before].prev;
    C[f].next = before;
    if (C[before].prev >= 0) C[C[before].prev].next = f; else S->head = f;
    C[before].prev = f;
}

static int facet_new(qh_state *S) {
    int f;
    if (S->free_head >= 0) {
        f = S->free_head;
        S->free_head = S->C[f].next;
    } else {
        if (S->nused >= S->ncap) return -1;
        f = S->nused++;
    }
    qh_facet *q = &S->F[f];
    qh_cold *c = &S->C[f];
    c->prev = c->next = -1;
    c->out_first = c->out_last = -1;
    c->has_out = 0;
    c->fdist = 0.0;
    c->replace = -1;
    q->visible = 0; q->isnew = 0; q->visit = 0; q->upper = 0; q->top = 0;
    q->nb[0] = q->nb[1] = q->nb[2] = -1;
    return f;
}

/* ---- outside sets ---- */
static inline void out_append(qh_state *S, qh_cold *f, int p) {
    S->onext[p] = -1; S->oprev[p] = f->out_last;
    if (f->out_last >= 0) S->onext[f->out_last] = p; else f->out_first = p;
    f->out_last = p;
}
static inline void out_insert_before_last(qh_state *S, qh_cold *f, int p) {
    const int l = f->out_last, pl = S->oprev[l];
    S->onext[p] = l; S->oprev[p] = pl;
    S->oprev[l] = p;
    if (pl >= 0) S->onext[pl] = p; else f->out_first = p;
}
static inline int out_pop(qh_state *S, qh_cold *f) {
    const int p = f->out_last, pl = S->oprev[p];
    f->out_last = pl;
    if (pl >= 0) S->onext[pl] = -1; else f->out_first = -1;
    return p;
}
static inline void add_outside(qh_state *S, qh_cold *f, int p, double d) {
    if (f->out_first < 0) { f->out_first = f->out_last = -1; out_append(S, f, p); f->fdist = d; f->has_out = 1; }
    else if (f->fdist < d) { out_append(S, f, p); f->fdist = d; }
    else out_insert_before_last(S, f, p);
}

/* ---- geometry, in Qhull's order of operations ---- */
static inline double dist_pf(const qh_state *S, int p, const qh_facet *f) {
    const double *pt = S->pt;
    return ((f->off + PX(p) * f->n0) + PY(p) * f->n1) + PZ(p) * f->n2;
}

static int set_plane(qh_state *S, qh_facet *f) {
    const double *pt = S->pt;
    const int p0 = S->vpoint[f->v[0]], p1 = S->vpoint[f->v[1]], p2 = S->vpoint[f->v[2]];
    const double dx1 = PX(p1) - PX(p0), dy1 = PY(p1) - PY(p0), dz1 = PZ(p1) - PZ(p0);
    const double dx2 = PX(p2) - PX(p0), dy2 = PY(p2) - PY(p0), dz2 = PZ(p2) - PZ(p0);
    double n0 = dy2 * dz1 - dz2 * dy1;
    double n1 = dx1 * dz2 - dz1 * dx2;
    double n2 = dx2 * dy1 - dy2 * dx1;
    double norm = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
    if (!(norm > 1e-290)) return QHH_DEGENERATE_FACET;
    if (!f->top) norm = -norm;
    n0 = n0 / norm; n1 = n1 / norm; n2 = n2 / norm;
    f->n0 = n0; f->n1 = n1; f->n2 = n2;
    f->off = -((PX(p0) * n0 + PY(p0) * n1) + PZ(p0) * n2);
    int gauss = 0;
    {
        double d = f->off + ((PX(p2) * n0 + PY(p2) * n1) + PZ(p2) * n2);
        if (d > S->distround || d < -S->distround) gauss = 1;
        else {
            d = f->off + ((PX(p1) * n0 + PY(p1) * n1) + PZ(p1) * n2);
            if (d > S->distround || d < -S->distround) gauss = 1;
        }
    }
    if (gauss) {
        /* qh_sethyperplane_gauss: elimination with partial pivoting on the two edge vectors, back substitution from normal[2] = -+1 */
        double r0[3] = {dx1, dy1, dz1}, r1[3] = {dx2, dy2, dz2};
        int sign = f->top ? 1 : 0;
        if (fabs(r1[0]) > fabs(r0[0])) {
            double t;
            t = r0[0]; r0[0] = r1[0]; r1[0] = t;
            t = r0[1]; r0[1] = r1[1]; r1[1] = t;
            t = r0[2]; r0[2] = r1[2]; r1[2] = t;
            sign = !sign;
        }
        if (fabs(r0[0]) <= S->nearzero[0]) return QHH_NEAR_ZERO_PIVOT;
        const double q = r1[0] / r0[0];
        r1[1] -= q * r0[1];
        r1[2] -= q * r0[2];
        if (fabs(r1[1]) <= S->nearzero[1]) return QHH_NEAR_ZERO_PIVOT;
        if (r1[1] < 0) sign = !sign;
        if (r0[0] < 0) sign = !sign;
        n2 = sign ? -1.0 : 1.0;
        n1 = 0.0;
        n1 -= r1[2] * n2;
        n1 /= r1[1];
        n0 = 0.0;
        n0 -= r0[1] * n1;
        n0 -= r0[2] * n2;
        n0 /= r0[0];
        norm = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
        n0 = n0 / norm; n1 = n1 / norm; n2 = n2 / norm;
        f->n0 = n0; f->n1 = n1; f->n2 = n2;
        double off = -(PX(p0) * n0);
        off -= PY(p0) * n1;
        off -= PZ(p0) * n2;
        f->off = off;
    }
    f->upper = n2 > -S->anground * 2.0;
    return 0;
}

static double det_of(const qh_state *S, int apex, const int *pts, int dim, int *near) {
    const double *pt = S->pt;
    if (dim == 2) {
        const int a = pts[0], b = pts[1];
        const double r00 = PX(a) - PX(apex), r01 = PY(a) - PY(apex);
        const double r10 = PX(b) - PX(apex), r11 = PY(b) - PY(apex);
        const double det = r00 * r11 - r01 * r10;
        *near = fabs(det) < 10 * S->nearzero[1];
        return det;
    }
    const int a = pts[0], b = pts[1], c = pts[2];
    const double a1 = PX(a) - PX(apex), a2 = PY(a) - PY(apex), a3 = PZ(a) - PZ(apex);
    const double b1 = PX(b) - PX(apex), b2 = PY(b) - PY(apex), b3 = PZ(b) - PZ(apex);
    const double c1 = PX(c) - PX(apex), c2 = PY(c) - PY(apex), c3 = PZ(c) - PZ(apex);
    const double det = (a1 * (b2 * c3 - b3 * c2) - b1 * (a2 * c3 - a3 * c2)) + c1 * (a2 * b3 - a3 * b2);
    *near = fabs(det) < 10 * S->nearzero[2];
    return det;
}

/* ---- the searches of a partition (qh_findbest with isnewfacets, qh_findbestnew, qh_findbesthorizon) ---- */
#define BAND(d_) do { if ((d_) > -S->guard && (d_) < S->guard) { S->why = QHH_PARTITION_ROUNDOFF; return -1; } } while (0)

static int find_best_horizon(qh_state *S, int p, int start, double *bestd_io) {
    qh_facet *F = S->F;
    const uint32_t vid = ++S->visit_id;
    int best = start;
    double bestd = *bestd_io;
    const double searchdist = 4 * S->distround;
    double minsearch = bestd - searchdist;
    int nstack = 0;
    F[start].visit = vid;
    int f = start, nextfacet = -1;
    for (;;) {
        for (int k = 0; k < 3; ++k) {
            const int g = F[f].nb[k];
            if (F[g].visit == vid) continue;
            F[g].visit = vid;
            const double d = dist_pf(S, p, &F[g]);
            BAND(d);
            if (d > bestd) {
                if (!F[g].upper || d >= S->minoutside) {
                    minsearch = d - searchdist;
                    if (d > bestd + searchdist) nstack = 0;
                    best = g; bestd = d;
                }
            } else if (d < minsearch) continue;
            if (nextfacet >= 0) {
                if (nstack >= S->listcap) { S->why = QHH_ROWS_OVERFLOW; return -1; }
                S->stack[nstack++] = nextfacet;
            }
            nextfacet = g;
        }
        f = nextfacet;
        if (f >= 0) nextfacet = -1;
        else if (!nstack) break;
        else f = S->stack[--nstack];
    }
    *bestd_io = bestd;
    return best;
}

static int find_best_new(qh_state *S, int p, int start, double *dout) {
    qh_facet *F = S->F;
    const qh_cold *C = S->C;
    const uint32_t vid = ++S->visit_id;
    int best = -1;
    double bestd = -QH_BIG;
    for (int i = 0; i < 2; ++i) {
        int f = i == 0 ? start : S->newlist;
        while (f != 0) {
            if (f == start && i) break;
            F[f].visit = vid;
            const double d = dist_pf(S, p, &F[f]);
            BAND(d);
            if (d > bestd && (!F[f].upper || d >= S->minoutside)) {
                best = f;
                if (d >= S->distoutside) { *dout = d; return f; }
                bestd = d;
            }
            f = C[f].next;
        }
    }
    best = find_best_horizon(S, p, best >= 0 ? best : start, &bestd);
    if (best < 0) return -1;
    if (bestd < S->minoutside) { S->why = QHH_ABOVE_NO_FACET; return -1; }
    *dout = bestd;
    return best;
}

static int cone_is_sharp(const qh_state *S) {
    const qh_facet *F = S->F;
    const qh_cold *C = S->C;
    int f = S->newlist;
    const int q0 = F[f].n0 > 0, q1 = F[f].n1 > 0, q2 = F[f].n2 > 0;
    for (f = C[f].next; f != 0; f = C[f].next)
        if (q0 != (F[f].n0 > 0) || q1 != (F[f].n1 > 0) || q2 != (F[f].n2 > 0)) return 1;
    return 0;
}

static int partition_point(qh_state *S, int p, int start, double *dout) {
    qh_facet *F = S->F;
    if (S->findbestnew) return find_best_new(S, p, start, dout);
    const uint32_t vid = ++S->visit_id;
    double d = dist_pf(S, p, &F[start]);
    BAND(d);
    if (d >= S->minoutside) { *dout = d; return start; }
    double bestd = d;
    int best = F[start].upper ? -1 : start;
    F[start].visit = vid;
    int f = start;
    while (f >= 0) {
        int nxt = -1;
        for (int k = 0; k < 3; ++k) {
            const int g = F[f].nb[k];
            if (!F[g].isnew || F[g].visit == vid) continue;
            F[g].visit = vid;
            d = dist_pf(S, p, &F[g]);
            BAND(d);
            if (d > bestd) {
                if (d >= S->minoutside) { *dout = d; return g; }
                if (!F[g].upper) { best = g; bestd = d; nxt = g; break; }
                else if (best < 0) { bestd = d; nxt = g; break; }
            }
        }
        f = nxt;
    }
    if (best < 0) return find_best_new(S, p, S->newlist, dout);
    if (!S->notsharp && bestd < -S->distround) {
        if (cone_is_sharp(S)) { S->findbestnew = 1; return find_best_new(S, p, best, dout); }
        S->notsharp = 1;
    }
    best = find_best_horizon(S, p, best, &bestd);
    if (best < 0) return -1;
    if (bestd < S->minoutside) { S->why = QHH_ABOVE_NO_FACET; return -1; }
    *dout = bestd;
    return best;
}

/* ---- the run ---- */
static int qh_run(qh_state *S) {
    const int m = S->m;
    double *pt = S->pt;
    qh_facet *F = S->F;
    qh_cold *C = S->C;
    /* 2. extreme points per coordinate (first strict maximum / minimum in input order, maximum tested first), ranges */
    int maxpoints[6];
    double maxabs = 0.0, maxwidth = 0.0, maxsum = 0.0, zlow = 0.0, zhigh = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double *c = pt + k;          /* coordinate k of point i: c[4 i] */
        int lo = 0, hi = 0;
        for (int i = 0; i < m; ++i) {
            if (c[4 * (size_t)hi] < c[4 * (size_t)i]) hi = i;
            else if (c[4 * (size_t)lo] > c[4 * (size_t)i]) lo = i;
        }
        double maxcoord;
        const double clo = c[4 * (size_t)lo], chi = c[4 * (size_t)hi];
        if (k == 2) { zlow = clo; zhigh = chi; maxcoord = maxabs; }
        else {
            maxcoord = chi > -clo ? chi : -clo;
            if (chi - clo > maxwidth) maxwidth = chi - clo;
        }
        if (maxcoord > maxabs) maxabs = maxcoord;
        maxsum += maxcoord;
        maxpoints[2 * k] = lo; maxpoints[2 * k + 1] = hi;
        S->nearzero[k] = 80 * maxsum * QH_EPS;
    }
    if (maxwidth <= 0.0) return QHH_ZERO_WIDTH;
    /* 3. 'Qbb': the last coordinate scaled to [0, max |x or y|] */
    {
        const double scale = maxabs / (zhigh - zlow);
        const double shift = 0.0 - zlow * scale;
        for (int i = 0; i < m; ++i) PZ(i) = PZ(i) * scale + shift;
    }
    /* 4. roundoff constants */
    {
        const double a = sqrt(3.0) * maxabs;
        const double maxdistsum = a < maxsum ? a : maxsum;
        S->distround = QH_EPS * (3 * maxdistsum * 1.01 + maxabs);
        S->anground = 1.01 * 3 * QH_EPS;
        S->minvisible = 2 * S->distround;
        S->minoutside = 2 * S->minvisible;
        S->distoutside = 2 * S->minoutside;
        S->maxwidth = maxwidth;
        S->guard = 64 * S->distround;
    }
    /* 5. initial simplex: the extreme points of min x and max x, then twice the extreme point with the largest |determinant| */
    int simplex[4], ns = 0;
    {
        double maxc = -QH_BIG, minc = QH_BIG;
        int maxx = -1, minx = -1;
        for (int k = 0; k < 6; ++k) {
            const int p = maxpoints[k];
            if (maxc < PX(p)) { maxc = PX(p); maxx = p; }
            if (minc > PX(p)) { minc = PX(p); minx = p; }
        }
        simplex[ns++] = minx;
        if (maxx != minx) simplex[ns++] = maxx;
        if (ns < 2) return QHH_ONE_EXTREME;
        double maxdet = maxc - minc;
        for (int i = 2; i < 4; ++i) {
            const double prevdet = maxdet;
            int maxpoint = -1, maxnear = 0;
            maxdet = -1.0;
            for (int k = 0; k < 6; ++k) {
                const int p = maxpoints[k];
                int in = 0;
                for (int j = 0; j < ns; ++j) in |= simplex[j] == p;
                if (in) continue;
                int near;
                double det = fabs(det_of(S, p, simplex, i, &near));
                if (det > maxdet) { maxdet = det; maxpoint = p; maxnear = near; }
            }
            const double targetdet = prevdet * S->maxwidth;
            if (maxpoint < 0 || maxnear || (maxdet > 0.0 && maxdet / targetdet < 0.02)) return QHH_SIMPLEX_SEARCH;
            simplex[ns++] = maxpoint;
        }
    }
    /* 6. four facets, each omitting one vertex, orientation alternating, flipped as a whole if the centre lies outside the first */
    S->vpoint[0] = -1;
    for (int i = 0; i < 4; ++i) S->vpoint[i + 1] = simplex[i];
    S->nvert = 5;
    S->visit_id = 0;
    C[0].prev = -1; C[0].next = -1; C[0].out_first = C[0].out_last = -1; C[0].has_out = 0; F[0].isnew = 0; F[0].visit = 0; F[0].upper = 1;
    S->nused = 1; S->free_head = -1;
    S->head = 0; S->facet_next = 0;
    int fs[4];
    {
        const int verts[4] = {4, 3, 2, 1};
        int top = 1;
        for (int i = 0; i < 4; ++i) {
            const int f = facet_new(S);
            if (f < 0) return QHH_ROWS_OVERFLOW;
            int c = 0;
            for (int j = 0; j < 4; ++j) if (j != i) F[f].v[c++] = verts[j];
            F[f].top = (uint8_t)top;
            top = !top;
            fl_append(S, f);
            fs[i] = f;
        }
        for (int i = 0; i < 4; ++i) { int c = 0; for (int j = 0; j < 4; ++j) if (j != i) F[fs[i]].nb[c++] = fs[j]; }
        double cx = 0.0, cy = 0.0, cz = 0.0;
        for (int j = 0; j < 4; ++j) { const int p = S->vpoint[verts[j]]; cx += PX(p); cy += PY(p); cz += PZ(p); }
        cx = cx / 4; cy = cy / 4; cz = cz / 4;
        int rc = set_plane(S, &F[fs[0]]);
        if (rc) return rc;
        const double d = ((F[fs[0]].off + cx * F[fs[0]].n0) + cy * F[fs[0]].n1) + cz * F[fs[0]].n2;
        if (d > S->distround) {
            for (int i = 0; i < 4; ++i) F[fs[i]].top = !F[fs[i]].top;
            if ((rc = set_plane(S, &F[fs[0]]))) return rc;
        } else if (d > -S->distround) return QHH_FLAT_SIMPLEX;
        for (int i = 1; i < 4; ++i) if ((rc = set_plane(S, &F[fs[i]]))) return rc;
        double minangle = 2.0;
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 3; ++k) {
                const qh_facet *a = &F[fs[i]], *b = &F[a->nb[k]];
                const double c = (a->n0 * b->n0 + a->n1 * b->n1) + a->n2 * b->n2;
                if (c < minangle) minangle = c;
            }
        if (minangle < -0.99999999) return QHH_NARROW_SIMPLEX;
    }
    /* 7. every other point to the FIRST facet (list order) it lies above by 8 DISTround; a facet keeps its furthest point last.
     * The point set shrinks from facet to facet: kept as a compact array (S->visible's room is free until the loop starts). */
    {
        int32_t *pointset = S->stack;          /* (m entries fit: listcap >= m) */
        int np = 0;
        for (int p = 0; p < m; ++p) {
            if (p == simplex[0] || p == simplex[1] || p == simplex[2] || p == simplex[3]) continue;
            pointset[np++] = p;
        }
        for (int f = S->head; f != 0; f = C[f].next) {
            int nrest = 0, best = -1;
            double bestd = 0.0;
            const qh_facet *q = &F[f];
            qh_cold *qc = &C[f];
            for (int i = 0; i < np; ++i) {
                const int p = pointset[i];
                const double d = dist_pf(S, p, q);
                if (d < S->distoutside) {
                    pointset[nrest++] = p;
                    if (d > -S->guard && d > S->distoutside - 2 * S->guard) return QHH_INITIAL_ROUNDOFF;
                } else {
                    if (best < 0) { best = p; bestd = d; }
                    else if (d > bestd) { out_append(S, qc, best); best = p; bestd = d; }
                    else out_append(S, qc, p);
                }
            }
            if (best >= 0) { out_append(S, qc, best); qc->fdist = bestd; qc->has_out = 1; }
            np = nrest;
        }
        if (np) return QHH_INSIDE_SIMPLEX;
    }
    /* 8. the facet with the furthest point of all moves to the head of the list */
    {
        int best = -1;
        double bestd = -QH_BIG;
        for (int f = S->head; f != 0; f = C[f].next)
            if (C[f].has_out && C[f].fdist > bestd) { best = f; bestd = C[f].fdist; }
        S->facet_next = S->head;
        if (best >= 0) {
            fl_remove(S, best);
            fl_prepend(S, best, S->facet_next);
            S->facet_next = best;
        }
    }
    /* 9. the loop */
    uint32_t ridge_generation = 0;
    for (;;) {
        int f = S->facet_next;
        while (f != 0 && C[f].out_first < 0) { C[f].has_out = 0; f = C[f].next; }
        S->facet_next = f;
        if (f == 0) break;
        const int p = out_pop(S, &C[f]);
        /* the visible facets, breadth first in neighbour order */
        fl_remove(S, f);
        fl_append(S, f);
        F[f].visible = 1; C[f].replace = -1;
        int nvis = 0;
        S->visible[nvis++] = f;
        const uint32_t vid = ++S->visit_id;
        for (int i = 0; i < nvis; ++i) {
            const int vis = S->visible[i];
            F[vis].visit = vid;
            for (int k = 0; k < 3; ++k) {
                const int g = F[vis].nb[k];
                if (F[g].visit == vid) continue;
                F[g].visit = vid;
                const double d = dist_pf(S, p, &F[g]);
                if (d > S->minvisible) {
                    if (d < S->guard) return QHH_VISIBILITY_ROUNDOFF;
                    fl_remove(S, g);
                    fl_append(S, g);
                    F[g].visible = 1; C[g].replace = -1;
                    if (nvis >= S->listcap) return QHH_ROWS_OVERFLOW;
                    S->visible[nvis++] = g;
                } else if (d >= -S->guard) return QHH_COPLANAR_HORIZON;
            }
        }
        /* the cone: per visible facet in that order, per horizon neighbour in neighbour order, a new facet (apex first) at the END */
        const int apex = S->nvert;
        S->vpoint[S->nvert++] = p;
        int nnew = 0;
        for (int i = 0; i < nvis; ++i) {
            const int vis = S->visible[i];
            int last = -1;
            for (int k = 0; k < 3; ++k) {
                const int g = F[vis].nb[k];
                if (F[g].visible) continue;
                const int skip = F[g].nb[0] == vis ? 0 : (F[g].nb[1] == vis ? 1 : 2);
                const int nf = facet_new(S);
                if (nf < 0 || nnew >= S->listcap) return QHH_ROWS_OVERFLOW;
                F = S->F;
                int c = 1;
                F[nf].v[0] = apex;
                for (int j = 0; j < 3; ++j) if (j != skip) F[nf].v[c++] = F[g].v[j];
                F[nf].top = F[g].top ? (uint8_t)(skip & 1) : (uint8_t)!(skip & 1);
                F[nf].nb[0] = g;
                F[nf].isnew = 1;
                fl_append(S, nf);
                F[g].nb[skip] = nf;
                S->newf[nnew++] = nf;
                last = nf;
            }
            if (last >= 0) C[vis].replace = last;
        }
        if (!nnew) return QHH_OPEN_CONE;
        S->newlist = S->newf[0];
        /* match: neighbour k (k = 1, 2) shares the ridge {apex, v[3 - k]} */
        {
            ++ridge_generation;
            int open = 0;
            for (int i = 0; i < nnew; ++i) {
                const int nf = S->newf[i];
                for (int k = 1; k <= 2; ++k) {
                    const int key = F[nf].v[3 - k];
                    if (S->ridge_gen[key] == (int32_t)ridge_generation && S->ridge_nf[key] >= 0) {
                        const int other = S->ridge_nf[key], ok = S->ridge_k[key];
                        F[nf].nb[k] = other;
                        F[other].nb[ok] = nf;
                        S->ridge_nf[key] = -1;
                        --open;
                    } else {
                        S->ridge_gen[key] = (int32_t)ridge_generation;
                        S->ridge_nf[key] = nf;
                        S->ridge_k[key] = (uint8_t)k;
                        ++open;
                    }
                }
            }
            if (open) return QHH_OPEN_CONE;
        }
        for (int i = 0; i < nnew; ++i) { const int rc = set_plane(S, &F[S->newf[i]]); if (rc) return rc; }
        /* convexity of the cone (Qhull would merge): the vertex of each neighbour opposite the shared ridge must lie below */
        for (int i = 0; i < nnew; ++i) {
            const int nf = S->newf[i];
            for (int k = 0; k < 3; ++k) {
                const int g = F[nf].nb[k];
                const int j = F[g].nb[0] == nf ? 0 : (F[g].nb[1] == nf ? 1 : 2);
                const int q = S->vpoint[F[g].v[j]];
                if (dist_pf(S, q, &F[nf]) > -S->guard) return QHH_CONE_NOT_CONVEX;
            }
        }
        /* the visible facets' points: to the first new facet a directed walk from the visible facet's replacement finds them above */
        S->findbestnew = 0; S->notsharp = 0;
        for (int i = 0; i < nvis; ++i) {
            const int vis = S->visible[i];
            if (C[vis].out_first < 0) continue;
            const int start = C[vis].replace >= 0 ? C[vis].replace : S->newlist;
            int q = C[vis].out_first;
            while (q >= 0) {
                const int qn = S->onext[q];
                double d;
                const int g = partition_point(S, q, start, &d);
                if (g < 0) return S->why;
                if (C[g].out_first < 0 && !F[g].isnew) { fl_remove(S, g); fl_append(S, g); }   /* an old facet takes a point */
                add_outside(S, &C[g], q, d);
                q = qn;
            }
            C[vis].out_first = C[vis].out_last = -1;
        }
        for (int i = 0; i < nvis; ++i) {
            const int vis = S->visible[i];
            fl_remove(S, vis);
            F[vis].visible = 0;
            C[vis].next = S->free_head;       /* (free list through `next`) */
            S->free_head = vis;
        }
        for (int i = 0; i < nnew; ++i) F[S->newf[i]].isnew = 0;
    }
    return 0;
}

/* rows: SciPy's — lower facets in list order; vertices by decreasing vertex id, first two swapped when NOT top-oriented */
static int emit_rows(const qh_state *S, int32_t *rows, int rows_cap) {
    const qh_facet *F = S->F;
    const qh_cold *C = S->C;
    int t = 0;
    for (int f = S->head; f != 0; f = C[f].next) {
        if (F[f].upper) continue;
        if (t >= rows_cap) return -1;
        const int a = S->vpoint[F[f].v[0]], b = S->vpoint[F[f].v[1]], c = S->vpoint[F[f].v[2]];
        if (F[f].top) { rows[3 * t] = a; rows[3 * t + 1] = b; }
        else { rows[3 * t] = b; rows[3 * t + 1] = a; }
        rows[3 * t + 2] = c;
        ++t;
    }
    return t;
}

/* include/mvosr.h */
int mvosr_qhull_rows_host(const double *points, int64_t n_points, int64_t stride_doubles, int32_t *rows, int64_t rows_cap,
                          int32_t *n_rows, int32_t *order) {
    if (!points || !rows || !n_rows || n_points < 0 || stride_doubles < 2 || n_points > 0x3fffffff) return QHH_ERR_ARG;
    *n_rows = 0;
    if (n_points < 3) return QHH_FEW_POINTS;
    const int n = (int)n_points, m = n + 1;
    qh_state S;
    memset(&S, 0, sizeof S);
    S.n = n; S.m = m;
    /* a run creates ~5.6 facets per point and keeps ~2 alive: slots are reused, 4 n + 64 never ran out on 60 000 sets (else: declined) */
    S.ncap = 4 * m + 64;
    S.listcap = 2 * m + 64;
    {
        char *raw = (char *)ws_get(0, sizeof(double) * 4 * (size_t)m + 64);
        S.pt = raw ? (double *)(((uintptr_t)raw + 63) & ~(uintptr_t)63) : NULL;
    }
    S.onext = (int32_t *)ws_get(1, sizeof(int32_t) * 2 * (size_t)m);
    S.vpoint = (int32_t *)ws_get(2, sizeof(int32_t) * ((size_t)m + 8));
    {
        char *raw = (char *)ws_get(3, sizeof(qh_facet) * (size_t)S.ncap + 64);
        S.F = raw ? (qh_facet *)(((uintptr_t)raw + 63) & ~(uintptr_t)63) : NULL;
    }
    S.C = (qh_cold *)ws_get(7, sizeof(qh_cold) * (size_t)S.ncap);
    S.visible = (int32_t *)ws_get(4, sizeof(int32_t) * 3 * (size_t)S.listcap);
    S.ridge_nf = (int32_t *)ws_get(5, sizeof(int32_t) * 2 * ((size_t)m + 8));
    S.ridge_k = (uint8_t *)ws_get(6, (size_t)m + 8);
    if (!S.pt || !S.onext || !S.vpoint || !S.F || !S.C || !S.visible || !S.ridge_nf || !S.ridge_k) return QHH_ERR_ALLOC;
    double *pt = S.pt;
    S.oprev = S.onext + m;
    S.newf = S.visible + S.listcap; S.stack = S.newf + S.listcap;
    S.ridge_gen = S.ridge_nf + (m + 8);
    memset(S.ridge_gen, 0, sizeof(int32_t) * ((size_t)m + 8));
    /* 1. lift: z = x*x + y*y; the point "at infinity" of 'Qz' = (mean x, mean y, 1.1 * max z), id n */
    {
        double sx = 0.0, sy = 0.0, maxb = -QH_BIG;
        for (int i = 0; i < n; ++i) {
            const double px = points[(size_t)i * (size_t)stride_doubles], py = points[(size_t)i * (size_t)stride_doubles + 1];
            if (!(px == px) || !(py == py) || fabs(px) > 1e150 || fabs(py) > 1e150) return QHH_ZERO_WIDTH;   /* NaN / inf / overflowing squares */
            const double pz = px * px + py * py;
            PX(i) = px; PY(i) = py; PZ(i) = pz; pt[4 * (size_t)i + 3] = 0.0;
            sx += px; sy += py;
            if (pz > maxb) maxb = pz;
        }
        PX(n) = sx / n; PY(n) = sy / n; PZ(n) = maxb * 1.1; pt[4 * (size_t)n + 3] = 0.0;
    }
    const int rc = qh_run(&S);
    if (rc) return rc;
    const int t = emit_rows(&S, rows, (int)(rows_cap > 0x3fffffff ? 0x3fffffff : rows_cap));
    if (t < 0) return QHH_ROWS_OVERFLOW;
    *n_rows = t;
    if (order)   /* insertion step of every site (the initial simplex first; the point at infinity counts as a step) */
        for (int v = 1; v < S.nvert; ++v) if (S.vpoint[v] < n) order[S.vpoint[v]] = v - 1;
    return 0;
}
