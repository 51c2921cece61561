// mvosr_delaunay.hip — batched 2-D Delaunay triangulation on the GPU: the two
// scipy.spatial.Delaunay calls of /root/reference/src/scale_calculator.py:257-258,266-267, which cost 3-3.5 ms each
// on a host core and bound the end-to-end rate of the drop-in path, as a device stage whose rows stay on the device.
//
// What it returns: for points in general position the Delaunay triangle SET is unique, and this kernel returns exactly
// that set (tested against SciPy) in a CANONICAL row form — vertex ids ascending inside a row, rows in lexicographic
// order — a function of the set alone.  What it cannot return is Qhull's rotation of each row, a by-product of Qhull's
// processing order that the reference's depth-order vote depends on (scale_calculator.py:113-115 sets flag[1] where
// flag[2] is meant).  The rows are therefore meant for the order-invariant vote (check_triangle="fixed",
// mvosr_params.vote_mode = MVOSR_VOTE_FIXED), with which SciPy's rows and these rows give bit-identical results; with the
// reference's flag pattern they are a measured deviation.
//
// Algorithm (round 3; the round-2 kernel scanned all points of the frame for every point): one workgroup of 8
// wavefronts per frame, the frame's points in LDS in fp64, counting-sorted into a uniform grid of ~1.5 points per cell
// (cell id = row-major, so one row of a cell box is one contiguous range of the sorted array).  Every point builds
// its own Delaunay star, independently of all others (no shared mutable structure, nothing ordered between stars):
//   ONE LANE PER POINT, the lanes persistent (a lane whose star is complete takes the next point).  A star starts at
//     the point's nearest neighbour within its 5x5 cell block (a Delaunay neighbour) and is wrapped counter-clockwise:
//     the apex of the triangle on the left of the directed edge (p, q) is the candidate c that sees the edge under the
//     largest angle (smallest cot = (c-p).(c-q) / cross(q-p, c-p), compared by cross-multiplication: no division or
//     square root per candidate; one branch-free step per candidate).  Every iteration of the loop is one scan step
//     per lane — up to five cell rows, cut down to the wanted side of the edge, and 32 candidates — so lanes in
//     different stages of different stars share every iteration.  A completion whose circumcircle lies within the
//     block is final; otherwise the search goes on over the circle's cell box, and where the block holds nothing on
//     the wanted side over the whole frame (hull vertices, points next to long hull slivers: the star is then wrapped
//     clockwise from its first neighbour as well);
//   a GROUP OF 16 LANES (one DPP row) per point for the few points that own more rows or have a larger star than a
//     lane keeps in registers.
// A triangle is written by its smallest vertex, so it appears once; a point's rows are sorted and the points' row counts
// prefix-summed in point order, so the output does not depend on scheduling.
//
// Robustness: plain fp64 predicates with guard bands.  A frame in which a decision falls inside a band — two
// candidates with (nearly) the same cot: four cocircular points; a point (nearly) on the line through an edge where
// that can decide a triangle (between the edge's ends; beyond them only on a hull edge, and not where the collinearity
// is exact: dt_step<true>, round 6 — sites on a pixel grid are collinear in threes everywhere without being
// degenerate); duplicate points — or whose rows fail Euler's relation (2n - 2 - h rows, 3 star triangles per row) is flagged
// MVOSR_DT_DEGENERATE and left to the host's Qhull (SciPy resolves such inputs by its own rules).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mvosr.h"
#include "mvosr_device.hpp"
#include "mvosr_host.hpp"

namespace mvosr {

constexpr int kDtWaves = 8;
constexpr int kDtBlock = kDtWaves * kWave;
#ifndef MVOSR_DT_SMALL_LADDER
#define MVOSR_DT_SMALL_LADDER 1
#endif
constexpr int kDtLadderMinFrames = 512;      // two eight-wavefront frames on each of 256 CUs
constexpr bool kDtSmallLadder = MVOSR_DT_SMALL_LADDER != 0;
#ifndef MVOSR_DT_ARENA_OUT
#define MVOSR_DT_ARENA_OUT 1
#endif
constexpr bool kDtArenaOut = MVOSR_DT_ARENA_OUT != 0;
#ifndef MVOSR_DT_WIDE16
#define MVOSR_DT_WIDE16 1
#endif
constexpr bool kDtWide16 = MVOSR_DT_WIDE16 != 0;          // sixteen wavefronts per frame for launches of a few frames (see the launcher)
constexpr int kDtWide16MaxFrames = 128;   // three four-wavefront frames per CU with the rows' arena in global memory (see the launcher)   // 2- and 4-wavefront instantiations for small frames (see the launcher)
#ifndef MVOSR_DT_R
#define MVOSR_DT_R 2
#endif
#ifndef MVOSR_DT_PER_CELL
#define MVOSR_DT_PER_CELL 1.5
#endif
constexpr int kDtR = MVOSR_DT_R;         // a point's candidates: the (2R+1)^2 cell block around its cell
#ifndef MVOSR_DT_COLOUR
#define MVOSR_DT_COLOUR 1
#endif
constexpr bool kDtColour = MVOSR_DT_COLOUR != 0;    // points taken colour by colour of their cells ((x & 1, y & 1)): see the kernel
constexpr double kDtPerCell = MVOSR_DT_PER_CELL;   // target points per cell (measured trade-off: profiles/micro/dt_proto.py)
constexpr int kDtMaxCells = 4096;
#ifndef MVOSR_DT_LANE_ROWS
#define MVOSR_DT_LANE_ROWS 12
#endif
constexpr int kDtLaneRows = MVOSR_DT_LANE_ROWS;   // rows a point may own on the lane path (more: the group pass)
constexpr int kDtLaneDeg = 24;           // star degree on the lane path
#ifndef MVOSR_DT_BUDGET
#define MVOSR_DT_BUDGET 48
#endif
#ifndef MVOSR_DT_RWIDE
#define MVOSR_DT_RWIDE 8
#endif
constexpr int kDtBudget = MVOSR_DT_BUDGET;   // candidates per lane and scan step
#ifndef MVOSR_DT_COOP_CELLS
#define MVOSR_DT_COOP_CELLS 36
#endif
constexpr int kDtCoopCells = MVOSR_DT_COOP_CELLS;    // a circumcircle's cell box larger than this is scanned by the whole wavefront
constexpr int kDtRWide = MVOSR_DT_RWIDE;              // the block a search is widened to before it takes the whole frame
constexpr int kDtWaveRows = 32;          // rows a point may own at all
constexpr int kDtWaveDeg = 60;
constexpr int kDtHardCap = 256;          // points left to the group pass
constexpr int kDtArenaSlack = 64;
constexpr double kDtTieTol = 1e-9;       // relative guard band on cot differences
constexpr double kDtColTol = 1e-12;      // relative guard band on collinearity (|cross| <= tol |a| |b|)
constexpr double kDtColSeg = (1.0 + 1e-9) / (kDtColTol * kDtColTol);      // a2col * kDtColSeg = |a|^2 (1 + 1e-9): a collinear candidate with a larger dot lies beyond q

// why a frame was declined (status bits 8..)
enum { DT_WHY_DUP = 1, DT_WHY_TIE = 2, DT_WHY_COLLINEAR = 4, DT_WHY_DEGREE = 8, DT_WHY_ROWS = 16, DT_WHY_EULER = 32,
       DT_WHY_HARD = 64, DT_WHY_SIZE = 128 };

struct DtArgs {
    int64_t n_frames;
    const int64_t *pts_off; const int32_t *pts_cnt;      // [F] the frame's points in u/v
    const double *u, *v;
    const int32_t *keep;                                 // laid out like u, or null: a point takes part iff keep[i] >= 0
    const int64_t *tri_off;                              // [F] start of the frame's rows in `tri` (capacity 2*n rows)
    int32_t *tri;                                        // rows (a, b, c), a < b < c
    int32_t *tri_cnt;                                    // [F] rows written
    int32_t *n_used;                                     // [F] points triangulated (null: not wanted)
    int32_t *status;                                     // [F] MVOSR_DT_*
    int max_pts;
    char *ws;                                            // GLOBAL variant: a slice of dt_plan().big bytes per frame
    char *aws;                                           // ARENA_OUT variant: a slice of dt_plan().out_bytes per frame (arena + starts)
    uint32_t *hints;                                     // LDS variant: kDtHintK (+1: see seeds) words per point and frame (all ones = empty), or null
    // seeds (mvosr_delaunay_batch_seeded): rows of a triangulation of ALL the frame's points (ids = positions in u/v).  A
    // triangle of it whose three vertices are kept is a triangle of this one — its circumcircle was empty among more
    // points — so its three corners go into the hint caches before the first star is started.
    const int64_t *seed_off; const int32_t *seed_tri; const int32_t *seed_cnt;
    // Per point of the SEED triangulation (laid out like u), written by the launch that built it (info_out) and read by the
    // seeded launch (seed_info): rows owned | star degree << 6 | open << 15 in the low half, index of the point's first row
    // within the frame's rows in the high half.  A kept point none of whose seed triangles lost a vertex has the SAME star in
    // this triangulation — the triangles stay Delaunay among fewer points and still close the fan — so its rows are copied
    // (ids mapped to ranks) and its star is not walked at all: at 95 % kept points that is three stars in four.
    const uint32_t *seed_info; uint32_t *info_out;
    // PARTS variant (a launch of a few frames — the per-frame call): `parts` workgroups per frame, each with the whole frame in
    // its LDS, each building the stars of its own strip of cell columns; what they found meets in `pg` (DtPartsPlan, one slice per frame)
    // and the last workgroup to arrive writes the rows.
    int parts; char *pg; unsigned int *ph;       // ph: 16 words per frame — [0] parts arrived, [1] their decline flags (zero between launches)
#ifdef MVOSR_STAMPS
    unsigned long long *stamps;                          // diagnostic builds: 16 values per frame (phase boundaries, list lengths)
#endif
};

#ifdef MVOSR_STAMPS
static unsigned long long *g_dt_stamps = nullptr;
// (PARTS: a row of 64 values per part)
#define DT_STAMP(i) do { if (tid == 0 && a.stamps) a.stamps[64 * (PARTS ? (int64_t)blockIdx.x : f) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DT_NOTE(i, v) do { if (tid == 0 && a.stamps) a.stamps[64 * (PARTS ? (int64_t)blockIdx.x : f) + (i)] = (unsigned long long)(v); } while (0)
#else
#define DT_STAMP(i) do {} while (0)
#define DT_NOTE(i, v) do {} while (0)
#endif

// The per-frame arrays: the BIG part (points, ids, per-point row bookkeeping, cell index, row arena: ~33 B per point)
// lives in the workgroup's LDS for frames up to ~4700 points, and in a per-frame slice of the context's workspace (read
// through L1/L2) for larger ones (the GLOBAL kernel variant: dense frames, config C5); the small part always in LDS.
// Every Delaunay triangle is found from each of its three vertices' stars.  The first finder tells the other two: a
// triangle (p, q, c) found counter-clockwise in p's star says "after c comes p" in q's star and "after p comes q" in c's
// — one word (from << 16 | to, indices into the sorted array) dropped into a small direct-mapped cache per point in global
// memory (slot = from mod kDtHintK; a collision overwrites: lossy, never wrong — the reader matches `from` exactly).  A lane
// that completes an edge of its star looks the next edge up before it searches; what it finds was certified by the lane
// that published it (its circumcircle lay within the cells that lane had scanned).  Relaxed device-scope loads and stores:
// a hint that is not visible yet is a search done twice, nothing else.  Measured at 2000 points: 43 % of the 11 936
// triangle corners of a set are taken from a hint, 5.7 busy scan steps per point instead of 9.0.
#ifndef MVOSR_DT_HINTS
#define MVOSR_DT_HINTS 16
#endif
constexpr int kDtHintK = MVOSR_DT_HINTS;       // 0: no hints
#ifndef MVOSR_DT_SCOPE
#define MVOSR_DT_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
constexpr int kDtScope = MVOSR_DT_SCOPE;
#ifndef MVOSR_DT_CHAIN
#define MVOSR_DT_CHAIN 8
#endif
#ifndef MVOSR_DT_HINT_START
#define MVOSR_DT_HINT_START 1
#endif
#ifndef MVOSR_DT_SERVE_ROWS
#define MVOSR_DT_SERVE_ROWS 4
#endif
constexpr int kDtServeRows = MVOSR_DT_SERVE_ROWS;      // (a power of two)
#ifndef MVOSR_DT_COOP
#define MVOSR_DT_COOP 1
#endif
constexpr int kDtHintChain = MVOSR_DT_CHAIN;
#ifndef MVOSR_DT_GLOBAL_HINTS
#define MVOSR_DT_GLOBAL_HINTS 1
#endif
template <bool GLOBAL> constexpr bool kDtHintsOn = kDtHintK > 0 && (!GLOBAL || MVOSR_DT_GLOBAL_HINTS);
#ifndef MVOSR_DT_GLOBAL_COOP
#define MVOSR_DT_GLOBAL_COOP 0
#endif
template <bool GLOBAL> constexpr bool kDtCoop = MVOSR_DT_COOP && (!GLOBAL || MVOSR_DT_GLOBAL_COOP);    // (frames in global memory: 26 k -> 19 k sets/s with it at 20 000 points)   // hinted triangles taken in a row before the lane goes back to searching

struct DtPlan { uint32_t S, oid, od, astart, cs, arena, big, hard, wrows, red, misc, wsl, aff, total, out_bytes; int max_cells, arena_cap; };
constexpr int kDtMaxCellsGlobal = 32768;
constexpr int kDtMaxPointsGlobal = 32000;      // (row arena indices and point ids are 16-bit)

__host__ __device__ inline int dt_cell_cap(int max_pts, bool global) {
    int c = (int)((double)max_pts / kDtPerCell * 1.15) + 64;        // (an elongated frame's grid may want more: it is then made coarser, see the kernel)
    const int cap = global ? kDtMaxCellsGlobal : kDtMaxCells;
    return c > cap ? cap : c;
}
// arena_out: the rows' arena and the points' starts in it live in a per-frame slice of GLOBAL memory (out_bytes; offsets astart and
// arena are into that slice) instead of LDS: 20 KB less per 2000-point frame, which is what lets THREE such frames share a CU
__host__ __device__ inline DtPlan dt_plan(int max_pts, bool global = false, int waves = kDtWaves, bool arena_out = false) {
    DtPlan p;
    const uint32_t npad = (uint32_t)((max_pts + 7) & ~7);
    p.max_cells = dt_cell_cap(max_pts, global);
    p.arena_cap = (int)(2u * npad) + kDtArenaSlack;
    p.S = 0;                                             // double2 per point (sorted by cell)
    p.oid = p.S + 16u * npad;                            // u16: sorted index -> id of the point
    p.od = p.oid + 2u * npad;                            // u16 per id: rows owned | star degree << 6 | listed << 14 | open << 15
    p.astart = p.od + 2u * npad;                         // u16 per id: the point's rows in the arena
    p.cs = p.astart + 2u * npad + 4u;                    // u32 per cell (+1): end of the cell in the sorted array; cs[-1] = 0 (the start of cell 0)
    p.arena = p.cs + 4u * (uint32_t)(p.max_cells + 8);   // u32 rows (b << 16 | c) in the order they were found
    p.big = (p.arena + 4u * (uint32_t)p.arena_cap + 255u) & ~255u;
    p.out_bytes = 0u;
    if (arena_out) {
        p.cs = p.od + 2u * npad + 4u;
        p.big = (p.cs + 4u * (uint32_t)(p.max_cells + 8) + 255u) & ~255u;
        p.astart = 0u;
        p.arena = (2u * npad + 15u) & ~15u;
        p.out_bytes = (p.arena + 4u * (uint32_t)p.arena_cap + 255u) & ~255u;
    }
    p.hard = p.big;                                      // u16 sorted indices
    p.wrows = p.hard + 2u * kDtHardCap;                  // u32 [groups of 16 lanes][kDtWaveRows] (phase 2)
    {
        const uint32_t rows2 = 4u * (uint32_t)(waves * kWave / 16) * kDtWaveRows;
        // (the flags of the stars to walk — aff, a byte per point, LDS variant — are dead when phase 1 begins, phase 2's rows
        // are not alive before: they share their room)
        const uint32_t affb = global ? 0u : ((npad + 7u) & ~7u);
        p.aff = p.wrows;
        p.red = p.wrows + (rows2 > affb ? rows2 : affb);  // doubles: block reductions
    }
    p.misc = p.red + 8u * 4u * (uint32_t)waves;
    p.wsl = p.misc + 4u * 64u;                           // int [4][16]: per-wavefront counts and partial sums (up to 16 wavefronts)
    p.total = p.wsl + 4u * 64u;
    return p;
}

// PARTS: a frame's meeting place in global memory — per point its `od` word and the start of its rows; the parts' arenas one
// after the other
struct DtPartsPlan { uint32_t od, start, arena, total; };
__host__ __device__ inline DtPartsPlan dt_parts_plan(int max_pts, int parts) {
    DtPartsPlan p;
    const uint32_t npad = (uint32_t)((max_pts + 7) & ~7);
    p.od = 0u;
    p.start = p.od + 2u * npad;
    p.arena = p.start + 4u * npad;
    p.total = (p.arena + 4u * (uint32_t)parts * (uint32_t)(2u * npad + kDtArenaSlack) + 255u) & ~255u;
    return p;
}
constexpr int kDtPartsMaxFrames = 16;      // launches of up to this many frames take the PARTS variant
constexpr int kDtPartsPoints = 128;        // points per part (four wavefronts: at most a star per lane)
constexpr int kDtPartsMax = 16;

enum { DM_FLAGS = 0, DM_ARENA = 1, DM_VQ = 2, DM_NHARD = 3, DM_NEXT = 4, DM_TICKET = 5, DM_PFLAGS = 6 };
enum { DW_CNT = 0, DW_SUM = 16, DW_SUM2 = 32, DW_SUM3 = 48 };        // wsl[]: 16 slots each

struct DtGrid {
    double lo_u, lo_v, ix, iy, sx, sy;
    int gx, gy;
    const uint32_t *cs;
    __device__ __forceinline__ int cellx(double x) const { return (int)fmin(fmax((x - lo_u) * ix, 0.0), (double)(gx - 1)); }
    __device__ __forceinline__ int celly(double y) const { return (int)fmin(fmax((y - lo_v) * iy, 0.0), (double)(gy - 1)); }
    __device__ __forceinline__ int row_begin(int cy, int cxa) const { return (int)cs[cy * gx + cxa - 1]; }     // (cs[-1] = 0)
    __device__ __forceinline__ int row_end(int cy, int cxb) const { return (int)cs[cy * gx + cxb]; }
};

struct DtBox { int xa, xb, ya, yb; };

// the directed edge p -> q = p + a of a completion.  sgn = +1: the apex is wanted on the left, -1: on the right.
// (k, side) describe the wanted half-plane per cell row: side = +1: u below px + k (v - py), -1: above, 0: no statement
struct DtEdge {
    double px, py, ax, ay, a2col, sgn, k;
    double sax, say, cr_add;       // lane pass: sgn * a, and what is added to the cross product (0; 1 in the nearest-neighbour mode)
    int i, iq, side;
    __device__ __forceinline__ void set(double2 p, double2 q, int i_, int iq_, double sgn_) {
        px = p.x; py = p.y; ax = q.x - p.x; ay = q.y - p.y; i = i_; iq = iq_; sgn = sgn_;
        a2col = kDtColTol * kDtColTol * (ax * ax + ay * ay);
        sax = sgn * ax; say = sgn * ay; cr_add = 0.0;
        side = say > 0.0 ? 1 : (say < 0.0 ? -1 : 0);
        k = side ? sax / say : 0.0;
    }
    // the nearest-neighbour search as the same minimisation: with a = 0 the step's numerator is |c - p|^2, its cross
    // product 0 + 1, nothing is "on the line" (a2col < 0) and no row is cut
    __device__ __forceinline__ void set_nn(double2 p, int i_) {
        px = p.x; py = p.y; ax = 0.0; ay = 0.0; sax = 0.0; say = 0.0; cr_add = 1.0; a2col = -1.0; sgn = 1.0; k = 0.0;
        i = i_; iq = -1; side = 0;
    }
};

// The best apex seen so far as a fraction num / cr (cr > 0): t = cot of the angle under which the candidate sees the
// edge.  Lane pass: `tie` = a candidate came within the guard band of the best OF ITS TIME — which covers every
// candidate within the band of the FINAL best (a better one that arrives later is compared with the best of its own
// time, itself at least as good as the earlier candidate) but also chance encounters with an intermediate best, so a
// set flag is only a reason to look again (dt_confirm_tie).  Group passes keep the runner-up and decide exactly.
struct DtAcc {
    double n1, c1, s1, n2, c2;      // (lane pass: best + `tie`; group passes: best and runner-up)
    int b1, flag, tie;
    __device__ __forceinline__ void reset() { n1 = 1e300; c1 = 1.0; s1 = kDtTieTol * 1e300; n2 = 1e300; c2 = 1.0; b1 = -1; flag = 0; tie = 0; }
};

constexpr int kDtGroup = 16;            // lanes that share a completion in the group passes: one DPP row

// one candidate of a group pass, branch-free.
// TWO (the pass over the hard points; round 6): the collinearity flag in two kinds.  Bit 0: a candidate (nearly) on the line ahead of p
// UP TO q — on the segment, or a duplicate of p or q: the edge is no Delaunay edge, or the sign of cr decides a triangle.  Bit 1:
// BEYOND q: such a candidate sees the edge under a zero angle, it is never the apex where anything else lies on the wanted side,
// whichever side of the line rounding puts it on — it only matters where nothing does (a hull edge: a sliver triangle or not?), and
// the completion looks at it only then (dt_col_declines) — and not at all where the collinearity is EXACT (below).  Coordinates
// quantised to 1/16 px: two frames in three hold such a triple (three sites on a grid line within a cell block), nearly always of the
// second kind: 65 % of the frames were declined; 5 % with the interior edges let through, 1.8 % with the exactly collinear hull
// points too (what is left: repeated sites, cocircular quadruples).  The lane pass and the wide searches keep the one flag and hand a
// flagged point to this pass.
// d == x - y computed in fp64: was that subtraction exact?  (Knuth's TwoSum error term; -ffp-contract=off keeps it as written)
__device__ __forceinline__ bool dt_exact_diff(double x, double y, double d) {
    const double bb = d - x;
    return (x - (d - bb)) - (y + bb) == 0.0;
}

template <bool TWO = false>
__device__ __forceinline__ void dt_step(DtAcc &A, const DtEdge &E, int j, double2 c, double2 q = double2{0.0, 0.0}) {
    const double bx = c.x - E.px, by = c.y - E.py;
    const double cr = E.sgn * __builtin_fma(E.ax, by, -(E.ay * bx));
    const double b2 = __builtin_fma(bx, bx, by * by);
    const double dot = __builtin_fma(bx, E.ax, by * E.ay);
    const double num = b2 - dot;                                        // (c - p).(c - q)
    const bool skip = (j == E.i) | (j == E.iq);
    const bool col = cr * cr <= E.a2col * b2;                           // (nearly) on the line through the edge ...
    if constexpr (TWO) {
        const bool ahead = !skip & col & ((dot > 0.0) | (b2 == 0.0));
        const bool beyond = dot > E.a2col * kDtColSeg;
        // EXACTLY collinear (every difference exact, the cross product exactly zero: t = ay bx rounded, ax by - t == 0 and ay bx - t == 0
        // as FMAs, i.e. without rounding): beyond q of a hull edge that is no sliver in anybody's arithmetic — q is a hull vertex between
        // p and c, no triangle (Qhull: the facet through the three lifted points is vertical, not a lower one).  Coordinates on a grid.
        A.flag |= (ahead & !beyond) ? 1 : 0;
        if (ahead & beyond) {
            asm volatile("");                      // (a region the wavefront skips — never entered on points in general position —, not a select:
                                                   // the pass over the hard points runs for a few points of EVERY frame)
            const double t = E.ay * bx;
            const bool exact = dt_exact_diff(q.x, E.px, E.ax) & dt_exact_diff(q.y, E.py, E.ay) & dt_exact_diff(c.x, E.px, bx) & dt_exact_diff(c.y, E.py, by) &
                               (__builtin_fma(E.ax, by, -t) == 0.0) & (__builtin_fma(E.ay, bx, -t) == 0.0);
            A.flag |= exact ? 0 : 2;
        }
    } else
    A.flag |= (!skip & col & ((dot > 0.0) | (b2 == 0.0))) ? 1 : 0;      // ... ahead of p: the sign of cr would decide a triangle
    const bool ok = !skip & !col & (cr > 0.0);
    const double d = __builtin_fma(num, A.c1, -(A.n1 * cr));            // num / cr < n1 / c1  <=>  d < 0
    const bool better = ok & (d < 0.0);
    const bool second = ok & !better & (__builtin_fma(num, A.c2, -(A.n2 * cr)) < 0.0);
    A.n2 = better ? A.n1 : (second ? num : A.n2); A.c2 = better ? A.c1 : (second ? cr : A.c2);
    A.n1 = better ? num : A.n1; A.c1 = better ? cr : A.c1; A.b1 = better ? j : A.b1;
}

// one candidate of the lane pass (as dt_step; best + `tie` only).  m1 = the lane is wrapping its star; otherwise it is
// looking for its point's nearest neighbour, which the edge's set_nn() turns into the same arithmetic.
__device__ __forceinline__ void dt_step_lane(DtAcc &A, const DtEdge &E, bool m1, int j, double2 c) {
    const double bx = c.x - E.px, by = c.y - E.py;
    const double cr = __builtin_fma(E.sax, by, __builtin_fma(-E.say, bx, E.cr_add));
    const double b2 = __builtin_fma(bx, bx, by * by);
    const double dot = __builtin_fma(bx, E.ax, by * E.ay);
    const double num = b2 - dot;                                        // (c - p).(c - q)
    const bool skip = (j == E.i) | (j == E.iq);
    const bool col = cr * cr <= E.a2col * b2;
    A.flag |= (!skip & col & ((dot > 0.0) | (b2 == 0.0))) ? 1 : 0;
    const bool ok = !skip & !col & (cr > 0.0);
    const double d = __builtin_fma(num, A.c1, -(A.n1 * cr));
    const bool better = ok & (d < 0.0);
    A.tie |= (m1 & ok & (fabs(d) <= (kDtTieTol * (fabs(A.n1) + A.c1)) * cr)) ? 1 : 0;
    A.n1 = better ? num : A.n1; A.c1 = better ? cr : A.c1; A.b1 = better ? j : A.b1;
}

// the part of cell row y, columns xa..xb, that can hold points on the wanted side of the edge, as a range of the sorted array
__device__ __forceinline__ void dt_row_range(const DtGrid &G, const DtEdge &E, int y, int xa, int xb, int &j0, int &j1) {
    if (E.side) {
        const double v0 = G.lo_v + ((double)y - 1e-6) * G.sy, v1 = G.lo_v + ((double)(y + 1) + 1e-6) * G.sy;
        const double t0 = E.k * (v0 - E.py), t1 = E.k * (v1 - E.py);
        if (E.side > 0) xb = min(xb, G.cellx(E.px + fmax(t0, t1)) + 1);
        else xa = max(xa, G.cellx(E.px + fmin(t0, t1)) - 1);
    }
    j0 = 0; j1 = 0;
    if (xa <= xb) { j0 = G.row_begin(y, xa); j1 = G.row_end(y, xb); }
}

// The same for the lane pass, where every lane has its own edge: without branches (lanes with side > 0, side < 0 and side = 0
// share a wavefront: the branches ran one after the other, 55 instructions per row and five rows per scan step), the same
// arithmetic.  `ok` = the row is wanted at all; a row that is not, or is cut away, is the empty range 0..0.
__device__ __forceinline__ void dt_row_range_lane(const DtGrid &G, const DtEdge &E, int y, bool ok, int xa, int xb, int &j0, int &j1) {
    const double v0 = G.lo_v + ((double)y - 1e-6) * G.sy, v1 = G.lo_v + ((double)(y + 1) + 1e-6) * G.sy;
    const double t0 = E.k * (v0 - E.py), t1 = E.k * (v1 - E.py);
    const double m = E.side > 0 ? fmax(t0, t1) : fmin(t0, t1);
    const int c = G.cellx(E.px + m) + E.side;
    xb = E.side > 0 ? min(xb, c) : xb;
    xa = E.side < 0 ? max(xa, c) : xa;
    const int row = min(y, G.gy - 1) * G.gx;
    const int b = (int)G.cs[row + xa - 1], e = (int)G.cs[row + xb];
    ok = ok && xa <= xb;
    j0 = ok ? b : 0; j1 = ok ? e : 0;
}

// Lane pass, cold: is any candidate of the block other than the winner within the guard band of the winner?
// (Everything by VALUE: a reference parameter of an out-of-line function gives its argument a home in scratch memory, and the
// loop around the call kept the edge, the box and the grid up to date there — a dozen scratch stores per scan step.)
__device__ __attribute__((noinline)) bool dt_confirm_tie(const double2 *S, const uint32_t *cs, int gx, int xa, int xb, int ya, int yb,
                                                         double px, double py, double ax, double ay, double sgn, double a2col,
                                                         int ei, int eiq, int b1, double n1, double c1) {
    const double s1 = kDtTieTol * (fabs(n1) + c1);
    bool tie = false;
    for (int y = ya; y <= yb; ++y) {
        const int j1 = (int)cs[y * gx + xb];
        for (int j = (int)cs[y * gx + xa - 1]; j < j1; ++j) {
            if (j == ei || j == eiq || j == b1) continue;
            const double2 c = S[j];
            const double bx = c.x - px, by = c.y - py;
            const double cr = sgn * __builtin_fma(ax, by, -(ay * bx));
            const double b2 = __builtin_fma(bx, bx, by * by);
            if (cr * cr <= a2col * b2 || !(cr > 0.0)) continue;
            const double num = b2 - __builtin_fma(bx, ax, by * ay);
            if (fabs(__builtin_fma(num, c1, -(n1 * cr))) <= s1 * cr) tie = true;
        }
    }
    return tie;
}

// a cell box, GL lanes striding over each row
template <int GL, bool TWO = false>
__device__ __forceinline__ void dt_scan_box(DtAcc &A, const double2 *S, const DtGrid &G, const DtBox &B, const DtEdge &E, double2 q = double2{0.0, 0.0}) {
    const int gl = lane_id() & (GL - 1);
    for (int y = B.ya; y <= B.yb; ++y) {
        int j0, j1;
        dt_row_range(G, E, y, B.xa, B.xb, j0, j1);
        for (int j = j0 + gl; j < j1; j += GL) dt_step<TWO>(A, E, j, S[j], q);
    }
}

// Does a collinearity flag of the two-kind form decline the frame?  Bit 0 always; bit 1 only where the search found nothing on the
// wanted side (id < 0).
__device__ __forceinline__ bool dt_col_declines(int flag, int id) { return (flag & 1) || ((flag & 2) && id < 0); }

constexpr int kDtRows = 2 * kDtR + 1;

// ---- groups of 16 lanes (one DPP row): min, ballot, broadcast
__device__ __forceinline__ double dt_group_min(double x) {
    x = fmin(x, dpp_mov<kDppXor1>(x)); x = fmin(x, dpp_mov<kDppXor2>(x)); x = fmin(x, dpp_mov<kDppHalfMirror>(x)); x = fmin(x, dpp_mov<kDppMirror>(x));
    return x;
}
__device__ __forceinline__ unsigned dt_group_ballot(bool b) { return (unsigned)((__ballot(b) >> (lane_id() & 48)) & 0xFFFFull); }
__device__ __forceinline__ int dt_group_shfl(int v, int src) { return __shfl(v, (lane_id() & 48) | src); }

__device__ __forceinline__ double dt_wave_min(double x) {
    x = dt_group_min(x);
    return fmin(fmin(readlane_d(x, 0), readlane_d(x, 16)), fmin(readlane_d(x, 32), readlane_d(x, 48)));
}

struct DtPick { int id, tie, flag; };
// the group's answer from its lanes' accumulators
__device__ __forceinline__ DtPick dt_group_pick(const DtAcc &A) {
    const double t1 = A.b1 >= 0 ? A.n1 / A.c1 : INFINITY;
    const double m = dt_group_min(t1);
    DtPick r;
    const unsigned who = dt_group_ballot(A.b1 >= 0 && t1 == m);
    r.id = dt_group_shfl(A.b1, who ? (int)__ffs((int)who) - 1 : 0);
    if (!who) r.id = -1;
    const double band = kDtTieTol * (fabs(m) + 1.0);
    const bool close = (A.b1 >= 0 && A.b1 != r.id && t1 - m <= band) || (A.n2 / A.c2 - m <= band);
    r.tie = (r.id >= 0) && dt_group_ballot(close) != 0u;
    r.flag = (dt_group_ballot((A.flag & 1) != 0) != 0u ? 1 : 0) | (dt_group_ballot((A.flag & 2) != 0) != 0u ? 2 : 0);
    return r;
}

// the wavefront's answer (a wide search that all 64 lanes scan together)
__device__ __forceinline__ DtPick dt_wave_pick(const DtAcc &A) {
    const double t1 = A.b1 >= 0 ? A.n1 / A.c1 : INFINITY;
    const double m = dt_wave_min(t1);
    DtPick r;
    const unsigned long long who = __ballot(A.b1 >= 0 && t1 == m);
    r.id = who ? __builtin_amdgcn_readlane(A.b1, (int)__ffsll((long long)who) - 1) : -1;
    const double band = kDtTieTol * (fabs(m) + 1.0);
    const bool close = (A.b1 >= 0 && A.b1 != r.id && t1 - m <= band) || (A.n2 / A.c2 - m <= band);
    r.tie = (r.id >= 0) && __ballot(close) != 0ull;
    r.flag = __ballot(A.flag != 0) != 0ull;
    return r;
}

// 1 / x and sqrt(x) to ~1e-9 (the hardware's estimate and one Newton step: 3 and 5 instructions against 16 and 22 for the
// correctly rounded ones).  For the cell boxes below, which only have to CONTAIN their circle: the radius is stretched by 1e-6.
__device__ __forceinline__ double dt_rcp_fast(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double dt_sqrt_fast(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s = x * y, h = 0.5 * y;
    return __builtin_fma(s, __builtin_fma(-h, s, 0.5), s);
}
constexpr double kDtBoxStretch = 1.0 + 1e-6;

// cell box of the circle through p, q, c (any orientation), clamped to the grid
__device__ __forceinline__ DtBox dt_circle_box(const DtGrid &G, double px, double py, double2 q, double2 c) {
    const double ax = q.x - px, ay = q.y - py, bx = c.x - px, by = c.y - py;
    const double cr = ax * by - ay * bx, a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
    const double inv = 0.5 * dt_rcp_fast(cr);
    const double ox = (by * a2 - ay * b2) * inv, oy = (ax * b2 - bx * a2) * inv;
    const double r = dt_sqrt_fast(ox * ox + oy * oy) * kDtBoxStretch;
    DtBox B;
    B.xa = G.cellx(px + ox - r); B.xb = G.cellx(px + ox + r); B.ya = G.celly(py + oy - r); B.yb = G.celly(py + oy + r);
    if (!(r < INFINITY)) { B.xa = 0; B.xb = G.gx - 1; B.ya = 0; B.yb = G.gy - 1; }      // (NaN / overflow / a zero radius: everything)
    return B;
}
__device__ __forceinline__ DtBox dt_disc_box(const DtGrid &G, double px, double py, double d2) {
    const double r = dt_sqrt_fast(d2) * kDtBoxStretch;
    DtBox B;
    B.xa = G.cellx(px - r); B.xb = G.cellx(px + r); B.ya = G.celly(py - r); B.yb = G.celly(py + r);
    if (!(r < INFINITY)) { B.xa = 0; B.xb = G.gx - 1; B.ya = 0; B.yb = G.gy - 1; }      // (d2 = 0: a duplicate point, the frame is declined anyway)
    return B;
}
__device__ __forceinline__ bool dt_inside(const DtBox &B, const DtBox &blk) {
    return B.xa >= blk.xa && B.xb <= blk.xb && B.ya >= blk.ya && B.yb <= blk.yb;
}

__device__ __forceinline__ int dt_incl_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}

template <bool GLOBAL, int WAVES = kDtWaves, bool ARENA_OUT = false, bool PARTS = false>
__global__ __launch_bounds__(WAVES *kWave, PARTS ? 1 : ((ARENA_OUT && WAVES == 4) ? 3 : 4)) void delaunay_kernel(const DtArgs a) {
    static_assert(!PARTS || (!GLOBAL && !ARENA_OUT), "the PARTS variant keeps a frame in every part's LDS");
    constexpr int BLOCK = WAVES * kWave;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t f = PARTS ? (int64_t)(blockIdx.x / (unsigned)a.parts) : (int64_t)blockIdx.x;
    const int part = PARTS ? (int)(blockIdx.x % (unsigned)a.parts) : 0;
    const int64_t hslice = PARTS ? (int64_t)blockIdx.x : f;          // (every part sorts the points its own way: caches of its own)
    const int n_in = a.pts_cnt[f];
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const DtPlan L = dt_plan(a.max_pts, GLOBAL, WAVES, ARENA_OUT);
    char *big = GLOBAL ? a.ws + (size_t)f * L.big : smem;                  // the frame's big arrays
    char *small = GLOBAL ? smem - L.big : smem;                            // (the plan's offsets of the small ones start at L.big)
    double2 *S = reinterpret_cast<double2 *>(big + L.S);
    uint16_t *oid = reinterpret_cast<uint16_t *>(big + L.oid);
    uint16_t *od = reinterpret_cast<uint16_t *>(big + L.od);
    char *outp = ARENA_OUT ? a.aws + (size_t)f * L.out_bytes : big;           // (the arena and the starts: LDS, or the frame's global slice)
    uint16_t *astart = reinterpret_cast<uint16_t *>(outp + L.astart);
    uint32_t *cs = reinterpret_cast<uint32_t *>(big + L.cs);
    uint32_t *arena = reinterpret_cast<uint32_t *>(outp + L.arena);
    uint16_t *hard = reinterpret_cast<uint16_t *>(small + L.hard);
    double *red = reinterpret_cast<double *>(small + L.red);
    int *misc = reinterpret_cast<int *>(small + L.misc);
    int *wsl = reinterpret_cast<int *>(small + L.wsl);
    uint8_t *aff = reinterpret_cast<uint8_t *>(small + L.aff);
    const size_t hint_pts = (size_t)((a.max_pts + 7) & ~7);
    uint32_t *hints = (kDtHintsOn<GLOBAL> && a.hints) ? a.hints + (size_t)hslice * ((size_t)(kDtHintK + 3) * hint_pts) : nullptr;
    uint32_t *start = hints ? hints + (size_t)(kDtHintK + 1) * hint_pts : nullptr;               // one known triangle per point: its star starts there
    uint32_t *inv = (hints && a.seed_tri) ? hints + (size_t)kDtHintK * hint_pts : nullptr;       // position in u/v -> sorted index (seeds only)
    uint32_t *order = (hints && kDtColour && !GLOBAL) ? hints + (size_t)(kDtHintK + 2) * hint_pts : nullptr; // the order in which the points are taken
    if (hints) {
        // all ones = empty.  (One hipMemsetAsync over the launch's caches instead held the HOST for the GPU's queue above
        // 256 MB: the chunk loop around this kernel ran at 146 k instead of 228 k frames/s.)
        const int n_words = kDtHintK * min(n_in, (int)hint_pts);
        uint4 ones; ones.x = ones.y = ones.z = ones.w = 0xFFFFFFFFu;
        for (int k = tid; k < (n_words + 3) / 4; k += BLOCK) reinterpret_cast<uint4 *>(hints)[k] = ones;
        for (int k = tid; k < min(n_in, (int)hint_pts); k += BLOCK) start[k] = 0xFFFFFFFFu;
        if (inv) for (int k = tid; k < min(n_in, (int)hint_pts); k += BLOCK) inv[k] = 0xFFFFFFFFu;
    }
    // "in x's star, after `key` comes ...": slot key mod kDtHintK of x's cache; x's known triangle
    auto hint_put = [&](uint32_t x, uint32_t key, uint32_t val) { __hip_atomic_store(hints + (size_t)x * kDtHintK + (key % kDtHintK), val, __ATOMIC_RELAXED, kDtScope); };
    auto hint_get = [&](uint32_t x, uint32_t key) -> uint32_t { return __hip_atomic_load(hints + (size_t)x * kDtHintK + (key % kDtHintK), __ATOMIC_RELAXED, kDtScope); };
    auto start_put = [&](uint32_t x, uint32_t val) { __hip_atomic_store(start + x, val, __ATOMIC_RELAXED, kDtScope); };
    auto start_get = [&](uint32_t x) -> uint32_t { return __hip_atomic_load(start + x, __ATOMIC_RELAXED, kDtScope); };

    auto decline = [&](int why, int n_used) {
        if (tid == 0) { a.tri_cnt[f] = 0; a.status[f] = MVOSR_DT_DEGENERATE | (why << 8); if (a.n_used) a.n_used[f] = n_used; }
    };
    if (n_in > a.max_pts || n_in < 0) { decline(DT_WHY_SIZE, 0); return; }
    DT_STAMP(0);
    const int64_t off = a.pts_off[f];
    const double *gu = a.u + off, *gv = a.v + off;
    const int32_t *gk = a.keep ? a.keep + off : nullptr;

    // ---- pass 0: survivors per wavefront slice (ids are ranks among the survivors, in order), bounding box
    const int per = ((n_in + BLOCK - 1) / BLOCK) * kWave;          // slice of a wavefront: a multiple of 64
    const int s_begin = w * per, s_end = min(n_in, s_begin + per);
    double lo_u = INFINITY, hi_u = -INFINITY, lo_v = INFINITY, hi_v = -INFINITY;
    int wcnt = 0;
    // A lane's points — up to kDtOwn of them: 2 048 points on four wavefronts — stay in registers for the three passes that need
    // them (bounding box, count, scatter): loaded once, all at once.  (Each pass loaded them again, one dependent global load
    // per point and pass: the scatter alone was 73 k of a carried-over second triangulation's 930 k cycles.)
    constexpr int kDtOwn = 8;
    const bool own = per <= kDtOwn * kWave;
    double own_u[kDtOwn], own_v[kDtOwn];
    int own_pos[kDtOwn];                                        // (where the scatter put them)
    unsigned own_k = 0u;
    if (own) {
#pragma unroll
        for (int r = 0; r < kDtOwn; ++r) {
            const int i = s_begin + r * kWave + lane;
            bool k = i < s_end;
            if (k && gk) k = gk[i] >= 0;
            own_k |= k ? (1u << r) : 0u;
            own_u[r] = k ? gu[i] : 0.0; own_v[r] = k ? gv[i] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < kDtOwn; ++r) {
            const bool k = (own_k >> r) & 1u;
            if (k) { lo_u = fmin(lo_u, own_u[r]); hi_u = fmax(hi_u, own_u[r]); lo_v = fmin(lo_v, own_v[r]); hi_v = fmax(hi_v, own_v[r]); }
            wcnt += __popcll(__ballot(k));
        }
    } else
    for (int i0 = s_begin; i0 < s_end; i0 += kWave) {
        const int i = i0 + lane;
        bool k = i < s_end;
        if (k && gk) k = gk[i] >= 0;
        if (k) {
            const double pu = gu[i], pv = gv[i];
            lo_u = fmin(lo_u, pu); hi_u = fmax(hi_u, pu); lo_v = fmin(lo_v, pv); hi_v = fmax(hi_v, pv);
        }
        wcnt += __popcll(__ballot(k));
    }
    {
        const double a0 = dt_wave_min(lo_u), a1 = dt_wave_min(-hi_u), a2 = dt_wave_min(lo_v), a3 = dt_wave_min(-hi_v);
        if (lane == 0) { red[4 * w] = a0; red[4 * w + 1] = a1; red[4 * w + 2] = a2; red[4 * w + 3] = a3; wsl[DW_CNT + w] = wcnt; }
        if (tid < 8) misc[tid] = tid == DM_NEXT ? BLOCK : 0;
#ifdef MVOSR_STAMPS
        if ((tid >= 40 && tid < 64) || (tid >= 24 && tid < 40)) misc[tid] = 0;    // (24..39: scratch of the last stage, free until then)
#endif
    }
    __syncthreads();
    int n = 0, rank_base = 0;
    lo_u = INFINITY; hi_u = INFINITY; lo_v = INFINITY; hi_v = INFINITY;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        const int c = wsl[DW_CNT + i];
        if (i < w) rank_base += c;
        n += c;
        lo_u = fmin(lo_u, red[4 * i]); hi_u = fmin(hi_u, red[4 * i + 1]); lo_v = fmin(lo_v, red[4 * i + 2]); hi_v = fmin(hi_v, red[4 * i + 3]);
    }
    hi_u = -hi_u; hi_v = -hi_v;
    if (n < 3) { decline(DT_WHY_SIZE, n); return; }
    const double W = hi_u - lo_u, H = hi_v - lo_v;
    if (!(W > 0.0 && H > 0.0 && W < INFINITY && H < INFINITY)) { decline(DT_WHY_COLLINEAR, n); return; }   // (also NaN coordinates)

    // ---- the grid
    DtGrid G;
    {
        const double s = sqrt(W * H * kDtPerCell / (double)n);
        double fx = ceil(W / s), fy = ceil(H / s);
        fx = fmin(fmax(fx, 1.0), (double)L.max_cells); fy = fmin(fmax(fy, 1.0), (double)L.max_cells);
        if (fx * fy > (double)L.max_cells) {
            const double k = sqrt((double)L.max_cells / (fx * fy));
            fx = fmax(1.0, floor(fx * k)); fy = fmax(1.0, floor(fy * k));
            while (fx * fy > (double)L.max_cells) { if (fx >= fy) fx -= 1.0; else fy -= 1.0; }
        }
        G.gx = (int)fx; G.gy = (int)fy;
        G.lo_u = lo_u; G.lo_v = lo_v; G.ix = fx / W; G.iy = fy / H; G.sx = W / fx; G.sy = H / fy; G.cs = cs;
    }
    const int ncell = G.gx * G.gy;
    // PARTS: the part a cell's stars belong to — a strip of cell columns, so that every part has its share of the long top and
    // bottom hulls (their stars are the long ones: bands of cell rows left the first and the last part with twice the others'
    // time: 277 against 227 us per launch of a 2000-point frame in eight parts)
    auto part_of = [&](int c) -> int { return min(a.parts - 1, (int)(((int64_t)(c % G.gx) * a.parts) / G.gx)); };
    DT_STAMP(1);
    DT_STAMP(48);
    for (int c = tid - 1; c <= ncell; c += BLOCK) cs[c] = 0u;          // (from cs[-1] on)
    for (int i = tid; i < ((n + 1) >> 1); i += BLOCK) reinterpret_cast<uint32_t *>(od)[i] = 0u;
    __syncthreads();

    // ---- pass 1: points per cell
    if (own) {
#pragma unroll
        for (int r = 0; r < kDtOwn; ++r) if ((own_k >> r) & 1u) atomicAdd(&cs[G.celly(own_v[r]) * G.gx + G.cellx(own_u[r])], 1u);
    } else
    for (int i0 = s_begin; i0 < s_end; i0 += kWave) {
        const int i = i0 + lane;
        bool k = i < s_end;
        if (k && gk) k = gk[i] >= 0;
        if (k) atomicAdd(&cs[G.celly(gv[i]) * G.gx + G.cellx(gu[i])], 1u);
    }
    __syncthreads();
    DT_STAMP(49);
    // exclusive scan over the cells (cs[c] = start of cell c; pass 2 advances it to the cell's end)
    {
        const int cper = (ncell + BLOCK - 1) / BLOCK;
        const int c0 = tid * cper, c1 = min(ncell, c0 + cper);
        int mine = 0;
        for (int c = c0; c < c1; ++c) mine += (int)cs[c];
        const int incl = dt_incl_scan(mine);
        if (lane == kWave - 1) wsl[DW_SUM + w] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) if (i < w) base += wsl[DW_SUM + i];
        int at = base + incl - mine;
        for (int c = c0; c < c1; ++c) { const int k = (int)cs[c]; cs[c] = (uint32_t)at; at += k; }
    }
    __syncthreads();
    DT_STAMP(50);
    // ---- pass 2: scatter (the order inside a cell is whatever the atomics give: no output depends on it)
    {
        int rank = rank_base;
        if (own) {
#pragma unroll
            for (int r = 0; r < kDtOwn; ++r) {
                const bool k = (own_k >> r) & 1u;
                const unsigned long long m = __ballot(k);
                if (k) {
                    double2 p; p.x = own_u[r]; p.y = own_v[r];
                    const int pos = (int)atomicAdd(&cs[G.celly(p.y) * G.gx + G.cellx(p.x)], 1u);
                    S[pos] = p;
                    oid[pos] = (uint16_t)(rank + __popcll(m & ((1ull << lane) - 1ull)));
                    if (inv) __hip_atomic_store(inv + (s_begin + r * kWave + lane), (uint32_t)pos, __ATOMIC_RELAXED, kDtScope);
                    own_pos[r] = pos;
                }
                rank += __popcll(m);
            }
        } else
        for (int i0 = s_begin; i0 < s_end; i0 += kWave) {
            const int i = i0 + lane;
            bool k = i < s_end;
            if (k && gk) k = gk[i] >= 0;
            const unsigned long long m = __ballot(k);
            if (k) {
                double2 p; p.x = gu[i]; p.y = gv[i];
                const int pos = (int)atomicAdd(&cs[G.celly(p.y) * G.gx + G.cellx(p.x)], 1u);
                S[pos] = p;
                oid[pos] = (uint16_t)(rank + __popcll(m & ((1ull << lane) - 1ull)));
                if (inv) __hip_atomic_store(inv + i, (uint32_t)pos, __ATOMIC_RELAXED, kDtScope);
            }
            rank += __popcll(m);
        }
    }
    __syncthreads();
    DT_STAMP(51);
    // carry: stars without a lost neighbour are copied from the seed triangulation instead of walked (LDS variant, seeds with info)
    const bool carry = !GLOBAL && inv && order && a.seed_info && a.seed_cnt[f] > 0;
    int n_work = n;                                           // points whose star phase 1 builds
    if (inv) {
        const int32_t *st = a.seed_tri + 3 * a.seed_off[f];
        const int ns = a.seed_cnt[f];
        if (carry) {
            if constexpr (PARTS) {
                // 2: the star belongs to another part (neither walked nor carried here)
                for (int c = tid; c < ncell; c += BLOCK) {
                    const int b = c ? (int)cs[c - 1] : 0, e = (int)cs[c];
                    const uint8_t v_ = part_of(c) == part ? 0 : 2;
                    for (int j = b; j < e; ++j) aff[j] = v_;
                }
            } else
            for (int j = tid; j < n; j += BLOCK) aff[j] = 0;
            __syncthreads();
            // a seed row that lost a vertex: its other vertices' stars change.  (Four rows at a time: their twelve ids, then their
            // twelve positions, are in flight together — one row at a time was fifteen times two dependent global loads per lane.)
            for (int r0 = tid; r0 < ns; r0 += 4 * BLOCK) {
                int ids[4][3];
                uint32_t ps[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = min(r0 + q * BLOCK, ns - 1);
                    ids[q][0] = st[3 * r]; ids[q][1] = st[3 * r + 1]; ids[q][2] = st[3 * r + 2];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        ps[q][c] = (unsigned)ids[q][c] < (unsigned)n_in ? __hip_atomic_load(inv + ids[q][c], __ATOMIC_RELAXED, kDtScope) : 0xFFFFFFFEu;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (r0 + q * BLOCK >= ns) continue;
                    if (ps[q][0] == 0xFFFFFFFEu || ps[q][1] == 0xFFFFFFFEu || ps[q][2] == 0xFFFFFFFEu) continue;     // (an id out of range: not a row of this frame)
                    const bool ka = ps[q][0] < (uint32_t)n, kb = ps[q][1] < (uint32_t)n, kc = ps[q][2] < (uint32_t)n;
                    if (ka && kb && kc) continue;
                    if (ka && (!PARTS || aff[ps[q][0]] != 2)) aff[ps[q][0]] = 1;
                    if (kb && (!PARTS || aff[ps[q][1]] != 2)) aff[ps[q][1]] = 1;
                    if (kc && (!PARTS || aff[ps[q][2]] != 2)) aff[ps[q][2]] = 1;
                }
            }
            __syncthreads();
            DT_STAMP(52);
            // the unchanged stars' bookkeeping: row count, degree and hull flag as they were, room for the rows
            const uint32_t *info = a.seed_info + off;
            if (own) {
                // (a lane's own points again: their positions are in registers, their facts loaded together, and the room for
                // a wavefront's rows is one prefix sum and one LDS atomic instead of one atomic per point)
                uint32_t wds[kDtOwn];
#pragma unroll
                for (int r = 0; r < kDtOwn; ++r) wds[r] = ((own_k >> r) & 1u) ? info[s_begin + r * kWave + lane] : 0u;
                unsigned carried = 0u;
                int mine_rows = 0;
#pragma unroll
                for (int r = 0; r < kDtOwn; ++r) {
                    const bool k = ((own_k >> r) & 1u) && !aff[own_pos[r]];
                    carried |= k ? (1u << r) : 0u;
                    mine_rows += k ? (int)(wds[r] & 63u) : 0;
                }
                const int incl = dt_incl_scan(mine_rows);
                const int tot = __builtin_amdgcn_readlane(incl, kWave - 1);
                int base = 0;
                if (lane == 0 && tot) base = atomicAdd(&misc[DM_ARENA], tot);
                int at = __builtin_amdgcn_readfirstlane(base) + incl - mine_rows;
#pragma unroll
                for (int r = 0; r < kDtOwn; ++r) {
                    if (!((carried >> r) & 1u)) continue;
                    const int nown = (int)(wds[r] & 63u);
                    if (at + nown > L.arena_cap) { atomicOr(&misc[DM_FLAGS], (int)DT_WHY_ROWS); aff[own_pos[r]] = 1; continue; }
                    const int o = oid[own_pos[r]];
                    od[o] = (uint16_t)(wds[r] & 0xFFFFu);
                    astart[o] = (uint16_t)at;
                    at += nown;
                }
            } else
            for (int i0 = tid; i0 < n_in; i0 += BLOCK) {
                const uint32_t pos = __hip_atomic_load(inv + i0, __ATOMIC_RELAXED, kDtScope);
                if (pos >= (uint32_t)n || aff[pos]) continue;
                const uint32_t wd = info[i0];
                const int nown = (int)(wd & 63u);
                const int at = nown ? atomicAdd(&misc[DM_ARENA], nown) : 0;
                if (at + nown > L.arena_cap) { atomicOr(&misc[DM_FLAGS], (int)DT_WHY_ROWS); aff[pos] = 1; continue; }
                const int o = oid[pos];
                od[o] = (uint16_t)(wd & 0xFFFFu);
                astart[o] = (uint16_t)at;
            }
            __syncthreads();
            DT_STAMP(53);
        }
        // the seeds' corners into the hint caches (orientation from the points: the rows are in canonical, not in
        // counter-clockwise order); with carry: only where a star that will be walked reads them, and the unchanged
        // stars' rows straight into the arena
        for (int r = tid; r < ns; r += BLOCK) {
            const int ra = st[3 * r], rb = st[3 * r + 1], rc = st[3 * r + 2];
            if ((unsigned)ra >= (unsigned)n_in || (unsigned)rb >= (unsigned)n_in || (unsigned)rc >= (unsigned)n_in) continue;
            const uint32_t pa = __hip_atomic_load(inv + ra, __ATOMIC_RELAXED, kDtScope);
            uint32_t pb = __hip_atomic_load(inv + rb, __ATOMIC_RELAXED, kDtScope);
            uint32_t pc = __hip_atomic_load(inv + rc, __ATOMIC_RELAXED, kDtScope);
            if (pa >= (uint32_t)n || pb >= (uint32_t)n || pc >= (uint32_t)n) continue;          // a vertex that is not kept (all ones)
            bool ha = true, hb = true, hc = true;
            if (carry) {
                ha = aff[pa] == 1; hb = aff[pb] == 1; hc = aff[pc] == 1;
                if (aff[pa] == 0) {                                   // ra is the row's smallest id: its owner, and the ranks keep the order
                    const int idx = r - (int)(a.seed_info[off + ra] >> 16);
                    if (idx >= 0 && idx < (int)(od[oid[pa]] & 63u)) arena[astart[oid[pa]] + idx] = ((uint32_t)oid[pb] << 16) | (uint32_t)oid[pc];
                    else atomicOr(&misc[DM_FLAGS], (int)DT_WHY_ROWS);
                }
                if (!(ha || hb || hc)) continue;
            }
            const double2 A_ = S[pa], B_ = S[pb], C_ = S[pc];
            const double cr = (B_.x - A_.x) * (C_.y - A_.y) - (B_.y - A_.y) * (C_.x - A_.x);
            if (!(cr != 0.0)) continue;
            if (cr < 0.0) { const uint32_t t = pb; pb = pc; pc = t; const bool tb = hb; hb = hc; hc = tb; }   // (pa, pb, pc) counter-clockwise now
            if (ha) {
                hint_put(pa, pb, (pb << 16) | pc);
                start_put(pa, (pb << 16) | pc);
            }
            if (hb) {
                hint_put(pb, pc, (pc << 16) | pa);
                start_put(pb, (pc << 16) | pa);
            }
            if (hc) {
                hint_put(pc, pa, (pa << 16) | pb);
                start_put(pc, (pa << 16) | pb);
            }
        }
        __syncthreads();
        DT_STAMP(54);
    }
    if (order) {
        // The points are taken cell colour by cell colour — (x & 1, y & 1): all even/even cells first, and so on.  In the
        // sorted order 512 consecutive points are in flight at once, a band of five or six cell rows in which every point's
        // neighbours are being worked on at the same moment: the triangle one of them finds cannot be handed to the others
        // in time (1.67 searches per triangle).  Cells of one colour do not touch: most of a point's neighbours are either
        // done — their triangles wait in its hint cache — or not started.  +5.5 % at 2000 points, nothing at 600-1200; not in the
        // global-memory variant, whose point reads want the sorted order's locality (20 000 points: 26.6 k -> 16.7 k sets/s).
        // (and the points of one cell — neighbours, as a rule — in different rounds: class = 4 * min(position in the cell, 3) + colour)
        int *ccnt = reinterpret_cast<int *>(red);            // 16 counters (the reductions' scratch is free here)
        if (tid < 16) ccnt[tid] = 0;
        __syncthreads();
        for (int c = tid; c < ncell; c += BLOCK) {
            const int cx = c % G.gx, cy = c / G.gx, col = (cx & 1) | ((cy & 1) << 1);
            const int b = c ? (int)cs[c - 1] : 0, e = (int)cs[c];
            if (PARTS && part_of(c) != part) continue;     // (the stars of this part's strip of cells only)
            int k = 0;                                        // (with carry: only the points whose star is walked)
            for (int j = b; j < e; ++j) {
                if (carry && aff[j] != 1) continue;
                atomicAdd(&ccnt[4 * min(k, 3) + col], 1);
                ++k;
            }
        }
        __syncthreads();
        int mine = tid < 16 ? ccnt[tid] : 0, base = 0;
        __syncthreads();
        int total_ = 0;
        for (int q = 0; q < 16; ++q) { const int v = __shfl(mine, q); if (q < tid) base += v; total_ += v; }
        if (tid < 16) ccnt[tid] = base;
        if (tid == 0) misc[DM_VQ] = total_;
        __syncthreads();
        n_work = misc[DM_VQ];
        for (int c = tid; c < ncell; c += BLOCK) {
            const int cx = c % G.gx, cy = c / G.gx, col = (cx & 1) | ((cy & 1) << 1);
            const int b = c ? (int)cs[c - 1] : 0, e = (int)cs[c];
            if (PARTS && part_of(c) != part) continue;
            int k = 0;
            for (int j = b; j < e; ++j) {
                if (carry && aff[j] != 1) continue;
                const int at = atomicAdd(&ccnt[4 * min(k, 3) + col], 1);
                __hip_atomic_store(order + at, (uint32_t)j, __ATOMIC_RELAXED, kDtScope);
                ++k;
            }
        }
        __syncthreads();
    }
    // GLOBAL: the cell index was read (the scan) and then rewritten by other wavefronts' stores and atomics — this CU's L1
    // may hold the old lines
    if constexpr (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    DT_STAMP(2);
    int degenerate = 0;
    // ---- phase 1: one lane per point.  Lanes are persistent: a lane whose star is complete takes the next point (stars
    // have 3 to 10+ triangles: in rounds of 64 points the wavefront would wait for its largest star).  Every iteration of
    // the loop is ONE SCAN STEP per lane — up to five cell rows and kDtBudget candidates of the lane's current search:
    // the nearest-neighbour search that starts a star (the same minimisation in another mode), a completion within the
    // point's block, or the continuation of a WIDE search (the cell box of a circumcircle that leaves the block, or the
    // whole frame cut down to the wanted side of the edge, when the block holds nothing on that side: hull vertices and
    // their neighbours).  Lanes in different stages of different stars share every iteration; none waits for another.
    {
        // Points are taken from both ends of the sorted array towards its middle: the first and the last cell rows hold
        // the hull and the points next to it, whose stars need wide searches — the long tasks start first, the short
        // ones fill the tail.  (All boundary cells' points first — an order array built per frame — measured the same.)
        auto point_of = [&](int idx) {
            if (order) return (int)__hip_atomic_load(order + idx, __ATOMIC_RELAXED, kDtScope);
            return (idx & 1) ? n - 1 - (idx >> 1) : (idx >> 1);
        };
        int i = tid < n_work ? point_of(tid) : -1;       // (misc[DM_NEXT] starts at BLOCK)
        bool exhausted = tid >= n_work;
        int nn_level = 0;                           // nearest-neighbour search: 3x3 block, then 5x5, then the frame
        int mode = 0, oi = 0, q0 = -1, iq = -1, deg = 0, nown = 0, open = 0;
        int coop = 0;
        int wide = 0, y_next = 0, j_resume = 0;     // the current search: box rows from y_next on, the first of them from j_resume
        double sgn = 1.0;
        DtBox box = {0, 0, 0, -1};
        DtAcc A;
        A.reset();
        double2 p;
        p.x = 0.0; p.y = 0.0;
        uint32_t rows[kDtLaneRows];
#pragma unroll
        for (int k = 0; k < kDtLaneRows; ++k) rows[k] = 0xFFFFFFFFu;
#ifdef MVOSR_STAMPS
        int n_by_search = 0, n_by_hint = 0, n_iter = 0, n_busy = 0, steps_pt = 0;
#endif
        auto block_r = [&](double2 pt, int R) {
            const int cx = G.cellx(pt.x), cy = G.celly(pt.y);
            DtBox b_;
            b_.xa = max(cx - R, 0); b_.xb = min(cx + R, G.gx - 1); b_.ya = max(cy - R, 0); b_.yb = min(cy + R, G.gy - 1);
            return b_;
        };
        auto block_of = [&](double2 pt) { return block_r(pt, kDtR); };
        // w_: the box is final by construction (a circumcircle's cell box, the whole frame); c_: many candidates — the
        // wavefront scans it together (below) instead of the lane on its own
        auto begin_search = [&](const DtBox &b_, int w_, int c_ = -1) {
            box = b_; wide = w_; y_next = b_.ya; j_resume = 0; A.reset();
            coop = (kDtCoop<GLOBAL> && mode == 1 && (c_ < 0 ? w_ : c_)) ? 1 : 0;
        };
        // A point that already holds a triangle of its star — "after `from` comes `to`" (a seed, or a neighbour's find) — starts
        // there: both are Delaunay neighbours, the nearest-neighbour search (a scan step or two) is not needed.  The state is
        // that of a finished final search whose answer is `to`: the completion code takes it from there.
        auto start_from_hint = [&]() {
#if MVOSR_DT_HINT_START
            if constexpr (kDtHintsOn<GLOBAL>) {
                if (hints && i >= 0) {
                    const uint32_t h = start_get((uint32_t)i);
                    const int from = (int)(h >> 16), to = (int)(h & 0xFFFFu);
                    if (h != 0xFFFFFFFFu && from < n && to < n && from != to && from != i && to != i) {
                        mode = 1; q0 = from; iq = from; nn_level = 0;
                        box.xa = 0; box.xb = 0; box.ya = 0; box.yb = -1; wide = 1; coop = 0; y_next = 0; j_resume = 0;
                        A.reset(); A.b1 = to;
                    }
                }
            }
#endif
        };
        if (i >= 0) { p = S[i]; oi = oid[i]; begin_search(block_r(p, 1), 0); start_from_hint(); }
        // A WIDE search of a star (the cell box of a circumcircle that leaves the point's block, the frame's half beside
        // a hull edge: hundreds of candidates) is not walked by its lane — 32 candidates per scan step, with 63 lanes
        // waiting on it at the end of the frame: the hull vertices' stars took 36 steps each against 5.4 for an
        // interior point and set the workgroup's critical path — but scanned by the whole wavefront at once, one
        // such search after the other, each lane taking every 64th candidate of a cell row.
        auto serve_wide = [&]() -> bool {
            bool any = false;
            if constexpr (kDtCoop<GLOBAL>) {
            for (unsigned long long todo = __ballot(i >= 0 && mode == 1 && coop); todo; todo &= todo - 1ull) {
                const int src = (int)__ffsll((long long)todo) - 1;
                const int bi = __builtin_amdgcn_readlane(i, src), biq = __builtin_amdgcn_readlane(iq, src);
                const int bneg = __builtin_amdgcn_readlane(sgn < 0.0 ? 1 : 0, src);
                DtBox bb;
                bb.xa = __builtin_amdgcn_readlane(box.xa, src); bb.xb = __builtin_amdgcn_readlane(box.xb, src);
                bb.ya = __builtin_amdgcn_readlane(box.ya, src); bb.yb = __builtin_amdgcn_readlane(box.yb, src);
                DtEdge E2;
                E2.set(S[bi], S[biq], bi, biq, bneg ? -1.0 : 1.0);
                DtAcc A2;
                A2.reset();
                // The rows that can hold a point on the wanted side at all: sax (v - py) > say (X - px) for some X within the box's
                // columns is a half-line in v.  The whole-frame searches beside hull edges — two per hull vertex, a quarter of all
                // shared scans and, at ten row passes each, most of their row passes — look at the two or three rows along the
                // edge instead of all of them (a row more on either side: points on the line itself must be seen, they decline
                // the frame).
                if (E2.sax != 0.0) {
                    const double X0 = G.lo_u + ((double)bb.xa - 1e-6) * G.sx, X1 = G.lo_u + ((double)(bb.xb + 1) + 1e-6) * G.sx;
                    const double w_ = fmin(E2.say * (X0 - E2.px), E2.say * (X1 - E2.px));
                    const int yc = G.celly(E2.py + w_ / E2.sax);
                    if (E2.sax > 0.0) bb.ya = max(bb.ya, __builtin_amdgcn_readfirstlane(yc) - 1);
                    else bb.yb = min(bb.yb, __builtin_amdgcn_readfirstlane(yc) + 1);
                }
                // kDtServeRows cell rows at a time, 64 / kDtServeRows lanes each: a wide box is many rows of a dozen candidates,
                // and a row's range costs as much as its candidates (one row at a time with all 64 lanes: 648 k sets/s at
                // 2000 points; four: 741 k)
                constexpr int kLanesPerRow = kWave / kDtServeRows;
#ifdef MVOSR_STAMPS
                int my_trips = 0;
#endif
                for (int y0 = bb.ya; y0 <= bb.yb; y0 += kDtServeRows) {
                    const int y = y0 + lane / kLanesPerRow;
                    int j0 = 0, j1 = 0;
                    if (y <= bb.yb) dt_row_range(G, E2, y, bb.xa, bb.xb, j0, j1);
#ifdef MVOSR_STAMPS
                    int t_ = 0;
                    for (int j = j0 + (lane & (kLanesPerRow - 1)); j < j1; j += kLanesPerRow) { dt_step(A2, E2, j, S[j]); ++t_; }
                    my_trips += wave_max(t_);
#else
                    for (int j = j0 + (lane & (kLanesPerRow - 1)); j < j1; j += kLanesPerRow) dt_step(A2, E2, j, S[j]);
#endif
                }
#ifdef MVOSR_STAMPS
                if (lane == 0) { atomicAdd(&misc[41], my_trips); atomicAdd(&misc[42], 1); atomicAdd(&misc[43], (bb.yb - bb.ya + 1 + kDtServeRows - 1) / kDtServeRows); }
#endif
                const DtPick pk = dt_wave_pick(A2);
                if (pk.tie) degenerate |= DT_WHY_TIE;
                any = true;
                // (a collinear candidate ahead of p: the lane's completion hands its point to the group pass, which tells the kinds apart)
                if (lane == src) { A.reset(); A.b1 = pk.id; A.flag = pk.flag ? 1 : 0; y_next = box.yb + 1; j_resume = 0; coop = 0; }
            }
            }
            return any;
        };
#ifdef MVOSR_STAMPS
        unsigned long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();      // take a point / ranges / scan / completion
#define DT_SEC(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); t_sec[k] += now_ - t_last; t_last = now_; } while (0)
#else
#define DT_SEC(k) do {} while (0)
#endif
#ifdef MVOSR_DT_MARKS
#define DT_MARK(n) asm volatile("; DTMARK " #n ::: "memory")
#else
#define DT_MARK(n) do {} while (0)
#endif
        for (;;) {
            DT_MARK(loop_top);
            if (i < 0 && !exhausted) {
                const int idx = atomicAdd(&misc[DM_NEXT], 1);
                if (idx < n_work) {
                    i = point_of(idx); p = S[i]; oi = oid[i]; mode = 0; deg = 0; nown = 0; iq = -1; open = 0; sgn = 1.0; nn_level = 0;
#pragma unroll
                    for (int k = 0; k < kDtLaneRows; ++k) rows[k] = 0xFFFFFFFFu;
                    begin_search(block_r(p, 1), 0);
                    start_from_hint();
                } else exhausted = true;
            }
            if (__ballot(i >= 0) == 0ull) break;
            const bool act = i >= 0, m1 = mode == 1;
#ifdef MVOSR_STAMPS
            ++n_iter; n_busy += act ? 1 : 0; steps_pt += act ? 1 : 0;
#endif
            DT_SEC(0);
            DT_MARK(serve1);
            serve_wide();
            DT_SEC(1);
            DT_MARK(edge_ranges);
            DtEdge E;
            if (m1) E.set(p, S[max(iq, 0)], i, iq, sgn); else E.set_nn(p, i);
            // up to five rows of the search's box as ranges of the sorted array, walked as ONE loop (a loop per row would
            // run for the longest row of any lane, five times over)
            int budget = kDtBudget;
            {
                // the rows' ranges, one word each (start | end << 16: at most 32 000 points), the non-empty ones first: when a
                // lane's row ends, the walk takes the next word and shifts the rest down — 5 moves.  (Ten registers of starts and
                // ends, shifted in a loop that skipped the empty rows, were 17-40 moves in nested branches whenever ANY lane of the
                // wavefront ended a row: every trip, as good as.)
                uint32_t sg[kDtRows];
#pragma unroll
                for (int r = 0; r < kDtRows; ++r) {
                    const int y = y_next + r;
                    int b_, e_;
                    dt_row_range_lane(G, E, y, act && y <= box.yb, box.xa, box.xb, b_, e_);
                    if (r == 0) b_ = max(b_, j_resume);
                    sg[r] = b_ < e_ ? ((uint32_t)b_ | ((uint32_t)e_ << 16)) : 0u;
                }
#pragma unroll
                for (int pass = 0; pass < kDtRows - 1; ++pass)
#pragma unroll
                    for (int r = 0; r < kDtRows - 1 - pass; ++r) {
                        const bool z = sg[r] == 0u;
                        sg[r] = z ? sg[r + 1] : sg[r]; sg[r + 1] = z ? 0u : sg[r + 1];
                    }
                int j = (int)(sg[0] & 0xFFFFu), je = (int)(sg[0] >> 16);
                DT_SEC(2);
                DT_MARK(scan_loop);
#pragma unroll 2
                while (j < je && budget > 0) {
                    const double2 c = S[j];
                    const int jc = j;
                    ++j; --budget;
                    if (j >= je) {
                        const uint32_t nx = sg[1];
#pragma unroll
                        for (int r = 1; r < kDtRows - 1; ++r) sg[r] = sg[r + 1];
                        sg[kDtRows - 1] = 0u;
                        j = (int)(nx & 0xFFFFu); je = (int)(nx >> 16);
                    }
                    dt_step_lane(A, E, m1, jc, c);
                }
                DT_MARK(scan_done);
#ifdef MVOSR_STAMPS
                { const int t_ = wave_max(kDtBudget - budget); if (lane == 0) { atomicAdd(&misc[40], t_); atomicAdd(&misc[44], 1); }
                  atomicAdd(&misc[45], kDtBudget - budget);
                  if (lane == 0) atomicAdd(&misc[36 + min(t_ / 12, 3)], 1);
                  atomicAdd(&misc[32 + (kDtBudget - budget == 0 ? 0 : min((kDtBudget - budget + 11) / 12, 3))], 1); }
#endif
                // out of budget: the next step goes on from candidate j, in the cell row it lies in
                if (j < je) { y_next = G.celly(S[j].y); j_resume = j; }
                else { y_next += kDtRows; j_resume = 0; }
            }
            DT_SEC(3);
            DT_MARK(completion);
            // (twice: a wide search the first pass raises is scanned by the wavefront at once and completed in the same step)
            for (int rep = 0; rep < 2; ++rep) {
            if (i >= 0 && y_next > box.yb) {                         // the search is finished
            // ---- the search is complete.  A search that does not certify its answer is widened: the point's 3x3 block
            // (nearest neighbour only), its 5x5 block, an 11x11 block, then — final by construction — the cell box of the
            // answer's circumcircle, or the whole frame when there is no answer yet.
            const DtBox blk = block_of(p);
            const DtBox all = {0, G.gx - 1, 0, G.gy - 1};
            int state = 0;                                           // 1: finished, 2: hard
            int accept = -1;
            if (!m1) {
                DT_MARK(c_nn);
                // the nearest neighbour is a Delaunay neighbour — certified when its disc lies within what was searched
                q0 = A.b1;
#ifdef MVOSR_STAMPS
                atomicAdd(&misc[63], 1);            // a nearest-neighbour search completed (one widening level)
#endif
                if (q0 >= 0 && A.n1 == 0.0) degenerate |= DT_WHY_DUP;
                if (q0 >= 0 && (wide || dt_inside(dt_disc_box(G, p.x, p.y, A.n1), box))) {
                    iq = q0; mode = 1; nn_level = 0;
                    begin_search(blk, 0);
                } else if (wide) state = 2;                          // (a frame of one point: cannot happen, n >= 3)
                else {
                    ++nn_level;
                    if (nn_level == 1) begin_search(blk, 0);
                    else if (nn_level == 2) begin_search(block_r(p, kDtRWide), 0);
                    else begin_search(all, 1);
                }
            } else {
                DT_MARK(c_m1);
                const int ic = A.b1;
                // a collinear candidate ahead of p was seen: the point goes to the group pass (hard list), whose steps tell a candidate
                // on the segment from one beyond q (dt_step<true>) — nothing here, in the loop every point of every frame runs through
                if (A.flag) state = 2;
                else if (wide) {
                    if (ic >= 0) accept = ic;
                    else {
                        // a hull edge: the star is open.  Counter-clockwise done: clockwise from the first neighbour
                        open = 1;
                        if (sgn > 0.0) { sgn = -1.0; iq = q0; nn_level = 0; begin_search(blk, 0); }
                        else state = 1;
                    }
                } else if (ic >= 0) {
                    const DtBox cb = dt_circle_box(G, p.x, p.y, S[iq], S[ic]);
                    if (dt_inside(cb, box)) accept = ic;
                    // the circumcircle leaves what was searched: its cell box decides.  A small box — the block and a row or
                    // a column more, as a rule — is one more scan step of this lane, walked under the other lanes' steps; only a
                    // large one (a sliver's circle) is worth stopping the wavefront for
                    else begin_search(cb, 1, (cb.xb - cb.xa + 1) * (cb.yb - cb.ya + 1) > kDtCoopCells ? 1 : 0);
                }
                // nothing on that side within the block: the frame's half beside the edge, scanned by the wavefront (an
                // intermediate 17 x 17 block first was measured: equal at 2000 points, 20 % slower at 300-600); the lane
                // on its own (global-memory variant) looks at 17 x 17 cells first
                else if (!kDtCoop<GLOBAL> && nn_level == 0) { nn_level = 1; begin_search(block_r(p, kDtRWide), 0); }
                else begin_search(all, 1);
                DT_SEC(4);
                DT_MARK(c_accept);
                if (accept >= 0) {
                    if (A.tie && dt_confirm_tie(S, cs, G.gx, box.xa, box.xb, box.ya, box.yb, E.px, E.py, E.ax, E.ay, E.sgn, E.a2col, E.i, E.iq, A.b1, A.n1, A.c1))
                        degenerate |= DT_WHY_TIE;
                    if constexpr (kDtHintsOn<GLOBAL>) {
                        if (hints) {
                            // counter-clockwise walk: (p, iq, accept) is the triangle; clockwise: (p, accept, iq)
                            const uint32_t a_ = (uint32_t)(sgn > 0.0 ? iq : accept), c_ = (uint32_t)(sgn > 0.0 ? accept : iq);
                            hint_put(a_, c_, (c_ << 16) | (uint32_t)i);                  // in a's star: after c comes p
                            hint_put(c_, (uint32_t)i, ((uint32_t)i << 16) | a_);         // in c's star: after p comes a
#if MVOSR_DT_HINT_START
                            start_put(a_, (c_ << 16) | (uint32_t)i);
                            start_put(c_, ((uint32_t)i << 16) | a_);
#endif
                        }
                    }
                    int chain = 0;
#ifdef MVOSR_STAMPS
                    ++n_by_search;
                    if (hints && sgn > 0.0) {       // did the hint arrive while the search ran?
                        const uint32_t h_ = hint_get((uint32_t)i, (uint32_t)iq);
                        if ((h_ >> 16) == (uint32_t)iq) atomicAdd(&misc[61], 1);
                        else if (h_ != 0xFFFFFFFFu) atomicAdd(&misc[62], 1);      // the slot holds another neighbour's hint
                    }
#endif
                    DT_MARK(c_chain);
                    // The chain as a loop of the whole (sub-)wavefront: it runs while ANY lane is still taking hinted triangles and
                    // every lane makes every round, a lane that is out making it without effect — its state is updated by
                    // selects, and what it inserts into its sorted rows is all ones, which changes nothing.  (As a loop that
                    // each lane left on its own, the rows of the lanes that had left were kept in a second set of registers and
                    // copied back every round: 12 moves, and 18 selects for the search the leaving lanes begin.)
                    bool search_on = false, in_chain = true;
                    for (;;) {
                        deg += in_chain ? 1 : 0;
                        if (in_chain && deg > kDtLaneDeg) state = 2;
                        const int oq = oid[iq], oc = oid[max(accept, 0)];
                        const bool own = in_chain && oi < oq && oi < oc;
                        if (own && nown == kDtLaneRows) state = 2;
                        const bool put = own && nown < kDtLaneRows;
                        uint32_t key = put ? (((uint32_t)min(oq, oc) << 16) | (uint32_t)max(oq, oc)) : 0xFFFFFFFFu;
#pragma unroll
                        for (int k = 0; k < kDtLaneRows; ++k) { const uint32_t hi = max(key, rows[k]); rows[k] = min(key, rows[k]); key = hi; }
                        nown += put ? 1 : 0;
                        iq = in_chain ? accept : iq;
                        bool go = in_chain && state == 0;
                        if (go && sgn > 0.0 && iq == q0) { state = 1; go = false; }      // closed
                        // the next edge of the star: already known from a neighbour's star?
                        int nxt = -1;
                        if constexpr (kDtHintsOn<GLOBAL>) {
                            if (go && hints && sgn > 0.0 && chain < kDtHintChain) {
                                const uint32_t h = hint_get((uint32_t)i, (uint32_t)iq);
                                if ((h >> 16) == (uint32_t)iq && (int)(h & 0xFFFFu) < n) nxt = (int)(h & 0xFFFFu);
                            }
                        }
                        search_on = search_on || (go && nxt < 0);
                        in_chain = go && nxt >= 0;
                        accept = in_chain ? nxt : accept;
                        chain += in_chain ? 1 : 0;
#ifdef MVOSR_STAMPS
                        n_by_hint += in_chain ? 1 : 0;
#endif
                        if (__ballot(in_chain) == 0ull) break;
                    }
                    nn_level = 0;
                    if (search_on) begin_search(blk, 0);               // (once, after the chain: inside the loop every lane paid for it in every round)
                }
            }
#ifdef MVOSR_STAMPS
            if (state != 0) {
                atomicMax(&misc[52], steps_pt);
                atomicAdd(&misc[53 + (steps_pt <= 6 ? 0 : steps_pt <= 10 ? 1 : steps_pt <= 16 ? 2 : steps_pt <= 28 ? 3 : 4)], 1);
                if (open) atomicAdd(&misc[58], steps_pt); else atomicAdd(&misc[59], steps_pt);
                if (open) atomicAdd(&misc[60], 1);
                steps_pt = 0;
            }
#endif
            DT_SEC(5);
            DT_MARK(c_state);
            if (state == 1) {
                const int at = nown ? atomicAdd(&misc[DM_ARENA], nown) : 0;
                if (at + nown > L.arena_cap) degenerate |= DT_WHY_ROWS;
                else {
#pragma unroll
                    for (int k = 0; k < kDtLaneRows; ++k) if (k < nown) arena[at + k] = rows[k];
                    od[oi] = (uint16_t)(nown | (deg << 6) | (open << 15));
                    astart[oi] = (uint16_t)at;
                }
                i = -1;
            } else if (state == 2) {
                // more rows / a larger star than a lane keeps: the group pass below takes the point
                od[oi] = 0x4000;
                const int pos = atomicAdd(&misc[DM_NHARD], 1);
                if (pos < kDtHardCap) hard[pos] = (uint16_t)i; else degenerate |= DT_WHY_HARD;
                i = -1;
            }
            }
            DT_SEC(6);
            DT_MARK(serve2);
            if (rep == 1 || !serve_wide()) break;
            }
            DT_MARK(loop_end);
            DT_SEC(7);
        }
#ifdef MVOSR_STAMPS
        if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&misc[24 + k], (int)(t_sec[k] >> 4));   // (sixteenths of a cycle count: 32-bit sums)
        atomicAdd(&misc[48], n_by_search); atomicAdd(&misc[49], n_by_hint); atomicAdd(&misc[50], n_iter); atomicAdd(&misc[51], n_busy);
#endif
    }
    __syncthreads();
    DT_STAMP(3);
#ifdef MVOSR_STAMPS
    DT_NOTE(10, misc[48]); DT_NOTE(11, misc[49]); DT_NOTE(12, misc[50]); DT_NOTE(13, misc[51]);
    DT_NOTE(26, misc[61]); DT_NOTE(27, misc[62]); DT_NOTE(28, misc[63]);
    if (tid == 0 && a.stamps) for (int k = 0; k < 16; ++k) a.stamps[64 * (PARTS ? (int64_t)blockIdx.x : f) + 32 + k] = (unsigned long long)misc[24 + k];
    DT_NOTE(29, misc[40]); DT_NOTE(30, misc[41]); DT_NOTE(31, misc[42]); DT_NOTE(7, misc[43]); DT_NOTE(8, misc[44]); DT_NOTE(14, misc[45]);
    if (tid == 0 && a.stamps) for (int k = 52; k < 61; ++k) a.stamps[64 * (PARTS ? (int64_t)blockIdx.x : f) + 16 + (k - 52)] = (unsigned long long)misc[k];
#endif

    // The two passes below work in GROUPS of 16 lanes (a DPP row): a completion has a few dozen candidates at most, so
    // four of them share a wavefront.  Lanes of a group stay together; groups diverge freely.
    const int gl = lane & (kDtGroup - 1), grp = tid / kDtGroup;
    constexpr int kGroups = BLOCK / kDtGroup;
    __syncthreads();
    DT_STAMP(4);

    // ---- phase 2: hard points, one group each
    {
        const int nh = min(misc[DM_NHARD], kDtHardCap);
        uint32_t *grows = reinterpret_cast<uint32_t *>(small + L.wrows) + grp * kDtWaveRows;
        const DtBox all = {0, G.gx - 1, 0, G.gy - 1};
        for (int h = grp; h < nh; h += kGroups) {
            const int i = hard[h];
            const double2 p = S[i];
            const int oi = oid[i];
            const int cx = G.cellx(p.x), cy = G.celly(p.y);
            DtBox blk;
            blk.xa = max(cx - kDtR, 0); blk.xb = min(cx + kDtR, G.gx - 1); blk.ya = max(cy - kDtR, 0); blk.yb = min(cy + kDtR, G.gy - 1);
            // nearest neighbour (block, then everything)
            int q0 = -1;
            double dmin = INFINITY;
            for (int pass = 0; pass < 2; ++pass) {
                const DtBox &B = pass ? all : blk;
                double bd = INFINITY;
                int bq = -1;
                for (int y = B.ya; y <= B.yb; ++y) {
                    const int j1 = G.row_end(y, B.xb);
                    for (int j = G.row_begin(y, B.xa) + gl; j < j1; j += kDtGroup) {
                        const double2 c = S[j];
                        const double dx = c.x - p.x, dy = c.y - p.y, d2 = dx * dx + dy * dy;
                        if (j != i && d2 < bd) { bd = d2; bq = j; }
                    }
                }
                dmin = dt_group_min(bd);
                const unsigned who = dt_group_ballot(bq >= 0 && bd == dmin);
                q0 = dt_group_shfl(bq, who ? (int)__ffs((int)who) - 1 : 0);
                if (!who) q0 = -1;
                if (q0 >= 0 && dt_inside(dt_disc_box(G, p.x, p.y, dmin), B)) break;
            }
            if (q0 < 0) { degenerate |= DT_WHY_EULER; continue; }
            if (dmin == 0.0) { degenerate |= DT_WHY_DUP; continue; }
            int nrows = 0, deg = 0, open = 0, bad = 0;
            for (int dir = 0; dir < 2 && !bad; ++dir) {
                const double sgn = dir ? -1.0 : 1.0;
                int iq = q0;
                while (true) {
                    const double2 q = S[iq];
                    DtEdge E;
                    E.set(p, q, i, iq, sgn);
                    DtAcc A;
                    A.reset();
                    dt_scan_box<kDtGroup, true>(A, S, G, blk, E, q);
                    DtPick pk = dt_group_pick(A);
                    if (pk.id < 0 || !dt_inside(dt_circle_box(G, p.x, p.y, q, S[max(pk.id, 0)]), blk)) {
                        // nothing on that side within the block, or a circumcircle that leaves it: search the circle's
                        // cell box, or the whole frame (row by row, each row cut down to the wanted side of the edge)
                        const DtBox B = pk.id < 0 ? all : dt_circle_box(G, p.x, p.y, q, S[pk.id]);
                        const int seg_seen = pk.flag & 1;           // (a candidate on the segment p..q stays one in whatever box)
                        A.reset();
                        dt_scan_box<kDtGroup, true>(A, S, G, B, E, q);
                        pk = dt_group_pick(A);
                        pk.flag |= seg_seen;
                    }
                    if (dt_col_declines(pk.flag, pk.id)) degenerate |= DT_WHY_COLLINEAR;
                    if (pk.id < 0) { open = 1; break; }                    // a hull edge
                    if (pk.tie) degenerate |= DT_WHY_TIE;
                    if (++deg > kDtWaveDeg) { degenerate |= DT_WHY_DEGREE; bad = 1; break; }
                    const int oq = oid[iq], oc = oid[pk.id];
                    if (oi < oq && oi < oc) {
                        if (nrows == kDtWaveRows) { degenerate |= DT_WHY_ROWS; bad = 1; break; }
                        if (gl == 0) grows[nrows] = ((uint32_t)min(oq, oc) << 16) | (uint32_t)max(oq, oc);
                        ++nrows;
                    }
                    iq = pk.id;
                    if (iq == q0) break;                                   // closed
                }
                if (!open) break;
            }
            if (bad) continue;
            // the point's rows, sorted (rank by counting), to the arena
            int base = 0;
            if (gl == 0 && nrows > 0) base = atomicAdd(&misc[DM_ARENA], nrows);
            base = dt_group_shfl(base, 0);
            if (base + nrows > L.arena_cap) { degenerate |= DT_WHY_ROWS; continue; }
            for (int k0 = gl; k0 < nrows; k0 += kDtGroup) {
                const uint32_t key = grows[k0];
                int rank = 0;
                for (int k = 0; k < nrows; ++k) rank += grows[k] < key ? 1 : 0;
                arena[base + rank] = key;
            }
            if (gl == 0) {
                od[oi] = (uint16_t)(nrows | (deg << 6) | (open << 15));
                astart[oi] = (uint16_t)base;
            }
        }
    }
    if (degenerate) atomicOr(&misc[DM_FLAGS], degenerate);
    __syncthreads();
    DT_STAMP(5);
    DT_NOTE(9, misc[DM_NHARD]);

    // PARTS: this part's stars — the `od` words and row starts of its band's points, its arena — to the frame's meeting place;
    // the last part to arrive goes on and writes the rows from there
    const uint16_t *OD = od;
    const uint32_t *AR = arena;
    const uint32_t *ST32 = nullptr;
    int part_flags = 0;
    if constexpr (PARTS) {
        const DtPartsPlan PP = dt_parts_plan(a.max_pts, a.parts & 0xFF);
        char *pg = a.pg + (size_t)f * PP.total;
        uint32_t *g_head = a.ph + 16 * f;
        uint16_t *g_od = reinterpret_cast<uint16_t *>(pg + PP.od);
        uint32_t *g_start = reinterpret_cast<uint32_t *>(pg + PP.start);
        uint32_t *g_arena = reinterpret_cast<uint32_t *>(pg + PP.arena);
        const uint32_t abase = (uint32_t)part * (uint32_t)L.arena_cap;
        const int used = min(misc[DM_ARENA], L.arena_cap);
        for (int k = tid; k < used; k += BLOCK) g_arena[abase + k] = arena[k];
        for (int c = tid; c < ncell; c += BLOCK) {
            const int b = c ? (int)cs[c - 1] : 0, e = (int)cs[c];
            if (part_of(c) != part) continue;
            for (int j = b; j < e; ++j) { const int o = oid[j]; g_od[o] = od[o]; g_start[o] = abase + astart[o]; }
        }
        if (tid == 0 && misc[DM_FLAGS]) atomicOr(&g_head[1], (uint32_t)misc[DM_FLAGS]);
        __threadfence();
        __syncthreads();
        if (tid == 0) misc[DM_TICKET] = (int)atomicAdd(&g_head[0], 1u);
        __syncthreads();
        if (misc[DM_TICKET] != a.parts - 1) return;
        __threadfence();
        // (one thread reads the parts' flags, every wavefront takes them from LDS, and only then are they reset for the next launch:
        // a wavefront that loaded them itself could come after thread 0's store, see zero, and go on with a frame the others decline)
        if (tid == 0) misc[DM_PFLAGS] = (int)__hip_atomic_load(&g_head[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        part_flags = misc[DM_PFLAGS];
        __syncthreads();
        if (tid == 0) { g_head[0] = 0u; g_head[1] = 0u; }            // (every part of this launch has been here)
        OD = g_od; AR = g_arena; ST32 = g_start;
    }
    auto start_of = [&](int o) -> int { if constexpr (PARTS) return (int)ST32[o]; else return (int)astart[o]; };

    // ---- rows in point order: block prefix over the points' row counts; Euler's relation
    {
        const int pper = (n + BLOCK - 1) / BLOCK;
        const int o0 = tid * pper, o1 = min(n, o0 + pper);
        int mine = 0, sdeg = 0, hull = 0;
        for (int o = o0; o < o1; ++o) { const int d = OD[o]; mine += d & 63; sdeg += (d >> 6) & 63; hull += (d >> 15) & 1; }
        const int incl = dt_incl_scan(mine);
        const int wdeg = wave_sum(sdeg), whull = wave_sum(hull);
        if (lane == kWave - 1) wsl[DW_SUM + w] = incl;
        if (lane == 0) { wsl[DW_SUM2 + w] = wdeg; wsl[DW_SUM3 + w] = whull; }
        __syncthreads();
        int base = 0, total = 0, tdeg = 0, thull = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) {
            const int c = wsl[DW_SUM + i];
            if (i < w) base += c;
            total += c; tdeg += wsl[DW_SUM2 + i]; thull += wsl[DW_SUM3 + i];
        }
        int why = PARTS ? part_flags : misc[DM_FLAGS];
        if (total != 2 * n - 2 - thull || tdeg != 3 * total) why |= DT_WHY_EULER;
        if (why) { decline(why, n); return; }
        int32_t *rows = a.tri + 3 * a.tri_off[f];
        int at = base + incl - mine;
        // (a thread's points eight at a time: their starts, then their rows four at a time, are in flight together — with the
        // arena in global memory every one of them is a load the next step waits for)
        for (int ob = o0; ob < o1; ob += 8) {
            int ks[8], sts[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int o = min(ob + q, o1 - 1);
                ks[q] = ob + q < o1 ? (OD[o] & 63) : 0;
                sts[q] = start_of(o);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int o = ob + q;
                if (o >= o1) break;
                if (a.info_out && !gk) a.info_out[off + o] = (uint32_t)OD[o] | ((uint32_t)at << 16);    // (no keep mask: rank o = position o)
                const uint32_t *src = AR + sts[q];
                for (int j = 0; j < ks[q]; j += 4) {
                    uint32_t key[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) key[c] = src[min(j + c, ks[q] - 1)];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (j + c >= ks[q]) break;
                        rows[3 * at] = o; rows[3 * at + 1] = (int32_t)(key[c] >> 16); rows[3 * at + 2] = (int32_t)(key[c] & 0xFFFFu);
                        ++at;
                    }
                }
            }
        }
        if (tid == 0) { a.tri_cnt[f] = total; a.status[f] = MVOSR_DT_OK; if (a.n_used) a.n_used[f] = n; }
        DT_STAMP(6);
    }
}

}  // namespace mvosr

using namespace mvosr;

#ifdef MVOSR_STAMPS
extern "C" void mvosr_debug_dt_stamps(void *dptr) { g_dt_stamps = reinterpret_cast<unsigned long long *>(dptr); }
#endif

// (MVOSR_DT_PARTS=0: the per-frame launch as one workgroup, for A/B runs and the tests that compare the two)
static int dt_parts_env() {          // 0: off, n: that many parts, -1: by the frame's size
    const char *e = getenv("MVOSR_DT_PARTS");
    if (!e) return -1;
    const int v = atoi(e);
    return v < 0 ? -1 : (v > kDtPartsMax ? kDtPartsMax : v);
}

static int dt_lds_points() {
    int lo = 3, hi = 65535;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (dt_plan(mid).total <= 160u * 1024u) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The launcher's instantiation for a batch (of 512 frames and more) whose largest frame has max_pts points: wavefronts per frame,
// whether the rows' arena lives in global memory, and how many such frames share a CU.
static void dt_ladder(int max_pts, int &waves, bool &arena_out, int &per_cu) {
    waves = kDtWaves; arena_out = false;
    {
        // (LDS is handed out in granules of 1 280 bytes: 128 per CU)
        auto fits = [](uint32_t frames, uint32_t bytes) { return frames * ((bytes + 1279u) / 1280u) <= 128u; };
        if (fits(8u, dt_plan(max_pts, false, 2).total)) waves = 2;
        else if (fits(3u, dt_plan(max_pts, false, 4).total)) waves = 4;
        // up to ~2 100 points three frames still share a CU when the rows' arena and the points' starts in it move to global
        // memory (written once per star, read once at the end): 52 KB of LDS per 2000-point frame instead of 74.  Three
        // four-wavefront frames against two eight-wavefront ones: 12 wavefronts per CU with 170 registers each (no spills), a
        // quarter fewer points of a frame in flight at once (more triangles arrive as hints), and a third frame's work
        // under the dependent steps of the second triangulation, which is bound by its longest star
        else if (kDtArenaOut && fits(3u, dt_plan(max_pts, false, 4, true).total)) { waves = 4; arena_out = true; }
        // beyond that, two eight-wavefront frames per CU as long as they fit — with the arena in global memory up to ~3 100
        // points instead of ~2 350 (one frame per CU is half the wavefronts)
        else if (kDtArenaOut && !fits(2u, dt_plan(max_pts, false, 8).total) && fits(2u, dt_plan(max_pts, false, 8, true).total)) arena_out = true;
        // (where three four-wavefront frames fit a CU either way — 1 100 to 1 480 points — the arena-out build, which is compiled
        // for three wavefronts per SIMD: 154 registers, no spills: +3-5 %)
        if (kDtArenaOut && waves == 4 && !arena_out && !fits(4u, dt_plan(max_pts, false, 4).total)) arena_out = true;
    }
    auto fits_n = [](uint32_t frames, uint32_t bytes) { return frames * ((bytes + 1279u) / 1280u) <= 128u; };
    const uint32_t bytes = dt_plan(max_pts, false, waves, arena_out).total;
    per_cu = waves == 2 ? 8 : 1;
    // (registers: the four-wavefront builds run 16 wavefronts per CU — the arena-out one, with 154 registers, 12 —, the eight-wavefront ones 16)
    if (waves != 2) for (int k = waves == 4 ? (arena_out ? 3 : 4) : 2; k >= 1; --k) if (fits_n((uint32_t)k, bytes)) { per_cu = k; break; }
}

extern "C" int mvosr_delaunay_frames_per_cu(int max_pts) {
    if (max_pts < 3) max_pts = 3;
    if (max_pts > dt_lds_points()) return 1;
    int waves, per_cu; bool arena_out;
    dt_ladder(max_pts, waves, arena_out, per_cu);
    return per_cu;
}

extern "C" int mvosr_delaunay_max_points(void) { return kDtMaxPointsGlobal; }
extern "C" int mvosr_delaunay_lds_points(void) { return dt_lds_points(); }

extern "C" int mvosr_delaunay_batch(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                    const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                                    int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status) {
    return mvosr_delaunay_batch_seeded(ctx, n_frames, pts_off, pts_cnt, u, v, keep, max_pts, tri_off, tri, tri_cnt, n_used, status,
                                       nullptr, nullptr, nullptr);
}

extern "C" int mvosr_delaunay_batch_seeded(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                           const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                                           int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status,
                                           const int64_t *seed_off, const int32_t *seed_tri, const int32_t *seed_cnt) {
    return mvosr_delaunay_batch_ex(ctx, n_frames, pts_off, pts_cnt, u, v, keep, max_pts, tri_off, tri, tri_cnt, n_used, status,
                                   seed_off, seed_tri, seed_cnt, nullptr, nullptr);
}

extern "C" int mvosr_delaunay_batch_ex(mvosr_ctx *ctx, int64_t n_frames, const int64_t *pts_off, const int32_t *pts_cnt,
                                       const double *u, const double *v, const int32_t *keep, int max_pts, const int64_t *tri_off,
                                       int32_t *tri, int32_t *tri_cnt, int32_t *n_used, int32_t *status,
                                       const int64_t *seed_off, const int32_t *seed_tri, const int32_t *seed_cnt,
                                       const uint32_t *seed_info, uint32_t *info_out) {
    if (seed_info && !seed_tri) return set_error(MVOSR_ERR_ARG, "delaunay_batch_ex: seed_info without seeds");
    if (info_out && keep) return set_error(MVOSR_ERR_ARG, "delaunay_batch_ex: info_out describes a triangulation of ALL the points (no keep mask)");
    if (!ctx || !pts_off || !pts_cnt || !u || !v || !tri_off || !tri || !tri_cnt || !status)
        return set_error(MVOSR_ERR_ARG, "delaunay_batch: null argument");
    if ((seed_off || seed_tri || seed_cnt) && !(seed_off && seed_tri && seed_cnt))
        return set_error(MVOSR_ERR_ARG, "delaunay_batch_seeded: seed_off, seed_tri and seed_cnt go together");
    if (seed_tri == tri) return set_error(MVOSR_ERR_ARG, "delaunay_batch_seeded: the seeds' rows and the output rows are the same array");
    if (max_pts < 0) return set_error(MVOSR_ERR_ARG, "delaunay_batch: max_pts < 0");
    if (n_frames <= 0) return MVOSR_OK;
    int rc = ctx_activate(ctx);
    if (rc) return rc;
    if (max_pts < 3) max_pts = 3;
    if (max_pts > kDtMaxPointsGlobal)
        return set_error(MVOSR_ERR_TOO_LARGE, "delaunay_batch: %d points per frame (limit: %d)", max_pts, kDtMaxPointsGlobal);
    const bool global = max_pts > dt_lds_points();
    const DtPlan L = dt_plan(max_pts, global);
    DtArgs a;
    a.n_frames = n_frames; a.pts_off = pts_off; a.pts_cnt = pts_cnt; a.u = u; a.v = v; a.keep = keep; a.tri_off = tri_off; a.tri = tri;
    a.tri_cnt = tri_cnt; a.n_used = n_used; a.status = status; a.max_pts = max_pts; a.ws = nullptr; a.aws = nullptr; a.hints = nullptr;
    a.seed_off = seed_off; a.seed_tri = seed_tri; a.seed_cnt = seed_cnt;
    a.seed_info = seed_info; a.info_out = info_out; a.parts = 1; a.pg = nullptr; a.ph = nullptr;
#ifdef MVOSR_STAMPS
    a.stamps = g_dt_stamps;
#endif
    size_t lds = L.total;
    if (global) {
        // frames beyond the LDS capacity: the big arrays in the context's workspace, one slice per frame
        void *ws = nullptr;
        const size_t big_bytes = ((size_t)n_frames * L.big + 255) & ~(size_t)255;
        const size_t hint_bytes = kDtHintsOn<true> ? (size_t)n_frames * (size_t)(kDtHintK + 3) * (size_t)((max_pts + 7) & ~7) * sizeof(uint32_t) : 0;
        if ((rc = ctx_workspace_bytes(ctx, big_bytes + hint_bytes, &ws))) return rc;
        a.ws = reinterpret_cast<char *>(ws);
        if (hint_bytes) a.hints = reinterpret_cast<uint32_t *>(a.ws + big_bytes);
        lds = L.total - L.big;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(delaunay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_hip_error("hipFuncSetAttribute(delaunay_kernel)", e);
        hipLaunchKernelGGL(delaunay_kernel<true>, dim3((unsigned)n_frames), dim3(kDtBlock), lds, ctx_stream(ctx), a);
        return check_launch("delaunay_kernel (global-memory variant)");
    }
    // Small frames: fewer wavefronts per frame and more frames per CU.  A 600-point set gives 512 lanes little more than one
    // point each and its length is that of its longest stars (the hull's); with 256 lanes the same critical path carries
    // twice the points per lane.  Four wavefronts while THREE frames' arrays fit a CU's LDS (12 wavefronts per CU instead of
    // 16: equal at 1500 points, +4 % at 1200, +20-50 % below 1000 — and a ragged batch is sized by its largest frame), two
    // while eight fit (up to ~500 points: +20-50 % over four).
    // A launch that cannot fill the GPU with eight-wavefront frames (a per-frame call: ONE frame) keeps all eight: there the
    // lanes per frame are what shortens the call (900 points: 0.96 against 1.26 ms per frame call).
    int waves = kDtWaves, per_cu = 1;
    bool arena_out = false;
    // A launch of a few frames (the per-frame call of /root/reference/src/main.py:110-113: ONE) leaves 255 CUs idle and lasts as long as
    // one frame on one CU (276 us at 2000 points, 95 % of it the stars): several workgroups per frame instead, each with the
    // whole frame in its LDS and a strip of the cells to build the stars of — four wavefronts, at most a star per lane —, the last one to
    // finish writing the rows (the PARTS instantiation)
    if (kDtHintK > 0 && kDtColour && n_frames <= kDtPartsMaxFrames && max_pts >= 2 * kDtPartsPoints && dt_parts_env() &&
        dt_plan(max_pts, false, 4).total <= 160u * 1024u) {
        const int parts = dt_parts_env() > 0 ? dt_parts_env() : min(kDtPartsMax, (max_pts + kDtPartsPoints - 1) / kDtPartsPoints);
        const DtPlan LQ = dt_plan(max_pts, false, 4);
        const size_t plds = (size_t)LQ.total;
        const DtPartsPlan PP = dt_parts_plan(max_pts, parts);
        const size_t hint_bytes = ((size_t)n_frames * parts * (size_t)(kDtHintK + 3) * (size_t)((max_pts + 7) & ~7) * sizeof(uint32_t) + 255) & ~(size_t)255;
        void *ws = nullptr;
        if ((rc = ctx_workspace_bytes(ctx, hint_bytes + (size_t)n_frames * PP.total, &ws))) return rc;
        a.hints = reinterpret_cast<uint32_t *>(ws);
        a.pg = reinterpret_cast<char *>(ws) + hint_bytes;
        a.parts = parts;
        if (!ctx->dt_parts_head) {
            hipError_t eh = hipMalloc(reinterpret_cast<void **>(&ctx->dt_parts_head), 16 * sizeof(unsigned int) * kDtPartsMaxFrames);
            if (eh == hipSuccess) eh = hipMemset(ctx->dt_parts_head, 0, 16 * sizeof(unsigned int) * kDtPartsMaxFrames);
            if (eh != hipSuccess) return set_hip_error("delaunay_batch: the parts' counters", eh);
        }
        a.ph = ctx->dt_parts_head;
        const void *kp = reinterpret_cast<const void *>(delaunay_kernel<false, 4, false, true>);
        hipError_t e2 = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
        if (e2 != hipSuccess) return set_hip_error("hipFuncSetAttribute(delaunay_kernel)", e2);
        hipLaunchKernelGGL((delaunay_kernel<false, 4, false, true>), dim3((unsigned)(n_frames * parts)), dim3(4 * kWave), plds, ctx_stream(ctx), a);
        return check_launch("delaunay_kernel (parts)");
    }
    if (kDtSmallLadder && n_frames >= kDtLadderMinFrames) dt_ladder(max_pts, waves, arena_out, per_cu);
    // A launch of a few frames (the per-frame call of /root/reference/src/main.py:110-113: ONE) leaves most CUs idle and its length is
    // one frame's: sixteen wavefronts per frame — four per SIMD instead of two: the lanes' dependent steps overlap, and a
    // lane walks two stars instead of four
    else if (kDtWide16 && n_frames <= kDtWide16MaxFrames && max_pts >= 256 && dt_plan(max_pts, false, 16).total <= 160u * 1024u) waves = 16;   // (its phase-2 rows are 6 KB more: the largest LDS frames keep eight)
    const DtPlan LP = dt_plan(max_pts, false, waves, arena_out);
    lds = LP.total;
    const void *kfn = waves == 16 ? reinterpret_cast<const void *>(delaunay_kernel<false, 16>)
                    : (arena_out && waves == 4) ? reinterpret_cast<const void *>(delaunay_kernel<false, 4, true>)
                    : arena_out ? reinterpret_cast<const void *>(delaunay_kernel<false, 8, true>)
                    : waves == 2 ? reinterpret_cast<const void *>(delaunay_kernel<false, 2>)
                    : waves == 4 ? reinterpret_cast<const void *>(delaunay_kernel<false, 4>) : reinterpret_cast<const void *>(delaunay_kernel<false>);
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_hip_error("hipFuncSetAttribute(delaunay_kernel)", e);
    {
        // the stars' hint caches (see kDtHintK): 4 * (kDtHintK + 3) bytes per point; behind them the arena slices
        void *ws = nullptr;
        const size_t hint_bytes = ((size_t)n_frames * (size_t)(kDtHintK + 3) * (size_t)((max_pts + 7) & ~7) * sizeof(uint32_t) + 255) & ~(size_t)255;
        const size_t out_bytes = arena_out ? (size_t)n_frames * LP.out_bytes : 0;
        if (hint_bytes + out_bytes) {
            if ((rc = ctx_workspace_bytes(ctx, hint_bytes + out_bytes, &ws))) return rc;
            if (kDtHintK > 0) a.hints = reinterpret_cast<uint32_t *>(ws);   // (every workgroup empties its own frame's caches: no memset of the whole block)
            a.aws = reinterpret_cast<char *>(ws) + hint_bytes;
        }
    }
    if (waves == 16) hipLaunchKernelGGL((delaunay_kernel<false, 16>), dim3((unsigned)n_frames), dim3(16 * kWave), lds, ctx_stream(ctx), a);
    else if (arena_out && waves == 4) hipLaunchKernelGGL((delaunay_kernel<false, 4, true>), dim3((unsigned)n_frames), dim3(4 * kWave), lds, ctx_stream(ctx), a);
    else if (arena_out) hipLaunchKernelGGL((delaunay_kernel<false, 8, true>), dim3((unsigned)n_frames), dim3(kDtBlock), lds, ctx_stream(ctx), a);
    else if (waves == 2) hipLaunchKernelGGL((delaunay_kernel<false, 2>), dim3((unsigned)n_frames), dim3(2 * kWave), lds, ctx_stream(ctx), a);
    else if (waves == 4) hipLaunchKernelGGL((delaunay_kernel<false, 4>), dim3((unsigned)n_frames), dim3(4 * kWave), lds, ctx_stream(ctx), a);
    else hipLaunchKernelGGL(delaunay_kernel<false>, dim3((unsigned)n_frames), dim3(kDtBlock), lds, ctx_stream(ctx), a);
    return check_launch("delaunay_kernel");
}
