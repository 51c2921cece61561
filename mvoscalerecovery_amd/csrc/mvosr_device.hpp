// mvosr_device.hpp — device-side building blocks shared by the scale-recovery kernels (gfx950).
//
// Wave-level helpers (64-lane DPP reductions / ballots), block-level fixed-order reductions, the
// 169-bin bit-set helpers, and the per-triangle 3x3 solve.  All arithmetic is IEEE binary64 and
// the file is compiled with -ffp-contract=off: a fused multiply-add appears only where it is
// written explicitly (the LU elimination and the divide-by-3), so that e.g. the histogram edges
// k*0.1 and the remove_single bounds (k+1)*0.1-0.1 are the very doubles NumPy produces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvosr {

constexpr int kWave = 64;
constexpr int kBins = 169;                  // np.histogram bins, scale_calculator.py:326
constexpr int kCounterBias = 0x8000;        // 16-bit vote counters are stored biased

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// one row of scipy's `simplices`: three int32 vertex ids, loaded with one global_load_dwordx3
struct TriIds { int a, b, c; };
// `tri` is the FRAME's first row (a wave-uniform base) and `t` a 32-bit row index, so that the load
// takes the scalar-base + 32-bit-offset addressing form instead of 64-bit VGPR arithmetic.
__device__ __forceinline__ TriIds load_tri(const int32_t *tri, int t) {
    const int32_t *p = tri + 3 * t;
    TriIds r;
    r.a = p[0]; r.b = p[1]; r.c = p[2];
    return r;
}

// ---- wave reductions on DPP (no LDS crossbar traffic) -----------------------------------------
// Four DPP steps (quad xor 1, quad xor 2, row_half_mirror, row_mirror) leave every lane with the
// sum of its row of 16; the four row sums are then read with v_readlane and added in a fixed
// order, so every lane ends with the same (bitwise) total and results are run-to-run identical.
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror
constexpr int kDppMirror = 0x140;       // row_mirror

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    return ((readlane_d(v, 0) + readlane_d(v, 16)) + readlane_d(v, 32)) + readlane_d(v, 48);
}
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, dpp_mov<kDppXor1>(v));
    v = max(v, dpp_mov<kDppXor2>(v));
    v = max(v, dpp_mov<kDppHalfMirror>(v));
    v = max(v, dpp_mov<kDppMirror>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_mov<kDppXor1>(v));
    v = min(v, dpp_mov<kDppXor2>(v));
    v = min(v, dpp_mov<kDppHalfMirror>(v));
    v = min(v, dpp_mov<kDppMirror>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- block reductions over WAVES wavefronts, fixed order => run-to-run bit-identical ---------
// Every call site owns its own scratch slot (2*WAVES doubles) and a workgroup handles exactly
// one frame, so a slot is written once: ONE barrier per reduction, none to protect reuse.
template <int WAVES>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *slot) {
    a = wave_sum(a);
    b = wave_sum(b);
    if constexpr (WAVES > 1) {
        const int w = wave_id();
        if (lane_id() == 0) { slot[2 * w] = a; slot[2 * w + 1] = b; }
        __syncthreads();
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) { ta += slot[2 * i]; tb += slot[2 * i + 1]; }
        a = ta; b = tb;
    }
}
// three sums at once: (a, b) through `slot`, c through `slot_c` (WAVES doubles, another site's unused scratch)
template <int WAVES>
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *slot, double *slot_c) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if constexpr (WAVES > 1) {
        const int w = wave_id();
        if (lane_id() == 0) { slot[2 * w] = a; slot[2 * w + 1] = b; slot_c[w] = c; }
        __syncthreads();
        double ta = 0.0, tb = 0.0, tc = 0.0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) { ta += slot[2 * i]; tb += slot[2 * i + 1]; tc += slot_c[i]; }
        a = ta; b = tb; c = tc;
    }
}
// four int counters at once (`slot` reinterpreted: 4*WAVES ints = 2*WAVES doubles)
template <int WAVES>
__device__ __forceinline__ void block_sum4i(int &a, int &b, int &c, int &d, double *slot_d) {
    int *slot = reinterpret_cast<int *>(slot_d);
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
    if constexpr (WAVES > 1) {
        const int w = wave_id();
        if (lane_id() == 0) { slot[4 * w] = a; slot[4 * w + 1] = b; slot[4 * w + 2] = c; slot[4 * w + 3] = d; }
        __syncthreads();
        int ta = 0, tb = 0, tc = 0, td = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) { ta += slot[4 * i]; tb += slot[4 * i + 1]; tc += slot[4 * i + 2]; td += slot[4 * i + 3]; }
        a = ta; b = tb; c = tc; d = td;
    }
}

// ---- 169-bit sets held as three wave-uniform 64-bit ballots -----------------------------------
struct Bits192 {
    unsigned long long w[3];
    __device__ __forceinline__ bool any() const { return (w[0] | w[1] | w[2]) != 0ull; }
    __device__ __forceinline__ bool test(int i) const {
        const unsigned long long x = i < 64 ? w[0] : (i < 128 ? w[1] : w[2]);
        return (x >> (i & 63)) & 1ull;
    }
    // per-lane test of a lane-varying index in 0..191 (branch-free)
    __device__ __forceinline__ bool test_lane(int i) const {
        const unsigned long long x = i < 64 ? w[0] : (i < 128 ? w[1] : w[2]);
        return (x >> (i & 63)) & 1ull;
    }
    // highest set index < limit, -1 if none
    __device__ __forceinline__ int highest_below(int limit) const {
        for (int c = 2; c >= 0; --c) {
            const int lo = c * 64;
            if (limit <= lo) continue;
            unsigned long long x = w[c];
            if (limit - lo < 64) x &= (1ull << (limit - lo)) - 1ull;
            if (x) return lo + 63 - __clzll((long long)x);
        }
        return -1;
    }
    // lowest set index >= start, -1 if none
    __device__ __forceinline__ int lowest_from(int start) const {
        for (int c = 0; c < 3; ++c) {
            const int lo = c * 64;
            if (start >= lo + 64) continue;
            unsigned long long x = w[c];
            if (start > lo) x &= ~((1ull << (start - lo)) - 1ull);
            if (x) return lo + __ffsll((long long)x) - 1;
        }
        return -1;
    }
    // number of maximal runs of consecutive set bits
    __device__ __forceinline__ int runs() const {
        int r = 0;
        unsigned long long carry = 0ull;          // bit 63 of the previous word
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned long long prev = (w[c] << 1) | carry;
            r += __popcll(w[c] & ~prev);
            carry = w[c] >> 63;
        }
        return r;
    }
};

// Bin edge k of np.array(range(0,170))*0.1 (scale_calculator.py:326): int -> double, one multiply.
__device__ __forceinline__ double bin_edge(int k) { return (double)k * 0.1; }

// np.histogram(y, bins=edges) bin of y: edges[k] <= y < edges[k+1], last bin closed, -1 outside.
// floor(10 y) is off by at most one bin from the bin defined by the edges k*0.1 (one rounding in
// 10*y, one in k*0.1), so one correction step each way settles it, branch-free.
__device__ __forceinline__ int bin_of(double y) {
    if (!(y >= 0.0 && y <= bin_edge(kBins))) return -1;
    int k = (int)(y * 10.0);
    k = min(k, kBins - 1);
    k += (k < kBins - 1 && y >= bin_edge(k + 1)) ? 1 : 0;
    k -= (k > 0 && y < bin_edge(k)) ? 1 : 0;
    return k;
}

// x/3 correctly rounded (Markstein: q = RN(x*c), r = x - 3q exact by FMA, RN(q + r*c) with
// c = RN(1/3)); equals the IEEE quotient NumPy's true_divide produces for np.mean(...,1)
// (scale_calculator.py:238).  Falls back to a real division outside the safe exponent range.
__device__ __forceinline__ double div3(double x) {
    const double c = 0x1.5555555555555p-2;
    const double ax = fabs(x);
    const double q = x * c;
    const double r = __builtin_fma(-3.0, q, x);
    double res = __builtin_fma(r, c, q);
    if (!((ax > 0x1p-900) & (ax < 0x1p900))) {
        // rare, and it has to stay a skipped region: without the asm the compiler evaluates the division for every
        // triangle and selects (some 17 instructions more per row, one of them a quarter-rate v_rcp_f64)
        asm volatile("" : "+v"(x));
        res = x / 3.0;
    }
    return res;
}

// Solve A n = (1,1,1)^T for the 3x3 matrix whose ROWS are the triangle's vertices — the plane
// n.p = 1 through them — by LU with partial (row) pivoting, the algorithm LAPACK's gesv applies
// when the reference evaluates np.matrix(A).I @ ones (scale_calculator.py:229-230).
// Returns false when a pivot is exactly zero (LAPACK info > 0 -> numpy LinAlgError).
__device__ __forceinline__ bool plane_normal(double a00, double a01, double a02,
                                             double a10, double a11, double a12,
                                             double a20, double a21, double a22,
                                             double &nx, double &ny, double &nz) {
    double b0 = 1.0, b1 = 1.0, b2 = 1.0;
    // column 0: pivot = largest |a_i0| (first maximum wins, like idamax)
    {
        const double m0 = fabs(a00), m1 = fabs(a10), m2 = fabs(a20);
        const bool p2 = (m2 > m0) && (m2 > m1);
        const bool p1 = !p2 && (m1 > m0);
        if (p1) { double t; t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; t = a02; a02 = a12; a12 = t; }
        if (p2) { double t; t = a00; a00 = a20; a20 = t; t = a01; a01 = a21; a21 = t; t = a02; a02 = a22; a22 = t; }
        // (right-hand side is all ones: swapping its rows changes nothing yet)
    }
    bool ok = (a00 != 0.0);
    const double r0 = 1.0 / a00;
    const double l10 = a10 * r0, l20 = a20 * r0;
    a11 = __builtin_fma(-l10, a01, a11); a12 = __builtin_fma(-l10, a02, a12); b1 = __builtin_fma(-l10, b0, b1);
    a21 = __builtin_fma(-l20, a01, a21); a22 = __builtin_fma(-l20, a02, a22); b2 = __builtin_fma(-l20, b0, b2);
    // column 1
    if (fabs(a21) > fabs(a11)) {
        double t; t = a11; a11 = a21; a21 = t; t = a12; a12 = a22; a22 = t; t = b1; b1 = b2; b2 = t;
    }
    ok = ok && (a11 != 0.0);
    const double r1 = 1.0 / a11;
    const double l21 = a21 * r1;
    a22 = __builtin_fma(-l21, a12, a22); b2 = __builtin_fma(-l21, b1, b2);
    ok = ok && (a22 != 0.0);
    // back substitution
    nz = b2 / a22;
    ny = __builtin_fma(-a12, nz, b1) * r1;
    nx = __builtin_fma(-a02, nz, __builtin_fma(-a01, ny, b0)) * r0;
    return ok;
}

}  // namespace mvosr
