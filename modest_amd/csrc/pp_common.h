// Shared device helpers for the PP-score (ephemerality) kernels.
#pragma once
#include "common.h"

namespace modest {

constexpr int PP_MAX_TRAV = 128;

// Traversal prefix offsets, passed to kernels by value (kernarg segment).
struct TravOffsets {
    long long off[PP_MAX_TRAV + 1];
    int n;
};

// Uniform 2-D grid over the live scan (x,y), cell edge c = r*(1+2^-10) so that two points within
// r of each other always land in cells whose coordinates differ by at most 1.
// The coordinate is evaluated in float32: t = fl(fl(v - o) * inv_c).  For a point inside the
// grid (|v - o| <= 640 c) the subtraction is exact (Sterbenz) or has relative error 2^-24, the
// product another 2^-24, i.e. at most 7.7e-5 cells in total; two points within r differ by at
// most r/c = 1 - 9.76e-4 cells, so their computed coordinates differ by less than 1 and the
// floors by at most 1.  v -> cell is monotone (both operations are), so clamping to the grid keeps
// the property for points outside it.  All kernels use this one function.
struct PPGrid {
    float ox, oy, inv_c;
};

__device__ __forceinline__ int pp_cell_coord(float v, float o, float inv_c, int n) {
    float f = floorf((v - o) * inv_c);
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));   // NaN -> 0
    return (int)f;
}

// scipy cKDTree (MinkowskiDistP2): r = 0; r += diff*diff for x, y, z in
// float64, compared `<= r*r`.  Compiled with -ffp-contract=off; the explicit
// *_rn intrinsics document that no FMA may be formed here.
__device__ __forceinline__ bool pp_within(double hx, double hy, double hz, float lx,
                                          float ly, float lz, double r2) {
    const double dx = (double)lx - hx;
    const double dy = (double)ly - hy;
    const double dz = (double)lz - hz;
    const double d =
        __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    return d <= r2;
}

// ---- entropy (compute_ephe_score, pre_compute_pp_score.py:68-75) -----------------------------
__device__ __forceinline__ double pp_term(int c, double denom) {
    const double P = (double)c / denom;
    return (-P) * log(P + 1e-8);
}

// numpy's pairwise summation order for a contiguous run of n <= 128 doubles
// (8 interleaved accumulators, then the remainder sequentially).
__device__ __forceinline__ void pp_entropy_kernel_body(const int *__restrict__ counts, int n, int T, float *__restrict__ H, const unsigned bx, const unsigned gx) {
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int *c = counts + (size_t)i * T;
    long long s = 0;
    for (int t = 0; t < T; ++t) s += c[t];
    const double denom = (double)s + 1e-8;
    double res;
    if (T < 8) {
        res = 0.0;
        for (int t = 0; t < T; ++t) res += pp_term(c[t], denom);
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pp_term(c[j], denom);
        int t = 8;
        for (; t < T - (T % 8); t += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += pp_term(c[t + j], denom);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; t < T; ++t) res += pp_term(c[t], denom);
    }
    H[i] = (float)(res / log((double)T));
}

}  // namespace modest
