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

}  // namespace modest
