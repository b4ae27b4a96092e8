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

// Uniform 2-D grid over the live scan (x,y), cell edge c = r*(1+2^-10) so that
// two points within r of each other always land in cells whose coordinates
// differ by at most 1 (cell coordinates are computed in float64; the map
// v -> cell is monotone, so clamping to the grid keeps that property).
struct PPGrid {
    double ox, oy, inv_c;
};

__device__ __forceinline__ int pp_cell_coord(float v, double o, double inv_c, int n) {
    double f = floor(((double)v - o) * inv_c);
    f = fmin(fmax(f, 0.0), (double)(n - 1));
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
