// above_plane + range mask predicate (utils/pointcloud_utils.py:68-81, generate_mask.py:57-65),
// shared by mask_kernel (plane.hip) and the fused mask + grid-count kernel (cluster.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace modest {

struct MaskParams {
    double n0, n1, n2, d, norm, offset;
    // range bounds are compared in float32, as numpy compares a float32 array
    // with Python scalars (weak-scalar promotion)
    float ox0, ox1, oy0, oy1;   // only_range (strict)
    float lx0, lx1, ly0, ly1;   // limit_range (lo, hi]
    int use_only_range;
};

__device__ __forceinline__ bool mask_keep(const MaskParams &P, float x, float y, float z) {
    // ptc @ plane[:3] + plane[3], float64: x*n0, fma(y,n1,.), fma(z,n2,.), then + d, then / norm
    double dist = (double)x * P.n0;
    dist = fma((double)y, P.n1, dist);
    dist = fma((double)z, P.n2, dist);
    dist = dist + P.d;
    dist = dist / P.norm;
    bool below = dist < P.offset;
    if (P.use_only_range) below = below && (x < P.ox1) && (x > P.ox0) && (y < P.oy1) && (y > P.oy0);
    const bool range = (x <= P.lx1) && (x > P.lx0) && (y <= P.ly1) && (y > P.ly0);
    return (!below) && range;
}

// host side: the parameter block from the arguments of the C ABI
inline void mask_params_fill(MaskParams &P, const double *plane4, double offset, const double *only_range4,
                             const double *limit_range4) {
    P.n0 = plane4[0];
    P.n1 = plane4[1];
    P.n2 = plane4[2];
    P.d = plane4[3];
    // np.sqrt((plane[:3]**2).sum()): squares summed left to right (n < 8 -> sequential)
    P.norm = sqrt((plane4[0] * plane4[0] + plane4[1] * plane4[1]) + plane4[2] * plane4[2]);
    P.offset = offset;
    P.use_only_range = only_range4 != nullptr;
    if (only_range4) {
        P.ox0 = (float)only_range4[0];
        P.ox1 = (float)only_range4[1];
        P.oy0 = (float)only_range4[2];
        P.oy1 = (float)only_range4[3];
    } else {
        P.ox0 = P.ox1 = P.oy0 = P.oy1 = 0;
    }
    P.lx0 = (float)limit_range4[0];
    P.lx1 = (float)limit_range4[1];
    P.ly0 = (float)limit_range4[2];
    P.ly1 = (float)limit_range4[3];
}

}  // namespace modest
