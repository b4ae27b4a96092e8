// Device-side descriptor of one frame of the frame store, shared by the two neighbour-count paths
// that read frames through a descriptor table (pp_count.hip: V3 streaming kernels with the
// transform fused; pp_frames.hip: the gather-join).
#pragma once
#include "pp_common.h"

namespace modest {

constexpr int F_FLAG_CENTER = 1;   // remove_center (pre_compute_pp_score.py:48-52)

struct FrameDev {   // 80 bytes
    const float *xyz;      // (n,3) raw points (tile-sorted by the frame store)
    const unsigned *tab;   // tile prefix table of the frame store
    int n, TX0, TY0;       // table anchor in global tile coordinates
    int trav_flags;        // traversal | flags << 16
    float rel[12];         // rows 0..2 of the float32 relative pose (get_relative_pose)
};
static_assert(sizeof(FrameDev) == 80, "descriptor layout");

// transform_points (utils/pointcloud_utils.py:11-19): the float32 BLAS rounding pinned by
// tests/golden/transform.npz (csrc/transform.hip has the same chain).
__device__ __forceinline__ void rel_apply(const float *__restrict__ T, float x, float y, float z, float *o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float acc = x * T[4 * r + 0];
        acc = fmaf(y, T[4 * r + 1], acc);
        acc = fmaf(z, T[4 * r + 2], acc);
        o[r] = acc + T[4 * r + 3];
    }
}

__device__ __forceinline__ bool in_center_box(float x, float y) {
    return (x < 1.75f) && (x >= -1.15f) && (y < 0.65f) && (y >= -0.65f);
}

}  // namespace modest

// host side of the frame path of the V3 kernels (pp_count.hip), called by modest_pp_score_frames
int modest_pp3_frames(modest_ctx *ctx, const modest_pp_frame *live, const uint32_t *live_perm_dev,
                      const modest_pp_frame *frames, int n_frames, int n_trav, double radius,
                      int32_t *counts_dev, float *H_dev, hipStream_t stream);
// the same chain once for a batch of scans (every kernel takes the scan as blockIdx.y), called by
// modest_pp_score_frames_batch
int modest_pp3_frames_batch(modest_ctx *ctx, int n_scans, const modest_pp_frame *const *live,
                            const uint32_t *const *live_perm_dev, const modest_pp_frame *const *frames,
                            const int *n_frames, int n_trav, double radius, int32_t *const *counts_dev,
                            float *const *H_dev, hipStream_t stream);
