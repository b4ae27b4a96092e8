// Rotated BEV IoU / overlap / NMS for gfx950.
//
// Replaces the reference's `iou3d_nms_cuda` extension
// (generate_cluster_mask/utils/iou3d_nms/src/iou3d_nms_kernel.cu:236-372 and
// the host glue src/iou3d_nms.cpp:41-186).  The arithmetic contract is the
// float32 operation order of the reference's rotated-rectangle clipping
// (edge intersections + contained corners with a 1e-2 margin, angular sort of
// at most 24 vertices, fan area); this file is compiled with
// -ffp-contract=off so no multiply-add is fused.  K (boxes per scan) is tens,
// so the design goal here is exactness and one launch, not bandwidth:
// one lane per (a,b) pair, 64-lane wavefront blocks, the clipped polygon of each
// pair in an LDS column; the NMS mask kernel also stages its 64 column boxes in LDS.
#include "common.h"
#include "trig_f32.h"
#include <vector>
#include <cstring>

namespace {

constexpr float IOU_EPS = 1e-8f;
constexpr int NMS_TPB = 64;  // one wavefront per block, 64-bit suppression words

struct Pt {
    float x, y;
};

__device__ __forceinline__ float cross2(const Pt &a, const Pt &b) { return a.x * b.y - a.y * b.x; }

__device__ __forceinline__ float cross3(const Pt &p1, const Pt &p2, const Pt &p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(const Pt &p1, const Pt &p2, const Pt &q1, const Pt &q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// Point-in-rotated-box with the reference's 1e-2 margin; (c,s) = cos/sin(-heading).
__device__ __forceinline__ bool in_box2d(const float *box, float c, float s, const Pt &p) {
    const float MARGIN = 1e-2f;
    const float cx = box[0], cy = box[1];
    const float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    const float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

__device__ __forceinline__ bool seg_intersection(const Pt &p1, const Pt &p0, const Pt &q1,
                                                 const Pt &q0, Pt &ans) {
    if (!rect_cross(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ void rot_center(const Pt &ctr, float c, float s, Pt &p) {
    const float nx = (p.x - ctr.x) * c + (p.y - ctr.y) * (-s) + ctr.x;
    const float ny = (p.x - ctr.x) * s + (p.y - ctr.y) * c + ctr.y;
    p.x = nx;
    p.y = ny;
}

// The clipped polygon (at most 24 vertices) and its polar angles live in LDS, one column per lane
// ([vertex][lane], conflict-free): as indexed per-thread arrays they went to scratch memory, which a
// kernel pays for at dispatch even before the first access.
constexpr int POLY_MAX = 24;
struct PolyLds {
    float x[POLY_MAX][NMS_TPB], y[POLY_MAX][NMS_TPB], ang[POLY_MAX][NMS_TPB];
};

__device__ float box_overlap(const float *a, const float *b, PolyLds &L) {
    const int ln = threadIdx.x & (NMS_TPB - 1);
    const float a_ang = a[6], b_ang = b[6];
    const float a_dxh = a[3] / 2, b_dxh = b[3] / 2, a_dyh = a[4] / 2, b_dyh = b[4] / 2;
    const float ax1 = a[0] - a_dxh, ay1 = a[1] - a_dyh, ax2 = a[0] + a_dxh, ay2 = a[1] + a_dyh;
    const float bx1 = b[0] - b_dxh, by1 = b[1] - b_dyh, bx2 = b[0] + b_dxh, by2 = b[1] + b_dyh;
    const Pt ca{a[0], a[1]}, cb{b[0], b[1]};
    Pt A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
    Pt B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
    const float a_cos = modest::cos_f32(a_ang), a_sin = modest::sin_f32(a_ang);
    const float b_cos = modest::cos_f32(b_ang), b_sin = modest::sin_f32(b_ang);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rot_center(ca, a_cos, a_sin, A[k]);
        rot_center(cb, b_cos, b_sin, B[k]);
    }
    A[4] = A[0];
    B[4] = B[0];

    Pt center{0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pt x;
            if (seg_intersection(A[i + 1], A[i], B[j + 1], B[j], x)) {
                center.x = center.x + x.x;
                center.y = center.y + x.y;
                L.x[cnt][ln] = x.x;
                L.y[cnt][ln] = x.y;
                ++cnt;
            }
        }
    // the containment tests rotate by the opposite angle: cos(-h), sin(-h)
    const float a_ncos = modest::cos_f32(-a_ang), a_nsin = modest::sin_f32(-a_ang);
    const float b_ncos = modest::cos_f32(-b_ang), b_nsin = modest::sin_f32(-b_ang);
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, a_ncos, a_nsin, B[k])) {
            center.x = center.x + B[k].x;
            center.y = center.y + B[k].y;
            L.x[cnt][ln] = B[k].x;
            L.y[cnt][ln] = B[k].y;
            ++cnt;
        }
        if (in_box2d(b, b_ncos, b_nsin, A[k])) {
            center.x = center.x + A[k].x;
            center.y = center.y + A[k].y;
            L.x[cnt][ln] = A[k].x;
            L.y[cnt][ln] = A[k].y;
            ++cnt;
        }
    }
    center.x /= cnt;
    center.y /= cnt;

    // bubble sort by polar angle around the centroid, descending swaps as in the reference
    for (int i = 0; i < cnt; ++i) L.ang[i][ln] = modest::atan2_f32(L.y[i][ln] - center.y, L.x[i][ln] - center.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float a0 = L.ang[i][ln], a1 = L.ang[i + 1][ln];
            if (a0 > a1) {
                const float tx = L.x[i][ln], ty = L.y[i][ln];
                L.x[i][ln] = L.x[i + 1][ln];
                L.y[i][ln] = L.y[i + 1][ln];
                L.x[i + 1][ln] = tx;
                L.y[i + 1][ln] = ty;
                L.ang[i][ln] = a1;
                L.ang[i + 1][ln] = a0;
            }
        }
    float area = 0.f;
    const float x0 = cnt > 0 ? L.x[0][ln] : 0.f, y0 = cnt > 0 ? L.y[0][ln] : 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const Pt u{L.x[k][ln] - x0, L.y[k][ln] - y0};
        const Pt v{L.x[k + 1][ln] - x0, L.y[k + 1][ln] - y0};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float *a, const float *b, PolyLds &L) {
    const float sa = a[3] * a[4];
    const float sb = b[3] * b[4];
    const float so = box_overlap(a, b, L);
    return so / fmaxf(sa + sb - so, IOU_EPS);
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {
    // src/iou3d_nms_kernel.cu:314-325 (axis-aligned)
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, IOU_EPS);
}

template <bool IOU>
__global__ __launch_bounds__(NMS_TPB) void pair_kernel(const float *__restrict__ A, int na,
                                                       const float *__restrict__ B, int nb,
                                                       float *__restrict__ out) {
    __shared__ PolyLds L;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)na * nb) return;
    const int ia = (int)(id / nb), ib = (int)(id % nb);
    float a[7], b[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        a[k] = A[(size_t)ia * 7 + k];
        b[k] = B[(size_t)ib * 7 + k];
    }
    out[id] = IOU ? iou_bev(a, b, L) : box_overlap(a, b, L);
}

// 64x64 tile suppression words (src/iou3d_nms_kernel.cu:267-311, :328-372).
template <bool ROTATED>
__global__ __launch_bounds__(NMS_TPB) void nms_mask_kernel(int n, float thresh,
                                                           const float *__restrict__ boxes,
                                                           unsigned long long *__restrict__ mask) {
    const int row_start = blockIdx.y, col_start = blockIdx.x;
    const int row_size = min(n - row_start * NMS_TPB, NMS_TPB);
    const int col_size = min(n - col_start * NMS_TPB, NMS_TPB);
    __shared__ float sb[NMS_TPB * 7];
    __shared__ PolyLds L;
    if ((int)threadIdx.x < col_size)
        for (int k = 0; k < 7; ++k)
            sb[threadIdx.x * 7 + k] = boxes[(size_t)(NMS_TPB * col_start + threadIdx.x) * 7 + k];
    __syncthreads();
    if ((int)threadIdx.x < row_size) {
        const int cur = NMS_TPB * row_start + threadIdx.x;
        float c[7];
        for (int k = 0; k < 7; ++k) c[k] = boxes[(size_t)cur * 7 + k];
        unsigned long long t = 0;
        int start = (row_start == col_start) ? threadIdx.x + 1 : 0;
        for (int i = start; i < col_size; ++i) {
            const float v = ROTATED ? iou_bev(c, sb + i * 7, L) : iou_normal(c, sb + i * 7);
            if (v > thresh) t |= 1ULL << i;
        }
        const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
        mask[(size_t)cur * col_blocks + col_start] = t;
    }
}

int pair_launch(bool iou, const float *a, int na, const float *b, int nb, float *out, void *stream) {
    MODEST_REQUIRE(na >= 0 && nb >= 0, "negative box count");
    if (na == 0 || nb == 0) return MODEST_OK;
    MODEST_REQUIRE(a && b && out, "NULL buffer");
    const long long total = (long long)na * nb;
    const int blocks = (int)((total + NMS_TPB - 1) / NMS_TPB);
    if (iou)
        pair_kernel<true><<<blocks, NMS_TPB, 0, as_stream(stream)>>>(a, na, b, nb, out);
    else
        pair_kernel<false><<<blocks, NMS_TPB, 0, as_stream(stream)>>>(a, na, b, nb, out);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

int nms_impl(bool rotated, modest_ctx *ctx, const float *boxes, int n, float thresh, int64_t *keep,
             int *num_keep, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0, "n < 0");
    MODEST_REQUIRE(num_keep != nullptr, "num_keep is NULL");
    *num_keep = 0;
    if (n == 0) return MODEST_OK;
    MODEST_REQUIRE(boxes && keep, "NULL buffer");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = as_stream(stream_);
    const int cb = (n + NMS_TPB - 1) / NMS_TPB;
    const size_t words = (size_t)n * cb;
    int rc = modest_ctx_reserve(ctx, arena_sz(words * 8));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, words * 8);
    if (rc) return rc;
    unsigned long long *dmask = reinterpret_cast<unsigned long long *>(ctx->scratch);
    unsigned long long *hmask = reinterpret_cast<unsigned long long *>(ctx->pinned);
    dim3 grid(cb, cb);
    if (rotated)
        nms_mask_kernel<true><<<grid, NMS_TPB, 0, stream>>>(n, thresh, boxes, dmask);
    else
        nms_mask_kernel<false><<<grid, NMS_TPB, 0, stream>>>(n, thresh, boxes, dmask);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipMemcpyAsync(hmask, dmask, words * 8, hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    // sequential reduction over the sorted boxes (src/iou3d_nms.cpp:116-133)
    std::vector<unsigned long long> remv(cb, 0ULL);
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const int nblock = i / NMS_TPB, inblock = i % NMS_TPB;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep[kept++] = i;
            const unsigned long long *p = hmask + (size_t)i * cb;
            for (int j = nblock; j < cb; ++j) remv[j] |= p[j];
        }
    }
    *num_keep = kept;
    return MODEST_OK;
}

}  // namespace

extern "C" int modest_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out,
                                        void *stream) {
    return pair_launch(false, a, na, b, nb, out, stream);
}

extern "C" int modest_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out,
                                    void *stream) {
    return pair_launch(true, a, na, b, nb, out, stream);
}

extern "C" int modest_nms_bev(modest_ctx *ctx, const float *boxes, int n, float thresh,
                              int64_t *keep, int *num_keep, void *stream) {
    return nms_impl(true, ctx, boxes, n, thresh, keep, num_keep, stream);
}

extern "C" int modest_nms_normal(modest_ctx *ctx, const float *boxes, int n, float thresh,
                                 int64_t *keep, int *num_keep, void *stream) {
    return nms_impl(false, ctx, boxes, n, thresh, keep, num_keep, stream);
}

extern "C" int modest_boxes_iou_bev_host(modest_ctx *ctx, const float *a_host, int na,
                                         const float *b_host, int nb, float *out_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(na >= 0 && nb >= 0, "negative box count");
    if (na == 0 || nb == 0) return MODEST_OK;
    MODEST_REQUIRE(a_host && b_host && out_host, "NULL buffer");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    // The boxes are staged in the context's pinned block and the kernel reads them and writes the
    // matrix there directly (a few hundred bytes each way): one launch, one stream sync, no copies.
    const size_t ba = arena_sz((size_t)na * 28), bb = arena_sz((size_t)nb * 28);
    const size_t bo = arena_sz((size_t)na * nb * 4);
    int rc = modest_ctx_reserve_pinned(ctx, ba + bb + bo);
    if (rc) return rc;
    float *pa = reinterpret_cast<float *>(ctx->pinned);
    float *pb = reinterpret_cast<float *>(ctx->pinned + ba);
    float *pout = reinterpret_cast<float *>(ctx->pinned + ba + bb);
    memcpy(pa, a_host, (size_t)na * 28);
    memcpy(pb, b_host, (size_t)nb * 28);
    rc = pair_launch(true, pa, na, pb, nb, pout, stream_);
    if (rc) return rc;
    MODEST_HIP_CHECK(hipStreamSynchronize(as_stream(stream_)));
    memcpy(out_host, pout, (size_t)na * nb * 4);
    return MODEST_OK;
}
