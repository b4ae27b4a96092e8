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
#include <cmath>
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr float IOU_EPS = 1e-8f;
constexpr int NMS_TPB = 64;  // one wavefront per block, 64-bit suppression words
constexpr size_t NMS_CHUNK_BYTES = 32u << 20;   // suppression words per D2H chunk of the host walk (nms_impl)

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

// ta / tb = (cos h, sin h, cos(-h), sin(-h)) of the two headings
__device__ float box_overlap(const float *a, const float *b, const float4 ta, const float4 tb, PolyLds &L) {
    const int ln = threadIdx.x & (NMS_TPB - 1);
    const float a_dxh = a[3] / 2, b_dxh = b[3] / 2, a_dyh = a[4] / 2, b_dyh = b[4] / 2;
    const float ax1 = a[0] - a_dxh, ay1 = a[1] - a_dyh, ax2 = a[0] + a_dxh, ay2 = a[1] + a_dyh;
    const float bx1 = b[0] - b_dxh, by1 = b[1] - b_dyh, bx2 = b[0] + b_dxh, by2 = b[1] + b_dyh;
    const Pt ca{a[0], a[1]}, cb{b[0], b[1]};
    Pt A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
    Pt B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
    const float a_cos = ta.x, a_sin = ta.y, b_cos = tb.x, b_sin = tb.y;
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
    const float a_ncos = ta.z, a_nsin = ta.w, b_ncos = tb.z, b_nsin = tb.w;
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

// trig of a heading evaluated on the device: float64 evaluation rounded once (trig_f32.h)
__device__ __forceinline__ float4 device_trig(float h) {
    return make_float4(modest::cos_f32(h), modest::sin_f32(h), modest::cos_f32(-h), modest::sin_f32(-h));
}

__device__ __forceinline__ float iou_bev(const float *a, const float *b, const float4 ta, const float4 tb, PolyLds &L) {
    const float sa = a[3] * a[4];
    const float sb = b[3] * b[4];
    const float so = box_overlap(a, b, ta, tb, L);
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

// HOSTTRIG: rows of 11 floats, the box followed by cos h, sin h, cos(-h), sin(-h) evaluated by the
// HOST's libm (modest_boxes_iou_bev_host: objs_nms orders boxes by float noise of their self-IoU,
// SURVEY H6, so the label path uses the very cosf / sinf the reference's CPU path calls).
template <bool IOU, bool HOSTTRIG>
__device__ __forceinline__ void pair_kernel_body(const float *__restrict__ A, int na, const float *__restrict__ B, int nb,
                                                 float *__restrict__ out, const unsigned bx) {
    __shared__ PolyLds L;
    const long long id = (long long)bx * blockDim.x + threadIdx.x;
    if (id >= (long long)na * nb) return;
    const int ia = (int)(id / nb), ib = (int)(id % nb);
    constexpr int STRIDE = HOSTTRIG ? 11 : 7;
    float a[7], b[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        a[k] = A[(size_t)ia * STRIDE + k];
        b[k] = B[(size_t)ib * STRIDE + k];
    }
    float4 ta, tb;
    if (HOSTTRIG) {
        const float *pa = A + (size_t)ia * STRIDE + 7, *pb = B + (size_t)ib * STRIDE + 7;
        ta = make_float4(pa[0], pa[1], pa[2], pa[3]);
        tb = make_float4(pb[0], pb[1], pb[2], pb[3]);
    } else {
        ta = device_trig(a[6]);
        tb = device_trig(b[6]);
    }
    out[id] = IOU ? iou_bev(a, b, ta, tb, L) : box_overlap(a, b, ta, tb, L);
}
template <bool IOU, bool HOSTTRIG>
__global__ __launch_bounds__(NMS_TPB) void pair_kernel(const float *__restrict__ A, int na,
                                                       const float *__restrict__ B, int nb,
                                                       float *__restrict__ out) {
    pair_kernel_body<IOU, HOSTTRIG>(A, na, B, nb, out, blockIdx.x);
}

// the self-IoU matrices of several box sets in one launch (the label stage of a chain of scans): set = blockIdx.y
struct PairSet {
    const float *boxes;   // rows of 11 floats (box + host trig)
    float *out;           // (n, n)
    int n, pad;
};
__global__ __launch_bounds__(NMS_TPB) void pair_sets_kernel(const PairSet *__restrict__ sets) {
    const PairSet S = sets[blockIdx.y];
    if ((long long)blockIdx.x * NMS_TPB >= (long long)S.n * S.n) return;
    pair_kernel_body<true, true>(S.boxes, S.n, S.boxes, S.n, S.out, blockIdx.x);
}

// ---- NMS over score-sorted boxes (semantics: src/iou3d_nms.cpp:90-136, nms_gpu / nms_normal_gpu) ----
// Built for detector scale (thousands of boxes) on this chip, not the reference's structure:
//   nms_trig      heading -> (cos, sin, cos(-h), sin(-h)) once per box (the pair tests reuse them);
//   nms_tiles     UPPER-TRIANGULAR grid of 64 x 64 tiles (linear tile id -> (row block, column block
//                 >= row block): no launched-then-exited blocks), one wavefront per tile, the 64 column
//                 boxes and their trig in LDS; bit j of word (i, cb) = box i suppresses box 64 cb + j;
//   the walk      the sorted order is resolved on the HOST (src/iou3d_nms.cpp:116-135), the suppression words
//                 crossing PCIe in row chunks of <= 32 MB.  A device-side walk (one wavefront, 64 boxes at a
//                 time from cross-lane reads of the diagonal words) was built in round 2 and removed in
//                 round 3: it is a serial chain of ~10 us per 64 boxes and lost to the host loop at every
//                 size (profiles/r03_nms_bench.txt: 5.4 vs 2.2 ms at 20 000 axis-aligned boxes).
__global__ void nms_trig(const float *__restrict__ boxes, int n, float4 *__restrict__ trig) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) trig[i] = device_trig(boxes[(size_t)i * 7 + 6]);
}

template <bool ROTATED>
__global__ __launch_bounds__(NMS_TPB) void nms_tiles(int n, int cb, float thresh, const float *__restrict__ boxes,
                                                     const float4 *__restrict__ trig,
                                                     unsigned long long *__restrict__ mask) {
    // linear id over the upper triangle, row by row: row r holds cb - r tiles
    const long long t = blockIdx.x;
    int r = (int)(((2.0 * cb + 1.0) - sqrt((2.0 * cb + 1.0) * (2.0 * cb + 1.0) - 8.0 * (double)t)) * 0.5);
    auto first = [&](int rr) { return (long long)rr * cb - (long long)rr * (rr - 1) / 2; };
    while (r > 0 && first(r) > t) --r;
    while (first(r + 1) <= t) ++r;
    const int c = r + (int)(t - first(r));
    __shared__ float sb[NMS_TPB * 7];
    __shared__ float4 st[NMS_TPB];
    __shared__ PolyLds L;
    const int lane = threadIdx.x;
    const int col_size = min(n - c * NMS_TPB, NMS_TPB);
    if (lane < col_size) {
        for (int k = 0; k < 7; ++k) sb[lane * 7 + k] = boxes[(size_t)(NMS_TPB * c + lane) * 7 + k];
        if (ROTATED) st[lane] = trig[NMS_TPB * c + lane];
    }
    __syncthreads();
    const int cur = NMS_TPB * r + lane;
    if (cur >= n) return;
    float bx[7];
    for (int k = 0; k < 7; ++k) bx[k] = boxes[(size_t)cur * 7 + k];
    const float4 tc = ROTATED ? trig[cur] : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long w = 0;
    for (int i = (r == c) ? lane + 1 : 0; i < col_size; ++i) {
        const float v = ROTATED ? iou_bev(bx, sb + i * 7, tc, st[i], L) : iou_normal(bx, sb + i * 7);
        if (v > thresh) w |= 1ULL << i;
    }
    mask[(size_t)cur * cb + c] = w;
}

int pair_launch(bool iou, const float *a, int na, const float *b, int nb, float *out, void *stream) {
    MODEST_REQUIRE(na >= 0 && nb >= 0, "negative box count");
    if (na == 0 || nb == 0) return MODEST_OK;
    MODEST_REQUIRE(a && b && out, "NULL buffer");
    const long long total = (long long)na * nb;
    const int blocks = (int)((total + NMS_TPB - 1) / NMS_TPB);
    if (iou)
        pair_kernel<true, false><<<blocks, NMS_TPB, 0, as_stream(stream)>>>(a, na, b, nb, out);
    else
        pair_kernel<false, false><<<blocks, NMS_TPB, 0, as_stream(stream)>>>(a, na, b, nb, out);
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
    MODEST_REQUIRE(cb <= 6144, "at most 393216 boxes");
    const size_t trigB = arena_sz((size_t)n * 16), maskB = arena_sz(words * 8);
    int rc = modest_ctx_reserve(ctx, trigB + maskB);
    if (rc) return rc;
    // the suppression words cross PCIe in chunks of whole rows (<= NMS_CHUNK_BYTES of pinned memory)
    const size_t rowB = (size_t)cb * 8;
    const int rowsPer = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, NMS_CHUNK_BYTES / rowB));
    rc = modest_ctx_reserve_pinned(ctx, (size_t)rowsPer * rowB + 64);
    if (rc) return rc;
    float4 *trig = reinterpret_cast<float4 *>(ctx->scratch);
    unsigned long long *dmask = reinterpret_cast<unsigned long long *>(ctx->scratch + trigB);
    const long long tiles = (long long)cb * (cb + 1) / 2;
    MODEST_REQUIRE(tiles < (1LL << 31), "too many boxes");
    if (rotated) {
        nms_trig<<<(n + 255) / 256, 256, 0, stream>>>(boxes, n, trig);
        nms_tiles<true><<<(unsigned)tiles, NMS_TPB, 0, stream>>>(n, cb, thresh, boxes, trig, dmask);
    } else {
        nms_tiles<false><<<(unsigned)tiles, NMS_TPB, 0, stream>>>(n, cb, thresh, boxes, trig, dmask);
    }
    MODEST_HIP_CHECK(hipGetLastError());
    // src/iou3d_nms.cpp:116-135: box i survives unless an earlier survivor suppressed it; its row then
    // suppresses later boxes (only the column blocks >= i / 64 exist: the tile grid is upper triangular)
    unsigned long long *hmask = reinterpret_cast<unsigned long long *>(ctx->pinned + 64);
    std::vector<unsigned long long> remv((size_t)cb, 0ULL);
    int kept = 0;
    for (int r0 = 0; r0 < n; r0 += rowsPer) {
        const int r1 = std::min(n, r0 + rowsPer);
        MODEST_HIP_CHECK(hipMemcpyAsync(hmask, dmask + (size_t)r0 * cb, (size_t)(r1 - r0) * rowB, hipMemcpyDeviceToHost, stream));
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        for (int i = r0; i < r1; ++i) {
            const int nb = i / NMS_TPB, inb = i % NMS_TPB;
            if (remv[(size_t)nb] & (1ULL << inb)) continue;
            keep[kept++] = i;
            const unsigned long long *row = hmask + (size_t)(i - r0) * cb;
            for (int j = nb; j < cb; ++j) remv[(size_t)j] |= row[j];
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

// the (n_s, n_s) self-IoU matrix of every box set of a chain (objs_nms of B scans): boxes, host trig, the set table and
// the results all live in the context's pinned block; one launch, one synchronise
int modest_boxes_self_iou_bev_host_batch(modest_ctx *ctx, const float *const *boxes_host, const int *n, int B,
                                         float *const *out_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && boxes_host && n && out_host && B >= 1, "bad arguments");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    size_t need = arena_sz((size_t)B * sizeof(PairSet));
    long long maxPairs = 0;
    for (int s = 0; s < B; ++s) {
        MODEST_REQUIRE(n[s] >= 0, "negative box count");
        need += arena_sz((size_t)n[s] * 44) + arena_sz((size_t)n[s] * n[s] * 4);
        maxPairs = std::max(maxPairs, (long long)n[s] * n[s]);
    }
    if (maxPairs == 0) return MODEST_OK;
    int rc = modest_ctx_reserve_pinned(ctx, need);
    if (rc) return rc;
    PairSet *sets = reinterpret_cast<PairSet *>(ctx->pinned);
    size_t off = arena_sz((size_t)B * sizeof(PairSet));
    std::vector<float *> outs((size_t)B, nullptr);
    for (int s = 0; s < B; ++s) {
        float *pb = reinterpret_cast<float *>(ctx->pinned + off);
        off += arena_sz((size_t)n[s] * 44);
        float *po = reinterpret_cast<float *>(ctx->pinned + off);
        off += arena_sz((size_t)n[s] * n[s] * 4);
        MODEST_REQUIRE(n[s] == 0 || (boxes_host[s] && out_host[s]), "NULL buffer");
        for (int i = 0; i < n[s]; ++i) {   // the host's libm evaluates the trig (modest_boxes_iou_bev_host)
            float *d = pb + (size_t)i * 11;
            memcpy(d, boxes_host[s] + (size_t)i * 7, 28);
            const float h = d[6];
            d[7] = cosf(h);
            d[8] = sinf(h);
            d[9] = cosf(-h);
            d[10] = sinf(-h);
        }
        sets[s].boxes = pb;
        sets[s].out = po;
        sets[s].n = n[s];
        sets[s].pad = 0;
        outs[(size_t)s] = po;
    }
    pair_sets_kernel<<<dim3((unsigned)((maxPairs + NMS_TPB - 1) / NMS_TPB), (unsigned)B), NMS_TPB, 0, as_stream(stream_)>>>(sets);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(as_stream(stream_)));
    for (int s = 0; s < B; ++s)
        if (n[s] > 0) memcpy(out_host[s], outs[(size_t)s], (size_t)n[s] * n[s] * 4);
    return MODEST_OK;
}

extern "C" int modest_boxes_iou_bev_host(modest_ctx *ctx, const float *a_host, int na,
                                         const float *b_host, int nb, float *out_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(na >= 0 && nb >= 0, "negative box count");
    if (na == 0 || nb == 0) return MODEST_OK;
    MODEST_REQUIRE(a_host && b_host && out_host, "NULL buffer");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    // The boxes are staged in the context's pinned block -- each followed by cos / sin of its heading
    // evaluated HERE, by the host's libm: objs_nms ranks boxes by the float noise of their self-IoU
    // (pointcloud_utils.py:335-336), so the label path must use the cosf / sinf the reference's CPU
    // path (iou3d_cpu.cpp:128-133) calls, not the device's -- and the kernel reads them and writes the
    // matrix there directly (a few hundred bytes each way): one launch, one stream sync, no copies.
    const size_t ba = arena_sz((size_t)na * 44), bb = arena_sz((size_t)nb * 44);
    const size_t bo = arena_sz((size_t)na * nb * 4);
    int rc = modest_ctx_reserve_pinned(ctx, ba + bb + bo);
    if (rc) return rc;
    float *pa = reinterpret_cast<float *>(ctx->pinned);
    float *pb = reinterpret_cast<float *>(ctx->pinned + ba);
    float *pout = reinterpret_cast<float *>(ctx->pinned + ba + bb);
    auto stage = [](float *dst, const float *src, int n) {
        for (int i = 0; i < n; ++i) {
            float *d = dst + (size_t)i * 11;
            memcpy(d, src + (size_t)i * 7, 28);
            const float h = d[6];
            d[7] = cosf(h);
            d[8] = sinf(h);
            d[9] = cosf(-h);
            d[10] = sinf(-h);
        }
    };
    stage(pa, a_host, na);
    stage(pb, b_host, nb);
    {
        const long long total = (long long)na * nb;
        pair_kernel<true, true><<<(int)((total + NMS_TPB - 1) / NMS_TPB), NMS_TPB, 0, as_stream(stream_)>>>(pa, na, pb, nb, pout);
        MODEST_HIP_CHECK(hipGetLastError());
    }
    MODEST_HIP_CHECK(hipStreamSynchronize(as_stream(stream_)));
    memcpy(out_host, pout, (size_t)na * nb * 4);
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_iou3d(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(nms_trig));
}
