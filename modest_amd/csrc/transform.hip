// transform_points (utils/pointcloud_utils.py:11-19) + remove_center
// (pre_compute_pp_score.py:48-52) for gfx950: raw KITTI frames stay resident
// in HBM as (n,4) float32 and are transformed into the common frame on the fly.
//
// Arithmetic contract (pinned by tests/golden/transform.npz): the reference's
// float32 BLAS product [x y z 1] . T^T rounds as
//     acc = x*T[r][0]; acc = fmaf(y,T[r][1],acc); acc = fmaf(z,T[r][2],acc);
//     out = acc + T[r][3]
// (the trailing 1*T[r][3] is an exact product, so that last fma is an add).
#include "common.h"

namespace {

struct Mat34 {
    float m[12];
};

__device__ __forceinline__ bool in_center(float x, float y) {
    return (x < 1.75f) && (x >= -1.15f) && (y < 0.65f) && (y >= -0.65f);
}

__device__ __forceinline__ void apply(const Mat34 &T, float x, float y, float z, float *o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float acc = x * T.m[4 * r + 0];
        acc = fmaf(y, T.m[4 * r + 1], acc);
        acc = fmaf(z, T.m[4 * r + 2], acc);
        o[r] = acc + T.m[4 * r + 3];
    }
}

__global__ __launch_bounds__(256) void transform_kernel(const float *__restrict__ in, long long n,
                                                        int stride, Mat34 T,
                                                        float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float *p = in + i * stride;
        float o[3];
        apply(T, p[0], p[1], p[2], o);
        out[3 * i + 0] = o[0];
        out[3 * i + 1] = o[1];
        out[3 * i + 2] = o[2];
    }
}

// Order-preserving compaction (np boolean-mask semantics): per-block ballot
// prefix + a chained global offset.  A block takes its LOGICAL index from a
// ticket counter when it starts, so it only ever waits on blocks that are
// already running: no assumption about dispatch order or residency.
__global__ __launch_bounds__(1024) void transform_filter_kernel(const float *__restrict__ in,
                                                                long long n, int stride, Mat34 T,
                                                                float *__restrict__ out,
                                                                unsigned long long *state,
                                                                long long *n_out) {
    __shared__ unsigned wave_cnt[16];
    __shared__ unsigned long long base_s;
    __shared__ unsigned ticket_s;
    if (threadIdx.x == 0) ticket_s = atomicAdd(reinterpret_cast<unsigned *>(state + 1), 1u);
    __syncthreads();
    const unsigned blk = ticket_s;
    const long long i = (long long)blk * 1024 + threadIdx.x;
    bool keep = false;
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float *p = in + i * stride;
        x = p[0];
        y = p[1];
        z = p[2];
        keep = !in_center(x, y);
    }
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned before = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) wave_cnt[w] = __popcll(bal);
    __syncthreads();
    unsigned woff = 0, total = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w) woff += wave_cnt[k];
        total += wave_cnt[k];
    }
    if (threadIdx.x == 0) {
        // state = (#blocks published << 40) | running total ; blocks publish in index order
        unsigned long long s;
        do {
            s = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((s >> 40) != (unsigned long long)blk) __builtin_amdgcn_s_sleep(2);
        } while ((s >> 40) != (unsigned long long)blk);
        base_s = s & ((1ULL << 40) - 1ULL);
        const unsigned long long ns = ((unsigned long long)(blk + 1) << 40) | (base_s + total);
        __hip_atomic_store(state, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blk == gridDim.x - 1 && n_out) *n_out = (long long)(base_s + total);
    }
    __syncthreads();
    if (keep) {
        float o[3];
        apply(T, x, y, z, o);
        const unsigned long long dst = base_s + woff + before;
        out[3 * dst + 0] = o[0];
        out[3 * dst + 1] = o[1];
        out[3 * dst + 2] = o[2];
    }
}

// project_velo_to_rect (utils/kitti_util.py:327-329): rect = (R0 @ ([p,1] @ V2C^T)^T)^T in float64.
// numpy hands both products to dgemm, whose k loop is one fused multiply-add chain per output
// element (first product rounded, then fma per further term; the appended 1 makes the last term
// of the first product an exact addend).
struct RectMats {
    double v[12];   // V2C, row major 3x4
    double r[9];    // R0, row major 3x3
};
__global__ __launch_bounds__(256) void velo_to_rect_kernel(const float *__restrict__ in, int n, int stride, RectMats M,
                                                           double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = in + (size_t)i * stride;
    const double x = p[0], y = p[1], z = p[2];
    double ref[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double acc = __dmul_rn(x, M.v[4 * j]);
        acc = fma(y, M.v[4 * j + 1], acc);
        acc = fma(z, M.v[4 * j + 2], acc);
        ref[j] = fma(1.0, M.v[4 * j + 3], acc);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double acc = __dmul_rn(M.r[3 * j], ref[0]);
        acc = fma(M.r[3 * j + 1], ref[1], acc);
        out[3 * (size_t)i + j] = fma(M.r[3 * j + 2], ref[2], acc);
    }
}

// the same for up to RECT_SETS scans in one launch (the box tail of a chain of scans): the scan is blockIdx.y
constexpr int RECT_SETS = 8;
struct RectSets {
    const float *in[RECT_SETS];
    double *out[RECT_SETS];
    int n[RECT_SETS], stride[RECT_SETS];
};
__global__ __launch_bounds__(256) void velo_to_rect_sets_kernel(RectSets S, RectMats M) {
    const int s = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n[s]) return;
    const float *p = S.in[s] + (size_t)i * S.stride[s];
    const double x = p[0], y = p[1], z = p[2];
    double ref[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double acc = __dmul_rn(x, M.v[4 * j]);
        acc = fma(y, M.v[4 * j + 1], acc);
        acc = fma(z, M.v[4 * j + 2], acc);
        ref[j] = fma(1.0, M.v[4 * j + 3], acc);
    }
    double *out = S.out[s];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double acc = __dmul_rn(M.r[3 * j], ref[0]);
        acc = fma(M.r[3 * j + 1], ref[1], acc);
        out[3 * (size_t)i + j] = fma(M.r[3 * j + 2], ref[2], acc);
    }
}

}  // namespace

int modest_project_velo_to_rect_multi(modest_ctx *ctx, const float *const *pts, const int *n, const int *stride,
                                      double *const *out, int B, const double *V2C12, const double *R09, void *stream_) {
    MODEST_REQUIRE(ctx && pts && n && stride && out && V2C12 && R09 && B >= 1, "bad arguments");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    RectMats M;
    for (int q = 0; q < 12; ++q) M.v[q] = V2C12[q];
    for (int q = 0; q < 9; ++q) M.r[q] = R09[q];
    for (int s0 = 0; s0 < B; s0 += RECT_SETS) {
        const int m = B - s0 < RECT_SETS ? B - s0 : RECT_SETS;
        RectSets S{};
        int maxn = 1;
        for (int s = 0; s < m; ++s) {
            MODEST_REQUIRE(pts[s0 + s] && out[s0 + s] && n[s0 + s] >= 1 && (stride[s0 + s] == 3 || stride[s0 + s] == 4), "bad scan");
            S.in[s] = pts[s0 + s], S.out[s] = out[s0 + s], S.n[s] = n[s0 + s], S.stride[s] = stride[s0 + s];
            maxn = maxn > n[s0 + s] ? maxn : n[s0 + s];
        }
        velo_to_rect_sets_kernel<<<dim3((unsigned)((maxn + 255) / 256), (unsigned)m), 256, 0, as_stream(stream_)>>>(S, M);
    }
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_transform_points(modest_ctx *ctx, const float *in, int64_t n, int in_stride,
                                       const float *T16, int remove_center, float *out,
                                       int64_t *n_out, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0, "n < 0");
    MODEST_REQUIRE(in_stride == 3 || in_stride == 4, "in_stride must be 3 or 4");
    MODEST_REQUIRE(T16 != nullptr, "T16 is NULL");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    Mat34 T;
    for (int k = 0; k < 12; ++k) T.m[k] = T16[k];
    if (n == 0) {
        if (remove_center && n_out) MODEST_HIP_CHECK(hipMemsetAsync(n_out, 0, sizeof(int64_t), stream));
        return MODEST_OK;
    }
    MODEST_REQUIRE(in != nullptr && out != nullptr, "NULL point buffer");
    if (!remove_center) {
        long long blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        transform_kernel<<<(int)blocks, 256, 0, stream>>>(in, n, in_stride, T, out);
    } else {
        int rc = modest_ctx_reserve(ctx, 256);
        if (rc) return rc;
        unsigned long long *state = reinterpret_cast<unsigned long long *>(ctx->scratch);
        MODEST_HIP_CHECK(hipMemsetAsync(state, 0, 16, stream));  // [0] chain word, [1] ticket
        const long long blocks = (n + 1023) / 1024;
        MODEST_REQUIRE(blocks < (1 << 23), "frame too large for remove_center path");
        transform_filter_kernel<<<(int)blocks, 1024, 0, stream>>>(in, n, in_stride, T, out, state,
                                                                  reinterpret_cast<long long *>(n_out));
    }
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_project_velo_to_rect(modest_ctx *ctx, const float *pts, int n, int stride, const double *V2C12,
                                           const double *R09, double *out, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && V2C12 && R09, "NULL argument");
    MODEST_REQUIRE(n >= 0 && (stride == 3 || stride == 4), "bad n/stride");
    if (n == 0) return MODEST_OK;
    MODEST_REQUIRE(pts && out, "NULL buffer");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    RectMats M;
    for (int q = 0; q < 12; ++q) M.v[q] = V2C12[q];
    for (int q = 0; q < 9; ++q) M.r[q] = R09[q];
    velo_to_rect_kernel<<<(n + 255) / 256, 256, 0, stream>>>(pts, n, stride, M, out);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_transform(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(transform_kernel));
}
