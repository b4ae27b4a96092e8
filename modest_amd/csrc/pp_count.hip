// PP-score neighbour counting (reference: pre_compute_pp_score.py:54-60,188-193)
// and entropy (pre_compute_pp_score.py:68-75) for gfx950.
//
// Inversion of the reference's "KD-tree over 10.8 M history points, query with
// 30 k live points": the small live scan is indexed (cell-sorted, L2 resident),
// the history is streamed ONCE with coalesced 16-byte loads and rejected early
// against an LDS-resident dilated occupancy bitmap of the live scan.
#include "pp_common.h"
#include <cmath>

using namespace modest;

namespace {

constexpr int PP_NX = 640;
constexpr int PP_NY = 640;
constexpr int PP_NCELL = PP_NX * PP_NY;       // 409,600 cells (192 m at r=0.3)
constexpr int PP_BITWORDS = PP_NCELL / 32;    // 12,800 words = 51,200 B of LDS
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_PER = PP_NCELL / SCAN_THREADS;  // 400
static_assert(PP_NCELL % SCAN_THREADS == 0, "scan tiling");

// ---- live-scan index build -------------------------------------------------

__global__ __launch_bounds__(1024) void pp_live_bbox(const float *__restrict__ live, int n,
                                                     double c, PPGrid *g) {
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = live[3 * (size_t)i], y = live[3 * (size_t)i + 1];
        mnx = fminf(mnx, x);
        mxx = fmaxf(mxx, x);
        mny = fminf(mny, y);
        mxy = fmaxf(mxy, y);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o));
        mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mny = fminf(mny, __shfl_xor(mny, o));
        mxy = fmaxf(mxy, __shfl_xor(mxy, o));
    }
    __shared__ float s[4][16];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
        s[0][w] = mnx;
        s[1][w] = mxx;
        s[2][w] = mny;
        s[3][w] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            s[0][0] = fminf(s[0][0], s[0][k]);
            s[1][0] = fmaxf(s[1][0], s[1][k]);
            s[2][0] = fminf(s[2][0], s[2][k]);
            s[3][0] = fmaxf(s[3][0], s[3][k]);
        }
        double cx = 0.5 * ((double)s[0][0] + (double)s[1][0]);
        double cy = 0.5 * ((double)s[2][0] + (double)s[3][0]);
        if (!(cx == cx) || fabs(cx) > 1e30) cx = 0.0;  // NaN / inf guard
        if (!(cy == cy) || fabs(cy) > 1e30) cy = 0.0;
        g->ox = cx - 0.5 * PP_NX * c;
        g->oy = cy - 0.5 * PP_NY * c;
        g->inv_c = 1.0 / c;
    }
}

__global__ void pp_live_count(const float *__restrict__ live, int n, const PPGrid *g,
                              unsigned *cellCount, unsigned *bitmap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ox = g->ox, oy = g->oy, inv = g->inv_c;
    const int cx = pp_cell_coord(live[3 * (size_t)i], ox, inv, PP_NX);
    const int cy = pp_cell_coord(live[3 * (size_t)i + 1], oy, inv, PP_NY);
    atomicAdd(&cellCount[cy * PP_NX + cx], 1u);
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, PP_NY - 1); ++yy)
        for (int xx = max(cx - 1, 0); xx <= min(cx + 1, PP_NX - 1); ++xx) {
            const int bit = yy * PP_NX + xx;
            atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
        }
}

// Exclusive scan of the PP_NCELL cell counters (one workgroup; the table is
// 1.6 MB and L2 resident, this is a few microseconds of a 30 k-point prologue).
__global__ __launch_bounds__(SCAN_THREADS) void pp_cell_scan(const unsigned *__restrict__ cnt,
                                                             unsigned *__restrict__ start) {
    __shared__ unsigned part[SCAN_THREADS];
    const int tid = threadIdx.x;
    const unsigned *p = cnt + (size_t)tid * SCAN_PER;
    unsigned s = 0;
    for (int k = 0; k < SCAN_PER; ++k) s += p[k];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
        unsigned v = (tid >= o) ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = part[tid] - s;  // exclusive prefix of this thread's segment
    unsigned *q = start + (size_t)tid * SCAN_PER;
    for (int k = 0; k < SCAN_PER; ++k) {
        q[k] = run;
        run += p[k];
    }
    if (tid == SCAN_THREADS - 1) start[PP_NCELL] = run;
}

__global__ void pp_live_scatter(const float *__restrict__ live, int n, const PPGrid *g,
                                const unsigned *__restrict__ start, unsigned *fill,
                                float4 *__restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ox = g->ox, oy = g->oy, inv = g->inv_c;
    const float x = live[3 * (size_t)i], y = live[3 * (size_t)i + 1], z = live[3 * (size_t)i + 2];
    const int cell = pp_cell_coord(y, oy, inv, PP_NY) * PP_NX + pp_cell_coord(x, ox, inv, PP_NX);
    const unsigned slot = start[cell] + atomicAdd(&fill[cell], 1u);
    sorted[slot] = make_float4(x, y, z, __int_as_float(i));
}

// ---- history stream --------------------------------------------------------

__device__ __forceinline__ int pp_find_trav(const TravOffsets &tr, long long p) {
    int t = 0;
    while (t + 1 < tr.n && p >= tr.off[t + 1]) ++t;
    return t;
}

__device__ __forceinline__ void pp_process_point(float x, float y, float z, long long p,
                                                 const unsigned *sbits, double ox, double oy,
                                                 double inv, const TravOffsets &tr,
                                                 const unsigned *__restrict__ cellStart,
                                                 const float4 *__restrict__ sorted,
                                                 int *counts, int T, double r2) {
    const int cx = pp_cell_coord(x, ox, inv, PP_NX);
    const int cy = pp_cell_coord(y, oy, inv, PP_NY);
    const int bit = cy * PP_NX + cx;
    if (!((sbits[bit >> 5] >> (bit & 31)) & 1u)) return;
    const int t = pp_find_trav(tr, p);
    const double hx = x, hy = y, hz = z;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, PP_NX - 1);
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, PP_NY - 1); ++yy) {
        const unsigned s = cellStart[yy * PP_NX + x0];
        const unsigned e = cellStart[yy * PP_NX + x1 + 1];
        for (unsigned j = s; j < e; ++j) {
            const float4 q = sorted[j];
            if (pp_within(hx, hy, hz, q.x, q.y, q.z, r2))
                atomicAdd(&counts[(size_t)__float_as_int(q.w) * T + t], 1);
        }
    }
}

// One lane owns 4 consecutive history points = 48 contiguous bytes, read as
// three 16-byte loads when the base is 16-byte aligned.
template <bool ALIGNED16>
__global__ __launch_bounds__(256) void pp_stream_v1(const float *__restrict__ hist, long long m0,
                                                    long long m1, TravOffsets tr,
                                                    const PPGrid *g,
                                                    const unsigned *__restrict__ bitmap,
                                                    const unsigned *__restrict__ cellStart,
                                                    const float4 *__restrict__ sorted,
                                                    int *counts, int T, double r2) {
    __shared__ unsigned sbits[PP_BITWORDS];
    for (int i = threadIdx.x; i < PP_BITWORDS; i += 256) sbits[i] = bitmap[i];
    __syncthreads();
    const double ox = g->ox, oy = g->oy, inv = g->inv_c;
    const long long M = m1 - m0;
    const long long nchunks = (M + 1023) / 1024;
    for (long long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const long long p0 = m0 + chunk * 1024 + (long long)threadIdx.x * 4;
        if (p0 >= m1) continue;
        float v[12];
        if (ALIGNED16 && p0 + 4 <= m1) {
            const float4 *src = reinterpret_cast<const float4 *>(hist + 3 * p0);
            const float4 a = src[0], b = src[1], c = src[2];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        } else {
            const long long left = m1 - p0;
            for (int k = 0; k < 12; ++k) v[k] = (k / 3 < left) ? hist[3 * p0 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (p0 + k < m1)
                pp_process_point(v[3 * k], v[3 * k + 1], v[3 * k + 2], p0 + k, sbits, ox, oy, inv,
                                 tr, cellStart, sorted, counts, T, r2);
        }
    }
}

// ---- entropy ---------------------------------------------------------------

__device__ __forceinline__ double pp_term(int c, double denom) {
    const double P = (double)c / denom;
    return (-P) * log(P + 1e-8);
}

// numpy's pairwise summation order for a contiguous run of n <= 128 doubles
// (8 interleaved accumulators, then the remainder sequentially).
__global__ void pp_entropy_kernel(const int *__restrict__ counts, int n, int T,
                                  float *__restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
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

int check_offsets(const int64_t *off, int n_trav, TravOffsets &tr) {
    MODEST_REQUIRE(off != nullptr, "trav_offsets is NULL");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= PP_MAX_TRAV, "1 <= n_trav <= 128");
    for (int t = 0; t <= n_trav; ++t) {
        MODEST_REQUIRE(off[t] >= 0, "negative offset");
        if (t) MODEST_REQUIRE(off[t] >= off[t - 1], "offsets must be non-decreasing");
        tr.off[t] = off[t];
    }
    MODEST_REQUIRE(off[n_trav] < (1LL << 31), "history must hold fewer than 2^31 points");
    tr.n = n_trav;
    return MODEST_OK;
}

}  // namespace

// `extra_bytes` of arena are reserved behind this call's own carve and
// returned in *extra (used by modest_pp_score for its private counts).
static int pp_count_impl(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                         const int64_t *trav_offsets, int n_trav, double radius,
                         int32_t *counts, void *stream_, size_t extra_bytes, void **extra) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_live >= 0, "n_live < 0");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    TravOffsets tr;
    int rc = check_offsets(trav_offsets, n_trav, tr);
    if (rc) return rc;
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t zero_words = (size_t)(PP_NCELL + 1) + PP_NCELL + PP_BITWORDS;
    size_t need = arena_sz(sizeof(PPGrid)) + arena_sz(zero_words * 4) +
                  arena_sz((size_t)(PP_NCELL + 1) * 4) + arena_sz((size_t)n_live * 16);
    rc = modest_ctx_reserve(ctx, need + arena_sz(extra_bytes));
    if (rc) return rc;
    if (extra) {
        *extra = ctx->scratch + need;
        if (!counts) counts = static_cast<int32_t *>(*extra);
    }
    if (n_live == 0) return MODEST_OK;
    MODEST_REQUIRE(counts != nullptr, "counts is NULL");
    MODEST_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n_live * n_trav * sizeof(int32_t), stream));
    const long long m0 = tr.off[0], m1 = tr.off[n_trav];
    if (m1 == m0) return MODEST_OK;
    MODEST_REQUIRE(live != nullptr && hist != nullptr, "NULL point buffer");
    Arena A(ctx->scratch);
    PPGrid *g = A.take<PPGrid>(1);
    unsigned *zeroed = A.take<unsigned>(zero_words);
    unsigned *cellCount = zeroed;
    unsigned *fill = zeroed + (PP_NCELL + 1);
    unsigned *bitmap = fill + PP_NCELL;
    unsigned *cellStart = A.take<unsigned>(PP_NCELL + 1);
    float4 *sorted = A.take<float4>(n_live);

    MODEST_HIP_CHECK(hipMemsetAsync(zeroed, 0, zero_words * 4, stream));
    const double c = radius * (1.0 + 1.0 / 1024.0);
    const double r2 = radius * radius;
    pp_live_bbox<<<1, 1024, 0, stream>>>(live, n_live, c, g);
    const int nb = (n_live + 255) / 256;
    pp_live_count<<<nb, 256, 0, stream>>>(live, n_live, g, cellCount, bitmap);
    pp_cell_scan<<<1, SCAN_THREADS, 0, stream>>>(cellCount, cellStart);
    pp_live_scatter<<<nb, 256, 0, stream>>>(live, n_live, g, cellStart, fill, sorted);

    const long long nchunks = (m1 - m0 + 1023) / 1024;
    long long grid = (long long)ctx->num_cus * 3;
    if (grid > nchunks) grid = nchunks;
    const bool aligned = ((reinterpret_cast<uintptr_t>(hist) & 15) == 0) && ((m0 & 3) == 0);
    modest_prof_mark(ctx, stream, 0);
    if (aligned)
        pp_stream_v1<true><<<(int)grid, 256, 0, stream>>>(hist, m0, m1, tr, g, bitmap, cellStart,
                                                          sorted, counts, n_trav, r2);
    else
        pp_stream_v1<false><<<(int)grid, 256, 0, stream>>>(hist, m0, m1, tr, g, bitmap, cellStart,
                                                           sorted, counts, n_trav, r2);
    modest_prof_mark(ctx, stream, 1);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_pp_count(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                               const int64_t *trav_offsets, int n_trav, double radius,
                               int32_t *counts, void *stream_) {
    MODEST_REQUIRE(counts != nullptr || n_live == 0, "counts is NULL");
    return pp_count_impl(ctx, live, n_live, hist, trav_offsets, n_trav, radius, counts, stream_, 0,
                         nullptr);
}

extern "C" int modest_pp_entropy(modest_ctx *ctx, const int32_t *counts, int n_live, int n_trav,
                                 float *H, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_live >= 0, "n_live < 0");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= PP_MAX_TRAV, "1 <= n_trav <= 128");
    if (n_live == 0) return MODEST_OK;
    MODEST_REQUIRE(counts != nullptr && H != nullptr, "NULL buffer");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    pp_entropy_kernel<<<(n_live + 255) / 256, 256, 0, as_stream(stream_)>>>(counts, n_live, n_trav, H);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_pp_score(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                               const int64_t *trav_offsets, int n_trav, double radius,
                               int32_t *counts, float *H, void *stream_) {
    MODEST_REQUIRE(n_live >= 0 && n_trav >= 1 && n_trav <= PP_MAX_TRAV, "bad sizes");
    void *tail = nullptr;
    const size_t extra = counts ? 0 : (size_t)n_live * n_trav * sizeof(int32_t);
    int rc = pp_count_impl(ctx, live, n_live, hist, trav_offsets, n_trav, radius, counts, stream_,
                           extra, &tail);
    if (rc) return rc;
    const int32_t *c = counts ? counts : static_cast<const int32_t *>(tail);
    return modest_pp_entropy(ctx, c, n_live, n_trav, H, stream_);
}
