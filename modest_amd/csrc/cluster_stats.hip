// Per-cluster statistics for filter_labels / is_valid_cluster
// (generate_cluster_mask/utils/clustering_utils.py:94-135): member count, min / max signed
// distance to the ground plane, and the two order statistics of the PP score that
// numpy.percentile(pp, q) interpolates between (method 'linear': virtual index (n-1)*q).
// The reference evaluates these with one boolean mask + numpy reductions per cluster on the
// host; here the members of every cluster are gathered by a counting sort and one workgroup
// per cluster does the reductions and an exact radix select.
#include "common.h"
#include "radix_select.h"
#include <cmath>
#include <cstdlib>

namespace {

// label histogram; a few clusters hold most points, so same-address global atomics (~12 ns
// each) are first folded per wavefront: lanes with equal labels elect one adder
__global__ void cs_count(const int *__restrict__ labels, int n, int C, unsigned *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int l = (i < n) ? labels[i] : -1;
    if (l >= C) l = -1;
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int ls = __shfl(l, src);
        const unsigned long long same = __ballot(l == ls);
        if ((int)(threadIdx.x & 63) == src) atomicAdd(&cnt[ls], (unsigned)__popcll(same));
        todo &= ~same;
    }
}

__global__ void cs_scan(const unsigned *__restrict__ cnt, int C, unsigned *__restrict__ start) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned run = 0;
    for (int c = 0; c < C; ++c) {
        start[c] = run;
        run += cnt[c];
    }
    start[C] = run;
}

__global__ void cs_scatter(const int *__restrict__ labels, int n, int C, const unsigned *__restrict__ start,
                           unsigned *fill, int *__restrict__ members) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = labels[i];
    if (l >= 0 && l < C) members[start[l] + atomicAdd(&fill[l], 1u)] = i;
}

// cs_count + cs_scan + cs_scatter in one workgroup for the usual case (a scan has tens of clusters
// and 30 k labels): LDS histogram, block scan, LDS cursors -- no same-line global atomics at all
// (they made the three-kernel path take 130 us).
constexpr int CS_GROUP_MAXC = 8192;
constexpr int CS_LDS_KEYS = 8192;
__global__ __launch_bounds__(1024) void cs_group(const int *__restrict__ labels, int n, int C,
                                                 unsigned *__restrict__ start, int *__restrict__ members) {
    __shared__ unsigned hist[CS_GROUP_MAXC];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int c = tid; c < C; c += 1024) hist[c] = 0;
    __syncthreads();
    // eight labels per thread and round, loaded before the atomics (one load latency per round)
    for (int base = 0; base < n; base += 8 * 1024) {
        int l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 1024 + tid;
            l[u] = i < n ? labels[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l[u] >= 0 && l[u] < C) atomicAdd(&hist[l[u]], 1u);
    }
    __syncthreads();
    // exclusive scan of hist[0..C): `per` consecutive bins per thread
    const int per = (C + 1023) / 1024, b0 = min(tid * per, C), b1 = min(b0 + per, C);
    unsigned s = 0;
    for (int c = b0; c < b1; ++c) s += hist[c];
    unsigned inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned run = inc - s;
    for (int k = 0; k < w; ++k) run += wsum[k];
    for (int c = b0; c < b1; ++c) {
        const unsigned v = hist[c];
        start[c] = run;
        hist[c] = run;   // becomes the cursor
        run += v;
    }
    if (tid == 1023) start[C] = run;
    __syncthreads();
    for (int base = 0; base < n; base += 8 * 1024) {
        int l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 1024 + tid;
            l[u] = i < n ? labels[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l[u] >= 0 && l[u] < C) members[atomicAdd(&hist[l[u]], 1u)] = base + u * 1024 + tid;
    }
}

struct PlaneP {
    double n0, n1, n2, d, norm, q;
};

// out[c*6 + {0:n, 1:dmin, 2:dmax, 3:a, 4:b, 5:gamma}]  (doubles)
constexpr int CSS_THREADS = 1024;   // the largest cluster sets the kernel time: more lanes, fewer gather rounds
__global__ __launch_bounds__(CSS_THREADS) void cs_stats(const float *__restrict__ pts, int stride,
                                                       const float *__restrict__ pp,
                                                       const int *__restrict__ members,
                                                       const unsigned *__restrict__ start, PlaneP P,
                                                       double *__restrict__ out) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wsum[CSS_THREADS / 64], sel[2];
    __shared__ double rmin[CSS_THREADS / 64], rmax[CSS_THREADS / 64];
    // PP keys of the members, gathered once (each of the up to seven select passes otherwise
    // repeats the members -> pp chain of dependent loads)
    __shared__ unsigned keys[CS_LDS_KEYS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int n = (int)(start[c + 1] - start[c]);
    const int *mem = members + start[c];
    const bool cached = n <= CS_LDS_KEYS;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < n; i += CSS_THREADS) {
        const int m = mem[i];
        if (cached) keys[i] = cs_key(pp[m]);
        const float *p = pts + (size_t)m * stride;
        // ptc @ plane[:3] + plane[3] then / norm: the float64 rounding of numpy's product
        double dist = (double)p[0] * P.n0;
        dist = fma((double)p[1], P.n1, dist);
        dist = fma((double)p[2], P.n2, dist);
        dist = (dist + P.d) / P.norm;
        mn = fmin(mn, dist);
        mx = fmax(mx, dist);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if ((tid & 63) == 0) {
        rmin[tid >> 6] = mn;
        rmax[tid >> 6] = mx;
    }
    __syncthreads();
    double a = 0.0, b = 0.0, gamma = 0.0;
    if (n > 0) {
        // numpy 'linear' on float32 data works in float32: virtual index (n-1)*q, neighbours
        // floor / floor+1; at or beyond the last index both neighbours are the maximum
        const float qf = (float)P.q;
        const float vi = (float)(n - 1) * qf;
        const float fl = floorf(vi);
        int prev = (int)fl, next = prev + 1;
        if (vi >= (float)(n - 1)) prev = next = n - 1;
        if (vi < 0.f) prev = next = 0;
        next = min(next, n - 1);
        gamma = (double)(vi - fl);
        if (cached) {
            const auto key_at = [&](int i) { return keys[i]; };
            const float af = cs_select_keys<CSS_THREADS>(key_at, n, (unsigned)prev, hist, wsum, sel);
            a = (double)af;
            b = a;
            if (next != prev) {
                // the neighbour rank prev + 1 without a second descent: it is the same value when more
                // than `next` keys are <= a, the smallest key above a otherwise (one pass over LDS)
                const unsigned akey = cs_key(af);
                if (tid == 0) {
                    sel[0] = 0;
                    sel[1] = 0xffffffffu;
                }
                __syncthreads();
                unsigned cle = 0, mgt = 0xffffffffu;
                for (int i = tid; i < n; i += CSS_THREADS) {
                    const unsigned k = keys[i];
                    cle += k <= akey ? 1u : 0u;
                    mgt = k > akey ? min(mgt, k) : mgt;
                }
                for (int o = 32; o > 0; o >>= 1) {
                    cle += __shfl_xor(cle, o);
                    mgt = min(mgt, (unsigned)__shfl_xor((int)mgt, o));
                }
                if ((tid & 63) == 0) {
                    atomicAdd(&sel[0], cle);
                    atomicMin(&sel[1], mgt);
                }
                __syncthreads();
                b = sel[0] > (unsigned)next ? a : (double)cs_unkey(sel[1]);
            }
        } else {
            a = (double)cs_select<CSS_THREADS>(pp, mem, n, (unsigned)prev, hist, wsum, sel);
            b = (next == prev) ? a : (double)cs_select<CSS_THREADS>(pp, mem, n, (unsigned)next, hist, wsum, sel);
        }
    }
    if (tid == 0) {
        out[6 * c + 0] = (double)n;
        double mn_ = rmin[0], mx_ = rmax[0];
        for (int q = 1; q < CSS_THREADS / 64; ++q) {
            mn_ = fmin(mn_, rmin[q]);
            mx_ = fmax(mx_, rmax[q]);
        }
        out[6 * c + 1] = mn_;
        out[6 * c + 2] = mx_;
        out[6 * c + 3] = a;
        out[6 * c + 4] = b;
        out[6 * c + 5] = gamma;
    }
}


// The same statistics WITHOUT the grouping pass: workgroup c walks the scan's labels itself (30 k labels = 120 KB out of
// the L2, coalesced), folds the plane distance of every member into its extremes and appends the member's PP key to LDS
// (the order of the keys does not matter to an order statistic).  One launch instead of two; a cluster with more
// than CS_LDS_KEYS members raises *overflow and the caller takes the grouped path.
__device__ __forceinline__ void cs_stats_direct_body(const float *__restrict__ pts, int stride,
                                                     const float *__restrict__ pp, const int *__restrict__ labels,
                                                     int nAll, const PlaneP &P, double *__restrict__ out, int *overflow,
                                                     const int c) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wsum[CSS_THREADS / 64], sel[2];
    __shared__ double rmin[CSS_THREADS / 64], rmax[CSS_THREADS / 64];
    __shared__ unsigned keys[CS_LDS_KEYS];
    __shared__ int midx[CS_LDS_KEYS];
    __shared__ unsigned nKeys;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) nKeys = 0;
    __syncthreads();
    double mn = INFINITY, mx = -INFINITY;
    // 1. the members' point numbers, compacted into LDS: eight labels per thread and round, loaded before they are
    //    looked at (one load latency per round), one LDS atomic per wavefront and label slot
    for (int base = 0; base < nAll; base += 8 * CSS_THREADS) {
        int l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * CSS_THREADS + tid;
            l[u] = i < nAll ? labels[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool mine = l[u] == c;
            const unsigned long long bal = __ballot(mine);
            if (bal == 0ULL) continue;   // wave-uniform
            unsigned pos = 0;
            if (lane == 0) pos = atomicAdd(&nKeys, (unsigned)__popcll(bal));
            pos = __builtin_amdgcn_readfirstlane(pos) + (unsigned)__popcll(bal & ((1ULL << lane) - 1ULL));
            if (mine && pos < (unsigned)CS_LDS_KEYS) midx[pos] = base + u * CSS_THREADS + tid;
        }
    }
    __syncthreads();
    // 2. keys and plane distances of the members, all lanes busy
    {
        const int nm = min((int)nKeys, CS_LDS_KEYS);
        for (int i = tid; i < nm; i += CSS_THREADS) {
            const int m = midx[i];
            keys[i] = cs_key(pp[m]);
            const float *p = pts + (size_t)m * stride;
            // ptc @ plane[:3] + plane[3] then / norm: the float64 rounding of numpy's product
            double dist = (double)p[0] * P.n0;
            dist = fma((double)p[1], P.n1, dist);
            dist = fma((double)p[2], P.n2, dist);
            dist = (dist + P.d) / P.norm;
            mn = fmin(mn, dist);
            mx = fmax(mx, dist);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
        rmin[tid >> 6] = mn;
        rmax[tid >> 6] = mx;
    }
    __syncthreads();
    const int n = (int)nKeys;
    if (n > CS_LDS_KEYS) {   // uniform
        if (tid == 0) *reinterpret_cast<volatile int *>(overflow) = 1;   // pinned host word: a plain store, any writer writes 1
        return;
    }
    double a = 0.0, b = 0.0, gamma = 0.0;
    if (n > 0) {
        const float qf = (float)P.q;
        const float vi = (float)(n - 1) * qf;
        const float fl = floorf(vi);
        int prev = (int)fl, next = prev + 1;
        if (vi >= (float)(n - 1)) prev = next = n - 1;
        if (vi < 0.f) prev = next = 0;
        next = min(next, n - 1);
        gamma = (double)(vi - fl);
        const auto key_at = [&](int i) { return keys[i]; };
        const float af = cs_select_keys<CSS_THREADS>(key_at, n, (unsigned)prev, hist, wsum, sel);
        a = (double)af;
        b = a;
        if (next != prev) {
            const unsigned akey = cs_key(af);
            if (tid == 0) {
                sel[0] = 0;
                sel[1] = 0xffffffffu;
            }
            __syncthreads();
            unsigned cle = 0, mgt = 0xffffffffu;
            for (int i = tid; i < n; i += CSS_THREADS) {
                const unsigned k = keys[i];
                cle += k <= akey ? 1u : 0u;
                mgt = k > akey ? min(mgt, k) : mgt;
            }
            for (int o = 32; o > 0; o >>= 1) {
                cle += __shfl_xor(cle, o);
                mgt = min(mgt, (unsigned)__shfl_xor((int)mgt, o));
            }
            if (lane == 0) {
                atomicAdd(&sel[0], cle);
                atomicMin(&sel[1], mgt);
            }
            __syncthreads();
            b = sel[0] > (unsigned)next ? a : (double)cs_unkey(sel[1]);
        }
    }
    if (tid == 0) {
        out[6 * c + 0] = (double)n;
        double mn_ = rmin[0], mx_ = rmax[0];
        for (int q = 1; q < CSS_THREADS / 64; ++q) {
            mn_ = fmin(mn_, rmin[q]);
            mx_ = fmax(mx_, rmax[q]);
        }
        out[6 * c + 1] = mn_;
        out[6 * c + 2] = mx_;
        out[6 * c + 3] = a;
        out[6 * c + 4] = b;
        out[6 * c + 5] = gamma;
    }
}
__global__ __launch_bounds__(CSS_THREADS) void cs_stats_direct(const float *__restrict__ pts, int stride,
                                                              const float *__restrict__ pp,
                                                              const int *__restrict__ labels, int nAll, PlaneP P,
                                                              double *__restrict__ out, int *overflow) {
    cs_stats_direct_body(pts, stride, pp, labels, nAll, P, out, overflow, (int)blockIdx.x);
}

// a chain of scans: the scan is blockIdx.y, its pointers come from a device table
struct CSB {
    const float *pts, *pp;
    const int *labels;
    double *out;
    int *overflow;
    PlaneP P;
    int stride, nAll, n_clusters, pad;
};
__global__ __launch_bounds__(CSS_THREADS) void csb_stats(const CSB *__restrict__ tab) {
    const CSB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.n_clusters) return;
    cs_stats_direct_body(S.pts, S.stride, S.pp, S.labels, S.nAll, S.P, S.out, S.overflow, (int)blockIdx.x);
}

}  // namespace

extern "C" int modest_cluster_stats(modest_ctx *ctx, const float *pts, int n, int stride, const float *pp,
                                    const int32_t *labels, int n_clusters, const double *plane4,
                                    double quantile, double *out_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0 && n_clusters >= 0 && (stride == 3 || stride == 4), "bad sizes");
    MODEST_REQUIRE(quantile >= 0.0 && quantile <= 1.0, "quantile must be in [0,1]");
    if (n_clusters == 0) return MODEST_OK;
    MODEST_REQUIRE(pts && pp && labels && plane4 && out_host, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t b_cnt = arena_sz((size_t)2 * n_clusters * 4), b_start = arena_sz((size_t)(n_clusters + 1) * 4);
    const size_t b_mem = arena_sz((size_t)n * 4), b_out = arena_sz((size_t)n_clusters * 48);
    int rc = modest_ctx_reserve(ctx, b_cnt + b_start + b_mem + b_out);
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, (size_t)n_clusters * 48 + 64);
    if (rc) return rc;
    unsigned *cnt = reinterpret_cast<unsigned *>(ctx->scratch);
    unsigned *fill = cnt + n_clusters;
    unsigned *start = reinterpret_cast<unsigned *>(ctx->scratch + b_cnt);
    int *members = reinterpret_cast<int *>(ctx->scratch + b_cnt + b_start);
    double *d_out = reinterpret_cast<double *>(ctx->scratch + b_cnt + b_start + b_mem);
    PlaneP P;
    P.n0 = plane4[0];
    P.n1 = plane4[1];
    P.n2 = plane4[2];
    P.d = plane4[3];
    P.norm = sqrt((plane4[0] * plane4[0] + plane4[1] * plane4[1]) + plane4[2] * plane4[2]);
    P.q = quantile;
    // the usual case in one launch: every cluster's workgroup finds its members itself; the overflow word sits behind
    // the results in pinned memory
    int *h_over = reinterpret_cast<int *>(ctx->pinned + (size_t)n_clusters * 48);
    if (!getenv("MODEST_CS_GROUPED")) {
        *h_over = 0;
        cs_stats_direct<<<n_clusters, CSS_THREADS, 0, stream>>>(pts, stride, pp, labels, n, P,
                                                                reinterpret_cast<double *>(ctx->pinned), h_over);
        MODEST_HIP_CHECK(hipGetLastError());
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        if (*h_over == 0) {
            const double *hd = reinterpret_cast<const double *>(ctx->pinned);
            for (size_t i = 0; i < (size_t)n_clusters * 6; ++i) out_host[i] = hd[i];
            return MODEST_OK;
        }
    }
    if (n_clusters <= CS_GROUP_MAXC) {
        cs_group<<<1, 1024, 0, stream>>>(labels, n, n_clusters, start, members);
    } else {
        const int nb = (n + 255) / 256;
        MODEST_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)2 * n_clusters * 4, stream));
        if (n > 0) cs_count<<<nb, 256, 0, stream>>>(labels, n, n_clusters, cnt);
        cs_scan<<<1, 64, 0, stream>>>(cnt, n_clusters, start);
        if (n > 0) cs_scatter<<<nb, 256, 0, stream>>>(labels, n, n_clusters, start, fill, members);
    }
    // six doubles per cluster, written by thread 0 of each workgroup straight into pinned host memory
    (void)d_out;
    cs_stats<<<n_clusters, CSS_THREADS, 0, stream>>>(pts, stride, pp, members, start, P,
                                                     reinterpret_cast<double *>(ctx->pinned));
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    const double *h = reinterpret_cast<const double *>(ctx->pinned);
    for (size_t i = 0; i < (size_t)n_clusters * 6; ++i) out_host[i] = h[i];
    return MODEST_OK;
}

#include "mask_chain.h"
#include <cstring>
#include <vector>

int modest_cluster_stats_chain(modest_stats_chain_scan *S, int B, double quantile, hipStream_t stream) {
    MODEST_REQUIRE(S != nullptr && B >= 1 && B <= 64, "bad chain");
    MODEST_REQUIRE(quantile >= 0.0 && quantile <= 1.0, "quantile must be in [0,1]");
    std::vector<CSB> tab((size_t)B);
    int maxC = 0;
    for (int s = 0; s < B; ++s) {
        modest_stats_chain_scan &q = S[s];
        CSB &b = tab[(size_t)s];
        memset(&b, 0, sizeof(b));
        if (q.n_clusters <= 0) continue;
        MODEST_REQUIRE(q.ctx && q.pts && q.pp && q.labels && q.plane4 && q.out_host, "NULL buffer");
        int rc = modest_ctx_reserve_pinned(q.ctx, (size_t)q.n_clusters * 48 + 64);
        if (rc) return rc;
        b.pts = q.pts;
        b.pp = q.pp;
        b.labels = q.labels;
        b.out = reinterpret_cast<double *>(q.ctx->pinned);
        b.overflow = reinterpret_cast<int *>(q.ctx->pinned + (size_t)q.n_clusters * 48);
        *b.overflow = 0;
        b.P.n0 = q.plane4[0];
        b.P.n1 = q.plane4[1];
        b.P.n2 = q.plane4[2];
        b.P.d = q.plane4[3];
        b.P.norm = sqrt((q.plane4[0] * q.plane4[0] + q.plane4[1] * q.plane4[1]) + q.plane4[2] * q.plane4[2]);
        b.P.q = quantile;
        b.stride = q.stride;
        b.nAll = q.n;
        b.n_clusters = q.n_clusters;
        maxC = maxC > q.n_clusters ? maxC : q.n_clusters;
    }
    if (maxC > 0) {
        char *d = nullptr, *h = nullptr;
        const size_t bytes = tab.size() * sizeof(CSB);
        int rc = modest_ctx_chain_tab(S[0].ctx, bytes, &d);
        if (rc) return rc;
        rc = modest_ctx_stage_slot(S[0].ctx, bytes, reinterpret_cast<void **>(&h));
        if (rc) return rc;
        memcpy(h, tab.data(), bytes);
        MODEST_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
        rc = modest_ctx_stage_commit(S[0].ctx, stream);
        if (rc) return rc;
        csb_stats<<<dim3((unsigned)maxC, (unsigned)B), CSS_THREADS, 0, stream>>>(reinterpret_cast<const CSB *>(d));
        MODEST_HIP_CHECK(hipGetLastError());
    }
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int s = 0; s < B; ++s) {
        modest_stats_chain_scan &q = S[s];
        if (q.n_clusters <= 0) continue;
        const CSB &b = tab[(size_t)s];
        if (*b.overflow) {   // a cluster of more than CS_LDS_KEYS points: the grouped path, for this scan alone
            int rc = modest_cluster_stats(q.ctx, q.pts, q.n, q.stride, q.pp, q.labels, q.n_clusters, q.plane4, quantile,
                                          q.out_host, stream);
            if (rc) return rc;
            continue;
        }
        const double *hd = reinterpret_cast<const double *>(q.ctx->pinned);
        for (size_t i = 0; i < (size_t)q.n_clusters * 6; ++i) q.out_host[i] = hd[i];
    }
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_cluster_stats(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(cs_count));
}
