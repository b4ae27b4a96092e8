// PP-weighted mutual-kNN graph + DBSCAN on the implicit graph, for gfx950.
//
// Replaces precompute_affinity_matrix(..., 'radius_mutual_knn', 'l1', k, r)
// (utils/clustering_utils.py:7-60: sklearn kneighbors_graph o transpose o
// radius_neighbors_graph, then |pp_i - pp_j|) followed by
// sklearn.cluster.DBSCAN(metric='precomputed', eps, min_samples)
// (generate_mask.py:75-81).  No sparse matrix is materialised:
//
//   r2k(i)    = squared distance from i to its k-th nearest other point
//               (+inf when fewer than k lie within `radius`)
//   edge(i,j) = d2(i,j) <= min(r2k(i), r2k(j), radius^2)
//               and (double)(float)|pp_i - pp_j| <= eps
//   core(i)   = deg(i) + 1 >= min_samples          (DBSCAN counts the point itself)
//   cluster   = connected component of the core-core edges, numbered by the
//               rank of its smallest core index (sklearn's DFS visits points in
//               index order); a border point takes the smallest cluster id among
//               its adjacent cores (the first cluster that reaches it); else -1.
//
// d2 is the float64 sum dx*dx + dy*dy + dz*dz of float32 coordinates, the value
// sklearn's KDTree (float64) compares.  Region queries run on a uniform grid
// with cell edge = radius: one 64-lane wavefront per query point sweeps the 3x3
// cell rows (contiguous in the cell-sorted array, so loads are coalesced).
//
// modest_mask_cluster puts generate_mask.py:57-65 in front of it (above_plane +
// limit_range mask, ordered compaction of the kept rows, their cell counts on a
// grid fixed around limit_range) and `labels[ptc_mask] = ...` behind it, as one
// call with one synchronise in the middle (the kept count sizes the launches).
#include "common.h"
#include "compact.h"
#include "mask_pred.h"
#include <cmath>

using namespace modest;

namespace {

constexpr int CG = 128;              // grid is CG x CG cells
constexpr int CG_CELLS = CG * CG;
constexpr int WPB = 4;               // wavefronts per block
constexpr int KNN_KEYS = 1024;       // in-radius candidates of a query whose float32 keys stay in LDS between the radix passes (4 KB per wavefront)
constexpr double D_INF = __builtin_huge_val();

struct CGrid {
    double ox, oy, inv_c;
};

__device__ __forceinline__ int cg_coord(float v, double o, double inv_c) {
    double f = floor(((double)v - o) * inv_c);
    f = fmin(fmax(f, 0.0), (double)(CG - 1));
    return (int)f;
}

__global__ __launch_bounds__(1024) void cg_bbox(const float *__restrict__ xyz, int n, double c, CGrid *g,
                                                unsigned *__restrict__ zeroed, int zero_words) {
    // the cell counters of this call are cleared here (one memset launch less);
    // the next kernel on the stream is their first user
    for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zeroed[i] = 0u;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1];
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
        for (int k = 1; k < 16; ++k) {
            s[0][0] = fminf(s[0][0], s[0][k]);
            s[1][0] = fmaxf(s[1][0], s[1][k]);
            s[2][0] = fminf(s[2][0], s[2][k]);
            s[3][0] = fmaxf(s[3][0], s[3][k]);
        }
        double cx = 0.5 * ((double)s[0][0] + (double)s[1][0]);
        double cy = 0.5 * ((double)s[2][0] + (double)s[3][0]);
        if (!(cx == cx) || fabs(cx) > 1e30) cx = 0.0;
        if (!(cy == cy) || fabs(cy) > 1e30) cy = 0.0;
        g->ox = cx - 0.5 * CG * c;
        g->oy = cy - 0.5 * CG * c;
        g->inv_c = 1.0 / c;
    }
}

__global__ void cg_count(const float *__restrict__ xyz, int n, const CGrid *g, unsigned *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = cg_coord(xyz[3 * (size_t)i], g->ox, g->inv_c);
    const int cy = cg_coord(xyz[3 * (size_t)i + 1], g->oy, g->inv_c);
    atomicAdd(&cnt[cy * CG + cx], 1u);
}

// exclusive scan of `n` counters into out[0..n] by one 1024-thread workgroup, 8192 counters per
// round.  A wavefront owns 512 consecutive counters: it loads them as 8 coalesced rows, turns
// them through its LDS tile so that every lane holds 8 CONSECUTIVE counters (row stride 9 words:
// next to no bank conflicts either way), scans lane sums with shuffles, and turns the prefixes back for
// coalesced stores.  One pass over the data, one barrier per round.
// CELLS (the grid's cell counters): also writes the scatter cursors (= the prefixes), clears the
// counters behind itself when `clear_in` is set (the persistent counters of the fused mask call
// stay zero between calls) and clears the four flag words of the call.
constexpr int SCAN_IPT = 8;
template <bool CELLS>
__device__ __forceinline__ void scan_u32_body(const unsigned *in, unsigned *__restrict__ out, int n,
                                              unsigned *__restrict__ total_copy, const int *__restrict__ flag_src,
                                              unsigned *__restrict__ cursor, unsigned *clear_in, unsigned *flags) {
    __shared__ unsigned tile[16][64 * (SCAN_IPT + 1)];
    __shared__ unsigned wtot[2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned *T = tile[w];
    unsigned carry = 0;
    int buf = 0;
    for (int t0 = 0; t0 < n; t0 += 1024 * SCAN_IPT, buf ^= 1) {
        const int seg = t0 + w * 64 * SCAN_IPT;
        unsigned v[SCAN_IPT];
#pragma unroll
        for (int q = 0; q < SCAN_IPT; ++q) {
            const int e = q * 64 + lane;
            v[q] = seg + e < n ? in[seg + e] : 0u;
        }
#pragma unroll
        for (int q = 0; q < SCAN_IPT; ++q) {
            const int e = q * 64 + lane;
            T[(e / SCAN_IPT) * (SCAN_IPT + 1) + (e % SCAN_IPT)] = v[q];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned mine = 0;
#pragma unroll
        for (int q = 0; q < SCAN_IPT; ++q) {
            v[q] = T[lane * (SCAN_IPT + 1) + q];
            mine += v[q];
        }
        unsigned inc = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wtot[buf][w] = inc;
        __syncthreads();
        unsigned before = carry, all = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned x = wtot[buf][q];
            before += q < w ? x : 0u;
            all += x;
        }
        unsigned run = before + inc - mine;
#pragma unroll
        for (int q = 0; q < SCAN_IPT; ++q) {
            T[lane * (SCAN_IPT + 1) + q] = run;
            run += v[q];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < SCAN_IPT; ++q) {
            const int e = q * 64 + lane;
            const unsigned x = T[(e / SCAN_IPT) * (SCAN_IPT + 1) + (e % SCAN_IPT)];
            if (seg + e < n) {
                out[seg + e] = x;
                if (CELLS) {
                    cursor[seg + e] = x;
                    if (clear_in) clear_in[seg + e] = 0u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        carry += all;
    }
    if (CELLS && flags && threadIdx.x < 4) flags[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        out[n] = carry;
        // pinned host memory: [overflow flag, total] are there when the stream has been synchronised
        if (total_copy) {
            total_copy[0] = flag_src ? (unsigned)*flag_src : 0u;
            total_copy[1] = carry;
        }
    }
}
template <bool CELLS>
__global__ __launch_bounds__(1024) void scan_u32(const unsigned *in, unsigned *__restrict__ out, int n,
                                                 unsigned *__restrict__ total_copy = nullptr,
                                                 const int *__restrict__ flag_src = nullptr,
                                                 unsigned *__restrict__ cursor = nullptr,
                                                 unsigned *clear_in = nullptr, unsigned *flags = nullptr) {
    scan_u32_body<CELLS>(in, out, n, total_copy, flag_src, cursor, clear_in, flags);
}

// idx == NULL: pp (and intensity) are per input point.  idx != NULL (fused mask call): the input
// points are the kept rows of a scan, idx their row numbers; pp is the scan's array and the
// intensity is column 3 of the scan rows (`rows`, `stride` floats apart).
__device__ __forceinline__ void cg_scatter_body(const float *__restrict__ xyz, const float *__restrict__ pp, int n, const CGrid *g, unsigned *cursor, float4 *__restrict__ sorted, int *__restrict__ sidx, const int *__restrict__ idx, const float *__restrict__ inten_src, const float *__restrict__ rows, int stride, float *__restrict__ sortedI, const unsigned bx, const unsigned gx) {
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    const int cell = cg_coord(y, g->oy, g->inv_c) * CG + cg_coord(x, g->ox, g->inv_c);
    const unsigned slot = atomicAdd(&cursor[cell], 1u);
    const int src = idx ? idx[i] : i;
    sorted[slot] = make_float4(x, y, z, pp[src]);
    sidx[slot] = i;
    if (sortedI) sortedI[slot] = rows ? rows[(size_t)src * stride + 3] : inten_src[src];
}
__global__  void cg_scatter(const float *__restrict__ xyz, const float *__restrict__ pp, int n, const CGrid *g, unsigned *cursor, float4 *__restrict__ sorted, int *__restrict__ sidx, const int *__restrict__ idx, const float *__restrict__ inten_src, const float *__restrict__ rows, int stride, float *__restrict__ sortedI) {
    cg_scatter_body(xyz, pp, n, g, cursor, sorted, sidx, idx, inten_src, rows, stride, sortedI, blockIdx.x, gridDim.x);
}

// Fused first kernel of modest_mask_cluster: above_plane + range mask of every scan row, ordered
// compaction of the kept rows, their cell counts in the fixed grid G, labels = -1 everywhere.
__device__ __forceinline__ void mask_count_kernel_body(const float *__restrict__ pts, int n, int stride, MaskParams P, CGrid G, CGrid *__restrict__ g, unsigned *cnt, int *__restrict__ labels, float *__restrict__ kept, int *__restrict__ kept_idx, unsigned long long *state, int *n_kept, const unsigned bx, const unsigned gx) {
    const unsigned blk = compact_ticket(state);
    const long long i = (long long)blk * 1024 + threadIdx.x;
    if (blk == 0 && threadIdx.x == 0) *g = G;
    bool keep = false;
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float *p = pts + i * stride;
        x = p[0];
        y = p[1];
        z = p[2];
        keep = mask_keep(P, x, y, z);
        labels[i] = -1;
        if (keep) atomicAdd(&cnt[cg_coord(y, G.oy, G.inv_c) * CG + cg_coord(x, G.ox, G.inv_c)], 1u);
    }
    const unsigned long long dst = compact_offset(keep, blk, gx, state, n_kept);
    if (keep) {
        kept[3 * dst + 0] = x;
        kept[3 * dst + 1] = y;
        kept[3 * dst + 2] = z;
        kept_idx[dst] = (int)i;
    }
}
__global__ __launch_bounds__(1024) void mask_count_kernel(const float *__restrict__ pts, int n, int stride, MaskParams P, CGrid G, CGrid *__restrict__ g, unsigned *cnt, int *__restrict__ labels, float *__restrict__ kept, int *__restrict__ kept_idx, unsigned long long *state, int *n_kept) {
    mask_count_kernel_body(pts, n, stride, P, G, g, cnt, labels, kept, kept_idx, state, n_kept, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ double dist2(const float4 &a, const float4 &b) {
    const double dx = (double)a.x - (double)b.x;
    const double dy = (double)a.y - (double)b.y;
    const double dz = (double)a.z - (double)b.z;
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// Sweep helper: the three cell rows around (cx,cy) as contiguous [s,e) ranges; rows outside the
// grid are empty.  Fixed size and constant indices only: a row count with indexed stores put the
// struct into scratch memory, and that alone cost every sweep kernel 25 us.
struct Rows {
    static constexpr int n = 3;
    unsigned s[3], e[3];
};
__device__ __forceinline__ Rows rows_of(const float4 &q, const CGrid *g, const unsigned *__restrict__ start) {
    const int cx = cg_coord(q.x, g->ox, g->inv_c), cy = cg_coord(q.y, g->oy, g->inv_c);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, CG - 1);
    Rows r;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = cy - 1 + d;
        const bool ok = yy >= 0 && yy < CG;
        r.s[d] = ok ? start[yy * CG + x0] : 0u;
        r.e[d] = ok ? start[yy * CG + x1 + 1] : 0u;
    }
    return r;
}

// ---- pass A: k-th neighbour squared distance ------------------------------------------
// Wave-level radix select.  (float)d2 is a monotone map of the float64 distances, so the k-th
// smallest float32 key belongs to the k-th smallest distance: four byte passes over the float32
// keys find it (instead of eight over the float64 bit pattern; the distances are recomputed in
// every pass), and the exact float64 value is then taken from the candidates that share the key
// -- one more pass when the key is unique, a min/count loop over the ties otherwise.
__device__ __forceinline__ void knn_kth_kernel_body(const float4 *__restrict__ sorted, int n, const CGrid *g, const unsigned *__restrict__ start, int k, double r2, double *__restrict__ kthS, const unsigned bx, const unsigned gx) {
    __shared__ unsigned hist_all[WPB][256];
    __shared__ unsigned keys_all[WPB][KNN_KEYS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = bx * WPB + w;
    if (s >= n) return;   // whole wave exits together
    unsigned *hist = hist_all[w];
    unsigned *keys = keys_all[w];
    const float4 q = sorted[s];
    const Rows R = rows_of(q, g, start);
    unsigned prefix = 0, mask = 0, cntF = 0;
    int kk = k - 1;
    bool enough = true;
    unsigned nkeys = 0;   // in-radius candidates seen by the first walk (their float32 keys are in LDS while there are at most KNN_KEYS)
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        if (shift == 24) {
            // the first walk computes every distance once and keeps the keys of the candidates inside the radius: the other three
            // byte passes read those from LDS instead of walking the rows (and their float64 distances) again -- in chains of 16
            // scans the kernel is bound by that arithmetic, not by the latency of the walk (a single scan was: round 3)
    #pragma unroll
    for (int r = 0; r < R.n; ++r)
            for (unsigned base = R.s[r]; base < R.e[r]; base += 64) {   // wave-uniform trip count
                const unsigned j = base + lane;
                bool inr = false;
                unsigned key = 0;
                if (j < R.e[r] && (int)j != s) {
                    const double d2 = dist2(q, sorted[j]);
                    inr = d2 <= r2;
                    key = __float_as_uint((float)d2);
                }
                const unsigned long long bal = __ballot(inr);
                if (inr) {
                    atomicAdd(&hist[(key >> 24) & 255u], 1u);
                    const unsigned pos = nkeys + (unsigned)__popcll(bal & ((1ULL << lane) - 1ULL));
                    if (pos < (unsigned)KNN_KEYS) keys[pos] = key;
                }
                nkeys += (unsigned)__popcll(bal);
            }
        } else if (nkeys <= (unsigned)KNN_KEYS) {
            for (unsigned i = lane; i < nkeys; i += 64) {
                const unsigned key = keys[i];
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
        } else {
    #pragma unroll
    for (int r = 0; r < R.n; ++r)
            for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
                if ((int)j == s) continue;
                const double d2 = dist2(q, sorted[j]);
                if (d2 <= r2) {
                    const unsigned key = __float_as_uint((float)d2);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // lane l owns bins 4l..4l+3
        const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2],
                       c3 = hist[4 * lane + 3];
        const unsigned mine = c0 + c1 + c2 + c3;
        unsigned inc = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        const unsigned total = __shfl(inc, 63);
        if (shift == 24 && total < (unsigned)k) {
            enough = false;   // fewer than k neighbours inside the radius
            break;
        }
        const unsigned long long bal = __ballot(inc > (unsigned)kk);
        const int owner = __ffsll((long long)bal) - 1;
        const unsigned excl = __shfl(inc - mine, owner);
        const unsigned o0 = __shfl(c0, owner), o1 = __shfl(c1, owner), o2 = __shfl(c2, owner),
                       o3 = __shfl(c3, owner);
        int rem = kk - (int)excl;
        int bin = 4 * owner;
        cntF = o0;
        if (rem >= (int)o0) {
            rem -= o0;
            ++bin;
            cntF = o1;
            if (rem >= (int)o1) {
                rem -= o1;
                ++bin;
                cntF = o2;
                if (rem >= (int)o2) {
                    rem -= o2;
                    ++bin;
                    cntF = o3;
                }
            }
        }
        kk = rem;
        prefix |= (unsigned)bin << shift;
        mask |= 255u << shift;
        __builtin_amdgcn_wave_barrier();
    }
    double result = D_INF;
    if (enough) {
        // kk = rank (0-based) among the cntF in-radius candidates whose float32 key is `prefix`
        double lo = -1.0;
        for (;;) {
            double m = D_INF;
        #pragma unroll
    for (int r = 0; r < R.n; ++r)
                for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
                    if ((int)j == s) continue;
                    const double d2 = dist2(q, sorted[j]);
                    if (d2 <= r2 && d2 > lo && __float_as_uint((float)d2) == prefix) m = fmin(m, d2);
                }
            for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o));
            if (cntF == 1) {
                result = m;
                break;
            }
            unsigned c = 0;
        #pragma unroll
    for (int r = 0; r < R.n; ++r)
                for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64)
                    if ((int)j != s && dist2(q, sorted[j]) == m) ++c;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (kk < (int)c || c == 0) {
                result = m;
                break;
            }
            kk -= (int)c;
            lo = m;
        }
    }
    if (lane == 0) kthS[s] = result;
}
__global__ __launch_bounds__(64 * WPB) void knn_kth_kernel(const float4 *__restrict__ sorted, int n, const CGrid *g, const unsigned *__restrict__ start, int k, double r2, double *__restrict__ kthS) {
    knn_kth_kernel_body(sorted, n, g, start, k, r2, kthS, blockIdx.x, gridDim.x);
}

// Graph / weight variants of precompute_affinity_matrix (utils/clustering_utils.py:16-56):
//   use_knn = 1: radius_mutual_knn (default)   edge <=> d2 <= min(r2_k(i), r2_k(j), radius^2)
//   use_knn = 0: radius                         edge <=> d2 <= radius^2
//   affinity 0 'l1'             |pp_i - pp_j|                          (float32, as numpy)
//            1 'exp'            exp((pp_i - pp_j)^2)                   (float32 square, float32 exp)
//            2 '3d_l2_distance' norm of the 4-column row difference    (float32: ((dx2+dy2)+dz2)+di2, sqrt)
//                               -- the reference passes the (n,4) scan rows, so intensity takes part
struct EdgeP {
    double r2, eps;
    int use_knn, affinity;
    const float *inten;   // intensity in sorted order (affinity 2 only)
};

__device__ __forceinline__ bool weight_ok(const float4 &q, const float4 &c, const EdgeP &ep, int s, unsigned j) {
    float w;
    if (ep.affinity == 0) {
        w = fabsf(q.w - c.w);
    } else if (ep.affinity == 1) {
        const float d = q.w - c.w;
        w = (float)exp((double)(d * d));   // float32 exp evaluated in float64 and rounded once
    } else {
        const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z, di = ep.inten[s] - ep.inten[j];
        w = sqrtf(((dx * dx + dy * dy) + dz * dz) + di * di);
    }
    return (double)w <= ep.eps;
}

__device__ __forceinline__ bool edge_ok(const float4 &q, double kq, const float4 &c, double kc, const EdgeP &ep,
                                        int s, unsigned j) {
    const double d2 = dist2(q, c);
    const double lim = ep.use_knn ? fmin(fmin(kq, kc), ep.r2) : ep.r2;
    if (!(d2 <= lim)) return false;
    return weight_ok(q, c, ep, s, j);
}

// ---- pass B: degrees / core flags ------------------------------------------------
__global__ __launch_bounds__(64 * WPB) void degree_kernel(const float4 *__restrict__ sorted, int n,
                                                          const CGrid *g,
                                                          const unsigned *__restrict__ start,
                                                          const double *__restrict__ kthS, EdgeP ep, int min_samples,
                                                          unsigned char *__restrict__ coreS) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const Rows R = rows_of(q, g, start);
    unsigned cnt = 0;
#pragma unroll
    for (int r = 0; r < R.n; ++r)
        for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
            if ((int)j == s) continue;
            if (edge_ok(q, kq, sorted[j], kthS[j], ep, s, j)) ++cnt;
        }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) coreS[s] = ((int)cnt + 1 >= min_samples) ? 1 : 0;
}

// ---- pass C: union-find over core-core edges (root = smallest original index) ----
__device__ __forceinline__ int uf_load(int *p, int i) {
    return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// find with path halving: every visited node is re-pointed at its grandparent (a plain
// racy store is fine: parents only ever move towards the root, i.e. to smaller indices)
__device__ __forceinline__ int uf_find(int *parent, int x) {
    int p = uf_load(parent, x);
    while (p != x) {
        const int gp = uf_load(parent, p);
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_unite(int *parent, int a, int b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        // a < b: hang root b under a
        const int old = atomicCAS(parent + b, b, a);
        if (old == b) return;
    }
}

__global__ void uf_init(int *parent, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}

// ECL-CC style initialisation: every core point first hangs under its smallest-index core
// neighbour (no atomics: one writer per entry, and parent[v] <= v keeps the forest acyclic).
__global__ __launch_bounds__(64 * WPB) void hook_min_kernel(const float4 *__restrict__ sorted, int n,
                                                            const CGrid *g,
                                                            const unsigned *__restrict__ start,
                                                            const double *__restrict__ kthS,
                                                            const unsigned char *__restrict__ coreS,
                                                            const int *__restrict__ sidx, EdgeP ep, int *parent) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    if (!coreS[s]) return;
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const int me = sidx[s];
    const Rows R = rows_of(q, g, start);
    int best = me;
#pragma unroll
    for (int r = 0; r < R.n; ++r)
        for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
            if ((int)j == s || !coreS[j]) continue;
            const int other = sidx[j];
            if (other < best && edge_ok(q, kq, sorted[j], kthS[j], ep, s, j)) best = other;
        }
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    if (lane == 0) parent[me] = best;
}

__global__ __launch_bounds__(64 * WPB) void union_kernel(const float4 *__restrict__ sorted, int n,
                                                         const CGrid *g,
                                                         const unsigned *__restrict__ start,
                                                         const double *__restrict__ kthS,
                                                         const unsigned char *__restrict__ coreS,
                                                         const int *__restrict__ sidx, EdgeP ep, int *parent) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    if (!coreS[s]) return;
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const int me = sidx[s];
    const Rows R = rows_of(q, g, start);
#pragma unroll
    for (int r = 0; r < R.n; ++r)
        for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
            if ((int)j <= s) continue;          // each unordered pair once
            if (!coreS[j]) continue;
            if (edge_ok(q, kq, sorted[j], kthS[j], ep, s, j)) uf_unite(parent, me, sidx[j]);
        }
}

__global__ void compress_kernel(int *parent, const unsigned char *__restrict__ coreS,
                                const int *__restrict__ sidx, int n, int *__restrict__ root,
                                unsigned *__restrict__ isroot) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = sidx[s];
    const int r = uf_find(parent, i);
    root[i] = r;
    isroot[i] = (coreS[s] && r == i) ? 1u : 0u;
}

__global__ __launch_bounds__(64 * WPB) void label_kernel(const float4 *__restrict__ sorted, int n,
                                                         const CGrid *g,
                                                         const unsigned *__restrict__ start,
                                                         const double *__restrict__ kthS,
                                                         const unsigned char *__restrict__ coreS,
                                                         const int *__restrict__ sidx,
                                                         const int *__restrict__ root,
                                                         const unsigned *__restrict__ rank, EdgeP ep, int *__restrict__ labels,
                                                         const int *__restrict__ oidx) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    const int me = sidx[s];
    if (coreS[s]) {
        if (lane == 0) labels[oidx ? oidx[me] : me] = (int)rank[root[me]];
        return;
    }
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const Rows R = rows_of(q, g, start);
    int best = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < R.n; ++r)
        for (unsigned j = R.s[r] + lane; j < R.e[r]; j += 64) {
            if ((int)j == s || !coreS[j]) continue;
            if (edge_ok(q, kq, sorted[j], kthS[j], ep, s, j)) best = min(best, root[sidx[j]]);
        }
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    if (lane == 0) labels[oidx ? oidx[me] : me] = (best == 0x7fffffff) ? -1 : (int)rank[best];
}

// ---- explicit adjacency (the fast path) ----------------------------------------------
// An edge needs j in kNN_k(i), so a point has at most k (+ exact ties) edges: the degree pass
// stores them (cell-sorted indices) in a fixed-stride table and every later pass walks the
// table instead of re-evaluating the float64 predicate.  More than ADJ edges (only possible
// with dozens of exactly tied k-th distances) raises `overflow` and the host re-runs the
// recomputing kernels below.
constexpr int ADJ = 128;
constexpr int AG = 16;    // lanes that share one adjacency row in the passes below

// ORIG (the radius graphs): the rows hold ORIGINAL point numbers (sidx[j]) and a non-core point gets
// the parent UF_NONE, so that the hooking / union / label passes below need one dependent load per
// edge (parent or root of the neighbour) instead of three (core flag, point number, parent).
constexpr int UF_NONE = 0x7fffffff;
template <bool ORIG>
__device__ __forceinline__ void degree_adj_kernel_body(const float4 *__restrict__ sorted, int n, const CGrid *g,
                                                       const unsigned *__restrict__ start,
                                                       const double *__restrict__ kthS, const EdgeP &ep, int min_samples,
                                                       unsigned char *__restrict__ coreS, int *__restrict__ deg,
                                                       int *__restrict__ adj, int *overflow,
                                                       const int *__restrict__ sidx, int *__restrict__ parent,
                                                       const unsigned bx) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = bx * WPB + w;
    if (s >= n) return;
    if (!ORIG && lane == 0) parent[sidx[s]] = sidx[s];   // uf_init, one launch less (sidx is a permutation)
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const Rows R = rows_of(q, g, start);
    unsigned total = 0;
    // two batches of 64 candidates per trip, all four loads issued (on clamped indices) before the first is used: a row of three
    // 2 m cells holds a few hundred kept points, and one batch per trip was a chain of dependent load rounds per query
    // (179 -> 153 us per chain of 16 scans)
#pragma unroll
    for (int r = 0; r < R.n; ++r)
        for (unsigned base = R.s[r]; base < R.e[r]; base += 128) {   // wave-uniform trip count
            const unsigned j0 = base + lane, j1 = j0 + 64;
            const bool in0 = j0 < R.e[r], in1 = j1 < R.e[r];
            const unsigned a0 = in0 ? j0 : (unsigned)s, a1 = in1 ? j1 : (unsigned)s;
            const float4 c0 = sorted[a0], c1 = sorted[a1];
            const double k0 = kthS[a0], k1 = kthS[a1];
            const bool e0 = in0 && (int)j0 != s && edge_ok(q, kq, c0, k0, ep, s, j0);
            const bool e1 = in1 && (int)j1 != s && edge_ok(q, kq, c1, k1, ep, s, j1);
            const unsigned long long b0 = __ballot(e0), b1 = __ballot(e1);
            const unsigned p0 = total + __popcll(b0 & ((1ULL << lane) - 1ULL));
            const unsigned p1 = total + __popcll(b0) + __popcll(b1 & ((1ULL << lane) - 1ULL));
            if (e0 && p0 < (unsigned)ADJ) adj[(size_t)s * ADJ + p0] = ORIG ? sidx[j0] : (int)j0;
            if (e1 && p1 < (unsigned)ADJ) adj[(size_t)s * ADJ + p1] = ORIG ? sidx[j1] : (int)j1;
            total += __popcll(b0) + __popcll(b1);
        }
    if (lane == 0) {
        deg[s] = (int)min(total, (unsigned)ADJ);
        coreS[s] = ((int)total + 1 >= min_samples) ? 1 : 0;
        if (ORIG) parent[sidx[s]] = ((int)total + 1 >= min_samples) ? sidx[s] : UF_NONE;
        if (total > (unsigned)ADJ) atomicOr(overflow, 1);
    }
}
template <bool ORIG>
__global__ __launch_bounds__(64 * WPB) void degree_adj_kernel(const float4 *__restrict__ sorted, int n,
                                                              const CGrid *g,
                                                              const unsigned *__restrict__ start,
                                                              const double *__restrict__ kthS, EdgeP ep, int min_samples,
                                                              unsigned char *__restrict__ coreS,
                                                              int *__restrict__ deg, int *__restrict__ adj,
                                                              int *overflow, const int *__restrict__ sidx,
                                                              int *__restrict__ parent) {
    degree_adj_kernel_body<ORIG>(sorted, n, g, start, kthS, ep, min_samples, coreS, deg, adj, overflow, sidx, parent, blockIdx.x);
}

// one round of min-root hooking: the root of every core point is hung under the smallest root
// found among its core neighbours (parents only decrease, so the forest stays acyclic)
__global__ void hook_adj_kernel(int n, int stride, const unsigned char *__restrict__ coreS, const int *__restrict__ deg,
                                const int *__restrict__ adj, const int *__restrict__ sidx, int *parent) {
    // AG lanes per point: the row is read coalesced and the coreS -> sidx -> parent chains of the
    // edges run side by side (one thread per point made this 40+ dependent loads deep, and a scan
    // has too few points to hide that with occupancy)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    const bool on = s < n && coreS[s];
    int rme = 0, m = 0x7fffffff;
    if (on) {
        rme = parent[sidx[s]];
        m = rme;
        const int *row = adj + (size_t)s * stride;
        const int d = deg[s];
        for (int e = sub; e < d; e += AG) {
            const int j = row[e];
            if (coreS[j]) m = min(m, parent[sidx[j]]);
        }
    }
    for (int o = AG / 2; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if (on && sub == 0 && m < rme) atomicMin(parent + rme, m);
}

__global__ void flatten_kernel(int *parent, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = uf_load(parent, i);
    for (;;) {
        const int p = uf_load(parent, r);
        if (p == r) break;
        r = p;
    }
    __hip_atomic_store(parent + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exact clean-up: unite whatever the hooking rounds left apart (pairs that already share a
// parent are skipped with two plain loads)
__global__ void union_adj_kernel(int n, int stride, const unsigned char *__restrict__ coreS, const int *__restrict__ deg,
                                 const int *__restrict__ adj, const int *__restrict__ sidx, int *parent) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    if (s >= n || !coreS[s]) return;
    const int me = sidx[s];
    const int *row = adj + (size_t)s * stride;
    const int d = deg[s];
    for (int e = sub; e < d; e += AG) {
        const int j = row[e];
        if (j <= s || !coreS[j]) continue;
        const int other = sidx[j];
        if (uf_load(parent, me) != uf_load(parent, other)) uf_unite(parent, me, other);
    }
}

__global__ void label_adj_kernel(int n, int stride, const unsigned char *__restrict__ coreS, const int *__restrict__ deg,
                                 const int *__restrict__ adj, const int *__restrict__ sidx,
                                 const int *__restrict__ root, const unsigned *__restrict__ rank,
                                 int *__restrict__ labels, const int *__restrict__ oidx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    const bool on = s < n;
    const bool core = on && coreS[s];
    int best = 0x7fffffff;
    if (on && !core) {
        const int *row = adj + (size_t)s * stride;
        const int d = deg[s];
        for (int e = sub; e < d; e += AG) {
            const int j = row[e];
            if (coreS[j]) best = min(best, root[sidx[j]]);
        }
    }
    for (int o = AG / 2; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    if (!on || sub) return;
    const int me = sidx[s];
    labels[oidx ? oidx[me] : me] = core ? (int)rank[root[me]] : (best == 0x7fffffff) ? -1 : (int)rank[best];
}

// ---- the same passes over rows of original point numbers (degree_adj_kernel<true>) ----------
__device__ __forceinline__ void hook_orig_kernel_body(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, int *parent, const unsigned bx, const unsigned gx) {
    const int t = bx * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    const bool on = s < n && coreS[s];
    int rme = 0, m = UF_NONE;
    if (on) {
        rme = parent[sidx[s]];
        m = rme;
        const int *row = adj + (size_t)s * ADJ;
        const int d = deg[s];
        for (int e = sub; e < d; e += AG) m = min(m, parent[row[e]]);   // UF_NONE for a non-core neighbour
    }
    for (int o = AG / 2; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if (on && sub == 0 && m < rme) atomicMin(parent + rme, m);
}
__global__  void hook_orig_kernel(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, int *parent) {
    hook_orig_kernel_body(n, coreS, deg, adj, sidx, parent, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void flatten_orig_kernel_body(int *parent, int n, const unsigned bx, const unsigned gx) {
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = uf_load(parent, i);
    if (r == UF_NONE) return;
    for (;;) {
        const int p = uf_load(parent, r);
        if (p == r) break;
        r = p;
    }
    __hip_atomic_store(parent + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__  void flatten_orig_kernel(int *parent, int n) {
    flatten_orig_kernel_body(parent, n, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void union_orig_kernel_body(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, int *parent, const unsigned bx, const unsigned gx) {
    const int t = bx * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    if (s >= n || !coreS[s]) return;
    const int me = sidx[s];
    const int *row = adj + (size_t)s * ADJ;
    const int d = deg[s];
    for (int e = sub; e < d; e += AG) {
        const int other = row[e];
        if (other <= me) continue;   // every core-core edge is in both rows
        const int po = uf_load(parent, other);
        if (po == UF_NONE) continue;
        if (uf_load(parent, me) != po) uf_unite(parent, me, other);
    }
}
__global__  void union_orig_kernel(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, int *parent) {
    union_orig_kernel_body(n, coreS, deg, adj, sidx, parent, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void compress_orig_kernel_body(int *parent, const unsigned char *__restrict__ coreS, const int *__restrict__ sidx, int n, int *__restrict__ root, unsigned *__restrict__ isroot, const unsigned bx, const unsigned gx) {
    const int s = bx * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = sidx[s];
    const int r = coreS[s] ? uf_find(parent, i) : UF_NONE;
    root[i] = r;
    isroot[i] = r == i ? 1u : 0u;
}
__global__  void compress_orig_kernel(int *parent, const unsigned char *__restrict__ coreS, const int *__restrict__ sidx, int n, int *__restrict__ root, unsigned *__restrict__ isroot) {
    compress_orig_kernel_body(parent, coreS, sidx, n, root, isroot, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void label_orig_kernel_body(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, const int *__restrict__ root, const unsigned *__restrict__ rank, int *__restrict__ labels, const int *__restrict__ oidx, const unsigned bx, const unsigned gx) {
    const int t = bx * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    const bool on = s < n;
    const bool core = on && coreS[s];
    int best = UF_NONE;
    if (on && !core) {
        const int *row = adj + (size_t)s * ADJ;
        const int d = deg[s];
        for (int e = sub; e < d; e += AG) best = min(best, root[row[e]]);   // UF_NONE for a non-core neighbour
    }
    for (int o = AG / 2; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    if (!on || sub) return;
    const int me = sidx[s];
    labels[oidx ? oidx[me] : me] = core ? (int)rank[root[me]] : (best == UF_NONE) ? -1 : (int)rank[best];
}
__global__  void label_orig_kernel(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ deg, const int *__restrict__ adj, const int *__restrict__ sidx, const int *__restrict__ root, const unsigned *__restrict__ rank, int *__restrict__ labels, const int *__restrict__ oidx) {
    label_orig_kernel_body(n, coreS, deg, adj, sidx, root, rank, labels, oidx, blockIdx.x, gridDim.x);
}

// ---- k-NN graphs without a radius bound (neighbor_type knn / sym_knn / mutual_knn) ------------
// The grid search grows a square of cells around the query: after all cells with Chebyshev
// distance <= R have been scanned every unseen point is farther than R*c (its cell differs by at
// least R+1 in x or y; clamped border cells only hold points that are farther still).

// points in the (2R+1)^2 cells around (cx, cy), summed by the wavefront from the prefix table
__device__ __forceinline__ unsigned square_count(const unsigned *__restrict__ start, int cx, int cy, int R, int lane) {
    const int x0 = max(cx - R, 0), x1 = min(cx + R, CG - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, CG - 1);
    unsigned c = 0;
    for (int yy = y0 + lane; yy <= y1; yy += 64) c += start[yy * CG + x1 + 1] - start[yy * CG + x0];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    return c;
}

// k-th smallest squared distance (self excluded) among the points of the square: wave-level radix
// select on the float64 bit pattern, 8 byte passes (same scheme as knn_kth_kernel)
__device__ double square_kth(const float4 *__restrict__ sorted, const unsigned *__restrict__ start, const float4 &q,
                             int s, int cx, int cy, int R, int k, unsigned *hist, int lane) {
    const int x0 = max(cx - R, 0), x1 = min(cx + R, CG - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, CG - 1);
    unsigned long long prefix = 0, mask = 0;
    int kk = k - 1;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int yy = y0; yy <= y1; ++yy) {
            const unsigned rs = start[yy * CG + x0], re = start[yy * CG + x1 + 1];
            for (unsigned j = rs + lane; j < re; j += 64) {
                if ((int)j == s) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(dist2(q, sorted[j]));
                if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const unsigned mine = c0 + c1 + c2 + c3;
        unsigned inc = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        const unsigned long long bal = __ballot(inc > (unsigned)kk);
        const int owner = __ffsll((long long)bal) - 1;
        const unsigned excl = __shfl(inc - mine, owner);
        const unsigned o0 = __shfl(c0, owner), o1 = __shfl(c1, owner), o2 = __shfl(c2, owner);
        int rem = kk - (int)excl, bin = 4 * owner;
        if (rem >= (int)o0) {
            rem -= o0;
            ++bin;
            if (rem >= (int)o1) {
                rem -= o1;
                ++bin;
                if (rem >= (int)o2) {
                    rem -= o2;
                    ++bin;
                }
            }
        }
        kk = rem;
        prefix |= (unsigned long long)bin << shift;
        mask |= 255ULL << shift;
        __builtin_amdgcn_wave_barrier();
    }
    return __longlong_as_double((long long)prefix);
}

// exact squared distance to the k-th nearest OTHER point (the host guarantees n > k)
__global__ __launch_bounds__(64 * WPB) void knn_kth_unbounded_kernel(const float4 *__restrict__ sorted, int n,
                                                                     const CGrid *g,
                                                                     const unsigned *__restrict__ start, int k,
                                                                     double c, double *__restrict__ kthS) {
    __shared__ unsigned hist_all[WPB][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    const float4 q = sorted[s];
    const int cx = cg_coord(q.x, g->ox, g->inv_c), cy = cg_coord(q.y, g->oy, g->inv_c);
    int R = 1;
    while (R < CG && square_count(start, cx, cy, R, lane) < (unsigned)k + 1u) R = min(2 * R, CG);
    double kth = square_kth(sorted, start, q, s, cx, cy, R, k, hist_all[w], lane);
    const double reach = ((double)R - 1e-6) * c;   // everything closer than this has been seen
    if (R < CG && !(kth <= reach * reach)) {
        const int R2 = min(CG, (int)ceil(sqrt(kth) / c + 1e-6));
        kth = square_kth(sorted, start, q, s, cx, cy, R2, k, hist_all[w], lane);
    }
    if (lane == 0) kthS[s] = kth;
}

// out-neighbours of every point inside its own k-NN ball (d2 <= kth_i), weight <= eps;
// mode 1 (mutual_knn) additionally requires d2 <= kth_j.  adj rows have `stride` slots.
__global__ __launch_bounds__(64 * WPB) void adj_knn_kernel(const float4 *__restrict__ sorted, int n, const CGrid *g,
                                                           const unsigned *__restrict__ start,
                                                           const double *__restrict__ kthS, EdgeP ep, double c,
                                                           int mutual, int stride, int *__restrict__ deg,
                                                           int *__restrict__ adj, int *overflow) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= n) return;
    const float4 q = sorted[s];
    const double kq = kthS[s];
    const int cx = cg_coord(q.x, g->ox, g->inv_c), cy = cg_coord(q.y, g->oy, g->inv_c);
    const int R = min(CG, (int)ceil(sqrt(kq) / c + 1e-6));
    const int x0 = max(cx - R, 0), x1 = min(cx + R, CG - 1);
    unsigned total = 0;
    for (int yy = max(cy - R, 0); yy <= min(cy + R, CG - 1); ++yy) {
        const unsigned rs = start[yy * CG + x0], re = start[yy * CG + x1 + 1];
        for (unsigned base = rs; base < re; base += 64) {   // wave-uniform trip count
            const unsigned j = base + lane;
            bool e = false;
            if (j < re && (int)j != s) {
                const float4 cj = sorted[j];
                const double d2 = dist2(q, cj);
                e = d2 <= kq && (!mutual || d2 <= kthS[j]) && weight_ok(q, cj, ep, s, j);
            }
            const unsigned long long bal = __ballot(e);
            const unsigned pos = total + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (e && pos < (unsigned)stride) adj[(size_t)s * stride + pos] = (int)j;
            total += __popcll(bal);
        }
    }
    if (lane == 0) {
        deg[s] = (int)min(total, (unsigned)stride);
        if (total > (unsigned)stride) atomicOr(overflow, 1);
    }
}

// sym_knn: graph + graph.T -- append i to the row of every out-neighbour j that does not hold it yet
// (i is in j's own k-NN ball iff d2 <= kth_j; the weight test is symmetric)
__global__ void adj_symmetrize_kernel(const float4 *__restrict__ sorted, int n, const double *__restrict__ kthS,
                                      int stride, const int *__restrict__ degOut, int *deg, int *adj,
                                      int *overflow) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float4 q = sorted[s];
    for (int e = 0; e < degOut[s]; ++e) {
        const int j = adj[(size_t)s * stride + e];
        if (dist2(q, sorted[j]) <= kthS[j]) continue;   // already an out-neighbour of j
        const int pos = atomicAdd(&deg[j], 1);
        if (pos < stride) adj[(size_t)j * stride + pos] = s;
        else atomicOr(overflow, 1);
    }
}

__global__ void core_from_deg_kernel(int n, const int *__restrict__ deg, int stride, int min_samples,
                                     unsigned char *__restrict__ coreS, int *deg_clamped) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    coreS[s] = (deg[s] + 1 >= min_samples) ? 1 : 0;
    deg_clamped[s] = min(deg[s], stride);
}

// Directed graph (neighbor_type knn).  sklearn's sequential DBSCAN labels a point with the first
// cluster that reaches it; clusters are seeded in index order and expand along the rows (out-
// neighbours) of core points only.  That is: label(v) = the smallest ORIGINAL index of a core point
// that reaches v through core points -- a fixpoint of min-propagation along core out-edges
// (a core point with a smaller index reaching the minimiser would reach v too).
__global__ void dir_init_kernel(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ sidx,
                                int *__restrict__ L) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) L[s] = coreS[s] ? sidx[s] : 0x7fffffff;
}
__global__ void dir_push_kernel(int n, int stride, const unsigned char *__restrict__ coreS,
                                const int *__restrict__ deg, const int *__restrict__ adj, int *L, int *changed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / AG, sub = t % AG;
    if (s >= n || !coreS[s]) return;
    const int mine = __hip_atomic_load(L + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ch = false;
    const int d = deg[s];
    for (int e = sub; e < d; e += AG) {
        const int j = adj[(size_t)s * stride + e];
        if (atomicMin(L + j, mine) > mine) ch = true;
    }
    if (ch) *changed = 1;
}
__global__ void dir_roots_kernel(int n, const unsigned char *__restrict__ coreS, const int *__restrict__ sidx,
                                 const int *__restrict__ L, int *__restrict__ root, unsigned *__restrict__ isroot) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = sidx[s];
    root[i] = L[s];
    isroot[i] = (coreS[s] && L[s] == i) ? 1u : 0u;
}
__global__ void dir_label_kernel(int n, const int *__restrict__ root, const unsigned *__restrict__ rank,
                                 int *__restrict__ labels, const int *__restrict__ oidx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) labels[oidx ? oidx[i] : i] = (root[i] == 0x7fffffff) ? -1 : (int)rank[root[i]];
}

__global__ void scatter_kth(const double *__restrict__ kthS, const int *__restrict__ sidx, int n,
                            double *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[sidx[s]] = kthS[s];
}

}  // namespace

// Rounds of min-root hooking before the exact union pass.  Measured on a 9 k-point scan
// (hook + flatten 14 us per round): union_adj takes 305 / 220 / 59 / 19 us after 0 / 1 / 2 / 3 rounds.
static int hook_rounds() {   // hooking rounds ahead of the exact union pass (accelerators only: the union pass makes any number exact)
    static const int v = [] {
        const char *e = getenv("MODEST_HOOK_ROUNDS");
        return e ? atoi(e) : 3;
    }();
    return v;
}
#define HOOK_ROUNDS hook_rounds()

namespace {

// Input of the fused mask call: the kept rows were compacted, binned into the grid `G` (a fixed grid
// around limit_range, no bounding-box pass) and counted per cell by mask_count_kernel.
struct PreGrid {
    const int *idx;      // [dev] row number of every kept point
    const float *rows;   // [dev] the scan rows (intensity = column 3)
    int stride;
    CGrid *g;            // [dev] grid descriptor, written by mask_count_kernel
    unsigned *cnt;       // [dev] persistent per-cell counters, left zeroed by the cell scan
};

// more than ADJ edges at some point (dozens of exactly tied k-th distances): the passes that re-walk the
// cell rows instead of reading adjacency rows
void cluster_overflow_path(const float4 *sorted, int n, const CGrid *g, const unsigned *start, const double *kthS,
                           const EdgeP &ep, int min_samples, unsigned char *coreS, int *parent, const int *sidx, int *root,
                           unsigned *isroot, unsigned *rank, unsigned *h_res, int *overflow, int32_t *labels,
                           const int *oidx, hipStream_t stream) {
    const int nb = (n + 255) / 256, nw = (n + WPB - 1) / WPB;
    degree_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, ep, min_samples, coreS);
    uf_init<<<nb, 256, 0, stream>>>(parent, n);
    hook_min_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, coreS, sidx, ep, parent);
    union_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, coreS, sidx, ep, parent);
    compress_kernel<<<nb, 256, 0, stream>>>(parent, coreS, sidx, n, root, isroot);
    scan_u32<false><<<1, 1024, 0, stream>>>(isroot, rank, n, h_res, overflow);
    label_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, coreS, sidx, root, rank, ep, labels, oidx);
}

size_t cluster_arena_bytes(int n, int neighbor_type, int k_neighbors) {
    const bool unbounded = neighbor_type >= MODEST_GRAPH_KNN;
    const int ustride = k_neighbors + (neighbor_type == MODEST_GRAPH_SYM_KNN ? 3 * k_neighbors : 0) + 8;
    return arena_sz(sizeof(CGrid)) + arena_sz((CG_CELLS + 4) * 4) + 2 * arena_sz((CG_CELLS + 1) * 4) +
           arena_sz((size_t)n * 16) + arena_sz((size_t)n * 4) + arena_sz((size_t)n * 8) + arena_sz((size_t)n) +
           arena_sz((size_t)n * 4) * 3 + arena_sz((size_t)(n + 1) * 4) + arena_sz((size_t)n * 4) +
           arena_sz((size_t)n * (unbounded ? (ustride > ADJ ? ustride : ADJ) : ADJ) * 4) + arena_sz((size_t)n * 4) * 3;
}

// The arena (ctx->scratch + arena_off, cluster_arena_bytes(n) reserved by the caller) holds every
// intermediate.  pre == NULL: xyz / pp / intensity are per input point and labels[i] is written for
// every input point; pre != NULL: see PreGrid, labels[pre->idx[i]] is written.
int cluster_impl(modest_ctx *ctx, size_t arena_off, const float *xyz, const float *pp, const float *intensity, int n,
                 int neighbor_type, int affinity_type, int k_neighbors, double radius, double eps, int min_samples,
                 int32_t *labels, double *kth_d2, int32_t *n_clusters, hipStream_t stream, const PreGrid *pre) {
    const bool unbounded = neighbor_type >= MODEST_GRAPH_KNN;
    const int ustride = k_neighbors + (neighbor_type == MODEST_GRAPH_SYM_KNN ? 3 * k_neighbors : 0) + 8;   // k-NN graph rows
    const size_t zero_words = (size_t)CG_CELLS + 4;   // cell counters, overflow flag, changed flag
    Arena A(ctx->scratch + arena_off);
    CGrid *g = A.take<CGrid>(1);
    unsigned *zeroed = A.take<unsigned>(zero_words);
    unsigned *cnt = zeroed;
    unsigned *start = A.take<unsigned>(CG_CELLS + 1);
    unsigned *cursor = A.take<unsigned>(CG_CELLS + 1);
    float4 *sorted = A.take<float4>(n);
    int *sidx = A.take<int>(n);
    double *kthS = A.take<double>(n);
    unsigned char *coreS = A.take<unsigned char>(n);
    int *parent = A.take<int>(n);
    int *root = A.take<int>(n);
    unsigned *isroot = A.take<unsigned>(n);
    unsigned *rank = A.take<unsigned>(n + 1);
    int *deg = A.take<int>(n);
    int *adj = A.take<int>((size_t)n * (unbounded ? (ustride > ADJ ? ustride : ADJ) : ADJ));
    float *sortedI = A.take<float>(n);
    int *degOut = A.take<int>(n);
    int *Lmin = A.take<int>(n);
    int *overflow = reinterpret_cast<int *>(zeroed + CG_CELLS);
    unsigned *h_res = reinterpret_cast<unsigned *>(ctx->pinned);   // [overflow, number of clusters], written by the last scan
    const int *oidx = pre ? pre->idx : nullptr;

    const double c = radius * (1.0 + 1.0 / 1024.0);
    const double r2 = radius * radius;
    const int nb = (n + 255) / 256, nw = (n + WPB - 1) / WPB;
    const int nbA = (int)(((long long)n * AG + 255) / 256);
    const bool l2 = affinity_type == MODEST_AFFINITY_L2_4D;
    if (pre) {
        g = pre->g;
        scan_u32<true><<<1, 1024, 0, stream>>>(pre->cnt, start, CG_CELLS, nullptr, nullptr, cursor, pre->cnt,
                                               reinterpret_cast<unsigned *>(overflow));
        cg_scatter<<<nb, 256, 0, stream>>>(xyz, pp, n, g, cursor, sorted, sidx, pre->idx, nullptr, pre->rows, pre->stride,
                                           l2 ? sortedI : nullptr);
    } else {
        cg_bbox<<<1, 1024, 0, stream>>>(xyz, n, c, g, zeroed, (int)zero_words);
        cg_count<<<nb, 256, 0, stream>>>(xyz, n, g, cnt);
        scan_u32<true><<<1, 1024, 0, stream>>>(cnt, start, CG_CELLS, nullptr, nullptr, cursor, nullptr, nullptr);
        cg_scatter<<<nb, 256, 0, stream>>>(xyz, pp, n, g, cursor, sorted, sidx, nullptr, intensity, nullptr, 0,
                                           l2 ? sortedI : nullptr);
    }
    EdgeP ep;
    ep.r2 = r2;
    ep.eps = eps;
    ep.use_knn = neighbor_type == MODEST_GRAPH_RADIUS_MUTUAL_KNN;
    ep.affinity = affinity_type;
    ep.inten = sortedI;
    if (unbounded) {
        // ---- knn / sym_knn / mutual_knn: exact k-th neighbour distance without a radius bound ----
        int *changed = overflow + 1;
        knn_kth_unbounded_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, k_neighbors, c, kthS);
        adj_knn_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, ep, c,
                                                   neighbor_type == MODEST_GRAPH_MUTUAL_KNN, ustride, degOut, adj,
                                                   overflow);
        MODEST_HIP_CHECK(hipMemcpyAsync(deg, degOut, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
        if (neighbor_type == MODEST_GRAPH_SYM_KNN)
            adj_symmetrize_kernel<<<nb, 256, 0, stream>>>(sorted, n, kthS, ustride, degOut, deg, adj, overflow);
        core_from_deg_kernel<<<nb, 256, 0, stream>>>(n, deg, ustride, min_samples, coreS, deg);
        if (neighbor_type == MODEST_GRAPH_KNN) {
            dir_init_kernel<<<nb, 256, 0, stream>>>(n, coreS, sidx, Lmin);
            for (int it = 0; it < 4096; ++it) {   // a propagation step per launch, convergence checked every 8
                dir_push_kernel<<<nbA, 256, 0, stream>>>(n, ustride, coreS, deg, adj, Lmin, changed);
                if (it % 8 == 7) {
                    MODEST_HIP_CHECK(hipMemcpyAsync(ctx->pinned + 16, changed, 4, hipMemcpyDeviceToHost, stream));
                    MODEST_HIP_CHECK(hipMemsetAsync(changed, 0, 4, stream));
                    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
                    if (!*reinterpret_cast<int *>(ctx->pinned + 16)) break;
                }
            }
            dir_roots_kernel<<<nb, 256, 0, stream>>>(n, coreS, sidx, Lmin, root, isroot);
            scan_u32<false><<<1, 1024, 0, stream>>>(isroot, rank, n, h_res, overflow);
            dir_label_kernel<<<nb, 256, 0, stream>>>(n, root, rank, labels, oidx);
        } else {
            uf_init<<<nb, 256, 0, stream>>>(parent, n);
            for (int round = 0; round < HOOK_ROUNDS; ++round) {
                hook_adj_kernel<<<nbA, 256, 0, stream>>>(n, ustride, coreS, deg, adj, sidx, parent);
                flatten_kernel<<<nb, 256, 0, stream>>>(parent, n);
            }
            union_adj_kernel<<<nbA, 256, 0, stream>>>(n, ustride, coreS, deg, adj, sidx, parent);
            compress_kernel<<<nb, 256, 0, stream>>>(parent, coreS, sidx, n, root, isroot);
            scan_u32<false><<<1, 1024, 0, stream>>>(isroot, rank, n, h_res, overflow);
            label_adj_kernel<<<nbA, 256, 0, stream>>>(n, ustride, coreS, deg, adj, sidx, root, rank, labels, oidx);
        }
        MODEST_HIP_CHECK(hipGetLastError());
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        MODEST_REQUIRE(!h_res[0],
                       "k-NN graph: a row exceeded its capacity (massively tied k-th distances or a hub point)");
        if (n_clusters) *n_clusters = (int32_t)h_res[1];
        if (kth_d2) scatter_kth<<<nb, 256, 0, stream>>>(kthS, sidx, n, kth_d2);
        MODEST_HIP_CHECK(hipGetLastError());
        return MODEST_OK;
    }
    if (ep.use_knn || kth_d2)
        knn_kth_kernel<<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, k_neighbors, r2, kthS);
    degree_adj_kernel<true><<<nw, 64 * WPB, 0, stream>>>(sorted, n, g, start, kthS, ep, min_samples, coreS, deg, adj,
                                                        overflow, sidx, parent);
    for (int round = 0; round < HOOK_ROUNDS; ++round) {   // accelerators only: union_orig_kernel makes the result exact
        hook_orig_kernel<<<nbA, 256, 0, stream>>>(n, coreS, deg, adj, sidx, parent);
        flatten_orig_kernel<<<nb, 256, 0, stream>>>(parent, n);
    }
    union_orig_kernel<<<nbA, 256, 0, stream>>>(n, coreS, deg, adj, sidx, parent);
    compress_orig_kernel<<<nb, 256, 0, stream>>>(parent, coreS, sidx, n, root, isroot);
    scan_u32<false><<<1, 1024, 0, stream>>>(isroot, rank, n, h_res, overflow);
    label_orig_kernel<<<nbA, 256, 0, stream>>>(n, coreS, deg, adj, sidx, root, rank, labels, oidx);
    MODEST_HIP_CHECK(hipGetLastError());
    {   // more than ADJ edges at some point (dozens of exactly tied k-th distances): recompute path
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        if (h_res[0]) {
            cluster_overflow_path(sorted, n, g, start, kthS, ep, min_samples, coreS, parent, sidx, root, isroot, rank, h_res,
                                  overflow, labels, oidx, stream);
            MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        }
        if (n_clusters) *n_clusters = (int32_t)h_res[1];
    }
    if (kth_d2) scatter_kth<<<nb, 256, 0, stream>>>(kthS, sidx, n, kth_d2);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

}  // namespace

extern "C" int modest_cluster_dbscan_ex(modest_ctx *ctx, const float *xyz, const float *pp,
                                        const float *intensity, int n, int neighbor_type, int affinity_type,
                                        int k_neighbors, double radius, double eps, int min_samples,
                                        int32_t *labels, double *kth_d2, int32_t *n_clusters,
                                        void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0, "n < 0");
    MODEST_REQUIRE(k_neighbors >= 1 && radius > 0.0 && min_samples >= 1, "bad parameters");
    MODEST_REQUIRE(neighbor_type >= MODEST_GRAPH_RADIUS_MUTUAL_KNN && neighbor_type <= MODEST_GRAPH_MUTUAL_KNN,
                   "neighbor_type out of range");
    const bool unbounded = neighbor_type >= MODEST_GRAPH_KNN;
    MODEST_REQUIRE(!unbounded || n == 0 || n > k_neighbors, "k-NN graph: n_neighbors must be < n (sklearn raises too)");
    MODEST_REQUIRE(affinity_type >= MODEST_AFFINITY_L1 && affinity_type <= MODEST_AFFINITY_L2_4D,
                   "affinity_type must be l1, exp or 3d_l2_distance");
    MODEST_REQUIRE(affinity_type != MODEST_AFFINITY_L2_4D || intensity != nullptr || n == 0,
                   "3d_l2_distance needs the intensity column (the reference takes the norm of the (n,4) rows)");
    if (n_clusters) *n_clusters = 0;
    if (n == 0) return MODEST_OK;
    MODEST_REQUIRE(xyz && pp && labels, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = modest_ctx_reserve(ctx, cluster_arena_bytes(n, neighbor_type, k_neighbors));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, 64);
    if (rc) return rc;
    return cluster_impl(ctx, 0, xyz, pp, intensity, n, neighbor_type, affinity_type, k_neighbors, radius, eps,
                        min_samples, labels, kth_d2, n_clusters, stream, nullptr);
}

// phase & 1: enqueue the mask / compaction / cell-count kernel (no synchronise); phase & 2: continue with
// the kept count (after the caller's synchronise, or this function's own when both bits are set).  The
// kept count sits at byte 192 of the context's pinned block (clear of a RANSAC refit's 128 bytes: the two
// may be in flight together, scan_driver.hip).
int modest_mask_cluster_phase(modest_ctx *ctx, const float *pts, int n, int stride, const float *pp,
                              const double *plane4, double offset, const double *only_range4,
                              const double *limit_range4, int neighbor_type, int affinity_type, int k_neighbors,
                              double radius, double eps, int min_samples, int32_t *labels, int32_t *n_kept,
                              int32_t *n_clusters, void *stream_, int phase) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0 && (stride == 3 || stride == 4), "bad n/stride");
    MODEST_REQUIRE(plane4 && limit_range4 && n_kept, "NULL argument");
    MODEST_REQUIRE(k_neighbors >= 1 && radius > 0.0 && min_samples >= 1, "bad parameters");
    MODEST_REQUIRE(neighbor_type >= MODEST_GRAPH_RADIUS_MUTUAL_KNN && neighbor_type <= MODEST_GRAPH_MUTUAL_KNN,
                   "neighbor_type out of range");
    MODEST_REQUIRE(affinity_type >= MODEST_AFFINITY_L1 && affinity_type <= MODEST_AFFINITY_L2_4D,
                   "affinity_type must be l1, exp or 3d_l2_distance");
    MODEST_REQUIRE(affinity_type != MODEST_AFFINITY_L2_4D || stride == 4,
                   "3d_l2_distance needs the intensity column of the scan rows");
    if (phase & 1) {
        *n_kept = 0;
        if (n_clusters) *n_clusters = 0;
    }
    if (n == 0) return MODEST_OK;
    MODEST_REQUIRE(pts && pp && labels, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    // [kept xyz | kept idx | grid descriptor | n_kept word] then the clustering arena, reserved for the
    // worst case (every row kept) so that nothing moves between the two halves of the call
    const size_t own = arena_sz((size_t)n * 12) + arena_sz((size_t)n * 4) + arena_sz(sizeof(CGrid));
    int rc = modest_ctx_reserve(ctx, own + cluster_arena_bytes(n, neighbor_type, k_neighbors));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, 256);
    if (rc) return rc;
    Arena A(ctx->scratch);
    float *kept = A.take<float>((size_t)n * 3);
    int *kept_idx = A.take<int>(n);
    CGrid *g = A.take<CGrid>(1);
    int *h_kept = reinterpret_cast<int *>(ctx->pinned + 192);   // pinned: there after the sync, no copy
    unsigned *cnt = nullptr;
    static_assert(CG_CELLS == (int)MODEST_ZW_CELLS, "persistent cell counters");
    if (phase & 1) {
        rc = modest_ctx_zero_words(ctx, stream, &cnt);
        if (rc) return rc;
        ctx->zwords_dirty = 1;   // until the cell scan behind the counters has been enqueued
        ctx->zwords_live = 1;    // ... and nobody may clear them in between
    } else {
        cnt = ctx->zwords;
    }
    MaskParams P;
    mask_params_fill(P, plane4, offset, only_range4, limit_range4);
    // the kept rows lie inside limit_range: a fixed grid around it needs no bounding-box pass (cells
    // beyond the grid are clamped; the sweeps stay exact, see cg_coord / rows_of)
    const double c = radius * (1.0 + 1.0 / 1024.0);
    double cx = 0.5 * ((double)P.lx0 + (double)P.lx1), cy = 0.5 * ((double)P.ly0 + (double)P.ly1);
    if (!(fabs(cx) <= 1e30)) cx = 0.0;
    if (!(fabs(cy) <= 1e30)) cy = 0.0;
    CGrid G;
    G.ox = cx - 0.5 * CG * c;
    G.oy = cy - 0.5 * CG * c;
    G.inv_c = 1.0 / c;
    if (phase & 1) {
        const int nblk = (n + 1023) / 1024;
        unsigned long long *state = nullptr;
        rc = modest_ctx_compact_state(ctx, (size_t)nblk, stream, &state);
        if (rc) return rc;
        mask_count_kernel<<<nblk, 1024, 0, stream>>>(pts, n, stride, P, G, g, cnt, labels, kept, kept_idx, state, h_kept);
        MODEST_HIP_CHECK(hipGetLastError());
    }
    if (!(phase & 2)) return MODEST_OK;
    if (phase & 1) MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    ctx->zwords_live = 0;
    const int m = *h_kept;
    *n_kept = m;
    if (m == 0) {
        ctx->zwords_dirty = 0;   // nothing was counted
        return MODEST_OK;
    }
    if (neighbor_type != MODEST_GRAPH_RADIUS && m <= k_neighbors) {   // sklearn's kneighbors raises
        modest_set_error("modest_mask_cluster: Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, "
                         "n_samples_fit = %d, n_samples = %d", k_neighbors + 1, m, m);
        return MODEST_ERR_ARG;
    }
    PreGrid pre{kept_idx, pts, stride, g, cnt};
    rc = cluster_impl(ctx, own, kept, pp, nullptr, m, neighbor_type, affinity_type, k_neighbors, radius, eps, min_samples,
                      labels, nullptr, n_clusters, stream, &pre);
    if (rc == MODEST_OK) ctx->zwords_dirty = 0;
    return rc;
}

extern "C" int modest_mask_cluster(modest_ctx *ctx, const float *pts, int n, int stride, const float *pp,
                                   const double *plane4, double offset, const double *only_range4,
                                   const double *limit_range4, int neighbor_type, int affinity_type, int k_neighbors,
                                   double radius, double eps, int min_samples, int32_t *labels, int32_t *n_kept,
                                   int32_t *n_clusters, void *stream_) {
    return modest_mask_cluster_phase(ctx, pts, n, stride, pp, plane4, offset, only_range4, limit_range4, neighbor_type,
                                     affinity_type, k_neighbors, radius, eps, min_samples, labels, n_kept, n_clusters,
                                     stream_, 3);
}

extern "C" int modest_cluster_dbscan(modest_ctx *ctx, const float *xyz, const float *pp, int n,
                                     int k_neighbors, double radius, double eps, int min_samples,
                                     int32_t *labels, double *kth_d2, int32_t *n_clusters,
                                     void *stream_) {
    return modest_cluster_dbscan_ex(ctx, xyz, pp, nullptr, n, MODEST_GRAPH_RADIUS_MUTUAL_KNN, MODEST_AFFINITY_L1,
                                    k_neighbors, radius, eps, min_samples, labels, kth_d2, n_clusters, stream_);
}

// ---- the mask / graph / DBSCAN block for a CHAIN of scans ---------------------------------------------
// What modest_mask_cluster_phase does for one scan, done once for several: every kernel takes the scan as blockIdx.y
// and reads that scan's pointers from a device table (MCB); the bodies are the functions the single-scan kernels call,
// the arenas are carved exactly as the single-scan call carves them (each scan in its OWN context: scratch, pinned
// words and persistent cell counters are per scan).  15 launches and 2 round trips per chain instead of per scan.
// Radius graphs only (the default and `radius`); other configurations take the single-scan calls.
#include "mask_chain.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {
struct MCB {
    const float *pts, *pp;
    MaskParams P;
    CGrid G;
    CGrid *g;
    unsigned *cnt;
    int *labels;
    float *kept;
    int *kept_idx;
    unsigned long long *state;
    int *h_kept;
    unsigned *start, *cursor, *flags;
    float4 *sorted;
    int *sidx;
    double *kthS;
    unsigned char *coreS;
    int *parent, *root;
    unsigned *isroot, *rank;
    int *deg, *adj;
    unsigned *h_res;
    const double *plane_dev;   // when set: the plane of the mask is read from here (written earlier in the stream)
    int n, stride, nblk, m, nb, nw, nbA, pad;
};

__global__ __launch_bounds__(1024) void mcb_mask_count(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nblk) return;
    MaskParams P = S.P;
    if (S.plane_dev) {   // (mask_params_fill's plane part)
        P.n0 = S.plane_dev[0];
        P.n1 = S.plane_dev[1];
        P.n2 = S.plane_dev[2];
        P.d = S.plane_dev[3];
        P.norm = sqrt((P.n0 * P.n0 + P.n1 * P.n1) + P.n2 * P.n2);
    }
    mask_count_kernel_body(S.pts, S.n, S.stride, P, S.G, S.g, S.cnt, S.labels, S.kept, S.kept_idx, S.state, S.h_kept,
                           blockIdx.x, (unsigned)S.nblk);
}
template <bool CELLS>
__global__ __launch_bounds__(1024) void mcb_scan(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if (S.m <= 0) return;
    if (CELLS)
        scan_u32_body<true>(S.cnt, S.start, CG_CELLS, nullptr, nullptr, S.cursor, S.cnt, S.flags);
    else
        scan_u32_body<false>(S.isroot, S.rank, S.m, S.h_res, reinterpret_cast<const int *>(S.flags), nullptr, nullptr, nullptr);
}
__global__ void mcb_cg_scatter(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nb) return;
    cg_scatter_body(S.kept, S.pp, S.m, S.g, S.cursor, S.sorted, S.sidx, S.kept_idx, nullptr, S.pts, S.stride, nullptr,
                    blockIdx.x, (unsigned)S.nb);
}
__global__ __launch_bounds__(64 * WPB) void mcb_knn_kth(const MCB *__restrict__ tab, int k, double r2) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nw) return;
    knn_kth_kernel_body(S.sorted, S.m, S.g, S.start, k, r2, S.kthS, blockIdx.x, (unsigned)S.nw);
}
__global__ __launch_bounds__(64 * WPB) void mcb_degree_adj(const MCB *__restrict__ tab, EdgeP ep, int min_samples) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nw) return;
    degree_adj_kernel_body<true>(S.sorted, S.m, S.g, S.start, S.kthS, ep, min_samples, S.coreS, S.deg, S.adj,
                                 reinterpret_cast<int *>(S.flags), S.sidx, S.parent, blockIdx.x);
}
__global__ void mcb_hook(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nbA) return;
    hook_orig_kernel_body(S.m, S.coreS, S.deg, S.adj, S.sidx, S.parent, blockIdx.x, (unsigned)S.nbA);
}
__global__ void mcb_flatten(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nb) return;
    flatten_orig_kernel_body(S.parent, S.m, blockIdx.x, (unsigned)S.nb);
}
__global__ void mcb_union(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nbA) return;
    union_orig_kernel_body(S.m, S.coreS, S.deg, S.adj, S.sidx, S.parent, blockIdx.x, (unsigned)S.nbA);
}
__global__ void mcb_compress(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nb) return;
    compress_orig_kernel_body(S.parent, S.coreS, S.sidx, S.m, S.root, S.isroot, blockIdx.x, (unsigned)S.nb);
}
__global__ void mcb_label(const MCB *__restrict__ tab) {
    const MCB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nbA) return;
    label_orig_kernel_body(S.m, S.coreS, S.deg, S.adj, S.sidx, S.root, S.rank, S.labels, S.kept_idx, blockIdx.x,
                           (unsigned)S.nbA);
}

// the table travels through a staging slot of the first scan's context into that context's chain table
int mcb_upload(modest_ctx *ctx0, const std::vector<MCB> &tab, hipStream_t stream, const MCB **dev) {
    char *d = nullptr, *h = nullptr;
    const size_t bytes = tab.size() * sizeof(MCB);
    int rc = modest_ctx_chain_tab(ctx0, bytes, &d);
    if (rc) return rc;
    rc = modest_ctx_stage_slot(ctx0, bytes, reinterpret_cast<void **>(&h));
    if (rc) return rc;
    memcpy(h, tab.data(), bytes);
    MODEST_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx0, stream);
    if (rc) return rc;
    *dev = reinterpret_cast<const MCB *>(d);
    return MODEST_OK;
}
}  // namespace

bool modest_mask_chain_supported(int neighbor_type, int affinity_type) {
    return (neighbor_type == MODEST_GRAPH_RADIUS_MUTUAL_KNN || neighbor_type == MODEST_GRAPH_RADIUS) &&
           affinity_type != MODEST_AFFINITY_L2_4D;
}

struct modest_mask_chain_state {
    std::vector<MCB> tab;
    std::vector<size_t> own;
};

int modest_mask_chain_count(modest_mask_chain_scan *S, int B, double offset, const double *only_range4,
                            const double *limit_range4, int neighbor_type, int k_neighbors, double radius,
                            modest_mask_chain_state **state_out, hipStream_t stream) {
    MODEST_REQUIRE(S != nullptr && B >= 1 && B <= 64 && limit_range4 != nullptr && state_out != nullptr, "bad chain");
    auto *st = new modest_mask_chain_state;
    *state_out = st;
    st->tab.resize((size_t)B);
    st->own.resize((size_t)B);
    const double c = radius * (1.0 + 1.0 / 1024.0);
    int maxblk = 1;
    for (int s = 0; s < B; ++s) {
        modest_mask_chain_scan &q = S[s];
        MCB &b = st->tab[(size_t)s];
        memset(&b, 0, sizeof(b));
        q.n_kept = q.n_clusters = 0;
        q.alone = 0;
        MODEST_REQUIRE(q.ctx && q.pts && q.pp && q.labels && (q.plane4 || q.plane4_dev) && q.n >= 1 && (q.stride == 3 || q.stride == 4),
                       "bad scan of the chain");
        modest_ctx *ctx = q.ctx;
        const size_t own = arena_sz((size_t)q.n * 12) + arena_sz((size_t)q.n * 4) + arena_sz(sizeof(CGrid));
        st->own[(size_t)s] = own;
        int rc = modest_ctx_reserve(ctx, own + cluster_arena_bytes(q.n, neighbor_type, k_neighbors));
        if (rc) return rc;
        rc = modest_ctx_reserve_pinned(ctx, 256);
        if (rc) return rc;
        Arena A(ctx->scratch);
        b.kept = A.take<float>((size_t)q.n * 3);
        b.kept_idx = A.take<int>(q.n);
        b.g = A.take<CGrid>(1);
        b.h_kept = reinterpret_cast<int *>(ctx->pinned + 192);
        unsigned *cnt = nullptr;
        rc = modest_ctx_zero_words(ctx, stream, &cnt);
        if (rc) return rc;
        ctx->zwords_dirty = 1;
        ctx->zwords_live = 1;
        b.cnt = cnt;
        const double no_plane[4] = {0.0, 0.0, 1.0, 0.0};
        mask_params_fill(b.P, q.plane4_dev ? no_plane : q.plane4, offset, only_range4, limit_range4);
        b.plane_dev = q.plane4_dev;
        double cx = 0.5 * ((double)b.P.lx0 + (double)b.P.lx1), cy = 0.5 * ((double)b.P.ly0 + (double)b.P.ly1);
        if (!(fabs(cx) <= 1e30)) cx = 0.0;
        if (!(fabs(cy) <= 1e30)) cy = 0.0;
        b.G.ox = cx - 0.5 * CG * c;
        b.G.oy = cy - 0.5 * CG * c;
        b.G.inv_c = 1.0 / c;
        b.nblk = (q.n + 1023) / 1024;
        rc = modest_ctx_compact_state(ctx, (size_t)b.nblk, stream, &b.state);
        if (rc) return rc;
        b.pts = q.pts;
        b.pp = q.pp;
        b.labels = q.labels;
        b.n = q.n;
        b.stride = q.stride;
        maxblk = std::max(maxblk, b.nblk);
    }
    const MCB *dev = nullptr;
    int rc = mcb_upload(S[0].ctx, st->tab, stream, &dev);
    if (rc) return rc;
    mcb_mask_count<<<dim3((unsigned)maxblk, (unsigned)B), 1024, 0, stream>>>(dev);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

void modest_mask_chain_free(modest_mask_chain_state *st) { delete st; }

// after the caller's synchronise: kept counts -> the chain of graph / DBSCAN launches -> synchronise -> cluster counts
int modest_mask_chain_cluster(modest_mask_chain_scan *S, int B, modest_mask_chain_state *st, int neighbor_type,
                              int affinity_type, int k_neighbors, double radius, double eps, int min_samples,
                              hipStream_t stream) {
    MODEST_REQUIRE(S != nullptr && st != nullptr && (int)st->tab.size() == B, "bad chain");
    const double r2 = radius * radius;
    EdgeP ep;
    ep.r2 = r2;
    ep.eps = eps;
    ep.use_knn = neighbor_type == MODEST_GRAPH_RADIUS_MUTUAL_KNN;
    ep.affinity = affinity_type;
    ep.inten = nullptr;
    int maxnb = 1, maxnw = 1, maxnbA = 1, live = 0;
    for (int s = 0; s < B; ++s) {
        modest_mask_chain_scan &q = S[s];
        MCB &b = st->tab[(size_t)s];
        modest_ctx *ctx = q.ctx;
        ctx->zwords_live = 0;
        const int m = *b.h_kept;
        q.n_kept = m;
        b.m = m;
        if (q.alone) {   // taken out of the chain by the caller: the cell counters keep this scan's counts (dirty)
            b.m = 0;
            continue;
        }
        if (m == 0) {
            ctx->zwords_dirty = 0;   // nothing was counted
            continue;
        }
        if (neighbor_type != MODEST_GRAPH_RADIUS && m <= k_neighbors) {   // sklearn's kneighbors raises: the caller decides
            q.alone = 1;
            b.m = 0;
            continue;
        }
        ++live;
        // the clustering arena of cluster_impl(ctx, own, kept, ..., m, ...) in the same order
        Arena A(ctx->scratch + st->own[(size_t)s]);
        A.take<CGrid>(1);
        unsigned *zeroed = A.take<unsigned>((size_t)CG_CELLS + 4);
        b.start = A.take<unsigned>(CG_CELLS + 1);
        b.cursor = A.take<unsigned>(CG_CELLS + 1);
        b.sorted = A.take<float4>(m);
        b.sidx = A.take<int>(m);
        b.kthS = A.take<double>(m);
        b.coreS = A.take<unsigned char>(m);
        b.parent = A.take<int>(m);
        b.root = A.take<int>(m);
        b.isroot = A.take<unsigned>(m);
        b.rank = A.take<unsigned>(m + 1);
        b.deg = A.take<int>(m);
        b.adj = A.take<int>((size_t)m * ADJ);
        b.flags = zeroed + CG_CELLS;
        b.h_res = reinterpret_cast<unsigned *>(ctx->pinned);
        b.nb = (m + 255) / 256;
        b.nw = (m + WPB - 1) / WPB;
        b.nbA = (int)(((long long)m * AG + 255) / 256);
        maxnb = std::max(maxnb, b.nb);
        maxnw = std::max(maxnw, b.nw);
        maxnbA = std::max(maxnbA, b.nbA);
    }
    if (live == 0) return MODEST_OK;
    const MCB *dev = nullptr;
    int rc = mcb_upload(S[0].ctx, st->tab, stream, &dev);
    if (rc) return rc;
    const unsigned Bu = (unsigned)B;
    mcb_scan<true><<<dim3(1, Bu), 1024, 0, stream>>>(dev);
    mcb_cg_scatter<<<dim3((unsigned)maxnb, Bu), 256, 0, stream>>>(dev);
    if (ep.use_knn) mcb_knn_kth<<<dim3((unsigned)maxnw, Bu), 64 * WPB, 0, stream>>>(dev, k_neighbors, r2);
    mcb_degree_adj<<<dim3((unsigned)maxnw, Bu), 64 * WPB, 0, stream>>>(dev, ep, min_samples);
    for (int round = 0; round < HOOK_ROUNDS; ++round) {
        mcb_hook<<<dim3((unsigned)maxnbA, Bu), 256, 0, stream>>>(dev);
        mcb_flatten<<<dim3((unsigned)maxnb, Bu), 256, 0, stream>>>(dev);
    }
    mcb_union<<<dim3((unsigned)maxnbA, Bu), 256, 0, stream>>>(dev);
    mcb_compress<<<dim3((unsigned)maxnb, Bu), 256, 0, stream>>>(dev);
    mcb_scan<false><<<dim3(1, Bu), 1024, 0, stream>>>(dev);
    mcb_label<<<dim3((unsigned)maxnbA, Bu), 256, 0, stream>>>(dev);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    bool again = false;
    for (int s = 0; s < B; ++s) {
        MCB &b = st->tab[(size_t)s];
        if (b.m == 0) continue;
        if (b.h_res[0]) {   // a row of more than ADJ edges: the passes that re-walk the cell rows, for this scan alone
            cluster_overflow_path(b.sorted, b.m, b.g, b.start, b.kthS, ep, min_samples, b.coreS, b.parent, b.sidx, b.root,
                                  b.isroot, b.rank, b.h_res, reinterpret_cast<int *>(b.flags), b.labels, b.kept_idx, stream);
            again = true;
        }
    }
    if (again) MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int s = 0; s < B; ++s) {
        MCB &b = st->tab[(size_t)s];
        if (b.m == 0) continue;
        S[s].n_clusters = (int32_t)b.h_res[1];
        S[s].ctx->zwords_dirty = 0;
    }
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_cluster(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(cg_count));
}
