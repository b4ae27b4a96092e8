// Box fitting for gfx950: the 901-angle "closeness to edge" search
// (utils/pointcloud_utils.py:167-187) and get_lowest_point_rect (:278-290).
//
// The reference runs a pure-Python loop of 901 angles per cluster (29 % of its
// mask stage).  Here every (cluster, angle) pair is one lane; the cluster's
// points are read as wave-uniform (broadcast) loads.  Arithmetic contract,
// float64 throughout:
//   projection  p0 = fma(z, s, x*c),  p1 = fma(z, c, x*(-s))   (the 2-term BLAS
//               product [x z] @ [[c,s],[-s,c]]^T as an FMA chain)
//   beta_i      = 1 / max(min(min(p0-minx, maxx-p0), min(p1-miny, maxy-p1)), d0)
//   sum         in numpy's pairwise order (8 accumulators per <=128 block,
//               halves split at a multiple of 8), so the first-strict-maximum
//               over the angles is reproducible bit for bit.
// The (cos, sin) table comes from the host so that device libm differences
// cannot move the chosen angle.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct BetaCtx {
    const double *pts;   // (n,2) x,z of this cluster
    double c, s, ns;
    double minx, maxx, miny, maxy, d0;
};

__device__ __forceinline__ double beta_at(const BetaCtx &B, int i) {
    const double x = B.pts[2 * (size_t)i], z = B.pts[2 * (size_t)i + 1];
    const double p0 = fma(z, B.s, x * B.c);
    const double p1 = fma(z, B.c, x * B.ns);
    const double dx = fmin(p0 - B.minx, B.maxx - p0);
    const double dy = fmin(p1 - B.miny, B.maxy - p1);
    double b = fmin(dx, dy);
    b = fmax(b, B.d0);
    return 1.0 / b;
}

// numpy pairwise-sum leaf: n <= 128
__device__ double pw_leaf(const BetaCtx &B, int off, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += beta_at(B, off + i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = beta_at(B, off + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += beta_at(B, off + i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += beta_at(B, off + i);
    return res;
}

// the same leaf on eight adjacent lanes (j = lane & 7): lane j owns numpy's accumulator r[j], the
// three shuffle-adds rebuild ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) (every level adds the same two
// values on both sides, so all eight lanes end up with numpy's bits), lane 0's tail loop follows.
// Must be called by all eight lanes; the result is valid on every one of them.
__device__ __forceinline__ double pw_leaf8(const BetaCtx &B, int off, int n, int j) {
    if (n < 8) {   // wave-divergence free: every lane of the group takes the same branch
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += beta_at(B, off + i);
        return res;
    }
    const int body = n - (n % 8);
    double r = beta_at(B, off + j);
    for (int i = 8 + j; i < body; i += 8) r += beta_at(B, off + i);
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 4);
    for (int i = body; i < n; ++i) r += beta_at(B, off + i);
    return r;
}

// explicit-stack form of: n<=128 ? leaf : sum(a,n2) + sum(a+n2,n-n2), n2 = n/2 - (n/2)%8
__device__ double pw_tree(const BetaCtx &B, int off0, int n) {
    int f_off[32], f_n[32], f_stage[32];
    double f_left[32];
    int sp = 0;
    f_off[0] = off0;
    f_n[0] = n;
    f_stage[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int off = f_off[sp], len = f_n[sp];
        if (len <= 128) {
            ret = pw_leaf(B, off, len);
            --sp;
            continue;
        }
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (f_stage[sp] == 0) {
            f_stage[sp] = 1;
            ++sp;
            f_off[sp] = off;
            f_n[sp] = n2;
            f_stage[sp] = 0;
        } else if (f_stage[sp] == 1) {
            f_left[sp] = ret;
            f_stage[sp] = 2;
            ++sp;
            f_off[sp] = off + n2;
            f_n[sp] = len - n2;
            f_stage[sp] = 0;
        } else {
            ret = f_left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

// numpy.add.reduce walks a contiguous array in buffers of NP_BUFSIZE elements: each buffer is summed
// pairwise and added to the running total, ((p(c0) + p(c1)) + p(c2)) + ... (measured against
// numpy 2.2: the plain tree over the whole array differs in the last bits from n ~ 20 k on)
constexpr int NP_BUFSIZE = 8192;
__device__ double pw_sum(const BetaCtx &B, int n) {
    double total = 0.0;
    for (int off = 0; off < n; off += NP_BUFSIZE) total += pw_tree(B, off, min(NP_BUFSIZE, n - off));
    return total;
}

__global__ __launch_bounds__(128) void closeness_kernel(const double *__restrict__ pts,
                                                        const int *__restrict__ offsets,
                                                        const double *__restrict__ cossin,
                                                        int n_angles, double d0,
                                                        double *__restrict__ beta) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (a >= n_angles) return;
    const int b = offsets[c], e = offsets[c + 1];
    const int n = e - b;
    BetaCtx B;
    B.pts = pts + 2 * (size_t)b;
    B.c = cossin[2 * a];
    B.s = cossin[2 * a + 1];
    B.ns = -B.s;
    B.d0 = d0;
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int i = 0; i < n; ++i) {
        const double x = B.pts[2 * (size_t)i], z = B.pts[2 * (size_t)i + 1];
        const double p0 = fma(z, B.s, x * B.c);
        const double p1 = fma(z, B.c, x * B.ns);
        mnx = fmin(mnx, p0);
        mxx = fmax(mxx, p0);
        mny = fmin(mny, p1);
        mxy = fmax(mxy, p1);
    }
    B.minx = mnx;
    B.maxx = mxx;
    B.miny = mny;
    B.maxy = mxy;
    beta[(size_t)c * n_angles + a] = (n > 0) ? pw_sum(B, n) : 0.0;
}

// first strict maximum over the angles (pointcloud_utils.py:185-187) = largest value, ties to
// the smallest index; one wavefront per cluster
__device__ __forceinline__ int argmax_wave(const double *__restrict__ row, int n_angles, int lane) {
    double mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int a = lane; a < n_angles; a += 64) {
        const double v = __hip_atomic_load(row + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v > mx) {
            mx = v;
            arg = a;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(mx, o);
        const int oa = __shfl_xor(arg, o);
        if (ov > mx || (ov == mx && oa < arg)) {
            mx = ov;
            arg = oa;
        }
    }
    return (arg == 0x7fffffff) ? -1 : arg;
}

__global__ __launch_bounds__(64) void argmax_kernel(const double *__restrict__ beta, int n_clusters,
                                                    int n_angles, int *__restrict__ best) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= n_clusters) return;
    const int arg = argmax_wave(beta + (size_t)c * n_angles, n_angles, lane);
    if (lane == 0) best[c] = arg;
}

// The same sum, spread over the lanes of a block.  numpy's pairwise sum is a fixed binary tree
// over the index range (of every 8192-element reduction buffer) whose leaves hold 65..128 elements (or everything when n <= 128); the host
// lists the leaves of every cluster and the post-order "program" (0 = next leaf, 1 = add) that
// combines them, so the leaves can be summed side by side and combined in numpy's order.
// A block serves one cluster and 256/P angles; P lanes share an angle: they split the min/max
// pass by points and the sum pass by leaves, then lane 0 runs the program on an LDS stack.
// (One lane per (cluster, angle), above, walks the whole cluster serially -- a 2000-point
// cluster made that kernel take 76 us while most of the chip idled.)
constexpr int CT_STACK = 40;
template <int P>
__global__ __launch_bounds__(256) void closeness_tree_kernel(const double *__restrict__ pts,
                                                             const int *__restrict__ offsets,
                                                             const double *__restrict__ cossin, int n_angles,
                                                             double d0, const int *__restrict__ leafBase,
                                                             const int2 *__restrict__ leaves,
                                                             const int *__restrict__ progBase,
                                                             const unsigned char *__restrict__ prog,
                                                             int maxLeaves, double *__restrict__ beta,
                                                             unsigned *tickets, int *__restrict__ best_host,
                                                             const double *__restrict__ cossin90,
                                                             double *__restrict__ rect_host) {
    constexpr int AB = 256 / P;
    extern __shared__ double ct_lsum[];          // [AB][maxLeaves]
    __shared__ double stk[AB][CT_STACK];
    const int c = blockIdx.y, al = threadIdx.x / P, sub = threadIdx.x % P;
    const int a = blockIdx.x * AB + al;
    const int aa = min(a, n_angles - 1);
    const int b = offsets[c], n = offsets[c + 1] - b;
    const int lb = leafBase[c], nl = leafBase[c + 1] - lb;
    BetaCtx B;
    B.pts = pts + 2 * (size_t)b;
    B.c = cossin[2 * aa];
    B.s = cossin[2 * aa + 1];
    B.ns = -B.s;
    B.d0 = d0;
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int i = sub; i < n; i += P) {
        const double x = B.pts[2 * (size_t)i], z = B.pts[2 * (size_t)i + 1];
        const double p0 = fma(z, B.s, x * B.c);
        const double p1 = fma(z, B.c, x * B.ns);
        mnx = fmin(mnx, p0);
        mxx = fmax(mxx, p0);
        mny = fmin(mny, p1);
        mxy = fmax(mxy, p1);
    }
#pragma unroll
    for (int o = P / 2; o > 0; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    B.minx = mnx;
    B.maxx = mxx;
    B.miny = mny;
    B.maxy = mxy;
    double *ls = ct_lsum + (size_t)al * maxLeaves;
    // eight lanes per leaf (one per numpy accumulator): a scan's clusters have a handful of leaves,
    // one lane per leaf left most lanes idle behind a 128-element serial chain of divisions
    for (int l0 = 0; l0 < nl; l0 += P / 8) {   // block-uniform trip count (pw_leaf8 shuffles)
        const int l = l0 + sub / 8;
        const int2 lf = l < nl ? leaves[lb + l] : make_int2(0, 0);
        const double v = pw_leaf8(B, lf.x, lf.y, sub & 7);
        if (l < nl && (sub & 7) == 0) ls[l] = v;
    }
    __syncthreads();
    if (sub == 0 && a < n_angles) {
        double *st = stk[al];
        int sp = 0, li = 0;
        for (int t = progBase[c]; t < progBase[c + 1]; ++t) {
            if (prog[t] == 0) {
                st[sp++] = ls[li++];
            } else {
                const double r = st[--sp], l = st[--sp];
                st[sp++] = l + r;
            }
        }
        __hip_atomic_store(beta + (size_t)c * n_angles + a, (n > 0) ? st[0] : 0.0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // the block that finishes a cluster last picks the angle (tickets: one word per cluster, left at zero)
    // (no __threadfence(): on this multi-XCD part it writes back / invalidates a whole L2; the betas are
    // agent-scope write-through stores, drained by every wavefront before the barrier (common.h), the last
    // block reads them with agent-scope loads)
    if (!tickets) return;
    __shared__ unsigned last_s;
    modest_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(tickets + c, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s) return;
    __shared__ int arg_s;
    {   // first strict maximum over the angles, all four wavefronts: larger value wins, equal values the smaller index
        __shared__ double wmx[4];
        __shared__ int warg[4];
        const double *row = beta + (size_t)c * n_angles;
        double mx = -INFINITY;
        int arg = 0x7fffffff;
        for (int a2 = threadIdx.x; a2 < n_angles; a2 += 256) {
            const double v = __hip_atomic_load(row + a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v > mx) {
                mx = v;
                arg = a2;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(mx, o);
            const int oa = __shfl_xor(arg, o);
            if (ov > mx || (ov == mx && oa < arg)) {
                mx = ov;
                arg = oa;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            wmx[threadIdx.x >> 6] = mx;
            warg[threadIdx.x >> 6] = arg;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 1; w2 < 4; ++w2)
                if (wmx[w2] > mx || (wmx[w2] == mx && warg[w2] < arg)) {
                    mx = wmx[w2];
                    arg = warg[w2];
                }
            arg = (arg == 0x7fffffff) ? -1 : arg;
            best_host[c] = arg;
            tickets[c] = 0u;
            arg_s = arg;
        }
    }
    if (!rect_host) return;
    // ... and the extents of the cluster along the chosen heading and along heading + pi/2
    // (rectangle_at_angle, pointcloud_utils.py:188-216: projection = pts @ [[c, s], [-s, c]]^T, its
    // column minima / maxima; the host keeps the scalar tail).  cossin90 holds cos / sin of
    // angle + pi/2 as the host's numpy evaluates them.
    __syncthreads();
    const int arg = arg_s;
    __shared__ double ext[8][4];   // [value][wavefront]
    double e[8] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
    if (arg >= 0) {
        const double c0 = cossin[2 * arg], s0 = cossin[2 * arg + 1], c1 = cossin90[2 * arg], s1 = cossin90[2 * arg + 1];
        for (int i = threadIdx.x; i < n; i += 256) {
            const double x = B.pts[2 * (size_t)i], z = B.pts[2 * (size_t)i + 1];
            const double p0 = fma(z, s0, x * c0), p1 = fma(z, c0, x * -s0);
            const double q0 = fma(z, s1, x * c1), q1 = fma(z, c1, x * -s1);
            e[0] = fmin(e[0], p0);
            e[1] = fmax(e[1], p0);
            e[2] = fmin(e[2], p1);
            e[3] = fmax(e[3], p1);
            e[4] = fmin(e[4], q0);
            e[5] = fmax(e[5], q0);
            e[6] = fmin(e[6], q1);
            e[7] = fmax(e[7], q1);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        for (int o = 32; o > 0; o >>= 1) {
            const double v = __shfl_xor(e[q], o);
            e[q] = (q & 1) ? fmax(e[q], v) : fmin(e[q], v);
        }
        if ((threadIdx.x & 63) == 0) ext[q][threadIdx.x >> 6] = e[q];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int q = threadIdx.x;
        double v = ext[q][0];
        for (int w = 1; w < 4; ++w) v = (q & 1) ? fmax(v, ext[q][w]) : fmin(v, ext[q][w]);
        rect_host[(size_t)c * 8 + q] = v;   // min_x, max_x, min_y, max_y at the heading; the same at heading + pi/2
    }
}

// leaves and combine program of numpy's pairwise sum over n elements (host side)
static void pw_tree_host(int off, int n, std::vector<int> &leaf, std::vector<unsigned char> &prog) {
    if (n <= 128) {
        leaf.push_back(off);
        leaf.push_back(n);
        prog.push_back(0);
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_tree_host(off, n2, leaf, prog);
    pw_tree_host(off + n2, n - n2, leaf, prog);
    prog.push_back(1);
}

// ---- fit_method = 'variance_to_edge' (utils/pointcloud_utils.py:218-275) ---------------
// Same 901-angle search, criterion  -var(Dx[Dx<Dy]) - var(Dy[Dy<Dx])  with numpy's var:
// mean = pairwise_sum(x)/n, var = pairwise_sum((x-mean)*(x-mean))/n.  The subsets are never
// materialised: a forward iterator yields their elements in index order, which is the order
// in which numpy's pairwise summation consumes them.
struct VarIt {
    const double *pts;
    double c, s, ns, minx, maxx, miny, maxy, shift;
    int i, sel, sq;
};

__device__ __forceinline__ double var_next(VarIt &it) {
    for (;;) {
        const double x = it.pts[2 * (size_t)it.i], z = it.pts[2 * (size_t)it.i + 1];
        ++it.i;
        const double p0 = fma(z, it.s, x * it.c);
        const double p1 = fma(z, it.c, x * it.ns);
        const double dx = fmin(p0 - it.minx, it.maxx - p0);
        const double dy = fmin(p1 - it.miny, it.maxy - p1);
        const bool take = it.sel == 0 ? (dx < dy) : (dy < dx);
        if (!take) continue;
        const double v = it.sel == 0 ? dx : dy;
        if (!it.sq) return v;
        const double t = v - it.shift;
        return t * t;
    }
}

__device__ double pw_leaf_it(VarIt &it, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += var_next(it);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = var_next(it);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += var_next(it);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += var_next(it);
    return res;
}

__device__ double pw_tree_it(VarIt &it, int n) {
    int f_n[32], f_stage[32];
    double f_left[32];
    int sp = 0;
    f_n[0] = n;
    f_stage[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int len = f_n[sp];
        if (len <= 128) {
            ret = pw_leaf_it(it, len);
            --sp;
            continue;
        }
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (f_stage[sp] == 0) {
            f_stage[sp] = 1;
            ++sp;
            f_n[sp] = n2;
            f_stage[sp] = 0;
        } else if (f_stage[sp] == 1) {
            f_left[sp] = ret;
            f_stage[sp] = 2;
            ++sp;
            f_n[sp] = len - n2;
            f_stage[sp] = 0;
        } else {
            ret = f_left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__device__ double pw_sum_it(VarIt &it, int n) {   // buffers of NP_BUFSIZE, as numpy reduces them
    double total = 0.0;
    for (int off = 0; off < n; off += NP_BUFSIZE) total += pw_tree_it(it, min(NP_BUFSIZE, n - off));
    return total;
}

__global__ __launch_bounds__(128) void variance_kernel(const double *__restrict__ pts,
                                                       const int *__restrict__ offsets,
                                                       const double *__restrict__ cossin, int n_angles,
                                                       double *__restrict__ crit) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (a >= n_angles) return;
    const int b = offsets[c], n = offsets[c + 1] - b;
    VarIt it;
    it.pts = pts + 2 * (size_t)b;
    it.c = cossin[2 * a];
    it.s = cossin[2 * a + 1];
    it.ns = -it.s;
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int i = 0; i < n; ++i) {
        const double x = it.pts[2 * (size_t)i], z = it.pts[2 * (size_t)i + 1];
        const double p0 = fma(z, it.s, x * it.c), p1 = fma(z, it.c, x * it.ns);
        mnx = fmin(mnx, p0);
        mxx = fmax(mxx, p0);
        mny = fmin(mny, p1);
        mxy = fmax(mxy, p1);
    }
    it.minx = mnx;
    it.maxx = mxx;
    it.miny = mny;
    it.maxy = mxy;
    int cnt[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const double x = it.pts[2 * (size_t)i], z = it.pts[2 * (size_t)i + 1];
        const double p0 = fma(z, it.s, x * it.c), p1 = fma(z, it.c, x * it.ns);
        const double dx = fmin(p0 - mnx, mxx - p0), dy = fmin(p1 - mny, mxy - p1);
        cnt[0] += dx < dy;
        cnt[1] += dy < dx;
    }
    double var = 0.0;
    for (int sel = 0; sel < 2; ++sel) {
        const int m = cnt[sel];
        if (m == 0) continue;
        it.sel = sel;
        it.i = 0;
        it.sq = 0;
        it.shift = 0.0;
        const double mean = pw_sum_it(it, m) / (double)m;
        it.i = 0;
        it.sq = 1;
        it.shift = mean;
        var += -(pw_sum_it(it, m) / (double)m);
    }
    crit[(size_t)c * n_angles + a] = (n > 0) ? var : -INFINITY;
}

// ---- fit_method = 'PCA' (utils/pointcloud_utils.py:189-206) ------------------------------
// sklearn PCA(n_components=2).components_ of the (n,2) cluster: principal axes of the centred
// points (closed form for the 2x2 scatter matrix), signs fixed as svd_flip(u_based_decision=False)
// does (largest |entry| of every axis positive); then the extent of the raw points along them.
// out[c*8 + {0..3: components row-major, 4: min0, 5: max0, 6: min1, 7: max1}]
__global__ __launch_bounds__(64) void pca_kernel(const double *__restrict__ pts, const int *__restrict__ offsets,
                                                 double *__restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int b = offsets[c], n = offsets[c + 1] - b;
    const double *p = pts + 2 * (size_t)b;
    double sx = 0.0, sz = 0.0;
    for (int i = lane; i < n; i += 64) {
        sx += p[2 * (size_t)i];
        sz += p[2 * (size_t)i + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sz += __shfl_xor(sz, o);
    }
    const double mx = n ? sx / n : 0.0, mz = n ? sz / n : 0.0;
    double a = 0.0, bb = 0.0, d = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double x = p[2 * (size_t)i] - mx, z = p[2 * (size_t)i + 1] - mz;
        a += x * x;
        bb += x * z;
        d += z * z;
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        bb += __shfl_xor(bb, o);
        d += __shfl_xor(d, o);
    }
    // leading eigenvector of [[a, bb], [bb, d]]
    double v0, v1;
    if (bb == 0.0) {
        v0 = a >= d ? 1.0 : 0.0;
        v1 = a >= d ? 0.0 : 1.0;
    } else {
        const double h = 0.5 * (a - d), r = sqrt(h * h + bb * bb);
        if (h >= 0.0) {   // lambda1 - d = h + r (no cancellation)
            v0 = h + r;
            v1 = bb;
        } else {          // lambda1 - a = r - h
            v0 = bb;
            v1 = r - h;
        }
        const double nrm = sqrt(v0 * v0 + v1 * v1);
        v0 /= nrm;
        v1 /= nrm;
    }
    double w0 = -v1, w1 = v0;
    if ((fabs(v0) >= fabs(v1) ? v0 : v1) < 0.0) {
        v0 = -v0;
        v1 = -v1;
    }
    if ((fabs(w0) >= fabs(w1) ? w0 : w1) < 0.0) {
        w0 = -w0;
        w1 = -w1;
    }
    double mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const double x = p[2 * (size_t)i], z = p[2 * (size_t)i + 1];
        const double q0 = fma(z, v1, x * v0), q1 = fma(z, w1, x * w0);   // cluster @ components.T
        mn0 = fmin(mn0, q0);
        mx0 = fmax(mx0, q0);
        mn1 = fmin(mn1, q1);
        mx1 = fmax(mx1, q1);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn0 = fmin(mn0, __shfl_xor(mn0, o));
        mx0 = fmax(mx0, __shfl_xor(mx0, o));
        mn1 = fmin(mn1, __shfl_xor(mn1, o));
        mx1 = fmax(mx1, __shfl_xor(mx1, o));
    }
    if (lane == 0) {
        double *o8 = out + 8 * (size_t)c;
        o8[0] = v0;
        o8[1] = v1;
        o8[2] = w0;
        o8[3] = w1;
        o8[4] = mn0;
        o8[5] = mx0;
        o8[6] = mn1;
        o8[7] = mx1;
    }
}

struct Box6 {
    double cx, cz, l, w, c, s;
};

// grid (LOW_SPLIT point slices, boxes): a block scans its slice of the points for one box and
// reduces in registers / LDS; the block that finishes a box last (ticket per box, left at zero)
// combines the LOW_SPLIT partial maxima and writes the result to pinned host memory.  The box
// table is read from pinned host memory as well: no copy or memset launches around the kernel.
constexpr int LOW_SPLIT = 8;
struct BoxSrc {   // boxes of several scans in one launch: the points a box looks at
    const double *pts;
    long long n;
};
__global__ __launch_bounds__(1024) void lowest_kernel(const double *__restrict__ pts, int n,
                                                      const Box6 *__restrict__ boxes, int n_boxes,
                                                      double *__restrict__ partial, unsigned *tickets,
                                                      double *__restrict__ out_host,
                                                      const BoxSrc *__restrict__ src = nullptr) {
    const Box6 b = boxes[blockIdx.y];
    if (src) {
        pts = src[blockIdx.y].pts;
        n = (int)src[blockIdx.y].n;
    }
    const double hl = b.l / 2, hw = b.w / 2, ns = -b.s;
    double best = -INFINITY;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += LOW_SPLIT * 1024) {
        const double dx = pts[3 * (size_t)i] - b.cx, dz = pts[3 * (size_t)i + 2] - b.cz;
        // [dx dz] @ [[c,-s],[s,c]]^T
        const double q0 = fma(dz, ns, dx * b.c);
        const double q1 = fma(dz, b.c, dx * b.s);
        if (q0 > -hl && q0 < hl && q1 > -hw && q1 < hw) best = fmax(best, pts[3 * (size_t)i + 1]);
    }
    for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o));
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) best = fmax(best, red[k]);
        __hip_atomic_store(partial + (size_t)blockIdx.y * LOW_SPLIT + blockIdx.x, best, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        modest_drain_stores();   // the write-through store has landed before the ticket
        if (atomicAdd(tickets + blockIdx.y, 1u) == LOW_SPLIT - 1) {
            double m = -INFINITY;
            for (int k = 0; k < LOW_SPLIT; ++k)
                m = fmax(m, __hip_atomic_load(partial + (size_t)blockIdx.y * LOW_SPLIT + k, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT));
            out_host[blockIdx.y] = m;   // -inf: no point inside
            tickets[blockIdx.y] = 0u;
        }
    }
}

}  // namespace

// pts_host != NULL: the points are host memory and travel in the same upload block as the tables
// (one copy for the whole call); otherwise pts_xz is device memory.
static int fit_boxes_angles(modest_ctx *ctx, int variance, const double *pts_xz, const int32_t *offsets_host,
                            int n_clusters, const double *cossin_host, int n_angles, double d0,
                            int32_t *best_angle_host, double *beta_host, void *stream_,
                            const double *pts_host = nullptr, const double *cossin90_host = nullptr,
                            double *rect_host_out = nullptr) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_clusters >= 0 && n_angles >= 1 && n_angles <= 65536, "bad sizes");
    if (n_clusters == 0) return MODEST_OK;
    MODEST_REQUIRE((pts_xz || pts_host) && offsets_host && cossin_host && best_angle_host, "NULL buffer");
    MODEST_REQUIRE(n_clusters <= 65535, "too many clusters");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    for (int i = 0; i <= n_clusters; ++i)
        MODEST_REQUIRE(offsets_host[i] >= 0 && (i == 0 || offsets_host[i] >= offsets_host[i - 1]),
                       "offsets must be non-decreasing");
    // closeness: leaf list + combine program of every cluster's pairwise-sum tree
    std::vector<int> leaf, leafBase, progBase;
    std::vector<unsigned char> prog;
    int maxLeaves = 1, maxN = 0;
    if (!variance) {
        leafBase.push_back(0);
        progBase.push_back(0);
        for (int c = 0; c < n_clusters; ++c) {
            const int n = offsets_host[c + 1] - offsets_host[c];
            for (int off = 0; off < n; off += NP_BUFSIZE) {   // numpy's reduction buffers, added left to right
                pw_tree_host(off, std::min(NP_BUFSIZE, n - off), leaf, prog);
                if (off > 0) prog.push_back(1);
            }
            leafBase.push_back((int)(leaf.size() / 2));
            progBase.push_back((int)prog.size());
            maxLeaves = std::max(maxLeaves, leafBase[c + 1] - leafBase[c]);
            maxN = std::max(maxN, n);
        }
    }
    // upload block: [offsets | cos,sin | leafBase | progBase | leaves | prog], one copy
    const size_t u_off = 0, u_cs = arena_sz((size_t)(n_clusters + 1) * 4);
    const size_t u_lb = u_cs + arena_sz((size_t)n_angles * 16);
    const size_t u_pb = u_lb + arena_sz(leafBase.size() * 4);
    const size_t u_lf = u_pb + arena_sz(progBase.size() * 4);
    const size_t u_pg = u_lf + arena_sz(leaf.size() * 4);
    const size_t u_c90 = u_pg + arena_sz(prog.size());
    const size_t u_pts = u_c90 + (cossin90_host ? arena_sz((size_t)n_angles * 16) : 0);
    const size_t b_up = u_pts + (pts_host ? arena_sz((size_t)offsets_host[n_clusters] * 16) : 0);
    const size_t b_rect = rect_host_out ? arena_sz((size_t)n_clusters * 64) : 0;
    const size_t b_beta = arena_sz((size_t)n_clusters * n_angles * 8), b_best = arena_sz((size_t)n_clusters * 4);
    int rc = modest_ctx_reserve(ctx, b_up + b_beta + b_best);
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, b_up + b_best + (beta_host ? b_beta : 0) + b_rect);
    if (rc) return rc;
    char *d = ctx->scratch, *h = ctx->pinned;
    int *d_off = reinterpret_cast<int *>(d + u_off);
    double *d_cs = reinterpret_cast<double *>(d + u_cs);
    double *d_beta = reinterpret_cast<double *>(d + b_up);
    int *d_best = reinterpret_cast<int *>(d + b_up + b_beta);
    int *h_best = reinterpret_cast<int *>(h + b_up);
    double *h_beta = reinterpret_cast<double *>(h + b_up + b_best);
    double *h_rect = rect_host_out ? reinterpret_cast<double *>(h + b_up + b_best + (beta_host ? b_beta : 0)) : nullptr;
    const double *d_c90 = cossin90_host ? reinterpret_cast<const double *>(d + u_c90) : nullptr;
    MODEST_REQUIRE(!rect_host_out || (cossin90_host && !variance), "extents need the heading + pi/2 table (closeness only)");
    memcpy(h + u_off, offsets_host, (size_t)(n_clusters + 1) * 4);
    memcpy(h + u_cs, cossin_host, (size_t)n_angles * 16);
    if (!variance) {
        memcpy(h + u_lb, leafBase.data(), leafBase.size() * 4);
        memcpy(h + u_pb, progBase.data(), progBase.size() * 4);
        if (!leaf.empty()) memcpy(h + u_lf, leaf.data(), leaf.size() * 4);
        if (!prog.empty()) memcpy(h + u_pg, prog.data(), prog.size());
    }
    if (cossin90_host) memcpy(h + u_c90, cossin90_host, (size_t)n_angles * 16);
    if (pts_host) {
        memcpy(h + u_pts, pts_host, (size_t)offsets_host[n_clusters] * 16);
        pts_xz = reinterpret_cast<const double *>(d + u_pts);
    }
    MODEST_HIP_CHECK(hipMemcpyAsync(d, h, b_up, hipMemcpyHostToDevice, stream));
    dim3 grid((n_angles + 127) / 128, n_clusters);
    unsigned *tickets = nullptr;   // the tree kernel picks the best angle itself (a ticket word per cluster)
    if (!variance && n_clusters <= (int)MODEST_ZW_TICKETS) {
        unsigned *zw = nullptr;
        rc = modest_ctx_zero_words(ctx, stream, &zw);
        if (rc) return rc;
        tickets = modest_tickets(zw);
    }
    bool picked = false;
    const int P = maxN > 4096 ? 64 : 16, AB = 256 / P;
    const size_t lds = (size_t)AB * maxLeaves * 8;
    if (variance) {
        variance_kernel<<<grid, 128, 0, stream>>>(pts_xz, d_off, d_cs, n_angles, d_beta);
    } else if (lds <= 48 * 1024) {
        const dim3 g2((n_angles + AB - 1) / AB, n_clusters);
        const int *d_lb = reinterpret_cast<const int *>(d + u_lb), *d_pb = reinterpret_cast<const int *>(d + u_pb);
        const int2 *d_lf = reinterpret_cast<const int2 *>(d + u_lf);
        const unsigned char *d_pg = reinterpret_cast<const unsigned char *>(d + u_pg);
        if (P == 64)
            closeness_tree_kernel<64><<<g2, 256, lds, stream>>>(pts_xz, d_off, d_cs, n_angles, d0, d_lb, d_lf, d_pb,
                                                                d_pg, maxLeaves, d_beta, tickets, h_best, d_c90, h_rect);
        else
            closeness_tree_kernel<16><<<g2, 256, lds, stream>>>(pts_xz, d_off, d_cs, n_angles, d0, d_lb, d_lf, d_pb,
                                                                d_pg, maxLeaves, d_beta, tickets, h_best, d_c90, h_rect);
        picked = tickets != nullptr;
    } else {   // a cluster of > 90 k points: one lane per (cluster, angle)
        closeness_kernel<<<grid, 128, 0, stream>>>(pts_xz, d_off, d_cs, n_angles, d0, d_beta);
    }
    (void)d_best;
    if (!picked) argmax_kernel<<<n_clusters, 64, 0, stream>>>(d_beta, n_clusters, n_angles, h_best);   // pinned host memory
    MODEST_HIP_CHECK(hipGetLastError());
    if (beta_host)
        MODEST_HIP_CHECK(hipMemcpyAsync(h_beta, d_beta, (size_t)n_clusters * n_angles * 8,
                                        hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n_clusters; ++i) best_angle_host[i] = h_best[i];
    if (rect_host_out) {
        MODEST_REQUIRE(picked, "extents are produced by the tree kernel only (cluster too large / too many clusters)");
        memcpy(rect_host_out, h_rect, (size_t)n_clusters * 64);
    }
    if (beta_host)
        for (size_t i = 0; i < (size_t)n_clusters * n_angles; ++i) beta_host[i] = h_beta[i];
    return MODEST_OK;
}

extern "C" int modest_fit_boxes_closeness(modest_ctx *ctx, const double *pts_xz,
                                          const int32_t *offsets_host, int n_clusters,
                                          const double *cossin_host, int n_angles, double d0,
                                          int32_t *best_angle_host, double *beta_host,
                                          void *stream_) {
    return fit_boxes_angles(ctx, 0, pts_xz, offsets_host, n_clusters, cossin_host, n_angles, d0, best_angle_host,
                            beta_host, stream_);
}

extern "C" int modest_fit_boxes_closeness_host(modest_ctx *ctx, const double *pts_xz_host,
                                               const int32_t *offsets_host, int n_clusters,
                                               const double *cossin_host, int n_angles, double d0,
                                               int32_t *best_angle_host, const double *cossin90_host,
                                               double *extents_host, void *stream_) {
    return fit_boxes_angles(ctx, 0, nullptr, offsets_host, n_clusters, cossin_host, n_angles, d0, best_angle_host,
                            nullptr, stream_, pts_xz_host, cossin90_host, extents_host);
}

extern "C" int modest_fit_boxes_variance(modest_ctx *ctx, const double *pts_xz,
                                         const int32_t *offsets_host, int n_clusters,
                                         const double *cossin_host, int n_angles,
                                         int32_t *best_angle_host, double *crit_host, void *stream_) {
    return fit_boxes_angles(ctx, 1, pts_xz, offsets_host, n_clusters, cossin_host, n_angles, 0.0, best_angle_host,
                            crit_host, stream_);
}

extern "C" int modest_fit_boxes_pca(modest_ctx *ctx, const double *pts_xz, const int32_t *offsets_host,
                                    int n_clusters, double *out8_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_clusters >= 0, "bad sizes");
    if (n_clusters == 0) return MODEST_OK;
    MODEST_REQUIRE(pts_xz && offsets_host && out8_host, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t b_off = arena_sz((size_t)(n_clusters + 1) * 4), b_out = arena_sz((size_t)n_clusters * 64);
    int rc = modest_ctx_reserve(ctx, b_off + b_out);
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, b_off + b_out);
    if (rc) return rc;
    int *d_off = reinterpret_cast<int *>(ctx->scratch);
    double *d_out = reinterpret_cast<double *>(ctx->scratch + b_off);
    int *h_off = reinterpret_cast<int *>(ctx->pinned);
    double *h_out = reinterpret_cast<double *>(ctx->pinned + b_off);
    for (int i = 0; i <= n_clusters; ++i) {
        MODEST_REQUIRE(offsets_host[i] >= 0 && (i == 0 || offsets_host[i] >= offsets_host[i - 1]),
                       "offsets must be non-decreasing");
        h_off[i] = offsets_host[i];
    }
    MODEST_HIP_CHECK(hipMemcpyAsync(d_off, h_off, (size_t)(n_clusters + 1) * 4, hipMemcpyHostToDevice, stream));
    pca_kernel<<<n_clusters, 64, 0, stream>>>(pts_xz, d_off, d_out);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)n_clusters * 64, hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < (size_t)n_clusters * 8; ++i) out8_host[i] = h_out[i];
    return MODEST_OK;
}

// the lowest point inside each of n_boxes footprints whose points are different arrays (the boxes of a chain of
// scans): pts_rect[k] / n_pts[k] of box k; one launch, one synchronise
int modest_lowest_point_multi(modest_ctx *ctx, const double *const *pts_rect, const int *n_pts, const double *boxes6_host,
                              int n_boxes, double *bottom_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && n_boxes >= 0, "bad arguments");
    if (n_boxes == 0) return MODEST_OK;
    MODEST_REQUIRE(pts_rect && n_pts && boxes6_host && bottom_host, "NULL buffer");
    MODEST_REQUIRE(n_boxes <= (int)MODEST_ZW_TICKETS, "too many boxes for one launch");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t b_box = arena_sz((size_t)n_boxes * sizeof(Box6)), b_out = arena_sz((size_t)n_boxes * 8);
    const size_t b_src = arena_sz((size_t)n_boxes * sizeof(BoxSrc));
    int rc = modest_ctx_reserve(ctx, arena_sz((size_t)n_boxes * LOW_SPLIT * 8));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, b_box + b_out + b_src);
    if (rc) return rc;
    double *d_part = reinterpret_cast<double *>(ctx->scratch);
    Box6 *h_box = reinterpret_cast<Box6 *>(ctx->pinned);
    double *h_out = reinterpret_cast<double *>(ctx->pinned + b_box);
    BoxSrc *h_src = reinterpret_cast<BoxSrc *>(ctx->pinned + b_box + b_out);
    memcpy(h_box, boxes6_host, (size_t)n_boxes * sizeof(Box6));
    for (int i = 0; i < n_boxes; ++i) {
        MODEST_REQUIRE(n_pts[i] >= 1 && pts_rect[i], "a box without points to look at");
        h_src[i].pts = pts_rect[i];
        h_src[i].n = n_pts[i];
    }
    unsigned *zw = nullptr;
    rc = modest_ctx_zero_words(ctx, stream, &zw);
    if (rc) return rc;
    lowest_kernel<<<dim3(LOW_SPLIT, n_boxes), 1024, 0, stream>>>(nullptr, 0, h_box, n_boxes, d_part, modest_tickets(zw), h_out,
                                                                h_src);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n_boxes; ++i) bottom_host[i] = h_out[i];
    return MODEST_OK;
}

extern "C" int modest_lowest_point(modest_ctx *ctx, const double *pts_rect, int n,
                                   const double *boxes6_host, int n_boxes, double *bottom_host,
                                   void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0 && n_boxes >= 0, "bad sizes");
    if (n_boxes == 0) return MODEST_OK;
    MODEST_REQUIRE(boxes6_host && bottom_host, "NULL buffer");
    MODEST_REQUIRE(n == 0 || pts_rect, "NULL points");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t b_box = arena_sz((size_t)n_boxes * sizeof(Box6)), b_out = arena_sz((size_t)n_boxes * 8);
    int rc = modest_ctx_reserve(ctx, arena_sz((size_t)n_boxes * LOW_SPLIT * 8));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, b_box + b_out);
    if (rc) return rc;
    double *d_part = reinterpret_cast<double *>(ctx->scratch);
    Box6 *h_box = reinterpret_cast<Box6 *>(ctx->pinned);
    double *h_out = reinterpret_cast<double *>(ctx->pinned + b_box);
    memcpy(h_box, boxes6_host, (size_t)n_boxes * sizeof(Box6));
    if (n == 0) {
        for (int i = 0; i < n_boxes; ++i) bottom_host[i] = -INFINITY;
        return MODEST_OK;
    }
    unsigned *zw = nullptr;
    rc = modest_ctx_zero_words(ctx, stream, &zw);
    if (rc) return rc;
    for (int b0 = 0; b0 < n_boxes; b0 += (int)MODEST_ZW_TICKETS) {   // a ticket word per box of the launch
        const int nb = std::min(n_boxes - b0, (int)MODEST_ZW_TICKETS);
        lowest_kernel<<<dim3(LOW_SPLIT, nb), 1024, 0, stream>>>(pts_rect, n, h_box + b0, nb, d_part + (size_t)b0 * LOW_SPLIT,
                                                               modest_tickets(zw), h_out + b0);
    }
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n_boxes; ++i) bottom_host[i] = h_out[i];
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_boxfit(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(argmax_kernel));
}
