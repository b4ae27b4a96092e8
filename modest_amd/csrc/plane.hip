// Ground-plane stage for gfx950: estimate_plane's candidate selection and the
// RANSAC inner loops (utils/pointcloud_utils.py:44-65 -> sklearn
// RANSACRegressor defaults: residual threshold = MAD(z), inliers = |z - pred|
// <= thr, all float32), the final least-squares refit, and above_plane +
// range mask (pointcloud_utils.py:68-81, generate_mask.py:57-65).
//
// Every O(N) or O(N*trials) loop is a kernel with deterministic (fixed-order)
// reductions.  The sequential accept rule and the random triplets are host code
// (a few dozen scalar decisions): of the library since round 2 (ransac_host.h,
// modest_ransac_plane: numpy's MT19937 stream as sklearn consumes it), or of the
// caller (modest_ransac_trials / _refit with its own triplets).
#include "common.h"
#include "compact.h"
#include "mask_chain.h"
#include "mask_pred.h"
#include "ransac_host.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <utility>
#include <cstring>
#include <vector>

using namespace modest;

namespace {

// ---- candidate selection -----------------------------------------------------
__global__ __launch_bounds__(1024) void candidates_kernel(const float *__restrict__ pts, int n,
                                                          int stride, float max_hs, float xlo,
                                                          float xhi, float ylo, float yhi,
                                                          float *__restrict__ cand,
                                                          int *__restrict__ cand_idx,
                                                          unsigned long long *state, int *n_cand) {
    const unsigned blk = compact_ticket(state);
    const long long i = (long long)blk * 1024 + threadIdx.x;
    bool keep = false;
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float *p = pts + i * stride;
        x = p[0];
        y = p[1];
        z = p[2];
        keep = (z < max_hs) && (x > xlo) && (x < xhi) && (y > ylo) && (y < yhi);
    }
    const unsigned long long dst = compact_offset(keep, blk, gridDim.x, state, n_cand);
    if (keep) {
        cand[3 * dst + 0] = x;
        cand[3 * dst + 1] = y;
        cand[3 * dst + 2] = z;
        if (cand_idx) cand_idx[dst] = (int)i;
    }
}

// Both candidate selections of a scan (estimate_plane's and filter_labels') in one pass over the
// rows: two ordered compactions side by side (own state block each, one shared block ticket).
struct CandSpec {
    float max_hs, xlo, xhi, ylo, yhi;
};
__device__ __forceinline__ void candidates2_kernel_body(const float *__restrict__ pts, int n, int stride, const CandSpec &A,
                                                        const CandSpec &B, float *__restrict__ candA,
                                                        float *__restrict__ candB, unsigned long long *stateA,
                                                        unsigned long long *stateB, int *n_out /* [2] */,
                                                        const unsigned gx) {
    const unsigned blk = compact_ticket(stateA);
    const long long i = (long long)blk * 1024 + threadIdx.x;
    bool ka = false, kb = false;
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float *p = pts + i * stride;
        x = p[0];
        y = p[1];
        z = p[2];
        ka = (z < A.max_hs) && (x > A.xlo) && (x < A.xhi) && (y > A.ylo) && (y < A.yhi);
        kb = (z < B.max_hs) && (x > B.xlo) && (x < B.xhi) && (y > B.ylo) && (y < B.yhi);
    }
    const unsigned long long da = compact_offset(ka, blk, gx, stateA, n_out);
    if (ka) {
        candA[3 * da + 0] = x;
        candA[3 * da + 1] = y;
        candA[3 * da + 2] = z;
    }
    const unsigned long long db = compact_offset(kb, blk, gx, stateB, n_out + 1);
    if (kb) {
        candB[3 * db + 0] = x;
        candB[3 * db + 1] = y;
        candB[3 * db + 2] = z;
    }
}
__global__ __launch_bounds__(1024) void candidates2_kernel(const float *__restrict__ pts, int n, int stride, CandSpec A,
                                                           CandSpec B, float *__restrict__ candA,
                                                           float *__restrict__ candB, unsigned long long *stateA,
                                                           unsigned long long *stateB, int *n_out /* [2] */) {
    candidates2_kernel_body(pts, n, stride, A, B, candA, candB, stateA, stateB, n_out, gridDim.x);
}

// the selections of a chain of scans in one launch: the scan is blockIdx.y
struct CandLaunch {
    const float *pts;
    float *candA, *candB;
    unsigned long long *stateA, *stateB;
    int *n_out;
    int n, stride, nblk, pad;
};
__global__ __launch_bounds__(1024) void cdb_candidates2(const CandLaunch *__restrict__ tab, CandSpec A, CandSpec B) {
    const CandLaunch &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nblk) return;   // before the block ticket: only the scan's own blocks take one
    candidates2_kernel_body(S.pts, S.n, S.stride, A, B, S.candA, S.candB, S.stateA, S.stateB, S.n_out, (unsigned)S.nblk);
}

// ---- MAD(z): median and median absolute deviation, one workgroup ---------------------
// One workgroup = one CU, and this kernel is bound by that CU's instruction issue (64 lanes x
// 1 instruction per cycle for 16 wavefronts), so everything is about instructions per element
// and pass: the order-preserving keys are computed once per median and kept in registers, a
// select is two or three 2048-bin histogram rounds of three barriers each, the scan zeroes the
// histogram for the next round as it reads it.
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

constexpr int MAD_R = 32;   // element r*1024 + tid of thread tid, up to 32768 candidates in registers
struct KeysReg {
    unsigned k[MAD_R];
};
struct KeysGlobal {   // fallback for more candidates: keys recomputed from memory in every pass
    const float *cand;
    float center;
    int mode;
};

// calls f(key, valid) for every element slot of this thread, wave-uniform trip count
template <class F>
__device__ __forceinline__ void for_keys(const KeysReg &K, int n, F f) {
#pragma unroll
    for (int r = 0; r < MAD_R; ++r) {
        if (r * 1024 >= n) break;
        f(K.k[r], r * 1024 + (int)threadIdx.x < n);
    }
}
template <class F>
__device__ __forceinline__ void for_keys(const KeysGlobal &K, int n, F f) {
    for (int base = 0; base < n; base += 8 * 1024) {   // eight strided loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 1024 + (int)threadIdx.x;
            v[u] = i < n ? K.cand[3 * (size_t)i + 2] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (base + u * 1024 >= n) break;   // wave-uniform
            f(f2key(K.mode ? fabsf(v[u] - K.center) : v[u]), base + u * 1024 + (int)threadIdx.x < n);
        }
    }
}

struct MadShared {
    unsigned hist[2048];   // zero between rounds
    unsigned wsum[16];
    unsigned sel[3], knew[3];
    unsigned cntLess, maxLessKey;
    unsigned kmin, kmax;   // 0xffffffff / 0 between selects
};

// exact k-th smallest key (0-based).  S.hist is zero on entry and on exit.
// Radix select over the RANGE the keys actually span: the first round spreads [min key, max key]
// over the 2048 bins, every later round the selected bin, until a bin is one key value (at most
// three rounds; two for ground heights).  Fixed bit fields put a scan's ground heights -- one or
// two float exponents -- into a handful of bins, where same-address LDS atomics serialise; spread
// bins take one plain atomic per element.
template <class KS>
__device__ __forceinline__ unsigned select_kth(const KS &K, int n, unsigned k, MadShared &S) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned mn = 0xffffffffu, mx = 0u;
    for_keys(K, n, [&](unsigned key, bool valid) {
        if (valid) {
            mn = min(mn, key);
            mx = max(mx, key);
        }
    });
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    }
    if (lane == 0) {
        atomicMin(&S.kmin, mn);
        atomicMax(&S.kmax, mx);
    }
    __syncthreads();
    unsigned lo = S.kmin;
    const unsigned range = S.kmax - lo;
    int shift = max(0, 32 - (int)__clz(range) - 11);   // (range >> shift) < 2048
    for (int ps = 0;; ++ps) {
        for_keys(K, n, [&](unsigned key, bool valid) {
            const unsigned b = (key - lo) >> shift;
            if (valid && key >= lo && b < 2048u) atomicAdd(&S.hist[b], 1u);
        });
        __syncthreads();
        if (ps == 0 && tid == 0) {   // everybody has read the range: reset it for the next select
            S.kmin = 0xffffffffu;
            S.kmax = 0u;
        }
        // block scan, two bins per thread; the bins are cleared for the next round on the way
        const unsigned v0 = S.hist[2 * tid], v1 = S.hist[2 * tid + 1];
        S.hist[2 * tid] = 0;
        S.hist[2 * tid + 1] = 0;
        unsigned inc = v0 + v1;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) S.wsum[w] = inc;
        __syncthreads();
        unsigned base = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) base += q < w ? S.wsum[q] : 0u;
        const unsigned incl = base + inc, excl = incl - v0 - v1;
        if (k >= excl && k < excl + v0) {
            S.sel[ps] = 2 * tid;
            S.knew[ps] = k - excl;
        } else if (k >= excl + v0 && k < incl) {
            S.sel[ps] = 2 * tid + 1;
            S.knew[ps] = k - excl - v0;
        }
        __syncthreads();
        lo += S.sel[ps] << shift;
        k = S.knew[ps];
        if (shift == 0) break;
        shift = max(0, shift - 11);
    }
    return lo;
}

// numpy.median: odd -> middle element; even -> float32 mean of the two middle ones.  The lower
// middle is the largest value below the upper middle b unless b is duplicated across the middle.
template <class KS>
__device__ __forceinline__ float median_np(const KS &K, int n, MadShared &S) {
    const unsigned bkey = select_kth(K, n, (unsigned)(n / 2), S);
    const float b = key2f(bkey);
    if (n & 1) return b;
    unsigned c = 0, mk = 0;
    for_keys(K, n, [&](unsigned key, bool valid) {
        if (valid && key < bkey) {
            ++c;
            mk = max(mk, key);
        }
    });
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o);
        mk = max(mk, (unsigned)__shfl_xor((int)mk, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&S.cntLess, c);
        atomicMax(&S.maxLessKey, mk);
    }
    __syncthreads();
    const float a = (S.cntLess == (unsigned)(n / 2)) ? key2f(S.maxLessKey) : b;
    __syncthreads();
    if (threadIdx.x == 0) {   // for the next median (its select has barriers before these are used)
        S.cntLess = 0;
        S.maxLessKey = 0;
    }
    return (a + b) / 2.0f;
}

// up to MAD_SETS candidate sets per launch, one workgroup (= one CU) each: the two ground-plane fits
// of a scan get their thresholds from one launch, side by side
constexpr int MAD_SETS = 16;   // (the two sets of every scan of a chain of 8 in one launch)
struct MadArgs {
    const float *cand[MAD_SETS];
    int n[MAD_SETS];
    const int *n_dev[MAD_SETS];   // when set: the size is read from here (written by an earlier kernel)
    int *n_host[MAD_SETS];        // when set: ... and mirrored to this pinned host word
    float *out[MAD_SETS];   // [median, mad] each; device or pinned host memory
};

__global__ __launch_bounds__(1024) void mad_kernel(MadArgs A) {
    __shared__ MadShared S;
    const float *__restrict__ cand = A.cand[blockIdx.x];
    const int n = A.n_dev[blockIdx.x] ? *A.n_dev[blockIdx.x] : A.n[blockIdx.x];
    float *out = A.out[blockIdx.x];
    const int tid = threadIdx.x;
    if (A.n_host[blockIdx.x] && tid == 0) *A.n_host[blockIdx.x] = n;   // pinned mirror of a device-side size
    if (n <= 0) {   // an empty set has no threshold (the fit raises on the host, as the reference does)
        if (tid == 0) out[0] = out[1] = NAN;
        return;
    }
    S.hist[2 * tid] = 0;
    S.hist[2 * tid + 1] = 0;
    if (tid == 0) {
        S.cntLess = 0;
        S.maxLessKey = 0;
        S.kmin = 0xffffffffu;
        S.kmax = 0u;
    }
    float med, mad;
    if (n <= MAD_R * 1024) {
        KeysReg K;
#pragma unroll
        for (int r = 0; r < MAD_R; ++r) {
            const int i = r * 1024 + tid;
            K.k[r] = f2key(i < n ? cand[3 * (size_t)i + 2] : 0.f);
        }
        __syncthreads();
        med = median_np(K, n, S);
#pragma unroll
        for (int r = 0; r < MAD_R; ++r) K.k[r] = f2key(fabsf(key2f(K.k[r]) - med));   // float32, as numpy (the key map is a bijection)
        mad = median_np(K, n, S);
    } else {
        __syncthreads();
        med = median_np(KeysGlobal{cand, 0.f, 0}, n, S);
        mad = median_np(KeysGlobal{cand, med, 1}, n, S);
    }
    if (tid == 0) {
        out[0] = med;
        out[1] = mad;
    }
}

// ---- trial scoring -------------------------------------------------------------
constexpr int SCORE_THREADS = 256;
constexpr int SCORE_WAVES = SCORE_THREADS / 64;

__device__ __forceinline__ float plane_pred(float x, float y, float c0, float c1, float b) {
    // float32 X @ coef + intercept as the BLAS gemv rounds it: fma chain, then add
    return fmaf(y, c1, x * c0) + b;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// exact-fit plane through three candidates (see fit_kernel)
__device__ __forceinline__ void fit_triplet(const float *__restrict__ cand, int n, int i0, int i1, int i2, float *c0,
                                            float *c1, float *b) {
    const int idx[3] = {i0, i1, i2};
    double x[3], y[3], z[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = min(max(idx[j], 0), n - 1);
        x[j] = cand[3 * (size_t)i];
        y[j] = cand[3 * (size_t)i + 1];
        z[j] = cand[3 * (size_t)i + 2];
    }
    const double mx = (x[0] + x[1] + x[2]) / 3.0, my = (y[0] + y[1] + y[2]) / 3.0, mz = (z[0] + z[1] + z[2]) / 3.0;
    double sxx = 0, sxy = 0, syy = 0, sxz = 0, syz = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double dx = x[j] - mx, dy = y[j] - my, dz = z[j] - mz;
        sxx += dx * dx;
        sxy += dx * dy;
        syy += dy * dy;
        sxz += dx * dz;
        syz += dy * dz;
    }
    const double det = sxx * syy - sxy * sxy;
    double a0 = 0.0, a1 = 0.0;
    if (fabs(det) > 1e-12 * fmax(sxx * syy, 1e-300)) {
        a0 = (sxz * syy - syz * sxy) / det;
        a1 = (syz * sxx - sxz * sxy) / det;
    } else if (sxx + syy > 0.0) {
        // Collinear in xy (a duplicated point in the triplet, in practice): sklearn's LinearRegression ->
        // lstsq returns the MINIMUM-NORM solution of the rank-one centred system, a legitimate trial model
        // that can win (a line on flat ground extends to the ground plane).  w = v (v . X'z) / lambda with
        // (lambda, v) the leading eigenpair of X'X.  (Three coincident xy: w = 0, b = mean z.)
        const double h = 0.5 * (sxx - syy), l1 = 0.5 * (sxx + syy) + sqrt(h * h + sxy * sxy);
        double vx = l1 - syy, vy = sxy;
        if (fabs(l1 - sxx) > fabs(l1 - syy)) {
            vx = sxy;
            vy = l1 - sxx;
        }
        const double nv = sqrt(vx * vx + vy * vy);
        if (nv > 0.0) {
            vx /= nv;
            vy /= nv;
            const double pr = (vx * sxz + vy * syz) / l1;
            a0 = vx * pr;
            a1 = vy * pr;
        }
    }
    *c0 = (float)a0;
    *c1 = (float)a1;
    *b = (float)(mz - a0 * mx - a1 * my);
}

// The triplets of a batch travel as a kernel argument (no upload, no separate fit launch).
constexpr int TRIP_MAX = 64;
struct TripArg {
    int t[3 * TRIP_MAX];
};

// One block scores SCORE_PTS candidates against SCORE_KG trials: every thread keeps its
// SCORE_PPT points in registers, sums its own inliers first and the wavefront reduces once per
// trial; wavefronts write their own partial rows (no block barrier anywhere).
// partial[((blk*SCORE_WAVES + w)*K + k)*4 + {0:count,1:sse,2:sy,3:syy}]
constexpr int SCORE_PPT = 4;
constexpr int SCORE_PTS = SCORE_THREADS * SCORE_PPT;
constexpr int SCORE_KG = 8;
template <bool FUSED, class TRIP>
__device__ __forceinline__ bool score_kernel_body(const float *__restrict__ cand, int n,
                                                              const float *__restrict__ models,
                                                              int K, const float *__restrict__ thr_ptr,
                                                              float thr_val /* used when thr_ptr is NULL */,
                                                              double *partial, const TRIP &trip,
                                                              float *__restrict__ models_host, unsigned *ticket,
                                                              double *__restrict__ out /* K*4, may be pinned host memory */,
                                                              const float *__restrict__ thr_src,
                                                              float *__restrict__ thr_dst, const unsigned bx, const unsigned by, const unsigned gx, const unsigned gy) {
    __shared__ float sm[SCORE_KG][3];
    __shared__ double red[SCORE_WAVES][SCORE_KG][4];
    __shared__ unsigned last_s;
    if (FUSED) {   // the block fits its own SCORE_KG planes; the first block column reports them
        const int k = by * SCORE_KG + (int)threadIdx.x;
        if (threadIdx.x < SCORE_KG && k < K) {
            float c0, c1, b;
            fit_triplet(cand, n, trip.t[3 * k], trip.t[3 * k + 1], trip.t[3 * k + 2], &c0, &c1, &b);
            sm[threadIdx.x][0] = c0;
            sm[threadIdx.x][1] = c1;
            sm[threadIdx.x][2] = b;
            if (bx == 0) {
                models_host[3 * k] = c0;
                models_host[3 * k + 1] = c1;
                models_host[3 * k + 2] = b;
            }
        }
        __syncthreads();
    }
    const float thr = thr_ptr ? *thr_ptr : thr_val;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float x[SCORE_PPT], y[SCORE_PPT], z[SCORE_PPT];
    bool valid[SCORE_PPT];
#pragma unroll
    for (int p = 0; p < SCORE_PPT; ++p) {
        // consecutive lanes take consecutive points, as the summation order has always been
        const int i = bx * SCORE_PTS + (w * SCORE_PPT + p) * 64 + lane;
        valid[p] = i < n;
        x[p] = valid[p] ? cand[3 * (size_t)i] : 0.f;
        y[p] = valid[p] ? cand[3 * (size_t)i + 1] : 0.f;
        z[p] = valid[p] ? cand[3 * (size_t)i + 2] : 0.f;
    }
    const int k0 = by * SCORE_KG, k1 = min(k0 + SCORE_KG, K);
    double *row = partial + ((size_t)bx * K) * 4;   // one row per block (the wavefronts combine in LDS)
    for (int k = k0; k < k1; ++k) {
        const float c0 = FUSED ? sm[k - k0][0] : models[3 * k], c1 = FUSED ? sm[k - k0][1] : models[3 * k + 1];
        const float b = FUSED ? sm[k - k0][2] : models[3 * k + 2];
        unsigned cnt = 0;
        double sse = 0.0, sy = 0.0, syy = 0.0;
#pragma unroll
        for (int p = 0; p < SCORE_PPT; ++p) {
            const float res = fabsf(z[p] - plane_pred(x[p], y[p], c0, c1, b));
            const bool in = valid[p] && (res <= thr);
            const double r = in ? (double)res : 0.0, zz = in ? (double)z[p] : 0.0;
            cnt += (unsigned)__popcll(__ballot(in));
            sse += r * r;
            sy += zz;
            syy += zz * zz;
        }
        sse = wave_sum(sse);
        sy = wave_sum(sy);
        syy = wave_sum(syy);
        if (lane == 0) {
            red[w][k - k0][0] = (double)cnt;
            red[w][k - k0][1] = sse;
            red[w][k - k0][2] = sy;
            red[w][k - k0][3] = syy;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < (k1 - k0) * 4) {   // the four wavefronts' sums, in wavefront order
        const int kk = threadIdx.x >> 2, q = threadIdx.x & 3;
        double s = 0.0;
        for (int ww = 0; ww < SCORE_WAVES; ++ww) s += red[ww][kk][q];
        __hip_atomic_store(row + 4 * (k0 + kk) + q, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        modest_drain_stores();
    }
    // The block that finishes last (ticket, left at zero) adds the partial rows, one output per thread in
    // row order (consecutive threads read consecutive words), and writes the totals.  No __threadfence():
    // on this multi-XCD part it writes back / invalidates a whole L2.  The partial rows are agent-scope
    // atomic stores (write-through), drained by the storing wavefronts before the barrier (common.h); the
    // last block reads them with agent-scope loads, eight in flight per thread.
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(ticket, 1u) == gx * gy - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s) return false;
    const int nrows = gx;
    for (int id = threadIdx.x; id < K * 4; id += SCORE_THREADS) {
        double s = 0.0;
        for (int r0 = 0; r0 < nrows; r0 += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = r0 + u < nrows ? __hip_atomic_load(partial + (size_t)(r0 + u) * K * 4 + id, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        out[id] = s;
    }
    if (threadIdx.x < 2 && thr_dst) thr_dst[threadIdx.x] = thr_src[threadIdx.x];
    if (threadIdx.x == 0) *ticket = 0u;
    return true;   // (the block that wrote the totals)
}
template <bool FUSED>
__global__ __launch_bounds__(SCORE_THREADS) void score_kernel(const float *__restrict__ cand, int n,
                                                              const float *__restrict__ models,
                                                              int K, const float *__restrict__ thr_ptr,
                                                              float thr_val /* used when thr_ptr is NULL */,
                                                              double *partial, TripArg trip,
                                                              float *__restrict__ models_host, unsigned *ticket,
                                                              double *__restrict__ out /* K*4, may be pinned host memory */,
                                                              const float *__restrict__ thr_src,
                                                              float *__restrict__ thr_dst) {
    score_kernel_body<FUSED>(cand, n, models, K, thr_ptr, thr_val, partial, trip, models_host, ticket, out, thr_src, thr_dst,
                             blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// exact-fit plane z = c0 x + c1 y + b through the three candidates of every trial: float64,
// centred 2x2 normal equations, rounded to float32 (what LinearRegression stores for float32
// data); a degenerate (collinear in xy) triplet gets lstsq's minimum-norm model (fit_triplet)
__global__ void fit_kernel(const float *__restrict__ cand, int n, const int *__restrict__ trip, int K,
                           float *__restrict__ models, float *__restrict__ models_host) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float c0, c1, b;
    fit_triplet(cand, n, trip[3 * k], trip[3 * k + 1], trip[3 * k + 2], &c0, &c1, &b);
    models[3 * k] = c0;
    models[3 * k + 1] = c1;
    models[3 * k + 2] = b;
    if (models_host) {   // pinned host memory: the caller reads it after the stream sync, no copy kernel
        models_host[3 * k] = c0;
        models_host[3 * k + 1] = c1;
        models_host[3 * k + 2] = b;
    }
}

// ---- refit ---------------------------------------------------------------------
// One pass: count and the first and second moments of the inliers about the pivot (0, 0, b) in
// float64 (|x|, |y| <= a few hundred metres, n <= 1e6: centring the moments afterwards loses
// ~1e-9 relative at worst, the plane is compared at 1e-4).  Every block writes its partial sums;
// the block that finishes last (ticket, left at zero) adds the partials in block order and writes
// the totals to pinned host memory.
constexpr int REFIT_NV = 9;   // n, Sx, Sy, Sz, Sxx, Sxy, Syy, Sxz, Syz
__device__ __forceinline__ bool refit_kernel_body(const float *__restrict__ cand, int n, float c0, float c1, float b,
                                                  float thr, double *partial, unsigned *ticket,
                                                  double *__restrict__ out_host, const unsigned bx, const unsigned gx) {
    __shared__ double red[SCORE_WAVES][REFIT_NV];
    __shared__ unsigned last_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = bx * SCORE_THREADS + threadIdx.x;
    double v[REFIT_NV] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n) {
        const float x = cand[3 * (size_t)i], y = cand[3 * (size_t)i + 1], z = cand[3 * (size_t)i + 2];
        if (fabsf(z - plane_pred(x, y, c0, c1, b)) <= thr) {
            const double dx = (double)x, dy = (double)y, dz = (double)z - (double)b;
            v[0] = 1.0;
            v[1] = dx;
            v[2] = dy;
            v[3] = dz;
            v[4] = dx * dx;
            v[5] = dx * dy;
            v[6] = dy * dy;
            v[7] = dx * dz;
            v[8] = dy * dz;
        }
    }
#pragma unroll
    for (int q = 0; q < REFIT_NV; ++q) {
        const double s = wave_sum(v[q]);
        if (lane == 0) red[w][q] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < REFIT_NV) {
        double s = 0.0;
        for (int ww = 0; ww < SCORE_WAVES; ++ww) s += red[ww][threadIdx.x];
        __hip_atomic_store(partial + (size_t)bx * REFIT_NV + threadIdx.x, s, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        modest_drain_stores();   // the write-through stores have landed before the ticket (no __threadfence(): see score_kernel)
    }
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(ticket, 1u) == gx - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s || w != 0) return false;
    double s[REFIT_NV] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int bb = lane; bb < (int)gx; bb += 64) {
        double t[REFIT_NV];
#pragma unroll
        for (int q = 0; q < REFIT_NV; ++q)
            t[q] = __hip_atomic_load(partial + (size_t)bb * REFIT_NV + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < REFIT_NV; ++q) s[q] += t[q];
    }
#pragma unroll
    for (int q = 0; q < REFIT_NV; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < REFIT_NV; ++q) out_host[q] = s[q];
        *ticket = 0u;
    }
    return lane == 0;   // (the lane that wrote the totals)
}
__global__ __launch_bounds__(SCORE_THREADS) void refit_kernel(const float *__restrict__ cand, int n, float c0, float c1,
                                                              float b, float thr, double *partial, unsigned *ticket,
                                                              double *__restrict__ out_host) {
    refit_kernel_body(cand, n, c0, c1, b, thr, partial, ticket, out_host, blockIdx.x, gridDim.x);
}

// ---- the same two kernels for a chain of scans: the fit is blockIdx.z, its arguments come from a device table ----
struct ScoreLaunch {
    const float *cand;
    double *partial;
    float *models_host;
    unsigned *ticket;
    double *out;
    int n, K, gx, gy;
    float thr_val;
    TripArg trip;
};
struct RefitLaunch {
    const float *cand;
    double *partial;
    unsigned *ticket;
    double *out_host;
    int n, gx;
    float c0, c1, b, thr;
};
__global__ __launch_bounds__(SCORE_THREADS) void scb_score(const ScoreLaunch *__restrict__ tab) {
    const ScoreLaunch &S = tab[blockIdx.z];
    if ((int)blockIdx.x >= S.gx || (int)blockIdx.y >= S.gy) return;
    score_kernel_body<true>(S.cand, S.n, nullptr, S.K, nullptr, S.thr_val, S.partial, S.trip, S.models_host, S.ticket, S.out,
                            nullptr, nullptr, blockIdx.x, blockIdx.y, (unsigned)S.gx, (unsigned)S.gy);
}
__global__ __launch_bounds__(SCORE_THREADS) void scb_refit(const RefitLaunch *__restrict__ tab) {
    const RefitLaunch &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.gx) return;
    refit_kernel_body(S.cand, S.n, S.c0, S.c1, S.b, S.thr, S.partial, S.ticket, S.out_host, blockIdx.x, (unsigned)S.gx);
}

// ---- the trial loops on the device (chains of scans, scan_driver.hip) ------------------------------------------
// sklearn's RANSAC loop has three data dependent steps that used to cost a host round trip each: how many trials run
// (_dynamic_max_trials after every accepted model), which triplets they use (the generator is consumed by the
// EXECUTED trials only, and the second fit continues where the first one stopped) and the refit of the winner.  Here
// all of it stays in the stream: rsd_draw runs numpy's MT19937 (masked rejection, tracking selection: ransac_host.h)
// for max_trials triplets and remembers the generator position behind each, rsd_score scores all of them (the block
// that finishes last replays the sequential accept rule and fits the winner), rsd_refit sums the consensus set (its
// last block solves the plane), the second fit starts from the generator position behind the first fit's last
// executed trial, rsd_final writes the advanced generator and the results to pinned host memory.  Results are those
// of the host loop, bit for bit; a trial bound that sits within 1e-7 of an integer (libm of host and device could
// round it to different sides) hands the scan back to the host (MODEST_STAGE_HOST_RULE).
constexpr int RSD_KMAX = 128;
struct RsdFit {   // device work area of one fit
    int trip[3 * RSD_KMAX];
    unsigned used[RSD_KMAX];        // generator words consumed up to and including triplet k (from the fit's start)
    float models[3 * RSD_KMAX];
    double sums[4 * RSD_KMAX];
    double refit[REFIT_NV + 1];
    double plane4[4];
    float best[3];
    int n_trials, status, drawn;
    unsigned words;                 // generator words consumed by the executed trials
};
struct RsdScan {   // one scan of the chain (device table, uploaded per call)
    const float *cand[2];
    const int *n_dev;         // [2] candidate counts (written by the selection kernel)
    const float *mad_dev;     // [2][2] (median, MAD) of the two sets
    RsdFit *fit;              // [2]
    double *partial[2];
    unsigned *ticket;
    modest_rsd_result *res;   // pinned host memory
    double stop_probability;
    int pos0, max_trials;
    unsigned key0[624];
};
struct PtrTrip {
    const int *t;
};

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// one regeneration of the 624 state words by a workgroup of 256 threads: the serial recurrence reads key[kk + 1] (old)
// and key[kk + 397 mod 624] (old for kk < 227, new after), so three sweeps of independent elements + the last word
__device__ __forceinline__ void mt_refill(unsigned *key, int tid) {
    const unsigned U = 0x80000000u, L = 0x7fffffffu, A = 0x9908b0dfu;
    const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 623};
#pragma unroll
    for (int sw = 0; sw < 3; ++sw) {
        const int kk = lo[sw] + tid;
        unsigned v = 0;
        if (kk < hi[sw]) {
            const unsigned y = (key[kk] & U) | (key[kk + 1] & L);
            v = key[sw == 0 ? kk + 397 : kk - 227] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        __syncthreads();
        if (kk < hi[sw]) key[kk] = v;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned y = (key[623] & U) | (key[0] & L);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    __syncthreads();
}
// the generator `words` draws further: numpy's (key, pos), pos = 624 meaning "regenerate before the next draw"
__device__ __forceinline__ int mt_advance(unsigned *key, int pos, unsigned words, int tid) {
    const unsigned total = (unsigned)pos + words;
    const unsigned r = total >= 1u ? (total - 1u) / 624u : 0u;
    for (unsigned q = 0; q < r; ++q) mt_refill(key, tid);
    return (int)(total - 624u * r);
}

// max_trials triplets of fit `which` of every scan: sample_without_replacement(n, 3) by tracking selection =
// RandomState.randint(n) until three distinct indices are found; randint = next word & mask until the value is < n.
// The workgroup tempers and filters 256 words at a time (order kept), until 3 K values are there; triplet k is then
// values 3k..3k+2 -- unless some triplet holds a repeated index (probability ~ 3/n each): then one thread walks the
// values in order, as the generator's consumer does.
constexpr int RSD_VALS = 3 * RSD_KMAX + 512;
__global__ __launch_bounds__(256) void rsd_draw(RsdScan *__restrict__ tab, int which) {
    __shared__ unsigned key[624];
    __shared__ unsigned vals[RSD_VALS], vpos[RSD_VALS], wtot[4];
    __shared__ int s_ntrip, s_stop;
    RsdScan &S = tab[blockIdx.x];
    RsdFit &F = S.fit[which];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = S.n_dev[which];
    if (tid == 0) {
        int st = 0;
        if (which == 1 && S.fit[0].status != 0) st = -1;                  // the first fit ended the scan
        else if (S.n_dev[0] <= 300 || S.n_dev[1] <= 300) st = MODEST_STAGE_SMALL_SET;   // sklearn's other selection methods: host
        F.status = st;
        F.n_trials = 0;
        F.words = 0u;
        F.drawn = 0;
        // a fit that ends without a plane (small set, no consensus, degenerate refit) leaves a DEFINED one behind: the chain's mask
        // kernel reads fit[0].plane4 of every scan before the host has seen the status and dropped the scan
        F.plane4[0] = 0.0, F.plane4[1] = 0.0, F.plane4[2] = 1.0, F.plane4[3] = 0.0;
        s_stop = st;
        s_ntrip = 0;
    }
    for (int i = tid; i < 624; i += 256) key[i] = S.key0[i];
    __syncthreads();
    if (s_stop != 0) return;
    int pos = S.pos0;
    if (which == 1) pos = mt_advance(key, pos, S.fit[0].words, tid);
    const int K = min(S.max_trials, RSD_KMAX);
    const unsigned rng = (unsigned)n - 1u;
    unsigned mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    unsigned consumed = 0;
    int nv = 0, need = 3 * K;
    bool first = true;
    for (;;) {
        while (nv < need && nv + 256 <= RSD_VALS) {
            if (pos >= 624) {
                mt_refill(key, tid);
                pos = 0;
            }
            const int m = min(256, 624 - pos);
            unsigned v = 0;
            bool ok = false;
            if (tid < m) {
                v = mt_temper(key[pos + tid]) & mask;
                ok = v <= rng;
            }
            const unsigned long long bal = __ballot(ok);
            if (lane == 0) wtot[w] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned base = 0, all = 0;
            for (int q = 0; q < 4; ++q) {
                if (q < w) base += wtot[q];
                all += wtot[q];
            }
            if (ok) {
                const unsigned at = (unsigned)nv + base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                vals[at] = v;
                vpos[at] = consumed + (unsigned)tid + 1u;
            }
            __syncthreads();
            nv += (int)all;
            pos += m;
            consumed += (unsigned)m;
        }
        if (nv < need) {   // (more repeated indices than the value buffer has room for: never; the host loop takes the scan)
            if (tid == 0) F.status = MODEST_STAGE_HOST_RULE;
            return;
        }
        if (first) {
            unsigned a = 0, b = 1, c = 2;
            if (tid < K) a = vals[3 * tid], b = vals[3 * tid + 1], c = vals[3 * tid + 2];
            if (!__syncthreads_or(a == b || a == c || b == c)) {
                if (tid < K) {
                    F.trip[3 * tid] = (int)a;
                    F.trip[3 * tid + 1] = (int)b;
                    F.trip[3 * tid + 2] = (int)c;
                    F.used[tid] = vpos[3 * tid + 2];
                }
                break;
            }
            first = false;
        }
        if (tid == 0) {   // in order, skipping the repeats
            int have = 0, ntrip = 0, t3[3] = {0, 0, 0};
            for (int i = 0; i < nv && ntrip < K; ++i) {
                const int j = (int)vals[i];
                bool dup = false;
                for (int q = 0; q < have; ++q) dup = dup || t3[q] == j;
                if (dup) continue;
                t3[have++] = j;
                if (have == 3) {
                    F.trip[3 * ntrip] = t3[0];
                    F.trip[3 * ntrip + 1] = t3[1];
                    F.trip[3 * ntrip + 2] = t3[2];
                    F.used[ntrip] = vpos[i];
                    ++ntrip;
                    have = 0;
                }
            }
            s_ntrip = ntrip;
        }
        __syncthreads();
        if (s_ntrip >= K) break;
        need = nv + 3 * (K - s_ntrip);   // (every repeat costs one more value)
        __syncthreads();
    }
    if (tid == 0) F.drawn = K;
}

// all drawn trials of fit `which` of every scan (blockIdx.z) in one launch; the block that finishes last replays
// sklearn's sequential accept rule over the totals (ransac_host.h: RansacFit::finish_batch) and fits the winner
__global__ __launch_bounds__(SCORE_THREADS) void rsd_score(RsdScan *__restrict__ tab, int which) {
    RsdScan &S = tab[blockIdx.z];
    RsdFit &F = S.fit[which];
    if (F.status != 0) return;
    const int n = S.n_dev[which], K = F.drawn;
    const int gx = (n + SCORE_PTS - 1) / SCORE_PTS, gy = (K + SCORE_KG - 1) / SCORE_KG;
    if ((int)blockIdx.x >= gx || (int)blockIdx.y >= gy) return;
    const PtrTrip trip{F.trip};
    const bool last = score_kernel_body<true>(S.cand[which], n, nullptr, K, S.mad_dev + 2 * which + 1, 0.f, S.partial[which], trip,
                                              F.models, S.ticket, F.sums, nullptr, nullptr, blockIdx.x, blockIdx.y, (unsigned)gx,
                                              (unsigned)gy);
    if (!last) return;
    // every trial's r2 score and the trial bound it would set if it were accepted, in parallel; then one thread replays the
    // sequential rule over them (compares only)
    __shared__ double s_score[RSD_KMAX], s_dyn[RSD_KMAX];
    __shared__ int s_nk[RSD_KMAX];
    __shared__ unsigned char s_amb[RSD_KMAX];
    __syncthreads();   // the totals of this block's threads
    for (int k = threadIdx.x; k < K; k += SCORE_THREADS) {
        const int nk = (int)F.sums[4 * k];
        s_nk[k] = nk;
        s_score[k] = ransac_r2_from_sums(nk, F.sums[4 * k + 1], F.sums[4 * k + 2], F.sums[4 * k + 3]);
        s_dyn[k] = ransac_dynamic_max_trials(nk, n, S.stop_probability);
        // _dynamic_max_trials: the quotient before the ceil decides; next to an integer the host's libm has the word
        const double eps = 2.220446049250313e-16;
        const double ratio = (double)nk / (double)n;
        const double nom = fmax(eps, 1.0 - S.stop_probability), denom = fmax(eps, 1.0 - pow(ratio, 3.0));
        bool amb = false;
        if (nom != 1.0 && denom != 1.0) {
            const double q = log(nom) / log(denom);
            amb = fabs(q) < (double)S.max_trials + 1.0 && fabs(q - rint(q)) < 1e-7 * fmax(1.0, fabs(q));
        }
        s_amb[k] = amb ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int n_best = 1, n_trials = 0, kbest = -1, status = 0;
    double score_best = -INFINITY, limit = (double)S.max_trials;
    for (int k = 0; k < K; ++k) {
        if (!((double)n_trials < limit)) break;
        ++n_trials;
        const int nk = s_nk[k];
        if (nk < n_best) continue;
        const double score = s_score[k];
        if (nk == n_best && score < score_best) continue;
        n_best = nk;
        score_best = score;
        kbest = k;
        if (s_amb[k]) status = MODEST_STAGE_HOST_RULE;
        limit = fmin(limit, s_dyn[k]);
    }
    F.n_trials = n_trials;
    F.words = n_trials > 0 ? F.used[n_trials - 1] : 0u;
    if (kbest < 0 && status == 0) status = MODEST_STAGE_NO_CONSENSUS;
    if (kbest >= 0) fit_triplet(S.cand[which], n, F.trip[3 * kbest], F.trip[3 * kbest + 1], F.trip[3 * kbest + 2], &F.best[0], &F.best[1], &F.best[2]);
    F.status = status;
}

// the consensus set of the winner: moments (refit_kernel), the plane from them by the block that finishes last
// (modest_ransac_refit_phase's read-back half + scan_driver.hip: plane_from_model)
__global__ __launch_bounds__(SCORE_THREADS) void rsd_refit(RsdScan *__restrict__ tab, int which) {
    RsdScan &S = tab[blockIdx.y];
    RsdFit &F = S.fit[which];
    if (F.status != 0) return;
    const int n = S.n_dev[which];
    const int gx = (n + SCORE_THREADS - 1) / SCORE_THREADS;
    if ((int)blockIdx.x >= gx) return;
    const float b = F.best[2];
    if (!refit_kernel_body(S.cand[which], n, F.best[0], F.best[1], b, S.mad_dev[2 * which + 1], S.partial[which], S.ticket, F.refit,
                           blockIdx.x, (unsigned)gx))
        return;
    const double *h = F.refit;   // (written by this lane)
    const double cnt = h[0];
    const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
    const double mx = h[1] * inv, my = h[2] * inv, mzs = h[3] * inv;
    const double sxx = h[4] - h[1] * mx, sxy = h[5] - h[1] * my, syy = h[6] - h[2] * my;
    const double sxz = h[7] - h[1] * mzs, syz = h[8] - h[2] * mzs;
    const double mz = mzs + (double)b;
    const double det = sxx * syy - sxy * sxy;
    if (!(cnt >= 3.0) || !(fabs(det) > 1e-12 * fmax(sxx * syy, 1e-300))) {
        F.status = MODEST_STAGE_DEGENERATE;
        return;
    }
    const double a0 = (sxz * syy - syz * sxy) / det;
    const double a1 = (syz * sxx - sxz * sxy) / det;
    const double m2 = mz - a0 * mx - a1 * my;
    const double c0 = (double)(float)a0, c1 = (double)(float)a1;   // LinearRegression stores float32
    const float b32 = (float)m2;
    const double norm = sqrt((c0 * c0 + c1 * c1) + 1.0);
    F.plane4[0] = -(c0 / norm);
    F.plane4[1] = -(c1 / norm);
    F.plane4[2] = -(-1.0 / norm);
    F.plane4[3] = -((double)b32 / norm);
}

// results + the generator behind the executed trials of both fits -> pinned host memory
__global__ __launch_bounds__(256) void rsd_final(RsdScan *__restrict__ tab) {
    __shared__ unsigned key[624];
    RsdScan &S = tab[blockIdx.x];
    const RsdFit &A = S.fit[0], &B = S.fit[1];
    modest_rsd_result *R = S.res;
    const int tid = threadIdx.x;
    const int status = A.status != 0 ? A.status : B.status;
    if (tid == 0) {
        R->n_cand[0] = S.n_dev[0];
        R->n_cand[1] = S.n_dev[1];
        R->mad[0] = S.mad_dev[1];
        R->mad[1] = S.mad_dev[3];
        R->n_trials[0] = A.n_trials;
        R->n_trials[1] = B.n_trials;
        R->status = status;
        for (int q = 0; q < 4; ++q) {
            R->plane1[q] = A.plane4[q];
            R->plane2[q] = B.plane4[q];
        }
    }
    if (status != 0) return;   // (the caller's generator stays where it was: the host statement starts over)
    for (int i = tid; i < 624; i += 256) key[i] = S.key0[i];
    __syncthreads();
    const int pos = mt_advance(key, S.pos0, A.words + B.words, tid);
    for (int i = tid; i < 624; i += 256) R->mt_key[i] = key[i];
    if (tid == 0) R->mt_pos = pos;
}

// ---- above_plane + range mask ----------------------------------------------------
__global__ __launch_bounds__(1024) void mask_kernel(const float *__restrict__ pts, int n, int stride,
                                                    MaskParams P, unsigned char *__restrict__ mask,
                                                    float *__restrict__ kept, int *__restrict__ kept_idx,
                                                    unsigned long long *state, int *n_kept) {
    const unsigned blk = compact_ticket(state);
    const long long i = (long long)blk * 1024 + threadIdx.x;
    bool keep = false;
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float *p = pts + i * stride;
        x = p[0];
        y = p[1];
        z = p[2];
        keep = mask_keep(P, x, y, z);
        mask[i] = keep ? 1 : 0;
    }
    const unsigned long long dst = compact_offset(keep, blk, gridDim.x, state, n_kept);
    if (keep && kept) {
        kept[3 * dst + 0] = x;
        kept[3 * dst + 1] = y;
        kept[3 * dst + 2] = z;
        if (kept_idx) kept_idx[dst] = (int)i;
    }
}

}  // namespace

extern "C" int modest_plane_candidates(modest_ctx *ctx, const float *pts, int n, int stride,
                                       float max_hs, float xlo, float xhi, float ylo, float yhi,
                                       float *cand, int32_t *cand_idx, int32_t *n_cand,
                                       void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0 && (stride == 3 || stride == 4), "bad n/stride");
    MODEST_REQUIRE(n_cand != nullptr, "n_cand is NULL");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {   // otherwise the last block of the kernel writes the total
        MODEST_HIP_CHECK(hipMemsetAsync(n_cand, 0, sizeof(int32_t), stream));
        return MODEST_OK;
    }
    MODEST_REQUIRE(pts && cand, "NULL buffer");
    const int nblk = (n + 1023) / 1024;
    unsigned long long *state = nullptr;
    int rc = modest_ctx_compact_state(ctx, (size_t)nblk, stream, &state);
    if (rc) return rc;
    candidates_kernel<<<nblk, 1024, 0, stream>>>(pts, n, stride, max_hs, xlo, xhi, ylo, yhi,
                                                             cand, cand_idx, state, n_cand);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_plane_prepare(modest_ctx *ctx, const float *pts, int n, int stride, const float *specs10,
                                    float *candA, float *candB, int32_t *n_cand2_host, float *mad2_host,
                                    void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && specs10 && n_cand2_host && mad2_host, "NULL argument");
    MODEST_REQUIRE(n >= 0 && (stride == 3 || stride == 4), "bad n/stride");
    n_cand2_host[0] = n_cand2_host[1] = 0;
    mad2_host[0] = mad2_host[1] = NAN;
    if (n == 0) return MODEST_OK;
    MODEST_REQUIRE(pts && candA && candB, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = modest_ctx_reserve_pinned(ctx, 64);
    if (rc) return rc;
    const int nblk = (n + 1023) / 1024;
    unsigned long long *state = nullptr;
    rc = modest_ctx_compact_state(ctx, 2 * (size_t)nblk + 2, stream, &state);
    if (rc) return rc;
    rc = modest_ctx_reserve(ctx, 256);
    if (rc) return rc;
    int *d_n = reinterpret_cast<int *>(ctx->scratch);   // [2] counts: written by the selection, read by the MAD kernel
    CandSpec A{specs10[0], specs10[1], specs10[2], specs10[3], specs10[4]};
    CandSpec B{specs10[5], specs10[6], specs10[7], specs10[8], specs10[9]};
    int *h_n = reinterpret_cast<int *>(ctx->pinned);          // [2] counts, pinned: valid after the sync
    float *h_mad = reinterpret_cast<float *>(ctx->pinned + 16);   // [median, mad] x 2
    candidates2_kernel<<<nblk, 1024, 0, stream>>>(pts, n, stride, A, B, candA, candB, state, state + 2 + nblk, d_n);
    MadArgs M{};
    M.cand[0] = candA;
    M.cand[1] = candB;
    M.n_dev[0] = d_n;
    M.n_dev[1] = d_n + 1;
    M.n_host[0] = h_n;
    M.n_host[1] = h_n + 1;
    M.out[0] = h_mad;
    M.out[1] = h_mad + 2;
    mad_kernel<<<2, 1024, 0, stream>>>(M);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    n_cand2_host[0] = h_n[0];
    n_cand2_host[1] = h_n[1];
    mad2_host[0] = h_mad[1];
    mad2_host[1] = h_mad[3];
    return MODEST_OK;
}

// modest_plane_prepare for a chain of scans: every scan's selection is its own launch in its own context, the 2 B MAD
// thresholds come from ONE launch (a workgroup per set, up to MAD_SETS sets) and one synchronise
int modest_plane_prepare_chain(modest_ctx *const *ctxs, const float *const *pts, const int *n, const int *stride, int B,
                               const float *specs10, float *const *candA, float *const *candB, int32_t *n_cand2_host,
                               float *mad2_host, hipStream_t stream) {
    MODEST_REQUIRE(ctxs && pts && n && stride && specs10 && candA && candB && n_cand2_host && mad2_host && B >= 1, "bad chain");
    CandSpec SA{specs10[0], specs10[1], specs10[2], specs10[3], specs10[4]};
    CandSpec SB{specs10[5], specs10[6], specs10[7], specs10[8], specs10[9]};
    MadArgs M{};
    int used = 0, maxblk = 1;
    std::vector<CandLaunch> sel;
    std::vector<std::pair<MadArgs, int>> mads;
    for (int s = 0; s < B; ++s) {
        modest_ctx *ctx = ctxs[s];
        MODEST_REQUIRE(ctx && pts[s] && candA[s] && candB[s] && n[s] >= 1, "bad scan of the chain");
        int rc = modest_ctx_reserve_pinned(ctx, 64);
        if (rc) return rc;
        const int nblk = (n[s] + 1023) / 1024;
        unsigned long long *state = nullptr;
        rc = modest_ctx_compact_state(ctx, 2 * (size_t)nblk + 2, stream, &state);
        if (rc) return rc;
        rc = modest_ctx_reserve(ctx, 256);
        if (rc) return rc;
        int *d_n = reinterpret_cast<int *>(ctx->scratch);
        int *h_n = reinterpret_cast<int *>(ctx->pinned);
        float *h_mad = reinterpret_cast<float *>(ctx->pinned + 16);
        CandLaunch L;
        L.pts = pts[s], L.candA = candA[s], L.candB = candB[s], L.stateA = state, L.stateB = state + 2 + nblk, L.n_out = d_n;
        L.n = n[s], L.stride = stride[s], L.nblk = nblk, L.pad = 0;
        sel.push_back(L);
        maxblk = std::max(maxblk, nblk);
        if (used + 2 > MAD_SETS) {
            mads.push_back({M, used});
            M = MadArgs{};
            used = 0;
        }
        M.cand[used] = candA[s];
        M.cand[used + 1] = candB[s];
        M.n_dev[used] = d_n;
        M.n_dev[used + 1] = d_n + 1;
        M.n_host[used] = h_n;
        M.n_host[used + 1] = h_n + 1;
        M.out[used] = h_mad;
        M.out[used + 1] = h_mad + 2;
        used += 2;
    }
    if (used) mads.push_back({M, used});
    {   // every scan's selection in ONE launch (table through the first context), then the thresholds
        char *d = nullptr, *h = nullptr;
        const size_t bytes = sel.size() * sizeof(CandLaunch);
        int rc = modest_ctx_chain_tab(ctxs[0], bytes, &d);
        if (rc) return rc;
        rc = modest_ctx_stage_slot(ctxs[0], bytes, reinterpret_cast<void **>(&h));
        if (rc) return rc;
        memcpy(h, sel.data(), bytes);
        MODEST_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
        rc = modest_ctx_stage_commit(ctxs[0], stream);
        if (rc) return rc;
        cdb_candidates2<<<dim3((unsigned)maxblk, (unsigned)B), 1024, 0, stream>>>(reinterpret_cast<const CandLaunch *>(d), SA, SB);
    }
    for (auto &m : mads) mad_kernel<<<m.second, 1024, 0, stream>>>(m.first);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int s = 0; s < B; ++s) {
        const int *h_n = reinterpret_cast<const int *>(ctxs[s]->pinned);
        const float *h_mad = reinterpret_cast<const float *>(ctxs[s]->pinned + 16);
        n_cand2_host[2 * s] = h_n[0];
        n_cand2_host[2 * s + 1] = h_n[1];
        mad2_host[2 * s] = h_mad[1];
        mad2_host[2 * s + 1] = h_mad[3];
    }
    return MODEST_OK;
}

static size_t rsd_partial_bytes(int n, int K) {
    const size_t nb = ((size_t)(n > 0 ? n : 1) + SCORE_PTS - 1) / SCORE_PTS;
    const size_t nbr = ((size_t)(n > 0 ? n : 1) + SCORE_THREADS - 1) / SCORE_THREADS;
    return std::max(arena_sz(nb * (size_t)K * 32), arena_sz(nbr * REFIT_NV * 8));
}
size_t modest_rsd_work_bytes(int n, int max_trials) {   // [counts 256 | thresholds 256 | RsdFit x 2 | partial A | partial B]
    return 512 + arena_sz(2 * sizeof(RsdFit)) + 2 * rsd_partial_bytes(n, std::min(max_trials, RSD_KMAX));
}
static_assert(RSD_KMAX == MODEST_RSD_MAX_TRIALS, "mask_chain.h states the bound");

int modest_rsd_enqueue(modest_ctx *const *ctxs, const float *const *pts, const int *n, const int *stride, int B,
                       const float *specs10, float *const *candA, float *const *candB, const uint32_t *const *mt_key624,
                       const int32_t *mt_pos, int max_trials, double stop_probability, char *const *work_dev,
                       modest_rsd_result *const *res_host, const double **plane1_dev_out, hipStream_t stream) {
    MODEST_REQUIRE(ctxs && pts && n && stride && specs10 && candA && candB && mt_key624 && mt_pos && work_dev && res_host &&
                       plane1_dev_out && B >= 1, "bad chain");
    MODEST_REQUIRE(max_trials >= 1 && max_trials <= RSD_KMAX, "max_trials");
    CandSpec SA{specs10[0], specs10[1], specs10[2], specs10[3], specs10[4]};
    CandSpec SB{specs10[5], specs10[6], specs10[7], specs10[8], specs10[9]};
    std::vector<CandLaunch> sel((size_t)B);
    std::vector<RsdScan> tab((size_t)B);
    std::vector<std::pair<MadArgs, int>> mads;
    MadArgs M{};
    int used = 0, maxblk = 1, maxgx = 1, maxgr = 1;
    for (int s = 0; s < B; ++s) {
        modest_ctx *ctx = ctxs[s];
        MODEST_REQUIRE(ctx && pts[s] && candA[s] && candB[s] && n[s] >= 1 && work_dev[s] && res_host[s] && mt_key624[s] &&
                           mt_pos[s] >= 0 && mt_pos[s] <= 624, "bad scan of the chain");
        const int nblk = (n[s] + 1023) / 1024;
        unsigned long long *state = nullptr;
        int rc = modest_ctx_compact_state(ctx, 2 * (size_t)nblk + 2, stream, &state);
        if (rc) return rc;
        unsigned *zw = nullptr;
        rc = modest_ctx_zero_words(ctx, stream, &zw);
        if (rc) return rc;
        char *w = work_dev[s];
        int *d_n = reinterpret_cast<int *>(w);
        float *d_mad = reinterpret_cast<float *>(w + 256);
        RsdFit *fit = reinterpret_cast<RsdFit *>(w + 512);
        const size_t bp = rsd_partial_bytes(n[s], max_trials);
        char *part = w + 512 + arena_sz(2 * sizeof(RsdFit));
        CandLaunch &L = sel[(size_t)s];
        L.pts = pts[s], L.candA = candA[s], L.candB = candB[s], L.stateA = state, L.stateB = state + 2 + nblk, L.n_out = d_n;
        L.n = n[s], L.stride = stride[s], L.nblk = nblk, L.pad = 0;
        maxblk = std::max(maxblk, nblk);
        if (used + 2 > MAD_SETS) {
            mads.push_back({M, used});
            M = MadArgs{};
            used = 0;
        }
        M.cand[used] = candA[s];
        M.cand[used + 1] = candB[s];
        M.n_dev[used] = d_n;
        M.n_dev[used + 1] = d_n + 1;
        M.out[used] = d_mad;
        M.out[used + 1] = d_mad + 2;
        used += 2;
        RsdScan &R = tab[(size_t)s];
        R.cand[0] = candA[s], R.cand[1] = candB[s];
        R.n_dev = d_n, R.mad_dev = d_mad, R.fit = fit;
        R.partial[0] = reinterpret_cast<double *>(part), R.partial[1] = reinterpret_cast<double *>(part + bp);
        R.ticket = modest_tickets(zw);
        R.res = res_host[s];
        R.stop_probability = stop_probability;
        R.pos0 = mt_pos[s], R.max_trials = max_trials;
        memcpy(R.key0, mt_key624[s], sizeof(R.key0));
        res_host[s]->status = -1;
        plane1_dev_out[s] = fit[0].plane4;
        maxgx = std::max(maxgx, (n[s] + SCORE_PTS - 1) / SCORE_PTS);
        maxgr = std::max(maxgr, (n[s] + SCORE_THREADS - 1) / SCORE_THREADS);
    }
    if (used) mads.push_back({M, used});
    char *d = nullptr, *h = nullptr;
    const size_t bSel = arena_sz(sel.size() * sizeof(CandLaunch)), bTab = tab.size() * sizeof(RsdScan);
    int rc = modest_ctx_chain_tab(ctxs[0], bSel + bTab, &d);
    if (rc) return rc;
    rc = modest_ctx_stage_slot(ctxs[0], bSel + bTab, reinterpret_cast<void **>(&h));
    if (rc) return rc;
    memcpy(h, sel.data(), sel.size() * sizeof(CandLaunch));
    memcpy(h + bSel, tab.data(), bTab);
    MODEST_HIP_CHECK(hipMemcpyAsync(d, h, bSel + bTab, hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctxs[0], stream);
    if (rc) return rc;
    RsdScan *dt = reinterpret_cast<RsdScan *>(d + bSel);
    const unsigned Bu = (unsigned)B;
    const unsigned gy = (unsigned)((std::min(max_trials, RSD_KMAX) + SCORE_KG - 1) / SCORE_KG);
    cdb_candidates2<<<dim3((unsigned)maxblk, Bu), 1024, 0, stream>>>(reinterpret_cast<const CandLaunch *>(d), SA, SB);
    for (auto &m : mads) mad_kernel<<<m.second, 1024, 0, stream>>>(m.first);
    for (int which = 0; which < 2; ++which) {
        rsd_draw<<<Bu, 256, 0, stream>>>(dt, which);
        rsd_score<<<dim3((unsigned)maxgx, gy, Bu), SCORE_THREADS, 0, stream>>>(dt, which);
        rsd_refit<<<dim3((unsigned)maxgr, Bu), SCORE_THREADS, 0, stream>>>(dt, which);
    }
    rsd_final<<<Bu, 256, 0, stream>>>(dt);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_mad_threshold(modest_ctx *ctx, const float *cand, int n_cand, float *mad_host,
                                    void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && mad_host != nullptr, "NULL argument");
    MODEST_REQUIRE(n_cand >= 1, "need at least one candidate");
    MODEST_REQUIRE(cand != nullptr, "cand is NULL");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = modest_ctx_reserve(ctx, 256);
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, 64);
    if (rc) return rc;
    MadArgs A{};
    A.cand[0] = cand;
    A.n[0] = n_cand;
    A.out[0] = reinterpret_cast<float *>(ctx->pinned);   // pinned host memory, read after the sync
    mad_kernel<<<1, 1024, 0, stream>>>(A);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    *mad_host = reinterpret_cast<float *>(ctx->pinned)[1];
    return MODEST_OK;
}

extern "C" int modest_mad_threshold_batch(modest_ctx *ctx, const float *const *cand, const int32_t *n_cand,
                                          int count, float *mad_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && cand != nullptr && n_cand != nullptr && mad_host != nullptr, "NULL argument");
    MODEST_REQUIRE(count >= 1 && count <= 4, "1..4 candidate sets per call");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = modest_ctx_reserve_pinned(ctx, 64);
    if (rc) return rc;
    MadArgs A{};
    for (int i = 0; i < count; ++i) {
        MODEST_REQUIRE(n_cand[i] >= 1 && cand[i] != nullptr, "every set needs at least one candidate");
        A.cand[i] = cand[i];
        A.n[i] = n_cand[i];
        A.out[i] = reinterpret_cast<float *>(ctx->pinned) + 2 * i;
    }
    mad_kernel<<<count, 1024, 0, stream>>>(A);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < count; ++i) mad_host[i] = reinterpret_cast<float *>(ctx->pinned)[2 * i + 1];
    return MODEST_OK;
}

extern "C" int modest_ransac_score_trials(modest_ctx *ctx, const float *cand, int n_cand,
                                          const float *models_host, int K, float thr,
                                          int32_t *n_inliers, double *sse, double *sy, double *syy,
                                          void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_cand >= 1 && K >= 1 && K <= 4096, "bad n_cand / K");
    MODEST_REQUIRE(cand && models_host && n_inliers, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int nb = (n_cand + SCORE_PTS - 1) / SCORE_PTS, nrows = nb * SCORE_WAVES;
    const size_t b_models = arena_sz((size_t)K * 12 + 4), b_part = arena_sz((size_t)nrows * K * 32);
    const size_t b_out = arena_sz((size_t)K * 32);
    int rc = modest_ctx_reserve(ctx, b_models + b_part + b_out);
    if (rc) return rc;
    const size_t hoff = ((size_t)K * 12 + 4 + 63) & ~size_t(63);
    rc = modest_ctx_reserve_pinned(ctx, hoff + (size_t)K * 32);
    if (rc) return rc;
    float *dm = reinterpret_cast<float *>(ctx->scratch);
    double *dp = reinterpret_cast<double *>(ctx->scratch + b_models);
    double *dout = reinterpret_cast<double *>(ctx->scratch + b_models + b_part);
    float *hm = reinterpret_cast<float *>(ctx->pinned);
    double *hout = reinterpret_cast<double *>(ctx->pinned + hoff);
    for (int i = 0; i < K * 3; ++i) hm[i] = models_host[i];
    hm[K * 3] = thr;   // the kernel reads the threshold from device memory
    MODEST_HIP_CHECK(hipMemcpyAsync(dm, hm, (size_t)K * 12 + 4, hipMemcpyHostToDevice, stream));
    unsigned *zw = nullptr;
    rc = modest_ctx_zero_words(ctx, stream, &zw);
    if (rc) return rc;
    score_kernel<false><<<dim3(nb, (K + SCORE_KG - 1) / SCORE_KG), SCORE_THREADS, 0, stream>>>(
        cand, n_cand, dm, K, nullptr, thr, dp, TripArg{}, nullptr, modest_tickets(zw), dout, nullptr, nullptr);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipMemcpyAsync(hout, dout, (size_t)K * 32, hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int k = 0; k < K; ++k) {
        n_inliers[k] = (int32_t)hout[4 * k];
        if (sse) sse[k] = hout[4 * k + 1];
        if (sy) sy[k] = hout[4 * k + 2];
        if (syy) syy[k] = hout[4 * k + 3];
    }
    return MODEST_OK;
}

// phase & 1: enqueue (no synchronise); phase & 2: read the results (after the caller's synchronise, or
// after this function's own when both bits are set).  The results of a batch live in the context's
// pinned block from byte 576 on (K <= 64), those of a refit in its first 128 bytes: one of each may be
// in flight on a stream at the same time (scan_driver.hip).
// Launch capture (scan_driver.hip, chains of scans): while a capture is open on the calling thread, the enqueue
// halves of modest_ransac_trials_phase / modest_ransac_refit_phase record their kernel arguments instead of
// launching; modest_ransac_capture_launch then runs ALL recorded refits as one launch and all recorded trial
// batches as one launch (the fit is a grid dimension).  Everything else of the two functions -- buffers, pinned
// result areas, read-back halves -- is unchanged.
struct modest_ransac_capture {
    std::vector<ScoreLaunch> score;
    std::vector<RefitLaunch> refit;
};
static thread_local modest_ransac_capture *g_capture = nullptr;

modest_ransac_capture *modest_ransac_capture_begin() {
    g_capture = new modest_ransac_capture;
    return g_capture;
}

int modest_ransac_capture_launch(modest_ctx *ctx0, modest_ransac_capture *cap, hipStream_t stream) {
    g_capture = nullptr;
    std::unique_ptr<modest_ransac_capture> own(cap);
    MODEST_REQUIRE(ctx0 != nullptr && cap != nullptr, "NULL argument");
    const size_t bR = arena_sz(cap->refit.size() * sizeof(RefitLaunch)), bS = cap->score.size() * sizeof(ScoreLaunch);
    if (bR + bS == 0) return MODEST_OK;
    char *d = nullptr, *h = nullptr;
    int rc = modest_ctx_chain_tab(ctx0, bR + bS, &d);
    if (rc) return rc;
    rc = modest_ctx_stage_slot(ctx0, bR + bS, reinterpret_cast<void **>(&h));
    if (rc) return rc;
    if (!cap->refit.empty()) memcpy(h, cap->refit.data(), cap->refit.size() * sizeof(RefitLaunch));
    if (!cap->score.empty()) memcpy(h + bR, cap->score.data(), bS);
    MODEST_HIP_CHECK(hipMemcpyAsync(d, h, bR + bS, hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx0, stream);
    if (rc) return rc;
    if (!cap->refit.empty()) {
        int gx = 1;
        for (const RefitLaunch &r : cap->refit) gx = std::max(gx, r.gx);
        scb_refit<<<dim3((unsigned)gx, (unsigned)cap->refit.size()), SCORE_THREADS, 0, stream>>>(
            reinterpret_cast<const RefitLaunch *>(d));
    }
    if (!cap->score.empty()) {
        int gx = 1, gy = 1;
        for (const ScoreLaunch &q : cap->score) {
            gx = std::max(gx, q.gx);
            gy = std::max(gy, q.gy);
        }
        scb_score<<<dim3((unsigned)gx, (unsigned)gy, (unsigned)cap->score.size()), SCORE_THREADS, 0, stream>>>(
            reinterpret_cast<const ScoreLaunch *>(d + bR));
    }
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

int modest_ransac_trials_phase(modest_ctx *ctx, const float *cand, int n_cand, const int32_t *trip_host, int K,
                               float *thr_inout, float *models_out, int32_t *n_inliers, double *sse, double *sy,
                               double *syy, void *stream_, int phase) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_cand >= 3 && K >= 1 && K <= 4096, "bad n_cand / K");
    MODEST_REQUIRE(cand && trip_host && thr_inout && models_out && n_inliers, "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int nb = (n_cand + SCORE_PTS - 1) / SCORE_PTS, nrows = nb * SCORE_WAVES;
    // device: [models K*3 f32 | thr pair 2 f32 | partial]; pinned host: [trip | results].
    // The kernels read the triplets from and write the results (sums, models, threshold) to the pinned
    // block directly: one stream sync, no copy kernels in either direction.
    const size_t n_out = (size_t)K * 32, n_models = (size_t)K * 12;
    const size_t b_models = arena_sz(n_models), b_thr = arena_sz(8);
    const size_t b_part = arena_sz((size_t)nrows * K * 32);
    int rc = modest_ctx_reserve(ctx, b_models + b_thr + b_part);
    if (rc) return rc;
    const size_t h_trip = std::max<size_t>(576, ((size_t)K * 12 + 63) & ~size_t(63));   // clear of a refit's 128 bytes
    rc = modest_ctx_reserve_pinned(ctx, h_trip + n_out + n_models + 8);
    if (rc) return rc;
    char *d = ctx->scratch;
    float *d_models = reinterpret_cast<float *>(d);
    float *d_thr = reinterpret_cast<float *>(d + b_models);
    double *d_part = reinterpret_cast<double *>(d + b_models + b_thr);
    int *h_tripp = reinterpret_cast<int *>(ctx->pinned);
    char *h_res = ctx->pinned + h_trip;
    double *h_outp = reinterpret_cast<double *>(h_res);
    float *h_models = reinterpret_cast<float *>(h_res + n_out);
    float *h_thr = reinterpret_cast<float *>(h_res + n_out + n_models);
    const bool thr_known = !(*thr_inout < 0.f);
    const float thr_val = *thr_inout;
    if (phase & 1) {
    if (K > TRIP_MAX)
        for (int i = 0; i < 3 * K; ++i) h_tripp[i] = trip_host[i];
    if (!thr_known) {   // residual threshold = MAD of the candidates, computed on the device
        MadArgs A{};
        A.cand[0] = cand;
        A.n[0] = n_cand;
        A.out[0] = d_thr;
        mad_kernel<<<1, 1024, 0, stream>>>(A);
    }
    const dim3 sgrid(nb, (K + SCORE_KG - 1) / SCORE_KG);
    unsigned *zw = nullptr;
    rc = modest_ctx_zero_words(ctx, stream, &zw);
    if (rc) return rc;
    const float *thr_src = thr_known ? nullptr : d_thr;
    float *thr_dst = thr_known ? nullptr : h_thr;
    if (K <= TRIP_MAX && thr_known && g_capture) {   // a chain of scans: launched together with the other fits' batches
        ScoreLaunch L;
        L.cand = cand, L.partial = d_part, L.models_host = h_models, L.ticket = modest_tickets(zw), L.out = h_outp;
        L.n = n_cand, L.K = K, L.gx = (int)sgrid.x, L.gy = (int)sgrid.y, L.thr_val = thr_val;
        for (int i = 0; i < 3 * K; ++i) L.trip.t[i] = trip_host[i];
        g_capture->score.push_back(L);
    } else if (K <= TRIP_MAX) {
        TripArg ta;
        for (int i = 0; i < 3 * K; ++i) ta.t[i] = trip_host[i];
        score_kernel<true><<<sgrid, SCORE_THREADS, 0, stream>>>(cand, n_cand, nullptr, K,
                                                               thr_known ? nullptr : d_thr + 1, thr_val, d_part, ta,
                                                               h_models, modest_tickets(zw), h_outp, thr_src, thr_dst);
    } else {
        fit_kernel<<<(K + 63) / 64, 64, 0, stream>>>(cand, n_cand, h_tripp, K, d_models, h_models);
        score_kernel<false><<<sgrid, SCORE_THREADS, 0, stream>>>(cand, n_cand, d_models, K,
                                                                thr_known ? nullptr : d_thr + 1, thr_val, d_part,
                                                                TripArg{}, nullptr, modest_tickets(zw), h_outp, thr_src,
                                                                thr_dst);
    }
    MODEST_HIP_CHECK(hipGetLastError());
    }
    if (!(phase & 2)) return MODEST_OK;
    if (phase & 1) MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    *thr_inout = thr_known ? thr_val : h_thr[1];
    for (int k = 0; k < K; ++k) {
        n_inliers[k] = (int32_t)h_outp[4 * k];
        if (sse) sse[k] = h_outp[4 * k + 1];
        if (sy) sy[k] = h_outp[4 * k + 2];
        if (syy) syy[k] = h_outp[4 * k + 3];
        models_out[3 * k] = h_models[3 * k];
        models_out[3 * k + 1] = h_models[3 * k + 1];
        models_out[3 * k + 2] = h_models[3 * k + 2];
    }
    return MODEST_OK;
}

extern "C" int modest_ransac_trials(modest_ctx *ctx, const float *cand, int n_cand, const int32_t *trip_host,
                                    int K, float *thr_inout, float *models_out, int32_t *n_inliers,
                                    double *sse, double *sy, double *syy, void *stream_) {
    return modest_ransac_trials_phase(ctx, cand, n_cand, trip_host, K, thr_inout, models_out, n_inliers, sse, sy, syy,
                                      stream_, 3);
}

// Device scratch a trial batch of K trials or a refit over at most n_cand candidates carves from the context arena.  A
// driver that RECORDS launches (scan_driver.hip: a refit of fit A and the first batch of fit B go out together) reserves
// this bound before the first recording: a later modest_ctx_reserve that had to grow the arena would free the block a
// recorded launch still points into.
size_t modest_ransac_scratch_bound(int n_cand, int K) {
    const size_t nb = ((size_t)(n_cand > 0 ? n_cand : 1) + SCORE_PTS - 1) / SCORE_PTS, nrows = nb * SCORE_WAVES;
    const size_t trials = arena_sz((size_t)K * 12) + arena_sz(8) + arena_sz(nrows * (size_t)K * 32);
    const size_t nbr = ((size_t)(n_cand > 0 ? n_cand : 1) + SCORE_THREADS - 1) / SCORE_THREADS;
    const size_t refit = arena_sz(nbr * REFIT_NV * 8);
    return trials > refit ? trials : refit;
}

int modest_ransac_refit_phase(modest_ctx *ctx, const float *cand, int n_cand, const float *model_host, float thr,
                              double *out_model, int32_t *n_inliers, void *stream_, int phase) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_cand >= 1, "bad n_cand");
    MODEST_REQUIRE(cand && model_host && (out_model || !(phase & 2)), "NULL buffer");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int nb = (n_cand + SCORE_THREADS - 1) / SCORE_THREADS;
    int rc = modest_ctx_reserve(ctx, arena_sz((size_t)nb * REFIT_NV * 8));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, 16 * 8);
    if (rc) return rc;
    double *dp = reinterpret_cast<double *>(ctx->scratch);
    const float c0 = model_host[0], c1 = model_host[1], b = model_host[2];
    double *h = reinterpret_cast<double *>(ctx->pinned);
    if (phase & 1) {
        // (not in the read-back phase: a context whose cell counters are live -- between the two phases of
        // modest_mask_cluster_phase -- is marked dirty, and this call would clear them)
        unsigned *zw = nullptr;
        rc = modest_ctx_zero_words(ctx, stream, &zw);
        if (rc) return rc;
        if (g_capture) {   // a chain of scans: launched together with the other scans' refits
            RefitLaunch L;
            L.cand = cand, L.partial = dp, L.ticket = modest_tickets(zw), L.out_host = h;
            L.n = n_cand, L.gx = nb, L.c0 = c0, L.c1 = c1, L.b = b, L.thr = thr;
            g_capture->refit.push_back(L);
        } else {
            refit_kernel<<<nb, SCORE_THREADS, 0, stream>>>(cand, n_cand, c0, c1, b, thr, dp, modest_tickets(zw), h);   // totals straight into pinned host memory
        }
        MODEST_HIP_CHECK(hipGetLastError());
    }
    if (!(phase & 2)) return MODEST_OK;
    if (phase & 1) MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    const double cnt = h[0];
    const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
    const double mx = h[1] * inv, my = h[2] * inv, mzs = h[3] * inv;   // mzs: mean of z - b
    const double sxx = h[4] - h[1] * mx, sxy = h[5] - h[1] * my, syy = h[6] - h[2] * my;
    const double sxz = h[7] - h[1] * mzs, syz = h[8] - h[2] * mzs;
    const double mz = mzs + (double)b;
    if (n_inliers) *n_inliers = (int32_t)cnt;
    const double det = sxx * syy - sxy * sxy;
    if (!(cnt >= 3.0) || !(fabs(det) > 1e-12 * fmax(sxx * syy, 1e-300))) {
        modest_set_error("modest_ransac_refit: degenerate inlier set (n=%d)", (int)cnt);
        return MODEST_ERR_ARG;
    }
    const double a0 = (sxz * syy - syz * sxy) / det;
    const double a1 = (syz * sxx - sxz * sxy) / det;
    out_model[0] = a0;
    out_model[1] = a1;
    out_model[2] = mz - a0 * mx - a1 * my;
    return MODEST_OK;
}

extern "C" int modest_ransac_refit(modest_ctx *ctx, const float *cand, int n_cand,
                                   const float *model_host, float thr, double *out_model,
                                   int32_t *n_inliers, void *stream_) {
    return modest_ransac_refit_phase(ctx, cand, n_cand, model_host, thr, out_model, n_inliers, stream_, 3);
}

// ---- the RANSAC driver: sklearn's trial loop on the host side of the library --------------------
// (generator, accept rule and the fit's state machine: ransac_host.h)
extern "C" int modest_mt19937_triplets(uint32_t *key624, int32_t *pos, uint32_t n_population, int n_trials,
                                       int32_t *triplets_out) {
    MODEST_REQUIRE(key624 && pos && triplets_out, "NULL argument");
    MODEST_REQUIRE(n_population > 300 && n_trials >= 0 && *pos >= 0 && *pos <= 624, "bad arguments");
    Mt19937 g;
    memcpy(g.key, key624, sizeof(g.key));
    g.pos = *pos;
    for (int k = 0; k < n_trials; ++k) g.triplet(n_population, triplets_out + 3 * k);
    memcpy(key624, g.key, sizeof(g.key));
    *pos = g.pos;
    return MODEST_OK;
}

extern "C" int modest_ransac_plane(modest_ctx *ctx, const float *cand, int n_cand, float thr, uint32_t *key624,
                                   int32_t *pos, int max_trials, double stop_probability, int batch,
                                   double *model64_out, float *best_model_out, int32_t *triplets_out,
                                   int32_t *n_trials_out, int32_t *n_inliers_out, int32_t *status_out, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && cand && key624 && pos && model64_out && best_model_out && n_trials_out && status_out,
                   "NULL argument");
    MODEST_REQUIRE(n_cand > 300, "fewer than 301 candidates: sklearn draws with another method there (host path)");
    MODEST_REQUIRE(max_trials >= 1 && max_trials <= 4096 && batch >= 1 && batch <= TRIP_MAX, "bad max_trials / batch");
    MODEST_REQUIRE(*pos >= 0 && *pos <= 624 && thr >= 0.f, "bad generator position / threshold");
    Mt19937 g;
    memcpy(g.key, key624, sizeof(g.key));
    g.pos = *pos;
    hipStream_t stream = as_stream(stream_);
    RansacFit fit;
    fit.init(ctx, cand, n_cand, thr, &g, max_trials, stop_probability, batch, stream_);
    fit.triplets_out = triplets_out;
    *status_out = 0;
    while (!fit.done()) {
        int rc = fit.enqueue_batch();
        if (rc) return rc;
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        rc = fit.finish_batch();
        if (rc) return rc;
    }
    memcpy(key624, g.key, sizeof(g.key));
    *pos = g.pos;
    *n_trials_out = fit.n_trials;
    if (!fit.have) {
        *status_out = 1;   // no consensus set: sklearn raises ValueError
        return MODEST_OK;
    }
    for (int q = 0; q < 3; ++q) best_model_out[q] = fit.best[q];
    int rc = fit.enqueue_refit();
    if (rc) return rc;
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    int32_t nfin = 0;
    bool degenerate = false;
    rc = fit.finish_refit(model64_out, &nfin, &degenerate);
    if (n_inliers_out) *n_inliers_out = nfin;
    if (degenerate) *status_out = 2;   // the caller takes the minimum-norm fit on the host
    return rc;
}

extern "C" int modest_plane_range_mask(modest_ctx *ctx, const float *pts, int n, int stride,
                                       const double *plane4, double offset,
                                       const double *only_range4, const double *limit_range4,
                                       uint8_t *mask, float *kept, int32_t *kept_idx,
                                       int32_t *n_kept, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n >= 0 && (stride == 3 || stride == 4), "bad n/stride");
    MODEST_REQUIRE(plane4 && limit_range4 && n_kept, "NULL argument");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {   // otherwise the last block of the kernel writes the total
        MODEST_HIP_CHECK(hipMemsetAsync(n_kept, 0, sizeof(int32_t), stream));
        return MODEST_OK;
    }
    MODEST_REQUIRE(pts && mask, "NULL buffer");
    MaskParams P;
    mask_params_fill(P, plane4, offset, only_range4, limit_range4);
    const int nblk = (n + 1023) / 1024;
    unsigned long long *state = nullptr;
    int rc = modest_ctx_compact_state(ctx, (size_t)nblk, stream, &state);
    if (rc) return rc;
    mask_kernel<<<nblk, 1024, 0, stream>>>(pts, n, stride, P, mask, kept, kept_idx, state,
                                                       n_kept);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_plane(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(fit_kernel));
}
