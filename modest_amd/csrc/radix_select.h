// Exact k-th smallest float32 of an indexed subset, one workgroup per subset:
// radix descent over the order-preserving key, 11 + 11 + 10 bits, LDS histograms.
// Used for the order statistics numpy.percentile interpolates between
// (clustering_utils.py:107-117 is_valid_cluster; combine_labels.py:41-60 filter_by_ppscore).
#pragma once
#include "common.h"

namespace {

constexpr int CS_THREADS = 256;

__device__ __forceinline__ unsigned cs_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float cs_unkey(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// k-th smallest (0-based) of the n values whose order-preserving keys key_at(i) yields
// (workgroup of NT threads).  Three 11/11/10-bit histogram rounds of three barriers each: the scan
// clears the bins it reads, so the next round (and the next call) finds the histogram zeroed.
template <int NT, class KeyAt>
__device__ __forceinline__ float cs_select_keys(KeyAt key_at, int n, unsigned k, unsigned *hist /* 2048 */,
                                                unsigned *wsum /* NT/64 */, unsigned *sel /* 2 */) {
    constexpr int BPT = 2048 / NT;   // bins per thread in the scan
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned prefix = 0, mask = 0;
    const int shifts[3] = {21, 10, 0}, bitsv[3] = {11, 11, 10};
    for (unsigned b = tid; b < 2048u; b += NT) hist[b] = 0;
    __syncthreads();
    for (int ps = 0; ps < 3; ++ps) {
        const int shift = shifts[ps];
        const unsigned nb = 1u << bitsv[ps];
        for (int i = tid; i < n; i += NT) {
            const unsigned key = key_at(i);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1u)], 1u);
        }
        __syncthreads();
        unsigned v[BPT], s = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            v[j] = hist[BPT * tid + j];
            hist[BPT * tid + j] = 0;
            s += v[j];
        }
        unsigned inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int q = 0; q < w; ++q) base += wsum[q];
        unsigned run = base + inc - s;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            if (k >= run && k < run + v[j]) {
                sel[0] = BPT * tid + j;
                sel[1] = k - run;
            }
            run += v[j];
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        mask |= (nb - 1u) << shift;
        k = sel[1];
        // sel is rewritten only after the next round's two barriers
    }
    __syncthreads();   // callers reuse sel / hist right away
    return cs_unkey(prefix);
}

// k-th smallest (0-based) pp among the members
template <int NT>
__device__ __forceinline__ float cs_select(const float *__restrict__ pp, const int *__restrict__ mem, int n,
                                           unsigned k, unsigned *hist, unsigned *wsum, unsigned *sel) {
    return cs_select_keys<NT>([=](int i) { return cs_key(pp[mem[i]]); }, n, k, hist, wsum, sel);
}

// numpy.percentile(x, q) ('linear') on float32 data of size n works in float32: virtual index
// (n-1)*q, neighbours floor / floor+1, both clamped to the last element.  Returns the two
// neighbour ranks and the interpolation weight.
__device__ __forceinline__ void cs_percentile_ranks(int n, float qf, int *prev, int *next, float *gamma) {
    const float vi = (float)(n - 1) * qf;
    const float fl = floorf(vi);
    int p = (int)fl, nx = p + 1;
    if (vi >= (float)(n - 1)) p = nx = n - 1;
    if (vi < 0.f) p = nx = 0;
    *prev = p;
    *next = min(nx, n - 1);
    *gamma = vi - fl;
}

}  // namespace
