// Host side of the RANSAC fits: numpy's legacy generator as sklearn consumes it, sklearn's accept rule,
// and one fit as a small state machine whose device round trips can be interleaved with other work
// (plane.hip: modest_ransac_plane; scan_driver.hip: two fits + the mask kernel of a scan).
#pragma once
#include "common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace modest {

// numpy's legacy generator (RandomState = MT19937) as sklearn's sample_without_replacement consumes
// it: for n_samples > 300 ("tracking selection") one trial = RandomState.randint(n) until three
// distinct indices are found, and randint(n) is the masked rejection `next_uint32 & mask` until
// the value is < n (numpy/random/_bounded_integers: use_masked, 32-bit range).
struct Mt19937 {
    uint32_t key[624];
    int pos;
    void refill() {
        const uint32_t U = 0x80000000u, L = 0x7fffffffu, A = 0x9908b0dfu;
        int kk = 0;
        for (; kk < 624 - 397; ++kk) {
            const uint32_t y = (key[kk] & U) | (key[kk + 1] & L);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; kk < 623; ++kk) {
            const uint32_t y = (key[kk] & U) | (key[kk + 1] & L);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        const uint32_t y = (key[623] & U) | (key[0] & L);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos >= 624) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t randint(uint32_t n) {   // RandomState.randint(n), 1 <= n <= 2^32 - 1
        const uint32_t rng = n - 1;
        if (rng == 0) return 0;
        uint32_t mask = rng;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        while ((v = next32() & mask) > rng) {
        }
        return v;
    }
    void triplet(uint32_t n, int32_t *t) {   // sample_without_replacement(n, 3), tracking selection
        int have = 0;
        while (have < 3) {
            const int32_t j = (int32_t)randint(n);
            bool dup = false;
            for (int q = 0; q < have; ++q) dup = dup || t[q] == j;
            if (!dup) t[have++] = j;
        }
    }
};

__host__ __device__ inline double ransac_dynamic_max_trials(int n_inliers, int n_samples, double probability) {   // sklearn _dynamic_max_trials, min_samples = 3
    const double eps = 2.220446049250313e-16;
    const double ratio = (double)n_inliers / (double)n_samples;
    const double nom = fmax(eps, 1.0 - probability);
    const double denom = fmax(eps, 1.0 - pow(ratio, 3.0));
    if (nom == 1.0) return 0.0;
    if (denom == 1.0) return INFINITY;
    return fabs(ceil(log(nom) / log(denom)));
}

__host__ __device__ inline double ransac_r2_from_sums(int n, double sse, double sy, double syy) {   // r2_score over the inliers
    if (n < 2) return NAN;
    const double den = syy - sy * sy / n;
    if (den <= 0.0) return sse == 0.0 ? 1.0 : 0.0;
    return 1.0 - sse / den;
}

// One RANSAC fit (sklearn RANSACRegressor defaults, residual threshold given).  The generator is shared
// with the caller and advanced by the EXECUTED trials only.
struct RansacFit {
    modest_ctx *ctx = nullptr;
    const float *cand = nullptr;
    int n_cand = 0, batch = 48;
    float thr = 0.f;
    double stop_probability = 0.99;
    Mt19937 *g = nullptr;
    void *stream = nullptr;
    int32_t *triplets_out = nullptr;   // optional (max_trials, 3)
    int n_best = 1, n_trials = 0, nb = 0;
    double score_best = -INFINITY, limit = 100.0;
    bool have = false;
    bool pending = false;   // a batch is in flight (drivers that keep several fits going: scan_driver.hip)
    float best[3] = {0, 0, 0};
    Mt19937 before;
    std::vector<int32_t> trip, n_in;
    std::vector<float> models;
    std::vector<double> sse, sy, syy;

    void init(modest_ctx *c, const float *cd, int n, float t, Mt19937 *gen, int max_trials, double p, int b, void *s) {
        ctx = c, cand = cd, n_cand = n, thr = t, g = gen, limit = (double)max_trials, stop_probability = p, batch = b, stream = s;
        trip.resize((size_t)3 * b), n_in.resize(b), models.resize((size_t)3 * b), sse.resize(b), sy.resize(b), syy.resize(b);
    }
    bool done() const { return !((double)n_trials < limit); }
    int enqueue_batch() {   // draw the triplets of the next batch and launch their scoring
        nb = (int)fmin((double)batch, limit - (double)n_trials);
        before = *g;
        for (int k = 0; k < nb; ++k) g->triplet((uint32_t)n_cand, trip.data() + 3 * k);
        float t = thr;
        return modest_ransac_trials_phase(ctx, cand, n_cand, trip.data(), nb, &t, models.data(), n_in.data(), sse.data(),
                                          sy.data(), syy.data(), stream, 1);
    }
    int finish_batch() {   // after a synchronise: results, sequential accept rule, dynamic trial bound
        float t = thr;
        int rc = modest_ransac_trials_phase(ctx, cand, n_cand, trip.data(), nb, &t, models.data(), n_in.data(),
                                            sse.data(), sy.data(), syy.data(), stream, 2);
        if (rc) return rc;
        int used = 0;
        for (int k = 0; k < nb; ++k) {
            if (done()) break;
            ++n_trials;
            ++used;
            const int nk = n_in[k];
            if (nk < n_best) continue;
            const double score = ransac_r2_from_sums(nk, sse[k], sy[k], syy[k]);
            if (nk == n_best && score < score_best) continue;
            n_best = nk;
            score_best = score;
            have = true;
            for (int q = 0; q < 3; ++q) best[q] = models[3 * k + q];
            limit = fmin(limit, ransac_dynamic_max_trials(n_best, n_cand, stop_probability));
        }
        if (triplets_out)
            for (int q = 0; q < 3 * used; ++q) triplets_out[3 * (n_trials - used) + q] = trip[q];
        if (used < nb) {   // the caller's stream advances by exactly the executed trials
            *g = before;
            int32_t scratch[3];
            for (int k = 0; k < used; ++k) g->triplet((uint32_t)n_cand, scratch);
        }
        return MODEST_OK;
    }
    int enqueue_refit() { return modest_ransac_refit_phase(ctx, cand, n_cand, best, thr, nullptr, nullptr, stream, 1); }
    // after a synchronise; *degenerate = the consensus set has no unique plane (host statement takes over)
    int finish_refit(double *model64, int32_t *n_inliers, bool *degenerate) {
        const int rc = modest_ransac_refit_phase(ctx, cand, n_cand, best, thr, model64, n_inliers, stream, 2);
        *degenerate = rc == MODEST_ERR_ARG;
        return *degenerate ? MODEST_OK : rc;
    }
};

}  // namespace modest
