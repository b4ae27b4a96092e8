// Per-scan driver of the mask stage (generate_mask.py:52-88 + clustering_utils.py:119-135): both
// ground fits, the plane / range mask, the affinity graph + DBSCAN, the cluster statistics, the
// is_valid_cluster rules and the relabelling behind ONE call, so that no interpreter runs between
// the device round trips of these steps.  Host code only: every device step is one of the library's
// own entry points (their temporaries come from the context arena at offset 0; what has to live
// across them -- candidate sets, the scan-sized labels -- sits in the context's hold buffers).
// modest_mask_stage_batch (end of the file) drives the same stage for a CHAIN of scans: per-scan contexts, the fits of
// all scans in lockstep, one launch per kernel for the whole chain (plane.hip launch capture, cluster.hip mcb_*,
// cluster_stats.hip csb_stats).
#include "common.h"
#include "ransac_host.h"
#include "mask_chain.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

using namespace modest;

namespace {

// plane_from_linear_model (utils/pointcloud_utils.py:53-62): w = (c0, c1, -1), h = b, both divided by
// np.linalg.norm(w) = sqrt(w.dot(w)) -- the squares of float32-valued coefficients are exact in
// float64, so the dot product rounds only in its two additions, left to right -- then negated.
void plane_from_model(const double *model64, double *plane4) {
    const double c0 = (double)(float)model64[0], c1 = (double)(float)model64[1];   // LinearRegression stores float32
    const float b32 = (float)model64[2];
    const double norm = sqrt((c0 * c0 + c1 * c1) + 1.0);
    plane4[0] = -(c0 / norm);
    plane4[1] = -(c1 / norm);
    plane4[2] = -(-1.0 / norm);
    plane4[3] = -((double)b32 / norm);
}

// is_valid_cluster (clustering_utils.py:94-117) + relabelling (:131-135) as one table look-up:
// table[l + 1] = rank of label l among the surviving values; -1 survives iff it occurs or a cluster is dropped
void finish_labels(const modest_mask_params *P, int n, int n_clusters, const std::vector<double> &st, const int32_t *labels_h,
                   int64_t *labels_out, int32_t *info_out, int32_t *members_out = nullptr, int32_t *n_members_out = nullptr) {
    std::vector<int64_t> table((size_t)n_clusters + 1, 0);
    bool has_neg = false;
    for (int i = 0; i < n && !has_neg; ++i) has_neg = labels_h[i] < 0;
    std::vector<char> valid(n_clusters > 0 ? n_clusters : 1, 0);
    int n_valid = 0;
    for (int c = 0; c < n_clusters; ++c) {
        const double cnt = st[6 * c], dmin = st[6 * c + 1], dmax = st[6 * c + 2];
        // numpy.percentile(float32 data, method='linear'): _lerp in float32, the upper form from t = 0.5 on
        const float a = (float)st[6 * c + 3], b = (float)st[6 * c + 4], t = (float)st[6 * c + 5];
        const float diff = b - a;
        const float lo = a + diff * t, hi = b - diff * (1.0f - t);
        const float pct = t >= 0.5f ? hi : lo;
        const bool ok = cnt >= (double)P->min_points && !(dmin > P->max_min_height) && !(dmax < P->min_max_height) &&
                        !(pct > P->min_percentile_pp_score) && cnt >= 1.0;
        valid[c] = ok ? 1 : 0;
        n_valid += ok;
        if (!ok && cnt >= 1.0) has_neg = true;
    }
    int64_t next = has_neg ? 1 : 0;
    for (int c = 0; c < n_clusters; ++c)
        if (valid[c]) table[(size_t)c + 1] = next++;
    if (members_out && n_clusters > 0) {   // the same pass also lists the points of the surviving clusters (for the box tail)
        int32_t m = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t l = table[(size_t)(labels_h[i] + 1)];
            labels_out[i] = l;
            members_out[m] = i;
            m += l > 0;
        }
        *n_members_out = m;
    } else {
        for (int i = 0; i < n; ++i) labels_out[i] = table[(size_t)(labels_h[i] + 1)];
        if (n_members_out) *n_members_out = members_out ? 0 : -1;
    }
    info_out[2] = n_valid > 0 ? (int32_t)(next - 1) : 0;   // largest final label = number of box candidates
    if (n_clusters == 0) {   // compact_labels of the raw labels: all -1 -> all 0
        for (int i = 0; i < n; ++i) labels_out[i] = 0;
        info_out[2] = 0;
    }
}

}  // namespace

extern "C" int modest_mask_stage(modest_ctx *ctx, const float *pts, int n, int stride, const float *pp,
                                 const modest_mask_params *P, uint32_t *mt_key624, int32_t *mt_pos,
                                 double *plane1_out, double *plane2_out, int64_t *labels_out, int32_t *info_out,
                                 void *stream_) {
    MODEST_REQUIRE(ctx && P && mt_key624 && mt_pos && plane1_out && plane2_out && labels_out && info_out, "NULL argument");
    MODEST_REQUIRE(n >= 1 && (stride == 3 || stride == 4) && pts && pp, "bad scan");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    for (int q = 0; q < 8; ++q) info_out[q] = 0;
    // hold: candidate sets A, B (n x 3 f32 each) | labels (n int32); pinned: labels mirror
    const size_t b_cand = arena_sz((size_t)n * 12), b_lab = arena_sz((size_t)n * 4);
    int rc = modest_ctx_reserve_hold(ctx, 2 * b_cand + b_lab, b_lab);
    if (rc) return rc;
    float *candA = reinterpret_cast<float *>(ctx->hold), *candB = reinterpret_cast<float *>(ctx->hold + b_cand);
    int32_t *labels_dev = reinterpret_cast<int32_t *>(ctx->hold + 2 * b_cand);
    int32_t *labels_h = reinterpret_cast<int32_t *>(ctx->hold_pinned);

    // 1. candidates + MAD thresholds of both fits
    const float specs[10] = {P->max_hs1, P->range1[0], P->range1[1], P->range1[2], P->range1[3],
                             P->max_hs2, P->range2[0], P->range2[1], P->range2[2], P->range2[3]};
    int32_t n_cand[2];
    float mad[2];
    rc = modest_plane_prepare(ctx, pts, n, stride, specs, candA, candB, n_cand, mad, stream_);
    if (rc) return rc;
    info_out[4] = n_cand[0];
    info_out[5] = n_cand[1];
    if (n_cand[0] <= 300 || n_cand[1] <= 300) {   // sklearn samples small populations with other methods: host path
        info_out[3] = MODEST_STAGE_SMALL_SET;
        return MODEST_OK;
    }
    // 2. the two RANSAC fits, in the order the reference draws from its generator, and the mask kernel.
    //    Round trips: [batch A]* , [refit A + first batch B] , [batch B]* , [refit B + mask / compaction /
    //    cell count with plane A] -- a refit rides along with the next independent launch (their result
    //    areas in the pinned block are disjoint, common.h).  The pinned block is sized here once: growing
    //    it between an enqueue and its read-back would free results.
    rc = modest_ctx_reserve_pinned(ctx, 16384);
    if (rc) return rc;
    rc = modest_ctx_reserve(ctx, modest_ransac_scratch_bound(n_cand[0] > n_cand[1] ? n_cand[0] : n_cand[1], P->batch > 64 ? P->batch : 64));   // (ditto: the arena)
    if (rc) return rc;
    Mt19937 g;
    memcpy(g.key, mt_key624, sizeof(g.key));
    g.pos = *mt_pos;
    RansacFit A, B;
    A.init(ctx, candA, n_cand[0], mad[0], &g, P->max_trials, P->stop_probability, P->batch, stream_);
    B.init(ctx, candB, n_cand[1], mad[1], &g, P->max_trials, P->stop_probability, P->batch, stream_);
    while (!A.done()) {
        if ((rc = A.enqueue_batch())) return rc;
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        if ((rc = A.finish_batch())) return rc;
    }
    info_out[6] = A.n_trials;
    if (!A.have) {
        info_out[3] = MODEST_STAGE_NO_CONSENSUS;
        return MODEST_OK;
    }
    if ((rc = A.enqueue_refit())) return rc;
    if ((rc = B.enqueue_batch())) return rc;
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    double model64[3];
    int32_t n_in = 0;
    bool degenerate = false;
    if ((rc = A.finish_refit(model64, &n_in, &degenerate))) return rc;
    if (degenerate) {   // the host statement handles it (never seen on LiDAR scans)
        info_out[3] = MODEST_STAGE_DEGENERATE;
        return MODEST_OK;
    }
    plane_from_model(model64, plane1_out);
    if ((rc = B.finish_batch())) return rc;
    while (!B.done()) {
        if ((rc = B.enqueue_batch())) return rc;
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        if ((rc = B.finish_batch())) return rc;
    }
    info_out[7] = B.n_trials;
    if (!B.have) {
        info_out[3] = MODEST_STAGE_NO_CONSENSUS;
        return MODEST_OK;
    }
    // 3. mask + graph + DBSCAN + labels[ptc_mask] = ...
    int32_t n_kept = 0, n_clusters = 0;
    const double *only = P->use_only_range ? P->only_range : nullptr;
    if ((rc = B.enqueue_refit())) return rc;
    rc = modest_mask_cluster_phase(ctx, pts, n, stride, pp, plane1_out, P->offset, only, P->limit_range, P->neighbor_type,
                                   P->affinity_type, P->k_neighbors, P->radius, P->eps, P->min_samples, labels_dev, &n_kept,
                                   &n_clusters, stream_, 1);
    if (rc) return rc;
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    if ((rc = B.finish_refit(model64, &n_in, &degenerate))) return rc;
    if (degenerate) {
        info_out[3] = MODEST_STAGE_DEGENERATE;
        ctx->zwords_dirty = 1;   // the cell counters hold this scan's counts
        ctx->zwords_live = 0;
        return MODEST_OK;
    }
    plane_from_model(model64, plane2_out);
    memcpy(mt_key624, g.key, sizeof(g.key));
    *mt_pos = g.pos;
    rc = modest_mask_cluster_phase(ctx, pts, n, stride, pp, plane1_out, P->offset, only, P->limit_range, P->neighbor_type,
                                   P->affinity_type, P->k_neighbors, P->radius, P->eps, P->min_samples, labels_dev, &n_kept,
                                   &n_clusters, stream_, 2);
    info_out[0] = n_kept;
    info_out[1] = n_clusters;
    if (rc) {
        if (n_kept > 0 && P->neighbor_type != MODEST_GRAPH_RADIUS && n_kept <= P->k_neighbors) {
            info_out[3] = MODEST_STAGE_TOO_FEW_KEPT;
            return MODEST_OK;
        }
        return rc;
    }
    // 4. labels to the host (the copy rides on the synchronise of the statistics call) + cluster statistics
    MODEST_HIP_CHECK(hipMemcpyAsync(labels_h, labels_dev, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    std::vector<double> st((size_t)(n_clusters > 0 ? n_clusters : 1) * 6);
    if (n_clusters > 0) {
        rc = modest_cluster_stats(ctx, pts, n, stride, pp, labels_dev, n_clusters, plane2_out, P->quantile, st.data(),
                                  stream_);
        if (rc) return rc;
    } else {
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    }
    // 5. is_valid_cluster + relabelling
    finish_labels(P, n, n_clusters, st, labels_h, labels_out, info_out);
    return MODEST_OK;
}

// ---- the same stage for a CHAIN of scans -----------------------------------------------------------------------
// The ground fits stay per scan (their trial loops are data dependent: three round trips each); from the mask kernel
// on the scans advance together: ONE launch per kernel of the mask / graph / DBSCAN block and of the cluster
// statistics for the whole chain (cluster.hip, cluster_stats.hip: the scan is blockIdx.y), three round trips per
// chain.  Every scan works in its own context; results are those of separate modest_mask_stage calls, bit for bit.
extern "C" int modest_mask_stage_batch(const modest_mask_stage_scan *scans, int n_scans, const modest_mask_params *P,
                                       void *stream_) {
    MODEST_REQUIRE(scans != nullptr && P != nullptr && n_scans >= 1 && n_scans <= 64, "bad chain");
    if (n_scans == 1 || !modest_mask_chain_supported(P->neighbor_type, P->affinity_type) || getenv("MODEST_MASK_NOCHAIN")) {
        for (int s = 0; s < n_scans; ++s) {
            const modest_mask_stage_scan &q = scans[s];
            int rc = modest_mask_stage(q.ctx, q.pts_dev, q.n, q.stride, q.pp_dev, P, q.mt_key624, q.mt_pos, q.plane1_out,
                                       q.plane2_out, q.labels_out, q.info_out, stream_);
            if (rc) return rc;
        }
        return MODEST_OK;
    }
    hipStream_t stream = as_stream(stream_);
    struct Run {
        Mt19937 g;
        RansacFit A, B;
        int32_t *labels_dev = nullptr, *labels_h = nullptr;
        bool alive = false;
    };
    std::vector<Run> R((size_t)n_scans);
    for (int s = 0; s < n_scans; ++s)
        for (int k = s + 1; k < n_scans; ++k) MODEST_REQUIRE(scans[s].ctx != scans[k].ctx, "every scan of a chain needs its own context");
    // 1. candidates + thresholds of both fits of every scan: one launch per selection, ONE for all thresholds
    const float specs[10] = {P->max_hs1, P->range1[0], P->range1[1], P->range1[2], P->range1[3],
                             P->max_hs2, P->range2[0], P->range2[1], P->range2[2], P->range2[3]};
    std::vector<modest_ctx *> ctxs((size_t)n_scans);
    std::vector<const float *> ptsv((size_t)n_scans);
    std::vector<int> nv((size_t)n_scans), sv((size_t)n_scans);
    std::vector<float *> cA((size_t)n_scans), cB((size_t)n_scans);
    std::vector<int32_t> n_cand((size_t)2 * n_scans);
    std::vector<float> mad((size_t)2 * n_scans);
    // the trial loops of the fits on the device (plane.hip: rsd_*): no round trip until the mask kernel has run
    const bool dev_loop = P->max_trials >= 1 && P->max_trials <= MODEST_RSD_MAX_TRIALS && !getenv("MODEST_RANSAC_HOST");
    std::vector<char *> work((size_t)n_scans, nullptr);
    std::vector<modest_rsd_result *> res((size_t)n_scans, nullptr);
    std::vector<const uint32_t *> keys((size_t)n_scans, nullptr);
    std::vector<int32_t> poss((size_t)n_scans, 0);
    std::vector<const double *> plane1_dev((size_t)n_scans, nullptr);
    for (int s = 0; s < n_scans; ++s)
        if (scans[s].n_members_out) *scans[s].n_members_out = -1;   // (-1: no member list; set by finish_labels)
    // the raw cluster labels of ALL scans of the chain live in one device block and one pinned block of the first scan's context:
    // they travel to the host as ONE copy (sixteen copies of 120 KB cost 14 us of stream time each)
    size_t tot_lab = 0;
    std::vector<size_t> lab_off((size_t)n_scans, 0);
    for (int s = 0; s < n_scans; ++s) {
        lab_off[(size_t)s] = tot_lab;
        tot_lab += arena_sz((size_t)(scans[s].n > 0 ? scans[s].n : 1) * 4);
    }
    char *chain_lab_dev = nullptr, *chain_lab_h = nullptr;
    for (int s = 0; s < n_scans; ++s) {
        const modest_mask_stage_scan &q = scans[s];
        Run &r = R[(size_t)s];
        modest_ctx *ctx = q.ctx;
        MODEST_REQUIRE(ctx && q.mt_key624 && q.mt_pos && q.plane1_out && q.plane2_out && q.labels_out && q.info_out, "NULL argument");
        MODEST_REQUIRE(q.n >= 1 && (q.stride == 3 || q.stride == 4) && q.pts_dev && q.pp_dev, "bad scan");
        MODEST_HIP_CHECK(hipSetDevice(ctx->device));
        for (int k = 0; k < 8; ++k) q.info_out[k] = 0;
        const size_t b_cand = arena_sz((size_t)q.n * 12), b_lab = arena_sz((size_t)q.n * 4);
        const size_t b_work = dev_loop ? arena_sz(modest_rsd_work_bytes(q.n, P->max_trials)) : 0;
        const size_t extra = s == 0 ? tot_lab : 0;
        int rc = modest_ctx_reserve_hold(ctx, 2 * b_cand + b_lab + b_work + extra, b_lab + arena_sz(sizeof(modest_rsd_result)) + extra);
        if (rc) return rc;
        rc = modest_ctx_reserve_pinned(ctx, 16384);   // sized once: growing it between an enqueue and its read-back would free results
        if (rc) return rc;
        if (s == 0) {
            chain_lab_dev = ctx->hold + 2 * b_cand + b_lab + b_work;
            chain_lab_h = ctx->hold_pinned + b_lab + arena_sz(sizeof(modest_rsd_result));
        }
        cA[(size_t)s] = reinterpret_cast<float *>(ctx->hold);
        cB[(size_t)s] = reinterpret_cast<float *>(ctx->hold + b_cand);
        r.labels_dev = reinterpret_cast<int32_t *>(chain_lab_dev + lab_off[(size_t)s]);
        r.labels_h = reinterpret_cast<int32_t *>(chain_lab_h + lab_off[(size_t)s]);
        work[(size_t)s] = ctx->hold + 2 * b_cand + b_lab;
        res[(size_t)s] = reinterpret_cast<modest_rsd_result *>(ctx->hold_pinned + b_lab);
        keys[(size_t)s] = q.mt_key624;
        poss[(size_t)s] = *q.mt_pos;
        ctxs[(size_t)s] = ctx;
        ptsv[(size_t)s] = q.pts_dev;
        nv[(size_t)s] = q.n;
        sv[(size_t)s] = q.stride;
    }
    if (dev_loop) {
        int rc = modest_rsd_enqueue(ctxs.data(), ptsv.data(), nv.data(), sv.data(), n_scans, specs, cA.data(), cB.data(), keys.data(),
                                    poss.data(), P->max_trials, P->stop_probability, work.data(), res.data(), plane1_dev.data(), stream);
        if (rc) return rc;
        for (int s = 0; s < n_scans; ++s) R[(size_t)s].alive = true;
    } else {
    {
        int rc = modest_plane_prepare_chain(ctxs.data(), ptsv.data(), nv.data(), sv.data(), n_scans, specs, cA.data(), cB.data(),
                                            n_cand.data(), mad.data(), stream);
        if (rc) return rc;
    }
    // 2. the trial loops of all scans in lockstep: every round trip carries one batch (or refit) of every scan that
    //    still needs one.  A scan's own generator is drawn from in the reference's order (fit A, then fit B).
    std::vector<char> inA((size_t)n_scans, 0), inB((size_t)n_scans, 0);
    for (int s = 0; s < n_scans; ++s) {
        const modest_mask_stage_scan &q = scans[s];
        Run &r = R[(size_t)s];
        q.info_out[4] = n_cand[(size_t)2 * s];
        q.info_out[5] = n_cand[(size_t)2 * s + 1];
        {   // the arena of the scan's context holds any batch or refit of either fit BEFORE a launch is recorded: a
            // reserve that grows it later would free the block a recorded launch points into
            const int big = n_cand[(size_t)2 * s] > n_cand[(size_t)2 * s + 1] ? n_cand[(size_t)2 * s] : n_cand[(size_t)2 * s + 1];
            int rc = modest_ctx_reserve(q.ctx, modest_ransac_scratch_bound(big, P->batch > 64 ? P->batch : 64));
            if (rc) return rc;
        }
        if (n_cand[(size_t)2 * s] <= 300 || n_cand[(size_t)2 * s + 1] <= 300) {
            q.info_out[3] = MODEST_STAGE_SMALL_SET;
            continue;
        }
        memcpy(r.g.key, q.mt_key624, sizeof(r.g.key));
        r.g.pos = *q.mt_pos;
        r.A.init(q.ctx, cA[(size_t)s], n_cand[(size_t)2 * s], mad[(size_t)2 * s], &r.g, P->max_trials, P->stop_probability, P->batch, stream_);
        r.B.init(q.ctx, cB[(size_t)s], n_cand[(size_t)2 * s + 1], mad[(size_t)2 * s + 1], &r.g, P->max_trials, P->stop_probability, P->batch,
                 stream_);
        inA[(size_t)s] = 1;
    }
    modest_ctx *ctx0 = scans[0].ctx;   // the chained launches' tables live there
    for (;;) {   // fit A
        int active = 0;
        modest_ransac_capture *cap = modest_ransac_capture_begin();
        for (int s = 0; s < n_scans; ++s)
            if (inA[(size_t)s] && !R[(size_t)s].A.done()) {
                int rc = R[(size_t)s].A.enqueue_batch();
                if (rc) {
                    (void)modest_ransac_capture_launch(ctx0, cap, stream);
                    return rc;
                }
                R[(size_t)s].A.pending = true;
                ++active;
            }
        {
            int rc = modest_ransac_capture_launch(ctx0, cap, stream);   // one launch for the batches of all scans
            if (rc) return rc;
        }
        if (!active) break;
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        for (int s = 0; s < n_scans; ++s)
            if (inA[(size_t)s] && R[(size_t)s].A.pending) {
                R[(size_t)s].A.pending = false;
                int rc = R[(size_t)s].A.finish_batch();
                if (rc) return rc;
            }
    }
    {   // refit A + first batch of B
        int active = 0;
        modest_ransac_capture *cap = modest_ransac_capture_begin();
        for (int s = 0; s < n_scans; ++s) {
            if (!inA[(size_t)s]) continue;
            const modest_mask_stage_scan &q = scans[s];
            Run &r = R[(size_t)s];
            q.info_out[6] = r.A.n_trials;
            if (!r.A.have) {
                q.info_out[3] = MODEST_STAGE_NO_CONSENSUS;
                continue;
            }
            int rc = r.A.enqueue_refit();
            if (!rc) rc = r.B.enqueue_batch();
            if (rc) {
                (void)modest_ransac_capture_launch(ctx0, cap, stream);
                return rc;
            }
            inB[(size_t)s] = 1;
            ++active;
        }
        {
            int rc = modest_ransac_capture_launch(ctx0, cap, stream);   // all refits as one launch, then all first batches as one
            if (rc) return rc;
        }
        if (active) MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        for (int s = 0; s < n_scans; ++s) {
            if (!inB[(size_t)s]) continue;
            const modest_mask_stage_scan &q = scans[s];
            Run &r = R[(size_t)s];
            double model64[3];
            int32_t n_in = 0;
            bool degenerate = false;
            int rc = r.A.finish_refit(model64, &n_in, &degenerate);
            if (rc) return rc;
            if (degenerate) {
                q.info_out[3] = MODEST_STAGE_DEGENERATE;
                inB[(size_t)s] = 0;
                continue;
            }
            plane_from_model(model64, q.plane1_out);
            if ((rc = r.B.finish_batch())) return rc;
        }
    }
    for (;;) {   // the rest of fit B
        int active = 0;
        modest_ransac_capture *cap = modest_ransac_capture_begin();
        for (int s = 0; s < n_scans; ++s)
            if (inB[(size_t)s] && !R[(size_t)s].B.done()) {
                int rc = R[(size_t)s].B.enqueue_batch();
                if (rc) {
                    (void)modest_ransac_capture_launch(ctx0, cap, stream);
                    return rc;
                }
                R[(size_t)s].B.pending = true;
                ++active;
            }
        {
            int rc = modest_ransac_capture_launch(ctx0, cap, stream);
            if (rc) return rc;
        }
        if (!active) break;
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        for (int s = 0; s < n_scans; ++s)
            if (inB[(size_t)s] && R[(size_t)s].B.pending) {
                R[(size_t)s].B.pending = false;
                int rc = R[(size_t)s].B.finish_batch();
                if (rc) return rc;
            }
    }
    modest_ransac_capture *capB = modest_ransac_capture_begin();
    for (int s = 0; s < n_scans; ++s) {   // the second refit stays in flight: it rides along with the mask kernel
        if (!inB[(size_t)s]) continue;
        const modest_mask_stage_scan &q = scans[s];
        Run &r = R[(size_t)s];
        q.info_out[7] = r.B.n_trials;
        if (!r.B.have) {
            q.info_out[3] = MODEST_STAGE_NO_CONSENSUS;
            continue;
        }
        int rc = r.B.enqueue_refit();
        if (rc) {
            (void)modest_ransac_capture_launch(ctx0, capB, stream);
            return rc;
        }
        r.alive = true;
    }
    {
        int rc = modest_ransac_capture_launch(ctx0, capB, stream);   // the second refits of all scans: one launch
        if (rc) return rc;
    }
    }   // (host trial loops)
    // 3. the mask kernel of every scan that got this far, one launch
    std::vector<modest_mask_chain_scan> C;
    std::vector<int> who;
    for (int s = 0; s < n_scans; ++s)
        if (R[(size_t)s].alive) {
            const modest_mask_stage_scan &q = scans[s];
            modest_mask_chain_scan c{};
            c.ctx = q.ctx;
            c.pts = q.pts_dev;
            c.n = q.n;
            c.stride = q.stride;
            c.pp = q.pp_dev;
            c.plane4 = q.plane1_out;
            c.plane4_dev = plane1_dev[(size_t)s];
            c.labels = R[(size_t)s].labels_dev;
            C.push_back(c);
            who.push_back(s);
        }
    if (C.empty()) return MODEST_OK;
    const int B = (int)C.size();
    const double *only = P->use_only_range ? P->only_range : nullptr;
    modest_mask_chain_state *cst = nullptr;
    int rc = modest_mask_chain_count(C.data(), B, P->offset, only, P->limit_range, P->neighbor_type, P->k_neighbors, P->radius,
                                     &cst, stream);
    struct Free {
        modest_mask_chain_state *p;
        ~Free() { modest_mask_chain_free(p); }
    } guard{cst};
    if (rc) return rc;
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < B; ++i) {
        const modest_mask_stage_scan &q = scans[who[(size_t)i]];
        Run &r = R[(size_t)who[(size_t)i]];
        if (dev_loop) {   // everything of the two fits arrived with this synchronise
            const modest_rsd_result &z = *res[(size_t)who[(size_t)i]];
            MODEST_REQUIRE(z.status >= 0, "the fits of a scan did not finish");
            q.info_out[4] = z.n_cand[0], q.info_out[5] = z.n_cand[1];
            q.info_out[6] = z.n_trials[0], q.info_out[7] = z.n_trials[1];
            if (z.status != 0) {
                q.info_out[3] = z.status;
                C[(size_t)i].alone = 1;
                continue;
            }
            memcpy(q.plane1_out, z.plane1, sizeof(z.plane1));
            memcpy(q.plane2_out, z.plane2, sizeof(z.plane2));
            memcpy(q.mt_key624, z.mt_key, sizeof(z.mt_key));
            *q.mt_pos = z.mt_pos;
            continue;
        }
        double model64[3];
        int32_t n_in = 0;
        bool degenerate = false;
        if ((rc = r.B.finish_refit(model64, &n_in, &degenerate))) return rc;
        if (degenerate) {
            q.info_out[3] = MODEST_STAGE_DEGENERATE;
            C[(size_t)i].alone = 1;   // out of the chain; its cell counters hold this scan's counts
            continue;
        }
        plane_from_model(model64, q.plane2_out);
        memcpy(q.mt_key624, r.g.key, sizeof(r.g.key));
        *q.mt_pos = r.g.pos;
    }
    // 4. graph + DBSCAN + labels[ptc_mask] = ... of the chain
    rc = modest_mask_chain_cluster(C.data(), B, cst, P->neighbor_type, P->affinity_type, P->k_neighbors, P->radius, P->eps,
                                   P->min_samples, stream);
    if (rc) return rc;
    // 5. labels to the host + cluster statistics of the chain (one launch, one synchronise)
    std::vector<modest_stats_chain_scan> T((size_t)B);
    std::vector<std::vector<double>> st((size_t)B);
    for (int i = 0; i < B; ++i) {
        const modest_mask_stage_scan &q = scans[who[(size_t)i]];
        Run &r = R[(size_t)who[(size_t)i]];
        modest_stats_chain_scan &t = T[(size_t)i];
        memset(&t, 0, sizeof(t));
        if (C[(size_t)i].alone) {
            if (q.info_out[3] == 0) {
                q.info_out[0] = C[(size_t)i].n_kept;
                q.info_out[3] = MODEST_STAGE_TOO_FEW_KEPT;
            }
            continue;
        }
        q.info_out[0] = C[(size_t)i].n_kept;
        q.info_out[1] = C[(size_t)i].n_clusters;
        st[(size_t)i].assign((size_t)(C[(size_t)i].n_clusters > 0 ? C[(size_t)i].n_clusters : 1) * 6, 0.0);
        t.ctx = q.ctx;
        t.pts = q.pts_dev;
        t.n = q.n;
        t.stride = q.stride;
        t.pp = q.pp_dev;
        t.labels = r.labels_dev;
        t.n_clusters = C[(size_t)i].n_clusters;
        t.plane4 = q.plane2_out;
        t.out_host = st[(size_t)i].data();
    }
    MODEST_HIP_CHECK(hipMemcpyAsync(chain_lab_h, chain_lab_dev, tot_lab, hipMemcpyDeviceToHost, stream));   // every scan's labels: one copy
    T[0].ctx = T[0].ctx ? T[0].ctx : scans[who[0]].ctx;   // the chain table lives in the first scan's context
    rc = modest_cluster_stats_chain(T.data(), B, P->quantile, stream);
    if (rc) return rc;
    timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int i = 0; i < B; ++i) {
        if (C[(size_t)i].alone) continue;
        const modest_mask_stage_scan &q = scans[who[(size_t)i]];
        finish_labels(P, q.n, C[(size_t)i].n_clusters, st[(size_t)i], R[(size_t)who[(size_t)i]].labels_h, q.labels_out, q.info_out,
                      q.members_out, q.n_members_out);
    }
    if (getenv("MODEST_CHAIN_TRACE")) {
        clock_gettime(CLOCK_MONOTONIC, &ts1);
        fprintf(stderr, "[mask_stage_batch %d scans] finish_labels %.3f ms\n", B, 1e3 * (double)(ts1.tv_sec - ts0.tv_sec) + 1e-6 * (double)(ts1.tv_nsec - ts0.tv_nsec));
    }
    return MODEST_OK;
}
