// Per-scan driver of the mask stage (generate_mask.py:52-88 + clustering_utils.py:119-135): both
// ground fits, the plane / range mask, the affinity graph + DBSCAN, the cluster statistics, the
// is_valid_cluster rules and the relabelling behind ONE call, so that no interpreter runs between
// the device round trips of these steps.  Host code only: every device step is one of the library's
// own entry points (their temporaries come from the context arena at offset 0; what has to live
// across them -- candidate sets, the scan-sized labels -- sits in the context's hold buffers).
#include "common.h"
#include "ransac_host.h"
#include <cmath>
#include <cstring>
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
    // 5. is_valid_cluster (clustering_utils.py:94-117) + relabelling (:131-135) as one table look-up:
    //    table[l + 1] = rank of label l among the surviving values; -1 survives iff it occurs or a
    //    cluster is dropped
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
    for (int i = 0; i < n; ++i) labels_out[i] = table[(size_t)(labels_h[i] + 1)];
    info_out[2] = n_valid > 0 ? (int32_t)(next - 1) : 0;   // largest final label = number of box candidates
    if (n_clusters == 0) {   // compact_labels of the raw labels: all -1 -> all 0
        for (int i = 0; i < n; ++i) labels_out[i] = 0;
        info_out[2] = 0;
    }
    return MODEST_OK;
}
