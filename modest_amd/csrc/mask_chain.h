// The mask / graph / DBSCAN block and the cluster statistics for a chain of scans (cluster.hip, cluster_stats.hip),
// used by modest_mask_stage_batch (scan_driver.hip).  Host-side plan of one scan of the chain:
#pragma once
#include "common.h"

struct modest_mask_chain_scan {
    modest_ctx *ctx;        // the scan's OWN context (scratch, pinned words, persistent cell counters)
    const float *pts;       // [dev] (n, stride) scan rows
    int n, stride;
    const float *pp;        // [dev] (n) PP scores
    const double *plane4;   // [host] the plane of the mask
    const double *plane4_dev;   // [dev] or NULL; when set the plane is read from here in the stream (plane4 is ignored)
    int32_t *labels;        // [dev] (n) int32
    int32_t n_kept, n_clusters;
    int alone;              // set when the scan has to be finished by the single-scan calls
};
struct modest_mask_chain_state;

bool modest_mask_chain_supported(int neighbor_type, int affinity_type);
// enqueue only: mask / compaction / cell count of every scan in one launch
int modest_mask_chain_count(modest_mask_chain_scan *S, int B, double offset, const double *only_range4,
                            const double *limit_range4, int neighbor_type, int k_neighbors, double radius,
                            modest_mask_chain_state **state_out, hipStream_t stream);
// after the caller's synchronise: the graph / DBSCAN launches of the chain, a synchronise, n_kept / n_clusters filled
int modest_mask_chain_cluster(modest_mask_chain_scan *S, int B, modest_mask_chain_state *st, int neighbor_type,
                              int affinity_type, int k_neighbors, double radius, double eps, int min_samples,
                              hipStream_t stream);
void modest_mask_chain_free(modest_mask_chain_state *st);

// cluster statistics of every scan of the chain in one launch (cluster_stats.hip); out_host[s]: 6 doubles per cluster
struct modest_stats_chain_scan {
    modest_ctx *ctx;
    const float *pts;
    int n, stride;
    const float *pp;
    const int32_t *labels;   // [dev]
    int n_clusters;
    const double *plane4;    // [host]
    double *out_host;        // (n_clusters, 6)
};
int modest_cluster_stats_chain(modest_stats_chain_scan *S, int B, double quantile, hipStream_t stream);

// candidates + MAD thresholds of both fits for every scan of a chain (plane.hip): one synchronise; n_cand2_host /
// mad2_host: 2 entries per scan
int modest_plane_prepare_chain(modest_ctx *const *ctxs, const float *const *pts, const int *n, const int *stride, int B,
                               const float *specs10, float *const *candA, float *const *candB, int32_t *n_cand2_host,
                               float *mad2_host, hipStream_t stream);

// The two ground fits of every scan of a chain with their trial loops ON THE DEVICE (plane.hip: rsd_*): selection,
// thresholds, triplets from the scan's own MT19937 state, all trials, sklearn's accept rule, the refits -- enqueue only,
// no synchronise.  work_dev[s]: modest_rsd_work_bytes(n[s], max_trials) bytes of device memory that stay untouched until
// the stream has passed the launches; res_host[s]: pinned host memory, valid after the caller's synchronise.
// plane1_dev_out[s]: where the first fit's plane (4 doubles) will be on the device (the mask kernel of the chain reads it).
struct modest_rsd_result {
    int32_t n_cand[2];
    float mad[2];
    int32_t n_trials[2];
    int32_t status;      // 0 or MODEST_STAGE_*: the scan goes back to the host statement, generator untouched
    int32_t mt_pos;
    double plane1[4], plane2[4];
    uint32_t mt_key[624];   // the generator behind the executed trials of both fits (status 0 only)
};
constexpr int MODEST_RSD_MAX_TRIALS = 128;
size_t modest_rsd_work_bytes(int n, int max_trials);
int modest_rsd_enqueue(modest_ctx *const *ctxs, const float *const *pts, const int *n, const int *stride, int B,
                       const float *specs10, float *const *candA, float *const *candB, const uint32_t *const *mt_key624,
                       const int32_t *mt_pos, int max_trials, double stop_probability, char *const *work_dev,
                       modest_rsd_result *const *res_host, const double **plane1_dev_out, hipStream_t stream);

// lowest point inside each box footprint, boxes of several scans in one launch (boxfit.hip)
int modest_lowest_point_multi(modest_ctx *ctx, const double *const *pts_rect, const int *n_pts, const double *boxes6_host,
                              int n_boxes, double *bottom_host, void *stream);

// RANSAC launches of a chain (plane.hip): between begin and launch the enqueue halves of the trial / refit phases on
// this thread are recorded; launch runs all recorded refits as ONE launch, then all recorded trial batches as ONE
struct modest_ransac_capture;
modest_ransac_capture *modest_ransac_capture_begin();
int modest_ransac_capture_launch(modest_ctx *ctx0, modest_ransac_capture *cap, hipStream_t stream);

// self-IoU matrices of several box sets in one launch (iou3d.hip): boxes_host[s] (n[s],7) float32, out_host[s] (n[s],n[s])
int modest_boxes_self_iou_bev_host_batch(modest_ctx *ctx, const float *const *boxes_host, const int *n, int B,
                                         float *const *out_host, void *stream);

// Calibration.project_velo_to_rect of several scans in one launch (transform.hip)
int modest_project_velo_to_rect_multi(modest_ctx *ctx, const float *const *pts, const int *n, const int *stride,
                                      double *const *out, int B, const double *V2C12, const double *R09, void *stream);
