/*
 * modest_hip.h — C ABI of libmodest_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the seed-label hot path of YurongYou/MODEST
 * (pre_compute_pp_score -> generate_mask -> gen_label_files).  Every entry
 * point names the reference interface it replaces (paths relative to the
 * reference checkout).  Conventions, all entry points:
 *
 *   - plain pointers and sizes; no torch / numpy types cross this boundary;
 *   - pointers marked [dev] are device (HBM) pointers, [host] are host
 *     pointers; the caller owns every buffer (reference convention:
 *     generate_cluster_mask/utils/iou3d_nms/iou3d_nms_utils.py:47,103);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     work is enqueued on it, the call does not synchronise unless the
 *     description says "blocking";
 *   - return value: 0 = ok, <0 = error (never exit(); the reference's
 *     CHECK_INPUT exits the process, src/iou3d_nms.cpp:14-26);
 *     modest_last_error() returns a thread-local message;
 *   - per-stream scratch lives in a modest_ctx (one per process x GPU).
 */
#ifndef MODEST_HIP_H
#define MODEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODEST_OK 0
#define MODEST_ERR_ARG (-1)
#define MODEST_ERR_HIP (-2)
#define MODEST_ERR_CAPACITY (-3)
#define MODEST_ERR_NODEVICE (-4)

typedef struct modest_ctx modest_ctx;

/* ---- library / context ------------------------------------------------ */
int modest_version(void);
const char *modest_last_error(void);
/* Number of visible HIP devices (0 when no GPU; never fails). */
int modest_device_count(void);
int modest_ctx_create(int device, modest_ctx **out);
int modest_ctx_destroy(modest_ctx *ctx);
/* Measurement hook (no reference counterpart): while enabled, the dominant
 * operation of the path -- the neighbour-count stage of modest_pp_count /
 * modest_pp_score / modest_pp_score_frames (every kernel of ONE scan between the
 * live index build and the last join) -- is bracketed by a HIP event pair on the launch stream; collect
 * returns the elapsed milliseconds of up to `cap` calls since begin and
 * disables the hook.                                                          */
/* The context's scratch arena grows on demand (a grow = device synchronise + free + allocate: tens of milliseconds when several
 * processes share the GPU).  A caller that knows what is coming -- the CLI before its first batch, which is shorter than the
 * later ones -- takes the arena in ONE driver call up front.  No reference counterpart (the reference allocates per call).  */
int modest_ctx_reserve_arena(modest_ctx *ctx, uint64_t bytes);
/* Loads the library's device code now (the HIP runtime loads a translation unit's code object at the first launch of one of its
 * kernels: ~0.1 s spread over the first scan of a process otherwise).  The CLIs call it before their loop clocks start.        */
int modest_warmup(modest_ctx *ctx);
int modest_ctx_profile_begin(modest_ctx *ctx, int capacity);
int modest_ctx_profile_collect(modest_ctx *ctx, float *ms_out_host, int cap,
                               int *n_out_host);

/* Calibration.project_velo_to_rect (utils/kitti_util.py:327-329) of the scan rows, float64:
 * out[i] = R0 @ (V2C @ [x y z 1]) with the rounding of numpy's two dgemm calls (one fma chain per
 * output element).  The rect-frame copy of the scan that get_obj's lowest-point search reads
 * (pointcloud_utils.py:278-290) is produced where it is consumed instead of being uploaded.
 * pts [dev] (n,stride) f32; V2C12 / R09 [host] row-major 3x4 / 3x3 f64; out [dev] (n,3) f64. */
int modest_project_velo_to_rect(modest_ctx *ctx, const float *pts_dev, int n, int stride,
                                const double *V2C12, const double *R09, double *out_dev,
                                void *stream);

/* ---- a4  transform_points  (utils/pointcloud_utils.py:11-19) ----------
 * out[i] = (x*T[r][0] (+fma) y*T[r][1] (+fma) z*T[r][2]) + T[r][3], float32,
 * i.e. [p,1] @ T^T, rows 0..2.  `in_stride` is 3 or 4 floats per point
 * (velodyne .bin frames are (n,4): load_velo_scan, pointcloud_utils.py:22-25).
 * T16 is a host pointer to a row-major 4x4 float32 matrix.
 * If remove_center != 0 (nuScenes, pre_compute_pp_score.py:48-52,141-142)
 * points with x in [-1.15,1.75) and y in [-0.65,0.65) of the UNtransformed
 * frame are dropped and the output is compacted in input order;
 * *n_out_dev [dev] receives the number written (may be NULL when
 * remove_center == 0).                                                     */
int modest_transform_points(modest_ctx *ctx, const float *in_dev, int64_t n,
                            int in_stride, const float *T16_host,
                            int remove_center, float *out_xyz_dev,
                            int64_t *n_out_dev, void *stream);

/* ---- a6+a7  cKDTree build + count_neighbors ---------------------------
 * (pre_compute_pp_score.py:54-60,188-193).  For every live point i and
 * traversal t: counts[i*T+t] = #{ h in hist[t] : |p_i-h|^2 <= radius^2 },
 * inclusive, predicate evaluated in float64 on the float32 coordinates
 * exactly as scipy's cKDTree.query_ball_point does.
 *   live_xyz [dev]  (n_live,3) f32, already in the common frame
 *   hist_xyz [dev]  (M,3) f32 stacked traversals, already transformed
 *   trav_offsets [host] (T+1) int64 prefix offsets into hist (points)
 *   counts [dev] (n_live,T) int32 (the reference returns int64; values
 *   are < 2^31 because M < 2^31 is required).                              */
int modest_pp_count(modest_ctx *ctx, const float *live_xyz_dev, int n_live,
                    const float *hist_xyz_dev, const int64_t *trav_offsets_host,
                    int n_trav, double radius, int32_t *counts_dev,
                    void *stream);

/* ---- a8  compute_ephe_score (pre_compute_pp_score.py:68-75) ------------
 * P = c/(sum_t c + 1e-8); H = sum_t -P ln(P+1e-8) / ln T, float64, stored
 * as float32 (np.save(...astype(float32)), :195-196).                      */
int modest_pp_entropy(modest_ctx *ctx, const int32_t *counts_dev, int n_live,
                      int n_trav, float *H_dev, void *stream);

/* a7+a8 fused: counts_dev may be NULL (scratch is used).                  */
int modest_pp_score(modest_ctx *ctx, const float *live_xyz_dev, int n_live,
                    const float *hist_xyz_dev, const int64_t *trav_offsets_host,
                    int n_trav, double radius, int32_t *counts_dev,
                    float *H_dev, void *stream);

/* ---- a1+a3+a4+a5+a6+a7 over a FRAME STORE (history assembly on the device) ------
 * The reference re-reads, transforms and stacks every history frame for every scan
 * (pre_compute_pp_score.py:132-150: load_velo_scan + remove_center + transform_points +
 * np.concatenate) and builds one cKDTree per traversal (:188-190).  Here a raw frame enters the
 * frame store ONCE: modest_frame_sort orders its points by the 8x8-cell tile of a world lattice
 * (cell edge c = r*(1+2^-10), the lattice is shared by all frames of a data set) and writes the
 * prefix table tile -> first point.  A scan then names its frames by descriptor
 * (modest_pp_frame: store buffers + traversal id + the float32 relative pose of
 * get_relative_pose, :27-28) and modest_pp_score_frames streams the frames through that table,
 * applies the pose on the fly (transform_points' float32 rounding) and counts neighbours: no
 * stacked, transformed copy of the history exists.
 *
 * W [host] 2x4 float64, rows x and y of (1/c) * (raw frame -> world lattice metres).
 * TX0, TY0: global tile coordinates of the table's first tile; the table covers
 * MODEST_FRAME_NTF x MODEST_FRAME_NTF tiles (callers centre it on the sensor origin).
 * Outputs [dev]: xyz (n,3) f32 tile-sorted raw points, perm (n) u32 sorted -> original index,
 * tab (NTF*NTF+1) u32 prefix offsets (row-major tiles); tab[NTF*NTF] = points inside the table,
 * the rest of xyz are outliers.  n_inside_host[k] receives that count (a scan whose frames have
 * outliers must use the stacked path, modest_pp_score).  Blocking.                             */
#define MODEST_FRAME_NTF 128
typedef struct {
    const float *raw_dev;   /* (n, stride) f32, stride 3 or 4 (velodyne .bin: load_velo_scan) */
    int32_t n, stride;
    int32_t TX0, TY0;
    double W[8];
    float *xyz_dev;
    uint32_t *perm_dev;
    uint32_t *tab_dev;
} modest_frame_sort_job;
int modest_frame_table_tiles(void);   /* MODEST_FRAME_NTF of the built library */
int modest_frame_sort(modest_ctx *ctx, const modest_frame_sort_job *jobs_host, int n_jobs,
                      int32_t *n_inside_host, void *stream);
/* The same sort, NOT blocking (an ingest thread keeps several batches in flight behind the compute stream's
 * kernels).  jobs_scratch_dev [dev]: n_jobs * MODEST_FRAME_SORT_JOB_BYTES, caller-owned, alive until the
 * launch has run; n_inside_pinned [PINNED host] (n_jobs) int32: written by the kernel itself, readable once
 * `stream` has passed the launch.  The grid sort of small batches also keeps per-point rank / bin words and its chunk
 * table in the CONTEXT's scratch arena, in stream order (and grows the arena on first use: a device synchronise): `ctx`
 * must not be shared with calls that run on another stream at the same time -- an ingest thread owns a context of its own
 * (pre_compute_pp_score.py: FrameLoader).                                                                           */
#define MODEST_FRAME_SORT_JOB_BYTES 128
int modest_frame_sort_async(modest_ctx *ctx, const modest_frame_sort_job *jobs_host, int n_jobs,
                            void *jobs_scratch_dev, int32_t *n_inside_pinned_host, void *stream);

#define MODEST_FRAME_REMOVE_CENTER 1   /* remove_center, pre_compute_pp_score.py:48-52 */
typedef struct {
    const float *xyz_dev;      /* store buffers of the frame */
    const uint32_t *tab_dev;
    int32_t n, TX0, TY0;
    int32_t trav;              /* traversal index in [0, n_trav) (ignored for the live scan) */
    int32_t flags;             /* MODEST_FRAME_REMOVE_CENTER */
    float rel[12];             /* rows 0..2 of get_relative_pose(...).astype(float32) */
} modest_pp_frame;
/* counts[i*T+t] / H[i] are indexed by the ORIGINAL point order of the live frame (perm).
 * A [host] 2x4 float64: rows x, y of (1/c) * (common frame -> world lattice metres); it must agree
 * with W * rel^-1 of every frame to better than r/1024 (callers check 1e-4 m).
 * counts_dev may be NULL (scratch) when only H is wanted; H_dev may be NULL.  n_trav <= 64.
 * Not blocking.                                                                              */
int modest_pp_score_frames(modest_ctx *ctx, const modest_pp_frame *live_host,
                           const uint32_t *live_perm_dev, const modest_pp_frame *frames_host,
                           int n_frames, int n_trav, const double *A8_host, double radius,
                           int32_t *counts_dev, float *H_dev, void *stream);

/* The same operation for SEVERAL scans in one chain of launches (the reference's scan loop,
 * pre_compute_pp_score.py:122-196, has no dependency between iterations): every kernel of the chain takes the
 * scan as a second grid dimension, so the nine sub-10 us launches are paid once per batch and the persistent
 * join workgroups take slices of any scan of the batch.  Results are those of n_scans separate calls, bit for
 * bit.  Arrays of n_scans entries [host]; counts_dev / H_dev (arrays or entries) may be NULL as above; scans the
 * batched kernels do not cover (no live points, no history) make the call fall back to separate calls.
 * Scratch: ~0.26 GB per Lyft-shape scan of the batch.  Not blocking.                                      */
int modest_pp_score_frames_batch(modest_ctx *ctx, int n_scans, const modest_pp_frame *const *live_host,
                                 const uint32_t *const *live_perm_dev, const modest_pp_frame *const *frames_host,
                                 const int32_t *n_frames, int n_trav, double radius,
                                 int32_t *const *counts_dev, float *const *H_dev, void *stream);

/* ---- the same operation for a BLOCK of scans that share history frames -------------------------
 * Consecutive scans of a shard share 35 of their 36 frames per traversal
 * (data_preprocessing/lyft/split_traintest.py:64,97): the reference re-stacks and re-indexes them scan
 * after scan (pre_compute_pp_score.py:132-150,188-190).  modest_pp_score_block takes the UNION of the
 * block's frames once -- every point is copied, raw, into the list of its world-lattice tile (list sizes
 * follow from the frames' prefix tables: no count pass) and every list is ordered by lattice cell -- and
 * then joins every scan against that store: records are read straight into registers, the scan's own
 * float32 relative pose (get_relative_pose, :27-28; transform_points' rounding) is applied per record,
 * neighbours are counted with scipy's float64 predicate (:54-60).  Results are those of
 * modest_pp_score_frames, bit for bit.
 *
 * frames_host [host] (n_frames): the union, one entry per (frame, occurrence): a frame that a scan's index list names
 *   k times (split_traintest.py:86-101; the reference stacks it k times, pre_compute_pp_score.py:132-150) is k entries
 *   with the same buffers, and that scan's member list names each of them once -- a member list that names one entry
 *   twice is refused (MODEST_ERR_ARG).  lat = the 2x4 float64 map the frame was
 *   sorted with (modest_frame_sort_job::W); flags: MODEST_FRAME_REMOVE_CENTER (all or none).  The ORDER of the table is
 *   the caller's: a scan reads, cell by cell, the records of the table range [its first frame, its last frame] (and masks
 *   the frames in between that are not its own), so an order in which every scan's frames are contiguous -- for the
 *   sliding windows of a shard: by (first scan, last scan) that uses the frame -- saves the other scans' records.
 * scans_host [host] (n_scans): the live frame of the store (xyz / perm / tab, lat as above, rel = its float32
 *   relative pose), the scan's history frames as (member_slot = index into frames_host, member_trav,
 *   member_rel = rows 0..2 of the frame's float32 relative pose IN THIS SCAN), outputs as above.
 * Requirements (the caller checks them; modest_pp_score_frames_batch has none of them): no frame has
 *   points outside its table; every pose agrees with the lattice to 1e-4 m (lat == (1/cell) * W and
 *   W ~ A * rel); cell >= radius * (1 + 2^-9); the live frames' tables lie within
 *   modest_pp_block_limits() tiles of each other.  n_trav <= 64.  Not blocking.                      */
typedef struct {
    const float *xyz_dev;
    const uint32_t *tab_dev;
    int32_t n, TX0, TY0, flags;
    double lat[8];
} modest_pp_block_frame;
typedef struct {
    const float *xyz_dev;
    const uint32_t *perm_dev;
    const uint32_t *tab_dev;
    int32_t n, TX0, TY0, n_members;
    double lat[8];
    float rel[12];
    const int32_t *member_slot;   /* [host] */
    const int32_t *member_trav;   /* [host] */
    const float *member_rel;      /* [host] (n_members, 12) */
    int32_t *counts_dev;          /* [dev] (n, n_trav) or NULL */
    float *H_dev;                 /* [dev] (n) or NULL */
} modest_pp_block_scan;
int modest_pp_block_limits(int32_t *max_window_tiles, int32_t *max_scans, int32_t *max_frames);
int modest_pp_score_block(modest_ctx *ctx, const modest_pp_block_frame *frames_host, int n_frames,
                          const modest_pp_block_scan *scans_host, int n_scans, int n_trav, double radius,
                          double cell, void *stream);
/* The same for scans with DIFFERENT numbers of traversals in one block: n_trav_scan [host] (n_scans), 1..64 each.
 * The reference accepts a traversal per scan (closest pose within 3 m, data_preprocessing/lyft/split_traintest.py:17,79)
 * and asks for at least two (:111; pre_compute_pp_score.py:125-126), so T changes along a sequence
 * (pre_compute_pp_score.py:132-150,181-194 run per scan with that scan's own list).  counts_dev of scan s is
 * (n, n_trav_scan[s]); member_trav of scan s < n_trav_scan[s]; H of scan s is normalised by ln n_trav_scan[s] (:72).       */
int modest_pp_score_block_mixed(modest_ctx *ctx, const modest_pp_block_frame *frames_host, int n_frames,
                                const modest_pp_block_scan *scans_host, int n_scans, const int32_t *n_trav_scan,
                                double radius, double cell, void *stream);

/* ---- the tables of a block call from a frame store's slot tables (host only) -------------------------------------------
 * What a caller of modest_pp_score_block[_mixed] has to build from the scans' index lists (valid_idx_info.pkl,
 * data_preprocessing/lyft/split_traintest.py:79-101; the reference stacks every scan's list anew, pre_compute_pp_score.py:132-150):
 * the union of the lists as (frame, occurrence) entries ordered by the middle of the interval of scans that use an entry, every scan's
 * members as indices into it, and the checks listed under modest_pp_score_block (no frame with points outside its table, every pose
 * within 1e-4 m of the lattice, the live tables within the block window, one remove_center flag).  store: slot -> modest_pp_frame record
 * (buffers and table origin), W (n_slots,4,4) float64 raw frame -> world, lat (n_slots,8) = modest_frame_sort_job::W, perm_dev
 * (n_slots) the frames' perm buffers as integers, clean (n_slots) bytes: 1 = no point outside the table.
 * Per scan s: live_host[s] = the live frame's record with rel = its float32 relative pose; members_host[s] = n_members[s] records
 * (trav, flags, rel = the frame's float32 relative pose in this scan); slots_host[s] = n_members[s] + 1 store slots: the members',
 * then the live frame's.  apply_rule != 0: the measured rule between block and chain (>= 4 scans, >= 12 entries per traversal and
 * scan, union <= 4 x a scan's entries and <= 2048).  Outputs [host]: frames_out (capacity frames_cap entries; *n_frames_out used),
 * scans_out (n_scans; counts_dev / H_dev left NULL for the caller), member_slot_out / member_trav_out (sum of n_members) and
 * member_rel_out (12 floats per member), which scans_out points into.
 * Returns MODEST_BLOCK_TABLES_BLOCK (tables built), _CHAIN (the block path does not apply: use modest_pp_score_frames_batch),
 * _SPLIT (worth the block path in two halves), -1 = bad arguments.                                                                 */
#define MODEST_BLOCK_TABLES_BLOCK 0
#define MODEST_BLOCK_TABLES_CHAIN 1
#define MODEST_BLOCK_TABLES_SPLIT 2
typedef struct {
    const modest_pp_frame *records;
    const double *W;
    const double *lat;
    const uint64_t *perm_dev;
    const uint8_t *clean;
    int64_t n_slots;
    double radius, cell;
    int32_t window_span;   /* tiles the live tables' origins may differ by (window of modest_pp_block_limits - table - 2) */
} modest_pp_store_view;
int modest_pp_block_tables(const modest_pp_store_view *store, int n_scans, const modest_pp_frame *const *live_host,
                           const modest_pp_frame *const *members_host, const int64_t *const *slots_host,
                           const int32_t *n_members, const int32_t *n_trav_scan, int apply_rule,
                           modest_pp_block_frame *frames_out, int32_t frames_cap, int32_t *n_frames_out,
                           modest_pp_block_scan *scans_out, int32_t *member_slot_out, int32_t *member_trav_out,
                           float *member_rel_out);

/* ---- a9  estimate_plane / RANSACRegressor inner loops -----------------
 * (utils/pointcloud_utils.py:44-65; sklearn RANSACRegressor defaults).
 * Candidate selection: z<max_hs, xlo<x<xhi, ylo<y<yhi (strict), compacted in
 * input order into cand_xyz [dev] (capacity n), *n_cand_dev [dev].
 * n_cand_dev may be any device-accessible word: device memory, or pinned host
 * memory (hipHostMalloc) -- then the count is on the host after a stream
 * synchronise, without a copy.  Same for n_kept_dev below.                   */
int modest_plane_candidates(modest_ctx *ctx, const float *pts_dev, int n,
                            int stride, float max_hs, float xlo, float xhi,
                            float ylo, float yhi, float *cand_xyz_dev,
                            int32_t *cand_idx_dev, int32_t *n_cand_dev,
                            void *stream);
/* MAD threshold: median(|z - median(z)|) over the candidates, float32
 * arithmetic, numpy.median semantics (mean of the two middle values when n
 * is even).  Blocking; result written to *mad_host.                         */
int modest_mad_threshold(modest_ctx *ctx, const float *cand_xyz_dev,
                         int n_cand, float *mad_host, void *stream);
/* The same for up to four candidate sets in ONE launch (one workgroup each): a scan's two plane fits
 * (generate_mask.py:55-56 and clustering_utils.py:126) take their thresholds from one call.
 * cand_xyz_dev / n_cand: host arrays of `count` device pointers / sizes (each >= 1); mad_host[count].
 * Blocking.                                                                 */
int modest_mad_threshold_batch(modest_ctx *ctx, const float *const *cand_xyz_dev,
                               const int32_t *n_cand, int count, float *mad_host,
                               void *stream);
/* sklearn.linear_model.RANSACRegressor().fit as estimate_plane calls it (pointcloud_utils.py:52-53:
 * LinearRegression, min_samples 3, residual threshold = MAD(z) passed in as `thr`, max_trials 100,
 * stop_probability 0.99) behind one call: triplets drawn from numpy's legacy MT19937 stream exactly
 * as sklearn's sample_without_replacement consumes it (key624 / pos = RandomState.get_state()[1:3],
 * advanced in place by the EXECUTED trials), batches of `batch` <= 64 trials scored per device round
 * trip, the sequential accept rule with the dynamic trial bound, the final least-squares refit.
 * n_cand must exceed 300 (below that sklearn samples with another method: host path).
 * model64_out[3] = (c0, c1, b) of the refit; best_model_out[3] the winning trial's float32 plane;
 * triplets_out (max_trials,3) optional.  status_out: 0 ok, 1 no consensus set (sklearn raises
 * ValueError), 2 degenerate consensus set (refit on the host from best_model_out).            */
int modest_ransac_plane(modest_ctx *ctx, const float *cand_xyz_dev, int n_cand, float thr,
                        uint32_t *mt_key624, int32_t *mt_pos, int max_trials,
                        double stop_probability, int batch, double *model64_out,
                        float *best_model_out, int32_t *triplets_out, int32_t *n_trials_out,
                        int32_t *n_inliers_out, int32_t *status_out, void *stream);
/* The generator half alone (host only, no device work): n_trials triplets of
 * sample_without_replacement(n_population > 300, 3) from the same stream.                    */
int modest_mt19937_triplets(uint32_t *mt_key624, int32_t *mt_pos, uint32_t n_population,
                            int n_trials, int32_t *triplets_out);

/* The RNG-independent part of the two ground fits of a scan (generate_mask.py:55-56 and
 * clustering_utils.py:126) in one call: both candidate selections from one pass over the rows
 * (modest_plane_candidates semantics, compacted in row order) and both MAD thresholds, two
 * launches and one stream sync.  specs10 [host] = {max_hs, xlo, xhi, ylo, yhi} x 2;
 * candA / candB [dev] (n,3) f32 outputs; n_cand2_host[2]; mad2_host[2] (NaN for an empty set). */
int modest_plane_prepare(modest_ctx *ctx, const float *pts_dev, int n, int stride,
                         const float *specs10, float *candA_dev, float *candB_dev,
                         int32_t *n_cand2_host, float *mad2_host, void *stream);
/* Score K trial models z = c0*x + c1*y + b (float32, pred = fma chain
 * fmaf(y,c1,x*c0)+b) against all candidates in ONE launch:
 *   n_inliers[k] = #{ |z - pred| <= thr },  sse[k], sy[k], syy[k] over the
 *   inliers in float64 (for the R^2 tie-break).  models [host] (K,3) f32.
 *   outputs [host], blocking.                                               */
int modest_ransac_score_trials(modest_ctx *ctx, const float *cand_xyz_dev,
                               int n_cand, const float *models_host, int K,
                               float thr, int32_t *n_inliers_host,
                               double *sse_host, double *sy_host,
                               double *syy_host, void *stream);
/* One round trip for a batch of RANSAC trials: (optional) MAD threshold, the
 * exact-fit plane through the three candidates `triplets[k]` of every trial
 * (float64 -> float32; NaN for a degenerate triplet) and the scoring of all K
 * planes.  *thr_inout [host]: < 0 on entry = compute the MAD threshold on the
 * device (returned on exit), otherwise the threshold to use.  triplets [host]
 * (K,3) int32 indices into the candidates; models_out [host] (K,3) float32.
 * Blocking.                                                                   */
int modest_ransac_trials(modest_ctx *ctx, const float *cand_xyz_dev, int n_cand,
                         const int32_t *triplets_host, int K, float *thr_inout_host,
                         float *models_out_host, int32_t *n_inliers_host,
                         double *sse_host, double *sy_host, double *syy_host,
                         void *stream);
/* Least-squares refit of z ~ x,y over the inliers of `model` (float64
 * normal equations, centred); out_model [host] (3) float64. Blocking.       */
int modest_ransac_refit(modest_ctx *ctx, const float *cand_xyz_dev, int n_cand,
                        const float *model_host, float thr,
                        double *out_model_host, int32_t *n_inliers_host,
                        void *stream);

/* ---- a10+a11 above_plane & range mask (pointcloud_utils.py:68-81,
 * generate_mask.py:57-65).  keep[i] = !(dist<offset && in only_range) &&
 * (lx0 < x <= lx1) && (ly0 < y <= ly1); dist = (p . n + d)/|n| in float64.
 * Compacts kept points (input order) to kept_xyz/kept_idx; mask is uint8.   */
int modest_plane_range_mask(modest_ctx *ctx, const float *pts_dev, int n,
                            int stride, const double *plane4_host,
                            double offset, const double *only_range4_host,
                            const double *limit_range4_host,
                            uint8_t *mask_dev, float *kept_xyz_dev,
                            int32_t *kept_idx_dev, int32_t *n_kept_dev,
                            void *stream);

/* ---- a12+a13 precompute_affinity_matrix('radius_mutual_knn','l1') +
 * DBSCAN(metric='precomputed') (utils/clustering_utils.py:7-60,
 * generate_mask.py:75-81), evaluated on the implicit graph:
 *   edge(i,j) <=> d2(i,j) <= min(r2_k(i), r2_k(j), radius^2)   (float64 d2)
 *                 and (double)(float)|pp_i - pp_j| <= eps
 *   core(i)   <=> deg(i) + 1 >= min_samples
 *   label     =  rank of the component's smallest core index; border point
 *                -> smallest label among adjacent cores; noise -> -1.
 * xyz [dev] (n,3) f32, pp [dev] (n) f32, labels [dev] (n) int32.
 * kth_d2 [dev] (n) float64 optional output (squared distance to the k-th
 * neighbour, +inf when fewer than k neighbours lie within radius).          */
int modest_cluster_dbscan(modest_ctx *ctx, const float *xyz_dev,
                          const float *pp_dev, int n, int k_neighbors,
                          double radius, double eps, int min_samples,
                          int32_t *labels_dev, double *kth_d2_dev,
                          int32_t *n_clusters_host, void *stream);

/* Non-default graph / weight branches of precompute_affinity_matrix (SURVEY §8f-3;
 * utils/clustering_utils.py:16-56), same DBSCAN semantics as above:
 *   neighbor_type  RADIUS_MUTUAL_KNN (configs default) | RADIUS (edge <=> d2 <= radius^2)
 *   affinity_type  L1 |pp_i - pp_j| (default) | EXP exp((pp_i - pp_j)^2) | L2_4D the reference's
 *                  '3d_l2_distance': float32 norm of the difference of the (n,4) scan rows,
 *                  i.e. xyz AND intensity (intensity_dev (n) float32 required)
 *   KNN (directed: the rows of kneighbors_graph), SYM_KNN (graph + graph.T), MUTUAL_KNN
 *   (graph .* graph.T): exact k-th neighbour distances without a radius bound (the grid search
 *   grows until the k-th distance is certified); `radius` only sizes the grid cells there.
 *   For KNN sklearn's sequential DBSCAN on the directed graph is reproduced exactly:
 *   label(v) = rank of the smallest-index core point that reaches v through core points.     */
#define MODEST_GRAPH_RADIUS_MUTUAL_KNN 0
#define MODEST_GRAPH_RADIUS 1
#define MODEST_GRAPH_KNN 2
#define MODEST_GRAPH_SYM_KNN 3
#define MODEST_GRAPH_MUTUAL_KNN 4
#define MODEST_AFFINITY_L1 0
#define MODEST_AFFINITY_EXP 1
#define MODEST_AFFINITY_L2_4D 2
int modest_cluster_dbscan_ex(modest_ctx *ctx, const float *xyz_dev, const float *pp_dev,
                             const float *intensity_dev, int n, int neighbor_type,
                             int affinity_type, int k_neighbors, double radius, double eps,
                             int min_samples, int32_t *labels_dev, double *kth_d2_dev,
                             int32_t *n_clusters_host, void *stream);

/* The per-scan body of generate_mask.py:57-88 as ONE call: above_plane + limit_range mask of the
 * scan rows (modest_plane_range_mask), the affinity graph and DBSCAN of the kept rows
 * (modest_cluster_dbscan_ex) and `labels[ptc_mask] = cluster labels` on an array preset to -1.
 * The first kernel masks, compacts, presets the labels and counts the kept rows per cell of a grid
 * fixed around limit_range (no bounding-box pass, no separate gathers of pp / intensity, no label
 * scatter launch); one stream sync inside (the kept count sizes the launches that follow).
 * pts [dev] (n,stride) f32 scan rows, pp [dev] (n) f32, labels [dev] (n) int32: -1 = masked out or
 * noise, else the DBSCAN label.  n_kept_host / n_clusters_host: host words.  When the graph needs
 * k neighbours and only n_kept <= k rows are kept the call fails with MODEST_ERR_ARG and the
 * message sklearn's kneighbors raises (n_kept_host is valid).                                 */
int modest_mask_cluster(modest_ctx *ctx, const float *pts_dev, int n, int stride, const float *pp_dev,
                        const double *plane4, double offset, const double *only_range4,
                        const double *limit_range4, int neighbor_type, int affinity_type,
                        int k_neighbors, double radius, double eps, int min_samples,
                        int32_t *labels_dev, int32_t *n_kept_host, int32_t *n_clusters_host,
                        void *stream);

/* ---- the mask stage of one scan behind one call ------------------------------------------
 * generate_mask.py:52-88 + clustering_utils.py:119-135: estimate_plane (RANSAC) for the mask and for
 * filter_labels, above_plane + limit_range mask, affinity graph + DBSCAN, labels[ptc_mask] = ...,
 * is_valid_cluster on every cluster, relabelling to 0 = background / 1..C -- i.e. everything of
 * generate_mask_scan up to `labels_filtered`, with no interpreter between the device round trips.
 * The pieces are the entry points above (modest_plane_prepare, modest_ransac_plane x 2,
 * modest_mask_cluster, modest_cluster_stats); the scalar glue (plane_from_linear_model, the four
 * comparisons per cluster incl. numpy's float32 percentile interpolation, the relabel table) is
 * restated in the library with numpy's rounding.
 * mt_key624 / mt_pos: numpy RandomState (MT19937) words, advanced by the executed trials of both fits
 * (pass copies and commit them when info_out[3] == 0).  labels_out [host] (n) int64 = the
 * `labels_filtered` array; plane1_out / plane2_out [host] (4) f64.
 * info_out[8]: {kept rows, DBSCAN clusters, largest final label, status, candidates of fit 1, of fit 2,
 * trials of fit 1, of fit 2}.  status != 0: nothing was committed, take the host path --
 * SMALL_SET (a candidate set of <= 300 points: sklearn samples those with another method),
 * NO_CONSENSUS, DEGENERATE (consensus set without a unique plane), TOO_FEW_KEPT (sklearn's
 * kneighbors raises).                                                                          */
typedef struct modest_mask_params {
    float max_hs1, range1[4];        /* plane_estimate.max_hs, .range {xlo, xhi, ylo, yhi}           */
    float max_hs2, range2[4];        /* filter_labels' hard-coded second fit (clustering_utils.py:126) */
    double offset;                   /* plane_estimate.offset                                        */
    int32_t use_only_range;
    double only_range[4];            /* above_plane only_range                                       */
    double limit_range[4];
    int32_t neighbor_type, affinity_type, k_neighbors, min_samples;
    double radius, eps;
    int32_t min_points;              /* filtering.*                                                  */
    double max_min_height, min_max_height;
    double quantile;                 /* np.true_divide(percentile, np.float32(100)) as a double      */
    float min_percentile_pp_score;
    int32_t max_trials, batch;       /* RANSACRegressor defaults: 100; trials per device round trip  */
    double stop_probability;
} modest_mask_params;
#define MODEST_STAGE_SMALL_SET 1
#define MODEST_STAGE_NO_CONSENSUS 2
#define MODEST_STAGE_DEGENERATE 3
#define MODEST_STAGE_TOO_FEW_KEPT 4
#define MODEST_STAGE_HOST_RULE 5   /* a RANSAC trial bound on a rounding boundary: the host loop (host libm) decides */
int modest_mask_stage(modest_ctx *ctx, const float *pts_dev, int n, int stride, const float *pp_dev,
                      const modest_mask_params *params, uint32_t *mt_key624, int32_t *mt_pos,
                      double *plane1_out, double *plane2_out, int64_t *labels_out, int32_t *info_out,
                      void *stream);

/* The same stage for a CHAIN of scans (the reference's scan loop, generate_mask.py:52, has no dependency between
 * iterations): the ground fits stay per scan, from the mask kernel on the scans advance together -- one launch per
 * kernel of the mask / graph / DBSCAN block and of the cluster statistics for the whole chain (the scan is a second
 * grid dimension), three round trips per chain instead of three per scan.  Every scan works in its OWN context
 * (scratch, pinned words, persistent counters); arguments per scan as in modest_mask_stage.  Results are those of
 * separate calls, bit for bit; configurations the chain does not cover (k-NN graphs without a radius,
 * 3d_l2_distance) and chains of one scan take the separate calls.  Blocking.
 * The trial loops of the two fits run ON THE DEVICE for max_trials <= 128 (csrc/plane.hip: rsd_*; numpy's MT19937 as
 * sklearn's sample_without_replacement consumes it, the sequential accept rule, _dynamic_max_trials, refit and plane,
 * the generator advanced by the executed trials): one synchronise for both fits + the mask kernel.  A scan whose trial
 * bound sits on a rounding boundary of log / pow comes back with info_out[3] = MODEST_STAGE_HOST_RULE and an untouched
 * generator, like the other hand-back statuses.  MODEST_RANSAC_HOST=1 in the environment keeps the loops on the host
 * (batches of params->batch trials per round trip).                                                                */
typedef struct modest_mask_stage_scan {
    modest_ctx *ctx;
    const float *pts_dev;
    int32_t n, stride;
    const float *pp_dev;
    uint32_t *mt_key624;
    int32_t *mt_pos;
    double *plane1_out, *plane2_out;
    int64_t *labels_out;
    int32_t *info_out;
    /* optional (NULL: not wanted): the indices, ascending, of the points whose labels_out is > 0 (capacity n) and their number --
     * what the box tail (modest_boxes_scan::members) walks instead of all n points                                            */
    int32_t *members_out;
    int32_t *n_members_out;
} modest_mask_stage_scan;
int modest_mask_stage_batch(const modest_mask_stage_scan *scans, int n_scans, const modest_mask_params *params,
                            void *stream);

/* ---- a14 filter_labels / is_valid_cluster statistics ------------------
 * (utils/clustering_utils.py:94-135).  For each label c in [0, n_clusters):
 * out[c*6 + {0: member count, 1: min, 2: max signed distance to `plane`
 * ((p . n + d)/|n|, float64), 3: a, 4: b, 5: gamma}] where a, b are the two
 * order statistics of the members' PP scores that numpy.percentile(pp,
 * 100*quantile) interpolates between (method 'linear': virtual index
 * (n-1)*quantile = floor + gamma).  labels [dev] int32, -1 = noise.
 * out [host] (n_clusters,6) float64.  Blocking.                              */
int modest_cluster_stats(modest_ctx *ctx, const float *pts_dev, int n, int stride,
                         const float *pp_dev, const int32_t *labels_dev,
                         int n_clusters, const double *plane4_host,
                         double quantile, double *out_host, void *stream);

/* ---- a15-a19, a21 the second half of a scan behind three calls (generate_mask.py:88-103,
 * gen_label_files.py:44-52) ------------------------------------------------------------------------
 * modest_scan_boxes: labels_inout [host] (n) int64 holds `labels_filtered` (0 = background, 1..n_lab, every
 * label with members) and receives the final labels (generate_mask.py:100-103).  For every cluster: its
 * rect-frame points (Calibration.project_velo_to_rect, kitti_util.py:327-329), the closeness fit
 * (pointcloud_utils.py:167-216; 901 headings on the device, rectangle_at_angle's tail on the host),
 * get_obj (:292-317) with the lowest point from the device, the volume gate (generate_mask.py:91-98).
 * pts_dev / pts_host: the same (n,stride) float32 scan in both memories.  angles / cossin / cossin90 [host]:
 * the caller's numpy tables (heading, (cos, sin) of it, (cos, sin) of heading + pi/2): the library takes
 * no cosine of its own, so the host's numpy defines them as in the reference.
 * objs_out [host] (n_lab,8) float64 rows {t0, t1, t2, l, w, h, ry, volume} of ALL clusters in label order,
 * keep_out [host] (n_lab) int32 = passed the volume gate.  info_out[2]: {boxes kept, status}; status != 0:
 * nothing was written, take the host statement (1: a cluster too large for the extents kernel, 2: a box
 * footprint without any scan point -- numpy raises there).  Blocking (two round trips).               */
typedef struct modest_boxes_params {
    double V2C[12], R0[9];          /* Calibration, row major 3x4 / 3x3                                  */
    const double *angles;           /* (n_angles) headings of closeness_rectangle (:170-175)             */
    const double *cossin;           /* (n_angles,2)                                                      */
    const double *cossin90;         /* (n_angles,2) of heading + pi/2                                    */
    int32_t n_angles;
    double d0;                      /* closeness criterion floor (1e-2)                                  */
    double min_volume, max_volume;  /* filtering.min_volume / max_volume                                 */
} modest_boxes_params;
int modest_scan_boxes(modest_ctx *ctx, const float *pts_dev, const float *pts_host, int n, int stride,
                      int64_t *labels_inout_host, int n_lab, const modest_boxes_params *params,
                      double *objs_out_host, int32_t *keep_out_host, int32_t *info_out_host, void *stream);
/* The same for a CHAIN of scans: the host phases per scan, ONE closeness launch over all clusters of the chain and ONE
 * lowest-point launch over all boxes (two round trips per chain instead of two per scan).  Every scan in its own
 * context; arguments per scan as above.  info_out[1] == 1 on every scan when some cluster of the chain is too large
 * for the extents kernel (go scan by scan then).  Results are those of separate calls.  Blocking.               */
typedef struct modest_boxes_scan {
    modest_ctx *ctx;
    const float *pts_dev, *pts_host;
    int32_t n, stride;
    int64_t *labels_inout;
    int32_t n_lab;
    double *objs_out;
    int32_t *keep_out, *info_out;
    /* optional (NULL: every point is looked at): the indices, ascending, of the points with labels_inout > 0, as the mask stage
     * reports them (modest_mask_stage_scan::members_out); the host passes then touch the members only                    */
    const int32_t *members;
    int32_t n_members;
} modest_boxes_scan;
int modest_scan_boxes_batch(const modest_boxes_scan *scans, int n_scans, const modest_boxes_params *params, void *stream);
/* objs_nms' boxes [t0, t2, 0, l, w, h, -ry] as float32 (pointcloud_utils.py:322-324) and their BEV IoU
 * matrix (iou3d_nms_utils.boxes_iou_bev) -> iou_out [host] (k,k) float32.  Blocking.                  */
int modest_objs_iou(modest_ctx *ctx, const double *objs8_host, int k, float *iou_out_host, void *stream);
/* ... of the box sets of a chain of scans: one launch and one round trip for all (k[s] boxes, (k[s],k[s]) out each). */
int modest_objs_iou_batch(modest_ctx *ctx, const double *const *objs8_host, const int32_t *k, int n_sets,
                          float *const *iou_out_host, void *stream);
/* ---- stages 2 + 3 of a CHAIN of scans behind one call (generate_mask.py:52-103 + the IoU matrix of gen_label_files.py:44-45,
 * pointcloud_utils.py:320-327): modest_mask_stage_batch, modest_scan_boxes_batch and modest_objs_iou_batch in sequence, results
 * leaving the library once.  Per scan (own context, as in those calls): labels_out [host] (n) int64 = the FINAL labels
 * (generate_mask.py:100-103); objs_out [host] (max_boxes,8): the first info_out[9] rows = the boxes that passed the volume gate,
 * in label order; iou_out [host]: their (k,k) float32 BEV IoU, row major, packed (NULL when nms_enable == 0);
 * members_scratch [host] (n) int32.  info_out[12]: [0..7] as modest_mask_stage; [8] clusters before the volume gate; [9] k;
 * [10] 0 = done, 1 = the mask stage handed the scan back (generator untouched, nothing written), 2 = the box tail handed it
 * back, 3 = more than max_boxes clusters -- for 2 and 3 labels_out holds `labels_filtered` and the generator is advanced: the
 * caller runs modest_scan_boxes / its host statement on them.  np.diag(iou).argsort() and the label text stay with the caller
 * (modest_label_lines).  Blocking.                                                                                          */
typedef struct modest_seed_scan {
    modest_ctx *ctx;
    const float *pts_dev, *pts_host;
    int32_t n, stride;
    const float *pp_dev;
    uint32_t *mt_key624;
    int32_t *mt_pos;
    double *plane1_out, *plane2_out;
    int64_t *labels_out;
    int32_t *members_scratch;
    double *objs_out;
    float *iou_out;
    int32_t *info_out;
} modest_seed_scan;
int modest_seed_chain(const modest_seed_scan *scans, int n_scans, const modest_mask_params *mask_params,
                      const modest_boxes_params *boxes_params, int max_boxes, int nms_enable, void *stream);
/* objs_nms' greedy walk (pointcloud_utils.py:329-343) in the caller's `order` -- the reference's
 * np.diag(iou).argsort()[::-1], a numpy call whose tie order is numpy's own --, is_within_fov (:373-379),
 * objs2label (:347-370).  cossin_ry [host] (k,2) = numpy's (cos, sin) of every obj.ry (roty, kitti_util.py:
 * 383-389).  kept_out [host] (k) int32: indices of the boxes written, text_out: the label file's text
 * (lines joined by '\n', no trailing newline, fields %.4f).  Pure host code, no context.               */
typedef struct modest_labels_params {
    double P[12];                   /* Calibration.P (P2), row major 3x4                                 */
    int32_t nms_enable;
    float nms_threshold;
    int32_t fov_only;
    double image_h, image_w;        /* image_shape = [h, w]                                              */
} modest_labels_params;
int modest_label_lines(const double *objs8_host, const double *cossin_ry_host, int k, const int64_t *order_host,
                       const float *iou_host, const modest_labels_params *params, int32_t *kept_out_host,
                       int32_t *n_kept_out_host, char *text_out_host, int32_t text_cap, int32_t *text_len_out_host);

/* ---- host-side ingest: the `.bin` files of a group of scans, back to back into one (pinned) staging buffer -------------
 * Replaces the per-frame np.fromfile of load_velo_scan (utils/pointcloud_utils.py:22-25; one call per history frame from
 * pre_compute_pp_score.py:137-146).  paths[k]: NUL-terminated; dst [host] capacity_bytes; sizes_out [host] (n_files) = every
 * file's size in bytes (file k lands at the sum of the sizes before it); n_threads host threads share the files.
 * Returns 0 = read; a positive number = the bytes the files need when that exceeds capacity_bytes (or dst is NULL): nothing was
 * read, sizes_out is filled, call again with a larger buffer; -(k + 2) = file k could not be opened or read (errno holds the
 * cause); -1 = bad arguments.  Pure host code: no context, no stream, no device.                                              */
int64_t modest_host_read_files(const char *const *paths, int n_files, void *dst, uint64_t capacity_bytes, uint64_t *sizes_out,
                               int n_threads);

/* ---- §8f-2 combine_labels.py: filter_by_ppscore (combine_labels.py:41-60) -------
 * For each detector box the reference masks the scan's rect-frame points (offsets from the box
 * centre rotated into the box frame by `ptc_xz @ rot.T`, strict half-extent tests in x and z,
 * y in (t_y - h, t_y]) and takes numpy.percentile of the masked PP scores.
 * rect_xyz [dev] (n,3) float64 (calib.project_velo_to_rect output), pp [dev] (n,) float32.
 * boxes12 [host] (n_boxes,12) float64, the scalars numpy compares against, evaluated by the
 * caller in the box fields' own dtypes: cx, cz, rot00, rot01, rot10, rot11, -l/2, l/2, -w/2,
 * w/2, t_y - h, t_y.
 * out [host] (n_boxes,4) float64: {0: points inside, 1: a, 2: b, 3: gamma} with a, b the two
 * order statistics numpy.percentile(pp[mask], 100*quantile) interpolates between ('linear' on
 * float32 data: virtual index (n-1)*quantile = floor + gamma, all in float32).  Blocking.   */
int modest_boxes_pp_stats(modest_ctx *ctx, const double *rect_xyz_dev, int n,
                          const float *pp_dev, const double *boxes12_host, int n_boxes,
                          double quantile, double *out_host, void *stream);

/* ---- a16 closeness_rectangle angle search (pointcloud_utils.py:167-187)
 * For each cluster c (points pts_xz[offsets[c]..offsets[c+1]) , float64 (x,z)
 * pairs) and each of n_angles (cos,sin) table entries: beta = sum over points
 * of 1/max(min(Dx,Dy),d0) accumulated in numpy's pairwise order; returns the
 * index of the first strict maximum per cluster and the betas.               */
int modest_fit_boxes_closeness(modest_ctx *ctx, const double *pts_xz_dev,
                               const int32_t *offsets_host, int n_clusters,
                               const double *cossin_host, int n_angles,
                               double d0, int32_t *best_angle_host,
                               double *beta_host /* optional (C,n_angles) */,
                               void *stream);

/* The same call for cluster points in HOST memory (get_obj's callers hold them there): the points
 * travel with the offset / angle / summation-order tables in one staged copy.
 * extents_host (optional, (C,8) f64) with cossin90_host = (cos, sin) of angle + pi/2 per table
 * entry: the first half of rectangle_at_angle (pointcloud_utils.py:188-216) -- min_x, max_x,
 * min_y, max_y of pts @ [[c, s], [-s, c]]^T at the chosen heading, then the same four at
 * heading + pi/2 -- computed by the block that picked the heading (dgemm rounding: one fma per
 * element); an error is returned when a cluster is too large for that kernel (> ~90 k points). */
int modest_fit_boxes_closeness_host(modest_ctx *ctx, const double *pts_xz_host,
                                    const int32_t *offsets_host, int n_clusters,
                                    const double *cossin_host, int n_angles, double d0,
                                    int32_t *best_angle_host, const double *cossin90_host,
                                    double *extents_host, void *stream);

/* fit_method = 'variance_to_edge' (utils/pointcloud_utils.py:218-275): the same
 * angle table, criterion -var(Dx[Dx<Dy]) - var(Dy[Dy<Dx]) with numpy's var (pairwise sums of the
 * subsets in index order); returns the first strict maximum per cluster and the criteria.   */
int modest_fit_boxes_variance(modest_ctx *ctx, const double *pts_xz_dev,
                              const int32_t *offsets_host, int n_clusters,
                              const double *cossin_host, int n_angles,
                              int32_t *best_angle_host, double *crit_host, void *stream);

/* fit_method = 'PCA' (utils/pointcloud_utils.py:189-206): per cluster the two principal axes of
 * the centred (x,z) points with sklearn's sign convention (svd_flip, u_based_decision=False) and
 * the extent of the points along them.  out8 [host] (n_clusters,8) float64:
 * components row-major (4), min0, max0, min1, max1.  Agrees with sklearn to ~1e-13 relative
 * (LAPACK's SVD and the closed form differ in the last bits).                                */
int modest_fit_boxes_pca(modest_ctx *ctx, const double *pts_xz_dev, const int32_t *offsets_host,
                         int n_clusters, double *out8_host, void *stream);

/* ---- a17 get_lowest_point_rect (pointcloud_utils.py:278-290) ----------
 * For each box b = (cx, cz, l, w, cos(ry), sin(ry)) float64 [host] the max
 * rect-y over all scan points strictly inside the rotated footprint
 * ([dx dz] @ [[c,-s],[s,c]]^T as an FMA chain); -inf when no point is
 * inside (numpy raises there).  cos/sin come from the host so that the
 * host's libm, not the device's, defines them.  Blocking.                   */
int modest_lowest_point(modest_ctx *ctx, const double *pts_rect_dev, int n,
                        const double *boxes6_host, int n_boxes,
                        double *bottom_host, void *stream);

/* ---- a20 iou3d_nms_cuda (utils/iou3d_nms/src/iou3d_nms.h:9-12,
 * iou3d_cpu.h:9, bound at src/iou3d_nms_api.cpp:11-17) -------------------
 * boxes: (n,7) f32 [x,y,z,dx,dy,dz,heading], out: (n_a,n_b) f32 row-major. */
int modest_boxes_overlap_bev(const float *boxes_a_dev, int n_a,
                             const float *boxes_b_dev, int n_b,
                             float *out_dev, void *stream);
int modest_boxes_iou_bev(const float *boxes_a_dev, int n_a,
                         const float *boxes_b_dev, int n_b, float *out_dev,
                         void *stream);
/* nms_gpu / nms_normal_gpu: boxes sorted by score [dev]; keep [host] int64
 * (n); returns the number kept in *num_keep_host.  Blocking (the reference
 * also blocks on a cudaMemcpy, src/iou3d_nms.cpp:111-112).                  */
int modest_nms_bev(modest_ctx *ctx, const float *boxes_dev, int n,
                   float thresh, int64_t *keep_host, int *num_keep_host,
                   void *stream);
int modest_nms_normal(modest_ctx *ctx, const float *boxes_dev, int n,
                      float thresh, int64_t *keep_host, int *num_keep_host,
                      void *stream);
/* boxes_iou_bev_cpu (src/iou3d_cpu.cpp:232-252): host pointers in and out.
 * Runs the same kernel on the boxes staged in the context's pinned block (the
 * kernel reads them and writes the matrix there directly; there is no CPU
 * arithmetic path in this library).  Blocking.                              */
int modest_boxes_iou_bev_host(modest_ctx *ctx, const float *boxes_a_host,
                              int n_a, const float *boxes_b_host, int n_b,
                              float *out_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MODEST_HIP_H */
