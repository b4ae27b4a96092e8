// Frame store: modest_frame_sort, and the frame entry point of the neighbour count.
//
// Reference steps replaced: pre_compute_pp_score.py:132-150 (per-frame transform_points +
// remove_center + np.concatenate), :188-190 (cKDTree per traversal), :54-60 (count_neighbors).
//
// Every raw frame is sorted ONCE, when it enters the frame store, by the 8x8-cell tile of a WORLD
// lattice that all frames share (a frame serves ~70 scans), and keeps a prefix table tile -> [start,
// end).  modest_pp_score_frames reads frames through a descriptor table: the V3 streaming kernels
// (pp_count.hip) apply the pose on the fly, no stacked history is built.  Tile-sorted input is what
// keeps their scatter pass near one list per wavefront.
//
// Two one-pass designs that gather a tile's runs instead of scattering survivors were built on this
// layout and measured slower than the streaming kernels (round 2: wave-autonomous gather-join,
// 0.54 ms; round 3: workgroup tile join over 32x32-cell tiles, 0.47-0.62 ms against 0.33 ms); their
// numbers are in DESIGN.md section 4.1 and profiles/r03_pp_tilejoin_experiment.md.
#include "pp_frames.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace modest;

namespace {

constexpr int F_TS = 8;                  // tile edge in cells
constexpr int F_NTF = MODEST_FRAME_NTF;  // tiles per table axis (128: +-153 m at r = 0.3)
constexpr int F_NTILE = F_NTF * F_NTF;
constexpr int F_MAXT = 64;               // traversals of the V3 join (one lane each in its segmented popcount)
static_assert(F_TS == 8, "frame_bin shifts cell coordinates by 3");

struct Map24 {
    double a[8];   // rows x, y of a 3x4 map into lattice CELL coordinates
};

// ---- frame store: sort a raw frame by world tile ---------------------------------------------
struct SortJob {
    const float *raw;
    int n, stride, TX0, TY0;
    Map24 W;   // raw frame -> lattice cells
    float *xyz;
    unsigned *perm;
    unsigned *tab;
    int *n_inside;   // one word per job, contiguous: read back with a single copy
};

__device__ __forceinline__ int frame_bin(const Map24 &W, float x, float y, float z, int TX0, int TY0) {
    const double lx = fma(W.a[2], (double)z, fma(W.a[1], (double)y, W.a[0] * (double)x)) + W.a[3];
    const double ly = fma(W.a[6], (double)z, fma(W.a[5], (double)y, W.a[4] * (double)x)) + W.a[7];
    if (!(fabs(lx) < 1.0e9) || !(fabs(ly) < 1.0e9)) return F_NTILE;   // NaN / inf / absurd: outlier bin
    const long long cx = (long long)floor(lx), cy = (long long)floor(ly);
    const long long tx = (cx >> 3) - TX0, ty = (cy >> 3) - TY0;
    if (tx < 0 || tx >= F_NTF || ty < 0 || ty >= F_NTF) return F_NTILE;
    return (int)(ty * F_NTF + tx);
}

// One workgroup per frame: LDS histogram over the tiles, scan, scatter.  The order inside a tile
// is the arrival order of the atomics (irrelevant: counts are order independent, `perm` maps back).
constexpr int SORT_PER = (F_NTILE + 1 + 1023) / 1024;
__global__ __launch_bounds__(1024) void frame_sort_kernel(const SortJob *__restrict__ jobs) {
    extern __shared__ unsigned hist[];   // F_NTILE + 1 bins (+ the outlier bin)
    __shared__ unsigned wsum[16];
    const SortJob J = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < F_NTILE + 1; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < J.n; i += 1024) {
        const float *p = J.raw + (size_t)i * J.stride;
        atomicAdd(&hist[frame_bin(J.W, p[0], p[1], p[2], J.TX0, J.TY0)], 1u);
    }
    __syncthreads();
    unsigned loc[SORT_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SORT_PER; ++k) {
        const int b = tid * SORT_PER + k;
        loc[k] = b <= F_NTILE ? hist[b] : 0u;
        s += loc[k];
    }
    unsigned inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned run = inc - s;
    for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < SORT_PER; ++k) {
        const int b = tid * SORT_PER + k;
        if (b <= F_NTILE) {
            hist[b] = run;
            J.tab[b] = run;   // tab[F_NTILE] = points inside the table = start of the outliers
            if (b == F_NTILE) *J.n_inside = (int)run;
        }
        run += loc[k];
    }
    __syncthreads();
    for (int i = tid; i < J.n; i += 1024) {
        const float *p = J.raw + (size_t)i * J.stride;
        const float x = p[0], y = p[1], z = p[2];
        const unsigned pos = atomicAdd(&hist[frame_bin(J.W, x, y, z, J.TX0, J.TY0)], 1u);
        J.xyz[3 * (size_t)pos] = x;
        J.xyz[3 * (size_t)pos + 1] = y;
        J.xyz[3 * (size_t)pos + 2] = z;
        J.perm[pos] = (unsigned)i;
    }
}

}  // namespace

extern "C" int modest_frame_table_tiles(void) { return F_NTF; }

namespace {
int fill_jobs(modest_ctx *ctx, const modest_frame_sort_job *jobs, int n_jobs, SortJob *hj, int *n_inside_dst) {
    for (int k = 0; k < n_jobs; ++k) {
        const modest_frame_sort_job &j = jobs[k];
        MODEST_REQUIRE(j.n >= 0 && (j.stride == 3 || j.stride == 4), "bad frame");
        MODEST_REQUIRE(j.n == 0 || (j.raw_dev && j.xyz_dev && j.perm_dev), "NULL frame buffer");
        MODEST_REQUIRE(j.tab_dev != nullptr, "NULL table");
        hj[k].raw = j.raw_dev;
        hj[k].n = j.n;
        hj[k].stride = j.stride;
        hj[k].TX0 = j.TX0;
        hj[k].TY0 = j.TY0;
        for (int q = 0; q < 8; ++q) hj[k].W.a[q] = j.W[q];
        hj[k].xyz = j.xyz_dev;
        hj[k].perm = j.perm_dev;
        hj[k].tab = j.tab_dev;
        hj[k].n_inside = n_inside_dst + k;
    }
    return MODEST_OK;
}
}  // namespace

extern "C" int modest_frame_sort(modest_ctx *ctx, const modest_frame_sort_job *jobs, int n_jobs,
                                 int32_t *n_inside_host, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_jobs >= 0, "n_jobs < 0");
    if (n_jobs == 0) return MODEST_OK;
    MODEST_REQUIRE(jobs != nullptr, "jobs is NULL");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    static_assert(sizeof(SortJob) % 8 == 0, "job layout");
    int rc = modest_ctx_reserve(ctx, arena_sz((size_t)n_jobs * sizeof(SortJob)) + arena_sz((size_t)n_jobs * 4));
    if (rc) return rc;
    rc = modest_ctx_reserve_pinned(ctx, (size_t)n_jobs * (sizeof(SortJob) + 4));
    if (rc) return rc;
    // the pinned staging area is reused by the next call: this entry point is blocking
    SortJob *hj = reinterpret_cast<SortJob *>(ctx->pinned);
    int32_t *hin = reinterpret_cast<int32_t *>(ctx->pinned + (size_t)n_jobs * sizeof(SortJob));
    rc = fill_jobs(ctx, jobs, n_jobs, hj, reinterpret_cast<int *>(ctx->scratch + arena_sz((size_t)n_jobs * sizeof(SortJob))));
    if (rc) return rc;
    SortJob *dj = reinterpret_cast<SortJob *>(ctx->scratch);
    MODEST_HIP_CHECK(hipMemcpyAsync(dj, hj, (size_t)n_jobs * sizeof(SortJob), hipMemcpyHostToDevice, stream));
    const size_t lds = (size_t)(F_NTILE + 1) * 4;
    // per call: the attribute is per device, a process may hold contexts on several (cheap)
    MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(frame_sort_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    frame_sort_kernel<<<n_jobs, 1024, lds, stream>>>(dj);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipMemcpyAsync(hin, ctx->scratch + arena_sz((size_t)n_jobs * sizeof(SortJob)), (size_t)n_jobs * 4,
                                    hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    if (n_inside_host)
        for (int k = 0; k < n_jobs; ++k) n_inside_host[k] = hin[k];
    return MODEST_OK;
}

// The same sort without a synchronise (an ingest thread keeps several batches in flight behind the compute
// stream's kernels): the job table travels in the launch's own device buffer `jobs_scratch_dev`
// (n_jobs * MODEST_FRAME_SORT_JOB_BYTES, caller-owned, alive until the launch has run) through one of the
// context's pinned staging slots, and every job's count of points inside its table is written by the kernel
// straight into `n_inside_pinned` (n_jobs int32 of PINNED host memory, readable once the stream has passed
// the launch).
extern "C" int modest_frame_sort_async(modest_ctx *ctx, const modest_frame_sort_job *jobs, int n_jobs,
                                       void *jobs_scratch_dev, int32_t *n_inside_pinned, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_jobs >= 0, "n_jobs < 0");
    if (n_jobs == 0) return MODEST_OK;
    MODEST_REQUIRE(jobs && jobs_scratch_dev && n_inside_pinned, "NULL argument");
    static_assert(sizeof(SortJob) <= MODEST_FRAME_SORT_JOB_BYTES, "header and kernel disagree on the job size");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    SortJob *hj = nullptr;
    int rc = modest_ctx_stage_slot(ctx, (size_t)n_jobs * sizeof(SortJob), reinterpret_cast<void **>(&hj));
    if (rc) return rc;
    rc = fill_jobs(ctx, jobs, n_jobs, hj, reinterpret_cast<int *>(n_inside_pinned));
    if (rc) return rc;
    SortJob *dj = reinterpret_cast<SortJob *>(jobs_scratch_dev);
    MODEST_HIP_CHECK(hipMemcpyAsync(dj, hj, (size_t)n_jobs * sizeof(SortJob), hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx, stream);
    if (rc) return rc;
    const size_t lds = (size_t)(F_NTILE + 1) * 4;
    MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(frame_sort_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    frame_sort_kernel<<<n_jobs, 1024, lds, stream>>>(dj);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_pp_score_frames(modest_ctx *ctx, const modest_pp_frame *live, const uint32_t *live_perm_dev,
                                      const modest_pp_frame *frames, int n_frames, int n_trav, const double *A8,
                                      double radius, int32_t *counts_dev, float *H_dev, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && live != nullptr && A8 != nullptr, "NULL argument");
    MODEST_REQUIRE(n_frames >= 0 && (n_frames == 0 || frames != nullptr), "bad frame list");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= F_MAXT, "1 <= n_trav <= 64 on the frame path");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    MODEST_REQUIRE(live->n >= 0 && live->n < (1 << 24), "live scan too large");
    const int N = live->n;
    if (N == 0) return MODEST_OK;
    MODEST_REQUIRE(live->xyz_dev && live->tab_dev && live_perm_dev, "NULL live buffer");
    MODEST_REQUIRE(counts_dev != nullptr || H_dev != nullptr, "no output requested");
    for (int f = 0; f < n_frames; ++f) {
        MODEST_REQUIRE(frames[f].trav >= 0 && frames[f].trav < n_trav, "frame traversal out of range");
        MODEST_REQUIRE(frames[f].n >= 0 && frames[f].tab_dev != nullptr, "bad frame");
        MODEST_REQUIRE(frames[f].n == 0 || frames[f].xyz_dev != nullptr, "NULL frame points");
    }
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    // The V3 streaming kernels read the frames through the descriptor table with the pose fused
    // (pp_count.hip); A8 (the lattice map of the scan) is part of the ABI for callers that check the store's
    // lattice against the poses, the kernels do not need it.
    return modest_pp3_frames(ctx, live, live_perm_dev, frames, n_frames, n_trav, radius, counts_dev, H_dev, stream);
}

extern "C" int modest_pp_score_frames_batch(modest_ctx *ctx, int n_scans, const modest_pp_frame *const *live,
                                            const uint32_t *const *live_perm_dev,
                                            const modest_pp_frame *const *frames, const int32_t *n_frames,
                                            int n_trav, double radius, int32_t *const *counts_dev,
                                            float *const *H_dev, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && live != nullptr && live_perm_dev != nullptr && frames != nullptr && n_frames != nullptr,
                   "NULL argument");
    MODEST_REQUIRE(n_scans >= 1 && n_scans <= 64, "1 <= n_scans <= 64");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= F_MAXT, "1 <= n_trav <= 64 on the frame path");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    MODEST_REQUIRE(counts_dev != nullptr || H_dev != nullptr, "no output requested");
    for (int s = 0; s < n_scans; ++s) {
        MODEST_REQUIRE(live[s] != nullptr && live[s]->n >= 0 && live[s]->n < (1 << 24), "bad live scan");
        MODEST_REQUIRE(live[s]->n == 0 || (live[s]->xyz_dev && live[s]->tab_dev && live_perm_dev[s]), "NULL live buffer");
        MODEST_REQUIRE(n_frames[s] >= 0 && (n_frames[s] == 0 || frames[s] != nullptr), "bad frame list");
        MODEST_REQUIRE((counts_dev && counts_dev[s]) || (H_dev && H_dev[s]) || live[s]->n == 0, "a scan without an output");
        for (int f = 0; f < n_frames[s]; ++f) {
            MODEST_REQUIRE(frames[s][f].trav >= 0 && frames[s][f].trav < n_trav, "frame traversal out of range");
            MODEST_REQUIRE(frames[s][f].n >= 0 && frames[s][f].tab_dev != nullptr, "bad frame");
            MODEST_REQUIRE(frames[s][f].n == 0 || frames[s][f].xyz_dev != nullptr, "NULL frame points");
        }
    }
    return modest_pp3_frames_batch(ctx, n_scans, live, live_perm_dev, frames, n_frames, n_trav, radius, counts_dev, H_dev,
                                   as_stream(stream_));
}
