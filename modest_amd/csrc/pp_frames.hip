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
    long long at;    // first point of the job in the batch's scratch arrays (rank / tile of every point)
};
static_assert(sizeof(SortJob) == MODEST_FRAME_SORT_JOB_BYTES, "job layout (callers size their job buffers with the header's constant)");

__device__ __forceinline__ int frame_bin(const Map24 &W, float x, float y, float z, int TX0, int TY0) {
    const double lx = fma(W.a[2], (double)z, fma(W.a[1], (double)y, W.a[0] * (double)x)) + W.a[3];
    const double ly = fma(W.a[6], (double)z, fma(W.a[5], (double)y, W.a[4] * (double)x)) + W.a[7];
    if (!(fabs(lx) < 1.0e9) || !(fabs(ly) < 1.0e9)) return F_NTILE;   // NaN / inf / absurd: outlier bin
    const long long cx = (long long)floor(lx), cy = (long long)floor(ly);
    const long long tx = (cx >> 3) - TX0, ty = (cy >> 3) - TY0;
    if (tx < 0 || tx >= F_NTF || ty < 0 || ty >= F_NTF) return F_NTILE;
    return (int)(ty * F_NTF + tx);
}

// The sort of a batch of frames is four small launches over a GRID of point chunks (round 5; one workgroup per frame took
// 235-485 us for a scan's 11 new frames: eleven workgroups on 256 CUs): clear the tables | every point adds itself to its
// tile's counter in the frame's table (global atomic, returning its rank in the tile) | per frame: exclusive prefix over the
// 16 385 counters, in place | every point goes to table[tile] + rank.  The order inside a tile is the arrival order of
// the atomics (irrelevant: counts are order independent, `perm` maps back).
constexpr int SORT_CH = 2048;   // points per workgroup of the two point passes
__global__ __launch_bounds__(256) void frame_clear_kernel(const SortJob *__restrict__ jobs) {
    const SortJob &J = jobs[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b <= F_NTILE) J.tab[b] = 0u;
}
__global__ __launch_bounds__(256) void frame_rank_kernel(const SortJob *__restrict__ jobs, const uint2 *__restrict__ chunks,
                                                         unsigned *__restrict__ rankAll, unsigned short *__restrict__ binAll) {
    const uint2 ck = chunks[blockIdx.x];   // (job, first point)
    const SortJob &J = jobs[ck.x];
    unsigned *rank = rankAll + J.at;         // rank of point i inside its tile
    unsigned short *binOf = binAll + J.at;   // tile of point i (F_NTILE: outside the table)
    float x[SORT_CH / 256], y[SORT_CH / 256], z[SORT_CH / 256];
#pragma unroll
    for (int u = 0; u < SORT_CH / 256; ++u) {   // (every load issued before the first use)
        const int i = (int)ck.y + u * 256 + threadIdx.x;
        const float *p = J.raw + (size_t)min(i, J.n - 1) * J.stride;
        x[u] = p[0], y[u] = p[1], z[u] = p[2];
    }
#pragma unroll
    for (int u = 0; u < SORT_CH / 256; ++u) {
        const int i = (int)ck.y + u * 256 + threadIdx.x;
        if (i < J.n) {
            const int bin = frame_bin(J.W, x[u], y[u], z[u], J.TX0, J.TY0);
            binOf[i] = (unsigned short)bin;
            rank[i] = atomicAdd(&J.tab[bin], 1u);
        }
    }
}
constexpr int SORT_PER = (F_NTILE + 1 + 1023) / 1024;
__global__ __launch_bounds__(1024) void frame_scan_kernel(const SortJob *__restrict__ jobs) {
    __shared__ unsigned wsum[16];
    const SortJob &J = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned loc[SORT_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SORT_PER; ++k) {
        const int b = tid * SORT_PER + k;
        loc[k] = b <= F_NTILE ? J.tab[b] : 0u;
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
            J.tab[b] = run;   // tab[F_NTILE] = points inside the table = start of the outliers
            if (b == F_NTILE) *J.n_inside = (int)run;
        }
        run += loc[k];
    }
}
__global__ __launch_bounds__(256) void frame_place_kernel(const SortJob *__restrict__ jobs, const uint2 *__restrict__ chunks,
                                                          const unsigned *__restrict__ rankAll, const unsigned short *__restrict__ binAll) {
    const uint2 ck = chunks[blockIdx.x];
    const SortJob &J = jobs[ck.x];
    const unsigned *rank = rankAll + J.at;
    const unsigned short *binOf = binAll + J.at;
#pragma unroll
    for (int u = 0; u < SORT_CH / 256; ++u) {
        const int i = (int)ck.y + u * 256 + threadIdx.x;
        const int ii = min(i, J.n - 1);
        const float *p = J.raw + (size_t)ii * J.stride;
        const float x = p[0], y = p[1], z = p[2];
        const unsigned pos = J.tab[binOf[ii]] + rank[ii];
        if (i < J.n) {
            J.xyz[3 * (size_t)pos] = x;
            J.xyz[3 * (size_t)pos + 1] = y;
            J.xyz[3 * (size_t)pos + 2] = z;
            J.perm[pos] = (unsigned)i;
        }
    }
}

// Large batches (a cold scan: 361 frames) keep the one-workgroup-per-frame form -- LDS histogram over the tiles, scan,
// scatter: with more frames than CUs its LDS atomics beat the grid form's global ones (372 against 850 us for 361 frames).
__global__ __launch_bounds__(1024) void frame_sort_wg_kernel(const SortJob *__restrict__ jobs) {
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
// scratch of a batch behind `head` bytes of the context's arena: [rank u32 x points][bin u16 x points][chunk table]
struct SortPlan {
    size_t oRank, oBin, oChunks, bytes;
    long long points, chunks;
};
SortPlan sort_plan(const modest_frame_sort_job *jobs, int n_jobs, size_t head) {
    SortPlan p;
    p.points = 0, p.chunks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        p.points += jobs[k].n;
        p.chunks += (jobs[k].n + SORT_CH - 1) / SORT_CH;
    }
    p.oRank = arena_sz(head);
    p.oBin = p.oRank + arena_sz((size_t)std::max<long long>(p.points, 1) * 4);
    p.oChunks = p.oBin + arena_sz((size_t)std::max<long long>(p.points, 1) * 2);
    p.bytes = p.oChunks + arena_sz((size_t)std::max<long long>(p.chunks, 1) * sizeof(uint2));
    return p;
}
// fills the jobs' scratch pointers and the chunk table (host side, staged with the job table)
void sort_assign(const modest_frame_sort_job *jobs, int n_jobs, SortJob *hj, uint2 *hc) {
    long long at = 0, c = 0;
    for (int k = 0; k < n_jobs; ++k) {
        hj[k].at = at;
        for (int p0 = 0; p0 < jobs[k].n; p0 += SORT_CH) hc[c++] = make_uint2((unsigned)k, (unsigned)p0);
        at += jobs[k].n;
    }
}
constexpr int SORT_WG_FROM = 96;   // jobs per batch from which the one-workgroup-per-frame kernel is used
int sort_launch(const SortJob *dj, const uint2 *dc, int n_jobs, long long chunks, char *arena, const SortPlan &p, hipStream_t stream) {
    if (n_jobs >= SORT_WG_FROM) {
        const size_t lds = (size_t)(F_NTILE + 1) * 4;
        // per call: the attribute is per device, a process may hold contexts on several (cheap)
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(frame_sort_wg_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        frame_sort_wg_kernel<<<n_jobs, 1024, lds, stream>>>(dj);
        MODEST_HIP_CHECK(hipGetLastError());
        return MODEST_OK;
    }
    unsigned *rank = reinterpret_cast<unsigned *>(arena + p.oRank);
    unsigned short *bin = reinterpret_cast<unsigned short *>(arena + p.oBin);
    frame_clear_kernel<<<dim3((F_NTILE + 1 + 255) / 256, (unsigned)n_jobs), 256, 0, stream>>>(dj);
    if (chunks > 0) frame_rank_kernel<<<(unsigned)chunks, 256, 0, stream>>>(dj, dc, rank, bin);
    frame_scan_kernel<<<(unsigned)n_jobs, 1024, 0, stream>>>(dj);
    if (chunks > 0) frame_place_kernel<<<(unsigned)chunks, 256, 0, stream>>>(dj, dc, rank, bin);
    MODEST_HIP_CHECK(hipGetLastError());
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
    const size_t head = arena_sz((size_t)n_jobs * sizeof(SortJob)) + arena_sz((size_t)n_jobs * 4);
    const SortPlan sp = sort_plan(jobs, n_jobs, head);
    MODEST_REQUIRE(sp.points < (1LL << 31), "too many points in one batch");
    int rc = modest_ctx_reserve(ctx, sp.bytes);
    if (rc) return rc;
    const size_t stageB = (size_t)n_jobs * sizeof(SortJob) + (size_t)std::max<long long>(sp.chunks, 1) * sizeof(uint2);
    rc = modest_ctx_reserve_pinned(ctx, stageB + (size_t)n_jobs * 4);
    if (rc) return rc;
    // the pinned staging area is reused by the next call: this entry point is blocking
    SortJob *hj = reinterpret_cast<SortJob *>(ctx->pinned);
    uint2 *hc = reinterpret_cast<uint2 *>(ctx->pinned + (size_t)n_jobs * sizeof(SortJob));
    int32_t *hin = reinterpret_cast<int32_t *>(ctx->pinned + stageB);
    rc = fill_jobs(ctx, jobs, n_jobs, hj, reinterpret_cast<int *>(ctx->scratch + arena_sz((size_t)n_jobs * sizeof(SortJob))));
    if (rc) return rc;
    sort_assign(jobs, n_jobs, hj, hc);
    SortJob *dj = reinterpret_cast<SortJob *>(ctx->scratch);
    uint2 *dc = reinterpret_cast<uint2 *>(ctx->scratch + sp.oChunks);
    MODEST_HIP_CHECK(hipMemcpyAsync(dj, hj, (size_t)n_jobs * sizeof(SortJob), hipMemcpyHostToDevice, stream));
    if (sp.chunks > 0) MODEST_HIP_CHECK(hipMemcpyAsync(dc, hc, (size_t)sp.chunks * sizeof(uint2), hipMemcpyHostToDevice, stream));
    rc = sort_launch(dj, dc, n_jobs, sp.chunks, ctx->scratch, sp, stream);
    if (rc) return rc;
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
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const SortPlan sp = sort_plan(jobs, n_jobs, 0);
    MODEST_REQUIRE(sp.points < (1LL << 31), "too many points in one batch");
    int rc = modest_ctx_reserve(ctx, sp.bytes);   // (rank / tile of every point + the chunk table: the context's arena, in stream order)
    if (rc) return rc;
    char *hs = nullptr;
    const size_t jobB = (size_t)n_jobs * sizeof(SortJob), stageB = jobB + (size_t)std::max<long long>(sp.chunks, 1) * sizeof(uint2);
    rc = modest_ctx_stage_slot(ctx, stageB, reinterpret_cast<void **>(&hs));
    if (rc) return rc;
    SortJob *hj = reinterpret_cast<SortJob *>(hs);
    uint2 *hc = reinterpret_cast<uint2 *>(hs + jobB);
    rc = fill_jobs(ctx, jobs, n_jobs, hj, reinterpret_cast<int *>(n_inside_pinned));
    if (rc) return rc;
    sort_assign(jobs, n_jobs, hj, hc);
    SortJob *dj = reinterpret_cast<SortJob *>(jobs_scratch_dev);
    uint2 *dc = reinterpret_cast<uint2 *>(ctx->scratch + sp.oChunks);
    MODEST_HIP_CHECK(hipMemcpyAsync(dj, hj, jobB, hipMemcpyHostToDevice, stream));
    if (sp.chunks > 0) MODEST_HIP_CHECK(hipMemcpyAsync(dc, hc, (size_t)sp.chunks * sizeof(uint2), hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx, stream);
    if (rc) return rc;
    return sort_launch(dj, dc, n_jobs, sp.chunks, ctx->scratch, sp, stream);
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

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_pp_frames(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(frame_clear_kernel));
}
