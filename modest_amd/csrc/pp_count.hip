// PP-score neighbour counting (reference: pre_compute_pp_score.py:54-60,188-193)
// and entropy (pre_compute_pp_score.py:68-75) for gfx950.
//
// Inversion of the reference's "KD-tree over 10.8 M history points, query with
// 30 k live points": the small live scan is indexed (cell-sorted), the history
// is streamed ONCE with coalesced 16-byte loads and rejected early against an
// LDS-resident dilated occupancy bitmap of the live scan.
//
// Measured on MI355X (Lyft shape, 10.8 M history points, 6.1 M of them survive the bitmap):
//   stream + bitmap test alone ............  25 us  (5.2 TB/s)
//   V1: survivors resolved against an L2-resident index, one global atomic per
//       pair ............................... 2090 us (630 us of divergent loads,
//                                            1430 us for 10.5 M device atomics)
//   V2 (removed; see git history and DESIGN.md): survivors binned in LDS by 32x32-cell tile,
//       resolved tile by tile against LDS-resident live points, one lane per record,
//       one LDS atomic per pair .............  ~400 us
//   V3 (pp_v3.h): records sorted by cell, wave-uniform candidate lists, ballot +
//       popcount instead of atomics ......... ~330 us
// V1 stays as the path for more than 64 traversals and for A/B runs (MODEST_PP_VARIANT=1).
#include "pp_frames.h"
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

using namespace modest;

namespace {

constexpr int PP_NX = 640;
constexpr int PP_NY = 640;
constexpr int PP_NCELL = PP_NX * PP_NY;       // 409,600 cells (192 m at r=0.3)
constexpr int PP_BITWORDS = PP_NCELL / 32;    // 12,800 words = 51,200 B of LDS
constexpr int SCAN_BLOCK = 1024;
constexpr int SCAN_NBLK = PP_NCELL / SCAN_BLOCK;   // 400
static_assert(PP_NCELL % SCAN_BLOCK == 0 && SCAN_NBLK <= 1024, "scan tiling");

// ---- live-scan index build -------------------------------------------------
// bbox words (zero-initialised): max key(x), max ~key(x), max key(y), max ~key(y)
__device__ __forceinline__ unsigned pp_fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pp_fkey_inv(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ PPGrid pp_grid(const unsigned *bb, double c) {
    const float mxx = pp_fkey_inv(bb[0]), mnx = pp_fkey_inv(~bb[1]);
    const float mxy = pp_fkey_inv(bb[2]), mny = pp_fkey_inv(~bb[3]);
    double cx = 0.5 * ((double)mnx + (double)mxx);
    double cy = 0.5 * ((double)mny + (double)mxy);
    if (!(cx == cx) || fabs(cx) > 1e30) cx = 0.0;   // NaN / inf guard
    if (!(cy == cy) || fabs(cy) > 1e30) cy = 0.0;
    PPGrid g;
    g.ox = (float)(cx - 0.5 * PP_NX * c);
    g.oy = (float)(cy - 0.5 * PP_NY * c);
    g.inv_c = (float)(1.0 / c);
    return g;
}

__global__ __launch_bounds__(256) void pp_live_bbox(const float *__restrict__ live, int n, unsigned *bb,
                                                    int *__restrict__ counts, size_t nCounts) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nCounts; i += (size_t)gridDim.x * 256) counts[i] = 0;
    unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    for (int r = 0; r < 4; ++r) {   // 1024 points per block
        const int i = blockIdx.x * 1024 + r * 256 + threadIdx.x;
        if (i < n) {
            const unsigned kx = pp_fkey(live[3 * (size_t)i]), ky = pp_fkey(live[3 * (size_t)i + 1]);
            k0 = max(k0, kx);
            k1 = max(k1, ~kx);
            k2 = max(k2, ky);
            k3 = max(k3, ~ky);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        k0 = max(k0, (unsigned)__shfl_xor((int)k0, o));
        k1 = max(k1, (unsigned)__shfl_xor((int)k1, o));
        k2 = max(k2, (unsigned)__shfl_xor((int)k2, o));
        k3 = max(k3, (unsigned)__shfl_xor((int)k3, o));
    }
    __shared__ unsigned red[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = k0;
        red[1][w] = k1;
        red[2][w] = k2;
        red[3][w] = k3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {   // one atomic per word per block (same-address atomics cost ~12 ns each)
        const unsigned m = max(max(red[threadIdx.x][0], red[threadIdx.x][1]),
                               max(red[threadIdx.x][2], red[threadIdx.x][3]));
        atomicMax(&bb[threadIdx.x], m);
    }
}

// bbPart != NULL (frame path): the bounding box arrives as per-block partial maxima of the key words
// (pp3_live_prep); every block combines them, block 0 publishes bb[] for the kernels that follow.
__device__ __forceinline__ void pp_live_count_body(const float *__restrict__ live, int n, unsigned *bb, double c, unsigned *cellCount, const unsigned *__restrict__ bbPart, int nPart, const unsigned bx, const unsigned gx) {
    __shared__ unsigned sbb[4];
    if (bbPart) {
        if (threadIdx.x < 4) sbb[threadIdx.x] = 0u;
        __syncthreads();
        for (int k = threadIdx.x; k < 4 * nPart; k += blockDim.x) atomicMax(&sbb[k & 3], bbPart[k]);
        __syncthreads();
        if (bx == 0 && threadIdx.x < 4) bb[threadIdx.x] = sbb[threadIdx.x];
    }
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PPGrid g = pp_grid(bbPart ? sbb : bb, c);
    const int cx = pp_cell_coord(live[3 * (size_t)i], g.ox, g.inv_c, PP_NX);
    const int cy = pp_cell_coord(live[3 * (size_t)i + 1], g.oy, g.inv_c, PP_NY);
    atomicAdd(&cellCount[cy * PP_NX + cx], 1u);
}
__global__ __launch_bounds__(256) void pp_live_count(const float *__restrict__ live, int n, unsigned *bb, double c, unsigned *cellCount, const unsigned *__restrict__ bbPart, int nPart) {
    pp_live_count_body(live, n, bb, c, cellCount, bbPart, nPart, blockIdx.x, gridDim.x);
}

// Dilated occupancy bitmap: bit(cell) = any live point in the 3x3 cells around it.
// One block per grid row: coalesced reads of the three neighbouring counter rows,
// wave ballots for the row's occupancy bits, shifts for the horizontal dilation.
static_assert(PP_NX % 64 == 0 && PP_NX <= 1024, "one thread per cell of a row, whole wavefronts");
__device__ __forceinline__ void pp_bitmap_row(const unsigned *__restrict__ cellCount, unsigned *__restrict__ bitmap,
                                              int cy) {
    __shared__ unsigned long long sb[PP_NX / 64 + 1];
    const int x = threadIdx.x;
    unsigned occ = 0;
    if (x < PP_NX) {
        occ = cellCount[(size_t)cy * PP_NX + x];
        if (cy > 0) occ |= cellCount[(size_t)(cy - 1) * PP_NX + x];
        if (cy + 1 < PP_NY) occ |= cellCount[(size_t)(cy + 1) * PP_NX + x];
    }
    const unsigned long long ball = __ballot(occ != 0u);
    if ((x & 63) == 0 && x < PP_NX) sb[x >> 6] = ball;
    if (x == 0) sb[PP_NX / 64] = 0ULL;
    __syncthreads();
    if (x < PP_NX / 32) {   // output word x covers columns [32x, 32x+32)
        const int p = 32 * x - 1;   // window bit t = occupancy of column p + t, t = 0..33
        unsigned long long w;
        if (p < 0) {
            w = sb[0] << 1;
        } else {
            const int idx = p >> 6, sh = p & 63;
            w = sb[idx] >> sh;
            if (sh) w |= sb[idx + 1] << (64 - sh);
        }
        const unsigned long long dil = w | (w >> 1) | (w >> 2);
        bitmap[(size_t)cy * (PP_NX / 32) + x] = (unsigned)(dil & 0xffffffffULL);
    }
}

// Exclusive scan of the cell counters in two coalesced launches.
__device__ __forceinline__ void pp_scan_block(const unsigned *__restrict__ cnt, unsigned *__restrict__ start,
                                              unsigned *__restrict__ blockSum, int blk) {
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t i = (size_t)blk * SCAN_BLOCK + tid;
    const unsigned v = cnt[i];
    unsigned inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    start[i] = base + inc - v;   // block-local exclusive prefix
    if (tid == SCAN_BLOCK - 1) blockSum[blk] = base + inc;
}

// One launch for the two independent consumers of the cell counters: blocks [0, SCAN_NBLK) scan,
// blocks [SCAN_NBLK, SCAN_NBLK + PP_NY) build one row of the dilated bitmap each.
__device__ __forceinline__ void pp_scan_bitmap_body(const unsigned *__restrict__ cnt, unsigned *__restrict__ start, unsigned *__restrict__ blockSum, unsigned *__restrict__ bitmap, const unsigned bx, const unsigned gx) {
    if (bx < SCAN_NBLK) pp_scan_block(cnt, start, blockSum, (int)bx);
    else pp_bitmap_row(cnt, bitmap, (int)bx - SCAN_NBLK);
}
__global__ __launch_bounds__(SCAN_BLOCK) void pp_scan_bitmap(const unsigned *__restrict__ cnt, unsigned *__restrict__ start, unsigned *__restrict__ blockSum, unsigned *__restrict__ bitmap) {
    pp_scan_bitmap_body(cnt, start, blockSum, bitmap, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void pp_scan_finish_body(unsigned *__restrict__ start, const unsigned *__restrict__ blockSum, const unsigned bx, const unsigned gx) {
    __shared__ unsigned red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned v = (tid < (int)bx) ? blockSum[tid] : 0u;   // sums of the earlier blocks
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[w] = v;
    __syncthreads();
    unsigned off = 0;
    for (int k = 0; k < 16; ++k) off += red[k];
    const size_t i = (size_t)bx * SCAN_BLOCK + tid;
    start[i] += off;
    if (bx == SCAN_NBLK - 1 && tid == SCAN_BLOCK - 1) start[PP_NCELL] = off + blockSum[SCAN_NBLK - 1];
}
__global__ __launch_bounds__(SCAN_BLOCK) void pp_scan_finish(unsigned *__restrict__ start, const unsigned *__restrict__ blockSum) {
    pp_scan_finish_body(start, blockSum, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void pp_live_scatter_one(int i, const float *__restrict__ live, int n, const unsigned *bb,
                                                    double c, const unsigned *__restrict__ start,
                                                    unsigned *fill, float4 *__restrict__ sorted) {
    if (i >= n) return;
    const PPGrid g = pp_grid(bb, c);
    const float x = live[3 * (size_t)i], y = live[3 * (size_t)i + 1], z = live[3 * (size_t)i + 2];
    const int cell = pp_cell_coord(y, g.oy, g.inv_c, PP_NY) * PP_NX + pp_cell_coord(x, g.ox, g.inv_c, PP_NX);
    const unsigned slot = start[cell] + atomicAdd(&fill[cell], 1u);
    sorted[slot] = make_float4(x, y, z, __int_as_float(i));
}

__global__ void pp_live_scatter(const float *__restrict__ live, int n, const unsigned *bb, double c,
                                const unsigned *__restrict__ start, unsigned *fill,
                                float4 *__restrict__ sorted) {
    pp_live_scatter_one(blockIdx.x * blockDim.x + threadIdx.x, live, n, bb, c, start, fill, sorted);
}

// ---- V1 history stream (kept for A/B: MODEST_PP_VARIANT=1) ---------------------
__device__ __forceinline__ int pp_find_trav(const TravOffsets &tr, long long p) {
    int t = 0;
    while (t + 1 < tr.n && p >= tr.off[t + 1]) ++t;
    return t;
}

__global__ __launch_bounds__(256) void pp_stream_v1(const float *__restrict__ hist, long long m0,
                                                    long long m1, TravOffsets tr, const unsigned *bb,
                                                    double c, const unsigned *__restrict__ bitmap,
                                                    const unsigned *__restrict__ cellStart,
                                                    const float4 *__restrict__ sorted, int *counts,
                                                    int T, double r2) {
    __shared__ unsigned sbits[PP_BITWORDS];
    for (int i = threadIdx.x; i < PP_BITWORDS; i += 256) sbits[i] = bitmap[i];
    __syncthreads();
    const PPGrid g = pp_grid(bb, c);
    for (long long p = m0 + (long long)blockIdx.x * 256 + threadIdx.x; p < m1; p += (long long)gridDim.x * 256) {
        const float x = hist[3 * p], y = hist[3 * p + 1], z = hist[3 * p + 2];
        const int cx = pp_cell_coord(x, g.ox, g.inv_c, PP_NX), cy = pp_cell_coord(y, g.oy, g.inv_c, PP_NY);
        const int bit = cy * PP_NX + cx;
        if (!((sbits[bit >> 5] >> (bit & 31)) & 1u)) continue;
        const int t = pp_find_trav(tr, p);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, PP_NX - 1);
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, PP_NY - 1); ++yy) {
            const unsigned s = cellStart[yy * PP_NX + x0], e = cellStart[yy * PP_NX + x1 + 1];
            for (unsigned j = s; j < e; ++j) {
                const float4 q = sorted[j];
                if (pp_within(x, y, z, q.x, q.y, q.z, r2))
                    atomicAdd(&counts[(size_t)__float_as_int(q.w) * T + t], 1);
            }
        }
    }
}

#include "pp_v3.h"

// ---- entropy: pp_entropy_kernel_body lives in pp_common.h (shared with pp_v4.hip) ----
__global__  void pp_entropy_kernel(const int *__restrict__ counts, int n, int T, float *__restrict__ H) {
    pp_entropy_kernel_body(counts, n, T, H, blockIdx.x, gridDim.x);
}

int check_offsets(const int64_t *off, int n_trav, TravOffsets &tr) {
    MODEST_REQUIRE(off != nullptr, "trav_offsets is NULL");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= PP_MAX_TRAV, "1 <= n_trav <= 128");
    for (int t = 0; t <= n_trav; ++t) {
        MODEST_REQUIRE(off[t] >= 0, "negative offset");
        if (t) MODEST_REQUIRE(off[t] >= off[t - 1], "offsets must be non-decreasing");
        tr.off[t] = off[t];
    }
    MODEST_REQUIRE(off[n_trav] < (1LL << 31), "history must hold fewer than 2^31 points");
    tr.n = n_trav;
    return MODEST_OK;
}

}  // namespace

namespace {
// live scan of the frame store (tile-sorted + perm) -> common frame, ORIGINAL point order
__global__ void pp3_live_transform(const float *__restrict__ xyz, const unsigned *__restrict__ perm, int n,
                                   FrameDev d, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3];
    rel_apply(d.rel, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], o);
    const size_t j = perm[i];
    out[3 * j] = o[0];
    out[3 * j + 1] = o[1];
    out[3 * j + 2] = o[2];
}
}  // namespace

namespace {
// First kernel of the frame path, one launch for what used to be two memsets and two kernels:
// the live scan of the frame store (tile-sorted + perm) goes to the common frame in ORIGINAL point
// order, every block that holds points leaves its bounding-box partial (pp_live_count combines
// them: no atomics, nothing to pre-zero), and all blocks clear the zero-initialised control block
// and the count matrix.
struct RelM {
    float m[12];
};
constexpr int PREP_BLOCKS = 512;
__device__ __forceinline__ void pp3_live_prep_body(const float *__restrict__ xyz, const unsigned *__restrict__ perm, int n, RelM rel, float *__restrict__ out, unsigned *__restrict__ bbPart, uint4 *__restrict__ zero16, size_t nZero16, int *__restrict__ counts, size_t nCounts, const unsigned bx, const unsigned gx) {
    const size_t gt = (size_t)bx * 256 + threadIdx.x, gs = (size_t)gx * 256;
    for (size_t i = gt; i < nZero16; i += gs) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = gt; i < nCounts; i += gs) counts[i] = 0;
    if ((long long)bx * 1024 >= n) return;
    unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    for (int r = 0; r < 4; ++r) {   // 1024 points per block
        const int i = bx * 1024 + r * 256 + threadIdx.x;
        if (i < n) {
            float o[3];
            rel_apply(rel.m, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], o);
            const size_t j = perm[i];
            out[3 * j] = o[0];
            out[3 * j + 1] = o[1];
            out[3 * j + 2] = o[2];
            const unsigned kx = pp_fkey(o[0]), ky = pp_fkey(o[1]);
            k0 = max(k0, kx);
            k1 = max(k1, ~kx);
            k2 = max(k2, ky);
            k3 = max(k3, ~ky);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        k0 = max(k0, (unsigned)__shfl_xor((int)k0, o));
        k1 = max(k1, (unsigned)__shfl_xor((int)k1, o));
        k2 = max(k2, (unsigned)__shfl_xor((int)k2, o));
        k3 = max(k3, (unsigned)__shfl_xor((int)k3, o));
    }
    __shared__ unsigned red[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = k0;
        red[1][w] = k1;
        red[2][w] = k2;
        red[3][w] = k3;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        bbPart[4 * bx + threadIdx.x] = max(max(red[threadIdx.x][0], red[threadIdx.x][1]),
                                                   max(red[threadIdx.x][2], red[threadIdx.x][3]));
}
__global__ __launch_bounds__(256) void pp3_live_prep(const float *__restrict__ xyz, const unsigned *__restrict__ perm, int n, RelM rel, float *__restrict__ out, unsigned *__restrict__ bbPart, uint4 *__restrict__ zero16, size_t nZero16, int *__restrict__ counts, size_t nCounts) {
    pp3_live_prep_body(xyz, perm, n, rel, out, bbPart, zero16, nZero16, counts, nCounts, blockIdx.x, gridDim.x);
}
}  // namespace

// Where the history comes from: a stacked (M,3) array with traversal offsets, or the frames of the
// frame store through a descriptor table (chunkTab: one entry per 4096-point chunk of a frame).
struct HistSrc {
    const float *hist = nullptr;
    TravOffsets tr;
    ChunkMap3 cm;
    const FrameDev *frames = nullptr;
    const uint2 *chunkTab = nullptr;
    int nchunks = 0;
    bool useFrames = false;
    long long totalPts = 0;
    // frame path: the live scan as the frame store holds it (tile-sorted + perm); pp_count_run's first
    // kernel transforms it into `liveOut` (original point order), in place of a separate launch
    const float *liveXyz = nullptr;
    const unsigned *livePerm = nullptr;
    float liveRel[12] = {0};
    float *liveOut = nullptr;
};

// `extra_bytes` of arena are reserved behind this call's own carve; `prepare` is called with that
// block after the reservation and before the first launch (it may fill in `live` / `src` / `counts`
// with buffers inside the block and enqueue work that produces them).
static int pp_count_run(modest_ctx *ctx, const float *live, int n_live, HistSrc &src, int n_trav, double radius,
                        int32_t *counts, hipStream_t stream, size_t extra_bytes,
                        const std::function<int(char *extra, const float **live, int32_t **counts)> &prepare) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_live >= 0, "n_live < 0");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int nchunks = src.nchunks;
    MODEST_REQUIRE(nchunks < (1 << 19), "history too large for the routed path");
    int nwg3 = 2 * ctx->num_cus < V3_MAXWG ? 2 * ctx->num_cus : V3_MAXWG;
    {
        const char *nw_env = getenv("MODEST_PP_NWG");
        if (nw_env && atoi(nw_env) > 0 && atoi(nw_env) <= V3_MAXWG) nwg3 = atoi(nw_env);
    }
    if (nwg3 > nchunks) nwg3 = nchunks > 0 ? nchunks : 1;
    const char *sr_env = getenv("MODEST_PP_SLICE");
    unsigned sliceCap = sr_env ? (unsigned)atoi(sr_env) : V3_SLICE_MAX;
    sliceCap = sliceCap < 256u ? 256u : (sliceCap > V3_SLICE_MAX ? V3_SLICE_MAX : sliceCap);
    const size_t maxSlices = (size_t)V3_NL + (size_t)nchunks * V3_CH / 64 + 2;
    // one contiguous zero-initialised block:
    // cellCount | fill | bbox[4] | ctrl3[4] | listTotal | tileBase | listLive | dense | denseBlock | blockLive | dbg
    const size_t zero_words = (size_t)(PP_NCELL + 1) + PP_NCELL + 4 + 4 + 3 * V3_NL + 2 * V3_DWORDS + V3_DMAX +
                              V3_NBLK + 68 + 2 * (64 + 3 * 1024);
    size_t need = arena_sz(zero_words * 4) + arena_sz((size_t)(PP_NCELL + 1) * 4) + arena_sz(SCAN_NBLK * 4) +
                  arena_sz(PP_BITWORDS * 4) + arena_sz((size_t)n_live * 16) +
                  arena_sz((size_t)nchunks * V3_CH * 16) + 2 * arena_sz((size_t)nwg3 * V3_NL * 4) +
                  arena_sz(maxSlices * 16) + arena_sz((size_t)PREP_BLOCKS * 16);
    int rc = modest_ctx_reserve(ctx, need + arena_sz(extra_bytes));
    if (rc) return rc;
    if (prepare) {
        rc = prepare(ctx->scratch + need, &live, &counts);
        if (rc) return rc;
    }
    if (n_live == 0) return MODEST_OK;
    MODEST_REQUIRE(counts != nullptr, "counts is NULL");
    if (src.totalPts == 0) {   // no history: all counts are zero
        MODEST_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n_live * n_trav * sizeof(int32_t), stream));
        return MODEST_OK;
    }
    MODEST_REQUIRE(live != nullptr && (src.useFrames ? src.frames != nullptr : src.hist != nullptr), "NULL point buffer");
    Arena A(ctx->scratch);
    unsigned *zeroed = A.take<unsigned>(zero_words);
    unsigned *cellCount = zeroed;
    unsigned *fill = cellCount + (PP_NCELL + 1);
    unsigned *bb = fill + PP_NCELL;
    unsigned *ctrl3 = bb + 4;
    unsigned *listTotal = ctrl3 + 4;
    unsigned *tileBase = listTotal + V3_NL;
    unsigned *listLive = tileBase + V3_NL;
    unsigned *dense = listLive + V3_NL;
    unsigned *denseBlock = dense + 2 * V3_DWORDS;
    unsigned *blockLive = denseBlock + V3_DMAX;
    unsigned *dbgStats = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(blockLive + V3_NBLK) + 7) & ~(uintptr_t)7);   // 16 x u64 (PROF builds)
    unsigned *cellStart = A.take<unsigned>(PP_NCELL + 1);
    unsigned *blockSum = A.take<unsigned>(SCAN_NBLK);
    unsigned *bitmap = A.take<unsigned>(PP_BITWORDS);
    float4 *sorted = A.take<float4>(n_live);
    float4 *rec = A.take<float4>((size_t)nchunks * V3_CH);
    unsigned *wgTile = A.take<unsigned>((size_t)nwg3 * V3_NL);
    unsigned *wgOff = A.take<unsigned>((size_t)nwg3 * V3_NL);
    uint4 *slices = A.take<uint4>(maxSlices);
    unsigned *bbPart = A.take<unsigned>((size_t)PREP_BLOCKS * 4);

    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of one scan
    const double c = radius * (1.0 + 1.0 / 1024.0);
    const double r2 = radius * radius;
    const int nb = (n_live + 255) / 256;
    const int nPart = (n_live + 1023) / 1024;
    if (src.liveXyz && nPart <= PREP_BLOCKS) {   // frame path: transform + bbox partials + clears in one launch
        RelM rm;
        for (int q = 0; q < 12; ++q) rm.m[q] = src.liveRel[q];
        pp3_live_prep<<<PREP_BLOCKS, 256, 0, stream>>>(src.liveXyz, src.livePerm, n_live, rm, src.liveOut, bbPart,
                                                       reinterpret_cast<uint4 *>(zeroed), arena_sz(zero_words * 4) / 16,
                                                       counts, (size_t)n_live * n_trav);
        pp_live_count<<<nb, 256, 0, stream>>>(live, n_live, bb, c, cellCount, bbPart, nPart);
    } else {
        if (src.liveXyz) {   // a live scan of more than 512 k points: separate transform
            FrameDev ld;
            memset(&ld, 0, sizeof(ld));
            for (int q = 0; q < 12; ++q) ld.rel[q] = src.liveRel[q];
            pp3_live_transform<<<nb, 256, 0, stream>>>(src.liveXyz, src.livePerm, n_live, ld, src.liveOut);
        }
        MODEST_HIP_CHECK(hipMemsetAsync(zeroed, 0, zero_words * 4, stream));
        pp_live_bbox<<<nPart, 256, 0, stream>>>(live, n_live, bb, counts, (size_t)n_live * n_trav);
        pp_live_count<<<nb, 256, 0, stream>>>(live, n_live, bb, c, cellCount, nullptr, 0);
    }
    pp_scan_bitmap<<<SCAN_NBLK + PP_NY, SCAN_BLOCK, 0, stream>>>(cellCount, cellStart, blockSum, bitmap);
    pp_scan_finish<<<SCAN_NBLK, SCAN_BLOCK, 0, stream>>>(cellStart, blockSum);
    const char *var_env = getenv("MODEST_PP_VARIANT");
    int var = var_env ? atoi(var_env) : 3;
    if (var != 1 || n_trav > V3_MAXT) var = n_trav > V3_MAXT ? 1 : 3;   // more than 64 traversals: the direct path
    MODEST_REQUIRE(var == 3 || !src.useFrames, "the frame path needs n_trav <= 64");
    if (var == 3)   // the extra blocks count the live points of every 8x8-cell block window
        pp3_scatter_blocklive<<<nb + (V3_NBLK + 255) / 256, 256, 0, stream>>>(live, n_live, bb, c, cellStart, fill,
                                                                             sorted, nb, blockLive);
    else
        pp_live_scatter<<<nb, 256, 0, stream>>>(live, n_live, bb, c, cellStart, fill, sorted);
    if (var == 1) {   // V1: per-point search in the L2-resident index (also kept for A/B measurements)
        pp_stream_v1<<<ctx->num_cus * 3, 256, 0, stream>>>(src.hist, src.tr.off[0], src.tr.off[n_trav], src.tr, bb, c,
                                                           bitmap, cellStart, sorted, counts, n_trav, r2);
        modest_prof_mark(ctx, stream, 1);
        MODEST_HIP_CHECK(hipGetLastError());
        return MODEST_OK;
    }
    const char *dbg_env = getenv("MODEST_PP_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    if (!ctx->pp_attr_done) {
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_scan),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             V3_MAXWG * V3_SCAN_L * 4));
        ctx->pp_attr_done = 1;
    }
    pp3_blocks<<<1, 1024, 0, stream>>>(blockLive, dense, denseBlock, listLive);
    const int pair = (nwg3 % 2 == 0 && nwg3 >= 4 && !getenv("MODEST_PP_NOPAIR")) ? 1 : 0;
    if (src.useFrames)
        pp3_stream<false, true><<<nwg3, 1024, 0, stream>>>(nullptr, src.tr, src.cm, src.frames, src.chunkTab, nchunks, bb,
                                                           c, bitmap, dense, wgTile, wgOff, tileBase, rec, pair);
    else
        pp3_stream<false, false><<<nwg3, 1024, 0, stream>>>(src.hist, src.tr, src.cm, nullptr, nullptr, nchunks, bb, c,
                                                            bitmap, dense, wgTile, wgOff, tileBase, rec, pair);
    pp3_scan<<<V3_NL / V3_SCAN_L, 1024, (size_t)nwg3 * V3_SCAN_L * 4, stream>>>(wgTile, wgOff, nwg3, listTotal);
    pp3_plan<<<1, 1024, 0, stream>>>(listTotal, listLive, n_trav, sliceCap, tileBase, slices, (unsigned)maxSlices, ctrl3);
    const int sgrid = pair ? nwg3 / 2 : nwg3;
    if (src.useFrames)
        pp3_stream<true, true><<<sgrid, 1024, 0, stream>>>(nullptr, src.tr, src.cm, src.frames, src.chunkTab, nchunks, bb,
                                                           c, bitmap, dense, wgTile, wgOff, tileBase, rec, pair);
    else
        pp3_stream<true, false><<<sgrid, 1024, 0, stream>>>(src.hist, src.tr, src.cm, nullptr, nullptr, nchunks, bb, c,
                                                            bitmap, dense, wgTile, wgOff, tileBase, rec, pair);
    if (dbg & 8)
        pp3_join<true><<<2 * ctx->num_cus, V3_JT, V3_JOIN_LDS_DYN, stream>>>(
            rec, slices, ctrl3, denseBlock, cellStart, sorted, counts, n_trav, r2, dbg,
            reinterpret_cast<unsigned long long *>(dbgStats));
    else
        pp3_join<false><<<2 * ctx->num_cus, V3_JT, V3_JOIN_LDS_DYN, stream>>>(
            rec, slices, ctrl3, denseBlock, cellStart, sorted, counts, n_trav, r2, 0,
            reinterpret_cast<unsigned long long *>(dbgStats));
    if (dbg & 8) {
        unsigned long long hs[32];
        unsigned hc[4];
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        MODEST_HIP_CHECK(hipMemcpy(hs, dbgStats, sizeof(hs), hipMemcpyDeviceToHost));
        MODEST_HIP_CHECK(hipMemcpy(hc, ctrl3, sizeof(hc), hipMemcpyDeviceToHost));
        fprintf(stderr, "[pp3] slices %u records %u | wg-time (10ns ticks, summed over WGs) load+hist %llu tables+scatter %llu band-load %llu join %llu flush %llu | max WG %llu | chunks %llu groups %llu iters %llu\n",
                hc[0], hc[2], hs[0], hs[1], hs[2], hs[3], hs[4], hs[6], hs[8], hs[9], hs[10]);
    }
    modest_prof_mark(ctx, stream, 1);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

// stacked history: `extra_bytes` behind the carve are returned in *extra (modest_pp_score's counts)
static int pp_count_impl(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                         const int64_t *trav_offsets, int n_trav, double radius,
                         int32_t *counts, void *stream_, size_t extra_bytes, void **extra) {
    HistSrc src;
    int rc = check_offsets(trav_offsets, n_trav, src.tr);
    if (rc) return rc;
    long long nch = 0;
    for (int t = 0; t < n_trav; ++t) {   // chunks never straddle traversals
        src.cm.cstart[t] = (int)nch;
        nch += (src.tr.off[t + 1] - src.tr.off[t] + V3_CH - 1) / V3_CH;
    }
    MODEST_REQUIRE(nch < (1LL << 19), "history too large for the routed path");
    src.cm.cstart[n_trav] = (int)nch;
    src.nchunks = (int)nch;
    src.hist = hist;
    src.totalPts = src.tr.off[n_trav] - src.tr.off[0];
    return pp_count_run(ctx, live, n_live, src, n_trav, radius, counts, as_stream(stream_), extra_bytes,
                        [&](char *tail, const float **, int32_t **cnt) {
                            if (extra) {
                                *extra = tail;
                                if (!*cnt) *cnt = reinterpret_cast<int32_t *>(tail);
                            }
                            return MODEST_OK;
                        });
}

// ---- frame path of the V3 kernels ------------------------------------------------------------
int modest_pp3_frames(modest_ctx *ctx, const modest_pp_frame *live, const uint32_t *live_perm_dev,
                      const modest_pp_frame *frames, int n_frames, int n_trav, double radius,
                      int32_t *counts_dev, float *H_dev, hipStream_t stream) {
    const int N = live->n, T = n_trav;
    HistSrc src;
    src.useFrames = true;
    src.tr.n = n_trav;
    for (int t = 0; t <= n_trav && t <= PP_MAX_TRAV; ++t) src.tr.off[t] = 0;
    long long nch = 0, pts = 0;
    for (int f = 0; f < n_frames; ++f) {
        nch += (frames[f].n + V3_CH - 1) / V3_CH;
        pts += frames[f].n;
    }
    MODEST_REQUIRE(nch < (1LL << 19) && pts < (1LL << 31), "history too large for the routed path");
    src.nchunks = (int)nch;
    src.totalPts = pts;
    const size_t descB = arena_sz((size_t)(n_frames > 0 ? n_frames : 1) * sizeof(FrameDev));
    const size_t tabB = arena_sz((size_t)(nch > 0 ? nch : 1) * sizeof(uint2));
    const size_t liveB = arena_sz((size_t)N * 12 + 16);
    const size_t cntB = counts_dev ? 0 : arena_sz((size_t)N * T * 4);
    int32_t *cnt_used = counts_dev;
    int rc = pp_count_run(
        ctx, nullptr, N, src, n_trav, radius, counts_dev, stream, descB + tabB + liveB + cntB,
        [&](char *tail, const float **livep, int32_t **cnt) -> int {
            FrameDev *dframes = reinterpret_cast<FrameDev *>(tail);
            uint2 *dtab = reinterpret_cast<uint2 *>(tail + descB);
            float *dlive = reinterpret_cast<float *>(tail + descB + tabB);
            if (!*cnt) *cnt = reinterpret_cast<int32_t *>(tail + descB + tabB + liveB);
            cnt_used = *cnt;
            // descriptors and the chunk table travel through one pinned staging slot
            char *hslot = nullptr;
            int r = modest_ctx_stage_slot(ctx, descB + tabB, reinterpret_cast<void **>(&hslot));
            if (r) return r;
            FrameDev *hd = reinterpret_cast<FrameDev *>(hslot);
            uint2 *ht = reinterpret_cast<uint2 *>(hslot + descB);
            size_t k = 0;
            for (int f = 0; f < n_frames; ++f) {
                FrameDev &d = hd[f];
                d.xyz = frames[f].xyz_dev;
                d.tab = frames[f].tab_dev;
                d.n = frames[f].n;
                d.TX0 = frames[f].TX0;
                d.TY0 = frames[f].TY0;
                d.trav_flags = frames[f].trav | (frames[f].flags << 16);
                for (int q = 0; q < 12; ++q) d.rel[q] = frames[f].rel[q];
                for (int p0 = 0; p0 < frames[f].n; p0 += V3_CH) ht[k++] = make_uint2((unsigned)f, (unsigned)p0);
            }
            MODEST_HIP_CHECK(hipMemcpyAsync(dframes, hslot, descB + tabB, hipMemcpyHostToDevice, stream));
            r = modest_ctx_stage_commit(ctx, stream);
            if (r) return r;
            src.liveXyz = live->xyz_dev;   // transformed into dlive by pp_count_run's first kernel
            src.livePerm = live_perm_dev;
            for (int q = 0; q < 12; ++q) src.liveRel[q] = live->rel[q];
            src.liveOut = dlive;
            *livep = dlive;
            src.frames = dframes;
            src.chunkTab = dtab;
            return MODEST_OK;
        });
    if (rc) return rc;
    if (H_dev && N > 0) return modest_pp_entropy(ctx, cnt_used, N, T, H_dev, stream);
    return MODEST_OK;
}


// ---- several scans per launch ---------------------------------------------------------
// The chain above, once per BATCH of scans: every kernel gets the scan as blockIdx.y and reads that
// scan's pointers from a device table (PPB); the bodies are the very functions the single-scan kernels
// call.  Per scan nothing changes (same lists, slices, records, counts); what changes is that the nine
// sub-10 us launches are paid once per batch, that the streaming workgroups of one scan start while
// those of another finish, and that the start-up and the tail of the persistent join are paid once.
namespace {
struct PPB {
    const float *liveXyz;
    const unsigned *livePerm;
    float *live;
    uint4 *zero16;
    unsigned long long nZero16, nCounts;
    unsigned *cellCount, *fill, *bb, *ctrl3, *listTotal, *tileBase, *listLive, *dense, *denseBlock, *blockLive;
    unsigned *cellStart, *blockSum, *bitmap, *bbPart, *wgTile, *wgOff;
    float4 *sorted, *rec;
    uint4 *slices;
    int *counts;
    float *H;
    const FrameDev *frames;
    const uint2 *chunkTab;
    float liveRel[12];
    int n_live, n_trav, nb, nPart, nchunks, nwg3, pair, sgrid;
    unsigned sliceCap, maxSlices;
};

__global__ __launch_bounds__(256) void ppb_live_prep(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    RelM rm;
#pragma unroll
    for (int q = 0; q < 12; ++q) rm.m[q] = S.liveRel[q];
    pp3_live_prep_body(S.liveXyz, S.livePerm, S.n_live, rm, S.live, S.bbPart, S.zero16, (size_t)S.nZero16, S.counts,
                       (size_t)S.nCounts, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void ppb_live_count(const PPB *__restrict__ tab, double c) {
    const PPB &S = tab[blockIdx.y];
    if ((int)blockIdx.x >= S.nb) return;
    pp_live_count_body(S.live, S.n_live, S.bb, c, S.cellCount, S.bbPart, S.nPart, blockIdx.x, (unsigned)S.nb);
}
__global__ __launch_bounds__(SCAN_BLOCK) void ppb_scan_bitmap(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    pp_scan_bitmap_body(S.cellCount, S.cellStart, S.blockSum, S.bitmap, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(SCAN_BLOCK) void ppb_scan_finish(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    pp_scan_finish_body(S.cellStart, S.blockSum, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void ppb_scatter_blocklive(const PPB *__restrict__ tab, double c) {
    const PPB &S = tab[blockIdx.y];
    pp3_scatter_blocklive_body(S.live, S.n_live, S.bb, c, S.cellStart, S.fill, S.sorted, S.nb, S.blockLive, blockIdx.x,
                               gridDim.x);
}
__global__ __launch_bounds__(1024) void ppb_blocks(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    pp3_blocks_body(S.blockLive, S.dense, S.denseBlock, S.listLive, blockIdx.x, gridDim.x);
}
__device__ TravOffsets ppb_no_tr;   // stacked-history arguments of the stream body: never read when FRAMES
__device__ ChunkMap3 ppb_no_cm;
template <bool SCATTER>
__global__ __launch_bounds__(1024, 8) void ppb_stream(const PPB *__restrict__ tab, double c) {
    const PPB &S = tab[blockIdx.y];
    const unsigned g = (unsigned)(SCATTER ? S.sgrid : S.nwg3);
    if (blockIdx.x >= g) return;
    pp3_stream_body<SCATTER, true>(nullptr, ppb_no_tr, ppb_no_cm, S.frames, S.chunkTab, S.nchunks, S.bb, c, S.bitmap, S.dense, S.wgTile,
                                   S.wgOff, S.tileBase, S.rec, S.pair, blockIdx.x, g);
}
__global__ __launch_bounds__(1024) void ppb_scan(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    pp3_scan_body(S.wgTile, S.wgOff, S.nwg3, S.listTotal, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(1024) void ppb_plan(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    pp3_plan_body(S.listTotal, S.listLive, S.n_trav, S.sliceCap, S.tileBase, S.slices, S.maxSlices, S.ctrl3, blockIdx.x,
                  gridDim.x);
}
// The join's pointers travel in the kernel-argument segment (loads from it are invariant: the compiler reloads them
// instead of keeping them in registers; read from the device table they cost the register that makes the kernel spill).
constexpr int PPB_MAX = 8;   // scans per batch
struct JoinArgs {
    const float4 *rec;
    const uint4 *slices;
    unsigned *ctrl3;
    const unsigned *denseBlock, *cellStart;
    const float4 *sorted;
    int *counts;
    int T, pad;
};
struct JoinArgsB {
    JoinArgs a[PPB_MAX];
};
__global__ __launch_bounds__(V3_JT, 8) void ppb_join(const JoinArgsB A, double r2) {
    // every scan keeps its own slice queue and its share of the persistent workgroups
    const JoinArgs &S = A.a[blockIdx.y];
    pp3_join_body<false>(S.rec, S.slices, S.ctrl3, S.denseBlock, S.cellStart, S.sorted, S.counts, S.T, r2, 0, nullptr,
                         blockIdx.x);
}
__global__ void ppb_entropy(const PPB *__restrict__ tab) {
    const PPB &S = tab[blockIdx.y];
    if (S.H == nullptr || (int)blockIdx.x >= S.nb) return;
    pp_entropy_kernel_body(S.counts, S.n_live, S.n_trav, S.H, blockIdx.x, gridDim.x);
}
}  // namespace

int modest_pp3_frames_batch(modest_ctx *ctx, int n_scans, const modest_pp_frame *const *live,
                            const uint32_t *const *live_perm_dev, const modest_pp_frame *const *frames,
                            const int *n_frames, int n_trav, double radius, int32_t *const *counts_dev,
                            float *const *H_dev, hipStream_t stream) {
    MODEST_REQUIRE(n_scans >= 1 && n_scans <= 64, "1 <= n_scans <= 64");
    if (n_scans > PPB_MAX) {   // groups of at most PPB_MAX scans
        for (int s0 = 0; s0 < n_scans; s0 += PPB_MAX) {
            const int m = n_scans - s0 < PPB_MAX ? n_scans - s0 : PPB_MAX;
            int rc = modest_pp3_frames_batch(ctx, m, live + s0, live_perm_dev + s0, frames + s0, n_frames + s0, n_trav, radius,
                                             counts_dev ? counts_dev + s0 : nullptr, H_dev ? H_dev + s0 : nullptr, stream);
            if (rc) return rc;
        }
        return MODEST_OK;
    }
    const int T = n_trav;
    bool plain = false;   // a scan the batched kernels do not cover: every scan takes the single-scan call
    for (int s = 0; s < n_scans; ++s) {
        long long pts = 0;
        for (int f = 0; f < n_frames[s]; ++f) pts += frames[s][f].n;
        if (live[s]->n == 0 || pts == 0 || (live[s]->n + 1023) / 1024 > PREP_BLOCKS) plain = true;
    }
    if (plain || n_scans == 1 || getenv("MODEST_PP_NOBATCH")) {
        for (int s = 0; s < n_scans; ++s) {
            int rc = modest_pp3_frames(ctx, live[s], live_perm_dev[s], frames[s], n_frames[s], n_trav, radius,
                                       counts_dev ? counts_dev[s] : nullptr, H_dev ? H_dev[s] : nullptr, stream);
            if (rc) return rc;
        }
        return MODEST_OK;
    }
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    int nwg3 = 2 * ctx->num_cus < V3_MAXWG ? 2 * ctx->num_cus : V3_MAXWG;
    {
        const char *nw_env = getenv("MODEST_PP_NWG");
        if (nw_env && atoi(nw_env) > 0 && atoi(nw_env) <= V3_MAXWG) nwg3 = atoi(nw_env);
    }
    if (!getenv("MODEST_PP_NWG") && !getenv("MODEST_PP_CHAIN_FULL")) {
        // the chain's streaming launches keep the footprint of a single scan's (2 workgroups per CU in total, dealt over
        // the scans): other processes' kernels find wave slots while a chain runs; alone on the GPU it costs nothing
        nwg3 = (nwg3 / n_scans) & ~1;
        if (nwg3 < 32) nwg3 = 32;
    }
    const char *sr_env = getenv("MODEST_PP_SLICE");
    unsigned sliceCap = sr_env ? (unsigned)atoi(sr_env) : V3_SLICE_MAX;
    sliceCap = sliceCap < 256u ? 256u : (sliceCap > V3_SLICE_MAX ? V3_SLICE_MAX : sliceCap);
    const size_t zero_words = (size_t)(PP_NCELL + 1) + PP_NCELL + 4 + 4 + 3 * V3_NL + 2 * V3_DWORDS + V3_DMAX + V3_NBLK + 68 +
                              2 * (64 + 3 * 1024);
    struct Lay {
        int N, nch, nwg, pair, sgrid;
        size_t maxSlices, descB, tabB, off, stageOff;
    };
    std::vector<Lay> L((size_t)n_scans);
    size_t total = 0, stageB = 0;
    int maxNb = 1, maxNwg = 1, maxSgrid = 1;
    for (int s = 0; s < n_scans; ++s) {
        Lay &l = L[(size_t)s];
        l.N = live[s]->n;
        long long nch = 0, pts = 0;
        for (int f = 0; f < n_frames[s]; ++f) {
            nch += (frames[s][f].n + V3_CH - 1) / V3_CH;
            pts += frames[s][f].n;
        }
        MODEST_REQUIRE(nch < (1LL << 19) && pts < (1LL << 31), "history too large for the routed path");
        l.nch = (int)nch;
        l.nwg = nwg3 > l.nch ? l.nch : nwg3;
        l.pair = (l.nwg % 2 == 0 && l.nwg >= 4 && !getenv("MODEST_PP_NOPAIR")) ? 1 : 0;
        l.sgrid = l.pair ? l.nwg / 2 : l.nwg;
        l.maxSlices = (size_t)V3_NL + (size_t)l.nch * V3_CH / 64 + 2;
        l.descB = arena_sz((size_t)n_frames[s] * sizeof(FrameDev));
        l.tabB = arena_sz((size_t)l.nch * sizeof(uint2));
        l.off = total;
        l.stageOff = stageB;
        stageB += l.descB + l.tabB;
        total += arena_sz(zero_words * 4) + arena_sz((size_t)(PP_NCELL + 1) * 4) + arena_sz(SCAN_NBLK * 4) +
                 arena_sz(PP_BITWORDS * 4) + arena_sz((size_t)l.N * 16) + arena_sz((size_t)l.nch * V3_CH * 16) +
                 2 * arena_sz((size_t)l.nwg * V3_NL * 4) + arena_sz(l.maxSlices * 16) + arena_sz((size_t)PREP_BLOCKS * 16) +
                 arena_sz((size_t)l.N * 12 + 16) + arena_sz((size_t)l.N * T * 4);
        maxNb = std::max(maxNb, (l.N + 255) / 256);
        maxNwg = std::max(maxNwg, l.nwg);
        maxSgrid = std::max(maxSgrid, l.sgrid);
    }
    const size_t tabOff = total, tableB = arena_sz((size_t)n_scans * sizeof(PPB));
    // device: [scan arenas][PPB table][staged block]; the staged block = [scan 0: frames | chunks] [scan 1: ...] ...
    // and the table travel through one pinned staging slot
    int rc = modest_ctx_reserve(ctx, total + tableB + stageB);
    if (rc) return rc;
    char *hslot = nullptr;
    rc = modest_ctx_stage_slot(ctx, stageB + tableB, reinterpret_cast<void **>(&hslot));
    if (rc) return rc;
    char *stageDev = ctx->scratch + total + tableB;
    PPB *htab = reinterpret_cast<PPB *>(hslot + stageB);
    for (int s = 0; s < n_scans; ++s) {
        const Lay &l = L[(size_t)s];
        FrameDev *hd = reinterpret_cast<FrameDev *>(hslot + l.stageOff);
        uint2 *ht = reinterpret_cast<uint2 *>(hslot + l.stageOff + l.descB);
        size_t k = 0;
        for (int f = 0; f < n_frames[s]; ++f) {
            const modest_pp_frame &fr = frames[s][f];
            FrameDev &d = hd[f];
            d.xyz = fr.xyz_dev;
            d.tab = fr.tab_dev;
            d.n = fr.n;
            d.TX0 = fr.TX0;
            d.TY0 = fr.TY0;
            d.trav_flags = fr.trav | (fr.flags << 16);
            for (int q = 0; q < 12; ++q) d.rel[q] = fr.rel[q];
            for (int p0 = 0; p0 < fr.n; p0 += V3_CH) ht[k++] = make_uint2((unsigned)f, (unsigned)p0);
        }
        Arena A(ctx->scratch + l.off);
        PPB &b = htab[s];
        memset(&b, 0, sizeof(b));
        unsigned *zeroed = A.take<unsigned>(zero_words);
        b.zero16 = reinterpret_cast<uint4 *>(zeroed);
        b.nZero16 = arena_sz(zero_words * 4) / 16;
        b.cellCount = zeroed;
        b.fill = b.cellCount + (PP_NCELL + 1);
        b.bb = b.fill + PP_NCELL;
        b.ctrl3 = b.bb + 4;
        b.listTotal = b.ctrl3 + 4;
        b.tileBase = b.listTotal + V3_NL;
        b.listLive = b.tileBase + V3_NL;
        b.dense = b.listLive + V3_NL;
        b.denseBlock = b.dense + 2 * V3_DWORDS;
        b.blockLive = b.denseBlock + V3_DMAX;
        b.cellStart = A.take<unsigned>(PP_NCELL + 1);
        b.blockSum = A.take<unsigned>(SCAN_NBLK);
        b.bitmap = A.take<unsigned>(PP_BITWORDS);
        b.sorted = A.take<float4>(l.N);
        b.rec = A.take<float4>((size_t)l.nch * V3_CH);
        b.wgTile = A.take<unsigned>((size_t)l.nwg * V3_NL);
        b.wgOff = A.take<unsigned>((size_t)l.nwg * V3_NL);
        b.slices = A.take<uint4>(l.maxSlices);
        b.bbPart = A.take<unsigned>((size_t)PREP_BLOCKS * 4);
        b.live = A.take<float>((size_t)l.N * 3 + 4);   // arena_sz(N * 12 + 16) above
        int32_t *cscr = A.take<int32_t>((size_t)l.N * T);
        b.counts = (counts_dev && counts_dev[s]) ? counts_dev[s] : cscr;
        b.nCounts = (unsigned long long)l.N * T;
        b.H = H_dev ? H_dev[s] : nullptr;
        b.frames = reinterpret_cast<const FrameDev *>(stageDev + l.stageOff);
        b.chunkTab = reinterpret_cast<const uint2 *>(stageDev + l.stageOff + l.descB);
        b.liveXyz = live[s]->xyz_dev;
        b.livePerm = live_perm_dev[s];
        for (int q = 0; q < 12; ++q) b.liveRel[q] = live[s]->rel[q];
        b.n_live = l.N;
        b.n_trav = T;
        b.nb = (l.N + 255) / 256;
        b.nPart = (l.N + 1023) / 1024;
        b.nchunks = l.nch;
        b.nwg3 = l.nwg;
        b.pair = l.pair;
        b.sgrid = l.sgrid;
        b.sliceCap = sliceCap;
        b.maxSlices = (unsigned)l.maxSlices;
    }
    // one copy: [frames | chunks]* then the table (device layout: table at tabOff, staged block behind it)
    MODEST_HIP_CHECK(hipMemcpyAsync(stageDev, hslot, stageB, hipMemcpyHostToDevice, stream));
    MODEST_HIP_CHECK(hipMemcpyAsync(ctx->scratch + tabOff, hslot + stageB, (size_t)n_scans * sizeof(PPB),
                                    hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx, stream);
    if (rc) return rc;
    const PPB *tab = reinterpret_cast<const PPB *>(ctx->scratch + tabOff);
    if (!ctx->pp_attr_done) {
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_scan),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_MAXWG * V3_SCAN_L * 4));
        ctx->pp_attr_done = 1;
    }
    if (!ctx->ppb_attr_done) {
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ppb_join),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ppb_scan),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, V3_MAXWG * V3_SCAN_L * 4));
        ctx->ppb_attr_done = 1;
    }
    const double c = radius * (1.0 + 1.0 / 1024.0), r2 = radius * radius;
    const unsigned B = (unsigned)n_scans;
    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of the batch
    ppb_live_prep<<<dim3(PREP_BLOCKS, B), 256, 0, stream>>>(tab);
    ppb_live_count<<<dim3((unsigned)maxNb, B), 256, 0, stream>>>(tab, c);
    ppb_scan_bitmap<<<dim3(SCAN_NBLK + PP_NY, B), SCAN_BLOCK, 0, stream>>>(tab);
    ppb_scan_finish<<<dim3(SCAN_NBLK, B), SCAN_BLOCK, 0, stream>>>(tab);
    ppb_scatter_blocklive<<<dim3((unsigned)maxNb + (V3_NBLK + 255) / 256, B), 256, 0, stream>>>(tab, c);
    ppb_blocks<<<dim3(1, B), 1024, 0, stream>>>(tab);
    ppb_stream<false><<<dim3((unsigned)maxNwg, B), 1024, 0, stream>>>(tab, c);
    ppb_scan<<<dim3(V3_NL / V3_SCAN_L, B), 1024, (size_t)maxNwg * V3_SCAN_L * 4, stream>>>(tab);
    ppb_plan<<<dim3(1, B), 1024, 0, stream>>>(tab);
    ppb_stream<true><<<dim3((unsigned)maxSgrid, B), 1024, 0, stream>>>(tab, c);
    // the join keeps the single-scan grid in total: two workgroups per CU, dealt over the scans
    unsigned jx = (unsigned)(2 * ctx->num_cus) / B;
    if (jx < 1) jx = 1;
    JoinArgsB ja;
    memset(&ja, 0, sizeof(ja));
    for (int s = 0; s < n_scans; ++s) {
        const PPB &b = htab[s];   // (the staging slot stays valid until its commit is overtaken)
        ja.a[s] = JoinArgs{b.rec, b.slices, b.ctrl3, b.denseBlock, b.cellStart, b.sorted, b.counts, T, 0};
    }
    ppb_join<<<dim3(jx, B), V3_JT, V3_JOIN_LDS_DYN, stream>>>(ja, r2);
    modest_prof_mark(ctx, stream, 1);
    ppb_entropy<<<dim3((unsigned)maxNb, B), 256, 0, stream>>>(tab);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_pp_count(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                               const int64_t *trav_offsets, int n_trav, double radius,
                               int32_t *counts, void *stream_) {
    MODEST_REQUIRE(counts != nullptr || n_live == 0, "counts is NULL");
    return pp_count_impl(ctx, live, n_live, hist, trav_offsets, n_trav, radius, counts, stream_, 0,
                         nullptr);
}

extern "C" int modest_pp_entropy(modest_ctx *ctx, const int32_t *counts, int n_live, int n_trav,
                                 float *H, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_live >= 0, "n_live < 0");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= PP_MAX_TRAV, "1 <= n_trav <= 128");
    if (n_live == 0) return MODEST_OK;
    MODEST_REQUIRE(counts != nullptr && H != nullptr, "NULL buffer");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    pp_entropy_kernel<<<(n_live + 255) / 256, 256, 0, as_stream(stream_)>>>(counts, n_live, n_trav, H);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

extern "C" int modest_pp_score(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                               const int64_t *trav_offsets, int n_trav, double radius,
                               int32_t *counts, float *H, void *stream_) {
    MODEST_REQUIRE(n_live >= 0 && n_trav >= 1 && n_trav <= PP_MAX_TRAV, "bad sizes");
    void *tail = nullptr;
    const size_t extra = counts ? 0 : (size_t)n_live * n_trav * sizeof(int32_t);
    int rc = pp_count_impl(ctx, live, n_live, hist, trav_offsets, n_trav, radius, counts, stream_,
                           extra, &tail);
    if (rc) return rc;
    const int32_t *cN = counts ? counts : static_cast<const int32_t *>(tail);
    return modest_pp_entropy(ctx, cN, n_live, n_trav, H, stream_);
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_pp_count(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(pp_live_scatter));
}
