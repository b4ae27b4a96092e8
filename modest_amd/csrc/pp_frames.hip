// Frame store (modest_frame_sort) and the one-pass gather-join over it (round 2).
//
// Reference steps replaced: pre_compute_pp_score.py:132-150 (per-frame transform_points +
// remove_center + np.concatenate), :188-190 (cKDTree per traversal), :54-60 (count_neighbors).
//
// Every raw frame is sorted ONCE, when it enters the frame store, by the 8x8-cell tile of a WORLD
// lattice that all frames share (a frame serves ~70 scans), and keeps a prefix table tile -> [start,
// end).  modest_pp_score_frames reads frames through a descriptor table; by default the V3 streaming
// kernels do (pp_count.hip: pose fused, no stacked history).  This file also holds the experimental
// one-pass path (MODEST_PP_FRAMES_PATH=gather-wave), in which the history crosses HBM once and
// nothing is written back:
//
//   pp5_live_index  one workgroup per four live tiles: cell-sort the live points inside their tile,
//                   (offset, count) per cell, occupancy mask and border sums per tile; zeroes the
//                   count matrix.
//   pp5_plan        one workgroup per tile: exact window activity, the tile's run in every frame's
//                   table, the window's cells, rectangle subdivision of dense tiles, items.
//   pp6_wave_join   persistent wavefronts, dynamic dequeue.  An item = a tile (or a rectangle of its
//                   cells) x a frame range: the wavefront gathers the runs of ITS tile from the frames
//                   (the runs of one tile in 360 frames are the records V3 has to scatter into a
//                   list), applies each frame's relative pose (the reference's float32 BLAS
//                   rounding), drops points whose cell has no live point in its 3x3 neighbourhood,
//                   queues the survivors and joins 64 of them at a time against the live points of
//                   the window, which sit in a private LDS slice.  No workgroup barrier.
//
// A workgroup-per-tile version of the join (rounds of 3072 records sorted in LDS, V3's candidate
// loops) was measured at 620-1900 us per scan and showed an intermittent memory fault in one
// configuration; it was removed (DESIGN.md section 4.1 keeps its numbers).
//
// Cells.  A point's TILE is the one it was sorted into (float64 world lattice, at insertion).
// Its CELL inside the tile is recomputed from its float32 common-frame coordinates through the
// scan-uniform map A (common frame -> lattice), clamped into the tile.  Live and history points
// use the same function, so two points within r differ by at most one cell (pp_common.h); the
// clamp is safe as long as the insertion lattice and A agree to better than c - r = r/1024
// (the host checks 1e-4 m per frame and falls back to the stacked path otherwise).
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
constexpr int F_NC = F_TS * F_TS;        // 64 cells per tile = sort keys
constexpr int F_W = F_TS + 2;            // window incl. halo
constexpr int F_MAXT = 64;               // one lane per traversal in the segmented popcount

static_assert(F_NC == 64, "cell key = 6 bits");

struct Map24 {
    double a[8];   // rows x, y of a 3x4 map into lattice CELL coordinates
};
struct Mat34f {
    float m[12];
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

// ---- shared per-point arithmetic (rel_apply: pp_frames.h) ------------------------------------
// cell of a common-frame point inside tile (gtx, gty) [global tile coordinates], clamped into it
__device__ __forceinline__ int cell_in_tile(const Map24 &A, float x, float y, float z, int gtx, int gty) {
    const double lx = fma(A.a[2], (double)z, fma(A.a[1], (double)y, A.a[0] * (double)x)) + A.a[3];
    const double ly = fma(A.a[6], (double)z, fma(A.a[5], (double)y, A.a[4] * (double)x)) + A.a[7];
    double fx = floor(lx) - 8.0 * (double)gtx, fy = floor(ly) - 8.0 * (double)gty;
    fx = fmin(fmax(fx, 0.0), 7.0);   // NaN -> 0
    fy = fmin(fmax(fy, 0.0), 7.0);
    return (int)fy * F_TS + (int)fx;
}

// ---- live index -------------------------------------------------------------------------------
// One workgroup per four tiles of the live frame's table, the whole workgroup on one tile at a
// time: counting sort by cell inside the tile.
//   sorted[a + pos] = (x, y, z in the common frame, original index)
//   cellPk[tile * 64 + k] = (offset of cell k inside the tile, live points in it)  [non-empty tiles]
//   tileOcc[tile] = bit k: cell k holds a live point                                 [every tile]
constexpr int LI_TILES = 4;
constexpr int PL_SHARDS = 32;   // shards of the item counters (pp5_plan)
static_assert(F_NTILE % LI_TILES == 0, "live index tiling");
__global__ __launch_bounds__(256) void pp5_live_index(const float *__restrict__ lxyz, const unsigned *__restrict__ lperm,
                                                      const unsigned *__restrict__ ltab, int LTX0, int LTY0,
                                                      Mat34f rel, Map24 A, float4 *__restrict__ sorted,
                                                      uint2 *__restrict__ cellPk,
                                                      unsigned long long *__restrict__ tileOcc,
                                                      uint4 *__restrict__ tileEdge /* 2 per tile, non-empty tiles */,
                                                      int *__restrict__ counts, size_t nCounts,
                                                      unsigned *__restrict__ ctrl) {
    __shared__ unsigned hist[F_NC];
    __shared__ unsigned tabv[LI_TILES + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < nCounts; i += (size_t)gridDim.x * 256) counts[i] = 0;
    if (blockIdx.x == 0 && tid < 8 + 2 * PL_SHARDS) ctrl[tid] = 0;
    const int t0 = blockIdx.x * LI_TILES;
    if (tid <= LI_TILES) tabv[tid] = ltab[t0 + tid];
    __syncthreads();
    for (int k = 0; k < LI_TILES; ++k) {
        const unsigned a = tabv[k], b = tabv[k + 1];
        const int tile = t0 + k;
        if (b == a) {   // uniform
            if (tid == 0) tileOcc[tile] = 0ULL;
            continue;
        }
        const int gtx = LTX0 + tile % F_NTF, gty = LTY0 + tile / F_NTF;
        if (tid < F_NC) hist[tid] = 0;
        __syncthreads();
        float o[3] = {0.f, 0.f, 0.f};
        int key0 = 0;
        for (unsigned base = a; base < b; base += 256) {
            const unsigned i = base + tid;
            if (i < b) {
                float q[3];
                rel_apply(rel.m, lxyz[3 * (size_t)i], lxyz[3 * (size_t)i + 1], lxyz[3 * (size_t)i + 2], q);
                const int key = cell_in_tile(A, q[0], q[1], q[2], gtx, gty);
                atomicAdd(&hist[key], 1u);
                if (base == a) {   // the first (usually only) pass stays in registers
                    o[0] = q[0];
                    o[1] = q[1];
                    o[2] = q[2];
                    key0 = key;
                }
            }
        }
        __syncthreads();
        if (tid < F_NC) {
            const unsigned c0 = hist[tid];
            unsigned inc = c0;
            for (int s = 1; s < 64; s <<= 1) {
                const unsigned u = __shfl_up(inc, s);
                if (lane >= s) inc += u;
            }
            cellPk[(size_t)tile * F_NC + tid] = make_uint2(inc - c0, c0);
            hist[tid] = inc - c0;
            const unsigned long long occ = __ballot(c0 != 0u);
            if (tid == 0) tileOcc[tile] = occ;
            // live points on the tile's border: what the 10x10-cell windows of the neighbours see
            // (top row, bottom row, left column, right column; corners 00, 70, 07, 77)
            const int kx = tid & 7, ky = tid >> 3;
            unsigned e0 = ky == 0 ? c0 : 0u, e1 = ky == 7 ? c0 : 0u, e2 = kx == 0 ? c0 : 0u, e3 = kx == 7 ? c0 : 0u;
            for (int o = 32; o > 0; o >>= 1) {
                e0 += __shfl_xor(e0, o);
                e1 += __shfl_xor(e1, o);
                e2 += __shfl_xor(e2, o);
                e3 += __shfl_xor(e3, o);
            }
            const unsigned k00 = __builtin_amdgcn_readlane(c0, 0), k70 = __builtin_amdgcn_readlane(c0, 7);
            const unsigned k07 = __builtin_amdgcn_readlane(c0, 56), k77 = __builtin_amdgcn_readlane(c0, 63);
            if (tid == 0) {
                tileEdge[2 * (size_t)tile] = make_uint4(e0, e1, e2, e3);
                tileEdge[2 * (size_t)tile + 1] = make_uint4(k00, k70, k07, k77);
            }
        }
        __syncthreads();
        for (unsigned base = a; base < b; base += 256) {
            const unsigned i = base + tid;
            if (i < b) {
                int key = key0;
                if (base != a) {
                    rel_apply(rel.m, lxyz[3 * (size_t)i], lxyz[3 * (size_t)i + 1], lxyz[3 * (size_t)i + 2], o);
                    key = cell_in_tile(A, o[0], o[1], o[2], gtx, gty);
                }
                const unsigned pos = atomicAdd(&hist[key], 1u);
                sorted[a + pos] = make_float4(o[0], o[1], o[2], __int_as_float((int)lperm[i]));
            }
        }
        __syncthreads();
    }
}

// ---- plan ----------------------------------------------------------------------------------
// One workgroup per tile of the live table.  A tile is active when its 10x10-cell window holds a live
// point (exact, from the occupancy masks of the 3x3 tiles).  An active tile looks up ITS run in the
// table of every history frame -- runs[tile * F + f] = (first point, length) --, stores the window's
// cells (count, start) for the wave items and cuts the frame list into items (tile, first frame, end
// frame, cell rectangle) of about `wmax` points.  Item slots come from sharded counters (one device
// atomic per item on ONE word would serialise at ~11 ns each):
//   ctrl[8 + s] class-H items of shard s, ctrl[8 + PL_SHARDS + s] class-L items; shard = tile % PL_SHARDS;
//   shard s owns items[s * shardCap ...).  Class H = the rectangles of dense tiles, dequeued first.
constexpr int PL_T = 256;
constexpr int PL_MAX_RECTS = 192;     // cell rectangles of a dense tile (wave path)
constexpr int W_CAP = 128;   // live points a single wavefront keeps in its LDS slice (wave path)
// wave item rectangle (item.w): all cells of the tile, all window rows
__host__ __device__ __forceinline__ unsigned pp6_full_rect() { return 0u | (7u << 3) | (0u << 6) | (7u << 9) | (0u << 12) | (9u << 16); }
constexpr int PL_ITEMS_PER_TILE = 128;   // >= 33 point cuts + (frames / fmax) frame cuts
constexpr unsigned PL_SHARD_CAP = (unsigned)(F_NTILE / PL_SHARDS) * PL_ITEMS_PER_TILE;
__global__ __launch_bounds__(PL_T) void pp5_plan(const FrameDev *__restrict__ frames, int nFrames,
                                                 const unsigned *__restrict__ ltab, int LTX0, int LTY0,
                                                 const unsigned long long *__restrict__ tileOcc,
                                                 const uint4 *__restrict__ tileEdge, const uint2 *__restrict__ cellPk,
                                                 unsigned wmax, unsigned denseItems, int fmax,
                                                 uint2 *__restrict__ runs,
                                                 uint4 *__restrict__ itemsH, uint4 *__restrict__ itemsL,
                                                 unsigned *__restrict__ itemPts /* [2][shards][cap] points per item */,
                                                 uint2 *__restrict__ winTab /* [tile][100] (count, start) of the window cells */,
                                                 unsigned *__restrict__ ctrl) {
    extern __shared__ unsigned pre[];   // nFrames + 1 prefix of the run lengths
    __shared__ unsigned wsum[PL_T / 64];
    __shared__ unsigned cuts[PL_ITEMS_PER_TILE + 2];
    __shared__ unsigned nCuts, slotBase;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tx = t % F_NTF, ty = t / F_NTF;
    constexpr unsigned long long COL0 = 0x0101010101010101ULL, COL7 = 0x8080808080808080ULL;
    constexpr unsigned long long ROW0 = 0xffULL, ROW7 = 0xffULL << 56;
    auto occ = [&](int x, int y) -> unsigned long long {
        return (x >= 0 && x < F_NTF && y >= 0 && y < F_NTF) ? tileOcc[y * F_NTF + x] : 0ULL;
    };
    const unsigned long long any = occ(tx, ty) | (occ(tx - 1, ty) & COL7) | (occ(tx + 1, ty) & COL0) |
                                   (occ(tx, ty - 1) & ROW7) | (occ(tx, ty + 1) & ROW0) |
                                   (occ(tx - 1, ty - 1) & (1ULL << 63)) | (occ(tx + 1, ty - 1) & (1ULL << 56)) |
                                   (occ(tx - 1, ty + 1) & (1ULL << 7)) | (occ(tx + 1, ty + 1) & 1ULL);
    if (!any || nFrames <= 0) return;   // uniform
    // exact number of live points in the 10x10-cell window (own tile + the neighbours' border cells)
    unsigned wlive = ltab[t + 1] - ltab[t];
    {
        auto edge = [&](int x, int y, int which) -> unsigned {
            if (x < 0 || x >= F_NTF || y < 0 || y >= F_NTF) return 0u;
            const int tt = y * F_NTF + x;
            if (!tileOcc[tt]) return 0u;
            const uint4 a = tileEdge[2 * (size_t)tt], b = tileEdge[2 * (size_t)tt + 1];
            switch (which) {
                case 0: return a.x;   // top row (ky = 0)
                case 1: return a.y;   // bottom row (ky = 7)
                case 2: return a.z;   // left column
                case 3: return a.w;   // right column
                case 4: return b.x;   // cell (0,0)
                case 5: return b.y;   // cell (7,0)
                case 6: return b.z;   // cell (0,7)
                default: return b.w;  // cell (7,7)
            }
        };
        wlive += edge(tx, ty - 1, 1) + edge(tx, ty + 1, 0) + edge(tx - 1, ty, 3) + edge(tx + 1, ty, 2) +
                 edge(tx - 1, ty - 1, 7) + edge(tx + 1, ty - 1, 6) + edge(tx - 1, ty + 1, 5) + edge(tx + 1, ty + 1, 4);
    }
    const int gtx = LTX0 + tx, gty = LTY0 + ty;
    const int per = (nFrames + PL_T - 1) / PL_T;
    const int fa = tid * per, fb = min(fa + per, nFrames);
    unsigned s = 0;
    for (int f = fa; f < fb; ++f) {
        const float4 *q = reinterpret_cast<const float4 *>(frames + f);
        const float4 h0 = q[0], h1 = q[1];
        const unsigned *tab = reinterpret_cast<const unsigned *>(
            ((unsigned long long)__float_as_uint(h0.w) << 32) | (unsigned long long)__float_as_uint(h0.z));
        const int ltx = gtx - __float_as_int(h1.y), lty = gty - __float_as_int(h1.z);
        unsigned a = 0, b = 0;
        if (ltx >= 0 && ltx < F_NTF && lty >= 0 && lty < F_NTF) {
            a = tab[lty * F_NTF + ltx];
            b = tab[lty * F_NTF + ltx + 1];
        }
        runs[(size_t)t * nFrames + f] = make_uint2(a, b - a);
        pre[f] = b - a;
        s += b - a;
    }
    unsigned inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned run = inc - s, total = 0;
    for (int k = 0; k < PL_T / 64; ++k) {
        if (k < w) run += wsum[k];
        total += wsum[k];
    }
    for (int f = fa; f < fb; ++f) {
        const unsigned v = pre[f];
        pre[f] = run;
        run += v;
    }
    if (tid == 0) pre[nFrames] = total;
    __syncthreads();
    if (total == 0) return;   // uniform
    // cuts: a new item starts at frame f when the points before it cross a multiple of `step`, or
    // at every multiple of fmax frames
    // Every item is a wave item.  A tile whose window
    // holds more than W_CAP live points is subdivided: its 8x8 cells are halved recursively until the
    // live points around a rectangle of cells (the rectangle grown by one cell) fit a wavefront's LDS
    // slice; a single cell that still does not fit is split by window row (and, beyond that, the
    // kernel walks its live points in slices).  Every rectangle re-gathers the tile's points and keeps
    // the records of its own cells, so rectangles x frame cuts is capped (`denseItems`).
    __shared__ unsigned rects[PL_MAX_RECTS];
    __shared__ unsigned nRects;
    __shared__ unsigned short wc[F_W * F_W];
    __shared__ unsigned W2[(F_W + 1) * (F_W + 1)];
    __shared__ unsigned stack[64];
    const bool dense = wlive > (unsigned)W_CAP;
    if (tid == 0) {
        rects[0] = pp6_full_rect();
        nRects = 1;
    }
    if (tid < F_W * F_W) {   // the window's cells, once per tile: wave items read them in one round trip
        const int r = tid / F_W, cc = tid - r * F_W;
        const int wx = tx * F_TS - 1 + cc, wy = ty * F_TS - 1 + r;
        unsigned c = 0, g = 0;
        if (wx >= 0 && wy >= 0 && wx < F_NTF * F_TS && wy < F_NTF * F_TS) {
            const int tile = (wy >> 3) * F_NTF + (wx >> 3);
            const unsigned ta = ltab[tile], tb = ltab[tile + 1];
            if (tb > ta) {
                const uint2 pk = cellPk[(size_t)tile * F_NC + (wy & 7) * F_TS + (wx & 7)];
                c = pk.y;
                g = ta + pk.x;
            }
        }
        wc[tid] = (unsigned short)min(c, 65535u);
        winTab[(size_t)t * (F_W * F_W) + tid] = make_uint2(c, g);
    }
    if (dense) {   // uniform
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < (F_W + 1) * (F_W + 1); ++i) W2[i] = 0;
            for (int r = 0; r < F_W; ++r)
                for (int cc = 0; cc < F_W; ++cc)
                    W2[(r + 1) * (F_W + 1) + cc + 1] = wc[r * F_W + cc] + W2[r * (F_W + 1) + cc + 1] +
                                                       W2[(r + 1) * (F_W + 1) + cc] - W2[r * (F_W + 1) + cc];
            auto live_of = [&](unsigned rc) -> unsigned {   // live points the item of rectangle rc holds
                const int cx0 = rc & 7, cx1 = (rc >> 3) & 7, cy0 = (rc >> 6) & 7, cy1 = (rc >> 9) & 7;
                const int ry0 = (rc >> 12) & 15, ry1 = (rc >> 16) & 15;
                const int r0 = max(cy0, ry0), r1 = min(cy1 + 2, ry1), c0 = cx0, c1 = cx1 + 2;
                if (r1 < r0) return 0u;
                return W2[(r1 + 1) * (F_W + 1) + c1 + 1] - W2[r0 * (F_W + 1) + c1 + 1] - W2[(r1 + 1) * (F_W + 1) + c0] +
                       W2[r0 * (F_W + 1) + c0];
            };
            unsigned sp = 0, n = 0;   // (the stack lives in LDS: an indexed per-thread array would be scratch memory)
            stack[sp++] = pp6_full_rect();
            while (sp) {
                const unsigned rc = stack[--sp];
                const unsigned lv = live_of(rc);
                if (lv == 0) continue;   // no live point around these cells: their records match nothing
                const int cx0 = rc & 7, cx1 = (rc >> 3) & 7, cy0 = (rc >> 6) & 7, cy1 = (rc >> 9) & 7;
                const bool rowSplit = ((rc >> 12) & 15) != 0 || ((rc >> 16) & 15) != 9;
                if (lv <= (unsigned)W_CAP || rowSplit || n + 4 > (unsigned)PL_MAX_RECTS || sp + 3 > 64) {
                    if (n < (unsigned)PL_MAX_RECTS) rects[n++] = rc;
                    continue;
                }
                if (cx1 > cx0 || cy1 > cy0) {   // halve the longer side
                    if (cx1 - cx0 >= cy1 - cy0) {
                        const int m = (cx0 + cx1) >> 1;
                        stack[sp++] = (rc & ~0x3fu) | cx0 | (m << 3);
                        stack[sp++] = (rc & ~0x3fu) | (m + 1) | (cx1 << 3);
                    } else {
                        const int m = (cy0 + cy1) >> 1;
                        stack[sp++] = (rc & ~0xfc0u) | (cy0 << 6) | (m << 9);
                        stack[sp++] = (rc & ~0xfc0u) | ((m + 1) << 6) | (cy1 << 9);
                    }
                } else {   // one cell: one item per window row of its 3x3 neighbourhood
                    for (int dr = 0; dr < 3; ++dr)
                        stack[sp++] = (rc & 0xfffu) | ((unsigned)(cy0 + dr) << 12) | ((unsigned)(cy0 + dr) << 16);
                }
            }
            nRects = n;
        }
        __syncthreads();
    }
    __syncthreads();
    const unsigned nR = nRects;
    if (nR == 0) return;   // uniform
    const bool hv = dense;   // class H is dequeued first: the rectangles of dense tiles are the long items
    unsigned step = dense ? max(wmax, (unsigned)(((unsigned long long)total * nR + denseItems - 1) / denseItems)) : wmax;
    step = max(step, (total + 31) / 32);
    unsigned nc = 0;
    for (int f = max(fa, 1); f < fb; ++f) nc += (pre[f] / step != pre[f - 1] / step) || (f % fmax == 0);
    unsigned cinc = nc;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(cinc, o);
        if (lane >= o) cinc += u;
    }
    __syncthreads();   // wsum is reused
    if (lane == 63) wsum[w] = cinc;
    __syncthreads();
    unsigned cpos = cinc - nc + 1, call = 1;   // cut 0 is frame 0
    for (int k = 0; k < PL_T / 64; ++k) {
        if (k < w) cpos += wsum[k];
        call += wsum[k];
    }
    if (tid == 0) cuts[0] = 0;
    for (int f = max(fa, 1); f < fb; ++f)
        if ((pre[f] / step != pre[f - 1] / step) || (f % fmax == 0)) {
            if (cpos < (unsigned)PL_ITEMS_PER_TILE) cuts[cpos] = (unsigned)f;
            ++cpos;
        }
    __syncthreads();
    unsigned n = min(call, (unsigned)PL_ITEMS_PER_TILE);
    if (tid == 0) {
        cuts[n] = (unsigned)nFrames;
        if (pre[nFrames] == pre[cuts[n - 1]]) --n;   // nothing after the last cut
        nCuts = n;
        slotBase = n ? atomicAdd(&ctrl[8 + (hv ? 0 : PL_SHARDS) + t % PL_SHARDS], n * nR) : 0u;
    }
    __syncthreads();
    n = nCuts;
    uint4 *dst = (hv ? itemsH : itemsL) + (size_t)(t % PL_SHARDS) * PL_SHARD_CAP + slotBase;
    for (unsigned idx = tid; idx < n * nR; idx += PL_T) {
        const unsigned k = idx / nR, rr = idx - k * nR;   // the rectangles of one frame cut are neighbours
        const unsigned f0 = cuts[k], f1 = (k + 1 == n) ? (unsigned)nFrames : cuts[k + 1];
        if (slotBase + idx < PL_SHARD_CAP) {
            dst[idx] = make_uint4((unsigned)t, f0, f1, rects[rr]);
            itemPts[((size_t)(hv ? 0 : 1) * PL_SHARDS + t % PL_SHARDS) * PL_SHARD_CAP + slotBase + idx] = pre[f1] - pre[f0];
        }
    }
}

__host__ __device__ __forceinline__ unsigned pp5_live_bytes(int T) { return 16u + 4u * (unsigned)((T + 1) >> 1); }

// ---- wave-autonomous join (MODEST_PP_FRAMES_PATH=gather-wave) ----------------------------------
// Workgroup-sized items were latency bound: chains of dependent round trips
// fenced by workgroup barriers, 16 wavefronts per CU.  Here ONE wavefront owns
// an item end to end: a tile (or a rectangle of its cells) x a frame range of about a thousand points
// x a range of window rows whose live points (<= W_CAP) and counters sit in a private LDS slice.
// There is no workgroup barrier; every wavefront of the chip works on its own item and their round
// trips overlap.
//   lanes = frames for the run look-up (64 at a time; run, pointer and pose stay in registers),
//   lanes = points for the gather (the owning frame's data comes through cross-lane reads; the
//           loads of batch b+1 are issued before batch b is transformed and joined),
//   lanes = records for the pair tests (every lane walks its own candidate list in the LDS slice).
// item.w = cx0 | cx1 << 3 | cy0 << 6 | cy1 << 9 | ry0 << 12 | ry1 << 16: the cells whose records the
// item joins and the window rows (0..9) whose live points it holds.
constexpr int W_WAVES = 4;   // wavefronts per workgroup (they never synchronise with each other)
constexpr int W_DEPTH = 4;   // batches of 64 points whose loads are in flight together
constexpr int W_FMAX = 384;  // frames per item (an item's frame list is walked 64 at a time; the cut keeps items short)
__host__ __device__ __forceinline__ unsigned pp6_slice_bytes(int T) {
    // live points + counters + cell counts (u16 [100]) + row tables (u16 [10][11], u32 [11]), 16-B multiple
    // ... + the queue of surviving records (128 x 16 B)
    return (((unsigned)W_CAP * pp5_live_bytes(T) + 200u + 220u + 44u + 15u) & ~15u) + 128u * 16u;
}

template <bool PROF>
__global__ __launch_bounds__(64 * W_WAVES, 3) void pp6_wave_join(const FrameDev *__restrict__ frames, int nFrames,
                                                                const uint2 *__restrict__ runs,
                                                                const uint4 *__restrict__ itemsH,
                                                                const uint4 *__restrict__ itemsL, unsigned *ctrl,
                                                                const uint2 *__restrict__ winTab, int LTX0, int LTY0,
                                                                const float4 *__restrict__ sorted, Map24 A, int *counts,
                                                                int T, double r2, unsigned sliceBytes,
                                                                unsigned long long *stats) {
    extern __shared__ __align__(16) unsigned char dynsm[];
    __shared__ unsigned shardEnd[W_WAVES][2 * PL_SHARDS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *slice = dynsm + (size_t)w * sliceBytes;
    const int Th = (T + 1) >> 1;
    float4 *live = reinterpret_cast<float4 *>(slice);
    unsigned *cntw = reinterpret_cast<unsigned *>(slice + (size_t)W_CAP * 16);
    unsigned short *ccnt = reinterpret_cast<unsigned short *>(slice + (size_t)W_CAP * pp5_live_bytes(T));
    unsigned short *ctab = ccnt + 100;
    unsigned *rowBase = reinterpret_cast<unsigned *>(ctab + 110);
    float4 *queue = reinterpret_cast<float4 *>(slice + ((W_CAP * pp5_live_bytes(T) + 464u + 15u) & ~15u));   // 128 records
    const float r2lo = (float)(r2 * (1.0 - 1e-6)), r2hi = (float)(r2 * (1.0 + 1e-6));
    unsigned long long tprof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define PP6_TICK(k)                                      \
    if (PROF) {                                          \
        const unsigned long long now_ = wall_clock64();  \
        tprof[k] += now_ - tlast;                        \
        tlast = now_;                                    \
    }
    {   // inclusive prefix of the shards' item counts, class H first (per wavefront copy: no workgroup barrier)
        const unsigned c0 = min(ctrl[8 + lane], PL_SHARD_CAP);
        unsigned inc = c0;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        shardEnd[w][lane] = inc;
    }
    static_assert(2 * PL_SHARDS == 64, "one lane per shard counter");
    __builtin_amdgcn_wave_barrier();
    const unsigned nItems = shardEnd[w][2 * PL_SHARDS - 1];
    if (PROF) tlast = wall_clock64();
    for (;;) {
        unsigned id = 0;
        if (lane == 0) id = atomicAdd(&ctrl[5], 1u);
        id = __builtin_amdgcn_readfirstlane(id);
        if (id >= nItems) break;
        int sh = 0;
        while (shardEnd[w][sh] <= id) ++sh;
        const uint4 *isrc = sh < PL_SHARDS ? itemsH + (size_t)sh * PL_SHARD_CAP : itemsL + (size_t)(sh - PL_SHARDS) * PL_SHARD_CAP;
        const uint4 it = isrc[id - (sh ? shardEnd[w][sh - 1] : 0u)];
        const int ttx = (int)(it.x % F_NTF), tty = (int)(it.x / F_NTF);
        const int gtx = LTX0 + ttx, gty = LTY0 + tty;
        const int cx0 = (int)(it.w & 7u), cx1 = (int)((it.w >> 3) & 7u), cy0 = (int)((it.w >> 6) & 7u);
        const int cy1 = (int)((it.w >> 9) & 7u), ry0 = (int)((it.w >> 12) & 15u), ry1 = (int)((it.w >> 16) & 15u);
        // live sub-window: window rows max(cy0, ry0) .. min(cy1 + 2, ry1), window columns cx0 .. cx1 + 2
        const int wr0 = max(cy0, ry0), wr1 = min(cy1 + 2, ry1);
        // ---- window: cell counts and starts (lane = window cell, two rounds)
        unsigned gst0 = 0, gst1 = 0;
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int e = lane + 64 * rnd;
            unsigned c = 0, g = 0;
            if (e < F_W * F_W) {
                const int r = e / F_W, cc = e - r * F_W;
                if (r >= wr0 && r <= wr1 && cc >= cx0 && cc <= cx1 + 2) {
                    const uint2 wt2 = winTab[(size_t)it.x * (F_W * F_W) + e];   // written by pp5_plan
                    c = wt2.x;
                    g = wt2.y;
                }
                ccnt[e] = (unsigned short)c;
            }
            if (rnd == 0) gst0 = g;
            else gst1 = g;
        }
        __builtin_amdgcn_wave_barrier();
        // row tables, all lanes: ctab[r][cc] = live points of row r before column cc; rowBase[r]
        for (int e = lane; e < F_W * (F_W + 1); e += 64) {
            const int r = e / (F_W + 1), cc = e - r * (F_W + 1);
            unsigned run = 0;
            for (int k = 0; k < cc; ++k) run += ccnt[r * F_W + k];
            ctab[e] = (unsigned short)run;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane <= F_W) {
            unsigned run = 0;
            for (int r = 0; r < lane; ++r) run += ctab[r * (F_W + 1) + F_W];
            rowBase[lane] = run;
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned Lw = rowBase[F_W];   // <= W_CAP by the plan
        unsigned long long occ;
        {
            const int kx = lane & 7, ky = lane >> 3;
            unsigned sOr = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) sOr |= ccnt[(ky + dy) * F_W + kx + dx];
            occ = __ballot(sOr != 0u && kx >= cx0 && kx <= cx1 && ky >= cy0 && ky <= cy1);
        }
        if (!occ) {
            PP6_TICK(0)
            continue;
        }
        // The plan sizes items so that Lw <= W_CAP; a sub-window that cannot be split further (one
        // window row of a single cell's neighbourhood) is walked in slices of W_CAP live points.
        for (unsigned lb0 = 0; lb0 < Lw; lb0 += W_CAP) {
        const unsigned Ls = min((unsigned)W_CAP, Lw - lb0);
        // live points -> LDS, one lane per point; the global starts of the window cells travel through
        // the (still unused) counter area
        __builtin_amdgcn_wave_barrier();
        if (lane < F_W * F_W) cntw[lane] = gst0;
        if (lane + 64 < F_W * F_W) cntw[lane + 64] = gst1;
        __builtin_amdgcn_wave_barrier();
        float4 lv[(W_CAP + 63) / 64];
#pragma unroll
        for (int k = 0; k < (W_CAP + 63) / 64; ++k) {
            const unsigned e = lb0 + lane + 64 * k;
            lv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < lb0 + Ls) {
                int r = 0;
                while (e >= rowBase[r + 1]) ++r;
                const unsigned pos = e - rowBase[r];
                const unsigned short *rw = ctab + r * (F_W + 1);
                int cc = 0;
                while (pos >= rw[cc + 1]) ++cc;
                lv[k] = sorted[cntw[r * F_W + cc] + (pos - rw[cc])];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < (W_CAP + 63) / 64; ++k)
            if (lane + 64u * k < Ls) live[lane + 64 * k] = lv[k];
        for (unsigned e = lane; e < Ls * Th; e += 64) cntw[e] = 0;
        __builtin_amdgcn_wave_barrier();
        PP6_TICK(0)

        // ---- frames, 64 at a time
        const int f0 = (int)it.y, f1 = (int)it.z;
        unsigned sinceFlush = 0;
        auto flush = [&](bool clear) {
            for (unsigned e = lane; e < Ls * Th; e += 64) {
                const unsigned cw = cntw[e];
                if (cw) {
                    const unsigned p = e / Th, tp = (e - p * Th) * 2;
                    const size_t rowi = (size_t)__float_as_int(live[p].w) * T;
                    if (cw & 0xffffu) atomicAdd(&counts[rowi + tp], (int)(cw & 0xffffu));
                    if (cw >> 16) atomicAdd(&counts[rowi + tp + 1], (int)(cw >> 16));
                    if (clear) cntw[e] = 0;
                }
            }
        };
        unsigned qn = 0;   // queued records (uniform)
        // pair tests of the first m queued records: every lane walks the candidates of its record (the
        // live points of the 3x3 cells around it).  Measured against a wave-uniform all-pairs loop over the
        // whole sub-window (one broadcast read per candidate, segmented popcounts instead of atomics): the
        // per-lane walk wins, 349 vs 517 ms of summed wave time -- a light tile's window holds ~10x more
        // live points than a record has candidates.
        auto pairs = [&](unsigned m) {
            const bool has = (unsigned)lane < m;
            const float4 rq = has ? queue[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int pk = has ? __float_as_int(rq.w) : 0;
            const int key = pk & (F_NC - 1);
            const unsigned trv = (unsigned)pk >> 16;
            const float hx = rq.x, hy = rq.y, hz = rq.z;
            const int lx = (key & (F_TS - 1)) + 1, ly = key / F_TS + 1;
            const unsigned short *row = ctab + (ly - 1) * (F_W + 1) + lx - 1;
            const unsigned c00 = row[0], c10 = row[F_W + 1], c20 = row[2 * (F_W + 1)];
            const unsigned n0 = row[3] - c00, n1 = row[F_W + 4] - c10, n2 = row[2 * (F_W + 1) + 3] - c20;
            const unsigned a0 = rowBase[ly - 1] + c00, n01 = n0 + n1, nAll = n01 + n2;
            const unsigned b1 = rowBase[ly] + c10 - n0, b2 = rowBase[ly + 1] + c20 - n01;
            const unsigned own = has ? nAll : 0u;
            const unsigned cword = trv >> 1, cinc = 1u << ((trv & 1u) * 16);
            for (unsigned p0 = 0; __any(p0 < own); p0 += 2) {
#pragma unroll
                for (unsigned u = 0; u < 2; ++u) {
                    const unsigned p = p0 + u;
                    const unsigned ca = p + (p < n0 ? a0 : (p < n01 ? b1 : b2)) - lb0;   // index inside the slice
                    const bool act = p < own && ca < Ls;
                    const unsigned ci = act ? ca : 0u;
                    const float4 q = live[ci];
                    const float fx = q.x - hx, fy = q.y - hy, fz = q.z - hz;
                    const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                    bool hit = act && d2 < r2lo;
                    if (act && !hit && d2 <= r2hi) hit = pp_within(hx, hy, hz, q.x, q.y, q.z, r2);   // exact re-test
                    if (hit) atomicAdd(&cntw[ci * Th + cword], cinc);
                }
            }
        };
        for (int c0 = f0; c0 < f1; c0 += 64) {
            const int f = c0 + lane;
            unsigned rstart = 0, rlen = 0, plo = 0, phi = 0;
            int tf = 0;
            float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
            if (f < f1) {
                const uint2 rn = runs[(size_t)it.x * nFrames + f];
                rstart = rn.x;
                rlen = rn.y;
                const float4 *q = reinterpret_cast<const float4 *>(frames + f);
                const float4 h0 = q[0];
                plo = __float_as_uint(h0.x);
                phi = __float_as_uint(h0.y);
                tf = __float_as_int(q[1].w);
                m0 = q[2];
                m1 = q[3];
                m2 = q[4];
            }
            unsigned inc = rlen;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            const unsigned pre = inc - rlen;   // exclusive
            const unsigned Pc = __builtin_amdgcn_readlane(inc, 63);
            PP6_TICK(1)
            // software pipeline: the raw point of batch b + 1 is in flight while batch b is processed
            auto issue = [&](unsigned b0, int &lo, bool &valid, float &x, float &y, float &z) {
                const unsigned i = b0 + lane;
                valid = i < Pc;
                lo = 0;   // largest lane whose prefix <= i
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int cand = lo + step;
                    const unsigned p = __shfl(pre, cand & 63);
                    if (cand < 64 && p <= i) lo = cand;
                }
                const unsigned sPre = __shfl(pre, lo), sStart = __shfl(rstart, lo);
                const unsigned sLo = __shfl(plo, lo), sHi = __shfl(phi, lo);
                x = y = z = 0.f;
                if (valid) {
                    const float *src = reinterpret_cast<const float *>(((unsigned long long)sHi << 32) | sLo) +
                                       3 * (size_t)(sStart + (i - sPre));
                    x = src[0];
                    y = src[1];
                    z = src[2];
                }
            };
            // the raw points of W_DEPTH batches are requested together: one round trip per W_DEPTH x 64
            // points (a wavefront that has 768 bytes in flight cannot hide a multi-microsecond latency)
            for (unsigned g0 = 0; g0 < Pc; g0 += 64u * W_DEPTH) {
                int blo[W_DEPTH];
                bool bvalid[W_DEPTH];
                float bx[W_DEPTH], by[W_DEPTH], bz[W_DEPTH];
#pragma unroll
                for (int k = 0; k < W_DEPTH; ++k) {
                    blo[k] = 0;
                    bvalid[k] = false;
                    bx[k] = by[k] = bz[k] = 0.f;
                    if (g0 + 64u * k < Pc) issue(g0 + 64u * k, blo[k], bvalid[k], bx[k], by[k], bz[k]);
                }
#pragma unroll
                for (int k = 0; k < W_DEPTH; ++k) {
                    if (g0 + 64u * k >= Pc) break;
                    const int lo = blo[k];
                    const bool valid = bvalid[k];
                    const float x = bx[k], y = by[k], z = bz[k];
                    if (sinceFlush > 60000u) {   // 16-bit counters
                        __builtin_amdgcn_wave_barrier();
                        if (qn) pairs(qn);
                        qn = 0;
                        __builtin_amdgcn_wave_barrier();
                        flush(true);
                        __builtin_amdgcn_wave_barrier();
                        sinceFlush = 0;
                    }
                    sinceFlush += 64;
                    const int sTf = __shfl(tf, lo);
                    float rel[12];
                    rel[0] = __shfl(m0.x, lo); rel[1] = __shfl(m0.y, lo); rel[2] = __shfl(m0.z, lo); rel[3] = __shfl(m0.w, lo);
                    rel[4] = __shfl(m1.x, lo); rel[5] = __shfl(m1.y, lo); rel[6] = __shfl(m1.z, lo); rel[7] = __shfl(m1.w, lo);
                    rel[8] = __shfl(m2.x, lo); rel[9] = __shfl(m2.y, lo); rel[10] = __shfl(m2.z, lo); rel[11] = __shfl(m2.w, lo);
                    bool keep = false;
                    float hx = 0.f, hy = 0.f, hz = 0.f;
                    int key = 0;
                    if (valid) {
                        const bool drop = ((sTf >> 16) & F_FLAG_CENTER) && in_center_box(x, y);
                        float o[3];
                        rel_apply(rel, x, y, z, o);
                        key = cell_in_tile(A, o[0], o[1], o[2], gtx, gty);
                        keep = !drop && ((occ >> key) & 1ULL);
                        hx = o[0];
                        hy = o[1];
                        hz = o[2];
                    }
                    // survivors join a wave-private queue; pair tests run on 64 queued records at a time, so
                    // that every lane has a record (a dense tile's rectangle keeps few points of a batch)
                    {
                        const unsigned long long bal = __ballot(keep);
                        if (keep) {
                            const unsigned trv = (unsigned)sTf & 0xffffu;
                            queue[qn + __popcll(bal & ((1ULL << lane) - 1ULL))] =
                                make_float4(hx, hy, hz, __int_as_float(key | (int)(trv << 16)));
                        }
                        qn += (unsigned)__popcll(bal);
                    }
                    PP6_TICK(2)
                    if (qn >= 64u) {
                        __builtin_amdgcn_wave_barrier();
                        pairs(64u);
                        __builtin_amdgcn_wave_barrier();
                        const float4 mv = (lane + 64u < qn) ? queue[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
                        __builtin_amdgcn_wave_barrier();
                        if (lane + 64u < qn) queue[lane] = mv;
                        qn -= 64u;
                        PP6_TICK(3)
                    }
                }
            }
        }
        if (qn) {
            __builtin_amdgcn_wave_barrier();
            pairs(qn);
            qn = 0;
            PP6_TICK(3)
        }
        __builtin_amdgcn_wave_barrier();
        flush(false);
        __builtin_amdgcn_wave_barrier();
        PP6_TICK(4)
        }   // live slices
    }
    if (PROF && lane == 0)
        for (int k = 0; k < 5; ++k) atomicAdd(&stats[8 + k], tprof[k]);
#undef PP6_TICK
}

// entropy of a count matrix (same arithmetic as pp_count.hip's kernel; duplicated so that the two
// translation units stay independent)
__device__ __forceinline__ double pp5_term(int c, double denom) {
    const double P = (double)c / denom;
    return (-P) * log(P + 1e-8);
}
__global__ void pp5_entropy_kernel(const int *__restrict__ counts, int n, int T, float *__restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int *c = counts + (size_t)i * T;
    long long s = 0;
    for (int t = 0; t < T; ++t) s += c[t];
    const double denom = (double)s + 1e-8;
    double res;
    if (T < 8) {
        res = 0.0;
        for (int t = 0; t < T; ++t) res += pp5_term(c[t], denom);
    } else {   // numpy's pairwise order for a run of <= 128 doubles
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pp5_term(c[j], denom);
        int t = 8;
        for (; t < T - (T % 8); t += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += pp5_term(c[t + j], denom);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; t < T; ++t) res += pp5_term(c[t], denom);
    }
    H[i] = (float)(res / log((double)T));
}

}  // namespace

extern "C" int modest_frame_table_tiles(void) { return F_NTF; }

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
        hj[k].n_inside = reinterpret_cast<int *>(ctx->scratch + arena_sz((size_t)n_jobs * sizeof(SortJob))) + k;
    }
    SortJob *dj = reinterpret_cast<SortJob *>(ctx->scratch);
    MODEST_HIP_CHECK(hipMemcpyAsync(dj, hj, (size_t)n_jobs * sizeof(SortJob), hipMemcpyHostToDevice, stream));
    const size_t lds = (size_t)(F_NTILE + 1) * 4;
    static bool attr = false;
    if (!attr) {
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(frame_sort_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    frame_sort_kernel<<<n_jobs, 1024, lds, stream>>>(dj);
    MODEST_HIP_CHECK(hipGetLastError());
    MODEST_HIP_CHECK(hipMemcpyAsync(hin, ctx->scratch + arena_sz((size_t)n_jobs * sizeof(SortJob)), (size_t)n_jobs * 4,
                                    hipMemcpyDeviceToHost, stream));
    MODEST_HIP_CHECK(hipStreamSynchronize(stream));
    if (n_inside_host)
        for (int k = 0; k < n_jobs; ++k) n_inside_host[k] = hin[k];
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
    const int N = live->n, T = n_trav;
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
    // Default: the V3 streaming kernels read the frames through the descriptor table with the pose
    // fused (pp_count.hip).  MODEST_PP_FRAMES_PATH=gather-wave selects the one-pass gather-join of this
    // file (wave-autonomous items; measured slower so far: DESIGN.md section 4.1).
    const char *path = getenv("MODEST_PP_FRAMES_PATH");
    if (!path || strcmp(path, "gather-wave") != 0)
        return modest_pp3_frames(ctx, live, live_perm_dev, frames, n_frames, n_trav, radius, counts_dev, H_dev, stream);

    const size_t shardItems = (size_t)PL_SHARDS * PL_SHARD_CAP;
    MODEST_REQUIRE(n_frames < (1 << 15), "too many frames");
    const size_t descBytes = arena_sz((size_t)(n_frames > 0 ? n_frames : 1) * sizeof(FrameDev));
    size_t need = descBytes + arena_sz((size_t)N * 16) + arena_sz((size_t)F_NTILE * F_NC * 8) +
                  arena_sz((size_t)F_NTILE * 8) + arena_sz((size_t)F_NTILE * 32) + arena_sz((size_t)F_NTILE * 100 * 8) +
                  arena_sz(1024) + 2 * arena_sz(shardItems * 16) + arena_sz(shardItems * 2 * 4) + arena_sz(64 * 8) +
                  arena_sz((size_t)F_NTILE * (n_frames > 0 ? n_frames : 1) * 8) + arena_sz((size_t)N * T * 4);
    int rc = modest_ctx_reserve(ctx, need);
    if (rc) return rc;
    Arena Ar(ctx->scratch);
    FrameDev *dframes = Ar.take<FrameDev>(n_frames > 0 ? n_frames : 1);
    float4 *sorted = Ar.take<float4>(N);
    uint2 *cellPk = Ar.take<uint2>((size_t)F_NTILE * F_NC);
    unsigned long long *tileOcc = Ar.take<unsigned long long>(F_NTILE);
    uint4 *tileEdge = Ar.take<uint4>((size_t)F_NTILE * 2);
    uint2 *winTab = Ar.take<uint2>((size_t)F_NTILE * 100);
    unsigned *ctrl = Ar.take<unsigned>(256);
    uint4 *itemsH = Ar.take<uint4>(shardItems);
    uint4 *itemsL = Ar.take<uint4>(shardItems);
    unsigned *itemPts = Ar.take<unsigned>(shardItems * 2);
    uint2 *runs = Ar.take<uint2>((size_t)F_NTILE * (n_frames > 0 ? n_frames : 1));
    unsigned long long *stats = Ar.take<unsigned long long>(64);
    int32_t *counts = counts_dev ? counts_dev : Ar.take<int32_t>((size_t)N * T);

    // descriptors travel through a ring of pinned staging slots (the copy is asynchronous)
    FrameDev *hslot = nullptr;
    rc = modest_ctx_stage_slot(ctx, (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(FrameDev),
                               reinterpret_cast<void **>(&hslot));
    if (rc) return rc;
    for (int f = 0; f < n_frames; ++f) {
        FrameDev &d = hslot[f];
        d.xyz = frames[f].xyz_dev;
        d.tab = frames[f].tab_dev;
        d.n = frames[f].n;
        d.TX0 = frames[f].TX0;
        d.TY0 = frames[f].TY0;
        d.trav_flags = frames[f].trav | (frames[f].flags << 16);
        for (int q = 0; q < 12; ++q) d.rel[q] = frames[f].rel[q];
    }
    if (n_frames > 0)
        MODEST_HIP_CHECK(hipMemcpyAsync(dframes, hslot, (size_t)n_frames * sizeof(FrameDev), hipMemcpyHostToDevice,
                                        stream));
    rc = modest_ctx_stage_commit(ctx, stream);
    if (rc) return rc;

    Map24 A;
    for (int q = 0; q < 8; ++q) A.a[q] = A8[q];
    Mat34f rel;
    for (int q = 0; q < 12; ++q) rel.m[q] = live->rel[q];
    const double r2 = radius * radius;

    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of one scan
    pp5_live_index<<<F_NTILE / LI_TILES, 256, 0, stream>>>(live->xyz_dev, live_perm_dev, live->tab_dev, live->TX0,
                                                           live->TY0, rel, A, sorted, cellPk, tileOcc, tileEdge, counts,
                                                           (size_t)N * T, ctrl);
    if (n_frames > 0) {
        const char *wm = getenv("MODEST_PP6_WMAX");
        const unsigned wmax = wm && atoi(wm) > 0 ? (unsigned)atoi(wm) : 768u;
        const char *dm = getenv("MODEST_PP6_DENSE_ITEMS");
        const unsigned denseItems = dm && atoi(dm) > 0 ? (unsigned)atoi(dm) : 256u;
        pp5_plan<<<F_NTILE, PL_T, (size_t)(n_frames + 1) * 4, stream>>>(dframes, n_frames, live->tab_dev, live->TX0,
                                                                        live->TY0, tileOcc, tileEdge, cellPk, wmax,
                                                                        denseItems, W_FMAX, runs, itemsH, itemsL, itemPts,
                                                                        winTab, ctrl);
        const char *pe = getenv("MODEST_PP5_PROF");
        const bool prof = pe && atoi(pe);
        if (prof) MODEST_HIP_CHECK(hipMemsetAsync(stats, 0, 64 * 8, stream));
        const unsigned slice = pp6_slice_bytes(T);
        static bool wattr = false;
        if (!wattr) {
            MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp6_wave_join<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2));
            MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp6_wave_join<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2));
            wattr = true;
        }
        MODEST_REQUIRE((size_t)slice * W_WAVES <= 80 * 1024, "too many traversals for the wave path's LDS slices");
        const char *wg = getenv("MODEST_PP6_WGS");
        const int wgs = (wg && atoi(wg) > 0 ? atoi(wg) : 4) * ctx->num_cus;
        if (prof)
            pp6_wave_join<true><<<wgs, 64 * W_WAVES, (size_t)slice * W_WAVES, stream>>>(
                dframes, n_frames, runs, itemsH, itemsL, ctrl, winTab, live->TX0, live->TY0, sorted, A, counts, T, r2, slice,
                stats);
        else
            pp6_wave_join<false><<<wgs, 64 * W_WAVES, (size_t)slice * W_WAVES, stream>>>(
                dframes, n_frames, runs, itemsH, itemsL, ctrl, winTab, live->TX0, live->TY0, sorted, A, counts, T, r2, slice,
                stats);
        if (prof) {
            unsigned long long hs[16];
            MODEST_HIP_CHECK(hipStreamSynchronize(stream));
            MODEST_HIP_CHECK(hipMemcpy(hs, stats, sizeof(hs), hipMemcpyDeviceToHost));
            unsigned sh[2 * PL_SHARDS];
            MODEST_HIP_CHECK(hipMemcpy(sh, ctrl + 8, sizeof(sh), hipMemcpyDeviceToHost));
            unsigned long long pts[2] = {0, 0}, mx[2] = {0, 0}, cnt[2] = {0, 0};
            std::vector<unsigned> buf(PL_SHARD_CAP);
            for (int c = 0; c < 2; ++c)
                for (int k = 0; k < PL_SHARDS; ++k) {
                    const unsigned n = sh[c * PL_SHARDS + k] < PL_SHARD_CAP ? sh[c * PL_SHARDS + k] : PL_SHARD_CAP;
                    cnt[c] += n;
                    if (!n) continue;
                    MODEST_HIP_CHECK(hipMemcpy(buf.data(), itemPts + ((size_t)c * PL_SHARDS + k) * PL_SHARD_CAP, n * 4,
                                               hipMemcpyDeviceToHost));
                    for (unsigned q = 0; q < n; ++q) {
                        pts[c] += buf[q];
                        if (buf[q] > mx[c]) mx[c] = buf[q];
                    }
                }
            fprintf(stderr, "[pp6] items: %llu of dense tiles (%llu points gathered, largest %llu) + %llu (%llu points, "
                            "largest %llu)\n", cnt[0], pts[0], mx[0], cnt[1], pts[1], mx[1]);
            fprintf(stderr, "[pp6] wave-time (10 ns ticks, summed over wavefronts) window %llu chunk-header %llu "
                            "gather+transform %llu pairs %llu flush %llu\n", hs[8], hs[9], hs[10], hs[11], hs[12]);
        }
    }
    modest_prof_mark(ctx, stream, 1);
    if (H_dev) pp5_entropy_kernel<<<(N + 255) / 256, 256, 0, stream>>>(counts, N, T, H_dev);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}
