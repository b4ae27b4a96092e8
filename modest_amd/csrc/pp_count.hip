// PP-score neighbour counting (reference: pre_compute_pp_score.py:54-60,188-193)
// and entropy (pre_compute_pp_score.py:68-75) for gfx950.
//
// Inversion of the reference's "KD-tree over 10.8 M history points, query with
// 30 k live points": the small live scan is indexed (cell-sorted), the history
// is streamed ONCE with coalesced 16-byte loads and rejected early against an
// LDS-resident dilated occupancy bitmap of the live scan.
//
// Measured on MI355X (Lyft shape, 10.8 M history points):
//   stream + bitmap test alone ............  25 us  (5.2 TB/s)
//   V1: survivors resolved against an L2-resident index, one global atomic per
//       pair ............................... 2090 us (630 us of divergent loads,
//                                            1430 us for 10.5 M device atomics)
// so V2 keeps the stream and moves the pair resolution and the counters to LDS:
//
//   route  (K1) stream once; survivors of the bitmap test are binned IN LDS by
//          32x32-cell tile (9.6 m) and written as ONE contiguous, tile-sorted
//          run per 4096-point chunk, plus an 8-byte run descriptor per
//          (tile, traversal, chunk);
//   tiles  (K2) a persistent grid dequeues balanced (tile, traversal, part)
//          work items; the tile's live points (+1 cell halo), a local cell
//          table and the counters live in LDS; a record is resolved with LDS
//          reads, a float32 pre-test with an exact float64 re-test inside a
//          1e-6 relative band around r^2, and LDS atomics; non-zero counters
//          are flushed with one global atomic each.
#include "pp_common.h"
#include <cmath>
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

constexpr int V2_TS = 32;                       // tile edge in cells
constexpr int V2_NT = PP_NX / V2_TS;            // 20 tiles per axis
constexpr int V2_NTILES = V2_NT * V2_NT;        // 400
constexpr int V2_CH = 4096;                     // history points per route chunk
constexpr int V2_LMAX = 6144;                   // live points (tile + halo) held in LDS
constexpr int V2_W = V2_TS + 2;                 // local tile width incl. halo (34)
static_assert(PP_NX % V2_TS == 0 && PP_NX == PP_NY, "tiling");

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

__global__ void pp_live_count(const float *__restrict__ live, int n, const unsigned *bb, double c,
                              unsigned *cellCount) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PPGrid g = pp_grid(bb, c);
    const int cx = pp_cell_coord(live[3 * (size_t)i], g.ox, g.inv_c, PP_NX);
    const int cy = pp_cell_coord(live[3 * (size_t)i + 1], g.oy, g.inv_c, PP_NY);
    atomicAdd(&cellCount[cy * PP_NX + cx], 1u);
}

// live points of every 34x34-cell tile window (tile + 1 cell halo) that pp2_tiles holds in LDS:
// one wavefront per tile, one lane per window row, from the prefix table
__global__ __launch_bounds__(64) void pp_tile_live(const unsigned *__restrict__ cellStart,
                                                   unsigned *__restrict__ tileLive) {
    const int tl = blockIdx.x, r = threadIdx.x;
    const int x0 = (tl % (PP_NX / 32)) * 32 - 1, gy = (tl / (PP_NX / 32)) * 32 - 1 + r;
    const int gx0 = max(x0, 0), gx1 = min(x0 + 34, PP_NX);
    unsigned v = 0;
    if (r < 34 && gy >= 0 && gy < PP_NY) v = cellStart[gy * PP_NX + gx1] - cellStart[gy * PP_NX + gx0];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (r == 0) tileLive[tl] = v;
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
__global__ __launch_bounds__(SCAN_BLOCK) void pp_scan_bitmap(const unsigned *__restrict__ cnt,
                                                             unsigned *__restrict__ start,
                                                             unsigned *__restrict__ blockSum,
                                                             unsigned *__restrict__ bitmap) {
    if (blockIdx.x < SCAN_NBLK) pp_scan_block(cnt, start, blockSum, (int)blockIdx.x);
    else pp_bitmap_row(cnt, bitmap, (int)blockIdx.x - SCAN_NBLK);
}

__global__ __launch_bounds__(SCAN_BLOCK) void pp_scan_finish(unsigned *__restrict__ start,
                                                             const unsigned *__restrict__ blockSum) {
    __shared__ unsigned red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned v = (tid < (int)blockIdx.x) ? blockSum[tid] : 0u;   // sums of the earlier blocks
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[w] = v;
    __syncthreads();
    unsigned off = 0;
    for (int k = 0; k < 16; ++k) off += red[k];
    const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + tid;
    start[i] += off;
    if (blockIdx.x == SCAN_NBLK - 1 && tid == SCAN_BLOCK - 1) start[PP_NCELL] = off + blockSum[SCAN_NBLK - 1];
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

// ---- V2 route (K1) ---------------------------------------------------------------
struct ChunkMap {
    int cstart[PP_MAX_TRAV + 1];   // first chunk id of each traversal (chunks never straddle)
};

__global__ __launch_bounds__(1024) void pp2_route(const float *__restrict__ hist, TravOffsets tr,
                                                  ChunkMap cm, int nchunks, const unsigned *bb, double c,
                                                  const unsigned *__restrict__ bitmap,
                                                  float4 *__restrict__ rec, uint2 *__restrict__ desc,
                                                  unsigned *descCount, unsigned *descRecs, int T,
                                                  int maxDesc, int dbg) {
    __shared__ unsigned sbits[PP_BITWORDS];
    __shared__ float4 stage[V2_CH];
    __shared__ unsigned thist[V2_NTILES];
    __shared__ unsigned tbase[V2_NTILES + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < PP_BITWORDS; i += 1024) sbits[i] = bitmap[i];
    const PPGrid g = pp_grid(bb, c);
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int t = 0;
        while (t + 1 < tr.n && chunk >= cm.cstart[t + 1]) ++t;
        const long long p0 = tr.off[t] + (long long)(chunk - cm.cstart[t]) * V2_CH;
        const long long pend = min(tr.off[t + 1], p0 + V2_CH);
        if (tid < V2_NTILES) thist[tid] = 0;
        __syncthreads();   // also orders the bitmap load before its first use
        const long long q0 = p0 + 4LL * tid;
        float v[12];
        const float *src = hist + 3 * q0;
        if (q0 + 4 <= pend && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            const float4 a = s4[0], b = s4[1], d = s4[2];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = d.x; v[9] = d.y; v[10] = d.z; v[11] = d.w;
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = (q0 + k / 3 < pend) ? src[k] : 0.f;
        }
        int tile[4], rank[4], lcell[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rank[k] = -1;
            tile[k] = 0;
            lcell[k] = 0;
            if (q0 + k < pend) {
                const int cx = pp_cell_coord(v[3 * k], g.ox, g.inv_c, PP_NX);
                const int cy = pp_cell_coord(v[3 * k + 1], g.oy, g.inv_c, PP_NY);
                const int bit = cy * PP_NX + cx;
                if ((sbits[bit >> 5] >> (bit & 31)) & 1u) {
                    tile[k] = (cy / V2_TS) * V2_NT + (cx / V2_TS);
                    lcell[k] = ((cy % V2_TS) << 8) | (cx % V2_TS);   // cell inside the tile, reused by pp2_tiles
                    rank[k] = (int)atomicAdd(&thist[tile[k]], 1u);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {   // one wavefront scans the 400 tile counters (7 per lane)
            unsigned loc[7], s = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid * 7 + j;
                loc[j] = (i < V2_NTILES) ? thist[i] : 0u;
                s += loc[j];
            }
            unsigned inc = s;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(inc, o);
                if (tid >= o) inc += u;
            }
            unsigned run = inc - s;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid * 7 + j;
                if (i < V2_NTILES) tbase[i] = run;
                run += loc[j];
            }
            if (tid == 63) tbase[V2_NTILES] = inc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (rank[k] >= 0)
                stage[tbase[tile[k]] + rank[k]] = make_float4(v[3 * k], v[3 * k + 1], v[3 * k + 2], __int_as_float(lcell[k]));
        __syncthreads();
        const unsigned total = tbase[V2_NTILES];
        float4 *dst = rec + (size_t)chunk * V2_CH;
        if (!(dbg & 64))
            for (unsigned i = tid; i < total; i += 1024) dst[i] = stage[i];
        if (tid < V2_NTILES && thist[tid] > 0 && !(dbg & 128)) {
            const unsigned list = (unsigned)tid * T + t;
            const unsigned d = atomicAdd(&descCount[list], 1u);
            atomicAdd(&descRecs[list], thist[tid]);
            desc[(size_t)list * maxDesc + d] = make_uint2((unsigned)chunk * V2_CH + tbase[tid], thist[tid]);
        }
        __syncthreads();
    }
}

// ---- V2 work list -------------------------------------------------------------------
// items[i]   = (list, part | nparts << 16): a (tile, traversal) list split into parts of at
//              most V2_ITEM_RECS records (one LDS chunk of pp2_tiles);
// entries    = runs of consecutive items of ONE tile with bounded estimated cost: the dequeue
//              unit of pp2_tiles (entryBegin[e] .. entryBegin[e+1]), at most 64 items each;
// cost model = records x (32 + live points of the tile): candidates per record grow with the
//              local live density (a record count alone left the dense centre tiles 10x
//              heavier than the rest).
// ctrl[0] = #items, ctrl[1] = dequeue head, ctrl[2] = #entries.
// One workgroup; every list is owned by one thread (blocked assignment), so after the single
// coalesced read of descRecs everything stays in registers / LDS: the kernel is a handful of
// dependent global round trips instead of one per phase and per 1024 lists.
constexpr int V2_WL_LPT = 8;                        // lists per thread: up to 8192 lists (20 traversals)
constexpr unsigned V2_ITEM_RECS = 4096;

__device__ __forceinline__ unsigned long long pp2_block_scan64(unsigned long long v, unsigned long long *sh,
                                                               int tid, unsigned long long *total) {
    const int lane = tid & 63, w = tid >> 6;   // inclusive, 1024 threads, two barriers
    unsigned long long inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();   // sh free
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
        const unsigned long long s = sh[k];
        if (k < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc;
}

__global__ __launch_bounds__(1024) void pp2_worklist(const unsigned *__restrict__ descCount,
                                                     const unsigned *__restrict__ descRecs,
                                                     const unsigned *__restrict__ tileLive, int nLists,
                                                     int T, int nWorkers, uint2 *__restrict__ items,
                                                     unsigned *__restrict__ entryBegin, unsigned maxItems,
                                                     unsigned *ctrl, int entryDiv) {
    __shared__ unsigned long long sh[16];
    __shared__ unsigned tileL[V2_NTILES];
    __shared__ unsigned long long lastPre[1024];   // cost prefix / tile of the last item owned by each thread
    __shared__ int lastTile[1024];
    const int tid = threadIdx.x;
    const int per = (nLists + 1023) / 1024;         // <= V2_WL_LPT (checked on the host)
    const int l0 = min(tid * per, nLists), l1 = min(l0 + per, nLists);
    unsigned recs[V2_WL_LPT], nd[V2_WL_LPT];
#pragma unroll
    for (int q = 0; q < V2_WL_LPT; ++q) {
        const int l = l0 + q;
        recs[q] = (q < per && l < l1) ? descRecs[l] : 0u;
        nd[q] = (q < per && l < l1) ? descCount[l] : 0u;
    }
    if (tid < V2_NTILES) tileL[tid] = tileLive[tid];   // live points per tile window (pp_live_count)
    __syncthreads();
    // per-list parts and costs
    unsigned long long totR = 0, totW = 0;
#pragma unroll
    for (int q = 0; q < V2_WL_LPT; ++q) {
        totR += recs[q];
        if (recs[q]) totW += (unsigned long long)recs[q] * (32u + tileL[(l0 + q) / T]);
    }
    unsigned long long sumR, sumW;
    pp2_block_scan64(totR, sh, tid, &sumR);
    pp2_block_scan64(totW, sh, tid, &sumW);
    unsigned long long itemT = V2_ITEM_RECS;
    {
        const unsigned long long floorT =
            (unsigned long long)((double)sumR / (double)(maxItems - (unsigned)nLists)) + 2ULL;
        if (itemT < floorT) itemT = floorT;   // never more than maxItems items
    }
    const unsigned itemT32 = (unsigned)min(itemT, 0x7fffffffULL);
    const unsigned long long entryT =
        (unsigned long long)((double)sumW / (double)((unsigned long long)nWorkers * (unsigned long long)entryDiv)) + 1ULL;
    // parts of list q and cost per part, recomputed where needed (keeps the register file small)
    // (64-bit integer division is a ~100-instruction software routine on this ISA: item counts use
    // 32-bit division, the per-part cost estimate uses a float division)
#define PP2_KK(q) ((recs[q] && nd[q]) ? max(1u, min((recs[q] + itemT32 - 1u) / itemT32, min(nd[q], 65535u))) : 0u)
#define PP2_WI(q, k) ((k) ? (unsigned long long)(((float)recs[q] * (float)(32u + tileL[(l0 + q) / T])) / (float)(k)) : 0ULL)
    unsigned long long myK = 0, myW = 0;
#pragma unroll
    for (int q = 0; q < V2_WL_LPT; ++q) {
        const unsigned k = PP2_KK(q);
        myK += k;
        myW += (unsigned long long)k * PP2_WI(q, k);
    }
    unsigned long long totK, totWW;
    const unsigned long long incK = pp2_block_scan64(myK, sh, tid, &totK);
    const unsigned long long incW = pp2_block_scan64(myW, sh, tid, &totWW);
    // items of this thread's lists; remember the last one for the neighbour's entry test
    unsigned it = (unsigned)(incK - myK);
    unsigned long long pre = incW - myW;
    int lt = -1;
    unsigned long long lp = 0;
#pragma unroll
    for (int q = 0; q < V2_WL_LPT; ++q) {
        const unsigned k = PP2_KK(q);
        const unsigned long long wi = PP2_WI(q, k);
        for (unsigned s = 0; s < k; ++s) {
            items[it++] = make_uint2((unsigned)(l0 + q), s | (k << 16));
            lt = (l0 + q) / T;
            lp = pre;
            pre += wi;
        }
    }
    lastTile[tid] = lt;
    lastPre[tid] = lp;
    __syncthreads();
    // predecessor of this thread's first item = last item of the nearest earlier thread that owns one
    int pt = -1;
    unsigned long long pp = 0;
    for (int b = tid - 1; b >= 0; --b)
        if (lastTile[b] >= 0) {
            pt = lastTile[b];
            pp = lastPre[b];
            break;
        }
    // entry starts: tile change, cost prefix crossing a multiple of entryT, or every 64th item
    const unsigned long long bound0 =
        ((unsigned long long)((double)pp / (double)entryT) + 1ULL) * entryT;   // one float64 division per thread
    unsigned nflag = 0;
    {
        unsigned i = (unsigned)(incK - myK);
        unsigned long long p2 = incW - myW;
        int ct = pt;
        unsigned long long nextB = bound0;   // first multiple of entryT above the predecessor's prefix
#pragma unroll
        for (int q = 0; q < V2_WL_LPT; ++q) {
            const unsigned k = PP2_KK(q);
            const unsigned long long wi = PP2_WI(q, k);
            const int tl = (l0 + q) / T;
            for (unsigned s = 0; s < k; ++s) {
                const bool cross = p2 >= nextB;
                while (p2 >= nextB) nextB += entryT;
                if (i == 0 || tl != ct || cross || (i & 63u) == 0u) ++nflag;
                ct = tl;
                p2 += wi;
                ++i;
            }
        }
    }
    unsigned long long totE;
    const unsigned long long incE = pp2_block_scan64(nflag, sh, tid, &totE);
    {
        unsigned e = (unsigned)(incE - nflag);
        unsigned i = (unsigned)(incK - myK);
        unsigned long long p2 = incW - myW;
        int ct = pt;
        unsigned long long nextB = bound0;
#pragma unroll
        for (int q = 0; q < V2_WL_LPT; ++q) {
            const unsigned k = PP2_KK(q);
            const unsigned long long wi = PP2_WI(q, k);
            const int tl = (l0 + q) / T;
            for (unsigned s = 0; s < k; ++s) {
                const bool cross = p2 >= nextB;
                while (p2 >= nextB) nextB += entryT;
                if (i == 0 || tl != ct || cross || (i & 63u) == 0u) entryBegin[e++] = i;
                ct = tl;
                p2 += wi;
                ++i;
            }
        }
#undef PP2_KK
#undef PP2_WI
    }
    if (tid == 0) {
        entryBegin[totE] = (unsigned)totK;
        ctrl[0] = (unsigned)totK;
        ctrl[1] = 0;
        ctrl[2] = (unsigned)totE;
    }
}

// ---- V2 tiles (K2) ------------------------------------------------------------------
// A workgroup dequeues one ENTRY (<= 64 consecutive work items of one tile).  Per entry
// there are only four dependent global round trips (entry -> items -> run descriptors ->
// records): the item headers and up to 1024 run descriptors at a time are staged in LDS,
// the tile's live points (+halo), its local cell table and the counters stay in LDS, and
// the counters are flushed when the (tile, traversal) list changes.
constexpr int V2_U = 4;      // records prefetched per thread
constexpr int V2_DC = 1024;  // run descriptors staged per chunk
constexpr int V2_EI = 64;    // max items per entry
constexpr unsigned V2_HEAVY = 64;   // more candidates than this: the whole wavefront helps

// exact float64 predicate, kept out of line so that the hot loop stays small
__device__ __noinline__ bool pp2_exact(float hx, float hy, float hz, float qx, float qy, float qz, double r2) {
    return pp_within(hx, hy, hz, qx, qy, qz, r2);
}

// One record per lane (invalid lanes have n = 0).  Lanes with up to `heavyT` candidates walk
// their own list four at a time: four independent LDS reads in flight, no branches in the
// loop body besides the predicated LDS add (the rare candidates inside the 1e-6 band around r^2 are queued
// in a bit mask and re-tested exactly in float64 afterwards).  Longer lists are shared by
// the whole wavefront.
__device__ __forceinline__ void pp2_resolve(const float4 *live, unsigned *cnt, float hx, float hy, float hz,
                                            unsigned a0, unsigned a1, unsigned a2, unsigned n0, unsigned n1,
                                            unsigned n, float r2lo, float r2hi, double r2, int lane,
                                            unsigned heavyT) {
    const unsigned n01 = n0 + n1;
    const unsigned b1 = a1 - n0, b2 = a2 - n01;
    const unsigned own = n > heavyT ? 0u : n;
    for (unsigned p0 = 0; __any(p0 < own); p0 += 4) {
        unsigned band = 0;
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const unsigned p = p0 + u;
            const bool act = p < own;
            const unsigned i = act ? p + (p < n0 ? a0 : (p < n01 ? b1 : b2)) : 0u;
            const float4 q = live[i];
            const float fx = q.x - hx, fy = q.y - hy, fz = q.z - hz;
            const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
            const bool hit = act && d2 < r2lo;
            band |= (act && !hit && d2 <= r2hi) ? (1u << u) : 0u;
            if (hit) atomicAdd(&cnt[i], 1u);
        }
        while (band) {   // practically never taken
            const unsigned u = __ffs((int)band) - 1;
            band &= band - 1;
            const unsigned p = p0 + u;
            const unsigned i = p + (p < n0 ? a0 : (p < n01 ? b1 : b2));
            const float4 q = live[i];
            if (pp2_exact(hx, hy, hz, q.x, q.y, q.z, r2)) atomicAdd(&cnt[i], 1u);
        }
    }
    unsigned long long heavy = __ballot(n > heavyT);
    while (heavy) {
        const int src = __ffsll((long long)heavy) - 1;
        heavy &= heavy - 1;
        const float sx = __shfl(hx, src), sy = __shfl(hy, src), sz = __shfl(hz, src);
        const unsigned sa0 = __shfl(a0, src), sb1 = __shfl(b1, src), sb2 = __shfl(b2, src);
        const unsigned sn0 = __shfl(n0, src), sn01 = __shfl(n01, src), sn = __shfl(n, src);
        for (unsigned p = lane; p < sn; p += 64) {
            const unsigned i = p + (p < sn0 ? sa0 : (p < sn01 ? sb1 : sb2));
            const float4 q = live[i];
            const float fx = q.x - sx, fy = q.y - sy, fz = q.z - sz;
            const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
            const bool hit = d2 < r2lo;
            if (hit) atomicAdd(&cnt[i], 1u);
            if (!hit && d2 <= r2hi && pp2_exact(sx, sy, sz, q.x, q.y, q.z, r2)) atomicAdd(&cnt[i], 1u);
        }
    }
}

// inclusive block scan (1024 threads) with two barriers; *total receives the block sum
__device__ __forceinline__ unsigned pp2_scan_u32(unsigned v, unsigned *wsum /* 16 */, int tid, unsigned *total) {
    const int lane = tid & 63, w = tid >> 6;
    unsigned inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();   // wsum free
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
        const unsigned s = wsum[k];
        if (k < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc;
}

__global__ __launch_bounds__(1024) void pp2_tiles(const float4 *__restrict__ rec,
                                                  const uint2 *__restrict__ desc,
                                                  const unsigned *__restrict__ descCount,
                                                  const uint2 *__restrict__ items,
                                                  const unsigned *__restrict__ entryBegin, unsigned *ctrl,
                                                  const unsigned *bb, double c,
                                                  const unsigned *__restrict__ cellStart,
                                                  const float4 *__restrict__ sorted, int *counts, int T,
                                                  int maxDesc, double r2, int dbg) {
    __shared__ float4 live[V2_LMAX];
    __shared__ unsigned cnt[V2_LMAX];
    __shared__ unsigned short ctab[V2_W * (V2_W + 1)];
    __shared__ unsigned segStart[V2_W], segLen[V2_W], rowBase[V2_W + 1];
    __shared__ unsigned dOff[V2_DC], dPre[V2_DC + 1];
    __shared__ unsigned iList[V2_EI], iDa[V2_EI], iBase[V2_EI + 1];
    __shared__ unsigned wsum[16];
    __shared__ unsigned s_item;
    __shared__ int s_L;
    const int tid = threadIdx.x, lane = tid & 63;
    const float r2lo = (float)(r2 * (1.0 - 1e-6)), r2hi = (float)(r2 * (1.0 + 1e-6));
    const unsigned nEntries = ctrl[2];
    const unsigned heavyT = (dbg >> 8) ? (unsigned)(dbg >> 8) : V2_HEAVY;
    int curTile = -1, L = 0;
    bool fits = false;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_item = atomicAdd(&ctrl[1], 1u);
        __syncthreads();
        const unsigned ent = s_item;
        if (ent >= nEntries) break;
        const unsigned ib = entryBegin[ent];
        const int nI = (int)(entryBegin[ent + 1] - ib);   // <= V2_EI
        if (tid < 64) {   // item headers -> descriptor ranges, one wavefront
            unsigned nd = 0, list = 0, da = 0;
            if (tid < nI) {
                const uint2 wi = items[ib + tid];
                list = wi.x;
                const unsigned part = wi.y & 0xffffu, nparts = wi.y >> 16;
                const unsigned nd_all = descCount[list];
                da = (unsigned)(((unsigned long long)nd_all * part) / nparts);
                nd = (unsigned)(((unsigned long long)nd_all * (part + 1)) / nparts) - da;
            }
            unsigned inc = nd;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(inc, o);
                if (tid >= o) inc += u;
            }
            iList[tid] = list;
            iDa[tid] = da;
            iBase[tid] = inc - nd;
            if (tid == 63) iBase[V2_EI] = inc;
        }
        __syncthreads();
        const unsigned ND = iBase[V2_EI];
        const int tile = (int)iList[0] / T;
        const int x0 = (tile % V2_NT) * V2_TS - 1, y0 = (tile / V2_NT) * V2_TS - 1;
        if (tile != curTile) {
            const int gx0 = max(x0, 0), gx1 = min(x0 + V2_W, PP_NX);
            if (tid < V2_W) {
                const int gy = y0 + tid;
                unsigned s = 0, e = 0;
                if (gy >= 0 && gy < PP_NY) {
                    s = cellStart[gy * PP_NX + gx0];
                    e = cellStart[gy * PP_NX + gx1];
                }
                segStart[tid] = s;
                segLen[tid] = e - s;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned run = 0;
                for (int r = 0; r < V2_W; ++r) {
                    rowBase[r] = run;
                    run += segLen[r];
                }
                rowBase[V2_W] = run;
                s_L = (int)run;
            }
            __syncthreads();
            L = s_L;
            fits = L <= V2_LMAX;
            if (fits) {
                for (int e = tid; e < V2_W * (V2_W + 1); e += 1024) {
                    const int r = e / (V2_W + 1), cc = e - r * (V2_W + 1);
                    const int gy = y0 + r;
                    unsigned val = rowBase[r];
                    if (gy >= 0 && gy < PP_NY) {
                        const int gx = min(max(x0 + cc, gx0), gx1);
                        val = cellStart[gy * PP_NX + gx] - segStart[r] + rowBase[r];
                    }
                    ctab[e] = (unsigned short)val;
                }
                for (int e = tid; e < L; e += 1024) {
                    int r = 0;
                    while (e >= (int)rowBase[r + 1]) ++r;
                    live[e] = sorted[segStart[r] + (e - rowBase[r])];
                }
                for (int e = tid; e < L; e += 1024) cnt[e] = 0;
            }
            curTile = tile;
        }
        int curList = -1, curT = 0;
        for (unsigned dc = 0; dc < ND; dc += V2_DC) {
            const unsigned ndc = min((unsigned)V2_DC, ND - dc);
            // stage run descriptors dc .. dc+ndc
            unsigned myCnt = 0;
            __syncthreads();   // previous chunk's dOff / dPre no longer read
            if ((unsigned)tid < ndc) {
                const unsigned gd = dc + tid;
                int lo = 0, hi = nI - 1;   // last k with iBase[k] <= gd
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (iBase[mid] <= gd) lo = mid; else hi = mid - 1;
                }
                const uint2 d = desc[(size_t)iList[lo] * maxDesc + iDa[lo] + (gd - iBase[lo])];
                dOff[tid] = d.x;
                myCnt = d.y;
            }
            unsigned totalRecs;
            const unsigned inc = pp2_scan_u32(myCnt, wsum, tid, &totalRecs);
            dPre[tid] = inc - myCnt;
            if (tid == 1023) dPre[V2_DC] = totalRecs;
            __syncthreads();
            // items that own descriptors of this chunk
            int k = 0;
            while (k + 1 < nI && iBase[k + 1] <= dc) ++k;
            for (; k < nI && iBase[k] < dc + ndc; ++k) {
                if (iBase[k + 1] == iBase[k]) continue;
                const int list = (int)iList[k];
                if (list != curList) {
                    __syncthreads();   // every record of the previous list is counted
                    if (fits)
                        for (int e = tid; e < L; e += 1024) {
                            const unsigned cN = cnt[e];
                            if (cN) {
                                if (!(dbg & 1))
                                    atomicAdd(&counts[(size_t)__float_as_int(live[e].w) * T + curT], (int)cN);
                                cnt[e] = 0;
                            }
                        }
                    __syncthreads();
                    curList = list;
                    curT = list - tile * T;
                }
                const unsigned dlo = max(iBase[k], dc) - dc, dhi = min(iBase[k + 1], dc + ndc) - dc;
                const unsigned r0 = dPre[dlo], r1 = (dhi == (unsigned)V2_DC) ? dPre[V2_DC] : dPre[dhi];
                // wave-uniform trip count: every lane takes part in the cooperative phase
                for (unsigned jb = r0 + (unsigned)(tid & ~63); jb < r1; jb += 1024 * V2_U) {
                    float4 hh[V2_U];
                    bool vv[V2_U];
#pragma unroll
                    for (int u = 0; u < V2_U; ++u) {   // issue all record loads before touching any
                        const unsigned j = jb + 1024u * u + lane;
                        vv[u] = j < r1;
                        hh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (vv[u]) {
                            int lo = (int)dlo, hi = (int)dhi - 1;   // last d with dPre[d] <= j
                            while (lo < hi) {
                                const int mid = (lo + hi + 1) >> 1;
                                if (dPre[mid] <= j) lo = mid; else hi = mid - 1;
                            }
                            hh[u] = (dbg & 4) ? make_float4(1.f, 1.f, 1.f, 0.f) : rec[dOff[lo] + (j - dPre[lo])];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < V2_U; ++u) {
                        if (jb + 1024u * u >= r1) break;   // wave-uniform
                        const float4 h = hh[u];
                        const bool valid = vv[u];
                        const int pk = __float_as_int(h.w);   // cell inside the tile, packed by pp2_route
                        if (fits) {
                            unsigned a0 = 0, a1 = 0, a2 = 0, n0 = 0, n1 = 0, n2 = 0;
                            if (valid && !(dbg & 2)) {
                                const int lcx = (pk & 255) + 1, lcy = (pk >> 8) + 1;   // in [1, V2_TS]
                                const unsigned short *row = ctab + (lcy - 1) * (V2_W + 1) + lcx - 1;
                                a0 = row[0];
                                n0 = row[3] - a0;
                                a1 = row[V2_W + 1];
                                n1 = row[V2_W + 4] - a1;
                                a2 = row[2 * (V2_W + 1)];
                                n2 = row[2 * (V2_W + 1) + 3] - a2;
                            }
                            pp2_resolve(live, cnt, h.x, h.y, h.z, a0, a1, a2, n0, n1, n0 + n1 + n2, r2lo, r2hi,
                                        r2, lane, heavyT);
                        } else if (valid) {   // tile too dense for LDS: resolve against the global index
                            const int cx = x0 + 1 + (pk & 255), cy = y0 + 1 + (pk >> 8);
                            const int xa = max(cx - 1, 0), xb = min(cx + 1, PP_NX - 1);
                            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, PP_NY - 1); ++yy) {
                                const unsigned a = cellStart[yy * PP_NX + xa], b = cellStart[yy * PP_NX + xb + 1];
                                for (unsigned i = a; i < b; ++i) {
                                    const float4 q = sorted[i];
                                    if (pp_within(h.x, h.y, h.z, q.x, q.y, q.z, r2))
                                        atomicAdd(&counts[(size_t)__float_as_int(q.w) * T + curT], 1);
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();   // flush the last list of the entry (counters return to zero)
        if (fits && curList >= 0)
            for (int e = tid; e < L; e += 1024) {
                const unsigned cN = cnt[e];
                if (cN) {
                    if (!(dbg & 1)) atomicAdd(&counts[(size_t)__float_as_int(live[e].w) * T + curT], (int)cN);
                    cnt[e] = 0;
                }
            }
    }
}

#include "pp_v3.h"

// ---- entropy ---------------------------------------------------------------
__device__ __forceinline__ double pp_term(int c, double denom) {
    const double P = (double)c / denom;
    return (-P) * log(P + 1e-8);
}

// numpy's pairwise summation order for a contiguous run of n <= 128 doubles
// (8 interleaved accumulators, then the remainder sequentially).
__global__ void pp_entropy_kernel(const int *__restrict__ counts, int n, int T,
                                  float *__restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int *c = counts + (size_t)i * T;
    long long s = 0;
    for (int t = 0; t < T; ++t) s += c[t];
    const double denom = (double)s + 1e-8;
    double res;
    if (T < 8) {
        res = 0.0;
        for (int t = 0; t < T; ++t) res += pp_term(c[t], denom);
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pp_term(c[j], denom);
        int t = 8;
        for (; t < T - (T % 8); t += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += pp_term(c[t + j], denom);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; t < T; ++t) res += pp_term(c[t], denom);
    }
    H[i] = (float)(res / log((double)T));
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

// `extra_bytes` of arena are reserved behind this call's own carve and
// returned in *extra (used by modest_pp_score for its private counts).
static int pp_count_impl(modest_ctx *ctx, const float *live, int n_live, const float *hist,
                         const int64_t *trav_offsets, int n_trav, double radius,
                         int32_t *counts, void *stream_, size_t extra_bytes, void **extra) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(n_live >= 0, "n_live < 0");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    TravOffsets tr;
    int rc = check_offsets(trav_offsets, n_trav, tr);
    if (rc) return rc;
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));

    // chunk map of the routed path (chunks never straddle traversals)
    ChunkMap cm;
    int nchunks = 0, maxDesc = 1;
    {
        long long nch = 0;
        for (int t = 0; t < n_trav; ++t) {
            cm.cstart[t] = (int)nch;
            const long long nc = (tr.off[t + 1] - tr.off[t] + V2_CH - 1) / V2_CH;
            nch += nc;
            if (nc > maxDesc) maxDesc = (int)nc;
        }
        MODEST_REQUIRE(nch < (1LL << 19), "history too large for the routed path");
        nchunks = (int)nch;
        cm.cstart[n_trav] = nchunks;
    }
    const int nLists = V2_NTILES * n_trav;
    const size_t maxItems = (size_t)nLists + 65536;
    int nwg3 = 2 * ctx->num_cus < V3_MAXWG ? 2 * ctx->num_cus : V3_MAXWG;
    if (nwg3 > nchunks) nwg3 = nchunks > 0 ? nchunks : 1;
    const char *sr_env = getenv("MODEST_PP_SLICE");
    unsigned sliceCap = sr_env ? (unsigned)atoi(sr_env) : V3_SLICE_MAX;
    sliceCap = sliceCap < 256u ? 256u : (sliceCap > V3_SLICE_MAX ? V3_SLICE_MAX : sliceCap);
    const size_t maxSlices = (size_t)V3_NL + (size_t)nchunks * V3_CH / 64 + 2;
    // one contiguous zero-initialised block: cellCount | fill | descCount | descRecs | ctrl[2] | bbox[4] | pad
    const size_t zero_words = (size_t)(PP_NCELL + 1) + PP_NCELL + 2 * (size_t)nLists + 8 + V2_NTILES + 4 + 3 * V3_NL + 2 * V3_DWORDS + V3_DMAX + V3_NBLK + 4 + 36;   // ... ctrl3[4] listTotal tileBase listLive dense denseBlock dbg   // ctrl[4] bbox[4] tileLive ctrl3[4]
    size_t need = arena_sz(zero_words * 4) + arena_sz((size_t)(PP_NCELL + 1) * 4) + arena_sz(SCAN_NBLK * 4) +
                  arena_sz(PP_BITWORDS * 4) + arena_sz((size_t)n_live * 16) +
                  arena_sz((size_t)nchunks * V2_CH * 16) + arena_sz((size_t)nLists * maxDesc * 8) +
                  arena_sz(maxItems * 8) + arena_sz((maxItems + 1) * 4) +
                  2 * arena_sz((size_t)nwg3 * V3_NL * 4) + arena_sz(maxSlices * 16);
    rc = modest_ctx_reserve(ctx, need + arena_sz(extra_bytes));
    if (rc) return rc;
    if (extra) {
        *extra = ctx->scratch + need;
        if (!counts) counts = static_cast<int32_t *>(*extra);
    }
    if (n_live == 0) return MODEST_OK;
    MODEST_REQUIRE(counts != nullptr, "counts is NULL");
    const long long m0 = tr.off[0], m1 = tr.off[n_trav];
    if (m1 == m0) {   // no history: all counts are zero
        MODEST_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n_live * n_trav * sizeof(int32_t), stream));
        return MODEST_OK;
    }
    MODEST_REQUIRE(live != nullptr && hist != nullptr, "NULL point buffer");
    Arena A(ctx->scratch);
    unsigned *zeroed = A.take<unsigned>(zero_words);
    unsigned *cellCount = zeroed;
    unsigned *fill = cellCount + (PP_NCELL + 1);
    unsigned *descCount = fill + PP_NCELL;
    unsigned *descRecs = descCount + nLists;
    unsigned *ctrl = descRecs + nLists;
    unsigned *bb = ctrl + 4;
    unsigned *tileLive = bb + 4;
    unsigned *ctrl3 = tileLive + V2_NTILES;
    unsigned *listTotal = ctrl3 + 4;
    unsigned *tileBase = listTotal + V3_NL;
    unsigned *listLive = tileBase + V3_NL;
    unsigned *dense = listLive + V3_NL;
    unsigned *denseBlock = dense + 2 * V3_DWORDS;
    unsigned *blockLive = denseBlock + V3_DMAX;
    unsigned *dbgStats = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(blockLive + V3_NBLK) + 7) & ~(uintptr_t)7);   // 16 x u64 (debug only)
    unsigned *cellStart = A.take<unsigned>(PP_NCELL + 1);
    unsigned *blockSum = A.take<unsigned>(SCAN_NBLK);
    unsigned *bitmap = A.take<unsigned>(PP_BITWORDS);
    float4 *sorted = A.take<float4>(n_live);
    float4 *rec = A.take<float4>((size_t)nchunks * V2_CH);
    uint2 *desc = A.take<uint2>((size_t)nLists * maxDesc);
    uint2 *items = A.take<uint2>(maxItems);
    unsigned *entryBegin = A.take<unsigned>(maxItems + 1);
    unsigned *wgTile = A.take<unsigned>((size_t)nwg3 * V3_NL);
    unsigned *wgOff = A.take<unsigned>((size_t)nwg3 * V3_NL);
    uint4 *slices = A.take<uint4>(maxSlices);

    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of one scan
    MODEST_HIP_CHECK(hipMemsetAsync(zeroed, 0, zero_words * 4, stream));
    const double c = radius * (1.0 + 1.0 / 1024.0);
    const double r2 = radius * radius;
    const int nb = (n_live + 255) / 256;
    pp_live_bbox<<<(n_live + 1023) / 1024, 256, 0, stream>>>(live, n_live, bb, counts, (size_t)n_live * n_trav);
    pp_live_count<<<nb, 256, 0, stream>>>(live, n_live, bb, c, cellCount);
    pp_scan_bitmap<<<SCAN_NBLK + PP_NY, SCAN_BLOCK, 0, stream>>>(cellCount, cellStart, blockSum, bitmap);
    pp_scan_finish<<<SCAN_NBLK, SCAN_BLOCK, 0, stream>>>(cellStart, blockSum);
    const char *var_env = getenv("MODEST_PP_VARIANT");
    int var = var_env ? atoi(var_env) : 3;
    if (var == 3 && n_trav > V3_MAXT) var = 2;
    if (var == 2 && nLists > 1024 * V2_WL_LPT) var = 1;   // > 20 traversals: beyond the work-list capacity, use the direct path
    if (var == 3)   // the extra blocks count the live points of every 8x8-cell block window
        pp3_scatter_blocklive<<<nb + (V3_NBLK + 255) / 256, 256, 0, stream>>>(live, n_live, bb, c, cellStart, fill,
                                                                             sorted, nb, blockLive);
    else
        pp_live_scatter<<<nb, 256, 0, stream>>>(live, n_live, bb, c, cellStart, fill, sorted);
    if (var == 1) {   // V1: per-point search in the L2-resident index (also kept for A/B measurements)
        pp_stream_v1<<<ctx->num_cus * 3, 256, 0, stream>>>(hist, m0, m1, tr, bb, c, bitmap, cellStart,
                                                           sorted, counts, n_trav, r2);
        modest_prof_mark(ctx, stream, 1);
        MODEST_HIP_CHECK(hipGetLastError());
        return MODEST_OK;
    }
    const char *dbg_env = getenv("MODEST_PP_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    if (var == 3) {
        ChunkMap3 cm3;
        for (int t = 0; t <= n_trav; ++t) cm3.cstart[t] = cm.cstart[t];
        static bool attr_done = false;
        if (!attr_done) {
            MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
            MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_join<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, V3_JOIN_LDS_DYN));
            MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pp3_scan),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 V3_MAXWG * V3_SCAN_L * 4));
            attr_done = true;
        }
        pp3_blocks<<<1, 1024, 0, stream>>>(blockLive, dense, denseBlock, listLive);
        pp3_stream<false><<<nwg3, 1024, 0, stream>>>(hist, tr, cm3, nchunks, bb, c, bitmap, dense, wgTile, wgOff,
                                                     tileBase, rec, dbg);
        pp3_scan<<<V3_NL / V3_SCAN_L, 1024, (size_t)nwg3 * V3_SCAN_L * 4, stream>>>(wgTile, wgOff, nwg3, listTotal);
        pp3_plan<<<1, 1024, 0, stream>>>(listTotal, listLive, n_trav, sliceCap, tileBase, slices,
                                         (unsigned)maxSlices, ctrl3);
        pp3_stream<true><<<nwg3, 1024, 0, stream>>>(hist, tr, cm3, nchunks, bb, c, bitmap, dense, wgTile, wgOff,
                                                    tileBase, rec, dbg);
        if (dbg & 8)
            pp3_join<true><<<2 * ctx->num_cus, V3_JT, V3_JOIN_LDS_DYN, stream>>>(
                rec, slices, ctrl3, denseBlock, cellStart, sorted, counts, n_trav, r2, dbg,
                reinterpret_cast<unsigned long long *>(dbgStats));
        else
            pp3_join<false><<<2 * ctx->num_cus, V3_JT, V3_JOIN_LDS_DYN, stream>>>(
                rec, slices, ctrl3, denseBlock, cellStart, sorted, counts, n_trav, r2, dbg,
                reinterpret_cast<unsigned long long *>(dbgStats));
        if (dbg & 8) {
            unsigned long long hs[16];
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
    const char *ed_env = getenv("MODEST_PP_ENTRY_DIV");
    const int entry_div = ed_env ? atoi(ed_env) : 4;
    pp_tile_live<<<V2_NTILES, 64, 0, stream>>>(cellStart, tileLive);
    const int grid1 = ctx->num_cus < nchunks ? ctx->num_cus : nchunks;
    pp2_route<<<grid1, 1024, 0, stream>>>(hist, tr, cm, nchunks, bb, c, bitmap, rec, desc, descCount,
                                          descRecs, n_trav, maxDesc, dbg);
    pp2_worklist<<<1, 1024, 0, stream>>>(descCount, descRecs, tileLive, nLists, n_trav, ctx->num_cus, items,
                                         entryBegin, (unsigned)maxItems, ctrl, entry_div);
    pp2_tiles<<<ctx->num_cus, 1024, 0, stream>>>(rec, desc, descCount, items, entryBegin, ctrl, bb, c, cellStart,
                                                 sorted, counts, n_trav, maxDesc, r2, dbg);
    modest_prof_mark(ctx, stream, 1);
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
