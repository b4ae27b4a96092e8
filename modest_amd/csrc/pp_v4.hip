// PP neighbour count, BLOCK path ("V4"): several consecutive scans of a shard in one call.
//
// Reference steps replaced: pre_compute_pp_score.py:132-150 (history stacking), :188-190 (cKDTree per
// traversal), :54-60 (count_neighbors) -- for a chain of scans at once.
//
// What the measurements of V3 (pp_v3.h) said: per scan it streams the 130 MB history twice (count +
// scatter), writes and re-reads 97 MB of survivor records and sorts every 4096-record slice in LDS before
// the pair phase -- yet consecutive scans of a Lyft shard share 35 of their 36 frames per traversal
// (data_preprocessing/lyft/split_traintest.py:64,97), so almost all of that work is repeated scan after
// scan.  The block path does the scan-independent part ONCE per chain of G scans:
//
//   * the frame store already keeps every frame sorted by the 8x8-cell tile of a WORLD lattice shared by all
//     frames (pp_frames.hip), with a prefix table per frame.  The size of every tile list of the UNION of the
//     chain's frames is therefore known before a single point is read (b4_counts / b4_lists / b4_bases:
//     a column sum over the frame tables) -- no count pass, no count matrix, no occupancy bitmap;
//   * b4_scatter streams the union frames once and copies every point, RAW (frame coordinates) plus its
//     frame slot and lattice cell, to its final position in its tile list (the position follows from the
//     tables: no atomics); tiles no scan of the chain has a live point near are skipped;
//   * b4_seg_hist / b4_seg_scan / b4_seg_scatter order every tile list by cell (counting sort over 4096-record
//     segments), so that the records of a cell are CONTIGUOUS in HBM for the whole chain;
//   * per scan: the live scan is cell-sorted on the same lattice (b4_live_*), b4_plan_* cuts the work into
//     items, and b4_join reads the records of the cells that have live points nearby straight into
//     registers -- no slice sort, no LDS staging of records -- applies the scan's own float32 pose to every
//     record (transform_points' rounding, per (scan, frame)) and tests it against the live points of the
//     3x3 cells around it.  A wavefront holds up to 512 records of ONE cell (eight per lane) against a
//     wave-uniform candidate: ballots + traversal-segmented popcounts, one LDS atomic per (candidate,
//     traversal).  Sparse cells (< 64 records) are packed 64 to a wavefront and walk their own candidates.
//
// Exactness: the lattice is only a conservative spatial filter.  Distances are evaluated exactly as V3
// does -- float32 pre-test, float64 re-test (scipy's predicate) inside a 1.5e-6 band around r^2 -- on
// coordinates produced by the reference's float32 chain from the scan's OWN relative poses.  Two points
// within r in a scan's common frame are within r + 2 * 1e-4 m on the lattice (the caller checks every
// pose against the lattice to 1e-4 m, frame_store.consistent), the lattice cell edge is at least
// r * (1 + 2^-9) (required below), so their lattice cells differ by at most one per axis.
#include "pp_frames.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace modest;

namespace {

constexpr int B4_NTF = MODEST_FRAME_NTF;   // tiles per axis of a frame table (128)
constexpr int B4_NTILE = B4_NTF * B4_NTF;
constexpr int B4_MAXW = 160;               // block window: at most this many tiles per axis
constexpr int B4_CH = 4096;                // points per streaming chunk (1024 threads x 4)
constexpr int B4_SEG = 4096;               // records per sort segment (512 threads x 8)
constexpr int B4_FG = 32;                  // frames per prefix group
constexpr int B4_QC = 16, B4_NC = B4_QC * B4_QC;   // a join item covers a QUAD of 2x2 tiles = 16x16 cells
constexpr int B4_W = B4_QC + 2, B4_W1 = B4_W + 1;   // window of a quad incl. halo (cells)
constexpr unsigned B4_HEAVY = 64;          // cells with at least this many records get tasks of their own
#ifndef B4_CPT_
#define B4_CPT_ 4
#endif
constexpr int B4_CPT = B4_CPT_;                  // 64-record chunks per task
constexpr unsigned B4_TASK = 64 * B4_CPT;  // 512 records
#ifndef B4_IT_
#define B4_IT_ 24
#endif
constexpr unsigned B4_IT = B4_IT_;             // tasks per item
#ifndef B4_WPE_
#define B4_WPE_ 4
#endif
#ifndef B4_JT_
#define B4_JT_ 256
#endif
constexpr int B4_JT = B4_JT_;                 // threads of a join workgroup
constexpr int B4_LDS_DYN = 30 * 1024;      // live points + counters of a band (4 workgroups per CU)
constexpr unsigned B4_LANE_MAX = 64;       // packed chunks: lanes with more candidates take the group loop
constexpr int B4_MAXT = 64;
constexpr int B4_POSE_LDS_MAX = 640;        // frames whose poses fit the LDS table of a join workgroup

struct UFrame {   // a frame of the union, device side (96 bytes)
    const float *xyz;
    const unsigned *tab;
    int n, TX0, TY0, flags;
    double lat[8];   // rows x, y of raw frame -> lattice cells (the map the frame was sorted with)
};
static_assert(sizeof(UFrame) == 96, "union frame layout");
static_assert(sizeof(modest_pp_block_frame) == 96 && sizeof(modest_pp_block_scan) == 192, "C ABI layout (frame_store.py mirrors it)");

struct PoseEnt {   // per (scan, union frame): 64 bytes = one cache line
    float rel[12];
    int trav;      // traversal of the frame in THIS scan; < 0: the frame is not part of the scan
    int pad[3];
};
static_assert(sizeof(PoseEnt) == 64, "pose entry layout");

struct Blk {   // block-wide device pointers and geometry (kernel argument)
    const UFrame *frames;
    const uint2 *chunkTab;
    unsigned *off, *gtot, *listTotal, *listBase, *segBase, *segList, *segHist, *segOff, *cellOff, *ctrl, *baseSum;
    uint2 *segRange;   // smallest / largest frame slot among a segment's records
    float4 *recA, *recB;
    int U, NG, nchunks, maxSegs;
    int BX0, BY0, BW, BH, BT, CW, CHc, NCpad, nScanBlk, G;
};

struct ScanDev {   // per scan (device table)
    const float *liveXyz;
    const unsigned *livePerm, *liveTab;
    unsigned *cellCount, *cellStart, *blockSum, *tileTasks, *ctrl;   // ctrl: [0] items, [1] queue head
    uint2 *cellRange;   // per (tile with tasks, cell): first record and count of the part of the cell this scan reads
    float4 *tmp, *sorted;
    uint4 *items;
    const PoseEnt *pose;
    int *counts;
    float *H;
    double lat[8];
    float rel[12];
    int n, TX0, TY0, T, maxItems, pad;
    int slotLo, slotHi;   // the scan's frames lie in [slotLo, slotHi] of the block's frame table
};

// lattice cell of a raw point: the arithmetic of frame_bin (pp_frames.hip) -- a frame's tile runs were
// made with exactly this chain, so the tile computed here is the run the point sits in
__device__ __forceinline__ bool b4_cell(const double *__restrict__ W, float x, float y, float z, long long *cx, long long *cy) {
    const double lx = fma(W[2], (double)z, fma(W[1], (double)y, W[0] * (double)x)) + W[3];
    const double ly = fma(W[6], (double)z, fma(W[5], (double)y, W[4] * (double)x)) + W[7];
    if (!(fabs(lx) < 1.0e9) || !(fabs(ly) < 1.0e9)) return false;
    *cx = (long long)floor(lx);
    *cy = (long long)floor(ly);
    return true;
}

// ---- list sizes from the frame tables ---------------------------------------------------------
// thread (block tile b, frame group g): exclusive prefix of the tile's point counts over the group's frames
__global__ __launch_bounds__(256) void b4_counts(Blk B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B.BT) return;
    const int g = blockIdx.y;
    const int gx = B.BX0 + b % B.BW, gy = B.BY0 + b / B.BW;
    const int f1 = min(B.U, (g + 1) * B4_FG);
    unsigned run = 0;
    for (int f = g * B4_FG; f < f1; ++f) {
        const UFrame &F = B.frames[f];   // wave-uniform: scalar loads
        const int lx = gx - F.TX0, ly = gy - F.TY0;
        unsigned c = 0;
        if (lx >= 0 && lx < B4_NTF && ly >= 0 && ly < B4_NTF) {
            const int k = ly * B4_NTF + lx;
            c = F.tab[k + 1] - F.tab[k];
        }
        B.off[(size_t)f * B.BT + b] = run;
        run += c;
    }
    B.gtot[(size_t)g * B.BT + b] = run;
}

// thread per block tile: is the tile needed (a live point of any scan in the 3x3 tiles around it)?  group
// bases; list total (0 for tiles nobody needs: their points are never written)
__global__ __launch_bounds__(256) void b4_lists(Blk B, const ScanDev *__restrict__ scans) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B.BT) return;
    const int gx = B.BX0 + b % B.BW, gy = B.BY0 + b / B.BW;
    bool needed = false;
#pragma unroll 4
    for (int s = 0; s < B.G; ++s) {
        const ScanDev &S = scans[s];
        const int l0 = max(gx - 1 - S.TX0, 0), l1 = min(gx + 1 - S.TX0, B4_NTF - 1);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {   // (the loads are unconditional on clamped indices: nothing waits inside a branch)
            const int ly = gy + dy - S.TY0;
            const bool in = S.n > 0 && ly >= 0 && ly < B4_NTF && l0 <= l1;
            const int row = in ? ly * B4_NTF : 0, a = in ? l0 : 0, e = in ? l1 + 1 : 0;
            const unsigned t0 = S.liveTab[row + a], t1 = S.liveTab[row + e];
            needed |= in && t1 > t0;
        }
    }
    unsigned run = 0;
    for (int g = 0; g < B.NG; ++g) {
        const unsigned t = B.gtot[(size_t)g * B.BT + b];
        B.gtot[(size_t)g * B.BT + b] = run;
        run += t;
    }
    B.listTotal[b] = needed ? run : 0u;
}

// list bases, segment bases, the segment -> list table; ctrl[0] = records, ctrl[1] = segments.  Two launches of a few
// workgroups (a tile per thread): local prefix sums + workgroup totals, then every workgroup adds the totals in front of it
// (at most 25 workgroups: a loop).  One workgroup walking all tiles took 41 us.
__global__ __launch_bounds__(1024) void b4_bases_local(Blk B) {
    __shared__ unsigned wa[16], wb[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x * 1024 + tid;
    const unsigned t = b < B.BT ? B.listTotal[b] : 0u, ns = (t + B4_SEG - 1) / B4_SEG;
    unsigned incA = t, incB = ns;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(incA, o), c = __shfl_up(incB, o);
        if (lane >= o) {
            incA += a;
            incB += c;
        }
    }
    if (lane == 63) {
        wa[w] = incA;
        wb[w] = incB;
    }
    __syncthreads();
    unsigned baseA = 0, baseB = 0, allA = 0, allB = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w) {
            baseA += wa[k];
            baseB += wb[k];
        }
        allA += wa[k];
        allB += wb[k];
    }
    if (b < B.BT) {
        B.listBase[b] = baseA + incA - t;   // (local to the workgroup until b4_bases_finish)
        B.segBase[b] = baseB + incB - ns;
    }
    if (tid == 0) {
        B.baseSum[2 * blockIdx.x] = allA;
        B.baseSum[2 * blockIdx.x + 1] = allB;
    }
}
__global__ __launch_bounds__(1024) void b4_bases_finish(Blk B) {
    const int tid = threadIdx.x;
    const int b = blockIdx.x * 1024 + tid;
    unsigned offA = 0, offB = 0;
    for (unsigned k = 0; k < blockIdx.x; ++k) {   // (wave-uniform: scalar loads)
        offA += B.baseSum[2 * k];
        offB += B.baseSum[2 * k + 1];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        B.ctrl[0] = offA + B.baseSum[2 * blockIdx.x];
        B.ctrl[1] = min(offB + B.baseSum[2 * blockIdx.x + 1], (unsigned)B.maxSegs);
    }
    if (b >= B.BT) return;
    const unsigned t = B.listTotal[b], ns = (t + B4_SEG - 1) / B4_SEG;
    const unsigned tB = B.listBase[b] + offA, sB = B.segBase[b] + offB;
    B.listBase[b] = tB;
    B.segBase[b] = sB;
    for (unsigned k = 0; k < ns; ++k)
        if (sB + k < (unsigned)B.maxSegs) B.segList[sB + k] = (unsigned)b;
}

// ---- the one pass over the union's points ------------------------------------------------------
// chunk = 4096 points of one frame (wave-uniform frame: scalar loads of its descriptor).  A point goes to
// listBase[tile] + (points of earlier frames in the tile) + (its rank in its frame's tile run).
__global__ __launch_bounds__(512) void b4_scatter(Blk B) {
    const int tid = threadIdx.x;
    for (int chunk = blockIdx.x; chunk < B.nchunks; chunk += gridDim.x) {
        const uint2 ct = B.chunkTab[chunk];
        const int f = (int)ct.x;
        const UFrame &F = B.frames[f];
        const int nin = (int)F.tab[B4_NTILE];   // points inside the frame's table (outliers are parked behind)
        const int g = f / B4_FG;
        // Four points per thread, every load issued whether or not the point turns out to be written (a lane without a
        // point reads the frame's first one; tiles outside the window read the table entries of tile 0): the kernel is two
        // dependent rounds of loads per point, and with the loads inside the `if`s of a point the compiler waited for each
        // round of each point in turn (185 us per block; 4 x 2 rounds in flight instead of 1).
#pragma unroll 1
        for (int half = 0; half < B4_CH / 2048; ++half) {   // (512 threads x 4 points; 68 registers: three workgroups per CU)
        float x[4], y[4], z[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (int)ct.y + (half * 4 + u) * 512 + tid;
            valid[u] = i < nin && i < (int)ct.y + B4_CH;
            const size_t ii = valid[u] ? (size_t)i : 0;   // (a chunk exists only for a frame with points)
            x[u] = F.xyz[3 * ii], y[u] = F.xyz[3 * ii + 1], z[u] = F.xyz[3 * ii + 2];
        }
        unsigned lt[4], lb[4], gt[4], of[4], tb[4];
        int key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long long cx = 0, cy = 0;
            const bool okc = b4_cell(F.lat, x[u], y[u], z[u], &cx, &cy);
            const long long tx = cx >> 3, ty = cy >> 3;
            const long long lx = tx - F.TX0, ly = ty - F.TY0;
            const long long bx = tx - B.BX0, by = ty - B.BY0;
            const bool in = okc && lx >= 0 && lx < B4_NTF && ly >= 0 && ly < B4_NTF   // (the first two cannot fail for i < nin)
                            && bx >= 0 && bx < B.BW && by >= 0 && by < B.BH;
            valid[u] = valid[u] && in;
            const int b = in ? (int)(by * B.BW + bx) : 0, k = in ? (int)(ly * B4_NTF + lx) : 0;
            lt[u] = B.listTotal[b];
            lb[u] = B.listBase[b];
            gt[u] = B.gtot[(size_t)g * B.BT + b];
            of[u] = B.off[(size_t)f * B.BT + b];
            tb[u] = F.tab[k];
            key[u] = (int)((cy & 7) * 8 + (cx & 7));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!valid[u] || lt[u] == 0u) continue;
            const int i = (int)ct.y + (half * 4 + u) * 512 + tid;
            const unsigned dest = lb[u] + gt[u] + of[u] + ((unsigned)i - tb[u]);
            // remove_center (pre_compute_pp_score.py:48-52,141-142) drops the point before the transform: a NaN
            // coordinate keeps its slot in the list and never passes a distance test
            float xs = x[u];
            if ((F.flags & F_FLAG_CENTER) && in_center_box(x[u], y[u])) xs = __int_as_float(0x7fc00000);
            B.recA[dest] = make_float4(xs, y[u], z[u], __int_as_float(key[u] | (f << 6)));
        }
        }
    }
}

// ---- tile lists -> cell order --------------------------------------------------------------------
__global__ __launch_bounds__(512) void b4_seg_hist(Blk B) {
    __shared__ unsigned hist[64], smin, smax;
    const unsigned seg = blockIdx.x;
    if (seg >= B.ctrl[1]) return;
    const int tid = threadIdx.x;
    if (tid < 64) hist[tid] = 0;
    if (tid == 64) smin = 0xffffffffu, smax = 0u;
    __syncthreads();
    unsigned mn = 0xffffffffu, mx = 0u;
    const unsigned b = B.segList[seg];
    const unsigned lo = B.listBase[b] + (seg - B.segBase[b]) * B4_SEG;
    const unsigned hi = min(B.listBase[b] + B.listTotal[b], lo + B4_SEG);
#pragma unroll
    for (int u = 0; u < B4_SEG / 512; ++u) {
        const unsigned i = lo + u * 512 + tid;
        if (i < hi) {
            const unsigned m = (unsigned)__float_as_int(B.recA[i].w);
            atomicAdd(&hist[m & 63], 1u);
            mn = min(mn, m >> 6);
            mx = max(mx, m >> 6);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    }
    if ((tid & 63) == 0) {
        atomicMin(&smin, mn);
        atomicMax(&smax, mx);
    }
    __syncthreads();
    if (tid < 64) B.segHist[(size_t)seg * 64 + tid] = hist[tid];
    if (tid == 0) B.segRange[seg] = make_uint2(smin, smax);   // (a tile list is in frame order: the ranges of its segments ascend)
}

// one wavefront per list, lane = cell: offsets of every (segment, cell) inside its cell run, cell bases
__global__ __launch_bounds__(256) void b4_seg_scan(Blk B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B.BT) return;
    const unsigned total = B.listTotal[b], base = B.listBase[b];
    const unsigned ns = (total + B4_SEG - 1) / B4_SEG, s0 = B.segBase[b];
    unsigned run = 0;
    for (unsigned k = 0; k < ns; ++k) {
        if (s0 + k >= (unsigned)B.maxSegs) break;
        const unsigned v = B.segHist[(size_t)(s0 + k) * 64 + lane];
        B.segOff[(size_t)(s0 + k) * 64 + lane] = run;
        run += v;
    }
    unsigned inc = run;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    B.cellOff[(size_t)b * 65 + lane] = base + inc - run;
    if (lane == 63) B.cellOff[(size_t)b * 65 + 64] = base + inc;
}

__global__ __launch_bounds__(512) void b4_seg_scatter(Blk B) {
    extern __shared__ __align__(16) unsigned char dynsm[];
    float4 *srec = reinterpret_cast<float4 *>(dynsm);   // the segment in cell order
    __shared__ unsigned cur[64], lstart[64], gbase[64];
    const unsigned seg = blockIdx.x;
    if (seg >= B.ctrl[1]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned b = B.segList[seg];
    const unsigned lo = B.listBase[b] + (seg - B.segBase[b]) * B4_SEG;
    const unsigned hi = min(B.listBase[b] + B.listTotal[b], lo + B4_SEG);
    const unsigned n = hi - lo;
    if (tid < 64) {
        const unsigned v = B.segHist[(size_t)seg * 64 + tid];
        unsigned inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        cur[tid] = inc - v;
        lstart[tid] = inc - v;
        gbase[tid] = B.cellOff[(size_t)b * 65 + tid] + B.segOff[(size_t)seg * 64 + tid];
    }
    float4 r[B4_SEG / 512];
#pragma unroll
    for (int u = 0; u < B4_SEG / 512; ++u) {
        const unsigned i = u * 512 + tid;
        r[u] = i < n ? B.recA[lo + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < B4_SEG / 512; ++u) {
        const unsigned i = u * 512 + tid;
        if (i < n) srec[atomicAdd(&cur[__float_as_int(r[u].w) & 63], 1u)] = r[u];
    }
    __syncthreads();
    for (unsigned p = tid; p < n; p += 512) {
        const float4 v = srec[p];
        const int key = __float_as_int(v.w) & 63;
        B.recB[gbase[key] + (p - lstart[key])] = v;
    }
}

// ---- live scans on the block's cells ----------------------------------------------------------
// grid.y = scan.  tmp[i] = (x, y, z in the scan's common frame, cell); the cell counters were cleared
__global__ __launch_bounds__(256) void b4_live_count(Blk B, const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    {   // the scan's counts start at zero (one launch for all scans instead of a memset per scan)
        const size_t nc = (size_t)S.n * S.T;
        for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < nc; c += (size_t)gridDim.x * 256) S.counts[c] = 0;
    }
    if (i >= S.n) return;
    const float x = S.liveXyz[3 * (size_t)i], y = S.liveXyz[3 * (size_t)i + 1], z = S.liveXyz[3 * (size_t)i + 2];
    float o[3];
    rel_apply(S.rel, x, y, z, o);
    long long cx = 0, cy = 0;
    (void)b4_cell(S.lat, x, y, z, &cx, &cy);
    cx -= 8LL * B.BX0;
    cy -= 8LL * B.BY0;
    // (a clean live frame lies inside its own table, which the block window contains: the clamp only keeps a
    // corrupt input from writing out of bounds)
    const int bx = (int)min(max(cx, 0LL), (long long)B.CW - 1), by = (int)min(max(cy, 0LL), (long long)B.CHc - 1);
    const int cell = by * B.CW + bx;
    atomicAdd(&S.cellCount[cell], 1u);
    S.tmp[i] = make_float4(o[0], o[1], o[2], __int_as_float(cell));
}

// exclusive scan of the cell counters (4096 per workgroup, four per thread); the counters stay: the scatter counts
// them back down to zero (its cursor), which saves a 4-byte store per cell of the window here
constexpr int B4_SCAN = 4096;
__global__ __launch_bounds__(1024) void b4_scan_local(Blk B, const ScanDev *__restrict__ scans) {
    __shared__ unsigned wsum[16];
    const ScanDev &S = scans[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint4 v = reinterpret_cast<const uint4 *>(S.cellCount)[(size_t)blockIdx.x * 1024 + tid];
    const unsigned s4 = v.x + v.y + v.z + v.w;
    unsigned inc = s4;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = inc - s4;
    for (int k = 0; k < w; ++k) base += wsum[k];
    reinterpret_cast<uint4 *>(S.cellStart)[(size_t)blockIdx.x * 1024 + tid] = make_uint4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
    if (tid == 1023) S.blockSum[blockIdx.x] = base + s4;
}
// The cell starts stay LOCAL to their 4096-cell scan block; this kernel (one workgroup per scan) turns the block sums into
// block offsets, and whoever reads a cell start adds the offset of the cell's block (b4_cs): no second pass over the
// window's 1.2 M cells (9.5 MB per scan less than adding the offsets in place).  blockSum[nScanBlk] = all live points.
__global__ __launch_bounds__(1024) void b4_scan_finish(Blk B, const ScanDev *__restrict__ scans) {
    __shared__ unsigned wsum[16];
    const ScanDev &S = scans[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned run = 0;   // (window of at most 160 x 160 tiles: 400 scan blocks; the loop is for larger constants)
    for (int base = 0; base < B.nScanBlk; base += 1024) {
        const int k = base + tid;
        const unsigned v = k < B.nScanBlk ? S.blockSum[k] : 0u;
        unsigned inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        __syncthreads();
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        unsigned off = run, all = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < w) off += wsum[q];
            all += wsum[q];
        }
        if (k < B.nScanBlk) S.blockSum[k] = off + inc - v;
        run += all;
    }
    if (tid == 0) {
        S.blockSum[B.nScanBlk] = run;
        S.cellStart[B.NCpad] = 0u;
    }
}
// start of cell `c` in the scan's cell-sorted live points
__device__ __forceinline__ unsigned b4_cs(const unsigned *__restrict__ cellStart, const unsigned *__restrict__ blockOff, size_t c) {
    return cellStart[c] + blockOff[c / B4_SCAN];
}
__global__ __launch_bounds__(256) void b4_live_scatter(Blk B, const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S.n) return;
    const float4 t = S.tmp[i];
    const int cell = __float_as_int(t.w);
    const unsigned slot = b4_cs(S.cellStart, S.blockSum, (size_t)cell) + (atomicSub(&S.cellCount[cell], 1u) - 1u);   // (the order inside a cell is free)
    S.sorted[slot] = make_float4(t.x, t.y, t.z, __int_as_float((int)S.livePerm[i]));
    // counts are indexed by the ORIGINAL point order of the live frame
}

// ---- plan ------------------------------------------------------------------------------------------
// A tile's work for one scan, in TASKS of at most 512 records: a cell with >= 64 records and live points in
// the 3x3 cells around it ("heavy") gets tasks of its own, ceil(chunks / 8) of them; the records of all other
// cells with live points nearby are packed ("light" virtual order) into tasks of 512.  One wavefront per
// tile, lane = cell.
__device__ __forceinline__ unsigned b4_cell_cand(const unsigned *__restrict__ cellStart, const unsigned *__restrict__ blockOff, int CW, int CHc,
                                                 int cx, int cy) {
    const int xa = max(cx - 1, 0), xb = min(cx + 1, CW - 1);
    unsigned c = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {   // (three rows, all loads issued: rows outside the window read the cell's own row and count nothing)
        const int yy = cy + dy;
        const bool in = yy >= 0 && yy < CHc;
        const size_t row = (size_t)(in ? yy : cy) * CW;
        const unsigned a = b4_cs(cellStart, blockOff, row + xa), e = b4_cs(cellStart, blockOff, row + xb + 1);
        c += in ? e - a : 0u;
    }
    return c;
}
__global__ __launch_bounds__(256) void b4_plan_tiles(Blk B, const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B.BT) return;
    unsigned th = 0, lv = 0;
    const unsigned total = B.listTotal[b];
    if (total != 0u) {
        // The part of every cell this scan reads.  The tile list is in the order of the block's frame table and its
        // segments were sorted one by one, so a cell's records are [run of segment 0 | run of segment 1 | ...] with ascending
        // frame slots from run to run: the runs of the segments that overlap [slotLo, slotHi] hold every record of the
        // scan's own frames (and, in the two boundary runs, some of the block's other scans' frames: masked per lane).
        const unsigned ns = (total + B4_SEG - 1) / B4_SEG, s0 = B.segBase[b];
        unsigned kf = ns, kl = 0;
        bool any = false;
        for (unsigned k = 0; k < ns; ++k) {   // (wave-uniform: scalar loads)
            const uint2 r = B.segRange[s0 + k];
            if (r.y >= (unsigned)S.slotLo && r.x <= (unsigned)S.slotHi) {
                kf = min(kf, k);
                kl = k;
                any = true;
            }
        }
        // (every load below is issued whether or not its value is used: the compiler waits for a load inside the branch
        // that holds it, and this kernel is nothing but dependent loads)
        if (!any) kf = 0, kl = 0;
        const unsigned base = B.cellOff[(size_t)b * 65 + lane], next = B.cellOff[(size_t)b * 65 + lane + 1];
        const unsigned oFirst = B.segOff[(size_t)(s0 + kf) * 64 + lane];
        const unsigned oEnd = B.segOff[(size_t)(s0 + min(kl + 1, ns - 1)) * 64 + lane];
        const int cx = (b % B.BW) * 8 + (lane & 7), cy = (b / B.BW) * 8 + (lane >> 3);
        const unsigned cand = b4_cell_cand(S.cellStart, S.blockSum, B.CW, B.CHc, cx, cy);
        const unsigned start = base + oFirst;
        const unsigned end = kl + 1 < ns ? base + oEnd : next;
        const unsigned n = any ? end - start : 0u;
        const bool active = n != 0u && cand != 0u;
        if (active) {
            if (n >= B4_HEAVY) th = (((n + 63) >> 6) + B4_CPT - 1) / B4_CPT;
            else lv = n;
        }
        for (int o = 32; o > 0; o >>= 1) {
            th += __shfl_xor(th, o);
            lv += __shfl_xor(lv, o);
        }
        if (th + lv) S.cellRange[(size_t)b * 64 + lane] = make_uint2(start, active ? n : 0u);   // (only tiles with work are read back)
    }
    if (lane == 0) S.tileTasks[b] = (th << 12) | lv;   // lv <= 64 * 63
}
// one workgroup per scan.  An item covers (a part of) a QUAD of 2x2 tiles: the per-item costs of the join (cell table,
// live window, barriers) are paid once for four tiles.  items = (quad, first task, end task); the full items first.
constexpr int B4_QPT = ((B4_MAXW / 2) * (B4_MAXW / 2) + 1023) / 1024;   // quads per thread
__device__ __forceinline__ unsigned b4_quad_tasks(const Blk &B, const unsigned *__restrict__ tileTasks, int qd, int QW) {
    const int qx = qd % QW, qy = qd / QW;
    unsigned th = 0, lv = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tx = 2 * qx + (s & 1), ty = 2 * qy + (s >> 1);
        if (tx < B.BW && ty < B.BH) {
            const unsigned w = tileTasks[ty * B.BW + tx];
            th += w >> 12;
            lv += w & 4095u;
        }
    }
    return th + (lv + B4_TASK - 1) / B4_TASK;
}
__global__ __launch_bounds__(1024) void b4_plan_items(Blk B, const ScanDev *__restrict__ scans) {
    __shared__ unsigned wa[16], wb[16];
    const ScanDev &S = scans[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int QW = (B.BW + 1) / 2, NQ = QW * ((B.BH + 1) / 2);
    unsigned sumF = 0, sumR = 0;
    for (int j = 0; j < B4_QPT; ++j) {
        const int qd = tid * B4_QPT + j;
        const unsigned t = qd < NQ ? b4_quad_tasks(B, S.tileTasks, qd, QW) : 0u;
        sumF += t / B4_IT;
        sumR += (t % B4_IT) ? 1u : 0u;
    }
    unsigned incA = sumF, incB = sumR;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(incA, o), c = __shfl_up(incB, o);
        if (lane >= o) {
            incA += a;
            incB += c;
        }
    }
    if (lane == 63) {
        wa[w] = incA;
        wb[w] = incB;
    }
    __syncthreads();
    unsigned baseA = 0, baseB = 0, allA = 0, allB = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w) {
            baseA += wa[k];
            baseB += wb[k];
        }
        allA += wa[k];
        allB += wb[k];
    }
    if (tid == 0) {
        S.ctrl[0] = min(allA + allB, (unsigned)S.maxItems);
        S.ctrl[1] = 0u;
    }
    unsigned fB = baseA + incA - sumF, rB = allA + baseB + incB - sumR;
    for (int j = 0; j < B4_QPT; ++j) {
        const int qd = tid * B4_QPT + j;
        if (qd >= NQ) break;
        const unsigned t = b4_quad_tasks(B, S.tileTasks, qd, QW), full = t / B4_IT;
        for (unsigned k = 0; k < full; ++k)
            if (fB + k < (unsigned)S.maxItems) S.items[fB + k] = make_uint4((unsigned)qd, k * B4_IT, (k + 1) * B4_IT, 0u);
        if ((t % B4_IT) && rB < (unsigned)S.maxItems) {
            S.items[rB] = make_uint4((unsigned)qd, full * B4_IT, t, 0u);
            ++rB;
        }
        fB += full;
    }
}

// ---- join ------------------------------------------------------------------------------------------
struct B4Shared {
    unsigned cst[B4_W * B4_W1];          // cellStart of the window cells
    unsigned short ctab[B4_W * B4_W1];   // live points of window row r before column cc
    unsigned segStart[B4_W], rowBase[B4_W1];   // rowBase: of the CURRENT band (LDS index of the first live point of window row r)
    unsigned colOff[B4_W];                     // ... live points of window row r left of the band
    unsigned recStart[B4_NC], recN[B4_NC];     // records of the quad's cells (cell c = row * 16 + column of the quad)
    unsigned thEnd[B4_NC];                     // one-cell tasks up to and including cell c
    unsigned lvStart[B4_NC], lvEnd[B4_NC];     // packed (light) order: first / end virtual record of cell c
    unsigned band[B4_NC];   // ya | yb << 5 | xa << 10 | xb << 15 | slow << 20 (cell rows / columns 1..16 of the window, inclusive)
    unsigned wsA[4], wsB[4];
    uint4 task[B4_IT];   // the item's one-cell tasks, decoded once: (cell, first record, end record, -)
    unsigned nBands, ticket, TH, LV, itemId, nextId;
    uint4 item, nextItem;
};

// traversal segment masks of one 64-record chunk: lane t keeps the lanes whose record belongs to traversal t
__device__ __forceinline__ void b4_segmask(unsigned trv, bool valid, int T, int lq, unsigned *lo, unsigned *hi) {
    const unsigned sel0 = (lq & 1) ? 0u : ~0u, sel1 = (lq & 2) ? 0u : ~0u;
    const unsigned sel2 = (lq & 4) ? 0u : ~0u, sel3 = (lq & 8) ? 0u : ~0u, sel4 = (lq & 16) ? 0u : ~0u;
    unsigned long long seg = __ballot(valid);
    const unsigned long long B0 = __ballot(trv & 1u), B1 = __ballot(trv & 2u);
    const unsigned long long B2 = __ballot(trv & 4u), B3 = __ballot(trv & 8u), B4 = __ballot(trv & 16u);
    const unsigned long long s0 = ((unsigned long long)sel0 << 32) | sel0;
    const unsigned long long s1 = ((unsigned long long)sel1 << 32) | sel1;
    const unsigned long long s2 = ((unsigned long long)sel2 << 32) | sel2;
    const unsigned long long s3 = ((unsigned long long)sel3 << 32) | sel3;
    const unsigned long long s4 = ((unsigned long long)sel4 << 32) | sel4;
    seg &= (B0 ^ s0) & (B1 ^ s1) & (B2 ^ s2) & (B3 ^ s3) & (B4 ^ s4);
    if (T > 32) {
        const unsigned sel5 = (lq & 32) ? 0u : ~0u;
        seg &= __ballot(trv & 32u) ^ (((unsigned long long)sel5 << 32) | sel5);
    }
    *lo = (unsigned)seg;
    *hi = (unsigned)(seg >> 32);
}

// pointers that reach a kernel through a device table are generic to the compiler (flat_load: counted against
// the LDS counter as well, so every LDS wait also waits for them); the join states that they are global
typedef float v4f __attribute__((ext_vector_type(4)));      // (HIP's float4 is a class: no address-space qualified copies)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define B4_GLOBAL(T) const __attribute__((address_space(1))) T *
template <typename T> __device__ __forceinline__ B4_GLOBAL(T) b4_global(const T *p) {
    return (B4_GLOBAL(T))(p);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// The pair phase of a one-cell task: 2 * NP chunks of 64 records in registers against the live points [ia, ie) of
// one window row (LDS, wave-uniform).  The chunks are held in PAIRS (v2f: chunk 2p in .x, chunk 2p+1 in .y), so that
// v_pk_add / v_pk_mul / v_pk_fma_f32 test a candidate against two chunks per instruction.  Counts the pairs with
// d2 < r2lo and re-tests the pairs inside the band [r2lo, r2hi] exactly (float64, scipy's predicate; practically
// never).  Lane t adds the hits of traversal t (segment masks sLo / sHi per chunk).
// popcount(x) + acc in ONE instruction (the compiler prefers independent popcounts and a three-operand add tree: three more
// VALU instructions per candidate in a loop that is bound by them)
__device__ __forceinline__ unsigned b4_bcnt(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
template <int NP>
__device__ __forceinline__ unsigned long long b4_pairs(const float4 *__restrict__ live, unsigned *__restrict__ cntw, unsigned ia,
                                                       unsigned ie, const v2f *hx, const v2f *hy, const v2f *hz,
                                                       const unsigned *sLo, const unsigned *sHi, float r2lo, float r2hi, int Th,
                                                       int lq, int T) {
    unsigned long long band = 0;
#pragma unroll 1
    for (unsigned i = ia; i < ie; ++i) {
        const float4 q = live[i];
        const v2f qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
        unsigned acc = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const v2f dx = qx - hx[p], dy = qy - hy[p], dz = qz - hz[p];
            v2f d2 = dx * dx;
            d2 = __builtin_elementwise_fma(dy, dy, d2);
            d2 = __builtin_elementwise_fma(dz, dz, d2);
            const unsigned long long hA = __ballot(d2.x < r2lo), mA = __ballot(d2.x <= r2hi);
            const unsigned long long hB = __ballot(d2.y < r2lo), mB = __ballot(d2.y <= r2hi);
            band |= (hA ^ mA) | (hB ^ mB);
            acc = b4_bcnt((unsigned)hA & sLo[2 * p], acc);
            acc = b4_bcnt((unsigned)(hA >> 32) & sHi[2 * p], acc);
            acc = b4_bcnt((unsigned)hB & sLo[2 * p + 1], acc);
            acc = b4_bcnt((unsigned)(hB >> 32) & sHi[2 * p + 1], acc);
        }
        if (lq < T && acc) atomicAdd(&cntw[i * Th + ((unsigned)lq >> 1)], acc << ((lq & 1) * 16));
    }
    return band;
}
// the pairs inside the band around r^2 (a separate pass over the task's candidates, entered practically never: its
// float64 temporaries must not live in the registers of the loop above)
template <int NP>
__device__ __forceinline__ void b4_pairs_band(const float4 *__restrict__ live, unsigned *__restrict__ cntw, unsigned ia, unsigned ie,
                                              const v2f *hx, const v2f *hy, const v2f *hz, const unsigned *sLo,
                                              const unsigned *sHi, float r2lo, float r2hi, double r2, int Th, int lq, int T) {
#pragma unroll 1
    for (unsigned i = ia; i < ie; ++i) {
        const float4 q = live[i];
        unsigned acc = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float fx = q.x - hx[p].x, fy = q.y - hy[p].x, fz = q.z - hz[p].x;
            const float dA = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
            fx = q.x - hx[p].y, fy = q.y - hy[p].y, fz = q.z - hz[p].y;
            const float dB = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
            const bool bA = !(dA < r2lo) && dA <= r2hi, bB = !(dB < r2lo) && dB <= r2hi;
            const unsigned long long xA = __ballot(bA && pp_within(hx[p].x, hy[p].x, hz[p].x, q.x, q.y, q.z, r2));
            const unsigned long long xB = __ballot(bB && pp_within(hx[p].y, hy[p].y, hz[p].y, q.x, q.y, q.z, r2));
            acc += __popc((unsigned)xA & sLo[2 * p]) + __popc((unsigned)(xA >> 32) & sHi[2 * p]);
            acc += __popc((unsigned)xB & sLo[2 * p + 1]) + __popc((unsigned)(xB >> 32) & sHi[2 * p + 1]);
        }
        if (lq < T && acc) atomicAdd(&cntw[i * Th + ((unsigned)lq >> 1)], acc << ((lq & 1) * 16));
    }
}
template <int NP>
__device__ __forceinline__ void b4_pairs_rows(const float4 *__restrict__ live, unsigned *__restrict__ cntw, const unsigned *aR,
                                              const unsigned *nR, const v2f *hx, const v2f *hy, const v2f *hz,
                                              const unsigned *sLo, const unsigned *sHi, float r2lo, float r2hi, double r2, int Th,
                                              int lq, int T) {
    unsigned long long band = 0;
#pragma unroll 1
    for (int rr = 0; rr < 3; ++rr) {
        const unsigned ia = __builtin_amdgcn_readfirstlane(aR[rr]);
        const unsigned ie = ia + __builtin_amdgcn_readfirstlane(nR[rr]);
        band |= b4_pairs<NP>(live, cntw, ia, ie, hx, hy, hz, sLo, sHi, r2lo, r2hi, Th, lq, T);
    }
    if (band) {
#pragma unroll 1
        for (int rr = 0; rr < 3; ++rr) {
            const unsigned ia = __builtin_amdgcn_readfirstlane(aR[rr]);
            const unsigned ie = ia + __builtin_amdgcn_readfirstlane(nR[rr]);
            b4_pairs_band<NP>(live, cntw, ia, ie, hx, hy, hz, sLo, sHi, r2lo, r2hi, r2, Th, lq, T);
        }
    }
}

// LDS pose table of a join workgroup (persistent: loaded once): 48 bytes of pose + 1 byte of traversal per frame
__host__ __device__ __forceinline__ unsigned b4_pose_bytes(int U) {
    return U <= B4_POSE_LDS_MAX ? (unsigned)(((U * 48 + 15) & ~15) + ((U + 15) & ~15)) : 0u;
}

template <bool LPOSE, bool PROF>
__global__ __launch_bounds__(B4_JT, B4_WPE_) void b4_join(Blk B, const ScanDev *__restrict__ scans, double r2, int dbg, unsigned long long *prof) {
    extern __shared__ __align__(16) unsigned char dynsm[];
    __shared__ B4Shared S;
    const int tid = threadIdx.x, lane = tid & 63;
    int lq = lane;
    asm volatile("" : "+v"(lq));   // (an opaque copy of the lane id: keeps the mask constants out of long-lived registers)
    const ScanDev &SC = scans[blockIdx.y];
    const int T = SC.T, Th = (T + 1) >> 1;
    const unsigned liveBytes = 16u + 4u * (unsigned)Th;
    const unsigned lcap = (unsigned)B4_LDS_DYN / liveBytes;
    const unsigned poseB = LPOSE ? b4_pose_bytes(B.U) : 0u;
    const float4 *poseL = reinterpret_cast<const float4 *>(dynsm);
    const signed char *travL = reinterpret_cast<const signed char *>(dynsm + ((B.U * 48 + 15) & ~15));
    float4 *live = reinterpret_cast<float4 *>(dynsm + poseB);
    unsigned *cntw = reinterpret_cast<unsigned *>(live + lcap);
    const float r2lo = (float)(r2 * (1.0 - 1e-6)), r2hi = (float)(r2 * (1.0 + 1e-6));
    B4_GLOBAL(unsigned) cellStart = b4_global(SC.cellStart);
    B4_GLOBAL(unsigned) blockOff = b4_global(SC.blockSum);   // (b4_scan_finish: offsets of the 4096-cell scan blocks)
    B4_GLOBAL(v4f) sorted = b4_global(reinterpret_cast<const v4f *>(SC.sorted));
    B4_GLOBAL(v4f) rec = b4_global(reinterpret_cast<const v4f *>(B.recB));
    B4_GLOBAL(v4f) pose = b4_global(reinterpret_cast<const v4f *>(SC.pose));
    B4_GLOBAL(unsigned) tileTasks = b4_global(SC.tileTasks);
    B4_GLOBAL(v2u) cellRange = b4_global(reinterpret_cast<const v2u *>(SC.cellRange));
    B4_GLOBAL(v4u) items = b4_global(reinterpret_cast<const v4u *>(SC.items));
    int *counts = SC.counts;
    const int CW = B.CW, CHc = B.CHc;
    const unsigned nItems = SC.ctrl[0];
    // PROF: cycles of this wavefront by phase (0 item set-up, 1 band set-up, 2 unit fetch + decode, 3 record wait + transform +
    // masks, 4 pair phase, 5 packed chunks, 6 wait at the end of a band, 7 flush), counts 8 items 9 bands 10 tasks 11 chunks
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = PROF ? clock64() : 0ULL;
#define B4_TICK(kk)                                  \
    if (PROF) {                                      \
        const unsigned long long now_ = clock64();   \
        pacc[kk] += now_ - plast;                    \
        plast = now_;                                \
    }

    if (LPOSE) {   // the scan's poses: read once per workgroup (the workgroups are persistent)
        float4 *pw = reinterpret_cast<float4 *>(dynsm);
        signed char *tw = reinterpret_cast<signed char *>(dynsm + ((B.U * 48 + 15) & ~15));
        for (int f = tid; f < B.U; f += B4_JT) {
            const v4f a = pose[4 * (size_t)f], b = pose[4 * (size_t)f + 1], c = pose[4 * (size_t)f + 2], d = pose[4 * (size_t)f + 3];
            pw[3 * f] = make_float4(a.x, a.y, a.z, a.w);
            pw[3 * f + 1] = make_float4(b.x, b.y, b.z, b.w);
            pw[3 * f + 2] = make_float4(c.x, c.y, c.z, c.w);
            tw[f] = (signed char)__float_as_int(d.x);
        }
    }
    // one record -> the scan's frame (transform_points' float32 chain); records of frames that are not part of the scan
    // and lanes without a record end up 1e30 away, *trv < 0
    auto xform = [&](const v4f R, bool valid, float *hx, float *hy, float *hz, int *trv) {
        const int slot = __float_as_int(R.w) >> 6;
        float4 p0, p1, p2;
        int t;
        if (LPOSE) {
            p0 = poseL[3 * slot], p1 = poseL[3 * slot + 1], p2 = poseL[3 * slot + 2];
            t = travL[slot];
        } else {
            const v4f a = pose[4 * (size_t)slot], b = pose[4 * (size_t)slot + 1], c = pose[4 * (size_t)slot + 2];
            p0 = make_float4(a.x, a.y, a.z, a.w), p1 = make_float4(b.x, b.y, b.z, b.w), p2 = make_float4(c.x, c.y, c.z, c.w);
            t = __float_as_int(pose[4 * (size_t)slot + 3].x);
        }
        float ax = R.x * p0.x;
        ax = fmaf(R.y, p0.y, ax);
        ax = fmaf(R.z, p0.z, ax);
        ax = ax + p0.w;
        float ay = R.x * p1.x;
        ay = fmaf(R.y, p1.y, ay);
        ay = fmaf(R.z, p1.z, ay);
        ay = ay + p1.w;
        float az = R.x * p2.x;
        az = fmaf(R.y, p2.y, az);
        az = fmaf(R.z, p2.z, az);
        az = az + p2.w;
        const bool ok = valid && t >= 0;
        *hx = ok ? ax : 1.0e30f;
        *hy = ok ? ay : 0.f;
        *hz = ok ? az : 0.f;
        *trv = ok ? t : -1;
    };
    // a record against the live points around its cell in GLOBAL memory (bands that do not fit the LDS: never on LiDAR)
    auto slow_walk = [&](float sx, float sy, float sz, int st, int cx, int cy) {
        const int xa = max(cx - 1, 0), xb = min(cx + 1, CW - 1);
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, CHc - 1); ++yy) {
            const size_t ia_ = (size_t)yy * CW + xa, ie_ = (size_t)yy * CW + xb + 1;
            const unsigned a = cellStart[ia_] + blockOff[ia_ / B4_SCAN], e = cellStart[ie_] + blockOff[ie_ / B4_SCAN];
            for (unsigned i = a; i < e; ++i) {
                const v4f qq = sorted[i];
                if (pp_within(sx, sy, sz, qq.x, qq.y, qq.z, r2)) atomicAdd(&counts[(size_t)__float_as_int(qq.w) * T + st], 1);
            }
        }
    };

    if (tid == 0) {
        const unsigned first = atomicAdd(&SC.ctrl[1], 1u);
        S.itemId = first;
        if (first < nItems) {
            const v4u w = items[first];
            S.item = make_uint4(w.x, w.y, w.z, w.w);
        }
    }
    for (;;) {
        __syncthreads();
        const unsigned iid = S.itemId;
        if (iid >= nItems) break;
        const uint4 it = S.item;   // (the header is rewritten at the end of the item only: behind the barriers below)
        if (tid == 0) S.nextId = atomicAdd(&SC.ctrl[1], 1u);   // in flight during the loads below
        const int qd = (int)it.x;
        const unsigned q0 = it.y, q1 = it.z;
        const int QW = (B.BW + 1) / 2;
        const int tbx = (qd % QW) * 2, tby = (qd / QW) * 2;   // first tile of the quad
        const int x0 = tbx * 8 - 1, y0 = tby * 8 - 1;
        const int gx0 = max(x0, 0), gx1 = min(x0 + B4_W, CW);
        // ---- (a) the window's cell table, the record offsets of the quad's cells -------------------
        for (int e = tid; e < B4_W * B4_W1; e += B4_JT) {
            const int r = e / B4_W1, cc = e - r * B4_W1;
            const int gy = y0 + r;
            unsigned val = 0;
            if (gy >= 0 && gy < CHc) {
                const size_t ci = (size_t)gy * CW + min(max(x0 + cc, gx0), gx1);
                val = cellStart[ci] + blockOff[ci / B4_SCAN];
            }
            S.cst[e] = val;
        }
        {
            static_assert(B4_JT == B4_NC, "one thread per cell of the quad");
            const int cxq = tid & 15, cyq = tid >> 4;
            const int tx = tbx + (cxq >> 3), ty = tby + (cyq >> 3);
            unsigned rs = 0, rn = 0;
            if (tx < B.BW && ty < B.BH) {
                const int bt = ty * B.BW + tx;
                // the plan wrote the ranges of the tiles that have work for this scan; both words are requested at once (the
                // range of a tile without work is whatever the arena held: read, not used -- one round trip instead of two)
                const unsigned tt = tileTasks[bt];
                const v2u r2_ = cellRange[(size_t)bt * 64 + (cyq & 7) * 8 + (cxq & 7)];
                rs = tt != 0u ? r2_.x : 0u;
                rn = tt != 0u ? r2_.y : 0u;
            }
            S.recStart[tid] = rs;
            S.recN[tid] = rn;
        }
        __syncthreads();
        B4_TICK(12)
        if (tid == 0 && S.nextId < nItems) {
            const v4u w = items[S.nextId];
            S.nextItem = make_uint4(w.x, w.y, w.z, w.w);
        }
        // ---- (b) tasks of the quad, window tables, bands ------------------------------------------
        unsigned thMine, lvMine, incA, incB;
        {
            const int lx = (tid & 15) + 1, ly = (tid >> 4) + 1;
            const unsigned n = S.recN[tid];
            unsigned cand = 0;
#pragma unroll
            for (int r = -1; r <= 1; ++r) cand += S.cst[(ly + r) * B4_W1 + lx + 2] - S.cst[(ly + r) * B4_W1 + lx - 1];
            const bool active = n > 0 && cand > 0, heavy = active && n >= B4_HEAVY;
            thMine = heavy ? (((n + 63) >> 6) + B4_CPT - 1) / B4_CPT : 0u;
            lvMine = (active && !heavy) ? n : 0u;
            incA = thMine, incB = lvMine;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned a = __shfl_up(incA, o), c = __shfl_up(incB, o);
                if (lane >= o) {
                    incA += a;
                    incB += c;
                }
            }
            if (lane == 63) {
                S.wsA[tid >> 6] = incA;
                S.wsB[tid >> 6] = incB;
            }
        }
        if (tid == 64) {   // (a lane of the second wavefront: next to the prefix sums and the window tables of the others)
            // bands: sub-rectangles of the quad whose live window (one cell of halo) fits the LDS budget.  Whole cell rows
            // first; a single row that does not fit (the dense rings next to the sensor) is cut into column halves,
            // quarters, ...; a single cell whose 3x3 neighbourhood does not fit takes the slow path.
            auto lenr = [&](int r) { return S.cst[r * B4_W1 + B4_W] - S.cst[r * B4_W1]; };   // live points of window row r
            auto cnt3 = [&](int ya, int xa, int xb) {   // live points of window rows ya-1..ya+1, columns xa-1..xb+1
                unsigned c = 0;
                for (int r = ya - 1; r <= ya + 1; ++r) c += S.cst[r * B4_W1 + xb + 2] - S.cst[r * B4_W1 + xa - 1];
                return c;
            };
            unsigned nb = 0;
            int ya = 1;
            while (ya <= B4_QC) {
                unsigned sum = lenr(ya - 1) + lenr(ya) + lenr(ya + 1);
                if (sum <= lcap) {
                    int yb = ya;
                    while (yb < B4_QC) {
                        const unsigned nx = lenr(yb + 2);
                        if (sum + nx > lcap) break;
                        sum += nx;
                        ++yb;
                    }
                    S.band[nb++] = (unsigned)ya | ((unsigned)yb << 5) | (1u << 10) | ((unsigned)B4_QC << 15);
                    ya = yb + 1;
                    continue;
                }
                int xa = 1, w = B4_QC / 2;   // pieces of width w (a power of two), left to right, as wide as fits
                while (xa <= B4_QC) {
                    while (w > 1 && (((xa - 1) & (w - 1)) != 0 || cnt3(ya, xa, xa + w - 1) > lcap)) w >>= 1;
                    const bool fits = cnt3(ya, xa, xa + w - 1) <= lcap;
                    S.band[nb++] = (unsigned)ya | ((unsigned)ya << 5) | ((unsigned)xa << 10) | ((unsigned)(xa + w - 1) << 15) |
                                   (fits ? 0u : 1u << 20);
                    xa += w;
                    w = B4_QC / 2;
                }
                ++ya;
            }
            S.nBands = nb;
        }
        for (int e = tid; e < B4_W * B4_W1; e += B4_JT) {
            const int r = e / B4_W1;
            S.ctab[e] = (unsigned short)min(S.cst[e] - S.cst[r * B4_W1], 65535u);
            if (e == r * B4_W1) S.segStart[r] = S.cst[e];
        }
        __syncthreads();
        B4_TICK(13)
        {
            unsigned bA = 0, bB = 0;
            for (int k = 0; k < (tid >> 6); ++k) {
                bA += S.wsA[k];
                bB += S.wsB[k];
            }
            S.thEnd[tid] = bA + incA;
            S.lvStart[tid] = bB + incB - lvMine;
            S.lvEnd[tid] = bB + incB;
            if (tid == B4_JT - 1) {
                S.TH = bA + incA;
                S.LV = bB + incB;
            }
        }
        __syncthreads();
        B4_TICK(14)
        const unsigned TH = S.TH, LV = S.LV;
        const unsigned nBands = S.nBands;
        // the item's units: one-cell tasks [qh0, qh1) and 64-record chunks of the packed order [l0, l1)
        const unsigned qh0 = min(q0, TH), qh1 = min(q1, TH), nH = qh1 - qh0;
        const unsigned l0 = (max(q0, TH) - TH) * B4_TASK, l1 = min(LV, (max(q1, TH) - TH) * B4_TASK);
        const unsigned nUnits = nH + (l1 > l0 ? (l1 - l0 + 63) / 64 : 0u);
        if (tid < (int)nH) {   // every task decoded once, in parallel: first cell whose task count (inclusive prefix) exceeds q
            const unsigned q = qh0 + tid;
            int lo = 0, hi = B4_NC - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (S.thEnd[mid] > q) hi = mid;
                else lo = mid + 1;
            }
            const unsigned cbeg = S.recStart[lo], cn = S.recN[lo];
            const unsigned nt = (((cn + 63) >> 6) + B4_CPT - 1) / B4_CPT;   // tasks of the cell
            const unsigned st = cbeg + (q - (S.thEnd[lo] - nt)) * B4_TASK;
            S.task[tid] = make_uint4((unsigned)lo, st, min(cbeg + cn, st + B4_TASK), 0u);
        }
        B4_TICK(0)
        if (PROF) ++pacc[8];
        for (unsigned bd = 0; bd < nBands; ++bd) {
            if (PROF) ++pacc[9];
            const unsigned bw = S.band[bd];
            const int ya = (int)(bw & 31u), yb = (int)((bw >> 5) & 31u);   // cell rows / columns of the band (window coordinates 1..16)
            const int xa = (int)((bw >> 10) & 31u), xb = (int)((bw >> 15) & 31u);
            const bool slow = (bw >> 20) != 0;
            if (bd > 0) __syncthreads();   // previous band's flush complete (the item's first band: the barriers of the set-up)
            if (tid <= yb - ya + 2) {   // rows ya-1 .. yb+1 of the window, columns xa-1 .. xb+1
                unsigned run = 0;
                for (int r = ya - 1; r < ya - 1 + tid; ++r) run += S.cst[r * B4_W1 + xb + 2] - S.cst[r * B4_W1 + xa - 1];
                const int r = ya - 1 + tid;
                S.rowBase[r] = run;
                S.colOff[r] = S.cst[r * B4_W1 + xa - 1] - S.cst[r * B4_W1];
            }
            __syncthreads();
            const unsigned Lb = slow ? 0u : S.rowBase[yb + 1] + (S.cst[(yb + 1) * B4_W1 + xb + 2] - S.cst[(yb + 1) * B4_W1 + xa - 1]);
            if (!slow) {
                for (unsigned e = tid; e < Lb; e += B4_JT) {
                    int r = ya - 1;
                    while (r < yb + 1 && e >= S.rowBase[r + 1]) ++r;
                    const v4f w = sorted[S.segStart[r] + S.colOff[r] + (e - S.rowBase[r])];
                    live[e] = make_float4(w.x, w.y, w.z, w.w);
                }
                for (unsigned e = tid; e < Lb * Th; e += B4_JT) cntw[e] = 0;
            }
            if (tid == 0) S.ticket = 0;
            __syncthreads();

            // ---- the tasks of the item, dealt to the wavefronts.  The records of a wavefront's NEXT one-cell task are
            // requested before the pair phase of the current one: the loads overlap the pair phase ----------------
            int ty = 0, k = 0;   // ty: 0 no unit left, 1 one-cell task (records requested), 2 a chunk of the packed order, 3 not in this band
            unsigned start = 0, end = 0, v0 = 0, v1 = 0;
            v4f R[B4_CPT];
            unsigned tnext = (unsigned)tid >> 6;   // units are dealt round robin to the wavefronts (no ticket: a returning LDS
            auto prep = [&]() {                     // atomic per unit is a latency of its own in a loop that is all latency)
                const unsigned t = __builtin_amdgcn_readfirstlane(tnext);
                tnext += B4_JT / 64;
                ty = 0;
                if (t >= nUnits) return;
                if (t >= nH) {   // a chunk of the packed order
                    ty = 2;
                    v0 = l0 + (t - nH) * 64u;
                    v1 = min(l1, v0 + 64u);
                    if (dbg & 2) v1 = v0;
                    return;
                }
                const uint4 tk4 = S.task[t];
                k = __builtin_amdgcn_readfirstlane((int)tk4.x);
                const int lcx = (k & 15) + 1, lcy = (k >> 4) + 1;
                ty = 3;
                if (lcy < ya || lcy > yb || lcx < xa || lcx > xb) return;
                ty = 1;
                start = __builtin_amdgcn_readfirstlane(tk4.y);
                end = __builtin_amdgcn_readfirstlane(tk4.z);
                if (dbg & 8) end = start;
                if (!slow && end > start) {   // (a lane without a record re-reads the last one: no load sits inside a branch)
#pragma unroll
                    for (int u = 0; u < B4_CPT; ++u) R[u] = __builtin_nontemporal_load(&rec[min(start + u * 64 + lane, end - 1)]);
                }
            };
            B4_TICK(1)
            prep();
            B4_TICK(2)
            while (ty != 0) {
                if (ty == 1) {
                    const int lcx = (k & 15) + 1, lcy = (k >> 4) + 1;
                    if (end <= start) {
                        prep();
                        continue;
                    }
                    if (slow) {
                        const int cx = tbx * 8 + (k & 15), cy = tby * 8 + (k >> 4);
                        for (unsigned ib = start; ib < end; ib += 64) {
                            float sx, sy, sz;
                            int st;
                            xform(rec[min(ib + lane, end - 1)], ib + lane < end, &sx, &sy, &sz, &st);
                            if (st >= 0) slow_walk(sx, sy, sz, st, cx, cy);
                        }
                        prep();
                        continue;
                    }
                    const int nch = (int)((end - start + 63) >> 6);
                    v2f hx[B4_CPT / 2], hy[B4_CPT / 2], hz[B4_CPT / 2];   // chunk 2p in .x, chunk 2p+1 in .y
                    unsigned sLo[B4_CPT], sHi[B4_CPT];
                    int lqm = lane;
                    asm volatile("" : "+v"(lqm));   // (per task: hoisted out of the loops, the mask constants are six registers that spill)
#pragma unroll
                    for (int u = 0; u < B4_CPT; ++u) {
                        int tv;
                        float ax, ay, az;
                        xform(R[u], start + u * 64 + lane < end, &ax, &ay, &az, &tv);
                        if (u & 1) hx[u / 2].y = ax, hy[u / 2].y = ay, hz[u / 2].y = az;
                        else hx[u / 2].x = ax, hy[u / 2].x = ay, hz[u / 2].x = az;
                        b4_segmask((unsigned)tv & 63u, tv >= 0, T, lqm, &sLo[u], &sHi[u]);
                    }
                    const unsigned short *row = S.ctab + (lcy - 1) * B4_W1 + lcx - 1;
                    const unsigned c00 = row[0], c03 = row[3], c10 = row[B4_W1], c13 = row[B4_W1 + 3];
                    const unsigned c20 = row[2 * B4_W1], c23 = row[2 * B4_W1 + 3];
                    const unsigned aR[3] = {S.rowBase[lcy - 1] + c00 - S.colOff[lcy - 1], S.rowBase[lcy] + c10 - S.colOff[lcy],
                                            S.rowBase[lcy + 1] + c20 - S.colOff[lcy + 1]};
                    const unsigned nR[3] = {c03 - c00, c13 - c10, c23 - c20};
                    B4_TICK(3)
                    if (PROF) ++pacc[10];
                    prep();   // the next task: its record loads are in flight during the pair phase below
                    B4_TICK(2)
                    if (!(dbg & 1)) {
                        static_assert(B4_CPT == 4, "one specialisation of the pair loop per number of chunk pairs");
                        if (nch <= 2) b4_pairs_rows<1>(live, cntw, aR, nR, hx, hy, hz, sLo, sHi, r2lo, r2hi, r2, Th, lq, T);
                        else b4_pairs_rows<2>(live, cntw, aR, nR, hx, hy, hz, sLo, sHi, r2lo, r2hi, r2, Th, lq, T);
                    }
                    B4_TICK(4)
                    continue;
                }
                if (ty == 3) {
                    prep();
                    B4_TICK(2)
                    continue;
                }
                // ======== a chunk of the packed records of sparse cells: every lane walks its own candidates ========
                do {
                    const unsigned vb = v0, ve = v1;
                    if (PROF) ++pacc[11];
                    if (ve <= vb) break;
                    const unsigned v = vb + lane;
                    const bool valid = v < ve;
                    int kc = 0;
                    if (valid) {   // first cell whose end lies behind v (cells outside the packed order have no extent)
                        int lo = 0, hi = B4_NC - 1;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (S.lvEnd[mid] > v) hi = mid;
                            else lo = mid + 1;
                        }
                        kc = lo;
                    }
                    const int lx = (kc & 15) + 1, ly = (kc >> 4) + 1;
                    const bool inband = valid && ly >= ya && ly <= yb && lx >= xa && lx <= xb;
                    if (!__any(inband)) break;
                    float hx, hy, hz;
                    int trv;
                    xform(rec[S.recStart[kc] + (valid ? v - S.lvStart[kc] : 0u)], inband, &hx, &hy, &hz, &trv);
                    const bool on = trv >= 0;
                    if (slow) {
                        if (on) slow_walk(hx, hy, hz, trv, tbx * 8 + (kc & 15), tby * 8 + (kc >> 4));
                        break;
                    }
                    const unsigned short *row = S.ctab + (ly - 1) * B4_W1 + lx - 1;
                    const unsigned c00 = row[0], c10 = row[B4_W1], c20 = row[2 * B4_W1];
                    const unsigned n0 = row[3] - c00, n1 = row[B4_W1 + 3] - c10, n2 = row[2 * B4_W1 + 3] - c20;
                    const int lyc = inband ? ly : ya;   // (lanes outside the band read the band's own tables)
                    const unsigned a0 = S.rowBase[lyc - 1] + c00 - S.colOff[lyc - 1];
                    const unsigned n01 = n0 + n1, nAll = n01 + n2;
                    const unsigned b1 = S.rowBase[lyc] + c10 - S.colOff[lyc] - n0;
                    const unsigned b2 = S.rowBase[lyc + 1] + c20 - S.colOff[lyc + 1] - n01;
                    const bool mine = on && nAll <= B4_LANE_MAX;
                    const unsigned own = mine ? nAll : 0u;
                    const unsigned cword = ((unsigned)trv & 63u) >> 1, cinc = 1u << ((trv & 1) * 16);
                    for (unsigned p0 = 0; __any(p0 < own); p0 += 2) {
                        unsigned bandBits = 0;
#pragma unroll
                        for (unsigned u = 0; u < 2; ++u) {
                            const unsigned p = p0 + u;
                            const bool act = p < own;
                            const unsigned i = act ? p + (p < n0 ? a0 : (p < n01 ? b1 : b2)) : 0u;
                            const float4 qq = live[i];
                            const float fx = qq.x - hx, fy = qq.y - hy, fz = qq.z - hz;
                            const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            const bool hit = act && d2 < r2lo;
                            bandBits |= (act && !hit && d2 <= r2hi) ? (1u << u) : 0u;
                            if (hit) atomicAdd(&cntw[i * Th + cword], cinc);
                        }
                        while (bandBits) {   // practically never: exact float64 re-test
                            const unsigned u = __ffs((int)bandBits) - 1;
                            bandBits &= bandBits - 1;
                            const unsigned p = p0 + u;
                            const unsigned i = p + (p < n0 ? a0 : (p < n01 ? b1 : b2));
                            const float4 qq = live[i];
                            if (pp_within(hx, hy, hz, qq.x, qq.y, qq.z, r2)) atomicAdd(&cntw[i * Th + cword], cinc);
                        }
                    }
                    // lanes with long candidate lists: one cell group at a time, wave-uniform candidates
                    unsigned long long todo = __ballot(on && !mine);
                    if (todo) {
                        unsigned sLo, sHi;
                        b4_segmask((unsigned)trv & 63u, on, T, lq, &sLo, &sHi);
                        while (todo) {
                            const int src = __ffsll((long long)todo) - 1;
                            const int gk = __builtin_amdgcn_readlane(kc, src);
                            const unsigned long long grp = __ballot(on && !mine && kc == gk);
                            todo &= ~grp;
                            const unsigned ga0 = __builtin_amdgcn_readlane(a0, src), gn0 = __builtin_amdgcn_readlane(n0, src);
                            const unsigned gb1 = __builtin_amdgcn_readlane(b1, src), gn01 = __builtin_amdgcn_readlane(n01, src);
                            const unsigned gb2 = __builtin_amdgcn_readlane(b2, src), gnAll = __builtin_amdgcn_readlane(nAll, src);
                            for (unsigned p = 0; p < gnAll; ++p) {
                                const unsigned i = p + (p < gn0 ? ga0 : (p < gn01 ? gb1 : gb2));
                                const float4 qq = live[i];
                                const float fx = qq.x - hx, fy = qq.y - hy, fz = qq.z - hz;
                                const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                                unsigned long long hb = __ballot(d2 < r2lo) & grp;
                                const unsigned long long mb = __ballot(d2 <= r2hi) & grp;
                                if (hb != mb) {
                                    const bool inBand = !(d2 < r2lo) && d2 <= r2hi;
                                    hb |= __ballot(inBand && pp_within(hx, hy, hz, qq.x, qq.y, qq.z, r2)) & grp;
                                }
                                if (hb) {
                                    const unsigned cN = __popc((unsigned)hb & sLo) + __popc((unsigned)(hb >> 32) & sHi);
                                    if (lane < T && cN) atomicAdd(&cntw[i * Th + ((unsigned)lq >> 1)], cN << ((lq & 1) * 16));
                                }
                            }
                        }
                    }
                } while (false);
                B4_TICK(5)
                prep();   // (after the chunk: requested records of a one-cell task would live across the walk)
                B4_TICK(2)
            }
            B4_TICK(5)
            __syncthreads();
            B4_TICK(6)
            if (!slow)
                for (unsigned e = tid; e < Lb * Th; e += B4_JT) {
                    const unsigned cw = cntw[e];
                    if (cw) {
                        const unsigned pp = e / Th, tp = (e - pp * Th) * 2;
                        const size_t rowi = (size_t)__float_as_int(live[pp].w) * T;
                        if (cw & 0xffffu) atomicAdd(&counts[rowi + tp], (int)(cw & 0xffffu));
                        if (cw >> 16) atomicAdd(&counts[rowi + tp + 1], (int)(cw >> 16));
                    }
                }
        }
        B4_TICK(7)
        if (tid == 0) {
            S.itemId = S.nextId;
            S.item = S.nextItem;
        }
    }
    if (PROF && lane == 0)
        for (int kk = 0; kk < 16; ++kk) atomicAdd(&prof[kk], pacc[kk]);
#undef B4_TICK
}

__global__ void b4_entropy(const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    if (S.H == nullptr || (int)(blockIdx.x * blockDim.x) >= S.n) return;
    pp_entropy_kernel_body(S.counts, S.n, S.T, S.H, blockIdx.x, gridDim.x);
}

}  // namespace

extern "C" int modest_pp_block_limits(int32_t *max_window_tiles, int32_t *max_scans, int32_t *max_frames) {
    if (max_window_tiles) *max_window_tiles = B4_MAXW;
    if (max_scans) *max_scans = 64;
    if (max_frames) *max_frames = 1 << 16;
    return MODEST_OK;
}

extern "C" int modest_pp_score_block(modest_ctx *ctx, const modest_pp_block_frame *frames, int n_frames,
                                     const modest_pp_block_scan *scans, int n_scans, int n_trav, double radius,
                                     double cell, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && scans != nullptr, "NULL argument");
    MODEST_REQUIRE(n_scans >= 1 && n_scans <= 64, "1 <= n_scans <= 64");
    MODEST_REQUIRE(n_frames >= 0 && n_frames < (1 << 16) && (n_frames == 0 || frames != nullptr), "bad frame table");
    MODEST_REQUIRE(n_trav >= 1 && n_trav <= B4_MAXT, "1 <= n_trav <= 64 on the block path");
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    // the lattice cell must leave room for the difference between distances on the lattice and in a scan's frame
    MODEST_REQUIRE(cell >= radius * (1.0 + 1.0 / 512.0) && cell <= radius * 1.25,
                   "lattice cell edge must be in [r (1 + 2^-9), 1.25 r]");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int G = n_scans, U = n_frames, T = n_trav;
    long long ntot = 0, nch = 0;
    for (int f = 0; f < U; ++f) {
        MODEST_REQUIRE(frames[f].n >= 0 && frames[f].tab_dev != nullptr, "bad frame");
        MODEST_REQUIRE(frames[f].n == 0 || frames[f].xyz_dev != nullptr, "NULL frame points");
        ntot += frames[f].n;
        nch += (frames[f].n + B4_CH - 1) / B4_CH;
    }
    MODEST_REQUIRE(ntot < (1LL << 31) && nch < (1LL << 24), "union of frames too large");
    int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0, maxN = 0;
    bool any = false;
    std::vector<int> seenBy((size_t)std::max(U, 1), -1);   // the last scan that named a union slot
    for (int s = 0; s < G; ++s) {
        const modest_pp_block_scan &sc = scans[s];
        MODEST_REQUIRE(sc.n >= 0 && sc.n < (1 << 24), "bad live scan");
        MODEST_REQUIRE(sc.n == 0 || (sc.xyz_dev && sc.perm_dev && sc.tab_dev), "NULL live buffer");
        MODEST_REQUIRE(sc.n == 0 || sc.counts_dev || sc.H_dev, "a scan without an output");
        MODEST_REQUIRE(sc.n_members >= 0 && (sc.n_members == 0 || (sc.member_slot && sc.member_trav && sc.member_rel)),
                       "bad member list");
        for (int m = 0; m < sc.n_members; ++m) {
            MODEST_REQUIRE(sc.member_slot[m] >= 0 && sc.member_slot[m] < U, "member slot out of range");
            MODEST_REQUIRE(sc.member_trav[m] >= 0 && sc.member_trav[m] < T, "frame traversal out of range");
            // one pose entry per (scan, union slot): a frame a scan lists twice (pre_compute_pp_score.py:132-150 stacks it
            // twice) must come as two union entries -- FrameStore.block_tables gives every occurrence a slot of its own
            MODEST_REQUIRE(seenBy[(size_t)sc.member_slot[m]] != s, "a scan names a union slot twice: repeated frames need a union entry per occurrence");
            seenBy[(size_t)sc.member_slot[m]] = s;
        }
        if (sc.n == 0) continue;
        if (!any) {
            bx0 = sc.TX0, by0 = sc.TY0, bx1 = sc.TX0 + B4_NTF, by1 = sc.TY0 + B4_NTF;
            any = true;
        } else {
            bx0 = std::min(bx0, sc.TX0), by0 = std::min(by0, sc.TY0);
            bx1 = std::max(bx1, sc.TX0 + B4_NTF), by1 = std::max(by1, sc.TY0 + B4_NTF);
        }
        maxN = std::max(maxN, sc.n);
    }
    if (!any) return MODEST_OK;
    // one more tile on every side: a live point in the outermost cells of its table still finds the history points of the
    // cells next to it (b4_scatter drops what lies outside the window)
    bx0 -= 1, by0 -= 1, bx1 += 1, by1 += 1;
    const int BW = bx1 - bx0, BH = by1 - by0;
    MODEST_REQUIRE(BW <= B4_MAXW && BH <= B4_MAXW, "the live scans of a block must lie within 30 tiles of each other");
    const int BT = BW * BH, CW = 8 * BW, CHc = 8 * BH;
    const int NC = CW * CHc, NCpad = (NC + B4_SCAN - 1) / B4_SCAN * B4_SCAN, nScanBlk = NCpad / B4_SCAN;
    const int NG = (U + B4_FG - 1) / B4_FG;
    const size_t maxSegs = (size_t)BT + (size_t)(ntot / B4_SEG) + 1;
    const size_t maxItems = (size_t)(ntot / (B4_TASK * B4_IT)) + (size_t)(ntot / (64 * B4_IT)) + 2 * (size_t)BT + 16;

    // ---- arena ------------------------------------------------------------------------------------
    size_t need = 0;
    auto take = [&](size_t bytes) {
        const size_t o = need;
        need += arena_sz(bytes);
        return o;
    };
    const size_t oOff = take((size_t)std::max(U, 1) * BT * 4), oGtot = take((size_t)std::max(NG, 1) * BT * 4);
    const size_t oTotal = take((size_t)BT * 4), oBase = take((size_t)BT * 4), oSegBase = take((size_t)BT * 4);
    const size_t oSegList = take(maxSegs * 4), oSegHist = take(maxSegs * 64 * 4), oSegOff = take(maxSegs * 64 * 4);
    const size_t oCellOff = take((size_t)BT * 65 * 4), oCtrl = take(256), oSegRange = take(maxSegs * 8);
    const size_t oBaseSum = take((size_t)((BT + 1023) / 1024) * 8);
    const size_t oRecA = take((size_t)std::max<long long>(ntot, 1) * 16), oRecB = take((size_t)std::max<long long>(ntot, 1) * 16);
    const size_t oCellCount = take((size_t)G * (NCpad + 4) * 4);   // contiguous over the scans: one memset
    struct ScanOff {
        size_t cellStart, blockSum, tileTasks, ctrl, tmp, sorted, items, counts, cellRange;
    };
    std::vector<ScanOff> so((size_t)G);
    for (int s = 0; s < G; ++s) {
        const int n = scans[s].n;
        so[(size_t)s].cellStart = take((size_t)(NCpad + 4) * 4);
        so[(size_t)s].blockSum = take((size_t)(nScanBlk + 1) * 4);
        so[(size_t)s].tileTasks = take((size_t)BT * 4);
        so[(size_t)s].ctrl = take(256);
        so[(size_t)s].tmp = take((size_t)std::max(n, 1) * 16);
        so[(size_t)s].sorted = take((size_t)std::max(n, 1) * 16);
        so[(size_t)s].items = take(maxItems * 16);
        so[(size_t)s].counts = take((size_t)std::max(n, 1) * T * 4);
        so[(size_t)s].cellRange = take((size_t)BT * 64 * 8);
    }
    // staged block: [UFrame x U][chunkTab][ScanDev x G][PoseEnt x G x U]
    const size_t stFrames = 0, stChunks = arena_sz((size_t)std::max(U, 1) * sizeof(UFrame));
    const size_t stScans = stChunks + arena_sz((size_t)std::max<long long>(nch, 1) * sizeof(uint2));
    const size_t stPose = stScans + arena_sz((size_t)G * sizeof(ScanDev));
    const size_t stageB = stPose + arena_sz((size_t)G * std::max(U, 1) * sizeof(PoseEnt));
    const size_t oStage = take(stageB);
    int rc = modest_ctx_reserve(ctx, need);
    if (rc) return rc;
    char *hs = nullptr;
    rc = modest_ctx_stage_slot(ctx, stageB, reinterpret_cast<void **>(&hs));
    if (rc) return rc;
    char *base = ctx->scratch, *dstage = base + oStage;
    UFrame *hf = reinterpret_cast<UFrame *>(hs + stFrames);
    uint2 *hc = reinterpret_cast<uint2 *>(hs + stChunks);
    ScanDev *hsc = reinterpret_cast<ScanDev *>(hs + stScans);
    PoseEnt *hp = reinterpret_cast<PoseEnt *>(hs + stPose);
    size_t kc = 0;
    for (int f = 0; f < U; ++f) {
        UFrame &d = hf[f];
        d.xyz = frames[f].xyz_dev;
        d.tab = frames[f].tab_dev;
        d.n = frames[f].n;
        d.TX0 = frames[f].TX0;
        d.TY0 = frames[f].TY0;
        d.flags = frames[f].flags;
        for (int q = 0; q < 8; ++q) d.lat[q] = frames[f].lat[q];
        for (int p0 = 0; p0 < frames[f].n; p0 += B4_CH) hc[kc++] = make_uint2((unsigned)f, (unsigned)p0);
    }
    for (size_t i = 0; i < (size_t)G * std::max(U, 1); ++i) {
        memset(&hp[i], 0, sizeof(PoseEnt));
        hp[i].trav = -1;
    }
    for (int s = 0; s < G; ++s) {
        const modest_pp_block_scan &sc = scans[s];
        const ScanOff &o = so[(size_t)s];
        ScanDev &d = hsc[s];
        memset(&d, 0, sizeof(d));
        d.liveXyz = sc.xyz_dev;
        d.livePerm = sc.perm_dev;
        d.liveTab = sc.tab_dev;
        d.cellCount = reinterpret_cast<unsigned *>(base + oCellCount) + (size_t)s * (NCpad + 4);
        d.cellStart = reinterpret_cast<unsigned *>(base + o.cellStart);
        d.blockSum = reinterpret_cast<unsigned *>(base + o.blockSum);
        d.tileTasks = reinterpret_cast<unsigned *>(base + o.tileTasks);
        d.ctrl = reinterpret_cast<unsigned *>(base + o.ctrl);
        d.tmp = reinterpret_cast<float4 *>(base + o.tmp);
        d.sorted = reinterpret_cast<float4 *>(base + o.sorted);
        d.items = reinterpret_cast<uint4 *>(base + o.items);
        d.cellRange = reinterpret_cast<uint2 *>(base + o.cellRange);
        d.pose = reinterpret_cast<const PoseEnt *>(dstage + stPose) + (size_t)s * std::max(U, 1);
        d.counts = sc.counts_dev ? sc.counts_dev : reinterpret_cast<int *>(base + o.counts);
        d.H = sc.H_dev;
        for (int q = 0; q < 8; ++q) d.lat[q] = sc.lat[q];
        for (int q = 0; q < 12; ++q) d.rel[q] = sc.rel[q];
        d.n = sc.n;
        d.TX0 = sc.TX0;
        d.TY0 = sc.TY0;
        d.T = T;
        d.maxItems = (int)maxItems;
        PoseEnt *pe = hp + (size_t)s * std::max(U, 1);
        d.slotLo = U, d.slotHi = -1;
        for (int m = 0; m < sc.n_members; ++m) {
            d.slotLo = std::min(d.slotLo, sc.member_slot[m]);
            d.slotHi = std::max(d.slotHi, sc.member_slot[m]);
        }
        if (d.slotHi < 0) d.slotLo = 0, d.slotHi = 0;
        for (int m = 0; m < sc.n_members; ++m) {
            PoseEnt &e = pe[sc.member_slot[m]];
            for (int q = 0; q < 12; ++q) e.rel[q] = sc.member_rel[(size_t)m * 12 + q];
            e.trav = sc.member_trav[m];
        }
    }
    MODEST_HIP_CHECK(hipMemcpyAsync(dstage, hs, stageB, hipMemcpyHostToDevice, stream));
    rc = modest_ctx_stage_commit(ctx, stream);
    if (rc) return rc;

    Blk B;
    memset(&B, 0, sizeof(B));
    B.frames = reinterpret_cast<const UFrame *>(dstage + stFrames);
    B.chunkTab = reinterpret_cast<const uint2 *>(dstage + stChunks);
    B.off = reinterpret_cast<unsigned *>(base + oOff);
    B.gtot = reinterpret_cast<unsigned *>(base + oGtot);
    B.listTotal = reinterpret_cast<unsigned *>(base + oTotal);
    B.listBase = reinterpret_cast<unsigned *>(base + oBase);
    B.segBase = reinterpret_cast<unsigned *>(base + oSegBase);
    B.segList = reinterpret_cast<unsigned *>(base + oSegList);
    B.segHist = reinterpret_cast<unsigned *>(base + oSegHist);
    B.segOff = reinterpret_cast<unsigned *>(base + oSegOff);
    B.cellOff = reinterpret_cast<unsigned *>(base + oCellOff);
    B.segRange = reinterpret_cast<uint2 *>(base + oSegRange);
    B.ctrl = reinterpret_cast<unsigned *>(base + oCtrl);
    B.baseSum = reinterpret_cast<unsigned *>(base + oBaseSum);
    B.recA = reinterpret_cast<float4 *>(base + oRecA);
    B.recB = reinterpret_cast<float4 *>(base + oRecB);
    B.U = U, B.NG = NG, B.nchunks = (int)nch, B.maxSegs = (int)maxSegs;
    B.BX0 = bx0, B.BY0 = by0, B.BW = BW, B.BH = BH, B.BT = BT, B.CW = CW, B.CHc = CHc, B.NCpad = NCpad;
    B.nScanBlk = nScanBlk, B.G = G;
    const ScanDev *dsc = reinterpret_cast<const ScanDev *>(dstage + stScans);

    static bool attr_done[64] = {false};
    if (!attr_done[ctx->device & 63]) {
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_seg_scatter),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, B4_SEG * 16));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<false, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, B4_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<false, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, B4_LDS_DYN));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<true, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             B4_LDS_DYN + (int)b4_pose_bytes(B4_POSE_LDS_MAX)));
        attr_done[ctx->device & 63] = true;
    }
    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of the block
    MODEST_HIP_CHECK(hipMemsetAsync(base + oCellCount, 0, (size_t)G * (NCpad + 4) * 4, stream));
    const unsigned gBT = (unsigned)((BT + 255) / 256), gN = (unsigned)((maxN + 255) / 256);
    if (U > 0 && ntot > 0) {
        b4_counts<<<dim3(gBT, (unsigned)NG), 256, 0, stream>>>(B);
        b4_lists<<<gBT, 256, 0, stream>>>(B, dsc);
        b4_bases_local<<<(unsigned)((BT + 1023) / 1024), 1024, 0, stream>>>(B);
        b4_bases_finish<<<(unsigned)((BT + 1023) / 1024), 1024, 0, stream>>>(B);
        const int swg = std::min((int)nch, 3 * ctx->num_cus);
        b4_scatter<<<swg, 512, 0, stream>>>(B);
        b4_seg_hist<<<(unsigned)maxSegs, 512, 0, stream>>>(B);
        b4_seg_scan<<<(unsigned)((BT + 3) / 4), 256, 0, stream>>>(B);
        b4_seg_scatter<<<(unsigned)maxSegs, 512, B4_SEG * 16, stream>>>(B);
        b4_live_count<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(B, dsc);
        b4_scan_local<<<dim3((unsigned)nScanBlk, (unsigned)G), 1024, 0, stream>>>(B, dsc);
        b4_scan_finish<<<(unsigned)G, 1024, 0, stream>>>(B, dsc);
        b4_live_scatter<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(B, dsc);
        b4_plan_tiles<<<dim3((unsigned)((BT + 3) / 4), (unsigned)G), 256, 0, stream>>>(B, dsc);
        b4_plan_items<<<(unsigned)G, 1024, 0, stream>>>(B, dsc);
        const char *jw_env = getenv("MODEST_PP4_JWG");
        unsigned jx = (unsigned)((jw_env ? atoi(jw_env) : (B4_JT >= 512 ? 2 : 4)) * ctx->num_cus) / (unsigned)G;
        if (jx < 8) jx = 8;
        const char *dbg_env = getenv("MODEST_PP4_DBG");
        const int dbg = dbg_env ? atoi(dbg_env) : 0;
        const unsigned poseB = b4_pose_bytes(U);
        if (dbg & 512) {   // MODEST_PP4_DBG=512: per-phase wavefront cycles of the join (blocking; diagnostics only)
            unsigned long long *dprof = reinterpret_cast<unsigned long long *>(base + oCtrl + 64), hprof[16];
            MODEST_HIP_CHECK(hipMemsetAsync(dprof, 0, sizeof(hprof), stream));
            b4_join<false, true><<<dim3(jx, (unsigned)G), B4_JT, B4_LDS_DYN, stream>>>(B, dsc, radius * radius, dbg, dprof);
            MODEST_HIP_CHECK(hipStreamSynchronize(stream));
            MODEST_HIP_CHECK(hipMemcpy(hprof, dprof, sizeof(hprof), hipMemcpyDeviceToHost));
            const double wv = (double)jx * G * (B4_JT / 64), us = 1.0 / 100.0;   // s_memtime ticks at 100 MHz
            fprintf(stderr, "[b4_join] per wavefront, us: item set-up (decode) %.1f | band set-up %.1f | fetch+decode %.1f | record wait+transform %.1f | "
                            "pairs %.1f | packed %.1f | band-end wait %.1f | flush %.1f (item set-up = %.1f to the tables' barrier + %.1f prefix / window tables + %.1f band search + decode) || per scan: items %.0f bands %.0f tasks %.0f chunks %.0f\n",
                    hprof[0] * us / wv, hprof[1] * us / wv, hprof[2] * us / wv, hprof[3] * us / wv, hprof[4] * us / wv, hprof[5] * us / wv,
                    hprof[6] * us / wv, hprof[7] * us / wv, hprof[12] * us / wv, hprof[13] * us / wv, hprof[14] * us / wv, hprof[8] / 4.0 / G, hprof[9] / 4.0 / G, (double)hprof[10] / G, (double)hprof[11] / G);
        } else if (poseB && B4_JT >= 512 && !(dbg & 256))
            b4_join<true, false><<<dim3(jx, (unsigned)G), B4_JT, B4_LDS_DYN + poseB, stream>>>(B, dsc, radius * radius, dbg, nullptr);
        else
            b4_join<false, false><<<dim3(jx, (unsigned)G), B4_JT, B4_LDS_DYN, stream>>>(B, dsc, radius * radius, dbg, nullptr);
    } else {   // no history: every count is zero
        for (int sc = 0; sc < G; ++sc)
            if (scans[sc].n > 0) MODEST_HIP_CHECK(hipMemsetAsync(hsc[sc].counts, 0, (size_t)scans[sc].n * T * 4, stream));
    }
    modest_prof_mark(ctx, stream, 1);
    b4_entropy<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(dsc);
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}
