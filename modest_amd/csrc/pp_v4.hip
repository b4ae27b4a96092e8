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
//     chain's frames is therefore known before a single point is read (b4_need / b4_counts / b4_lists / b4_bases:
//     a column sum over the frame tables, for the tiles some scan needs) -- no count pass, no count matrix, no
//     occupancy bitmap;
//   * b4_scatter streams the union frames once and copies every point, RAW (frame coordinates) plus its
//     frame slot and lattice cell, to its final position in its tile list (the position follows from the
//     tables: no atomics); tiles no scan of the chain has a live point near are skipped;
//   * b4_seg_hist / b4_seg_scan / b4_seg_scatter order every tile list by cell (counting sort over 2048-record
//     segments), so that the records of a cell are CONTIGUOUS in HBM for the whole chain;
//   * per scan: the live scan is cell-sorted on the same lattice (b4_live_*), b4_plan writes a flat list of
//     self-contained tasks (<= 256 records of ONE cell + the three runs of live points around it), and b4_join
//     -- every wavefront on its own, no workgroup barrier, no LDS window -- reads a task's records straight into
//     registers (tasks dealt by tickets of the scan's queue), applies the scan's own float32 pose to every record
//     (transform_points' rounding, per (scan, frame)) and tests them against one wave-uniform candidate per step (read with a scalar load: the live
//     point arrives in SGPRs): ballots + traversal-segmented popcounts, one global atomic per (candidate,
//     task).  Sparse cells (< 64 records) go four to a task, a cell per 64-lane chunk, each against its own candidates.
//     (Round 4 dealt ITEMS of 24 tasks to whole workgroups, with the live window of a quad of tiles in LDS: four
//     barriers and four dependent rounds of loads per item -- 34 of the join's 99 us per scan went there.)
//
// Exactness: the lattice is only a conservative spatial filter.  Distances are evaluated exactly as V3
// does -- float32 pre-test, float64 re-test (scipy's predicate) inside a 1.5e-6 band around r^2 -- on
// coordinates produced by the reference's float32 chain from the scan's OWN relative poses.  Two points
// within r in a scan's common frame are within r + 2 * 1e-4 m on the lattice (the caller checks every
// pose against the lattice to 1e-4 m, frame_store.consistent), the lattice cell edge is at least
// r * (1 + 2^-9) (required below), so their lattice cells differ by at most one per axis.
#include "pp_frames.h"
#include <algorithm>
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
#ifndef B4_SEG_
#define B4_SEG_ 2048
#endif
constexpr int B4_SEG = B4_SEG_;            // records per sort segment.  A scan reads whole segments, and the two at the ends of its slot range also
                                           // hold other scans' frames (masked per lane: 17 % of the join's pair tests at 4096): 4096 / 2048 / 1024 records
                                           // = 84.5 / 81.7 / ~82 us per scan (the sort passes grow by 6 / 35 us per block)
constexpr int B4_FG = 32;                  // frames per prefix group
constexpr unsigned B4_HEAVY = 64;          // cells with at least this many records get tasks of their own
#ifndef B4_CPT_
#define B4_CPT_ 4
#endif
constexpr int B4_CPT = B4_CPT_;                  // 64-record chunks per task
constexpr unsigned B4_TASK = 64 * B4_CPT;  // 256 records
// the task lists' capacities in pp_block_impl (ntot / 64 + 16 one-cell tasks, the four-cell bound) are derived for these values
static_assert(B4_HEAVY == 64 && B4_CPT == 4, "maxTasks / maxLight in pp_block_impl assume cells of >= 64 records in tasks of 4 chunks");
#ifndef B4_WPE_
#define B4_WPE_ 4
#endif
#ifndef B4_JT_
#define B4_JT_ 1024
#endif
constexpr int B4_JT = B4_JT_;                 // threads of a join workgroup
constexpr int B4_MAXT = 64;
constexpr int B4_POSE_LDS_MAX = 2048;       // union entries whose poses fit the LDS table of a join workgroup (100 KB of the CU's 160: one workgroup per CU)

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
    unsigned *needList;   // the tiles some scan of the block needs AND that hold points, in no particular order; ctrl[2] = their number
    unsigned *needCand;   // the tiles some scan needs (ctrl[3] of them)
    unsigned long long *needMask;   // per window tile: the scans with a live point in the 3x3 tiles around it
    uint2 *segRange;   // smallest / largest frame slot among a segment's records
    unsigned *deal;     // b4_deal: [0, G] first join workgroup of every scan (G: their number), then the scan of every join workgroup
    float4 *recA, *recB;
    int U, NG, nchunks, maxSegs;
    int compact;        // b4_join packs the member records of a one-cell task (LDS: 4 KB per wavefront on top of the pose table and the masks)
    int BX0, BY0, BW, BH, BT, CW, CHc, NCpad, nScanBlk, G;
};

struct ScanDev {   // per scan (device table)
    const float *liveXyz;
    const unsigned *livePerm, *liveTab;
    unsigned *cellCount, *cellStart, *blockSum, *ctrl;   // ctrl: [0] one-cell tasks, [1] four-cell tasks (b4_plan's cursors)
    float4 *tmp, *sorted;
    void *tasks;       // B4Task x maxTasks
    uint4 *ltHead;     // four-cell tasks: (first record, records) x 4 per task = 2 x uint4, x maxLight
    uint4 *ltSegs;     // ... (a0, n0, a1, n1), (a2, n2, -, -) per cell = 8 x uint4 per task
    const PoseEnt *pose;
    int *counts;
    float *H;
    double lat[8];
    float rel[12];
    int n, TX0, TY0, T, maxTasks, maxLight;
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

// The live scans are indexed on the cells of the CROP: the bounding box of the block's needed tiles plus one tile on every side,
// found by b4_need (ctrl[8..11]: maxima of BW - 1 - x, x, BH - 1 - y, y over the needed tiles -- zero-initialised by the call's
// one memset).  The window (160 x 160 tiles for 16 Lyft scans: the union of the frame tables) is 2.8 x larger than what the live
// points occupy, and the dense per-scan cell arrays -- counters, starts -- were cleared, scanned and read over all of it.
struct B4Crop {
    int x0, y0, CW, CH, nBlk;   // first tile (relative to the window), cells per axis, 4096-cell scan blocks
};
__device__ __forceinline__ B4Crop b4_crop(const Blk &B) {
    const int c0 = (int)B.ctrl[8], c1 = (int)B.ctrl[9], c2 = (int)B.ctrl[10], c3 = (int)B.ctrl[11];
    int x0 = B.BW - 1 - c0, x1 = c1, y0 = B.BH - 1 - c2, y1 = c3;
    if (x0 > x1 || y0 > y1) x0 = x1 = y0 = y1 = 0;   // (no needed tile at all)
    x0 = max(x0 - 1, 0), y0 = max(y0 - 1, 0), x1 = min(x1 + 1, B.BW - 1), y1 = min(y1 + 1, B.BH - 1);
    B4Crop c;
    c.x0 = x0, c.y0 = y0, c.CW = 8 * (x1 - x0 + 1), c.CH = 8 * (y1 - y0 + 1);
    c.nBlk = (c.CW * c.CH + 4095) / 4096;
    return c;
}

// ---- list sizes from the frame tables ---------------------------------------------------------
// thread per block tile: is the tile needed (a live point of any scan in the 3x3 tiles around it), and by which scans?  Needed
// tiles are LISTED (7 k of the window's 25.6 k): the column sums below, the plan and the join only ever look at those.
// (two launches: eight scans per thread and grid row -- one thread walking all 32 scans of a block was 41 us of dependent loads --,
// then the list)
constexpr int B4_NEED_SCANS = 8;
__global__ __launch_bounds__(256) void b4_need(Blk B, const ScanDev *__restrict__ scans) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B.BT) return;
    const int gx = B.BX0 + b % B.BW, gy = B.BY0 + b / B.BW;
    unsigned long long mask = 0ULL;
    const int s0 = blockIdx.y * B4_NEED_SCANS, s1 = min(B.G, s0 + B4_NEED_SCANS);
#pragma unroll 8
    for (int s = s0; s < s1; ++s) {
        const ScanDev &S = scans[s];
        const int l0 = max(gx - 1 - S.TX0, 0), l1 = min(gx + 1 - S.TX0, B4_NTF - 1);
        bool needed = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {   // (the loads are unconditional on clamped indices: nothing waits inside a branch)
            const int ly = gy + dy - S.TY0;
            const bool in = S.n > 0 && ly >= 0 && ly < B4_NTF && l0 <= l1;
            const int row = in ? ly * B4_NTF : 0, a = in ? l0 : 0, e = in ? l1 + 1 : 0;
            const unsigned t0 = S.liveTab[row + a], t1 = S.liveTab[row + e];
            needed |= in && t1 > t0;
        }
        mask |= (unsigned long long)needed << s;
    }
    if (mask) atomicOr(&B.needMask[b], mask);   // (the masks were cleared with the block's cursors)
}
__global__ __launch_bounds__(256) void b4_need_list(Blk B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const bool tile = b < B.BT;   // (no early exit: the wave reductions below want every lane of the last wavefront present)
    const unsigned long long mask = tile ? B.needMask[b] : 0ULL;
    if (tile) B.listTotal[b] = 0u;   // (b4_lists writes the totals of the needed tiles that hold points)
    // one cursor atomic per wavefront, the wavefront's tiles in order: neighbours in the list are neighbours in a tile row, and the
    // column sums below read and write neighbouring table entries
    const unsigned long long bal = __ballot(mask != 0ULL);
    if (bal) {
        const int lane = threadIdx.x & 63;
        {   // the bounding box of the needed tiles (b4_crop)
            int a = mask ? B.BW - 1 - b % B.BW : 0, c = mask ? b % B.BW : 0, d = mask ? B.BH - 1 - b / B.BW : 0, e = mask ? b / B.BW : 0;
            for (int o = 32; o > 0; o >>= 1) {
                a = max(a, __shfl_xor(a, o)), c = max(c, __shfl_xor(c, o));
                d = max(d, __shfl_xor(d, o)), e = max(e, __shfl_xor(e, o));
            }
            if (lane == 0) {
                atomicMax(&B.ctrl[8], (unsigned)a), atomicMax(&B.ctrl[9], (unsigned)c);
                atomicMax(&B.ctrl[10], (unsigned)d), atomicMax(&B.ctrl[11], (unsigned)e);
            }
        }
        unsigned pos = 0;
        if (lane == 0) pos = atomicAdd(&B.ctrl[3], (unsigned)__popcll(bal));
        pos = __shfl(pos, 0);
        if (mask) B.needCand[pos + (unsigned)__popcll(bal & ((1ULL << lane) - 1ULL))] = (unsigned)b;
    }
}

// thread (needed tile, frame group g): exclusive prefix of the tile's point counts over the group's frames
__global__ __launch_bounds__(256) void b4_counts(Blk B) {
    const unsigned q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B.ctrl[3]) return;
    const int b = (int)B.needCand[q];
    const int g = blockIdx.y;
    const int gx = B.BX0 + b % B.BW, gy = B.BY0 + b / B.BW;
    const int f1 = min(B.U, (g + 1) * B4_FG);
    unsigned run = 0;
    // eight frames per trip, every table entry requested before the first is used (one frame per trip was 32 dependent rounds of a
    // descriptor load + a table load: 32 us for 7 k tiles)
    for (int f0 = g * B4_FG; f0 < f1; f0 += 8) {
        unsigned t0[8], t1[8];
        bool in[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const UFrame &F = B.frames[min(f0 + u, f1 - 1)];   // wave-uniform: scalar loads
            const int lx = gx - F.TX0, ly = gy - F.TY0;
            in[u] = f0 + u < f1 && lx >= 0 && lx < B4_NTF && ly >= 0 && ly < B4_NTF;
            const int k = in[u] ? ly * B4_NTF + lx : 0;
            t0[u] = F.tab[k], t1[u] = F.tab[k + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (f0 + u < f1) B.off[(size_t)(f0 + u) * B.BT + b] = run;
            run += in[u] ? t1[u] - t0[u] : 0u;
        }
    }
    B.gtot[(size_t)g * B.BT + b] = run;
}

// thread per needed tile: group bases, list total; the tiles with points go to the list b4_plan walks
__global__ __launch_bounds__(256) void b4_lists(Blk B) {
    const unsigned q = blockIdx.x * 256 + threadIdx.x;
    if (q >= B.ctrl[3]) return;
    const int b = (int)B.needCand[q];
    unsigned run = 0;
    for (int g = 0; g < B.NG; ++g) {
        const unsigned t = B.gtot[(size_t)g * B.BT + b];
        B.gtot[(size_t)g * B.BT + b] = run;
        run += t;
    }
    B.listTotal[b] = run;
    if (run != 0u) B.needList[atomicAdd(&B.ctrl[2], 1u)] = (unsigned)b;   // b4_plan walks this list instead of the window
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
            // remove_center (pre_compute_pp_score.py:48-52,141-142) drops the point before the transform: an x of 1e30 keeps
            // its slot in the list and never passes a distance test (a rotation has no zero column: some coordinate of the
            // transformed point is ~1e30, its squared distance overflows to +inf)
            float xs = x[u];
            if ((F.flags & F_FLAG_CENTER) && in_center_box(x[u], y[u])) xs = 1.0e30f;   // (not a NaN: the join keeps a running minimum of |d2 - r2|)
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
// grid.y = scan.  The cell counters of the crop start at zero (launched over the window's scan blocks: the workgroups behind the
// crop leave at once)
__global__ __launch_bounds__(1024) void b4_zero_cells(Blk B, const ScanDev *__restrict__ scans) {
    const B4Crop C = b4_crop(B);
    if ((int)blockIdx.x >= C.nBlk) return;
    reinterpret_cast<uint4 *>(scans[blockIdx.y].cellCount)[(size_t)blockIdx.x * 1024 + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
}
// grid.y = scan.  tmp[i] = (x, y, z in the scan's common frame, cell); the cell counters were cleared
__global__ __launch_bounds__(256) void b4_live_count(Blk B, const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    const B4Crop C = b4_crop(B);
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
    cx -= 8LL * (B.BX0 + C.x0);
    cy -= 8LL * (B.BY0 + C.y0);
    // (a live point's tile is a needed tile, i.e. inside the crop: the clamp only keeps a corrupt input from writing out of bounds)
    const int bx = (int)min(max(cx, 0LL), (long long)C.CW - 1), by = (int)min(max(cy, 0LL), (long long)C.CH - 1);
    const int cell = by * C.CW + bx;
    atomicAdd(&S.cellCount[cell], 1u);
    S.tmp[i] = make_float4(o[0], o[1], o[2], __int_as_float(cell));
}

// exclusive scan of the cell counters (4096 per workgroup, four per thread); the counters stay: the scatter counts
// them back down to zero (its cursor), which saves a 4-byte store per cell of the window here
constexpr int B4_SCAN = 4096;
__global__ __launch_bounds__(1024) void b4_scan_local(Blk B, const ScanDev *__restrict__ scans) {
    __shared__ unsigned wsum[16];
    const ScanDev &S = scans[blockIdx.y];
    if ((int)blockIdx.x >= b4_crop(B).nBlk) return;   // (wave-uniform)
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
    const int nBlk = b4_crop(B).nBlk;
    unsigned run = 0;   // (window of at most 160 x 160 tiles: 400 scan blocks; the loop is for larger constants)
    for (int base = 0; base < nBlk; base += 1024) {
        const int k = base + tid;
        const unsigned v = k < nBlk ? S.blockSum[k] : 0u;
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
        if (k < nBlk) S.blockSum[k] = off + inc - v;
        run += all;
    }
    if (tid == 0) {
        S.blockSum[nBlk] = run;
        S.cellStart[(size_t)nBlk * B4_SCAN] = 0u;
        S.ctrl[0] = 0u, S.ctrl[1] = 0u, S.ctrl[40] = 0u, S.ctrl[41] = 0u;   // b4_plan's cursors (from the front / from the back of the lists)
        S.ctrl[32] = 0u, S.ctrl[48] = 0u;   // b4_join's tickets (one-cell / four-cell tasks)
        S.ctrl[42] = 0u;                    // b4_plan's overflow word
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
    // counts are indexed by the ORIGINAL point order of the live frame: w = the BYTE offset of the point's row of counts (n, T) --
    // the join adds it to the counts pointer with two scalar instructions per atomic (a row number cost a 64-bit multiply and shift)
    S.sorted[slot] = make_float4(t.x, t.y, t.z, __int_as_float((int)(S.livePerm[i] * (unsigned)S.T * 4u)));
}

// ---- plan ------------------------------------------------------------------------------------------
// The join's work for one scan is a flat list of self-contained TASKS.  A one-cell task: at most 256 records of ONE cell
// of the block store that has live points in the 3x3 cells around it ("heavy" cell: >= 64 records of the scan's own frames)
// together with its candidates -- three runs of the scan's cell-sorted live points (the cell rows cy-1, cy, cy+1; a row's
// three cells are contiguous).  All other cells with live points nearby ("light": < 64 records) go FOUR to a task -- a cell per
// 64-lane chunk, each with its own three candidate runs (header: first record + count per cell; body: the runs).  One
// wavefront per tile of the block's needed-tile list, lane = cell; list positions come from two atomic counters per scan:
// the order of the lists is free, every count is an integer sum.
struct B4Task {   // 32 bytes
    unsigned recStart, recEnd;           // records [recStart, recEnd) of the block store
    unsigned a0, n0, a1, n1, a2, n2;     // candidates: live points [a, a + n) of the scan's sorted array, per cell row
};
static_assert(sizeof(B4Task) == 32, "task layout");

// the live points around cell (cx, cy), row by row
__device__ __forceinline__ void b4_cell_segs(const unsigned *__restrict__ cellStart, const unsigned *__restrict__ blockOff, int CW, int CHc,
                                             int cx, int cy, unsigned *a, unsigned *n) {
    const int xa = max(cx - 1, 0), xb = min(cx + 1, CW - 1);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {   // (three rows, all loads issued: rows outside the window read the cell's own row and count nothing)
        const int yy = cy + dy;
        const bool in = yy >= 0 && yy < CHc;
        const size_t row = (size_t)(in ? yy : cy) * CW;
        const unsigned s = b4_cs(cellStart, blockOff, row + xa), e = b4_cs(cellStart, blockOff, row + xb + 1);
        a[dy + 1] = s;
        n[dy + 1] = in ? e - s : 0u;
    }
}
__global__ __launch_bounds__(256) void b4_plan(Blk B, const ScanDev *__restrict__ scans) {
    const ScanDev &S = scans[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const unsigned nNeed = B.ctrl[2];
    const B4Crop C = b4_crop(B);
    // (a wavefront per tile of the list, eight wavefronts per SIMD: the window has four times as many tiles as the list, and a
    // launch over all of them spent more time starting wavefronts that leave at once than on the tiles with work)
    // Every trip is a chain of dependent load rounds; the first two (the tile, its list header) are requested a trip ahead, and
    // everything that depends on the tile alone (cell bases, the live runs around every cell) is requested before the round that
    // finds the scan's segments: three rounds per trip instead of five.
    const unsigned stride = gridDim.x * 4;
    unsigned it = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned bC = it < nNeed ? B.needList[it] : 0u, bN = it + stride < nNeed ? B.needList[it + stride] : 0u;
    unsigned long long mC = B.needMask[bC];
    unsigned tC = B.listTotal[bC], sC = B.segBase[bC];
#pragma unroll 1
    for (; it < nNeed; it += stride) {
    const unsigned bNN = it + 2 * stride < nNeed ? B.needList[it + 2 * stride] : 0u;
    const unsigned long long mN = B.needMask[bN];
    const unsigned tN = B.listTotal[bN], sN = B.segBase[bN];
    do {
    const int b = (int)bC;
    if (!((mC >> blockIdx.y) & 1ULL)) break;   // (no live point of THIS scan around the tile: wave-uniform)
    const unsigned total = tC;
    const unsigned base = B.cellOff[(size_t)b * 65 + lane], next = B.cellOff[(size_t)b * 65 + lane + 1];
    const int cx = (b % B.BW - C.x0) * 8 + (lane & 7), cy = (b / B.BW - C.y0) * 8 + (lane >> 3);   // (on the crop)
    unsigned sa[3], sn[3];
    b4_cell_segs(S.cellStart, S.blockSum, C.CW, C.CH, cx, cy, sa, sn);
    // non-empty runs first: the join streams a task's candidates run after run, the first trip of a run requested during the
    // last trip of the run before it, and stops at the first empty run
    if (sn[0] == 0u) sa[0] = sa[1], sn[0] = sn[1], sa[1] = sa[2], sn[1] = sn[2], sn[2] = 0u;
    if (sn[0] == 0u) sa[0] = sa[1], sn[0] = sn[1], sn[1] = 0u;
    if (sn[1] == 0u) sa[1] = sa[2], sn[1] = sn[2], sn[2] = 0u;
    // The part of every cell this scan reads.  The tile list is in the order of the block's frame table and its
    // segments were sorted one by one, so a cell's records are [run of segment 0 | run of segment 1 | ...] with ascending
    // frame slots from run to run: the runs of the segments that overlap [slotLo, slotHi] hold every record of the
    // scan's own frames (and, in the two boundary runs, some of the block's other scans' frames: masked per lane).
    const unsigned ns = (total + B4_SEG - 1) / B4_SEG, s0 = sC;
    unsigned kf = ns, kl = 0;
    bool any = false;
    for (unsigned k0 = 0; k0 < ns; k0 += 64) {   // (lane = segment: one round of loads for up to 64 segments)
        const unsigned k = k0 + (unsigned)lane;
        const uint2 r = B.segRange[s0 + min(k, ns - 1)];
        const unsigned long long hit = __ballot(k < ns && r.y >= (unsigned)S.slotLo && r.x <= (unsigned)S.slotHi);
        if (hit) {
            if (!any) kf = k0 + (unsigned)__ffsll((long long)hit) - 1u;
            kl = k0 + 63u - (unsigned)__clzll((long long)hit);
            any = true;
        }
    }
    // (every load is issued whether or not its value is used: the compiler waits for a load inside the branch
    // that holds it, and this kernel is nothing but dependent loads)
    if (!any) kf = 0, kl = 0;
    const unsigned oFirst = B.segOff[(size_t)(s0 + kf) * 64 + lane];
    const unsigned oEnd = B.segOff[(size_t)(s0 + min(kl + 1, ns - 1)) * 64 + lane];
    const unsigned cand = sn[0] + sn[1] + sn[2];
    const unsigned start = base + oFirst;
    const unsigned end = kl + 1 < ns ? base + oEnd : next;
    const unsigned n = any ? end - start : 0u;
    const bool active = n != 0u && cand != 0u;
    const unsigned th = (active && n >= B4_HEAVY) ? (((n + 63) >> 6) + B4_CPT - 1) / B4_CPT : 0u;
    const unsigned lv = (active && n < B4_HEAVY) ? n : 0u;
    unsigned incT = th, incL = lv;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned x = __shfl_up(incT, o), y = __shfl_up(incL, o);
        if (lane >= o) {
            incT += x;
            incL += y;
        }
    }
    const unsigned TH = __shfl(incT, 63), LV = __shfl(incL, 63);
    if (TH + LV == 0u) break;
    // sparse cells: FOUR cells to a task (a cell per 64-lane chunk of the wavefront, each with its own three candidate runs)
    const bool isL = lv != 0u;
    const unsigned long long lmask = __ballot(isL);
    const unsigned nLc = (unsigned)__popcll(lmask), li = (unsigned)__popcll(lmask & ((1ULL << lane) - 1ULL));
    const unsigned nL = (nLc + 3) >> 2;   // <= 16
    // ONE cursor atomic per tile: both list positions in a 64-bit word (ctrl[0] one-cell tasks | ctrl[1] four-cell tasks).  The
    // cursors of a scan are one address for every wavefront of its grid row, and atomics on one address are served one after
    // the other.  (Ablation, us per block of 16: loads 27, + prefix sums 34, + cursors and descriptor writes 71 -- the loop below
    // writes a cell's tasks one per trip with only that cell's lane active.)
    unsigned tb = 0, lb = 0;
    if (lane == 0) {
        // (... and TWO such words per scan: the tiles with an even list position fill the task lists from the front, the others from the
        // back -- the lists' capacity is a bound on their sum, so the two ends never meet.  A scan's cursor is ONE address for all the
        // wavefronts of its grid row, atomics on one address are served one after the other (~14 ns), and 5 k tiles per scan made that
        // queue this kernel's length whatever the number of scans.)
        const bool down = it & 1u;
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(S.ctrl + (down ? 40 : 0)), (unsigned long long)TH | ((unsigned long long)nL << 32));
        tb = (unsigned)old, lb = (unsigned)(old >> 32);
        if (down) tb = (unsigned)S.maxTasks - tb - TH, lb = (unsigned)S.maxLight - lb - nL;
    }
    tb = __shfl(tb, 0), lb = __shfl(lb, 0);
    // (the capacities are upper bounds of the lists' lengths; a position beyond them would mean a dropped task and silently wrong counts:
    // it raises the scan's overflow word, which the host can read back -- MODEST_PP4_CHECK=1 -- and the tests do)
    if (lane == 0 && (tb + TH > (unsigned)S.maxTasks || lb + nL > (unsigned)S.maxLight || (int)tb < 0 || (int)lb < 0)) atomicOr(&S.ctrl[42], 1u);
    {
        unsigned *head = reinterpret_cast<unsigned *>(S.ltHead);
        if (isL) {
            const unsigned task = lb + (li >> 2), slot = li & 3u;
            if (task < (unsigned)S.maxLight) {
                *reinterpret_cast<uint2 *>(head + (size_t)task * 8 + slot * 2) = make_uint2(start, n);
                S.ltSegs[((size_t)task * 4 + slot) * 2] = make_uint4(sa[0], sn[0], sa[1], sn[1]);
                S.ltSegs[((size_t)task * 4 + slot) * 2 + 1] = make_uint4(sa[2], sn[2], 0u, 0u);
            }
        }
        const unsigned padN = nL * 4 - nLc;   // empty cells of the tile's last task
        if ((unsigned)lane < padN) {
            const unsigned q = nLc + (unsigned)lane, task = lb + (q >> 2);
            if (task < (unsigned)S.maxLight) *reinterpret_cast<uint2 *>(head + (size_t)task * 8 + (q & 3u) * 2) = make_uint2(0u, 0u);
        }
    }
    uint4 *out = reinterpret_cast<uint4 *>(S.tasks);
    const unsigned first = tb + incT - th;
    for (unsigned k = 0; k < th; ++k) {
        const unsigned idx = first + k;
        if (idx < (unsigned)S.maxTasks) {
            const unsigned st = start + k * B4_TASK;
            out[2 * (size_t)idx] = make_uint4(st, min(start + n, st + B4_TASK), sa[0], sn[0]);
            out[2 * (size_t)idx + 1] = make_uint4(sa[1], sn[1], sa[2], sn[2]);
        }
    }
    } while (false);
    bC = bN, bN = bNN, mC = mN, tC = tN, sC = sN;
    }
}

// ---- which scan a join workgroup works on --------------------------------------------------------
// The scans of a block differ by up to 1.5 x in work (a scan in the middle of the block reads two boundary segments per cell, the
// scenery changes along the shard); with the same number of workgroups per scan the grid rows of the light scans ended early and
// their CUs idled: the wavefronts were present for 88 % of the join's span on average.  One wavefront (lane = scan) shares the NW
// workgroups out in proportion to the planned work (one-cell and four-cell tasks, weighted by their measured cost), at least one each.
constexpr int B4_DEAL_WH = 4, B4_DEAL_WL = 5;   // relative cost of a one-cell / a four-cell task (MODEST_PP4_DBG=512: 41 / 50 ns per wavefront)
__global__ __launch_bounds__(64) void b4_deal(Blk B, const ScanDev *__restrict__ scans, unsigned NW) {
    const int lane = threadIdx.x, G = B.G;
    unsigned long long w = 0;
    if (lane < G) {
        const ScanDev &S = scans[lane];
        const unsigned nHu = min(S.ctrl[0], (unsigned)S.maxTasks), nLu = min(S.ctrl[1], (unsigned)S.maxLight);
        const unsigned nH = nHu + min(S.ctrl[40], (unsigned)S.maxTasks - nHu), nL = nLu + min(S.ctrl[41], (unsigned)S.maxLight - nLu);
        if (S.ctrl[0] + S.ctrl[40] > (unsigned)S.maxTasks || S.ctrl[1] + S.ctrl[41] > (unsigned)S.maxLight) atomicOr(&S.ctrl[42], 2u);   // (the lists' two ends met)
        w = (unsigned long long)nH * B4_DEAL_WH + (unsigned long long)nL * B4_DEAL_WL + 1ULL;
    }
    unsigned long long cum = w;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(cum, o);
        if (lane >= o) cum += u;
    }
    const unsigned long long total = __shfl(cum, 63);
    // end of the scan's range of workgroups: proportional, then strictly increasing (one workgroup each at least) and <= NW
    long long b = lane < G ? (long long)((cum * NW + total / 2) / total) : (long long)NW;
    if (lane == G - 1) b = NW;
    long long c = max(b, 1LL) - lane;   // prefix maximum of (b - lane), + lane: strictly increasing
    for (int o = 1; o < 64; o <<= 1) {
        const long long u = __shfl_up(c, o);
        if (lane >= o) c = max(c, u);
    }
    const unsigned end = (unsigned)min(c + lane, (long long)NW - (G - 1 - lane));
    unsigned first = __shfl_up(end, 1);
    if (lane == 0) first = 0;
    if (lane < G) {
        B.deal[lane] = first;
        if (lane == G - 1) B.deal[G] = NW;
        for (unsigned i = first; i < end; ++i) B.deal[65 + i] = (unsigned)lane;
    }
}

// ---- join ------------------------------------------------------------------------------------------
// pointers that reach a kernel through a device table are generic to the compiler (flat_load: counted against
// the LDS counter as well, so every LDS wait also waits for them); the join states that they are global -- and, for what
// an earlier kernel of the call wrote and this one only reads through wave-uniform addresses (task descriptors, the scan's
// sorted live points), CONSTANT: those become scalar loads (s_load: the candidate of a pair step arrives in SGPRs)
typedef float v4f __attribute__((ext_vector_type(4)));      // (HIP's float4 is a class: no address-space qualified copies)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define B4_GLOBAL(T) const __attribute__((address_space(1))) T *
#define B4_CONST(T) const __attribute__((address_space(4))) T *
typedef __attribute__((address_space(1))) int *B4_CNT;   // the scan's counts: global atomics, no return value
__device__ __forceinline__ void b4_count_add(B4_CNT p, int v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// counts[live point][lane's traversal] += v with the row's address in SGPRs (the live point is wave-uniform) and the lane's byte
// offset in one VGPR: the compiler's own form adds the scalar row offset to a per-lane 64-bit pointer with a VALU instruction
// per candidate -- in a loop that is bound by them
__device__ __forceinline__ void b4_count_add_row(B4_CNT counts, int rowBytes, unsigned laneBytes, int v) {
    B4_CNT row = (B4_CNT)((__attribute__((address_space(1))) char *)counts + (unsigned)rowBytes);
    asm volatile("global_atomic_add %0, %1, %2" : : "v"(laneBytes), "v"(v), "s"(row) : "memory");
}
template <typename T> __device__ __forceinline__ B4_GLOBAL(T) b4_global(const T *p) {
    return (B4_GLOBAL(T))(p);
}
template <typename T> __device__ __forceinline__ B4_CONST(T) b4_const(const T *p) {
    return (B4_CONST(T))(p);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned *B4_TICKET;
constexpr unsigned B4_TK = 2;   // tasks per ticket (1: 48 k atomics per scan on ONE word, ~10 ns each -- they serialise with few scans; 4 and more:
                                // consecutive tasks share a cell and its candidates, i.e. their cost: the balance suffers.  Measured 1 / 2 / 4 / 8 / 16:
                                // 86.7 / 84.1 / 91.6 / 95.5 / 106.9 us per scan in blocks of 16, 241 / 189 / 193 / 214 / 273 in blocks of 4)
// one ticket of a scan's task queue (lane 0 asks; the value is read when the wavefront's current batch runs out)
__device__ __forceinline__ unsigned b4_ticket_request(B4_TICKET p, int lane) {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// popcount(x) + acc in ONE instruction (the compiler prefers independent popcounts and a three-operand add tree: three more
// VALU instructions per candidate in a loop that is bound by them)
// min(m, |a|, |b|) in ONE instruction (the compiler's fminf(fabsf()) canonicalises both operands first: seven instructions)
__device__ __forceinline__ float b4_min3abs(float m, float a, float b) {
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned b4_bcnt(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// The pair phase of a one-cell task: 2 * NP chunks of 64 records in registers against the live points [ia, ie) of the scan's
// sorted array -- one candidate per step, read with a SCALAR load one step ahead (wave-uniform address: the point arrives in
// SGPRs; no LDS window, no barrier).  The chunks are held in PAIRS (v2f: chunk 2p in .x, chunk 2p+1 in .y), so that
// v_pk_add / v_pk_mul / v_pk_fma_f32 test a candidate against two chunks per instruction.  Counts the pairs with
// d2 < r2lo and reports the pairs inside the band [r2lo, r2hi] (re-tested exactly by b4_pairs_band; practically never).
// Lane t adds the hits of traversal t (segment masks sLo / sHi per chunk) to counts[live point][t]: one global atomic
// instruction per (candidate, task), at most T lanes of it active.
typedef float v8f __attribute__((ext_vector_type(8)));
// One candidate (q: wave-uniform, in SGPRs) against the task's chunks.  e = |q - h|^2 - r^2 comes out of the fused chain
// fma(dz, dz, fma(dy, dy, fma(dx, dx, -r2f))) (three packed fma per chunk pair: the threshold costs no instruction), a pair
// with e < -eps is a neighbour for certain (float32 evaluation is within 3.6e-7 r^2 of the float64 value scipy's predicate
// compares -- the subtraction's, the three fma's and r2f's roundings -- and eps = 2e-6 r^2), a pair with e > eps is none, and the
// pairs in between are found by ONE per-lane minimum of |e| per chunk pair (v_min3_f32 with |.| modifiers: no second compare,
// no scalar mask logic per candidate) that is looked at once per task: b4_pairs_band then re-tests, exactly, every pair the loop
// did not count.  Returns the hit count of this lane's traversal.
template <int NP, bool ODD>
__device__ __forceinline__ unsigned b4_pair_step(float qx_, float qy_, float qz_, const v2f *hx, const v2f *hy, const v2f *hz,
                                                 const unsigned *sLo, const unsigned *sHi, v2f nr2, float neps, float *bm) {
    const v2f qx = {qx_, qx_}, qy = {qy_, qy_}, qz = {qz_, qz_};
    unsigned acc = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const v2f dx = qx - hx[p], dy = qy - hy[p], dz = qz - hz[p];
        v2f e = __builtin_elementwise_fma(dx, dx, nr2);
        e = __builtin_elementwise_fma(dy, dy, e);
        e = __builtin_elementwise_fma(dz, dz, e);
        const unsigned long long hA = __ballot(e.x < neps), hB = __ballot(e.y < neps);
        *bm = b4_min3abs(*bm, e.x, e.y);
        // (a chunk pair without a hit could skip these eight instructions behind a scalar branch: measured, 5 % slower)
        acc = b4_bcnt((unsigned)hA & sLo[2 * p], acc);
        acc = b4_bcnt((unsigned)(hA >> 32) & sHi[2 * p], acc);
        acc = b4_bcnt((unsigned)hB & sLo[2 * p + 1], acc);
        acc = b4_bcnt((unsigned)(hB >> 32) & sHi[2 * p + 1], acc);
    }
    if (ODD) {   // a task of one or three chunks: the last chunk on its own
        const float dx = qx_ - hx[NP].x, dy = qy_ - hy[NP].x, dz = qz_ - hz[NP].x;
        const float e = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, nr2.x)));
        const unsigned long long hA = __ballot(e < neps);
        *bm = b4_min3abs(*bm, e, e);
        acc = b4_bcnt((unsigned)hA & sLo[2 * NP], acc);
        acc = b4_bcnt((unsigned)(hA >> 32) & sHi[2 * NP], acc);
    }
    return acc;
}
// the pairs the loop above did not count (e >= -eps), exactly: scipy's float64 predicate on those with e <= eps (the others are
// further than r for certain).  A separate pass over the task's candidates, entered practically never (a few tasks per scan): its
// float64 temporaries must not live in the registers of the loop above.  e is the SAME float32 chain as above (v_pk_fma_f32 and
// v_fma_f32 round alike), so the two passes split the pairs without overlap.
__device__ __forceinline__ bool b4_band_pair(float hx, float hy, float hz, float qx, float qy, float qz, float nr2, float neps, double r2) {
    const float fx = qx - hx, fy = qy - hy, fz = qz - hz;
    const float e = fmaf(fz, fz, fmaf(fy, fy, fmaf(fx, fx, nr2)));
    return !(e < neps) && e <= -neps && pp_within(hx, hy, hz, qx, qy, qz, r2);
}
template <int NP, bool ODD>
__device__ __forceinline__ void b4_pairs_band(B4_CONST(v4f) sorted, B4_CNT counts, unsigned ia, unsigned ie, const v2f *hx,
                                              const v2f *hy, const v2f *hz, const unsigned *sLo, const unsigned *sHi, float nr2,
                                              float neps, double r2, unsigned laneBytes) {
#pragma unroll 1
    for (unsigned i = ia; i < ie; ++i) {
        const v4f q = sorted[i];
        unsigned acc = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned long long xA = __ballot(b4_band_pair(hx[p].x, hy[p].x, hz[p].x, q.x, q.y, q.z, nr2, neps, r2));
            const unsigned long long xB = __ballot(b4_band_pair(hx[p].y, hy[p].y, hz[p].y, q.x, q.y, q.z, nr2, neps, r2));
            acc += __popc((unsigned)xA & sLo[2 * p]) + __popc((unsigned)(xA >> 32) & sHi[2 * p]);
            acc += __popc((unsigned)xB & sLo[2 * p + 1]) + __popc((unsigned)(xB >> 32) & sHi[2 * p + 1]);
        }
        if (ODD) {
            const unsigned long long xA = __ballot(b4_band_pair(hx[NP].x, hy[NP].x, hz[NP].x, q.x, q.y, q.z, nr2, neps, r2));
            acc += __popc((unsigned)xA & sLo[2 * NP]) + __popc((unsigned)(xA >> 32) & sHi[2 * NP]);
        }
        if (acc) b4_count_add_row(counts, __float_as_int(q.w), laneBytes, (int)acc);
    }
}
// The pair phase of a one-cell task: 2 * NP (+ 1) chunks of 64 records in registers -- in PAIRS (v2f: chunk 2p in .x, chunk
// 2p+1 in .y), so that v_pk_add / v_pk_fma_f32 test a candidate against two chunks per instruction -- against the task's
// candidates: up to three runs [aR, aR + nR) of the scan's sorted live points, the non-empty ones first (b4_plan).  One
// candidate per step, two per trip from ONE 32-byte SCALAR load (wave-uniform address: the points arrive in SGPRs; no LDS
// window, no barrier) requested a trip ahead -- across the runs: the last trip of a run requests the first candidates of the
// next one, and the task's very first load (q) was issued before the transform.  (Rounds 4-5 started every run with a load
// the loop then waited for: 2.3 candidates per run on average, i.e. most trips began with a wait.)
// Lane t adds the hits of traversal t (segment masks sLo / sHi per chunk: zero for lanes >= T) to counts[live point][t]: one
// global atomic instruction per (candidate, task), at most T lanes of it active.
template <int NP, bool ODD>
__device__ __forceinline__ void b4_pairs_rows(B4_CONST(v4f) sorted, B4_CNT counts, v8f q, const unsigned *aR, const unsigned *nR,
                                              const v2f *hx, const v2f *hy, const v2f *hz, const unsigned *sLo,
                                              const unsigned *sHi, float r2f, float eps, double r2, unsigned laneBytes) {
    const v2f nr2 = {-r2f, -r2f};
    const float neps = -eps;
    float bm = 3.0e38f;
#pragma unroll 1
    for (int rr = 0; rr < 3; ++rr) {
        const unsigned ia = __builtin_amdgcn_readfirstlane(aR[rr]), n = __builtin_amdgcn_readfirstlane(nR[rr]);
        if (n == 0u) break;
        const unsigned ie = ia + n;
        // where the last trip of this run prefetches: the next run's first candidates (the run's own last one if there is none)
        const unsigned nn = rr < 2 ? __builtin_amdgcn_readfirstlane(nR[rr < 2 ? rr + 1 : 2]) : 0u;
        const unsigned nxt = nn ? __builtin_amdgcn_readfirstlane(aR[rr < 2 ? rr + 1 : 2]) : ie - 1u;
#pragma unroll 1
        for (unsigned i = ia; i < ie; i += 2) {
            const v8f qn = *(B4_CONST(v8f))(sorted + (i + 2 < ie ? i + 2 : nxt));   // in flight during this trip
            const unsigned accA = b4_pair_step<NP, ODD>(q[0], q[1], q[2], hx, hy, hz, sLo, sHi, nr2, neps, &bm);
            if (accA) b4_count_add_row(counts, __float_as_int(q[3]), laneBytes, (int)accA);
            if (i + 1 < ie) {
                const unsigned accB = b4_pair_step<NP, ODD>(q[4], q[5], q[6], hx, hy, hz, sLo, sHi, nr2, neps, &bm);
                if (accB) b4_count_add_row(counts, __float_as_int(q[7]), laneBytes, (int)accB);
            }
            q = qn;
        }
    }
    if (__ballot(bm <= eps)) {
#pragma unroll 1
        for (int rr = 0; rr < 3; ++rr) {
            const unsigned ia = __builtin_amdgcn_readfirstlane(aR[rr]), ie = ia + __builtin_amdgcn_readfirstlane(nR[rr]);
            b4_pairs_band<NP, ODD>(sorted, counts, ia, ie, hx, hy, hz, sLo, sHi, -r2f, neps, r2, laneBytes);
        }
    }
}

// one chunk (a sparse cell: < 64 records) against the live points [ia, ie): the candidate in SGPRs, one per trip
__device__ __forceinline__ unsigned long long b4_pairs1(B4_CONST(v4f) sorted, B4_CNT counts, unsigned ia, unsigned ie, float hx, float hy,
                                                        float hz, unsigned sLo, unsigned sHi, float r2lo, float r2hi, int lq, int T) {
    unsigned long long band = 0;
    if (ia >= ie) return band;
    v4f q = sorted[ia];
#pragma unroll 1
    for (unsigned i = ia; i < ie; ++i) {
        const v4f qn = sorted[min(i + 1, ie - 1)];
        const float dx = q.x - hx, dy = q.y - hy, dz = q.z - hz;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const unsigned long long hA = __ballot(d2 < r2lo), mA = __ballot(d2 <= r2hi);
        band |= hA ^ mA;
        unsigned acc = b4_bcnt((unsigned)hA & sLo, 0u);
        acc = b4_bcnt((unsigned)(hA >> 32) & sHi, acc);
        if (lq < T && acc) b4_count_add_row(counts, __float_as_int(q.w), 4u * (unsigned)lq, (int)acc);
        q = qn;
    }
    return band;
}
__device__ __forceinline__ void b4_pairs1_band(B4_CONST(v4f) sorted, B4_CNT counts, unsigned ia, unsigned ie, float hx, float hy, float hz,
                                               unsigned sLo, unsigned sHi, float r2lo, float r2hi, double r2, int lq, int T) {
#pragma unroll 1
    for (unsigned i = ia; i < ie; ++i) {
        const v4f q = sorted[i];
        const float fx = q.x - hx, fy = q.y - hy, fz = q.z - hz;
        const float dA = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
        const bool bA = !(dA < r2lo) && dA <= r2hi;
        const unsigned long long xA = __ballot(bA && pp_within(hx, hy, hz, q.x, q.y, q.z, r2));
        const unsigned acc = __popc((unsigned)xA & sLo) + __popc((unsigned)(xA >> 32) & sHi);
        if (lq < T && acc) b4_count_add_row(counts, __float_as_int(q.w), 4u * (unsigned)lq, (int)acc);
    }
}
// a sparse cell of a four-cell task: its three candidate runs
__device__ __forceinline__ void b4_cell_rows(B4_CONST(v4f) sorted, B4_CNT counts, v4u g0, v4u g1, float hx, float hy, float hz, unsigned sLo,
                                             unsigned sHi, float r2lo, float r2hi, double r2, int lq, int T) {
    const unsigned aR[3] = {g0.x, g0.z, g1.x}, nR[3] = {g0.y, g0.w, g1.y};
    unsigned long long band = 0;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) band |= b4_pairs1(sorted, counts, aR[rr], aR[rr] + nR[rr], hx, hy, hz, sLo, sHi, r2lo, r2hi, lq, T);
    if (band) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) b4_pairs1_band(sorted, counts, aR[rr], aR[rr] + nR[rr], hx, hy, hz, sLo, sHi, r2lo, r2hi, r2, lq, T);
    }
}

constexpr int B4_JW = B4_JT / 64;   // wavefronts of a join workgroup: they share the scan's pose table and nothing else
// LDS of a join workgroup: [pose table of the scan: 48 bytes of pose + 1 byte of traversal per union frame | per wavefront:
// traversal masks of a task's chunks (B4_CPT x T words of 64 bits)]
__host__ __device__ __forceinline__ unsigned b4_pose_bytes(int U) {
    return (unsigned)(((U * 48 + 15) & ~15) + ((U + 15) & ~15));
}
constexpr unsigned B4_LDS_LIMIT = 160u * 1024u;   // per workgroup (one per CU)
__host__ __device__ __forceinline__ unsigned b4_join_lds(int U, int T, bool lpose, bool compact = false) {
    return (lpose ? b4_pose_bytes(U) : 0u) + (unsigned)B4_JW * (unsigned)(B4_CPT * T * 8) + (compact ? (unsigned)B4_JW * 64u * B4_CPT * 16u : 0u);
}
// Every wavefront works on its own: tasks from the scan's queue (a wavefront's first two by position, the others by ticket),
// no workgroup barrier after the pose table is in place, no LDS window of live points.  A workgroup is as large as a CU holds
// wavefronts of this kernel (1024 threads at 128 registers): one pose table per CU.
template <bool LPOSE, bool PROF>
__global__ __launch_bounds__(B4_JT, B4_WPE_) void b4_join(Blk B, const ScanDev *__restrict__ scans, double r2, int dbg, unsigned long long *prof) {
    extern __shared__ __align__(16) unsigned char dynsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int lq = lane;
    asm volatile("" : "+v"(lq));   // (an opaque copy of the lane id: keeps the mask constants out of long-lived registers)
    const int scanIdx = __builtin_amdgcn_readfirstlane((int)B.deal[65 + blockIdx.x]);   // (b4_deal: workgroups in proportion to the scan's planned work)
    const unsigned wgFirst = B.deal[scanIdx], wgCount = B.deal[scanIdx + 1] - wgFirst;
    const int T = scans[scanIdx].T;   // (the scan's own: a block may mix traversal counts, the LDS is sized with the largest)
    const unsigned poseB = LPOSE ? b4_pose_bytes(B.U) : 0u;
    const float4 *poseL = reinterpret_cast<const float4 *>(dynsm);
    const signed char *travL = reinterpret_cast<const signed char *>(dynsm + ((B.U * 48 + 15) & ~15));
    unsigned long long *smask = reinterpret_cast<unsigned long long *>(dynsm + poseB) + (size_t)wv * (B4_CPT * T);
    // the wavefront's slot for packing a task's member records (4 KB; B.compact: the workgroup's LDS has room for 16 of them)
    float4 *compactL = B.compact ? reinterpret_cast<float4 *>(dynsm + poseB + (size_t)B4_JW * (B4_CPT * T * 8)) + (size_t)wv * (64 * B4_CPT) : nullptr;
    const float r2lo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)(r2 * (1.0 - 1e-6)))));   // (the four-cell tasks' band)
    const float r2hi = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)(r2 * (1.0 + 1e-6)))));
    // (the one-cell tasks': b4_pair_step; wave-uniform values the compiler computes on the vector side -- stated scalar, they
    // are SGPR operands of the packed fma and the compares instead of three registers per lane)
    const float r2f = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)r2)));
    const float epsf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)(r2 * 2e-6))));
    B4_GLOBAL(v4f) rec = b4_global(reinterpret_cast<const v4f *>(B.recB));
    const unsigned TK = (dbg >> 16) & 255 ? (unsigned)((dbg >> 16) & 255) : B4_TK;   // (MODEST_PP4_TK: tasks per ticket, experiments)
    const unsigned W = (unsigned)__builtin_amdgcn_readfirstlane((int)(wgCount * B4_JW));
    const unsigned w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x - wgFirst) * B4_JW + (unsigned)wv));   // (wave-uniform: task descriptors and candidates are scalar loads)

    // PROF (MODEST_PP4_DBG=512): wall time of this wavefront by phase -- 0 pose table, 1 one-cell tasks: transform (incl. the wait
    // for the records), 2 masks, 3 pair phase, 4 four-cell tasks: segs + transform, 5 masks, 6 pair phase; 8 / 9 task counts
    unsigned long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = PROF ? wall_clock64() : 0ULL;
    const unsigned long long pstart = plast;
#define B4_TICK(kk)                                                 \
    if (PROF) {                                                     \
        const unsigned long long now_ = wall_clock64(); \
        pacc[kk] += now_ - plast;                                   \
        plast = now_;                                               \
    }
    // (A workgroup that moved on to the scan with the most tasks left once its own queues were empty -- a scan's work differs by up
    // to 1.5 x inside a block -- balanced the wavefronts' end times to 90 % of the span and changed nothing: the kernel is bound by
    // instruction issue, not by its tail; it cost 15 registers.  Measured and removed.)
    {
    const ScanDev &SC = scans[scanIdx];
    B4_CONST(v4f) sortedC = b4_const(reinterpret_cast<const v4f *>(SC.sorted));
    B4_GLOBAL(v4f) pose = b4_global(reinterpret_cast<const v4f *>(SC.pose));
    B4_CONST(v4u) tasks = b4_const(reinterpret_cast<const v4u *>(SC.tasks));
    B4_CNT counts = (B4_CNT)(SC.counts);
    B4_TICKET ticketH = (B4_TICKET)(SC.ctrl + 32), ticketL = (B4_TICKET)(SC.ctrl + 48);
    // the task lists were filled from both ends (b4_plan): queue position -> list position
    const unsigned nHu = (unsigned)__builtin_amdgcn_readfirstlane((int)min(SC.ctrl[0], (unsigned)SC.maxTasks));
    const unsigned nLu = (unsigned)__builtin_amdgcn_readfirstlane((int)min(SC.ctrl[1], (unsigned)SC.maxLight));
    const unsigned nH = nHu + (unsigned)__builtin_amdgcn_readfirstlane((int)min(SC.ctrl[40], (unsigned)SC.maxTasks - nHu));
    const unsigned nL = nLu + (unsigned)__builtin_amdgcn_readfirstlane((int)min(SC.ctrl[41], (unsigned)SC.maxLight - nLu));
    const unsigned backH = (unsigned)SC.maxTasks - 1u + nHu, backL = (unsigned)SC.maxLight - 1u + nLu;
    auto posH = [&](unsigned t) { return (size_t)(t < nHu ? t : backH - t); };
    auto posL = [&](unsigned t) { return (size_t)(t < nLu ? t : backL - t); };
    if (LPOSE) {   // the scan's poses: read once per workgroup
        float4 *pw = reinterpret_cast<float4 *>(dynsm);
        signed char *tw = reinterpret_cast<signed char *>(dynsm + ((B.U * 48 + 15) & ~15));
        for (int f = tid; f < B.U; f += B4_JT) {
            const v4f a = pose[4 * (size_t)f], b = pose[4 * (size_t)f + 1], c = pose[4 * (size_t)f + 2], d = pose[4 * (size_t)f + 3];
            pw[3 * f] = make_float4(a.x, a.y, a.z, a.w);
            pw[3 * f + 1] = make_float4(b.x, b.y, b.z, b.w);
            pw[3 * f + 2] = make_float4(c.x, c.y, c.z, c.w);
            tw[f] = (signed char)__float_as_int(d.x);
        }
        __syncthreads();
    }
    B4_TICK(0)
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

    // The two queues in an order that depends on the wavefront: the first LF wavefronts of a workgroup start with the four-cell
    // tasks, the others with the one-cell tasks.  Blocks of 16 scans: LF = 0 (measured 0 / 2 / 4 / 6 / 8: 82.8 / 84.6 / 86.7 / 85.1 /
    // 87.2 us per scan); blocks of 4, where a scan's queues are spread over 64 workgroups: 191 / 192 / 186 / 184 / 177 -> LF = 8.
    const bool lightFirst = __builtin_amdgcn_readfirstlane(wv) < ((dbg >> 24) & 31);   // (wave-uniform: stated, it lives in a scalar register)
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
    const bool doLight = (ph == 0) == lightFirst;
    // ======== one-cell tasks: the records of the NEXT task are requested before the pair phase of the current one ========
    if (!doLight && !(dbg & 8)) {
        // The deal is DYNAMIC: a wavefront's first two tasks are w and w + W, every later one is a ticket of the scan's queue
        // (2 W + the counter; a ticket is a batch of B4_TK consecutive tasks), asked for a batch ahead.
        // (A static deal w, w + W, ... left the wavefronts ending anywhere between 50 % and 100 % of the kernel's span: a task's
        // cost is its number of candidates, and a wavefront's 130 tasks do not average that out.)
        unsigned bnext = 0, bleft = 0;   // (the rest of the wavefront's current batch)
        unsigned tk = b4_ticket_request(ticketH, lane);
        auto next_task = [&]() {
            if (bleft == 0u) {   // (the ticket was asked for a batch ago)
                bnext = 2u * W + TK * (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
                bleft = TK;
                tk = b4_ticket_request(ticketH, lane);
            }
            --bleft;
            return bnext++;
        };
        unsigned t = w0, tn = w0 + W;
        v4u c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, n0 = {0, 0, 0, 0}, n1 = {0, 0, 0, 0};
        v4f R[B4_CPT];
        if (t < nH) {
            c0 = tasks[2 * posH(t)], c1 = tasks[2 * posH(t) + 1];
            const unsigned start = c0.x, end = c0.y;
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u) R[u] = __builtin_nontemporal_load(&rec[min(start + u * 64 + lane, end - 1)]);
            if (tn < nH) n0 = tasks[2 * posH(tn)], n1 = tasks[2 * posH(tn) + 1];
        }
        while (t < nH) {
            const unsigned start = c0.x, end = c0.y;
            int nch = (int)((end - start + 63) >> 6);
            // the task's first two candidates: requested before the transform and the mask rounds, used after them
            const v8f q0 = *(B4_CONST(v8f))(sortedC + c0.z);
            // ... and the descriptor of the task after next
            const unsigned tnn = next_task();
            v4u f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
            if (tnn < nH) f0 = tasks[2 * posH(tnn)], f1 = tasks[2 * posH(tnn) + 1];
            v2f hx[B4_CPT / 2], hy[B4_CPT / 2], hz[B4_CPT / 2];   // chunk 2p in .x, chunk 2p+1 in .y
            unsigned sLo[B4_CPT], sHi[B4_CPT];
            int tv[B4_CPT];
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u) {
                float ax, ay, az;
                xform(R[u], start + u * 64 + lane < end, &ax, &ay, &az, &tv[u]);
                if (u & 1) hx[u / 2].y = ax, hy[u / 2].y = ay, hz[u / 2].y = az;
                else hx[u / 2].x = ax, hy[u / 2].x = ay, hz[u / 2].x = az;
                // (poses from memory: 13 registers per chunk in flight -- two chunks at a time keep the variant free of scratch)
                if (!LPOSE && u == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (PROF) {   // (the transform's results exist: the wait for the records ends here)
                asm volatile("" ::"v"(hx[0].x), "v"(hx[1].y));
                ++pacc[8];
            }
            B4_TICK(1)
            // Records of frames that are not the scan's own sit between its records wherever windows are chosen as the reference chooses
            // them (every third frame of a slow traversal, traversals that are absent for this scan: 2.2 x the scan's entries in its
            // slot range on bench.py's realistic shard, profiles/r06_*): the task's member records are packed into as few chunks as
            // they need -- through the wavefront's LDS slot, positions by ballot prefix -- and a task without any is left at once.
            if (compactL != nullptr) {
                unsigned long long mb[B4_CPT];
                unsigned total = 0, cb[B4_CPT];
#pragma unroll
                for (int u = 0; u < B4_CPT; ++u) {
                    mb[u] = __ballot(tv[u] >= 0);
                    cb[u] = total;
                    total += (unsigned)__popcll(mb[u]);
                }
                const int nchNew = (int)((total + 63u) >> 6);
                if (nchNew < nch) {   // (wave-uniform)
                    if (nchNew > 0) {
#pragma unroll
                        for (int u = 0; u < B4_CPT; ++u) {
                            const unsigned pos = cb[u] + __builtin_amdgcn_mbcnt_hi((unsigned)(mb[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb[u], 0u));
                            const float x = (u & 1) ? hx[u / 2].y : hx[u / 2].x, y = (u & 1) ? hy[u / 2].y : hy[u / 2].x, z = (u & 1) ? hz[u / 2].y : hz[u / 2].x;
                            if (tv[u] >= 0) compactL[pos] = make_float4(x, y, z, __int_as_float(tv[u]));
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int u = 0; u < B4_CPT; ++u) {
                            const unsigned idx = (unsigned)(u * 64 + lane);
                            const float4 v = compactL[min(idx, total - 1u)];
                            const bool ok = idx < total;
                            const float x = ok ? v.x : 1.0e30f, y = ok ? v.y : 0.f, z = ok ? v.z : 0.f;
                            tv[u] = ok ? __float_as_int(v.w) : -1;
                            if (u & 1) hx[u / 2].y = x, hy[u / 2].y = y, hz[u / 2].y = z;
                            else hx[u / 2].x = x, hy[u / 2].x = y, hz[u / 2].x = z;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    nch = nchNew;
                }
            }
            // traversal masks of the chunks through LDS: every record ORs its lane bit into the word of its traversal, lane t
            // reads the word of traversal t (the LDS executes a wavefront's instructions in order; three rounds for the four
            // chunks: clear, OR, read)
            if (lq < T) {
#pragma unroll
                for (int u = 0; u < B4_CPT; ++u) smask[u * T + lq] = 0ULL;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u)
                if (tv[u] >= 0) atomicOr(reinterpret_cast<unsigned *>(&smask[u * T + tv[u]]) + (lane >> 5), 1u << (lane & 31));   // (32-bit: the word of the lane's half)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u) {
                const unsigned long long mv = lq < T ? smask[u * T + lq] : 0ULL;
                sLo[u] = (unsigned)mv;
                sHi[u] = (unsigned)(mv >> 32);
            }
            __builtin_amdgcn_wave_barrier();
            if (PROF) asm volatile("" ::"v"(sLo[0]), "v"(sHi[3]));
            B4_TICK(2)
            const unsigned aR[3] = {c0.z, c1.x, c1.z}, nR[3] = {c0.w, c1.y, c1.w};
            // the next task: its record loads are in flight during the pair phase below; the task after it: its descriptor
            if (tn < nH) {
                const unsigned s2 = n0.x, e2 = n0.y;
#pragma unroll
                for (int u = 0; u < B4_CPT; ++u) R[u] = __builtin_nontemporal_load(&rec[min(s2 + u * 64 + lane, e2 - 1)]);
            }
            if (!(dbg & 1) && nch > 0) {
                static_assert(B4_CPT == 4, "one specialisation of the pair loop per number of chunk pairs");
                const unsigned laneBytes = 4u * (unsigned)lq;
                // (the masks' LDS reads and the first candidates' scalar load are waited for HERE: left to the compiler, the wait sits inside
                // the pair loop -- the counter is shared with the loop's own prefetch, which it would then wait for in the same trip)
                asm volatile("" ::"v"(sLo[0]), "v"(sLo[1]), "v"(sLo[2]), "v"(sLo[3]), "v"(sHi[0]), "v"(sHi[1]), "v"(sHi[2]), "v"(sHi[3]), "s"(q0[0]));
                if (nch == 4) b4_pairs_rows<2, false>(sortedC, counts, q0, aR, nR, hx, hy, hz, sLo, sHi, r2f, epsf, r2, laneBytes);
                else if (nch == 2) b4_pairs_rows<1, false>(sortedC, counts, q0, aR, nR, hx, hy, hz, sLo, sHi, r2f, epsf, r2, laneBytes);
                else if (nch == 3) b4_pairs_rows<1, true>(sortedC, counts, q0, aR, nR, hx, hy, hz, sLo, sHi, r2f, epsf, r2, laneBytes);
                else b4_pairs_rows<0, true>(sortedC, counts, q0, aR, nR, hx, hy, hz, sLo, sHi, r2f, epsf, r2, laneBytes);
            }
            B4_TICK(3)
            t = tn, tn = tnn;
            c0 = n0, c1 = n1;
            n0 = f0, n1 = f1;
        }
    }
    // ======== sparse cells (< 64 records of the scan): four cells to a task, a cell per chunk, each with its own candidates ========
    if (doLight && !(dbg & 2)) {
        B4_CONST(v4u) lhead = b4_const(reinterpret_cast<const v4u *>(SC.ltHead));
        B4_CONST(v4u) lsegs = b4_const(reinterpret_cast<const v4u *>(SC.ltSegs));
        unsigned bnext = 0, bleft = 0;   // (the same dynamic deal)
        unsigned tk = b4_ticket_request(ticketL, lane);
        auto next_task = [&]() {
            if (bleft == 0u) {
                bnext = 2u * W + TK * (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
                bleft = TK;
                tk = b4_ticket_request(ticketL, lane);
            }
            --bleft;
            return bnext++;
        };
        unsigned t = w0, tn = w0 + W;
        v4u h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0}, m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
        v4f R[B4_CPT];
        auto request = [&](const v4u a, const v4u b) {   // a lane without a record re-reads the cell's last one (an empty cell: record 0)
            R[0] = __builtin_nontemporal_load(&rec[a.x + min((unsigned)lane, max(a.y, 1u) - 1u)]);
            R[1] = __builtin_nontemporal_load(&rec[a.z + min((unsigned)lane, max(a.w, 1u) - 1u)]);
            R[2] = __builtin_nontemporal_load(&rec[b.x + min((unsigned)lane, max(b.y, 1u) - 1u)]);
            R[3] = __builtin_nontemporal_load(&rec[b.z + min((unsigned)lane, max(b.w, 1u) - 1u)]);
        };
        if (t < nL) {
            h0 = lhead[2 * posL(t)], h1 = lhead[2 * posL(t) + 1];
            request(h0, h1);
            if (tn < nL) m0 = lhead[2 * posL(tn)], m1 = lhead[2 * posL(tn) + 1];
        }
        while (t < nL) {
            const unsigned nn[B4_CPT] = {h0.y, h0.w, h1.y, h1.w};
            float hx[B4_CPT], hy[B4_CPT], hz[B4_CPT];
            unsigned sLo[B4_CPT], sHi[B4_CPT];
            int tv[B4_CPT];
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u) {
                xform(R[u], (unsigned)lane < nn[u], &hx[u], &hy[u], &hz[u], &tv[u]);
                if (!LPOSE && u == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (PROF) {
                asm volatile("" ::"v"(hx[0]), "v"(hx[3]));
                ++pacc[9];
            }
            B4_TICK(4)
            int lq4 = lane;
            asm volatile("" : "+v"(lq4));   // (an opaque lane id: the mask words' addresses are formed here, per task, instead of living in registers across the whole kernel)
            if (lq4 < T) {
#pragma unroll
                for (int u = 0; u < B4_CPT; ++u) smask[u * T + lq4] = 0ULL;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u)
                if (tv[u] >= 0) atomicOr(reinterpret_cast<unsigned *>(&smask[u * T + tv[u]]) + (lane >> 5), 1u << (lane & 31));   // (32-bit: the word of the lane's half)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < B4_CPT; ++u) {
                const unsigned long long mv = lq4 < T ? smask[u * T + lq4] : 0ULL;
                sLo[u] = (unsigned)mv;
                sHi[u] = (unsigned)(mv >> 32);
            }
            __builtin_amdgcn_wave_barrier();
            if (PROF) asm volatile("" ::"v"(sLo[0]), "v"(sHi[3]));
            B4_TICK(5)
            v4u g[2 * B4_CPT];
#pragma unroll
            for (int u = 0; u < 2 * B4_CPT; ++u) g[u] = lsegs[8 * posL(t) + u];
            if (tn < nL) request(m0, m1);
            const unsigned tnn = next_task();
            v4u f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
            if (tnn < nL) f0 = lhead[2 * posL(tnn)], f1 = lhead[2 * posL(tnn) + 1];
            if (!(dbg & 1)) {
#pragma unroll
                for (int u = 0; u < B4_CPT; ++u)
                    if (nn[u]) b4_cell_rows(sortedC, counts, g[2 * u], g[2 * u + 1], hx[u], hy[u], hz[u], sLo[u], sHi[u], r2lo, r2hi, r2, lq, T);
            }
            B4_TICK(6)
            t = tn, tn = tnn;
            h0 = m0, h1 = m1;
            m0 = f0, m1 = f1;
        }
    }
    }
    }
    if (PROF && lane == 0) {
        for (int kk = 0; kk < 10; ++kk) atomicAdd(&prof[kk], pacc[kk]);
        // (recA is free once the store is sorted: absolute start / end of every wavefront, for the tail statistics)
        unsigned long long *wt = reinterpret_cast<unsigned long long *>(B.recA) + 2 * ((size_t)blockIdx.x * B4_JW + wv);
        wt[0] = pstart, wt[1] = wall_clock64();
    }
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

static int pp_block_impl(modest_ctx *ctx, const modest_pp_block_frame *frames, int n_frames, const modest_pp_block_scan *scans,
                         int n_scans, const int32_t *Ts, double radius, double cell, void *stream_);

extern "C" int modest_pp_score_block(modest_ctx *ctx, const modest_pp_block_frame *frames, int n_frames,
                                     const modest_pp_block_scan *scans, int n_scans, int n_trav, double radius,
                                     double cell, void *stream_) {
    MODEST_REQUIRE(n_scans >= 1 && n_scans <= 64, "1 <= n_scans <= 64");
    int32_t Ts[64];
    for (int s = 0; s < n_scans; ++s) Ts[s] = n_trav;
    return pp_block_impl(ctx, frames, n_frames, scans, n_scans, Ts, radius, cell, stream_);
}

// The scans of a block need not have the same number of traversals: the reference accepts a traversal PER SCAN (closest pose
// within 3 m, data_preprocessing/lyft/split_traintest.py:17,79) and only asks for two of them (:111), so T changes along a
// sequence.  Everything per-scan on the device already reads its scan's own T (counts (n, T_s), the entropy's ln T_s, the
// traversal masks of the join); the join's LDS is sized with the block's largest T.
extern "C" int modest_pp_score_block_mixed(modest_ctx *ctx, const modest_pp_block_frame *frames, int n_frames,
                                           const modest_pp_block_scan *scans, int n_scans, const int32_t *n_trav_scan,
                                           double radius, double cell, void *stream_) {
    MODEST_REQUIRE(n_trav_scan != nullptr, "NULL argument");
    return pp_block_impl(ctx, frames, n_frames, scans, n_scans, n_trav_scan, radius, cell, stream_);
}

static int pp_block_impl(modest_ctx *ctx, const modest_pp_block_frame *frames, int n_frames, const modest_pp_block_scan *scans,
                         int n_scans, const int32_t *Ts, double radius, double cell, void *stream_) {
    MODEST_REQUIRE(ctx != nullptr && scans != nullptr, "NULL argument");
    MODEST_REQUIRE(n_scans >= 1 && n_scans <= 64, "1 <= n_scans <= 64");
    MODEST_REQUIRE(n_frames >= 0 && n_frames < (1 << 16) && (n_frames == 0 || frames != nullptr), "bad frame table");
    int Tmax = 1;
    for (int s = 0; s < n_scans; ++s) {
        MODEST_REQUIRE(Ts[s] >= 1 && Ts[s] <= B4_MAXT, "1 <= n_trav <= 64 on the block path");
        Tmax = std::max(Tmax, (int)Ts[s]);
    }
    MODEST_REQUIRE(radius > 0.0 && radius < 1e6, "radius must be positive and finite");
    // the lattice cell must leave room for the difference between distances on the lattice and in a scan's frame
    MODEST_REQUIRE(cell >= radius * (1.0 + 1.0 / 512.0) && cell <= radius * 1.25,
                   "lattice cell edge must be in [r (1 + 2^-9), 1.25 r]");
    hipStream_t stream = as_stream(stream_);
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int G = n_scans, U = n_frames;
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
            MODEST_REQUIRE(sc.member_trav[m] >= 0 && sc.member_trav[m] < Ts[s], "frame traversal out of range");
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
    // a cell with n >= 64 records of the scan gives ceil(ceil(n / 64) / 4) <= n / 64 tasks
    const size_t maxTasks = (size_t)(ntot / 64) + 16;
    // a sparse cell has a live point in the 3x3 cells around it: at most 9 cells per live point, four cells to a task (+ one partly
    // filled task per tile)
    const size_t maxLight = std::min<size_t>((size_t)(ntot / 4), (size_t)9 * (size_t)maxN / 4) + (size_t)BT + 16;

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
    const size_t oCellOff = take((size_t)BT * 65 * 4), oSegRange = take(maxSegs * 8);
    const size_t oBaseSum = take((size_t)((BT + 1023) / 1024) * 8), oNeed = take((size_t)BT * 4), oNeedCand = take((size_t)BT * 4);
    const size_t oRecA = take((size_t)std::max<long long>(ntot, 1) * 16), oRecB = take((size_t)std::max<long long>(ntot, 1) * 16);
    const size_t oDeal = take((size_t)(65 + 4 * ctx->num_cus + 64) * 4);   // b4_deal's tables
    const size_t oCtrl = take(256);   // the block's cursors and crop ...
    const size_t oNeedMask = take((size_t)BT * 8);   // ... and, directly behind them, the tiles' scan masks: one memset clears both
    const size_t oCellCount = take((size_t)G * (NCpad + 4) * 4);   // contiguous over the scans: one memset
    struct ScanOff {
        size_t cellStart, blockSum, ctrl, tmp, sorted, tasks, ltHead, ltSegs, counts;
    };
    std::vector<ScanOff> so((size_t)G);
    for (int s = 0; s < G; ++s) {
        const int n = scans[s].n;
        so[(size_t)s].cellStart = take((size_t)(NCpad + 4) * 4);
        so[(size_t)s].blockSum = take((size_t)(nScanBlk + 1) * 4);
        so[(size_t)s].ctrl = take(256);
        so[(size_t)s].tmp = take((size_t)std::max(n, 1) * 16);
        so[(size_t)s].sorted = take((size_t)(std::max(n, 1) + 2) * 16);   // (b4_pairs reads candidates in pairs: up to one point past a run)
        so[(size_t)s].tasks = take(maxTasks * sizeof(B4Task));
        so[(size_t)s].ltHead = take(maxLight * 32);
        so[(size_t)s].ltSegs = take(maxLight * 128);
        so[(size_t)s].counts = take((size_t)std::max(n, 1) * Ts[s] * 4);
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
        d.ctrl = reinterpret_cast<unsigned *>(base + o.ctrl);
        d.tmp = reinterpret_cast<float4 *>(base + o.tmp);
        d.sorted = reinterpret_cast<float4 *>(base + o.sorted);
        d.tasks = base + o.tasks;
        d.ltHead = reinterpret_cast<uint4 *>(base + o.ltHead);
        d.ltSegs = reinterpret_cast<uint4 *>(base + o.ltSegs);
        d.pose = reinterpret_cast<const PoseEnt *>(dstage + stPose) + (size_t)s * std::max(U, 1);
        d.counts = sc.counts_dev ? sc.counts_dev : reinterpret_cast<int *>(base + o.counts);
        d.H = sc.H_dev;
        for (int q = 0; q < 8; ++q) d.lat[q] = sc.lat[q];
        for (int q = 0; q < 12; ++q) d.rel[q] = sc.rel[q];
        d.n = sc.n;
        d.TX0 = sc.TX0;
        d.TY0 = sc.TY0;
        d.T = Ts[s];
        d.maxTasks = (int)maxTasks;
        d.maxLight = (int)maxLight;
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
    B.needList = reinterpret_cast<unsigned *>(base + oNeed);
    B.needCand = reinterpret_cast<unsigned *>(base + oNeedCand);
    B.needMask = reinterpret_cast<unsigned long long *>(base + oNeedMask);
    B.deal = reinterpret_cast<unsigned *>(base + oDeal);
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
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS_LIMIT));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS_LIMIT));
        MODEST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(b4_join<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS_LIMIT));
        attr_done[ctx->device & 63] = true;
    }
    modest_prof_mark(ctx, stream, 0);   // bench.py: the whole neighbour-count stage of the block
    MODEST_HIP_CHECK(hipMemsetAsync(base + oCtrl, 0, 256 + (size_t)BT * 8, stream));   // (the block's cursors and crop, the tiles' scan masks)
    const unsigned gBT = (unsigned)((BT + 255) / 256), gN = (unsigned)((maxN + 255) / 256);
    if (U > 0 && ntot > 0) {
        b4_need<<<dim3(gBT, (unsigned)((G + B4_NEED_SCANS - 1) / B4_NEED_SCANS)), 256, 0, stream>>>(B, dsc);
        b4_need_list<<<gBT, 256, 0, stream>>>(B);
        b4_counts<<<dim3(gBT, (unsigned)NG), 256, 0, stream>>>(B);   // (over the needed tiles: the workgroups behind them leave at once)
        b4_lists<<<gBT, 256, 0, stream>>>(B);
        b4_bases_local<<<(unsigned)((BT + 1023) / 1024), 1024, 0, stream>>>(B);
        b4_bases_finish<<<(unsigned)((BT + 1023) / 1024), 1024, 0, stream>>>(B);
        const int swg = std::min((int)nch, 3 * ctx->num_cus);
        b4_scatter<<<swg, 512, 0, stream>>>(B);
        b4_seg_hist<<<(unsigned)maxSegs, 512, 0, stream>>>(B);
        b4_seg_scan<<<(unsigned)((BT + 3) / 4), 256, 0, stream>>>(B);
        b4_seg_scatter<<<(unsigned)maxSegs, 512, B4_SEG * 16, stream>>>(B);
        b4_zero_cells<<<dim3((unsigned)nScanBlk, (unsigned)G), 1024, 0, stream>>>(B, dsc);
        b4_live_count<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(B, dsc);
        b4_scan_local<<<dim3((unsigned)nScanBlk, (unsigned)G), 1024, 0, stream>>>(B, dsc);
        b4_scan_finish<<<(unsigned)G, 1024, 0, stream>>>(B, dsc);
        b4_live_scatter<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(B, dsc);
        b4_plan<<<dim3(std::max(8u, (unsigned)(8 * ctx->num_cus) / (unsigned)G), (unsigned)G), 256, 0, stream>>>(B, dsc);
        const char *jw_env = getenv("MODEST_PP4_JWG");
        unsigned jx = (unsigned)((jw_env ? std::min(atoi(jw_env), 4) : 1) * ctx->num_cus) / (unsigned)G;   // one workgroup of 16 wavefronts per CU
        if (jx < 2) jx = 2;
        const unsigned NW = jx * (unsigned)G;   // (<= 4 * num_cus + 64: the deal table's size)
        b4_deal<<<1, 64, 0, stream>>>(B, dsc, NW);
        const char *dbg_env = getenv("MODEST_PP4_DBG");   // ablations: 1 no pair phase, 2 no four-cell tasks, 8 no one-cell tasks, 256 poses from memory, 512 phase times
        const char *tk_env = getenv("MODEST_PP4_TK");
        const char *lf_env = getenv("MODEST_PP4_LF");   // wavefronts of a join workgroup that start with the four-cell tasks
        const int lf = lf_env ? atoi(lf_env) & 31 : (G <= 6 ? 8 : 0);
        const int dbg = (dbg_env ? atoi(dbg_env) : 0) | ((tk_env ? atoi(tk_env) & 255 : 0) << 16) | (lf << 24);
        const bool lpose = U <= B4_POSE_LDS_MAX && !(dbg & 256);
        const char *cp_env = getenv("MODEST_PP4_COMPACT");   // (0: no packing of member records -- A/B)
        // Packing pays where a scan's slot range holds many entries that are not its own (reference-rule windows: every third frame of a
        // slow traversal, absent traversals; +2 ... +7 % on such shards) and costs 1 % where it holds none (windows of frames i..i+F-1)
        double rangeSum = 0, memberSum = 0;
        for (int sc = 0; sc < G; ++sc)
            if (scans[sc].n_members > 0) rangeSum += hsc[sc].slotHi - hsc[sc].slotLo + 1, memberSum += scans[sc].n_members;
        const bool sparse = cp_env ? atoi(cp_env) != 0 : rangeSum > 1.2 * memberSum;
        B.compact = sparse && b4_join_lds(U, Tmax, lpose, true) <= B4_LDS_LIMIT;
        const unsigned ldsB = b4_join_lds(U, Tmax, lpose, B.compact != 0);
        if ((dbg & 512) && lpose) {   // MODEST_PP4_DBG=512: wall time of the join's wavefronts by phase (blocking; diagnostics only)
            unsigned long long *dprof = reinterpret_cast<unsigned long long *>(base + oCtrl + 64), hprof[10];
            MODEST_HIP_CHECK(hipMemsetAsync(dprof, 0, sizeof(hprof), stream));
            b4_join<true, true><<<NW, B4_JT, ldsB, stream>>>(B, dsc, radius * radius, dbg, dprof);
            MODEST_HIP_CHECK(hipStreamSynchronize(stream));
            MODEST_HIP_CHECK(hipMemcpy(hprof, dprof, sizeof(hprof), hipMemcpyDeviceToHost));
            {   // when do the wavefronts end?  (share of the kernel's span a wavefront is present, per scan and overall)
                const size_t nw = (size_t)jx * G * B4_JW;
                const double us = 1.0 / 100.0;   // wall_clock64 ticks at 100 MHz
                if (nw * 16 <= (size_t)ntot * 16) {
                    std::vector<unsigned long long> wt(2 * nw);
                    MODEST_HIP_CHECK(hipMemcpy(wt.data(), base + oRecA, nw * 16, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ULL, t1 = 0;
                    for (size_t i = 0; i < nw; ++i) t0 = std::min(t0, wt[2 * i]), t1 = std::max(t1, wt[2 * i + 1]);
                    std::vector<double> en(nw);
                    double sum = 0, sumStart = 0;
                    for (size_t i = 0; i < nw; ++i) en[i] = (double)(wt[2 * i + 1] - t0), sum += en[i], sumStart += (double)(wt[2 * i] - t0);
                    std::sort(en.begin(), en.end());
                    fprintf(stderr, "[b4_join] span %.1f us; wavefront end: mean %.1f, p10 %.1f, p50 %.1f, p90 %.1f, p99 %.1f us; mean start %.1f us\n",
                            (double)(t1 - t0) * us, sum / nw * us, en[nw / 10] * us, en[nw / 2] * us, en[nw * 9 / 10] * us, en[nw * 99 / 100] * us, sumStart / nw * us);
                }
            }
            const double wv = (double)jx * G * B4_JW, us = 1.0 / 100.0;   // s_memtime ticks at 100 MHz
            fprintf(stderr, "[b4_join] per wavefront, us: pose table %.1f | one-cell tasks: records + transform %.1f, masks %.1f, pairs %.1f | "
                            "four-cell tasks: records + transform %.1f, masks %.1f, pairs %.1f || per scan: one-cell tasks %.0f, four-cell tasks %.0f\n",
                    hprof[0] * us / wv, hprof[1] * us / wv, hprof[2] * us / wv, hprof[3] * us / wv, hprof[4] * us / wv, hprof[5] * us / wv,
                    hprof[6] * us / wv, (double)hprof[8] / G, (double)hprof[9] / G);
        } else if (lpose) b4_join<true, false><<<NW, B4_JT, ldsB, stream>>>(B, dsc, radius * radius, dbg, nullptr);
        else b4_join<false, false><<<NW, B4_JT, ldsB, stream>>>(B, dsc, radius * radius, dbg, nullptr);
    } else {   // no history: every count is zero
        for (int sc = 0; sc < G; ++sc)
            if (scans[sc].n > 0) MODEST_HIP_CHECK(hipMemsetAsync(hsc[sc].counts, 0, (size_t)scans[sc].n * Ts[sc] * 4, stream));
    }
    modest_prof_mark(ctx, stream, 1);
    b4_entropy<<<dim3(gN, (unsigned)G), 256, 0, stream>>>(dsc);
    if (const char *chk = getenv("MODEST_PP4_CHECK"); chk && atoi(chk) && U > 0 && ntot > 0) {   // (blocking; tests: no task list overflowed)
        MODEST_HIP_CHECK(hipStreamSynchronize(stream));
        for (int sc = 0; sc < G; ++sc) {
            unsigned ov = 0;
            MODEST_HIP_CHECK(hipMemcpy(&ov, reinterpret_cast<unsigned *>(base + so[(size_t)sc].ctrl) + 42, 4, hipMemcpyDeviceToHost));
            MODEST_REQUIRE(ov == 0u, "b4_plan: a task list overflowed its capacity");
        }
    }
    MODEST_HIP_CHECK(hipGetLastError());
    return MODEST_OK;
}

// modest_warmup (ctx.hip): resolving one kernel of this translation unit makes the runtime load its code object now
extern "C" void modest_warm_pp_v4(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(b4_entropy));
}
