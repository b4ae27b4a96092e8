// PP neighbour count, V3: list scatter + LDS-resident slice sort + cell-uniform join.
// Included by pp_count.hip (inside its anonymous namespace, after the index-build kernels).
//
// What the measurements of V2 said: only 59 M (record, live point) pairs have to be tested for a
// Lyft-shape scan (6.1 M surviving history records x 9.7 candidates on average), but with records
// in arrival order every wavefront runs to its slowest lane (2.7x the useful tests) and the
// per-record set-up (descriptor search, cell table, candidate index arithmetic) costs more
// instructions than the tests.  V3 therefore SORTS the records so that a wavefront holds
// records of one cell, which makes the candidate list wave-uniform: live points are read once
// per wavefront (LDS broadcast), hits are counted with ballots + popcounts instead of one LDS
// atomic per pair, and there is no per-lane control flow at all.
//
//   pp3_blocks    one workgroup: live points in the 10x10-cell window of every 8x8-cell block;
//                 the densest blocks are split into their four 4x4-cell quadrants (own lists).
//   pp3_stream<0> stream the history (HBM, coalesced 48 B per lane), test the dilated occupancy
//                 bitmap in LDS, histogram the survivors by list (6400 blocks + quadrants of the
//                 dense ones) in LDS; one row of counts per workgroup.  No barrier in the loop.
//   pp3_scan      exclusive scan of the count matrix along the workgroup axis (LDS transpose);
//   pp3_plan      one workgroup: list bases and the slice list (dense lists first, then ring by
//                 ring from the grid centre, so that the expensive slices are dequeued first).
//   pp3_stream<1> stream the history again (Infinity Cache: 130 MB < 256 MB), same static
//                 chunk -> workgroup map: records go straight to their exact position in the
//                 list-contiguous array (LDS cursor per list, 16-byte stores).  No barrier either.
//   pp3_join      persistent grid, one slice at a time, everything in LDS: the slice's records
//                 (<= 4096, loaded once, kept in registers across the histogram) are counting-
//                 sorted by cell into LDS, the block's live points (+1 cell halo) sit next to them,
//                 64-record chunks of the sorted slice are dealt to wavefronts; a chunk is joined one
//                 cell group at a time (chunks spanning many cells: one lane per record instead).
//                 Per-(live point, traversal) counters are 16-bit pairs in LDS, flushed with one
//                 global atomic per non-zero counter.  Two dependent global round trips per slice.
//
// Why small lists: the cost of a slice is (chunks) x (candidates of the cells they touch), so
// lists must be small for balance, yet a slice must hold >= 64 records PER CELL for the
// wavefronts to be full: 4096 records over 64 cells, 2048 over the 16 cells of a dense quadrant.
//
// Blocks whose live points (+halo) do not fit next to the records are processed in bands of cell
// rows; a band that does not fit even as a single row falls back to per-record loops over the
// global index (correct, slow, never seen on LiDAR-shaped input).

constexpr int V3_TS = 8;                    // block edge in cells
constexpr int V3_NT = PP_NX / V3_TS;        // 80
constexpr int V3_NBLK = V3_NT * V3_NT;      // 6400 blocks = base lists
constexpr int V3_DMAX = 192;                // dense blocks that get one list per quadrant
constexpr int V3_NL = V3_NBLK + 4 * V3_DMAX;   // 7168 lists
constexpr int V3_NC = V3_TS * V3_TS;        // 64 cells per block = sort keys of a slice
constexpr int V3_W = V3_TS + 2;             // 10: window incl. halo
constexpr int V3_CH = 4096;                 // history points per chunk (1024 threads x 4)
constexpr int V3_MAXT = 64;                 // traversals handled by the routed path (6 bits, one lane each)
constexpr int V3_MAXWG = 512;               // stream workgroups
constexpr int V3_JT = 1024;                 // threads of a join workgroup
constexpr int V3_RPT = 4;                   // records per join thread
constexpr unsigned V3_SLICE_MAX = V3_JT * V3_RPT;   // 4096 records per slice (16-bit counters cannot overflow)
constexpr unsigned V3_SLICE_MIN = 1024;     // records always guaranteed to fit next to a band
constexpr int V3_JOIN_LDS_DYN = 76 * 1024;  // records + live points + counters
constexpr unsigned V3_DENSE_LIVE = 192;     // window live points above which a block is split
constexpr unsigned V3_LANE_GROUPS = 2;      // chunks spanning at least this many cells take the per-lane path
constexpr unsigned V3_LANE_MAX = 64;        // ... for the lanes with at most this many candidates
constexpr int V3_DWORDS = V3_NBLK / 32;     // 200 words of dense-block flags
static_assert(PP_NX % V3_TS == 0 && V3_NC == 64, "cell key = 6 bits, one lane per cell in the scan");
static_assert(V3_NL % 64 == 0 && V3_NL == 7 * 1024, "scan / plan tiling");
static_assert(V3_NBLK % 32 == 0, "dense flag words");

struct ChunkMap3 {
    int cstart[PP_MAX_TRAV + 1];   // first chunk id of each traversal (chunks never straddle)
};

// Base lists are numbered ring by ring from the grid centre outwards (the grid is centred on the
// live scan, whose dense part is the centre): slices are generated and dequeued in list order,
// so the expensive ones are started first and the cheap border lists fill the tail.
// Ring k (0..39) holds the blocks with max(|dx|,|dy|) = k and starts at list 4k^2.
static_assert(V3_NT % 2 == 0, "ring numbering assumes an even number of blocks per axis");
__device__ __forceinline__ int pp3_list_of(int tx, int ty) {
    constexpr int H = V3_NT / 2;
    const int a = tx < H ? H - 1 - tx : tx - H, b = ty < H ? H - 1 - ty : ty - H;
    const int k = max(a, b), e = 2 * k + 1;
    const int u = tx - (H - 1 - k), v = ty - (H - 1 - k);   // in [0, e]
    int pos;
    if (v == 0) pos = u;
    else if (v == e) pos = e + 1 + u;
    else pos = 2 * (e + 1) + (u == 0 ? 0 : e - 1) + (v - 1);
    return 4 * k * k + pos;
}
__device__ __forceinline__ void pp3_tile_of(int list, int *tx, int *ty) {
    constexpr int H = V3_NT / 2;
    int k = (int)(sqrtf((float)list) * 0.5f);
    while (4 * (k + 1) * (k + 1) <= list) ++k;
    while (4 * k * k > list) --k;
    const int e = 2 * k + 1, pos = list - 4 * k * k;
    int u, v;
    if (pos <= e) {
        u = pos;
        v = 0;
    } else if (pos < 2 * (e + 1)) {
        u = pos - (e + 1);
        v = e;
    } else {
        const int q = pos - 2 * (e + 1);
        u = q < e - 1 ? 0 : e;
        v = (q < e - 1 ? q : q - (e - 1)) + 1;
    }
    *tx = u + (H - 1 - k);
    *ty = v + (H - 1 - k);
}

// ---- blocks ---------------------------------------------------------------------------
// dense[0 .. 200)   flag words (bit b = block b, row-major, is split into quadrants)
// dense[200 .. 400) number of dense blocks before each word
// denseBlock[d]     block of dense index d;   listLive[l] window live points of list l's block
__device__ __forceinline__ void pp3_block_live_one(int b, const unsigned *__restrict__ cellStart,
                                                   unsigned *__restrict__ blockLive) {
    if (b >= V3_NBLK) return;
    const int bx = b % V3_NT, by = b / V3_NT;
    const int gx0 = max(bx * V3_TS - 1, 0), gx1 = min(bx * V3_TS + V3_TS + 1, PP_NX);
    unsigned lw = 0;
#pragma unroll
    for (int r = 0; r < V3_W; ++r) {   // twenty independent loads
        const int gy = by * V3_TS - 1 + r;
        const int gyc = min(max(gy, 0), PP_NY - 1);
        const unsigned v = cellStart[gyc * PP_NX + gx1] - cellStart[gyc * PP_NX + gx0];
        lw += (gy == gyc) ? v : 0u;
    }
    blockLive[b] = lw;
}

// last step of the live index build, one launch for two independent jobs: blocks [0, nb) place
// the live points into their cells, the others count the live points of every block window
__device__ __forceinline__ void pp3_scatter_blocklive_body(const float *__restrict__ live, int n, const unsigned *bb, double c, const unsigned *__restrict__ cellStart, unsigned *fill, float4 *__restrict__ sorted, int nb, unsigned *__restrict__ blockLive, const unsigned bx, const unsigned gx) {
    if ((int)bx < nb)
        pp_live_scatter_one(bx * 256 + threadIdx.x, live, n, bb, c, cellStart, fill, sorted);
    else
        pp3_block_live_one(((int)bx - nb) * 256 + threadIdx.x, cellStart, blockLive);
}
__global__ __launch_bounds__(256) void pp3_scatter_blocklive(const float *__restrict__ live, int n, const unsigned *bb, double c, const unsigned *__restrict__ cellStart, unsigned *fill, float4 *__restrict__ sorted, int nb, unsigned *__restrict__ blockLive) {
    pp3_scatter_blocklive_body(live, n, bb, c, cellStart, fill, sorted, nb, blockLive, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void pp3_blocks_body(const unsigned *__restrict__ blockLive, unsigned *__restrict__ dense, unsigned *__restrict__ denseBlock, unsigned *__restrict__ listLive, const unsigned bx, const unsigned gx) {
    __shared__ unsigned bits[V3_DWORDS];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < V3_DWORDS) bits[tid] = 0;
    unsigned lw[7], nd = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int b = tid * 7 + j;
        lw[j] = b < V3_NBLK ? blockLive[b] : 0u;
        nd += lw[j] > V3_DENSE_LIVE;
    }
    unsigned inc = nd;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();   // also orders the zeroing of bits[]
    unsigned d = inc - nd;
    for (int k = 0; k < w; ++k) d += wsum[k];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int b = tid * 7 + j;
        if (b < V3_NBLK) {
            listLive[pp3_list_of(b % V3_NT, b / V3_NT)] = lw[j];
            if (lw[j] > V3_DENSE_LIVE) {
                if (d < (unsigned)V3_DMAX) {
                    atomicOr(&bits[b >> 5], 1u << (b & 31));
                    denseBlock[d] = (unsigned)b;
                    for (int q = 0; q < 4; ++q) listLive[V3_NBLK + 4 * d + q] = lw[j];
                }
                ++d;
            }
        }
    }
    __syncthreads();
    if (tid < V3_DWORDS) {
        unsigned before = 0;
        for (int k = 0; k < tid; ++k) before += __popc(bits[k]);
        dense[tid] = bits[tid];
        dense[V3_DWORDS + tid] = before;
    }
}
__global__ __launch_bounds__(1024) void pp3_blocks(const unsigned *__restrict__ blockLive, unsigned *__restrict__ dense, unsigned *__restrict__ denseBlock, unsigned *__restrict__ listLive) {
    pp3_blocks_body(blockLive, dense, denseBlock, listLive, blockIdx.x, gridDim.x);
}

// Cell of a history point, tested against the dilated bitmap: returns the list or -1;
// *key = (cell row in the block << 3) | cell column in the block.
__device__ __forceinline__ int pp3_classify(float x, float y, const PPGrid &g, const unsigned *sbits,
                                            const unsigned *sdense, int *key) {
    const int cx = pp_cell_coord(x, g.ox, g.inv_c, PP_NX);
    const int cy = pp_cell_coord(y, g.oy, g.inv_c, PP_NY);
    const int bit = cy * PP_NX + cx;
    if (!((sbits[bit >> 5] >> (bit & 31)) & 1u)) return -1;
    *key = ((cy % V3_TS) * V3_TS) | (cx % V3_TS);
    const int bx = cx / V3_TS, by = cy / V3_TS, b = by * V3_NT + bx;
    const unsigned dw = sdense[b >> 5];
    if ((dw >> (b & 31)) & 1u) {
        const unsigned d = sdense[V3_DWORDS + (b >> 5)] + __popc(dw & ((1u << (b & 31)) - 1u));
        return V3_NBLK + 4 * (int)d + (((cy % V3_TS) >> 2) << 1) + ((cx % V3_TS) >> 2);
    }
    return pp3_list_of(bx, by);
}

// Loads the (up to) four points of this thread's part of a chunk.
__device__ __forceinline__ void pp3_load4(const float *__restrict__ hist, long long q0, long long pend, float v[12]) {
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
}

// FRAMES: the history is not a stacked array but the frames of the frame store, named by a
// descriptor table; chunkTab[chunk] = (frame, first point).  The frame's relative pose is applied on
// the fly (transform_points' float32 rounding, pp_frames.h; the pose is wave-uniform: scalar loads),
// remove_center drops points before the transform (pre_compute_pp_score.py:141-142).
template <bool SCATTER, bool FRAMES>
__device__ __forceinline__ void pp3_stream_body(const float *__restrict__ hist, const TravOffsets &tr, const ChunkMap3 &cm,
                                                const FrameDev *__restrict__ frames,
                                                const uint2 *__restrict__ chunkTab, int nchunks,
                                                const unsigned *bb, double c,
                                                const unsigned *__restrict__ bitmap,
                                                const unsigned *__restrict__ dense,
                                                unsigned *__restrict__ wgTile /* [grid][NL] counts */,
                                                const unsigned *__restrict__ wgOff /* [grid][NL] offsets in the list */,
                                                const unsigned *__restrict__ tileBase,
                                                float4 *__restrict__ rec, int pair, const unsigned bx, const unsigned gx) {
    __shared__ unsigned sbits[PP_BITWORDS];
    __shared__ unsigned cur[V3_NL];
    __shared__ unsigned sdense[2 * V3_DWORDS];
    const int tid = threadIdx.x;
    for (int i = tid; i < PP_BITWORDS; i += 1024) sbits[i] = bitmap[i];
    // `pair`: the count pass runs twice as many workgroups as the scatter pass (fewer (workgroup,
    // list) write streams let the L2 merge more of the 16-byte stores: 98 -> 80 us).  Scatter
    // workgroup v takes the chunks of count workgroups v and v + G/2, whose rows 2v and 2v+1 of the
    // matrix are adjacent, so its range of every list is contiguous.
    const size_t row = !pair ? bx
                             : (SCATTER ? 2u * bx : (bx % (gx / 2)) * 2u + bx / (gx / 2));
    for (int i = tid; i < V3_NL; i += 1024) cur[i] = SCATTER ? tileBase[i] + wgOff[row * V3_NL + i] : 0u;
    if (tid < 2 * V3_DWORDS) sdense[tid] = dense[tid];
    const PPGrid g = pp_grid(bb, c);
    __syncthreads();
    for (int chunk = bx; chunk < nchunks; chunk += gx) {
        int t = 0;
        long long p0, pend;
        const float *src = hist;
        const FrameDev *fd = nullptr;
        bool center = false;
        if (FRAMES) {
            const uint2 ct = chunkTab[chunk];
            fd = frames + ct.x;
            src = fd->xyz;
            p0 = ct.y;
            pend = min((long long)fd->n, p0 + V3_CH);
            t = fd->trav_flags & 0xffff;
            center = ((fd->trav_flags >> 16) & F_FLAG_CENTER) != 0;
        } else {
            while (t + 1 < tr.n && chunk >= cm.cstart[t + 1]) ++t;
            p0 = tr.off[t] + (long long)(chunk - cm.cstart[t]) * V3_CH;
            pend = min(tr.off[t + 1], p0 + V3_CH);
        }
        const long long q0 = p0 + 4LL * tid;
        if (q0 >= pend) continue;
        float v[12];
        pp3_load4(src, q0, pend, v);
        bool dropped[4] = {false, false, false, false};
        if (FRAMES) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dropped[k] = center && in_center_box(v[3 * k], v[3 * k + 1]);
                float o[3];
                rel_apply(fd->rel, v[3 * k], v[3 * k + 1], v[3 * k + 2], o);
                v[3 * k] = o[0];
                v[3 * k + 1] = o[1];
                v[3 * k + 2] = o[2];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (q0 + k < pend && !dropped[k]) {
                int key;
                const int list = pp3_classify(v[3 * k], v[3 * k + 1], g, sbits, sdense, &key);
                if (list >= 0) {
                    if (SCATTER) {
                        const unsigned pos = atomicAdd(&cur[list], 1u);
                        rec[pos] = make_float4(v[3 * k], v[3 * k + 1], v[3 * k + 2], __int_as_float(key | (t << 16)));
                    } else {
                        atomicAdd(&cur[list], 1u);
                    }
                }
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int i = tid; i < V3_NL; i += 1024) wgTile[row * V3_NL + i] = cur[i];
    }
}
template <bool SCATTER, bool FRAMES>
__global__ __launch_bounds__(1024, 8) void pp3_stream(const float *__restrict__ hist, TravOffsets tr, ChunkMap3 cm,
                                                      const FrameDev *__restrict__ frames,
                                                      const uint2 *__restrict__ chunkTab, int nchunks,
                                                      const unsigned *bb, double c,
                                                      const unsigned *__restrict__ bitmap,
                                                      const unsigned *__restrict__ dense,
                                                      unsigned *__restrict__ wgTile,
                                                      const unsigned *__restrict__ wgOff,
                                                      const unsigned *__restrict__ tileBase,
                                                      float4 *__restrict__ rec, int pair) {
    pp3_stream_body<SCATTER, FRAMES>(hist, tr, cm, frames, chunkTab, nchunks, bb, c, bitmap, dense, wgTile, wgOff, tileBase,
                                     rec, pair, blockIdx.x, gridDim.x);
}

// ctrl3: [0] = #slices, [1] = dequeue head, [2] = total records
// pp3_scan: a workgroup owns 64 consecutive lists.  The [workgroup][list] count matrix is read in
// 256-byte pieces (coalesced), transposed through LDS, scanned along the workgroup axis in place
// and written back as offsets inside each list; list totals go to listTotal.
constexpr int V3_SCAN_L = 64;   // lists per scan workgroup
__device__ __forceinline__ void pp3_scan_body(const unsigned *__restrict__ wgTile, unsigned *__restrict__ wgOff, int nwg, unsigned *__restrict__ listTotal, const unsigned bx, const unsigned gx) {
    extern __shared__ unsigned m[];             // [nwg][64]
    __shared__ unsigned segSum[16][V3_SCAN_L];
    const int tid = threadIdx.x, j = tid & 63, seg = tid >> 6;
    const int l0 = bx * V3_SCAN_L;
    for (int k = seg; k < nwg; k += 16) m[k * V3_SCAN_L + j] = wgTile[(size_t)k * V3_NL + l0 + j];
    __syncthreads();
    const int per = (nwg + 15) / 16, k0 = seg * per, k1 = min(k0 + per, nwg);
    unsigned s = 0;
    for (int k = k0; k < k1; ++k) s += m[k * V3_SCAN_L + j];
    segSum[seg][j] = s;
    __syncthreads();
    unsigned run = 0, all = 0;
    for (int q = 0; q < 16; ++q) {
        const unsigned v = segSum[q][j];
        if (q < seg) run += v;
        all += v;
    }
    for (int k = k0; k < k1; ++k) {
        const unsigned v = m[k * V3_SCAN_L + j];
        m[k * V3_SCAN_L + j] = run;
        run += v;
    }
    if (seg == 0) listTotal[l0 + j] = all;
    __syncthreads();
    for (int k = seg; k < nwg; k += 16) wgOff[(size_t)k * V3_NL + l0 + j] = m[k * V3_SCAN_L + j];
}
__global__ __launch_bounds__(1024) void pp3_scan(const unsigned *__restrict__ wgTile, unsigned *__restrict__ wgOff, int nwg, unsigned *__restrict__ listTotal) {
    pp3_scan_body(wgTile, wgOff, nwg, listTotal, blockIdx.x, gridDim.x);
}

// LDS bytes per live point of a band: float4 + T 16-bit counters (rounded up to a 32-bit pair)
__host__ __device__ __forceinline__ unsigned pp3_live_bytes(int T) { return 16u + 4u * (unsigned)((T + 1) >> 1); }

// pp3_plan: one workgroup; list bases and the slice list, seven lists per thread in PROCESS
// order (dense quadrant lists first, then the base lists centre-out).  A slice is sized so that
// its records and the block's live points (+counters) share the LDS of one join workgroup.
__device__ __forceinline__ void pp3_plan_body(const unsigned *__restrict__ listTotal, const unsigned *__restrict__ listLive, int T, unsigned sliceCap, unsigned *__restrict__ tileBase, uint4 *__restrict__ slices, unsigned maxSlices, unsigned *ctrl3, const unsigned bx, const unsigned gx) {
    __shared__ unsigned tot[16], nsl[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned lb = pp3_live_bytes(T);
    unsigned total[7], ns[7], sumT = 0, sumS = 0;
    int list[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int i = tid * 7 + j;   // process order
        list[j] = i < 4 * V3_DMAX ? V3_NBLK + i : i - 4 * V3_DMAX;
        total[j] = listTotal[list[j]];
        const unsigned liveB = min(listLive[list[j]] * lb, (unsigned)V3_JOIN_LDS_DYN - V3_SLICE_MIN * 16u);
        const unsigned fit = ((unsigned)V3_JOIN_LDS_DYN - liveB) / 16u;   // >= V3_SLICE_MIN
        unsigned cap = max(min(min(fit, sliceCap), V3_SLICE_MAX), 64u);
        if (list[j] >= V3_NBLK) cap = min(cap, 2048u);   // dense quadrants: 16 cells, >= 128 records per cell
        ns[j] = (total[j] + cap - 1) / cap;
        sumT += total[j];
        sumS += ns[j];
    }
    unsigned incA = sumT, incB = sumS;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(incA, o), b = __shfl_up(incB, o);
        if (lane >= o) {
            incA += a;
            incB += b;
        }
    }
    if (lane == 63) {
        tot[w] = incA;
        nsl[w] = incB;
    }
    __syncthreads();
    unsigned baseA = 0, baseB = 0, allA = 0, allB = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w) {
            baseA += tot[k];
            baseB += nsl[k];
        }
        allA += tot[k];
        allB += nsl[k];
    }
    if (tid == 0) {
        ctrl3[0] = min(allB, maxSlices);
        ctrl3[2] = allA;
    }
    unsigned tBase = baseA + incA - sumT, sBase = baseB + incB - sumS;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        tileBase[list[j]] = tBase;
        const unsigned per = ns[j] ? (total[j] + ns[j] - 1) / ns[j] : 0u;   // equal parts
        for (unsigned k = 0; k < ns[j]; ++k) {
            if (sBase + k >= maxSlices) break;
            slices[sBase + k] = make_uint4((unsigned)list[j], tBase + k * per, tBase + min((k + 1) * per, total[j]), 0u);
        }
        tBase += total[j];
        sBase += ns[j];
    }
}
__global__ __launch_bounds__(1024) void pp3_plan(const unsigned *__restrict__ listTotal, const unsigned *__restrict__ listLive, int T, unsigned sliceCap, unsigned *__restrict__ tileBase, uint4 *__restrict__ slices, unsigned maxSlices, unsigned *ctrl3) {
    pp3_plan_body(listTotal, listLive, T, sliceCap, tileBase, slices, maxSlices, ctrl3, blockIdx.x, gridDim.x);
}

struct JoinShared {
    unsigned cursor[V3_NC];                   // cell histogram -> running cursor -> cell END offsets in the slice
    unsigned cst[V3_W * (V3_W + 1)];          // cellStart of the window cells (one global round trip)
    unsigned short ctab[V3_W * (V3_W + 1)];   // live points of window row r before column cc
    unsigned segStart[V3_W], rowBase[V3_W + 1];
    unsigned bandA[V3_TS], bandB[V3_TS], bandSlow[V3_TS];
    unsigned nBands, ticket, sliceId, nextId;
    uint4 slice, nextSlice;
    unsigned long long prof[8], prev[8], tlast;   // PROF builds only
};

template <bool PROF>
__device__ __forceinline__ void pp3_join_body(const float4 *__restrict__ rec,
                                              const uint4 *__restrict__ slices, unsigned *ctrl3,
                                              const unsigned *__restrict__ denseBlock,
                                              const unsigned *__restrict__ cellStart,
                                              const float4 *__restrict__ sorted, int *counts, int T,
                                              double r2, int dbg, unsigned long long *stats, const unsigned bx) {
    extern __shared__ __align__(16) unsigned char dynsm[];
    __shared__ JoinShared S;
    float4 *srec = reinterpret_cast<float4 *>(dynsm);   // the slice, sorted by cell
    const int tid = threadIdx.x, lane = tid & 63;
    const int Th = (T + 1) >> 1;   // counter words per live point (two 16-bit counters each)
    const unsigned liveBytes = pp3_live_bytes(T);
    const float r2lo = (float)(r2 * (1.0 - 1e-6)), r2hi = (float)(r2 * (1.0 + 1e-6));
    const float r2f = (float)r2, bandw = (float)(r2 * 1.5e-6);   // |d2 - r2f| <= bandw covers [r2lo, r2hi]
    const unsigned nSlices = ctrl3[0];
    // per-lane constants of the traversal-segmented popcount: lane t < T owns traversal t;
    // selk = all ones if bit k of t is CLEAR (mask of traversal t = AND_k (B_k ^ selk))
    // (recomputed for every chunk from an opaque copy of the lane id: hoisted out of the slice loop
    // they are seven registers the allocator spills)

    constexpr bool prof = PROF;
    // ablation knobs (MODEST_PP_DBG) exist in the PROF instantiation only
    const unsigned laneGroups = (PROF && ((dbg >> 8) & 0xff)) ? ((dbg >> 8) & 0xff) : V3_LANE_GROUPS;
    const unsigned laneMax = (PROF && ((dbg >> 16) & 0xfff)) ? ((dbg >> 16) & 0xfff) : V3_LANE_MAX;
    unsigned nGroups = 0, nIter = 0, nChunk = 0;
    if (prof && tid < 8) {
        S.prof[tid] = 0;
        S.prev[tid] = 0;
    }
#define PP3_TICK(k)                                     \
    if (prof && tid == 0) {                              \
        const unsigned long long now_ = wall_clock64(); \
        S.prof[k] += now_ - S.tlast;                     \
        S.tlast = now_;                                  \
    }
    if (prof && tid == 0) stats[64 + bx] = wall_clock64();
    unsigned nMine = 0;
    if (tid == 0) {   // the first slice; later ones are fetched while the previous slice is processed
        const unsigned first = atomicAdd(&ctrl3[1], 1u);
        S.sliceId = first;
        if (first < nSlices) S.slice = slices[first];
    }
    for (;;) {
        __syncthreads();
        if (prof && tid == 0) {
            S.tlast = wall_clock64();
            S.prof[7] = S.tlast;
        }
        const unsigned sid = S.sliceId;
        if (sid >= nSlices) break;
        ++nMine;
        const uint4 sl = S.slice;
        if (tid < V3_NC) S.cursor[tid] = 0;
        __syncthreads();   // everyone holds the slice: thread 0 may overwrite the header below
        // the next slice header goes through LDS, not registers: values that live across the whole
        // slice get spilled, and a kernel that touches scratch memory at all pays for it at dispatch
        if (tid == 0) S.nextId = atomicAdd(&ctrl3[1], 1u);   // in flight during the loads below
        int ttx, tty;
        if (sl.x < (unsigned)V3_NBLK) {
            pp3_tile_of((int)sl.x, &ttx, &tty);
        } else {
            const unsigned b = denseBlock[(sl.x - V3_NBLK) >> 2];
            ttx = (int)(b % V3_NT);
            tty = (int)(b / V3_NT);
        }
        const unsigned lo = sl.y, n = sl.z - sl.y;   // n <= V3_SLICE_MAX
        const int x0 = ttx * V3_TS - 1, y0 = tty * V3_TS - 1;
        const int gx0 = max(x0, 0), gx1 = min(x0 + V3_W, PP_NX);
        // live points and counters of a band follow the records
        const unsigned lcap = ((unsigned)V3_JOIN_LDS_DYN - n * 16u) / liveBytes;
        float4 *live = srec + n;
        unsigned *cntw = reinterpret_cast<unsigned *>(live + lcap);

        // ---- (a) one round trip: the window's cell table and the slice's records --------
        float4 h4[V3_RPT];
#pragma unroll
        for (int u = 0; u < V3_RPT; ++u) {
            const unsigned i = tid + u * V3_JT;
            h4[u] = i < n ? rec[lo + i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        }
        if (tid < V3_W * (V3_W + 1)) {
            const int r = tid / (V3_W + 1), cc = tid - r * (V3_W + 1);
            const int gy = y0 + r;
            unsigned val = 0;
            if (gy >= 0 && gy < PP_NY) val = cellStart[gy * PP_NX + min(max(x0 + cc, gx0), gx1)];
            S.cst[tid] = val;
        }
#pragma unroll
        for (int u = 0; u < V3_RPT; ++u) {
            const int pk = __float_as_int(h4[u].w);
            if (pk >= 0) atomicAdd(&S.cursor[pk & (V3_NC - 1)], 1u);
        }
        if (tid == 0 && S.nextId < nSlices) S.nextSlice = slices[S.nextId];
        __syncthreads();
        PP3_TICK(0)
        // ---- (b) cell offsets, window tables, bands -----------------------------------
        if (tid < 64) {   // exclusive scan of the 64 cell counts
            const unsigned c0 = S.cursor[tid];
            unsigned inc = c0;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            S.cursor[tid] = inc - c0;
        } else if (tid < 64 + V3_W * (V3_W + 1)) {   // positions inside a window row
            const int e = tid - 64, r = e / (V3_W + 1);
            S.ctab[e] = (unsigned short)min(S.cst[e] - S.cst[r * (V3_W + 1)], 65535u);
            if (e == r * (V3_W + 1)) S.segStart[r] = S.cst[e];
        } else if (tid == V3_JT - 1) {   // window rows -> prefix, bands of cell rows that fit the LDS budget
            auto len = [&](int r) { return S.cst[r * (V3_W + 1) + V3_W] - S.cst[r * (V3_W + 1)]; };
            unsigned run = 0;
            for (int r = 0; r < V3_W; ++r) {
                S.rowBase[r] = run;
                run += len(r);
            }
            S.rowBase[V3_W] = run;
            unsigned nb = 0;
            int ya = 1;
            while (ya <= V3_TS) {
                unsigned sum = len(ya - 1) + len(ya) + len(ya + 1);
                int yb = ya + 1;
                const unsigned slow = sum > lcap;
                if (!slow)
                    while (yb <= V3_TS && sum + len(yb + 1) <= lcap) {
                        sum += len(yb + 1);
                        ++yb;
                    }
                S.bandA[nb] = (unsigned)ya;
                S.bandB[nb] = (unsigned)yb;
                S.bandSlow[nb] = slow;
                ++nb;
                ya = yb;
            }
            S.nBands = nb;
        }
        __syncthreads();
        // ---- (c) scatter the records into LDS (cell-sorted) -----------------------------
#pragma unroll
        for (int u = 0; u < V3_RPT; ++u) {
            const int pk = __float_as_int(h4[u].w);
            if (pk >= 0) srec[atomicAdd(&S.cursor[pk & (V3_NC - 1)], 1u)] = h4[u];
        }
        // cursor[k] becomes the END of cell k; cell k starts at cursor[k-1]
        PP3_TICK(1)

        // ---- (d) bands of cell rows -----------------------------------------------------
        const unsigned nBands = (PROF && (dbg & 4)) ? 0u : S.nBands;
        for (unsigned b = 0; b < nBands; ++b) {
            const int ya = (int)S.bandA[b], yb = (int)S.bandB[b];
            const unsigned lbase = S.rowBase[ya - 1];
            const unsigned Lb = S.rowBase[yb + 1] - lbase;
            const bool slow = S.bandSlow[b] != 0;
            __syncthreads();   // scatter / previous band's flush complete
            const unsigned kA = (unsigned)(ya - 1) * V3_TS, kB = (unsigned)(yb - 1) * V3_TS;
            const unsigned ra = kA ? S.cursor[kA - 1] : 0u, rb = S.cursor[kB - 1];
            if (ra == rb) continue;   // uniform: no record in these rows
            if (slow) {
                // a single row of cells whose three live rows exceed the LDS budget
                for (unsigned j = ra + tid; j < rb; j += V3_JT) {
                    const float4 h = srec[j];
                    const int pk = __float_as_int(h.w);
                    const int tr = pk >> 16;
                    const int cx = x0 + 1 + (pk & (V3_TS - 1)), cy = y0 + 1 + ((pk & (V3_NC - 1)) / V3_TS);
                    const int xa = max(cx - 1, 0), xb = min(cx + 1, PP_NX - 1);
                    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, PP_NY - 1); ++yy) {
                        const unsigned a = cellStart[yy * PP_NX + xa], e = cellStart[yy * PP_NX + xb + 1];
                        for (unsigned i = a; i < e; ++i) {
                            const float4 q = sorted[i];
                            if (pp_within(h.x, h.y, h.z, q.x, q.y, q.z, r2))
                                atomicAdd(&counts[(size_t)__float_as_int(q.w) * T + tr], 1);
                        }
                    }
                }
                continue;
            }
            for (unsigned e = tid; e < Lb; e += V3_JT) {
                int r = ya - 1;
                while (e + lbase >= S.rowBase[r + 1]) ++r;
                live[e] = sorted[S.segStart[r] + (e + lbase - S.rowBase[r])];
            }
            for (unsigned e = tid; e < Lb * Th; e += V3_JT) cntw[e] = 0;
            if (tid == 0) S.ticket = 0;
            __syncthreads();
            PP3_TICK(2)

            // 64-record chunks of the sorted band range, dealt to wavefronts
            const unsigned nChunks = (rb - ra + 63) / 64;
            for (;;) {
                unsigned ck = 0;
                if (lane == 0) ck = atomicAdd(&S.ticket, 1u);
                ck = __builtin_amdgcn_readfirstlane(ck);
                if (ck >= nChunks) break;
                if (prof) ++nChunk;
                const unsigned j = ra + ck * 64 + lane;
                const bool valid = j < rb;
                const float4 h = valid ? srec[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int pk = valid ? __float_as_int(h.w) : -1;
                const int key = pk & (V3_NC - 1);
                const unsigned trv = (unsigned)(pk >> 16) & 63u;
                int lq = lane;
                asm volatile("" : "+v"(lq));
                const unsigned sel0 = (lq & 1) ? 0u : ~0u, sel1 = (lq & 2) ? 0u : ~0u;
                const unsigned sel2 = (lq & 4) ? 0u : ~0u, sel3 = (lq & 8) ? 0u : ~0u, sel4 = (lq & 16) ? 0u : ~0u;
                const unsigned cshift = (lq & 1) * 16;
                const unsigned laneWord = (unsigned)lq >> 1;
                // traversal segment masks: lane t keeps the lanes whose record belongs to traversal t
                unsigned long long seg = __ballot(valid);
                {
                    const unsigned long long B0 = __ballot(trv & 1u), B1 = __ballot(trv & 2u);
                    const unsigned long long B2 = __ballot(trv & 4u), B3 = __ballot(trv & 8u), B4 = __ballot(trv & 16u);
                    const unsigned long long s0 = ((unsigned long long)sel0 << 32) | sel0;
                    const unsigned long long s1 = ((unsigned long long)sel1 << 32) | sel1;
                    const unsigned long long s2 = ((unsigned long long)sel2 << 32) | sel2;
                    const unsigned long long s3 = ((unsigned long long)sel3 << 32) | sel3;
                    const unsigned long long s4 = ((unsigned long long)sel4 << 32) | sel4;
                    seg &= (B0 ^ s0) & (B1 ^ s1) & (B2 ^ s2) & (B3 ^ s3) & (B4 ^ s4);
                    if (T > 32) {   // sixth traversal bit only when there are that many (wave-uniform)
                        const unsigned sel5 = (lq & 32) ? 0u : ~0u;
                        seg &= __ballot(trv & 32u) ^ (((unsigned long long)sel5 << 32) | sel5);
                    }
                }
                const unsigned segLo = (unsigned)seg, segHi = (unsigned)(seg >> 32);
                unsigned long long todo = __ballot(valid);
                // Sparse chunks (many cells, few records each) would pay the group set-up once per
                // cell: there every lane walks its own candidate list instead (four LDS reads in
                // flight, one LDS atomic per hit); lanes with long lists stay on the group path.
                {
                    const int prevKey = __shfl_up(key, 1);
                    const unsigned nG = __popcll(__ballot(valid && (lane == 0 || key != prevKey)));
                    if (nG >= laneGroups && !(PROF && (dbg & 2))) {
                        const int lx = (key & (V3_TS - 1)) + 1, ly = key / V3_TS + 1;
                        const unsigned short *row = S.ctab + (ly - 1) * (V3_W + 1) + lx - 1;
                        const unsigned c00 = row[0], c10 = row[V3_W + 1], c20 = row[2 * (V3_W + 1)];
                        const unsigned n0 = row[3] - c00, n1 = row[V3_W + 4] - c10;
                        const unsigned n2 = row[2 * (V3_W + 1) + 3] - c20;
                        const unsigned a0 = S.rowBase[ly - 1] - lbase + c00;
                        const unsigned n01 = n0 + n1, nAll = n01 + n2;
                        const unsigned b1 = S.rowBase[ly] - lbase + c10 - n0;
                        const unsigned b2 = S.rowBase[ly + 1] - lbase + c20 - n01;
                        const bool mine = valid && nAll <= laneMax;
                        const unsigned own = mine ? nAll : 0u;
                        const unsigned cword = trv >> 1, cinc = 1u << ((trv & 1u) * 16);
                        for (unsigned p0 = 0; __any(p0 < own); p0 += 4) {
                            unsigned bandBits = 0;
#pragma unroll
                            for (unsigned u = 0; u < 4; ++u) {
                                const unsigned p = p0 + u;
                                const bool act = p < own;
                                const unsigned i = act ? p + (p < n0 ? a0 : (p < n01 ? b1 : b2)) : 0u;
                                const float4 q = live[i];
                                const float fx = q.x - h.x, fy = q.y - h.y, fz = q.z - h.z;
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
                                const float4 q = live[i];
                                if (pp_within(h.x, h.y, h.z, q.x, q.y, q.z, r2)) atomicAdd(&cntw[i * Th + cword], cinc);
                            }
                        }
                        todo = __ballot(valid && !mine);
                    }
                }
                while (todo) {   // one cell group at a time (wave-uniform)
                    const int src = __ffsll((long long)todo) - 1;
                    const int gkey = __builtin_amdgcn_readlane(key, src);
                    const unsigned long long grp = __ballot(valid && key == gkey);
                    todo &= ~grp;
                    const int lcx = (gkey & (V3_TS - 1)) + 1, lcy = gkey / V3_TS + 1;   // window coordinates
                    if (prof) ++nGroups;
                    if (PROF && (dbg & 2)) continue;
                    // the three candidate runs (cell rows lcy-1 .. lcy+1, columns lcx-1 .. lcx+1): all scalar
                    const unsigned short *row = S.ctab + (lcy - 1) * (V3_W + 1) + lcx - 1;
                    const unsigned rb0 = S.rowBase[lcy - 1] - lbase, rb1 = S.rowBase[lcy] - lbase;
                    const unsigned rb2 = S.rowBase[lcy + 1] - lbase;
                    const unsigned c00 = row[0], c03 = row[3], c10 = row[V3_W + 1], c13 = row[V3_W + 4];
                    const unsigned c20 = row[2 * (V3_W + 1)], c23 = row[2 * (V3_W + 1) + 3];
                    const unsigned a0 = __builtin_amdgcn_readfirstlane(rb0 + c00);
                    const unsigned n0 = __builtin_amdgcn_readfirstlane(c03 - c00);
                    const unsigned a1 = __builtin_amdgcn_readfirstlane(rb1 + c10);
                    const unsigned n1 = __builtin_amdgcn_readfirstlane(c13 - c10);
                    const unsigned a2 = __builtin_amdgcn_readfirstlane(rb2 + c20);
                    const unsigned n2 = __builtin_amdgcn_readfirstlane(c23 - c20);
                    // Row by row, four contiguous candidates per step: one address register, LDS reads
                    // with immediate offsets, hit masks straight from the compares; reads past the end
                    // of a run see the next row or the counters (ignored: their masks are forced to 0).
                    // The rare pairs inside the band around r^2 are found through a running minimum of
                    // |d2 - r^2| and re-tested in float64 after the group.
                    float dmin = 3.0e38f;
#pragma unroll 1
                    for (int rr = 0; rr < 3; ++rr) {
                        const unsigned ra_ = rr == 0 ? a0 : (rr == 1 ? a1 : a2);
                        const unsigned re_ = ra_ + (rr == 0 ? n0 : (rr == 1 ? n1 : n2));
                        for (unsigned i = ra_; i < re_; i += 4) {
                            if (prof) ++nIter;
                            const float4 q0 = live[i], q1 = live[i + 1], q2 = live[i + 2], q3 = live[i + 3];
                            float fx, fy, fz;
                            fx = q0.x - h.x, fy = q0.y - h.y, fz = q0.z - h.z;
                            const float d0 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            fx = q1.x - h.x, fy = q1.y - h.y, fz = q1.z - h.z;
                            const float d1 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            fx = q2.x - h.x, fy = q2.y - h.y, fz = q2.z - h.z;
                            const float d2_ = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            fx = q3.x - h.x, fy = q3.y - h.y, fz = q3.z - h.z;
                            const float d3 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            const unsigned long long h0 = __ballot(d0 < r2lo) & grp;
                            const unsigned long long h1 = (i + 1 < re_) ? (__ballot(d1 < r2lo) & grp) : 0ULL;
                            const unsigned long long h2 = (i + 2 < re_) ? (__ballot(d2_ < r2lo) & grp) : 0ULL;
                            const unsigned long long h3 = (i + 3 < re_) ? (__ballot(d3 < r2lo) & grp) : 0ULL;
                            dmin = fminf(dmin, fminf(fminf(fabsf(d0 - r2f), fabsf(d1 - r2f)),
                                                     fminf(fabsf(d2_ - r2f), fabsf(d3 - r2f))));
                            if (h0 | h1 | h2 | h3) {
                                const unsigned c0 = __popc((unsigned)h0 & segLo) + __popc((unsigned)(h0 >> 32) & segHi);
                                const unsigned c1 = __popc((unsigned)h1 & segLo) + __popc((unsigned)(h1 >> 32) & segHi);
                                const unsigned c2 = __popc((unsigned)h2 & segLo) + __popc((unsigned)(h2 >> 32) & segHi);
                                const unsigned c3 = __popc((unsigned)h3 & segLo) + __popc((unsigned)(h3 >> 32) & segHi);
                                if (lane < T) {   // adding zero is cheaper than testing for it
                                    unsigned *cw = cntw + i * Th + laneWord;
                                    atomicAdd(cw, c0 << cshift);
                                    if (i + 1 < re_) atomicAdd(cw + Th, c1 << cshift);
                                    if (i + 2 < re_) atomicAdd(cw + 2 * Th, c2 << cshift);
                                    if (i + 3 < re_) atomicAdd(cw + 3 * Th, c3 << cshift);
                                }
                            }
                        }
                    }
                    if (__ballot(dmin <= bandw) & grp) {   // practically never: exact float64 re-test
                        const unsigned n01 = n0 + n1, nAll = n01 + n2;
                        const unsigned b1 = a1 - n0, b2 = a2 - n01;
                        for (unsigned i = 0; i < nAll; ++i) {
                            const unsigned p = i + (i < n0 ? a0 : (i < n01 ? b1 : b2));
                            const float4 qq = live[p];
                            const float fx = qq.x - h.x, fy = qq.y - h.y, fz = qq.z - h.z;
                            const float d2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                            const bool inBand = (d2 >= r2lo) && (fabsf(d2 - r2f) <= bandw);
                            const unsigned long long hx =
                                __ballot(inBand && pp_within(h.x, h.y, h.z, qq.x, qq.y, qq.z, r2)) & grp;
                            if (hx) {
                                const unsigned cN = __popc((unsigned)hx & segLo) + __popc((unsigned)(hx >> 32) & segHi);
                                if (lane < T && cN) atomicAdd(&cntw[p * Th + laneWord], cN << cshift);
                            }
                        }
                    }
                }
            }
            __syncthreads();
            PP3_TICK(3)
            if (!(PROF && (dbg & 1)))
                for (unsigned e = tid; e < Lb * Th; e += V3_JT) {
                    const unsigned cw = cntw[e];
                    if (cw) {
                        const unsigned p = e / Th, tp = (e - p * Th) * 2;
                        const size_t row = (size_t)__float_as_int(live[p].w) * T;
                        if (cw & 0xffffu) atomicAdd(&counts[row + tp], (int)(cw & 0xffffu));
                        if (cw >> 16) atomicAdd(&counts[row + tp + 1], (int)(cw >> 16));
                    }
                }
        }
        PP3_TICK(4)
        if (prof && tid == 0) {
            const unsigned long long dur = wall_clock64() - S.prof[7];
            const unsigned long long packed = (dur << 32) | sid;
            if (atomicMax(&stats[13], packed) < packed) {
                for (int k = 0; k < 5; ++k) stats[16 + k] = S.prof[k] - S.prev[k];
                stats[21] = S.prof[7] - stats[64 + bx];   // start of the slice, relative to this WG's start
            }
            for (int k = 0; k < 5; ++k) S.prev[k] = S.prof[k];
        }
        if (tid == 0) {
            S.sliceId = S.nextId;
            S.slice = S.nextSlice;
        }
    }
    if (prof && tid == 0) {   // per-workgroup start / end / slice count
        stats[64 + 1024 + bx] = wall_clock64();
        stats[64 + 2048 + bx] = nMine;
    }
    if (prof) {
        if (tid == 0) {
            unsigned long long all = 0;
            for (int k = 0; k < 5; ++k) {
                atomicAdd(&stats[k], S.prof[k]);
                all += S.prof[k];
            }
            atomicMax(&stats[6], all);
        }
        if (lane == 0) {
            atomicAdd(&stats[8], (unsigned long long)nChunk);
            atomicAdd(&stats[9], (unsigned long long)nGroups);
            atomicAdd(&stats[10], (unsigned long long)nIter);
        }
    }
#undef PP3_TICK
}
template <bool PROF>
__global__ __launch_bounds__(V3_JT, 8) void pp3_join(const float4 *__restrict__ rec,
                                                     const uint4 *__restrict__ slices, unsigned *ctrl3,
                                                     const unsigned *__restrict__ denseBlock,
                                                     const unsigned *__restrict__ cellStart,
                                                     const float4 *__restrict__ sorted, int *counts, int T,
                                                     double r2, int dbg, unsigned long long *stats) {
    pp3_join_body<PROF>(rec, slices, ctrl3, denseBlock, cellStart, sorted, counts, T, r2, dbg, stats, blockIdx.x);
}
