// Order-preserving stream compaction helper (numpy boolean-mask semantics).
//
// Each 1024-thread block takes a LOGICAL index from a ticket counter when it
// starts (state[1]) and publishes its own keep-count at once in state[2 + blk]
// (bit 40 = published).  Its first wavefront then sums the counts of all
// logically earlier blocks, 64 at a time: a block only ever waits on blocks that
// are already running (no assumption about dispatch order, residency or XCD
// placement), and nobody waits on a chain -- the 30 blocks of a scan used to hand
// a running total from one to the next, 0.6 us per hop.
// state[] (compact_state_bytes(nblocks)) must be zero before the launch; the block that finishes
// its look-back last (state[0] counts them) zeroes it again, so a buffer that is only ever used by
// these kernels (modest_ctx_compact_state) needs no memset per launch.
#pragma once
#include <hip/hip_runtime.h>

namespace modest {

struct CompactSlot {
    unsigned blk;             // logical block index (use instead of blockIdx.x)
    unsigned long long dst;   // output position of this thread's element (valid iff keep)
};

__host__ __device__ inline size_t compact_state_bytes(size_t nblocks) { return 16 + 8 * nblocks; }

__device__ __forceinline__ unsigned compact_ticket(unsigned long long *state) {
    __shared__ unsigned ticket_s;
    if (threadIdx.x == 0) ticket_s = atomicAdd(reinterpret_cast<unsigned *>(state + 1), 1u);
    __syncthreads();
    return ticket_s;
}

// Call from ALL 1024 threads of the block.  `total_out` (may be NULL) receives
// the grand total from the last logical block.
__device__ __forceinline__ unsigned long long compact_offset(bool keep, unsigned blk,
                                                             unsigned nblocks,
                                                             unsigned long long *state,
                                                             int *total_out) {
    __shared__ unsigned wave_cnt[16];
    __shared__ unsigned long long base_s;
    __shared__ unsigned last_s;
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned before = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) wave_cnt[w] = __popcll(bal);
    __syncthreads();
    unsigned woff = 0, total = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < w) woff += wave_cnt[k];
        total += wave_cnt[k];
    }
    if (w == 0) {
        constexpr unsigned long long FLAG = 1ULL << 40;
        if (lane == 0)
            __hip_atomic_store(state + 2 + blk, FLAG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long sum = 0;
        for (unsigned j0 = 0; j0 < blk; j0 += 64) {
            const unsigned j = j0 + lane;
            if (j < blk) {
                unsigned long long s;
                for (;;) {
                    s = __hip_atomic_load(state + 2 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s & FLAG) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                sum += s & (FLAG - 1ULL);
            }
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) {
            base_s = sum;
            if (blk == nblocks - 1 && total_out) *total_out = (int)(sum + total);
            // every block that has counted itself here is done reading state[]
            last_s = atomicAdd(state, 1ULL) == (unsigned long long)(nblocks - 1) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (last_s) {
        for (unsigned j = threadIdx.x; j < nblocks + 2; j += blockDim.x)
            __hip_atomic_store(state + j, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return base_s + woff + before;
}

}  // namespace modest
