// Order-preserving stream compaction helper (numpy boolean-mask semantics).
//
// Each 1024-thread block takes a LOGICAL index from a ticket counter when it
// starts and chains a running total through one 64-bit word
// (state[0] = (#blocks published << 40) | total, state[1] = ticket), so a block
// only ever waits on blocks that are already running: no assumption about
// dispatch order, residency or XCD placement.  state[] (16 bytes) must be
// zeroed on the stream before the launch.
#pragma once
#include <hip/hip_runtime.h>

namespace modest {

struct CompactSlot {
    unsigned blk;             // logical block index (use instead of blockIdx.x)
    unsigned long long dst;   // output position of this thread's element (valid iff keep)
};

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
    if (threadIdx.x == 0) {
        unsigned long long s;
        for (;;) {
            s = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((s >> 40) == (unsigned long long)blk) break;
            __builtin_amdgcn_s_sleep(2);
        }
        base_s = s & ((1ULL << 40) - 1ULL);
        const unsigned long long ns = ((unsigned long long)(blk + 1) << 40) | (base_s + total);
        __hip_atomic_store(state, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blk == nblocks - 1 && total_out) *total_out = (int)(base_s + total);
    }
    __syncthreads();
    return base_s + woff + before;
}

}  // namespace modest
