// Shared host-side helpers for libmodest_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "modest_hip.h"

// The "last block finishes the reduction" kernels (plane.hip, boxfit.hip, cluster.hip, compact.h) publish
// their partial results with relaxed agent-scope atomic stores, wait for them with the s_waitcnt of
// __syncthreads(), take a relaxed ticket and read the partials back with relaxed agent-scope loads -- no
// release/acquire edge in the HIP memory model.  It is correct on gfx950 because agent-scope (sc1) stores
// are write-through past the XCD's L2 and sc1 loads bypass the CU's L1 (MI355X_MICROARCH.md, workgroup
// dispatch section: "{sc0 sc1 stores and loads both sides}" is a valid hand-off form there); an agent-scope
// fence per block costs 1.7-3.5 us here (16 -> 57 us for one of these kernels).  The
// assumption is tied to the target: any other architecture must not compile these sources unchanged.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmodest_hip is written for gfx950 only: its last-block reductions rely on gfx950's write-through sc1 stores"
#endif

// ... and the publishing side must DRAIN its write-through stores before the workgroup barrier that precedes
// the ticket: __syncthreads() does not wait for global stores (hipcc emits `s_waitcnt lgkmcnt(0); s_barrier`
// there), so a ticket taken by wavefront 0 could overtake the partial results of wavefronts 1..3 -- found in
// round 3 as 11 differing results in 14 400 repeated scans under an 8-process load (tools/soak_mask.py).  Inline
// asm because the compiler drops a redundant-looking s_waitcnt (MI355X_MICROARCH.md, compiler hazard).
__device__ __forceinline__ void modest_drain_stores() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

constexpr int MODEST_STAGE_SLOTS = 8;

struct modest_ctx {
    int device;
    char *scratch;          // grow-only device arena
    size_t scratch_bytes;
    char *pinned;           // grow-only pinned host staging
    size_t pinned_bytes;
    int num_cus;
    // optional per-launch timing of the dominant (history stream) kernel
    int profiling;
    int prof_count;
    hipEvent_t *prof_ev;   // 2 * prof_cap events
    int prof_cap;
    int pp_attr_done;      // dynamic-LDS limits of the PP kernels raised on this device
    int ppb_attr_done;     // ... of the batched PP kernels
    // zeroed state words of the order-preserving compaction kernels (compact.h); the kernels leave
    // them zeroed, so no memset per launch
    unsigned long long *cstate;
    size_t cstate_blocks;
    // per-cell counters of the fused mask + cluster call (cluster.hip): zero between calls (the cell
    // scan clears them behind itself); `zwords_dirty` is set while a call is in flight on the host side
    // and makes the next call clear them if an error return left them undefined
    unsigned *zwords;
    size_t zwords_count;
    int zwords_dirty;
    int zwords_live;   // the cell counters hold a scan's counts (between the two phases of the fused mask call): never clear
    // buffers of the per-scan driver (scan_driver.hip) that live across its sub-calls, which carve their
    // own temporaries from `scratch` / `pinned` at offset 0: grow-only, device and pinned host
    char *hold;
    size_t hold_bytes;
    char *hold_pinned;
    size_t hold_pinned_bytes;
    // ring of pinned staging slots for small per-call tables that are copied to the device
    // asynchronously (frame descriptors of modest_pp_score_frames): a slot is reused only after the
    // event recorded behind its copy has completed
    char *stage[MODEST_STAGE_SLOTS];
    size_t stage_bytes[MODEST_STAGE_SLOTS];
    hipEvent_t stage_ev[MODEST_STAGE_SLOTS];
    int stage_used[MODEST_STAGE_SLOTS];
    int stage_next;
    // device table of a chain of scans (mask_chain.h): grow-only
    char *chain_tab;
    size_t chain_tab_bytes;
};

// next staging slot with at least `bytes` (waits for the slot's previous copy); commit records the
// event behind the copy that was just enqueued from it
int modest_ctx_stage_slot(modest_ctx *ctx, size_t bytes, void **out);
int modest_ctx_stage_commit(modest_ctx *ctx, hipStream_t stream);
int modest_ctx_chain_tab(modest_ctx *ctx, size_t bytes, char **out);

// persistent compaction state for `nblocks` blocks (allocated and zeroed on first use / growth)
int modest_ctx_compact_state(modest_ctx *ctx, size_t nblocks, hipStream_t stream, unsigned long long **out);

// persistent zeroed counter words (see modest_ctx::zwords); cleared here on first use or when the
// dirty mark is still set.  Layout: [0, MODEST_ZW_CELLS) cell counters of the fused mask call,
// [MODEST_ZW_CELLS, +MODEST_ZW_TICKETS) "last block done" tickets of kernels that finish their own
// reduction (every such kernel leaves its tickets at zero).
constexpr size_t MODEST_ZW_CELLS = 16384;
constexpr size_t MODEST_ZW_TICKETS = 4096;
int modest_ctx_zero_words(modest_ctx *ctx, hipStream_t stream, unsigned **out);
static inline unsigned *modest_tickets(unsigned *zwords) { return zwords + MODEST_ZW_CELLS; }

// Record an event pair around a kernel when profiling is on (no-ops otherwise).
void modest_prof_mark(modest_ctx *ctx, hipStream_t stream, int end);

void modest_set_error(const char *fmt, ...);

#define MODEST_HIP_CHECK(expr)                                                  \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            modest_set_error("%s failed: %s (%s:%d)", #expr,                    \
                             hipGetErrorString(_e), __FILE__, __LINE__);        \
            return MODEST_ERR_HIP;                                              \
        }                                                                       \
    } while (0)

#define MODEST_REQUIRE(cond, msg)                                               \
    do {                                                                        \
        if (!(cond)) {                                                          \
            modest_set_error("%s: requirement failed: %s (%s)", __func__, #cond, msg); \
            return MODEST_ERR_ARG;                                              \
        }                                                                       \
    } while (0)

// Ensure the context arena holds at least `bytes`; returns 0 or error code.
int modest_ctx_reserve(modest_ctx *ctx, size_t bytes);
int modest_ctx_reserve_pinned(modest_ctx *ctx, size_t bytes);
int modest_ctx_reserve_hold(modest_ctx *ctx, size_t dev_bytes, size_t pinned_bytes);

// Two-phase forms of three blocking entry points (phase & 1: enqueue, no synchronise; phase & 2: read the
// results after a synchronise -- the caller's, or the function's own when both bits are set).  The per-scan
// driver keeps a RANSAC refit in flight together with the next trial batch or the mask kernel; their result
// areas in the context's pinned block are disjoint (refit [0,128), kept count [192,196), batch [576,...)).
int modest_ransac_trials_phase(modest_ctx *ctx, const float *cand, int n_cand, const int32_t *trip_host, int K,
                               float *thr_inout, float *models_out, int32_t *n_inliers, double *sse, double *sy,
                               double *syy, void *stream, int phase);
int modest_ransac_refit_phase(modest_ctx *ctx, const float *cand, int n_cand, const float *model_host, float thr,
                              double *out_model, int32_t *n_inliers, void *stream, int phase);
size_t modest_ransac_scratch_bound(int n_cand, int K);   // plane.hip: arena bytes of a trial batch / refit, see there
int modest_mask_cluster_phase(modest_ctx *ctx, const float *pts, int n, int stride, const float *pp,
                              const double *plane4, double offset, const double *only_range4,
                              const double *limit_range4, int neighbor_type, int affinity_type, int k_neighbors,
                              double radius, double eps, int min_samples, int32_t *labels, int32_t *n_kept,
                              int32_t *n_clusters, void *stream, int phase);

// Bump allocator over the arena (256-byte aligned carves).
struct Arena {
    char *base;
    size_t off;
    explicit Arena(char *b) : base(b), off(0) {}
    template <typename T> T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};
static inline size_t arena_sz(size_t bytes) { return (bytes + 255) & ~size_t(255); }

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
