// Context, error string and arena management for libmodest_hip.so.
#include "common.h"
#include <cstdlib>
#include <ctime>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <new>

// MODEST_ALLOC_TRACE=1: every (re)allocation of a context prints a line -- a grow inside a timed region is a device
// synchronise + free + allocate, tens of milliseconds when several processes share the GPU
static void modest_alloc_note(const char *what, size_t bytes) {
    static const int on = getenv("MODEST_ALLOC_TRACE") != nullptr;
    if (on) {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        fprintf(stderr, "[modest alloc] t=%.3f s pid %d: %s %zu bytes\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, (int)getpid(), what, bytes);
    }
}

static thread_local char g_err[512] = "";

void modest_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int modest_version(void) { return 100; }

extern "C" const char *modest_last_error(void) { return g_err; }

extern "C" int modest_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int modest_ctx_create(int device, modest_ctx **out) {
    if (!out) {
        modest_set_error("modest_ctx_create: out is NULL");
        return MODEST_ERR_ARG;
    }
    *out = nullptr;
    int n = modest_device_count();
    if (n <= 0) {
        modest_set_error("modest_ctx_create: no HIP device visible (this library has no CPU path)");
        return MODEST_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) {
        modest_set_error("modest_ctx_create: device %d out of range [0,%d)", device, n);
        return MODEST_ERR_ARG;
    }
    MODEST_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MODEST_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    modest_ctx *c = new (std::nothrow) modest_ctx();
    if (!c) {
        modest_set_error("modest_ctx_create: out of host memory");
        return MODEST_ERR_CAPACITY;
    }
    c->device = device;
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // tuning knob: size the persistent PP grids for fewer CUs (leaves room for the small kernels of other
    // host processes that share the GPU)
    if (const char *e = getenv("MODEST_NUM_CUS")) {
        const int v = atoi(e);
        if (v >= 8 && v <= c->num_cus) c->num_cus = v;
    }
    c->profiling = 0;
    c->prof_count = 0;
    c->prof_ev = nullptr;
    c->prof_cap = 0;
    c->pp_attr_done = 0;
    c->ppb_attr_done = 0;
    c->cstate = nullptr;
    c->cstate_blocks = 0;
    c->zwords = nullptr;
    c->zwords_count = 0;
    c->zwords_dirty = 0;
    c->zwords_live = 0;
    c->hold = nullptr;
    c->hold_bytes = 0;
    c->hold_pinned = nullptr;
    c->hold_pinned_bytes = 0;
    c->chain_tab = nullptr;
    c->chain_tab_bytes = 0;
    for (int i = 0; i < MODEST_STAGE_SLOTS; ++i) {
        c->stage[i] = nullptr;
        c->stage_bytes[i] = 0;
        c->stage_ev[i] = nullptr;
        c->stage_used[i] = 0;
    }
    c->stage_next = 0;
    *out = c;
    return MODEST_OK;
}

extern "C" int modest_ctx_destroy(modest_ctx *ctx) {
    if (!ctx) return MODEST_OK;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->cstate) (void)hipFree(ctx->cstate);
    if (ctx->zwords) (void)hipFree(ctx->zwords);
    if (ctx->hold) (void)hipFree(ctx->hold);
    if (ctx->hold_pinned) (void)hipHostFree(ctx->hold_pinned);
    if (ctx->chain_tab) (void)hipFree(ctx->chain_tab);
    for (int i = 0; i < MODEST_STAGE_SLOTS; ++i) {
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
        if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
    }
    for (int i = 0; i < 2 * ctx->prof_cap; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
    delete ctx;
    return MODEST_OK;
}

int modest_ctx_compact_state(modest_ctx *ctx, size_t nblocks, hipStream_t stream, unsigned long long **out) {
    if (nblocks > ctx->cstate_blocks) {
        MODEST_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->cstate) MODEST_HIP_CHECK(hipFree(ctx->cstate));
        ctx->cstate = nullptr;
        ctx->cstate_blocks = 0;
        const size_t want = nblocks < 4096 ? 4096 : nblocks + nblocks / 4;
        void *p = nullptr;
        MODEST_HIP_CHECK(hipMalloc(&p, 16 + 8 * want));
        modest_alloc_note("compaction state", (size_t)(16 + 8 * want));
        MODEST_HIP_CHECK(hipMemsetAsync(p, 0, 16 + 8 * want, stream));
        ctx->cstate = static_cast<unsigned long long *>(p);
        ctx->cstate_blocks = want;
    }
    *out = ctx->cstate;
    return MODEST_OK;
}

int modest_ctx_zero_words(modest_ctx *ctx, hipStream_t stream, unsigned **out) {
    const size_t words = MODEST_ZW_CELLS + MODEST_ZW_TICKETS;
    if (words > ctx->zwords_count) {
        MODEST_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->zwords) MODEST_HIP_CHECK(hipFree(ctx->zwords));
        ctx->zwords = nullptr;
        ctx->zwords_count = 0;
        void *p = nullptr;
        MODEST_HIP_CHECK(hipMalloc(&p, words * 4));
        modest_alloc_note("counter words", (size_t)(words * 4));
        ctx->zwords = static_cast<unsigned *>(p);
        ctx->zwords_count = words;
        ctx->zwords_dirty = 1;
    }
    if (ctx->zwords_dirty && !ctx->zwords_live) {
        MODEST_HIP_CHECK(hipMemsetAsync(ctx->zwords, 0, ctx->zwords_count * 4, stream));
        ctx->zwords_dirty = 0;
    }
    *out = ctx->zwords;
    return MODEST_OK;
}

int modest_ctx_reserve(modest_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return MODEST_OK;
    // Growing frees the old arena: wait for work that may still use it.
    MODEST_HIP_CHECK(hipDeviceSynchronize());
    if (ctx->scratch) MODEST_HIP_CHECK(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + (1u << 20);
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {   // the 25 % head room is a convenience: without it the exact size still serves the call
        (void)hipGetLastError();
        want = bytes;
        p = nullptr;
        MODEST_HIP_CHECK(hipMalloc(&p, want));
    }
    modest_alloc_note("scratch arena", (size_t)(want));
    ctx->scratch = static_cast<char *>(p);
    ctx->scratch_bytes = want;
    return MODEST_OK;
}

extern "C" void modest_warm_boxfilter(void);
extern "C" void modest_warm_boxfit(void);
extern "C" void modest_warm_cluster(void);
extern "C" void modest_warm_cluster_stats(void);
extern "C" void modest_warm_iou3d(void);
extern "C" void modest_warm_plane(void);
extern "C" void modest_warm_pp_count(void);
extern "C" void modest_warm_pp_frames(void);
extern "C" void modest_warm_pp_v4(void);
extern "C" void modest_warm_transform(void);
extern "C" int modest_warmup(modest_ctx *ctx) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    modest_warm_boxfilter();
    modest_warm_boxfit();
    modest_warm_cluster();
    modest_warm_cluster_stats();
    modest_warm_iou3d();
    modest_warm_plane();
    modest_warm_pp_count();
    modest_warm_pp_frames();
    modest_warm_pp_v4();
    modest_warm_transform();
    (void)hipGetLastError();
    return MODEST_OK;
}

extern "C" int modest_ctx_reserve_arena(modest_ctx *ctx, uint64_t bytes) {
    MODEST_REQUIRE(ctx != nullptr, "ctx is NULL");
    MODEST_REQUIRE(bytes < (1ULL << 40), "absurd arena size");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    return modest_ctx_reserve(ctx, (size_t)bytes);
}

int modest_ctx_reserve_pinned(modest_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return MODEST_OK;
    MODEST_HIP_CHECK(hipDeviceSynchronize());
    if (ctx->pinned) MODEST_HIP_CHECK(hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    size_t want = bytes + bytes / 4 + (64u << 10);
    void *p = nullptr;
    MODEST_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    modest_alloc_note("pinned block", (size_t)(want));
    ctx->pinned = static_cast<char *>(p);
    ctx->pinned_bytes = want;
    return MODEST_OK;
}

int modest_ctx_reserve_hold(modest_ctx *ctx, size_t dev_bytes, size_t pinned_bytes) {
    if (dev_bytes > ctx->hold_bytes) {
        MODEST_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->hold) MODEST_HIP_CHECK(hipFree(ctx->hold));
        ctx->hold = nullptr;
        ctx->hold_bytes = 0;
        const size_t want = dev_bytes + dev_bytes / 4 + (1u << 20);
        void *p = nullptr;
        MODEST_HIP_CHECK(hipMalloc(&p, want));
        modest_alloc_note("hold (device)", (size_t)(want));
        ctx->hold = static_cast<char *>(p);
        ctx->hold_bytes = want;
    }
    if (pinned_bytes > ctx->hold_pinned_bytes) {
        MODEST_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->hold_pinned) MODEST_HIP_CHECK(hipHostFree(ctx->hold_pinned));
        ctx->hold_pinned = nullptr;
        ctx->hold_pinned_bytes = 0;
        const size_t want = pinned_bytes + pinned_bytes / 4 + (64u << 10);
        void *p = nullptr;
        MODEST_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        modest_alloc_note("hold (pinned)", (size_t)(want));
        ctx->hold_pinned = static_cast<char *>(p);
        ctx->hold_pinned_bytes = want;
    }
    return MODEST_OK;
}

int modest_ctx_stage_slot(modest_ctx *ctx, size_t bytes, void **out) {
    const int k = ctx->stage_next;
    if (ctx->stage_used[k]) {   // the copy that last read this slot must be over
        MODEST_HIP_CHECK(hipEventSynchronize(ctx->stage_ev[k]));
        ctx->stage_used[k] = 0;
    }
    if (bytes > ctx->stage_bytes[k]) {
        // ALL slots of the ring grow now: a caller that starts to send larger tables (a longer chain, the block path's
        // frame tables) pays the pinned allocations in one call -- its first, normally a warm-up -- and not one in each
        // of its next MODEST_STAGE_SLOTS calls (measured: a 10 ms timed window spent 7 ms in them under eight processes)
        const size_t want = bytes + bytes / 2 + 4096;
        for (int j = 0; j < MODEST_STAGE_SLOTS; ++j) {
            if (ctx->stage_bytes[j] >= want) continue;
            if (ctx->stage_used[j]) {
                MODEST_HIP_CHECK(hipEventSynchronize(ctx->stage_ev[j]));
                ctx->stage_used[j] = 0;
            }
            if (ctx->stage[j]) MODEST_HIP_CHECK(hipHostFree(ctx->stage[j]));
            ctx->stage[j] = nullptr;
            ctx->stage_bytes[j] = 0;
            void *p = nullptr;
            MODEST_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
            modest_alloc_note("staging ring slot", (size_t)(want));
            ctx->stage[j] = static_cast<char *>(p);
            ctx->stage_bytes[j] = want;
        }
    }
    if (!ctx->stage_ev[k]) MODEST_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    *out = ctx->stage[k];
    return MODEST_OK;
}

int modest_ctx_stage_commit(modest_ctx *ctx, hipStream_t stream) {
    const int k = ctx->stage_next;
    MODEST_HIP_CHECK(hipEventRecord(ctx->stage_ev[k], stream));
    ctx->stage_used[k] = 1;
    ctx->stage_next = (k + 1) % MODEST_STAGE_SLOTS;
    return MODEST_OK;
}

// ---- per-launch timing of the dominant kernel (bench.py's roofline leg) ------
void modest_prof_mark(modest_ctx *ctx, hipStream_t stream, int end) {
    if (!ctx->profiling || ctx->prof_count >= ctx->prof_cap) return;
    (void)hipEventRecord(ctx->prof_ev[2 * ctx->prof_count + (end ? 1 : 0)], stream);
    if (end) ++ctx->prof_count;
}

extern "C" int modest_ctx_profile_begin(modest_ctx *ctx, int capacity) {
    MODEST_REQUIRE(ctx != nullptr && capacity >= 0 && capacity <= (1 << 20), "bad arguments");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    if (capacity > ctx->prof_cap) {
        hipEvent_t *ev = new (std::nothrow) hipEvent_t[2 * (size_t)capacity];
        if (!ev) {
            modest_set_error("modest_ctx_profile_begin: out of host memory");
            return MODEST_ERR_CAPACITY;
        }
        for (int i = 0; i < 2 * ctx->prof_cap; ++i) ev[i] = ctx->prof_ev[i];
        for (int i = 2 * ctx->prof_cap; i < 2 * capacity; ++i) MODEST_HIP_CHECK(hipEventCreate(&ev[i]));
        delete[] ctx->prof_ev;
        ctx->prof_ev = ev;
        ctx->prof_cap = capacity;
    }
    ctx->prof_count = 0;
    ctx->profiling = capacity > 0;
    return MODEST_OK;
}

extern "C" int modest_ctx_profile_collect(modest_ctx *ctx, float *ms_out, int cap, int *n_out) {
    MODEST_REQUIRE(ctx != nullptr && n_out != nullptr, "bad arguments");
    MODEST_HIP_CHECK(hipSetDevice(ctx->device));
    const int n = ctx->prof_count < cap ? ctx->prof_count : cap;
    for (int i = 0; i < n; ++i) {
        MODEST_HIP_CHECK(hipEventSynchronize(ctx->prof_ev[2 * i + 1]));
        MODEST_HIP_CHECK(hipEventElapsedTime(&ms_out[i], ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    }
    *n_out = n;
    ctx->profiling = 0;
    return MODEST_OK;
}

// device table of a chain of scans: grow-only, never moved while a launch that reads it may be in flight
// (a grow synchronises the device first)
int modest_ctx_chain_tab(modest_ctx *ctx, size_t bytes, char **out) {
    MODEST_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    if (bytes > ctx->chain_tab_bytes) {
        MODEST_HIP_CHECK(hipSetDevice(ctx->device));
        MODEST_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->chain_tab) MODEST_HIP_CHECK(hipFree(ctx->chain_tab));
        ctx->chain_tab = nullptr;
        ctx->chain_tab_bytes = 0;
        const size_t want = bytes < (64u << 10) ? (64u << 10) : bytes;
        void *p = nullptr;
        MODEST_HIP_CHECK(hipMalloc(&p, want));
        modest_alloc_note("chain table", (size_t)(want));
        ctx->chain_tab = static_cast<char *>(p);
        ctx->chain_tab_bytes = want;
    }
    *out = ctx->chain_tab;
    return MODEST_OK;
}
