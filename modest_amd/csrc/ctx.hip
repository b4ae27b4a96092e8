// Context, error string and arena management for libmodest_hip.so.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <new>

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
    *out = c;
    return MODEST_OK;
}

extern "C" int modest_ctx_destroy(modest_ctx *ctx) {
    if (!ctx) return MODEST_OK;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
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
    MODEST_HIP_CHECK(hipMalloc(&p, want));
    ctx->scratch = static_cast<char *>(p);
    ctx->scratch_bytes = want;
    return MODEST_OK;
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
    ctx->pinned = static_cast<char *>(p);
    ctx->pinned_bytes = want;
    return MODEST_OK;
}
