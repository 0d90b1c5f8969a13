// context.hip -- error plumbing and the device/stream context of librlarm_hip.so.
#include "internal.h"

static thread_local char g_err[512] = "";

void hp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int hp_abi_version(void) { return HP_ABI_VERSION; }
const char *hp_last_error(void) { return g_err; }

int hp_ctx_create(int device_id, hp_ctx **out) {
    HP_REQUIRE(out, HP_ERR_INVALID, "hp_ctx_create: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        hp_set_error("hp_ctx_create: no HIP device visible (%s); this library has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return HP_ERR_NODEVICE;
    }
    HP_REQUIRE(device_id >= 0 && device_id < n, HP_ERR_INVALID, "hp_ctx_create: device %d not in [0,%d)", device_id, n);
    HP_CHECK_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HP_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        hp_set_error("hp_ctx_create: device %d is %s; the kernels in this library are built for gfx950 only",
                     device_id, prop.gcnArchName);
        return HP_ERR_NODEVICE;
    }
    hp_ctx *c = new hp_ctx();
    c->device = device_id;
    c->cu_count = prop.multiProcessorCount;
    snprintf(c->name, sizeof(c->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        hp_set_error("hp_ctx_create: hipStreamCreate failed: %s", hipGetErrorString(e));
        return HP_ERR_HIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return HP_OK;
}

int hp_ctx_set_stream(hp_ctx *ctx, void *hip_stream) {
    HP_REQUIRE(ctx, HP_ERR_INVALID, "hp_ctx_set_stream: null ctx");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return HP_OK;
}

int hp_ctx_synchronize(hp_ctx *ctx) {
    HP_REQUIRE(ctx, HP_ERR_INVALID, "hp_ctx_synchronize: null ctx");
    HP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return HP_OK;
}

int hp_ctx_device_name(hp_ctx *ctx, char *buf, size_t len) {
    HP_REQUIRE(ctx && buf && len > 0, HP_ERR_INVALID, "hp_ctx_device_name: bad argument");
    snprintf(buf, len, "%s", ctx->name);
    return HP_OK;
}

void hp_ctx_destroy(hp_ctx *ctx) {
    if (!ctx) return;
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

}  // extern "C"
