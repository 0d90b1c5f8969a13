// context.hip -- error plumbing and the device/stream context of librlarm_hip.so.
#include "internal.h"
#include <chrono>

static thread_local char g_err[512] = "";

// shader-clock probe: s_memtime ticks (shader cycles) per wall_clock64 tick (100 MHz constant clock)
__global__ void k_clock_probe(unsigned long long *out, int spin) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    unsigned long long c1 = c0;
    while ((long long)(c1 - c0) < spin) c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

__global__ void k_floor_probe(int *p) {
    if (threadIdx.x == 0) p[0] += 1;
}

// calibration probes (hp_ctx_calibrate): the two per-CU rates the update's chain kernel lives on
// (a) one workgroup streams a 256 KiB block (one 256 x 256 layer's weights) through LDS-DMA, `passes` times: 8 waves x 32 blocks
//     of 1 KiB, eight in flight per wave.  out[0] = wall-clock ticks (100 MHz)
__global__ __launch_bounds__(512) void k_cal_stream(const float *src, int passes, unsigned long long *out, float *sink) {
    __shared__ __attribute__((aligned(16))) float ring[8][8][256];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const float4 *base = reinterpret_cast<const float4 *>(src) + (size_t)wave * 32 * 64 + lane;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int b = 0; b < 8; ++b) __builtin_amdgcn_global_load_lds(base + (8 * g + b) * 64, &ring[wave][b][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = wall_clock64() - t0;
    if (ring[wave][lane & 7][lane] == 1.2345e-30f) sink[0] = 1.f;
}
// (b) one wave, a dependent chain of v_mfma_f32_4x4x1_16b_f32: out[1] = shader cycles for n of them
typedef float cal_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_cal_mfma(int n8, unsigned long long *out, float *sink) {
    cal_f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = (float)(threadIdx.x & 3), b = 1.0f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    }
    if (c[0] == 1.2345e-30f) sink[0] = c[1];
    if (threadIdx.x == 0) out[1] = __builtin_readcyclecounter() - c0;
}

thread_local std::vector<std::string> *hp_klog = nullptr;

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
    CtxGuard guard(ctx);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {
        // what the context enqueued so far stays ordered in front of what it enqueues next (a store on the old stream, a
        // sample on the new one): the new stream waits for an event behind the old stream's work, no host wait
        // (one event per context, re-recorded at every switch: an event destroyed with a wait still pending is not safe).
        // The special handles (hipStreamLegacy = 1, hipStreamPerThread = 2) take no event records on this runtime: a host wait
        // for the old stream instead -- a switch is a rare, set-up time call.
        const bool special = (uintptr_t)ctx->stream <= 2 || (uintptr_t)next <= 2;
        if (special) {
            HP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            if (!ctx->order_ev) HP_CHECK_HIP(hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
            HP_CHECK_HIP(hipEventRecord(ctx->order_ev, ctx->stream));
            HP_CHECK_HIP(hipStreamWaitEvent(next, ctx->order_ev, 0));
        }
    }
    ctx->stream = next;
    return HP_OK;
}

int hp_ctx_get_stream(hp_ctx *ctx, void **hip_stream) {
    HP_REQUIRE(ctx && hip_stream, HP_ERR_INVALID, "hp_ctx_get_stream: null argument");
    CtxGuard guard(ctx);
    *hip_stream = (void *)ctx->stream;
    return HP_OK;
}

// ---- a caller's stream for the duration of ONE call (hp_ctx_borrow_stream / hp_ctx_return_stream) ---------------------------------
// A host that hands device outputs to a framework wants them written on the framework's stream, in its order -- without rebinding
// the context (hp_ctx_set_stream), whose own stream carries the fused learner's cached graphs.  Between borrow and return the
// calling thread holds the context's lock and every launch of the library goes to the borrowed stream, ordered behind what the
// context's own stream held (a device-side wait, skipped when the own stream has been idle since the last borrow).  The way back is
// lazy: the first entry point that uses the own stream again orders it behind the borrowed stream's work (ctx_join_foreign, from
// CtxGuard).  Measured on a torch learner (tools/ubench/level1_gpu.py, us per update): sampler on torch's stream 2083; sampler on
// the own stream with an event fence each way 2206-2270 -- two active queues cost that loop ~190 us whatever orders them.
}  // extern "C" (re-opened below)
void ctx_join_foreign(hp_ctx *c) {
    hipStream_t f = c->foreign;
    c->foreign = nullptr;
    if (!f || f == c->stream) return;
    hipEvent_t &ev = c->fence_ev[1];
    // (the legacy default stream takes event records under its null-stream name only)
    if ((ev || hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) &&
        hipEventRecord(ev, f == hipStreamLegacy ? nullptr : f) == hipSuccess && hipStreamWaitEvent(c->stream, ev, 0) == hipSuccess)
        return;
    (void)hipGetLastError();
    (void)hipStreamSynchronize(f);      // no event on that stream: a host wait instead (rare: once per switch back)
}
extern "C" {

int hp_ctx_borrow_stream(hp_ctx *ctx, void *hip_stream) {
    HP_REQUIRE(ctx, HP_ERR_INVALID, "hp_ctx_borrow_stream: null ctx");
    ctx->mu.lock();                      // held until hp_ctx_return_stream (recursive: the calls in between lock it again)
    (void)hipSetDevice(ctx->device);
    auto fail = [&](const char *what, hipError_t e) {
        hp_set_error("hp_ctx_borrow_stream: %s: %s", what, hipGetErrorString(e));
        ctx->mu.unlock();
        return HP_ERR_HIP;
    };
    if (ctx->borrowed_from) {
        hp_set_error("hp_ctx_borrow_stream: a stream is borrowed already (hp_ctx_return_stream first)");
        ctx->mu.unlock();
        return HP_ERR_STATE;
    }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : hipStreamLegacy;   // NULL: the null stream, a framework's default
    if (s != ctx->stream) {
        if (ctx->foreign && ctx->foreign != s) ctx_join_foreign(ctx);
        if (ctx->own_dirty) {            // the borrowed stream's work waits for what the own stream holds
            hipEvent_t &ev = ctx->fence_ev[0];
            hipError_t e = ev ? hipSuccess : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess && (uintptr_t)ctx->stream > 2) e = hipEventRecord(ev, ctx->stream);
            else if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess && (uintptr_t)ctx->stream > 2) e = hipStreamWaitEvent(s == hipStreamLegacy ? nullptr : s, ev, 0);
            if (e != hipSuccess) return fail("ordering the borrowed stream behind the context's", e);
            ctx->own_dirty = false;
        }
    }
    ctx->borrowed_from = ctx->stream;
    ctx->stream = s;
    return HP_OK;
}

int hp_ctx_return_stream(hp_ctx *ctx) {
    HP_REQUIRE(ctx, HP_ERR_INVALID, "hp_ctx_return_stream: null ctx");
    HP_REQUIRE(ctx->borrowed_from, HP_ERR_STATE, "hp_ctx_return_stream: no stream is borrowed");   // (only the borrowing thread can get here with it set)
    if (ctx->stream != ctx->borrowed_from) ctx->foreign = ctx->stream;
    ctx->stream = ctx->borrowed_from;
    ctx->borrowed_from = nullptr;
    ctx->mu.unlock();
    return HP_OK;
}

int hp_ctx_synchronize(hp_ctx *ctx) {
    HP_REQUIRE(ctx, HP_ERR_INVALID, "hp_ctx_synchronize: null ctx");
    hipStream_t s;
    {   // wait outside the lock: a feeder thread may keep storing while this thread waits for a cycle
        CtxGuard guard(ctx);             // (also orders the own stream behind a stream borrowed earlier)
        s = ctx->stream;
    }
    // Short waits spin on the stream's status: a blocking hipStreamSynchronize parks the thread and its wake-up costs
    // 20-50 us -- 2-3 us per step of a 20-update call (the driver's bench invocation: 44.3 vs 41.2 us/step).  After 2 ms of
    // spinning the wait is a long one and the thread blocks as before.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return HP_OK;
        if (e != hipErrorNotReady) HP_CHECK_HIP(e);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

int hp_ctx_device_name(hp_ctx *ctx, char *buf, size_t len) {
    HP_REQUIRE(ctx && buf && len > 0, HP_ERR_INVALID, "hp_ctx_device_name: bad argument");
    snprintf(buf, len, "%s", ctx->name);
    return HP_OK;
}

// PCI bus id of the context's device ("0000:05:00.0"): what tells two ranks on ONE physical device apart from two GPUs
int hp_ctx_pci_bus_id(hp_ctx *ctx, char *buf, size_t len) {
    HP_REQUIRE(ctx && buf && len >= 16, HP_ERR_INVALID, "hp_ctx_pci_bus_id: bad argument");
    HP_CHECK_HIP(hipDeviceGetPCIBusId(buf, (int)len, ctx->device));
    return HP_OK;
}

// diagnostic: average cost of one dependent trivial kernel on the context's stream, as a captured
// hipGraph of n nodes (graph != 0) or n eager launches.  Used by DESIGN.md's launch-floor numbers.
int hp_ctx_launch_floor(hp_ctx *ctx, int n, int graph, double *us_per_kernel) {
    HP_REQUIRE(ctx && us_per_kernel && n > 0, HP_ERR_INVALID, "hp_ctx_launch_floor: bad argument");
    hipStream_t s = ctx->stream;
    int *d = nullptr;
    HP_CHECK_HIP(hipMalloc((void **)&d, 4));
    HP_CHECK_HIP(hipMemsetAsync(d, 0, 4, s));
    hipEvent_t e0, e1;
    HP_CHECK_HIP(hipEventCreate(&e0));
    HP_CHECK_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    if (graph) {
        hipGraph_t g;
        hipGraphExec_t ge;
        HP_CHECK_HIP(hipStreamSynchronize(s));
        HP_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_floor_probe, dim3(1), dim3(64), 0, s, d);
        HP_CHECK_HIP(hipStreamEndCapture(s, &g));
        HP_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HP_CHECK_HIP(hipGraphLaunch(ge, s));
        HP_CHECK_HIP(hipEventRecord(e0, s));
        HP_CHECK_HIP(hipGraphLaunch(ge, s));
        HP_CHECK_HIP(hipEventRecord(e1, s));
        HP_CHECK_HIP(hipEventSynchronize(e1));
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    } else {
        for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k_floor_probe, dim3(1), dim3(64), 0, s, d);
        HP_CHECK_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_floor_probe, dim3(1), dim3(64), 0, s, d);
        HP_CHECK_HIP(hipEventRecord(e1, s));
        HP_CHECK_HIP(hipEventSynchronize(e1));
    }
    HP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *us_per_kernel = 1e3 * ms / n;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return HP_OK;
}

// diagnostic: what a hipEvent pair with NOTHING between the two records reads on the context's stream (average of
// `reps` pairs, each behind a trivial kernel so that the stream is busy like in the profiling pass).  This is the
// bracketing overhead contained in every per-launch event measurement; rocprofv3 kernel durations do not have it.
int hp_ctx_event_pair_us(hp_ctx *ctx, int reps, double *us) {
    HP_REQUIRE(ctx && us && reps > 0, HP_ERR_INVALID, "hp_ctx_event_pair_us: bad argument");
    hipStream_t s = ctx->stream;
    int *d = nullptr;
    HP_CHECK_HIP(hipMalloc((void **)&d, 4));
    HP_CHECK_HIP(hipMemsetAsync(d, 0, 4, s));
    hipEvent_t e0, e1;
    HP_CHECK_HIP(hipEventCreate(&e0));
    HP_CHECK_HIP(hipEventCreate(&e1));
    double tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_floor_probe, dim3(1), dim3(64), 0, s, d);
        HP_CHECK_HIP(hipEventRecord(e0, s));
        HP_CHECK_HIP(hipEventRecord(e1, s));
        HP_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        HP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
    }
    *us = 1e3 * tot / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return HP_OK;
}

// diagnostic: shader clock (MHz) seen by a kernel enqueued right now on the context's stream
int hp_ctx_clock_mhz(hp_ctx *ctx, double *mhz) {
    HP_REQUIRE(ctx && mhz, HP_ERR_INVALID, "hp_ctx_clock_mhz: bad argument");
    unsigned long long *d = nullptr, h[2] = {0, 0};
    HP_CHECK_HIP(hipMalloc((void **)&d, 16));
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, ctx->stream, d, 40000);
    HP_CHECK_HIP(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    *mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return HP_OK;
}

// diagnostic: ~200 us of probes that characterise the box (rlarm_hip_debug.h)
int hp_ctx_calibrate(hp_ctx *ctx, double *out4) {
    HP_REQUIRE(ctx && out4, HP_ERR_INVALID, "hp_ctx_calibrate: bad argument");
    CtxGuard guard(ctx);
    // on a stream of its own: the context's may be a framework's (legacy streams cannot be captured) or busy
    hipStream_t s = nullptr, keep = ctx->stream;
    HP_CHECK_HIP(hipStreamSynchronize(keep));
    HP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ctx->stream = s;
    int st = hp_ctx_launch_floor(ctx, 200, 1, &out4[0]);
    float *src = nullptr, *sink = nullptr;
    unsigned long long *d = nullptr, h[2] = {0, 0};
    const int passes = 16, n8 = 512;
    hipError_t e = hipSuccess;
    if (st == HP_OK) {
        e = hipMalloc((void **)&src, 256 * 1024);
        if (e == hipSuccess) e = hipMalloc((void **)&sink, 16);
        if (e == hipSuccess) e = hipMalloc((void **)&d, 16);
        if (e == hipSuccess) e = hipMemsetAsync(src, 0, 256 * 1024, s);
        for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {   // the second run finds the block in L2 and the code in the instruction cache
            hipLaunchKernelGGL(k_cal_stream, dim3(1), dim3(512), 0, s, src, passes, d, sink);
            hipLaunchKernelGGL(k_cal_mfma, dim3(1), dim3(64), 0, s, n8, d, sink);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (st == HP_OK && e == hipSuccess) st = hp_ctx_clock_mhz(ctx, &out4[3]);
    }
    ctx->stream = keep;
    if (src) (void)hipFree(src);
    if (sink) (void)hipFree(sink);
    if (d) (void)hipFree(d);
    (void)hipStreamDestroy(s);
    if (st != HP_OK) return st;
    HP_CHECK_HIP(e);
    out4[1] = h[0] ? (double)passes * 256.0 * 1024.0 / ((double)h[0] * 10.0) : 0.0;   // bytes per ns = GB/s (a tick is 10 ns)
    out4[2] = (double)h[1] / (8.0 * n8);
    return HP_OK;
}

void hp_ctx_destroy(hp_ctx *ctx) {
    if (!ctx) return;
    ctx->reward_ws.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    for (hipEvent_t ev : ctx->fence_ev)
        if (ev) (void)hipEventDestroy(ev);
    delete ctx;
}

}  // extern "C"
