// internal.h -- shared declarations of librlarm_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rlarm_hip.h"
#include "rlarm_hip_debug.h"

// ------------------------------------------------------------------ error plumbing
void hp_set_error(const char *fmt, ...);

// launch log (hp_agent_update_kernels, rlarm_hip_debug.h): while non-null on the calling thread, every launch site on the path of
// an update sequence records the kernel it enqueues -- the names bench.py reports are read off the launch logic itself
extern thread_local std::vector<std::string> *hp_klog;
#define HP_KLOG(name) do { if (hp_klog) hp_klog->push_back(name); } while (0)

#define HP_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            hp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return HP_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define HP_REQUIRE(cond, code, ...)   \
    do {                              \
        if (!(cond)) {                \
            hp_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define HP_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != HP_OK) return _s; \
    } while (0)

// small RAII-less device buffer helper (grow-only)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return HP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        HP_CHECK_HIP(hipMalloc(&p, need));
        bytes = need;
        return HP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ------------------------------------------------------------------ context
struct hp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // stream kernels are enqueued on
    hipStream_t own_stream = nullptr;  // created by the context
    hipEvent_t order_ev = nullptr;     // hp_ctx_set_stream: orders the new stream behind the old one's work
    // hp_ctx_borrow_stream / hp_ctx_return_stream (context.hip): the stream to go back to while a caller's stream is borrowed; a
    // borrowed stream whose work the own stream has not been ordered behind yet; has the own stream been used since the last borrow?
    hipStream_t borrowed_from = nullptr, foreign = nullptr;
    bool own_dirty = true;
    hipEvent_t fence_ev[2] = {nullptr, nullptr};   // own -> borrowed, borrowed -> own
    int cu_count = 0;
    char name[128] = {0};
    // Every entry point that enqueues on the stream or touches a handle's host mirror holds this for the duration of the
    // call: the counterpart of the reference's per-object locks (replay_buffer.py:29,34,48; normalizer.py:22,27,42), so a
    // host feeder thread may call hp_buffer_store while another thread drives hp_agent_train_cycle.  Recursive because
    // entry points call each other.
    std::recursive_mutex mu;
    DevBuf reward_ws;                  // scratch of the host-array forms of hp_compute_reward / hp_is_success
};
void ctx_join_foreign(hp_ctx *c);   // context.hip: order the context's stream behind the work of a stream that was borrowed
struct CtxGuard {   // + the calling thread's current device: HIP keeps that per thread and a feeder thread starts on device 0
    std::lock_guard<std::recursive_mutex> g;
    explicit CtxGuard(hp_ctx *c) : g(c->mu) {
        (void)hipSetDevice(c->device);
        if (!c->borrowed_from) {         // (inside a borrow the launches go to the caller's stream: nothing to join, nothing dirtied)
            if (c->foreign) ctx_join_foreign(c);
            c->own_dirty = true;
        }
    }
};
#define HP_SERIALISE(handle) CtxGuard hp_serialise_guard_((handle)->ctx)

// pinned host staging for async H2D copies of caller-owned (pageable) arrays: the caller's memory
// is only touched by a CPU memcpy during the call; `fence` marks the last DMA that read the buffer.
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipEvent_t fence = nullptr;
    bool pending = false;
    int ensure(size_t need) {
        if (!fence) HP_CHECK_HIP(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
        if (pending) {
            HP_CHECK_HIP(hipEventSynchronize(fence));
            pending = false;
        }
        if (need <= bytes) return HP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        HP_CHECK_HIP(hipHostMalloc(&p, need, hipHostMallocDefault));
        bytes = need;
        return HP_OK;
    }
    int mark(hipStream_t s) {
        HP_CHECK_HIP(hipEventRecord(fence, s));
        pending = true;
        return HP_OK;
    }
    void release() {
        if (pending && fence) (void)hipEventSynchronize(fence);
        if (p) (void)hipHostFree(p);
        if (fence) (void)hipEventDestroy(fence);
        p = nullptr;
        fence = nullptr;
        bytes = 0;
        pending = false;
    }
};

// ------------------------------------------------------------------ random stream
#define MT_N 624
struct MtState {  // device-resident, same fields as numpy's legacy state tuple
    uint32_t key[MT_N];
    int32_t pos;
    int32_t pad[3];
};

struct hp_rng {
    hp_ctx *ctx = nullptr;
    MtState *d_state = nullptr;
    DevBuf scratch;  // test-hook outputs
};

// one drawn transition index record (her.py:24-33)
struct __attribute__((aligned(16))) PlanRec {
    int32_t e;    // episode index
    int32_t t;    // timestep
    int32_t fut;  // t + 1 + int(u2 * (T - t))
    int32_t her;  // u1 < future_p
};

// ------------------------------------------------------------------ replay buffer
struct BufMeta {  // device-resident mirror of replay_buffer's counters
    int64_t current_size;
    int64_t n_transitions_stored;
};

struct hp_buffer {
    hp_ctx *ctx = nullptr;
    int64_t size = 0;  // capacity in episodes
    int32_t T = 0, obs_dim = 0, goal_dim = 0, act_dim = 0;
    double *d_obs = nullptr, *d_ag = nullptr, *d_g = nullptr, *d_act = nullptr;
    BufMeta *d_meta = nullptr;
    // host mirror (slot policy is deterministic given n_new, so the host can track it)
    int64_t current_size = 0, n_transitions_stored = 0;
    // staging of the most recent store_episode batch (also the source of _update_normalizer)
    // one allocation (st_obs) holds obs | ag | g | actions of the staged batch, so the upload is ONE copy
    DevBuf st_obs, st_slots;
    double *st_ag = nullptr, *st_g = nullptr, *st_act = nullptr;
    PinnedBuf pin;
    int64_t staged_n = 0;
    // hp_buffer_store_pinned: tickets of the asynchronous copies out of caller-registered host blocks
    static constexpr int PIN_RING = 16;
    hipEvent_t pin_events[PIN_RING] = {nullptr};
    uint64_t pin_tickets = 0;
    // Throughput rows (hp_buffer_enable_f32_rows; SURVEY 8b `storage_dtype`): a float32 mirror of observations + actions packed one
    // (episode, timestep) per 128-byte line, and the goals (kept float64: rewards and relabelled goals stay bit-exact) packed
    // [ag_t | g_t] per 64-byte half line -- what hp_buffer_sample_dev_f32 reads.  Maintained behind every scatter.
    float *p_row = nullptr;      // [size][T + 1][row_w] floats: obs_t | actions_t (zeros at t = T) | 0
    double *p_goal = nullptr;    // [size][T + 1][goal_w] doubles: ag_t | g_t (zeros at t = T) | 0
    int32_t row_w = 0, goal_w = 0;
    // sampling scratch
    DevBuf plan, out;
    size_t ep_obs() const { return (size_t)(T + 1) * obs_dim; }
    size_t ep_ag() const { return (size_t)(T + 1) * goal_dim; }
    size_t ep_g() const { return (size_t)T * goal_dim; }
    size_t ep_act() const { return (size_t)T * act_dim; }
};
int buffer_launch_pack(hp_buffer *b, int64_t n_new);   // refresh the throughput rows of the episodes just scattered (no-op when off)

// ------------------------------------------------------------------ normalizer
#define NORM_MAX 256   // columns a normalizer can hold (bmirobot: 27 observations, 3 goals); hp_norm_create rejects more
struct NormDev {  // device-resident state; size <= NORM_MAX columns
    float local_sum[NORM_MAX], local_sumsq[NORM_MAX], local_count[4];
    float total_sum[NORM_MAX], total_sumsq[NORM_MAX], total_count[4];
    float sync[2 * NORM_MAX + 4];  // sum | sumsq | count snapshot exchanged between ranks
    float mean[NORM_MAX];
    double std[NORM_MAX];  // float64-valued (std_f32: float32 value widened)
};

struct hp_norm {
    hp_ctx *ctx = nullptr;
    int32_t size = 0;
    double eps = 1e-2, clip = 0;
    int32_t std_f32 = 0;
    NormDev *d = nullptr;
    DevBuf scratch, scratch2;
    PinnedBuf pin;
    bool in_recompute = false;
};

// Reward of a relabelled transition from the squared goal distance s (compute_reward, bmirobot_env_push_F.py:84-90):
//   sq_threshold >= 0: sparse, -(d > thr) == -(s >= sq_threshold) with sq_threshold the smallest double whose correctly
//                      rounded square root exceeds thr -- no square root on the device;
//   sq_threshold <  0: dense, -d (the env returns it in float64; the learner narrows it to float32, ddpg_agent.py:243).
#ifdef __HIPCC__
__device__ __forceinline__ float hp_reward(double s, double sq_threshold) {
    if (sq_threshold < 0.0) return (float)(-__dsqrt_rn(s));
    return (s >= sq_threshold) ? -1.0f : -0.0f;
}
// the value compute_reward itself returns: float32 widened (sparse) or float64 -d (dense)
__device__ __forceinline__ double hp_reward64(double s, double sq_threshold) {
    if (sq_threshold < 0.0) return -__dsqrt_rn(s);
    return (s >= sq_threshold) ? -1.0 : -0.0;
}
#endif

// rank exchange (comm.hip): one RCCL communicator bound to the context's stream
struct hp_comm {
    hp_ctx *ctx = nullptr;
    void *nccl = nullptr;   // ncclComm_t
    int rank = 0, world = 1;
};
int comm_allreduce_sum_f32(hp_comm *c, float *dev, size_t n);    // utils.py:43-48
int comm_allreduce_mean_f32(hp_comm *c, float *dev, size_t n);   // normalizer.py:60-64

// launchers implemented in the .hip files -------------------------------------------------
// rng.hip
int rng_launch_plan(hp_rng *rng, const BufMeta *d_meta, int64_t n_eps_fixed, int32_t T, int64_t batch,
                    int32_t n_batches, double future_p, PlanRec *d_plan, hipStream_t stream = nullptr);
int rng_launch_slots(hp_rng *rng, hp_buffer *buf, int64_t n_new, int64_t *d_slots);
// normalizer plan (batch_first transitions out of n_first staged episodes) + the first n_batches minibatch plans, one launch
int rng_launch_plan2(hp_rng *rng, int64_t n_first, int32_t T, int64_t batch_first, PlanRec *d_plan_first,
                     const BufMeta *d_meta, int64_t batch, int32_t n_batches, double future_p, PlanRec *d_plan);

// buffer.hip
int buffer_launch_gather_dict(hp_buffer *buf, const PlanRec *d_plan, int64_t batch, double sq_threshold,
                              double *d_out, float *d_r, double *d_r64);
int buffer_stage_and_store(hp_buffer *b, hp_rng *rng, const double *obs, const double *ag, const double *g,
                           const double *actions, int64_t n_new);
// the staging half alone (+ the host mirror of the counters): slots and scatter follow in k_cycle_open (cycle_open.hip)
int buffer_stage_for_cycle(hp_buffer *b, const double *obs, const double *ag, const double *g, const double *actions,
                           int64_t n_new);
// the same out of a device-registered host block (feeder ring); store_now: slots + scatter follow at once
int buffer_stage_pinned(hp_buffer *b, hp_rng *rng, const double *block, int64_t n_new, uint64_t *ticket, bool store_now);

// norm.hip
int norm_launch_update_from_plan(hp_norm *o, hp_norm *g, hp_buffer *b, const PlanRec *d_plan, int64_t rows,
                                 double clip_obs, bool recompute);
int norm_launch_begin(hp_norm *nz);
int norm_launch_end(hp_norm *nz);
