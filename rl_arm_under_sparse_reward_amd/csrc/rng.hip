// rng.hip -- device random stream object + the index-draw kernels of the HER sampler.
#include "mt19937_device.h"

// ------------------------------------------------------------------------------- kernels
// her.py:24-33 for `n_batches` consecutive minibatches: plan[b*batch + i] = (e, t, future_t, her).
// n_eps comes from the device-resident buffer counters when `meta` != nullptr (so a cached
// hipGraph keeps working while the buffer fills), else from n_eps_fixed.
__global__ __launch_bounds__(MT_THREADS) void k_draw_plan(MtState *st, const BufMeta *meta, long long n_eps_fixed,
                                                         int T, long long batch, int n_batches, double future_p,
                                                         PlanRec *plan) {
    __shared__ uint32_t ring[4][MT_N];
    __shared__ int ibuf[MT_IBUF];
    const long long n_eps = meta ? meta->current_size : n_eps_fixed;
    mt_her_plan(st, n_eps, T, batch, n_batches, future_p, plan, ring, ibuf);
}

// Cycle boundary: the index plan of ddpg_agent._update_normalizer (:187-212: T transitions out of the n_first episodes staged
// by the last store) and the first plans of the n_batches updates that follow (:145-147), drawn back to back from ONE load of
// the stream -- the same words in the same order as two k_draw_plan launches, one launch and one state round trip less.
__global__ __launch_bounds__(MT_THREADS) void k_draw_plan2(MtState *st, long long n_first, int T, long long batch_first,
                                                          PlanRec *plan_first, const BufMeta *meta, long long batch,
                                                          int n_batches, double future_p, PlanRec *plan) {
    __shared__ uint32_t ring[4][MT_N];
    __shared__ int ibuf[MT_IBUF];
    MtWg g;
    mt_load(g, st, ring, ibuf);
    mt_her_draw(g, n_first, T, batch_first, 1, future_p, plan_first);
    mt_her_draw(g, meta->current_size, T, batch, n_batches, future_p, plan);
    mt_store(g, st);
}

// replay_buffer._get_storage_idx (replay_buffer.py:57-71); updates the device counters.
__global__ __launch_bounds__(MT_THREADS) void k_draw_slots(MtState *st, BufMeta *meta, long long size, int T,
                                                          long long inc, long long *slots) {
    __shared__ uint32_t ring[4][MT_N];
    __shared__ int ibuf[MT_IBUF];
    const long long cur = meta->current_size;
    if (cur + inc <= size) {   // no draw: the stream is not even loaded
        MtWg g{};
        mt_draw_slots(g, cur, size, inc, slots);
    } else {
        MtWg g;
        mt_load(g, st, ring, ibuf);
        mt_draw_slots(g, cur, size, inc, slots);
        mt_store(g, st);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        meta->current_size = (cur + inc < size) ? cur + inc : size;
        meta->n_transitions_stored += (long long)T * inc;
    }
}

__global__ __launch_bounds__(MT_THREADS) void k_test_randint(MtState *st, long long low, uint32_t rng,
                                                            long long count, long long *out) {
    __shared__ uint32_t ring[4][MT_N];
    __shared__ int ibuf[MT_IBUF];
    MtWg g;
    mt_load(g, st, ring, ibuf);
    mt_draw_bounded(g, rng, count, [&](long long i, uint32_t v) { out[i] = low + (long long)v; });
    mt_store(g, st);
}

__global__ __launch_bounds__(MT_THREADS) void k_test_uniform(MtState *st, long long count, double *out) {
    __shared__ uint32_t ring[4][MT_N];
    __shared__ int ibuf[MT_IBUF];
    MtWg g;
    mt_load(g, st, ring, ibuf);
    mt_draw_double(g, count, [&](long long i, double u) { out[i] = u; });
    mt_store(g, st);
}

// ------------------------------------------------------------------------------ launchers
int rng_launch_plan(hp_rng *rng, const BufMeta *d_meta, int64_t n_eps_fixed, int32_t T, int64_t batch,
                    int32_t n_batches, double future_p, PlanRec *d_plan, hipStream_t stream) {
    HP_KLOG("k_draw_plan");
    hipLaunchKernelGGL(k_draw_plan, dim3(1), dim3(MT_THREADS), 0, stream ? stream : rng->ctx->stream, rng->d_state, d_meta,
                       (long long)n_eps_fixed, (int)T, (long long)batch, (int)n_batches, future_p, d_plan);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int rng_launch_plan2(hp_rng *rng, int64_t n_first, int32_t T, int64_t batch_first, PlanRec *d_plan_first,
                     const BufMeta *d_meta, int64_t batch, int32_t n_batches, double future_p, PlanRec *d_plan) {
    HP_KLOG("k_draw_plan2");
    hipLaunchKernelGGL(k_draw_plan2, dim3(1), dim3(MT_THREADS), 0, rng->ctx->stream, rng->d_state, (long long)n_first, (int)T,
                       (long long)batch_first, d_plan_first, d_meta, (long long)batch, (int)n_batches, future_p, d_plan);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int rng_launch_slots(hp_rng *rng, hp_buffer *buf, int64_t n_new, int64_t *d_slots) {
    hipLaunchKernelGGL(k_draw_slots, dim3(1), dim3(MT_THREADS), 0, rng->ctx->stream, rng->d_state, buf->d_meta,
                       (long long)buf->size, (int)buf->T, (long long)n_new, (long long *)d_slots);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// --------------------------------------------------------------------------------- C ABI
extern "C" {

int hp_rng_create(hp_ctx *ctx, hp_rng **out) {
    HP_REQUIRE(ctx && out, HP_ERR_INVALID, "hp_rng_create: null argument");
    hp_rng *r = new hp_rng();
    r->ctx = ctx;
    hipError_t e = hipMalloc((void **)&r->d_state, sizeof(MtState));
    if (e != hipSuccess) {
        delete r;
        hp_set_error("hp_rng_create: hipMalloc failed: %s", hipGetErrorString(e));
        return HP_ERR_HIP;
    }
    *out = r;
    return hp_rng_seed(r, 5489u);
}

// numpy _legacy_seeding(int) -> mt19937_seed(): init_genrand, pos = 624.
int hp_rng_seed(hp_rng *rng, uint32_t seed) {
    HP_REQUIRE(rng, HP_ERR_INVALID, "hp_rng_seed: null handle");
    HP_SERIALISE(rng);
    MtState h;
    memset(&h, 0, sizeof(h));
    for (int i = 0; i < MT_N; ++i) {
        h.key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    h.pos = MT_N;
    // stream-ordered with the kernels that use the state; the pageable source is copied before return
    HP_CHECK_HIP(hipMemcpyAsync(rng->d_state, &h, sizeof(h), hipMemcpyHostToDevice, rng->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(rng->ctx->stream));
    return HP_OK;
}

int hp_rng_set_state(hp_rng *rng, const uint32_t *key624, int32_t pos) {
    HP_REQUIRE(rng && key624, HP_ERR_INVALID, "hp_rng_set_state: null argument");
    HP_SERIALISE(rng);
    HP_REQUIRE(pos >= 0 && pos <= MT_N, HP_ERR_INVALID, "hp_rng_set_state: pos %d outside [0, 624]", pos);
    MtState h;
    memset(&h, 0, sizeof(h));
    memcpy(h.key, key624, sizeof(h.key));
    h.pos = pos;
    HP_CHECK_HIP(hipMemcpyAsync(rng->d_state, &h, sizeof(h), hipMemcpyHostToDevice, rng->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(rng->ctx->stream));
    return HP_OK;
}

int hp_rng_get_state(hp_rng *rng, uint32_t *key624, int32_t *pos) {
    HP_REQUIRE(rng && key624 && pos, HP_ERR_INVALID, "hp_rng_get_state: null argument");
    HP_SERIALISE(rng);
    MtState h;
    HP_CHECK_HIP(hipMemcpyAsync(&h, rng->d_state, sizeof(h), hipMemcpyDeviceToHost, rng->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(rng->ctx->stream));
    memcpy(key624, h.key, sizeof(h.key));
    *pos = h.pos;
    return HP_OK;
}

int hp_rng_randint(hp_rng *rng, int64_t low, int64_t high, int64_t count, int64_t *host_out) {
    HP_REQUIRE(rng && host_out, HP_ERR_INVALID, "hp_rng_randint: null argument");
    HP_SERIALISE(rng);
    HP_REQUIRE(high > low, HP_ERR_INVALID, "high <= 0");  // numpy's message for randint(0, 0)
    HP_REQUIRE(high - low - 1 < 0xFFFFFFFFll, HP_ERR_INVALID, "hp_rng_randint: range needs more than 32 bits");
    HP_REQUIRE(count >= 0, HP_ERR_INVALID, "hp_rng_randint: negative count");
    if (count == 0) return HP_OK;
    HP_TRY(rng->scratch.ensure((size_t)count * 8));
    hipLaunchKernelGGL(k_test_randint, dim3(1), dim3(MT_THREADS), 0, rng->ctx->stream, rng->d_state,
                       (long long)low, (uint32_t)(high - low - 1), (long long)count, rng->scratch.as<long long>());
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(host_out, rng->scratch.p, (size_t)count * 8, hipMemcpyDeviceToHost, rng->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(rng->ctx->stream));
    return HP_OK;
}

int hp_rng_uniform(hp_rng *rng, int64_t count, double *host_out) {
    HP_REQUIRE(rng && host_out, HP_ERR_INVALID, "hp_rng_uniform: null argument");
    HP_SERIALISE(rng);
    HP_REQUIRE(count >= 0, HP_ERR_INVALID, "hp_rng_uniform: negative count");
    if (count == 0) return HP_OK;
    HP_TRY(rng->scratch.ensure((size_t)count * 8));
    hipLaunchKernelGGL(k_test_uniform, dim3(1), dim3(MT_THREADS), 0, rng->ctx->stream, rng->d_state,
                       (long long)count, rng->scratch.as<double>());
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(host_out, rng->scratch.p, (size_t)count * 8, hipMemcpyDeviceToHost, rng->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(rng->ctx->stream));
    return HP_OK;
}

void hp_rng_destroy(hp_rng *rng) {
    if (!rng) return;
    if (rng->d_state) (void)hipFree(rng->d_state);
    rng->scratch.release();
    delete rng;
}

}  // extern "C"
