// norm.hip -- running mean/std normalizer on the device (reference: normalizer.py:5-70,
// ddpg_agent.py:187-217).
//
// Arithmetic contract (pinned bit-for-bit by tests/golden/normalizer.npz):
//   update:   float64 column sums, rows added in order (numpy's axis-0 reduction of a
//             C-contiguous array is a sequential row accumulation), then
//             float32 accumulator = float32(float64(accumulator) + column_sum)
//   totals:   float32 adds;  mean = float32 division
//   std:      sqrt(max(eps^2, sumsq/n - (sum/n)^2)), float32 inner expression, evaluated in
//             float64 (numpy >= 2) or float32 (numpy 1.19.2) -- see hp_norm_create.
// One lane per column: a row read is obs_dim consecutive doubles (coalesced), the column sums
// are independent, so no cross-lane reduction is needed and the result is order-exact.
// Built with -ffp-contract=off.
#include "norm_device.h"

__device__ __forceinline__ void norm_accumulate(NormDev *nz, int c, double s, double ss) {
    nz->local_sum[c] = (float)__dadd_rn((double)nz->local_sum[c], s);        // f32 += f64
    nz->local_sumsq[c] = (float)__dadd_rn((double)nz->local_sumsq[c], ss);
}

// normalizer.update(v) for v [rows, size] float64 on the device
__global__ void k_norm_update_rows(NormDev *nz, const double *__restrict__ v, long long rows, int size) {
    const int c = threadIdx.x;
    if (c < size) {
        double s = 0.0, ss = 0.0;
        for (long long r = 0; r < rows; ++r) {
            double x = v[r * size + c];
            s = __dadd_rn(s, x);
            ss = __dadd_rn(ss, __dmul_rn(x, x));
        }
        norm_accumulate(nz, c, s, ss);
    }
    if (c == 0) nz->local_count[0] = (float)((double)nz->local_count[0] + (double)rows);
}

// ddpg_agent._update_normalizer (:187-212) on the episodes staged by the last store (norm_device.h)
__global__ __launch_bounds__(NORM_THREADS) void k_norm_update_from_plan(
    NormDev *onz, NormDev *gnz, const PlanRec *__restrict__ plan, long long rows, const double *__restrict__ s_obs,
    const double *__restrict__ s_ag, const double *__restrict__ s_g, int T, int obs_dim, int goal_dim, double clip_obs,
    int recompute, double o_eps_sq, int o_std_f32, double g_eps_sq, int g_std_f32, int chunk_rows) {
    extern __shared__ __attribute__((aligned(16))) char norm_lds[];
    norm_update_from_plan_body<false>(onz, gnz, plan, rows, s_obs, s_ag, s_g, T, obs_dim, goal_dim, clip_obs, recompute, o_eps_sq,
                                      o_std_f32, g_eps_sq, g_std_f32, chunk_rows, norm_lds);
}

// recompute_stats part 1 (normalizer.py:41-48): snapshot + reset the local accumulators
__global__ void k_norm_begin(NormDev *nz, int size) {
    const int c = threadIdx.x;
    if (c < size) {
        nz->sync[c] = nz->local_sum[c];
        nz->sync[size + c] = nz->local_sumsq[c];
        nz->local_sum[c] = 0.f;
        nz->local_sumsq[c] = 0.f;
    }
    if (c == 0) {
        nz->sync[2 * size] = nz->local_count[0];
        nz->local_count[0] = 0.f;
    }
}

// recompute_stats part 2 (normalizer.py:51-57) on the (possibly rank-averaged) snapshot
__global__ void k_norm_end(NormDev *nz, int size, double eps_sq, int std_f32) {
    const int c = threadIdx.x;
    const float cnt = __fadd_rn(nz->total_count[0], nz->sync[2 * size]);
    if (c < size) {
        const float ts = __fadd_rn(nz->total_sum[c], nz->sync[c]);
        const float tss = __fadd_rn(nz->total_sumsq[c], nz->sync[size + c]);
        nz->total_sum[c] = ts;
        nz->total_sumsq[c] = tss;
        const float m = (float)__ddiv_rn((double)ts, (double)cnt);   // correctly rounded float32 quotient
        nz->mean[c] = m;
        const float var = __fsub_rn((float)__ddiv_rn((double)tss, (double)cnt), __fmul_rn(m, m));
        if (std_f32) {
            // float32 sqrt via float64: innocuous double rounding (53 >= 2*24+2), correctly rounded
            nz->std[c] = (double)(float)__dsqrt_rn((double)fmaxf((float)eps_sq, var));
        } else {
            nz->std[c] = __dsqrt_rn(fmax(eps_sq, (double)var));
        }
    }
    __syncthreads();
    if (c == 0) nz->total_count[0] = cnt;
}

// normalizer.normalize (:67-70): clip((v - mean) / std, -clip, clip), float64
__global__ void k_norm_normalize(const NormDev *nz, const double *__restrict__ v, long long n, int size, double clip,
                                 double *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % size);
    const double z = __ddiv_rn(__dsub_rn(v[i], (double)nz->mean[c]), nz->std[c]);
    out[i] = clipd(z, -clip, clip);
}

__global__ void k_norm_init(NormDev *nz) {
    const int c = threadIdx.x;
    if (c < NORM_MAX) {
        nz->local_sum[c] = nz->local_sumsq[c] = nz->total_sum[c] = nz->total_sumsq[c] = nz->mean[c] = 0.f;
        nz->std[c] = 1.0;  // normalizer.py:20
    }
    for (int k = c; k < 2 * NORM_MAX + 4; k += blockDim.x) nz->sync[k] = 0.f;
    if (c < 4) {
        nz->local_count[c] = 0.f;
        nz->total_count[c] = (c == 0) ? 1.f : 0.f;  // normalizer.py:17: total_count starts at ONE
    }
}

__global__ void k_norm_set(NormDev *nz, const float *mean, const double *std, int size) {
    const int c = threadIdx.x;
    if (c < size) {
        nz->mean[c] = mean[c];
        nz->std[c] = std[c];
    }
}

// ------------------------------------------------------------------------------ launchers
int norm_launch_update_from_plan(hp_norm *o, hp_norm *g, hp_buffer *b, const PlanRec *d_plan, int64_t rows,
                                 double clip_obs, bool recompute) {
    // rows per chunk: what 48 KB of LDS hold (plan record + one float64 per column and row), at most 256
    const int W = b->obs_dim + b->goal_dim;
    int chunk = (int)((48 * 1024) / (sizeof(PlanRec) + (size_t)W * 8));
    chunk = chunk > 256 ? 256 : chunk;
    chunk = rows < chunk ? (int)rows : chunk;
    const size_t lds = (size_t)chunk * (sizeof(PlanRec) + (size_t)W * 8);
    HP_KLOG("k_norm_update_from_plan");
    hipLaunchKernelGGL(k_norm_update_from_plan, dim3(1), dim3(NORM_THREADS), lds, o->ctx->stream, o->d, g->d, d_plan,
                       (long long)rows, b->st_obs.as<double>(), b->st_ag, b->st_g, (int)b->T,
                       (int)b->obs_dim, (int)b->goal_dim, clip_obs, recompute ? 1 : 0, o->eps * o->eps, o->std_f32,
                       g->eps * g->eps, g->std_f32, chunk);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int norm_launch_begin(hp_norm *nz) {
    HP_KLOG("k_norm_begin");
    hipLaunchKernelGGL(k_norm_begin, dim3(1), dim3(NORM_MAX), 0, nz->ctx->stream, nz->d, nz->size);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int norm_launch_end(hp_norm *nz) {
    HP_KLOG("k_norm_end");
    hipLaunchKernelGGL(k_norm_end, dim3(1), dim3(NORM_MAX), 0, nz->ctx->stream, nz->d, nz->size, nz->eps * nz->eps,
                       nz->std_f32);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// --------------------------------------------------------------------------------- C ABI
extern "C" {

int hp_norm_create(hp_ctx *ctx, int32_t size, double eps, double default_clip_range, int32_t std_f32, hp_norm **out) {
    HP_REQUIRE(ctx && out, HP_ERR_INVALID, "hp_norm_create: null argument");
    HP_REQUIRE(size > 0 && size <= NORM_MAX, HP_ERR_INVALID, "hp_norm_create: size=%d must be in [1, %d]", size, NORM_MAX);
    hp_norm *nz = new hp_norm();
    nz->ctx = ctx;
    nz->size = size;
    nz->eps = eps;
    nz->clip = default_clip_range;
    nz->std_f32 = std_f32 ? 1 : 0;
    hipError_t e = hipMalloc((void **)&nz->d, sizeof(NormDev));
    if (e != hipSuccess) {
        delete nz;
        hp_set_error("hp_norm_create: hipMalloc failed: %s", hipGetErrorString(e));
        return HP_ERR_HIP;
    }
    hipLaunchKernelGGL(k_norm_init, dim3(1), dim3(NORM_MAX), 0, ctx->stream, nz->d);
    *out = nz;
    return HP_OK;
}

int hp_norm_update(hp_norm *nz, const double *v_host, int64_t rows) {
    HP_REQUIRE(nz && v_host, HP_ERR_INVALID, "hp_norm_update: null argument");
    HP_SERIALISE(nz);
    HP_REQUIRE(rows >= 0, HP_ERR_INVALID, "hp_norm_update: negative rows");
    hipStream_t s = nz->ctx->stream;
    const size_t bytes = (size_t)rows * nz->size * 8;
    if (rows > 0) {
        HP_TRY(nz->scratch.ensure(bytes));
        HP_TRY(nz->pin.ensure(bytes));
        memcpy(nz->pin.p, v_host, bytes);
        HP_CHECK_HIP(hipMemcpyAsync(nz->scratch.p, nz->pin.p, bytes, hipMemcpyHostToDevice, s));
        HP_TRY(nz->pin.mark(s));
    }
    hipLaunchKernelGGL(k_norm_update_rows, dim3(1), dim3(NORM_MAX), 0, s, nz->d, nz->scratch.as<double>(), (long long)rows,
                       nz->size);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int hp_norm_recompute_begin(hp_norm *nz, void **dev_sync, int64_t *n_floats) {
    HP_REQUIRE(nz, HP_ERR_INVALID, "hp_norm_recompute_begin: null handle");
    HP_SERIALISE(nz);
    HP_REQUIRE(!nz->in_recompute, HP_ERR_STATE, "hp_norm_recompute_begin: previous recompute not ended");
    HP_TRY(norm_launch_begin(nz));
    nz->in_recompute = true;
    if (dev_sync) *dev_sync = nz->d->sync;
    if (n_floats) *n_floats = 2 * nz->size + 1;
    return HP_OK;
}

int hp_norm_recompute_end(hp_norm *nz) {
    HP_REQUIRE(nz, HP_ERR_INVALID, "hp_norm_recompute_end: null handle");
    HP_SERIALISE(nz);
    HP_REQUIRE(nz->in_recompute, HP_ERR_STATE, "hp_norm_recompute_end: begin was not called");
    nz->in_recompute = false;
    return norm_launch_end(nz);
}

int hp_norm_recompute(hp_norm *nz) {
    HP_REQUIRE(nz, HP_ERR_INVALID, "hp_norm_recompute: null handle");
    HP_SERIALISE(nz);
    HP_TRY(hp_norm_recompute_begin(nz, nullptr, nullptr));
    return hp_norm_recompute_end(nz);
}

int hp_norm_get(hp_norm *nz, float *mean, double *std, float *total_sum, float *total_sumsq, float *total_count,
                float *local_sum, float *local_sumsq, float *local_count) {
    HP_REQUIRE(nz, HP_ERR_INVALID, "hp_norm_get: null handle");
    HP_SERIALISE(nz);
    NormDev h;
    HP_CHECK_HIP(hipMemcpyAsync(&h, nz->d, sizeof(h), hipMemcpyDeviceToHost, nz->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(nz->ctx->stream));
    const int n = nz->size;
    if (mean) memcpy(mean, h.mean, n * 4);
    if (std) memcpy(std, h.std, n * 8);
    if (total_sum) memcpy(total_sum, h.total_sum, n * 4);
    if (total_sumsq) memcpy(total_sumsq, h.total_sumsq, n * 4);
    if (total_count) total_count[0] = h.total_count[0];
    if (local_sum) memcpy(local_sum, h.local_sum, n * 4);
    if (local_sumsq) memcpy(local_sumsq, h.local_sumsq, n * 4);
    if (local_count) local_count[0] = h.local_count[0];
    return HP_OK;
}

int hp_norm_set_stats(hp_norm *nz, const float *mean, const double *std) {
    HP_REQUIRE(nz && mean && std, HP_ERR_INVALID, "hp_norm_set_stats: null argument");
    HP_SERIALISE(nz);
    hipStream_t s = nz->ctx->stream;
    const int n = nz->size;
    constexpr size_t MB = NORM_MAX * 4, SB = NORM_MAX * 8;
    HP_TRY(nz->scratch2.ensure(MB + SB));
    HP_TRY(nz->pin.ensure(MB + SB));
    char *h = static_cast<char *>(nz->pin.p);
    memcpy(h, mean, n * 4);
    memcpy(h + MB, std, n * 8);
    HP_CHECK_HIP(hipMemcpyAsync(nz->scratch2.p, h, MB + SB, hipMemcpyHostToDevice, s));
    HP_TRY(nz->pin.mark(s));
    hipLaunchKernelGGL(k_norm_set, dim3(1), dim3(NORM_MAX), 0, s, nz->d, nz->scratch2.as<float>(),
                       reinterpret_cast<const double *>(nz->scratch2.as<char>() + MB), n);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int hp_norm_normalize(hp_norm *nz, const double *v_host, int64_t rows, double clip_range, double *out_host) {
    HP_REQUIRE(nz && v_host && out_host, HP_ERR_INVALID, "hp_norm_normalize: null argument");
    HP_SERIALISE(nz);
    HP_REQUIRE(rows >= 0, HP_ERR_INVALID, "hp_norm_normalize: negative rows");
    if (rows == 0) return HP_OK;
    hipStream_t s = nz->ctx->stream;
    const long long n = (long long)rows * nz->size;
    HP_TRY(nz->scratch.ensure((size_t)n * 16));
    double *d_in = nz->scratch.as<double>(), *d_out = d_in + n;
    HP_CHECK_HIP(hipMemcpyAsync(d_in, v_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
    const double clip = clip_range < 0 ? nz->clip : clip_range;
    hipLaunchKernelGGL(k_norm_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nz->d, d_in, n, nz->size,
                       clip, d_out);
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(out_host, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

int hp_norm_update_from_staged(hp_buffer *b, hp_rng *rng, hp_norm *o_norm, hp_norm *g_norm, double future_p,
                               double clip_obs) {
    HP_REQUIRE(b && rng && o_norm && g_norm, HP_ERR_INVALID, "hp_norm_update_from_staged: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(b->staged_n > 0, HP_ERR_STATE, "hp_norm_update_from_staged: no staged episodes (call hp_buffer_store first)");
    HP_REQUIRE(o_norm->size == b->obs_dim && g_norm->size == b->goal_dim, HP_ERR_INVALID,
               "hp_norm_update_from_staged: normalizer sizes do not match the buffer");
    const int64_t rows = b->T;  // ddpg_agent.py:194 -- num_transitions = T regardless of episode count
    HP_TRY(b->plan.ensure(rows * sizeof(PlanRec)));
    HP_TRY(rng_launch_plan(rng, nullptr, b->staged_n, b->T, rows, 1, future_p, b->plan.as<PlanRec>()));
    return norm_launch_update_from_plan(o_norm, g_norm, b, b->plan.as<PlanRec>(), rows, clip_obs, false);
}

void hp_norm_destroy(hp_norm *nz) {
    if (!nz) return;
    if (nz->d) (void)hipFree(nz->d);
    nz->scratch.release();
    nz->scratch2.release();
    nz->pin.release();
    delete nz;
}

}  // extern "C"
