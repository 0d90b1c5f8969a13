// norm_device.h -- device side of the running normalizer shared by norm.hip's kernels and the cycle-opening kernel
// (cycle_open.hip).  Arithmetic contract: norm.hip's header.
#pragma once
#include "internal.h"

__device__ __forceinline__ double clipd(double v, double lo, double hi) {
    // np.clip == minimum(maximum(v, lo), hi)
    return fmin(fmax(v, lo), hi);
}

// recompute_stats (normalizer.py:40-57) for one column, begin and end in one go (single rank: nothing to exchange in
// between).  ls / lss / lc: the local accumulators as they stood; same expressions as k_norm_begin + k_norm_end.
__device__ __forceinline__ void norm_recompute_column(NormDev *nz, int c, int size, float ls, float lss, float lc,
                                                      double eps_sq, int std_f32, float &cnt_out) {
    const float cnt = __fadd_rn(nz->total_count[0], lc);
    cnt_out = cnt;
    if (c < size) {
        nz->sync[c] = ls;
        nz->sync[size + c] = lss;
        nz->local_sum[c] = 0.f;
        nz->local_sumsq[c] = 0.f;
        const float ts = __fadd_rn(nz->total_sum[c], ls);
        const float tss = __fadd_rn(nz->total_sumsq[c], lss);
        nz->total_sum[c] = ts;
        nz->total_sumsq[c] = tss;
        const float m = (float)__ddiv_rn((double)ts, (double)cnt);
        nz->mean[c] = m;
        const float var = __fsub_rn((float)__ddiv_rn((double)tss, (double)cnt), __fmul_rn(m, m));
        if (std_f32) nz->std[c] = (double)(float)__dsqrt_rn((double)fmaxf((float)eps_sq, var));
        else nz->std[c] = __dsqrt_rn(fmax(eps_sq, (double)var));
    }
}

// ddpg_agent._update_normalizer (:187-212) on the episodes staged by the last store:
// rows are the HER-sampled transitions in `plan`; obs -> o_norm, (relabelled) g -> g_norm.  The column sums are
// sequential in the row index (order-exact), the loads are not: a chunk of rows costs three round trips -- all plan
// records into LDS, all clipped values into LDS (1024 threads, every load independent), then one thread per column
// adds its column in row order -- instead of two dependent round trips per handful of rows (it was 16.6 us per cycle with 10 rows
// per trip).  recompute != 0 (single rank): recompute_stats of both normalizers follows in the same launch.
#define NORM_THREADS 1024
// Executed by one NORM_THREADS-thread workgroup; norm_lds = chunk_rows * (sizeof(PlanRec) + (obs_dim + goal_dim) * 8) bytes.
template <bool SC1_PLAN>
__device__ __forceinline__ void norm_update_from_plan_body(
    NormDev *onz, NormDev *gnz, const PlanRec *__restrict__ plan, long long rows, const double *__restrict__ s_obs,
    const double *__restrict__ s_ag, const double *__restrict__ s_g, int T, int obs_dim, int goal_dim, double clip_obs,
    int recompute, double o_eps_sq, int o_std_f32, double g_eps_sq, int g_std_f32, int chunk_rows, char *norm_lds) {
    PlanRec *sp = reinterpret_cast<PlanRec *>(norm_lds);                                   // [chunk_rows]
    double *sv = reinterpret_cast<double *>(norm_lds + (size_t)chunk_rows * sizeof(PlanRec));   // [chunk_rows][W]
    const int c = threadIdx.x;
    const bool goal = c >= NORM_MAX;            // threads [0, NORM_MAX): observation columns, [NORM_MAX, 2 NORM_MAX): goal columns
    const int j = goal ? c - NORM_MAX : c;
    NormDev *nz = goal ? gnz : onz;
    const int size = goal ? goal_dim : obs_dim;
    const bool act = c < 2 * NORM_MAX && j < size;
    const int W = obs_dim + goal_dim;
    double s = 0.0, ss = 0.0;
    for (long long r0 = 0; r0 < rows; r0 += chunk_rows) {
        const int n = (int)((rows - r0) < chunk_rows ? (rows - r0) : chunk_rows);
        for (int r = c; r < n; r += NORM_THREADS) {
            if (SC1_PLAN) {   // written by another workgroup of this launch: agent-scope loads (never a stale line of this XCD)
                const unsigned long long *q = reinterpret_cast<const unsigned long long *>(plan + r0 + r);
                const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PlanRec rec;
                rec.e = (int)(unsigned)lo; rec.t = (int)(unsigned)(lo >> 32); rec.fut = (int)(unsigned)hi; rec.her = (int)(unsigned)(hi >> 32);
                sp[r] = rec;
            } else {
                sp[r] = plan[r0 + r];
            }
        }
        __syncthreads();
        for (int idx = c; idx < n * W; idx += NORM_THREADS) {
            const int r = idx / W, col = idx - r * W;
            const PlanRec p = sp[r];
            const int k = col - obs_dim;
            const double *src = col < obs_dim ? s_obs + ((long long)p.e * (T + 1) + p.t) * obs_dim + col
                                              : (p.her ? s_ag + ((long long)p.e * (T + 1) + p.fut) * goal_dim + k
                                                       : s_g + ((long long)p.e * T + p.t) * goal_dim + k);
            sv[idx] = clipd(*src, -clip_obs, clip_obs);
        }
        __syncthreads();
        if (act) {
            const double *col = sv + (goal ? obs_dim + j : j);
            for (int r = 0; r < n; ++r) {
                const double v = col[(size_t)r * W];
                s = __dadd_rn(s, v);
                ss = __dadd_rn(ss, __dmul_rn(v, v));
            }
        }
        __syncthreads();   // the chunk is consumed before the next one overwrites it
    }
    if (c >= 2 * NORM_MAX) return;   // whole wavefronts: the barriers below count the ones that remain
    // normalizer.update: float32 accumulators += float64 column sums; count += rows
    float ls = 0.f, lss = 0.f;
    if (act) {
        ls = (float)__dadd_rn((double)nz->local_sum[j], s);
        lss = (float)__dadd_rn((double)nz->local_sumsq[j], ss);
    }
    const float lc = (float)((double)nz->local_count[0] + (double)rows);
    __syncthreads();   // every lane has read local_count before lane 0 of its wave rewrites it
    if (!recompute) {
        if (act) {
            nz->local_sum[j] = ls;
            nz->local_sumsq[j] = lss;
        }
        if (j == 0) nz->local_count[0] = lc;
        return;
    }
    float cnt;
    norm_recompute_column(nz, j, size, ls, lss, lc, goal ? g_eps_sq : o_eps_sq, goal ? g_std_f32 : o_std_f32, cnt);
    __syncthreads();   // total_count is read by every lane of the wave's normalizer above
    if (j == 0) {
        nz->sync[2 * size] = lc;
        nz->local_count[0] = 0.f;
        nz->total_count[0] = cnt;
    }
}

// LDS rows per chunk of norm_update_from_plan_body and the bytes they take
inline int norm_plan_chunk(int obs_dim, int goal_dim, long long rows, size_t *lds_bytes) {
    const int W = obs_dim + goal_dim;
    int chunk = (int)((48 * 1024) / (sizeof(PlanRec) + (size_t)W * 8));
    chunk = chunk > 256 ? 256 : chunk;
    chunk = rows < chunk ? (int)rows : chunk;
    *lds_bytes = (size_t)chunk * (sizeof(PlanRec) + (size_t)W * 8);
    return chunk;
}
