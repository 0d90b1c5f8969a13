/*
 * rlarm_hip_debug.h -- diagnostic and test-hook entry points of librlarm_hip.so.
 *
 * NOT part of the stable surface HP_ABI_VERSION (rlarm_hip.h) names: these exist for the parity tests (tests/), the
 * micro-benchmarks (tools/ubench/) and bench.py's roofline / calibration fields, and may change with any build.  A host that
 * only drives the hot path never needs them.  Time-line builds (-DSLAB_TIMELINE, `make -C csrc timeline`) export three more
 * symbols that no header declares: hp_debug_gemm_wg_timeline, hp_debug_gemm_blk_timeline, hp_debug_split_timeline.
 */
#ifndef RLARM_HIP_DEBUG_H
#define RLARM_HIP_DEBUG_H

#include "rlarm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: average microseconds per dependent trivial kernel on the context's stream, measured
 * as an n-node captured hipGraph (graph != 0) or n eager launches (the launch floor in DESIGN.md) */
int hp_ctx_launch_floor(hp_ctx *ctx, int n, int graph, double *us_per_kernel);
/* diagnostic: microseconds a hipEvent pair reads with nothing between the two records (the bracketing overhead
 * inside every per-launch event measurement of hp_agent_profile; bench.py subtracts it) */
int hp_ctx_event_pair_us(hp_ctx *ctx, int reps, double *us);
/* diagnostic: shader clock in MHz observed by a probe kernel enqueued now (DVFS state under this load) */
int hp_ctx_clock_mhz(hp_ctx *ctx, double *mhz);

/* diagnostic: average device microseconds of the sampler's two kernels (index draw; gather + relabel + reward into the
 * reference's dict layout), `reps` back-to-back launches each, no host copies.  Consumes 1 + reps index draws. */
int hp_buffer_sample_device_us(hp_buffer *buf, hp_rng *rng, int64_t batch, double future_p, double sq_threshold,
                               int32_t reps, double *draw_us, double *gather_us);

/* the same for hp_buffer_sample_dev's fused kernel (gather + relabel + reward + clip + normalise -> float32, device outputs
 * into library scratch); f32_rows != 0: hp_buffer_sample_dev_f32's kernel on the throughput rows */
int hp_buffer_sample_dev_us(hp_buffer *buf, hp_rng *rng, hp_norm *o_norm, hp_norm *g_norm, int64_t batch, double future_p,
                            double sq_threshold, double clip_obs, int32_t reps, int32_t f32_rows, double *draw_us, double *gather_us);

/* diagnostic: device microseconds per hp_buffer_sample_dev_fast launch (index draw inside the gather kernel) */
int hp_buffer_sample_dev_fast_us(hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, int64_t batch, double future_p, double sq_threshold,
                                 double clip_obs, int32_t reps, int32_t f32_rows, double *us);

/* test hook: load torch.optim.Adam state (exp_avg, exp_avg_sq in the flat order of utils.py:18-27; either may be NULL) and the
 * number of optimizer steps already taken (shared by both optimizers, ddpg_agent.py:272,277 step together) */
int hp_agent_set_adam(hp_agent *ag, int32_t net, const float *m_host, const float *v_host, int64_t n, int64_t step);

/* diagnostic: microseconds per launch of ONE stage of the update, repeated n times in a captured
 * hipGraph (kind: 6 optimizer kernel, 8 polyak, 10 forward + backward of the active engine, 11 its chain kernel only,
 * 12 its weight-gradient launch + optimizer only) -- the per-stage numbers quoted in DESIGN.md / bench.py */
int hp_agent_debug_chain(hp_agent *ag, int32_t kind, int32_t n, double *us_per_launch);
/* diagnostic: the kernels a sequence of n_updates sampled updates (hp_agent_sample_and_update) enqueues on this agent as it is
 * now -- engine, RLARM_* switches, attached communicator / peer exchange --, read off the launch logic itself (it runs under a
 * stream capture that is discarded: nothing executes).  out = comma-separated names in launch order with markers "#open",
 * "#prologue", "#update" (one per update), "#close"; "rccl:ncclAllReduce" stands for RCCL's own kernel.  caller_exchanges != 0:
 * the host-driven form instead (hp_agent_forward_backward, the caller's all-reduce -- "host:all_reduce" --, hp_agent_apply), one
 * update.  bench.py's `config.engine.kernels_per_update` and the kernel its roofline names come from here. */
int hp_agent_update_kernels(hp_agent *ag, hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, hp_rng *rng, double future_p,
                            double sq_threshold, int32_t n_updates, int32_t caller_exchanges, char *out, int32_t out_len);
/* diagnostic: stage-boundary time stamps (100 MHz ticks) of the slab kernels; only a build with
 * -DSLAB_TIMELINE writes them (tools/ubench/), a production build returns zeros */
int hp_agent_debug_timeline(hp_agent *ag, uint64_t *out192);

/* diagnostic: a ~200 us calibration of the box this process landed on, so that a slow box can be told from a regression in a
 * bench line (about one box in seven of the pool ran every kernel of this path ~1.4 x slower at the same shader clock):
 *   out[0] launch floor, us per dependent trivial kernel in a captured hipGraph
 *   out[1] LDS-DMA stream rate of ONE compute unit, GB/s (global_load_lds_dwordx4 of a 256 KiB block resident in L2, the
 *          weight stream of one 256 x 256 layer)
 *   out[2] cycles per v_mfma_f32_4x4x1_16b_f32 in a dependent chain (shader clock ticks)
 *   out[3] shader clock during the probes, MHz */
int hp_ctx_calibrate(hp_ctx *ctx, double *out4);

#ifdef __cplusplus
}
#endif
#endif /* RLARM_HIP_DEBUG_H */
