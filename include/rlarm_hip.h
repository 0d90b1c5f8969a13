/*
 * rlarm_hip.h -- C ABI of librlarm_hip.so: the MI355X (gfx950) implementation of the
 * HER-replay + DDPG-update hot path of PiggyCh/RL_arm_under_sparse_reward.
 *
 * The reference has no FFI of its own (pure Python); its seam is the
 * set of duck-typed Python objects wired in ddpg_agent.py:24-53.  Each entry point below
 * names the reference interface it stands in for (file:line relative to the reference
 * root).  The Python mirror of those objects lives in rl_arm_under_sparse_reward_amd/ and
 * binds this header with ctypes; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative hp_status otherwise;
 *     hp_last_error() returns a thread-local message for the last failure;
 *   - "host" pointers are caller-owned and only read/written during the call (the
 *     reference's copy-in / copy-out semantics, replay_buffer.py:39-42, her.py:26);
 *   - "dev" pointers are HIP device addresses (e.g. torch.Tensor.data_ptr());
 *   - all kernels are enqueued on the context's stream (hp_ctx_set_stream), so work
 *     is ordered with whatever else the caller enqueues on that stream;
 *   - calls on the handles of one context are serialised by a per-context lock held for the
 *     duration of each call (the reference's per-object locks, replay_buffer.py:29,34,48 and
 *     normalizer.py:22,27,42): a host feeder thread may call hp_buffer_store while another
 *     thread drives hp_agent_train_cycle; the order in which the calls win the lock is the
 *     order of their work on the stream and of their draws from the shared hp_rng.
 *     hp_ctx_synchronize waits outside the lock.
 *   - there is NO CPU fallback: without a gfx950 device hp_ctx_create fails.
 */
#ifndef RLARM_HIP_H
#define RLARM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of the STABLE surface declared in this header.  2 (round 4): the diagnostic / test-hook entry points moved to
 * rlarm_hip_debug.h (no stability promise), hp_agent_fused_status is gone (round 3), hp_agent_train_cycle_pinned,
 * hp_peer_set_gate, hp_ctx_pci_bus_id, hp_agent_status were added.  3 (round 5): hp_buffer_sample_dev (device-output fused
 * sampler) was added.  4 (round 6): hp_ctx_get_stream and hp_ctx_borrow_stream / hp_ctx_return_stream were added (a host that hands
 * device outputs to a framework has them written on the framework's stream for that call instead of rebinding the context), and the sampler's throughput modes
 * (hp_buffer_enable_f32_rows, hp_buffer_sample_dev_f32, hp_buffer_sample_dev_fast).  hp_abi_version() returns the
 * library's value; a host must refuse a library whose version differs from the header it was built against. */
#define HP_ABI_VERSION 4

typedef enum {
    HP_OK = 0,
    HP_ERR_INVALID = -1,   /* bad argument (maps to ValueError in the Python mirror) */
    HP_ERR_EMPTY = -2,     /* sampling an empty buffer (numpy: "ValueError: high <= 0") */
    HP_ERR_HIP = -3,       /* a HIP runtime call failed */
    HP_ERR_NODEVICE = -4,  /* no gfx950 device visible */
    HP_ERR_STATE = -5      /* call sequence error */
} hp_status;

typedef struct hp_ctx hp_ctx;
typedef struct hp_rng hp_rng;
typedef struct hp_buffer hp_buffer;
typedef struct hp_norm hp_norm;
typedef struct hp_agent hp_agent;
typedef struct hp_comm hp_comm;
typedef struct hp_peer hp_peer;

int hp_abi_version(void);
const char *hp_last_error(void);

/* ---- context ---------------------------------------------------------------------- */
int hp_ctx_create(int device_id, hp_ctx **out);
/* stream == NULL selects a stream owned by the context; to run on the legacy default stream (e.g. to be ordered
 * with a framework that uses it) pass hipStreamLegacy, i.e. (void *)1.  A change of stream keeps the context's work in
 * order: the new stream waits (on the device, not the host) for what the context enqueued on the old one. */
int hp_ctx_set_stream(hp_ctx *ctx, void *hip_stream);
/* the stream the context enqueues on right now (its own one unless hp_ctx_set_stream changed it) */
int hp_ctx_get_stream(hp_ctx *ctx, void **hip_stream);
/* A caller's stream for the duration of ONE call sequence, without rebinding the context: between hp_ctx_borrow_stream and
 * hp_ctx_return_stream (same thread; the context's lock is held in between) every launch of the library goes to `hip_stream`
 * (NULL: the null stream, a framework's default), ordered on the device behind what the context's own stream held; the own
 * stream is ordered behind the borrowed stream's work when it is next used.  No host synchronisation.  What
 * replay_buffer.sample_device wraps hp_buffer_sample_dev in, so that torch consumes the outputs in ITS stream's order while a
 * fused learner on the same context keeps its own stream and its cached graphs. */
int hp_ctx_borrow_stream(hp_ctx *ctx, void *hip_stream);
int hp_ctx_return_stream(hp_ctx *ctx);
int hp_ctx_synchronize(hp_ctx *ctx);
int hp_ctx_device_name(hp_ctx *ctx, char *buf, size_t len);
int hp_ctx_pci_bus_id(hp_ctx *ctx, char *buf, size_t len);      /* "0000:05:00.0"; len >= 16 */
void hp_ctx_destroy(hp_ctx *ctx);

/* ---- random stream ------------------------------------------------------------------
 * Device-resident twin of numpy's legacy global RandomState, which the reference draws
 * from at her.py:24,25,29,31 and replay_buffer.py:64,67 (seeded at train.py:36).
 * State is interoperable with np.random.get_state()/set_state(): (key[624], pos). */
int hp_rng_create(hp_ctx *ctx, hp_rng **out);
int hp_rng_seed(hp_rng *rng, uint32_t seed);                              /* np.random.seed(int) */
int hp_rng_set_state(hp_rng *rng, const uint32_t *key624, int32_t pos);   /* np.random.set_state */
int hp_rng_get_state(hp_rng *rng, uint32_t *key624, int32_t *pos);        /* synchronises */
/* test hooks: the primitive draws, results copied to host (synchronises) */
int hp_rng_randint(hp_rng *rng, int64_t low, int64_t high, int64_t count, int64_t *host_out);
int hp_rng_uniform(hp_rng *rng, int64_t count, double *host_out);
void hp_rng_destroy(hp_rng *rng);

/* ---- episodic replay buffer ------------------------------------------------------------
 * replay_buffer.py:11-71.  Storage is float64 [size, T+1, obs], [size, T+1, goal],
 * [size, T, goal], [size, T, action] in HBM (same layout as the reference's numpy arrays). */
int hp_buffer_create(hp_ctx *ctx, int64_t size_episodes, int32_t T, int32_t obs_dim, int32_t goal_dim,
                     int32_t act_dim, hp_buffer **out);
/* replay_buffer.store_episode (:32-43) + _get_storage_idx (:57-71).  Host arrays, float64,
 * C-contiguous [n_new, T+1, obs] / [n_new, T+1, goal] / [n_new, T, goal] / [n_new, T, action].
 * Slot selection runs on the device and draws from `rng` exactly when the reference does. */
int hp_buffer_store(hp_buffer *buf, hp_rng *rng, const double *obs, const double *ag, const double *g,
                    const double *actions, int64_t n_new);
/* Upload n_new episodes into the buffer's device staging area WITHOUT storing them (no slot draw, counters untouched):
 * the temporary dict ddpg_agent._update_normalizer builds from the episodes it is handed (ddpg_agent.py:187-203).
 * hp_norm_update_from_staged then samples from exactly these episodes.  n_new == 0 -> "high <= 0" like numpy. */
int hp_buffer_stage(hp_buffer *buf, const double *obs, const double *ag, const double *g, const double *actions,
                    int64_t n_new);
/* Multi-process host feeder (the rollout workers of ddpg_agent.py:101-142 as processes): the episodes of a wave are
 * written by the workers into ONE host block [obs n x (T+1) x obs_dim | ag | g | actions] that the trainer registered
 * with the device (hp_host_register, e.g. a POSIX shared-memory segment).  hp_buffer_store_pinned is store_episode on
 * such a block without the CPU copy: the DMA reads the block directly and asynchronously (slot draw and scatter as in
 * hp_buffer_store); the block may be rewritten once hp_buffer_store_done reports its ticket done (wait != 0 blocks,
 * outside the context lock). */
int hp_host_register(hp_ctx *ctx, void *host, size_t bytes);
int hp_host_unregister(hp_ctx *ctx, void *host);
int hp_buffer_store_pinned(hp_buffer *buf, hp_rng *rng, const double *block, int64_t n_new, uint64_t *ticket);
int hp_buffer_store_done(hp_buffer *buf, uint64_t ticket, int32_t wait, int32_t *done);
int hp_buffer_info(hp_buffer *buf, int64_t *size, int64_t *current_size, int64_t *n_transitions_stored,
                   int32_t *T);
/* slots chosen by the most recent hp_buffer_store (parity tests); synchronises */
int hp_buffer_last_slots(hp_buffer *buf, int64_t *host_out, int64_t n);
/* copy episodes [first, first+n) of one array back to the host: which = 0 obs, 1 ag, 2 g, 3 actions */
int hp_buffer_read(hp_buffer *buf, int32_t which, int64_t first, int64_t n, double *host_out);

/* replay_buffer.sample (:46-55) -> her_sampler.sample_her_transitions (her.py:13-41) with the
 * sparse goal-distance reward of bmirobot_env_push_F.py:20-23,84-90 inlined.
 *   future_p      her.py:8  (1 - 1/(1+replay_k));  0 disables relabelling
 *   sq_threshold  smallest double s with sqrt(s) > distance_threshold (reward = -(s >= sq_threshold));
 *                 a NEGATIVE value selects the dense reward of compute_reward (:89-90): r = float32(-sqrt(s))
 * Any host output pointer may be NULL.  r is float32 [B] with bit patterns 0x80000000 / 0xBF800000.
 * e/t/future_t/her expose the drawn indices for parity tests (future_t is defined for every
 * sample; the reference uses it only where her != 0). */
typedef struct {
    double *obs, *ag, *g, *actions, *obs_next, *ag_next; /* [B, dim] float64 */
    float *r;                                            /* [B] */
    int64_t *e, *t, *future_t;                           /* [B] */
    uint8_t *her;                                        /* [B] */
    double *r64;                                         /* [B] what compute_reward itself returns: the float32 reward
                                                            widened (sparse) or -d in float64 (dense, :89-90) */
} hp_sample_out;
int hp_buffer_sample(hp_buffer *buf, hp_rng *rng, int64_t batch, double future_p, double sq_threshold,
                     const hp_sample_out *host_out);

/* replay_buffer.sample (:46-55) followed by the learner's own preprocessing of the minibatch (ddpg_agent.py:227-243:
 * _preproc_og clip :214-217, o_norm / g_norm .normalize normalizer.py:67-70, np.concatenate, torch.tensor(.., float32)) in ONE
 * gather kernel with DEVICE outputs -- the `hp_sample(... out device ptrs x, x', a, r)` of SURVEY.md section 8b.  What a
 * GPU learner consumes never crosses PCIe:
 *   x        float32 [B, obs+goal]  inputs_norm_tensor       (ddpg_agent.py:236,240)
 *   x_next   float32 [B, obs+goal]  inputs_next_norm_tensor  (:237,241)
 *   actions  float32 [B, act]       actions_tensor           (:242; unscaled, the critic divides by max_action itself)
 *   r        float32 [B]            r_tensor                 (:243; reshape to [B, 1])
 *   e, t, future_t (int64 [B]), her (uint8 [B]): the drawn indices, for parity tests
 * Any pointer may be NULL.  The normalizers' default_clip_range is the clip of normalize(); clip_obs is arguments.py:87.
 * Same random stream, same draws and same float64 arithmetic as hp_buffer_sample + hp_norm_normalize (bit-identical
 * float32 results).  Asynchronous on the context's stream; outputs are caller-owned device memory (torch tensors). */
typedef struct {
    float *x, *x_next, *actions, *r;
    int64_t *e, *t, *future_t;
    uint8_t *her;
} hp_sample_dev_out;
int hp_buffer_sample_dev(hp_buffer *buf, hp_rng *rng, hp_norm *o_norm, hp_norm *g_norm, int64_t batch, double future_p,
                         double sq_threshold, double clip_obs, const hp_sample_dev_out *dev_out);
/* Throughput mode of the sampler (SURVEY 8b's `storage_dtype` = fp32; opt-in, NOT bit-identical to the reference's float64 rows):
 * hp_buffer_enable_f32_rows builds -- from what the buffer holds now, and behind every later store -- a float32 mirror of the
 * observations and actions laid out for the gather (one (episode, timestep) per 128-byte line: obs_t | action_t), the goals
 * staying float64.  hp_buffer_sample_dev_f32 = hp_buffer_sample_dev reading that mirror: same stream, same draws, indices,
 * relabelled goals (her.py:35-36), rewards (:38) and goal columns of x / x_next BIT-identical; actions identical (the learner takes
 * float32(actions) either way); the observation columns are those of float32-rounded observations (replay_buffer.py:23-27 stores
 * float64), about half the bytes per transition.  Needs obs_dim + act_dim <= 32.  The float64 arrays stay the source of truth:
 * hp_buffer_sample / hp_buffer_sample_dev / the fused learner are unaffected. */
int hp_buffer_enable_f32_rows(hp_buffer *buf);
int hp_buffer_sample_dev_f32(hp_buffer *buf, hp_rng *rng, hp_norm *o_norm, hp_norm *g_norm, int64_t batch, double future_p,
                             double sq_threshold, double clip_obs, const hp_sample_dev_out *dev_out);
/* Fast draw (SURVEY 8b's `rng_mode` = Philox; opt-in, NOT the reference's random stream): the same device-output minibatch with the
 * index draw inside the gather kernel -- no hp_rng, no sequential draw launch.  Transition m of call `call` takes
 * (r0..r3) = Philox4x32-10(counter (m, call), key `seed`) [Random123] and draws her.py:24-33's four values from them:
 * e = floor(r0 N / 2^32), t = floor(r1 T / 2^32), her = r2 2^-32 < future_p, future_t = t + 1 + floor(r3 (T - t) / 2^32).
 * Deterministic in (seed, call): the caller owns the counter (a rank passes seed + rank and call = 0, 1, 2, ...).  f32_rows != 0
 * reads the throughput rows.  Gather, relabel, reward, clip and normalisation are those of hp_buffer_sample_dev, bit for bit. */
int hp_buffer_sample_dev_fast(hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, int64_t batch, double future_p, double sq_threshold,
                              double clip_obs, uint64_t seed, uint64_t call, int32_t f32_rows, const hp_sample_dev_out *dev_out);

/* ---- GoalEnv reward / success as batched device ops --------------------------------------------
 * compute_reward (bmirobot_env_push_F.py:84-90 -> goal_distance :20-23; byte-identical in
 * bmirobot_env_pickandplace_v2.py:84-90) and _is_success (:243-245) for n goal pairs [n][goal_dim] float64.
 *   dense == 0: sparse reward -(d > distance_threshold).astype(float32) -> sparse_out (bits 0x80000000 / 0xBF800000)
 *   dense != 0: -d in float64 -> dense_out
 *   hp_is_success: (d < distance_threshold).astype(float32)
 * The predicates are evaluated on the squared distance against the smallest double whose correctly rounded square
 * root passes the comparison, so they agree with numpy's sqrt-then-compare for every input (bit-exact for
 * goal_dim < 8, where numpy's add.reduce is a left-to-right sum).  *_dev: device pointers, asynchronous on the context's
 * stream; the plain forms take host arrays and synchronise. */
int hp_compute_reward_dev(hp_ctx *ctx, const double *ag_dev, const double *g_dev, int64_t n, int32_t goal_dim,
                          double distance_threshold, int32_t dense, float *sparse_out_dev, double *dense_out_dev);
int hp_is_success_dev(hp_ctx *ctx, const double *ag_dev, const double *g_dev, int64_t n, int32_t goal_dim,
                      double distance_threshold, float *out_dev);
int hp_compute_reward(hp_ctx *ctx, const double *ag_host, const double *g_host, int64_t n, int32_t goal_dim,
                      double distance_threshold, int32_t dense, float *sparse_out_host, double *dense_out_host);
int hp_is_success(hp_ctx *ctx, const double *ag_host, const double *g_host, int64_t n, int32_t goal_dim,
                  double distance_threshold, float *out_host);

/* ---- running normalizer ----------------------------------------------------------------
 * normalizer.py:5-70.  float32 accumulators/totals/mean; std is float64 when std_f32 == 0
 * (what numpy >= 2 computes for the reference's expression) or float32 when std_f32 != 0
 * (numpy 1.19.2, the version the reference pins). */
int hp_norm_create(hp_ctx *ctx, int32_t size, double eps, double default_clip_range, int32_t std_f32,
                   hp_norm **out);
int hp_norm_update(hp_norm *nz, const double *v_host, int64_t rows);          /* normalizer.update :25-31 */
/* normalizer.recompute_stats :40-57 split around the cross-rank mean (:34-38, :60-64):
 *   begin: snapshot + zero the local accumulators; *dev_sync -> float32[2*size+1] = sum|sumsq|count
 *   (caller all-reduces that vector and divides by world size -- RCCL -- or leaves it alone)
 *   end:   totals += sync; mean/std recomputed. */
int hp_norm_recompute_begin(hp_norm *nz, void **dev_sync, int64_t *n_floats);
int hp_norm_recompute_end(hp_norm *nz);
int hp_norm_recompute(hp_norm *nz);                                            /* single-rank: begin+end */
/* host copies of the state (any pointer may be NULL); synchronises */
int hp_norm_get(hp_norm *nz, float *mean, double *std, float *total_sum, float *total_sumsq, float *total_count,
                float *local_sum, float *local_sumsq, float *local_count);
int hp_norm_set_stats(hp_norm *nz, const float *mean, const double *std);    /* checkpoint load */
/* normalizer.normalize :67-70 on host data (rollout / test path).  clip_range < 0 -> default */
int hp_norm_normalize(hp_norm *nz, const double *v_host, int64_t rows, double clip_range, double *out_host);

/* ddpg_agent._update_normalizer (:187-212): HER-sample T transitions out of the n_new
 * episodes most recently passed to hp_buffer_store (still staged on the device), clip to
 * +-clip_obs (:214-217), update both normalizers.  Does NOT recompute (call hp_norm_recompute*). */
int hp_norm_update_from_staged(hp_buffer *buf, hp_rng *rng, hp_norm *o_norm, hp_norm *g_norm, double future_p,
                               double clip_obs);

/* ---- DDPG learner -----------------------------------------------------------------------
 * models.py:11-44 (actor 30-256-256-256-4, critic 34-256-256-256-1), ddpg_agent.py:220-277,
 * torch.optim.Adam defaults (ddpg_agent.py:42-43), utils.py:6-69 flat parameter order. */
typedef struct {
    int32_t obs_dim, goal_dim, act_dim, hidden;
    int32_t batch;           /* transitions per update (arguments.py:88) */
    int32_t grad_world_size; /* informational (grads are SUMmed across ranks, utils.py:47) */
    /* Python floats in the reference -> doubles here; they are narrowed to float32 exactly where
     * torch narrows them (tensor-scalar ops) so the arithmetic matches. */
    double max_action;       /* env action_max (bmirobot_env_push_F.py:75-78) */
    double gamma;            /* arguments.py:89 */
    double action_l2;        /* arguments.py:90 */
    double lr_actor, lr_critic; /* arguments.py:91-92 */
    double polyak;           /* arguments.py:93 */
    double clip_obs;         /* arguments.py:87 */
    double clip_range;       /* arguments.py:95 (normalizer default_clip_range) */
    double adam_beta1, adam_beta2, adam_eps; /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 */
} hp_agent_cfg;

enum { HP_NET_ACTOR = 0, HP_NET_CRITIC = 1, HP_NET_ACTOR_TARGET = 2, HP_NET_CRITIC_TARGET = 3 };

int hp_agent_create(hp_ctx *ctx, const hp_agent_cfg *cfg, hp_agent **out);
int64_t hp_agent_param_count(hp_agent *ag, int32_t net);   /* 140548 / 140801 for the reference sizes */
/* flat float32 vectors in named_parameters() order (utils.py:18-40): fc1.weight, fc1.bias, ... */
int hp_agent_set_params(hp_agent *ag, int32_t net, const float *flat_host, int64_t n);
int hp_agent_get_params(hp_agent *ag, int32_t net, float *flat_host, int64_t n);
/* gradients of the last hp_agent_update_minibatch / hp_agent_forward_backward (the sampled update loops of a single
 * rank consume the weight gradients in the optimizer epilogue without writing them out; bias gradients are always
 * written) */
int hp_agent_get_grads(hp_agent *ag, int32_t net, float *flat_host, int64_t n);  /* net = actor|critic */
int hp_agent_get_adam(hp_agent *ag, int32_t net, float *m_host, float *v_host, int64_t n, int64_t *step);

/* One ddpg_agent._update_network (:250-277) on a caller-provided, already normalised minibatch
 * (host float32: x [B,obs+goal], x_next [B,obs+goal], actions [B,act], r [B]).
 * losses_host[0] = actor_loss, [1] = critic_loss.  Synchronises. */
int hp_agent_update_minibatch(hp_agent *ag, const float *x, const float *x_next, const float *actions,
                              const float *r, float *losses_host);

/* The hot loop (ddpg_agent.py:145-147): n_updates x { sample B transitions, relabel, reward, clip,
 * normalise, 5 forwards, 3 backwards, Adam x2 } entirely on the device.  Asynchronous. */
int hp_agent_sample_and_update(hp_agent *ag, hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, hp_rng *rng,
                               double future_p, double sq_threshold, int32_t n_updates);
/* losses of the most recent updates, newest last: out[2*i] actor, out[2*i+1] critic.  Synchronises. */
int hp_agent_get_losses(hp_agent *ag, float *out_host, int32_t n_last);
/* ddpg_agent._soft_update_target_network (:220-222) for both nets.  Asynchronous. */
int hp_agent_soft_update(hp_agent *ag);
/* actor forward on host inputs (rollout side, ddpg_agent.py:114-116): x [rows, obs+goal] -> actions [rows, act] */
int hp_agent_actor_forward(hp_agent *ag, int32_t net, const float *x_host, int64_t rows, float *actions_host);
/* critic forward on host inputs (models.py:28-44: the critic scales the actions by 1/max_action itself):
 * x [rows, obs+goal] float32 (normalised), actions [rows, act] float32 -> q [rows] */
int hp_agent_critic_forward(hp_agent *ag, int32_t net, const float *x_host, const float *actions_host, int64_t rows,
                            float *q_host);
/* rollout side in one call (ddpg_agent._preproc_inputs :163-171 + actor, :114-116 / :288-292): float64 observation
 * and goal rows -> normalise with the two normalizers (clip at each normalizer's default_clip_range), float32, actor
 * forward -> actions [rows, act].  clip_obs > 0 additionally clips the raw values first (_preproc_og); the reference's
 * rollouts do not, so the drop-in passes 0.  rows = the environments of a vectorised feeder stepped in lockstep. */
int hp_agent_act(hp_agent *ag, hp_norm *o_norm, hp_norm *g_norm, int32_t net, const double *obs_host,
                 const double *g_host, int64_t rows, double clip_obs, float *actions_host);

/* Policy calls that do not queue behind training: hp_agent_policy_snapshot copies the online actor and both normalizers'
 * statistics (stream-ordered with the updates, no host wait); hp_agent_act_snapshot evaluates the most recent COMPLETE
 * snapshot on a second stream, so a feeder can step its environments while a training cycle runs (its policy lags the
 * learner by at most one snapshot interval; the reference's rollouts are synchronous, ddpg_agent.py:101-150).  Same
 * arithmetic as hp_agent_act.  Callable from another host thread. */
int hp_agent_policy_snapshot(hp_agent *ag, hp_norm *o_norm, hp_norm *g_norm);
int hp_agent_act_snapshot(hp_agent *ag, const double *obs_host, const double *g_host, int64_t rows, double clip_obs,
                          float *actions_host);

/* Split-phase update for data-parallel ranks (utils.sync_grads, utils.py:43-48):
 *   forward_backward: sample + forwards + backwards, leaves SUM-able gradients in one flat device
 *   vector (*dev_grads, both nets, padded layout -- every rank has the same layout);
 *   apply: Adam on whatever that vector holds after the caller's all-reduce. */
int hp_agent_forward_backward(hp_agent *ag, hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, hp_rng *rng,
                              double future_p, double sq_threshold);
int hp_agent_grad_buffer(hp_agent *ag, void **dev_grads, int64_t *n_floats);
int hp_agent_param_buffer(hp_agent *ag, void **dev_params, int64_t *n_floats); /* actor|critic, for Bcast (utils.py:6-15) */
int hp_agent_apply(hp_agent *ag);
/* ddpg_agent.py:33-34: targets := online nets.  Also the call that makes parameters written through
 * hp_agent_param_buffer (broadcast from rank 0) take effect in the kernels' weight copies. */
int hp_agent_sync_targets(hp_agent *ag);

/* ---- rank exchange on RCCL (one process per GPU; replaces mpi4py) -------------------------------
 * The three exchanges of the reference: sync_networks (utils.py:6-15, Bcast from rank 0), sync_grads
 * (utils.py:43-48, Allreduce SUM of the flat gradients, every update) and the normalizer's
 * _mpi_average (normalizer.py:60-64, Allreduce SUM / size).  Bootstrap: rank 0 calls
 * hp_comm_unique_id, the 128 bytes travel to the other ranks by any side channel (the Python mirror
 * uses torch.distributed), every rank calls hp_comm_create.  Collectives are enqueued on the
 * context's stream.  RCCL is resolved with dlopen at first use; without it these return HP_ERR_STATE. */
int hp_comm_unique_id(uint8_t *out128);
int hp_comm_create(hp_ctx *ctx, const uint8_t *id128, int32_t rank, int32_t world, hp_comm **out);
int hp_comm_info(hp_comm *comm, int32_t *rank, int32_t *world);
int hp_comm_allreduce_sum_f32(hp_comm *comm, void *dev, int64_t n);    /* in place, device pointer */
int hp_comm_allreduce_mean_f32(hp_comm *comm, void *dev, int64_t n);   /* SUM then / world (float32) */
int hp_comm_broadcast_f32(hp_comm *comm, void *dev, int64_t n, int32_t root);
void hp_comm_destroy(hp_comm *comm);
/* Attach (or detach with NULL) a communicator: hp_agent_sample_and_update and hp_agent_train_cycle then
 * all-reduce the gradients between backward and Adam inside the library (and train_cycle the
 * normalizer sums), so the data-parallel loop needs no host round trip per update and stays inside
 * the cycle's hipGraph.  The split-phase calls above keep working for callers with their own transport. */
int hp_agent_set_comm(hp_agent *ag, hp_comm *comm);
/* Reduction applied to the gradients when a communicator is attached: 0 (default) = SUM, exactly utils.py:47 -- N
 * ranks therefore step with N times the single-rank gradient; 1 = MEAN (SUM / world size), for callers who want the
 * learning rate to keep its single-rank meaning at a larger global batch. */
int hp_agent_set_grad_reduce(hp_agent *ag, int32_t mean);
/* diagnostic: 0 = no cycle built yet, 1 = hp_agent_train_cycle replays a cached hipGraph, 2 = it issues eager
 * launches because a capture containing collectives was refused by the runtime */
int hp_agent_cycle_mode(hp_agent *ag, int32_t *mode);

/* ---- rank exchange as one-shot all-reduces over peer memory (xGMI), fused with the optimizer ----------------------------
 * Same three exchanges as above, but without a collective library on the critical path of an update: every rank exports
 * one block of exchange memory (hipIpcGetMemHandle), maps the other ranks' blocks, and the kernel that applies Adam
 * reads the peers' gradient vectors itself and sums them in rank order (utils.py:43-48 SUM; bit-identical parameters on
 * every rank).  Bootstrap: hp_peer_create on every rank -> exchange the 64-byte handles through any side channel ->
 * hp_peer_connect(all handles, rank order).  Waits are bounded (RLARM_PEER_TIMEOUT_S, default 20 s) and a timeout is FATAL
 * for the exchange: the kernel whose wait gave up skips its sum / optimizer step, sets a sticky error word that makes every
 * later exchange kernel return at once, and mirrors it into pinned host memory -- hp_agent_train_cycle /
 * hp_agent_sample_and_update then fail with HP_ERR_STATE at their next call (no synchronisation needed), hp_peer_status
 * reads the word explicitly.  The reference's MPI Allreduce (utils.py:47) would block forever instead; silent divergence of
 * the replicas is the one outcome that is excluded.  One node only (ranks that can map each other's device memory). */
int hp_peer_create(hp_ctx *ctx, int32_t rank, int32_t world, int64_t n_grad_floats, hp_peer **out, uint8_t *handle64);
int hp_peer_connect(hp_peer *peer, const uint8_t *handles_world_x_64);
/* normalizer._mpi_average (normalizer.py:60-64) and other small vectors (<= 1024 floats): in place, SUM or SUM / world
 * (mean != 0), enqueued on the context's stream; collective */
int hp_peer_allreduce_f32(hp_peer *peer, void *dev, int64_t n, int32_t mean);
/* collective self-check of the gradient channel (flags, both buffer parities, system-scope loads of every peer's vector) on an
 * exactly representable pattern: *mismatches = number of wrong elements on this rank (top bit: a wait timed out) */
int hp_peer_selfcheck(hp_peer *peer, uint32_t *mismatches);
int hp_peer_status(hp_peer *peer, uint32_t *error);
/* Gate mode (every rank alike, before the first exchange): each wait of the gradient exchange runs in a ONE-WAVEFRONT kernel
 * of its own in front of the kernel that consumes the peers' data, instead of inside that kernel's hundreds of workgroups.
 * Needed when ranks share one physical device (rehearsals of an N-GPU job on one GPU): there a rank's progress depends on
 * its kernels being co-resident with the other ranks' WAITING kernels, and workgroups spinning by the hundred can starve
 * them of registers / LDS.  Costs one kernel boundary per wait; default off. */
int hp_peer_set_gate(hp_peer *peer, int32_t on);
/* 1: one-shot exchange (every rank reads every peer's whole gradient vector), 2: reduce-scatter + all-gather over the same
 * peer memory (default from 4 ranks; RLARM_PEER_PHASES=1|2).  Both sum in rank order: bit-identical results. */
int hp_peer_phases(hp_peer *p, int32_t *phases);
void hp_peer_destroy(hp_peer *peer);
/* Attach (or detach with NULL): hp_agent_sample_and_update / hp_agent_train_cycle then exchange the gradients through
 * peer memory inside the optimizer kernel and the normalizer sums through the mailboxes.  hp_agent_grad_buffer-style
 * host-driven exchange keeps working.  n_grad_floats of the peer must equal hp_agent_grad_buffer's length. */
int hp_agent_set_peer(hp_agent *ag, hp_peer *peer);

/* One training cycle's learner half (ddpg_agent.py:143-150) as one cached hipGraph:
 *   store n_new episodes -> update normalizers (+recompute) -> n_batches updates -> soft update.
 * Requires single-rank operation (no cross-rank exchange inside the graph). Asynchronous. */
int hp_agent_train_cycle(hp_agent *ag, hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, hp_rng *rng,
                         const double *obs, const double *ag_host, const double *g, const double *actions,
                         int64_t n_new, double future_p, double sq_threshold, int32_t n_batches);
/* The same cycle (ddpg_agent.py:143-150) on episodes that lie in a host block registered with the device (hp_host_register;
 * layout and ticket as hp_buffer_store_pinned): what a multi-process rollout feeder (ddpg_agent.py:101-144 spread over worker
 * processes) calls per wave -- the store is an asynchronous DMA out of the shared ring, no CPU copy. */
int hp_agent_train_cycle_pinned(hp_agent *ag, hp_buffer *buf, hp_norm *o_norm, hp_norm *g_norm, hp_rng *rng,
                                const double *block, int64_t n_new, double future_p, double sq_threshold,
                                int32_t n_batches, uint64_t *ticket);

/* timing hook for bench.py: average device time (ms) of the kernels tagged `which` over the
 * last recorded region; see DESIGN.md "Measurement". */
/* Which kernels hp_agent_sample_and_update / hp_agent_train_cycle run for this agent (chosen at creation from the batch
 * size and the RLARM_* switches): *engine = 0 layer-per-launch, 8 thin row slabs (slab8.h, *slab_rows = 4 / 8 / 16),
 * 32 the 32-row engine (slab32.h); *dw_split = 0 for 32 x 32 weight-gradient tiles
 * (gemm_lds.h), else the number of batch-row slices per 64 x 64 tile (dw64.h).  Any pointer may be null. */
int hp_agent_engine(hp_agent *ag, int32_t *engine, int32_t *slab_rows, int32_t *dw_split);
/* Launch structure of a sequence of n_updates sampled updates WITH their optimizer steps (ddpg_agent.py:145-147;
 * hp_agent_sample_and_update, hp_agent_train_cycle) on this agent as it is now: *form = 0 chain launch + weight-gradient launch per
 * update; 1 split launch (target networks one update ahead, the critic's weight gradients inside the chain launch) + the actor's
 * weight-gradient launch -- single rank: optimizer steps in both launches' epilogues; data-parallel ranks on a device each: the same
 * with the tile-wise rank exchange in front of every step, or gradients only + exchange + optimizer launches (RCCL, two-phase
 * peer memory).  Gradient-only calls (hp_agent_forward_backward) always take form 0.  The kernels by name:
 * hp_agent_update_kernels (rlarm_hip_debug.h). */
int hp_agent_update_form(hp_agent *ag, int32_t n_updates, int32_t *form);
/* Sticky health word of the learner, free for the host (pinned memory, no synchronisation): 0 = healthy.  Bit 0: a bounded
 * in-launch hand-off gave up; one bit per source beside it (several may be set): bit 4 critic chains -> weight-gradient tiles,
 * bit 5 actor chains -> critic optimizer step, bit 6 cycle-opening launch; the launches since skipped work, every later hp_agent_train_cycle* /
 * hp_agent_sample_and_update / hp_agent_get_losses fails with HP_ERR_STATE, the agent must be recreated. */
int hp_agent_status(hp_agent *ag, uint32_t *fault);
int hp_agent_profile(hp_agent *ag, int32_t enable);
int hp_agent_profile_read(hp_agent *ag, double *ms_out, int32_t n);

void hp_agent_destroy(hp_agent *ag);
void hp_norm_destroy(hp_norm *nz);
void hp_buffer_destroy(hp_buffer *buf);

#ifdef __cplusplus
}
#endif
#endif /* RLARM_HIP_H */
