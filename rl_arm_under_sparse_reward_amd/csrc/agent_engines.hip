// agent_engines.hip -- the update's device code and its launch logic on gfx950: row-slab chain kernels (slab8.h: 4/8/16-row
// slabs on v_mfma_f32_4x4x1, slab32.h: 32-row slabs on v_mfma_f32_32x32x2), weight-gradient launches (gemm_lds.h 32 x 32
// tiles, dw64.h split 64 x 64 tiles) with the optimizer in their epilogue and the sampler's look-ahead riding along,
// the peer-exchange optimizer kernels, polyak, and the forward-only entry points (policy / critic rows).
// Reference: models.py:11-44, ddpg_agent.py:214-277, torch.optim.Adam (ddpg_agent.py:42-43).
//
// ---- why it is shaped like this (measured) -------------------------------------
// One update at batch 256 is 0.7 GFLOP over ~16 strictly dependent layers: microseconds of FP32-MFMA time, so the cost
// is the number of dependent launches and the per-CU weight stream, not math.  Every product is fp32 MFMA (no TF32 on
// gfx950; the 1e-5 loss parity needs fp32); all state is device resident and every kernel argument is constant across
// updates, so a whole training cycle is one cached hipGraph (agent.hip).
#ifdef SLAB_TIMELINE
#define ADAM_TL 1
#endif
#include "agent_device.h"
#include "gemm_lds.h"

// gradients: barrier + rank-ordered sum + Adam.  n4 = arena floats / 4; u = index of the update in its sequence.
__global__ __launch_bounds__(256) void k_peer_adam(const PeerDev D, const AdamFuse F, int n4, int u, int mean) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_g, epoch);
    if (!peer_wait(D, D.flags_g[D.rank], epoch)) return;   // dead exchange: no step from a partial sum (peer.h)
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    split_reset_by_block0(F);
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= n4) return;
    bool quad;
    const int idx0 = adam_quad_remap(F.am, t0, quad), t = idx0 >> 2;   // which float4 of the arena this thread steps (agent_device.h)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t bytes = (size_t)n4 * 16;
    for (int q = 0; q < D.world; ++q) {   // rank order: the same float32 sum on every rank
        const float4 v = peer_load4(D.grad[q][par], bytes, (unsigned)t * 16u, q == D.rank);
        if (q == 0) acc = v;
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (mean) {   // SUM / world, float32 true division (what the RCCL path's k_scale_div does)
        const float w = (float)D.world;
        acc.x /= w; acc.y /= w; acc.z /= w; acc.w /= w;
    }
    const float g[4] = {acc.x, acc.y, acc.z, acc.w};
    if (F.keep_grads) *reinterpret_cast<float4 *>(const_cast<float *>(F.grads_base) + 4 * (size_t)t) = acc;
    if (quad) adam_step4_quad(F, idx0, g);
    else adam_apply4(F, idx0, g);
}


// two-phase exchange, phase 2 (phase 1 = k_peer_reduce_slice in peer.hip): every element's sum comes from the rank that
// owns its slice; Adam on all of them.  Same values as k_peer_adam computes itself: bit-identical.
__global__ __launch_bounds__(256) void k_peer_adam2(const PeerDev D, const AdamFuse F, int n4, int u) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_r, epoch);
    if (!peer_wait(D, D.flags_r[D.rank], epoch, 3u)) return;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    split_reset_by_block0(F);
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= n4) return;
    bool quad;
    const int idx0 = adam_quad_remap(F.am, t0, quad), t = idx0 >> 2;
    const int owner = t / peer_slice_len(D, n4);
    const float4 acc = peer_load4(D.red[owner][par], (size_t)n4 * 16, (unsigned)t * 16u, owner == D.rank);
    const float g[4] = {acc.x, acc.y, acc.z, acc.w};
    if (F.keep_grads) *reinterpret_cast<float4 *>(const_cast<float *>(F.grads_base) + 4 * (size_t)t) = acc;
    if (quad) adam_step4_quad(F, idx0, g);
    else adam_apply4(F, idx0, g);
}

static int peer_enqueue_adam(hp_peer *p, const AdamFuse &F, int n_arena, int u, bool mean) {
    const int n4 = n_arena / 4;
    if (p->phases == 2) {
        HP_TRY(peer_enqueue_reduce_slice(p, n4, u, mean));
        HP_TRY(peer_enqueue_gate(p, 3, u));
        HP_KLOG("k_peer_adam2");
        hipLaunchKernelGGL(k_peer_adam2, dim3((n4 + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev, F, n4, u);
    } else {
        HP_TRY(peer_enqueue_gate(p, 1, u));
        HP_KLOG("k_peer_adam");
        hipLaunchKernelGGL(k_peer_adam, dim3((n4 + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev, F, n4, u, mean ? 1 : 0);
    }
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// the 4x4x1 slab engine, compiled for two slab heights (see slab8.h)
#define S8_NRG 1
#define S8_NS s8r4
#include "slab8.h"
#undef S8_NRG
#undef S8_NS
#define S8_NRG 2
#define S8_NS s8r8
#include "slab8.h"
#undef S8_NRG
#undef S8_NS
#define S8_NRG 4
#define S8_NS s8r16
#include "slab8.h"
#undef S8_NRG
#undef S8_NS
#undef S8_ROWS
#undef S8_RING
#undef S8_RPW
#include "slab32.h"

// Weight-gradient tiles (+ optimizer) with the sampler's look-ahead riding along.  When the chain kernel occupies every
// CU (batch 1024: 256 chain workgroups) it has no room for its spare workgroups -- a workgroup appended to a full launch
// starts when the first chain ends and then runs alone (k_fb_slab8 51.5 instead of 37.8 us) -- so the index plan of
// update u + 2 and the gather of update u + 1's inputs move into THIS launch, whose 296 tile workgroups leave half of the
// CUs' slots free: blocks [tiles, tiles + n_plan) draw, the next n_ahead gather.  Same device functions as the chain
// kernel's spare workgroups, same order of draws in the stream: identical bits.
struct RideArgs {
    int n_plan, n_ahead;
    MtState *rng;
    const BufMeta *meta;
    PlanRec *next_plan;
    double future_p;
    int T, plan_batch;
    GatherSrc ahead;
    float *aXT, *aXA, *aXP;
    int ldx, act_off, act_dim;
    float max_action;
};

template <bool ADAM, bool UNI = false, bool PEER = false>
__device__ __forceinline__ void gemm_ride_body(const GemmGroup &grp, const AdamFuse *F, const RideArgs &R, int tiles,
                                               const PeerTile *PT = nullptr, const TileHead *TH = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[GL_LDS_FLOATS];
    __shared__ float bsum[GL_WAVES][32];
    if (ADAM && !PEER && grp.loss_wg && blockIdx.x == gridDim.x - 1) {   // behind the riders: the loss log's own workgroup (gemm_lds.h)
        gemm_loss_wg(*F);
        return;
    }
    if (grp.bias0 > 0 && (int)blockIdx.x >= grp.bias0 && (int)blockIdx.x < tiles) {   // (`tiles` counts the bias panels behind the tiles)
        gemm_bias_tile<ADAM>(grp, F, (int)blockIdx.x - grp.bias0, lds, bsum);
        return;
    }
    if ((int)blockIdx.x < tiles) {
        gemm_tile<ADAM, UNI, false, PEER>(grp, F, (int)blockIdx.x, lds, bsum, blockIdx.x == 0 && (PEER || !grp.loss_wg), PT, TH);
        return;
    }
    const int extra = (int)blockIdx.x - tiles;
#ifdef SLAB_TIMELINE   // riders stamp start / end like the tiles (slot 7: 1 = index plan, 2 = gather)
    if (threadIdx.x == 0 && blockIdx.x < 512) { g_gemm_tl_wg[blockIdx.x][0] = wall_clock64(); g_gemm_tl_wg[blockIdx.x][7] = extra < R.n_plan ? 1 : 2; }
#endif
    if (extra < R.n_plan) {
        if (threadIdx.x >= MT_THREADS) return;   // ended waves take no part in the barriers of the draw
        // the sequential draw is the longest single job of this launch at batch 1024 (as long as the tiles): let its waves
        // issue ahead of the tile workgroup that shares the CU
        __builtin_amdgcn_s_setprio(3);
        mt_her_plan(R.rng, R.meta->current_size, R.T, R.plan_batch, 1, R.future_p, R.next_plan,
                    reinterpret_cast<uint32_t(*)[MT_N]>(lds), reinterpret_cast<int *>(&bsum[0][0]));
    } else {
        s8r4::s8_gather_ahead(R.ahead, R.aXT, R.aXA, R.aXP, R.ldx, R.act_off, R.act_dim, R.max_action, extra - R.n_plan,
                              R.n_ahead);
    }
#ifdef SLAB_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 512) g_gemm_tl_wg[blockIdx.x][5] = wall_clock64();
#endif
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_ride(const GemmGroup grp, const RideArgs R, int tiles) {
    gemm_ride_body<false>(grp, nullptr, R, tiles);
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_ride(unsigned long long t03, unsigned long long t47, int tiles,
                                                                   const GemmGroup grp, const AdamFuse F, const RideArgs R) {
    const TileHead TH{t03, t47};   // (leading scalars: preloaded with the wave, gemm_lds.h)
    const unsigned sink = kernarg_prefetch<24 + sizeof(GemmGroup) + sizeof(AdamFuse) + sizeof(RideArgs)>();
    gemm_ride_body<true>(grp, &F, R, tiles, nullptr, &TH);
    kernarg_prefetch_keep(sink);
}

__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_ride_u(const GemmGroup grp, const RideArgs R, int tiles) {
    gemm_ride_body<false, true>(grp, nullptr, R, tiles);
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_ride_u(unsigned long long t03, unsigned long long t47, int tiles,
                                                                     const GemmGroup grp, const AdamFuse F, const RideArgs R) {
    const TileHead TH{t03, t47};
    const unsigned sink = kernarg_prefetch<24 + sizeof(GemmGroup) + sizeof(AdamFuse) + sizeof(RideArgs)>();
    gemm_ride_body<true, true>(grp, &F, R, tiles, nullptr, &TH);
    kernarg_prefetch_keep(sink);
}

// data-parallel ranks, tile-wise one-shot exchange (gemm_lds.h PEER): weight gradients + rank exchange + optimizer step in ONE
// launch, the riders behind the tiles as above.  Replaces k_gemm_lds -> k_peer_adam (its kernel boundary, its second pass over
// the gradients: +6 us per update at world size 1), and lets early tiles exchange while later ones still multiply.
// row0: first flag row of this launch's tiles (0; the actor's tile launch behind a split launch: the critic's tile count)
// t03 / t47: the tile -> problem table as leading scalars (preloaded with the wave, gemm_lds.h TileHead), as in k_gemm_lds_adam
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_peer(unsigned long long t03, unsigned long long t47, const GemmGroup grp,
                                                                   const AdamFuse F, const RideArgs R, int tiles, const PeerDev D, int u,
                                                                   int mean, int row0) {
    const TileHead TH{t03, t47};
    const PeerTile PT{&D, u, mean, row0};
    gemm_ride_body<true, false, true>(grp, &F, R, tiles, &PT, &TH);
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_peer_u(unsigned long long t03, unsigned long long t47, const GemmGroup grp,
                                                                     const AdamFuse F, const RideArgs R, int tiles, const PeerDev D, int u,
                                                                     int mean, int row0) {
    const TileHead TH{t03, t47};
    const PeerTile PT{&D, u, mean, row0};
    gemm_ride_body<true, true, true>(grp, &F, R, tiles, &PT, &TH);
}

// Large minibatches: 64 x 64 tiles with the batch rows split over workgroups (dw64.h), same riders behind the tiles
#include "dw64.h"
template <bool ADAM>
__device__ __forceinline__ void dw64_ride_body(const GemmGroup &grp, const AdamFuse *F, const RideArgs &R, const Dw64Args &X) {
    __shared__ __attribute__((aligned(16))) float lds[DW_LDS_FLOATS];
    __shared__ int aux[256];
    if ((int)blockIdx.x < X.n_wg) {
        dw64_tile<ADAM>(grp, F, X, (int)blockIdx.x, lds, aux);
        return;
    }
    const int extra = (int)blockIdx.x - X.n_wg;
    if (extra < R.n_plan) {
        if (threadIdx.x >= MT_THREADS) return;
        __builtin_amdgcn_s_setprio(3);
        mt_her_plan(R.rng, R.meta->current_size, R.T, R.plan_batch, 1, R.future_p, R.next_plan,
                    reinterpret_cast<uint32_t(*)[MT_N]>(lds), aux);
    } else {
        s8r4::s8_gather_ahead(R.ahead, R.aXT, R.aXA, R.aXP, R.ldx, R.act_off, R.act_dim, R.max_action, extra - R.n_plan,
                              R.n_ahead);
    }
}
__global__ __launch_bounds__(DW_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_dw64(const GemmGroup grp, const RideArgs R, const Dw64Args X) {
    dw64_ride_body<false>(grp, nullptr, R, X);
}
__global__ __launch_bounds__(DW_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_dw64_adam(const GemmGroup grp, const AdamFuse F, const RideArgs R, const Dw64Args X) {
    dw64_ride_body<true>(grp, &F, R, X);
}


// canonical arena -> fragment-ordered copies of the slab engines (hp_agent_set_params, sync_targets)
__global__ void k_relayout(const float *__restrict__ canon, float *fragF, float *fragD, int n, const ArenaMap am) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int of, od;
    frag_offsets_any(am, idx, of, od);
    const float v = canon[idx];
    if (of >= 0) fragF[of] = v;
    if (od >= 0 && fragD) fragD[od] = v;
}

__global__ __launch_bounds__(256) void k_gather_fused(const double *__restrict__ obs, const double *__restrict__ ag,
                                                      const double *__restrict__ g, const double *__restrict__ act,
                                                      const PlanRec *__restrict__ plan, int batch, int T, int obs_dim,
                                                      int goal_dim, int act_dim, double sq_threshold,
                                                      const NormDev *__restrict__ onz, const NormDev *__restrict__ gnz,
                                                      double clip_obs, double clip_range, float max_action, int ldx,
                                                      int act_off, float *XA, float *XP, float *XT, float *R) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= batch) return;
    const PlanRec rec = plan[i];
    const long long e = rec.e;
    const int t = rec.t;
    const double *obs_row = obs + (e * (T + 1) + t) * obs_dim;
    const double *ag_next = ag + (e * (T + 1) + t + 1) * goal_dim;
    const double *g_src = rec.her ? ag + (e * (T + 1) + rec.fut) * goal_dim : g + (e * T + t) * goal_dim;
    const double *act_row = act + (e * T + t) * act_dim;
    float *xa = XA + (long long)i * ldx, *xp = XP + (long long)i * ldx, *xt = XT + (long long)i * ldx;
    for (int c = lane; c < 2 * obs_dim; c += 64) {
        const int col = (c < obs_dim) ? c : c - obs_dim;
        double v = fmin(fmax(obs_row[c], -clip_obs), clip_obs);                       // _preproc_og
        v = __ddiv_rn(__dsub_rn(v, (double)onz->mean[col]), onz->std[col]);           // normalize
        const float x = (float)fmin(fmax(v, -clip_range), clip_range);
        if (c < obs_dim) {
            xa[col] = x;
            xp[col] = x;
        } else {
            xt[col] = x;
        }
    }
    for (int c = lane; c < goal_dim; c += 64) {
        double v = fmin(fmax(g_src[c], -clip_obs), clip_obs);
        v = __ddiv_rn(__dsub_rn(v, (double)gnz->mean[c]), gnz->std[c]);
        const float x = (float)fmin(fmax(v, -clip_range), clip_range);
        xa[obs_dim + c] = x;
        xp[obs_dim + c] = x;
        xt[obs_dim + c] = x;   // g_next := g (ddpg_agent.py:231)
    }
    for (int c = lane; c < act_dim; c += 64) xa[act_off + c] = (float)act_row[c] / max_action;  // models.py:38
    if (lane == 0) {
        double s = 0.0;
        for (int c = 0; c < goal_dim; ++c) {
            const double d = __dsub_rn(ag_next[c], g_src[c]);
            const double sq = __dmul_rn(d, d);
            s = (c == 0) ? sq : __dadd_rn(s, sq);
        }
        R[i] = hp_reward(s, sq_threshold);
    }
}

// slab engine: Adam that also refreshes the fragment-ordered copies and finishes the loss log
__global__ __launch_bounds__(256) void k_adam_frag(const AdamFuse F, const float *__restrict__ g, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    split_reset_by_block0(F);
    if (idx >= n) return;
    adam_apply(F, idx, g[idx]);
}

// 4 consecutive arena elements per thread (n % 4 == 0)
__global__ __launch_bounds__(256) void k_adam_frag4(const AdamFuse F, const float *__restrict__ g, int n4) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    split_reset_by_block0(F);
    if (t >= n4) return;
    bool quad;
    const int idx0 = adam_quad_remap(F.am, t, quad);   // the 256-wide layers dealt in 4-row x 64-column blocks (agent_device.h)
    const float4 g4 = *reinterpret_cast<const float4 *>(g + idx0);
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
    if (quad) adam_step4_quad(F, idx0, gv);
    else adam_apply4(F, idx0, gv);
}

__global__ void k_polyak_frag(float *__restrict__ tgt, const float *__restrict__ src, float *fragFT, int n,
                              float one_minus, float polyak, const ArenaMap am) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float t = __fadd_rn(__fmul_rn(one_minus, src[idx]), __fmul_rn(polyak, tgt[idx]));
    tgt[idx] = t;
    int of, od;
    frag_offsets_any(am, idx, of, od);
    if (of >= 0) fragFT[of] = t;
}

// actor forward for rollouts: x [rows, xdim] -> padded input rows
__global__ void k_pack_rows(const float *__restrict__ src, int rows, int width, float *dst, int ld, int col0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * width) return;
    const int r = idx / width, c = idx - r * width;
    dst[(long long)r * ld + col0 + c] = src[idx];
}

// rollout inputs (ddpg_agent._preproc_inputs :163-171): normalised, clipped observation | goal rows in float32, the same
// float64 arithmetic as the sampled minibatch rows (slab8.h s8_gather)
__global__ void k_policy_inputs(const double *__restrict__ obs, const double *__restrict__ g, int rows, int od, int gd,
                                const NormDev *__restrict__ onz, const NormDev *__restrict__ gnz, double clip_obs,
                                double clip_o, double clip_g, float *X, int ld) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = od + gd;
    if (idx >= (long long)rows * w) return;
    const int r = (int)(idx / w), c = (int)(idx - (long long)r * w);
    double v;
    if (c < od) {
        v = fmin(fmax(obs[(long long)r * od + c], -clip_obs), clip_obs);
        v = __ddiv_rn(__dsub_rn(v, (double)onz->mean[c]), onz->std[c]);
        v = fmin(fmax(v, -clip_o), clip_o);
    } else {
        const int j = c - od;
        v = fmin(fmax(g[(long long)r * gd + j], -clip_obs), clip_obs);
        v = __ddiv_rn(__dsub_rn(v, (double)gnz->mean[j]), gnz->std[j]);
        v = fmin(fmax(v, -clip_g), clip_g);
    }
    X[(long long)r * ld + c] = (float)v;
}

// critic input: action block = actions / max_action (models.py:38)
__global__ void k_pack_scaled_actions(const float *__restrict__ src, int rows, int act_dim, float *dst, int ld, int act_off,
                                      float max_action) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * act_dim) return;
    const int r = idx / act_dim, c = idx - r * act_dim;
    dst[(long long)r * ld + act_off + c] = src[idx] / max_action;
}

__global__ void k_unpack_actions(const float *__restrict__ X, int rows, int ld, int act_off, int act_dim,
                                 float max_action, float *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * act_dim) return;
    const int r = idx / act_dim, c = idx - r * act_dim;
    out[idx] = X[(long long)r * ld + act_off + c] * max_action;   // stored value is actions / max_action
}

// ------------------------------------------------------------------------------- host side
int launch_group(hp_agent *a, const Launch &L, int which) {
    ProfScope ps(a, which);
    HP_KLOG(L.g.uni ? "k_gemm_lds_u" : "k_gemm_lds");
    hipLaunchKernelGGL(L.g.uni ? k_gemm_lds_u : k_gemm_lds, dim3(L.tiles), dim3(GL_THREADS), 0, a->ctx->stream, L.g);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int enqueue_gather(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, const PlanRec *plan, double sq, int xset,
                   hipStream_t stream) {
    ProfScope ps(a, PROF_SAMPLE);
    HP_KLOG("k_gather_fused");
    hipLaunchKernelGGL(k_gather_fused, dim3((a->B + 3) / 4), dim3(256), 0, stream ? stream : a->ctx->stream, b->d_obs, b->d_ag,
                       b->d_g, b->d_act, plan, a->B, (int)b->T, (int)b->obs_dim, (int)b->goal_dim, (int)b->act_dim, sq, on->d,
                       gn->d, a->cfg.clip_obs, a->cfg.clip_range, (float)a->cfg.max_action, a->ldx, a->act_off,
                       xset ? a->XA2 : a->XA, xset ? a->XP2 : a->XP, xset ? a->XT2 : a->XT, xset ? a->R2 : a->R);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static AdamFuse adam_fuse(hp_agent *a);
static void fold_polyak(hp_agent *a, AdamFuse &F) {   // the optimizer launch also steps the targets (GatherCtx::polyak_after)
    F.tgt = a->targets; F.fragFT = a->fragFT;
    F.polyak = (float)a->cfg.polyak; F.one_minus = (float)(1.0 - a->cfg.polyak);
}

static ArenaMap arena_map(const hp_agent *a) {
    ArenaMap am;
    am.la = a->la;
    am.lc = a->lc;
    am.H = a->H;
    am.mode = a->slab32 ? 2 : 1;
    return am;
}

int enqueue_relayout(hp_agent *a, bool targets) {
    const int n = a->n_arena;
    hipLaunchKernelGGL(k_relayout, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream,
                       targets ? a->targets : a->params, targets ? a->fragFT : a->fragF,
                       targets ? (float *)nullptr : a->fragD, n, arena_map(a));
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// all weight gradients of one update: the only products that reduce over the batch (input sets sXA / sXP)
Launch build_dw_group(const hp_agent *a, const float *sXA, const float *sXP, float *grads) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    if (!grads) grads = a->grads;
    float *Ga = grads, *Gc = grads + la.total;
    Launch L;
    // the four 256 x 256 problems first: Launch::place_on_xcds gives each of them one pair of XCDs
    add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp, 3);
    add_dw(L, a->dA2, H, H, a->CA.h1, H, H, Gc + lc.w2, Gc + lc.b2, Mp, 2);
    add_dw(L, a->dK3, H, H, a->AP.h2, H, H, Ga + la.w3, Ga + la.b3, Mp, 3);
    add_dw(L, a->dK2, H, H, a->AP.h1, H, H, Ga + la.w2, Ga + la.b2, Mp, 2);
    // the narrow problems (heads, first layers: 40 tiles at the reference shapes) follow as second workgroups on CUs that
    // already hold a 256 x 256 tile; with a long reduction their batch rows are split over ks workgroups each, so that no CU
    // carries two full tiles (gemm_lds.h)
    const int ks = (!a->dw64 && a->gl_part && a->dw_ksplit > 1 && Mp >= GL_RING_MIN_K) ? a->dw_ksplit : 1;   // ring path only
    add_dw(L, a->dQA, 16, 16, a->CA.h3, H, H, Gc + lc.w4, Gc + lc.b4, Mp, 4);
    if (ks > 1) L.split_last(ks);
    add_dw(L, a->dA1, H, H, sXA, ldx, lc.K1, Gc + lc.w1, Gc + lc.b1, Mp, 1);
    if (ks > 1) L.split_last(ks);
    add_dw(L, a->dZ, 16, 16, a->AP.h3, H, H, Ga + la.w4, Ga + la.b4, Mp, 4);
    if (ks > 1) L.split_last(ks);
    add_dw(L, a->dK1, H, H, sXP, ldx, la.K1, Ga + la.w1, Ga + la.b1, Mp, 1);
    if (ks > 1) L.split_last(ks);
    L.g.part = a->gl_part;
    L.g.ticket = a->gl_ticket;
    L.g.uni = (Mp > 256 && Mp <= 640) ? 1 : 0;   // gemm_lds.h, note at `wave`
    L.place_on_xcds();
    return L;
}

// tile table + exchange buffers of the large-minibatch weight-gradient launch (dw64.h)
static int dw64_args(hp_agent *a, const Launch &L, Dw64Args &X) {
    memset(&X, 0, sizeof(X));
    X.S = a->dw_S;
    int K = 0, tiles = 0;
    for (int i = 0; i < L.g.n; ++i) {
        const GemmProb &p = L.g.p[i];
        HP_REQUIRE(p.a_si == 1 && p.b_sj == 1 && p.M % 8 == 0 && p.N % 8 == 0 && p.K % DW_KH == 0, HP_ERR_INVALID,
                   "dw64: operand layout");
        X.tile0[i] = tiles;
        X.tiles_n[i] = (p.N + 63) / 64;
        tiles += ((p.M + 63) / 64) * X.tiles_n[i];
        K = p.K > K ? p.K : K;
    }
    X.kslice = ((K + X.S - 1) / X.S + DW_KH - 1) / DW_KH * DW_KH;
    X.n_wg = X.S * tiles;
    // allocated by hp_agent_create (this runs under stream capture)
    HP_REQUIRE(a->dw_part.bytes >= (size_t)tiles * X.S * DW_PART * sizeof(float) && a->dw_ticket.bytes >= sizeof(unsigned long long) * (size_t)tiles,
               HP_ERR_INVALID, "dw64: exchange buffers too small");
    X.part = a->dw_part.as<float>();
    X.ticket = a->dw_ticket.as<unsigned long long>();
    return HP_OK;
}

// ---- split launch (slab8_split.h) ------------------------------------------------------------------------------------------
static int split_warmers(const hp_agent *a) {   // L2 warmers per XCD that still fit beside three kinds of chains + the spare workgroups
    const int per_xcd = a->ctx->cu_count / 8;
    const int nslab = a->Mp / a->s8_rows;
    const int chains = 3 * ((nslab + 7) / 8) + 1;   // worst XCD: its share of every kind of chain + a plan / gather workgroup
    return per_xcd - chains >= 2 ? 2 : (per_xcd - chains >= 1 ? 1 : 0);
}
// three kinds of chains of `rows`-row slabs + a plan / gather workgroup per XCD on this device's CUs?
bool split_fits_rows(const hp_agent *a, int rows) {
    if (a->ctx->cu_count % 8 != 0 || rows != 4) return false;   // (the kernel is compiled for 4-row slabs only)
    const int per_xcd = a->ctx->cu_count / 8, nslab = a->Mp / rows;
    return 3 * ((nslab + 7) / 8) + 1 <= per_xcd;
}
// Data-parallel ranks take the split launch too (round 6): with the tile-wise peer exchange the critic's in-launch tiles exchange
// and step by themselves (SPLIT_TILES_PEER), with any other transport -- RCCL, the two-phase or gated peer forms, a caller that
// exchanges itself -- they leave the gradients to the exchange + optimizer launches that follow (SPLIT_TILES_GRADS).
// Not when ranks SHARE a device (rehearsals on a 1-GPU box; hp_peer_set_gate is how the library is told): the in-launch waits of
// one rank's tiles for its own chains assume that the launch is resident as a whole, and eight ranks' launches on one device are
// not (measured: the bounded hand-off of an 8-rank rehearsal gave up, fault word 0x11) -- those keep the two-launch form.
bool split_fits(const hp_agent *a) {
    if (!a->slab8 || a->dw64 || !a->fuse_adam_ok) return false;
    if (a->peer && a->peer->gate) return false;
    if (a->Mp < GL_RING_MIN_K || a->dw_ksplit > 1) return false;   // the in-launch tiles take the ring path, unsplit
    return split_fits_rows(a, a->s8_rows);
}

// Who runs where.  Workgroup b of the launch lands on XCD b % 8, and within an XCD in index order: chains first, then the spare
// workgroups, the weight-gradient tiles last (they wait for chains).  The actor-side chains run ALONE on XCDs 0-3 -- the launch's
// critical path keeps an L2 to itself, 16 streams per XCD like in k_fb_slab8 -- and every short chain on XCDs 4-7, whose 32
// streams per XCD saturate their L2s (the short chains end 1.5 us later: they have 10 us of slack): 38.3 us/update at batch 256
// against 39.8 with every kind of chain spread evenly over the XCDs and 39.2 with the critic chains on both halves (round 4,
// profiles/r04_ab_split_place_b256.txt; those placements are gone).  Shapes whose chains do not divide that way are spread evenly.
static unsigned build_split_roles(const hp_agent *a, FbSplitArgs &Q, bool chains_ac, bool chains_t, int n_plan, int n_ahead,
                                  int n_tiles) {
    const int nslab = a->Mp / a->s8_rows, per_xcd = a->ctx->cu_count / 8;
    int n[8][SR_N];
    memset(n, 0, sizeof(n));
    auto spread = [&](int role, int count, int x0, int nx) {   // evenly over XCDs x0 .. x0 + nx - 1, remainder to the first ones
        for (int i = 0; i < nx; ++i) n[x0 + i][role] += count / nx + (i < count % nx ? 1 : 0);
    };
    const bool half = nslab % 8 == 0 && chains_ac && 3 * (nslab / 8) + 2 <= per_xcd && 2 * (nslab / 4) <= per_xcd;
    if (half) {   // the actor-side chains alone on XCDs 0-3 (as in k_fb_slab8), every short chain on XCDs 4-7
        spread(SR_A, nslab, 0, 4);
        spread(SR_C, nslab, 4, 4);
        if (chains_t) spread(SR_T, nslab, 4, 4);
    } else {
        if (chains_ac) { spread(SR_A, nslab, 0, 8); spread(SR_C, nslab, 0, 8); }
        if (chains_t) spread(SR_T, nslab, 0, 8);
    }
    for (int i = 0; i < n_plan; ++i) n[i % (half ? 4 : 8)][SR_PLAN] += 1;
    for (int i = 0; i < n_ahead; ++i) n[(n_plan + i) % (half ? 4 : 8)][SR_AHEAD] += 1;
    const int warm = split_warmers(a);
    Q.warm_side = 0u;
    for (int x = 0; x < 8; ++x) {
        int used = 0;
        for (int r = 0; r < SR_WARM; ++r) used += n[x][r];
        const int w = (per_xcd - used) < warm ? (per_xcd - used > 0 ? per_xcd - used : 0) : warm;
        n[x][SR_WARM] = (!chains_ac && !chains_t) ? 0 : w;
        // what this XCD's chains stream: 1 = the actor side's sets, 0 = the critic side's (targets + critic), 2 = all
        const unsigned side = half ? (x < 4 ? 1u : 0u) : (chains_ac ? 2u : 0u);
        Q.warm_side |= side << (4 * x);
    }
    // tiles: dealt to the XCDs in proportion to the CUs their short chains (C, T) and idle CUs leave them; the kernel numbers
    // them slot-major across the XCDs (slab8_split.h)
    if (n_tiles > 0) {
        int room[8], total = 0;
        for (int x = 0; x < 8; ++x) {
            room[x] = per_xcd - n[x][SR_A] - n[x][SR_PLAN] - n[x][SR_AHEAD] - n[x][SR_WARM];
            if (room[x] < 0) room[x] = 0;
            total += room[x];
        }
        int given = 0;
        for (int x = 0; x < 8; ++x) {
            n[x][SR_TILE] = total ? n_tiles * room[x] / total : 0;
            given += n[x][SR_TILE];
        }
        while (given < n_tiles) {   // the remainder: one at a time to the XCD with the most room left (never to one without, while any has)
            int best = 0;
            for (int x = 1; x < 8; ++x)
                if (room[x] - n[x][SR_TILE] > room[best] - n[best][SR_TILE]) best = x;
            n[best][SR_TILE] += 1;
            ++given;
        }
    }
    unsigned rows = 0;
    for (int x = 0; x < 8; ++x) {
        unsigned long long packed = 0ull;
        unsigned tot = 0;
        for (int r = 0; r < SR_N; ++r) {
            packed |= (unsigned long long)(n[x][r] & 0xff) << (8 * r);
            tot += (unsigned)n[x][r];
        }
        Q.nrole[x] = packed;
        rows = tot > rows ? tot : rows;
    }
    return 8u * rows;   // grid: workgroup b = 8 * slot + XCD; slots past an XCD's list exit at once
}

// the weight-gradient problems of one network (critic: inside the split launch; actor: the launch behind it)
static Launch build_dw_half(const hp_agent *a, bool critic, const float *sX, float *grads) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    if (!grads) grads = a->grads;
    float *Ga = grads, *Gc = grads + la.total;
    Launch L;
    if (critic) {
        // in the order the critic chains publish the operands (slab8_split.h: stages 0, 0, 1, 2)
        add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp, 3);
        add_dw(L, a->dQA, 16, 16, a->CA.h3, H, H, Gc + lc.w4, Gc + lc.b4, Mp, 4);
        add_dw(L, a->dA2, H, H, a->CA.h1, H, H, Gc + lc.w2, Gc + lc.b2, Mp, 2);
        add_dw(L, a->dA1, H, H, sX, ldx, lc.K1, Gc + lc.w1, Gc + lc.b1, Mp, 1);
    } else {
        add_dw(L, a->dK3, H, H, a->AP.h2, H, H, Ga + la.w3, Ga + la.b3, Mp, 3);
        add_dw(L, a->dK2, H, H, a->AP.h1, H, H, Ga + la.w2, Ga + la.b2, Mp, 2);
        add_dw(L, a->dZ, 16, 16, a->AP.h3, H, H, Ga + la.w4, Ga + la.b4, Mp, 4);
        add_dw(L, a->dK1, H, H, sX, ldx, la.K1, Ga + la.w1, Ga + la.b1, Mp, 1);
        // two 256 x 256 problems: four XCDs each (gemm_tile: placed2)
        if (L.g.p[0].tiles_n == 8 && L.g.p[0].M == 256 && L.g.p[1].tile0 == 64 && L.g.p[1].tiles_n == 8 && L.g.p[1].M == 256)
            L.g.xcd = 2;
    }
    L.g.uni = (Mp > 256 && Mp <= 640) ? 1 : 0;
    return L;
}

// slab engines: forwards + losses + backwards of one update (inputs in XA/XP/XT/R): chain kernel + weight-gradient launch
// only = 1 / 2: just the chain kernel / just the weight-gradient launch (timing diagnostics, hp_agent_debug_chain)
// argument blocks of the chain kernels for one update (gc as in enqueue_forward_backward_slab)
struct FbBuilt {
    FbSlabArgs P;
    int nslab, xs;
    float *sXA, *sXP, *sXT, *sR;
    bool ride_dw, ride;
};
static void build_fb_args(hp_agent *a, const GatherCtx *gc, FbBuilt &O) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    const int nslab = Mp / (a->slab8 ? a->s8_rows : S32_ROWS);
    FbSlabArgs &P = O.P;
    const int xs = gc ? gc->xset : 0;
    float *sXA = xs ? a->XA2 : a->XA, *sXP = xs ? a->XP2 : a->XP, *sXT = xs ? a->XT2 : a->XT, *sR = xs ? a->R2 : a->R;
    const SlabNetPtrs online = SlabNetPtrs{a->fragF, a->fragD, a->params};
    {
        FwdSlabArgs &A = P.f;
        A.tl = a->timeline;
        memset(&A.gs, 0, sizeof(A.gs));
        A.gs.plan_any = a->plan.as<PlanRec>();
        A.gs.B = a->B;
        if (gc) {
            hp_buffer *b = gc->b;
            A.gs.obs = b->d_obs; A.gs.ag = b->d_ag; A.gs.g = b->d_g; A.gs.act = b->d_act;
            A.gs.plan = gc->pregathered ? nullptr : gc->plan; A.gs.plan_any = gc->plan;
            A.gs.onz = gc->on->d; A.gs.gnz = gc->gn->d;
            A.gs.sq_threshold = gc->sq; A.gs.clip_obs = a->cfg.clip_obs; A.gs.clip_range = a->cfg.clip_range;
            A.gs.T = b->T; A.gs.obs_dim = b->obs_dim; A.gs.goal_dim = b->goal_dim; A.gs.B = a->B;
            A.gs.R = sR;
        }
        A.online = online;
        A.target = SlabNetPtrs{a->fragFT, nullptr, a->targets};
        A.la = la; A.lc = lc; A.H = H; A.ldx = ldx; A.act_off = a->act_off; A.act_dim = a->cfg.act_dim; A.Mp = Mp;
        A.max_action = (float)a->cfg.max_action;
        A.XA = sXA; A.XT = sXT; A.XP = sXP; A.TP = a->TP;
        A.CAh1 = a->CA.h1; A.CAh2 = a->CA.h2; A.CAh3 = a->CA.h3;
        A.APh1 = a->AP.h1; A.APh2 = a->AP.h2; A.APh3 = a->AP.h3;
        A.CPh1 = a->CP.h1; A.CPh2 = a->CP.h2; A.CPh3 = a->CP.h3;
        A.QT = a->QT; A.QA = a->QA; A.QP = a->QP;
    }
    const bool ride_dw = gc && gc->ride_in_dw;
    const bool ride = gc && gc->next_plan && gc->rng && !ride_dw;
    {
        BwdSlabArgs &A = P.b;
        A.tl = a->timeline + 96;
        A.online = online;
        A.la = la; A.lc = lc; A.H = H; A.ldx = ldx; A.act_off = a->act_off; A.act_dim = a->cfg.act_dim;
        A.B = a->B; A.Mp = Mp;
        A.max_action = (float)a->cfg.max_action; A.gamma = (float)a->cfg.gamma;
        A.clip_ret = (float)(1.0 / (1.0 - a->cfg.gamma)); A.action_l2 = (float)a->cfg.action_l2;
        A.QT = a->QT; A.QA = a->QA; A.QP = a->QP; A.R = sR; A.XP = sXP; A.TP = a->TP;
        A.CAh1 = a->CA.h1; A.CAh2 = a->CA.h2; A.CAh3 = a->CA.h3;
        A.APh1 = a->AP.h1; A.APh2 = a->AP.h2; A.APh3 = a->AP.h3;
        A.CPh1 = a->CP.h1; A.CPh2 = a->CP.h2; A.CPh3 = a->CP.h3;
        A.dQA = a->dQA; A.dA3 = a->dA3; A.dA2 = a->dA2; A.dA1 = a->dA1;
        A.dZ = a->dZ; A.dK3 = a->dK3; A.dK2 = a->dK2; A.dK1 = a->dK1;
        A.part = a->part; A.st = a->d_state; A.adam = adam_cfg(a);
        A.nslab = nslab;
        A.rng = ride ? gc->rng->d_state : nullptr;
        A.meta = ride ? gc->b->d_meta : nullptr;
        A.next_plan = ride ? gc->next_plan : nullptr;
        A.future_p = ride ? gc->future_p : 0.0;
        A.T = ride ? gc->b->T : 0;
        A.plan_batch = a->B;
    }
    O.nslab = nslab; O.xs = xs; O.sXA = sXA; O.sXP = sXP; O.sXT = sXT; O.sR = sR; O.ride_dw = ride_dw; O.ride = ride;
}

// bias gradients + their optimizer step in workgroups of their own behind the tiles (gemm_lds.h: gemm_bias_tile; the ring path's
// summation order only; inside the tn == 0 tiles otherwise, and in the launches that exchange tile-wise between ranks: same bits)
static bool sep_bias_on(const hp_agent *a) { return a->Mp >= GL_RING_MIN_K && !a->dw64; }

static int enqueue_split_update(hp_agent *a, const GatherCtx *gc, FbBuilt &built, bool fuse_adam, int only);
int enqueue_forward_backward_slab(hp_agent *a, const GatherCtx *gc, bool fuse_adam, int only) {
    const int ldx = a->ldx;
    hipStream_t s = a->ctx->stream;
    FbBuilt built;
    build_fb_args(a, gc, built);
    FbSlabArgs &P = built.P;
    const int nslab = built.nslab, xs = built.xs;
    float *sXA = built.sXA, *sXP = built.sXP;
    const bool ride_dw = built.ride_dw, ride = built.ride;
    if (gc && gc->split) return enqueue_split_update(a, gc, built, fuse_adam, only);
    if (only == 2) {
    } else if (a->slab8) {
        // one launch: each workgroup carries its rows through forward AND backward (k_fb_slab8)
        ProfScope ps(a, PROF_GEMM_FWD);
        P.n_plan = ride ? 1 : 0;
        P.n_ahead = 0;
        // chains split across XCD halves: measured (us/update, split vs not) 42.0 vs 43.5 at batch 128, 44.0 vs 45.2 at 256,
        // 46.6 vs 46.6 at 384, 48.0 vs 47.8 at 448, 77.3 vs 74.6 at 1024 -- it pays while the chains leave half of the CUs free
        const int n_chain = chain_wgs(a);
        P.xcd_split = (nslab % 4 == 0) && 4 * nslab <= a->ctx->cu_count;
        P.ahead = P.f.gs;
        P.aXT = P.aXA = P.aXP = nullptr;
        if (gc && gc->ahead_plan && !ride_dw) {   // next update's inputs into the other set
            P.n_ahead = S8_AHEAD_WGS;
            P.ahead.plan = gc->ahead_plan;
            P.ahead.plan_any = gc->ahead_plan;
            P.ahead.R = xs ? a->R : a->R2;
            P.aXT = xs ? a->XT : a->XT2; P.aXA = xs ? a->XA : a->XA2; P.aXP = xs ? a->XP : a->XP2;
        }
        // L2 warmers: spare workgroups (the same number on every XCD) while the launch still fits the CUs.  Measured (us/update, with
        // 8 vs without): 39.9 vs 42.4 at batch 128, 42.1 vs 44.4 at 256, 46.4 vs 46.6 at 384, 52.8 vs 54.2 at 512, 56.6 vs 57.1 at 768.
        // Two per XCD instead of one (round 3, after the entry reordering): 39.65 vs 39.82 at batch 256, 41.3 vs 43.0 at 384; 24 / 32: no
        // further gain (40.0 / 40.0 vs 39.95 with 16).
        {
            const int fit = a->ctx->cu_count - (n_chain + P.n_plan + P.n_ahead);
            const int want = fit >= 16 ? 16 : (fit >= 8 ? 8 : 0);
            P.n_pref = want;
        }
        const unsigned grid = n_chain + P.n_plan + P.n_ahead + P.n_pref;
        HP_KLOG("k_fb_slab8");
        if (a->s8_rows == 4)
            hipLaunchKernelGGL(s8r4::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        else if (a->s8_rows == 8)
            hipLaunchKernelGGL(s8r8::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        else
            hipLaunchKernelGGL(s8r16::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        HP_CHECK_HIP(hipGetLastError());
    } else {
        // 32-row slabs, forward + backward of a chain in one workgroup; inputs come gathered (enqueue_updates)
        ProfScope ps(a, PROF_GEMM_FWD);
        P.n_plan = ride ? 1 : 0;
        P.n_ahead = P.n_pref = P.xcd_split = 0;
        HP_KLOG("k_fb_slab32");
        hipLaunchKernelGGL(s32::k_fb_slab32, dim3(2 * nslab + P.n_plan), dim3(S32_THREADS), 0, s, P);
        HP_CHECK_HIP(hipGetLastError());
    }
    if (only != 1) {   // all weight gradients (+ the optimizer when no gradient exchange follows) as their own launch
        Launch L = build_dw_group(a, sXA, sXP, gc ? gc->grads_out : nullptr);
        const bool riders = ride_dw && ((gc->next_plan && gc->rng) || gc->ahead_plan);
        RideArgs R;
        memset(&R, 0, sizeof(R));
        if (riders) {
            if (gc->next_plan && gc->rng) {
                R.n_plan = 1;
                R.rng = gc->rng->d_state; R.meta = gc->b->d_meta; R.next_plan = gc->next_plan; R.future_p = gc->future_p;
                R.T = gc->b->T; R.plan_batch = a->B;
            }
            if (gc->ahead_plan) {
                // one pass of 32 rows (8 waves x 4 rows in flight) per gather workgroup: each pass is two dependent HBM
                // latencies, so fewer, longer workgroups made this launch 3 us longer than its tiles (61.0 vs 58.6 us/update at
                // batch 1024 with 8 vs 32 of them)
                R.n_ahead = (a->B + 31) / 32 < 64 ? (a->B + 31) / 32 : 64;
                R.ahead = P.f.gs;
                R.ahead.plan = gc->ahead_plan; R.ahead.plan_any = gc->ahead_plan;
                R.ahead.R = xs ? a->R : a->R2;
                R.aXT = xs ? a->XT : a->XT2; R.aXA = xs ? a->XA : a->XA2; R.aXP = xs ? a->XP : a->XP2;
                R.ldx = ldx; R.act_off = a->act_off; R.act_dim = a->cfg.act_dim; R.max_action = (float)a->cfg.max_action;
            }
        }
        if (a->dw64) {
            // large minibatch: 64 x 64 tiles, batch rows split over workgroups (dw64.h); the riders follow the tiles
            ProfScope ps(a, PROF_DW);
            Dw64Args X;
            HP_TRY(dw64_args(a, L, X));
            const unsigned grid = X.n_wg + R.n_plan + R.n_ahead;
            if (fuse_adam) {
                AdamFuse F = adam_fuse(a);
                F.keep_grads = (gc == nullptr || a->keep_grads_dbg) ? 1 : 0;
                if (gc && gc->polyak_after) fold_polyak(a, F);
                HP_KLOG("k_dw64_adam");
                hipLaunchKernelGGL(k_dw64_adam, dim3(grid), dim3(DW_THREADS), 0, s, L.g, F, R, X);
            } else {
                HP_KLOG("k_dw64");
                hipLaunchKernelGGL(k_dw64, dim3(grid), dim3(DW_THREADS), 0, s, L.g, R, X);
            }
            HP_CHECK_HIP(hipGetLastError());
        } else if (fuse_adam && gc && gc->peer_u >= 0) {
            // data-parallel ranks: the tiles exchange by themselves (gemm_lds.h PEER); riders, if any, behind them
            ProfScope ps(a, PROF_DW);
            HP_REQUIRE(a->peer && L.tiles <= HP_PEER_TILES, HP_ERR_STATE, "tile-wise exchange: %d tiles exceed the flag rows", L.tiles);
            const unsigned grid = L.tiles + R.n_plan + R.n_ahead;
            AdamFuse F = adam_fuse(a);
            F.keep_grads = a->keep_grads_dbg ? 1 : 0;
            if (gc->polyak_after) fold_polyak(a, F);
            HP_KLOG(L.g.uni ? "k_gemm_lds_adam_peer_u" : "k_gemm_lds_adam_peer");
            hipLaunchKernelGGL(L.g.uni ? k_gemm_lds_adam_peer_u : k_gemm_lds_adam_peer, dim3(grid), dim3(GL_THREADS), 0, s, L.head(0), L.head(1), L.g, F, R,
                               L.tiles, a->peer->dev, gc->peer_u, a->grad_mean ? 1 : 0, 0);
            HP_CHECK_HIP(hipGetLastError());
        } else if (riders) {
            ProfScope ps(a, PROF_DW);
            const int front = L.tiles + (sep_bias_on(a) ? L.separate_bias() : 0);   // tiles, then the bias panels, then the riders
            const unsigned grid = front + R.n_plan + R.n_ahead;
            if (fuse_adam) {
                AdamFuse F = adam_fuse(a);
                F.keep_grads = a->keep_grads_dbg ? 1 : 0;
                if (gc->polyak_after) fold_polyak(a, F);
                L.g.loss_wg = a->loss_wg;
                HP_KLOG(L.g.uni ? "k_gemm_lds_adam_ride_u" : "k_gemm_lds_adam_ride");
                hipLaunchKernelGGL(L.g.uni ? k_gemm_lds_adam_ride_u : k_gemm_lds_adam_ride, dim3(grid + L.g.loss_wg), dim3(GL_THREADS), 0, s, L.head(0), L.head(1),
                                   front, L.g, F, R);
            } else {
                HP_KLOG(L.g.uni ? "k_gemm_lds_ride_u" : "k_gemm_lds_ride");
                hipLaunchKernelGGL(L.g.uni ? k_gemm_lds_ride_u : k_gemm_lds_ride, dim3(grid), dim3(GL_THREADS), 0, s, L.g, R, front);
            }
            HP_CHECK_HIP(hipGetLastError());
        } else if (fuse_adam) {
            ProfScope ps(a, PROF_DW);
            // inside a sampled update loop nobody reads the gradient vector (hp_agent_get_grads documents this): 1.17 MB of
            // the ~6.7 MB this kernel leaves dirty in L2 for the end-of-kernel write-back
            AdamFuse F = adam_fuse(a);
            F.keep_grads = (gc == nullptr || a->keep_grads_dbg) ? 1 : 0;
            if (gc && gc->polyak_after) fold_polyak(a, F);
            const int front = L.tiles + (sep_bias_on(a) ? L.separate_bias() : 0);
            L.g.loss_wg = a->loss_wg;
            HP_KLOG(L.g.uni ? "k_gemm_lds_adam_u" : "k_gemm_lds_adam");
            hipLaunchKernelGGL(L.g.uni ? k_gemm_lds_adam_u : k_gemm_lds_adam, dim3(front + L.g.loss_wg), dim3(GL_THREADS), 0, s, L.head(0), L.head(1), L.g, F);
            HP_CHECK_HIP(hipGetLastError());
        } else {
            HP_TRY(launch_group(a, L, PROF_DW));
        }
    }
    return HP_OK;
}

// ---- one update in the split form: k_fb_split8 (chains + the critic's tiles and optimizer step [+ the actor's]), then -- in
// the two-launch form -- the actor's tiles
// k_fb_split8 with its role table as leading scalar arguments: word r = role r, byte x = its workgroups on XCD x
static void launch_split(unsigned grid, hipStream_t s, const FbSplitArgs &Q, int tiles_mode = SPLIT_TILES_ADAM) {
    unsigned long long w[SR_N];
    for (int r = 0; r < SR_N; ++r) {
        w[r] = 0ull;
        for (int x = 0; x < 8; ++x) w[r] |= ((Q.nrole[x] >> (8 * r)) & 0xffull) << (8 * x);
    }
    HP_KLOG(tiles_mode == SPLIT_TILES_PEER ? "k_fb_split8<1>" : (tiles_mode == SPLIT_TILES_GRADS ? "k_fb_split8<2>" : "k_fb_split8<0>"));
    if (tiles_mode == SPLIT_TILES_PEER)
        hipLaunchKernelGGL(s8r4::k_fb_split8<SPLIT_TILES_PEER>, dim3(grid), dim3(S8_THREADS), 0, s, w[0], w[1], w[2], w[3], w[4], w[5], w[6], Q);
    else if (tiles_mode == SPLIT_TILES_GRADS)
        hipLaunchKernelGGL(s8r4::k_fb_split8<SPLIT_TILES_GRADS>, dim3(grid), dim3(S8_THREADS), 0, s, w[0], w[1], w[2], w[3], w[4], w[5], w[6], Q);
    else
        hipLaunchKernelGGL(s8r4::k_fb_split8<SPLIT_TILES_ADAM>, dim3(grid), dim3(S8_THREADS), 0, s, w[0], w[1], w[2], w[3], w[4], w[5], w[6], Q);
}
static void split_common(hp_agent *a, FbSplitArgs &Q, int set) {
    Q.sync = a->k1_sync + (set & 1) * SPLIT_SET_WORDS;
    Q.sync_other = a->k1_sync + ((set + 1) & 1) * SPLIT_SET_WORDS;
    Q.fault = a->k1_sync + SPLIT_FAULT;
    Q.fault_host = a->fault_host_dev;
    Q.wait_ticks = 50000000ull;   // 0.5 s of the 100 MHz wall clock: three orders of magnitude beyond a launch
}

static int enqueue_split_update(hp_agent *a, const GatherCtx *gc, FbBuilt &built, bool fuse_adam, int only) {
    HP_REQUIRE(only == 0 && split_fits(a), HP_ERR_STATE, "split launch: not available for this engine / call");
    // what the in-launch tiles do behind their products (slab8_split_args.h): the optimizer step (single rank), the tile-wise rank
    // exchange + the step (data-parallel ranks on a device each, gemm_lds.h PEER), or nothing -- the caller exchanges the gradient
    // vector and steps in launches of its own (RCCL; two-phase / gated peer memory; utils.sync_grads from a host loop)
    const int mode = !fuse_adam ? SPLIT_TILES_GRADS : (gc->peer_u >= 0 ? SPLIT_TILES_PEER : SPLIT_TILES_ADAM);
    hipStream_t s = a->ctx->stream;
    FbSlabArgs &P = built.P;
    const int xs = built.xs, nslab = built.nslab;
    static FbSplitArgs Qz;   // zero template (the struct has padding the kernel never reads)
    FbSplitArgs Q = Qz;
    P.n_plan = built.ride ? 1 : 0;
    P.n_ahead = 0;
    P.xcd_split = 0;
    P.n_pref = 0;
    P.ahead = P.f.gs;
    P.aXT = P.aXA = P.aXP = nullptr;
    if (gc->ahead_plan) {   // next update's inputs into the other set (the target side gathers its own rows)
        P.n_ahead = S8_AHEAD_WGS;
        P.ahead.plan = gc->ahead_plan;
        P.ahead.plan_any = gc->ahead_plan;
        P.ahead.R = xs ? a->R : a->R2;
        P.aXT = xs ? a->XT : a->XT2; P.aXA = xs ? a->XA : a->XA2; P.aXP = xs ? a->XP : a->XP2;
    }
    Q.tgs = P.f.gs;
    if (gc->t_plan) {
        Q.tgs.plan = gc->t_plan;
        Q.tgs.plan_any = gc->t_plan;
    }
    Q.tgs.R = nullptr;
    Q.qt_in = gc->qset ? a->QT2 : a->QT;
    Q.qt_out = gc->qset ? a->QT : a->QT2;
    split_common(a, Q, gc->qset);
    // tile problems in the order their operands are published: the critic's W3, W4 (stage 0), W2 (stage 1), W1 (stage 2).  Gates
    // of the optimizer steps: W3c after the actor-side chains' first critic dX layer (counter 4), W4c / W1c after their critic
    // forward (3), W2c after the second dX layer (5).  The actor's tiles are the launch behind this one (the form that held them
    // in this launch too, behind an "actor-side chains done" counter, measured 45.7 vs 38.0 us/update in round 4 and is gone).
    Launch L = build_dw_half(a, true, built.sXA, mode == SPLIT_TILES_GRADS ? gc->grads_out : nullptr);
    Launch La = build_dw_half(a, false, built.sXP, mode == SPLIT_TILES_GRADS ? gc->grads_out : nullptr);
    Q.tile_stage = 0u | (0u << 4) | (1u << 8) | (2u << 12);
    const unsigned gate_sel = 4u | (3u << 4) | (5u << 8) | (3u << 12);
    Q.tiles = L.g;
    Q.need_c = (unsigned)nslab;
    // time-line builds: the last launch with target chains AND a plan workgroup stamps
    Q.tl_mark = (gc->t_plan != nullptr && P.n_plan > 0) ? 1 : 0;
    AdamFuse F = adam_fuse(a);
    F.keep_grads = a->keep_grads_dbg ? 1 : 0;
    if (gc->polyak_after) fold_polyak(a, F);
    F.gate = Q.sync;
    F.gate_need = (unsigned)nslab;
    F.gate_sel = gate_sel;
    F.fault = Q.fault;
    F.fault_host = a->fault_host_dev;
    F.gate_ticks = Q.wait_ticks;
    F.tl_mark = Q.tl_mark;
    Q.adam = F;
    Q.s = P;
    if (mode == SPLIT_TILES_PEER) {
        HP_REQUIRE(a->peer && a->peer->d_dev && L.tiles + La.tiles <= HP_PEER_TILES, HP_ERR_STATE,
                   "tile-wise exchange in the split launch: %d + %d tiles exceed the flag rows", L.tiles, La.tiles);
        Q.peer = a->peer->d_dev;
        Q.peer_u = gc->peer_u;
        Q.peer_mean = a->grad_mean ? 1 : 0;
    }
    const unsigned grid = build_split_roles(a, Q, true, gc->t_plan != nullptr, P.n_plan, P.n_ahead, L.tiles);
    {
        ProfScope ps(a, PROF_GEMM_FWD);
        launch_split(grid, s, Q, mode);
        HP_CHECK_HIP(hipGetLastError());
    }
    {   // the actor's weight gradients (+ optimizer step): 144 tiles at the reference shapes, one per CU
        ProfScope ps(a, PROF_DW);
        if (mode == SPLIT_TILES_GRADS) {
            // gradients only (k_gemm_lds); the optimizer launch behind the exchange clears this launch's counter set (enqueue_adam /
            // enqueue_peer_adam take it from split_reset_pending)
            HP_KLOG(La.g.uni ? "k_gemm_lds_u" : "k_gemm_lds");
            hipLaunchKernelGGL(La.g.uni ? k_gemm_lds_u : k_gemm_lds, dim3(La.tiles), dim3(GL_THREADS), 0, s, La.g);
            HP_CHECK_HIP(hipGetLastError());
            a->split_reset_pending = Q.sync;
            return HP_OK;
        }
        AdamFuse Fa = adam_fuse(a);
        Fa.keep_grads = a->keep_grads_dbg ? 1 : 0;
        if (gc->polyak_after) fold_polyak(a, Fa);
        Fa.reset_sync = Q.sync;   // every split launch then starts from a clean set whatever the parity of the sequence before it
        if (mode == SPLIT_TILES_PEER) {
            // the same tiles exchanging by themselves (gemm_lds.h PEER), flag rows behind the critic's; tile 0 writes the loss log
            // and clears the counter set (the bias gradients stay inside the tn == 0 tiles, as in every tile-wise launch)
            RideArgs R;
            memset(&R, 0, sizeof(R));
            HP_KLOG(La.g.uni ? "k_gemm_lds_adam_peer_u" : "k_gemm_lds_adam_peer");
            hipLaunchKernelGGL(La.g.uni ? k_gemm_lds_adam_peer_u : k_gemm_lds_adam_peer, dim3(La.tiles), dim3(GL_THREADS), 0, s, La.head(0), La.head(1), La.g, Fa, R,
                               La.tiles, a->peer->dev, gc->peer_u, a->grad_mean ? 1 : 0, L.tiles);
        } else {
            const int front = La.tiles + (sep_bias_on(a) ? La.separate_bias() : 0);
            La.g.loss_wg = a->loss_wg;
            HP_KLOG(La.g.uni ? "k_gemm_lds_adam_u" : "k_gemm_lds_adam");
            hipLaunchKernelGGL(La.g.uni ? k_gemm_lds_adam_u : k_gemm_lds_adam, dim3(front + La.g.loss_wg), dim3(GL_THREADS), 0, s, La.head(0), La.head(1), La.g, Fa);
        }
        HP_CHECK_HIP(hipGetLastError());
    }
    return HP_OK;
}

// target chains of the sequence's first update (its plan: gc->plan) -> Q' set 0; also clears the first update's counter set
int enqueue_split_prologue(hp_agent *a, const GatherCtx *gc, int tiles_mode) {
    HP_REQUIRE(gc && split_fits(a), HP_ERR_STATE, "split launch: not available for this engine");
    FbBuilt built;
    GatherCtx g0 = *gc;
    g0.pregathered = false;
    g0.next_plan = nullptr;
    g0.rng = nullptr;
    build_fb_args(a, &g0, built);
    FbSlabArgs &P = built.P;
    static FbSplitArgs Qz;
    FbSplitArgs Q = Qz;
    P.n_plan = P.n_ahead = P.xcd_split = P.n_pref = 0;
    P.ahead = P.f.gs;
    P.aXT = P.aXA = P.aXP = nullptr;
    Q.tgs = P.f.gs;
    Q.tgs.plan = gc->plan;
    Q.tgs.plan_any = gc->plan;
    Q.tgs.R = nullptr;
    Q.qt_in = a->QT2;
    Q.qt_out = a->QT;
    split_common(a, Q, 1);        // sync_other = set 0, the first update's
    Q.need_c = 0u;
    Q.reset_sync = 1;
    Q.adam = adam_fuse(a);
    Q.s = P;
    const unsigned grid = build_split_roles(a, Q, false, true, 0, 0, 0);
    ProfScope ps(a, PROF_PLAN);   // (once per sequence, with the index draws: not an update's launch)
    launch_split(grid, a->ctx->stream, Q, tiles_mode);   // (no tiles here: the instantiation the sequence's updates run, so that a rank executes ONE k_fb_split8)
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static AdamFuse adam_fuse(hp_agent *a) {
    AdamFuse F;
    F.p = a->params; F.p_out = a->params; F.m = a->adam_m; F.v = a->adam_v; F.fragF = a->fragF; F.fragD = a->fragD;
    F.grads_base = a->grads; F.st = a->d_state; F.scal = &a->d_state->neg_step_actor; F.am = arena_map(a); F.n_actor = a->la.total;
    F.keep_grads = 1;
    F.tgt = nullptr; F.fragFT = nullptr; F.polyak = 0.f; F.one_minus = 0.f;
    F.gate = nullptr; F.gate_need = 0u; F.gate_sel = 0u; F.fault = nullptr; F.fault_host = nullptr; F.gate_ticks = 0ull; F.reset_sync = nullptr; F.tl_mark = 0;
    F.w = (float)(1.0 - a->cfg.adam_beta1); F.b2 = (float)a->cfg.adam_beta2;
    F.omb2 = (float)(1.0 - a->cfg.adam_beta2); F.eps = (float)a->cfg.adam_eps;
    F.part = a->part; F.nslab = a->Mp / (a->slab8 ? a->s8_rows : S32_ROWS); F.B = a->B;
    F.act_dim = a->cfg.act_dim;
    F.action_l2 = (float)a->cfg.action_l2; F.loss_log = a->loss_log;
    F.wt = a->Mp <= 768 ? 1 : 0;   // us/update without / with: 40.9 / 40.3 at 256, 44.9 / 44.4 at 384, 46.9 / 46.1 at 512 k8, 53.1 / 52.9 at 768, 55.6 / 55.8 at 1024
    return F;
}

int enqueue_adam(hp_agent *a, bool polyak_after) {
    ProfScope ps(a, PROF_ADAM);
    const int n = a->n_arena;
    if (a->slab) {
        AdamFuse F = adam_fuse(a);
        if (polyak_after) fold_polyak(a, F);
        F.reset_sync = a->split_reset_pending;   // behind a split launch whose tiles wrote gradients only (enqueue_split_update)
        a->split_reset_pending = nullptr;
        // plain stores in the stand-alone optimizer kernels: write-through (adam_fuse: small minibatches) pays inside a tile launch,
        // where other workgroups still multiply while the stepped state drains; a kernel that does nothing else only waits for its
        // own acknowledgements (forced data-parallel world 1, us/update: RCCL form 44.7 -> 43.5, separate peer exchange 45.3 -> 44.9)
        F.wt = 0;
        const bool by4 = n % 4 == 0 && a->la.total % 4 == 0;
        HP_KLOG(by4 ? "k_adam_frag4" : "k_adam_frag");
        if (by4)
            hipLaunchKernelGGL(k_adam_frag4, dim3((n / 4 + 255) / 256), dim3(256), 0, a->ctx->stream, F, a->grads, n / 4);
        else
            hipLaunchKernelGGL(k_adam_frag, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, F, a->grads, n);
        HP_CHECK_HIP(hipGetLastError());
        return HP_OK;
    }
    return layers_enqueue_adam(a);
}

int enqueue_polyak(hp_agent *a) {
    ProfScope ps(a, PROF_ADAM);
    const int n = a->n_arena;
    const double om = 1.0 - a->cfg.polyak;
    if (a->slab) {
        HP_KLOG("k_polyak_frag");
        hipLaunchKernelGGL(k_polyak_frag, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->targets, a->params,
                           a->fragFT, n, (float)om, (float)a->cfg.polyak, arena_map(a));
        HP_CHECK_HIP(hipGetLastError());
        return HP_OK;
    }
    return layers_enqueue_polyak(a);
}

// one update's forwards + backwards.  gc == nullptr: the minibatch is already staged in XA/XP/XT/R.
// fuse_adam: the optimizer step follows immediately on this rank (no gradient exchange): the slab engines then apply
// it in the weight-gradient GEMM's epilogue and the caller must NOT enqueue Adam again (returns that via *fused).
int enqueue_forward_backward(hp_agent *a, const GatherCtx *gc, bool fuse_adam, bool *fused) {
    // Adam in the weight-gradient GEMM's epilogue (k_gemm_lds_adam).  The first version (one element at a time: four
    // serialised cold round trips per thread for p/m/v) measured 92.9 vs 67.5 us per update and was parked; with four
    // elements per thread (one float4 load per state array, float4 store into the forward fragment copy) it is the
    // faster path, 57.2 vs 60.5 us, and the default.  RLARM_FUSE_ADAM=0 keeps the separate k_adam_frag4 launch for A/B.
    fuse_adam = fuse_adam && a->fuse_adam_ok;
    if (fused) *fused = a->slab && fuse_adam;
    if (a->slab) return enqueue_forward_backward_slab(a, gc, fuse_adam);   // gather fused into the forward kernel
    if (gc) HP_TRY(enqueue_gather(a, gc->b, gc->on, gc->gn, gc->plan, gc->sq));
    return enqueue_forward_backward_layers(a);
}

int enqueue_peer_adam(hp_agent *a, int u, bool polyak_after) {
    AdamFuse F = adam_fuse(a);
    if (polyak_after) fold_polyak(a, F);
    F.grads_base = a->grads;
    F.keep_grads = a->keep_grads_dbg ? 1 : 0;   // RLARM_KEEP_GRADS=1: hp_agent_get_grads then returns the exchanged sum
    F.reset_sync = a->split_reset_pending;
    a->split_reset_pending = nullptr;
    F.wt = 0;   // (plain stores in the stand-alone optimizer kernels: enqueue_adam)
    ProfScope ps(a, PROF_ADAM);
    return peer_enqueue_adam(a->peer, F, a->n_arena, u, a->grad_mean);
}

// actor rows on the device.  Scratch layout: [head_bytes of caller data] | X rows | h1 | h2 | h3 | tanh | actions; `fill`
// enqueues whatever turns the caller data into X (zeroed beforehand).
template <typename Fill>
static int actor_rows(hp_agent *a, int32_t net, int64_t rows, size_t head_bytes, float *actions_host, Fill fill) {
    const int H = a->H, ldx = a->ldx, ad = a->cfg.act_dim;
    const int Mp = roundup((int)rows, 32);
    hipStream_t s = a->ctx->stream;
    const size_t nX = (size_t)Mp * ldx, nH = (size_t)Mp * H, nT = (size_t)Mp * 16;
    head_bytes = (head_bytes + 15) & ~(size_t)15;
    HP_TRY(a->fwd_ws.ensure(head_bytes + (nX + 3 * nH + nT + (size_t)rows * ad) * 4));
    char *head = a->fwd_ws.as<char>();
    float *X = reinterpret_cast<float *>(head + head_bytes), *h1 = X + nX, *h2 = h1 + nH, *h3 = h2 + nH, *tp = h3 + nH,
          *outp = tp + nT;
    HP_CHECK_HIP(hipMemsetAsync(X, 0, nX * 4, s));
    HP_TRY(fill(head, X, s));
    const NetLayout &l = a->la;
    const float *P = (net == HP_NET_ACTOR) ? a->params : a->targets;
    { Launch L; add_fwd(L, X, ldx, l.K1, P + l.w1, P + l.b1, h1, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    { Launch L; add_fwd(L, h1, H, H, P + l.w2, P + l.b2, h2, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    { Launch L; add_fwd(L, h2, H, H, P + l.w3, P + l.b3, h3, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    {
        Launch L;
        add_fwd(L, h3, H, H, P + l.w4, P + l.b4, X + a->act_off, ldx, Mp, 16, EPI_BIAS_TANH);
        L.g.p[0].n_store = ad; L.g.p[0].C2 = tp; L.g.p[0].ldc2 = 16; L.g.p[0].max_action = (float)a->cfg.max_action;
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    // actions = max_action * tanh(.)  (models.py:24); tp holds tanh
    hipLaunchKernelGGL(k_unpack_actions, dim3((unsigned)((rows * ad + 255) / 256)), dim3(256), 0, s, tp, (int)rows, 16, 0, ad,
                       (float)a->cfg.max_action, outp);
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(actions_host, outp, (size_t)rows * ad * 4, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

// stand-alone critic rows (models.py:28-44): Q(x, a) for host inputs, on the layer-per-launch GEMMs
static int critic_rows(hp_agent *a, int32_t net, int64_t rows, const float *x_host, const float *act_host, float *q_host) {
    const int H = a->H, ldx = a->ldx, xd = a->xdim, ad = a->cfg.act_dim;
    const int Mp = roundup((int)rows, 32);
    hipStream_t s = a->ctx->stream;
    const size_t n_x = (size_t)rows * xd, n_a = (size_t)rows * ad, nX = (size_t)Mp * ldx, nH = (size_t)Mp * H,
                 nT = (size_t)Mp * 16;
    const size_t head = ((n_x + n_a) * 4 + 15) & ~(size_t)15;
    HP_TRY(a->fwd_ws.ensure(head + (nX + 3 * nH + nT + (size_t)rows) * 4));
    char *base = a->fwd_ws.as<char>();
    float *raw_x = reinterpret_cast<float *>(base), *raw_a = raw_x + n_x;
    float *X = reinterpret_cast<float *>(base + head), *h1 = X + nX, *h2 = h1 + nH, *h3 = h2 + nH, *q16 = h3 + nH,
          *outp = q16 + nT;
    HP_CHECK_HIP(hipMemsetAsync(X, 0, nX * 4, s));
    HP_CHECK_HIP(hipMemcpyAsync(raw_x, x_host, n_x * 4, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(raw_a, act_host, n_a * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((n_x + 255) / 256)), dim3(256), 0, s, raw_x, (int)rows, xd, X, ldx, 0);
    hipLaunchKernelGGL(k_pack_scaled_actions, dim3((unsigned)((n_a + 255) / 256)), dim3(256), 0, s, raw_a, (int)rows, ad, X,
                       ldx, a->act_off, (float)a->cfg.max_action);
    HP_CHECK_HIP(hipGetLastError());
    const NetLayout &l = a->lc;
    const float *P = ((net == HP_NET_CRITIC) ? a->params : a->targets) + a->la.total;
    { Launch L; add_fwd(L, X, ldx, l.K1, P + l.w1, P + l.b1, h1, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    { Launch L; add_fwd(L, h1, H, H, P + l.w2, P + l.b2, h2, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    { Launch L; add_fwd(L, h2, H, H, P + l.w3, P + l.b3, h3, H, Mp, H, EPI_BIAS_RELU); HP_TRY(launch_group(a, L, PROF_GEMM_FWD)); }
    {
        Launch L;
        add_fwd(L, h3, H, H, P + l.w4, P + l.b4, q16, 16, Mp, 16, EPI_BIAS);
        L.g.p[0].n_store = 1;
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    hipLaunchKernelGGL(k_unpack_actions, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, q16, (int)rows, 16, 0, 1, 1.0f, outp);
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(q_host, outp, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

// slab engines: the whole policy call is one launch (k_policy_slab8).  `head` = float32 inputs (x != null) or the float64
// observation rows followed by the goal rows.
static int policy_rows_slab(hp_agent *a, hp_norm *on, hp_norm *gn, int32_t net, int64_t rows, const void *host_a,
                            size_t bytes_a, const void *host_b, size_t bytes_b, bool f32_inputs, double clip_obs,
                            float *actions_host) {
    hipStream_t s = a->ctx->stream;
    const int ad = a->cfg.act_dim;
    const size_t head = (bytes_a + bytes_b + 15) & ~(size_t)15;
    HP_TRY(a->fwd_ws.ensure(head + (size_t)rows * ad * 4));
    char *d = a->fwd_ws.as<char>();
    float *d_act = reinterpret_cast<float *>(d + head);
    HP_CHECK_HIP(hipMemcpyAsync(d, host_a, bytes_a, hipMemcpyHostToDevice, s));
    if (bytes_b) HP_CHECK_HIP(hipMemcpyAsync(d + bytes_a, host_b, bytes_b, hipMemcpyHostToDevice, s));
    PolicyArgs P;
    memset(&P, 0, sizeof(P));
    if (f32_inputs) {
        P.x = reinterpret_cast<const float *>(d);
        P.od = a->xdim; P.gd = 0;
    } else {
        P.obs = reinterpret_cast<const double *>(d);
        P.g = reinterpret_cast<const double *>(d + bytes_a);
        P.od = on->size; P.gd = gn->size;
        P.onz = on->d; P.gnz = gn->d;
        P.clip_obs = clip_obs; P.clip_o = on->clip; P.clip_g = gn->clip;
    }
    P.rows = (int)rows;
    P.net = (net == HP_NET_ACTOR) ? SlabNetPtrs{a->fragF, a->fragD, a->params} : SlabNetPtrs{a->fragFT, nullptr, a->targets};
    P.la = a->la; P.H = a->H; P.act_dim = ad; P.max_action = (float)a->cfg.max_action;
    P.actions = d_act;
    hipLaunchKernelGGL(s8r4::k_policy_slab8, dim3((unsigned)((rows + 3) / 4)), dim3(S8_THREADS), 0, s, P);
    HP_CHECK_HIP(hipGetLastError());
    HP_CHECK_HIP(hipMemcpyAsync(actions_host, d_act, (size_t)rows * ad * 4, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

extern "C" {

int hp_agent_actor_forward(hp_agent *a, int32_t net, const float *x_host, int64_t rows, float *actions_host) {
    HP_REQUIRE(a && x_host && actions_host, HP_ERR_INVALID, "hp_agent_actor_forward: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(net == HP_NET_ACTOR || net == HP_NET_ACTOR_TARGET, HP_ERR_INVALID, "hp_agent_actor_forward: net must be an actor");
    HP_REQUIRE(rows > 0 && rows < (1 << 24), HP_ERR_INVALID, "hp_agent_actor_forward: rows out of range");
    const int xd = a->xdim, ldx = a->ldx;
    const size_t n_raw = (size_t)rows * xd;
    if (a->slab8) return policy_rows_slab(a, nullptr, nullptr, net, rows, x_host, n_raw * 4, nullptr, 0, true, 0.0, actions_host);
    return actor_rows(a, net, rows, n_raw * 4, actions_host, [&](char *head, float *X, hipStream_t s) -> int {
        float *raw = reinterpret_cast<float *>(head);
        HP_CHECK_HIP(hipMemcpyAsync(raw, x_host, n_raw * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((n_raw + 255) / 256)), dim3(256), 0, s, raw, (int)rows, xd, X, ldx, 0);
        HP_CHECK_HIP(hipGetLastError());
        return (int)HP_OK;
    });
}

int hp_agent_critic_forward(hp_agent *a, int32_t net, const float *x_host, const float *actions_host, int64_t rows,
                            float *q_host) {
    HP_REQUIRE(a && x_host && actions_host && q_host, HP_ERR_INVALID, "hp_agent_critic_forward: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(net == HP_NET_CRITIC || net == HP_NET_CRITIC_TARGET, HP_ERR_INVALID, "hp_agent_critic_forward: net must be a critic");
    HP_REQUIRE(rows > 0 && rows < (1 << 24), HP_ERR_INVALID, "hp_agent_critic_forward: rows out of range");
    return critic_rows(a, net, rows, x_host, actions_host, q_host);
}

int hp_agent_act(hp_agent *a, hp_norm *on, hp_norm *gn, int32_t net, const double *obs_host, const double *g_host,
                 int64_t rows, double clip_obs, float *actions_host) {
    HP_REQUIRE(a && on && gn && obs_host && g_host && actions_host, HP_ERR_INVALID, "hp_agent_act: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(on->ctx == a->ctx && gn->ctx == a->ctx, HP_ERR_INVALID, "hp_agent_act: handles belong to different contexts");
    HP_REQUIRE(net == HP_NET_ACTOR || net == HP_NET_ACTOR_TARGET, HP_ERR_INVALID, "hp_agent_act: net must be an actor");
    HP_REQUIRE(rows > 0 && rows < (1 << 24), HP_ERR_INVALID, "hp_agent_act: rows out of range");
    const int od = on->size, gd = gn->size;
    HP_REQUIRE(od + gd == a->xdim, HP_ERR_INVALID, "hp_agent_act: normalizer sizes %d+%d do not match the actor input %d", od,
               gd, a->xdim);
    const size_t nb_o = (size_t)rows * od * 8, nb_g = (size_t)rows * gd * 8;
    const double co = clip_obs > 0 ? clip_obs : INFINITY;
    if (a->slab8) return policy_rows_slab(a, on, gn, net, rows, obs_host, nb_o, g_host, nb_g, false, co, actions_host);
    return actor_rows(a, net, rows, nb_o + nb_g, actions_host, [&](char *head, float *X, hipStream_t s) -> int {
        double *d_obs = reinterpret_cast<double *>(head), *d_g = reinterpret_cast<double *>(head + nb_o);
        HP_CHECK_HIP(hipMemcpyAsync(d_obs, obs_host, nb_o, hipMemcpyHostToDevice, s));
        HP_CHECK_HIP(hipMemcpyAsync(d_g, g_host, nb_g, hipMemcpyHostToDevice, s));
        const long long n = (long long)rows * (od + gd);
        hipLaunchKernelGGL(k_policy_inputs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_obs, d_g, (int)rows, od, gd,
                           on->d, gn->d, co, on->clip, gn->clip, X, a->ldx);
        HP_CHECK_HIP(hipGetLastError());
        return (int)HP_OK;
    });
}

}  // extern "C"

extern "C" {

int hp_agent_policy_snapshot(hp_agent *a, hp_norm *on, hp_norm *gn) {
    HP_REQUIRE(a && on && gn, HP_ERR_INVALID, "hp_agent_policy_snapshot: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(a->slab8, HP_ERR_STATE, "hp_agent_policy_snapshot: needs the fused policy kernel (slab8 engine)");
    HP_REQUIRE(on->size + gn->size == a->xdim, HP_ERR_INVALID, "hp_agent_policy_snapshot: normalizer sizes do not match the actor");
    hipStream_t s = a->ctx->stream;
    if (!a->act_stream) {
        HP_CHECK_HIP(hipStreamCreateWithFlags(&a->act_stream, hipStreamNonBlocking));
        HP_CHECK_HIP(hipEventCreateWithFlags(&a->act_done, hipEventDisableTiming));
        for (auto &ps : a->snap) {
            HP_CHECK_HIP(hipMalloc((void **)&ps.params, sizeof(float) * a->la.total));
            HP_CHECK_HIP(hipMalloc((void **)&ps.fragF, sizeof(float) * a->la.total));
            HP_CHECK_HIP(hipMalloc((void **)&ps.on, sizeof(NormDev)));
            HP_CHECK_HIP(hipMalloc((void **)&ps.gn, sizeof(NormDev)));
            HP_CHECK_HIP(hipEventCreateWithFlags(&ps.ready, hipEventDisableTiming));
        }
    }
    const int target = (a->snap_cur == 0) ? 1 : 0;        // never the set policy calls are reading
    hp_agent::PolicySnap &ps = a->snap[target];
    if (a->act_recorded) HP_CHECK_HIP(hipStreamWaitEvent(s, a->act_done, 0));
    const size_t nb = sizeof(float) * a->la.total;
    HP_CHECK_HIP(hipMemcpyAsync(ps.params, a->params, nb, hipMemcpyDeviceToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(ps.fragF, a->fragF, nb, hipMemcpyDeviceToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(ps.on, on->d, sizeof(NormDev), hipMemcpyDeviceToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(ps.gn, gn->d, sizeof(NormDev), hipMemcpyDeviceToDevice, s));
    ps.clip_o = on->clip; ps.clip_g = gn->clip; ps.od = on->size; ps.gd = gn->size;
    HP_CHECK_HIP(hipEventRecord(ps.ready, s));
    a->snap_pending = target;
    return HP_OK;
}

int hp_agent_act_snapshot(hp_agent *a, const double *obs_host, const double *g_host, int64_t rows, double clip_obs,
                          float *actions_host) {
    HP_REQUIRE(a && obs_host && g_host && actions_host, HP_ERR_INVALID, "hp_agent_act_snapshot: null argument");
    HP_REQUIRE(rows > 0 && rows < (1 << 24), HP_ERR_INVALID, "hp_agent_act_snapshot: rows out of range");
    hipStream_t s = nullptr;
    {
        HP_SERIALISE(a);
        HP_REQUIRE(a->snap_cur >= 0 || a->snap_pending >= 0, HP_ERR_STATE, "hp_agent_act_snapshot: no snapshot taken yet");
        if (a->snap_pending >= 0) {
            hipEvent_t ev = a->snap[a->snap_pending].ready;
            hipError_t q = hipEventQuery(ev);
            if (q == hipErrorNotReady && a->snap_cur < 0) {   // the very first snapshot: nothing older to fall back to
                HP_CHECK_HIP(hipEventSynchronize(ev));
                q = hipSuccess;
            }
            (void)hipGetLastError();
            if (q == hipSuccess) {
                a->snap_cur = a->snap_pending;
                a->snap_pending = -1;
            }
        }
        const hp_agent::PolicySnap &ps = a->snap[a->snap_cur];
        s = a->act_stream;
        const int ad = a->cfg.act_dim;
        const size_t nb_o = (size_t)rows * ps.od * 8, nb_g = (size_t)rows * ps.gd * 8;
        const size_t head = (nb_o + nb_g + 15) & ~(size_t)15;
        HP_TRY(a->act_ws.ensure(head + (size_t)rows * ad * 4));
        char *d = a->act_ws.as<char>();
        float *d_act = reinterpret_cast<float *>(d + head);
        HP_CHECK_HIP(hipMemcpyAsync(d, obs_host, nb_o, hipMemcpyHostToDevice, s));
        HP_CHECK_HIP(hipMemcpyAsync(d + nb_o, g_host, nb_g, hipMemcpyHostToDevice, s));
        PolicyArgs P;
        memset(&P, 0, sizeof(P));
        P.obs = reinterpret_cast<const double *>(d);
        P.g = reinterpret_cast<const double *>(d + nb_o);
        P.od = ps.od; P.gd = ps.gd;
        P.onz = ps.on; P.gnz = ps.gn;
        P.clip_obs = clip_obs > 0 ? clip_obs : INFINITY; P.clip_o = ps.clip_o; P.clip_g = ps.clip_g;
        P.rows = (int)rows;
        P.net = SlabNetPtrs{ps.fragF, nullptr, ps.params};
        P.la = a->la; P.H = a->H; P.act_dim = ad; P.max_action = (float)a->cfg.max_action;
        P.actions = d_act;
        hipLaunchKernelGGL(s8r4::k_policy_slab8, dim3((unsigned)((rows + 3) / 4)), dim3(S8_THREADS), 0, s, P);
        HP_CHECK_HIP(hipGetLastError());
        HP_CHECK_HIP(hipMemcpyAsync(actions_host, d_act, (size_t)rows * ad * 4, hipMemcpyDeviceToHost, s));
        HP_CHECK_HIP(hipEventRecord(a->act_done, s));
        a->act_recorded = true;
    }
    HP_CHECK_HIP(hipStreamSynchronize(s));   // outside the context lock: the trainer keeps enqueueing meanwhile
    return HP_OK;
}

}  // extern "C"

extern "C" {

// diagnostic: time `n` back-to-back launches of ONE stage of the update as a captured hipGraph.
//   kind 6: optimizer kernel   8: polyak   10: forward + backward of the active engine (inputs as staged)
//   11: chain kernel only   12: weight-gradient launch (+ optimizer) only
int hp_agent_debug_chain(hp_agent *a, int32_t kind, int32_t n, double *us_per_launch) {
    HP_REQUIRE(a && us_per_launch && n > 0, HP_ERR_INVALID, "hp_agent_debug_chain: bad argument");
    HP_SERIALISE(a);
    hipStream_t s = a->ctx->stream;
    auto one = [&]() -> int {
        switch (kind) {
            case 6: return enqueue_adam(a);
            case 8: return enqueue_polyak(a);
            case 10: return enqueue_forward_backward(a);   // whole forward+backward of the active engine
            case 11: return a->slab ? enqueue_forward_backward_slab(a, nullptr, true, 1) : (int)HP_ERR_STATE;  // chain kernel(s) only
            case 12: return a->slab ? enqueue_forward_backward_slab(a, nullptr, true, 2) : (int)HP_ERR_STATE;  // weight gradients + Adam only
            default: hp_set_error("hp_agent_debug_chain: unknown kind %d", kind); return HP_ERR_INVALID;
        }
    };
    HP_CHECK_HIP(hipStreamSynchronize(s));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HP_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int st = HP_OK;
    for (int i = 0; i < n && st == HP_OK; ++i) st = one();
    hipError_t e = hipStreamEndCapture(s, &g);
    if (st != HP_OK) return st;
    HP_CHECK_HIP(e);
    HP_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HP_CHECK_HIP(hipGraphLaunch(ge, s));
    HP_CHECK_HIP(hipEventRecord(a->ev0, s));
    HP_CHECK_HIP(hipGraphLaunch(ge, s));
    HP_CHECK_HIP(hipEventRecord(a->ev1, s));
    HP_CHECK_HIP(hipEventSynchronize(a->ev1));
    float ms = 0.f;
    HP_CHECK_HIP(hipEventElapsedTime(&ms, a->ev0, a->ev1));
    *us_per_launch = 1e3 * ms / n;
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return HP_OK;
}

// diagnostic: stage-boundary time stamps (100 MHz ticks) written by a -DSLAB_TIMELINE build of the slab
// kernels: out[chain * 32 + k] for the forward kernel, out[96 + chain * 32 + k] for the backward kernel
int hp_agent_debug_timeline(hp_agent *a, uint64_t *out192) {
    HP_REQUIRE(a && out192, HP_ERR_INVALID, "hp_agent_debug_timeline: bad argument");
    HP_SERIALISE(a);
    HP_CHECK_HIP(hipMemcpyAsync(out192, a->timeline, 192 * 8, hipMemcpyDeviceToHost, a->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
#ifdef SLAB_TIMELINE   // weight-gradient GEMM stamps: first workgroup at [160..175], last at [176..191]
    HP_CHECK_HIP(hipMemcpyFromSymbol(out192 + 160, HIP_SYMBOL(g_gemm_tl), 32 * 8));
#endif
    return HP_OK;
}

#ifdef SLAB_TIMELINE
// time-line builds only (not part of the ABI): stage stamps of every workgroup of the last 32 x 32-tile launch
int hp_debug_gemm_wg_timeline(uint64_t *out4096) {
    HP_CHECK_HIP(hipDeviceSynchronize());
    HP_CHECK_HIP(hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_gemm_tl_wg), 512 * 8 * 8));
    return HP_OK;
}
// every workgroup of the last k_fb_split8 launch: out[5 * b + 0..3] = {start, hand-off point, gate reached (tiles), end}, [4] = role
int hp_debug_split_timeline(uint64_t *out5120) {
    HP_CHECK_HIP(hipDeviceSynchronize());
    static unsigned long long tl[1024][4], gate[1024][2];
    static int role[1024];
    HP_CHECK_HIP(hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_split_tl), sizeof(tl)));
    HP_CHECK_HIP(hipMemcpyFromSymbol(gate, HIP_SYMBOL(g_split_tl_gate), sizeof(gate)));
    HP_CHECK_HIP(hipMemcpyFromSymbol(role, HIP_SYMBOL(g_split_role), sizeof(role)));
    for (int b = 0; b < 1024; ++b) {
        out5120[5 * b + 0] = tl[b][0]; out5120[5 * b + 1] = tl[b][1]; out5120[5 * b + 2] = gate[b][0]; out5120[5 * b + 3] = tl[b][3];
        out5120[5 * b + 4] = (uint64_t)role[b] | (gate[b][1] << 8);   // role | gate passed << 8 (the stamps are < 2^56)
    }
    return HP_OK;
}
int hp_debug_split_entry(uint64_t *out1024) {   // wall clock at each workgroup's first instruction (before any kernel argument is read)
    HP_CHECK_HIP(hipDeviceSynchronize());
    HP_CHECK_HIP(hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_split_entry), 1024 * 8));
    return HP_OK;
}
int hp_debug_gemm_blk_timeline(uint64_t *out512) {   // [wave][block]{landed, issued} of one workgroup's product loop
    HP_CHECK_HIP(hipDeviceSynchronize());
    HP_CHECK_HIP(hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_gemm_tl_blk), 8 * 32 * 2 * 8));
    return HP_OK;
}
#endif

}  // extern "C"
