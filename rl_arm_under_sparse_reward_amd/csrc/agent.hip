// agent.hip -- the DDPG learner on gfx950: fused HER-sample/normalise kernel, grouped FP32-MFMA
// GEMMs for the five forward and three backward passes, loss kernel, fused Adam and polyak.
// Reference: models.py:11-44, ddpg_agent.py:214-277, torch.optim.Adam (ddpg_agent.py:42-43).
//
// ---- why it is shaped like this --------------------------------------------------------
// One update at batch 256 is 0.7 GFLOP over ~20 strictly dependent small matrix products:
// microseconds of FP32-MFMA time, so the cost is the number of dependent launches, not math.
//   * independent products of one dependency level go into ONE grouped launch
//     (actor_target | critic | actor forward layer k together; dX and dW of a backward layer
//     together), 19 launches per update;
//   * every product uses v_mfma_f32_16x16x4_f32 (exact fp32 == fmaf chain; there is no
//     TF32 on gfx950 and 1e-5 loss parity needs fp32).  A 256-thread workgroup owns one
//     32x32 output tile; its 4 wavefronts split the reduction dimension 4 ways (short
//     dependent MFMA chains = low latency), partial tiles are combined through LDS in a fixed
//     order (deterministic), and bias / ReLU / tanh / ReLU-mask run in the epilogue;
//   * all state (weights, targets, Adam moments, step counter, normalizer statistics,
//     RNG, buffer counters) is device resident and every kernel argument is constant across
//     updates, so a whole training cycle (store -> normalizer -> 40 updates -> polyak) is one
//     cached hipGraph launch.
//
// ---- HBM layout ------------------------------------------------------------------------
// Parameter "arena" (float32): [actor | critic], each  W1[H][K1] b1[H] W2[H][H] b2[H] W3[H][H]
// b3[H] W4[16][H] b4[16];  K1 = 32 for the actor, 48 for the critic, rows/cols beyond the real
// sizes are zero and stay zero (their gradients are exactly zero).  Gradients, Adam m, Adam v
// and the target networks use the same layout, so Adam and polyak are one elementwise pass.
// Network inputs are rows of 48 floats: [ x (obs+goal = 30) | 0 0 | a/max_action (4) | 0.. ]:
// the actor reads columns 0..31, the critic 0..47 (its W1 columns are permuted to match).
#include "internal.h"

#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------- structures
struct NetLayout {     // offsets in floats inside one net's arena segment
    int K1;            // padded input width (multiple of 16)
    int w1, b1, w2, b2, w3, b3, w4, b4, total;
};

enum { EPI_NONE = 0, EPI_BIAS_RELU = 1, EPI_BIAS = 2, EPI_BIAS_TANH = 3, EPI_MASK = 4 };

struct GemmProb {
    const float *A, *B;
    float *C;
    const float *bias;   // EPI_BIAS*
    const float *mask;   // EPI_MASK: gate on mask[m][n] > 0
    float *bias_grad;    // non-null: also emit column sums of the A operand (db) from tile column 0
    float *C2;           // EPI_BIAS_TANH: raw tanh output (needed by the backward pass)
    int a_si, a_sk;      // A element strides: output-row index / reduction index
    int b_sj, b_sk;      // B element strides: output-col index / reduction index
    int ldc, ldmask, ldc2;
    int M, N, K;         // output rows, output cols (multiples of 16), reduction length (multiple of 16)
    int n_store;         // only columns < n_store are written
    int epi;
    int tile0, tiles_n;  // first workgroup of this problem, tiles along N
    float max_action;    // EPI_BIAS_TANH
};

#define MAX_PROBS 8
struct GemmGroup {
    int n;
    int pipe;   // per-wave LDS-DMA rings for k-major operands with K > 256 (RLARM_GEMM_PIPE=0: workgroup-staged chunks, for A/B)
    int xcd;    // problems 0..3 have 8 x 8 tiles each and own one pair of XCDs (Launch::place_on_xcds)
    int pad_;
    GemmProb p[MAX_PROBS];
};

struct Pass {  // hidden activations of one forward pass
    float *h1, *h2, *h3;
};

struct AgentDevState {      // small device-resident scalars
    long long step;         // Adam step counter (both optimizers step together)
    long long n_logged;     // number of loss pairs written
    // per-step Adam scalars (torch computes them in Python doubles and narrows where used)
    float neg_step_actor, neg_step_critic, bc2_sqrt, pad;
};

struct AdamCfg {
    double lr_actor, lr_critic, beta1, beta2, eps;
};

// bias corrections of torch.optim.Adam for the step that is about to be applied
__device__ __forceinline__ void adam_prepare(AgentDevState *st, const AdamCfg c) {
    const double step = (double)st->step;
    const double bc1 = 1.0 - pow(c.beta1, step);
    const double bc2 = 1.0 - pow(c.beta2, step);
    st->neg_step_actor = (float)(-(c.lr_actor / bc1));
    st->neg_step_critic = (float)(-(c.lr_critic / bc1));
    st->bc2_sqrt = (float)sqrt(bc2);
}

#define LOSS_LOG 4096

#include "slab.h"
// Adam (torch.optim.Adam, _single_tensor_adam, no weight decay / amsgrad) on one arena element, plus the
// fragment-ordered copies of the slab engines.  Shared by k_adam_frag and the weight-gradient GEMM epilogue
// (single-rank runs fuse the optimizer into the GEMM; data-parallel runs all-reduce the gradients in between).
struct AdamFuse {
    const float *p;                   // parameters the step starts from (canonical arena)
    float *p_out;                     // ... and where the stepped parameters go: p itself, or the other set of the fused
                                      // single-launch update, whose chains still read p / its fragment copies while tiles finish
    float *m, *v, *fragF, *fragD;     // fragF / fragD: fragment-ordered copies of p_out
    const float *grads_base;          // arena origin of the gradient buffer the GEMM writes
    AgentDevState *st;
    const float *scal;                // {-lr_actor / bc1, -lr_critic / bc1, sqrt(bc2)} of the step being applied: the three
                                      // scalars in *st (written by the kernel before), or this update's row of the fused
                                      // path's per-sequence table (k_seq_begin)
    ArenaMap am;
    int n_actor;
    float w, b2, omb2, eps;
    const float *part;                // per-slab loss partials
    int nslab, B, act_dim;
    float action_l2;
    float *loss_log;
    int keep_grads;                   // also write the gradient out (the fused epilogue itself does not need it in memory)
};

__device__ __forceinline__ void adam_apply(const AdamFuse &F, int idx, float gi) {
    const float neg_step_size = F.scal[idx < F.n_actor ? 0 : 1];
    const float bc2_sqrt = F.scal[2];
    float mi = F.m[idx], vi = F.v[idx];
    mi = __fadd_rn(mi, __fmul_rn(F.w, __fsub_rn(gi, mi)));                      // exp_avg.lerp_(grad, 1 - beta1)
    vi = __fadd_rn(__fmul_rn(vi, F.b2), __fmul_rn(__fmul_rn(F.omb2, gi), gi));  // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float sq = __fsqrt_rn(vi);                             // correctly rounded float32 sqrt
    const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), F.eps);
    const float pn = __fadd_rn(F.p[idx], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    F.p_out[idx] = pn;
    F.m[idx] = mi;
    F.v[idx] = vi;
    int of, od;
    frag_offsets_any(F.am, idx, of, od);
    if (of >= 0) F.fragF[of] = pn;
    if (od >= 0) F.fragD[od] = pn;
}

// four consecutive arena elements at once (idx0 a multiple of 4): one vector load per state array, so the cold-cache
// latency of p / m / v is paid once, not once per element (scalar version: the store to p[idx] may alias the next
// element's load, which serialises them)
struct AdamState4 {   // optimizer state of 4 consecutive elements + the step scalars, fetched ahead of the gradient
    float4 p, m, v;
    float neg_step_size, bc2_sqrt;
};
__device__ __forceinline__ void adam_fetch4(AdamState4 &S, const AdamFuse &F, int idx0) {
    S.neg_step_size = F.scal[idx0 < F.n_actor ? 0 : 1];
    S.bc2_sqrt = F.scal[2];
    S.p = *reinterpret_cast<const float4 *>(F.p + idx0);
    S.m = *reinterpret_cast<const float4 *>(F.m + idx0);
    S.v = *reinterpret_cast<const float4 *>(F.v + idx0);
}
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4], const AdamState4 &S);
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4]) {
    AdamState4 S;
    adam_fetch4(S, F, idx0);
    adam_apply4(F, idx0, g, S);
}
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4], const AdamState4 &S) {
    const float neg_step_size = S.neg_step_size;
    const float bc2_sqrt = S.bc2_sqrt;
    const float4 p4 = S.p, m4 = S.m, v4 = S.v;
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mm[j] = __fadd_rn(mm[j], __fmul_rn(F.w, __fsub_rn(g[j], mm[j])));
        vv[j] = __fadd_rn(__fmul_rn(vv[j], F.b2), __fmul_rn(__fmul_rn(F.omb2, g[j]), g[j]));
        const float sq = __fsqrt_rn(vv[j]);
        const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), F.eps);
        pp[j] = __fadd_rn(pp[j], __fdiv_rn(__fmul_rn(neg_step_size, mm[j]), denom));
    }
    *reinterpret_cast<float4 *>(F.p_out + idx0) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4 *>(F.m + idx0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4 *>(F.v + idx0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (F.am.mode == 1 || F.am.mode == 2) {
        // slab8 / slab32 fragment orders: 4 consecutive reduction indices of one output row (idx0 % 4 == 0, every tensor's
        // row length is a multiple of 4) are ONE float4 of the forward copy and 4 dwords 16 B apart in the dX copy
        int of, od;
        if (F.am.mode == 1) frag8_offsets(F.am, idx0, of, od);
        else frag32_offsets(F.am, idx0, of, od);
        if (of >= 0) *reinterpret_cast<float4 *>(F.fragF + of) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (od >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) F.fragD[od + 4 * j] = pp[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int of, od;
            frag_offsets_any(F.am, idx0 + j, of, od);
            if (of >= 0) F.fragF[of] = pp[j];
            if (od >= 0) F.fragD[od] = pp[j];
        }
    }
}

// loss means from the per-slab partial sums: one wavefront, fixed reduction tree (deterministic)
__device__ __forceinline__ void loss_finalize(const AdamFuse &F) {
    const int lane = threadIdx.x;
    float tc = 0.f, tq = 0.f, tl = 0.f;
    for (int s = lane; s < F.nslab; s += 64) {   // agent-scope loads: in the fused kernel the chains of this launch wrote them
        tc += __hip_atomic_load(F.part + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tq += __hip_atomic_load(F.part + F.nslab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tl += __hip_atomic_load(F.part + 2 * F.nslab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int o = 32; o > 0; o >>= 1) {
        tc += __shfl_down(tc, o);
        tq += __shfl_down(tq, o);
        tl += __shfl_down(tl, o);
    }
    if (lane == 0) {
        const float invB = 1.0f / (float)F.B;
        const long long k = F.st->n_logged;
        F.loss_log[(k % LOSS_LOG) * 2 + 0] = -(tq * invB) + F.action_l2 * (tl / (float)(F.B * F.act_dim));
        F.loss_log[(k % LOSS_LOG) * 2 + 1] = tc * invB;
        F.st->n_logged = k + 1;
    }
}

#include "gemm_lds.h"
#include "peer.h"

// gradients: barrier + rank-ordered sum + Adam.  n4 = arena floats / 4; u = index of the update in its sequence.
__global__ __launch_bounds__(256) void k_peer_adam(const PeerDev D, const AdamFuse F, int n4, int u, int mean) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_g, epoch);
    if (!peer_wait(D, D.flags_g[D.rank], epoch)) return;   // dead exchange: no step from a partial sum (peer.h)
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
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
    adam_apply4(F, 4 * t, g);
}


// two-phase exchange, phase 2 (phase 1 = k_peer_reduce_slice in peer.hip): every element's sum comes from the rank that
// owns its slice; Adam on all of them.  Same values as k_peer_adam computes itself: bit-identical.
__global__ __launch_bounds__(256) void k_peer_adam2(const PeerDev D, const AdamFuse F, int n4, int u) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_r, epoch);
    if (!peer_wait(D, D.flags_r[D.rank], epoch, 3u)) return;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const int owner = t / peer_slice_len(D, n4);
    const float4 acc = peer_load4(D.red[owner][par], (size_t)n4 * 16, (unsigned)t * 16u, owner == D.rank);
    const float g[4] = {acc.x, acc.y, acc.z, acc.w};
    if (F.keep_grads) *reinterpret_cast<float4 *>(const_cast<float *>(F.grads_base) + 4 * (size_t)t) = acc;
    adam_apply4(F, 4 * t, g);
}

static int peer_enqueue_adam(hp_peer *p, const AdamFuse &F, int n_arena, int u, bool mean) {
    const int n4 = n_arena / 4;
    if (p->phases == 2) {
        HP_TRY(peer_enqueue_reduce_slice(p, n4, u, mean));
        HP_TRY(peer_enqueue_gate(p, 3, u));
        hipLaunchKernelGGL(k_peer_adam2, dim3((n4 + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev, F, n4, u);
    } else {
        HP_TRY(peer_enqueue_gate(p, 1, u));
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

template <bool ADAM>
__device__ __forceinline__ void gemm_ride_body(const GemmGroup &grp, const AdamFuse *F, const RideArgs &R, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[GL_LDS_FLOATS];
    __shared__ float bsum[GL_WAVES][32];
    if ((int)blockIdx.x < tiles) {
        gemm_tile<ADAM>(grp, F, (int)blockIdx.x, lds, bsum, blockIdx.x == 0);
        return;
    }
    const int extra = (int)blockIdx.x - tiles;
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
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_ride(const GemmGroup grp, const RideArgs R, int tiles) {
    gemm_ride_body<false>(grp, nullptr, R, tiles);
}
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_ride(const GemmGroup grp, const AdamFuse F, const RideArgs R,
                                                                   int tiles) {
    gemm_ride_body<true>(grp, &F, R, tiles);
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


enum { PROF_SAMPLE = 0, PROF_GEMM_FWD = 1, PROF_GEMM_BWD = 2, PROF_LOSS = 3, PROF_ADAM = 4, PROF_PLAN = 5, PROF_DW = 6, PROF_N = 7 };

struct hp_agent {
    hp_ctx *ctx = nullptr;
    hp_agent_cfg cfg;
    int H = 256, B = 0, Mp = 0;
    int xdim = 0, act_off = 0, ldx = 0;  // obs+goal, column of the action block, row stride of X buffers
    NetLayout la, lc;                    // actor / critic layouts; critic segment starts at la.total
    int n_arena = 0;
    float *params = nullptr, *targets = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float *XA = nullptr, *XP = nullptr, *XT = nullptr, *R = nullptr, *TP = nullptr;
    float *XA2 = nullptr, *XP2 = nullptr, *XT2 = nullptr, *R2 = nullptr;   // second input set (gather-ahead ping-pong)
    Pass AT, CT, CA, AP, CP;
    float *QT = nullptr, *QA = nullptr, *QP = nullptr, *dQA = nullptr, *dQP = nullptr;
    float *dA3 = nullptr, *dA2 = nullptr, *dA1 = nullptr;  // critic-loss path
    float *dP3 = nullptr, *dP2 = nullptr, *dP1 = nullptr, *dXP = nullptr;  // actor-loss path through the critic
    float *dZ = nullptr, *dK3 = nullptr, *dK2 = nullptr, *dK1 = nullptr;   // actor
    float *loss_log = nullptr;
    AgentDevState *d_state = nullptr;
    // row-slab engine: fragment-ordered weight copies (online forward / online dX / target forward), loss partials
    float *fragF = nullptr, *fragD = nullptr, *fragFT = nullptr, *part = nullptr;
    unsigned long long *timeline = nullptr;   // debug builds (SLAB_TIMELINE) stamp stage boundaries here
    bool slab = true;      // a row-slab engine (false: layer-per-launch engine)
    bool slab8 = true;     // thin slabs on the 4x4x1 MFMA (false: 16-row slabs on 16x16x4, or slab32)
    bool slab32 = false;   // 32-row slabs on the 32x32x2 MFMA, forward + backward in one kernel (slab32.h: large batches)
    int s8_rows = 4;       // slab height of that engine: 4 rows up to batch 448, 8 up to 1280, 16 beyond (RLARM_SLAB_ROWS overrides)
    bool fuse_adam_ok = true;   // Adam in the weight-gradient GEMM's epilogue (RLARM_FUSE_ADAM=0: separate launch, for A/B)
    // A/B switches, read once in hp_agent_create: RLARM_GEMM_PIPE, RLARM_GEMM_XCD (0 = off), RLARM_FB_XCD,
    // RLARM_FB_PREFETCH (-1 = by size, 0 = off, 1 = on)
    bool gemm_pipe = true, gemm_xcd = true;
    int fb_xcd = -1, fb_prefetch = -1;
    // fused single-launch update (FuseArgs in slab8.h): second parameter set the optimizer epilogue writes while the chains
    // of the same launch still read the first, hand-off counters, per-sequence Adam scalars, device copies of the
    // weight-gradient problem table (one per input set)
    float *params_b = nullptr, *fragF_b = nullptr, *fragD_b = nullptr;
    FuseSync *fsync = nullptr;
    GemmGroup *d_grp = nullptr;          // [2]
    int dw_tiles = 0;
    DevBuf adam_tab;                     // float[4] per update of a sequence (sized with the index plan)
    bool fuse_dw_ok = false;             // RLARM_FUSE_DW=1: single-launch updates (default: chain kernel + tile kernel)
    long long fused_launches = 0;
    // large-minibatch weight gradients (dw64.h): 64 x 64 tiles, batch rows split over dw_S workgroups per tile
    bool dw64 = false;                   // RLARM_DW64: default from batch 1536
    int dw_S = 3;                        // RLARM_DW_SPLIT
    DevBuf dw_part, dw_ticket;           // partial tiles / arrival counters
    bool keep_grads_dbg = false;   // RLARM_KEEP_GRADS=1 (parity tests): the peer optimizer kernels also write the summed gradients out
    bool upd_graph_ok = true;   // hp_agent_sample_and_update replays cached graphs (RLARM_UPDATE_GRAPH=0: eager launches, for A/B)
    bool gather_ahead = true;   // merged kernel: gather update u+1's inputs during update u (RLARM_AHEAD=0: off, for A/B)
    DevBuf plan, norm_plan;
    int plan_batches = 0;
    DevBuf fwd_ws;          // actor_forward scratch
    // policy snapshots for a feeder that steps environments while cycles run (hp_agent_policy_snapshot / _act_snapshot)
    struct PolicySnap {
        float *params = nullptr, *fragF = nullptr;   // actor segment of the arenas
        NormDev *on = nullptr, *gn = nullptr;
        double clip_o = 0, clip_g = 0;
        int od = 0, gd = 0;
        hipEvent_t ready = nullptr;
    } snap[2];
    int snap_cur = -1, snap_pending = -1;
    // index plans of later updates drawn on a second stream, concurrently with the chain kernel, when the launch has no
    // spare CU for a ride-along plan workgroup (enqueue_updates)
    hipStream_t plan_stream = nullptr;
    hipEvent_t plan_fork = nullptr, plan_join = nullptr;
    int plan_side = -1;                  // RLARM_PLAN_SIDE: -1 by occupancy, 0 never, 1 always
    hipStream_t act_stream = nullptr;
    hipEvent_t act_done = nullptr;
    bool act_recorded = false;
    DevBuf act_ws;
    PinnedBuf pin;
    std::vector<void *> owned;
    // rank exchange inside the library (hp_agent_set_comm); nullptr: single rank, or the caller exchanges
    hp_comm *comm = nullptr;
    hp_peer *peer = nullptr;      // one-shot exchange over peer memory (hp_agent_set_peer); takes precedence over comm
    bool grad_mean = false;       // divide the all-reduced gradients by the world size (default: SUM, like the reference)
    bool comm_warm = false;       // the collectives of a cycle have each run once outside a capture
    bool graph_refused = false;   // capturing the cycle with collectives failed once: stay on eager launches
    // graphs of hp_agent_sample_and_update(n_updates), one per distinct argument set (a training loop that does not use
    // hp_agent_train_cycle replays its inner loop instead of issuing 2 launches per update)
    struct UpdGraph {
        hipGraphExec_t exec;
        int n_updates;
        hp_buffer *b;
        hp_norm *on, *gn;
        hp_rng *rng;
        double future_p, sq;
    };
    std::vector<UpdGraph> upd_graphs;
    // cycle graph cache
    hipGraphExec_t graph = nullptr;
    hp_buffer *g_buf = nullptr;
    hp_norm *g_on = nullptr, *g_gn = nullptr;
    hp_rng *g_rng = nullptr;
    int64_t g_n_new = -1;
    int g_n_batches = -1;
    double g_future_p = -1, g_sq = -1;
    void *g_stage = nullptr;
    // profiling
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double prof_ms[PROF_N] = {0};
    long long prof_cnt[PROF_N] = {0};
    long long host_steps = 0;
};

// ---------------------------------------------------------------------------------- kernels
// Grouped GEMM on v_mfma_f32_16x16x4_f32.  C[m][n] = sum_k A(m,k) * B(n,k) with generic element
// strides, which covers   forward  Y = X W^T          (A = X,  a_sk = 1;  B = W,  b_sk = 1)
//                         dX = dY W                   (A = dY, a_sk = 1;  B = W,  b_sj = 1, b_sk = ldw)
//                         dW = dY^T X                 (A = dY, a_si = 1, a_sk = ldy;  B = X, b_sj = 1, b_sk = ldx)
// MFMA operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; accumulator register r holds D[row = 4 * (l >> 4) + r][col = l & 15].
struct Acc {
    f32x4 c00, c01, c10, c11;
    float as0, as1;
};

template <int S>
__device__ __forceinline__ void mma_chunk(Acc &acc, const float *a0, const float *b0, long long a16, long long b16,
                                          long long astep, long long bstep, bool vm1, bool vn1) {
    float av0[S], av1[S], bv0[S], bv1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        av0[s] = a0[s * astep];
        bv0[s] = b0[s * bstep];
        av1[s] = vm1 ? a0[a16 + s * astep] : 0.f;
        bv1[s] = vn1 ? b0[b16 + s * bstep] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        acc.c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[s], bv0[s], acc.c00, 0, 0, 0);
        acc.c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[s], bv1[s], acc.c01, 0, 0, 0);
        acc.c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[s], bv0[s], acc.c10, 0, 0, 0);
        acc.c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[s], bv1[s], acc.c11, 0, 0, 0);
        acc.as0 += av0[s];
        acc.as1 += av1[s];
    }
}

__global__ __launch_bounds__(256) void k_gemm_group(const GemmGroup grp) {
    __shared__ float red[4][32 * 33];
    __shared__ float bsum[4][32];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROBS; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.p[i].tile0) pi = i;
    const GemmProb &p = grp.p[pi];
    const int t = blockIdx.x - p.tile0;
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const bool vm1 = (m0 + 16) < p.M, vn1 = (n0 + 16) < p.N;
    const int ksl = p.K >> 2;  // reduction slice of this wavefront
    const int kbeg = wave * ksl + q;
    const float *a0 = p.A + (long long)(m0 + i) * p.a_si + (long long)kbeg * p.a_sk;
    const float *b0 = p.B + (long long)(n0 + i) * p.b_sj + (long long)kbeg * p.b_sk;
    const long long a16 = 16ll * p.a_si, b16 = 16ll * p.b_sj;
    const long long astep = 4ll * p.a_sk, bstep = 4ll * p.b_sk;
    Acc acc;
    acc.c00 = acc.c01 = acc.c10 = acc.c11 = f32x4{0, 0, 0, 0};
    acc.as0 = acc.as1 = 0.f;
    // epilogue operands do not depend on the products: fetch them now so their (cold-cache) latency
    // overlaps the operand loads instead of following the LDS reduction
    const int erow = tid >> 3, ecol = (tid & 7) * 4;
    const int em = m0 + erow, en = n0 + ecol;
    float ev[4] = {0.f, 0.f, 0.f, 0.f};
    if (em < p.M && en < p.N) {
        if (p.epi == EPI_BIAS_RELU || p.epi == EPI_BIAS || p.epi == EPI_BIAS_TANH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ev[j] = p.bias[en + j];
        } else if (p.epi == EPI_MASK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ev[j] = p.mask[(long long)em * p.ldmask + en + j];
        }
    }
    int steps = ksl >> 2;
    // every kernel starts with cold caches (the boundary invalidates them), so a dependent load costs
    // ~0.7 us: issue a whole chunk of operand loads before the first MFMA consumes any of them
    while (steps >= 16) {
        mma_chunk<16>(acc, a0, b0, a16, b16, astep, bstep, vm1, vn1);
        a0 += 16 * astep; b0 += 16 * bstep; steps -= 16;
    }
    if (steps >= 8) { mma_chunk<8>(acc, a0, b0, a16, b16, astep, bstep, vm1, vn1); a0 += 8 * astep; b0 += 8 * bstep; steps -= 8; }
    if (steps >= 4) { mma_chunk<4>(acc, a0, b0, a16, b16, astep, bstep, vm1, vn1); a0 += 4 * astep; b0 += 4 * bstep; steps -= 4; }
    if (steps >= 2) { mma_chunk<2>(acc, a0, b0, a16, b16, astep, bstep, vm1, vn1); a0 += 2 * astep; b0 += 2 * bstep; steps -= 2; }
    if (steps >= 1) { mma_chunk<1>(acc, a0, b0, a16, b16, astep, bstep, vm1, vn1); }
    const f32x4 c00 = acc.c00, c01 = acc.c01, c10 = acc.c10, c11 = acc.c11;
    float as0 = acc.as0, as1 = acc.as1;
    // partial tiles -> LDS (row stride 33 spreads the 4 row groups over banks)
    float *my = red[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        my[row * 33 + i] = c00[r];
        my[row * 33 + 16 + i] = c01[r];
        my[(16 + row) * 33 + i] = c10[r];
        my[(16 + row) * 33 + 16 + i] = c11[r];
    }
    if (p.bias_grad != nullptr && tn == 0) {  // wave-uniform
        as0 += __shfl_xor(as0, 16);
        as0 += __shfl_xor(as0, 32);
        as1 += __shfl_xor(as1, 16);
        as1 += __shfl_xor(as1, 32);
        if (q == 0) {
            bsum[wave][i] = as0;
            bsum[wave][16 + i] = as1;
        }
    }
    __syncthreads();
    if (p.bias_grad != nullptr && tn == 0 && tid < 32 && m0 + tid < p.M)
        p.bias_grad[m0 + tid] = (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]);
    // each thread finishes 4 consecutive columns of one row
    const int row = tid >> 3, col = (tid & 7) * 4;
    const int m = m0 + row;
    if (m >= p.M) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = row * 33 + col + j;
        v[j] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    }
    const int n = n0 + col;
    if (n >= p.N) return;
    switch (p.epi) {
        case EPI_BIAS_RELU:
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] + ev[j], 0.f);
            break;
        case EPI_BIAS:
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] + ev[j];
            break;
        case EPI_MASK:
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (ev[j] > 0.f) ? v[j] : 0.f;
            break;
        case EPI_BIAS_TANH: {
            // models.py:24: actions = max_action * tanh(.); the critic consumes actions / max_action
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j < p.n_store) {
                    const float th = tanhf(v[j] + ev[j]);
                    p.C2[(long long)m * p.ldc2 + n + j] = th;
                    p.C[(long long)m * p.ldc + n + j] = (p.max_action * th) / p.max_action;
                }
            }
            return;
        }
        default: break;
    }
    if (n + 3 < p.n_store) {
        *reinterpret_cast<float4 *>(p.C + (long long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < p.n_store) p.C[(long long)m * p.ldc + n + j] = v[j];
    }
}

// HER gather + relabel + reward + clip + normalise, straight into the network input rows.
// One wavefront per transition (64 lanes ~ 54 obs + 3 + 3 + 4 values of a bmirobot transition).
// Reference: her.py:26-38, ddpg_agent.py:228-243, normalizer.py:67-70.  float64 in, float32 out
// (torch.tensor(..., dtype=float32) rounds to nearest even, as the cast below does).
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

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    // fixed-order tree: deterministic run to run
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return tot;
}

// ddpg_agent.py:255-267: targets, both losses and their first derivatives.  One workgroup.
__global__ __launch_bounds__(256) void k_loss(const float *__restrict__ QT, const float *__restrict__ QA,
                                              const float *__restrict__ QP, const float *__restrict__ R,
                                              const float *__restrict__ XP, int ldx, int act_off, int act_dim, int B,
                                              int Mp, float gamma, float clip_ret, float action_l2, float *dQA,
                                              float *dQP, float *loss_log, AgentDevState *st, const AdamCfg adam) {
    __shared__ float sh[4];
    float sc = 0.f, sq = 0.f, sl2 = 0.f;
    const float invB = 1.0f / (float)B;
    for (int i = threadIdx.x; i < Mp; i += 256) {
        if (i < B) {
            float y = R[i] + gamma * QT[i * 16];          // target_q = r + gamma * q_next
            y = fminf(fmaxf(y, -clip_ret), 0.f);          // clamp(-1/(1-gamma), 0)
            const float d = y - QA[i * 16];
            sc += d * d;
            dQA[i * 16] = -2.f * d * invB;                // d/dq mean((y-q)^2)
            sq += QP[i * 16];
            dQP[i * 16] = -invB;                          // d/dq (-mean(q))
            for (int j = 0; j < act_dim; ++j) {
                const float u = XP[i * ldx + act_off + j];
                sl2 += u * u;
            }
        } else {
            dQA[i * 16] = 0.f;
            dQP[i * 16] = 0.f;
        }
    }
    const float tc = block_sum_256(sc, sh);
    const float tq = block_sum_256(sq, sh);
    const float tl = block_sum_256(sl2, sh);
    if (threadIdx.x == 0) {
        const long long k = st->n_logged;
        const float critic_loss = tc * invB;
        const float actor_loss = -(tq * invB) + action_l2 * (tl / (float)(B * act_dim));
        loss_log[(k % LOSS_LOG) * 2 + 0] = actor_loss;
        loss_log[(k % LOSS_LOG) * 2 + 1] = critic_loss;
        st->n_logged = k + 1;
        st->step += 1;
        adam_prepare(st, adam);
    }
}

// actor head backward (autograd of ddpg_agent.py:265-267 w.r.t. the pre-tanh output):
//   grad_u = action_l2 * 2u/(B*act_dim) + dXP[:, action block];  grad_pi = grad_u / max_action;
//   grad_tanh = grad_pi * max_action;  dZ = grad_tanh * (1 - tanh^2)
__global__ void k_actor_head(const float *__restrict__ dXP, const float *__restrict__ XP, const float *__restrict__ TP,
                             int ldx, int act_off, int act_dim, int B, float action_l2, float max_action, float *dZ) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * act_dim) return;
    const int i = idx / act_dim, j = idx - i * act_dim;
    const float u = XP[i * ldx + act_off + j];
    const float th = TP[i * 16 + j];
    const float gu = action_l2 * (2.f * u / (float)(B * act_dim)) + dXP[i * ldx + act_off + j];
    const float gt = (gu / max_action) * max_action;
    dZ[i * 16 + j] = gt * (1.f - th * th);
}

// torch.optim.Adam (_single_tensor_adam, no weight decay / amsgrad) over the whole arena.
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g,
                                              float *__restrict__ m, float *__restrict__ v, int n, int n_actor,
                                              float w, float b2, float omb2, float epsf, const AgentDevState *st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float neg_step_size = (idx < n_actor) ? st->neg_step_actor : st->neg_step_critic;
    const float bc2_sqrt = st->bc2_sqrt;
    const float gi = g[idx];
    float mi = m[idx], vi = v[idx];
    mi = __fadd_rn(mi, __fmul_rn(w, __fsub_rn(gi, mi)));                 // exp_avg.lerp_(grad, 1 - beta1)
    vi = __fadd_rn(__fmul_rn(vi, b2), __fmul_rn(__fmul_rn(omb2, gi), gi));  // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float sq = __fsqrt_rn(vi);                     // correctly rounded float32 sqrt
    const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), epsf);
    p[idx] = __fadd_rn(p[idx], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    m[idx] = mi;
    v[idx] = vi;
}

// ddpg_agent.py:220-222: target = (1 - polyak) * param + polyak * target
__global__ void k_polyak(float *__restrict__ tgt, const float *__restrict__ src, int n, float one_minus, float polyak) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    tgt[idx] = __fadd_rn(__fmul_rn(one_minus, src[idx]), __fmul_rn(polyak, tgt[idx]));
}

// Fused single-launch updates, once per sequence of n updates: zero the hand-off counters and tabulate the Adam step
// scalars of every update of the sequence (bias corrections of step + u + 1: torch computes them in Python doubles),
// so that no workgroup of the fused launches has to wait for a pow().  k_seq_end moves the step counter on.
__global__ void k_seq_begin(const AgentDevState *st, FuseSync *sync, float *tab, int n, const AdamCfg c) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u == 0) {
        sync->chains_done = 0ull;
        // `error` is sticky: hp_agent_fused_status reports it
    }
    if (u >= n) return;
    const double step = (double)(st->step + u + 1);
    const double bc1 = 1.0 - pow(c.beta1, step);
    const double bc2 = 1.0 - pow(c.beta2, step);
    tab[4 * u + 0] = (float)(-(c.lr_actor / bc1));
    tab[4 * u + 1] = (float)(-(c.lr_critic / bc1));
    tab[4 * u + 2] = (float)sqrt(bc2);
    tab[4 * u + 3] = 0.f;
}
__global__ void k_seq_end(AgentDevState *st, int n, const AdamCfg c) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->step += n;
        adam_prepare(st, c);   // keeps the scalars in *st those of the last applied step, as the two-launch path leaves them
    }
}

// slab engine: Adam that also refreshes the fragment-ordered copies and finishes the loss log
__global__ __launch_bounds__(256) void k_adam_frag(const AdamFuse F, const float *__restrict__ g, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    if (idx >= n) return;
    adam_apply(F, idx, g[idx]);
}

// 4 consecutive arena elements per thread (n % 4 == 0)
__global__ __launch_bounds__(256) void k_adam_frag4(const AdamFuse F, const float *__restrict__ g, int n4) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) loss_finalize(F);
    if (t >= n4) return;
    const float4 g4 = *reinterpret_cast<const float4 *>(g + 4 * t);
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
    adam_apply4(F, 4 * t, gv);
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
static AdamCfg adam_cfg(const hp_agent *a) {
    return AdamCfg{a->cfg.lr_actor, a->cfg.lr_critic, a->cfg.adam_beta1, a->cfg.adam_beta2, a->cfg.adam_eps};
}

static NetLayout make_layout(int K1, int H) {
    NetLayout l;
    l.K1 = K1;
    int o = 0;
    l.w1 = o; o += H * K1;
    l.b1 = o; o += H;
    l.w2 = o; o += H * H;
    l.b2 = o; o += H;
    l.w3 = o; o += H * H;
    l.b3 = o; o += H;
    l.w4 = o; o += 16 * H;
    l.b4 = o; o += 16;
    l.total = o;
    return l;
}

static int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct Launch {  // builds one grouped launch
    GemmGroup g;
    int tiles = 0;
    Launch() {
        g.n = 0;
        g.pipe = 1;   // the agent's switches are applied at launch (apply_switches)
        g.xcd = 0;
        g.pad_ = 0;
    }
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and everything a weight-gradient tile reads
    // was written by the previous kernel on other XCDs: it comes through the fabric once per XCD that touches it.  In
    // row-major tile order every XCD reads 9 of the 16 operand panels of every problem (6.75 x the unique bytes in
    // total); with one 256 x 256 problem per pair of XCDs (half of the row panels each) the fabric carries 1.5 x.
    void place_on_xcds() {
        if (g.n < 4) return;
        for (int i = 0; i < 4; ++i)
            if (g.p[i].tiles_n != 8 || g.p[i].M != 256 || g.p[i].tile0 != 64 * i) return;
        g.xcd = 1;
    }
    GemmProb &add(int M, int N, int K) {
        GemmProb &p = g.p[g.n++];
        memset(&p, 0, sizeof(p));
        p.M = M; p.N = N; p.K = K;
        p.n_store = N;
        p.tiles_n = (N + 31) / 32;
        p.tile0 = tiles;
        tiles += ((M + 31) / 32) * p.tiles_n;
        return p;
    }
};

struct ProfScope {
    hp_agent *a;
    int which;
    ProfScope(hp_agent *ag, int w) : a(ag), which(w) {
        if (a->prof) (void)hipEventRecord(a->ev0, a->ctx->stream);
    }
    ~ProfScope() {
        if (a->prof) {
            (void)hipEventRecord(a->ev1, a->ctx->stream);
            (void)hipEventSynchronize(a->ev1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, a->ev0, a->ev1);
            a->prof_ms[which] += ms;
            a->prof_cnt[which] += 1;
        }
    }
};

static bool use_direct_gemm() {  // A/B switch: RLARM_GEMM=direct selects the first (global-fed) kernel
    static const bool v = [] { const char *e = getenv("RLARM_GEMM"); return e && strcmp(e, "direct") == 0; }();
    return v;
}

static int launch_group(hp_agent *a, const Launch &L, int which) {
    ProfScope ps(a, which);
    if (use_direct_gemm())
        hipLaunchKernelGGL(k_gemm_group, dim3(L.tiles), dim3(256), 0, a->ctx->stream, L.g);
    else
        hipLaunchKernelGGL(k_gemm_lds, dim3(L.tiles), dim3(GL_THREADS), 0, a->ctx->stream, L.g);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// forward layer Y = act(X W^T + b)
static void add_fwd(Launch &L, const float *X, int ldx, int K, const float *W, const float *bias, float *Y, int ldy,
                    int M, int N, int epi) {
    GemmProb &p = L.add(M, N, K);
    p.A = X; p.a_si = ldx; p.a_sk = 1;
    p.B = W; p.b_sj = K; p.b_sk = 1;
    p.C = Y; p.ldc = ldy;
    p.bias = bias;
    p.epi = epi;
}

// dX = (dY W) * relu'(gate)
static void add_dx(Launch &L, const float *dY, int ldy, int Nout, const float *W, int Kin, float *dX, int lddx, int M,
                   const float *gate, int ldgate) {
    GemmProb &p = L.add(M, Kin, Nout);
    p.A = dY; p.a_si = ldy; p.a_sk = 1;
    p.B = W; p.b_sj = 1; p.b_sk = Kin;
    p.C = dX; p.ldc = lddx;
    p.mask = gate; p.ldmask = ldgate;
    p.epi = gate ? EPI_MASK : EPI_NONE;
}

// dW = dY^T X, db = column sums of dY
static void add_dw(Launch &L, const float *dY, int ldy, int Nout, const float *X, int ldx, int Kin, float *dW,
                   float *db, int Mrows) {
    GemmProb &p = L.add(Nout, Kin, Mrows);
    p.A = dY; p.a_si = 1; p.a_sk = ldy;
    p.B = X; p.b_sj = 1; p.b_sk = ldx;
    p.C = dW; p.ldc = Kin;
    p.bias_grad = db;
    p.epi = EPI_NONE;
}

static int enqueue_gather(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, const PlanRec *plan, double sq,
                          int xset = 0, hipStream_t stream = nullptr) {
    ProfScope ps(a, PROF_SAMPLE);
    hipLaunchKernelGGL(k_gather_fused, dim3((a->B + 3) / 4), dim3(256), 0, stream ? stream : a->ctx->stream, b->d_obs, b->d_ag,
                       b->d_g, b->d_act, plan, a->B, (int)b->T, (int)b->obs_dim, (int)b->goal_dim, (int)b->act_dim, sq, on->d,
                       gn->d, a->cfg.clip_obs, a->cfg.clip_range, (float)a->cfg.max_action, a->ldx, a->act_off,
                       xset ? a->XA2 : a->XA, xset ? a->XP2 : a->XP, xset ? a->XT2 : a->XT, xset ? a->R2 : a->R);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

struct GatherCtx {   // where the minibatch comes from (nullptr plan = inputs already staged in XA/XP/XT/R)
    hp_buffer *b;
    hp_norm *on, *gn;
    const PlanRec *plan;
    double sq;
    // slab engine: draw a LATER update's index plan in a spare workgroup of this update's kernel
    hp_rng *rng = nullptr;
    PlanRec *next_plan = nullptr;
    double future_p = 0.0;
    // merged slab8 kernel: input sets ping-pong between updates.  xset = the set this update reads (and, when it
    // gathers in-kernel, writes); pregathered = a previous launch already filled it; ahead_plan = plan of the NEXT
    // update, gathered by spare workgroups of this launch into the other set.
    int xset = 0;
    bool pregathered = false;
    const PlanRec *ahead_plan = nullptr;
    // fused single-launch update: index of this update in its sequence (-1: chain kernel + tile kernel as two launches)
    int fuse_u = -1;
    // full chain launch: the plan draw (next_plan) and the look-ahead gather (ahead_plan) ride in the weight-gradient
    // launch instead of the chain kernel
    bool ride_in_dw = false;
    // data-parallel ranks exchanging through peer memory: this update's gradients go straight into the exchange buffer
    float *grads_out = nullptr;
};

static int enqueue_forward_backward_slab(hp_agent *a, const GatherCtx *gc, bool fuse_adam, int only = 0);
static AdamFuse adam_fuse(hp_agent *a);

// layer-per-launch engine: forwards + losses + backwards of one update, inputs in XA/XP/XT/R (18 launches)
static int enqueue_forward_backward_layers(hp_agent *a) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    float *Pa = a->params, *Pc = a->params + la.total;
    float *Ta = a->targets, *Tc = a->targets + la.total;
    float *Ga = a->grads, *Gc = a->grads + la.total;
    const float maxa = (float)a->cfg.max_action;
    hipStream_t s = a->ctx->stream;
    {   // level 1-3: hidden layers of actor_target(x'), critic(x,a), actor(x)
        Launch L;
        add_fwd(L, a->XT, ldx, la.K1, Ta + la.w1, Ta + la.b1, a->AT.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XA, ldx, lc.K1, Pc + lc.w1, Pc + lc.b1, a->CA.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XP, ldx, la.K1, Pa + la.w1, Pa + la.b1, a->AP.h1, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->AT.h1, H, H, Ta + la.w2, Ta + la.b2, a->AT.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CA.h1, H, H, Pc + lc.w2, Pc + lc.b2, a->CA.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->AP.h1, H, H, Pa + la.w2, Pa + la.b2, a->AP.h2, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->AT.h2, H, H, Ta + la.w3, Ta + la.b3, a->AT.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CA.h2, H, H, Pc + lc.w3, Pc + lc.b3, a->CA.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->AP.h2, H, H, Pa + la.w3, Pa + la.b3, a->AP.h3, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 4: heads.  tanh outputs land in the action block of the critic inputs
        Launch L;
        add_fwd(L, a->AT.h3, H, H, Ta + la.w4, Ta + la.b4, a->XT + a->act_off, ldx, Mp, 16, EPI_BIAS_TANH);
        L.g.p[0].n_store = a->cfg.act_dim; L.g.p[0].C2 = a->TP + 16 * (size_t)Mp; L.g.p[0].ldc2 = 16; L.g.p[0].max_action = maxa;
        add_fwd(L, a->CA.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QA, 16, Mp, 16, EPI_BIAS);
        add_fwd(L, a->AP.h3, H, H, Pa + la.w4, Pa + la.b4, a->XP + a->act_off, ldx, Mp, 16, EPI_BIAS_TANH);
        L.g.p[2].n_store = a->cfg.act_dim; L.g.p[2].C2 = a->TP; L.g.p[2].ldc2 = 16; L.g.p[2].max_action = maxa;
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 5-8: critic_target(x', a') and critic(x, pi(x))
        Launch L;
        add_fwd(L, a->XT, ldx, lc.K1, Tc + lc.w1, Tc + lc.b1, a->CT.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XP, ldx, lc.K1, Pc + lc.w1, Pc + lc.b1, a->CP.h1, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h1, H, H, Tc + lc.w2, Tc + lc.b2, a->CT.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CP.h1, H, H, Pc + lc.w2, Pc + lc.b2, a->CP.h2, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h2, H, H, Tc + lc.w3, Tc + lc.b3, a->CT.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CP.h2, H, H, Pc + lc.w3, Pc + lc.b3, a->CP.h3, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h3, H, H, Tc + lc.w4, Tc + lc.b4, a->QT, 16, Mp, 16, EPI_BIAS);
        add_fwd(L, a->CP.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QP, 16, Mp, 16, EPI_BIAS);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 9: losses and dL/dq
        ProfScope ps(a, PROF_LOSS);
        const double clip_ret = 1.0 / (1.0 - a->cfg.gamma);  // ddpg_agent.py:259
        hipLaunchKernelGGL(k_loss, dim3(1), dim3(256), 0, s, a->QT, a->QA, a->QP, a->R, a->XP, ldx, a->act_off,
                           (int)a->cfg.act_dim, a->B, Mp, (float)a->cfg.gamma, (float)clip_ret,
                           (float)a->cfg.action_l2, a->dQA, a->dQP, a->loss_log, a->d_state, adam_cfg(a));
        HP_CHECK_HIP(hipGetLastError());
    }
    {   // level 10-13: backward through the critic, for the critic loss (dX + dW) and for the actor loss (dX only)
        Launch L;
        add_dx(L, a->dQA, 16, 16, Pc + lc.w4, H, a->dA3, H, Mp, a->CA.h3, H);
        add_dw(L, a->dQA, 16, 16, a->CA.h3, H, H, Gc + lc.w4, Gc + lc.b4, Mp);
        add_dx(L, a->dQP, 16, 16, Pc + lc.w4, H, a->dP3, H, Mp, a->CP.h3, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dA3, H, H, Pc + lc.w3, H, a->dA2, H, Mp, a->CA.h2, H);
        add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp);
        add_dx(L, a->dP3, H, H, Pc + lc.w3, H, a->dP2, H, Mp, a->CP.h2, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dA2, H, H, Pc + lc.w2, H, a->dA1, H, Mp, a->CA.h1, H);
        add_dw(L, a->dA2, H, H, a->CA.h1, H, H, Gc + lc.w2, Gc + lc.b2, Mp);
        add_dx(L, a->dP2, H, H, Pc + lc.w2, H, a->dP1, H, Mp, a->CP.h1, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dw(L, a->dA1, H, H, a->XA, ldx, lc.K1, Gc + lc.w1, Gc + lc.b1, Mp);
        add_dx(L, a->dP1, H, H, Pc + lc.w1, lc.K1, a->dXP, ldx, Mp, nullptr, 0);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {   // level 14: through tanh and the action penalty
        ProfScope ps(a, PROF_LOSS);
        const int n = a->B * a->cfg.act_dim;
        hipLaunchKernelGGL(k_actor_head, dim3((n + 255) / 256), dim3(256), 0, s, a->dXP, a->XP, a->TP, ldx, a->act_off,
                           (int)a->cfg.act_dim, a->B, (float)a->cfg.action_l2, maxa, a->dZ);
        HP_CHECK_HIP(hipGetLastError());
    }
    {   // level 15-18: actor backward
        Launch L;
        add_dx(L, a->dZ, 16, 16, Pa + la.w4, H, a->dK3, H, Mp, a->AP.h3, H);
        add_dw(L, a->dZ, 16, 16, a->AP.h3, H, H, Ga + la.w4, Ga + la.b4, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dK3, H, H, Pa + la.w3, H, a->dK2, H, Mp, a->AP.h2, H);
        add_dw(L, a->dK3, H, H, a->AP.h2, H, H, Ga + la.w3, Ga + la.b3, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dK2, H, H, Pa + la.w2, H, a->dK1, H, Mp, a->AP.h1, H);
        add_dw(L, a->dK2, H, H, a->AP.h1, H, H, Ga + la.w2, Ga + la.b2, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dw(L, a->dK1, H, H, a->XP, ldx, la.K1, Ga + la.w1, Ga + la.b1, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    return HP_OK;
}

// one update's forwards + backwards.  gc == nullptr: the minibatch is already staged in XA/XP/XT/R.
// fuse_adam: the optimizer step follows immediately on this rank (no gradient exchange): the slab engines then apply
// it in the weight-gradient GEMM's epilogue and the caller must NOT enqueue Adam again (returns that via *fused).
static int enqueue_forward_backward(hp_agent *a, const GatherCtx *gc = nullptr, bool fuse_adam = false,
                                    bool *fused = nullptr) {
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

static ArenaMap arena_map(const hp_agent *a) {
    ArenaMap am;
    am.la = a->la;
    am.lc = a->lc;
    am.H = a->H;
    am.mode = a->slab8 ? 1 : (a->slab32 ? 2 : 0);
    return am;
}

static int enqueue_relayout(hp_agent *a, bool targets) {
    const int n = a->n_arena;
    hipLaunchKernelGGL(k_relayout, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream,
                       targets ? a->targets : a->params, targets ? a->fragFT : a->fragF,
                       targets ? (float *)nullptr : a->fragD, n, arena_map(a));
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// all weight gradients of one update: the only products that reduce over the batch (input sets sXA / sXP)
static Launch build_dw_group(const hp_agent *a, const float *sXA, const float *sXP, float *grads = nullptr) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    if (!grads) grads = a->grads;
    float *Ga = grads, *Gc = grads + la.total;
    Launch L;
    // the four 256 x 256 problems first: Launch::place_on_xcds gives each of them one pair of XCDs
    add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp);
    add_dw(L, a->dA2, H, H, a->CA.h1, H, H, Gc + lc.w2, Gc + lc.b2, Mp);
    add_dw(L, a->dK3, H, H, a->AP.h2, H, H, Ga + la.w3, Ga + la.b3, Mp);
    add_dw(L, a->dK2, H, H, a->AP.h1, H, H, Ga + la.w2, Ga + la.b2, Mp);
    add_dw(L, a->dQA, 16, 16, a->CA.h3, H, H, Gc + lc.w4, Gc + lc.b4, Mp);
    add_dw(L, a->dA1, H, H, sXA, ldx, lc.K1, Gc + lc.w1, Gc + lc.b1, Mp);
    add_dw(L, a->dZ, 16, 16, a->AP.h3, H, H, Ga + la.w4, Ga + la.b4, Mp);
    add_dw(L, a->dK1, H, H, sXP, ldx, la.K1, Ga + la.w1, Ga + la.b1, Mp);
    if (a->gemm_xcd) L.place_on_xcds();
    L.g.pipe = a->gemm_pipe ? 1 : 0;
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

// slab engine: forwards + losses + backwards of one update (inputs in XA/XP/XT/R): 3 launches
// only = 1 / 2: just the chain kernel / just the weight-gradient launch (timing diagnostics, hp_agent_debug_chain)
static int enqueue_forward_backward_slab(hp_agent *a, const GatherCtx *gc, bool fuse_adam, int only) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    hipStream_t s = a->ctx->stream;
    const int nslab = Mp / (a->slab8 ? a->s8_rows : (a->slab32 ? S32_ROWS : SL_ROWS));
    FbSlabArgs P;
    const int xs = gc ? gc->xset : 0;
    float *sXA = xs ? a->XA2 : a->XA, *sXP = xs ? a->XP2 : a->XP, *sXT = xs ? a->XT2 : a->XT, *sR = xs ? a->R2 : a->R;
    // fused single-launch update: this update's chains read parameter set (u & 1), its optimizer epilogue writes the other
    const bool fused = a->slab8 && fuse_adam && gc && gc->fuse_u >= 0;
    const int rset = fused ? (gc->fuse_u & 1) : 0;
    const SlabNetPtrs online = rset ? SlabNetPtrs{a->fragF_b, a->fragD_b, a->params_b}
                                    : SlabNetPtrs{a->fragF, a->fragD, a->params};
    memset(&P.fuse, 0, sizeof(P.fuse));
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
    if (only == 2) {
    } else if (a->slab8) {
        // one launch: each workgroup carries its rows through forward AND backward (k_fb_slab8)
        ProfScope ps(a, PROF_GEMM_FWD);
        P.n_plan = ride ? 1 : 0;
        P.n_ahead = 0;
        // chains split across XCD halves: measured (us/update, split vs not) 42.0 vs 43.5 at batch 128, 44.0 vs 45.2 at 256,
        // 46.6 vs 46.6 at 384, 48.0 vs 47.8 at 448, 77.3 vs 74.6 at 1024 -- it pays while the chains leave half of the CUs free
        P.xcd_split = (nslab % 4 == 0) && (a->fb_xcd >= 0 ? a->fb_xcd == 1 : 4 * nslab <= a->ctx->cu_count);
        P.ahead = P.f.gs;
        P.aXT = P.aXA = P.aXP = nullptr;
        if (gc && gc->ahead_plan && !ride_dw) {   // next update's inputs into the other set
            P.n_ahead = S8_AHEAD_WGS;
            P.ahead.plan = gc->ahead_plan;
            P.ahead.plan_any = gc->ahead_plan;
            P.ahead.R = xs ? a->R : a->R2;
            P.aXT = xs ? a->XT : a->XT2; P.aXA = xs ? a->XA : a->XA2; P.aXP = xs ? a->XP : a->XP2;
        }
        // L2 warmers: one spare workgroup per XCD while the launch still fits the CUs.  Measured (us/update, with vs
        // without): 39.9 vs 42.4 at batch 128, 42.1 vs 44.4 at 256, 46.4 vs 46.6 at 384, 52.8 vs 54.2 at 512, 56.6 vs 57.1 at 768
        P.n_pref = (a->fb_prefetch >= 0 ? a->fb_prefetch == 1
                                        : 2 * nslab + P.n_plan + P.n_ahead + 8 <= a->ctx->cu_count) ? 8 : 0;
        unsigned grid = 2 * nslab + P.n_plan + P.n_ahead + P.n_pref;
        if (fused) {
            AdamFuse F = adam_fuse(a);
            F.p = online.canon;
            F.p_out = rset ? a->params : a->params_b;
            F.fragF = rset ? a->fragF : a->fragF_b;
            F.fragD = rset ? a->fragD : a->fragD_b;
            F.scal = a->adam_tab.as<float>() + 4 * (size_t)gc->fuse_u;
            F.keep_grads = a->keep_grads_dbg ? 1 : 0;   // nobody reads the gradient vector inside a sampled update loop (RLARM_KEEP_GRADS=1: parity tests do)
            P.fuse.on = 1;
            P.fuse.u = gc->fuse_u;
            P.fuse.n_tiles = a->dw_tiles;
            P.fuse.sync = a->fsync;
            P.fuse.grp = a->d_grp + xs;
            P.fuse.adam = F;
            // idle CUs hold pure tile workers: one workgroup per CU (LDS), chains first in dispatch order
            const unsigned cus = (unsigned)a->ctx->cu_count;
            if (grid < cus) grid = cus;
            a->fused_launches += 1;
        }
        if (a->s8_rows == 4)
            hipLaunchKernelGGL(s8r4::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        else if (a->s8_rows == 8)
            hipLaunchKernelGGL(s8r8::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        else
            hipLaunchKernelGGL(s8r16::k_fb_slab8, dim3(grid), dim3(S8_THREADS), 0, s, P);
        HP_CHECK_HIP(hipGetLastError());
    } else if (a->slab32) {
        // 32-row slabs, forward + backward of a chain in one workgroup; inputs come gathered (enqueue_updates)
        ProfScope ps(a, PROF_GEMM_FWD);
        P.n_plan = ride ? 1 : 0;
        P.n_ahead = P.n_pref = P.xcd_split = 0;
        hipLaunchKernelGGL(s32::k_fb_slab32, dim3(2 * nslab + P.n_plan), dim3(S32_THREADS), 0, s, P);
        HP_CHECK_HIP(hipGetLastError());
    } else {
        {
            ProfScope ps(a, PROF_GEMM_FWD);
            hipLaunchKernelGGL(k_fwd_slab, dim3(nslab, 3), dim3(SL_THREADS), 0, s, P.f);
            HP_CHECK_HIP(hipGetLastError());
        }
        {
            ProfScope ps(a, PROF_GEMM_BWD);
            hipLaunchKernelGGL(k_bwd_slab, dim3(2 * nslab + (ride ? 1 : 0)), dim3(SL_THREADS), 0, s, P.b);
            HP_CHECK_HIP(hipGetLastError());
        }
    }
    if (!fused && only != 1) {   // all weight gradients (+ the optimizer when no gradient exchange follows) as their own launch
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
                hipLaunchKernelGGL(k_dw64_adam, dim3(grid), dim3(DW_THREADS), 0, s, L.g, F, R, X);
            } else {
                hipLaunchKernelGGL(k_dw64, dim3(grid), dim3(DW_THREADS), 0, s, L.g, R, X);
            }
            HP_CHECK_HIP(hipGetLastError());
        } else if (riders) {
            ProfScope ps(a, PROF_DW);
            const unsigned grid = L.tiles + R.n_plan + R.n_ahead;
            if (fuse_adam) {
                AdamFuse F = adam_fuse(a);
                F.keep_grads = a->keep_grads_dbg ? 1 : 0;
                hipLaunchKernelGGL(k_gemm_lds_adam_ride, dim3(grid), dim3(GL_THREADS), 0, s, L.g, F, R, L.tiles);
            } else {
                hipLaunchKernelGGL(k_gemm_lds_ride, dim3(grid), dim3(GL_THREADS), 0, s, L.g, R, L.tiles);
            }
            HP_CHECK_HIP(hipGetLastError());
        } else if (fuse_adam) {
            ProfScope ps(a, PROF_DW);
            // inside a sampled update loop nobody reads the gradient vector (hp_agent_get_grads documents this): 1.17 MB of
            // the ~6.7 MB this kernel leaves dirty in L2 for the end-of-kernel write-back
            AdamFuse F = adam_fuse(a);
            F.keep_grads = (gc == nullptr || a->keep_grads_dbg) ? 1 : 0;
            hipLaunchKernelGGL(k_gemm_lds_adam, dim3(L.tiles), dim3(GL_THREADS), 0, s, L.g, F);
            HP_CHECK_HIP(hipGetLastError());
        } else {
            HP_TRY(launch_group(a, L, PROF_DW));
        }
    }
    return HP_OK;
}

static AdamFuse adam_fuse(hp_agent *a) {
    AdamFuse F;
    F.p = a->params; F.p_out = a->params; F.m = a->adam_m; F.v = a->adam_v; F.fragF = a->fragF; F.fragD = a->fragD;
    F.grads_base = a->grads; F.st = a->d_state; F.scal = &a->d_state->neg_step_actor; F.am = arena_map(a); F.n_actor = a->la.total;
    F.keep_grads = 1;
    F.w = (float)(1.0 - a->cfg.adam_beta1); F.b2 = (float)a->cfg.adam_beta2;
    F.omb2 = (float)(1.0 - a->cfg.adam_beta2); F.eps = (float)a->cfg.adam_eps;
    F.part = a->part; F.nslab = a->Mp / (a->slab8 ? a->s8_rows : (a->slab32 ? S32_ROWS : SL_ROWS)); F.B = a->B;
    F.act_dim = a->cfg.act_dim;
    F.action_l2 = (float)a->cfg.action_l2; F.loss_log = a->loss_log;
    return F;
}

static int enqueue_adam(hp_agent *a) {
    ProfScope ps(a, PROF_ADAM);
    const int n = a->n_arena;
    if (a->slab) {
        if (n % 4 == 0 && a->la.total % 4 == 0)
            hipLaunchKernelGGL(k_adam_frag4, dim3((n / 4 + 255) / 256), dim3(256), 0, a->ctx->stream, adam_fuse(a), a->grads, n / 4);
        else
            hipLaunchKernelGGL(k_adam_frag, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, adam_fuse(a), a->grads, n);
        HP_CHECK_HIP(hipGetLastError());
        return HP_OK;
    }
    hipLaunchKernelGGL(k_adam, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->params, a->grads, a->adam_m,
                       a->adam_v, n, a->la.total, (float)(1.0 - a->cfg.adam_beta1), (float)a->cfg.adam_beta2,
                       (float)(1.0 - a->cfg.adam_beta2), (float)a->cfg.adam_eps, a->d_state);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static int enqueue_polyak(hp_agent *a) {
    ProfScope ps(a, PROF_ADAM);
    const int n = a->n_arena;
    const double om = 1.0 - a->cfg.polyak;
    if (a->slab) {
        hipLaunchKernelGGL(k_polyak_frag, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->targets, a->params,
                           a->fragFT, n, (float)om, (float)a->cfg.polyak, arena_map(a));
        HP_CHECK_HIP(hipGetLastError());
        return HP_OK;
    }
    hipLaunchKernelGGL(k_polyak, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->targets, a->params, n, (float)om,
                       (float)a->cfg.polyak);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static void drop_graph(hp_agent *a);

static int ensure_plan(hp_agent *a, int n_batches) {
    if (n_batches > a->plan_batches) {
        // the cached cycle graph has the plan's address baked into its draw / gather / ride-along kernels: growing the
        // plan frees that memory, so the graph goes with it (rebuilt by the next hp_agent_train_cycle)
        drop_graph(a);
        HP_TRY(a->plan.ensure((size_t)n_batches * a->B * sizeof(PlanRec)));
        HP_TRY(a->adam_tab.ensure((size_t)n_batches * 4 * sizeof(float)));
        a->plan_batches = n_batches;
    }
    return HP_OK;
}

static int check_handles(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, const char *who) {
    HP_REQUIRE(a && b && on && gn && rng, HP_ERR_INVALID, "%s: null handle", who);
    HP_REQUIRE(b->obs_dim == a->cfg.obs_dim && b->goal_dim == a->cfg.goal_dim && b->act_dim == a->cfg.act_dim,
               HP_ERR_INVALID, "%s: buffer dimensions do not match the agent", who);
    HP_REQUIRE(on->size == a->cfg.obs_dim && gn->size == a->cfg.goal_dim, HP_ERR_INVALID,
               "%s: normalizer sizes do not match the agent", who);
    return HP_OK;
}

// n_updates x (sample + update); the index plan for all of them is drawn by one kernel up front
// (nothing else consumes the stream in between, exactly like the reference's inner loop).
static int enqueue_updates(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p, double sq,
                           int n_updates, bool with_adam) {
    // slab engine: only the first minibatch's indices are drawn up front; update u draws the plan of update u+1 in
    // a spare workgroup of its backward kernel (same stream order of draws, so the same indices).  Layer engine:
    // one kernel draws all of them (nothing else consumes the stream in between, like the reference's inner loop).
    const bool ride = a->slab && with_adam;
    // merged slab8 kernel: plans are drawn TWO updates ahead so that spare workgroups of update u can gather the inputs
    // of update u+1 from a plan that an earlier launch finished (the order of draws in the stream is unchanged)
    // Single-launch updates: the weight-gradient tiles and the optimizer run as a second phase of the chain kernel
    // (slab8.h FuseArgs).  Needs the optimizer to follow the gradients directly (one rank) and every chain workgroup
    // resident at once (one per CU: the tile phase starts when ALL chains have published).
    const int chains = 2 * (a->Mp / a->s8_rows);
    const bool fuse_dw = a->slab8 && with_adam && !a->comm && !a->peer && a->fuse_adam_ok && a->fuse_dw_ok && a->d_grp &&
                         chains <= a->ctx->cu_count;
    // ... and when the launch is full (no CU for spare workgroups) both jobs move out of the chain kernel
    const bool full = a->slab8 && chains + 1 + S8_AHEAD_WGS > a->ctx->cu_count;
    // slab8 engine, full launch: both jobs ride in the weight-gradient launch (k_gemm_lds_adam_ride); RLARM_PLAN_SIDE=2
    // keeps the second-stream variant for A/B (its cross-queue graph edges cost ~4 us each: 65.5 vs 71.4 us at batch 1024)
    // slab32 engine: its chain kernel never gathers and its launches fill the CUs from batch 4096, so both jobs ride in the
    // weight-gradient launch by default (measured at batch 4096: beside the chain kernel on a second stream k_draw_plan
    // took 83 us instead of 3.4 and the chain kernel 97 us); RLARM_PLAN_SIDE=2 keeps the second stream, 0 runs the gather in
    // front of every launch.  Profiling brackets every launch with events on the main stream: serial as well.
    const bool s32_ride = ride && a->slab32 && !a->prof && a->plan_side != 0 && a->plan_side != 2;
    const bool s32_side = ride && a->slab32 && !a->prof && a->plan_side == 2;
    const bool s32_serial = a->slab32 && !s32_side && !s32_ride;
    const bool want_offload = s32_ride || (ride && a->slab8 && !fuse_dw && a->plan_side != 0 && (a->plan_side >= 1 || full) &&
                                           (getenv("RLARM_AHEAD") ? a->gather_ahead : true));
    const bool side_gather = (want_offload && a->slab8 && a->plan_side == 2 && !a->prof) || s32_side;
    const bool dw_ride = want_offload && !side_gather;
    const bool ahead = ride && ((a->slab8 && (a->gather_ahead || side_gather || dw_ride)) || s32_side || s32_ride);
    const int lead = ahead ? 2 : 1;
    if (fuse_dw) {
        hipLaunchKernelGGL(k_seq_begin, dim3((n_updates + 63) / 64), dim3(64), 0, a->ctx->stream, a->d_state, a->fsync,
                           a->adam_tab.as<float>(), n_updates, adam_cfg(a));
        HP_CHECK_HIP(hipGetLastError());
    }
    {
        ProfScope ps(a, PROF_PLAN);
        HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, a->B, ride ? (n_updates < lead ? n_updates : lead) : n_updates,
                               future_p, a->plan.as<PlanRec>()));
    }
    // Where the plan of update u + lead is drawn: by a spare workgroup of update u's own launch while that launch leaves
    // a CU free -- or, when the chains occupy every CU (batch 1024: 256 chain workgroups; the 16-row engine beyond), by
    // k_draw_plan on a second stream next to the chain kernel.  A workgroup appended to a full launch only starts when
    // the first chain ends and then runs its sequential MT19937 draw alone: measured 51.5 vs 37.8 us for k_fb_slab8 at
    // batch 1024 and 96 vs 51 us for k_bwd_slab at batch 4096 (profiles/r02_large_batch_traces.txt).
    const int spare_cus = a->ctx->cu_count - (a->slab8 ? chains + (ahead && !side_gather ? S8_AHEAD_WGS : 0)
                                              : (a->slab32 ? 2 * (a->Mp / S32_ROWS) : 3 * (a->Mp / SL_ROWS)));
    const bool side = side_gather || (ride && !dw_ride && !a->prof && (a->plan_side >= 0 ? a->plan_side >= 1 : spare_cus < 1));
    if (side && !a->plan_stream) {
        HP_CHECK_HIP(hipStreamCreateWithFlags(&a->plan_stream, hipStreamNonBlocking));
        HP_CHECK_HIP(hipEventCreateWithFlags(&a->plan_fork, hipEventDisableTiming));
        HP_CHECK_HIP(hipEventCreateWithFlags(&a->plan_join, hipEventDisableTiming));
    }
    bool join_pending = false;
    for (int u = 0; u < n_updates; ++u) {
        GatherCtx gc{b, on, gn, a->plan.as<PlanRec>() + (size_t)u * a->B, sq};
        hipStream_t ms = a->ctx->stream;
        if (join_pending) {   // the plan drawn beside the previous update is what this launch gathers from
            HP_CHECK_HIP(hipStreamWaitEvent(ms, a->plan_join, 0));
            join_pending = false;
        }
        const bool gather_beside = side_gather && u + 1 < n_updates;
        if (side && (gather_beside || (ride && u + lead < n_updates))) {
            // fork: ordered behind everything enqueued so far (the previous draws included), concurrent with update u
            HP_CHECK_HIP(hipEventRecord(a->plan_fork, ms));
            HP_CHECK_HIP(hipStreamWaitEvent(a->plan_stream, a->plan_fork, 0));
            if (gather_beside)   // inputs of update u + 1 into the other input set, from the plan an earlier draw finished
                HP_TRY(enqueue_gather(a, b, on, gn, a->plan.as<PlanRec>() + (size_t)(u + 1) * a->B, sq, (u + 1) & 1,
                                      a->plan_stream));
            if (ride && u + lead < n_updates)
                HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, a->B, 1, future_p,
                                       a->plan.as<PlanRec>() + (size_t)(u + lead) * a->B, a->plan_stream));
            HP_CHECK_HIP(hipEventRecord(a->plan_join, a->plan_stream));
            join_pending = true;
        }
        if (ride && u + lead < n_updates) {
            PlanRec *next = a->plan.as<PlanRec>() + (size_t)(u + lead) * a->B;
            if (side) {
                (void)next;
            } else if (a->slab32 && s32_serial && 2 * (a->Mp / S32_ROWS) >= a->ctx->cu_count) {
                // serial mode (profiling, RLARM_PLAN_SIDE=0) of a launch that fills the CUs: a spare workgroup would only start
                // when the first chain ends (137 instead of 87 us per launch at batch 4096) -- draw in front of the launch
                HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, a->B, 1, future_p, next));
            } else {
                gc.rng = rng;
                gc.next_plan = next;
                gc.future_p = future_p;
            }
        }
        if (a->slab32 && (s32_serial || u == 0))   // nobody gathered this update's inputs beside the previous one
            HP_TRY(enqueue_gather(a, b, on, gn, gc.plan, sq, ahead ? (u & 1) : 0));
        if (ahead) {
            gc.xset = u & 1;
            gc.pregathered = u > 0;
            if (u + 1 < n_updates && !side_gather) gc.ahead_plan = a->plan.as<PlanRec>() + (size_t)(u + 1) * a->B;
        }
        gc.ride_in_dw = dw_ride;
        if (fuse_dw) gc.fuse_u = u;
        const bool via_peer = with_adam && a->peer != nullptr;
        if (via_peer) gc.grads_out = peer_grad_buffer(a->peer, u + 1);   // epoch base is even: parity of epoch base + u + 1
        bool fused = false;
        HP_TRY(enqueue_forward_backward(a, &gc, with_adam && !a->comm && !a->peer, &fused));
        if (via_peer) {
            // utils.sync_grads (utils.py:43-48) + both Adam steps in ONE kernel: every rank reads the peers' gradient
            // vectors over xGMI, sums them in rank order and steps (peer.hip)
            AdamFuse F = adam_fuse(a);
            F.grads_base = a->grads;
            F.keep_grads = a->keep_grads_dbg ? 1 : 0;   // RLARM_KEEP_GRADS=1: hp_agent_get_grads then returns the exchanged sum
            ProfScope ps(a, PROF_ADAM);
            HP_TRY(peer_enqueue_adam(a->peer, F, a->n_arena, u, a->grad_mean));
        } else if (with_adam) {
            // utils.sync_grads (utils.py:43-48): SUM over ranks between backward and the optimizer step; one
            // all-reduce covers both networks (the reference sends the actor's and the critic's separately)
            if (a->comm)
                HP_TRY(a->grad_mean ? comm_allreduce_mean_f32(a->comm, a->grads, (size_t)a->n_arena)
                                    : comm_allreduce_sum_f32(a->comm, a->grads, (size_t)a->n_arena));
            if (!fused) HP_TRY(enqueue_adam(a));
        }
    }
    if (fuse_dw) {
        hipStream_t s = a->ctx->stream;
        if (n_updates & 1) {
            // update u wrote parameter set (u + 1) & 1: after an odd count the live parameters sit in set b.  Between API
            // calls they always live in set a (what every other entry point, graph and kernel argument refers to).
            const size_t nb = sizeof(float) * (size_t)a->n_arena;
            HP_CHECK_HIP(hipMemcpyAsync(a->params, a->params_b, nb, hipMemcpyDeviceToDevice, s));
            HP_CHECK_HIP(hipMemcpyAsync(a->fragF, a->fragF_b, nb, hipMemcpyDeviceToDevice, s));
            HP_CHECK_HIP(hipMemcpyAsync(a->fragD, a->fragD_b, nb, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(k_seq_end, dim3(1), dim3(64), 0, s, a->d_state, n_updates, adam_cfg(a));
        HP_CHECK_HIP(hipGetLastError());
    }
    if (join_pending) HP_CHECK_HIP(hipStreamWaitEvent(a->ctx->stream, a->plan_join, 0));
    if (with_adam && a->peer) HP_TRY(peer_enqueue_seq_end(a->peer, n_updates));
    return HP_OK;
}

// reference flat order (utils.py:18-27): fc1.weight, fc1.bias, fc2.weight, ... out.weight, out.bias
static void pack_net(const hp_agent *a, bool critic, const float *flat, float *arena_seg) {
    const NetLayout &l = critic ? a->lc : a->la;
    const int H = a->H, xdim = a->xdim, act = a->cfg.act_dim;
    const int in1 = critic ? xdim + act : xdim;
    const int out4 = critic ? 1 : act;
    memset(arena_seg, 0, sizeof(float) * l.total);
    const float *src = flat;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < in1; ++c) arena_seg[l.w1 + r * l.K1 + (c < xdim ? c : a->act_off + (c - xdim))] = *src++;
    memcpy(arena_seg + l.b1, src, H * 4); src += H;
    memcpy(arena_seg + l.w2, src, H * H * 4); src += H * H;
    memcpy(arena_seg + l.b2, src, H * 4); src += H;
    memcpy(arena_seg + l.w3, src, H * H * 4); src += H * H;
    memcpy(arena_seg + l.b3, src, H * 4); src += H;
    memcpy(arena_seg + l.w4, src, out4 * H * 4); src += out4 * H;
    memcpy(arena_seg + l.b4, src, out4 * 4);
}

static void unpack_net(const hp_agent *a, bool critic, const float *arena_seg, float *flat) {
    const NetLayout &l = critic ? a->lc : a->la;
    const int H = a->H, xdim = a->xdim, act = a->cfg.act_dim;
    const int in1 = critic ? xdim + act : xdim;
    const int out4 = critic ? 1 : act;
    float *dst = flat;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < in1; ++c) *dst++ = arena_seg[l.w1 + r * l.K1 + (c < xdim ? c : a->act_off + (c - xdim))];
    memcpy(dst, arena_seg + l.b1, H * 4); dst += H;
    memcpy(dst, arena_seg + l.w2, H * H * 4); dst += H * H;
    memcpy(dst, arena_seg + l.b2, H * 4); dst += H;
    memcpy(dst, arena_seg + l.w3, H * H * 4); dst += H * H;
    memcpy(dst, arena_seg + l.b3, H * 4); dst += H;
    memcpy(dst, arena_seg + l.w4, out4 * H * 4); dst += out4 * H;
    memcpy(dst, arena_seg + l.b4, out4 * 4);
}

static int64_t flat_count(const hp_agent *a, bool critic) {
    const int H = a->H, act = a->cfg.act_dim;
    const int in1 = critic ? a->xdim + act : a->xdim;
    const int out4 = critic ? 1 : act;
    return (int64_t)H * in1 + H + 2ll * (H * H + H) + (int64_t)out4 * H + out4;
}

static int arena_read(hp_agent *a, const float *d_arena, bool critic, float *flat_host, int64_t n) {
    HP_REQUIRE(n == flat_count(a, critic), HP_ERR_INVALID, "flat vector has %lld elements, expected %lld", (long long)n,
               (long long)flat_count(a, critic));
    const NetLayout &l = critic ? a->lc : a->la;
    std::vector<float> seg(l.total);
    HP_CHECK_HIP(hipMemcpyAsync(seg.data(), d_arena + (critic ? a->la.total : 0), sizeof(float) * l.total,
                                hipMemcpyDeviceToHost, a->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
    unpack_net(a, critic, seg.data(), flat_host);
    return HP_OK;
}

template <class T> static int dev_alloc(hp_agent *a, T **p, size_t count) {
    void *q = nullptr;
    HP_CHECK_HIP(hipMalloc(&q, count * sizeof(T)));
    HP_CHECK_HIP(hipMemsetAsync(q, 0, count * sizeof(T), a->ctx->stream));
    a->owned.push_back(q);
    *p = static_cast<T *>(q);
    return HP_OK;
}

static void drop_graph(hp_agent *a) {   // every cached graph: they all bake in the plan address, the communicator, the switches
    if (a->graph) (void)hipGraphExecDestroy(a->graph);
    a->graph = nullptr;
    for (auto &u : a->upd_graphs) (void)hipGraphExecDestroy(u.exec);
    a->upd_graphs.clear();
}

// --------------------------------------------------------------------------------- C ABI
extern "C" {

int hp_agent_create(hp_ctx *ctx, const hp_agent_cfg *cfg, hp_agent **out) {
    HP_REQUIRE(ctx && cfg && out, HP_ERR_INVALID, "hp_agent_create: null argument");
    HP_REQUIRE(cfg->hidden > 0 && cfg->hidden % 32 == 0, HP_ERR_INVALID, "hp_agent_create: hidden must be a multiple of 32");
    HP_REQUIRE(cfg->batch > 0, HP_ERR_INVALID, "hp_agent_create: batch must be positive");
    HP_REQUIRE(cfg->obs_dim > 0 && cfg->goal_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 16, HP_ERR_INVALID,
               "hp_agent_create: need obs_dim, goal_dim > 0 and 0 < act_dim <= 16");
    HP_REQUIRE(cfg->max_action > 0, HP_ERR_INVALID, "hp_agent_create: max_action must be positive");
    hp_agent *a = new hp_agent();
    a->ctx = ctx;
    a->cfg = *cfg;
    a->H = cfg->hidden;
    a->B = cfg->batch;
    a->Mp = roundup(cfg->batch, 32);
    a->xdim = cfg->obs_dim + cfg->goal_dim;
    a->act_off = roundup(a->xdim, 16);
    a->ldx = roundup(a->act_off + cfg->act_dim, 16);
    a->la = make_layout(a->act_off, a->H);
    a->lc = make_layout(a->ldx, a->H);
    a->n_arena = a->la.total + a->lc.total;
    const size_t Mp = a->Mp, H = a->H, ldx = a->ldx;
    int st = HP_OK;
    auto A = [&](float **p, size_t n) { if (st == HP_OK) st = dev_alloc(a, p, n); };
    A(&a->params, a->n_arena); A(&a->targets, a->n_arena); A(&a->grads, a->n_arena);
    A(&a->adam_m, a->n_arena); A(&a->adam_v, a->n_arena);
    A(&a->XA, Mp * ldx); A(&a->XP, Mp * ldx); A(&a->XT, Mp * ldx); A(&a->R, Mp); A(&a->TP, 2 * Mp * 16);
    A(&a->XA2, Mp * ldx); A(&a->XP2, Mp * ldx); A(&a->XT2, Mp * ldx); A(&a->R2, Mp);
    for (Pass *ps : {&a->AT, &a->CT, &a->CA, &a->AP, &a->CP}) { A(&ps->h1, Mp * H); A(&ps->h2, Mp * H); A(&ps->h3, Mp * H); }
    A(&a->QT, Mp * 16); A(&a->QA, Mp * 16); A(&a->QP, Mp * 16); A(&a->dQA, Mp * 16); A(&a->dQP, Mp * 16);
    A(&a->dA3, Mp * H); A(&a->dA2, Mp * H); A(&a->dA1, Mp * H);
    A(&a->dP3, Mp * H); A(&a->dP2, Mp * H); A(&a->dP1, Mp * H); A(&a->dXP, Mp * ldx);
    A(&a->dZ, Mp * 16); A(&a->dK3, Mp * H); A(&a->dK2, Mp * H); A(&a->dK1, Mp * H);
    A(&a->loss_log, LOSS_LOG * 2);
    A(&a->fragF, a->n_arena); A(&a->fragD, a->n_arena); A(&a->fragFT, a->n_arena); A(&a->part, 3 * (Mp / 4));
    {
        // RLARM_ENGINE = slab8 (default) | slab16 | layers: the alternatives stay for A/B runs and debugging
        const char *e = getenv("RLARM_ENGINE");
        // the slab engines are specialised: 256-wide hidden layers, network inputs of at most 48 columns (obs + goal +
        // action, padded to 16) and at most 4 action components; any other shape takes the layer-per-launch engine
        const bool slab_shape = a->H == 256 && a->ldx <= 48 && cfg->act_dim <= 4;
        a->slab = !(e && strcmp(e, "layers") == 0) && slab_shape;
        // Thin slabs (4x4x1 MFMA, slab8.h) while their chains fit the CUs in one round (16-row slabs: batch <= 2048), 32-row
        // slabs on the 32x32x2 MFMA (slab32.h) beyond; the 16-row two-kernel engine (16x16x4 MFMA, slab.h) stays selectable
        // for A/B.  Measured us/update, slab8 / slab16 / slab32 (profiles/r02_large_batch_engines.txt):
        //   1024: 59.5 / 88.0 / 106.8    1536: 87.7 / 119.5 / 113.2    2048: 97.4 / 127.5 / 119.8
        //   3072: 163.3 / 173.6 / 132.9  4096: 180.1 / 202.7 / 146.7
        const int cus_e = a->ctx->cu_count > 0 ? a->ctx->cu_count : 256;
        const bool thin_fits = 2 * (a->Mp / 16) <= cus_e;
        a->slab8 = a->slab && (e ? (strcmp(e, "slab16") != 0 && strcmp(e, "slab32") != 0) : thin_fits);
        a->slab32 = a->slab && !a->slab8 && (e ? strcmp(e, "slab32") == 0 : true);
        // Thin slabs buy latency at small batches (more CUs busy, less matrix work per streamed weight block) and cost
        // L2 weight traffic per row.  The kernel's LDS footprint allows one workgroup per CU, so a launch with more chain
        // workgroups than CUs runs in two rounds: the rule is "the thinnest slab whose 2 * B / rows chains fit the CUs".
        // Measured, us/update (tools/ubench/rows_sweep2.sh, sweep_rows.sh): 4 vs 8 rows 41.4 vs 45 at batch 256, 46.8 vs 48.6
        // at 448, 47.5 vs 49.6 at 480, 48.4 vs 49.9 at 512 (256 chains: the look-ahead then rides in the weight-gradient
        // launch), 73.4 vs 52.0 at 544 (two rounds); 8 vs 16 rows 59.1 vs 81.4 at 1024, 98.3 vs 90.9 at 1280.
        const int cus = a->ctx->cu_count > 0 ? a->ctx->cu_count : 256;
        a->s8_rows = 2 * (a->Mp / 4) <= cus ? 4 : (2 * (a->Mp / 8) <= cus ? 8 : 16);
        if (const char *sr = getenv("RLARM_SLAB_ROWS")) {
            if (strcmp(sr, "4") == 0) a->s8_rows = 4;
            if (strcmp(sr, "8") == 0) a->s8_rows = 8;
            if (strcmp(sr, "16") == 0) a->s8_rows = 16;
        }
        const char *fa = getenv("RLARM_FUSE_ADAM");
        a->fuse_adam_ok = !(fa && fa[0] == '0');
        auto tri = [](const char *name) { const char *e = getenv(name); return e ? (e[0] != '0' ? 1 : 0) : -1; };
        a->gemm_pipe = tri("RLARM_GEMM_PIPE") != 0;
        a->gemm_xcd = tri("RLARM_GEMM_XCD") != 0;
        a->fb_xcd = tri("RLARM_FB_XCD");
        a->fb_prefetch = tri("RLARM_FB_PREFETCH");
        a->upd_graph_ok = tri("RLARM_UPDATE_GRAPH") != 0;
        a->keep_grads_dbg = tri("RLARM_KEEP_GRADS") == 1;
        // measured slower than two launches (48.2 vs 40.8 us/update at batch 256: the in-kernel hand-off costs ~4 us and a
        // tile ~7 us warm, profiles/r02_fused_single_launch.txt), so it is opt-in
        a->fuse_dw_ok = tri("RLARM_FUSE_DW") == 1;
        if (const char *ps = getenv("RLARM_PLAN_SIDE")) a->plan_side = atoi(ps);   // -1 auto, 0 off, 1 on, 2 on via a second stream
        // weight gradients: 64 x 64 tiles with split batch rows (dw64.h) where the 32 x 32 tiles are L2-bound
        // (us/update, 32 x 32 tiles vs dw64: 56.6 / 58.9 at batch 1024, 85.8 / 85.1 at 1536, 93.9 / 92.2 at 2048, 146 / 128 at 4096)
        a->dw64 = a->slab && (tri("RLARM_DW64") >= 0 ? tri("RLARM_DW64") == 1 : a->Mp >= 1536);
        if (const char *ds = getenv("RLARM_DW_SPLIT")) a->dw_S = atoi(ds) > 0 && atoi(ds) <= 16 ? atoi(ds) : a->dw_S;
        const char *ah = getenv("RLARM_AHEAD");
        a->gather_ahead = !(ah && ah[0] == '0');
        // ... and the spare workgroups of the gather-ahead only pay while they find free CUs next to the chains: at batch
        // 1024 (256 chains) they ran after them, 87.6 vs 76.1 us/update
        const int chains = 2 * (a->Mp / a->s8_rows);
        if (!ah && chains + 1 + S8_AHEAD_WGS > cus) a->gather_ahead = false;
    }
    if (st == HP_OK) st = dev_alloc(a, &a->d_state, 1);
    if (st == HP_OK) st = dev_alloc(a, &a->timeline, 192);
    if (st == HP_OK && a->dw64) {
        Launch L = build_dw_group(a, a->XA, a->XP);
        size_t tiles = 0;
        for (int i = 0; i < L.g.n; ++i) tiles += (size_t)((L.g.p[i].M + 63) / 64) * ((L.g.p[i].N + 63) / 64);
        if (a->dw_part.ensure(tiles * a->dw_S * DW_PART * sizeof(float)) != HP_OK || a->dw_ticket.ensure(tiles * sizeof(unsigned long long)) != HP_OK ||
            hipMemsetAsync(a->dw_ticket.p, 0, tiles * sizeof(unsigned long long), a->ctx->stream) != hipSuccess)
            st = HP_ERR_HIP;
    }
    if (st == HP_OK && a->slab8) {   // fused single-launch update (slab8.h FuseArgs)
        A(&a->params_b, a->n_arena); A(&a->fragF_b, a->n_arena); A(&a->fragD_b, a->n_arena);
        if (st == HP_OK) st = dev_alloc(a, &a->fsync, 1);
        if (st == HP_OK) st = dev_alloc(a, &a->d_grp, 2);
        if (st == HP_OK) {
            GemmGroup g2[2];
            for (int xs = 0; xs < 2; ++xs) {
                const Launch L = build_dw_group(a, xs ? a->XA2 : a->XA, xs ? a->XP2 : a->XP);
                g2[xs] = L.g;
                a->dw_tiles = L.tiles;
            }
            if (hipMemcpyAsync(a->d_grp, g2, sizeof(g2), hipMemcpyHostToDevice, a->ctx->stream) != hipSuccess ||
                hipStreamSynchronize(a->ctx->stream) != hipSuccess)
                st = HP_ERR_HIP;
        }
    }
    if (st == HP_OK && hipEventCreate(&a->ev0) != hipSuccess) st = HP_ERR_HIP;
    if (st == HP_OK && hipEventCreate(&a->ev1) != hipSuccess) st = HP_ERR_HIP;
    if (st == HP_OK) st = ensure_plan(a, 1);
    if (st != HP_OK) {
        hp_agent_destroy(a);
        return st;
    }
    *out = a;
    return HP_OK;
}

int64_t hp_agent_param_count(hp_agent *a, int32_t net) {
    if (!a) return -1;
    return flat_count(a, net == HP_NET_CRITIC || net == HP_NET_CRITIC_TARGET);
}

int hp_agent_set_params(hp_agent *a, int32_t net, const float *flat_host, int64_t n) {
    HP_REQUIRE(a && flat_host, HP_ERR_INVALID, "hp_agent_set_params: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(net >= 0 && net <= 3, HP_ERR_INVALID, "hp_agent_set_params: net=%d not in 0..3", net);
    const bool critic = (net == HP_NET_CRITIC || net == HP_NET_CRITIC_TARGET);
    const bool target = net >= 2;
    HP_REQUIRE(n == flat_count(a, critic), HP_ERR_INVALID, "hp_agent_set_params: got %lld values, expected %lld",
               (long long)n, (long long)flat_count(a, critic));
    const NetLayout &l = critic ? a->lc : a->la;
    std::vector<float> seg(l.total);
    pack_net(a, critic, flat_host, seg.data());
    float *dst = (target ? a->targets : a->params) + (critic ? a->la.total : 0);
    HP_CHECK_HIP(hipMemcpyAsync(dst, seg.data(), sizeof(float) * l.total, hipMemcpyHostToDevice, a->ctx->stream));
    HP_TRY(enqueue_relayout(a, target));
    HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
    return HP_OK;
}

int hp_agent_get_params(hp_agent *a, int32_t net, float *flat_host, int64_t n) {
    HP_REQUIRE(a && flat_host, HP_ERR_INVALID, "hp_agent_get_params: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(net >= 0 && net <= 3, HP_ERR_INVALID, "hp_agent_get_params: net=%d not in 0..3", net);
    return arena_read(a, net >= 2 ? a->targets : a->params, net == HP_NET_CRITIC || net == HP_NET_CRITIC_TARGET,
                      flat_host, n);
}

int hp_agent_get_grads(hp_agent *a, int32_t net, float *flat_host, int64_t n) {
    HP_REQUIRE(a && flat_host, HP_ERR_INVALID, "hp_agent_get_grads: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(net == HP_NET_ACTOR || net == HP_NET_CRITIC, HP_ERR_INVALID, "hp_agent_get_grads: net must be actor or critic");
    return arena_read(a, a->grads, net == HP_NET_CRITIC, flat_host, n);
}

int hp_agent_get_adam(hp_agent *a, int32_t net, float *m_host, float *v_host, int64_t n, int64_t *step) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_get_adam: null handle");
    HP_SERIALISE(a);
    HP_REQUIRE(net == HP_NET_ACTOR || net == HP_NET_CRITIC, HP_ERR_INVALID, "hp_agent_get_adam: net must be actor or critic");
    if (m_host) HP_TRY(arena_read(a, a->adam_m, net == HP_NET_CRITIC, m_host, n));
    if (v_host) HP_TRY(arena_read(a, a->adam_v, net == HP_NET_CRITIC, v_host, n));
    if (step) {
        AgentDevState h;
        HP_CHECK_HIP(hipMemcpyAsync(&h, a->d_state, sizeof(h), hipMemcpyDeviceToHost, a->ctx->stream));
        HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
        *step = h.step;
    }
    return HP_OK;
}

// test hook beside hp_agent_get_adam: load optimizer state (torch.optim.Adam's exp_avg / exp_avg_sq in the reference's flat
// order, and the number of steps already taken -- both optimizers step together) so that a teacher-forced comparison can
// restart every update from the oracle's exact state
int hp_agent_set_adam(hp_agent *a, int32_t net, const float *m_host, const float *v_host, int64_t n, int64_t step) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_set_adam: null handle");
    HP_SERIALISE(a);
    HP_REQUIRE(net == HP_NET_ACTOR || net == HP_NET_CRITIC, HP_ERR_INVALID, "hp_agent_set_adam: net must be actor or critic");
    HP_REQUIRE(step >= 0, HP_ERR_INVALID, "hp_agent_set_adam: step must be >= 0");
    const bool critic = net == HP_NET_CRITIC;
    HP_REQUIRE(n == flat_count(a, critic), HP_ERR_INVALID, "hp_agent_set_adam: got %lld values, expected %lld", (long long)n,
               (long long)flat_count(a, critic));
    const NetLayout &l = critic ? a->lc : a->la;
    std::vector<float> seg(l.total);
    for (int which = 0; which < 2; ++which) {
        const float *src = which ? v_host : m_host;
        if (!src) continue;
        pack_net(a, critic, src, seg.data());
        float *dst = (which ? a->adam_v : a->adam_m) + (critic ? a->la.total : 0);
        HP_CHECK_HIP(hipMemcpyAsync(dst, seg.data(), sizeof(float) * l.total, hipMemcpyHostToDevice, a->ctx->stream));
        HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
    }
    const long long st = step;
    HP_CHECK_HIP(hipMemcpyAsync(&a->d_state->step, &st, sizeof(st), hipMemcpyHostToDevice, a->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
    return HP_OK;
}

int hp_agent_sync_targets(hp_agent *a) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_sync_targets: null handle");
    HP_SERIALISE(a);
    HP_CHECK_HIP(hipMemcpyAsync(a->targets, a->params, sizeof(float) * a->n_arena, hipMemcpyDeviceToDevice, a->ctx->stream));
    // the online parameters may just have been overwritten through hp_agent_param_buffer (sync_networks on a rank other
    // than 0): their fragment-ordered copies are rebuilt here too, not only the targets'
    HP_TRY(enqueue_relayout(a, false));
    return enqueue_relayout(a, true);
}

int hp_agent_update_minibatch(hp_agent *a, const float *x, const float *x_next, const float *actions, const float *r,
                              float *losses_host) {
    HP_REQUIRE(a && x && x_next && actions && r, HP_ERR_INVALID, "hp_agent_update_minibatch: null argument");
    HP_SERIALISE(a);
    const int B = a->B, Mp = a->Mp, ldx = a->ldx, xd = a->xdim, ad = a->cfg.act_dim;
    std::vector<float> hxa((size_t)Mp * ldx, 0.f), hxp((size_t)Mp * ldx, 0.f), hxt((size_t)Mp * ldx, 0.f), hr(Mp, 0.f);
    const float maxa = (float)a->cfg.max_action;
    for (int i = 0; i < B; ++i) {
        for (int c = 0; c < xd; ++c) {
            hxa[(size_t)i * ldx + c] = x[(size_t)i * xd + c];
            hxp[(size_t)i * ldx + c] = x[(size_t)i * xd + c];
            hxt[(size_t)i * ldx + c] = x_next[(size_t)i * xd + c];
        }
        for (int c = 0; c < ad; ++c) hxa[(size_t)i * ldx + a->act_off + c] = actions[(size_t)i * ad + c] / maxa;
        hr[i] = r[i];
    }
    hipStream_t s = a->ctx->stream;
    HP_CHECK_HIP(hipMemcpyAsync(a->XA, hxa.data(), hxa.size() * 4, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(a->XP, hxp.data(), hxp.size() * 4, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(a->XT, hxt.data(), hxt.size() * 4, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(a->R, hr.data(), hr.size() * 4, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    bool fused = false;
    HP_TRY(enqueue_forward_backward(a, nullptr, true, &fused));
    if (!fused) HP_TRY(enqueue_adam(a));
    a->host_steps += 1;
    if (losses_host) HP_TRY(hp_agent_get_losses(a, losses_host, 1));
    else HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

int hp_agent_sample_and_update(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p,
                               double sq_threshold, int32_t n_updates) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_sample_and_update"));
    HP_SERIALISE(a);
    HP_REQUIRE(n_updates > 0, HP_ERR_INVALID, "hp_agent_sample_and_update: n_updates must be positive");
    HP_TRY(peer_check_alive(a->peer, "hp_agent_sample_and_update"));
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    HP_TRY(ensure_plan(a, n_updates));
    hipStream_t s = a->ctx->stream;
    // The n_updates x 2 launches have constant arguments (all state is device resident), so the call is replayed as a
    // cached hipGraph: same kernels, same order, same bits as the eager launches, ~1.5 us less boundary per launch.
    // Not under profiling (per-launch events), not on the legacy stream (cannot be captured), not after a refusal.
    const bool graphable = !a->prof && s != hipStreamLegacy && !a->graph_refused && a->upd_graph_ok;
    if (graphable) {
        for (auto &u : a->upd_graphs)
            if (u.n_updates == n_updates && u.b == b && u.on == on && u.gn == gn && u.rng == rng &&
                u.future_p == future_p && u.sq == sq_threshold) {
                HP_CHECK_HIP(hipGraphLaunch(u.exec, s));
                a->host_steps += n_updates;
                return HP_OK;
            }
        if (a->comm && !a->comm_warm) {   // RCCL sets its channels up lazily: first collective outside a capture
            a->comm_warm = true;
            HP_TRY(comm_allreduce_sum_f32(a->comm, a->grads, (size_t)a->n_arena));
            HP_TRY(comm_allreduce_sum_f32(a->comm, on->d->sync, (size_t)(2 * on->size + 1)));
            HP_TRY(comm_allreduce_sum_f32(a->comm, gn->d->sync, (size_t)(2 * gn->size + 1)));
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HP_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int st = enqueue_updates(a, b, on, gn, rng, future_p, sq_threshold, n_updates, true);
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (st == HP_OK && e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (st == HP_OK && e == hipSuccess) {
            if (a->upd_graphs.size() >= 8) {   // a loop uses one or two chunk lengths; keep the cache small
                (void)hipGraphExecDestroy(a->upd_graphs.front().exec);
                a->upd_graphs.erase(a->upd_graphs.begin());
            }
            a->upd_graphs.push_back({exec, n_updates, b, on, gn, rng, future_p, sq_threshold});
            HP_CHECK_HIP(hipGraphLaunch(exec, s));
            a->host_steps += n_updates;
            return HP_OK;
        }
        if (!a->comm) {
            if (st != HP_OK) return st;
            HP_CHECK_HIP(e);
        }
        (void)hipGetLastError();       // a capture with collectives was refused: nothing ran, fall through to eager launches
        a->graph_refused = true;
    }
    HP_TRY(enqueue_updates(a, b, on, gn, rng, future_p, sq_threshold, n_updates, true));
    a->host_steps += n_updates;
    return HP_OK;
}

int hp_agent_forward_backward(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p,
                              double sq_threshold) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_forward_backward"));
    HP_SERIALISE(a);
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    HP_TRY(ensure_plan(a, 1));
    return enqueue_updates(a, b, on, gn, rng, future_p, sq_threshold, 1, false);
}

int hp_agent_grad_buffer(hp_agent *a, void **dev_grads, int64_t *n_floats) {
    HP_REQUIRE(a && dev_grads && n_floats, HP_ERR_INVALID, "hp_agent_grad_buffer: null argument");
    HP_SERIALISE(a);
    *dev_grads = a->grads;
    *n_floats = a->n_arena;
    return HP_OK;
}

int hp_agent_param_buffer(hp_agent *a, void **dev_params, int64_t *n_floats) {
    HP_REQUIRE(a && dev_params && n_floats, HP_ERR_INVALID, "hp_agent_param_buffer: null argument");
    HP_SERIALISE(a);
    *dev_params = a->params;
    *n_floats = a->n_arena;
    return HP_OK;
}

int hp_agent_apply(hp_agent *a) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_apply: null handle");
    HP_SERIALISE(a);
    HP_TRY(enqueue_adam(a));
    a->host_steps += 1;
    return HP_OK;
}

int hp_agent_get_losses(hp_agent *a, float *out_host, int32_t n_last) {
    HP_REQUIRE(a && out_host, HP_ERR_INVALID, "hp_agent_get_losses: null argument");
    HP_SERIALISE(a);
    HP_REQUIRE(n_last > 0 && n_last <= LOSS_LOG, HP_ERR_INVALID, "hp_agent_get_losses: n_last must be in [1, %d]", LOSS_LOG);
    hipStream_t s = a->ctx->stream;
    AgentDevState h;
    std::vector<float> log(LOSS_LOG * 2);
    HP_CHECK_HIP(hipMemcpyAsync(&h, a->d_state, sizeof(h), hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipMemcpyAsync(log.data(), a->loss_log, log.size() * 4, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    HP_REQUIRE(h.n_logged >= n_last, HP_ERR_STATE, "hp_agent_get_losses: only %lld updates logged", h.n_logged);
    for (int i = 0; i < n_last; ++i) {
        const long long k = h.n_logged - n_last + i;
        out_host[2 * i] = log[(k % LOSS_LOG) * 2];
        out_host[2 * i + 1] = log[(k % LOSS_LOG) * 2 + 1];
    }
    return HP_OK;
}

int hp_agent_soft_update(hp_agent *a) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_soft_update: null handle");
    HP_SERIALISE(a);
    return enqueue_polyak(a);
}

}  // extern "C" (re-opened below)

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

// device part of one cycle after the episodes are staged: slots+scatter happen in buffer_stage_and_store
static int enqueue_cycle_tail(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p,
                              double sq, int n_batches, PlanRec *norm_plan) {
    // ddpg_agent._update_normalizer (:187-212)
    HP_TRY(rng_launch_plan(rng, nullptr, b->staged_n, b->T, b->T, 1, future_p, norm_plan));
    if (!a->comm && !a->peer) {   // single rank: update + recompute_stats of both normalizers in one launch
        HP_TRY(norm_launch_update_from_plan(on, gn, b, norm_plan, b->T, a->cfg.clip_obs, true));
    } else if (a->peer) {
        HP_TRY(norm_launch_update_from_plan(on, gn, b, norm_plan, b->T, a->cfg.clip_obs, false));
        HP_TRY(norm_launch_begin(on));
        HP_TRY(norm_launch_begin(gn));
        // normalizer._mpi_average (normalizer.py:60-64) through the mailboxes
        HP_TRY(peer_allreduce_small(a->peer, on->d->sync, (size_t)(2 * on->size + 1), true));
        HP_TRY(peer_allreduce_small(a->peer, gn->d->sync, (size_t)(2 * gn->size + 1), true));
        HP_TRY(norm_launch_end(on));
        HP_TRY(norm_launch_end(gn));
    } else {
        HP_TRY(norm_launch_update_from_plan(on, gn, b, norm_plan, b->T, a->cfg.clip_obs, false));
        HP_TRY(norm_launch_begin(on));
        HP_TRY(norm_launch_begin(gn));
        // normalizer._mpi_average (normalizer.py:60-64) on sum | sumsq | count of each normalizer
        HP_TRY(comm_allreduce_mean_f32(a->comm, on->d->sync, (size_t)(2 * on->size + 1)));
        HP_TRY(comm_allreduce_mean_f32(a->comm, gn->d->sync, (size_t)(2 * gn->size + 1)));
        HP_TRY(norm_launch_end(on));
        HP_TRY(norm_launch_end(gn));
    }
    // ddpg_agent.py:145-150
    HP_TRY(enqueue_updates(a, b, on, gn, rng, future_p, sq, n_batches, true));
    HP_TRY(enqueue_polyak(a));
    return HP_OK;
}

extern "C" {

int hp_agent_set_grad_reduce(hp_agent *a, int32_t mean) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_set_grad_reduce: null handle");
    HP_SERIALISE(a);
    if (a->grad_mean != (mean != 0)) drop_graph(a);
    a->grad_mean = mean != 0;
    return HP_OK;
}

// diagnostic: how hp_agent_train_cycle currently runs -- 0 nothing built yet, 1 cached hipGraph, 2 eager launches
// (a capture containing collectives was refused)
int hp_agent_cycle_mode(hp_agent *a, int32_t *mode) {
    HP_REQUIRE(a && mode, HP_ERR_INVALID, "hp_agent_cycle_mode: null argument");
    HP_SERIALISE(a);
    *mode = a->graph_refused ? 2 : (a->graph ? 1 : 0);
    return HP_OK;
}

int hp_agent_set_peer(hp_agent *a, hp_peer *peer) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_set_peer: null handle");
    HP_SERIALISE(a);
    HP_REQUIRE(!peer || peer->ctx == a->ctx, HP_ERR_INVALID, "hp_agent_set_peer: exchange belongs to another context");
    HP_REQUIRE(!peer || (peer->connected && peer->n_grad == (size_t)a->n_arena), HP_ERR_INVALID,
               "hp_agent_set_peer: exchange not connected, or its gradient length differs from the agent's (%d floats)", a->n_arena);
    HP_REQUIRE(!peer || a->slab, HP_ERR_INVALID, "hp_agent_set_peer: needs a slab engine (the optimizer kernel with fragment copies)");
    drop_graph(a);
    a->peer = peer;
    return HP_OK;
}

int hp_agent_set_comm(hp_agent *a, hp_comm *comm) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_set_comm: null handle");
    HP_SERIALISE(a);
    HP_REQUIRE(!comm || comm->ctx == a->ctx, HP_ERR_INVALID, "hp_agent_set_comm: communicator belongs to another context");
    drop_graph(a);
    a->graph_refused = false;
    a->comm_warm = false;
    a->comm = comm;
    return HP_OK;
}

int hp_agent_train_cycle(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, const double *obs,
                         const double *ag_host, const double *g, const double *actions, int64_t n_new,
                         double future_p, double sq_threshold, int32_t n_batches) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_train_cycle"));
    HP_SERIALISE(a);
    HP_REQUIRE(obs && ag_host && g && actions, HP_ERR_INVALID, "hp_agent_train_cycle: null episode array");
    HP_REQUIRE(n_new > 0 && n_batches > 0, HP_ERR_INVALID, "hp_agent_train_cycle: n_new and n_batches must be positive");
    HP_REQUIRE(!(b->current_size == 0 && n_new > b->size), HP_ERR_INVALID, "high <= 0");
    HP_REQUIRE(!a->prof, HP_ERR_STATE, "hp_agent_train_cycle: profiling mode uses the eager path (hp_agent_profile(0) first)");
    HP_TRY(peer_check_alive(a->peer, "hp_agent_train_cycle"));
    hipStream_t s = a->ctx->stream;
    // 1. episodes -> pinned -> device staging, slots, scatter (eager: the source pointers change per call)
    HP_TRY(buffer_stage_and_store(b, rng, obs, ag_host, g, actions, n_new));
    // 2. everything else is one graph; rebuild when a baked-in argument changes
    const bool same = a->graph && a->g_buf == b && a->g_on == on && a->g_gn == gn && a->g_rng == rng &&
                      a->g_n_new == n_new && a->g_n_batches == n_batches && a->g_future_p == future_p &&
                      a->g_sq == sq_threshold && a->g_stage == b->st_obs.p;
    if (s == hipStreamLegacy) a->graph_refused = true;   // the legacy default stream cannot be captured: eager launches
    if (a->graph_refused) {   // see below
        HP_TRY(ensure_plan(a, n_batches));
        HP_TRY(a->norm_plan.ensure((size_t)b->T * sizeof(PlanRec)));
        HP_TRY(enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>()));
        a->host_steps += n_batches;
        return HP_OK;
    }
    if (!same) {
        drop_graph(a);
        HP_TRY(ensure_plan(a, n_batches));
        HP_TRY(a->norm_plan.ensure((size_t)b->T * sizeof(PlanRec)));
        if (a->comm && !a->comm_warm) {
            a->comm_warm = true;   // once per attach, on the first cycle of every rank: stays symmetric across ranks
            // RCCL sets up its channels lazily on the first collective of a given kind: do that outside the capture
            // (the gradients are recomputed before they are read, the zeroed sync vectors are idle between cycles)
            HP_TRY(comm_allreduce_sum_f32(a->comm, a->grads, (size_t)a->n_arena));
            HP_TRY(comm_allreduce_sum_f32(a->comm, on->d->sync, (size_t)(2 * on->size + 1)));   // overwritten by
            HP_TRY(comm_allreduce_sum_f32(a->comm, gn->d->sync, (size_t)(2 * gn->size + 1)));   // k_norm_begin
        }
        HP_CHECK_HIP(hipStreamSynchronize(s));
        hipGraph_t graph = nullptr;
        HP_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int st = enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>());
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (st == HP_OK && e == hipSuccess) {
            e = hipGraphInstantiate(&a->graph, graph, nullptr, nullptr, 0);
            if (e != hipSuccess) a->graph = nullptr;
        }
        if (graph) (void)hipGraphDestroy(graph);
        if (st != HP_OK || e != hipSuccess) {
            if (!a->comm) {
                if (st != HP_OK) return st;
                HP_CHECK_HIP(e);
            }
            // A capture that contains collectives was refused (RCCL build without graph support, or a lazy allocation
            // inside the capture).  Nothing was executed -- a capture only records -- so the same work is issued as
            // ordinary launches from now on; the other ranks see the same sequence of collectives either way.
            (void)hipGetLastError();
            a->graph_refused = true;
            HP_TRY(enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>()));
            a->host_steps += n_batches;
            return HP_OK;
        }
        a->g_buf = b; a->g_on = on; a->g_gn = gn; a->g_rng = rng;
        a->g_n_new = n_new; a->g_n_batches = n_batches; a->g_future_p = future_p; a->g_sq = sq_threshold;
        a->g_stage = b->st_obs.p;
    }
    HP_CHECK_HIP(hipGraphLaunch(a->graph, s));
    a->host_steps += n_batches;
    return HP_OK;
}

// diagnostic: time `n` back-to-back launches of ONE stage of the update as a captured hipGraph.
//   kind 0: k_loss   1: k_actor_head   2: forward level 2 (3 x 256x256x256)   3: forward level 1 (K=32/48)
//   4: forward heads (N=16)   5: backward level (dX+dW+dX, 256^3)   6: k_adam   7: k_gather_fused is not
//   available here (needs a buffer); 8: k_polyak
int hp_agent_debug_chain(hp_agent *a, int32_t kind, int32_t n, double *us_per_launch) {
    HP_REQUIRE(a && us_per_launch && n > 0, HP_ERR_INVALID, "hp_agent_debug_chain: bad argument");
    HP_SERIALISE(a);
    hipStream_t s = a->ctx->stream;
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    float *Pa = a->params, *Pc = a->params + la.total, *Ta = a->targets, *Gc = a->grads + la.total;
    auto one = [&]() -> int {
        switch (kind) {
            case 0:
                hipLaunchKernelGGL(k_loss, dim3(1), dim3(256), 0, s, a->QT, a->QA, a->QP, a->R, a->XP, ldx, a->act_off,
                                   (int)a->cfg.act_dim, a->B, Mp, 0.98f, 50.f, 1.f, a->dQA, a->dQP, a->loss_log, a->d_state,
                                   adam_cfg(a));
                return HP_OK;
            case 1:
                hipLaunchKernelGGL(k_actor_head, dim3((a->B * a->cfg.act_dim + 255) / 256), dim3(256), 0, s, a->dXP, a->XP,
                                   a->TP, ldx, a->act_off, (int)a->cfg.act_dim, a->B, 1.f, 0.5f, a->dZ);
                return HP_OK;
            case 2: {
                Launch L;
                add_fwd(L, a->AT.h1, H, H, Ta + la.w2, Ta + la.b2, a->AT.h2, H, Mp, H, EPI_BIAS_RELU);
                add_fwd(L, a->CA.h1, H, H, Pc + lc.w2, Pc + lc.b2, a->CA.h2, H, Mp, H, EPI_BIAS_RELU);
                add_fwd(L, a->AP.h1, H, H, Pa + la.w2, Pa + la.b2, a->AP.h2, H, Mp, H, EPI_BIAS_RELU);
                return launch_group(a, L, PROF_GEMM_FWD);
            }
            case 3: {
                Launch L;
                add_fwd(L, a->XT, ldx, la.K1, Ta + la.w1, Ta + la.b1, a->AT.h1, H, Mp, H, EPI_BIAS_RELU);
                add_fwd(L, a->XA, ldx, lc.K1, Pc + lc.w1, Pc + lc.b1, a->CA.h1, H, Mp, H, EPI_BIAS_RELU);
                add_fwd(L, a->XP, ldx, la.K1, Pa + la.w1, Pa + la.b1, a->AP.h1, H, Mp, H, EPI_BIAS_RELU);
                return launch_group(a, L, PROF_GEMM_FWD);
            }
            case 4: {
                Launch L;
                add_fwd(L, a->CA.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QA, 16, Mp, 16, EPI_BIAS);
                add_fwd(L, a->CP.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QP, 16, Mp, 16, EPI_BIAS);
                return launch_group(a, L, PROF_GEMM_FWD);
            }
            case 5: {
                Launch L;
                add_dx(L, a->dA3, H, H, Pc + lc.w3, H, a->dA2, H, Mp, a->CA.h2, H);
                add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp);
                add_dx(L, a->dP3, H, H, Pc + lc.w3, H, a->dP2, H, Mp, a->CP.h2, H);
                return launch_group(a, L, PROF_GEMM_BWD);
            }
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

int hp_agent_engine(hp_agent *a, int32_t *engine, int32_t *slab_rows, int32_t *dw_split) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_engine: null handle");
    if (engine) *engine = !a->slab ? 0 : (a->slab8 ? 8 : (a->slab32 ? 32 : 16));
    if (slab_rows) *slab_rows = !a->slab ? 0 : (a->slab8 ? a->s8_rows : (a->slab32 ? S32_ROWS : SL_ROWS));
    if (dw_split) *dw_split = a->dw64 ? a->dw_S : 0;
    return HP_OK;
}

// diagnostic: fused single-launch updates issued so far, and the device's sticky hand-off error word (a bounded spin of
// the tile phase gave up: only possible when the chain workgroups of a launch were not all resident).  Synchronises.
int hp_agent_fused_status(hp_agent *a, int64_t *fused_launches, uint32_t *error) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_fused_status: null handle");
    HP_SERIALISE(a);
    if (fused_launches) *fused_launches = a->fused_launches;
    if (error) {
        *error = 0;
        if (a->fsync) {
            FuseSync h;
            HP_CHECK_HIP(hipMemcpyAsync(&h, a->fsync, sizeof(h), hipMemcpyDeviceToHost, a->ctx->stream));
            HP_CHECK_HIP(hipStreamSynchronize(a->ctx->stream));
            *error = h.error;
        }
    }
    return HP_OK;
}

int hp_agent_profile(hp_agent *a, int32_t enable) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_profile: null handle");
    HP_SERIALISE(a);
    a->prof = enable != 0;
    for (int i = 0; i < PROF_N; ++i) {
        a->prof_ms[i] = 0;
        a->prof_cnt[i] = 0;
    }
    return HP_OK;
}

// out[2*k] = total ms, out[2*k+1] = launches, k = sample, forward, backward(dX), loss(+head, layer engine),
// adam(+polyak), index plan, weight-gradient GEMM (slab engine)
int hp_agent_profile_read(hp_agent *a, double *ms_out, int32_t n) {
    HP_REQUIRE(a && ms_out, HP_ERR_INVALID, "hp_agent_profile_read: null argument");
    HP_SERIALISE(a);
    for (int i = 0; i < PROF_N && 2 * i + 1 < n; ++i) {
        ms_out[2 * i] = a->prof_ms[i];
        ms_out[2 * i + 1] = (double)a->prof_cnt[i];
    }
    return HP_OK;
}

void hp_agent_destroy(hp_agent *a) {
    if (!a) return;
    drop_graph(a);
    (void)hipStreamSynchronize(a->ctx->stream);
    for (void *p : a->owned) (void)hipFree(p);
    if (a->plan_stream) {
        (void)hipStreamSynchronize(a->plan_stream);
        (void)hipEventDestroy(a->plan_fork);
        (void)hipEventDestroy(a->plan_join);
        (void)hipStreamDestroy(a->plan_stream);
    }
    if (a->act_stream) {
        (void)hipStreamSynchronize(a->act_stream);
        for (auto &ps : a->snap) {
            if (ps.params) (void)hipFree(ps.params);
            if (ps.fragF) (void)hipFree(ps.fragF);
            if (ps.on) (void)hipFree(ps.on);
            if (ps.gn) (void)hipFree(ps.gn);
            if (ps.ready) (void)hipEventDestroy(ps.ready);
        }
        (void)hipEventDestroy(a->act_done);
        (void)hipStreamDestroy(a->act_stream);
    }
    a->act_ws.release();
    a->plan.release();
    a->adam_tab.release();
    a->dw_part.release();
    a->dw_ticket.release();
    a->norm_plan.release();
    a->fwd_ws.release();
    a->pin.release();
    if (a->ev0) (void)hipEventDestroy(a->ev0);
    if (a->ev1) (void)hipEventDestroy(a->ev1);
    delete a;
}

}  // extern "C"
