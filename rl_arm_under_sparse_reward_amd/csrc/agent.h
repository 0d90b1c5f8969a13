// agent.h -- the DDPG learner's host-side state and the declarations its translation units share (library-internal, not
// part of the C ABI):
//   agent.hip           lifecycle, parameter / optimizer-state access, update sequences, the cycle hipGraph, C ABI
//   agent_engines.hip   the row-slab engines (slab8.h, slab32.h), weight-gradient launches (gemm_lds.h, dw64.h), optimizer
//                       and exchange kernels, forward-only entry points
//   agent_layers.hip    layer-per-launch fallback engine for network shapes the slab engines do not cover
// Reference: models.py:11-44, ddpg_agent.py:214-277, torch.optim.Adam (ddpg_agent.py:42-43).
//
// ---- HBM layout ------------------------------------------------------------------------
// Parameter "arena" (float32): [actor | critic], each  W1[H][K1] b1[H] W2[H][H] b2[H] W3[H][H]
// b3[H] W4[16][H] b4[16];  K1 = 32 for the actor, 48 for the critic, rows/cols beyond the real
// sizes are zero and stay zero (their gradients are exactly zero).  Gradients, Adam m, Adam v
// and the target networks use the same layout, so Adam and polyak are one elementwise pass.
// Network inputs are rows of 48 floats: [ x (obs+goal = 30) | 0 0 | a/max_action (4) | 0.. ]:
// the actor reads columns 0..31, the critic 0..47 (its W1 columns are permuted to match).
#pragma once
#include "internal.h"

#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// engine geometry the host logic needs (the kernels' own headers define the rest)
#define S8_AHEAD_WGS 4                 // slab8.h: spare workgroups that gather the next update's inputs (2 and 8 measured the same or slower)
#define S32_ROWS 32                    // slab32.h: rows of a slab
#define DW_PART (64 * 64 + 64)         // dw64.h: floats of one partial tile in the exchange buffer (accumulators | column sums of dY)

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
    int ks, part0;       // ks > 1: the reduction of every tile is split over ks workgroups (slice-major behind tile0) that meet
                         // through GemmGroup::part / ticket, entries part0 + tile (gemm_lds.h, ring path only)
    float max_action;    // EPI_BIAS_TANH
    int frag_layer;      // weight gradients: layer 1-4 of the network C belongs to (its fragment copies' offsets follow from the tile's
                         // coordinates, gemm_lds.h); 0 = unknown, look them up from the arena index
};

#define MAX_PROBS 8
#define GL_PART (32 * 32 + 32)   // floats of one partial tile in the split-reduction exchange (gemm_lds.h): the tile, then its column sums
#define GL_MAX_SPLIT_TILES 64
#define GL_MAX_KS 8            // most reduction slices per split tile (RLARM_DW_KSPLIT; hp_agent_create allocates GL_MAX_SPLIT_TILES * 8 partials)
struct GemmGroup {
    int n;
    int xcd;    // problems 0..3 have 8 x 8 tiles each and own one pair of XCDs (Launch::place_on_xcds)
    int uni;    // ring loop with the wave index in a scalar register (gemm_lds.h)
    int bias0;  // > 0: the bias gradients + their optimizer step run in workgroups of their own from this block index on (one per
                // 32-row panel of every problem that has a bias vector: gemm_lds.h gemm_bias_tile); the tiles then skip them
    int loss_wg; // 1: the LAST workgroup of the launch writes the loss log (gemm_lds.h gemm_loss_wg), workgroup 0 is a plain tile
    float *part;                  // split tiles: GL_PART floats per (tile, slice)
    unsigned long long *ticket;   // split tiles: arrival counter per tile, monotonic over the life of the agent
    GemmProb p[MAX_PROBS];
};

struct Pass {  // hidden activations of one forward pass
    float *h1, *h2, *h3;
};

struct AgentDevState {      // small device-resident scalars
    long long step;         // Adam step counter (both optimizers step together)
    long long n_logged;     // number of loss pairs written
    // per-step Adam scalars (torch computes them in Python doubles and narrows where used)
    float neg_step_actor, neg_step_critic, bc2_sqrt;
    unsigned open_timeouts; // k_cycle_open: hand-off polls that gave up (always 0; reported by hp_agent_get_losses)
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

// hand-off counters of the split launch (slab8_split.h): two sets (the launch of update u counts in set u % 2 and clears the
// other one for its successor) of SPLIT_COUNTERS counters x 8 XCD copies, SPLIT_CTR_STRIDE words apart, then the learner's
// sticky fault word
#define SPLIT_CTR_STRIDE 64          // 256 bytes: another memory channel
#define SPLIT_COUNTERS 8             // 0-2 stages of the critic chains, 3-5 gates of the actor-side chains, 6 actor-side chains done
#define SPLIT_CTR_NONE 15u           // "no counter" in the 4-bit per-problem tables
#define SPLIT_SET_WORDS (SPLIT_COUNTERS * 8 * SPLIT_CTR_STRIDE)
#define SPLIT_FAULT (2 * SPLIT_SET_WORDS)
#define SPLIT_SYNC_WORDS (SPLIT_FAULT + 4)

// sticky fault word of in-launch hand-offs: bit 0 = a poll gave up, and ONE BIT PER SOURCE above it (the words of several
// give-ups are OR-ed together, so the sources must not share bits): bit 4 = source 1, critic chains -> weight-gradient tiles;
// bit 5 = source 2, actor chains -> critic optimizer step; bit 6 = source 3, cycle-opening launch.  Mirrored into pinned host
// memory so that the next host call fails loudly.
__device__ __forceinline__ void handoff_fault(unsigned *fault, unsigned *fault_host, unsigned which) {
    const unsigned word = 1u | (1u << (3u + which));   // which in 1..3
    __hip_atomic_fetch_or(fault, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fault_host) __hip_atomic_fetch_or(fault_host, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait until *ctr >= need (agent scope); the verdict reaches every thread of the workgroup (contains barriers: call
// from uniform control flow).  flag: one int of LDS nobody else touches between the two barriers.
__device__ __forceinline__ bool handoff_wait(const unsigned *ctr, unsigned need, unsigned long long ticks, unsigned *fault,
                                             unsigned *fault_host, unsigned which, int *flag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = wall_clock64();
        // Adaptive poll: while NOBODY has counted yet the producers are microseconds away -- look every ~1.7 us; once the first
        // count is in, the rest follow within a microsecond or two -- look every ~0.15 us.  (Hundreds of workgroups polling
        // every 0.3 us slowed the chains they were waiting for by 2 us: each poll is a round trip to the memory side.)
        for (;;) {
            const unsigned seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= need) break;
            if (seen == 0u) __builtin_amdgcn_s_sleep(64);
            else __builtin_amdgcn_s_sleep(6);
            if (wall_clock64() - t0 > ticks) {
                handoff_fault(fault, fault_host, which);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}
#define LOSS_LOG 4096

#include "slab_common.h"
#include "peer.h"

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
    float *QT2 = nullptr;                // split launch (slab8_split.h): Q' of the next update's minibatch (ping-pong with QT)
    float *dA3 = nullptr, *dA2 = nullptr, *dA1 = nullptr;  // critic-loss path
    float *dP3 = nullptr, *dP2 = nullptr, *dP1 = nullptr, *dXP = nullptr;  // actor-loss path through the critic
    float *dZ = nullptr, *dK3 = nullptr, *dK2 = nullptr, *dK1 = nullptr;   // actor
    float *loss_log = nullptr;
    AgentDevState *d_state = nullptr;
    // row-slab engine: fragment-ordered weight copies (online forward / online dX / target forward), loss partials
    float *fragF = nullptr, *fragD = nullptr, *fragFT = nullptr, *part = nullptr;
    unsigned long long *timeline = nullptr;   // debug builds (SLAB_TIMELINE) stamp stage boundaries here
    bool slab = true;      // a row-slab engine (false: layer-per-launch engine)
    bool slab8 = true;     // thin slabs on the 4x4x1 MFMA (false: slab32)
    bool slab32 = false;   // 32-row slabs on the 32x32x2 MFMA, forward + backward in one kernel (slab32.h: large batches)
    int s8_rows = 4;       // slab height of that engine: 4 rows up to batch 448, 8 up to 1280, 16 beyond (RLARM_SLAB_ROWS overrides)
    bool fuse_adam_ok = true;   // Adam in the weight-gradient GEMM's epilogue (RLARM_FUSE_ADAM=0: separate launch, for A/B)
    // large-minibatch weight gradients (dw64.h): 64 x 64 tiles, batch rows split over dw_S workgroups per tile
    bool dw64 = false;                   // RLARM_DW64: default from batch 1536
    int dw_S = 3;                        // (RLARM_DW64=s<n>)
    DevBuf dw_part, dw_ticket;           // partial tiles / arrival counters
    bool keep_grads_dbg = false;   // RLARM_KEEP_GRADS=1 (parity tests): the peer optimizer kernels also write the summed gradients out
    bool upd_graph_ok = true;   // hp_agent_sample_and_update replays cached graphs (RLARM_UPDATE_GRAPH=0: eager launches, for A/B)
    bool gather_ahead = true;   // merged kernel: spare workgroups gather update u+1's inputs during update u (while CUs are free)
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
    unsigned *open_sync = nullptr;       // k_cycle_open's flags (cycle_open.hip)
    int loss_wg = 1;                     // the loss log is written by a workgroup of its own behind the weight-gradient tiles (gemm_lds.h gemm_loss_wg)
    int dw_ksplit = 0;                   // reduction slices of the narrow weight-gradient problems (RLARM_DW_KSPLIT; 0/1: none)
    float *gl_part = nullptr;            // their partial tiles and arrival counters (gemm_lds.h)
    unsigned long long *gl_ticket = nullptr;
    // split launch (slab8_split.h): target chains one update ahead, the critic's weight gradients + optimizer step inside the
    // chain launch.  RLARM_SPLIT: unset = where it fits and the sequence has at least SPLIT_MIN_UPDATES updates, 0 = never,
    // 1 = wherever it fits (single updates too: parity tests)
    int split_mode = -1;                 // RLARM_SPLIT=0|1: never / also for short sequences (default: from SPLIT_MIN_UPDATES updates)
    unsigned *split_reset_pending = nullptr;   // a split launch whose tiles wrote gradients only went out: the next optimizer launch clears its counter set
    unsigned *k1_sync = nullptr;         // device: hand-off counters of the split launch, then the sticky fault word (SPLIT_FAULT)
    unsigned *fault_host = nullptr;      // pinned + mapped mirror of the fault word (agent_check_fault), and its device address
    unsigned *fault_host_dev = nullptr;
    bool cycle_open = true;              // RLARM_CYCLE_OPEN=0: slots / scatter / plans / normalizer as separate launches
    bool g_open = false;
    void *g_slots = nullptr;
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


// ------------------------------------------------------------------------------- host-side helpers
static inline AdamCfg adam_cfg(const hp_agent *a) {
    return AdamCfg{a->cfg.lr_actor, a->cfg.lr_critic, a->cfg.adam_beta1, a->cfg.adam_beta2, a->cfg.adam_eps};
}

static inline NetLayout make_layout(int K1, int H) {
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

static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct Launch {  // builds one grouped launch
    GemmGroup g;
    int tiles = 0;
    Launch() {
        g.n = 0;
        g.xcd = 0;
        g.uni = 0;
        g.bias0 = 0;
        g.loss_wg = 0;
        g.part = nullptr;
        g.ticket = nullptr;
    }
    int split_tiles = 0;   // tiles whose reduction is split (entries of part / ticket in use)
    // first tiles of the problems, 16 bits each (gemm_lds.h: TileHead -- leading scalar kernel arguments)
    unsigned long long head(int half) const {
        unsigned long long w = 0ull;
        for (int i = 0; i < 4; ++i) {
            const int pi = 4 * half + i;
            w |= (unsigned long long)(pi < g.n ? (g.p[pi].tile0 & 0xffff) : 0xffff) << (16 * i);
        }
        return w;
    }
    // bias vectors in workgroups of their own behind the tiles (call last); returns how many
    int separate_bias() {
        int nb = 0;
        for (int i = 0; i < g.n; ++i)
            if (g.p[i].bias_grad) nb += (g.p[i].M + 31) / 32;
        g.bias0 = tiles;
        return nb;
    }
    // split the reduction of the problem added last over `ks` workgroups per tile
    void split_last(int ks) {
        GemmProb &p = g.p[g.n - 1];
        const int nt = tiles - p.tile0;
        p.ks = ks;
        p.part0 = split_tiles;
        split_tiles += nt;
        tiles += (ks - 1) * nt;
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
        p.ks = 1;
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

// forward layer Y = act(X W^T + b)
static inline void add_fwd(Launch &L, const float *X, int ldx, int K, const float *W, const float *bias, float *Y, int ldy,
                    int M, int N, int epi) {
    GemmProb &p = L.add(M, N, K);
    p.A = X; p.a_si = ldx; p.a_sk = 1;
    p.B = W; p.b_sj = K; p.b_sk = 1;
    p.C = Y; p.ldc = ldy;
    p.bias = bias;
    p.epi = epi;
}

// dX = (dY W) * relu'(gate)
static inline void add_dx(Launch &L, const float *dY, int ldy, int Nout, const float *W, int Kin, float *dX, int lddx, int M,
                   const float *gate, int ldgate) {
    GemmProb &p = L.add(M, Kin, Nout);
    p.A = dY; p.a_si = ldy; p.a_sk = 1;
    p.B = W; p.b_sj = 1; p.b_sk = Kin;
    p.C = dX; p.ldc = lddx;
    p.mask = gate; p.ldmask = ldgate;
    p.epi = gate ? EPI_MASK : EPI_NONE;
}

// dW = dY^T X, db = column sums of dY
static inline void add_dw(Launch &L, const float *dY, int ldy, int Nout, const float *X, int ldx, int Kin, float *dW,
                   float *db, int Mrows, int layer = 0) {
    GemmProb &p = L.add(Nout, Kin, Mrows);
    p.frag_layer = layer;
    p.A = dY; p.a_si = 1; p.a_sk = ldy;
    p.B = X; p.b_sj = 1; p.b_sk = ldx;
    p.C = dW; p.ldc = Kin;
    p.bias_grad = db;
    p.epi = EPI_NONE;
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
    // full chain launch: the plan draw (next_plan) and the look-ahead gather (ahead_plan) ride in the weight-gradient
    // launch instead of the chain kernel
    bool ride_in_dw = false;
    // data-parallel ranks exchanging through peer memory: this update's gradients go straight into the exchange buffer
    float *grads_out = nullptr;
    // last update of a training cycle: the optimizer launch also applies the soft update of both target networks
    // (ddpg_agent.py:149-150) to the parameters it has just stepped -- no separate polyak launch
    bool polyak_after = false;
    // split launch (slab8_split.h): this update's Q' was computed one launch ahead (set qset of QT / QT2); t_plan = plan of the
    // NEXT update, whose Q' the target chains of this launch compute into the other set (nullptr: last update of the sequence)
    bool split = false;
    int qset = 0;
    // data-parallel ranks, tile-wise exchange: index of this update in its sequence (the exchange epoch), -1: not this form
    int peer_u = -1;
    const PlanRec *t_plan = nullptr;
};
// what the in-launch weight-gradient tiles of k_fb_split8 do behind their products (slab8_split.h: one instantiation per form)
enum { SPLIT_TILES_ADAM = 0,   // single rank: optimizer step of the critic inside the launch (behind the actor-side chains' gates)
       SPLIT_TILES_PEER = 1,   // data-parallel ranks, one device each: rank exchange tile by tile, then the same step (utils.py:43-48 + Adam)
       SPLIT_TILES_GRADS = 2 };// gradients only: exchange (RCCL, two-phase / gated peer memory) + optimizer follow as launches of their own
#define SPLIT_MIN_UPDATES 12  // shorter sequences keep the two-launch form: the prologue launch costs ~17 us per sequence, an update gains ~1.4

// workgroups of the chain kernel that carry chains (the spare ones -- index plan, look-ahead gather, L2 warmers -- follow)
static inline int chain_wgs(const hp_agent *a) { return 2 * (a->Mp / a->s8_rows); }

// ---- defined in agent_engines.hip
int launch_group(hp_agent *a, const Launch &L, int which);                       // one grouped k_gemm_lds launch
int enqueue_gather(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, const PlanRec *plan, double sq, int xset = 0,
                   hipStream_t stream = nullptr);
int enqueue_relayout(hp_agent *a, bool targets);
Launch build_dw_group(const hp_agent *a, const float *sXA, const float *sXP, float *grads = nullptr);
// forwards + losses + backwards of one update.  Inputs: gc == nullptr -> already in XA/XP/XT/R (minibatch API), else sampled
// (HER gather fused into the chain kernel / k_gather_fused).  fuse_adam: the caller wants the optimizer step applied too;
// the slab engines then do it in the weight-gradient launch's epilogue and the caller must NOT enqueue Adam again (*fused).
int enqueue_forward_backward(hp_agent *a, const GatherCtx *gc = nullptr, bool fuse_adam = false, bool *fused = nullptr);
// only = 1 / 2: just the chain kernel / just the weight-gradient launch (timing diagnostics, hp_agent_debug_chain)
int enqueue_forward_backward_slab(hp_agent *a, const GatherCtx *gc, bool fuse_adam, int only = 0);
// split launch: does this agent's shape fit it (4-row slabs, three kinds of chains + spare workgroups on the CUs)?
bool split_fits(const hp_agent *a);
bool split_fits_rows(const hp_agent *a, int rows);
// ... and its prologue: the target chains of a sequence's FIRST update (plan = that update's index plan) into Q' set 0
int enqueue_split_prologue(hp_agent *a, const GatherCtx *gc, int tiles_mode);   // tiles_mode: SPLIT_TILES_* of the sequence's updates
// a bounded in-launch hand-off gave up earlier (k_cycle_open, k_fb_split8): HP_ERR_STATE + message; free for the host
int agent_check_fault(const hp_agent *a, const char *who);
int enqueue_adam(hp_agent *a, bool polyak_after = false);   // polyak_after: see GatherCtx (slab engines; returns whether via *folded)
int enqueue_polyak(hp_agent *a);
// cycle_open.hip: slots + scatter + normalizer update + first minibatch plans of a cycle as one launch
bool cycle_open_fits(const hp_agent *a, int64_t n_new);
int cycle_open_launch(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, PlanRec *norm_plan, int n_first,
                      double future_p, bool recompute);
// utils.sync_grads (utils.py:43-48) + both Adam steps of update u as the peer exchange's optimizer kernel(s) (peer.hip)
int enqueue_peer_adam(hp_agent *a, int u, bool polyak_after = false);
// data-parallel ranks: do the weight-gradient tiles exchange by themselves (one launch: gradients + rank exchange + optimizer)?
// (a launch with more tiles than flag rows takes the separate exchange + optimizer kernel instead of failing)
static inline bool peer_tiles_ok(const hp_agent *a) {
    return a->peer && a->peer->tiles && a->peer->phases == 1 && !a->peer->gate && a->slab && !a->dw64 && a->fuse_adam_ok &&
           build_dw_group(a, a->XA, a->XP).tiles <= HP_PEER_TILES;
}
// can the optimizer launches of this agent apply the soft target update themselves?  (slab engines: yes)
static inline bool polyak_foldable(const hp_agent *a) { return a->slab; }
// ---- defined in agent_layers.hip
int enqueue_forward_backward_layers(hp_agent *a);
int layers_enqueue_adam(hp_agent *a);
int layers_enqueue_polyak(hp_agent *a);
