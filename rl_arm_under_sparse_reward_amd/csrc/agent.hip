// agent.hip -- the DDPG learner's lifecycle and control flow on gfx950: parameter / optimizer-state access in the reference's
// flat order, sequences of sampled updates (index plans drawn ahead, look-ahead gathers, exchange + optimizer), the training
// cycle as ONE cached hipGraph, and the C ABI around them.  The kernels live in agent_engines.hip / agent_layers.hip.
// Reference: ddpg_agent.py:143-150 (cycle), :225-277 (_update_network), utils.py:6-69 (exchange), torch.optim.Adam.
#include "agent.h"

#include <functional>

static void drop_graph(hp_agent *a);

// A bounded in-launch hand-off gave up in an earlier launch (k_cycle_open's flags, k_fb_split8's counters): work was skipped, so
// buffer / normalizer / parameters are no longer what the reference would hold.  The word is sticky and mirrored into pinned
// host memory by the kernel that gave up: reading it costs the host nothing, and every entry point that enqueues more work on
// this learner checks it first (as peer_check_alive does for the rank exchange).
int agent_check_fault(const hp_agent *a, const char *who) {
    const unsigned w = a->fault_host ? *(volatile const unsigned *)a->fault_host : 0u;
    if (w == 0u) return HP_OK;
    hp_set_error("%s: an in-launch hand-off gave up earlier (%s%s%s; word 0x%x): the launches since skipped work and the "
                 "learner's state is not valid -- recreate the agent", who, (w & 0x10u) ? "critic chains -> weight-gradient tiles; " : "",
                 (w & 0x20u) ? "actor chains -> critic optimizer step; " : "", (w & 0x40u) ? "cycle-opening launch; " : "", w);
    return HP_ERR_STATE;
}

static int ensure_plan(hp_agent *a, int n_batches) {
    if (n_batches > a->plan_batches) {
        // the cached cycle graph has the plan's address baked into its draw / gather / ride-along kernels: growing the
        // plan frees that memory, so the graph goes with it (rebuilt by the next hp_agent_train_cycle)
        drop_graph(a);
        HP_TRY(a->plan.ensure((size_t)n_batches * a->B * sizeof(PlanRec)));
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
// Cycle mode (`cyc` != nullptr, hp_agent_train_cycle): the first plans are drawn TOGETHER with the normalizer's index plan
// (one launch, k_draw_plan2), `cyc->between` enqueues the normalizer update behind that launch, and the last update's optimizer
// launch also applies the soft target update where the engine can (cyc->polyak_folded tells the caller).
struct CycleOpts {
    PlanRec *norm_plan;
    bool open;                       // slots, scatter, normalizer update and the draws as ONE launch (cycle_open.hip)
    bool recompute;                  // single rank: recompute_stats inside the normalizer update
    std::function<int()> between;    // what follows the draws and precedes the first update
    bool polyak_folded = false;
};
// Does a sequence of n_updates sampled updates WITH optimizer steps on this agent take the split form (slab8_split.h)?  ONE
// predicate for the launch logic (enqueue_updates: `ride` = with_adam on a slab engine is the only further condition, the others
// follow from the terms below) and for what hp_agent_update_form reports.
static bool update_takes_split_form(const hp_agent *a, int n_updates) {
    const bool full = a->slab8 && chain_wgs(a) + 1 + S8_AHEAD_WGS > a->ctx->cu_count;   // no CU left for the chain kernel's spare workgroups
    return a->slab8 && a->gather_ahead && !full && split_fits(a) &&
           (a->split_mode >= 0 ? a->split_mode == 1 : n_updates >= SPLIT_MIN_UPDATES);
}

static int enqueue_updates(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p, double sq,
                           int n_updates, bool with_adam, CycleOpts *cyc = nullptr) {
    // slab engine: only the first minibatch's indices are drawn up front; update u draws the plan of update u+1 in
    // a spare workgroup of its backward kernel (same stream order of draws, so the same indices).  Layer engine:
    // one kernel draws all of them (nothing else consumes the stream in between, like the reference's inner loop).
    const bool ride = a->slab && with_adam;
    // merged slab8 kernel: plans are drawn TWO updates ahead so that spare workgroups of update u can gather the inputs
    // of update u+1 from a plan that an earlier launch finished (the order of draws in the stream is unchanged)
    const int chains = chain_wgs(a);
    // ... and when the launch is full (no CU for spare workgroups) both jobs move out of the chain kernel
    const bool full = a->slab8 && chains + 1 + S8_AHEAD_WGS > a->ctx->cu_count;
    // slab8 engine, full launch: both jobs ride in the weight-gradient launch (k_gemm_lds_adam_ride).  (A second stream beside
    // the chain kernel was the alternative until round 4: its cross-queue graph edges cost ~4 us each, 65.5 vs 71.4 us at batch
    // 1024, and at batch 4096 k_draw_plan took 83 us instead of 3.4 next to the chain kernel; removed in round 5.)
    // slab32 engine: its chain kernel never gathers and its launches fill the CUs from batch 4096, so both jobs ride in the
    // weight-gradient launch as well.  Profiling brackets every launch with events: the gather then runs in front of every launch.
    const bool s32_ride = ride && a->slab32 && !a->prof;
    const bool s32_serial = a->slab32 && !s32_ride;
    const bool dw_ride = s32_ride || (ride && a->slab8 && full);
    const bool ahead = ride && ((a->slab8 && (a->gather_ahead || dw_ride)) || s32_ride);
    const int lead = ahead ? 2 : 1;
    // split form (slab8_split.h): target chains one update ahead, the critic's weight gradients + optimizer step inside the chain
    // launch.  Needs the chain kernel's own look-ahead (plans two updates ahead, gather workgroups) and the fused optimizer.
    const bool split = ride && ahead && !dw_ride && update_takes_split_form(a, n_updates);
    HP_KLOG("#open");
    {
        ProfScope ps(a, PROF_PLAN);
        const int first = ride ? (n_updates < lead ? n_updates : lead) : n_updates;
        if (cyc && cyc->open)
            HP_TRY(cycle_open_launch(a, b, on, gn, rng, cyc->norm_plan, first, future_p, cyc->recompute));
        else if (cyc)   // ddpg_agent._update_normalizer's plan (T transitions of the staged episodes), then the first minibatch plans
            HP_TRY(rng_launch_plan2(rng, b->staged_n, b->T, b->T, cyc->norm_plan, b->d_meta, a->B, first, future_p,
                                    a->plan.as<PlanRec>()));
        else
            HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, a->B, first, future_p, a->plan.as<PlanRec>()));
    }
    if (cyc && cyc->between) HP_TRY(cyc->between());
    const bool fold = cyc && with_adam && polyak_foldable(a);
    if (cyc) cyc->polyak_folded = fold;
    // split form: Q' of the FIRST update's minibatch comes from a prologue launch of target chains (every later one from the
    // launch before).  Round 4 also built the alternative -- the first update's critic chains run k_fb_slab8's whole critic side
    // and hand off to the tiles at their end, no prologue: 37.90 vs 37.92 us/update over cycles, 40.97 vs 41.14 in the driver's
    // 20-step form, and an intermittent optimizer mismatch in the teacher-forced test that the prologue form never showed;
    // removed.
    if (split) {
        HP_KLOG("#prologue");
        GatherCtx g0{b, on, gn, a->plan.as<PlanRec>(), sq};
        const bool fuse = with_adam && !a->comm && (!a->peer || peer_tiles_ok(a));   // (as for the updates below)
        HP_TRY(enqueue_split_prologue(a, &g0, !fuse ? SPLIT_TILES_GRADS : (a->peer ? SPLIT_TILES_PEER : SPLIT_TILES_ADAM)));
    }
    for (int u = 0; u < n_updates; ++u) {
        HP_KLOG("#update");
        GatherCtx gc{b, on, gn, a->plan.as<PlanRec>() + (size_t)u * a->B, sq};
        gc.split = split;
        gc.qset = u & 1;
        gc.t_plan = (split && u + 1 < n_updates) ? a->plan.as<PlanRec>() + (size_t)(u + 1) * a->B : nullptr;
        if (ride && u + lead < n_updates) {
            PlanRec *next = a->plan.as<PlanRec>() + (size_t)(u + lead) * a->B;
            if (a->slab32 && s32_serial && 2 * (a->Mp / S32_ROWS) >= a->ctx->cu_count) {
                // serial mode (profiling) of a launch that fills the CUs: a spare workgroup would only start
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
            if (u + 1 < n_updates) gc.ahead_plan = a->plan.as<PlanRec>() + (size_t)(u + 1) * a->B;
        }
        gc.ride_in_dw = dw_ride;
        const bool last_fold = fold && u == n_updates - 1;
        gc.polyak_after = last_fold;
        const bool peer_tiles = with_adam && peer_tiles_ok(a);
        const bool via_peer = with_adam && a->peer != nullptr && !peer_tiles;
        if (via_peer) gc.grads_out = peer_grad_buffer(a->peer, u + 1);   // epoch base is even: parity of epoch base + u + 1
        if (peer_tiles) gc.peer_u = u;   // the weight-gradient tiles exchange and step by themselves (gemm_lds.h PEER)
        bool fused = false;
        HP_TRY(enqueue_forward_backward(a, &gc, with_adam && !a->comm && (!a->peer || peer_tiles), &fused));
        HP_REQUIRE(!peer_tiles || fused, HP_ERR_STATE, "tile-wise exchange: the engine did not take the fused optimizer path");
        if (via_peer) {
            // utils.sync_grads (utils.py:43-48) + both Adam steps in ONE kernel: every rank reads the peers' gradient
            // vectors over xGMI, sums them in rank order and steps (peer.hip)
            HP_TRY(enqueue_peer_adam(a, u, last_fold));
        } else if (with_adam) {
            // utils.sync_grads (utils.py:43-48): SUM over ranks between backward and the optimizer step; one
            // all-reduce covers both networks (the reference sends the actor's and the critic's separately)
            if (a->comm)
                HP_TRY(a->grad_mean ? comm_allreduce_mean_f32(a->comm, a->grads, (size_t)a->n_arena)
                                    : comm_allreduce_sum_f32(a->comm, a->grads, (size_t)a->n_arena));
            if (!fused) HP_TRY(enqueue_adam(a, last_fold));
        }
    }
    HP_KLOG("#close");
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
    A(&a->QT, Mp * 16); A(&a->QT2, Mp * 16); A(&a->QA, Mp * 16); A(&a->QP, Mp * 16); A(&a->dQA, Mp * 16); A(&a->dQP, Mp * 16);
    A(&a->dA3, Mp * H); A(&a->dA2, Mp * H); A(&a->dA1, Mp * H);
    A(&a->dP3, Mp * H); A(&a->dP2, Mp * H); A(&a->dP1, Mp * H); A(&a->dXP, Mp * ldx);
    A(&a->dZ, Mp * 16); A(&a->dK3, Mp * H); A(&a->dK2, Mp * H); A(&a->dK1, Mp * H);
    A(&a->loss_log, LOSS_LOG * 2);
    A(&a->fragF, a->n_arena); A(&a->fragD, a->n_arena); A(&a->fragFT, a->n_arena); A(&a->part, 3 * (Mp / 4));
    {
        // RLARM_ENGINE = slab8 | slab32 | layers overrides the table below (A/B runs, debugging)
        const char *e = getenv("RLARM_ENGINE");
        // the slab engines are specialised: 256-wide hidden layers, network inputs of at most 48 columns (obs + goal +
        // action, padded to 16) and at most 4 action components; any other shape takes the layer-per-launch engine
        const bool slab_shape = a->H == 256 && a->ldx <= 48 && cfg->act_dim <= 4;
        a->slab = !(e && strcmp(e, "layers") == 0) && slab_shape;
        // Thin slabs (4x4x1 MFMA, slab8.h) while their chains fit the CUs in one round (16-row slabs: batch <= 2048), 32-row
        // slabs on the 32x32x2 MFMA (slab32.h) beyond.  Measured us/update, slab8 / slab32 (and the 16-row two-kernel engine
        // on the 16x16x4 MFMA that round 3 removed; profiles/r02_large_batch_engines.txt):
        //   1024: 59.5 / 106.8 (88.0)    1536: 87.7 / 113.2 (119.5)    2048: 97.4 / 119.8 (127.5)
        //   3072: 163.3 / 132.9 (173.6)  4096: 180.1 / 146.7 (202.7)
        const int cus_e = a->ctx->cu_count > 0 ? a->ctx->cu_count : 256;
        const bool thin_fits = 2 * (a->Mp / 16) <= cus_e;
        a->slab8 = a->slab && (e && (strcmp(e, "slab8") == 0 || strcmp(e, "slab32") == 0) ? strcmp(e, "slab8") == 0 : thin_fits);
        a->slab32 = a->slab && !a->slab8;
        // Thin slabs buy latency at small batches (more CUs busy, less matrix work per streamed weight block) and cost
        // L2 weight traffic per row.  The kernel's LDS footprint allows one workgroup per CU, so a launch with more chain
        // workgroups than CUs runs in two rounds: the rule is "the thinnest slab whose 2 * B / rows chains fit the CUs".
        // Measured, us/update (tools/ubench/rows_sweep2.sh, sweep_rows.sh): 4 vs 8 rows 41.4 vs 45 at batch 256, 46.8 vs 48.6
        // at 448, 47.5 vs 49.6 at 480, 48.4 vs 49.9 at 512 (256 chains: the look-ahead then rides in the weight-gradient
        // launch), 73.4 vs 52.0 at 544 (two rounds); 8 vs 16 rows 59.1 vs 81.4 at 1024, 98.3 vs 90.9 at 1280.
        const int cus = a->ctx->cu_count > 0 ? a->ctx->cu_count : 256;
        auto tri = [](const char *name) { const char *e = getenv(name); return e ? (e[0] != '0' ? 1 : 0) : -1; };
        a->s8_rows = 2 * (a->Mp / 4) <= cus ? 4 : (2 * (a->Mp / 8) <= cus ? 8 : 16);
        // (Round 4 also compiled the split launch for 8-row slabs and took them where three kinds of 4-row chains do not fit the CUs:
        // batch 384 k8: 44.7 vs 41.1 us/update with 4-row slabs in two launches, 512 k8: 45.6 vs 44.5, 640: 47.4 vs 47.4 -- an 8-row
        // layer is matrix-issue bound at 2.6-3.0 us, so the actor side gets longer, not shorter.  Removed)
        if (const char *sr = getenv("RLARM_SLAB_ROWS")) {
            if (strcmp(sr, "4") == 0) a->s8_rows = 4;
            if (strcmp(sr, "8") == 0) a->s8_rows = 8;
            if (strcmp(sr, "16") == 0) a->s8_rows = 16;
        }
        const char *fa = getenv("RLARM_FUSE_ADAM");
        a->fuse_adam_ok = !(fa && fa[0] == '0');
        a->upd_graph_ok = tri("RLARM_UPDATE_GRAPH") != 0;
        a->keep_grads_dbg = tri("RLARM_KEEP_GRADS") == 1;
        a->cycle_open = tri("RLARM_CYCLE_OPEN") != 0;
        a->split_mode = tri("RLARM_SPLIT");
        // weight gradients: 64 x 64 tiles with split batch rows (dw64.h) where the 32 x 32 tiles are L2-bound
        // (us/update, 32 x 32 tiles vs dw64: 56.6 / 58.9 at batch 1024, 85.8 / 85.1 at 1536, 93.9 / 92.2 at 2048, 146 / 128 at 4096)
        // RLARM_DW64 = 0 | 1 | s<n>: never / always (3 slices of the batch rows per tile) / always with n slices (1..16)
        a->dw64 = a->slab && (tri("RLARM_DW64") >= 0 ? tri("RLARM_DW64") == 1 : a->Mp >= 1536);
        if (const char *ds = getenv("RLARM_DW64"))
            if (ds[0] == 's' && atoi(ds + 1) > 0 && atoi(ds + 1) <= 16) a->dw_S = atoi(ds + 1);
        // the chain kernel's spare workgroups (index plan two updates ahead, gather of the next minibatch) only pay while they find
        // free CUs next to the chains: at batch 1024 (256 chains) they ran after them, 87.6 vs 76.1 us/update -- both jobs then
        // ride in the weight-gradient launch (agent.hip: enqueue_updates)
        a->gather_ahead = chain_wgs(a) + 1 + S8_AHEAD_WGS <= cus;
    }
    if (st == HP_OK) st = dev_alloc(a, &a->d_state, 1);
    if (st == HP_OK) st = dev_alloc(a, &a->open_sync, 4);
    if (st == HP_OK) st = dev_alloc(a, &a->k1_sync, SPLIT_SYNC_WORDS);
    if (st == HP_OK) {   // sticky fault word of the in-launch hand-offs, mirrored where the host can read it without a sync
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            hp_set_error("hp_agent_create: pinned fault word: %s", hipGetErrorString(hipGetLastError()));
            st = HP_ERR_HIP;
        } else {
            memset(hp, 0, 64);
            a->fault_host = static_cast<unsigned *>(hp);
            a->fault_host_dev = static_cast<unsigned *>(dp);
        }
    }
    if (st == HP_OK && a->slab) {   // split narrow weight-gradient tiles (gemm_lds.h): 4 slices from 768 batch rows (us/update with
        // 1 / 2 / 4 / 8 slices: 58.6 / 57.1 / 55.6 / 60.0 at batch 1024; 47.7 / 48.2 / 48.2 / 52.4 at 512 k8; 40.6 / 41.9 / 42.7 / - at 256;
        // 1 vs 4 slices: 54.2 / 53.1 at 768, 55.6 / 54.1 at 896, 82.1 / 78.9 at 1152, 83.6 / 79.2 at 1280)
        const char *e = getenv("RLARM_DW_KSPLIT");
        a->dw_ksplit = e ? atoi(e) : (a->Mp >= 768 ? 4 : 1);
        if (a->dw_ksplit < 1 || a->dw_ksplit > GL_MAX_KS) a->dw_ksplit = 1;
        const int H = a->H;
        const int narrow = 2 * ((H + 31) / 32) + ((H + 31) / 32) * (((a->lc.K1 + 31) / 32) + ((a->la.K1 + 31) / 32));
        if (narrow <= GL_MAX_SPLIT_TILES) {
            st = dev_alloc(a, &a->gl_part, (size_t)GL_MAX_SPLIT_TILES * GL_MAX_KS * GL_PART);
            if (st == HP_OK) st = dev_alloc(a, &a->gl_ticket, GL_MAX_SPLIT_TILES);
        }
    }
    if (st == HP_OK) st = dev_alloc(a, &a->timeline, 192);
    if (st == HP_OK && a->dw64) {
        Launch L = build_dw_group(a, a->XA, a->XP);
        size_t tiles = 0;
        for (int i = 0; i < L.g.n; ++i) tiles += (size_t)((L.g.p[i].M + 63) / 64) * ((L.g.p[i].N + 63) / 64);
        if (a->dw_part.ensure(tiles * a->dw_S * DW_PART * sizeof(float)) != HP_OK || a->dw_ticket.ensure(tiles * sizeof(unsigned long long)) != HP_OK ||
            hipMemsetAsync(a->dw_ticket.p, 0, tiles * sizeof(unsigned long long), a->ctx->stream) != hipSuccess)
            st = HP_ERR_HIP;
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
    HP_TRY(agent_check_fault(a, "hp_agent_sample_and_update"));
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
    HP_TRY(agent_check_fault(a, "hp_agent_get_losses"));
    HP_REQUIRE(h.open_timeouts == 0u, HP_ERR_STATE, "hp_agent_get_losses: %u hand-off polls of the cycle-opening launch gave up "
               "(k_cycle_open): the cycles since are not valid", h.open_timeouts);
    HP_REQUIRE(h.n_logged >= n_last, HP_ERR_STATE, "hp_agent_get_losses: only %lld updates logged", h.n_logged);
    for (int i = 0; i < n_last; ++i) {
        const long long k = h.n_logged - n_last + i;
        out_host[2 * i] = log[(k % LOSS_LOG) * 2];
        out_host[2 * i + 1] = log[(k % LOSS_LOG) * 2 + 1];
    }
    return HP_OK;
}

int hp_agent_status(hp_agent *a, uint32_t *fault) {
    HP_REQUIRE(a && fault, HP_ERR_INVALID, "hp_agent_status: null argument");
    *fault = a->fault_host ? *(volatile const unsigned *)a->fault_host : 0u;
    return HP_OK;
}

int hp_agent_soft_update(hp_agent *a) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_soft_update: null handle");
    HP_SERIALISE(a);
    return enqueue_polyak(a);
}

}  // extern "C" (re-opened below)

// device part of one cycle after the episodes are staged.  open: slots + scatter are part of it (k_cycle_open); otherwise they
// happened in buffer_stage_and_store
static int enqueue_cycle_tail(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p,
                              double sq, int n_batches, PlanRec *norm_plan, bool open) {
    // ddpg_agent._update_normalizer (:187-212) behind the merged index draw, then the updates (:145-147), then the soft target
    // update (:149-150) -- inside the last optimizer launch where the engine can, as its own launch otherwise
    CycleOpts cyc;
    cyc.norm_plan = norm_plan;
    cyc.open = open;
    cyc.recompute = !a->comm && !a->peer;   // single rank: update + recompute_stats of both normalizers in one launch
    cyc.between = [=]() -> int {
        if (!open) HP_TRY(norm_launch_update_from_plan(on, gn, b, norm_plan, b->T, a->cfg.clip_obs, !a->comm && !a->peer));
        if (!a->comm && !a->peer) {
        } else if (a->peer) {
            HP_TRY(norm_launch_begin(on));
            HP_TRY(norm_launch_begin(gn));
            // normalizer._mpi_average (normalizer.py:60-64) through the mailboxes
            HP_TRY(peer_allreduce_small(a->peer, on->d->sync, (size_t)(2 * on->size + 1), true));
            HP_TRY(peer_allreduce_small(a->peer, gn->d->sync, (size_t)(2 * gn->size + 1), true));
            HP_TRY(norm_launch_end(on));
            HP_TRY(norm_launch_end(gn));
        } else {
            HP_TRY(norm_launch_begin(on));
            HP_TRY(norm_launch_begin(gn));
            // normalizer._mpi_average (normalizer.py:60-64) on sum | sumsq | count of each normalizer
            HP_TRY(comm_allreduce_mean_f32(a->comm, on->d->sync, (size_t)(2 * on->size + 1)));
            HP_TRY(comm_allreduce_mean_f32(a->comm, gn->d->sync, (size_t)(2 * gn->size + 1)));
            HP_TRY(norm_launch_end(on));
            HP_TRY(norm_launch_end(gn));
        }
        return HP_OK;
    };
    // ddpg_agent.py:145-150
    HP_TRY(enqueue_updates(a, b, on, gn, rng, future_p, sq, n_batches, true, &cyc));
    if (!cyc.polyak_folded) HP_TRY(enqueue_polyak(a));
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

// everything of a cycle behind the staging of its episodes: one cached graph
static int train_cycle_staged(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, int64_t n_new, double future_p,
                              double sq_threshold, int32_t n_batches, bool open);

static int train_cycle_checks(hp_agent *a, hp_buffer *b, int64_t n_new, int32_t n_batches, const char *who) {
    HP_REQUIRE(n_new > 0 && n_batches > 0, HP_ERR_INVALID, "%s: n_new and n_batches must be positive", who);
    HP_REQUIRE(!(b->current_size == 0 && n_new > b->size), HP_ERR_INVALID, "high <= 0");
    HP_REQUIRE(!a->prof, HP_ERR_STATE, "%s: profiling mode uses the eager path (hp_agent_profile(0) first)", who);
    HP_TRY(agent_check_fault(a, who));
    return peer_check_alive(a->peer, who);
}

int hp_agent_train_cycle(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, const double *obs,
                         const double *ag_host, const double *g, const double *actions, int64_t n_new,
                         double future_p, double sq_threshold, int32_t n_batches) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_train_cycle"));
    HP_SERIALISE(a);
    HP_REQUIRE(obs && ag_host && g && actions, HP_ERR_INVALID, "hp_agent_train_cycle: null episode array");
    HP_TRY(train_cycle_checks(a, b, n_new, n_batches, "hp_agent_train_cycle"));
    // 1. episodes -> pinned -> device staging (eager: the source pointers change per call); slots and scatter: part of the
    // opening launch of the graph, or eager launches as well when that launch is switched off / would not fit the CUs
    const bool open = a->cycle_open && cycle_open_fits(a, n_new);
    if (open) HP_TRY(buffer_stage_for_cycle(b, obs, ag_host, g, actions, n_new));
    else HP_TRY(buffer_stage_and_store(b, rng, obs, ag_host, g, actions, n_new));
    return train_cycle_staged(a, b, on, gn, rng, n_new, future_p, sq_threshold, n_batches, open);
}

// The same cycle on episodes that lie in a host block registered with the device (hp_host_register: the feeder's shared-memory
// ring, layout of hp_buffer_store_pinned): the staging is an asynchronous DMA out of that block; `ticket` as there.
int hp_agent_train_cycle_pinned(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, const double *block,
                                int64_t n_new, double future_p, double sq_threshold, int32_t n_batches, uint64_t *ticket) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_train_cycle_pinned"));
    HP_SERIALISE(a);
    HP_REQUIRE(block, HP_ERR_INVALID, "hp_agent_train_cycle_pinned: null block");
    HP_TRY(train_cycle_checks(a, b, n_new, n_batches, "hp_agent_train_cycle_pinned"));
    const bool open = a->cycle_open && cycle_open_fits(a, n_new);
    HP_TRY(buffer_stage_pinned(b, rng, block, n_new, ticket, !open));
    return train_cycle_staged(a, b, on, gn, rng, n_new, future_p, sq_threshold, n_batches, open);
}

static int train_cycle_staged(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, int64_t n_new, double future_p,
                              double sq_threshold, int32_t n_batches, bool open) {
    hipStream_t s = a->ctx->stream;
    // 2. everything else is one graph; rebuild when a baked-in argument changes
    const bool same = a->graph && a->g_buf == b && a->g_on == on && a->g_gn == gn && a->g_rng == rng &&
                      a->g_n_new == n_new && a->g_n_batches == n_batches && a->g_future_p == future_p &&
                      a->g_sq == sq_threshold && a->g_stage == b->st_obs.p && a->g_slots == b->st_slots.p && a->g_open == open;
    // the legacy default stream cannot be captured: eager launches for THIS call only (the context may be back on a capturable
    // stream at the next one; only a refused capture with collectives, below, is sticky)
    if (a->graph_refused || s == hipStreamLegacy) {
        HP_TRY(ensure_plan(a, n_batches));
        HP_TRY(a->norm_plan.ensure((size_t)b->T * sizeof(PlanRec)));
        HP_TRY(enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>(), open));
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
        int st = enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>(), open);
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
            HP_TRY(enqueue_cycle_tail(a, b, on, gn, rng, future_p, sq_threshold, n_batches, a->norm_plan.as<PlanRec>(), open));
            a->host_steps += n_batches;
            return HP_OK;
        }
        a->g_buf = b; a->g_on = on; a->g_gn = gn; a->g_rng = rng;
        a->g_n_new = n_new; a->g_n_batches = n_batches; a->g_future_p = future_p; a->g_sq = sq_threshold;
        a->g_stage = b->st_obs.p;
        a->g_slots = b->st_slots.p;
        a->g_open = open;
    }
    HP_CHECK_HIP(hipGraphLaunch(a->graph, s));
    a->host_steps += n_batches;
    return HP_OK;
}

// which launch structure a sequence of n_updates sampled updates WITH optimizer steps takes on this agent: 0 = chain launch +
// weight-gradient launch, 1 = split launch (slab8_split.h) + the actor's tile launch (data-parallel ranks included, round 6).
// Gradient-only sequences (with_adam false) never split; hp_agent_update_kernels names the kernels of either.
int hp_agent_update_form(hp_agent *a, int32_t n_updates, int32_t *form) {
    HP_REQUIRE(a && form, HP_ERR_INVALID, "hp_agent_update_form: null argument");
    *form = update_takes_split_form(a, n_updates) ? 1 : 0;
    return HP_OK;
}

// The kernels a sequence of n_updates sampled updates enqueues on this agent AS IT IS NOW (engine, switches, attached communicator
// or peer exchange): the launch logic itself runs under a stream capture that is thrown away -- nothing executes, no state moves
// -- with the launch log on.  out: "#open,k..,#prologue,k..,#update,k..,k..,#update,...,#close,k.." (markers start with '#';
// "rccl:ncclAllReduce" stands for RCCL's own kernel).  What bench.py names in its line, instead of re-deriving the choice.
// caller_exchanges != 0: the host-driven form (hp_agent_forward_backward -> the caller's all-reduce -> hp_agent_apply), one update.
int hp_agent_update_kernels(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, double future_p, double sq_threshold,
                            int32_t n_updates, int32_t caller_exchanges, char *out, int32_t out_len) {
    HP_TRY(check_handles(a, b, on, gn, rng, "hp_agent_update_kernels"));
    HP_SERIALISE(a);
    HP_REQUIRE(out && out_len > 0 && n_updates > 0, HP_ERR_INVALID, "hp_agent_update_kernels: bad argument");
    HP_REQUIRE(!a->prof, HP_ERR_STATE, "hp_agent_update_kernels: not in profiling mode (its launches are bracketed by events)");
    // nothing of the capture ever runs, so it need not be taken on the stream the context is bound to: a context on the legacy
    // default stream (a host-driven torch.distributed loop), which cannot be captured, borrows its own stream for the log
    hipStream_t bound = a->ctx->stream, s = (bound == hipStreamLegacy || bound == hipStreamPerThread) ? a->ctx->own_stream : bound;
    HP_TRY(ensure_plan(a, n_updates));
    HP_CHECK_HIP(hipStreamSynchronize(bound));
    struct Rebind {   // every launch site reads ctx->stream
        hp_ctx *c; hipStream_t keep;
        ~Rebind() { c->stream = keep; }
    } rebind{a->ctx, bound};
    a->ctx->stream = s;
    std::vector<std::string> log;
    hipGraph_t graph = nullptr;
    unsigned *pending = a->split_reset_pending;
    HP_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hp_klog = &log;
    int st;
    if (caller_exchanges) {
        st = enqueue_updates(a, b, on, gn, rng, future_p, sq_threshold, 1, false);
        HP_KLOG("host:all_reduce");
        if (st == HP_OK) st = enqueue_adam(a);
    } else {
        st = enqueue_updates(a, b, on, gn, rng, future_p, sq_threshold, n_updates, true);
    }
    hp_klog = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    a->split_reset_pending = pending;
    if (e != hipSuccess) (void)hipGetLastError();   // (a capture with collectives may be refused: the log is complete all the same)
    if (st != HP_OK) return st;
    std::string j;
    for (size_t i = 0; i < log.size(); ++i) j += (i ? "," : "") + log[i];
    HP_REQUIRE((int)j.size() + 1 <= out_len, HP_ERR_INVALID, "hp_agent_update_kernels: %zu bytes needed", j.size() + 1);
    memcpy(out, j.c_str(), j.size() + 1);
    return HP_OK;
}

int hp_agent_engine(hp_agent *a, int32_t *engine, int32_t *slab_rows, int32_t *dw_split) {
    HP_REQUIRE(a, HP_ERR_INVALID, "hp_agent_engine: null handle");
    if (engine) *engine = !a->slab ? 0 : (a->slab8 ? 8 : 32);
    if (slab_rows) *slab_rows = !a->slab ? 0 : (a->slab8 ? a->s8_rows : S32_ROWS);
    if (dw_split) *dw_split = a->dw64 ? a->dw_S : 0;
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
    a->dw_part.release();
    a->dw_ticket.release();
    a->norm_plan.release();
    a->fwd_ws.release();
    a->pin.release();
    if (a->ev0) (void)hipEventDestroy(a->ev0);
    if (a->ev1) (void)hipEventDestroy(a->ev1);
    if (a->fault_host) (void)hipHostFree(a->fault_host);
    delete a;
}

}  // extern "C"
