// slab8_split.h -- the update restructured by its real dependency graph (round 4; included by slab8.h inside namespace s8r4).
//
// Within a sequence of updates the TARGET networks are constant (ddpg_agent.py:149-150 steps them after the n_batches updates),
// so y = clamp(r + gamma Q'(x', pi'(x'))) of update u + 1 depends on nothing update u writes (ddpg_agent.py:250-260), and the
// critic's own trajectory (critic(x, a) -> loss -> dX -> dW -> Adam, :262-263, :274-277) never reads the actor.  k_fb_slab8 carries
// both in ONE 8-layer "critic side" chain that is as long as the actor side's.  Here the launch of update u holds three kinds
// of 4-row chains, none of which waits for another:
//   A  actor side, unchanged (slab8_actor_side.inc): actor -> critic(x, pi(x)) -> dX through both              8 big layers
//   C  critic(x, a) -> Q, critic loss against the Q' the PREVIOUS launch left in qt_in, dX chain                 4 big layers
//   T  target actor -> target critic -> Q' of update u + 1's minibatch (gathered here from its plan) -> qt_out    4 big layers
// and, on the CUs the short chains leave early, the CRITIC's weight-gradient tiles + optimizer step (gemm_tile of gemm_lds.h,
// the same code as the stand-alone launch) behind two in-launch counters:
//   stage counters  C chains that have published (write-through stores drained) their dA3 / dA2 / dA1: a tile starts when all of
//            them have published what ITS problem reads (W3, W4 after the first stage, W2 after the second, W1 after the third),
//            and the tiles are numbered so that the ones dispatched first read the operands that come first;
//   gates    A chains that are past the critic's forward layers / its third layer's dX fragments / its second layer's: a tile's
//            optimizer epilogue overwrites its problem's parameters and fragment copies only after all of them are past the
//            last read of THAT layer (AdamFuse::gate, gate_sel: W4 and W1 after the forward, W3 after the first dX layer, W2
//            after the second).
// Every counter exists once per XCD (SPLIT_CTR_STRIDE words apart: a producer bumps all eight, a consumer polls its XCD's): a
// hundred workgroups polling ONE address saturate its memory channel and slow every weight stream that crosses it (measured:
// the actor-side chains went from 26.3 to 28.6 us with a single counter polled every 0.1 us).
// Forward progress of the waits: workgroup b runs on XCD b % 8, an XCD dispatches ITS workgroups in index order, chains come
// first on every XCD and tiles last -- but a tile may wait for chains of ANOTHER XCD with a larger index (the actor-side chains
// on XCDs 0-3 gate tiles everywhere; the critic chains on XCDs 4-7 feed tiles on XCDs 0-3).  What makes that safe is that every
// XCD's chains fit its CUs at once (split_fits_rows: they are all resident before any tile of that XCD is dispatched) and that
// the whole launch fits the device; with ranks sharing a device the split launch is not used at all (split_fits: no peer).
// Every poll is bounded all the same (FbSplitArgs::wait_ticks): a give-up skips the work, sets the sticky fault word and its
// pinned host mirror, and the next host call fails (agent.hip: agent_check_fault).
// The ACTOR's tiles are the launch behind this one (144 tiles, one per CU, k_gemm_lds_adam).  Holding them in this launch as well
// -- behind an "every actor-side chain has published" counter, one launch per update -- was built in round 4 and measured
// 45.7 vs 38.0 us/update (the tiles cannot start before the last chain ends, an in-launch tile takes 6-8 us against 5 in a
// fresh launch, and half of them queue behind critic tiles): removed in round 5, as was the form that carried them at the
// head of the NEXT launch (40.9 vs 38.6).  profiles/r05_ab_staged_actor_tiles.txt has the stage-by-stage arithmetic.
// Same device functions, same operands, same summation order as k_fb_slab8 + k_gemm_lds_adam: bit-identical results
// (tests/test_gpu_update.py::test_split_launch_is_bit_identical).

// counter `which` (0-2: stages of the C chains, 3-5: the A chains' gates), copy of XCD x
__device__ __forceinline__ unsigned *split_ctr(unsigned *sync, int which, int x) { return sync + (which * 8 + x) * SPLIT_CTR_STRIDE; }
__device__ __forceinline__ void split_bump(unsigned *sync, int which) {   // lanes 0-7 of one wave: one copy each
    if (threadIdx.x < 8) __hip_atomic_fetch_add(split_ctr(sync, which, (int)threadIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// publish a stage: every wave's write-through stores have completed, then one count per chain and copy
__device__ __forceinline__ void split_publish(unsigned *sync, int which) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s8_sync();
    split_bump(sync, which);
}

// Role table of the launch: word r = workgroups of role r, byte x = on XCD x (workgroup b runs on XCD b % 8, in role order within
// the XCD).  The seven words are the kernel's LEADING SCALAR arguments: they arrive in SGPRs with the wave (kernarg preload,
// Makefile; 14 dwords is the limit), so a workgroup knows its role at its first instruction and fetches what that role needs in
// its FIRST round trip to the argument block -- the fetch of the table itself cost every chain 0.45-0.75 us at the launch's start
// (tools/ubench/split_timeline.py: "first instruction -> role known").
struct SplitRoles {
    unsigned long long w[SR_N];
};
// role of this workgroup, its index among the workgroups of that role (XCD-major: all of XCD 0's first) and on its XCD
__device__ __forceinline__ int split_role(const SplitRoles &R, int &idx, int &in_xcd, int &n_in_xcd) {
    const int x = blockIdx.x & 7;
    int s = blockIdx.x >> 3, role = SR_N;
    unsigned long long mine = 0ull;   // the word of this workgroup's role
#pragma unroll
    for (int r = 0; r < SR_N; ++r) {
        const int n = (int)((R.w[r] >> (8 * x)) & 0xffull);
        if (role == SR_N) {
            if (s < n) { role = r; mine = R.w[r]; }
            else s -= n;
        }
    }
    int before = 0;
#pragma unroll
    for (int xx = 0; xx < 8; ++xx)
        if (xx < x) before += (int)((mine >> (8 * xx)) & 0xffull);
    in_xcd = s;
    n_in_xcd = role < SR_N ? (int)((mine >> (8 * x)) & 0xffull) : 0;
    idx = before + s;
    return role;
}

// the optimizer's step scalars, written through so that the tiles of THIS launch (other XCDs) can read them
__device__ __forceinline__ void adam_prepare_wt(AgentDevState *st, const AdamCfg c) {
    const long long stepi = st->step + 1;
    const double step = (double)stepi;
    const double bc1 = 1.0 - pow(c.beta1, step);
    const double bc2 = 1.0 - pow(c.beta2, step);
    __hip_atomic_store(&st->step, stepi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wt_store(&st->neg_step_actor, (float)(-(c.lr_actor / bc1)));
    wt_store(&st->neg_step_critic, (float)(-(c.lr_critic / bc1)));
    wt_store(&st->bc2_sqrt, (float)sqrt(bc2));
}

// TILES: what the in-launch tiles do behind their products (slab8_split_args.h: SPLIT_TILES_*); one kernel per form, so that the
// single-rank launch keeps its own register allocation and code placement
template <int TILES>
__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_fb_split8(unsigned long long r0, unsigned long long r1, unsigned long long r2, unsigned long long r3, unsigned long long r4,
                 unsigned long long r5, unsigned long long r6, const FbSplitArgs Q) {
    static_assert(SR_N == 7, "the role table is seven preloaded words");
    const SplitRoles R{{r0, r1, r2, r3, r4, r5, r6}};
    const FbSlabArgs &P = Q.s;
    const FwdSlabArgs &A = P.f;
    const BwdSlabArgs &Bk = P.b;
    __shared__ __attribute__((aligned(16))) float xin[S8_ROWS * S8_LDX];
    __shared__ __attribute__((aligned(16))) float xin2[S8_ROWS * S8_LDX];
    __shared__ __attribute__((aligned(16))) float bufA[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float bufB[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float pbuf[S8_ROWS * 256];
    __shared__ float dq[S8_ROWS];
    __shared__ float rows[3][S8_ROWS];          // per-row scalars: Q' | Q (or Q_pi) | reward
    __shared__ __attribute__((aligned(16))) float dz[S8_ROWS * 20];
    __shared__ __attribute__((aligned(16))) float w1t[4 * 256];
    __shared__ s8_mask_t msk[5][256];       // ReLU masks: critic h1, h2 | actor h1, h2, h3
    __shared__ __attribute__((aligned(16))) RingSlot wring[S8_WAVES][S8_RING];
    static_assert(sizeof(RingSlot) * S8_WAVES * S8_RING >= sizeof(float) * GL_LDS_FLOATS, "the weight ring must hold a tile's operand images");
    static_assert(S8_THREADS == GL_THREADS, "gemm_tile runs on the chain kernel's workgroup shape");
#ifdef SLAB_TIMELINE
    unsigned long long t_entry = wall_clock64();   // (needs no kernel argument: what the first argument fetch costs shows against stamp 0)
    asm volatile("" : "+s"(t_entry));
#endif
    int idx, in_xcd, n_in_xcd;
    const int role = split_role(R, idx, in_xcd, n_in_xcd);
    if (role == SR_N) return;

#ifdef SLAB_TIMELINE
    if (Q.tl_mark && threadIdx.x == 0 && blockIdx.x < 1024) {
        g_split_entry[blockIdx.x] = t_entry;
        g_split_role[blockIdx.x] = role;
        g_split_tl[blockIdx.x][1] = g_split_tl[blockIdx.x][2] = 0ull;
    }
#endif
    SPLIT_STAMP(0);
    const int nslab = A.Mp / S8_ROWS;
    const int slab = idx;
    const size_t row0 = (size_t)slab * S8_ROWS;
    const int tid = threadIdx.x, H = A.H;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const NetLayout &la = A.la, &lc = A.lc;
    const int ca = la.total, ad = A.act_dim;
    const float invB = 1.0f / (float)Bk.B;
    RingSlot *ring = wring[wave];
    int rbase = 0;
    unsigned long long *tl = nullptr;
#ifdef SLAB_TIMELINE
    if (slab == 0 && role <= SR_T) tl = A.tl + role * 32;
    if (role == SR_A && slab == nslab / 4) tl = A.tl + 96;   // the first actor-side chain of the SECOND XCD that holds them (placement 1 / 2)
#endif
    const SlabNetPtrs &on = A.online;
    if (role == SR_PLAN) {   // the index-plan workgroup ends here (ended waves take no part in barriers)
        if (tid >= MT_THREADS) return;
        mt_her_plan(Bk.rng, Bk.meta->current_size, Bk.T, Bk.plan_batch, 1, Bk.future_p, Bk.next_plan,
                    reinterpret_cast<uint32_t(*)[MT_N]>(&wring[0][0][0]), reinterpret_cast<int *>(pbuf));
        SPLIT_STAMP(3);
        return;
    } else if (role == SR_AHEAD) {
        s8_gather_ahead(P.ahead, P.aXT, P.aXA, P.aXP, A.ldx, A.act_off, A.act_dim, A.max_action, idx, P.n_ahead);
        SPLIT_STAMP(3);
    } else if (role == SR_WARM) {
        s8_l2_warm_at(P, (int)((Q.warm_side >> (4 * (blockIdx.x & 7))) & 15u), n_in_xcd, in_xcd, dq);
        SPLIT_STAMP(3);
    } else if (role == SR_TILE) {
        // ---- critic weight gradients + optimizer step of THIS update, on a CU a short chain has left (or that held none).
        // Tile ids are slot-major (the workgroups dispatched first on every XCD take the lowest ids): those are the tiles whose
        // operands the C chains publish first (the group lists W3, W4, W2, W1).
        float *tlds = reinterpret_cast<float *>(&wring[0][0][0]);
        float(*bsum)[32] = reinterpret_cast<float(*)[32]>(pbuf);
        // rank of (slot, XCD) among all tile workgroups in the order (slot, XCD): XCDs may hold different numbers of tiles
        int tile = 0;
        {
            const int x = (int)(blockIdx.x & 7);
#pragma unroll
            for (int xx = 0; xx < 8; ++xx) {
                const int nt = (int)((R.w[SR_TILE] >> (8 * xx)) & 0xffull);
                tile += nt < in_xcd ? nt : in_xcd;
                if (xx < x && nt > in_xcd) tile += 1;
            }
        }
        int pi = 0;
#pragma unroll
        for (int i = 1; i < MAX_PROBS; ++i)
            if (i < Q.tiles.n && tile >= Q.tiles.p[i].tile0) pi = i;
        const int stage = (int)((Q.tile_stage >> (4 * pi)) & 15u);
        if (!handoff_wait(split_ctr(Q.sync, stage, (int)(blockIdx.x & 7)), Q.need_c, Q.wait_ticks, Q.fault, Q.fault_host,
                          1u, reinterpret_cast<int *>(dq)))
            return;
        SPLIT_STAMP(1);
        if constexpr (TILES == SPLIT_TILES_GRADS) {
            // gradients into the buffer the exchange reads (plain stores: the kernel boundary publishes them); no parameter is
            // written in this launch, so no gate
            gemm_tile<false, false, true>(Q.tiles, nullptr, tile, tlds, bsum, false);
        } else if constexpr (TILES == SPLIT_TILES_PEER) {
            // the ranks' same tile meets through the exchange block's per-tile flag rows (rows 0 .. tiles - 1: the actor's tiles of
            // the launch behind this one use the rows that follow), summed in rank order, then the gate and the step
            const PeerTile PT{Q.peer, Q.peer_u, Q.peer_mean, 0};
            gemm_tile<true, false, true, true>(Q.tiles, &Q.adam, tile, tlds, bsum, false, &PT);
        } else {
            gemm_tile<true, false, true>(Q.tiles, &Q.adam, tile, tlds, bsum, false);
        }
        SPLIT_STAMP(3);
    } else if (role == SR_T) {
        // ------------------------------------------------------------------ target side, one update ahead
        // actor_target -> critic_target -> Q' (ddpg_agent.py:250-256) of the NEXT update's minibatch: its rows are gathered here
        // (her.py:26-38, ddpg_agent.py:228-243) from the plan an earlier launch drew; same arithmetic as k_fb_slab8's critic side
        S8_TSTAMP(tl, 0);
        const SlabNetPtrs &tn = A.target;
        const GatherSrc &G = Q.tgs;
        if (Q.reset_sync && slab == 0 && tid < SPLIT_COUNTERS * 8)   // prologue: the first update's set (nothing of THIS launch counts)
            __hip_atomic_store(Q.sync_other + tid * SPLIT_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const PlanRec rec = s8_plan_rec(G, row0);
        float4 wbaT[6], wbcT[6], whT[4], wqT[4];
        s8_ring_prologue<0, S8_PRO_FIRST>(ring, rbase, tn.wf + la.w2);
        s8_small_prefetch(tn.wf + la.w1, la.K1, wbaT);
        float ebT[3] = {0.f, 0.f, 0.f}, ebC[3] = {0.f, 0.f, 0.f};
        const int ecol_ = 64 * (wave & 3) + lane;
        const bool ekh0_ = S8_BOTH_HALVES || wave < 4;   // (8- / 16-row slabs: both reduction halves run epilogues)
        if (ekh0_) ebT[0] = tn.canon[la.b1 + ecol_];
        __builtin_amdgcn_sched_barrier(0);
        s8_gather(xin, G, rec, 0, row0, A.ldx, A.act_off, ad, A.max_action, nullptr);
        s8_ring_prologue<S8_PRO_FIRST, S8_RING>(ring, rbase, tn.wf + la.w2);
        __builtin_amdgcn_sched_barrier(0);
        s8_small_prefetch(tn.wf + ca + lc.w1, lc.K1, wbcT);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            whT[j] = *reinterpret_cast<const float4 *>(tn.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
        wqT[0] = *reinterpret_cast<const float4 *>(tn.canon + ca + lc.w4 + 4 * lane);
        const float bhT = tn.canon[la.b4 + (lane < ad ? lane : 0)];
        const float bqT = tn.canon[ca + lc.b4];
        if (ekh0_) {
            ebT[1] = tn.canon[la.b2 + ecol_]; ebT[2] = tn.canon[la.b3 + ecol_];
            ebC[0] = tn.canon[ca + lc.b1 + ecol_]; ebC[1] = tn.canon[ca + lc.b2 + ecol_]; ebC[2] = tn.canon[ca + lc.b3 + ecol_];
        }
        const float *pT = ebT, *pC = ebC;
        __builtin_amdgcn_sched_barrier(0);
        s8_sync();
        s8_trunk(xin, la, wbaT, tn.wf, tn.canon, H, bufA, bufB, pbuf, nullptr, nullptr, nullptr, row0, ring, rbase,
                 tn.wf + ca + lc.w2, tl, 1, nullptr, nullptr, nullptr, pT);
        {   // target actor head -> action block of the target critic's input (models.py:24)
#pragma unroll
            for (int i = 0; i < S8_RPW; ++i) {
                const int rr = wave + S8_WAVES * i;
                const float z = s8_rowdots(bufA, S8_LD, rr < S8_ROWS ? rr : 0, ad, whT);
                if (lane < ad && rr < S8_ROWS) {
                    const float th = tanhf(z + bhT);
                    xin[rr * S8_LDX + A.act_off + lane] = (A.max_action * th) / A.max_action;
                }
            }
        }
        s8_sync();
        S8_TSTAMP(tl, 7);
        s8_trunk(xin, lc, wbcT, tn.wf + ca, tn.canon + ca, H, bufA, bufB, pbuf, nullptr, nullptr, nullptr, row0, ring, rbase,
                 nullptr, tl, 8, nullptr, nullptr, nullptr, pC);
        {
#pragma unroll
            for (int i = 0; i < S8_RPW; ++i) {
                const int rr = wave + S8_WAVES * i;
                const float q = s8_rowdots(bufA, S8_LD, rr < S8_ROWS ? rr : 0, 1, wqT);
                if (lane == 0 && rr < S8_ROWS) wt_store(Q.qt_out + (row0 + rr) * 16, q + bqT);
            }
        }
        S8_TSTAMP(tl, 13);
        SPLIT_STAMP(3);
    } else if (role == SR_C) {
        // ------------------------------------------------------------------ critic side without its target half
        // critic(x, a) -> Q, critic loss against the Q' in qt_in, critic dX chain (ddpg_agent.py:257-263 up to the weight gradients)
        S8_TSTAMP(tl, 0);
        const PlanRec rec = s8_plan_rec(A.gs, row0);
        float4 wbcA[6], wqA[4];
        s8_ring_prologue<0, S8_PRO_FIRST>(ring, rbase, on.wf + ca + lc.w2);
        s8_small_prefetch(on.wf + ca + lc.w1, lc.K1, wbcA);
        float ebA[3] = {0.f, 0.f, 0.f};
        const int ecol_ = 64 * (wave & 3) + lane;
        const bool ekh0_ = S8_BOTH_HALVES || wave < 4;   // (8- / 16-row slabs: both reduction halves run epilogues)
        if (ekh0_) ebA[0] = on.canon[ca + lc.b1 + ecol_];
        __builtin_amdgcn_sched_barrier(0);
        if (A.gs.plan) {
            s8_gather(xin2, A.gs, rec, 1, row0, A.ldx, A.act_off, ad, A.max_action, const_cast<float *>(A.XA), rows[2]);
        } else {
            s8_load(xin2, S8_LDX, A.ldx, A.XA + row0 * A.ldx, A.ldx);
            if (tid < S8_ROWS) rows[2][tid] = Bk.R[row0 + tid];
        }
        if (tid < S8_ROWS) rows[0][tid] = Q.qt_in[(row0 + tid) * 16];
        s8_ring_prologue<S8_PRO_FIRST, S8_RING>(ring, rbase, on.wf + ca + lc.w2);
        __builtin_amdgcn_sched_barrier(0);
        wqA[0] = *reinterpret_cast<const float4 *>(on.canon + ca + lc.w4 + 4 * lane);
        const float bqA = on.canon[ca + lc.b4];
        const float w4c = on.canon[ca + lc.w4 + (tid & 255)];
        if (ekh0_) { ebA[1] = on.canon[ca + lc.b2 + ecol_]; ebA[2] = on.canon[ca + lc.b3 + ecol_]; }
        const float *pA = ebA;
        if (slab == 0 && tid == 0) adam_prepare_wt(Bk.st, Bk.adam);   // step scalars of this update's optimizer epilogues (this chain is not the launch's critical path)
        __builtin_amdgcn_sched_barrier(0);
        s8_sync();
        S8_TSTAMP(tl, 13);
        // critic(x, a): forward with global copies (weight gradients) and masks (dX chain below)
        s8_trunk(xin2, lc, wbcA, on.wf + ca, on.canon + ca, H, bufA, bufB, pbuf, A.CAh1, A.CAh2, A.CAh3, row0, ring, rbase,
                 on.wd + ca + lc.w3, tl, 14, msk[0], msk[1], nullptr, pA);
        {
#pragma unroll
            for (int i = 0; i < S8_RPW; ++i) {
                const int rr = wave + S8_WAVES * i;
                const float q = s8_rowdots(bufA, S8_LD, rr < S8_ROWS ? rr : 0, 1, wqA);
                if (lane == 0 && rr < S8_ROWS) {
                    rows[1][rr] = q + bqA;
                    A.QA[(row0 + rr) * 16] = q + bqA;
                }
            }
        }
        s8_sync();
        S8_TSTAMP(tl, 18);
        // ---- critic loss (ddpg_agent.py:255-263)
        float keep_g = 0.f, keep_a = 0.f;
        if (tid < S8_ROWS) {
            const size_t m = row0 + tid;
            float g = 0.f, sq = 0.f;
            if ((int)m < Bk.B) {
                float y = rows[2][tid] + Bk.gamma * rows[0][tid];
                y = fminf(fmaxf(y, -Bk.clip_ret), 0.f);
                const float d = y - rows[1][tid];
                sq = d * d;
                g = -2.f * d * invB;
            }
            sq = s8_rows_sum_to_lane0(sq);
            keep_g = g;
            keep_a = sq;
        }
        s8_head_bwd_inplace_td(rows, (int)row0, Bk.B, Bk.gamma, Bk.clip_ret, invB, w4c, bufA);   // bufA holds h3 of critic(x, a)
        s8_sync();
        S8_TSTAMP(tl, 19);
        s8_store(bufA, S8_LD, H, Bk.dA3 + row0 * H, H);
        if (tid < S8_ROWS) {
            wt_store(Bk.dQA + (row0 + tid) * 16, keep_g);
            if (tid == 0) wt_store(Bk.part + slab, keep_a);
        }
        // Stages are published ONE LAYER LATE, where it is free: a layer that has no successor to prefetch ends with every wave
        // at vmcnt(0) (s8_ring_step, last blocks), i.e. with all of the wave's earlier stores complete -- so after the barrier
        // behind the W3 layer, h1..h3, dA3 and the head's dQ (stored BEFORE that layer) are visible without a drain of their own.
        // (Draining right behind the stores and restarting the ring cold cost the critic chains 1 us per stage: they ended at
        // 17-18.6 instead of 15.6-16 us and the last tiles with them.)
        s8_big_layer(bufA, S8_LD, ring, rbase, on.wd + ca + lc.w3, nullptr, SE_MASK, nullptr, 0, pbuf, bufB, S8_LD,
                     msk[1], nullptr, nullptr, 0, Bk.dA2 + row0 * H);
        s8_sync();
        split_bump(Q.sync, 0);      // stage 0: W3's and W4's tiles may start
        s8_ring_prologue(ring, rbase, on.wd + ca + lc.w2);
        S8_TSTAMP(tl, 20);
        s8_big_layer(bufB, S8_LD, ring, rbase, on.wd + ca + lc.w2, nullptr, SE_MASK, nullptr, 0, pbuf, bufA, S8_LD, msk[0],
                     nullptr, nullptr, 0, Bk.dA1 + row0 * H);
        s8_sync();
        split_bump(Q.sync, 1);      // stage 1 (dA2, stored before the layer that just ended): W2's tiles
        S8_TSTAMP(tl, 21);
        split_publish(Q.sync, 2);   // stage 2: dA1 (and the inputs gathered at entry) behind a drain of its own: W1's tiles
        S8_TSTAMP(tl, 22);
        SPLIT_STAMP(3);
    } else {
        // ---------------------------------------------------------------------- actor side
        S8_TSTAMP(tl, 0);
        if (slab == 0 && tid < SPLIT_COUNTERS * 8)   // the NEXT launch's counter set (nobody of this launch touches it)
            __hip_atomic_store(Q.sync_other + tid * SPLIT_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define S8_AFTER_CRITIC_FWD do { split_bump(Q.sync, 3); } while (0)
#define S8_AFTER_CRITIC_DX1 do { split_bump(Q.sync, 4); } while (0)
#define S8_AFTER_CRITIC_DX do { split_bump(Q.sync, 5); SPLIT_STAMP(1); } while (0)
#include "slab8_actor_side.inc"
#undef S8_AFTER_CRITIC_FWD
#undef S8_AFTER_CRITIC_DX1
#undef S8_AFTER_CRITIC_DX
        SPLIT_STAMP(3);
    }
}
