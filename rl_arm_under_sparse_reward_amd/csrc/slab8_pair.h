// slab8_pair.h -- the thin-slab engine with every chain carried by a PAIR of workgroups on one XCD (included by slab8.h
// inside its 8-row namespace; small batches: the 4 * Mp / 8 chain workgroups must fit the CUs, batch <= 512).
//
// Why: a chain workgroup spends 2.3-2.4 us per 256 x 256 layer because ONE CU has
// to pull the layer's 256 KiB of weights through its LDS-DMA path (139 GB/s: 1.9 us), whatever the number of rows.  Two
// CUs that share an L2 split two consecutive layers Megatron-style and each streams half of both:
//     layer l     by OUTPUT COLUMNS: member p computes columns [128 p, 128 p + 128) of all 8 rows from the full input
//                 (128 KiB of weights; bias / ReLU / mask are local to a column)
//     layer l + 1 by REDUCTION INDEX: member p owns exactly the reduction indices it just produced, computes the partial
//                 sums of ALL 256 outputs over them (128 KiB of weights), the members exchange their 8 x 256 partial sums
//                 through the shared L2 (8 KiB each way), add them in member order (same bits on both) and both hold the
//                 full output again
// one exchange per TWO layers: measured 0.94 us per round trip for 8 KiB per side between workgroups i and i ^ 8 (plain
// stores, vmcnt(0), flag, agent-scope poll and loads: tools/ubench/pair_exchange.hip, profiles/r03_pair_exchange.txt),
// against 2 x ~1.1 us of weight streaming saved.  Everything that is not a 256 x 256 layer (first layers, heads, losses,
// the gather) is computed by both members redundantly; global outputs are written by one of them (or half each).
//
// Placement: workgroups are dealt round-robin to the 8 XCDs, so members b and b ^ 8 share an XCD (checked on the host
// with a probe launch before the engine is enabled, and in the kernel: every flag carries its writer's XCC id and a
// mismatch sets the sticky error word).  Exchange memory: per workgroup two 8 KiB buffers (exchange k uses buffer k & 1:
// a member that has seen flag k + 1 of its partner knows the partner has consumed buffer k & 1 ... of exchange k - 1) and
// one 64-bit flag = epoch; a workgroup reads its OWN flag at kernel entry as the launch's epoch base (both members did
// the same number of exchanges before), so replayed hipGraphs need no reset.  Every poll is bounded.
#if S8_NRG == 2

#define P8_SPIN_LIMIT 200000   // x (s_sleep 1 + one L2 round trip ~ 0.3 us): tens of ms, then the sticky error word

struct PairLink {
    int p;                               // which member of the pair this workgroup is
    float *mine;                         // [2][S8_ROWS * 256] partial sums this member publishes
    const float *theirs;                 // ... and the partner's
    unsigned long long *my_flag;
    const unsigned long long *their_flag;
    unsigned long long base;             // epoch before this launch
    int k;                               // exchanges done in this launch
    unsigned xcc;
    unsigned int *err;
};

// agent-scope loads (sc1): miss this CU's L1 and are served by the XCD's L2, where the partner's write-through stores are.
// Through the builtin, not inline assembly: the compiler must know when the value arrives.
__device__ __forceinline__ float p8_load_agent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long p8_load_agent64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void p8_store_flag(unsigned long long *p, unsigned long long v) {   // plain (write-through L1) store
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// ---- 16-block layers: a wave streams 16 KiB per layer instead of 32 ------------------------------------------------------
// block t of this wave = block b0 + t of column group cg; the ring is continuous across the two layers of a pair (the
// last blocks' slots take the first blocks of `nxt`, whose wave mapping is (ncg, nb0)), and drained before an exchange.
template <int T, bool HAS_NEXT>
__device__ __forceinline__ void p8_ring_step(f32x4 (&c)[S8_NRG], RingSlot *ring, int rbase, const float *wlayer, int cg, int b0,
                                             const float *nxt, int ncg, int nb0, const float (&a)[8][S8_NRG],
                                             const float4 bcur) {
    float4 bnext = bcur;
    if constexpr (T + 1 < 16) {
        constexpr int out = HAS_NEXT ? S8_RING - 1 : ((15 - T) < S8_RING - 1 ? (15 - T) : S8_RING - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(out - 1) : "memory");
        bnext = ring[(rbase + T + 1) % S8_RING][threadIdx.x & 63];
    }
    if constexpr (T + S8_RING < 16 || HAS_NEXT) {
        if constexpr (T + 1 < 16) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (T + S8_RING < 16) s8_dma16(s8_wblock(wlayer, cg, 64, b0 + T + S8_RING), &ring[(rbase + T + S8_RING) % S8_RING][0]);
        else s8_dma16(s8_wblock(nxt, ncg, 64, nb0 + T + S8_RING - 16), &ring[(rbase + T + S8_RING) % S8_RING][0]);
    }
    s8_mma<T % 4>(c, a[T / 4], bcur);
    if constexpr (T + 1 < 16) p8_ring_step<T + 1, HAS_NEXT>(c, ring, rbase, wlayer, cg, b0, nxt, ncg, nb0, a, bnext);
}

__device__ __forceinline__ void p8_prologue(RingSlot *ring, int rbase, const float *wlayer, int cg, int b0) {
#pragma unroll
    for (int t = 0; t < S8_RING; ++t) s8_dma16(s8_wblock(wlayer, cg, 64, b0 + t), &ring[(rbase + t) % S8_RING][0]);
}

// wave mappings: column-split layer (cg = 2 p + wave % 2, reduction quarter wave / 2), reduction-split layer (cg = wave % 4,
// the wave / 4 -th half of the member's 128 reduction indices)
__device__ __forceinline__ int p8c_cg(int p, int wave) { return 2 * p + (wave & 1); }
__device__ __forceinline__ int p8c_b0(int wave) { return (wave >> 1) * 16; }
__device__ __forceinline__ int p8k_cg(int wave) { return wave & 3; }
__device__ __forceinline__ int p8k_b0(int p, int wave) { return 32 * p + 16 * (wave >> 2); }

// Column-split layer.  lin: full input [8][ld_in]; writes columns [128 p, 128 p + 128) of lout (indexed by the GLOBAL column),
// of mask_out and of gout.  The first S8_RING blocks of `wlayer` (this mapping) are in flight on entry; on exit the first
// S8_RING blocks of `nxt` in the reduction-split mapping.  pbuf: 3 x [8][128] floats.
__device__ __forceinline__ void p8_layer_c(const float *lin, int ld_in, RingSlot *ring, int &rbase, const float *__restrict__ wlayer,
                                           const float *__restrict__ nxt, int p, int epi, const float *__restrict__ bias,
                                           float *pbuf, float *lout, int ld_out, const s8_mask_t *mask_in, s8_mask_t *mask_out,
                                           float *gout, unsigned long long *tl2 = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cg = p8c_cg(p, wave), kq = wave >> 1, b0 = p8c_b0(wave);
    const int coll = 64 * (wave & 1) + lane, col = 128 * p + coll;
    float e0 = 0.f;
    if (kq == 0 && epi == SE_BIAS_RELU) e0 = bias[col];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c[S8_NRG];
#pragma unroll
    for (int g = 0; g < S8_NRG; ++g) c[g] = f32x4{0, 0, 0, 0};
    float a[8][S8_NRG];
    s8_aload<4>(lin, ld_in, 4 * b0, a);
    S8_TSTAMP(tl2, 24);
    S8_WSTAMP(tl2, 64);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S8_RING - 1) : "memory");   // block 0 has landed
    const float4 bfirst = ring[rbase % S8_RING][lane];
    p8_ring_step<0, true>(c, ring, rbase, wlayer, cg, b0, nxt, p8k_cg(wave), p8k_b0(p, wave), a, bfirst);
    rbase = (rbase + 16) % S8_RING;
    __builtin_amdgcn_sched_barrier(0);
    S8_TSTAMP(tl2, 25);
    S8_WSTAMP(tl2, 72);
    if (kq > 0) {
#pragma unroll
        for (int g = 0; g < S8_NRG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[((kq - 1) * S8_ROWS + 4 * g + r) * 128 + coll] = c[g][r];
    }
    s8_sync();
    S8_TSTAMP(tl2, 11);
    if (kq == 0) {
        const unsigned bits = mask_in ? (unsigned)mask_in[col] : 0u;
        unsigned outbits = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < S8_NRG; ++g) {
                const int row = 4 * g + r;
                const float v = ((c[g][r] + pbuf[row * 128 + coll]) + pbuf[(S8_ROWS + row) * 128 + coll]) +
                                pbuf[(2 * S8_ROWS + row) * 128 + coll];
                float o;
                if (epi == SE_BIAS_RELU) {
                    o = fmaxf(v + e0, 0.f);
                    outbits |= (o > 0.f ? 1u : 0u) << row;
                } else {
                    o = ((bits >> row) & 1u) ? v : 0.f;
                }
                lout[row * ld_out + col] = o;
                if (gout) wt_store(gout + (size_t)row * 256 + col, o);
            }
        if (mask_out) mask_out[col] = (s8_mask_t)outbits;
    }
    S8_TSTAMP(tl2, 26);
}

// Reduction-split layer + exchange.  lin: [8][ld_in], only this member's 128 columns are read.  On exit lout / mask_out hold
// the FULL output on both members; gout gets this member's 4 rows.  The first S8_RING blocks of `wlayer` are in flight on
// entry; the ring is empty during the exchange; on exit the first S8_RING blocks of `next_c` (column-split mapping) are in
// flight when next_c != nullptr.  pbuf: [8][256] floats.
__device__ __forceinline__ void p8_layer_k(const float *lin, int ld_in, RingSlot *ring, int &rbase, const float *__restrict__ wlayer,
                                           const float *__restrict__ next_c, PairLink &L, int epi,
                                           const float *__restrict__ bias, float *pbuf, float *lout, int ld_out,
                                           const s8_mask_t *mask_in, s8_mask_t *mask_out, float *gout,
                                           unsigned long long *tl2 = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = L.p, cg = p8k_cg(wave), kh = wave >> 2, b0 = p8k_b0(p, wave), col = 64 * cg + lane;
    float e0 = 0.f;
    if (kh == 0 && epi == SE_BIAS_RELU) e0 = bias[col];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c[S8_NRG];
#pragma unroll
    for (int g = 0; g < S8_NRG; ++g) c[g] = f32x4{0, 0, 0, 0};
    float a[8][S8_NRG];
    s8_aload<4>(lin, ld_in, 4 * b0, a);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S8_RING - 1) : "memory");
    const float4 bfirst = ring[rbase % S8_RING][lane];
    S8_TSTAMP(tl2, 27);
    S8_WSTAMP(tl2, 80);
    p8_ring_step<0, false>(c, ring, rbase, wlayer, cg, b0, nullptr, 0, 0, a, bfirst);
    rbase = (rbase + 16) % S8_RING;
    __builtin_amdgcn_sched_barrier(0);
    S8_TSTAMP(tl2, 28);
    S8_WSTAMP(tl2, 88);
    if (kh == 1) {
#pragma unroll
        for (int g = 0; g < S8_NRG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[(4 * g + r) * 256 + col] = c[g][r];
    }
    s8_sync();
    S8_TSTAMP(tl2, 12);
    // ---- publish this member's partial sums
    float *mine = L.mine + (size_t)(L.k & 1) * (S8_ROWS * 256);
    const float *theirs = L.theirs + (size_t)(L.k & 1) * (S8_ROWS * 256);
    float v[S8_ROWS];
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < S8_NRG; ++g) {
                const int row = 4 * g + r;
                v[row] = c[g][r] + pbuf[row * 256 + col];
                mine[row * 256 + col] = v[row];
            }
    }
    S8_TSTAMP(tl2, 17);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring is empty: these are the stores above, acknowledged by the L2
    s8_sync();
    S8_TSTAMP(tl2, 29);
    const unsigned long long want = L.base + (unsigned long long)(L.k + 1);
    if (threadIdx.x == 0) {
        p8_store_flag(L.my_flag, ((unsigned long long)L.xcc << 56) | want);   // the partner reads it from the shared L2
        unsigned long long f = 0ull;
        int spins = 0;
        for (;;) {
            f = p8_load_agent64(L.their_flag);
            if ((f & 0x00ffffffffffffffull) >= want) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > P8_SPIN_LIMIT) {
                __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if ((unsigned)(f >> 56) != L.xcc && spins <= P8_SPIN_LIMIT)   // the partner runs on another XCD: its stores are not in OUR L2
            __hip_atomic_store(L.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (kh == 1 && next_c) {
        // the waves that take no part in the exchange start the next layer's weight stream under the poll
        p8_prologue(ring, rbase, next_c, p8c_cg(p, wave), p8c_b0(wave));
    }
    s8_sync();
    S8_TSTAMP(tl2, 30);
    if (kh == 0) {
        float t[S8_ROWS], sum[S8_ROWS];
#pragma unroll
        for (int row = 0; row < S8_ROWS; ++row) t[row] = p8_load_agent(theirs + row * 256 + col);
#pragma unroll
        for (int row = 0; row < S8_ROWS; ++row) sum[row] = (p == 0) ? v[row] + t[row] : t[row] + v[row];   // member 0's half first: same bits on both
        __builtin_amdgcn_sched_barrier(0);
        S8_TSTAMP(tl2, 23);
        // the next layer's weight stream starts once the partner's sums have arrived (while an LDS-DMA is pending the
        // compiler's own waits are vmcnt(0): issued earlier, the stream would have to land before the sums could be used)
        if (next_c) p8_prologue(ring, rbase, next_c, p8c_cg(p, wave), p8c_b0(wave));
        const unsigned bits = mask_in ? (unsigned)mask_in[col] : 0u;
        unsigned outbits = 0u;
#pragma unroll
        for (int row = 0; row < S8_ROWS; ++row) {
            const float s = sum[row];
            float o;
            if (epi == SE_BIAS_RELU) {
                o = fmaxf(s + e0, 0.f);
                outbits |= (o > 0.f ? 1u : 0u) << row;
            } else {
                o = ((bits >> row) & 1u) ? s : 0.f;
            }
            lout[row * ld_out + col] = o;
            if (gout && (row >> 2) == p) wt_store(gout + (size_t)row * 256 + col, o);
        }
        if (mask_out) mask_out[col] = (s8_mask_t)outbits;
    }
    S8_TSTAMP(tl2, 31);
    L.k += 1;
}

// this member's 4 rows of a full [8][width] LDS slab -> global (write-through: operand of the weight-gradient tiles)
__device__ __forceinline__ void p8_store_rows(const float *l, int ld, int width, float *g, int ldg, int p) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < 4 * per_row; f += S8_THREADS) {
        const int r = 4 * p + f / per_row, c4 = f % per_row;
        wt_store4(g + (size_t)r * ldg + 4 * c4, *reinterpret_cast<const float4 *>(l + r * ld + 4 * c4));
    }
}

// small layer (weights prefetched into registers) computed by both members; the global copy comes from member 0
__device__ __forceinline__ void p8_small_layer(const float *lin, int ld_in, int Kred, const float4 (&b)[6], int epi,
                                               const float *__restrict__ aux, float *pbuf, float *lout, int ld_out,
                                               const s8_mask_t *mask_in, s8_mask_t *mask_out, float *gout, int p) {
    s8_small_layer(lin, ld_in, Kred, b, epi, aux, 0, pbuf, lout, ld_out, mask_in, mask_out, p == 0 ? gout : nullptr);
}

__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fb_pair8(const FbSlabArgs P) {
    const FwdSlabArgs &A = P.f;
    const BwdSlabArgs &Bk = P.b;
    __shared__ __attribute__((aligned(16))) float xin[S8_ROWS * S8_LDX];
    __shared__ __attribute__((aligned(16))) float xin2[S8_ROWS * S8_LDX];
    __shared__ __attribute__((aligned(16))) float bufA[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float bufB[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float pbuf[3 * S8_ROWS * 128];
    __shared__ float dq[S8_ROWS];
    __shared__ float rows[3][S8_ROWS];          // per-row scalars: Q' | Q (or Q_pi) | reward
    __shared__ __attribute__((aligned(16))) float dz[S8_ROWS * 20];
    __shared__ __attribute__((aligned(16))) float w1t[4 * 256];
    __shared__ s8_mask_t msk[5][256];       // ReLU masks: critic h1, h2 | actor h1, h2, h3
    __shared__ __attribute__((aligned(16))) RingSlot wring[S8_WAVES][S8_RING];
    const int nslab = A.Mp / S8_ROWS;            // pairs per chain
    const int n_chain_wg = 4 * nslab;            // 2 chains x nslab pairs x 2 members
    const int tid = threadIdx.x, H = A.H;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const NetLayout &la = A.la, &lc = A.lc;
    const int ca = la.total, ad = A.act_dim;
    const float invB = 1.0f / (float)Bk.B;
    RingSlot *ring = wring[wave];
    int rbase = 0;
    unsigned long long *tl = nullptr;
    (void)tl;
    if ((int)blockIdx.x >= n_chain_wg) {   // spare workgroups, exactly as in k_fb_slab8
        const int extra = (int)blockIdx.x - n_chain_wg;
        if (extra < P.n_plan) {
            if (tid >= MT_THREADS) return;
            mt_her_plan(Bk.rng, Bk.meta->current_size, Bk.T, Bk.plan_batch, 1, Bk.future_p, Bk.next_plan,
                        reinterpret_cast<uint32_t(*)[MT_N]>(&wring[0][0][0]), reinterpret_cast<int *>(pbuf));
            return;
        } else if (extra < P.n_plan + P.n_ahead) {
            s8_gather_ahead(P.ahead, P.aXT, P.aXA, P.aXP, A.ldx, A.act_off, A.act_dim, A.max_action, extra - P.n_plan, P.n_ahead);
        } else if (extra < P.n_plan + P.n_ahead + P.n_pref) {
            s8_l2_warm(P, extra - P.n_plan - P.n_ahead, dq);
        }
        return;
    }
    // members b and b ^ 8 share an XCD; XCDs 0-3 carry the critic-side chains, 4-7 the actor-side chains (as xcd_split)
    const int b = blockIdx.x, p = (b >> 3) & 1, q = ((b >> 4) << 3) | (b & 7);
    const int chain = (q & 7) >> 2, slab = (q >> 3) * 4 + (q & 3);
    const size_t row0 = (size_t)slab * S8_ROWS;
#ifdef SLAB_TIMELINE
    if (slab == 0 && p == 0) tl = A.tl + chain * 32;
#endif
    PairLink L;
    L.p = p;
    L.mine = P.pair_exch + (size_t)b * (2 * S8_ROWS * 256);
    L.theirs = P.pair_exch + (size_t)(b ^ 8) * (2 * S8_ROWS * 256);
    L.my_flag = P.pair_flags + (size_t)b * 8;
    L.their_flag = P.pair_flags + (size_t)(b ^ 8) * 8;
    L.k = 0;
    L.err = P.pair_err;
    {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        L.xcc = id & 0xfu;
    }
    L.base = p8_load_agent64(L.my_flag) & 0x00ffffffffffffffull;   // written by this slot's previous launch (kernel boundary)
    const bool w0 = p == 0;                        // member 0 writes the global copies of everything computed redundantly
    S8_TSTAMP(tl, 0);
    const SlabNetPtrs &on = A.online;
    if (chain == 0) {
        // ------------------------------------------------------------------ critic side
        const SlabNetPtrs &tn = A.target;
        const PlanRec rec = s8_plan_rec(A.gs, row0);
        float4 wbaT[6], wbcT[6], wbcA[6], whT[4], wqT[4], wqA[4];
        s8_small_prefetch(tn.wf + la.w1, la.K1, wbaT);
        s8_small_prefetch(tn.wf + ca + lc.w1, lc.K1, wbcT);
        s8_small_prefetch(on.wf + ca + lc.w1, lc.K1, wbcA);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            whT[j] = *reinterpret_cast<const float4 *>(tn.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
        wqT[0] = *reinterpret_cast<const float4 *>(tn.canon + ca + lc.w4 + 4 * lane);
        wqA[0] = *reinterpret_cast<const float4 *>(on.canon + ca + lc.w4 + 4 * lane);
        const float bhT = tn.canon[la.b4 + (lane < ad ? lane : 0)];
        const float bqT = tn.canon[ca + lc.b4], bqA = on.canon[ca + lc.b4];
        const float w4c = on.canon[ca + lc.w4 + (tid & 255)];
        __builtin_amdgcn_sched_barrier(0);
        if (A.gs.plan) {
            s8_gather(xin, A.gs, rec, 0, row0, A.ldx, A.act_off, ad, A.max_action, nullptr);
            s8_gather(xin2, A.gs, rec, 1, row0, A.ldx, A.act_off, ad, A.max_action, w0 ? const_cast<float *>(A.XA) : nullptr, rows[2]);
        } else {
            s8_load(xin, S8_LDX, A.ldx, A.XT + row0 * A.ldx, A.ldx);
            s8_load(xin2, S8_LDX, A.ldx, A.XA + row0 * A.ldx, A.ldx);
            if (tid < S8_ROWS) rows[2][tid] = Bk.R[row0 + tid];
        }
        p8_prologue(ring, rbase, tn.wf + la.w2, p8c_cg(p, wave), p8c_b0(wave));
        s8_sync();
        // actor_target(x')
        S8_TSTAMP(tl, 1);
        p8_small_layer(xin, S8_LDX, la.K1, wbaT, SE_BIAS_RELU, tn.canon + la.b1, pbuf, bufA, S8_LD, nullptr, nullptr, nullptr, p);
        s8_sync();
        S8_TSTAMP(tl, 2);
        p8_layer_c(bufA, S8_LD, ring, rbase, tn.wf + la.w2, tn.wf + la.w3, p, SE_BIAS_RELU, tn.canon + la.b2, pbuf, bufB, S8_LD, nullptr,
                   nullptr, nullptr);
        s8_sync();
        S8_TSTAMP(tl, 3);
        p8_layer_k(bufB, S8_LD, ring, rbase, tn.wf + la.w3, tn.wf + ca + lc.w2, L, SE_BIAS_RELU, tn.canon + la.b3, pbuf, bufA, S8_LD,
                   nullptr, nullptr, nullptr);
        s8_sync();
        S8_TSTAMP(tl, 4);
        {   // target actor head -> action block of the target critic's input (models.py:24)
            const int rr = wave;
            const float z = s8_rowdots(bufA, S8_LD, rr, ad, whT);
            if (lane < ad) {
                const float th = tanhf(z + bhT);
                const float u = (A.max_action * th) / A.max_action;
                xin[rr * S8_LDX + A.act_off + lane] = u;
                if (w0) const_cast<float *>(A.XT)[(row0 + rr) * A.ldx + A.act_off + lane] = u;
            }
        }
        s8_sync();
        S8_TSTAMP(tl, 7);
        // critic_target(x', a')
        p8_small_layer(xin, S8_LDX, lc.K1, wbcT, SE_BIAS_RELU, tn.canon + ca + lc.b1, pbuf, bufA, S8_LD, nullptr, nullptr, nullptr, p);
        s8_sync();
        S8_TSTAMP(tl, 8);
        p8_layer_c(bufA, S8_LD, ring, rbase, tn.wf + ca + lc.w2, tn.wf + ca + lc.w3, p, SE_BIAS_RELU, tn.canon + ca + lc.b2, pbuf, bufB,
                   S8_LD, nullptr, nullptr, nullptr, tl);
        s8_sync();
        S8_TSTAMP(tl, 9);
        p8_layer_k(bufB, S8_LD, ring, rbase, tn.wf + ca + lc.w3, on.wf + ca + lc.w2, L, SE_BIAS_RELU, tn.canon + ca + lc.b3, pbuf, bufA,
                   S8_LD, nullptr, nullptr, nullptr, tl);
        s8_sync();
        S8_TSTAMP(tl, 10);
        {
            const int rr = wave;
            const float qv = s8_rowdots(bufA, S8_LD, rr, 1, wqT);
            if (lane == 0) {
                rows[0][rr] = qv + bqT;
                if (w0) A.QT[(row0 + rr) * 16] = qv + bqT;
            }
        }
        S8_TSTAMP(tl, 13);
        // critic(x, a): forward with global copies (weight gradients) and masks (dX chain below)
        p8_small_layer(xin2, S8_LDX, lc.K1, wbcA, SE_BIAS_RELU, on.canon + ca + lc.b1, pbuf, bufB, S8_LD, nullptr, msk[0],
                       A.CAh1 + row0 * H, p);
        s8_sync();
        S8_TSTAMP(tl, 14);
        p8_layer_c(bufB, S8_LD, ring, rbase, on.wf + ca + lc.w2, on.wf + ca + lc.w3, p, SE_BIAS_RELU, on.canon + ca + lc.b2, pbuf, bufA,
                   S8_LD, nullptr, msk[1], A.CAh2 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 15);
        p8_layer_k(bufA, S8_LD, ring, rbase, on.wf + ca + lc.w3, on.wd + ca + lc.w3, L, SE_BIAS_RELU, on.canon + ca + lc.b3, pbuf, bufB,
                   S8_LD, nullptr, nullptr, A.CAh3 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 16);
        {
            const int rr = wave;
            const float qv = s8_rowdots(bufB, S8_LD, rr, 1, wqA);
            if (lane == 0) {
                rows[1][rr] = qv + bqA;
                if (w0) A.QA[(row0 + rr) * 16] = qv + bqA;
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
            dq[tid] = g;
            for (int o = S8_ROWS / 2; o > 0; o >>= 1) sq += __shfl_down(sq, o, S8_ROWS);
            keep_g = g;
            keep_a = sq;
        }
        s8_sync();
        s8_head_bwd_inplace(dq, w4c, bufB);   // bufB holds h3 of critic(x, a)
        s8_sync();
        S8_TSTAMP(tl, 19);
        p8_store_rows(bufB, S8_LD, H, Bk.dA3 + row0 * H, H, p);
        p8_layer_c(bufB, S8_LD, ring, rbase, on.wd + ca + lc.w3, on.wd + ca + lc.w2, p, SE_MASK, nullptr, pbuf, bufA, S8_LD, msk[1], nullptr,
                   Bk.dA2 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 20);
        p8_layer_k(bufA, S8_LD, ring, rbase, on.wd + ca + lc.w2, nullptr, L, SE_MASK, nullptr, pbuf, bufB, S8_LD, msk[0], nullptr,
                   Bk.dA1 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 21);
        if (w0 && tid < S8_ROWS) {
            wt_store(Bk.dQA + (row0 + tid) * 16, keep_g);
            if (tid == 0) wt_store(Bk.part + slab, keep_a);
        }
        if (w0 && slab == 0 && tid == 0) {   // Adam step scalars for the optimizer kernel that follows
            Bk.st->step += 1;
            adam_prepare(Bk.st, Bk.adam);
        }
        S8_TSTAMP(tl, 22);
    } else {
        // ---------------------------------------------------------------------- actor side
        const PlanRec rec = s8_plan_rec(A.gs, row0);
        float4 wba[6], wbc[6], wh[4], wq[4], wb4[6];
        float w1n[4];
        s8_small_prefetch(on.wf + la.w1, la.K1, wba);
        s8_small_prefetch(on.wf + ca + lc.w1, lc.K1, wbc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wh[j] = *reinterpret_cast<const float4 *>(on.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
        wq[0] = *reinterpret_cast<const float4 *>(on.canon + ca + lc.w4 + 4 * lane);
        const float bh = on.canon[la.b4 + (lane < ad ? lane : 0)];
        const float bq = on.canon[ca + lc.b4];
        const float w4c = on.canon[ca + lc.w4 + (tid & 255)];
        {
            const float *w1 = on.canon + ca + lc.w1 + (size_t)(tid & 255) * lc.K1 + A.act_off;
#pragma unroll
            for (int j = 0; j < 4; ++j) w1n[j] = w1[j < ad ? j : ad - 1];
        }
        s8_small_prefetch(on.wd + la.w4, 16, wb4);
        __builtin_amdgcn_sched_barrier(0);
        if (A.gs.plan) s8_gather(xin, A.gs, rec, 2, row0, A.ldx, A.act_off, ad, A.max_action, w0 ? A.XP : nullptr);
        else s8_load(xin, S8_LDX, A.ldx, A.XP + row0 * A.ldx, A.ldx);
        p8_prologue(ring, rbase, on.wf + la.w2, p8c_cg(p, wave), p8c_b0(wave));
        s8_sync();
        // actor(x)
        S8_TSTAMP(tl, 1);
        p8_small_layer(xin, S8_LDX, la.K1, wba, SE_BIAS_RELU, on.canon + la.b1, pbuf, bufA, S8_LD, nullptr, msk[2], A.APh1 + row0 * H, p);
        s8_sync();
        S8_TSTAMP(tl, 2);
        p8_layer_c(bufA, S8_LD, ring, rbase, on.wf + la.w2, on.wf + la.w3, p, SE_BIAS_RELU, on.canon + la.b2, pbuf, bufB, S8_LD, nullptr,
                   msk[3], A.APh2 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 3);
        p8_layer_k(bufB, S8_LD, ring, rbase, on.wf + la.w3, on.wf + ca + lc.w2, L, SE_BIAS_RELU, on.canon + la.b3, pbuf, bufA, S8_LD,
                   nullptr, msk[4], A.APh3 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 5);
        float u_mine = 0.f, th_mine = 0.f;
        {   // actor head: tanh -> action block of the critic input (models.py:24, :38); lane j owns output j of row `wave`
            const int rr = wave;
            const float z = s8_rowdots(bufA, S8_LD, rr, ad, wh);
            if (lane < ad) {
                th_mine = tanhf(z + bh);
                u_mine = (A.max_action * th_mine) / A.max_action;
                xin[rr * S8_LDX + A.act_off + lane] = u_mine;
                if (w0) {
                    A.XP[(row0 + rr) * A.ldx + A.act_off + lane] = u_mine;
                    A.TP[(row0 + rr) * 16 + lane] = th_mine;
                }
            }
        }
        S8_TSTAMP(tl, 6);
        s8_sync();
        S8_TSTAMP(tl, 7);
        // critic(x, pi(x))
        p8_small_layer(xin, S8_LDX, lc.K1, wbc, SE_BIAS_RELU, on.canon + ca + lc.b1, pbuf, bufA, S8_LD, nullptr, msk[0], nullptr, p);
        s8_sync();
        S8_TSTAMP(tl, 8);
        p8_layer_c(bufA, S8_LD, ring, rbase, on.wf + ca + lc.w2, on.wf + ca + lc.w3, p, SE_BIAS_RELU, on.canon + ca + lc.b2, pbuf, bufB,
                   S8_LD, nullptr, msk[1], nullptr);
        s8_sync();
        S8_TSTAMP(tl, 9);
        p8_layer_k(bufB, S8_LD, ring, rbase, on.wf + ca + lc.w3, on.wd + ca + lc.w3, L, SE_BIAS_RELU, on.canon + ca + lc.b3, pbuf, bufA,
                   S8_LD, nullptr, nullptr, nullptr);
        s8_sync();
        S8_TSTAMP(tl, 10);
        {
            const int rr = wave;
            const float qv = s8_rowdots(bufA, S8_LD, rr, 1, wq);
            if (lane == 0) {
                rows[1][rr] = qv + bq;
                if (w0) A.QP[(row0 + rr) * 16] = qv + bq;
            }
        }
        if (tid < 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w1t[j * 256 + tid] = w1n[j];
        }
        s8_sync();
        S8_TSTAMP(tl, 13);
        // ---- actor loss (ddpg_agent.py:265-267)
        float keep_q = 0.f, keep_u = 0.f;
        if (tid < S8_ROWS) {
            const size_t m = row0 + tid;
            const bool live = (int)m < Bk.B;
            dq[tid] = live ? -invB : 0.f;
            float sq = live ? rows[1][tid] : 0.f, su = 0.f;
            if (live) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < ad) {
                        const float u = xin[tid * S8_LDX + A.act_off + j];
                        su += u * u;
                    }
            }
            for (int o = S8_ROWS / 2; o > 0; o >>= 1) {
                sq += __shfl_down(sq, o, S8_ROWS);
                su += __shfl_down(su, o, S8_ROWS);
            }
            keep_q = sq;
            keep_u = su;
        }
        s8_sync();
        s8_head_bwd_inplace(dq, w4c, bufA);   // bufA holds h3 of critic(x, pi(x))
        s8_sync();
        S8_TSTAMP(tl, 14);
        p8_layer_c(bufA, S8_LD, ring, rbase, on.wd + ca + lc.w3, on.wd + ca + lc.w2, p, SE_MASK, nullptr, pbuf, bufB, S8_LD, msk[1], nullptr,
                   nullptr);
        s8_sync();
        S8_TSTAMP(tl, 15);
        p8_layer_k(bufB, S8_LD, ring, rbase, on.wd + ca + lc.w2, on.wd + la.w3, L, SE_MASK, nullptr, pbuf, bufA, S8_LD, msk[0], nullptr,
                   nullptr);
        s8_sync();
        S8_TSTAMP(tl, 16);
        {   // d L / d(action block of the critic input), then through the L2 penalty and tanh; lane j owns action j
            float4 w1g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w1g[j] = *reinterpret_cast<const float4 *>(w1t + j * 256 + 4 * lane);
            const int rr = wave;
            const float sj = s8_rowdots(bufA, S8_LD, rr, ad, w1g);
            if (lane < 16) {
                const size_t m = row0 + rr;
                float v = 0.f;
                if (lane < ad && (int)m < Bk.B) {
                    const float gu = Bk.action_l2 * (2.f * u_mine / (float)(Bk.B * ad)) + sj;
                    const float gt = (gu / A.max_action) * A.max_action;
                    v = gt * (1.f - th_mine * th_mine);
                }
                dz[rr * 20 + lane] = v;
                if (w0) wt_store(Bk.dZ + m * 16 + lane, v);
            }
        }
        s8_sync();
        S8_TSTAMP(tl, 17);
        p8_small_layer(dz, 20, 16, wb4, SE_MASK, nullptr, pbuf, bufB, S8_LD, msk[4], nullptr, Bk.dK3 + row0 * H, p);
        s8_sync();
        S8_TSTAMP(tl, 18);
        p8_layer_c(bufB, S8_LD, ring, rbase, on.wd + la.w3, on.wd + la.w2, p, SE_MASK, nullptr, pbuf, bufA, S8_LD, msk[3], nullptr,
                   Bk.dK2 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 19);
        p8_layer_k(bufA, S8_LD, ring, rbase, on.wd + la.w2, nullptr, L, SE_MASK, nullptr, pbuf, bufB, S8_LD, msk[2], nullptr,
                   Bk.dK1 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 20);
        if (w0 && tid == 0) {
            wt_store(Bk.part + nslab + slab, keep_q);
            wt_store(Bk.part + 2 * nslab + slab, keep_u);
        }
        S8_TSTAMP(tl, 21);
    }
}

#endif  // S8_NRG == 2
