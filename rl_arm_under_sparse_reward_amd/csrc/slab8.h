// slab8.h -- row-slab engine on v_mfma_f32_4x4x1_16b_f32, S8_ROWS = 4 * S8_NRG (4, 8 or 16) batch rows per workgroup;
// included by agent_engines.hip once per slab height (the name dates from its 8-row first version).
//
// Why thin slabs: with 16-row slabs on the 16x16x4 MFMA (the first slab engine, removed in round 3) a 256x256 layer costs a workgroup 1024
// MFMAs = 3.4 us of its CU's matrix pipes, and at batch 256 only 48 workgroups exist.  The 4x4x1 instruction (16
// independent 4x4 outer products per issue, measured 8.7 cycles = 92 % of the 16x16x4 FLOP rate,
// tools/ubench/mfma4x4.hip) lets a slab be as thin as 4 rows: lane l of a wavefront owns output column 64*cg + l,
// accumulator register r owns row r, the A operand is the 4 activations of the slab at one reduction index (the same
// for all 16 blocks: block broadcast), the B operand is ONE weight per lane.  The per-workgroup time of a layer is
// set by streaming its 256 KiB of weights through the CU (2.0 us), not by the rows, so thinner slabs cost nothing
// per workgroup, shrink the matrix work that competes with the stream (8 rows: 1.9 us of MFMA issue per layer and
// 2.6 us for both together; 4 rows: 0.95 us and 2.0 us, tools/ubench/stream_bw3.hip modes 4 / 9) and double the
// number of busy CUs (128 at batch 256).
//   wave w of 8:  column group cg = w & 3 (64 output columns),  reduction half kh = w >> 2
//   per block (4 reduction indices): B = float4 per lane (1 KiB, LDS-DMA ring), 4 MFMAs into one accumulator
//   the two reduction halves meet in LDS (4 KiB), the kh == 0 waves run the epilogue.
// Fragment-ordered copies for this engine (same arena offsets as the canonical layout):
//   forward  Wf8[((n >> 6) * K/4 + (k >> 2)) * 256 + (n & 63) * 4 + (k & 3)] = W[n][k]     (layers 1-3)
//   dX       Wd8[((k >> 6) * N/4 + (n >> 2)) * 256 + (k & 63) * 4 + (n & 3)] = W[n][k]     (layers 2-4)
// The 4- and 1-wide heads and the 4 action columns of the critic's input gradient are S8_ROWS x 4 dot products of
// length 256: one wavefront per row, float4 per lane, shuffle tree -- no matrix pipe, canonical weights.
#ifndef RLARM_SLAB8_SHARED
#define RLARM_SLAB8_SHARED


#define S8_THREADS 512
#define S8_WAVES 8
#define S8_LD 260
#define S8_LDX 52
#ifdef SLAB_TIMELINE
#define S8_STAMP(k) do { if (slab == 0 && threadIdx.x == 0) A.tl[chain * 32 + (k)] = wall_clock64(); } while (0)
#define S8_TSTAMP(tl, k) do { if ((tl) && threadIdx.x == 0) (tl)[k] = wall_clock64(); } while (0)
#define S8_WSTAMP(tl, k) do { if ((tl) && (threadIdx.x & 63) == 0) (tl)[(k) + (threadIdx.x >> 6)] = wall_clock64(); } while (0)
#else
#define S8_STAMP(k) do { } while (0)
#define S8_TSTAMP(tl, k) do { } while (0)
#define S8_WSTAMP(tl, k) do { } while (0)
#endif

__host__ __device__ __forceinline__ int frag8_fwd_index(int n, int k, int K) {
    return (((n >> 6) * (K >> 2) + (k >> 2)) << 8) + ((n & 63) << 2) + (k & 3);
}
__host__ __device__ __forceinline__ int frag8_dx_index(int n, int k, int N) {
    return (((k >> 6) * (N >> 2) + (n >> 2)) << 8) + ((k & 63) << 2) + (n & 3);
}

// canonical arena index -> offsets of the slab8 fragment copies (-1: this tensor has none)
__host__ __device__ __forceinline__ void frag8_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d) {
    const bool critic = idx >= am.la.total;
    const NetLayout &l = critic ? am.lc : am.la;
    const int base = critic ? am.la.total : 0;
    const int r = idx - base;
    off_f = off_d = -1;
    int w0, N, K, layer;
    if (r < l.b1) { w0 = l.w1; N = am.H; K = l.K1; layer = 1; }
    else if (r < l.w2) return;
    else if (r < l.b2) { w0 = l.w2; N = am.H; K = am.H; layer = 2; }
    else if (r < l.w3) return;
    else if (r < l.b3) { w0 = l.w3; N = am.H; K = am.H; layer = 3; }
    else if (r < l.w4) return;
    else if (r < l.b4) { w0 = l.w4; N = 16; K = am.H; layer = 4; }
    else return;
    const int e = r - w0, n = e / K, k = e - n * K;
    if (layer <= 3) off_f = base + w0 + frag8_fwd_index(n, k, K);
    if (layer >= 2) off_d = base + w0 + frag8_dx_index(n, k, N);
}

// Chain outputs are stored WRITE-THROUGH (wt_store, agent_device.h): neutral on its own (40.8 vs 41.1 us/update), and what the
// split weight-gradient kernels' partial-tile exchange (dw64.h, gemm_lds.h) needs.  (Round 2 also ran the weight-gradient tiles
// as a second phase of this kernel -- one launch per update: 48.2 vs 40.8 us, removed in round 3; DESIGN.md 3.2,
// profiles/r02_fused_single_launch.txt.)

// arguments of k_fb_slab8 (both row counts)
struct FbSlabArgs {
    FwdSlabArgs f;
    BwdSlabArgs b;
    // Spare workgroups behind the 2 * nslab chain workgroups: n_plan (0/1) draws the index plan of a LATER update
    // (b.next_plan), n_ahead gather the NEXT update's network inputs from its already drawn plan into the other input
    // set, so that the next launch starts from a coalesced load instead of two dependent memory latencies
    // (plan record -> replay-buffer rows).  The plan they read was finished by an EARLIER launch: no in-kernel handshake.
    int n_plan, n_ahead;
    int xcd_split;   // chain = XCD half (needs nslab % 4 == 0)
    int n_pref;      // L2-warmer workgroups (a multiple of 8: the same number on every XCD)
    GatherSrc ahead;             // ahead.plan = plan of the next update; ahead.R = its reward vector
    float *aXT, *aXA, *aXP;      // its input sets (the chains of THIS launch use f.XT / f.XA / f.XP)
};

// Rollout-side policy call (hp_agent_act / hp_agent_actor_forward): one 4-row slab per workgroup through the actor
struct PolicyArgs {
    const double *obs, *g;        // raw float64 rows (hp_agent_act) ...
    const float *x;               // ... or already normalised float32 rows [rows][od + gd] (hp_agent_actor_forward)
    int rows, od, gd;
    const NormDev *onz, *gnz;
    double clip_obs, clip_o, clip_g;
    SlabNetPtrs net;              // forward fragments + canonical arena of the actor to evaluate
    NetLayout la;
    int H, act_dim;
    float max_action;
    float *actions;               // [rows][act_dim]
};

#include "slab8_split_args.h"

#endif  // RLARM_SLAB8_SHARED

// ---- everything below is compiled once per slab height: S8_NRG row groups of 4 (S8_NRG = 1: 4-row slabs, the
// small-batch choice; S8_NRG = 2 / 4: 8- / 16-row slabs, less weight traffic per row for batches that fill the chip),
// each time inside its own namespace S8_NS (agent.hip includes this file twice)
#undef S8_ROWS
#define S8_ROWS (4 * S8_NRG)
#undef S8_RING
#ifndef S8_RING1
#define S8_RING1 12   // ring depth of the 4-row build (-DS8_RING1=14|16 for experiments)
#endif
#ifndef S8_RING2
#define S8_RING2 12   // ring depth of the 8-row build
#endif
#define S8_RING (S8_NRG >= 4 ? 10 : (S8_NRG == 1 ? S8_RING1 : S8_RING2))   // 16-row slabs need the LDS for their activation buffers
#undef S8_BOTH_HALVES
#define S8_BOTH_HALVES (S8_NRG >= 2)   // both reduction halves finish rows (s8_finish); 4-row slabs: the kh 0 waves finish all four
#undef S8_RPW
#define S8_RPW ((S8_ROWS + S8_WAVES - 1) / S8_WAVES)   // rows per wavefront in the one-wavefront-per-row stages
namespace S8_NS {

typedef unsigned short s8_mask_t;   // ReLU mask of one column: one byte per reduction half, bit i = the i-th row that half finishes (s8_finish)

__device__ __forceinline__ void s8_sync() {   // barrier that leaves global loads / DMA in flight (__syncthreads() would also drain them: s_waitcnt vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// weight block b (4 reduction indices) of column group cg in a layer with nb4 blocks per column group
__device__ __forceinline__ const float4 *s8_wblock(const float *wlayer, int cg, int nb4, int b) {
    return reinterpret_cast<const float4 *>(wlayer) + ((size_t)cg * nb4 + b) * 64 + (threadIdx.x & 63);
}

// A operand through the MFMA's block broadcast (cbsz = 4: all 16 blocks use the A values of block `abid`):
// lane l = 4*blk + i keeps x[row group * 4 + i][k0 + 16 j + blk] in register j, so ONE register per lane serves
// 16 reduction indices and a wavefront loads its whole A operand for a layer once (16 ds_read_b32) instead of
// two ds_read_b128 per super-step -- the LDS pipe, shared with the weight DMA, was the limiter.
// TB = (block index within the wave's half) % 4 selects which quarter of register j the 4 indices of a block hit.
template <int TB>
__device__ __forceinline__ void s8_mma(f32x4 (&c)[S8_NRG], const float (&a)[S8_NRG], const float4 b) {
    const float bk[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < S8_NRG; ++g) {   // row groups alternate: independent accumulators back to back
            if (k == 0) c[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], bk[0], c[g], 4, 4 * TB + 0, 0);
            if (k == 1) c[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], bk[1], c[g], 4, 4 * TB + 1, 0);
            if (k == 2) c[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], bk[2], c[g], 4, 4 * TB + 2, 0);
            if (k == 3) c[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], bk[3], c[g], 4, 4 * TB + 3, 0);
        }
}

// load this wave's A operand for NJ groups of 16 reduction indices starting at index k0
template <int NJ>
__device__ __forceinline__ void s8_aload(const float *lin, int ld_in, int k0, float (&a)[8][S8_NRG]) {
    const int i = threadIdx.x & 3, blk = (threadIdx.x & 63) >> 2;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < S8_NRG; ++g) a[j][g] = lin[(4 * g + i) * ld_in + k0 + 16 * j + blk];
}

// The ring is CONTINUOUS across the 256x256 layers of a chain: block t of the current layer lives in slot
// (rbase + t) % S8_RING, and as soon as the current layer has no block left to issue the freed slots take the
// first blocks of the next layer (slot arithmetic stays consistent with rbase' = rbase + 32).  The DMA queue
// therefore never drains at a layer boundary.
// LDS-DMA of one 1 KiB block (16 B per lane).  Note for whoever tunes this next: the builtin is a FLAT-encoded
// instruction that touches memory and LDS, and while one is pending hipcc's wait-count pass turns every wait it
// inserts into vmcnt(0) / lgkmcnt(0) ("pending flat").  Issuing the transfer from inline assembly (s_mov_b32 m0 +
// global_load_lds_dwordx4) makes the compiler's counts exact -- verified in the ISA -- but measured no faster
// (57.1 vs 57.3 us per update) and needs M0 on the clobber list, which hipcc flags as reserved; the builtin stays.
__device__ __forceinline__ void s8_dma16(const void *gsrc, void *lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, lds_dst, 16, 0, 0);
}

__device__ __forceinline__ void s8_ring_issue(RingSlot *ring, int rbase, const float *wlayer, int cg, int b0, int t) {
    s8_dma16(s8_wblock(wlayer, cg, 64, b0 + t), &ring[(rbase + t) % S8_RING][0]);
}

// first S8_RING blocks of this wave's half of a 256-reduction layer (start of a chain)
template <int T0 = 0, int T1 = S8_RING>
__device__ __forceinline__ void s8_ring_prologue(RingSlot *ring, int rbase, const float *wlayer) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cg = wave & 3, b0 = (wave >> 2) * 32;
#pragma unroll
    for (int t = T0; t < T1; ++t) s8_ring_issue(ring, rbase, wlayer, cg, b0, t);
}
// blocks of the chain's first ring issued in front of the input loads (the rest behind them).  us/update with 0 / 4 / 8 / 12 in
// front: 40.09 / 39.70 / 39.56 / 39.79 at batch 256 and 46.12 / 45.00 / 44.69 / 45.30 at 512 k8 (4 rows), 54.63 / 54.87 / 54.76 /
// 54.29 at 1024 (8 rows)
#undef S8_PRO_FIRST
#define S8_PRO_FIRST (S8_NRG == 1 ? 8 : S8_RING)

// Block T of the 32 this wave consumes.  The weight operand is software-pipelined through registers: the LDS read of
// block T+1 is issued BEFORE the 8 MFMAs of block T, so its latency hides under this wave's own matrix work instead
// of being exposed once per block (with two waves per SIMD the other wave covered only part of it).
// (Store-aware waits -- allowing the write-through copies a wave issued at the end of the previous stage to stay outstanding
// while the blocks that were in flight before them are consumed -- were built and measured SLOWER: 55.7 vs 54.1 us/update at batch
// 1024, 46.2 vs 45.3 at 512 k8, 40.6 vs 40.3 at 256.  The stricter waits stay.)
template <int T, bool HAS_NEXT>
__device__ __forceinline__ void s8_ring_step(f32x4 (&c)[S8_NRG], RingSlot *ring, int rbase, const float *wlayer,
                                             const float *nxt, int cg, int b0, const float (&a)[8][S8_NRG],
                                             const float4 bcur) {
    float4 bnext = bcur;
    if constexpr (T + 1 < 32) {
        // DMA blocks possibly outstanding here: T+1 .. T+R-1 (fewer at the tail of a chain's last layer)
        constexpr int out = HAS_NEXT ? S8_RING - 1 : ((31 - T) < S8_RING - 1 ? (31 - T) : S8_RING - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(out - 1) : "memory");
        bnext = ring[(rbase + T + 1) % S8_RING][threadIdx.x & 63];
    }
    if constexpr (T + S8_RING < 32 || HAS_NEXT) {
        // slot of block T is free once ITS read (one LDS op older than bnext's) has returned
        if constexpr (T + 1 < 32) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (T + S8_RING < 32) s8_ring_issue(ring, rbase, wlayer, cg, b0, T + S8_RING);
        else s8_ring_issue(ring, rbase + 32, nxt, cg, b0, T + S8_RING - 32);
    }
    s8_mma<T % 4>(c, a[T / 4], bcur);
    if constexpr (T + 1 < 32) s8_ring_step<T + 1, HAS_NEXT>(c, ring, rbase, wlayer, nxt, cg, b0, a, bnext);
}

// combine the two reduction halves and run the epilogue.  c: this wave's partial [row][col = lane]
// gout (may be null): global [rows][256] copy of the output.
// mask_out (SE_BIAS_RELU, may be null): LDS word per column = the ReLU mask the backward stages of the SAME workgroup need;
// mask_in (SE_MASK, may be null): use such a word instead of the gate values in e[].
// BOTH halves finish: the kh-th wave of a column group owns half of the slab's rows (row groups [kh NRG/2, (kh + 1) NRG/2); with
// one row group its rows 2 kh, 2 kh + 1), hands the partial sums of the OTHER rows to its partner through pbuf and, behind the
// barrier, adds the partner's sums to its own rows (kh 0: c + partner, kh 1: partner + c -- the same two addends, the sum is kh 0's
// + kh 1's either way), applies the epilogue and writes them out.  Until round 5 the kh 0 waves did all of it while the kh 1 waves
// waited: at 8 rows the layer's tail was 1.2 us (profiles/r05_ab_chain_small_levers.txt).  A mask word keeps one BYTE per half
// (bit 8 kh + index of the row among the half's rows), so the two waves of a column write different bytes.
template <int HH>
__device__ __forceinline__ void s8_finish_half(const f32x4 (&c)[S8_NRG], int epi, const float *e, float *pbuf, float *lout,
                                               int ld_out, const s8_mask_t *mask_in, s8_mask_t *mask_out, float *gout, int col,
                                               unsigned long long *wtl) {
    constexpr int NOWN = 2 * S8_NRG;                     // rows a half owns
    auto own_g = [](int i) { return S8_NRG == 1 ? 0 : HH * (S8_NRG / 2) + i / 4; };        // i-th owned row -> row group, row in group
    auto own_r = [](int i) { return S8_NRG == 1 ? 2 * HH + i : i % 4; };
    auto oth_g = [](int i) { return S8_NRG == 1 ? 0 : (1 - HH) * (S8_NRG / 2) + i / 4; };
    auto oth_r = [](int i) { return S8_NRG == 1 ? 2 * (1 - HH) + i : i % 4; };
#pragma unroll
    for (int i = 0; i < NOWN; ++i) pbuf[(4 * oth_g(i) + oth_r(i)) * 256 + col] = c[oth_g(i)][oth_r(i)];
    s8_sync();
    S8_WSTAMP(wtl, 16);
    const unsigned bits = mask_in ? ((unsigned)mask_in[col] >> (8 * HH)) & 0xffu : 0u;
    unsigned outbits = 0u;
    // the partner's sums are all asked for before the first output is written: pbuf and lout are both LDS pointers the compiler
    // cannot tell apart, so a read behind a write would wait for it
    float pv[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) pv[i] = pbuf[(4 * own_g(i) + own_r(i)) * 256 + col];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
        const int g = own_g(i), r = own_r(i), row = 4 * g + r;
        const float v = HH == 0 ? c[g][r] + pv[i] : pv[i] + c[g][r];
        float o;
        if (epi == SE_BIAS_RELU) {
            o = fmaxf(v + e[0], 0.f);
            outbits |= (o > 0.f ? 1u : 0u) << i;
        } else if (mask_in) {
            o = ((bits >> i) & 1u) ? v : 0.f;
        } else {
            o = (e[row & 7] > 0.f) ? v : 0.f;
        }
        lout[row * ld_out + col] = o;
        // global copy for the weight-gradient GEMM straight from the register (a wavefront writes 64
        // consecutive floats of one row): no second pass over the LDS slab before the next layer can start
        if (gout) wt_store(gout + (size_t)row * 256 + col, o);   // operand of the weight-gradient tiles
    }
    if (mask_out) reinterpret_cast<unsigned char *>(mask_out)[2 * col + HH] = (unsigned char)outbits;
}
__device__ __forceinline__ void s8_finish(const f32x4 (&c)[S8_NRG], int epi, const float *e, float *pbuf, float *lout,
                                          int ld_out, const s8_mask_t *mask_in = nullptr,
                                          s8_mask_t *mask_out = nullptr, float *gout = nullptr,
                                          unsigned long long *wtl = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, cg = wave & 3, kh = wave >> 2;
    const int col = 64 * cg + lane;
    static_assert(2 * S8_NRG <= 8 && sizeof(s8_mask_t) == 2, "a half's rows fit one byte of a mask word");
#if S8_BOTH_HALVES
    if (kh == 0) s8_finish_half<0>(c, epi, e, pbuf, lout, ld_out, mask_in, mask_out, gout, col, wtl);
    else s8_finish_half<1>(c, epi, e, pbuf, lout, ld_out, mask_in, mask_out, gout, col, wtl);
#else
    // 4-row slabs: the kh 0 waves finish all four rows (measured with both halves finishing two each: 37.16 vs 37.13 us/update at batch
    // 256, 45.03 vs 44.42 at 512 k8 -- the layer is bound by its weight transfers there and four more waves issuing stores only add to
    // them; 8 rows: 51.9 vs 52.4 at batch 1024, 16 rows: 85.7 vs 88.8 at 2048)
    if (kh == 1) {
#pragma unroll
        for (int g = 0; g < S8_NRG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[(4 * g + r) * 256 + col] = c[g][r];
    }
    s8_sync();
    S8_WSTAMP(wtl, 16);
    if (kh == 0) {
        unsigned bits = mask_in ? (unsigned)mask_in[col] : 0u, outbits = 0u;
        // the other half's partial sums are all asked for before the first output is written: pbuf and lout are both LDS pointers the
        // compiler cannot tell apart, so a read behind a write waited for it -- S8_ROWS dependent LDS round trips per layer
        float pv[4][S8_NRG];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < S8_NRG; ++g) pv[r][g] = pbuf[(4 * g + r) * 256 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < S8_NRG; ++g) {
                const int row = 4 * g + r;
                const float v = c[g][r] + pv[r][g];
                float o;
                if (epi == SE_BIAS_RELU) {
                    o = fmaxf(v + e[0], 0.f);
                    outbits |= (o > 0.f ? 1u : 0u) << row;
                } else if (mask_in) {
                    o = ((bits >> row) & 1u) ? v : 0.f;
                } else {
                    o = (e[row & 7] > 0.f) ? v : 0.f;
                }
                lout[row * ld_out + col] = o;
                // global copy for the weight-gradient GEMM straight from the register (a wavefront writes 64
                // consecutive floats of one row): no second pass over the LDS slab before the next layer can start
                if (gout) wt_store(gout + (size_t)row * 256 + col, o);   // operand of the weight-gradient tiles
            }
        if (mask_out) mask_out[col] = (s8_mask_t)outbits;
    }
#endif
}

// epilogue operands of this wave's column (both reduction halves finish rows): bias, or the 8 gate values
__device__ __forceinline__ void s8_epi_load(float (&e)[8], int epi, const float *__restrict__ aux, int ldaux) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = 64 * (wave & 3) + (threadIdx.x & 63);
    if (!S8_BOTH_HALVES && (wave >> 2) != 0) return;
    if (epi == SE_BIAS_RELU) {
        e[0] = aux[col];
    } else {
        // (gate values instead of a mask word: slabs of at most 8 rows; every caller in this engine passes masks)
#pragma unroll
        for (int r = 0; r < (S8_ROWS < 8 ? S8_ROWS : 8); ++r) e[r] = aux[(size_t)r * ldaux + col];
    }
}

// out[8][256] = epi(in[8][256] . W): DMA-ring fed.  The first S8_RING blocks of `wlayer` must be in flight at
// ring base `rbase`; on return the first S8_RING blocks of `nxt` are (if nxt != nullptr) and rbase has advanced.
__device__ __forceinline__ void s8_big_layer(const float *lin, int ld_in, RingSlot *ring, int &rbase,
                                             const float *__restrict__ wlayer, const float *__restrict__ nxt, int epi,
                                             const float *__restrict__ aux, int ldaux, float *pbuf, float *lout,
                                             int ld_out, const s8_mask_t *mask_in = nullptr,
                                             s8_mask_t *mask_out = nullptr, unsigned long long *tl2 = nullptr,
                                             int k2 = 0, float *gout = nullptr, const float *pre_e = nullptr,
                                             unsigned long long *wtl = nullptr) {   // wtl: per-wave stamps (time-line build)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cg = wave & 3, b0 = (wave >> 2) * 32;
    float e[8];
    if (pre_e) e[0] = *pre_e;   // bias loaded at the trunk's start (s8_trunk)
    else if (!mask_in) s8_epi_load(e, epi, aux, ldaux);
    __builtin_amdgcn_sched_barrier(0);
    S8_TSTAMP(tl2, k2);
    S8_WSTAMP(wtl, 0);
    f32x4 c[S8_NRG];
#pragma unroll
    for (int g = 0; g < S8_NRG; ++g) c[g] = f32x4{0, 0, 0, 0};
    float a[8][S8_NRG];
    s8_aload<8>(lin, ld_in, 4 * b0, a);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S8_RING - 1) : "memory");   // block 0 has landed
    const float4 bfirst = ring[rbase % S8_RING][threadIdx.x & 63];
    if (nxt) s8_ring_step<0, true>(c, ring, rbase, wlayer, nxt, cg, b0, a, bfirst);
    else s8_ring_step<0, false>(c, ring, rbase, wlayer, nxt, cg, b0, a, bfirst);
    rbase = (rbase + 32) % S8_RING;
    __builtin_amdgcn_sched_barrier(0);
    S8_TSTAMP(tl2, k2 + 1);
    S8_WSTAMP(wtl, 8);
    s8_finish(c, epi, e, pbuf, lout, ld_out, mask_in, mask_out, gout, wtl);
    S8_TSTAMP(tl2, k2 + 2);
}

template <int T, int HALF>
__device__ __forceinline__ void s8_small_steps(f32x4 (&c)[S8_NRG], const float4 (&b)[6], const float (&a)[8][S8_NRG]) {
    s8_mma<T % 4>(c, a[T / 4], b[T]);
    if constexpr (T + 1 < HALF) s8_small_steps<T + 1, HALF>(c, b, a);
}

// All weight blocks of a small layer (reduction length Kred = 16 / 32 / 48 -> 2 / 4 / 6 blocks per reduction half) in
// ONE batch of loads, issued by the caller as early as it can (kernel entry): the loads are cold, and left to the
// compiler they end up as load -> wait -> MFMA once per block with one destination register.
__device__ __forceinline__ void s8_small_prefetch(const float *__restrict__ wlayer, int Kred, float4 (&b)[6]) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cg = wave & 3, kh = wave >> 2;
    const int nb4 = Kred >> 2, half = nb4 >> 1, b0 = kh * half;
    // branch free (blocks past `half` reload the last one): with no control flow between the loads the compiler
    // can count them, and waits for OLDER loads become vmcnt(n) instead of vmcnt(0)
#pragma unroll
    for (int t = 0; t < 6; ++t) b[t] = *s8_wblock(wlayer, cg, nb4, b0 + (t < half ? t : half - 1));
}

// small layer: weights already in registers (s8_small_prefetch), no ring
__device__ __forceinline__ void s8_small_layer(const float *lin, int ld_in, int Kred, const float4 (&b)[6], int epi,
                                               const float *__restrict__ aux, int ldaux, float *pbuf, float *lout,
                                               int ld_out, const s8_mask_t *mask_in = nullptr,
                                               s8_mask_t *mask_out = nullptr, float *gout = nullptr,
                                               const float *pre_e = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = wave >> 2;
    const int nb4 = Kred >> 2, half = nb4 >> 1, b0 = kh * half;
    float e[8];
    if (pre_e) e[0] = *pre_e;   // bias loaded at kernel entry (k_fb_slab8)
    else if (!mask_in) s8_epi_load(e, epi, aux, ldaux);
    f32x4 c[S8_NRG];
#pragma unroll
    for (int g = 0; g < S8_NRG; ++g) c[g] = f32x4{0, 0, 0, 0};
    float a[8][S8_NRG];
    s8_aload<2>(lin, ld_in, 4 * b0, a);   // at most 24 indices per half; lanes past the row end read unused padding
    switch (half) {
        case 2: s8_small_steps<0, 2>(c, b, a); break;
        case 4: s8_small_steps<0, 4>(c, b, a); break;
        case 6: s8_small_steps<0, 6>(c, b, a); break;
        default: break;   // other input widths are rejected on the host
    }
    s8_finish(c, epi, e, pbuf, lout, ld_out, mask_in, mask_out, gout);
}

// nout (<= 4) dot products of length 256 for ONE row (wavefront-wide; callers walk rows wave, wave + 8, ... and clamp
// the row index for wavefronts that have none): lane p takes the reduction indices 4p..4p+3;
// wv[j] = this lane's float4 of weights for output j (loaded by the caller well ahead of time).  On return lane j
// (j < nout) holds output j of the wave's row (other lanes: unspecified).
// lane i += lane i + O for O in {8, 4, 2, 1}: inside a 16-lane row this is a DPP row shift (VALU speed) instead of a
// trip through the LDS crossbar; lanes whose partner falls outside the row add 0, which only affects lanes the tree
// never reads.  Same pairs as __shfl_down, so the same bits in lane 0 of each row.
template <int O>
__device__ __forceinline__ float s8_row_shl_add(float s) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x100 + O, 0xf, 0xf, true);   // row_shl:O, zero fill
    return s + __int_as_float(moved);
}
// sum over the S8_ROWS first lanes of a wavefront into lane 0 (the per-slab loss partials): the pairs of __shfl_down(s, o, S8_ROWS) for
// o = S8_ROWS / 2 .. 1, as DPP row shifts (S8_ROWS <= 16: one DPP row)
__device__ __forceinline__ float s8_rows_sum_to_lane0(float s) {
    if (S8_ROWS >= 16) s = s8_row_shl_add<8>(s);
    if (S8_ROWS >= 8) s = s8_row_shl_add<4>(s);
    s = s8_row_shl_add<2>(s);
    s = s8_row_shl_add<1>(s);
    return s;
}
// lanes 32-63 onto lanes 0-31 and rows 1 / 3 onto rows 0 / 2 with gfx950's lane-swap VALU instructions instead of two trips through the
// LDS crossbar (ds_bpermute): v_permlane32_swap(a, a) leaves {lo, lo} and {hi, hi}, so lane i of the sum holds s[i % 32] + s[i % 32 + 32]
// -- for i < 32 exactly what s += __shfl_down(s, 32) left there (same two addends) -- and likewise one row further down.
__device__ __forceinline__ float s8_fold_halves(float s) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float s8_fold_rows(float s) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
// lane 0's value to every lane: callers run with all 64 lanes active, so the first active lane IS lane 0 (one SALU move, no crossbar)
__device__ __forceinline__ float s8_lane0(float s) {
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s)));
}
__device__ __forceinline__ float s8_wave_sum_to_lane0(float s) {
    s = s8_fold_halves(s);
    s = s8_fold_rows(s);
    s = s8_row_shl_add<8>(s);
    s = s8_row_shl_add<4>(s);
    s = s8_row_shl_add<2>(s);
    s = s8_row_shl_add<1>(s);
    return s;
}

__device__ __forceinline__ float s8_rowdots(const float *lin, int ld_in, int row, int nout, const float4 (&wv)[4]) {
    const int p = threadIdx.x & 63;
    const float4 h = *reinterpret_cast<const float4 *>(lin + row * ld_in + 4 * p);
    if (nout == 1) {
        const float s = s8_wave_sum_to_lane0((h.x * wv[0].x + h.y * wv[0].y) + (h.z * wv[0].z + h.w * wv[0].w));
        return s8_lane0(s);
    }
    // several outputs: the reduction trees advance TOGETHER (independent cross-lane moves in flight per level instead
    // of one dependent chain per output; same tree per output, so the same bits).  wv[j] for j >= nout holds a valid
    // duplicate row (callers clamp), its result is dropped.
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (h.x * wv[j].x + h.y * wv[j].y) + (h.z * wv[j].z + h.w * wv[j].w);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = s8_fold_halves(s[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = s8_fold_rows(s[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = s8_row_shl_add<1>(s8_row_shl_add<2>(s8_row_shl_add<4>(s8_row_shl_add<8>(s[j]))));
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float tot = s8_lane0(s[j]);
        if (p == j && j < nout) mine = tot;
    }
    return mine;
}

__device__ __forceinline__ void s8_store(const float *l, int ld, int width, float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < S8_ROWS * per_row; f += S8_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        wt_store4(g + (size_t)r * ldg + 4 * c4, *reinterpret_cast<const float4 *>(l + r * ld + 4 * c4));
    }
}

__device__ __forceinline__ void s8_load(float *l, int ld, int width, const float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < S8_ROWS * per_row; f += S8_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        *reinterpret_cast<float4 *>(l + r * ld + 4 * c4) = *reinterpret_cast<const float4 *>(g + (size_t)r * ldg + 4 * c4);
    }
}

// HER gather for the rows of a slab (same arithmetic as k_gather_fused); one wavefront per row
// this thread's row of the index plan: the first load of the kernel (everything else in the gather depends on it).
// Unconditional (rows past the batch re-read the last record, plan_any is never null): a load under a branch is
// merged with its default through a register copy, which makes the compiler wait for it on the spot.
__device__ __forceinline__ PlanRec s8_plan_rec(const GatherSrc &G, size_t row0, int r = -1) {
    if (r < 0) r = (int)((threadIdx.x >> 6) & (S8_ROWS - 1));   // this wavefront's first row
    const int m = (int)row0 + (r < S8_ROWS ? r : S8_ROWS - 1);
    return G.plan_any[m < G.B ? m : G.B - 1];
}

__device__ __forceinline__ void s8_gather(float *xin, const GatherSrc &G, const PlanRec rec, int which, size_t row0, int ldx,
                                          int act_off, int act_dim, float max_action, float *Xout,
                                          float *rew_lds = nullptr) {
    const int l = threadIdx.x & 63;
    for (int r = threadIdx.x >> 6, it = 0; r < S8_ROWS; r += S8_WAVES, ++it) {   // one wavefront per row
    const PlanRec rc = (it == 0) ? rec : s8_plan_rec(G, row0, r);
    const size_t m = row0 + r;
    const bool live = (int)m < G.B;
    const long long e = rc.e;
    const int t = rc.t, od = G.obs_dim, gd = G.goal_dim;
    const double *obs_row = G.obs + (e * (G.T + 1) + t + (which == 0 ? 1 : 0)) * od;
    const double *g_src = rc.her ? G.ag + (e * (G.T + 1) + rc.fut) * gd : G.g + (e * G.T + t) * gd;
    if (l < ldx) {
        const int c = l;
        float x = 0.f;
        if (live) {
            if (c < od) {
                double v = fmin(fmax(obs_row[c], -G.clip_obs), G.clip_obs);
                v = __ddiv_rn(__dsub_rn(v, (double)G.onz->mean[c]), G.onz->std[c]);
                x = (float)fmin(fmax(v, -G.clip_range), G.clip_range);
            } else if (c < od + gd) {
                const int j = c - od;
                double v = fmin(fmax(g_src[j], -G.clip_obs), G.clip_obs);
                v = __ddiv_rn(__dsub_rn(v, (double)G.gnz->mean[j]), G.gnz->std[j]);
                x = (float)fmin(fmax(v, -G.clip_range), G.clip_range);
            } else if (which == 1 && c >= act_off && c < act_off + act_dim) {
                x = (float)G.act[(e * G.T + t) * act_dim + (c - act_off)] / max_action;
            }
        }
        xin[r * S8_LDX + c] = x;
        if (Xout && (which == 1 || c < act_off)) wt_store(Xout + m * ldx + c, x);
    }
    if (which == 1 && l == 63) {
        float rew = 0.f;
        if (live) {
            const double *ag_next = G.ag + (e * (G.T + 1) + t + 1) * gd;
            double s = 0.0;
            for (int c = 0; c < gd; ++c) {
                const double d = __dsub_rn(ag_next[c], g_src[c]);
                const double sq = __dmul_rn(d, d);
                s = (c == 0) ? sq : __dadd_rn(s, sq);
            }
            rew = hp_reward(s, G.sq_threshold);
        }
        G.R[m] = rew;
        if (rew_lds) rew_lds[r] = rew;
    }
    }
}

// xin (K1 wide) -> h1 -> h2 -> h3.  Ring: layer 2 in flight on entry, `nxt` on exit.  m1..m3 (LDS, may be null):
// ReLU masks of h1..h3 for the backward stages of a merged kernel.
__device__ __forceinline__ void s8_trunk(const float *xin, const NetLayout &l, const float4 (&wb1)[6], const float *wf,
                                         const float *canon, int H,
                                         float *bufA, float *bufB, float *pbuf, float *g1, float *g2, float *g3,
                                         size_t row0, RingSlot *ring, int &rbase, const float *nxt,
                                         unsigned long long *tl, int tbase, s8_mask_t *m1 = nullptr,
                                         s8_mask_t *m2 = nullptr, s8_mask_t *m3 = nullptr, const float *pre3 = nullptr) {
    S8_TSTAMP(tl, tbase);
    // the biases of the two 256 x 256 layers come in now: a global load issued at a layer's start is younger than the ring's
    // transfers in flight and, loads retiring in order, would make every counted wait of that layer one block stricter
    // (-0.3 us/update at batch 256, -0.6 at 1024)
    float eb2 = 0.f, eb3 = 0.f;
    if (pre3) {   // k_fb_slab8 loaded all three at kernel entry: nothing comes in from global memory at a trunk's start
        eb2 = pre3[1];
        eb3 = pre3[2];
    } else {
        const int w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col_ = 64 * (w_ & 3) + (threadIdx.x & 63);
        if (S8_BOTH_HALVES || (w_ >> 2) == 0) { eb2 = canon[l.b2 + col_]; eb3 = canon[l.b3 + col_]; }
    }
    const float *pe2 = &eb2, *pe3 = &eb3;
    s8_small_layer(xin, S8_LDX, l.K1, wb1, SE_BIAS_RELU, canon + l.b1, 0, pbuf, bufA, S8_LD, nullptr, m1,
                   g1 ? g1 + row0 * H : nullptr, pre3);
    s8_sync();
    S8_TSTAMP(tl, tbase + 1);
    s8_big_layer(bufA, S8_LD, ring, rbase, wf + l.w2, wf + l.w3, SE_BIAS_RELU, canon + l.b2, 0, pbuf, bufB, S8_LD, nullptr, m2,
                 tl, 24, g2 ? g2 + row0 * H : nullptr, pe2);
    s8_sync();
    S8_TSTAMP(tl, tbase + 2);
    s8_big_layer(bufB, S8_LD, ring, rbase, wf + l.w3, nxt, SE_BIAS_RELU, canon + l.b3, 0, pbuf, bufA, S8_LD, nullptr, m3, nullptr,
                 0, g3 ? g3 + row0 * H : nullptr, pe3);
    s8_sync();
    S8_TSTAMP(tl, tbase + 3);
}

// ===================================================================================================================
// Merged forward + backward: one launch per update instead of two.
//   chain 0 (critic side):  actor_target -> critic_target -> Q';  critic(x, a) -> Q;  critic loss;  critic dX chain
//   chain 1 (actor side):   actor -> critic(x, pi(x)) -> Q_pi;  actor loss;  dX through the critic and the actor
//   spare workgroups: index plan of a later update, gather of the next update's inputs (FbSlabArgs)
// Both chains are 8 256x256 layers long, no workgroup waits for another.  What the split kernels hand over through
// global memory stays on chip here: Q / Q' / reward / action / tanh in LDS or registers, the top hidden layer in the
// LDS slab it was computed in, the ReLU masks as one byte per column (s8_finish); the weight ring runs on from the
// forward fragment copies into the dX copies without draining.  Arithmetic and summation order are those of
// the separate forward and backward kernels this one replaced (same device functions): it reproduced their results bit for bit.

// HER gather of rows [g * per, (g + 1) * per) of a minibatch into global input sets (same arithmetic as s8_gather:
// her.py:26-38, ddpg_agent.py:228-243, normalizer.py:67-70).  One wavefront per row, 4 rows in flight per wavefront.
__device__ __forceinline__ void s8_gather_ahead(const GatherSrc &G, float *XT, float *XA, float *XP, int ldx, int act_off,
                                                int act_dim, float max_action, int g, int ng) {
    const int wave = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int od = G.obs_dim, gd = G.goal_dim;
    const int per = (G.B + ng - 1) / ng, r_begin = g * per, r_end = (r_begin + per < G.B) ? r_begin + per : G.B;
    // this lane's normalizer statistics do not depend on the row
    const bool is_obs = c < od, is_goal = !is_obs && c < od + gd, is_act = c >= act_off && c < act_off + act_dim;
    float mu = 0.f;
    double sd = 1.0;
    if (is_obs) { mu = G.onz->mean[c]; sd = G.onz->std[c]; }
    if (is_goal) { mu = G.gnz->mean[c - od]; sd = G.gnz->std[c - od]; }
    for (int base = r_begin + wave * 4; base < r_end; base += S8_WAVES * 4) {
        PlanRec rec[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[k] = G.plan[(base + k < r_end) ? base + k : r_end - 1];
        double v0[4], v1[4], an[4], gs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = rec[k].e;
            const int t = rec[k].t;
            const double *obs0 = G.obs + (e * (G.T + 1) + t) * od;
            const double *g_src = rec[k].her ? G.ag + (e * (G.T + 1) + rec[k].fut) * gd : G.g + (e * G.T + t) * gd;
            const double *ag_next = G.ag + (e * (G.T + 1) + t + 1) * gd;
            // address-selected, unconditional loads (padding lanes re-read element 0)
            const double *p0 = is_obs ? obs0 + od + c : (is_goal ? g_src + (c - od) : obs0);
            const double *p1 = is_obs ? obs0 + c : (is_goal ? g_src + (c - od) : (is_act ? G.act + (e * G.T + t) * act_dim + (c - act_off) : obs0));
            v0[k] = *p0;
            v1[k] = *p1;
            const int cc = c < gd ? c : gd - 1;   // lanes 0..gd-1 carry the reward operands
            an[k] = ag_next[cc];
            gs[k] = g_src[cc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = base + k;
            if (m >= r_end) continue;
            if (c < ldx) {
                float x0 = 0.f, x1 = 0.f;
                if (is_obs || is_goal) {
                    double a = fmin(fmax(v0[k], -G.clip_obs), G.clip_obs);
                    a = __ddiv_rn(__dsub_rn(a, (double)mu), sd);
                    x0 = (float)fmin(fmax(a, -G.clip_range), G.clip_range);
                    double b = fmin(fmax(v1[k], -G.clip_obs), G.clip_obs);
                    b = __ddiv_rn(__dsub_rn(b, (double)mu), sd);
                    x1 = (float)fmin(fmax(b, -G.clip_range), G.clip_range);
                } else if (is_act) {
                    x1 = (float)v1[k] / max_action;
                }
                if (c < act_off) {
                    XT[(size_t)m * ldx + c] = x0;
                    XP[(size_t)m * ldx + c] = x1;
                }
                XA[(size_t)m * ldx + c] = x1;
            }
            // reward: d^2 summed in index order by lane 0 (values of lanes 0..gd-1 via shuffles, same operation order
            // as the in-kernel gather)
            const double d = __dsub_rn(an[k], gs[k]);
            const double sq = __dmul_rn(d, d);
            double ssum = 0.0;
            for (int j = 0; j < gd; ++j) {
                const double sj = __shfl(sq, j);
                ssum = (j == 0) ? sj : __dadd_rn(ssum, sj);
            }
            if (c == 0) G.R[m] = hp_reward(ssum, G.sq_threshold);
        }
    }
}

// Backward step of a 1-wide head in place: buf[r][c] = relu'(h) * dQ[r] * w4[c] (H == 256: a thread owns column c of rows r0, r0 + 2, ...),
// every LDS read before the first write (a read behind a write to an LDS pointer the compiler cannot tell apart waits for it).
// Actor loss (ddpg_agent.py:265): dQ of a live row is d_live = -1 / B, known without a trip through the LDS
__device__ __forceinline__ void s8_head_bwd_inplace_rows(float d_live, int row0, int B, float w4c, float *buf) {
    const int c = threadIdx.x & 255, r0 = threadIdx.x >> 8;
    float h[S8_ROWS / 2];
#pragma unroll
    for (int i = 0; i < S8_ROWS / 2; ++i) h[i] = buf[(r0 + 2 * i) * S8_LD + c];
#pragma unroll
    for (int i = 0; i < S8_ROWS / 2; ++i) {
        const float d = (row0 + r0 + 2 * i < B) ? d_live : 0.f;
        buf[(r0 + 2 * i) * S8_LD + c] = (h[i] > 0.f) ? d * w4c : 0.f;
    }
}

// the same for the critic loss (ddpg_agent.py:255-263): every thread derives dL/dQ of its rows from the per-row scalars already in the LDS
// (Q', Q, reward: written before the barrier in front of this stage) with the expression the loss block uses, instead of waiting behind
// a barrier for that block to hand it over
__device__ __forceinline__ void s8_head_bwd_inplace_td(const float (*rows)[S8_ROWS], int row0, int B, float gamma, float clip_ret,
                                                       float invB, float w4c, float *buf) {
    const int c = threadIdx.x & 255, r0 = threadIdx.x >> 8;
    float h[S8_ROWS / 2], qt[S8_ROWS / 2], qa[S8_ROWS / 2], rw[S8_ROWS / 2];
#pragma unroll
    for (int i = 0; i < S8_ROWS / 2; ++i) {
        const int r = r0 + 2 * i;
        h[i] = buf[r * S8_LD + c];
        qt[i] = rows[0][r]; qa[i] = rows[1][r]; rw[i] = rows[2][r];
    }
#pragma unroll
    for (int i = 0; i < S8_ROWS / 2; ++i) {
        float g = 0.f;
        if (row0 + r0 + 2 * i < B) {
            float y = rw[i] + gamma * qt[i];
            y = fminf(fmaxf(y, -clip_ret), 0.f);
            const float d = y - qa[i];
            g = -2.f * d * invB;
        }
        buf[(r0 + 2 * i) * S8_LD + c] = (h[i] > 0.f) ? g * w4c : 0.f;
    }
}

// L2 warmer `widx` (of P.n_pref: a multiple of 8, the same number on every XCD) of this workgroup's XCD: touches the weight
// fragments the XCD's chains will stream, in the order they use them, one dword per 128-byte line, so that the chains find them
// in their L2 instead of behind the fabric
// side: 0 = the sets of the critic-side chains, 1 = of the actor-side chains, 2 = all; per / mine: warmers on this XCD, my index
__device__ __forceinline__ void s8_l2_warm_at(const FbSlabArgs &P, int side, int per, int mine, float *sink) {
    const FwdSlabArgs &A = P.f;
    const int tid = threadIdx.x;
    const int na = A.la.total, nall = na + A.lc.total;
    const float *r0 = side == 1 ? A.online.wf : A.target.wf;
    const int n0 = nall;
    const float *r1 = side == 1 ? A.online.wd + na : A.online.wf + (side == 0 ? na : 0);
    const int n1 = side == 2 ? nall : A.lc.total;
    const float *r2 = side == 1 ? A.online.wd : A.online.wd + (side == 0 ? na : 0);
    const int n2 = side == 1 ? na : (side == 0 ? A.lc.total : nall);
    const float *rs[3] = {r0, r1, r2};
    const int ns[3] = {n0, n1, n2};
    float acc = 0.f;
    if (tid < 256) {   // first the few lines every layer epilogue and head reads from the canonical arenas: biases, head rows
        const int grp = tid >> 6, i = tid & 63;
        const float *canon = (grp >> 1) ? A.online.canon : A.target.canon;
        const NetLayout &l = (grp & 1) ? A.lc : A.la;
        const float *net = canon + ((grp & 1) ? na : 0);
        const int tail = (l.total - l.w4 + 31) >> 5;
        int o = -1;
        if (i < 8) o = l.b1 + 32 * i;
        else if (i < 16) o = l.b2 + 32 * (i - 8);
        else if (i < 24) o = l.b3 + 32 * (i - 16);
        else if (i - 24 < tail) o = l.w4 + 32 * (i - 24);
        if (o >= 0 && o < l.total) acc += net[o];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int step = per * S8_THREADS * 32;
        int off = (mine * S8_THREADS + tid) * 32;
        for (; off + 7 * step < ns[r]; off += 8 * step) {   // 8 lines in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = rs[r][off + u * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; off < ns[r]; off += step) acc += rs[r][off];
    }
    if (acc == 1.2345678e-33f) sink[0] = acc;   // keeps the loads; never true in practice, harmless if it is
}
__device__ __forceinline__ void s8_l2_warm(const FbSlabArgs &P, int widx, float *sink) {
    // L2 warmer of this workgroup's XCD (workgroups are dealt round-robin to the XCDs): touches the weight fragments the XCD's
    // chains will stream, in the order they use them, one dword per 128-byte line
    s8_l2_warm_at(P, P.xcd_split ? (int)((blockIdx.x & 7) >> 2) : 2, P.n_pref >> 3, widx >> 3, sink);
}

__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fb_slab8(const FbSlabArgs P) {
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
    const int nslab = A.Mp / S8_ROWS;
    int chain = blockIdx.x / nslab, slab = blockIdx.x - chain * nslab;
    if (P.xcd_split && (int)blockIdx.x < 2 * nslab) {
        // workgroups are dealt round-robin to the 8 XCDs: critic-side chains on XCDs 0-3, actor-side chains on 4-7, so
        // that an XCD's L2 pulls 4 of the 6 weight-fragment sets through the fabric instead of all of them
        const int x = blockIdx.x & 7;
        chain = x >> 2;
        slab = (blockIdx.x >> 3) * 4 + (x & 3);
    }
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
    if (slab == 0 && chain < 2) tl = A.tl + chain * 32;
#endif
    if ((int)blockIdx.x >= 2 * nslab) {   // spare workgroups (see FbSlabArgs)
        const int extra = (int)blockIdx.x - 2 * nslab;
        if (extra < P.n_plan) {   // the index-plan workgroup ends here (ended waves take no part in barriers)
            if (tid >= MT_THREADS) return;
            mt_her_plan(Bk.rng, Bk.meta->current_size, Bk.T, Bk.plan_batch, 1, Bk.future_p, Bk.next_plan,
                        reinterpret_cast<uint32_t(*)[MT_N]>(&wring[0][0][0]), reinterpret_cast<int *>(pbuf));
            return;
        } else if (extra < P.n_plan + P.n_ahead) {
            s8_gather_ahead(P.ahead, P.aXT, P.aXA, P.aXP, A.ldx, A.act_off, A.act_dim, A.max_action, extra - P.n_plan, P.n_ahead);
        } else if (extra < P.n_plan + P.n_ahead + P.n_pref) {
            s8_l2_warm(P, extra - P.n_plan - P.n_ahead, dq);
        }
    } else {
    S8_TSTAMP(tl, 0);
    const SlabNetPtrs &on = A.online;
    if (chain == 0) {
        // ------------------------------------------------------------------ critic side
        const SlabNetPtrs &tn = A.target;
        const PlanRec rec = s8_plan_rec(A.gs, row0);
        float4 wbaT[6], wbcT[6], wbcA[6], whT[4], wqT[4], wqA[4];
        s8_ring_prologue<0, S8_PRO_FIRST>(ring, rbase, tn.wf + la.w2);   // first: the transfers fly while the inputs and first-layer weights come in
        s8_small_prefetch(tn.wf + la.w1, la.K1, wbaT);
#if S8_NRG == 1   // 4-row slabs: -0.4 us/update at batch 256, -0.2 at 512 k8 (8 rows: +0.2 at 1024, not taken there)
        // biases of all three trunks (epilogue lanes: the reduction-half-0 waves, column 64 (wave % 4) + lane): the first layer's
        // with its weights, the rest behind the inputs -- a trunk's start then waits for nothing from global memory
        float ebT[3] = {0.f, 0.f, 0.f}, ebC[3] = {0.f, 0.f, 0.f}, ebA[3] = {0.f, 0.f, 0.f};
        const int ecol_ = 64 * (wave & 3) + lane;
        const bool ekh0_ = S8_BOTH_HALVES || wave < 4;   // (8- / 16-row slabs: both reduction halves run epilogues)
        if (ekh0_) ebT[0] = tn.canon[la.b1 + ecol_];
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (A.gs.plan) {
            s8_gather(xin, A.gs, rec, 0, row0, A.ldx, A.act_off, ad, A.max_action, nullptr);
            s8_gather(xin2, A.gs, rec, 1, row0, A.ldx, A.act_off, ad, A.max_action, const_cast<float *>(A.XA), rows[2]);
        } else {
            s8_load(xin, S8_LDX, A.ldx, A.XT + row0 * A.ldx, A.ldx);
            s8_load(xin2, S8_LDX, A.ldx, A.XA + row0 * A.ldx, A.ldx);
            if (tid < S8_ROWS) rows[2][tid] = Bk.R[row0 + tid];
        }
        s8_ring_prologue<S8_PRO_FIRST, S8_RING>(ring, rbase, tn.wf + la.w2);
        // everything the FIRST layer does not need goes out behind the input loads (loads return in order: the inputs would wait
        // for all of it): -0.4 us/update at batch 256, -0.2 at 1024 with the critic side alone
        __builtin_amdgcn_sched_barrier(0);
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
#if S8_NRG == 1   // 4-row slabs: -0.4 us/update at batch 256, -0.2 at 512 k8 (8 rows: +0.2 at 1024, not taken there)
        if (ekh0_) {
            ebT[1] = tn.canon[la.b2 + ecol_]; ebT[2] = tn.canon[la.b3 + ecol_];
            ebC[0] = tn.canon[ca + lc.b1 + ecol_]; ebC[1] = tn.canon[ca + lc.b2 + ecol_]; ebC[2] = tn.canon[ca + lc.b3 + ecol_];
            ebA[0] = on.canon[ca + lc.b1 + ecol_]; ebA[1] = on.canon[ca + lc.b2 + ecol_]; ebA[2] = on.canon[ca + lc.b3 + ecol_];
        }
        const float *pT = ebT, *pC = ebC, *pA = ebA;
#else
        const float *pT = nullptr, *pC = nullptr, *pA = nullptr;
#endif
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
                    const float u = (A.max_action * th) / A.max_action;
                    xin[rr * S8_LDX + A.act_off + lane] = u;
                    const_cast<float *>(A.XT)[(row0 + rr) * A.ldx + A.act_off + lane] = u;
                }
            }
        }
        s8_sync();
        S8_TSTAMP(tl, 7);
        s8_trunk(xin, lc, wbcT, tn.wf + ca, tn.canon + ca, H, bufA, bufB, pbuf, nullptr, nullptr, nullptr, row0, ring, rbase,
                 on.wf + ca + lc.w2, tl, 8, nullptr, nullptr, nullptr, pC);
        {
#pragma unroll
            for (int i = 0; i < S8_RPW; ++i) {
                const int rr = wave + S8_WAVES * i;
                const float q = s8_rowdots(bufA, S8_LD, rr < S8_ROWS ? rr : 0, 1, wqT);
                if (lane == 0 && rr < S8_ROWS) {
                    rows[0][rr] = q + bqT;
                    A.QT[(row0 + rr) * 16] = q + bqT;
                }
            }
        }
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
        s8_big_layer(bufA, S8_LD, ring, rbase, on.wd + ca + lc.w3, on.wd + ca + lc.w2, SE_MASK, nullptr, 0, pbuf, bufB, S8_LD,
                     msk[1], nullptr, nullptr, 0, Bk.dA2 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 20);
        s8_big_layer(bufB, S8_LD, ring, rbase, on.wd + ca + lc.w2, nullptr, SE_MASK, nullptr, 0, pbuf, bufA, S8_LD, msk[0],
                     nullptr, nullptr, 0, Bk.dA1 + row0 * H);
        s8_sync();
        S8_TSTAMP(tl, 21);
        if (tid < S8_ROWS) {
            wt_store(Bk.dQA + (row0 + tid) * 16, keep_g);
            if (tid == 0) wt_store(Bk.part + slab, keep_a);
        }
        if (slab == 0 && tid == 0) {   // Adam step scalars for the optimizer kernel that follows
            Bk.st->step += 1;
            adam_prepare(Bk.st, Bk.adam);
        }
        S8_TSTAMP(tl, 22);
    } else {
    // ---------------------------------------------------------------------- actor side
#define S8_AFTER_CRITIC_FWD do { } while (0)
#define S8_AFTER_CRITIC_DX1 do { } while (0)
#define S8_AFTER_CRITIC_DX do { } while (0)
#include "slab8_actor_side.inc"
#undef S8_AFTER_CRITIC_FWD
#undef S8_AFTER_CRITIC_DX1
#undef S8_AFTER_CRITIC_DX
    }
    }
}

#if S8_NRG == 1
// actions = max_action * tanh(actor(normalise(obs | g)))  (ddpg_agent._preproc_inputs :163-171, models.py:19-26): the actor
// half of the chain kernel above as its own launch -- same device functions, so the same bits as the training forward
__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_policy_slab8(const PolicyArgs P) {
    __shared__ __attribute__((aligned(16))) float xin[S8_ROWS * S8_LDX];
    __shared__ __attribute__((aligned(16))) float bufA[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float bufB[S8_ROWS * S8_LD];
    __shared__ __attribute__((aligned(16))) float pbuf[S8_ROWS * 256];
    __shared__ __attribute__((aligned(16))) RingSlot wring[S8_WAVES][S8_RING];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t row0 = (size_t)blockIdx.x * S8_ROWS;
    const NetLayout &la = P.la;
    const int ad = P.act_dim, H = P.H, w = P.od + P.gd;
    RingSlot *ring = wring[wave];
    int rbase = 0;
    float4 wba[6], wh[4];
    s8_small_prefetch(P.net.wf + la.w1, la.K1, wba);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        wh[j] = *reinterpret_cast<const float4 *>(P.net.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
    const float bh = P.net.canon[la.b4 + (lane < ad ? lane : 0)];
    for (int idx = tid; idx < S8_ROWS * S8_LDX; idx += S8_THREADS) {
        const int r = idx / S8_LDX, c = idx - r * S8_LDX;
        const size_t m = row0 + r;
        float v = 0.f;
        if ((int)m < P.rows && c < w) {
            if (P.x) {
                v = P.x[m * w + c];
            } else if (c < P.od) {
                double t = fmin(fmax(P.obs[m * P.od + c], -P.clip_obs), P.clip_obs);
                t = __ddiv_rn(__dsub_rn(t, (double)P.onz->mean[c]), P.onz->std[c]);
                v = (float)fmin(fmax(t, -P.clip_o), P.clip_o);
            } else {
                const int j = c - P.od;
                double t = fmin(fmax(P.g[m * P.gd + j], -P.clip_obs), P.clip_obs);
                t = __ddiv_rn(__dsub_rn(t, (double)P.gnz->mean[j]), P.gnz->std[j]);
                v = (float)fmin(fmax(t, -P.clip_g), P.clip_g);
            }
        }
        xin[idx] = v;
    }
    s8_ring_prologue(ring, rbase, P.net.wf + la.w2);
    s8_sync();
    s8_trunk(xin, la, wba, P.net.wf, P.net.canon, H, bufA, bufB, pbuf, nullptr, nullptr, nullptr, row0, ring, rbase, nullptr,
             nullptr, 0);
#pragma unroll
    for (int i = 0; i < S8_RPW; ++i) {
        const int rr = wave + S8_WAVES * i;
        const float z = s8_rowdots(bufA, S8_LD, rr < S8_ROWS ? rr : 0, ad, wh);
        if (lane < ad && rr < S8_ROWS && (int)(row0 + rr) < P.rows)
            P.actions[(row0 + rr) * ad + lane] = P.max_action * tanhf(z + bh);
    }
}
#endif

#if S8_NRG == 1   // the split launch exists for 4-row slabs (8-row: built, measured, lost -- agent.hip at the slab-height table)
#include "slab8_split.h"
#endif

}  // namespace S8_NS
