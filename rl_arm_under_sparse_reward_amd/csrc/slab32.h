// slab32.h -- row-slab engine for LARGE minibatches on v_mfma_f32_32x32x2_f32: 32 batch rows per workgroup, forward and
// backward (dX) of a chain in one kernel (the structure of slab8.h's k_fb_slab8).  Included by agent.hip.
//
// Why a third slab shape.  From 2048 rows per GPU on, the update stops being a latency chain and becomes matrix work:
// 11.3 GFLOP at batch 4096 = 72 us of FP32 MFMA.  The 16-row engine on the 16x16x4 MFMA (slab.h, removed in round 3) reads three LDS operands
// of 1 KiB for every 64 matrix-pipe cycles per wave, which with 8 waves is more than the LDS delivers, and runs 768 + 512
// workgroups of one per CU whose epilogues idle the pipes (44 % / 40 %, profiles/r02_kernel_trace_b4096_k4.txt).  On the
// 32x32x2 instruction a wavefront owns a 32 x 32 output tile: each of the 8 waves of a workgroup computes 32 columns of a
// 256-wide layer for all 32 rows of the slab, reads 2 KiB of LDS per 256 matrix cycles (4x less), needs no cross-wave
// reduction, and the whole 256 KiB weight matrix is streamed once per 32 rows.  A 256x256 layer is then 6.8 us of
// back-to-back MFMAs per workgroup against 1.9 us of weight stream: matrix bound.  Batch 4096 = 128 slabs x 2 chains = 256
// workgroups, exactly one round.
//
//   lane l of a wave: row i = l & 31, reduction half h = l >> 5.  Block b (8 reduction indices) = 4 MFMAs; MFMA c uses
//   k = 8b + 4h + c, so the A operand of a block is ONE float4 per lane out of a row-major LDS slab
//   (x[i][8b + 4h .. + 3], stride 260 floats: conflict free) and the B operand ONE float4 per lane of a fragment-ordered copy
//     forward  Wf32[((n >> 5) * K/8 + (k >> 3)) * 256 + ((((k >> 2) & 1) << 5) + (n & 31)) * 4 + (k & 3)] = W[n][k]
//     dX       Wd32[((k >> 5) * N/8 + (n >> 3)) * 256 + ((((n >> 2) & 1) << 5) + (k & 31)) * 4 + (n & 3)] = W[n][k]
//   streamed global -> LDS ring (8 x 1 KiB per wave, continuous across layers) by LDS-DMA like slab8.h.
//   accumulator register r of lane l: row 8 (r >> 2) + 4 h + (r & 3), column 32 wave + (l & 31).
// The minibatch is gathered by k_gather_fused into the input sets (beside the previous update on a second stream), the
// chains load their rows from there.  Heads, losses and the action gradient are the slab8 formulation (one wavefront per
// row, shuffle trees) on 32 rows.  Summation order differs from the other engines (per output: blocks in order, inside a
// block c = 0..3 with the two reduction halves of each MFMA), results are deterministic; tested against the oracle like
// every engine (tests/test_gpu_update.py).
#pragma once

#define S32_THREADS 512
#define S32_WAVES 8
#define S32_LD 260
#define S32_LDX 52
#define S32_RING 8     // x 1 KiB per wave; must divide the 32 blocks of a layer (the ring base is 0 at every layer start)
#define S32_RPW (S32_ROWS / S32_WAVES)   // rows per wavefront in the one-wavefront-per-row stages

typedef unsigned int s32_mask_t;   // ReLU mask of one column: bit r = row r of the slab

__host__ __device__ __forceinline__ int frag32_fwd_index(int n, int k, int K) {
    return (((n >> 5) * (K >> 3) + (k >> 3)) << 8) + (((((k >> 2) & 1) << 5) + (n & 31)) << 2) + (k & 3);
}
__host__ __device__ __forceinline__ int frag32_dx_index(int n, int k, int N) {
    return (((k >> 5) * (N >> 3) + (n >> 3)) << 8) + (((((n >> 2) & 1) << 5) + (k & 31)) << 2) + (n & 3);
}

// canonical arena index -> offsets of the slab32 fragment copies (-1: this tensor has none)
__host__ __device__ __forceinline__ void frag32_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d) {
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
    if (layer <= 3) off_f = base + w0 + frag32_fwd_index(n, k, K);
    if (layer >= 2) off_d = base + w0 + frag32_dx_index(n, k, N);
}

#ifdef SLAB_TIMELINE   // debug build: slab 0 of each chain stamps the 100 MHz wall clock (tl[chain * 32 + k])
#define S32_STAMP(k) do { if (slab == 0 && tid == 0) A.tl[chain * 32 + (k)] = wall_clock64(); } while (0)
// per-wave stamps inside one layer: wtl[4 * wave + k]
#define S32_WSTAMP(wtl, k) do { if ((wtl) && (threadIdx.x & 63) == 0) (wtl)[4 * (threadIdx.x >> 6) + (k)] = wall_clock64(); } while (0)
#else
#define S32_STAMP(k) do { } while (0)
#define S32_WSTAMP(wtl, k) do { } while (0)
#endif

namespace s32 {

__device__ __forceinline__ void sync() {   // barrier that leaves global loads / DMA in flight (__syncthreads() would also drain them: s_waitcnt vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// weight block b (8 reduction indices) of this wave's 32 output columns in a layer with nb blocks per column group
__device__ __forceinline__ const float4 *wblock(const float *wlayer, int wave, int nb, int b) {
    return reinterpret_cast<const float4 *>(wlayer) + ((size_t)wave * nb + b) * 64 + (threadIdx.x & 63);
}

__device__ __forceinline__ void ring_issue32(RingSlot *ring, const float *wlayer, int wave, int t) {
    __builtin_amdgcn_global_load_lds(wblock(wlayer, wave, 32, t), &ring[t % S32_RING][0], 16, 0, 0);
}

__device__ __forceinline__ void ring_prologue32(RingSlot *ring, const float *wlayer) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int t = 0; t < S32_RING; ++t) ring_issue32(ring, wlayer, wave, t);
}

// One accumulator per wave: two alternating accumulators (a dependent MFMA always one instruction behind) measured the same
// (98.0 vs 96.7 us per launch at batch 4096), the two wavefronts of a SIMD already cover the result latency.
__device__ __forceinline__ void mma432(f32x16 &c, const float4 a, const float4 b) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
}
__device__ __forceinline__ void acc_zero(f32x16 &c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
}

// Block T of the 32 this wave consumes in a 256-reduction layer (32 blocks = 4 ring turns, so the ring base is 0 at every
// layer start).  Operands of block T + 1 are read from LDS before the MFMAs of block T (their latency hides under them).
// YS = global stores this wave issued AFTER the first S32_RING blocks of this layer went out and before its first step
// (the previous layer's epilogue copies): vmcnt retires in order, so while the block waited for is one of those first
// blocks the stores are younger than it and may stay in flight -- without the allowance every layer would start by
// draining its predecessor's stores to memory (~1-2 us).
template <int T, bool HAS_NEXT, int YS>
__device__ __forceinline__ void ring_step32(f32x16 &c, RingSlot *ring, const float *wlayer, const float *nxt, int wave,
                                          const float *arow, const float4 acur, const float4 bcur) {
    float4 anext = acur, bnext = bcur;
    if constexpr (T + 1 < 32) {
        // DMA blocks possibly outstanding here: T+1 .. T+R-1 (fewer at the tail of a chain's last layer)
        constexpr int out = HAS_NEXT ? S32_RING - 1 : ((31 - T) < S32_RING - 1 ? (31 - T) : S32_RING - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(out - 1 + (T + 1 < S32_RING ? YS : 0)) : "memory");
        bnext = ring[(T + 1) % S32_RING][threadIdx.x & 63];
        anext = *reinterpret_cast<const float4 *>(arow + 8 * (T + 1));
    }
    if constexpr (T + S32_RING < 32 || HAS_NEXT) {
        // slot of block T is free once ITS read has returned (bcur was read one step ago; the reads just issued are younger)
        if constexpr (T + 1 < 32) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (T + S32_RING < 32) ring_issue32(ring, wlayer, wave, T + S32_RING);
        else ring_issue32(ring, nxt, wave, T + S32_RING - 32);
    }
    mma432(c, acur, bcur);
    if constexpr (T + 1 < 32) ring_step32<T + 1, HAS_NEXT, YS>(c, ring, wlayer, nxt, wave, arow, anext, bnext);
}

// epilogue of a layer: this wave's 32 x 32 tile -> LDS slab (row major), optional global copy, ReLU masks
//   SE_BIAS_RELU: o = max(v + bias[col], 0), mask_out[col] bit r = (o > 0)        SE_MASK: o = mask_in[col] bit r ? v : 0
__device__ __forceinline__ void finish(const f32x16 &acc, int epi, const float *__restrict__ bias, float *lout, int ld_out,
                                       const s32_mask_t *mask_in, s32_mask_t *mask_out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = 32 * wave + (lane & 31), h = lane >> 5;
    const float b = (epi == SE_BIAS_RELU) ? bias[col] : 0.f;
    const unsigned bits = mask_in ? mask_in[col] : 0u;
    unsigned outbits = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * h + (r & 3);
        const float v = acc[r];
        float o;
        if (epi == SE_BIAS_RELU) {
            o = fmaxf(v + b, 0.f);
            outbits |= (o > 0.f ? 1u : 0u) << row;
        } else {
            o = ((bits >> row) & 1u) ? v : 0.f;
        }
        lout[row * ld_out + col] = o;
    }
    if (mask_out) {
        outbits |= __shfl_xor(outbits, 32);            // the two row halves of a column live in lanes l and l + 32
        if (h == 0) mask_out[col] = outbits;
    }
}

__device__ __forceinline__ void slab_store32(const float *l, int ld, int width, float *g, int ldg);

// out[32][256] = epi(in[32][256] . W), DMA-ring fed.  The first S32_RING blocks of `wlayer` must be in flight; on return
// the first S32_RING blocks of `nxt` are (if nxt != nullptr).
template <int YS = 0>
__device__ __forceinline__ void big_layer32(const float *lin, RingSlot *ring, const float *__restrict__ wlayer,
                                          const float *__restrict__ nxt, int epi, const float *__restrict__ bias, float *lout,
                                          const s32_mask_t *mask_in, s32_mask_t *mask_out, float *gprev,
                                          unsigned long long *wtl = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    S32_WSTAMP(wtl, 0);
    // global copy of the INPUT slab (operand of the weight-gradient launch): 4 float4 stores per lane that complete under
    // this layer's products.  The epilogue used to store its 16 accumulator registers one dword per lane each; in the
    // per-wave timeline of one layer (-DSLAB_TIMELINE) that epilogue took 2.3 us for the waves that finish their products
    // first (global stores are issue-bound) and the layer 9.2 us; with the copy here 1.0 us and 8.5 us.  The whole update did
    // not move (132.6 vs 132.4 us at batch 4096: the next layer's start absorbs it), kept for the 4x fewer store instructions.
    if (gprev) slab_store32(lin, S32_LD, 256, gprev, 256);
    f32x16 c;
    acc_zero(c);
    const float *arow = lin + (lane & 31) * S32_LD + 4 * (lane >> 5);
    const float4 afirst = *reinterpret_cast<const float4 *>(arow);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S32_RING - 1 + YS) : "memory");   // block 0 has landed
    const float4 bfirst = ring[0][lane];
    S32_WSTAMP(wtl, 1);
    if (nxt) ring_step32<0, true, YS>(c, ring, wlayer, nxt, wave, arow, afirst, bfirst);
    else ring_step32<0, false, YS>(c, ring, wlayer, nxt, wave, arow, afirst, bfirst);
    __builtin_amdgcn_sched_barrier(0);
    S32_WSTAMP(wtl, 2);
    finish(c, epi, bias, lout, S32_LD, mask_in, mask_out);
    S32_WSTAMP(wtl, 3);
}

// small layer (reduction length Kred = 16 / 32 / 48 -> 2 / 4 / 6 blocks): weights straight from global into registers
// (issued by the caller at kernel entry through small_prefetch), A operand from an LDS slab with row stride ld_in
__device__ __forceinline__ void small_prefetch(const float *__restrict__ wlayer, int Kred, float4 (&b)[6]) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = Kred >> 3;
#pragma unroll
    for (int t = 0; t < 6; ++t) b[t] = *wblock(wlayer, wave, nb, t < nb ? t : nb - 1);   // branch free
}
__device__ __forceinline__ void small_layer(const float *lin, int ld_in, int Kred, const float4 (&b)[6], int epi,
                                            const float *__restrict__ bias, float *lout, const s32_mask_t *mask_in,
                                            s32_mask_t *mask_out) {
    const int lane = threadIdx.x & 63;
    const int nb = Kred >> 3;
    f32x16 c;
    acc_zero(c);
    const float *arow = lin + (lane & 31) * ld_in + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 6; ++t)
        if (t < nb) mma432(c, *reinterpret_cast<const float4 *>(arow + 8 * t), b[t]);
    finish(c, epi, bias, lout, S32_LD, mask_in, mask_out);
}

// Head layers (4 or 1 outputs of reduction length 256) for the S32_RPW = 4 rows a wavefront owns: lane p takes reduction
// indices 4p..4p+3 of every row, then ALL the sums are reduced together.  multi_sum halves the list of sums each step (mask
// 32, 16, ...: the lane whose bit is clear keeps the lower half and receives its partner's partials of it), then finishes
// with plain butterflies: 17 cross-lane moves for 16 sums and 7 for 4, where one shuffle tree per sum is 7 dependent
// ds_bpermute round trips each -- measured 5 us per 4-output head at batch 4096 with the trees, ~16 us of a chain.
// On return every lane holds sum number sum_index<N>(lane), complete (bit-identical in the lanes that share a sum).
template <int N>
__device__ __forceinline__ int sum_index(int lane) {
    if constexpr (N == 16) return (((lane >> 5) & 1) << 3) | (((lane >> 4) & 1) << 2) | (((lane >> 3) & 1) << 1) | ((lane >> 2) & 1);
    else return (((lane >> 5) & 1) << 1) | ((lane >> 4) & 1);
}
template <int N, int MASK>
__device__ __forceinline__ void multi_sum_halve(float (&p)[16], int lane) {
    if constexpr (N > 1) {
        const bool hi = (lane & MASK) != 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const float send = hi ? p[k] : p[k + N / 2];
            const float keep = hi ? p[k + N / 2] : p[k];
            p[k] = keep + __shfl_xor(send, MASK);
        }
        multi_sum_halve<N / 2, MASK / 2>(p, lane);
    }
}
template <int N>
__device__ __forceinline__ float multi_sum(float (&p)[16]) {
    static_assert(N == 16 || N == 4, "4 rows x (4 or 1) outputs");
    multi_sum_halve<N, 32>(p, threadIdx.x & 63);
    float s = p[0];
#pragma unroll
    for (int m = (N == 16 ? 2 : 8); m > 0; m >>= 1) s += __shfl_xor(s, m);
    return s;
}
__device__ __forceinline__ float dot4(const float4 h, const float4 w) { return (h.x * w.x + h.y * w.y) + (h.z * w.z + h.w * w.w); }
// 4 outputs per row: sum (row slot i, output j) has number 4 i + j, row = wave + S32_WAVES * i
__device__ __forceinline__ float head4(const float *lin, int ld_in, const float4 (&wv)[4]) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = threadIdx.x & 63;
    float part[16];
#pragma unroll
    for (int i = 0; i < S32_RPW; ++i) {
        const float4 hv = *reinterpret_cast<const float4 *>(lin + (wave + S32_WAVES * i) * ld_in + 4 * p);
#pragma unroll
        for (int j = 0; j < 4; ++j) part[4 * i + j] = dot4(hv, wv[j]);
    }
    return multi_sum<16>(part);
}
// 1 output per row: sum number i is row wave + S32_WAVES * i
__device__ __forceinline__ float head1(const float *lin, int ld_in, const float4 w) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), p = threadIdx.x & 63;
    float part[16];
#pragma unroll
    for (int i = 0; i < S32_RPW; ++i)
        part[i] = dot4(*reinterpret_cast<const float4 *>(lin + (wave + S32_WAVES * i) * ld_in + 4 * p), w);
    return multi_sum<4>(part);
}

__device__ __forceinline__ void slab_store32(const float *l, int ld, int width, float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < S32_ROWS * per_row; f += S32_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        *reinterpret_cast<float4 *>(g + (size_t)r * ldg + 4 * c4) = *reinterpret_cast<const float4 *>(l + r * ld + 4 * c4);
    }
}
__device__ __forceinline__ void slab_load32(float *l, int ld, int width, const float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < S32_ROWS * per_row; f += S32_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        *reinterpret_cast<float4 *>(l + r * ld + 4 * c4) = *reinterpret_cast<const float4 *>(g + (size_t)r * ldg + 4 * c4);
    }
}

// xin (K1 wide) -> h1 -> h2 -> h3 (left in bufA).  Ring: layer 2 in flight on entry, `nxt` on exit.
template <bool KEEP>   // KEEP: g1..g3 are non-null (global copies for the weight gradients): each slab is copied out of LDS by
                       // the layer that consumes it (big_layer32 gprev), h3 at the end: 4 float4 stores per lane and slab
__device__ __forceinline__ void trunk(const float *xin, const NetLayout &l, const float4 (&wb1)[6], const float *wf,
                                      const float *canon, float *bufA, float *bufB, float *g1, float *g2, float *g3,
                                      size_t row0, RingSlot *ring, const float *nxt, s32_mask_t *m1, s32_mask_t *m2,
                                      s32_mask_t *m3) {
    small_layer(xin, S32_LDX, l.K1, wb1, SE_BIAS_RELU, canon + l.b1, bufA, nullptr, m1);
    sync();
    big_layer32<KEEP ? 4 : 0>(bufA, ring, wf + l.w2, wf + l.w3, SE_BIAS_RELU, canon + l.b2, bufB, nullptr, m2,
                              g1 ? g1 + row0 * 256 : nullptr);
    sync();
    big_layer32<KEEP ? 4 : 0>(bufB, ring, wf + l.w3, nxt, SE_BIAS_RELU, canon + l.b3, bufA, nullptr, m3,
                              g2 ? g2 + row0 * 256 : nullptr);
    sync();
    if (g3) slab_store32(bufA, S32_LD, 256, g3 + row0 * 256, 256);
}

__device__ __forceinline__ void head_bwd_inplace(const float *dq_rows, float w4c, float *buf) {
    const int c = threadIdx.x & 255, r0 = threadIdx.x >> 8;   // a thread owns column c of rows r0, r0 + 2, ...
#pragma unroll
    for (int i = 0; i < S32_ROWS / 2; ++i) {
        const int r = r0 + 2 * i;
        const float h = buf[r * S32_LD + c];
        buf[r * S32_LD + c] = (h > 0.f) ? dq_rows[r] * w4c : 0.f;
    }
}

// chain 0 (critic side):  actor_target -> critic_target -> Q';  critic(x, a) -> Q;  critic loss;  critic dX chain
// chain 1 (actor side):   actor -> critic(x, pi(x)) -> Q_pi;  actor loss;  dX through the critic and the actor
// spare workgroup (n_plan): index plan of a later update.  Inputs come gathered (f.XT / f.XA / f.XP, b.R).
__global__ __launch_bounds__(S32_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fb_slab32(const FbSlabArgs P) {
    const FwdSlabArgs &A = P.f;
    const BwdSlabArgs &Bk = P.b;
    __shared__ __attribute__((aligned(16))) float xin[S32_ROWS * S32_LDX];
    // one 6.5 KB region, two tenants: the critic side's second input slab | the actor side's w1 columns + head gradient
    // (the kernel's LDS must leave ~10 KB of the CU's 160 KB for a co-resident k_draw_plan on the second stream)
    __shared__ __attribute__((aligned(16))) float xin2[S32_ROWS * S32_LDX];
    float *const w1t = xin2;                 // [4][256]
    float *const dz = xin2 + 4 * 256;        // [S32_ROWS][20]
    static_assert(4 * 256 + S32_ROWS * 20 <= S32_ROWS * S32_LDX, "actor-side scratch must fit the second input slab");
    __shared__ __attribute__((aligned(16))) float bufA[S32_ROWS * S32_LD];
    __shared__ __attribute__((aligned(16))) float bufB[S32_ROWS * S32_LD];
    __shared__ float dq[S32_ROWS];
    __shared__ float rows[3][S32_ROWS];          // per-row scalars: Q' | Q (or Q_pi) | reward
    __shared__ s32_mask_t msk[5][256];           // ReLU masks: critic h1, h2 | actor h1, h2, h3
    __shared__ __attribute__((aligned(16))) RingSlot wring[S32_WAVES][S32_RING];
    const int nslab = A.Mp / S32_ROWS;
    const int tid = threadIdx.x, H = A.H;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if ((int)blockIdx.x >= 2 * nslab) {   // index-plan workgroup (the only spare role of this engine)
        if (tid >= MT_THREADS) return;
        mt_her_plan(Bk.rng, Bk.meta->current_size, Bk.T, Bk.plan_batch, 1, Bk.future_p, Bk.next_plan,
                    reinterpret_cast<uint32_t(*)[MT_N]>(&wring[0][0][0]), reinterpret_cast<int *>(bufA));
        return;
    }
    const int chain = blockIdx.x / nslab, slab = blockIdx.x - chain * nslab;
    // (critic-side chains on XCDs 0-3 and actor-side chains on 4-7, as the thin-slab engine does at small batches, measured
    // the same here: 133.3 vs 132.7 us/update at batch 4096 -- the weight stream is not what an XCD's L2 misses on)
    // the head sums this lane ends up holding (multi_sum): row hr / output hj of a 4-output head, row qr of a 1-output head
    const int hj = sum_index<16>(lane) & 3, hr = wave + S32_WAVES * (sum_index<16>(lane) >> 2);
    const int qr = wave + S32_WAVES * sum_index<4>(lane);
    const size_t row0 = (size_t)slab * S32_ROWS;
    const NetLayout &la = A.la, &lc = A.lc;
    const int ca = la.total, ad = A.act_dim;
    const float invB = 1.0f / (float)Bk.B;
    RingSlot *ring = wring[wave];
    const SlabNetPtrs &on = A.online;
    if (chain == 0) {
        // ------------------------------------------------------------------ critic side
        const SlabNetPtrs &tn = A.target;
        float4 wbaT[6], wbcT[6], wbcA[6], whT[4], wqT[4], wqA[4];
        small_prefetch(tn.wf + la.w1, la.K1, wbaT);
        small_prefetch(tn.wf + ca + lc.w1, lc.K1, wbcT);
        small_prefetch(on.wf + ca + lc.w1, lc.K1, wbcA);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            whT[j] = *reinterpret_cast<const float4 *>(tn.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
        wqT[0] = *reinterpret_cast<const float4 *>(tn.canon + ca + lc.w4 + 4 * lane);
        wqA[0] = *reinterpret_cast<const float4 *>(on.canon + ca + lc.w4 + 4 * lane);
        const float bhT = tn.canon[la.b4 + (hj < ad ? hj : 0)];
        const float bqT = tn.canon[ca + lc.b4], bqA = on.canon[ca + lc.b4];
        const float w4c = on.canon[ca + lc.w4 + (tid & 255)];
        __builtin_amdgcn_sched_barrier(0);
        slab_load32(xin, S32_LDX, A.ldx, A.XT + row0 * A.ldx, A.ldx);
        slab_load32(xin2, S32_LDX, A.ldx, A.XA + row0 * A.ldx, A.ldx);
        if (tid < S32_ROWS) rows[2][tid] = Bk.R[row0 + tid];
        ring_prologue32(ring, tn.wf + la.w2);
        S32_STAMP(0);
        sync();
        S32_STAMP(1);
        trunk<false>(xin, la, wbaT, tn.wf, tn.canon, bufA, bufB, nullptr, nullptr, nullptr, row0, ring, tn.wf + ca + lc.w2,
                     nullptr, nullptr, nullptr);
        {   // target actor head -> action block of the target critic's input (models.py:24)
            const float z = head4(bufA, S32_LD, whT);
            if ((lane & 3) == 0 && hj < ad) {
                const float th = tanhf(z + bhT);
                const float u = (A.max_action * th) / A.max_action;
                xin[hr * S32_LDX + A.act_off + hj] = u;
                const_cast<float *>(A.XT)[(row0 + hr) * A.ldx + A.act_off + hj] = u;
            }
        }
        sync();
        S32_STAMP(2);
        trunk<false>(xin, lc, wbcT, tn.wf + ca, tn.canon + ca, bufA, bufB, nullptr, nullptr, nullptr, row0, ring,
                     on.wf + ca + lc.w2, nullptr, nullptr, nullptr);
        {
            const float q = head1(bufA, S32_LD, wqT[0]);
            if ((lane & 15) == 0) {
                rows[0][qr] = q + bqT;
                A.QT[(row0 + qr) * 16] = q + bqT;
            }
        }
        sync();   // the q' head has read bufA before the next trunk's first layer overwrites it
        // critic(x, a): forward with global copies (weight gradients) and masks (dX chain below)
        S32_STAMP(3);
        trunk<true>(xin2, lc, wbcA, on.wf + ca, on.canon + ca, bufA, bufB, A.CAh1, A.CAh2, A.CAh3, row0, ring,
                    on.wd + ca + lc.w3, msk[0], msk[1], nullptr);
        {
            const float q = head1(bufA, S32_LD, wqA[0]);
            if ((lane & 15) == 0) {
                rows[1][qr] = q + bqA;
                A.QA[(row0 + qr) * 16] = q + bqA;
            }
        }
        sync();
        // ---- critic loss (ddpg_agent.py:255-263)
        S32_STAMP(4);
        if (tid < S32_ROWS) {
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
            Bk.dQA[m * 16] = g;
            for (int o = S32_ROWS / 2; o > 0; o >>= 1) sq += __shfl_down(sq, o, S32_ROWS);
            if (tid == 0) Bk.part[slab] = sq;
        }
        sync();
        head_bwd_inplace(dq, w4c, bufA);   // bufA holds h3 of critic(x, a)
        sync();
        slab_store32(bufA, S32_LD, H, Bk.dA3 + row0 * H, H);
        S32_STAMP(5);
        // (4 = the float4 stores per wave of the slab copy just above)
        big_layer32<4>(bufA, ring, on.wd + ca + lc.w3, on.wd + ca + lc.w2, SE_MASK, nullptr, bufB, msk[1], nullptr, nullptr,
                       slab == 0 ? A.tl + 96 : nullptr);
        sync();
        S32_STAMP(6);
        S32_WSTAMP(slab == 0 ? A.tl + 128 : nullptr, 0);
        big_layer32<4>(bufB, ring, on.wd + ca + lc.w2, nullptr, SE_MASK, nullptr, bufA, msk[0], nullptr, Bk.dA2 + row0 * H);
        sync();
        slab_store32(bufA, S32_LD, H, Bk.dA1 + row0 * H, H);
        S32_STAMP(7);
        if (slab == 0 && tid == 0) {   // Adam step scalars for the optimizer kernel that follows
            Bk.st->step += 1;
            adam_prepare(Bk.st, Bk.adam);
        }
        return;
    }
    // ---------------------------------------------------------------------- actor side
    float4 wba[6], wbc[6], wh[4], wq[4], wb4[6];
    float w1n[4];
    small_prefetch(on.wf + la.w1, la.K1, wba);
    small_prefetch(on.wf + ca + lc.w1, lc.K1, wbc);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        wh[j] = *reinterpret_cast<const float4 *>(on.canon + la.w4 + (j < ad ? j : ad - 1) * H + 4 * lane);
    wq[0] = *reinterpret_cast<const float4 *>(on.canon + ca + lc.w4 + 4 * lane);
    const float bh = on.canon[la.b4 + (hj < ad ? hj : 0)];
    const float bq = on.canon[ca + lc.b4];
    const float w4c = on.canon[ca + lc.w4 + (tid & 255)];
    {
        const float *w1 = on.canon + ca + lc.w1 + (size_t)(tid & 255) * lc.K1 + A.act_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) w1n[j] = w1[j < ad ? j : ad - 1];
    }
    small_prefetch(on.wd + la.w4, 16, wb4);
    __builtin_amdgcn_sched_barrier(0);
    slab_load32(xin, S32_LDX, A.ldx, A.XP + row0 * A.ldx, A.ldx);
    ring_prologue32(ring, on.wf + la.w2);
    S32_STAMP(0);
    sync();
    S32_STAMP(1);
    trunk<true>(xin, la, wba, on.wf, on.canon, bufA, bufB, A.APh1, A.APh2, A.APh3, row0, ring, on.wf + ca + lc.w2, msk[2],
                msk[3], msk[4]);
    float u_mine = 0.f, th_mine = 0.f;   // of (row hr, action hj): the head sum this lane ends up holding
    {   // actor head: tanh -> action block of the critic input (models.py:24, :38)
        const float z = head4(bufA, S32_LD, wh);
        if ((lane & 3) == 0 && hj < ad) {
            th_mine = tanhf(z + bh);
            u_mine = (A.max_action * th_mine) / A.max_action;
            xin[hr * S32_LDX + A.act_off + hj] = u_mine;
            A.XP[(row0 + hr) * A.ldx + A.act_off + hj] = u_mine;
            A.TP[(row0 + hr) * 16 + hj] = th_mine;
        }
    }
    sync();
    S32_STAMP(2);
    trunk<false>(xin, lc, wbc, on.wf + ca, on.canon + ca, bufA, bufB, nullptr, nullptr, nullptr, row0, ring,
                 on.wd + ca + lc.w3, msk[0], msk[1], nullptr);
    {
        const float q = head1(bufA, S32_LD, wq[0]);
        if ((lane & 15) == 0) {
            rows[1][qr] = q + bq;
            A.QP[(row0 + qr) * 16] = q + bq;
        }
    }
    if (tid < 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w1t[j * 256 + tid] = w1n[j];
    }
    sync();
    // ---- actor loss (ddpg_agent.py:265-267)
    S32_STAMP(3);
    if (tid < S32_ROWS) {
        const size_t m = row0 + tid;
        const bool live = (int)m < Bk.B;
        dq[tid] = live ? -invB : 0.f;
        float sq = live ? rows[1][tid] : 0.f, su = 0.f;
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ad) {
                    const float u = xin[tid * S32_LDX + A.act_off + j];
                    su += u * u;
                }
        }
        for (int o = S32_ROWS / 2; o > 0; o >>= 1) {
            sq += __shfl_down(sq, o, S32_ROWS);
            su += __shfl_down(su, o, S32_ROWS);
        }
        if (tid == 0) {
            Bk.part[nslab + slab] = sq;
            Bk.part[2 * nslab + slab] = su;
        }
    }
    sync();
    head_bwd_inplace(dq, w4c, bufA);   // bufA holds h3 of critic(x, pi(x))
    sync();
    S32_STAMP(4);
    big_layer32(bufA, ring, on.wd + ca + lc.w3, on.wd + ca + lc.w2, SE_MASK, nullptr, bufB, msk[1], nullptr, nullptr);
    sync();
    S32_STAMP(5);
    big_layer32(bufB, ring, on.wd + ca + lc.w2, on.wd + la.w3, SE_MASK, nullptr, bufA, msk[0], nullptr, nullptr);
    sync();
    S32_STAMP(6);
    {   // d L / d(action block of the critic input), then through the L2 penalty and tanh; lane j owns action j
        float4 w1g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w1g[j] = *reinterpret_cast<const float4 *>(w1t + j * 256 + 4 * lane);
        const float sj = head4(bufA, S32_LD, w1g);
        const size_t m = row0 + hr;
        if ((lane & 3) == 0 && hj < ad) {
            float v = 0.f;
            if ((int)m < Bk.B) {
                const float gu = Bk.action_l2 * (2.f * u_mine / (float)(Bk.B * ad)) + sj;
                const float gt = (gu / A.max_action) * A.max_action;
                v = gt * (1.f - th_mine * th_mine);
            }
            dz[hr * 20 + hj] = v;
            Bk.dZ[m * 16 + hj] = v;
        }
        if (lane >= ad && lane < 16) {   // padding columns of the 16-wide head gradient (reduction length of the layer below)
#pragma unroll
            for (int i = 0; i < S32_RPW; ++i) {
                dz[(wave + S32_WAVES * i) * 20 + lane] = 0.f;
                Bk.dZ[(row0 + wave + S32_WAVES * i) * 16 + lane] = 0.f;
            }
        }
    }
    sync();
    S32_STAMP(7);
    small_layer(dz, 20, 16, wb4, SE_MASK, nullptr, bufB, msk[4], nullptr);
    sync();
    S32_STAMP(8);
    big_layer32<4>(bufB, ring, on.wd + la.w3, on.wd + la.w2, SE_MASK, nullptr, bufA, msk[3], nullptr, Bk.dK3 + row0 * H);
    sync();
    S32_STAMP(9);
    big_layer32<4>(bufA, ring, on.wd + la.w2, nullptr, SE_MASK, nullptr, bufB, msk[2], nullptr, Bk.dK2 + row0 * H);
    sync();
    slab_store32(bufB, S32_LD, H, Bk.dK1 + row0 * H, H);
    S32_STAMP(10);
}

}  // namespace s32
