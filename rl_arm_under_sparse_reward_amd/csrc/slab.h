// slab.h -- "row-slab" engine for the DDPG update (included by agent.hip).
//
// Measured on MI355X (profiles/, DESIGN.md "launch floor"): a dependent kernel boundary costs
// ~1.55 us for a trivial kernel and 4-6 us for any kernel that has to pull its operands through
// cold caches (the boundary write-backs/invalidates the non-coherent per-XCD L2s).  The layer-per-
// launch engine needs 20 such boundaries per update, ~130 us, for 4.6 us worth of FP32-MFMA math.
// Every product in the forward passes and in the dX half of the backward passes is ROW-independent:
// row m of layer l+1 depends only on row m of layer l.  So one workgroup can carry a slab of 16
// batch rows through an entire chain of layers with the activations parked in LDS, and no other
// workgroup ever needs its intermediate results:
//     k_fwd_slab   chain T: actor_target(x') -> critic_target(x', a')            -> Q'
//                  chain A: critic(x, a)                                          -> Q,  h1..h3 kept
//                  chain P: actor(x) -> critic(x, pi(x))                          -> Qpi, h1..h3 of both kept
//     k_bwd_slab   chain A: dL_c/dq -> dX through critic layers 4,3,2              (dY of every layer kept)
//                  chain P: dL_a/dq -> critic 4,3,2,1 -> tanh/L2 head -> actor 4,3,2 (dY of actor layers kept)
// Only the weight gradients reduce over the batch; they are one grouped GEMM launch (gemm_lds.h)
// over the kept activations / dY, followed by the fused Adam.  5 launches per update instead of 20.
//
// Per layer a workgroup streams the whole weight matrix once (256 KB for 256x256) from L2; to make
// that stream full-line and conflict free the weights are kept in a second, fragment-ordered copy:
//   forward  Wf[(nf * K/16 + S) * 256 + lane * 4 + c] = W[16 nf + (lane & 15)][16 S + 4 (lane >> 4) + c]
//   dX       Wd[(kf * N/16 + S) * 256 + lane * 4 + c] = W[16 S + 4 (lane >> 4) + c][16 kf + (lane & 15)]
// so the B operand of v_mfma_f32_16x16x4_f32 for super-step S (16 reduction indices, 4 MFMAs) is ONE
// float4 per lane, 1 KiB contiguous per wavefront.  Adam / polyak write these copies together with
// the canonical row-major parameters.  A operands (activations, dY) are read from LDS rows of 260
// floats, one ds_read_b128 per lane per super-step (conflict free, see gemm_lds.h).
#pragma once
#include "mt19937_device.h"

#ifndef SL_WAVES
#define SL_WAVES 8         // wavefronts per slab workgroup: 8 -> two output fragments per wave, 256-VGPR budget
#endif
#define SL_THREADS (64 * SL_WAVES)
#define SL_FR (16 / SL_WAVES)   // output fragments per wave in a 256-wide layer (1 or 2)
#define SL_ROWS 16
#define SL_LD 260  // LDS row stride of a 256-wide activation slab (floats)
#define SL_LDX 52  // LDS row stride of the 48-wide network-input slab

// Workgroup barrier that does NOT drain outstanding global loads.  __syncthreads() makes hipcc emit
// s_waitcnt vmcnt(0) first, which would stall on the weight prefetch of the NEXT layer at every layer
// boundary; LDS traffic only needs lgkmcnt(0) (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void slab_sync() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__host__ __device__ __forceinline__ int frag_fwd_index(int n, int k, int K) {
    return (((n >> 4) * (K >> 4) + (k >> 4)) << 8) + (((n & 15) + 16 * ((k & 15) >> 2)) << 2) + (k & 3);
}
__host__ __device__ __forceinline__ int frag_dx_index(int n, int k, int N) {
    return (((k >> 4) * (N >> 4) + (n >> 4)) << 8) + (((k & 15) + 16 * ((n & 15) >> 2)) << 2) + (n & 3);
}

struct ArenaMap {  // enough of the arena geometry to find (layer, n, k) of a flat index on the device
    NetLayout la, lc;
    int H;
    int mode;      // 0: 16-row slab fragment order (slab.h), 1: thin-slab order (slab8.h), 2: 32-row order (slab32.h)
};

// canonical arena index -> (offset of the forward-fragment copy, offset of the dX-fragment copy); -1 for biases
__host__ __device__ __forceinline__ void frag_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d) {
    const bool critic = idx >= am.la.total;
    const NetLayout &l = critic ? am.lc : am.la;
    const int base = critic ? am.la.total : 0;
    const int r = idx - base;
    int w0, N, K;
    if (r < l.b1) { w0 = l.w1; N = am.H; K = l.K1; }
    else if (r < l.w2) { off_f = off_d = -1; return; }
    else if (r < l.b2) { w0 = l.w2; N = am.H; K = am.H; }
    else if (r < l.w3) { off_f = off_d = -1; return; }
    else if (r < l.b3) { w0 = l.w3; N = am.H; K = am.H; }
    else if (r < l.w4) { off_f = off_d = -1; return; }
    else if (r < l.b4) { w0 = l.w4; N = 16; K = am.H; }
    else { off_f = off_d = -1; return; }
    const int e = r - w0, n = e / K, k = e - n * K;
    off_f = base + w0 + frag_fwd_index(n, k, K);
    off_d = base + w0 + frag_dx_index(n, k, N);
}

__host__ __device__ __forceinline__ void frag8_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d);
__host__ __device__ __forceinline__ void frag32_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d);

__host__ __device__ __forceinline__ void frag_offsets_any(const ArenaMap &am, int idx, int &off_f, int &off_d) {
    if (am.mode == 1) frag8_offsets(am, idx, off_f, off_d);
    else if (am.mode == 2) frag32_offsets(am, idx, off_f, off_d);
    else frag_offsets(am, idx, off_f, off_d);
}

__global__ void k_relayout(const float *__restrict__ canon, float *fragF, float *fragD, int n, const ArenaMap am) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int of, od;
    frag_offsets_any(am, idx, of, od);
    const float v = canon[idx];
    if (of >= 0) fragF[of] = v;
    if (od >= 0 && fragD) fragD[od] = v;
}

// ------------------------------------------------------------------------------------------------
enum { SE_BIAS_RELU = 0, SE_MASK = 1 };

__device__ __forceinline__ void slab_mma8(f32x4 &c0, f32x4 &c1, const float4 a, const float4 b0, const float4 b1) {
#ifdef SLAB_ABLATE_MFMA   // ablation build: keep every operand live, skip the matrix pipe
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w), "v"(b1.x),
                 "v"(b1.y), "v"(b1.z), "v"(b1.w));
    c0[0] += a.x;
    c1[0] += b0.x + b1.x;
    return;
#endif
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c1, 0, 0, 0);
}

template <int NS>
__device__ __forceinline__ void slab_mma(f32x4 &c0, f32x4 &c1, const float4 *__restrict__ w0,
                                         const float4 *__restrict__ w1, const float *ap) {
    float4 b0[NS], b1[NS];
#pragma unroll
    for (int S = 0; S < NS; ++S) {
        b0[S] = w0[S * 64];
        b1[S] = w1[S * 64];
    }
    // hipcc otherwise sinks every load next to its MFMA (load, wait, use): pin the load phase
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int S = 0; S < NS; ++S) {
        const float4 a = *reinterpret_cast<const float4 *>(ap + 16 * S);
        slab_mma8(c0, c1, a, b0[S], b1[S]);
    }
}

// One dense layer on a 16-row slab held in LDS:  out[16][16*nfrag] = epi(in[16][K] . Wfrag)
//   lin   LDS input rows (stride ld_in), K a multiple of 16
//   wf    fragment-ordered weights of this layer (forward or dX copy)
//   epi   SE_BIAS_RELU: out = max(acc + aux[col], 0)         (aux = bias vector)
//         SE_MASK:      out = aux[row * ldaux + col] > 0 ? acc : 0   (aux = kept activation of the layer below, global)
//   lout  LDS output rows (stride ld_out)
// Each wavefront owns output fragments nf = wave, wave + 8, ...; both of its fragments share the A read.
__device__ __forceinline__ void slab_layer(const float *lin, int ld_in, int K, const float *__restrict__ wf, int nfrag,
                                           int epi, const float *__restrict__ aux, int ldaux, float *lout,
                                           int ld_out) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int nS = K >> 4;
    for (int nf0 = wave; nf0 < nfrag; nf0 += 2 * SL_WAVES) {
        const int nf1 = nf0 + SL_WAVES;
        const bool two = nf1 < nfrag;
        const float4 *w0 = reinterpret_cast<const float4 *>(wf) + (size_t)nf0 * nS * 64 + lane;
        const float4 *w1 = reinterpret_cast<const float4 *>(wf) + (size_t)(two ? nf1 : nf0) * nS * 64 + lane;
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const float *ap = lin + i * ld_in + 4 * q;
        switch (nS) {   // small layers only (K = 16 / 32 / 48); the 256-wide ones go through big_layer
            case 3: slab_mma<3>(c0, c1, w0, w1, ap); break;
            case 2: slab_mma<2>(c0, c1, w0, w1, ap); break;
            case 1: slab_mma<1>(c0, c1, w0, w1, ap); break;
            default:
                for (int S = 0; S < nS; ++S) {
                    const float4 b0 = w0[S * 64];
                    const float4 b1 = w1[S * 64];
                    const float4 a = *reinterpret_cast<const float4 *>(ap + 16 * S);
                    slab_mma8(c0, c1, a, b0, b1);
                }
        }
        // accumulator register r holds out[row = 4q + r][col = 16 nf + i]
        const int col0 = 16 * nf0 + i, col1 = 16 * nf1 + i;
        if (epi == SE_BIAS_RELU) {
            const float bb0 = aux[col0], bb1 = two ? aux[col1] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lout[(4 * q + r) * ld_out + col0] = fmaxf(c0[r] + bb0, 0.f);
                if (two) lout[(4 * q + r) * ld_out + col1] = fmaxf(c1[r] + bb1, 0.f);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q + r;
                lout[row * ld_out + col0] = (aux[(size_t)row * ldaux + col0] > 0.f) ? c0[r] : 0.f;
                if (two) lout[row * ld_out + col1] = (aux[(size_t)row * ldaux + col1] > 0.f) ? c1[r] : 0.f;
            }
        }
    }
}

// ---- 256 -> 256 layer fed by an LDS-DMA weight ring ----------------------------------------------
// Measured (tools/ubench/stream_bw2.hip): a CU pulls only ~38 GB/s through global_load_dwordx4 -> VGPR, but
// ~141 GB/s through LDS-DMA (global_load_lds_dwordx4).  A wavefront needs 32 KiB of weights per layer
// (2 fragments x 16 super-steps x 1 KiB blocks, already lane-linear in the fragment-ordered copy, which is
// exactly the image LDS-DMA writes: wave-uniform base + lane * 16).  Each wavefront owns a private ring of
// SL_RING 1-KiB slots: block t lives in slot t % SL_RING, SL_RING blocks are always in flight, a block is
// consumed with one ds_read_b128 per lane after a COUNTED s_waitcnt vmcnt (the DMA is ordered for the
// issuing wave by vmcnt alone; the ring is wave-private, so no workgroup barrier is involved), and the freed
// slot is refilled at once.  After the last block of a layer the first SL_RING blocks of the NEXT 256x256
// layer of the chain are issued, so they fly during the epilogue, the barrier and any small stage in between.
#define SL_RING 12
#define SL_BLOCKS (16 * SL_FR)     // 1-KiB weight blocks a wavefront consumes per 256x256 layer
typedef float4 RingSlot[64];

__device__ __forceinline__ const float4 *wave_wptr(const float *wlayer, int f) {
    // the wave index is uniform: say so, and the block base stays in SGPRs (32-bit lane offset only)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    return reinterpret_cast<const float4 *>(wlayer) + (size_t)(wave + f * SL_WAVES) * 16 * 64 + lane;
}

// block t of a layer = (super-step S = t / SL_FR, fragment f = t % SL_FR)
__device__ __forceinline__ void ring_issue(RingSlot *ring, const float *wlayer, int t) {
#ifndef SLAB_ABLATE_LOAD
    const float4 *src = wave_wptr(wlayer, t % SL_FR) + (t / SL_FR) * 64;
    __builtin_amdgcn_global_load_lds(src, &ring[t % SL_RING][0], 16, 0, 0);
#endif
}

__device__ __forceinline__ void ring_prologue(RingSlot *ring, const float *wlayer) {
#pragma unroll
    for (int t = 0; t < SL_RING; ++t) ring_issue(ring, wlayer, t);
}

__device__ __forceinline__ void mma4(f32x4 &c, const float4 a, const float4 b) {
#ifdef SLAB_ABLATE_MFMA
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
    c[0] += a.x + b.x;
    return;
#endif
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
}

template <int S>
__device__ __forceinline__ void ring_step(f32x4 (&c)[SL_FR], RingSlot *ring, const float *wlayer, const float *ap) {
    constexpr int t0 = S * SL_FR;                               // first block of this super-step
    constexpr int left = SL_BLOCKS - t0;                        // blocks not yet consumed
    constexpr int inflight = left < SL_RING ? left : SL_RING;   // of those, issued and possibly still flying
    // vmcnt retires in order and the blocks needed now are the oldest outstanding operations
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight - SL_FR) : "memory");
    const int lane = threadIdx.x & 63;
    float4 b[SL_FR];
#pragma unroll
    for (int f = 0; f < SL_FR; ++f) {
#ifdef SLAB_ABLATE_LOAD
        b[f] = make_float4((float)S, 1.f, 2.f, (float)lane);
#else
        b[f] = ring[(t0 + f) % SL_RING][lane];
#endif
    }
    const float4 a = *reinterpret_cast<const float4 *>(ap + 16 * S);
    if constexpr (t0 + SL_RING < SL_BLOCKS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slot reads done before the DMA refills them
#pragma unroll
        for (int f = 0; f < SL_FR; ++f) ring_issue(ring, wlayer, t0 + SL_RING + f);
    }
#if SL_FR == 2
    slab_mma8(c[0], c[1], a, b[0], b[1]);   // interleaved accumulators (dependent latency 40 > 32-cycle issue)
#else
    mma4(c[0], a, b[0]);
#endif
    if constexpr (S + 1 < 16) ring_step<S + 1>(c, ring, wlayer, ap);
}

// out[16][256] = epi(in[16][256] . W).  The first SL_RING blocks of `wlayer` must already be in flight
// (ring_prologue or the previous big_layer's `nxt`); on return the first SL_RING blocks of `nxt` are.
__device__ __forceinline__ void big_layer(const float *lin, int ld_in, RingSlot *ring, const float *__restrict__ wlayer,
                                          const float *__restrict__ nxt, int epi, const float *__restrict__ aux,
                                          int ldaux, float *lout, int ld_out) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    float e[SL_FR][4];
#pragma unroll
    for (int f = 0; f < SL_FR; ++f) {   // epilogue operands first: their latency hides behind the products
        const int col = 16 * (wave + f * SL_WAVES) + i;
        if (epi == SE_BIAS_RELU) {
            e[f][0] = aux[col];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) e[f][r] = aux[(size_t)(4 * q + r) * ldaux + col];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c[SL_FR];
#pragma unroll
    for (int f = 0; f < SL_FR; ++f) c[f] = f32x4{0, 0, 0, 0};
    const float *ap = lin + i * ld_in + 4 * q;
    ring_step<0>(c, ring, wlayer, ap);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nxt) ring_prologue(ring, nxt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < SL_FR; ++f) {
        const int col = 16 * (wave + f * SL_WAVES) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = (epi == SE_BIAS_RELU) ? fmaxf(c[f][r] + e[f][0], 0.f) : ((e[f][r] > 0.f) ? c[f][r] : 0.f);
            lout[(4 * q + r) * ld_out + col] = v;
        }
    }
}

// 16-output head on a 16-row slab: the 8 wavefronts split the reduction, partials meet in LDS.
// Returns (to threads 0..255: row = tid >> 4, col = tid & 15) the raw sum; caller adds bias etc.
__device__ __forceinline__ float slab_head(const float *lin, int ld_in, int K, const float *__restrict__ wf,
                                           float *scratch /* >= 8*256 floats */) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int nS = K >> 4;
    const float4 *w0 = reinterpret_cast<const float4 *>(wf) + lane;
    f32x4 c0 = {0, 0, 0, 0};
    const float *ap = lin + i * ld_in + 4 * q;
    for (int S = wave; S < nS; S += SL_WAVES) {
        const float4 b0 = w0[S * 64];
        const float4 a = *reinterpret_cast<const float4 *>(ap + 16 * S);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[wave * 256 + (4 * q + r) * 16 + i] = c0[r];
    slab_sync();
    float s = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < SL_WAVES; ++w) s += scratch[w * 256 + tid];
    }
    return s;
}

// copy a 16 x width LDS slab (stride ld) to global rows (stride ldg), float4, all threads
__device__ __forceinline__ void slab_store(const float *l, int ld, int width, float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < SL_ROWS * per_row; f += SL_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        *reinterpret_cast<float4 *>(g + (size_t)r * ldg + 4 * c4) = *reinterpret_cast<const float4 *>(l + r * ld + 4 * c4);
    }
}

__device__ __forceinline__ void slab_load(float *l, int ld, int width, const float *g, int ldg) {
    const int per_row = width >> 2;
    for (int f = threadIdx.x; f < SL_ROWS * per_row; f += SL_THREADS) {
        const int r = f / per_row, c4 = f - r * per_row;
        *reinterpret_cast<float4 *>(l + r * ld + 4 * c4) = *reinterpret_cast<const float4 *>(g + (size_t)r * ldg + 4 * c4);
    }
}

struct SlabNetPtrs {
    const float *wf;     // forward-fragment copy of the whole arena this net lives in
    const float *wd;     // dX-fragment copy (online nets only)
    const float *canon;  // canonical arena (biases, head rows)
};

#ifdef SLAB_TIMELINE   // debug build: wave 0 of slab 0 stamps the 100 MHz wall clock at stage boundaries
#define SLAB_STAMP(tl, k) do { if (slab_stamp_slab() == 0 && threadIdx.x == 0) (tl)[slab_stamp_chain() * 32 + (k)] = wall_clock64(); } while (0)
__device__ __forceinline__ int slab_stamp_chain() { return gridDim.y > 1 ? blockIdx.y : blockIdx.x / (gridDim.x / 3); }
__device__ __forceinline__ int slab_stamp_slab() { return gridDim.y > 1 ? blockIdx.x : blockIdx.x % (gridDim.x / 3); }
#else
#define SLAB_STAMP(tl, k) do { } while (0)
#endif

struct GatherSrc {   // replay buffer + index plan + normalizer statistics: everything k_gather_fused takes
    const double *obs, *ag, *g, *act;
    const PlanRec *plan;              // nullptr: the network inputs are already in XA / XP / XT (minibatch API)
    const PlanRec *plan_any;          // never null (>= B records): lets the kernels load their record without a branch
    const NormDev *onz, *gnz;
    double sq_threshold, clip_obs, clip_range;
    int T, obs_dim, goal_dim, B;
    float *R;
};

struct FwdSlabArgs {
    unsigned long long *tl;
    GatherSrc gs;
    SlabNetPtrs online, target;   // arenas: [actor | critic]
    NetLayout la, lc;
    int H, ldx, act_off, act_dim, Mp;
    float max_action;
    const float *XA, *XT;
    float *XP;                    // x part read, action block written
    float *TP;                    // raw tanh of the online actor
    float *CAh1, *CAh2, *CAh3, *APh1, *APh2, *APh3, *CPh1, *CPh2, *CPh3;
    float *QT, *QA, *QP;          // [Mp][16], column 0
};

// trunk of one network on the slab: xin (K1 wide) -> h1 -> h2 -> h3, optionally keeping copies in global.
// On entry the ring holds (in flight) the first blocks of this net's layer 2; on exit those of `nxt`
// (the next 256x256 layer of the chain), or nothing when nxt == nullptr.
__device__ __forceinline__ void slab_trunk(const float *xin, const NetLayout &l, const float *wf, const float *canon,
                                           int H, float *bufA, float *bufB, float *g1, float *g2, float *g3,
                                           size_t row0, RingSlot *ring, const float *nxt, unsigned long long *tl,
                                           int tbase) {
    SLAB_STAMP(tl, tbase);
    slab_layer(xin, SL_LDX, l.K1, wf + l.w1, H >> 4, SE_BIAS_RELU, canon + l.b1, 0, bufA, SL_LD);
    slab_sync();
    SLAB_STAMP(tl, tbase + 1);
    if (g1) slab_store(bufA, SL_LD, H, g1 + row0 * H, H);
    big_layer(bufA, SL_LD, ring, wf + l.w2, wf + l.w3, SE_BIAS_RELU, canon + l.b2, 0, bufB, SL_LD);
    slab_sync();
    SLAB_STAMP(tl, tbase + 2);
    if (g2) slab_store(bufB, SL_LD, H, g2 + row0 * H, H);
    big_layer(bufB, SL_LD, ring, wf + l.w3, nxt, SE_BIAS_RELU, canon + l.b3, 0, bufA, SL_LD);
    slab_sync();
    SLAB_STAMP(tl, tbase + 3);
    if (g3) slab_store(bufA, SL_LD, H, g3 + row0 * H, H);
}

// HER gather + relabel + reward + clip + normalise for the 16 rows of this slab, straight into the LDS input
// slab (and to the global copies the backward pass / weight-gradient GEMM read later).  Same arithmetic as
// k_gather_fused (her.py:26-38, ddpg_agent.py:228-243, normalizer.py:67-70): float64 in, float32 out.
//   which = 0: x' (obs_next, g)   1: (x, a / max_action) + reward   2: x
__device__ __forceinline__ void slab_gather(float *xin, const GatherSrc &G, int which, size_t row0, int ldx, int act_off,
                                            int act_dim, float max_action, float *Xout) {
    const int tid = threadIdx.x, r = tid / (SL_THREADS / SL_ROWS), l = tid % (SL_THREADS / SL_ROWS);
    const size_t m = row0 + r;
    const bool live = (int)m < G.B;
    PlanRec rec = {0, 0, 1, 0};
    if (live) rec = G.plan[m];
    const long long e = rec.e;
    const int t = rec.t, od = G.obs_dim, gd = G.goal_dim;
    const double *obs_row = G.obs + (e * (G.T + 1) + t + (which == 0 ? 1 : 0)) * od;
    const double *g_src = rec.her ? G.ag + (e * (G.T + 1) + rec.fut) * gd : G.g + (e * G.T + t) * gd;
    for (int c = l; c < ldx; c += SL_THREADS / SL_ROWS) {
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
        xin[r * SL_LDX + c] = x;
        if (Xout && (which == 1 || c < act_off)) Xout[m * ldx + c] = x;   // chain P: the action block is written by the head
    }
    if (which == 1 && l == 0) {
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
    }
}

__global__ __launch_bounds__(SL_THREADS) __attribute__((amdgpu_waves_per_eu(SL_WAVES / 4, SL_WAVES / 4))) void k_fwd_slab(const FwdSlabArgs A) {
    __shared__ __attribute__((aligned(16))) float xin[SL_ROWS * SL_LDX];
    __shared__ __attribute__((aligned(16))) float bufA[SL_ROWS * SL_LD];
    __shared__ __attribute__((aligned(16))) float bufB[SL_ROWS * SL_LD];
    __shared__ float scratch[SL_WAVES * 256];
    __shared__ __attribute__((aligned(16))) RingSlot wring[SL_WAVES][SL_RING];
    const int slab = blockIdx.x, chain = blockIdx.y;
    const size_t row0 = (size_t)slab * SL_ROWS;
    const int tid = threadIdx.x;
    const int H = A.H;
    RingSlot *ring = wring[__builtin_amdgcn_readfirstlane(tid >> 6)];
    SLAB_STAMP(A.tl, 0);
    const NetLayout &la = A.la, &lc = A.lc;
    const int ca = la.total;  // critic segment offset inside an arena
    if (chain == 1) {
        // critic(x, a)
        // inputs before the weight prefetch: vmcnt retires in order
        if (A.gs.plan) slab_gather(xin, A.gs, 1, row0, A.ldx, A.act_off, A.act_dim, A.max_action, const_cast<float *>(A.XA));
        else slab_load(xin, SL_LDX, A.ldx, A.XA + row0 * A.ldx, A.ldx);
        ring_prologue(ring, A.online.wf + ca + lc.w2);
        slab_sync();
        slab_trunk(xin, lc, A.online.wf + ca, A.online.canon + ca, H, bufA, bufB, A.CAh1, A.CAh2, A.CAh3, row0, ring,
                   nullptr, A.tl, 1);
        const float s = slab_head(bufA, SL_LD, H, A.online.wf + ca + lc.w4, scratch);
        if (tid < 256 && (tid & 15) == 0) A.QA[(row0 + (tid >> 4)) * 16] = s + A.online.canon[ca + lc.b4];
        return;
    }
    const bool tgt = (chain == 0);
    const SlabNetPtrs &net = tgt ? A.target : A.online;
    float *X = tgt ? const_cast<float *>(A.XT) : A.XP;
    if (A.gs.plan) slab_gather(xin, A.gs, tgt ? 0 : 2, row0, A.ldx, A.act_off, A.act_dim, A.max_action, tgt ? nullptr : X);
    else slab_load(xin, SL_LDX, A.ldx, X + row0 * A.ldx, A.ldx);
    ring_prologue(ring, net.wf + la.w2);
    slab_sync();
    // actor (its last layer prefetches the critic's first 256x256 layer)
    slab_trunk(xin, la, net.wf, net.canon, H, bufA, bufB, tgt ? nullptr : A.APh1, tgt ? nullptr : A.APh2,
               tgt ? nullptr : A.APh3, row0, ring, net.wf + ca + lc.w2, A.tl, 1);   // critic layer 2 now in flight
    SLAB_STAMP(A.tl, 5);
    {
        const float s = slab_head(bufA, SL_LD, H, net.wf + la.w4, scratch);
        if (tid < 256) {
            const int r = tid >> 4, c = tid & 15;
            if (c < A.act_dim) {
                const float th = tanhf(s + net.canon[la.b4 + c]);
                const float u = (A.max_action * th) / A.max_action;   // models.py:24 then :38
                xin[r * SL_LDX + A.act_off + c] = u;
                X[(row0 + r) * A.ldx + A.act_off + c] = u;
                if (!tgt) A.TP[(row0 + r) * 16 + c] = th;
            }
        }
    }
    slab_sync();
    SLAB_STAMP(A.tl, 7);
    // critic on (x, pi(x))
    slab_trunk(xin, lc, net.wf + ca, net.canon + ca, H, bufA, bufB, tgt ? nullptr : A.CPh1, tgt ? nullptr : A.CPh2,
               tgt ? nullptr : A.CPh3, row0, ring, nullptr, A.tl, 8);
    {
        const float s = slab_head(bufA, SL_LD, H, net.wf + ca + lc.w4, scratch);
        float *Q = tgt ? A.QT : A.QP;
        if (tid < 256 && (tid & 15) == 0) Q[(row0 + (tid >> 4)) * 16] = s + net.canon[ca + lc.b4];
    }
    SLAB_STAMP(A.tl, 13);
}

struct BwdSlabArgs {
    unsigned long long *tl;
    // the index plan of the NEXT update is drawn by one spare workgroup (blockIdx.y == 2) while the backward
    // pass runs: the sequential MT19937 draw (3.6 us per minibatch) leaves the critical path entirely
    MtState *rng;
    const BufMeta *meta;
    PlanRec *next_plan;               // nullptr: nothing to draw
    double future_p;
    int T, plan_batch, nslab;
    SlabNetPtrs online;
    NetLayout la, lc;
    int H, ldx, act_off, act_dim, B, Mp;
    float max_action, gamma, clip_ret, action_l2;
    const float *QT, *QA, *QP, *R, *XP, *TP;
    const float *CAh1, *CAh2, *CAh3, *APh1, *APh2, *APh3, *CPh1, *CPh2, *CPh3;
    float *dQA;                       // [Mp][16] col 0 (for dW4 of the critic)
    float *dA3, *dA2, *dA1;           // critic-loss dY of critic layers 3,2,1
    float *dZ, *dK3, *dK2, *dK1;      // actor dY: head (16 wide), layers 3,2,1
    float *part;                      // [3][nslab] partial sums: sum (y-q)^2, sum q_pi, sum u^2
    AgentDevState *st;
    AdamCfg adam;
};

// dY of the top hidden layer from a scalar-per-row head gradient: d3[m][n] = dq[m] * w4[n] * (h3[m][n] > 0)
__device__ __forceinline__ void slab_head_bwd(const float *dq_rows /*LDS [16]*/, const float *__restrict__ w4row,
                                              const float *__restrict__ h3, int H, float *lout) {
    for (int f = threadIdx.x; f < SL_ROWS * H; f += SL_THREADS) {
        const int r = f / H, c = f - r * H;
        lout[r * SL_LD + c] = (h3[(size_t)r * H + c] > 0.f) ? dq_rows[r] * w4row[c] : 0.f;
    }
}

__global__ __launch_bounds__(SL_THREADS) __attribute__((amdgpu_waves_per_eu(SL_WAVES / 4, SL_WAVES / 4))) void k_bwd_slab(const BwdSlabArgs A) {
    __shared__ __attribute__((aligned(16))) float bufA[SL_ROWS * SL_LD];
    __shared__ __attribute__((aligned(16))) float bufB[SL_ROWS * SL_LD];
    __shared__ float scratch[SL_WAVES * 256];
    __shared__ float dq[SL_ROWS];
    __shared__ __attribute__((aligned(16))) float dz[SL_ROWS * 20];
    __shared__ __attribute__((aligned(16))) RingSlot wring[SL_WAVES][SL_RING];
    const int chain = blockIdx.x / A.nslab, slab = blockIdx.x - chain * A.nslab, nslab = A.nslab;
    const size_t row0 = (size_t)slab * SL_ROWS;
    const int tid = threadIdx.x, H = A.H;
    const NetLayout &la = A.la, &lc = A.lc;
    const int ca = la.total;
    const float invB = 1.0f / (float)A.B;
    RingSlot *ring = wring[__builtin_amdgcn_readfirstlane(tid >> 6)];
    if (chain == 2) {   // plan workgroup (only launched when next_plan != nullptr)
        if (tid >= MT_THREADS) return;   // ended waves do not take part in barriers
        mt_her_plan(A.rng, A.meta->current_size, A.T, A.plan_batch, 1, A.future_p, A.next_plan,
                    reinterpret_cast<uint32_t(*)[MT_N]>(&wring[0][0][0]), reinterpret_cast<int *>(scratch));
        return;
    }
    if (slab == 0 && chain == 0 && tid == 0) {  // bookkeeping for the optimizer step that follows
        A.st->step += 1;
        adam_prepare(A.st, A.adam);
    }
    if (chain == 0) {
        // ---- critic loss: y = clamp(r + gamma q', -1/(1-gamma), 0); L = mean((y - q)^2)   (ddpg_agent.py:255-263)
        ring_prologue(ring, A.online.wd + ca + lc.w3);
        if (tid < SL_ROWS) {
            const size_t m = row0 + tid;
            float g = 0.f, sq = 0.f;
            if ((int)m < A.B) {
                float y = A.R[m] + A.gamma * A.QT[m * 16];
                y = fminf(fmaxf(y, -A.clip_ret), 0.f);
                const float d = y - A.QA[m * 16];
                sq = d * d;
                g = -2.f * d * invB;
            }
            dq[tid] = g;
            A.dQA[m * 16] = g;
            for (int o = 8; o > 0; o >>= 1) sq += __shfl_down(sq, o, 16);
            if (tid == 0) A.part[slab] = sq;
        }
        slab_sync();
        slab_head_bwd(dq, A.online.canon + ca + lc.w4, A.CAh3 + row0 * H, H, bufA);
        slab_sync();
        slab_store(bufA, SL_LD, H, A.dA3 + row0 * H, H);
        big_layer(bufA, SL_LD, ring, A.online.wd + ca + lc.w3, A.online.wd + ca + lc.w2, SE_MASK, A.CAh2 + row0 * H, H,
                  bufB, SL_LD);
        slab_sync();
        slab_store(bufB, SL_LD, H, A.dA2 + row0 * H, H);
        big_layer(bufB, SL_LD, ring, A.online.wd + ca + lc.w2, nullptr, SE_MASK, A.CAh1 + row0 * H, H, bufA, SL_LD);
        slab_sync();
        slab_store(bufA, SL_LD, H, A.dA1 + row0 * H, H);
        return;
    }
    // ---- actor loss: L = -mean(Q(x, pi(x))) + action_l2 * mean((pi/max_action)^2)   (ddpg_agent.py:265-267)
    ring_prologue(ring, A.online.wd + ca + lc.w3);
    if (tid < SL_ROWS) {
        const size_t m = row0 + tid;
        const bool live = (int)m < A.B;
        dq[tid] = live ? -invB : 0.f;
        float sq = live ? A.QP[m * 16] : 0.f, su = 0.f;
        if (live)
            for (int j = 0; j < A.act_dim; ++j) {
                const float u = A.XP[m * A.ldx + A.act_off + j];
                su += u * u;
            }
        for (int o = 8; o > 0; o >>= 1) {
            sq += __shfl_down(sq, o, 16);
            su += __shfl_down(su, o, 16);
        }
        if (tid == 0) {
            A.part[nslab + slab] = sq;
            A.part[2 * nslab + slab] = su;
        }
    }
    slab_sync();
    slab_head_bwd(dq, A.online.canon + ca + lc.w4, A.CPh3 + row0 * H, H, bufA);
    slab_sync();
    big_layer(bufA, SL_LD, ring, A.online.wd + ca + lc.w3, A.online.wd + ca + lc.w2, SE_MASK, A.CPh2 + row0 * H, H, bufB,
              SL_LD);
    slab_sync();
    // its second half prefetches the actor's layer-3 dX, which stays in flight across the small stages below
    // its tail issues the actor's layer-3 dX blocks, which stay in flight across the small stages below
    big_layer(bufB, SL_LD, ring, A.online.wd + ca + lc.w2, A.online.wd + la.w3, SE_MASK, A.CPh1 + row0 * H, H, bufA,
              SL_LD);
    slab_sync();
    {
        // dL/d(input) of the critic, action block only: fragment kf = act_off/16 of the dX copy of W1
        const int nSred = H >> 4;
        const float s = slab_head(bufA, SL_LD, H, A.online.wd + ca + lc.w1 + (size_t)(A.act_off >> 4) * nSred * 256, scratch);
        if (tid < 256) {
            const int r = tid >> 4, c = tid & 15;
            float v = 0.f;
            const size_t m = row0 + r;
            if (c < A.act_dim && (int)m < A.B) {
                const float u = A.XP[m * A.ldx + A.act_off + c];
                const float th = A.TP[m * 16 + c];
                const float gu = A.action_l2 * (2.f * u / (float)(A.B * A.act_dim)) + s;
                const float gt = (gu / A.max_action) * A.max_action;
                v = gt * (1.f - th * th);
            }
            dz[r * 20 + c] = v;
            A.dZ[m * 16 + c] = v;
        }
    }
    slab_sync();
    // actor layer 4 backward: reduction over the 16 (padded) head outputs -> one super-step
    slab_layer(dz, 20, 16, A.online.wd + la.w4, H >> 4, SE_MASK, A.APh3 + row0 * H, H, bufB, SL_LD);
    slab_sync();
    slab_store(bufB, SL_LD, H, A.dK3 + row0 * H, H);
    big_layer(bufB, SL_LD, ring, A.online.wd + la.w3, A.online.wd + la.w2, SE_MASK, A.APh2 + row0 * H, H, bufA, SL_LD);
    slab_sync();
    slab_store(bufA, SL_LD, H, A.dK2 + row0 * H, H);
    big_layer(bufA, SL_LD, ring, A.online.wd + la.w2, nullptr, SE_MASK, A.APh1 + row0 * H, H, bufB, SL_LD);
    slab_sync();
    slab_store(bufB, SL_LD, H, A.dK1 + row0 * H, H);
}
