// gemm_lds.h -- grouped FP32-MFMA GEMM, LDS-staged operands (included by agent_engines.hip; the layer-per-launch engine of agent_layers.hip launches it through launch_group).
//
// Why a second kernel: the first version fed v_mfma_f32_16x16x4_f32 straight from global memory
// with one dword per lane, i.e. 16 rows x 16 B per wave-instruction for K-contiguous operands.
// Measured on MI355X (rocprofv3, profiles/): a 3 x [256x256x256] level took 10.5 us and even the
// 16-column head level 8.5 us -- texture-address bound, not math bound.  Here every operand chunk
// is brought in with full-line float4 loads by all 512 threads (one cold-cache round trip for the
// whole tile), parked in LDS, and the MFMA fragments are read from there:
//   * operand with a contiguous reduction index ("rowK": X, W in the forward pass, dY in dX)
//       LDS image [32][KC + 4] floats; lane (i, q) reads ONE float4 = A[i][16S + 4q .. +3] per
//       super-step S and feeds its 4 components to 4 consecutive MFMAs (the reduction index is
//       permuted identically for both operands, so the sum is the same set of products);
//       row stride 260 floats puts the 16 lanes of a ds_read_b128 group on 16 different
//       4-bank slots (conflict free).
//   * operand with a strided reduction index ("kmajor": W in dX, dY and X in dW)
//       LDS image [KC][36] floats; lane (i, q) reads lds[16S + 4q + c][i], c = 0..3: lanes i are
//       consecutive banks and the q groups are 4 rows = 144 floats = 16 banks apart.
// Reductions longer than 256 with both operands k-major (weight gradients of minibatches > 256) do not stage chunks for the
// workgroup at all: every wave brings its own 8-row blocks in through a private LDS-DMA ring and accumulates the whole tile
// on v_mfma_f32_32x32x2 (gemm_tile, ring path).
// 8 wavefronts split the super-steps of a chunk (short dependent MFMA chains), partial tiles are
// combined through LDS in a fixed order, bias / ReLU / tanh / ReLU-mask run in the epilogue, whose
// operands are prefetched before the products (every kernel starts cache-cold).
#pragma once

#ifndef ADAM_QUAD8
#define ADAM_QUAD8 1   // -DADAM_QUAD8=0: the dX copies as four dword stores per lane (A/B builds)
#endif
#define GL_THREADS 512
#define GL_WAVES 8
#define GL_KC 256
#define GL_ROWK_LD (GL_KC + 4)
#define GL_KMAJ_LD 36
#define GL_TS 36   // row length of a wave's partial 32 x 32 tile in the LDS (floats)
#ifndef GL_RING_MIN_K
#define GL_RING_MIN_K 64   // k-major x k-major reductions of at least this length take the per-wave ring path (gemm_tile)
#endif
#define GL_OPERAND_FLOATS (GL_KC * GL_KMAJ_LD)  // 9216 floats = larger of the two images (rowK: 32*260 = 8320)

#ifdef SLAB_TIMELINE   // debug build: first and last workgroup stamp the 100 MHz wall clock at stage boundaries
__device__ unsigned long long g_split_tl_gate[1024][2];   // in-launch tiles of k_fb_split8: {products done: about to wait at the gate, gate passed}
__device__ unsigned long long g_gemm_tl[32];
__device__ unsigned long long g_gemm_tl_blk[8][32][2];   // workgroup GL_BLK_WG's product loop: [wave][block]{operands landed, MFMAs issued}
#ifndef GL_BLK_WG
#define GL_BLK_WG 100
#endif
#define GL_BLK_STAMP(i, j) do { if (blockIdx.x == GL_BLK_WG && (threadIdx.x & 63) == 0 && (i) < 32) g_gemm_tl_blk[wave][(i)][(j)] = wall_clock64(); } while (0)
__device__ unsigned long long g_gemm_tl_wg[512][8];   // every tile workgroup's stamps (hp_debug_gemm_wg_timeline, time-line builds only)
#define GL_STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_gemm_tl[(blockIdx.x ? 16 : 0) + (k)] = wall_clock64(); \
                          if (threadIdx.x == 0 && blockIdx.x < 512) g_gemm_tl_wg[blockIdx.x][(k)] = wall_clock64(); \
                          if (threadIdx.x == 0 && blockIdx.x < 512 && (k) == 0) g_gemm_tl_wg[blockIdx.x][6] = __builtin_amdgcn_s_getreg(63492);   /* HW_REG_HW_ID: which CU / SE the workgroup landed on */ \
                          if (threadIdx.x == 0 && blockIdx.x == 0 && (k) < 2) g_gemm_tl[24 + (k)] = __builtin_readcyclecounter(); } while (0)   /* shader cycles over the product loop */
#else
#define GL_BLK_STAMP(i, j) do { } while (0)
#define GL_STAMP(k) do { } while (0)
#endif

// stage one 32 x kc operand chunk into LDS.  `valid` = number of real rows/cols (16 or 32).
__device__ __forceinline__ void gl_stage(float *lds, const float *base, long long s_idx, long long s_k, int valid,
                                         int kc, bool rowk) {
    const int tid = threadIdx.x;
    if (rowk) {
        const int per_row = kc >> 2;  // float4 per row
        const int total = 32 * per_row;
        for (int f = tid; f < total; f += GL_THREADS) {
            const int idx = f / per_row, k4 = f - idx * per_row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < valid) v = *reinterpret_cast<const float4 *>(base + idx * s_idx + 4 * k4);
            *reinterpret_cast<float4 *>(lds + idx * GL_ROWK_LD + 4 * k4) = v;
        }
    } else {
        // k-major operand: LDS-DMA (global_load_lds_dwordx4, ~3.6x the per-CU rate of global_load -> VGPR -> ds_write).
        // A wave instruction fills 8 consecutive 128-byte LDS rows; which element lands where is chosen per lane through
        // the SOURCE address, so the image can be swizzled for conflict-free fragment reads without padding:
        //   element (k, col) lives at row k ^ ((k >> 3) & 1), 16-byte chunk (col >> 2) ^ (((k >> 2) & 1) << 2)
        // (gl_frag: the 4 q-groups of a wavefront then hit 4 different 16-bank groups).  Columns past `valid`
        // re-read the last valid chunk: their products are computed and never stored.
        const int wave = tid >> 6, lane = tid & 63;
        const int nchunk = valid >> 2;
        for (int R = 8 * wave; R < kc; R += 8 * GL_WAVES) {
            const int rowp = R + (lane >> 3);
            const int k = rowp ^ ((rowp >> 3) & 1);
            int gch = (lane & 7) ^ (((k >> 2) & 1) << 2);
            gch = gch < nchunk ? gch : nchunk - 1;
            __builtin_amdgcn_global_load_lds(base + k * s_k + 4 * gch, lds + R * 32, 16, 0, 0);
        }
    }
}

__device__ __forceinline__ float4 gl_frag(const float *lds, bool rowk, int frag, int S, int i, int q) {
    if (rowk) return *reinterpret_cast<const float4 *>(lds + (frag * 16 + i) * GL_ROWK_LD + 16 * S + 4 * q);
    // swizzled k-major image (gl_stage): k = 16 S + 4 q + c
    const int fr = frag ^ (q & 1);
    const float *p = lds + fr * 16 + i;
    const int k0 = 16 * S + 4 * q, x = q >> 1;
    return make_float4(p[((k0 + 0) ^ x) * 32], p[((k0 + 1) ^ x) * 32], p[((k0 + 2) ^ x) * 32], p[((k0 + 3) ^ x) * 32]);
}

// LDS-DMA issued from inline assembly: the compiler tracks the builtin's LDS write and, alias scopes or not, puts
// s_waitcnt vmcnt(0) in front of the first LDS read that follows it -- which would drain the blocks in flight before the
// products of the previous one start.  The ring below orders the two itself with counted waits.  (M0 holds the LDS base of
// the transfer; hipcc reloads M0 in front of every instruction of its own that reads it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one LDS-DMA wave instruction (64 lanes x 16 B -> 1 KB at dst)
// SC1: agent-scope load -- the operands were written (write-through) by workgroups of the SAME launch on other XCDs
// (k_fb_split8's in-launch weight-gradient tiles), so they must not be served from this XCD's L2
template <bool SC1 = false>
__device__ __forceinline__ void gl_dma(float *dst, const float *src) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)dst);
    if constexpr (SC1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" ::"s"(m0v), "v"(src) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
}
#pragma clang diagnostic pop

// products of one staged chunk: wave w takes the 16-row blocks w, w + 8, ... of the chunk
__device__ __forceinline__ void gl_products(const float *ldsA, const float *ldsB, bool a_rowk, bool b_rowk, int kc, int wave,
                                            int i, int q, f32x4 &c00, f32x4 &c01, f32x4 &c10, f32x4 &c11, float &as0,
                                            float &as1) {
    const int nS = kc >> 4;
    for (int S = wave; S < nS; S += GL_WAVES) {
        const float4 a0 = gl_frag(ldsA, a_rowk, 0, S, i, q), a1 = gl_frag(ldsA, a_rowk, 1, S, i, q);
        const float4 b0 = gl_frag(ldsB, b_rowk, 0, S, i, q), b1 = gl_frag(ldsB, b_rowk, 1, S, i, q);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[c], bv0[c], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[c], bv1[c], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[c], bv0[c], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[c], bv1[c], c11, 0, 0, 0);
            as0 += av0[c];
            as1 += av1[c];
        }
    }
}

// ADAM = true (single-rank weight-gradient launch of the slab engines): the epilogue applies the optimizer step to
// the tile it just produced (gradient still written out for inspection), and workgroup 0 finishes the loss log --
// one launch and one cold pass over p/m/v less per update.  Data-parallel runs use ADAM = false + k_adam_frag so the
// gradients can be all-reduced in between.
// One 32 x 32 output tile.  bx = the tile's index in the group's launch order (blockIdx.x of the stand-alone kernels);
// lds / bsum = GL_LDS_FLOATS / GL_WAVES * 32 floats of workgroup LDS.
#define GL_LDS_FLOATS (2 * GL_OPERAND_FLOATS)
// UNI: the wave index lives in a scalar register, so the ring loop below gets scalar branches instead of exec-mask ones
// (GemmGroup::uni, set for reductions of at most 640 rows: see the note at `wave` below).
// SC1 (k_fb_split8 only): the tile runs inside the launch that produces its operands -- agent-scope operand loads, and the
// optimizer epilogue waits at AdamFuse::gate (the chains of that launch that still read the parameters it is about to step).
// PEER (data-parallel ranks, one-shot form inside this launch): the tile's gradient sums go out through the rank's exchange
// vector, the workgroup signals its slot of the per-tile flags on every peer, waits for the peers' same tile, adds the
// ranks' tiles in rank order (utils.sync_grads: SUM, utils.py:43-48) and steps -- no separate exchange + optimizer launch.
// s + (s of the lane 32 / 16 away), in every lane: gfx950's lane-swap VALU instructions instead of a trip through the LDS crossbar
// (__shfl_xor); swap(a, a) leaves {lo, lo} and {hi, hi} (or the even / odd rows), the two addends of every lane are the old pair
__device__ __forceinline__ float gl_fold32(float s) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float gl_fold16(float s) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
struct PeerTile {
    const PeerDev *D;
    int u, mean;
    int row0;   // first flag row of this launch's tiles: the split form exchanges the critic's tiles (rows 0 ..) inside the chain
                // launch and the actor's (rows behind them) in the launch that follows, within the same epoch
};
// First tile of problems 0-3 / 4-7 (16 bits each, 0xffff: no such problem) and the first bias panel, handed to the kernel as LEADING
// SCALAR arguments: those are preloaded into SGPRs when the wave starts (gfx942+ kernarg preload, Makefile), so the
// workgroup knows its problem without waiting for a first fetch of the argument block (0.4-0.7 us under load,
// tools/ubench/kernarg_preload.hip) and the problem's own fields are fetched in the FIRST round trip, not the second.
struct TileHead {
    unsigned long long t03, t47;
};
__host__ __device__ __forceinline__ int tile_head_prob(const TileHead &H, int bx) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROBS; ++i) {
        const int t0 = (int)(((i < 4 ? H.t03 >> (16 * i) : H.t47 >> (16 * (i - 4)))) & 0xffffull);
        if (bx >= t0) pi = i;
    }
    return pi;
}
template <bool ADAM, bool UNI = false, bool SC1 = false, bool PEER = false>
__device__ __forceinline__ void gemm_tile(const GemmGroup &grp, const AdamFuse *F_arg, int bx, float *lds, float (*bsum)[32],
                                          bool finalize_loss, const PeerTile *PT = nullptr, const TileHead *TH = nullptr) {
    AdamFuse F_pinned;
    const AdamFuse *F = F_arg;
    if constexpr (ADAM && SC1) {   // (agent_device.h: adam_pinned)
        F_pinned = adam_pinned(*F_arg);
        F = &F_pinned;
    }
    int pi = 0;
    if (TH) {
        pi = tile_head_prob(*TH, bx);
    } else {
#pragma unroll
        for (int i = 1; i < MAX_PROBS; ++i)
            if (i < grp.n && bx >= grp.p[i].tile0) pi = i;
    }
    const bool placed = grp.xcd == 1 && bx < 256;   // Launch::place_on_xcds: four 256 x 256 problems, one pair of XCDs each
    const bool placed2 = grp.xcd == 2 && bx < 128;  // ... two of them (actor-only launch behind k_fb_split8): four XCDs each
    if (placed) pi = (bx & 7) >> 1;
    if (placed2) pi = (bx & 7) >> 2;
    const GemmProb &p = grp.p[pi];
    int t = bx - p.tile0, slice = 0;
    if (p.ks > 1) {   // slice-major: the slices of a tile are a whole problem apart in the launch order
        const int nt = ((p.M + 31) >> 5) * p.tiles_n;
        slice = t / nt;
        t -= slice * nt;
    }
    int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    if (placed) {   // XCD x = bx & 7: problem x / 2, row-panel half x % 2, all 8 column panels
        const int slot = bx >> 3;
        tm = (bx & 1) * 4 + (slot >> 3);
        tn = slot & 7;
    }
    if (placed2) {  // XCD x = bx & 7: problem x / 4, row-panel quarter x % 4, all 8 column panels
        const int slot = bx >> 3;
        tm = (bx & 3) * 2 + (slot >> 3);
        tn = slot & 7;
    }
    const int m0 = tm * 32, n0 = tn * 32;
    // How the wave index is held decides how the two waves of a SIMD fall into step in the ring loop below, and the better form
    // depends on the length of the reduction (us/update, same box, three alternating runs each; bit-identical results):
    // scalar (readfirstlane) vs vector: 44.4 / 44.8 at batch 384, 46.8 / 47.7 at 512 k8, 53.5 / 53.0 at 768, 56.3 / 55.6 at 1024
    // (61.8 / 58.5 there without the split narrow tiles), 40.7 / 40.7 at 256.  Both are compiled as kernels of their own (k_gemm_lds*_u); the launch picks (GemmGroup::uni).
    const int tid = threadIdx.x, wave = UNI ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lane = tid & 63, i = lane & 15, q = lane >> 4;
    GL_STAMP(0);
    if (ADAM && finalize_loss && tid < 64) loss_finalize(*F);
    if (ADAM && finalize_loss && tid >= 64 && tid < 64 + SPLIT_COUNTERS * 8 && F->reset_sync)   // the launch BEHIND a split launch clears
        __hip_atomic_store(F->reset_sync + (tid - 64) * SPLIT_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the set that launch counted in
    const int vm = (p.M - m0) < 32 ? (p.M - m0) : 32, vn = (p.N - n0) < 32 ? (p.N - n0) : 32;
    const bool a_rowk = (p.a_sk == 1), b_rowk = (p.b_sk == 1);
    float *ldsA = lds, *ldsB = lds + GL_OPERAND_FLOATS;
    // epilogue operands first (cold-cache latency overlaps the operand staging)
    const int erow = tid >> 3, ecol = (tid & 7) * 4;
    const int em = m0 + erow, en = n0 + ecol;
    const bool etile = tid < 256 && erow < vm && ecol < vn;
    float ev[4] = {0.f, 0.f, 0.f, 0.f};
    if (etile) {
        if (p.epi == EPI_BIAS_RELU || p.epi == EPI_BIAS || p.epi == EPI_BIAS_TANH) {
            const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + en);
            ev[0] = b4.x; ev[1] = b4.y; ev[2] = b4.z; ev[3] = b4.w;
        } else if (p.epi == EPI_MASK) {
            const float4 b4 = *reinterpret_cast<const float4 *>(p.mask + (long long)em * p.ldmask + en);
            ev[0] = b4.x; ev[1] = b4.y; ev[2] = b4.z; ev[3] = b4.w;
        }
    }
    // ADAM: the optimizer state of this thread's 4 output elements comes in now as well -- cold loads whose latency
    // would otherwise sit between the last MFMA and the parameter store
    AdamState4 ast;
    const bool adam_vec = ADAM && etile && (en + 3 < p.n_store);
    if (ADAM) {
        if (adam_vec) adam_fetch4<SC1>(ast, *F, (int)(p.C - F->grads_base) + em * p.ldc + en);
    }
    f32x4 c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
    float as0 = 0.f, as1 = 0.f;
    const float *Abase = p.A + (long long)m0 * p.a_si;
    const float *Bbase = p.B + (long long)n0 * p.b_sj;
    bool ring_path = false;
    f32x16 cr;
    float asr = 0.f;
    if (!a_rowk && !b_rowk && p.K >= GL_RING_MIN_K) {
        // Weight gradients of a large minibatch (reduction = batch rows > 256).  Every wave owns blocks of 8 batch rows (block
        // i of wave w: rows 64 i + 8 w .. + 7), brings them in through its OWN ring of 4 blocks (A rows | B rows, 2 KB, one
        // LDS-DMA instruction per operand) and accumulates the whole 32 x 32 tile on v_mfma_f32_32x32x2: no barrier in the
        // loop, 3 blocks in flight per wave.  The version before (128-row chunks staged by the workgroup, double buffered, one
        // barrier per chunk; removed in round 3) waited 1.4-1.75 us per chunk for its transfer: 11-14 us for the
        // 1024 rows of batch 1024.
        ring_path = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) cr[r] = 0.f;
        float *ring = lds + wave * (4 * 512);
        const int h = lane >> 5, l = lane & 31;
        const int rsub = lane >> 3, chunk = lane & 7;
        const int gchA = chunk < (vm >> 2) ? chunk : (vm >> 2) - 1, gchB = chunk < (vn >> 2) ? chunk : (vn >> 2) - 1;
        // a split tile's workgroup walks its slice of the batch rows (whole turns of 64 rows)
        const int kslice = p.ks > 1 ? (((p.K + p.ks - 1) / p.ks + 63) & ~63) : p.K;
        const int k_begin = slice * kslice;
        const int k_len = (k_begin + kslice < p.K ? k_begin + kslice : p.K) - k_begin;   // may be <= 0 for a trailing slice
        const float *srcA = Abase + (long long)(k_begin + 8 * wave + rsub) * p.a_sk + 4 * gchA;
        const float *srcB = Bbase + (long long)(k_begin + 8 * wave + rsub) * p.b_sk + 4 * gchB;
        const long long stepA = 64LL * p.a_sk, stepB = 64LL * p.b_sk;
        const int nblk = k_len > 8 * wave ? (k_len - 8 * wave + 63) >> 6 : 0;   // K is a multiple of 8
        for (int i2 = 0; i2 < 3 && i2 < nblk; ++i2) {
            gl_dma<SC1>(ring + (i2 & 3) * 512, srcA + i2 * stepA);
            gl_dma<SC1>(ring + (i2 & 3) * 512 + 256, srcB + i2 * stepB);
        }
        for (int i2 = 0; i2 < nblk; ++i2) {
            const int ahead = i2 + 3;
            if (ahead < nblk) {   // into the slot of block i2 - 1, whose operands the MFMAs of the previous turn have consumed
                gl_dma<SC1>(ring + (ahead & 3) * 512, srcA + ahead * stepA);
                gl_dma<SC1>(ring + (ahead & 3) * 512 + 256, srcB + ahead * stepB);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                const int rem = nblk - 1 - i2;
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            GL_BLK_STAMP(i2, 0);
            const float *blk = ring + (i2 & 3) * 512 + h * 32 + l;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const float av = blk[kp * 64], bv = blk[256 + kp * 64];
                cr = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, cr, 0, 0, 0);
                asr += av;
            }
            GL_BLK_STAMP(i2, 1);
        }
        GL_STAMP(1);
    } else
    for (int k0 = 0; k0 < p.K; k0 += GL_KC) {
        const int kc = (p.K - k0) < GL_KC ? (p.K - k0) : GL_KC;
        if (k0 > 0) __syncthreads();  // previous chunk fully consumed
        gl_stage(ldsA, Abase + (long long)k0 * p.a_sk, p.a_si, p.a_sk, vm, kc, a_rowk);
        gl_stage(ldsB, Bbase + (long long)k0 * p.b_sk, p.b_sj, p.b_sk, vn, kc, b_rowk);
        __syncthreads();
        GL_STAMP(1);
        gl_products(ldsA, ldsB, a_rowk, b_rowk, kc, wave, i, q, c00, c01, c10, c11, as0, as1);
    }
    GL_STAMP(2);
    __syncthreads();  // operand images are dead: reuse the LDS for the partial tiles
    // partial tiles in rows of GL_TS = 36 floats: a thread's 4 columns of a row are one aligned float4, so the reduction over the 8 waves
    // is 8 LDS reads per thread instead of 32 (the count of reads in flight per wave is capped at 15: 32 went in three round trips)
    float *my = lds + wave * (32 * GL_TS);
    const bool want_bias_grad = p.bias_grad != nullptr && tn == 0 && grp.bias0 == 0;   // (bias0 > 0: gemm_bias_tile does it)
    if (ring_path) {   // 32 x 32 accumulator layout: register r -> row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32
        const int h = lane >> 5, l = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) my[(8 * (r >> 2) + 4 * h + (r & 3)) * GL_TS + l] = cr[r];
        if (want_bias_grad) {
            asr = gl_fold32(asr);
            if (h == 0) bsum[wave][l] = asr;
        }
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        my[row * GL_TS + i] = c00[r];
        my[row * GL_TS + 16 + i] = c01[r];
        my[(16 + row) * GL_TS + i] = c10[r];
        my[(16 + row) * GL_TS + 16 + i] = c11[r];
    }
    }
    if (want_bias_grad && !ring_path) {
        as0 = gl_fold16(as0);
        as0 = gl_fold32(as0);
        as1 = gl_fold16(as1);
        as1 = gl_fold32(as1);
        if (q == 0) {
            bsum[wave][i] = as0;
            bsum[wave][16 + i] = as1;
        }
    }
    __syncthreads();
    GL_STAMP(3);
    float sb = 0.f;
    if (want_bias_grad && tid < vm) {
#pragma unroll
        for (int w = 0; w < GL_WAVES; ++w) sb += bsum[w][tid];
    }
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (etile) {
        float4 pw[GL_WAVES];
#pragma unroll
        for (int w = 0; w < GL_WAVES; ++w) pw[w] = *reinterpret_cast<const float4 *>(lds + w * (32 * GL_TS) + erow * GL_TS + ecol);
#pragma unroll
        for (int w = 0; w < GL_WAVES; ++w) {   // wave order, as before: s = ((0 + w0) + w1) + ...
            v[0] += pw[w].x; v[1] += pw[w].y; v[2] += pw[w].z; v[3] += pw[w].w;
        }
    }
    if (ring_path && p.ks > 1) {
        // Split reduction (the narrow problems of a large minibatch: without it 40 CUs carry two full tiles and the launch is as
        // long as those): every slice writes its partial tile write-through, takes a ticket, and the LAST one to arrive sums the
        // ks partials in slice order and runs the epilogue -- nobody waits, the order of the sum is fixed (dw64.h has the same
        // hand-off: sc1 stores, drained, one relaxed agent-scope atomic; the reader uses agent-scope loads).
        float *mine = grp.part + ((size_t)(p.part0 + t) * p.ks + slice) * GL_PART;
        if (tid < 256) wt_store4(mine + erow * 32 + ecol, make_float4(v[0], v[1], v[2], v[3]));
        if (tid < 32) wt_store(mine + 1024 + tid, sb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have completed
        __syncthreads();                                   // (also: every read of bsum above is done)
        int *flag = reinterpret_cast<int *>(&bsum[0][0]);
        if (tid == 0) {
            const unsigned long long old = __hip_atomic_fetch_add(grp.ticket + p.part0 + t, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = ((old + 1ull) % (unsigned long long)p.ks == 0ull) ? 1 : 0;
        }
        __syncthreads();
        if (!*flag) return;   // somebody else finishes this tile
        const float *q = grp.part + (size_t)(p.part0 + t) * p.ks * GL_PART;
        // every slice's partial is asked for BEFORE the first one is used: one round trip under load instead of ks dependent ones
        // (a loop over a run-time ks waited for each slice in turn: 4 us of the finisher's 5.5 at batch 1024); summed in slice order
        static_assert(GL_MAX_KS == 8, "the unrolled fetch below covers 8 slices");
        unsigned long long plo[GL_MAX_KS], phi[GL_MAX_KS];
        float pb[GL_MAX_KS];
#pragma unroll
        for (int sl = 0; sl < GL_MAX_KS; ++sl) {
            plo[sl] = phi[sl] = 0ull;
            pb[sl] = 0.f;
            if (sl < p.ks) {
                const float *src = q + (size_t)sl * GL_PART;
                if (tid < 256) {
                    const unsigned long long *s2 = reinterpret_cast<const unsigned long long *>(src + erow * 32 + ecol);
                    plo[sl] = __hip_atomic_load(s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    phi[sl] = __hip_atomic_load(s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid < 32) pb[sl] = __hip_atomic_load(src + 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int sl = 0; sl < GL_MAX_KS; ++sl) {
            if (sl < p.ks) {
                const float w[4] = {__uint_as_float((unsigned)plo[sl]), __uint_as_float((unsigned)(plo[sl] >> 32)),
                                    __uint_as_float((unsigned)phi[sl]), __uint_as_float((unsigned)(phi[sl] >> 32))};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = sl ? v[j] + w[j] : w[j];
                sb = sl ? sb + pb[sl] : pb[sl];
            }
        }
    }
    if constexpr (ADAM && PEER) {
        const PeerDev &D = *PT->D;
        const unsigned long long epoch = D.epoch[0] + (unsigned long long)PT->u + 1ull;
        const int par = (int)(epoch & 1ull);
        float *G = D.grad[D.rank][par];                                     // arena layout, like F->grads_base
        const int gbase = (int)(p.C - F->grads_base) + em * p.ldc + en;    // this thread's 4 elements
        const int bbase = want_bias_grad ? (int)(p.bias_grad - F->grads_base) + m0 + tid : 0;
        const bool vec = etile && (en + 3 < p.n_store);
        // 1. my sums -> my exchange vector (system scope: what the peers' loads must find), drained before the flag
        if (vec) peer_store4(G + gbase, make_float4(v[0], v[1], v[2], v[3]));
        else if (etile) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (en + j < p.n_store) peer_store1(G + gbase + j, v[j]);
        }
        if (want_bias_grad && tid < vm) peer_store1(G + bbase, sb);
        if (D.world > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // 2. my slot of this tile's flag row on every peer; wait for theirs (bounded; a dead exchange steps nothing)
            // The flag row names the TILE, not the workgroup: with a split reduction (p.ks > 1) the workgroup that gets here is
            // whichever slice arrived last -- another slice on another rank, so bx would differ from rank to rank.  tile0 + t
            // lies inside the problem's own range of the launch order (slice 0's workgroup index) and is the same everywhere.
            const int row = PT->row0 + (p.ks > 1 ? p.tile0 + t : bx);
            peer_signal_row(D, D.flags_t, row, epoch);
            if (!peer_wait(D, D.flags_t[D.rank] + (size_t)row * HP_PEER_MAX, epoch, 1u)) return;
            // 3. rank-ordered sum (the same float32 expression on every rank: the replicas stay bit-identical)
            const size_t gbytes = (size_t)F->am.la.total * 4 + (size_t)F->am.lc.total * 4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb = 0.f;
            for (int q = 0; q < D.world; ++q) {
                float w[4] = {v[0], v[1], v[2], v[3]}, wb = sb;
                if (q != D.rank) {
                    if (vec) {
                        const float4 t4 = peer_load4(D.grad[q][par], gbytes, (unsigned)gbase * 4u, false);
                        w[0] = t4.x; w[1] = t4.y; w[2] = t4.z; w[3] = t4.w;
                    } else if (etile) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (en + j < p.n_store) w[j] = __hip_atomic_load(D.grad[q][par] + gbase + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (want_bias_grad && tid < vm) wb = __hip_atomic_load(D.grad[q][par] + bbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = q ? acc[j] + w[j] : w[j];
                accb = q ? accb + wb : wb;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[j];
            sb = accb;
        }
        if (PT->mean) {   // SUM / world, float32 true division (what k_peer_adam does)
            const float wn = (float)D.world;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] /= wn;
            sb /= wn;
        }
    }
    if constexpr (ADAM && SC1) {
        // the chains of this launch that still read the parameters stepped below (AdamFuse::gate) must be past them
#ifdef SLAB_TIMELINE
        if (threadIdx.x == 0 && blockIdx.x < 1024 && F->tl_mark) g_split_tl_gate[blockIdx.x][0] = wall_clock64();
#endif
        if (!adam_gate_wait(*F, pi, reinterpret_cast<int *>(&bsum[0][0]))) return;
#ifdef SLAB_TIMELINE
        if (threadIdx.x == 0 && blockIdx.x < 1024 && F->tl_mark) g_split_tl_gate[blockIdx.x][1] = wall_clock64();
#endif
    }
    if (want_bias_grad && tid < vm) {
        if (!SC1 || !ADAM || F->keep_grads) p.bias_grad[m0 + tid] = sb;
        if (ADAM) adam_apply<SC1>(*F, (int)(p.bias_grad - F->grads_base) + m0 + tid, sb);
    }
    if (!etile) return;
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
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (en + j < p.n_store) {
                    const float th = tanhf(v[j] + ev[j]);
                    p.C2[(long long)em * p.ldc2 + en + j] = th;
                    p.C[(long long)em * p.ldc + en + j] = (p.max_action * th) / p.max_action;
                }
            }
            return;
        }
        default: break;
    }
    GL_STAMP(4);
    if (ADAM && !F->keep_grads) {
        // gradient consumed by the optimizer epilogue below, not materialised
    } else if (en + 3 < p.n_store) {
        *reinterpret_cast<float4 *>(p.C + (long long)em * p.ldc + en) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (en + j < p.n_store) p.C[(long long)em * p.ldc + en + j] = v[j];
    }
    if (ADAM) {
        const int base = (int)(p.C - F->grads_base) + em * p.ldc + en;
        GL_STAMP(7);
        if (adam_vec) {
            // the fragment copies' offsets from the tile's own coordinates (row em, column en of layer frag_layer's weight matrix): the
            // generic look-up walks the arena layout and divides by the row length, per thread, at the launch's very end
            int of = ADAM_FRAG_LOOKUP, od = -1;
            if (F->am.mode == 1 && p.frag_layer > 0) {
                const int tw0 = (int)(p.C - F->grads_base);
                of = p.frag_layer <= 3 ? tw0 + frag8_fwd_index(em, en, p.ldc) : -1;
                od = p.frag_layer >= 2 ? tw0 + frag8_dx_index(em, en, p.M) : -1;
            }
            // whole 4-row groups with a dX copy (layers 2-4): the four lanes of a group hand their quads round (adam_apply4, quad8)
            const bool quad8 = ADAM_QUAD8 && od >= 0 && (vm & 3) == 0;
            adam_apply4(*F, base, v, ast, of, od, quad8);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (en + j < p.n_store) adam_apply<SC1>(*F, base + j, v[j]);
        }
    }
    GL_STAMP(5);
}

// Bias gradient (column sums of dY over the batch rows) + its optimizer step for ONE 32-row panel of one problem, as a workgroup
// of its own (GemmGroup::bias0).  In gemm_tile the tile with tn == 0 carries them, and those tiles are every launch's tail: at
// batch 512 they end 4 us after the median tile (11.8 vs 7.6 us in the workgroup), at 256 2.6 us after (7.6 vs 5.0) -- the
// scalar optimizer step with its three cold state loads sits behind the reduction, in front of the tile's own epilogue.  Here:
// the same A-operand stream through the same per-wave ring and the SAME summation order as gemm_tile's `asr` (block by block,
// kp = 0..3, halves, waves 0..7; a split problem slice by slice, the slices' sums added in slice order) -- the bits do not
// change -- with nothing else to do, on a CU slot the tiles leave free, and the state loads issued at entry.
template <bool ADAM, bool SC1 = false>
__device__ __forceinline__ void gemm_bias_tile(const GemmGroup &grp, const AdamFuse *F_arg, int bidx, float *lds, float (*bsum)[32]) {
    AdamFuse F_pinned;
    const AdamFuse *F = F_arg;
    if constexpr (ADAM && SC1) {
        F_pinned = adam_pinned(*F_arg);
        F = &F_pinned;
    }
    int pi = 0, first = 0, acc = 0;
#pragma unroll
    for (int i = 0; i < MAX_PROBS; ++i) {
        const int nb = (i < grp.n && grp.p[i].bias_grad) ? (grp.p[i].M + 31) >> 5 : 0;
        if (bidx >= acc && bidx < acc + nb) { pi = i; first = acc; }
        acc += nb;
    }
    const GemmProb &p = grp.p[pi];
    const int tm = bidx - first, m0 = tm * 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int vm = (p.M - m0) < 32 ? (p.M - m0) : 32;
    const int gidx = ADAM ? (int)(p.bias_grad - F->grads_base) + m0 + tid : 0;
    AdamState1 st{0.f, 0.f, 0.f};
    if (ADAM && tid < vm) { st.p = F->p[gidx]; st.m = F->m[gidx]; st.v = F->v[gidx]; }
    const float *Abase = p.A + (long long)m0 * p.a_si;
    float *ring = lds + wave * (4 * 512);
    const int h = lane >> 5, l = lane & 31, rsub = lane >> 3, chunk = lane & 7;
    const int gchA = chunk < (vm >> 2) ? chunk : (vm >> 2) - 1;
    const long long stepA = 64LL * p.a_sk;
    const int ks = p.ks > 1 ? p.ks : 1;
    const int kslice = p.ks > 1 ? (((p.K + p.ks - 1) / p.ks + 63) & ~63) : p.K;
    float sb = 0.f;
    for (int slice = 0; slice < ks; ++slice) {
        const int k_begin = slice * kslice;
        const int k_len = (k_begin + kslice < p.K ? k_begin + kslice : p.K) - k_begin;
        const float *srcA = Abase + (long long)(k_begin + 8 * wave + rsub) * p.a_sk + 4 * gchA;
        const int nblk = k_len > 8 * wave ? (k_len - 8 * wave + 63) >> 6 : 0;
        float asr = 0.f;
        for (int i2 = 0; i2 < 3 && i2 < nblk; ++i2) gl_dma<SC1>(ring + (i2 & 3) * 512, srcA + i2 * stepA);
        for (int i2 = 0; i2 < nblk; ++i2) {
            const int ahead = i2 + 3;
            if (ahead < nblk) {
                gl_dma<SC1>(ring + (ahead & 3) * 512, srcA + ahead * stepA);
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                const int rem = nblk - 1 - i2;
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const float *blk = ring + (i2 & 3) * 512 + h * 32 + l;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) asr += blk[kp * 64];
        }
        asr = gl_fold32(asr);
        if (slice) __syncthreads();   // the sums of the slice before have been read
        if (h == 0) bsum[wave][l] = asr;
        __syncthreads();
        if (tid < vm) {
            float s1 = 0.f;
#pragma unroll
            for (int w = 0; w < GL_WAVES; ++w) s1 += bsum[w][tid];
            sb = slice ? sb + s1 : s1;
        }
    }
    if constexpr (ADAM && SC1) {
        if (!adam_gate_wait(*F, pi, reinterpret_cast<int *>(&bsum[0][0]))) return;
    }
    if (tid < vm) {
        if (!SC1 || !ADAM || F->keep_grads) p.bias_grad[m0 + tid] = sb;
        if (ADAM) adam_apply<SC1>(*F, gidx, sb, &st);
    }
}

// A launch's kernel-argument block is fetched field by field through the scalar cache, each field where the compiler first
// uses it; the first workgroups of a launch to reach a field take the round trip to memory for it, in every stage (time line of
// the actor's tile launch: the first five workgroups per XCD ended 0.5-2 us after the others).  Every workgroup therefore
// touches the whole block with ONE vector load per 128-byte line at entry, so that the lines are on their way into the XCD's L2
// before the scalar loads ask for them: -0.2 us/update at batch 256 (profiles/r05_ab_actor_tile_launch_tail.txt).  The same trick on
// the kernel's CODE (s_getpc, 36 KB) changed nothing and is not kept.  The destination register must stay reserved until the load
// has returned: the kernel keeps it alive to its end (kernarg_prefetch_keep).
#ifndef GL_KERNARG_PREFETCH
#define GL_KERNARG_PREFETCH 1
#endif
template <int BYTES>
__device__ __forceinline__ unsigned kernarg_prefetch() {
    unsigned sink = 0;
#if GL_KERNARG_PREFETCH
    static_assert(BYTES <= GL_THREADS * 128, "one load per thread covers the block");
    const char *ka = (const char *)(unsigned long long)(size_t)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)threadIdx.x * 128;
    if ((int)threadIdx.x * 128 < BYTES) asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(ka) : "memory");
#endif
    return sink;
}
__device__ __forceinline__ void kernarg_prefetch_keep(unsigned sink) {
#if GL_KERNARG_PREFETCH
    asm volatile("" ::"v"(sink));
#endif
}

// The loss log of the update (and, behind a split launch, the reset of the counter set that launch counted in) used to be
// workgroup 0's first job: its wave 0 went through two dependent cold round trips before it issued its first operand block, so
// that tile was the launch's last to end (7.6 us against a median of 5.4).  GemmGroup::loss_wg: the launch has one more
// workgroup, its last, that does nothing else.
__device__ __forceinline__ void gemm_loss_wg(const AdamFuse &F) {
    const int tid = threadIdx.x;
    if (tid < 64) loss_finalize(F);
    else if (tid < 64 + SPLIT_COUNTERS * 8 && F.reset_sync)
        __hip_atomic_store(F.reset_sync + (tid - 64) * SPLIT_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool ADAM, bool UNI = false>
__device__ __forceinline__ void gemm_lds_body(const GemmGroup &grp, const AdamFuse *F, const TileHead *TH = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[GL_LDS_FLOATS];  // A image | B image; reused for the reduction
    __shared__ float bsum[GL_WAVES][32];
    if (ADAM && grp.loss_wg && blockIdx.x == gridDim.x - 1) {
        gemm_loss_wg(*F);
        return;
    }
    const int bx = (int)blockIdx.x;
    if (grp.bias0 > 0 && bx >= grp.bias0) {
        gemm_bias_tile<ADAM>(grp, F, bx - grp.bias0, lds, bsum);
        return;
    }
    gemm_tile<ADAM, UNI>(grp, F, bx, lds, bsum, bx == 0 && !grp.loss_wg, nullptr, TH);
}

__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds(const GemmGroup grp) { gemm_lds_body<false>(grp, nullptr); }

__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam(unsigned long long t03, unsigned long long t47, const GemmGroup grp,
                                                              const AdamFuse F) {
    const TileHead TH{t03, t47};
    const unsigned sink = kernarg_prefetch<16 + sizeof(GemmGroup) + sizeof(AdamFuse)>();
    gemm_lds_body<true>(grp, &F, &TH);
    kernarg_prefetch_keep(sink);
}
// the same kernels with the wave index in a scalar register (gemm_tile's UNI; kernels of their own so that each form keeps
// its own register allocation: both forms inside one kernel cost either of them 0.3-0.6 us/update)
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_u(const GemmGroup grp) { gemm_lds_body<false, true>(grp, nullptr); }
__global__ __launch_bounds__(GL_THREADS) void k_gemm_lds_adam_u(unsigned long long t03, unsigned long long t47, const GemmGroup grp,
                                                                const AdamFuse F) {
    const TileHead TH{t03, t47};
    const unsigned sink = kernarg_prefetch<16 + sizeof(GemmGroup) + sizeof(AdamFuse)>();
    gemm_lds_body<true, true>(grp, &F, &TH);
    kernarg_prefetch_keep(sink);
}
