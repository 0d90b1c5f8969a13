// dw64.h -- weight gradients of a LARGE minibatch (+ the optimizer epilogue), included by agent_engines.hip.
//
// dW = dY^T X over B batch rows (ddpg_agent.py:262-271 through autograd: fc*.weight.grad, fc*.bias.grad).  gemm_lds.h
// gives every 32 x 32 output tile to one workgroup that walks ALL the batch rows: 8 flop per operand byte, and beyond
// ~1024 rows the launch is bound by what the L2s deliver (~6.5 TB/s measured; 58 us at batch 4096 = 41 TFLOP/s).  Here:
//   * 64 x 64 output tiles on v_mfma_f32_32x32x2_f32: 16 flop per operand byte, half the L2 -> LDS traffic;
//   * the batch rows of a tile are split over S workgroups (80 tiles x S = 3: 240 workgroups, one per CU and one round -- the
//     product loop is bound by what a CU can stream, so more workgroups per CU buy nothing and every extra slice adds
//     a partial tile to exchange: us/update at batch 4096 for S = 2 / 3 / 4 / 5 / 6 / 8: 136.4 / 128.1 / 141.1 / 134.6 / 132.1 / 145.9);
//     every workgroup writes its partial tile write-through, takes a ticket, and the LAST one to arrive sums the S
//     partials in slice order and runs the epilogue (optimizer step or gradient store).  Nobody waits for anybody, the
//     sum order is fixed: deterministic.  Hand-off as in slab8.h (cdna_hip_programming.md Guideline 16 form R1):
//     sc1 stores, drained, one relaxed device-scope atomic; the reader uses device-scope (sc1) loads.
//   * operands come in by LDS-DMA into per-wave rings: wave w owns blocks of 4 batch rows (rows 32 i + 4 w .. + 3 of the
//     slice), 3 blocks in flight, and accumulates the WHOLE 64 x 64 tile (4 accumulators): no barrier in the product
//     loop, one LDS read per MFMA.  (First version: 64-row stages shared by the workgroup, one barrier per stage, one
//     stage ahead -- every stage waited ~2 us for its transfer, 31.8 us for the 16 stages of a slice at batch 4096.)
//     64 KB of LDS: two workgroups per CU.  Block image [4 rows][64 columns] per operand: the 32 lanes of a half
//     wavefront read 32 consecutive floats of one batch row (an MFMA operand: row = reduction index, lane = output).
//   * the 8 partial tiles of a workgroup meet in LDS in a fixed tree.
#pragma once

#define DW_THREADS 512
#define DW_KH 32                       // batch rows per turn of the 8 waves (4 each): slices are multiples of this
#define DW_BLK 512                     // floats of one ring block: 4 rows x 64 columns of A, then of B
#define DW_RING 4                      // blocks per wave (power of two): 3 in flight
#define DW_LDS_FLOATS (8 * DW_RING * DW_BLK + 4 * 64)   // rings (64 KB; later the partial tiles) + column sums
#define DW_RED_LD 68                   // floats per row of a quarter in LDS (16-byte aligned rows, 3 x 64 x 68 x 4 B = 52 KB)

#ifdef SLAB_TIMELINE   // debug build: first and last tile workgroup stamp the 100 MHz wall clock (g_gemm_tl of gemm_lds.h)
#define DW_STAMP(k) do { if (threadIdx.x == 0 && (bx == 0 || bx == X.n_wg - 1)) g_gemm_tl[(bx ? 16 : 0) + (k)] = wall_clock64(); \
                          if (threadIdx.x == 0 && ((k) == 1 || (k) == 3 || (k) == 5)) atomicMax(&g_gemm_tl[8 + (k)], (wall_clock64() << 12) | (unsigned long long)bx); } while (0)
#else
#define DW_STAMP(k) do { } while (0)
#endif

struct Dw64Args {
    int S;                  // workgroups (batch-row slices) per tile
    int kslice;             // batch rows per slice (multiple of DW_KH)
    int n_wg;               // tile workgroups of the launch = S * tiles
    int tile0[MAX_PROBS];   // first 64 x 64 tile of each problem
    int tiles_n[MAX_PROBS]; // tiles along the columns
    float *part;            // exchange buffer: DW_PART floats per (tile, slice)
    unsigned long long *ticket;   // one arrival counter per tile, monotonic over the life of the agent (64-bit: a 32-bit
                            // count would wrap after 2^32 / S launches -- a day of training -- and S = 6 does not divide 2^32)
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one LDS-DMA wave instruction: 64 lanes x 16 B -> 1 KB at `dst` (4 rows of 64 floats).  Inline assembly for the reason
// given at gl_dma (gemm_lds.h): the compiler would drain the transfer in front of the next LDS read.
__device__ __forceinline__ void dw_dma(float *dst, const float *src) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
}
#pragma clang diagnostic pop

// One (tile, slice) workgroup.  lds = DW_LDS_FLOATS floats; flag = one int of LDS.
template <bool ADAM>
__device__ __forceinline__ void dw64_tile(const GemmGroup &grp, const AdamFuse *F, const Dw64Args &X, int bx, float *lds, int *flag) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int pi, s, t;
    {   // problem-major, slice-major within a problem.  (Placing all tiles of a (problem, slice) group on one XCD so that every
        // operand byte crosses the fabric once measured the same: 132.06 vs 132.16 us/update at batch 4096, split 6.)
        pi = 0;
#pragma unroll
        for (int i = 1; i < MAX_PROBS; ++i)
            if (i < grp.n && bx >= X.S * X.tile0[i]) pi = i;
        const int local = bx - X.S * X.tile0[pi];
        const int nt = (pi + 1 < grp.n ? X.tile0[pi + 1] : X.n_wg / X.S) - X.tile0[pi];
        s = local / nt;
        t = local - s * nt;
    }
    const GemmProb &p = grp.p[pi];
    const int tiles_n = X.tiles_n[pi];
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int vm = (p.M - m0) < 64 ? (p.M - m0) : 64, vn = (p.N - n0) < 64 ? (p.N - n0) : 64;
    const int tile = X.tile0[pi] + t;
    DW_STAMP(0);
    if (ADAM && bx == 0 && tid < 64) loss_finalize(*F);
    const int k_begin = s * X.kslice;
    const int k_end = (k_begin + X.kslice) < p.K ? (k_begin + X.kslice) : p.K;
    const float *Abase = p.A + (long long)m0 * p.a_si;
    const float *Bbase = p.B + (long long)n0 * p.b_sj;
    // ---- products.  Every wave owns blocks of 4 batch rows (block i of wave w: rows 32 i + 4 w .. + 3 of the slice), brings
    // them in through its OWN ring of DW_RING blocks (A rows | B rows, 2 KB) and accumulates the whole 64 x 64 tile: no
    // barrier in the loop, DW_RING - 1 blocks in flight per wave, one LDS read per MFMA.
    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;
    float as0 = 0.f, as1 = 0.f;
    const int h = lane >> 5, l = lane & 31;
    const bool wm = vm > 32, wn = vn > 32;   // narrow tiles (head rows, first-layer columns) skip the empty accumulators
    {
        float *ring = lds + wave * (DW_RING * DW_BLK);
        const int nblk = k_begin < k_end ? (k_end - k_begin) >> 5 : 0;
        const int rsub = lane >> 4, chunk = lane & 15;
        const int gchA = chunk < (vm >> 2) ? chunk : (vm >> 2) - 1, gchB = chunk < (vn >> 2) ? chunk : (vn >> 2) - 1;
        const float *srcA = Abase + (long long)(k_begin + 4 * wave + rsub) * p.a_sk + 4 * gchA;
        const float *srcB = Bbase + (long long)(k_begin + 4 * wave + rsub) * p.b_sk + 4 * gchB;
        const long long stepA = 32LL * p.a_sk, stepB = 32LL * p.b_sk;
        for (int i = 0; i < DW_RING - 1 && i < nblk; ++i) {
            dw_dma(ring + (i & (DW_RING - 1)) * DW_BLK, srcA + i * stepA);
            dw_dma(ring + (i & (DW_RING - 1)) * DW_BLK + 256, srcB + i * stepB);
        }
        for (int i = 0; i < nblk; ++i) {
            const int ahead = i + DW_RING - 1;
            if (ahead < nblk) {   // into the slot of block i - 1, whose operands the MFMAs of the previous turn have consumed
                dw_dma(ring + (ahead & (DW_RING - 1)) * DW_BLK, srcA + ahead * stepA);
                dw_dma(ring + (ahead & (DW_RING - 1)) * DW_BLK + 256, srcB + ahead * stepB);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DW_RING - 1)) : "memory");
            } else {
                const int rem = nblk - 1 - i;   // blocks behind block i still in flight
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const float *blk = ring + (i & (DW_RING - 1)) * DW_BLK + h * 64 + l;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const float a0 = blk[kp * 128], a1 = blk[kp * 128 + 32], b0 = blk[256 + kp * 128], b1 = blk[256 + kp * 128 + 32];
                c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
                if (wn) c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
                if (wm) {
                    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
                    if (wn) c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
                }
                as0 += a0;
                as1 += a1;
            }
        }
    }
    DW_STAMP(1);
    as0 = gl_fold32(as0);
    as1 = gl_fold32(as1);
    // ---- the 8 partial tiles meet in LDS, fixed tree: ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
    float *bs = lds + DW_RING * DW_BLK * 8;   // [4][64] column sums, behind the rings
    // (slot layout of the intermediate turns = the accumulators' own: float4 number (16 a + q) * 64 + lane holds registers
    // 4 q .. 4 q + 3 of accumulator a -- wave w and wave w + half hold the same elements in the same lanes, so one
    // ds_write_b128 / ds_read_b128 per register quad, conflict free)
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        __syncthreads();   // rings (first turn) / the slots read in the previous turn are free
        if (wave >= half && wave < 2 * half) {
            float4 *o = reinterpret_cast<float4 *>(lds + (wave - half) * 4096) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[(0 + q) * 64] = make_float4(c00[4 * q], c00[4 * q + 1], c00[4 * q + 2], c00[4 * q + 3]);
                o[(4 + q) * 64] = make_float4(c01[4 * q], c01[4 * q + 1], c01[4 * q + 2], c01[4 * q + 3]);
                o[(8 + q) * 64] = make_float4(c10[4 * q], c10[4 * q + 1], c10[4 * q + 2], c10[4 * q + 3]);
                o[(12 + q) * 64] = make_float4(c11[4 * q], c11[4 * q + 1], c11[4 * q + 2], c11[4 * q + 3]);
            }
            if (h == 0) {
                bs[(wave - half) * 64 + l] = as0;
                bs[(wave - half) * 64 + 32 + l] = as1;
            }
        }
        __syncthreads();
        if (wave < half) {
            const float4 *o = reinterpret_cast<const float4 *>(lds + wave * 4096) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x0 = o[(0 + q) * 64], x1 = o[(4 + q) * 64], x2 = o[(8 + q) * 64], x3 = o[(12 + q) * 64];
                c00[4 * q] += x0.x; c00[4 * q + 1] += x0.y; c00[4 * q + 2] += x0.z; c00[4 * q + 3] += x0.w;
                c01[4 * q] += x1.x; c01[4 * q + 1] += x1.y; c01[4 * q + 2] += x1.z; c01[4 * q + 3] += x1.w;
                c10[4 * q] += x2.x; c10[4 * q + 1] += x2.y; c10[4 * q + 2] += x2.z; c10[4 * q + 3] += x2.w;
                c11[4 * q] += x3.x; c11[4 * q + 1] += x3.y; c11[4 * q + 2] += x3.z; c11[4 * q + 3] += x3.w;
            }
            as0 += bs[wave * 64 + l];
            as1 += bs[wave * 64 + 32 + l];
        }
    }
    float *red = lds;   // [64][DW_RED_LD] row-major tile: over slot 0 (read by wave 0 only, just now) and the start of slot 1
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * h + (r & 3);
            red[row * DW_RED_LD + l] = c00[r];
            red[row * DW_RED_LD + 32 + l] = c01[r];
            red[(32 + row) * DW_RED_LD + l] = c10[r];
            red[(32 + row) * DW_RED_LD + 32 + l] = c11[r];
        }
        if (h == 0) {
            bs[l] = as0;
            bs[32 + l] = as1;
        }
    }
    __syncthreads();
    // row-major from here: thread -> row tid / 8, columns 8 (tid % 8) .. + 7 (two float4 of every state array)
    const int erow = tid >> 3, ecol = (tid & 7) * 8;
    float v[8];
    {
        const float4 lo = *reinterpret_cast<const float4 *>(red + erow * DW_RED_LD + ecol);
        const float4 hi = *reinterpret_cast<const float4 *>(red + erow * DW_RED_LD + ecol + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    float sb = (tid < 64) ? bs[tid] : 0.f;
    const int em = m0 + erow, en = n0 + ecol;
    const bool live = erow < vm && ecol < vn;          // this thread's 8 columns hold outputs (vn is a multiple of 8 or 16)
    const bool live_hi = live && ecol + 4 < vn;
    const int base = (int)(p.C - (ADAM ? F->grads_base : p.C)) + em * p.ldc + en;   // arena index of v[0] (ADAM)
    const bool vec_lo = live && en + 3 < p.n_store, vec_hi = live_hi && en + 7 < p.n_store;
    DW_STAMP(2);
    if (X.S > 1) {
        float *mine = X.part + ((size_t)tile * X.S + s) * DW_PART;
        wt_store4(mine + erow * 64 + ecol, make_float4(v[0], v[1], v[2], v[3]));
        wt_store4(mine + erow * 64 + ecol + 4, make_float4(v[4], v[5], v[6], v[7]));
        if (tid < 64) wt_store(mine + 4096 + tid, sb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have completed
        __syncthreads();
        if (tid == 0) {
            const unsigned long long old = __hip_atomic_fetch_add(X.ticket + tile, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = ((old + 1ull) % (unsigned long long)X.S == 0ull) ? 1 : 0;
        }
        __syncthreads();
        DW_STAMP(3);
        if (!*flag) return;   // somebody else finishes this tile
    }
    // optimizer state of this thread's elements: cold loads, in flight together with the partial tiles
    AdamState4 st_lo, st_hi;
    if (ADAM) {
        if (vec_lo) adam_fetch4(st_lo, *F, base);
        if (vec_hi) adam_fetch4(st_hi, *F, base + 4);
    }
    if (X.S > 1) {   // slice order, whoever arrived last; device-scope (sc1) loads
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const float *q = X.part + (size_t)tile * X.S * DW_PART;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(q), 0, (int)(X.S * DW_PART * sizeof(float)), 0x00020000);
        const int off = (erow * 64 + ecol) * 4;
        for (int sl = 0; sl < X.S; ++sl) {
            const int o = sl * DW_PART * 4;
            const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + off, 0, 16);
            const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + off + 16, 0, 16);
            const float b = (tid < 64) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o + (4096 + (tid & 63)) * 4, 0, 16)) : 0.f;
            const float w[8] = {__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w),
                                __uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sl ? v[j] + w[j] : w[j];
            sb = sl ? sb + b : b;
        }
    }
    DW_STAMP(4);
    if (p.bias_grad != nullptr && tn == 0 && tid < vm) {
        p.bias_grad[m0 + tid] = sb;
        if (ADAM) adam_apply(*F, (int)(p.bias_grad - F->grads_base) + m0 + tid, sb);
    }
    if (live) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half && !live_hi) break;
            const float g[4] = {v[4 * half], v[4 * half + 1], v[4 * half + 2], v[4 * half + 3]};
            const bool vec = half ? vec_hi : vec_lo;
            float *c = p.C + (long long)em * p.ldc + en + 4 * half;
            if (!ADAM || F->keep_grads) {
                if (vec) *reinterpret_cast<float4 *>(c) = make_float4(g[0], g[1], g[2], g[3]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (en + 4 * half + j < p.n_store) c[j] = g[j];
                }
            }
            if (ADAM) {
                if (vec) adam_apply4(*F, base + 4 * half, g, half ? st_hi : st_lo);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (en + 4 * half + j < p.n_store) adam_apply(*F, base + 4 * half + j, g[j]);
                }
            }
        }
    }
    DW_STAMP(5);
}
