// mt19937_device.h -- workgroup-cooperative MT19937 + numpy-legacy draws for gfx950.
//
// Stands in for numpy's legacy global RandomState on the reference's hot path
// (her.py:24,25,29,31; replay_buffer.py:64,67).  The sequence of 32-bit words is that of
// numpy/random/src/mt19937/mt19937.c; the derived draws are
//   randint(low, high)  : masked rejection on single words (distributions.c,
//                         random_bounded_uint64_fill, use_masked, rng <= 0xFFFFFFFF)
//   uniform()/random    : ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53
// Rejection makes the number of words per sample data dependent and the four draws of
// one HER batch are consecutive in ONE stream, so the draw is inherently sequential in
// the stream position.  Design: ONE MT_THREADS-thread workgroup owns the stream.
//   * the 624-word key blocks live in an LDS ring of 4 blocks; block j+1 is produced from
//     block j by a 3-phase parallel twist (k<227 reads only old words, 227<=k<454 reads the
//     first phase's outputs, k>=454 the second's);
//   * a draw consumes the stream in chunks of MT_THREADS (bounded) or 2 x MT_THREADS (double) words:
//     each thread tempers one candidate, a ballot/popcount prefix sum compacts the
//     accepted ones, and the position of the last accepted word advances the cursor;
//   * the final (key, pos) is written back in numpy's own representation, so
//     np.random.get_state() / set_state() interoperate at any point.
// All control flow below is workgroup-uniform; only `tid`-indexed work differs.
#pragma once
#include "internal.h"

#define MT_M 397
#define MT_THREADS 512     // one candidate word per thread and chunk: fewer, larger chunks = fewer barriers per drawn index
#define MT_WAVES (MT_THREADS / 64)
#define MT_IBUF (2 * MT_WAVES + 1)   // LDS ints: wave totals, two sets used alternately [0 .. 2 MT_WAVES), last-accept position [2 MT_WAVES]

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)),
// and the draws below store their results to global memory between barriers: every barrier then waited a memory round
// trip for stores nobody in this kernel reads back -- ~190 barriers per 4096-sample HER batch, 37 us for a draw whose
// twists and compactions are ~10 us of work.  The one place where a thread reads what another thread stored (p[i].t in
// mt_her_plan) keeps the full barrier.
__device__ __forceinline__ void mt_sync() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct MtWg {
    uint32_t (*blk)[MT_N];  // LDS ring [4][624]
    int *ibuf;              // LDS ints [MT_IBUF]: wave totals (two sets), last-accept position
    int flip;               // which set of wave totals the next mt_prefix uses
    long long cursor;       // absolute stream index of the next unconsumed word (block 0 = loaded key)
    int nblk;               // blocks generated so far (ring holds blocks nblk-4 .. nblk-1)
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t mt_twist_one(const uint32_t *src, const uint32_t *dst, int k) {
    uint32_t nxt = (k + 1 < MT_N) ? src[k + 1] : dst[0];
    uint32_t y = (src[k] & 0x80000000u) | (nxt & 0x7fffffffu);
    uint32_t far = (k + MT_M < MT_N) ? src[k + MT_M] : dst[k + MT_M - MT_N];
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// produce block nblk from block nblk-1 (workgroup-wide)
__device__ __forceinline__ void mt_generate_block(MtWg &g) {
    const uint32_t *src = g.blk[(g.nblk - 1) & 3];
    uint32_t *dst = g.blk[g.nblk & 3];
    const int tid = threadIdx.x;
    if (tid < MT_N - MT_M) dst[tid] = mt_twist_one(src, dst, tid);  // k in [0,227)
    mt_sync();
    if (tid < MT_N - MT_M) dst[227 + tid] = mt_twist_one(src, dst, 227 + tid);  // k in [227,454)
    mt_sync();
    if (tid < MT_N - 454) dst[454 + tid] = mt_twist_one(src, dst, 454 + tid);  // k in [454,624)
    mt_sync();
    g.nblk += 1;
}

__device__ __forceinline__ void mt_ensure(MtWg &g, long long abs_end) {
    while ((long long)g.nblk * MT_N < abs_end) mt_generate_block(g);
}

__device__ __forceinline__ uint32_t mt_word(const MtWg &g, long long abs) {
    int b = (int)(abs / MT_N);
    int o = (int)(abs - (long long)b * MT_N);
    return mt_temper(g.blk[b & 3][o]);
}

__device__ __forceinline__ void mt_load(MtWg &g, const MtState *st, uint32_t (*ring)[MT_N], int *ibuf) {
    g.blk = ring;
    g.ibuf = ibuf;
    for (int k = threadIdx.x; k < MT_N; k += MT_THREADS) ring[0][k] = st->key[k];
    g.cursor = st->pos;
    g.nblk = 1;
    g.flip = 0;
    __syncthreads();
}

__device__ __forceinline__ void mt_store(const MtWg &g, MtState *st) {
    // numpy keeps (block, pos) with pos in [0,624]; a cursor on a block boundary belongs to the
    // block just finished (pos == 624 -> "twist before the next word").
    long long c = g.cursor;
    int b, pos;
    if (c > 0 && c % MT_N == 0) {
        b = (int)(c / MT_N) - 1;
        pos = MT_N;
    } else {
        b = (int)(c / MT_N);
        pos = (int)(c % MT_N);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < MT_N; k += MT_THREADS) st->key[k] = g.blk[b & 3][k];
    if (threadIdx.x == 0) st->pos = pos;
}

// exclusive prefix of a predicate over the MT_THREADS-thread workgroup; returns this thread's rank among the
// accepting threads and the workgroup total.  ONE barrier: consecutive calls alternate between two sets of wave totals, so
// a fast wave writing the next call's totals cannot overtake a slow wave still reading this call's (it would have to pass
// the next call's barrier first, which the slow wave has not reached).
__device__ __forceinline__ int mt_prefix(MtWg &g, bool acc, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *tot_w = g.ibuf + g.flip * MT_WAVES;
    g.flip ^= 1;
    unsigned long long m = __ballot(acc);
    int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) tot_w[wave] = __popcll(m);
    mt_sync();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MT_WAVES; ++w) {
        const int c = tot_w[w];
        off += (w < wave) ? c : 0;
        tot += c;
    }
    total = tot;
    return off + within;
}

// legacy randint: `count` values uniform in [0, rng] by masked rejection; emit(i, value).
template <class Emit>
__device__ __forceinline__ void mt_draw_bounded(MtWg &g, uint32_t rng, long long count, Emit emit) {
    if (rng == 0u) {  // numpy consumes nothing
        for (long long i = threadIdx.x; i < count; i += MT_THREADS) emit(i, 0u);
        return;
    }
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    long long produced = 0;
    while (produced < count) {
        mt_ensure(g, g.cursor + MT_THREADS);
        uint32_t v = mt_word(g, g.cursor + threadIdx.x) & mask;
        bool acc = v <= rng;
        int total;
        int rank = mt_prefix(g, acc, total);
        long long idx = produced + rank;
        if (acc && idx < count) emit(idx, v);
        if (produced + total >= count) {
            if (acc && idx == count - 1) g.ibuf[2 * MT_WAVES] = threadIdx.x;
            mt_sync();
            g.cursor += g.ibuf[2 * MT_WAVES] + 1;
            produced = count;
            mt_sync();
        } else {
            g.cursor += MT_THREADS;
            produced += total;
        }
    }
}

// `count` doubles in [0,1): two words each; emit(i, u).
template <class Emit>
__device__ __forceinline__ void mt_draw_double(MtWg &g, long long count, Emit emit) {
    long long produced = 0;
    while (produced < count) {
        long long n = count - produced;
        if (n > MT_THREADS) n = MT_THREADS;
        mt_ensure(g, g.cursor + 2 * n);
        if ((long long)threadIdx.x < n) {
            uint32_t a = mt_word(g, g.cursor + 2 * threadIdx.x) >> 5;
            uint32_t b = mt_word(g, g.cursor + 2 * threadIdx.x + 1) >> 6;
            double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            emit(produced + threadIdx.x, u);
        }
        g.cursor += 2 * n;
        produced += n;
    }
}

// Stores of a draw's results.  WT: write-through (agent-scope relaxed atomic = sc1 store), for results that ANOTHER workgroup of
// the same launch reads after a flag (k_cycle_open); plain otherwise (the reader is a later launch).
template <bool WT> __device__ __forceinline__ void mt_put(int *p, int v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool WT> __device__ __forceinline__ void mt_put(long long *p, long long v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// replay_buffer._get_storage_idx (replay_buffer.py:57-71) for `inc` new episodes on top of `cur` stored ones: consecutive slots
// while the buffer has room, uniform random ones for the overflow (the only branch that consumes the stream).
template <bool WT = false>
__device__ __forceinline__ void mt_draw_slots(MtWg &g, long long cur, long long size, long long inc, long long *slots) {
    if (cur + inc <= size) {
        for (long long i = threadIdx.x; i < inc; i += MT_THREADS) mt_put<WT>(slots + i, cur + i);
    } else if (cur < size) {
        const long long head = size - cur, overflow = inc - head;
        for (long long i = threadIdx.x; i < head; i += MT_THREADS) mt_put<WT>(slots + i, cur + i);
        mt_draw_bounded(g, (uint32_t)(cur - 1), overflow, [&](long long i, uint32_t v) { mt_put<WT>(slots + head + i, (long long)v); });
    } else {
        mt_draw_bounded(g, (uint32_t)(size - 1), inc, [&](long long i, uint32_t v) { mt_put<WT>(slots + i, (long long)v); });
    }
}

// her.py:24-33 for `n_batches` consecutive minibatches (shared by k_draw_plan and the plan workgroup that rides
// along with the backward slab kernel): plan[b*batch + i] = (e, t, future_t, her).  Must be executed by exactly
// MT_THREADS threads of one workgroup (threadIdx.x < MT_THREADS); `ring` = uint32[4][624], `ibuf` = int[MT_IBUF] in LDS.
// the draws of mt_her_plan on a stream already loaded into LDS (several plans in one kernel: k_draw_plan2)
template <bool WT = false>
__device__ __forceinline__ void mt_her_draw(MtWg &g, long long n_eps, int T, long long batch, int n_batches, double future_p,
                                            PlanRec *plan) {
    if (n_eps <= 0 || T <= 0) return;  // host refuses this case (ValueError: high <= 0)
    for (int b = 0; b < n_batches; ++b) {
        PlanRec *p = plan + (long long)b * batch;
        mt_draw_bounded(g, (uint32_t)(n_eps - 1), batch, [&](long long i, uint32_t v) { mt_put<WT>(&p[i].e, (int)v); });
        mt_draw_bounded(g, (uint32_t)(T - 1), batch, [&](long long i, uint32_t v) { mt_put<WT>(&p[i].t, (int)v); });
        mt_draw_double(g, batch, [&](long long i, double u) { mt_put<WT>(&p[i].her, (u < future_p) ? 1 : 0); });
        __syncthreads();  // p[i].t may have been written by another thread
        mt_draw_double(g, batch, [&](long long i, double u) {
            int t = p[i].t;
            double off = u * (double)(T - t);  // her.py:31 (float64 * int64)
            mt_put<WT>(&p[i].fut, t + 1 + (int)off);   // her.py:32-33 (astype(int) truncates)
        });
        __syncthreads();
    }
}

__device__ __forceinline__ void mt_her_plan(MtState *st, long long n_eps, int T, long long batch, int n_batches,
                                            double future_p, PlanRec *plan, uint32_t (*ring)[MT_N], int *ibuf) {
    MtWg g;
    mt_load(g, st, ring, ibuf);
    if (n_eps <= 0 || T <= 0) return;  // host refuses this case (ValueError: high <= 0)
    mt_her_draw(g, n_eps, T, batch, n_batches, future_p, plan);
    mt_store(g, st);
}
