// peer.h -- one-shot all-reduce over peer memory (peer.hip): shared declarations of the library (not part of the C ABI)
#pragma once
#include "internal.h"

#define HP_PEER_MAX 16        // ranks of one node
#define HP_PEER_SMALL 1024    // floats of a mailbox (the normalizer exchanges 55 and 7)
#define HP_PEER_TILES 512     // weight-gradient tiles of one launch that can exchange by themselves (flags_t)

struct PeerDev {              // by-value kernel argument: where every rank's exchange memory is mapped in THIS process
    int rank, world;
    unsigned long long *flags_g[HP_PEER_MAX];   // rank q's gradient-channel flags (slot w written by rank w)
    unsigned long long *flags_s[HP_PEER_MAX];   // rank q's mailbox-channel flags
    float *small[HP_PEER_MAX][2];               // rank q's mailboxes (ping-pong by epoch parity)
    float *grad[HP_PEER_MAX][2];                // rank q's gradient vectors (ping-pong)
    unsigned long long *flags_r[HP_PEER_MAX];   // two-phase exchange: rank q's "reduced slice ready" flags
    float *red[HP_PEER_MAX][2];                 // two-phase exchange: rank q's buffer of reduced sums (it fills ITS slice)
    unsigned long long *flags_t[HP_PEER_MAX];   // tile-wise exchange: rank q's [HP_PEER_TILES][HP_PEER_MAX] epochs (slot [t][w] written by rank w)
    unsigned long long *epoch;                  // local: [0] gradient channel base, [1] mailbox channel
    unsigned int *error;                        // local, sticky: a wait timed out (every later exchange kernel returns at once)
    unsigned int *error_host;                   // the same word in pinned host memory: the host reads it without a sync
    unsigned long long timeout_ticks;           // 100 MHz ticks
};

struct hp_peer {
    hp_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    size_t n_grad = 0, bytes = 0;
    void *local = nullptr;
    void *remote[HP_PEER_MAX] = {nullptr};
    unsigned long long *d_epoch = nullptr;
    unsigned int *h_error = nullptr;           // pinned + mapped: set by the kernel whose wait timed out
    unsigned char handle[64] = {0};
    bool connected = false;
    int phases = 1;            // 1: every rank reads all peers' whole vectors; 2: reduce-scatter + all-gather (peer.hip)
    bool gate = false;         // every wait of the gradient exchange in a one-wavefront kernel of its own (hp_peer_set_gate)
    bool tiles = true;         // one-shot form inside the weight-gradient launch: every tile exchanges by itself (RLARM_PEER_TILES=0: off)
    PeerDev dev;
    PeerDev *d_dev = nullptr;  // device copy of `dev` (hp_peer_connect): what kernels whose argument block has no room for it read (k_fb_split8)
};

float *peer_grad_buffer(hp_peer *p, int parity);
int peer_enqueue_seq_end(hp_peer *p, int n_updates);
int peer_enqueue_reduce_slice(hp_peer *p, int n4, int u, bool mean);   // phase 1 of the two-phase exchange
int peer_allreduce_small(hp_peer *p, float *dev, size_t n, bool mean);
int peer_enqueue_gate(hp_peer *p, int channel, int u);                 // no-op unless p->gate; channel 1 gradients, 3 reduced slices
// a wait of an earlier exchange kernel timed out: optimizer steps were skipped, the replicas are no longer in step.  Free
// for the host (a read of pinned memory); what every entry point that enqueues exchange kernels checks first.
inline bool peer_failed(const hp_peer *p) { return p && p->h_error && *(volatile const unsigned int *)p->h_error != 0u; }
int peer_check_alive(const hp_peer *p, const char *who);   // HP_ERR_STATE + message when peer_failed

#ifdef __HIPCC__
// ---- device helpers shared by peer.hip (mailboxes) and agent.hip (k_peer_adam: gradients + optimizer) ----------------
__device__ __forceinline__ unsigned long long now_ticks() { return wall_clock64(); }   // 100 MHz

__device__ __forceinline__ void peer_signal(const PeerDev &D, unsigned long long *const *flags, unsigned long long epoch) {
    // one lane per peer: my slot in the peer's flag array
    const int q = threadIdx.x;
    if (q < D.world && q != D.rank) __hip_atomic_store(flags[q] + D.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... the same for one slot row of a per-tile flag array: my slot of row `row` on every peer
__device__ __forceinline__ void peer_signal_row(const PeerDev &D, unsigned long long *const *flags, int row, unsigned long long epoch) {
    const int q = threadIdx.x;
    if (q < D.world && q != D.rank)
        __hip_atomic_store(flags[q] + (size_t)row * HP_PEER_MAX + D.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every peer has signalled `epoch`?  lane q of the first wave polls the local slot of peer q.  Returns the same answer to
// every thread of the workgroup (it contains a barrier: call it from uniform control flow).  false = the exchange is dead:
// a wait timed out now or in an earlier kernel (sticky) -- the caller must NOT consume the peers' buffers or step the
// optimizer; the host sees the pinned error word at its next call (peer_failed) and every later exchange kernel returns at
// once instead of stalling for another timeout.
__device__ __forceinline__ bool peer_wait(const PeerDev &D, unsigned long long *mine, unsigned long long epoch, unsigned channel = 1u) {
    __shared__ int s_peer_ok;
    __syncthreads();   // a second call in one kernel must not overwrite the verdict a slow wave of the first call has yet to read
    if (threadIdx.x < 64) {
        const int q = threadIdx.x & 63;
        const bool poll = q < D.world && q != D.rank;
        bool ok = __hip_atomic_load(D.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        if (ok) {
            const unsigned long long t0 = now_ticks();
            for (;;) {
                bool here = true;
                if (poll) here = __hip_atomic_load(mine + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch;
                if (__all(here)) break;
                __builtin_amdgcn_s_sleep(16);
                if (now_ticks() - t0 > D.timeout_ticks) {
                    // error word: bit 0 = dead, bits 4-7 = channel (1 gradients, 2 mailboxes, 3 reduced slices),
                    // bits 8-23 = ranks whose flag had not arrived, bits 24-31 = low bits of the epoch waited for
                    const unsigned missing = (unsigned)(__ballot(!here) & 0xffffull);
                    if (threadIdx.x == 0) {
                        const unsigned word = 1u | (channel << 4) | (missing << 8) | ((unsigned)(epoch & 0xffull) << 24);
                        __hip_atomic_store(D.error, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(D.error_host, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    ok = false;
                    break;
                }
            }
        }
        if (threadIdx.x == 0) s_peer_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return s_peer_ok != 0;
}

// write-through to system scope: what a peer GPU's system-scope load must find (own exchange memory)
__device__ __forceinline__ void peer_store4(float *p, const float4 v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void peer_store1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// float4s [lo, lo + per) of the vector are rank r's slice in the two-phase exchange
__device__ __forceinline__ int peer_slice_len(const PeerDev &D, int n4) { return (n4 + D.world - 1) / D.world; }

// 4 consecutive floats of rank q's vector: the own vector with a plain load (written by the previous kernel), a peer's
// with a system-scope load (sc0 sc1: never served from a stale line of this GPU's caches)
__device__ __forceinline__ float4 peer_load4(const float *base, size_t bytes, unsigned off_bytes, bool own) {
    if (own) return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_bytes, 0, /*sc0 | sc1*/ 17);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}


#endif
