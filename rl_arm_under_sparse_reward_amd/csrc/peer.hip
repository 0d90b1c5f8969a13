// peer.hip -- the rank exchange of the hot path as ONE-SHOT all-reduces over peer memory (xGMI), fused with the optimizer.
//
// What the reference exchanges per update (utils.py:43-48, via ddpg_agent.py:271,276): Allreduce(SUM) of the flat
// gradients of both networks, 1.17 MB here, between backward and the Adam steps -- on the critical path of every ~41 us
// update, because the next forward needs the stepped weights.  A ring all-reduce of that size is pure latency (2(N-1)
// dependent hops).  xGMI is a full mesh: every GPU reaches every peer's HBM directly, so each rank can simply READ the
// other ranks' gradient vectors and sum them itself, in rank order (deterministic, and bit-identical on every rank), and
// apply Adam to the sum in the same kernel: one hop, one launch, no second pass over the gradients.
//
//   rank r, update with epoch e:
//     k_gemm_lds   weight gradients -> G_r[e & 1]                    (own buffer; the kernel boundary makes it visible)
//     k_peer_adam  (a) flags_q[r] := e on every peer q               (system-scope stores over the fabric)
//                  (b) wait until flags_r[q] >= e for every q         (local polls; bounded by wall clock)
//                  (c) g = G_0[e&1][i] + G_1[e&1][i] + ... in rank order (peers: system-scope loads), Adam(g)
//   Two gradient buffers alternate: a rank that passes the barrier of epoch e + 1 has finished reading epoch e's
//   buffers everywhere it matters, so G[e & 1] may be rewritten at epoch e + 2 without a second barrier.
//   Epochs live in device memory (base + index of the update in its sequence), so a captured hipGraph replays correctly.
//
// Two-phase variant (default from 4 ranks, RLARM_PEER_PHASES=1|2): the one-shot form pulls every peer's WHOLE vector over
// its link (1.17 MB per link and update whatever the world size); with W ranks a reduce-scatter + all-gather moves 2/W of
// that per link -- rank r sums slice r of all vectors in rank order into its `red` buffer (k_peer_reduce_slice), a second
// barrier, then every rank reads the W reduced slices from their owners and steps (k_peer_adam2 in agent.hip).  Same sums in
// the same order: bit-identical to the one-shot form; one more kernel boundary and flag round trip (~4-5 us), so it only
// pays once the link time it saves exceeds that (8 ranks: 2 x 146 KB instead of 1.17 MB per link).  Not measured on a
// multi-GPU node (none available to the build); both forms are exercised by two processes sharing one device.
//
// The per-cycle normalizer exchange (normalizer.py:60-64, 62 floats) uses the same mechanism through small mailboxes
// (k_peer_small: copy in, flag, poll, sum in rank order, optional / world).
//
// Memory: one fine-grained device allocation per rank ([flags | mailboxes | 2 x gradients | 2 x reduced sums]) exported with
// hipIpcGetMemHandle; the 64-byte handles travel through any side channel (the Python mirror uses torch.distributed).
// RCCL (comm.hip) stays as the fallback transport; utils.Communicator picks this one when a self-check passes.
#include "internal.h"

#include <cstdlib>

#include "peer.h"

// the gradient channel's epoch base moves on by an EVEN count per sequence, so that the buffer parity of update u is
// known on the host when the launches are recorded
__global__ void k_peer_seq_end(const PeerDev D, int n_updates) {
    if (blockIdx.x == 0 && threadIdx.x == 0) D.epoch[0] += (unsigned long long)(n_updates + (n_updates & 1));
}

// Gate (hp_peer_set_gate): signal + wait of one channel as ONE wavefront.  The kernel behind it signals the same epoch again
// (idempotent) and finds every flag there, so its own wait falls through.  On one shared device this is what keeps a rank
// that is still computing from being starved by the other ranks' waiting workgroups.
__global__ __launch_bounds__(64) void k_peer_gate(const PeerDev D, int channel, int u) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    unsigned long long *const *flags = channel == 3 ? D.flags_r : D.flags_g;
    peer_signal(D, flags, epoch);
    (void)peer_wait(D, flags[D.rank], epoch, (unsigned)channel);
}

// small vectors (normalizer sums): one workgroup.  vec[n] := sum over ranks (/ world if mean), in place.
__global__ __launch_bounds__(256) void k_peer_small(const PeerDev D, float *vec, int n, int mean) {
    const unsigned long long epoch = D.epoch[1] + 1ull;
    const int par = (int)(epoch & 1ull);
    float *mine = D.small[D.rank][par];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(mine + i, vec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through, system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: stores complete before the flag goes out
    __syncthreads();
    peer_signal(D, D.flags_s, epoch);
    if (!peer_wait(D, D.flags_s[D.rank], epoch, 2u)) return;   // dead exchange: leave vec alone, the host raises (peer_failed)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float acc = 0.f;
        for (int q = 0; q < D.world; ++q) {
            const float v = (q == D.rank) ? vec[i]
                                          : __hip_atomic_load(D.small[q][par] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            acc = (q == 0) ? v : acc + v;
        }
        vec[i] = mean ? acc / (float)D.world : acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) D.epoch[1] = epoch;
}

// Two-phase exchange, phase 1: this rank reduces ITS slice of the gradient vectors, in rank order, into red[rank][par]
__global__ __launch_bounds__(256) void k_peer_reduce_slice(const PeerDev D, int n4, int u, int mean) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_g, epoch);
    if (!peer_wait(D, D.flags_g[D.rank], epoch)) return;
    const int per = peer_slice_len(D, n4);
    const int lo = D.rank * per, hi = (lo + per) < n4 ? (lo + per) : n4;
    const int t = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= hi) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t bytes = (size_t)n4 * 16;
    for (int q = 0; q < D.world; ++q) {   // rank order: the same float32 sum as the one-shot form
        const float4 v = peer_load4(D.grad[q][par], bytes, (unsigned)t * 16u, q == D.rank);
        if (q == 0) acc = v;
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (mean) {
        const float w = (float)D.world;
        acc.x /= w; acc.y /= w; acc.z /= w; acc.w /= w;
    }
    *reinterpret_cast<float4 *>(D.red[D.rank][par] + 4 * (size_t)t) = acc;   // own memory; the kernel boundary publishes it
}

// Self-check of the GRADIENT channel (the path k_peer_adam uses: flags, buffer parity, system-scope 16-byte loads of the
// peers' vectors): every rank fills its own buffer with an exactly representable pattern, the ranks reduce it like
// k_peer_adam does and count the elements that differ from the known sum.  Run at attach time for both buffer parities.
__global__ void k_peer_check_fill(const PeerDev D, int n, int par) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) D.grad[D.rank][par][i] = (float)(D.rank + 1) * (float)((i % 1021) + 1);
}
__global__ __launch_bounds__(256) void k_peer_check_reduce(const PeerDev D, int n4, int u, unsigned int *bad) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_g, epoch);
    if (!peer_wait(D, D.flags_g[D.rank], epoch)) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < D.world; ++q) {
        const float4 v = peer_load4(D.grad[q][par], (size_t)n4 * 16, (unsigned)t * 16u, q == D.rank);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float w = (float)(D.world * (D.world + 1) / 2);
    const float got[4] = {acc.x, acc.y, acc.z, acc.w};
    int wrong = 0;
    for (int j = 0; j < 4; ++j) wrong += got[j] != w * (float)(((4 * t + j) % 1021) + 1);
    if (wrong) atomicAdd(bad, (unsigned)wrong);
}

// ... and of the two-phase form: k_peer_reduce_slice (above) followed by this gather + compare
__global__ __launch_bounds__(256) void k_peer_check_gather(const PeerDev D, int n4, int u, unsigned int *bad) {
    const unsigned long long epoch = D.epoch[0] + (unsigned long long)u + 1ull;
    const int par = (int)(epoch & 1ull);
    if (blockIdx.x == 0) peer_signal(D, D.flags_r, epoch);
    if (!peer_wait(D, D.flags_r[D.rank], epoch, 3u)) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const int owner = t / peer_slice_len(D, n4);
    const float4 acc = peer_load4(D.red[owner][par], (size_t)n4 * 16, (unsigned)t * 16u, owner == D.rank);
    const float w = (float)(D.world * (D.world + 1) / 2);
    const float got[4] = {acc.x, acc.y, acc.z, acc.w};
    int wrong = 0;
    for (int j = 0; j < 4; ++j) wrong += got[j] != w * (float)(((4 * t + j) % 1021) + 1);
    if (wrong) atomicAdd(bad, (unsigned)wrong);
}

// ---- internal entry points used by agent.hip -----------------------------------------------------------------------
float *peer_grad_buffer(hp_peer *p, int parity) { return p->dev.grad[p->rank][parity & 1]; }

int peer_check_alive(const hp_peer *p, const char *who) {
    if (!peer_failed(p)) return HP_OK;
    const unsigned w = *(volatile const unsigned int *)p->h_error;
    static const char *chan[4] = {"?", "gradient", "normalizer mailbox", "reduced-slice"};
    HP_REQUIRE(false, HP_ERR_STATE,
               "%s: the peer-memory exchange is dead -- rank %d waited longer than the bound (RLARM_PEER_TIMEOUT_S) on the %s "
               "channel for ranks 0x%x (epoch ..%u): late or gone.  The optimizer steps from that update on were skipped on this "
               "rank and the replicas are no longer in step",
               who, p->rank, chan[(w >> 4) & 3u], (w >> 8) & 0xffffu, w >> 24);
    return HP_OK;
}

int peer_enqueue_seq_end(hp_peer *p, int n_updates) {
    HP_KLOG("k_peer_seq_end");
    hipLaunchKernelGGL(k_peer_seq_end, dim3(1), dim3(64), 0, p->ctx->stream, p->dev, n_updates);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int peer_enqueue_gate(hp_peer *p, int channel, int u) {
    if (!p->gate) return HP_OK;
    HP_KLOG("k_peer_gate");
    hipLaunchKernelGGL(k_peer_gate, dim3(1), dim3(64), 0, p->ctx->stream, p->dev, channel, u);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int peer_enqueue_reduce_slice(hp_peer *p, int n4, int u, bool mean) {
    HP_TRY(peer_enqueue_gate(p, 1, u));
    const int per = (n4 + p->world - 1) / p->world;
    HP_KLOG("k_peer_reduce_slice");
    hipLaunchKernelGGL(k_peer_reduce_slice, dim3((per + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev, n4, u, mean ? 1 : 0);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int peer_allreduce_small(hp_peer *p, float *dev, size_t n, bool mean) {
    HP_REQUIRE(n <= HP_PEER_SMALL, HP_ERR_INVALID, "peer all-reduce: %zu floats exceed the mailbox (%d)", n, HP_PEER_SMALL);
    HP_KLOG("k_peer_small");
    hipLaunchKernelGGL(k_peer_small, dim3(1), dim3(256), 0, p->ctx->stream, p->dev, dev, (int)n, mean ? 1 : 0);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// --------------------------------------------------------------------------------------------------------- C ABI
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PeerLayout {
    size_t flags_g, flags_s, flags_r, small, grad, red, flags_t, total;
    explicit PeerLayout(size_t n_grad) {
        size_t o = 0;
        flags_g = o; o += HP_PEER_MAX * 8;
        flags_s = o; o += HP_PEER_MAX * 8;
        flags_r = o; o += HP_PEER_MAX * 8;
        o = align_up(o, 256);
        small = o; o += 2 * (size_t)HP_PEER_SMALL * 4;
        o = align_up(o, 256);
        grad = o; o += 2 * align_up(n_grad * 4, 256);
        red = o; o += 2 * align_up(n_grad * 4, 256);
        flags_t = o; o += (size_t)HP_PEER_TILES * HP_PEER_MAX * 8;
        total = align_up(o, 4096);
    }
};

static void peer_fill_dev(hp_peer *p) {
    const PeerLayout L(p->n_grad);
    const size_t gstride = align_up(p->n_grad * 4, 256);
    for (int q = 0; q < p->world; ++q) {
        char *b = static_cast<char *>(p->remote[q]);
        p->dev.flags_g[q] = reinterpret_cast<unsigned long long *>(b + L.flags_g);
        p->dev.flags_s[q] = reinterpret_cast<unsigned long long *>(b + L.flags_s);
        p->dev.flags_r[q] = reinterpret_cast<unsigned long long *>(b + L.flags_r);
        p->dev.flags_t[q] = reinterpret_cast<unsigned long long *>(b + L.flags_t);
        for (int k = 0; k < 2; ++k) {
            p->dev.small[q][k] = reinterpret_cast<float *>(b + L.small) + (size_t)k * HP_PEER_SMALL;
            p->dev.grad[q][k] = reinterpret_cast<float *>(b + L.grad + k * gstride);
            p->dev.red[q][k] = reinterpret_cast<float *>(b + L.red + k * gstride);
        }
    }
}

// Failure injection for the start-up paths that only a multi-GPU node can fail for real (tests/test_gpu_bench_contract.py):
// RLARM_PEER_INJECT=ipc[@rank] -> hp_peer_connect fails on that rank (default 0) as if hipIpcOpenMemHandle had refused a peer's
// handle; =selfcheck[@rank] -> hp_peer_selfcheck reports mismatches there.  Either way every rank must agree to drop the
// exchange and take the next transport down (utils.Communicator.attach_peer), and the run must still print its line.
static bool peer_inject(const hp_peer *p, const char *what) {
    const char *e = getenv("RLARM_PEER_INJECT");
    const size_t n = strlen(what);
    if (!e || strncmp(e, what, n) != 0 || (e[n] != '\0' && e[n] != '@')) return false;
    return p->rank == (e[n] == '@' ? atoi(e + n + 1) : 0);
}

extern "C" {

void hp_peer_destroy(hp_peer *p);

int hp_peer_create(hp_ctx *ctx, int32_t rank, int32_t world, int64_t n_grad_floats, hp_peer **out, uint8_t *handle64) {
    HP_REQUIRE(ctx && out && handle64, HP_ERR_INVALID, "hp_peer_create: null argument");
    HP_REQUIRE(world >= 1 && world <= HP_PEER_MAX && rank >= 0 && rank < world, HP_ERR_INVALID,
               "hp_peer_create: rank %d of %d (at most %d ranks)", rank, world, HP_PEER_MAX);
    HP_REQUIRE(n_grad_floats > 0 && n_grad_floats % 4 == 0, HP_ERR_INVALID, "hp_peer_create: gradient length must be a positive multiple of 4");
    CtxGuard guard(ctx);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hp_peer *p = new hp_peer();
    p->ctx = ctx;
    p->rank = rank;
    p->world = world;
    p->n_grad = (size_t)n_grad_floats;
    p->phases = world >= 4 ? 2 : 1;
    if (const char *ph = getenv("RLARM_PEER_PHASES")) p->phases = atoi(ph) == 2 ? 2 : (atoi(ph) == 1 ? 1 : p->phases);
    if (const char *t = getenv("RLARM_PEER_TILES")) p->tiles = t[0] != '0';
    const PeerLayout L(p->n_grad);
    p->bytes = L.total;
    // fine-grained: coherent with the peers' system-scope accesses; plain device memory if the runtime refuses
    if (hipExtMallocWithFlags(&p->local, p->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        p->local = nullptr;
        if (hipMalloc(&p->local, p->bytes) != hipSuccess) {
            hp_set_error("hp_peer_create: cannot allocate %zu bytes of exchange memory", p->bytes);
            delete p;
            return HP_ERR_HIP;
        }
    }
    hipError_t e = hipMemset(p->local, 0, p->bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_epoch, 2 * sizeof(unsigned long long) + 16);
    if (e == hipSuccess) e = hipMemset(p->d_epoch, 0, 2 * sizeof(unsigned long long) + 16);
    void *h_err_dev = nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_error, 64, hipHostMallocMapped);
    if (e == hipSuccess) { *p->h_error = 0u; e = hipHostGetDevicePointer(&h_err_dev, p->h_error, 0); }
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p->local);
    if (e != hipSuccess) {
        hp_set_error("hp_peer_create: %s", hipGetErrorString(e));
        hp_peer_destroy(p);
        return HP_ERR_HIP;
    }
    memcpy(handle64, &h, 64);
    memcpy(p->handle, &h, 64);
    memset(&p->dev, 0, sizeof(p->dev));
    p->dev.rank = rank;
    p->dev.world = world;
    p->dev.epoch = p->d_epoch;
    p->dev.error = reinterpret_cast<unsigned int *>(p->d_epoch + 2);
    p->dev.error_host = static_cast<unsigned int *>(h_err_dev);
    double secs = 20.0;   // a rank may legitimately be late by a host-side pause (checkpoint, graph capture); a dead one must not hang us
    if (const char *t = getenv("RLARM_PEER_TIMEOUT_S")) secs = atof(t) > 0 ? atof(t) : secs;
    p->dev.timeout_ticks = (unsigned long long)(secs * 1e8);
    for (int q = 0; q < world; ++q) p->remote[q] = nullptr;
    p->remote[rank] = p->local;
    *out = p;
    return HP_OK;
}

int hp_peer_connect(hp_peer *p, const uint8_t *handles) {
    HP_REQUIRE(p && handles, HP_ERR_INVALID, "hp_peer_connect: null argument");
    CtxGuard guard(p->ctx);
    HP_REQUIRE(!peer_inject(p, "ipc"), HP_ERR_HIP, "hp_peer_connect: hipIpcOpenMemHandle(rank %d) failed: injected (RLARM_PEER_INJECT=ipc)",
               (p->rank + 1) % p->world);
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank) continue;
        HP_REQUIRE(memcmp(handles + 64 * q, p->handle, 64) != 0, HP_ERR_INVALID, "hp_peer_connect: rank %d sent this rank's own handle", q);
        hipIpcMemHandle_t h;
        memcpy(&h, handles + 64 * q, 64);
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            hp_set_error("hp_peer_connect: hipIpcOpenMemHandle(rank %d) failed: %s", q, hipGetErrorString(e));
            return HP_ERR_HIP;
        }
        p->remote[q] = ptr;
    }
    peer_fill_dev(p);
    // the same table in device memory, for the kernels that take it by pointer (written once, here; read-only afterwards)
    if (!p->d_dev) HP_CHECK_HIP(hipMalloc((void **)&p->d_dev, sizeof(PeerDev)));
    HP_CHECK_HIP(hipMemcpy(p->d_dev, &p->dev, sizeof(PeerDev), hipMemcpyHostToDevice));
    p->connected = true;
    return HP_OK;
}

int hp_peer_allreduce_f32(hp_peer *p, void *dev, int64_t n, int32_t mean) {
    HP_REQUIRE(p && dev && n >= 0, HP_ERR_INVALID, "hp_peer_allreduce_f32: bad argument");
    CtxGuard guard(p->ctx);
    HP_REQUIRE(p->connected, HP_ERR_STATE, "hp_peer_allreduce_f32: hp_peer_connect first");
    if (n == 0) return HP_OK;
    return peer_allreduce_small(p, static_cast<float *>(dev), (size_t)n, mean != 0);
}

int hp_peer_selfcheck(hp_peer *p, uint32_t *mismatches) {
    HP_REQUIRE(p && mismatches, HP_ERR_INVALID, "hp_peer_selfcheck: null argument");
    CtxGuard guard(p->ctx);
    HP_REQUIRE(p->connected, HP_ERR_STATE, "hp_peer_selfcheck: hp_peer_connect first");
    hipStream_t s = p->ctx->stream;
    unsigned int *bad = p->dev.error + 1;   // scratch word behind the sticky error word
    HP_CHECK_HIP(hipMemsetAsync(bad, 0, 4, s));
    const int n = (int)p->n_grad, n4 = n / 4;
    // two consecutive epochs = both buffer parities, exactly as two updates of a sequence would use them
    for (int u = 0; u < 2; ++u) {
        hipLaunchKernelGGL(k_peer_check_fill, dim3((n + 255) / 256), dim3(256), 0, s, p->dev, n, (u + 1) & 1);
        if (p->phases == 2) {
            HP_TRY(peer_enqueue_reduce_slice(p, n4, u, false));
            HP_TRY(peer_enqueue_gate(p, 3, u));
            hipLaunchKernelGGL(k_peer_check_gather, dim3((n4 + 255) / 256), dim3(256), 0, s, p->dev, n4, u, bad);
        } else {
            HP_TRY(peer_enqueue_gate(p, 1, u));
            hipLaunchKernelGGL(k_peer_check_reduce, dim3((n4 + 255) / 256), dim3(256), 0, s, p->dev, n4, u, bad);
        }
    }
    HP_CHECK_HIP(hipGetLastError());
    HP_TRY(peer_enqueue_seq_end(p, 2));
    uint32_t h[2] = {0, 0};
    HP_CHECK_HIP(hipMemcpyAsync(h, p->dev.error, 8, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    *mismatches = h[1] + (h[0] ? 0x80000000u : 0u);   // top bit: a wait timed out
    if (peer_inject(p, "selfcheck")) *mismatches += 7u;
    return HP_OK;
}

int hp_peer_status(hp_peer *p, uint32_t *error) {
    HP_REQUIRE(p && error, HP_ERR_INVALID, "hp_peer_status: null argument");
    CtxGuard guard(p->ctx);
    HP_CHECK_HIP(hipMemcpyAsync(error, p->dev.error, 4, hipMemcpyDeviceToHost, p->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
    if (peer_failed(p)) *error |= *(volatile const unsigned int *)p->h_error;
    return HP_OK;
}

int hp_peer_set_gate(hp_peer *p, int32_t on) {
    HP_REQUIRE(p, HP_ERR_INVALID, "hp_peer_set_gate: null handle");
    p->gate = on != 0;
    return HP_OK;
}

int hp_peer_phases(hp_peer *p, int32_t *phases) {
    HP_REQUIRE(p && phases, HP_ERR_INVALID, "hp_peer_phases: null argument");
    *phases = p->phases;
    return HP_OK;
}

void hp_peer_destroy(hp_peer *p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (int q = 0; q < p->world; ++q)
        if (q != p->rank && p->remote[q]) (void)hipIpcCloseMemHandle(p->remote[q]);
    if (p->local) (void)hipFree(p->local);
    if (p->d_epoch) (void)hipFree(p->d_epoch);
    if (p->d_dev) (void)hipFree(p->d_dev);
    if (p->h_error) (void)hipHostFree(p->h_error);
    delete p;
}

}  // extern "C"
