// cycle_open.hip -- everything a training cycle does before its first update, as ONE launch.
//
// ddpg_agent.py:143-147 per cycle: store_episode (replay_buffer.py:32-43: slots from _get_storage_idx, then the scatter),
// _update_normalizer (:187-212: an index plan over the new episodes, then the column sums and recompute_stats), and the first
// minibatch draws of _update_network (her.py:24-33).  All three draws come out of ONE MT19937 stream in that order, so they
// are one sequential job; the scatter only needs the slots and the normalizer only needs its plan.  As four launches that was
// 29.6 us of kernels per cycle (k_draw_slots 5.1 + k_store_scatter 6.4 + k_draw_plan2 11.3 + k_norm_update_from_plan 6.9, the
// first two outside the cycle graph).  Here:
//   workgroup 0            loads the stream once, draws slots -> flag, normalizer plan -> flag, first minibatch plans, stores the
//                          stream and the buffer counters;
//   workgroup 1            waits for the second flag, then runs the normalizer update (norm_device.h);
//   workgroups 2 ..        wait for the first flag, then scatter their share of a staged episode (store_device.h).
// The same device functions as the separate kernels: same words, same sums, same bits.
// Hand-off inside the launch (cdna_hip_programming.md Guideline 16, write-through form): the producer's results are sc1 stores
// (mt_put<true>), every wave drains them, barrier, one relaxed agent-scope flag store; consumers poll relaxed, then read the
// results with agent-scope loads.  The launch is 2 + OPEN_PARTS * n_new workgroups -- far fewer than the CUs, all resident at
// once, so the polls cannot starve the producer; they are bounded all the same (a timeout is counted in AgentDevState and
// reported by hp_agent_get_losses).  The last workgroup to leave clears the flags for the next replay of the graph.
#include "mt19937_device.h"
#include "norm_device.h"
#include "store_device.h"
#include "agent.h"

#define OPEN_THREADS NORM_THREADS
#define OPEN_PARTS 2   // workgroups per staged episode (2 x 1024 threads: the 8 x 256 of k_store_scatter)
static_assert(MT_THREADS == 2 * NORM_MAX, "the draw and the normalizer columns use the same 512 threads");

struct OpenArgs {
    // draws
    MtState *st;
    BufMeta *meta;
    long long size, inc;          // buffer capacity, staged episodes
    int T;
    long long *slots;
    PlanRec *norm_plan;           // T records over the staged episodes
    PlanRec *plan;                // first minibatch plans
    long long batch;
    int n_first;
    double future_p;
    // scatter
    const double *s_obs, *s_ag, *s_g, *s_act;
    double *obs, *ag, *g, *act;
    long long ep_obs, ep_ag, ep_g, ep_act;
    // normalizer
    NormDev *onz, *gnz;
    int obs_dim, goal_dim;
    double clip_obs;
    int recompute;
    double o_eps_sq, g_eps_sq;
    int o_std_f32, g_std_f32;
    int chunk_rows;
    // hand-off
    unsigned *sync;               // [0] slots ready, [1] normalizer plan ready, [2] workgroups that have left
    AgentDevState *dev;
    unsigned *fault, *fault_host; // the agent's sticky fault word and its pinned host mirror (agent.h: handoff_fault)
};

__device__ __forceinline__ void open_publish(unsigned *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have completed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// false (to every thread of the workgroup): the poll gave up -- the caller must NOT consume what it waited for (slots or plan
// that are not ready would corrupt the replay buffer / the normalizer silently); the fault word makes the next host call fail
__device__ __forceinline__ bool open_wait(unsigned *flag, const OpenArgs &A) {
    __shared__ int s_open_ok;
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(2);
        if (spins >= (1 << 22)) {
            atomicAdd(&A.dev->open_timeouts, 1u);
            handoff_fault(A.fault, A.fault_host, 3u);
        }
        s_open_ok = spins < (1 << 22) ? 1 : 0;
    }
    __syncthreads();
    return s_open_ok != 0;
}

__device__ __forceinline__ void open_leave(unsigned *sync) {   // thread 0 of every workgroup, as its last act
    const unsigned old = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == gridDim.x) {
        __hip_atomic_store(sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(OPEN_THREADS) void k_cycle_open(const OpenArgs A) {
    extern __shared__ __attribute__((aligned(16))) char open_lds[];
    const int role = blockIdx.x;
    if (role == 0) {
        if (threadIdx.x >= MT_THREADS) return;   // ended waves take no part in the barriers of the draws
        uint32_t(*ring)[MT_N] = reinterpret_cast<uint32_t(*)[MT_N]>(open_lds);
        int *ibuf = reinterpret_cast<int *>(open_lds + 4 * MT_N * sizeof(uint32_t));
        MtWg g;
        mt_load(g, A.st, ring, ibuf);
        const long long cur = A.meta->current_size;
        const long long now = (cur + A.inc < A.size) ? cur + A.inc : A.size;   // replay_buffer.py:68
        mt_draw_slots<true>(g, cur, A.size, A.inc, A.slots);
        open_publish(A.sync + 0);
        mt_her_draw<true>(g, A.inc, A.T, A.T, 1, A.future_p, A.norm_plan);
        open_publish(A.sync + 1);
        mt_her_draw(g, now, A.T, A.batch, A.n_first, A.future_p, A.plan);
        mt_store(g, A.st);
        if (threadIdx.x == 0) {
            A.meta->current_size = now;
            A.meta->n_transitions_stored += (long long)A.T * A.inc;   // replay_buffer.py:43
            open_leave(A.sync);
        }
    } else if (role == 1) {
        if (open_wait(A.sync + 1, A))
            norm_update_from_plan_body<true>(A.onz, A.gnz, A.norm_plan, (long long)A.T, A.s_obs, A.s_ag, A.s_g, A.T, A.obs_dim,
                                             A.goal_dim, A.clip_obs, A.recompute, A.o_eps_sq, A.o_std_f32, A.g_eps_sq, A.g_std_f32,
                                             A.chunk_rows, open_lds);
        if (threadIdx.x == 0) open_leave(A.sync);
    } else {
        const int w = role - 2;
        if (open_wait(A.sync + 0, A))
        store_scatter_share([&](long long j) { return __hip_atomic_load(A.slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
                            w / OPEN_PARTS, w % OPEN_PARTS, OPEN_PARTS, A.inc, A.s_obs, A.s_ag, A.s_g, A.s_act, A.obs, A.ag, A.g,
                            A.act, A.ep_obs, A.ep_ag, A.ep_g, A.ep_act);
        if (threadIdx.x == 0) open_leave(A.sync);
    }
}

// can the cycle open as one launch?  (every workgroup must be resident at once: the consumers poll)
bool cycle_open_fits(const hp_agent *a, int64_t n_new) {
    return 2 + OPEN_PARTS * n_new <= (int64_t)a->ctx->cu_count;
}

// The staged episodes of `b` (buffer_stage) -> slots, scatter, normalizer update, first `n_first` minibatch plans.
int cycle_open_launch(hp_agent *a, hp_buffer *b, hp_norm *on, hp_norm *gn, hp_rng *rng, PlanRec *norm_plan, int n_first,
                      double future_p, bool recompute) {
    OpenArgs A;
    A.st = rng->d_state;
    A.meta = b->d_meta;
    A.size = b->size;
    A.inc = b->staged_n;
    A.T = b->T;
    A.slots = b->st_slots.as<long long>();
    A.norm_plan = norm_plan;
    A.plan = a->plan.as<PlanRec>();
    A.batch = a->B;
    A.n_first = n_first;
    A.future_p = future_p;
    A.s_obs = b->st_obs.as<double>();
    A.s_ag = b->st_ag;
    A.s_g = b->st_g;
    A.s_act = b->st_act;
    A.obs = b->d_obs;
    A.ag = b->d_ag;
    A.g = b->d_g;
    A.act = b->d_act;
    A.ep_obs = b->ep_obs();
    A.ep_ag = b->ep_ag();
    A.ep_g = b->ep_g();
    A.ep_act = b->ep_act();
    A.onz = on->d;
    A.gnz = gn->d;
    A.obs_dim = b->obs_dim;
    A.goal_dim = b->goal_dim;
    A.clip_obs = a->cfg.clip_obs;
    A.recompute = recompute ? 1 : 0;
    A.o_eps_sq = on->eps * on->eps;
    A.g_eps_sq = gn->eps * gn->eps;
    A.o_std_f32 = on->std_f32;
    A.g_std_f32 = gn->std_f32;
    size_t norm_bytes = 0;
    A.chunk_rows = norm_plan_chunk(b->obs_dim, b->goal_dim, b->T, &norm_bytes);
    A.sync = a->open_sync;
    A.dev = a->d_state;
    A.fault = a->k1_sync + SPLIT_FAULT;
    A.fault_host = a->fault_host_dev;
    const size_t mt_bytes = 4 * MT_N * sizeof(uint32_t) + MT_IBUF * sizeof(int);
    const size_t lds = norm_bytes > mt_bytes ? norm_bytes : mt_bytes;
    HP_KLOG("k_cycle_open");
    hipLaunchKernelGGL(k_cycle_open, dim3((unsigned)(2 + OPEN_PARTS * b->staged_n)), dim3(OPEN_THREADS), lds, a->ctx->stream, A);
    HP_CHECK_HIP(hipGetLastError());
    return buffer_launch_pack(b, b->staged_n);   // throughput rows of the episodes just scattered (nothing unless the buffer has them)
}
