// slab8_split_args.h -- declarations the split launch (slab8_split.h, compiled once per slab height inside its namespace) shares with
// its host side (agent_engines.hip): roles, the argument block, the time-line build's stamp arrays.  Included once, outside the
// per-height namespaces, by slab8.h.
#pragma once
#include "peer.h"

#ifdef SLAB_TIMELINE   // time-line builds: every workgroup of the last split launch stamps {start, hand-off point, -, end}
__device__ unsigned long long g_split_tl[1024][4];
__device__ int g_split_role[1024];
__device__ unsigned long long g_split_entry[1024];   // wall clock at the kernel's first instruction, before any kernel argument is read
#define SPLIT_STAMP(k) do { if (Q.tl_mark && threadIdx.x == 0 && blockIdx.x < 1024) g_split_tl[blockIdx.x][(k)] = wall_clock64(); } while (0)
#else
#define SPLIT_STAMP(k) do { } while (0)
#endif

enum { SR_A = 0, SR_C = 1, SR_T = 2, SR_PLAN = 3, SR_AHEAD = 4, SR_WARM = 5, SR_TILE = 6, SR_N = 7 };

struct FbSplitArgs {
    FbSlabArgs s;                    // what the chains and the spare workgroups of k_fb_slab8 take (s.n_plan / n_ahead / n_pref: totals)
    unsigned long long nrole[8];     // HOST side only (build_split_roles): byte r of nrole[x] = workgroups of role r on XCD x; the kernel takes the
                                     // transposed table as leading scalar arguments (slab8_split.h: SplitRoles)
    unsigned warm_side;              // 4 bits per XCD: what its warmers touch (s8_l2_warm_at)
    GatherSrc tgs;                   // T chains: replay buffer, normalizers and the plan of the NEXT update
    const float *qt_in;              // C chains: Q' of this update's minibatch, [Mp][16] column 0
    float *qt_out;                   // T chains: Q' of the next update's minibatch
    unsigned *sync;                  // this launch's counter set (split_ctr), sync_other: the set it clears for the next launch
    unsigned *sync_other;
    unsigned *fault;                 // the learner's sticky fault word ...
    unsigned *fault_host;            // ... and its pinned host mirror
    unsigned long long wait_ticks;   // bound of every poll (100 MHz)
    unsigned need_c;                 // C chains of this launch (0: no tiles)
    unsigned tile_stage;             // 4 bits per problem of `tiles`: the counter that says its operands are published
    int reset_sync;                  // prologue launch of a sequence (target chains only): clear the counters
    int tl_mark;                     // time-line builds: this launch records its per-workgroup stamps (the last one WITH target chains)
    GemmGroup tiles;                 // weight-gradient problems: the critic's four
    AdamFuse adam;                   // their optimizer epilogue
    // data-parallel ranks, tile-wise exchange inside this launch (k_fb_split8<SPLIT_TILES_PEER>; gemm_lds.h PEER): the rank's exchange
    // block as mapped in this process -- a DEVICE copy of hp_peer::dev (by value it would push this block past the 4 KB kernarg
    // segment) --, index of the update in its sequence (the exchange epoch), SUM / MEAN
    const PeerDev *peer;
    int peer_u, peer_mean;
};
static_assert(sizeof(FbSplitArgs) <= 4096, "kernel arguments of k_fb_split8 exceed the 4 KB kernarg segment");

