// slab_common.h -- what the row-slab engines (slab8.h: 4/8/16-row slabs on v_mfma_f32_4x4x1, slab32.h: 32-row slabs on
// v_mfma_f32_32x32x2) and their host side (agent_*.hip) share: the arena map that locates a canonical parameter in the
// fragment-ordered weight copies, the re-layout kernel, and the argument blocks of the chain kernels.
//
// Why row slabs (measured on MI355X): a dependent kernel boundary costs ~1.55 us for a trivial kernel and
// 4-6 us for any kernel that pulls its operands through cold caches, and the layer-per-launch engine needs 20 of them per
// update (~130 us for 4.6 us of FP32-MFMA math).  Every product of the forward passes and of the dX half of the backward
// passes is ROW-independent, so one workgroup carries a slab of batch rows through a whole chain of layers with the
// activations in LDS; only the weight gradients reduce over the batch (one grouped GEMM launch, gemm_lds.h / dw64.h).
// The first engine of this kind (16-row slabs on v_mfma_f32_16x16x4, two kernels per update, 87 us at batch 256) was
// superseded at every batch size by the thin-slab and 32-row engines and removed in round 3 (profiles/r02_large_batch_engines.txt
// holds its last numbers).
#pragma once
#include "mt19937_device.h"

struct ArenaMap {  // enough of the arena geometry to find (layer, n, k) of a flat index on the device
    NetLayout la, lc;
    int H;
    int mode;      // 1: thin-slab fragment order (slab8.h), 2: 32-row order (slab32.h)
};

enum { SE_BIAS_RELU = 0, SE_MASK = 1 };

// one 1-KiB slot of a per-wave LDS-DMA weight ring (global_load_lds_dwordx4 writes wave-uniform base + lane * 16)
typedef float4 RingSlot[64];

struct SlabNetPtrs {
    const float *wf;     // forward-fragment copy of the whole arena this net lives in
    const float *wd;     // dX-fragment copy (online nets only)
    const float *canon;  // canonical arena (biases, head rows)
};

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
