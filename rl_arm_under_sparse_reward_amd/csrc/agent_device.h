// agent_device.h -- device-side optimizer / loss-log helpers shared by the kernels of agent_engines.hip
// (gemm_lds.h and dw64.h epilogues, k_adam_frag*, k_peer_adam*).
#pragma once
#include "agent.h"

// canonical arena index -> (offset of the forward-fragment copy, offset of the dX-fragment copy); -1 for biases
__host__ __device__ __forceinline__ void frag8_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d);
__host__ __device__ __forceinline__ int frag8_fwd_index(int n, int k, int K);   // (slab8.h)
__host__ __device__ __forceinline__ int frag8_dx_index(int n, int k, int N);
__host__ __device__ __forceinline__ void frag32_offsets(const ArenaMap &am, int idx, int &off_f, int &off_d);

__host__ __device__ __forceinline__ void frag_offsets_any(const ArenaMap &am, int idx, int &off_f, int &off_d) {
    if (am.mode == 2) frag32_offsets(am, idx, off_f, off_d);
    else frag8_offsets(am, idx, off_f, off_d);
}

// Adam (torch.optim.Adam, _single_tensor_adam, no weight decay / amsgrad) on one arena element, plus the
// fragment-ordered copies of the slab engines.  Shared by k_adam_frag and the weight-gradient GEMM epilogue
// (single-rank runs fuse the optimizer into the GEMM; data-parallel runs all-reduce the gradients in between).
struct AdamFuse {
    const float *p;                   // parameters the step starts from (canonical arena)
    float *p_out;                     // ... and where the stepped parameters go (p itself)
    float *m, *v, *fragF, *fragD;     // fragF / fragD: fragment-ordered copies of p_out
    const float *grads_base;          // arena origin of the gradient buffer the GEMM writes
    AgentDevState *st;
    const float *scal;                // {-lr_actor / bc1, -lr_critic / bc1, sqrt(bc2)} of the step being applied: the three
                                      // scalars in *st, written by the chain kernel of the same update
    ArenaMap am;
    int n_actor;
    float w, b2, omb2, eps;
    const float *part;                // per-slab loss partials
    int nslab, B, act_dim;
    float action_l2;
    float *loss_log;
    int keep_grads;                   // also write the gradient out (the fused epilogue itself does not need it in memory)
    // soft update of the target networks folded into this step (last update of a cycle; nullptr otherwise):
    // tgt = (1 - polyak) * p_new + polyak * tgt (ddpg_agent.py:220-222), fragFT = forward-fragment copy of the targets
    float *tgt, *fragFT;
    float polyak, one_minus;
    int wt;                           // write-through stores for the stepped state (adam_apply4)
    // tiles that run INSIDE the chain launch (k_fb_split8, slab8_split.h): the step may only be stored once `gate_need` chains
    // of that launch have counted themselves past the parameters it overwrites (*gate, agent scope); a poll that gives up
    // after gate_ticks (100 MHz) skips the step, sets the sticky fault word and its pinned host mirror
    const unsigned *gate;             // counter 0, copy of XCD 0, of this launch's set; XCD x polls the copy x * SPLIT_CTR_STRIDE words on
    unsigned gate_need;
    unsigned gate_sel;                // 4 bits per problem of the group: the counter its step waits for (SPLIT_CTR_NONE: none)
    unsigned *fault, *fault_host;
    unsigned long long gate_ticks;
    unsigned *reset_sync;             // the launch BEHIND k_fb_split8: its workgroup 0 clears the counter set that launch counted in
    int tl_mark;                      // time-line builds: record this launch's gate stamps
};

// Everything the optimizer epilogue reads of its argument block, loaded NOW and held in scalar registers.  Inside k_fb_split8 the
// block sits at the end of a 2.5 KB kernel-argument segment and the compiler loads each field where it is first used: behind the
// gate that meant four dependent scalar-cache misses (~0.5 us each) between "gate passed" and the first store -- the epilogue
// took 2.4 us instead of the stand-alone launch's 1.3 (round 4 time line).  Called at tile entry, where the latency hides under
// the operand transfers.
template <class T> __device__ __forceinline__ T sgpr_pin(T v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ void sgpr_pin_layout(NetLayout &l) {
    l.K1 = sgpr_pin(l.K1); l.w1 = sgpr_pin(l.w1); l.b1 = sgpr_pin(l.b1); l.w2 = sgpr_pin(l.w2); l.b2 = sgpr_pin(l.b2);
    l.w3 = sgpr_pin(l.w3); l.b3 = sgpr_pin(l.b3); l.w4 = sgpr_pin(l.w4); l.b4 = sgpr_pin(l.b4); l.total = sgpr_pin(l.total);
}
__device__ __forceinline__ AdamFuse adam_pinned(const AdamFuse &F) {
    AdamFuse L = F;
    L.p = sgpr_pin(L.p); L.p_out = sgpr_pin(L.p_out); L.m = sgpr_pin(L.m); L.v = sgpr_pin(L.v);
    L.fragF = sgpr_pin(L.fragF); L.fragD = sgpr_pin(L.fragD); L.grads_base = sgpr_pin(L.grads_base); L.scal = sgpr_pin(L.scal);
    sgpr_pin_layout(L.am.la); sgpr_pin_layout(L.am.lc);
    L.am.H = sgpr_pin(L.am.H); L.am.mode = sgpr_pin(L.am.mode);
    L.n_actor = sgpr_pin(L.n_actor);
    L.w = sgpr_pin(L.w); L.b2 = sgpr_pin(L.b2); L.omb2 = sgpr_pin(L.omb2); L.eps = sgpr_pin(L.eps);
    L.keep_grads = sgpr_pin(L.keep_grads);
    L.tgt = sgpr_pin(L.tgt); L.fragFT = sgpr_pin(L.fragFT); L.polyak = sgpr_pin(L.polyak); L.one_minus = sgpr_pin(L.one_minus);
    L.wt = sgpr_pin(L.wt);
    L.gate = sgpr_pin(L.gate); L.gate_need = sgpr_pin(L.gate_need); L.gate_sel = sgpr_pin(L.gate_sel);
    L.fault = sgpr_pin(L.fault); L.fault_host = sgpr_pin(L.fault_host); L.gate_ticks = sgpr_pin(L.gate_ticks);
    return L;
}

__device__ __forceinline__ bool adam_gate_wait(const AdamFuse &F, int prob, int *flag) {
    if (!F.gate) return true;
    const unsigned which = (F.gate_sel >> (4 * prob)) & 15u;
    if (which == SPLIT_CTR_NONE) return true;
    return handoff_wait(F.gate + (which * 8 + (blockIdx.x & 7)) * SPLIT_CTR_STRIDE, F.gate_need, F.gate_ticks, F.fault, F.fault_host, 2u,
                        flag);
}


// optimizer state of ONE element fetched ahead (gemm_bias_tile: cold loads that otherwise sit behind the reduction)
struct AdamState1 {
    float p, m, v;
};

// AGENT: the step scalars were written (write-through) by a workgroup of the SAME launch: agent-scope loads
template <bool AGENT>
__device__ __forceinline__ float adam_scal(const AdamFuse &F, int i) {
    if constexpr (AGENT) return __hip_atomic_load(F.scal + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return F.scal[i];
}
template <bool AGENT = false>
__device__ __forceinline__ void adam_apply(const AdamFuse &F, int idx, float gi, const AdamState1 *pre = nullptr) {
    const float neg_step_size = adam_scal<AGENT>(F, idx < F.n_actor ? 0 : 1);
    const float bc2_sqrt = adam_scal<AGENT>(F, 2);
    float mi = pre ? pre->m : F.m[idx], vi = pre ? pre->v : F.v[idx];
    mi = __fadd_rn(mi, __fmul_rn(F.w, __fsub_rn(gi, mi)));                      // exp_avg.lerp_(grad, 1 - beta1)
    vi = __fadd_rn(__fmul_rn(vi, F.b2), __fmul_rn(__fmul_rn(F.omb2, gi), gi));  // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float sq = __fsqrt_rn(vi);                             // correctly rounded float32 sqrt
    const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), F.eps);
    const float pn = __fadd_rn(pre ? pre->p : F.p[idx], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    F.p_out[idx] = pn;
    F.m[idx] = mi;
    F.v[idx] = vi;
    int of, od;
    frag_offsets_any(F.am, idx, of, od);
    if (of >= 0) F.fragF[of] = pn;
    if (od >= 0) F.fragD[od] = pn;
    if (F.tgt) {   // same expression as k_polyak_frag
        const float t = __fadd_rn(__fmul_rn(F.one_minus, pn), __fmul_rn(F.polyak, F.tgt[idx]));
        F.tgt[idx] = t;
        if (of >= 0) F.fragFT[of] = t;
    }
}

// write-through (sc1) stores: visible to other XCDs once drained (s_waitcnt vmcnt(0)); readers use agent-scope loads
__device__ __forceinline__ void wt_store(float *p, float v) {   // write-through (sc1) store: visible to other XCDs once drained
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_store4(float *p, const float4 v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    // s_nop 1 INSIDE the string: the store reads its data registers a few cycles after issue and hipcc pads nothing behind an
    // asm statement -- without it the next instruction may overwrite x before the store has read it (cdna_hip_programming.md
    // 5.7 item 1; found when an experiment put this store in front of arithmetic that reused the registers: NaNs)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}

// four consecutive arena elements at once (idx0 a multiple of 4): one vector load per state array, so the cold-cache
// latency of p / m / v is paid once, not once per element (scalar version: the store to p[idx] may alias the next
// element's load, which serialises them)
struct AdamState4 {   // optimizer state of 4 consecutive elements + the step scalars, fetched ahead of the gradient
    float4 p, m, v;
    float neg_step_size, bc2_sqrt;
};
template <bool AGENT = false>
__device__ __forceinline__ void adam_fetch4(AdamState4 &S, const AdamFuse &F, int idx0) {
    S.neg_step_size = adam_scal<AGENT>(F, idx0 < F.n_actor ? 0 : 1);
    S.bc2_sqrt = adam_scal<AGENT>(F, 2);
    S.p = *reinterpret_cast<const float4 *>(F.p + idx0);
    S.m = *reinterpret_cast<const float4 *>(F.m + idx0);
    S.v = *reinterpret_cast<const float4 *>(F.v + idx0);
}
#ifdef ADAM_TL   // (agent_engines.hip, time-line build): stamps of workgroup 0's optimizer step in g_gemm_tl[8..11] (timeline slots 168-171)
extern __device__ unsigned long long g_gemm_tl[32];
#define ADAM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_gemm_tl[8 + (k)] = wall_clock64(); } while (0)
#else
#define ADAM_STAMP(k) do { } while (0)
#endif
#define ADAM_FRAG_LOOKUP (-2)   // of_known: find the fragment offsets from the arena index (frag8_offsets: a walk over the layout + a division)
// quad8 (gemm_tile's epilogue): the lanes l ^ 8, l ^ 16, l ^ 24 of this lane's 32 hold rows (n & 3) ^ 1, ^ 2, ^ 3 of the same 4-row
// group at the same 4 reduction indices, and all four call with od_known >= 0 -- the dX copy then goes out as ONE float4 per lane
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4], const AdamState4 &S,
                                            int of_known = ADAM_FRAG_LOOKUP, int od_known = -1, bool quad8 = false);

// 4 x 4 transposition across the lanes l, l ^ 8, l ^ 16, l ^ 24 (i = this lane's row of the four = (l >> 3) & 3): lane i ends up with
// {a_0[i], a_1[i], a_2[i], a_3[i]}.  ds_swizzle in bit-mask mode (xor within 32 lanes): LDS crossbar, no memory, no address register.
__device__ __forceinline__ float lane_pick4(const float (&a)[4], int idx) {
    const float lo = (idx & 1) ? a[1] : a[0], hi = (idx & 1) ? a[3] : a[2];
    return (idx & 2) ? hi : lo;
}
__device__ __forceinline__ float4 quad8_transpose4(const float (&a)[4], int i) {
    float r[4];   // r[x]: element i of the lane whose row is i ^ x
    r[0] = lane_pick4(a, i);
    r[1] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(lane_pick4(a, i ^ 1)), 0x201F));   // xor 8
    r[2] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(lane_pick4(a, i ^ 2)), 0x401F));   // xor 16
    r[3] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(lane_pick4(a, i ^ 3)), 0x601F));   // xor 24
    return make_float4(lane_pick4(r, i), lane_pick4(r, i ^ 1), lane_pick4(r, i ^ 2), lane_pick4(r, i ^ 3));   // out[j] = r[i ^ j]
}
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4]) {
    AdamState4 S;
    adam_fetch4(S, F, idx0);
    adam_apply4(F, idx0, g, S);
}
__device__ __forceinline__ void adam_apply4(const AdamFuse &F, int idx0, const float (&g)[4], const AdamState4 &S, int of_known,
                                            int od_known, bool quad8) {
    const float neg_step_size = S.neg_step_size;
    const float bc2_sqrt = S.bc2_sqrt;
    ADAM_STAMP(0);
    const float4 p4 = S.p, m4 = S.m, v4 = S.v;
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mm[j] = __fadd_rn(mm[j], __fmul_rn(F.w, __fsub_rn(g[j], mm[j])));
        vv[j] = __fadd_rn(__fmul_rn(vv[j], F.b2), __fmul_rn(__fmul_rn(F.omb2, g[j]), g[j]));
        const float sq = __fsqrt_rn(vv[j]);
        const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), F.eps);
        pp[j] = __fadd_rn(pp[j], __fdiv_rn(__fmul_rn(neg_step_size, mm[j]), denom));
    }
    ADAM_STAMP(1);
    // F.wt (small minibatches): write-through stores.  The optimizer leaves 5.6 MB dirty in the L2s, which the end of the kernel
    // has to write back before the next launch may start; written through while other workgroups still multiply, that tail is
    // gone: 40.7 -> 40.1 us/update at batch 256.  At batch 1024 / 4096 the launch is long enough to hide the write-back itself
    // and the stores only add traffic under load (55.6 -> 55.8, 127.5 -> 129.0): plain stores there.  (System scope, sc0 sc1,
    // measured 44.5 vs 41.8 in round 2: agent scope is what the other XCDs need.)
    if (F.wt) {
        wt_store4(F.p_out + idx0, make_float4(pp[0], pp[1], pp[2], pp[3]));
        wt_store4(F.m + idx0, make_float4(mm[0], mm[1], mm[2], mm[3]));
        wt_store4(F.v + idx0, make_float4(vv[0], vv[1], vv[2], vv[3]));
    } else {
        *reinterpret_cast<float4 *>(F.p_out + idx0) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        *reinterpret_cast<float4 *>(F.m + idx0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4 *>(F.v + idx0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    ADAM_STAMP(2);
    float tt[4] = {0.f, 0.f, 0.f, 0.f};
    if (F.tgt) {   // same expression as k_polyak_frag, on the parameters just stepped
        const float4 t4 = *reinterpret_cast<const float4 *>(F.tgt + idx0);
        const float told[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) tt[j] = __fadd_rn(__fmul_rn(F.one_minus, pp[j]), __fmul_rn(F.polyak, told[j]));
        *reinterpret_cast<float4 *>(F.tgt + idx0) = make_float4(tt[0], tt[1], tt[2], tt[3]);
    }
    if (F.am.mode == 1 || F.am.mode == 2) {
        // slab8 / slab32 fragment orders: 4 consecutive reduction indices of one output row (idx0 % 4 == 0, every tensor's
        // row length is a multiple of 4) are ONE float4 of the forward copy and 4 dwords 16 B apart in the dX copy
        int of, od;
        if (F.am.mode == 1 && of_known != ADAM_FRAG_LOOKUP) { of = of_known; od = od_known; }   // (the caller knows its tensor: gemm_tile)
        else if (F.am.mode == 1) frag8_offsets(F.am, idx0, of, od);
        else frag32_offsets(F.am, idx0, of, od);
        if (of >= 0) {
            if (F.wt) wt_store4(F.fragF + of, make_float4(pp[0], pp[1], pp[2], pp[3]));
            else *reinterpret_cast<float4 *>(F.fragF + of) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        }
        if (of >= 0 && F.tgt) *reinterpret_cast<float4 *>(F.fragFT + of) = make_float4(tt[0], tt[1], tt[2], tt[3]);
        if (quad8) {
            // rows n .. n + 3 at reduction index k are ONE float4 of the dX copy (frag8_dx_index): this lane (row i of its group) takes
            // k = k0 + i of the four lanes' 4 x 4 block -- 16 dword stores at 16-byte stride become four 16-byte stores, 64 B in a row
            const int i = (int)(threadIdx.x >> 3) & 3;
            const float4 t = quad8_transpose4(pp, i);
            if (F.wt) wt_store4(F.fragD + od + 3 * i, t);
            else *reinterpret_cast<float4 *>(F.fragD + od + 3 * i) = t;
        } else if (od >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F.wt) wt_store(F.fragD + od + 4 * j, pp[j]);
                else F.fragD[od + 4 * j] = pp[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int of, od;
            frag_offsets_any(F.am, idx0 + j, of, od);
            if (of >= 0) F.fragF[of] = pp[j];
            if (od >= 0) F.fragD[od] = pp[j];
            if (of >= 0 && F.tgt) F.fragFT[of] = tt[j];
        }
    }
    ADAM_STAMP(3);
}

// ---- the stand-alone optimizer kernels (k_adam_frag4, k_peer_adam, k_peer_adam2: every transport that exchanges the whole
// gradient vector) ------------------------------------------------------------------------------------------------------------------
// Thread t of those kernels used to step arena floats 4 t .. 4 t + 3: 64 consecutive threads = 256 consecutive reduction indices of
// ONE row.  For the 256-wide layers that made the forward-fragment store a 16-byte piece every 1 KiB per lane and the dX-fragment
// store four 4-byte pieces at 16-byte stride: 1.2 M partial L2 writes per step (k_adam_frag4: 8.9 us for 10.5 MB of traffic).
// adam_quad_remap deals the float4s of such a layer in blocks of 4 rows x 64 columns instead -- lane = (row & 3) + 4 * (column
// group) -- so the four lanes of a quad hold the same four columns of four consecutive rows: their forward-fragment float4s are
// 64 contiguous bytes, and a 4 x 4 transpose inside the quad (DPP quad_perm, no LDS) turns the dX-fragment pieces into one float4
// per lane, 64 contiguous bytes per quad again.  Same elements, same arithmetic: only who stores what changes.
// returns the arena index this thread steps; quad: it sits in a dealt block (quad lanes = rows n .. n + 3 of the same columns)
__device__ __forceinline__ int adam_quad_remap(const ArenaMap &am, int t, bool &quad) {
    const int idx = 4 * t;
    quad = false;
    if (am.mode != 1) return idx;
    const bool critic = idx >= am.la.total;
    const NetLayout &l = critic ? am.lc : am.la;
    const int base = critic ? am.la.total : 0, r = idx - base, K = am.H;
    int w0 = -1;
    if (r >= l.w2 && r < l.b2) w0 = l.w2;
    else if (r >= l.w3 && r < l.b3) w0 = l.w3;
    else if (r >= l.w4 && r < l.b4) w0 = l.w4;
    if (w0 < 0 || (K & 63) || ((base + w0) & 15)) return idx;      // (16 floats = the 4 threads of a quad start together)
    const int local = (r - w0) >> 2, blk = local >> 6, w = local & 63, kblocks = K >> 6;
    const int row = 4 * (blk / kblocks) + (w & 3), k = 64 * (blk % kblocks) + 4 * (w >> 2);
    quad = true;
    return base + w0 + row * K + k;
}
// lane (quad position nn) gives v[0..3]; returns in out[c] what quad lane c held at index nn: the 4 x 4 transpose of the quad
__device__ __forceinline__ void quad_transpose4(const float (&v)[4], float (&out)[4], int nn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // round r: lane s sends v[(s + r) & 3] to lane (s + r) & 3, which files it under out[s]
        const float send = ((nn + r) & 3) == 0 ? v[0] : ((nn + r) & 3) == 1 ? v[1] : ((nn + r) & 3) == 2 ? v[2] : v[3];
        float got;
        const int bits = __float_as_int(send);
        // quad_perm: destination lane d reads source lane (d - r) & 3
        if (r == 0) got = send;
        else if (r == 1) got = __int_as_float(__builtin_amdgcn_mov_dpp(bits, 0x93 /* [3,0,1,2] */, 0xf, 0xf, true));
        else if (r == 2) got = __int_as_float(__builtin_amdgcn_mov_dpp(bits, 0x4e /* [2,3,0,1] */, 0xf, 0xf, true));
        else got = __int_as_float(__builtin_amdgcn_mov_dpp(bits, 0x39 /* [1,2,3,0] */, 0xf, 0xf, true));
        const int src = (nn - r) & 3;
        out[0] = src == 0 ? got : out[0];
        out[1] = src == 1 ? got : out[1];
        out[2] = src == 2 ? got : out[2];
        out[3] = src == 3 ? got : out[3];
    }
}
// torch.optim.Adam on 4 consecutive arena elements of a dealt block + their fragment copies (+ the folded soft update): the
// arithmetic of adam_apply4, the stores as described above.  Call from wave-uniform control flow over whole quads.
__device__ __forceinline__ void adam_step4_quad(const AdamFuse &F, int idx0, const float (&g)[4]) {
    AdamState4 S;
    adam_fetch4(S, F, idx0);
    const float4 p4 = S.p, m4 = S.m, v4 = S.v;
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mm[j] = __fadd_rn(mm[j], __fmul_rn(F.w, __fsub_rn(g[j], mm[j])));
        vv[j] = __fadd_rn(__fmul_rn(vv[j], F.b2), __fmul_rn(__fmul_rn(F.omb2, g[j]), g[j]));
        const float sq = __fsqrt_rn(vv[j]);
        const float denom = __fadd_rn(__fdiv_rn(sq, S.bc2_sqrt), F.eps);
        pp[j] = __fadd_rn(pp[j], __fdiv_rn(__fmul_rn(S.neg_step_size, mm[j]), denom));
    }
    *reinterpret_cast<float4 *>(F.p_out + idx0) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4 *>(F.m + idx0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4 *>(F.v + idx0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    float tt[4] = {0.f, 0.f, 0.f, 0.f};
    if (F.tgt) {   // same expression as k_polyak_frag, on the parameters just stepped
        const float4 t4 = *reinterpret_cast<const float4 *>(F.tgt + idx0);
        const float told[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) tt[j] = __fadd_rn(__fmul_rn(F.one_minus, pp[j]), __fmul_rn(F.polyak, told[j]));
        *reinterpret_cast<float4 *>(F.tgt + idx0) = make_float4(tt[0], tt[1], tt[2], tt[3]);
    }
    int of, od;
    frag8_offsets(F.am, idx0, of, od);
    if (of >= 0) {
        *reinterpret_cast<float4 *>(F.fragF + of) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (F.tgt) *reinterpret_cast<float4 *>(F.fragFT + of) = make_float4(tt[0], tt[1], tt[2], tt[3]);
    }
    if (od >= 0) {
        // od = this row's slot (n & 3 = nn) of column k; the quad's transposed float4 for column k + nn starts 3 nn floats on
        const int nn = (int)(threadIdx.x & 3);
        float col[4] = {0.f, 0.f, 0.f, 0.f};
        quad_transpose4(pp, col, nn);
        *reinterpret_cast<float4 *>(F.fragD + od + 3 * nn) = make_float4(col[0], col[1], col[2], col[3]);
    }
}

// optimizer kernels that follow a split launch whose tiles left the step to them (SPLIT_TILES_GRADS): block 0, threads 64 .. 127 clear
// the counter set that launch counted in, as the actor's tile launch does in the fused forms (gemm_lds.h)
__device__ __forceinline__ void split_reset_by_block0(const AdamFuse &F) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && F.reset_sync && tid >= 64 && tid < 64 + SPLIT_COUNTERS * 8)
        __hip_atomic_store(F.reset_sync + (tid - 64) * SPLIT_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// loss means from the per-slab partial sums: one wavefront, fixed reduction tree (deterministic)
__device__ __forceinline__ void loss_finalize(const AdamFuse &F) {
    const int lane = threadIdx.x;
    float tc = 0.f, tq = 0.f, tl = 0.f;
    for (int s = lane; s < F.nslab; s += 64) {   // agent-scope loads of write-through stores of the chain kernel
        tc += __hip_atomic_load(F.part + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tq += __hip_atomic_load(F.part + F.nslab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tl += __hip_atomic_load(F.part + 2 * F.nslab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int o = 32; o > 0; o >>= 1) {
        tc += __shfl_down(tc, o);
        tq += __shfl_down(tq, o);
        tl += __shfl_down(tl, o);
    }
    if (lane == 0) {
        const float invB = 1.0f / (float)F.B;
        const long long k = F.st->n_logged;
        F.loss_log[(k % LOSS_LOG) * 2 + 0] = -(tq * invB) + F.action_l2 * (tl / (float)(F.B * F.act_dim));
        F.loss_log[(k % LOSS_LOG) * 2 + 1] = tc * invB;
        F.st->n_logged = k + 1;
    }
}
