// agent_layers.hip -- layer-per-launch fallback engine: every dependency level of the update as one grouped GEMM launch
// (gemm_lds.h through launch_group), loss / head / optimizer as small kernels: ~20 launches per update.  Selected
// automatically for network shapes the slab engines are not specialised for (hidden != 256, inputs wider than 48 columns,
// more than 4 action components) and by RLARM_ENGINE=layers; held to the same oracle bar (test_other_env_shapes_track_oracle).
// Reference: models.py:11-44, ddpg_agent.py:250-277.
#include "agent.h"

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    // fixed-order tree: deterministic run to run
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return tot;
}

// ddpg_agent.py:255-267: targets, both losses and their first derivatives.  One workgroup.
__global__ __launch_bounds__(256) void k_loss(const float *__restrict__ QT, const float *__restrict__ QA,
                                              const float *__restrict__ QP, const float *__restrict__ R,
                                              const float *__restrict__ XP, int ldx, int act_off, int act_dim, int B,
                                              int Mp, float gamma, float clip_ret, float action_l2, float *dQA,
                                              float *dQP, float *loss_log, AgentDevState *st, const AdamCfg adam) {
    __shared__ float sh[4];
    float sc = 0.f, sq = 0.f, sl2 = 0.f;
    const float invB = 1.0f / (float)B;
    for (int i = threadIdx.x; i < Mp; i += 256) {
        if (i < B) {
            float y = R[i] + gamma * QT[i * 16];          // target_q = r + gamma * q_next
            y = fminf(fmaxf(y, -clip_ret), 0.f);          // clamp(-1/(1-gamma), 0)
            const float d = y - QA[i * 16];
            sc += d * d;
            dQA[i * 16] = -2.f * d * invB;                // d/dq mean((y-q)^2)
            sq += QP[i * 16];
            dQP[i * 16] = -invB;                          // d/dq (-mean(q))
            for (int j = 0; j < act_dim; ++j) {
                const float u = XP[i * ldx + act_off + j];
                sl2 += u * u;
            }
        } else {
            dQA[i * 16] = 0.f;
            dQP[i * 16] = 0.f;
        }
    }
    const float tc = block_sum_256(sc, sh);
    const float tq = block_sum_256(sq, sh);
    const float tl = block_sum_256(sl2, sh);
    if (threadIdx.x == 0) {
        const long long k = st->n_logged;
        const float critic_loss = tc * invB;
        const float actor_loss = -(tq * invB) + action_l2 * (tl / (float)(B * act_dim));
        loss_log[(k % LOSS_LOG) * 2 + 0] = actor_loss;
        loss_log[(k % LOSS_LOG) * 2 + 1] = critic_loss;
        st->n_logged = k + 1;
        st->step += 1;
        adam_prepare(st, adam);
    }
}

// actor head backward (autograd of ddpg_agent.py:265-267 w.r.t. the pre-tanh output):
//   grad_u = action_l2 * 2u/(B*act_dim) + dXP[:, action block];  grad_pi = grad_u / max_action;
//   grad_tanh = grad_pi * max_action;  dZ = grad_tanh * (1 - tanh^2)
__global__ void k_actor_head(const float *__restrict__ dXP, const float *__restrict__ XP, const float *__restrict__ TP,
                             int ldx, int act_off, int act_dim, int B, float action_l2, float max_action, float *dZ) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * act_dim) return;
    const int i = idx / act_dim, j = idx - i * act_dim;
    const float u = XP[i * ldx + act_off + j];
    const float th = TP[i * 16 + j];
    const float gu = action_l2 * (2.f * u / (float)(B * act_dim)) + dXP[i * ldx + act_off + j];
    const float gt = (gu / max_action) * max_action;
    dZ[i * 16 + j] = gt * (1.f - th * th);
}

// torch.optim.Adam (_single_tensor_adam, no weight decay / amsgrad) over the whole arena.
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g,
                                              float *__restrict__ m, float *__restrict__ v, int n, int n_actor,
                                              float w, float b2, float omb2, float epsf, const AgentDevState *st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float neg_step_size = (idx < n_actor) ? st->neg_step_actor : st->neg_step_critic;
    const float bc2_sqrt = st->bc2_sqrt;
    const float gi = g[idx];
    float mi = m[idx], vi = v[idx];
    mi = __fadd_rn(mi, __fmul_rn(w, __fsub_rn(gi, mi)));                 // exp_avg.lerp_(grad, 1 - beta1)
    vi = __fadd_rn(__fmul_rn(vi, b2), __fmul_rn(__fmul_rn(omb2, gi), gi));  // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float sq = __fsqrt_rn(vi);                     // correctly rounded float32 sqrt
    const float denom = __fadd_rn(__fdiv_rn(sq, bc2_sqrt), epsf);
    p[idx] = __fadd_rn(p[idx], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    m[idx] = mi;
    v[idx] = vi;
}

// ddpg_agent.py:220-222: target = (1 - polyak) * param + polyak * target
__global__ void k_polyak(float *__restrict__ tgt, const float *__restrict__ src, int n, float one_minus, float polyak) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    tgt[idx] = __fadd_rn(__fmul_rn(one_minus, src[idx]), __fmul_rn(polyak, tgt[idx]));
}

// layer-per-launch engine: forwards + losses + backwards of one update, inputs in XA/XP/XT/R (18 launches)
int enqueue_forward_backward_layers(hp_agent *a) {
    const int H = a->H, Mp = a->Mp, ldx = a->ldx;
    const NetLayout &la = a->la, &lc = a->lc;
    float *Pa = a->params, *Pc = a->params + la.total;
    float *Ta = a->targets, *Tc = a->targets + la.total;
    float *Ga = a->grads, *Gc = a->grads + la.total;
    const float maxa = (float)a->cfg.max_action;
    hipStream_t s = a->ctx->stream;
    {   // level 1-3: hidden layers of actor_target(x'), critic(x,a), actor(x)
        Launch L;
        add_fwd(L, a->XT, ldx, la.K1, Ta + la.w1, Ta + la.b1, a->AT.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XA, ldx, lc.K1, Pc + lc.w1, Pc + lc.b1, a->CA.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XP, ldx, la.K1, Pa + la.w1, Pa + la.b1, a->AP.h1, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->AT.h1, H, H, Ta + la.w2, Ta + la.b2, a->AT.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CA.h1, H, H, Pc + lc.w2, Pc + lc.b2, a->CA.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->AP.h1, H, H, Pa + la.w2, Pa + la.b2, a->AP.h2, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->AT.h2, H, H, Ta + la.w3, Ta + la.b3, a->AT.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CA.h2, H, H, Pc + lc.w3, Pc + lc.b3, a->CA.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->AP.h2, H, H, Pa + la.w3, Pa + la.b3, a->AP.h3, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 4: heads.  tanh outputs land in the action block of the critic inputs
        Launch L;
        add_fwd(L, a->AT.h3, H, H, Ta + la.w4, Ta + la.b4, a->XT + a->act_off, ldx, Mp, 16, EPI_BIAS_TANH);
        L.g.p[0].n_store = a->cfg.act_dim; L.g.p[0].C2 = a->TP + 16 * (size_t)Mp; L.g.p[0].ldc2 = 16; L.g.p[0].max_action = maxa;
        add_fwd(L, a->CA.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QA, 16, Mp, 16, EPI_BIAS);
        add_fwd(L, a->AP.h3, H, H, Pa + la.w4, Pa + la.b4, a->XP + a->act_off, ldx, Mp, 16, EPI_BIAS_TANH);
        L.g.p[2].n_store = a->cfg.act_dim; L.g.p[2].C2 = a->TP; L.g.p[2].ldc2 = 16; L.g.p[2].max_action = maxa;
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 5-8: critic_target(x', a') and critic(x, pi(x))
        Launch L;
        add_fwd(L, a->XT, ldx, lc.K1, Tc + lc.w1, Tc + lc.b1, a->CT.h1, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->XP, ldx, lc.K1, Pc + lc.w1, Pc + lc.b1, a->CP.h1, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h1, H, H, Tc + lc.w2, Tc + lc.b2, a->CT.h2, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CP.h1, H, H, Pc + lc.w2, Pc + lc.b2, a->CP.h2, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h2, H, H, Tc + lc.w3, Tc + lc.b3, a->CT.h3, H, Mp, H, EPI_BIAS_RELU);
        add_fwd(L, a->CP.h2, H, H, Pc + lc.w3, Pc + lc.b3, a->CP.h3, H, Mp, H, EPI_BIAS_RELU);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {
        Launch L;
        add_fwd(L, a->CT.h3, H, H, Tc + lc.w4, Tc + lc.b4, a->QT, 16, Mp, 16, EPI_BIAS);
        add_fwd(L, a->CP.h3, H, H, Pc + lc.w4, Pc + lc.b4, a->QP, 16, Mp, 16, EPI_BIAS);
        HP_TRY(launch_group(a, L, PROF_GEMM_FWD));
    }
    {   // level 9: losses and dL/dq
        ProfScope ps(a, PROF_LOSS);
        const double clip_ret = 1.0 / (1.0 - a->cfg.gamma);  // ddpg_agent.py:259
        hipLaunchKernelGGL(k_loss, dim3(1), dim3(256), 0, s, a->QT, a->QA, a->QP, a->R, a->XP, ldx, a->act_off,
                           (int)a->cfg.act_dim, a->B, Mp, (float)a->cfg.gamma, (float)clip_ret,
                           (float)a->cfg.action_l2, a->dQA, a->dQP, a->loss_log, a->d_state, adam_cfg(a));
        HP_CHECK_HIP(hipGetLastError());
    }
    {   // level 10-13: backward through the critic, for the critic loss (dX + dW) and for the actor loss (dX only)
        Launch L;
        add_dx(L, a->dQA, 16, 16, Pc + lc.w4, H, a->dA3, H, Mp, a->CA.h3, H);
        add_dw(L, a->dQA, 16, 16, a->CA.h3, H, H, Gc + lc.w4, Gc + lc.b4, Mp);
        add_dx(L, a->dQP, 16, 16, Pc + lc.w4, H, a->dP3, H, Mp, a->CP.h3, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dA3, H, H, Pc + lc.w3, H, a->dA2, H, Mp, a->CA.h2, H);
        add_dw(L, a->dA3, H, H, a->CA.h2, H, H, Gc + lc.w3, Gc + lc.b3, Mp);
        add_dx(L, a->dP3, H, H, Pc + lc.w3, H, a->dP2, H, Mp, a->CP.h2, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dA2, H, H, Pc + lc.w2, H, a->dA1, H, Mp, a->CA.h1, H);
        add_dw(L, a->dA2, H, H, a->CA.h1, H, H, Gc + lc.w2, Gc + lc.b2, Mp);
        add_dx(L, a->dP2, H, H, Pc + lc.w2, H, a->dP1, H, Mp, a->CP.h1, H);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dw(L, a->dA1, H, H, a->XA, ldx, lc.K1, Gc + lc.w1, Gc + lc.b1, Mp);
        add_dx(L, a->dP1, H, H, Pc + lc.w1, lc.K1, a->dXP, ldx, Mp, nullptr, 0);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {   // level 14: through tanh and the action penalty
        ProfScope ps(a, PROF_LOSS);
        const int n = a->B * a->cfg.act_dim;
        hipLaunchKernelGGL(k_actor_head, dim3((n + 255) / 256), dim3(256), 0, s, a->dXP, a->XP, a->TP, ldx, a->act_off,
                           (int)a->cfg.act_dim, a->B, (float)a->cfg.action_l2, maxa, a->dZ);
        HP_CHECK_HIP(hipGetLastError());
    }
    {   // level 15-18: actor backward
        Launch L;
        add_dx(L, a->dZ, 16, 16, Pa + la.w4, H, a->dK3, H, Mp, a->AP.h3, H);
        add_dw(L, a->dZ, 16, 16, a->AP.h3, H, H, Ga + la.w4, Ga + la.b4, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dK3, H, H, Pa + la.w3, H, a->dK2, H, Mp, a->AP.h2, H);
        add_dw(L, a->dK3, H, H, a->AP.h2, H, H, Ga + la.w3, Ga + la.b3, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dx(L, a->dK2, H, H, Pa + la.w2, H, a->dK1, H, Mp, a->AP.h1, H);
        add_dw(L, a->dK2, H, H, a->AP.h1, H, H, Ga + la.w2, Ga + la.b2, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    {
        Launch L;
        add_dw(L, a->dK1, H, H, a->XP, ldx, la.K1, Ga + la.w1, Ga + la.b1, Mp);
        HP_TRY(launch_group(a, L, PROF_GEMM_BWD));
    }
    return HP_OK;
}

int layers_enqueue_adam(hp_agent *a) {
    const int n = a->n_arena;
    hipLaunchKernelGGL(k_adam, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->params, a->grads, a->adam_m,
                       a->adam_v, n, a->la.total, (float)(1.0 - a->cfg.adam_beta1), (float)a->cfg.adam_beta2,
                       (float)(1.0 - a->cfg.adam_beta2), (float)a->cfg.adam_eps, a->d_state);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

int layers_enqueue_polyak(hp_agent *a) {
    const int n = a->n_arena;
    const double om = 1.0 - a->cfg.polyak;
    hipLaunchKernelGGL(k_polyak, dim3((n + 255) / 256), dim3(256), 0, a->ctx->stream, a->targets, a->params, n, (float)om,
                       (float)a->cfg.polyak);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}
