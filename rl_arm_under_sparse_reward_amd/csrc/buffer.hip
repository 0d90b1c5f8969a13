// buffer.hip -- episodic replay storage in HBM + the HER gather/relabel/reward kernel
// (reference: replay_buffer.py:11-71, her.py:26-39, bmirobot_env_push_F.py:20-23,84-90).
//
// HBM layout (float64, identical to the reference's numpy arrays so that episodes can be
// memcpy'd in and read back verbatim):
//   obs  [size][T+1][obs_dim]   ag [size][T+1][goal_dim]
//   g    [size][T][goal_dim]    actions [size][T][act_dim]
// A sampled transition touches obs[e][t..t+1] (one contiguous 2*obs_dim run), ag[e][t..t+1],
// g[e][t], actions[e][t] and, when relabelled, ag[e][future_t].
//
// Built with -ffp-contract=off: the reward must round exactly like numpy's
// (x0*x0 + x1*x1) + x2*x2 in float64, so no multiply-add may be fused.
#include "store_device.h"

#include <cmath>

// ----------------------------------------------------------------------------- kernels
// One wavefront per transition; lanes stride over the row elements (coalesced 8-byte loads).
// dict-mode output = the arrays her_sampler.sample_her_transitions returns (her.py:39).
__global__ __launch_bounds__(256) void k_gather_dict(const double *__restrict__ obs, const double *__restrict__ ag,
                                                     const double *__restrict__ g, const double *__restrict__ act,
                                                     const PlanRec *__restrict__ plan, long long batch, int T,
                                                     int obs_dim, int goal_dim, int act_dim, double sq_threshold,
                                                     double *o_obs, double *o_ag, double *o_g, double *o_act,
                                                     double *o_obs_next, double *o_ag_next, float *o_r,
                                                     double *o_r64) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= batch) return;
    const PlanRec rec = plan[i];
    const long long e = rec.e;
    const int t = rec.t;
    const double *obs_row = obs + (e * (T + 1) + t) * obs_dim;        // rows t and t+1 are adjacent
    const double *ag_row = ag + (e * (T + 1) + t) * goal_dim;
    const double *g_src = rec.her ? ag + (e * (T + 1) + rec.fut) * goal_dim  // her.py:35-36
                                  : g + (e * T + t) * goal_dim;
    const double *act_row = act + (e * T + t) * act_dim;
    for (int c = lane; c < obs_dim; c += 64) {
        o_obs[i * obs_dim + c] = obs_row[c];
        o_obs_next[i * obs_dim + c] = obs_row[obs_dim + c];
    }
    for (int c = lane; c < goal_dim; c += 64) {
        o_ag[i * goal_dim + c] = ag_row[c];
        o_ag_next[i * goal_dim + c] = ag_row[goal_dim + c];
        o_g[i * goal_dim + c] = g_src[c];
    }
    for (int c = lane; c < act_dim; c += 64) o_act[i * act_dim + c] = act_row[c];
    if (lane == 0) {
        // goal_distance: sqrt(sum((ag_next - g)^2)) > thr  <=>  sum >= sq_threshold  (sqrt is monotone;
        // sq_threshold is the smallest double whose correctly rounded sqrt exceeds thr).  numpy's
        // add.reduce over a contiguous axis of < 8 elements is a plain left-to-right sum.
        double s = 0.0;
        for (int c = 0; c < goal_dim; ++c) {
            double d = __dsub_rn(ag_row[goal_dim + c], g_src[c]);
            double sq = __dmul_rn(d, d);
            s = (c == 0) ? sq : __dadd_rn(s, sq);
        }
        o_r[i] = hp_reward(s, sq_threshold);  // -(d > thr).astype(float32), or float32(-d)
        // what compute_reward itself returns (:87-90): the float32 above widened (sparse) or -d in float64 (dense)
        if (o_r64) o_r64[i] = hp_reward64(s, sq_threshold);
    }
}

// The learner's view of a minibatch (ddpg_agent.py:227-243) straight out of the replay buffer: gather (her.py:26), relabel
// (:35-36), reward (:38), _preproc_og clip (:214-217), both normalizers (normalizer.py:67-70), concatenate, float32 -- the
// arithmetic of the chain kernels' in-launch gather (slab8.h s8_gather), as a kernel of its own with DEVICE outputs:
//   x [B][od+gd] = norm(clip(obs[e][t])) | norm(clip(g'))      x_next = norm(clip(obs[e][t+1])) | norm(clip(g'))
//   a [B][ad]    = float32(actions[e][t])  (the critic divides by max_action itself, models.py:38)     r [B]
// One wavefront per transition, FS_FLIGHT transitions in flight per wavefront.  A transition's source elements are
// [obs t | obs t+1 | ag t+1 | g' | action] = 2 od + 2 gd + ad doubles -- exactly 64 for the bmirobot shapes (27, 3, 4), i.e.
// ONE 8-byte load per lane, of which the first 54 lanes read one contiguous 432-byte run.  HBM-bound: 67 doubles read,
// 65 floats written per transition.
struct FusedSampleArgs {
    const double *obs, *ag, *g, *act;
    const PlanRec *plan;
    const NormDev *onz, *gnz;
    long long batch;
    int T, od, gd, ad;
    double sq_threshold, clip_obs, clip_o, clip_g;
    float *x, *xn, *a, *r;
    long long *o_e, *o_t, *o_fut;
    unsigned char *o_her;
    // fast draw (hp_buffer_sample_dev_fast; SURVEY 8b rng_mode = Philox): no plan, the index record of transition m of call
    // `fast_call` is Philox4x32-10((m, call); seed) -- fs_fast_rec below
    unsigned long long fast_seed, fast_call;
    int fast_n_eps;
    double fast_future_p;
};

// ---- fast draw: counter-based indices instead of the reference's sequential MT19937 stream (opt-in, NOT the reference's draws) ------
// her.py:24-33 draws e = randint(N), t = randint(T), u1, u2 from ONE global stream, which makes the index draw of a large batch a
// sequential kernel (2.0 ms per 2^18 transitions against 59 us for their gather).  Fast mode keys every transition by its own counter:
//   (r0, r1, r2, r3) = Philox4x32-10(counter = (m lo, m hi, call lo, call hi), key = (seed lo, seed hi))      [Random123, Salmon et al. SC11]
//   e = floor(r0 N / 2^32), t = floor(r1 T / 2^32), her = r2 2^-32 < future_p, future_t = t + 1 + floor(r3 (T - t) / 2^32)
// -- her.py's four draws with multiply-shift bounded integers (bias <= N / 2^32) -- so the draw costs a few dozen integer
// instructions inside the gather kernel.  Deterministic in (seed, call, m); pinned by Random123's known-answer vectors and an
// numpy twin in the test infrastructure (draw_her_indices_fast).
__device__ __forceinline__ void fs_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                                 unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1;
        c3 = (unsigned)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ PlanRec fs_fast_rec(const FusedSampleArgs &A, long long m) {
    unsigned r[4];
    fs_philox4x32_10((unsigned)m, (unsigned)((unsigned long long)m >> 32), (unsigned)A.fast_call, (unsigned)(A.fast_call >> 32),
                     (unsigned)A.fast_seed, (unsigned)(A.fast_seed >> 32), r);
    PlanRec rec;
    rec.e = (int)(((unsigned long long)r[0] * (unsigned)A.fast_n_eps) >> 32);
    rec.t = (int)(((unsigned long long)r[1] * (unsigned)A.T) >> 32);
    rec.her = ((double)r[2] * 0x1p-32 < A.fast_future_p) ? 1 : 0;
    rec.fut = rec.t + 1 + (int)(((unsigned long long)r[3] * (unsigned)(A.T - rec.t)) >> 32);
    return rec;
}
// the records of the 2 x FLIGHT transitions of one pass: lane j draws transition base + j, every lane picks up its half's
template <int FLIGHT>
__device__ __forceinline__ void fs_fast_recs(const FusedSampleArgs &A, long long base, int lane, int h, PlanRec (&rec)[FLIGHT]) {
    const long long mj = base + (lane & (2 * FLIGHT - 1));
    const PlanRec mine = fs_fast_rec(A, mj < A.batch ? mj : A.batch - 1);
#pragma unroll
    for (int k = 0; k < FLIGHT; ++k) {
        const int src = 2 * k + h;
        rec[k].e = __shfl(mine.e, src);
        rec[k].t = __shfl(mine.t, src);
        rec[k].fut = __shfl(mine.fut, src);
        rec[k].her = __shfl(mine.her, src);
    }
}

// FLIGHT transitions per wavefront and pass: 1 for small batches (one minibatch = two dependent memory latencies, as many
// wavefronts as transitions), 4 from 16 Ki transitions on (more bytes in flight per wavefront, grid-stride).  Every load of a
// pass -- the lanes' source elements AND the reward's operands -- is issued before the first value is used.
template <int FLIGHT>
__global__ __launch_bounds__(256) void k_gather_fused(const FusedSampleArgs A) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int od = A.od, gd = A.gd, ad = A.ad, ldx = od + gd, Q = 2 * od + 2 * gd + ad;
    // what this lane's source element is in the first strip of 64 (the only one for the bmirobot shapes): 0 obs t, 1 obs t+1,
    // 2 ag t+1 (reward only), 3 g', 4 action, 5 nothing
    auto kind_of = [&](int q) { return q < od ? 0 : q < 2 * od ? 1 : q < 2 * od + gd ? 2 : q < 2 * od + 2 * gd ? 3 : q < Q ? 4 : 5; };
    auto col_of = [&](int q, int kind) { return kind == 0 ? q : kind == 1 ? q - od : kind == 2 ? q - 2 * od : kind == 3 ? q - 2 * od - gd : q - 2 * od - 2 * gd; };
    const int rc = lane < gd ? lane : gd - 1;      // reward operands: lanes < gd, one goal component each
    for (long long base = wave * FLIGHT; base < A.batch; base += n_waves * FLIGHT) {
        PlanRec rec[FLIGHT];
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) rec[k] = A.plan[base + k < A.batch ? base + k : A.batch - 1];
        const double *obs0[FLIGHT], *g_src[FLIGHT];
        double ra[FLIGHT], rg[FLIGHT];
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {
            const long long e = rec[k].e;
            const int t = rec[k].t;
            obs0[k] = A.obs + (e * (A.T + 1) + t) * od;
            g_src[k] = rec[k].her ? A.ag + (e * (A.T + 1) + rec[k].fut) * gd : A.g + (e * A.T + t) * gd;   // her.py:35-36
            ra[k] = A.ag[(e * (A.T + 1) + t + 1) * gd + rc];
            rg[k] = g_src[k][rc];
        }
        for (int q0 = 0; q0 < Q; q0 += 64) {   // one pass for the bmirobot shapes
            const int q = q0 + lane, kind = kind_of(q), j = col_of(q, kind);
            float mu = 0.f;
            double sd = 1.0, clip = 0.0;
            if (kind <= 1) { mu = A.onz->mean[j]; sd = A.onz->std[j]; clip = A.clip_o; }
            if (kind == 3) { mu = A.gnz->mean[j]; sd = A.gnz->std[j]; clip = A.clip_g; }
            double v[FLIGHT];
#pragma unroll
            for (int k = 0; k < FLIGHT; ++k) {
                const long long e = rec[k].e;
                const int t = rec[k].t;
                // address-selected, unconditional load (idle lanes re-read element 0 of the observation row)
                const double *p = kind <= 1 ? obs0[k] + q : kind == 2 ? A.ag + (e * (A.T + 1) + t + 1) * gd + j
                                : kind == 3 ? g_src[k] + j : kind == 4 ? A.act + (e * A.T + t) * ad + j : obs0[k];
                v[k] = *p;
            }
#pragma unroll
            for (int k = 0; k < FLIGHT; ++k) {
                const long long m = base + k;
                if (m >= A.batch) continue;
                if (kind <= 1 || kind == 3) {
                    double c = fmin(fmax(v[k], -A.clip_obs), A.clip_obs);                 // _preproc_og
                    c = __ddiv_rn(__dsub_rn(c, (double)mu), sd);                          // normalizer.normalize
                    const float x = (float)fmin(fmax(c, -clip), clip);
                    if (kind == 0) { if (A.x) A.x[m * ldx + j] = x; }
                    else if (kind == 1) { if (A.xn) A.xn[m * ldx + j] = x; }
                    else {
                        if (A.x) A.x[m * ldx + od + j] = x;
                        if (A.xn) A.xn[m * ldx + od + j] = x;
                    }
                } else if (kind == 4) {
                    if (A.a) A.a[m * ad + j] = (float)v[k];
                }
            }
        }
        // reward: lanes < gd hold (ag_next - g')^2 of their component, lane 0 adds them in index order (numpy's add.reduce
        // over < 8 contiguous elements is a left-to-right sum)
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {
            const long long m = base + k;
            const double d = __dsub_rn(ra[k], rg[k]);
            const double sq = __dmul_rn(d, d);
            double s = 0.0;
            for (int c = 0; c < gd; ++c) {
                const double sc = __shfl(sq, c);
                s = (c == 0) ? sc : __dadd_rn(s, sc);
            }
            if (m < A.batch && lane == 0) {
                if (A.r) A.r[m] = hp_reward(s, A.sq_threshold);
                if (A.o_e) A.o_e[m] = rec[k].e;
                if (A.o_t) A.o_t[m] = rec[k].t;
                if (A.o_fut) A.o_fut[m] = rec[k].fut;
                if (A.o_her) A.o_her[m] = (unsigned char)rec[k].her;
            }
        }
    }
}

// The same kernel with 16-byte loads and TWO transitions per wavefront instruction (round 6).  A transition's sources are three
// vectors of doubles -- the contiguous [obs t | obs t+1] run (2 od), g' (gd), the action (ad) -- cut into units of two doubles: od +
// ceil(gd / 2) + ceil(ad / 2) units = 31 for the bmirobot shapes (27, 3, 4), so 32 lanes carry a transition with ONE 16-byte load
// each (rows are 8-byte aligned; the last unit of an odd-length vector loads [len - 2, len - 1] and keeps its second double, so no
// load reads past a row) and a wavefront instruction carries two.  Against k_gather_fused: half the load instructions per
// transition for the same bytes, 2 x FLIGHT transitions in flight per wavefront instead of FLIGHT, the normalizers' mean / std
// fetched once per lane instead of once per pass, g' of the reward taken from the units already loaded (one extra 8-byte load per
// transition pair for ag[t + 1] instead of two).  Same arithmetic per element: identical bits.  Shapes that do not fit 32 lanes
// take k_gather_fused.
// (us per 2^18 / 2^20 transitions, float64 rows | float32 rows, branch-free kernels: FLIGHT 2: 56.1 / 234.6 | 62.3 / 242.9; 4: 59.2 / 233.6 |
// 58.2 / 224.0; 8: 77.1 / 291.4 | 74.8 / 283.8)
#ifndef FS2_FLIGHT
#define FS2_FLIGHT 4
#endif
typedef double fs_d2 __attribute__((ext_vector_type(2), aligned(8)));
template <int FLIGHT, bool FAST = false>
__global__ __launch_bounds__(256) void k_gather_fused2(const FusedSampleArgs A) {
    const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int od = A.od, gd = A.gd, ad = A.ad, ldx = od + gd;
    const int ug = (gd + 1) >> 1, ua = (ad + 1) >> 1;
    // this lane's unit: 0 = two doubles of the [obs t | obs t+1] run, 1 = of g', 2 = of the action, 3 = none
    const int unit = l < od ? 0 : l < od + ug ? 1 : l < od + ug + ua ? 2 : 3;
    const int j = unit == 0 ? l : unit == 1 ? l - od : l - od - ug;
    const int len = unit == 0 ? 2 * od : unit == 1 ? gd : ad;
    const bool tail = unit != 3 && 2 * j + 1 >= len;      // last unit of an odd-length vector
    const int e0 = unit == 3 ? 0 : (tail ? len - 2 : 2 * j);
    // per slot (the unit's two doubles): where it goes -- a primary output row (x, x_next or actions; nullptr: nowhere) and, for g',
    // x_next as well -- and the constants of ONE branch-free expression for every lane: normalised elements clip at clip_obs, subtract
    // their mean, divide by their std, clip at the normaliser's range; an action passes through it unchanged ((v - 0) / 1 between
    // infinite clips is v, exactly).  (The first version branched per destination: 142 branches in the kernel, as many scalar
    // instructions as vector ones.)
    float *p0[2], *p1[2];
    int stride0[2], off[2];
    float mu[2] = {0.f, 0.f};
    double sd[2] = {1.0, 1.0}, clip[2] = {INFINITY, INFINITY}, cobs[2] = {INFINITY, INFINITY};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int el = e0 + s;
        const bool ok = unit != 3 && !(tail && s == 0);
        p0[s] = p1[s] = nullptr;
        stride0[s] = ldx;
        off[s] = 0;
        if (ok && unit == 0) {
            const int col = el < od ? el : el - od;
            p0[s] = el < od ? A.x : A.xn;
            off[s] = col;
            mu[s] = A.onz->mean[col]; sd[s] = A.onz->std[col]; clip[s] = A.clip_o; cobs[s] = A.clip_obs;
        } else if (ok && unit == 1) {
            p0[s] = A.x;
            p1[s] = A.xn;               // g_next := g (ddpg_agent.py:231)
            off[s] = od + el;
            mu[s] = A.gnz->mean[el]; sd[s] = A.gnz->std[el]; clip[s] = A.clip_g; cobs[s] = A.clip_obs;
        } else if (ok) {
            p0[s] = A.a;
            stride0[s] = ad;
            off[s] = el;
        }
    }
    // reward operands: lanes l < gd of each half hold component l.  g'[c] sits in lane od + (c / 2) -- the last unit for the last
    // component of an odd gd -- slot c % 2 (slot 1 there)
    const int rc = l < gd ? l : gd - 1;
    const bool rc_tail = (gd & 1) && rc == gd - 1;
    const int g_lane = (h << 5) + od + (rc_tail ? ug - 1 : (rc >> 1)), g_slot = rc_tail ? 1 : (rc & 1);
    for (long long base = wave * (2 * FLIGHT); base < A.batch; base += n_waves * (2 * FLIGHT)) {
        PlanRec rec[FLIGHT];
        if constexpr (FAST) {
            fs_fast_recs<FLIGHT>(A, base, lane, h, rec);
        } else {
#pragma unroll
            for (int k = 0; k < FLIGHT; ++k) {
                const long long m = base + 2 * k + h;
                rec[k] = A.plan[m < A.batch ? m : A.batch - 1];
            }
        }
        fs_d2 v[FLIGHT];
        double ra[FLIGHT];
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {      // every load of the pass before the first use
            const long long e = rec[k].e;
            const int t = rec[k].t;
            const double *obs0 = A.obs + (e * (A.T + 1) + t) * od;
            const double *g_src = rec[k].her ? A.ag + (e * (A.T + 1) + rec[k].fut) * gd : A.g + (e * A.T + t) * gd;   // her.py:35-36
            const double *p = unit == 0 ? obs0 + e0 : unit == 1 ? g_src + e0 : unit == 2 ? A.act + (e * A.T + t) * ad + e0 : obs0;
            v[k] = *reinterpret_cast<const fs_d2 *>(p);
            ra[k] = A.ag[(e * (A.T + 1) + t + 1) * gd + rc];
        }
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {
            const long long m = base + 2 * k + h;
            const bool live = m < A.batch;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const double val = s ? v[k].y : v[k].x;
                double c = fmin(fmax(val, -cobs[s]), cobs[s]);                           // _preproc_og
                c = __ddiv_rn(__dsub_rn(c, (double)mu[s]), sd[s]);                       // normalizer.normalize
                const float x = (float)fmin(fmax(c, -clip[s]), clip[s]);
                if (live && p0[s]) p0[s][m * stride0[s] + off[s]] = x;
                if (live && p1[s]) p1[s][m * ldx + off[s]] = x;
            }
            // reward (her.py:38): (ag_next - g')^2 per component in lanes l < gd, lane 0 of the half adds them left to right
            const double gx = __shfl(v[k].x, g_lane), gy = __shfl(v[k].y, g_lane);
            const double d = __dsub_rn(ra[k], g_slot ? gy : gx);
            const double sq = __dmul_rn(d, d);
            double sum = 0.0;
            for (int c = 0; c < gd; ++c) {
                const double sc = __shfl(sq, (h << 5) + c);
                sum = (c == 0) ? sc : __dadd_rn(sum, sc);
            }
            if (live && l == 0) {
                if (A.r) A.r[m] = hp_reward(sum, A.sq_threshold);
                if (A.o_e) A.o_e[m] = rec[k].e;
                if (A.o_t) A.o_t[m] = rec[k].t;
                if (A.o_fut) A.o_fut[m] = rec[k].fut;
                if (A.o_her) A.o_her[m] = (unsigned char)rec[k].her;
            }
        }
    }
}

// ---- throughput rows (SURVEY 8b storage_dtype = fp32; hp_buffer_enable_f32_rows / hp_buffer_sample_dev_f32) -----------------------
// The float64 arrays above are the reference's layout and stay the source of truth.  The mirror below is laid out for the
// gather instead: one (episode, timestep) = ONE 128-byte line of float32 [obs_t | action_t | 0], so a transition's
// observations and action are two adjacent, line-aligned lines (the float64 rows are a 432-byte run at any 8-byte offset:
// 4.4 lines, plus a line for the action); the goals stay float64 -- the reward and the relabelled goal are bit-exact as before --
// as [ag_t | g_t | 0] per 64 bytes, so ag[t + 1] and g[t] share 128 bytes.  PMC, 2^18 transitions: 947 B fetched per transition
// from the float64 rows (7.4 lines), 5xx from these.  Observations and actions are rounded to float32 at store time (actions
// lose nothing the learner sees: it consumes float32(actions)); the normaliser arithmetic on them is unchanged (float64).
__global__ __launch_bounds__(256) void k_pack_rows(const long long *__restrict__ slots, long long n, long long first,
                                                   const double *__restrict__ obs, const double *__restrict__ ag,
                                                   const double *__restrict__ g, const double *__restrict__ act, float *p_row,
                                                   double *p_goal, int T, int od, int gd, int ad, int row_w, int goal_w) {
    // one workgroup per (episode, block of 8 timesteps); slots == nullptr: episodes first .. first + n - 1
    const long long i = blockIdx.x / ((T + 8) / 8);
    const int t0 = (int)(blockIdx.x % ((T + 8) / 8)) * 8;
    const long long e = slots ? slots[i] : first + i;
    for (int idx = threadIdx.x; idx < 8 * row_w; idx += blockDim.x) {
        const int t = t0 + idx / row_w, c = idx % row_w;
        if (t > T) continue;
        float v = 0.f;
        if (c < od) v = (float)obs[(e * (T + 1) + t) * od + c];
        else if (c < od + ad && t < T) v = (float)act[(e * T + t) * ad + (c - od)];
        p_row[(e * (T + 1) + t) * row_w + c] = v;
    }
    for (int idx = threadIdx.x; idx < 8 * goal_w; idx += blockDim.x) {
        const int t = t0 + idx / goal_w, c = idx % goal_w;
        if (t > T) continue;
        double v = 0.0;
        if (c < gd) v = ag[(e * (T + 1) + t) * gd + c];
        else if (c < 2 * gd && t < T) v = g[(e * T + t) * gd + (c - gd)];
        p_goal[(e * (T + 1) + t) * goal_w + c] = v;
    }
}

// hp_buffer_sample_dev on the throughput rows.  32 lanes per transition, two transitions per wavefront instruction, ONE 16-byte
// load per lane: lanes 0-14 the float4s of the two adjacent row lines (row t: obs | action, row t + 1: obs), lanes 15 and 31 the
// two halves of g' (= ag[future_t] when relabelled, her.py:35-36, else the g half of goal row t), lanes 16 and 17 ag[t + 1] for the
// reward.  Each row lane then hands components 2, 3 of its float4 to the lane 16 above it, so that all 32 lanes carry two
// elements through ONE pass of the float64 clip / normalise arithmetic (the kernel is bound by that arithmetic as much as by
// bytes: a first version that kept four components per lane ran 104 us per 2^18 transitions against 68 for the float64 rows).
// Needs obs_dim <= 28, obs_dim + act_dim <= 32, goal_dim <= 4 (hp_buffer_enable_f32_rows checks).
struct PackedSampleArgs {
    const float *p_row;
    const double *p_goal;
    int row_w, goal_w;
    FusedSampleArgs f;
};
template <int FLIGHT, bool FAST = false>
__global__ __launch_bounds__(256) void k_gather_packed(const PackedSampleArgs P) {
    const FusedSampleArgs &A = P.f;
    const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int od = A.od, gd = A.gd, ad = A.ad, ldx = od + gd, rw = P.row_w, gw = P.goal_w;
    // what this lane LOADS: 0 a row float4 (lanes 0-14: column 4 l of the 64-float [row t | row t + 1] pair), 1 a g' half (lane
    // 15: doubles 0, 1; lane 31: doubles 2, 3), 2 an ag[t + 1] half (lanes 16, 17), 3 nothing
    const int load = l < 15 ? 0 : (l == 15 || l == 31) ? 1 : (l == 16 || l == 17) ? 2 : 3;
    const int gj = l == 31 ? 1 : 0;                        // which half of g'
    // what this lane COMPUTES, two slots: lanes 0-14 columns 4 l + s, lanes 16-30 columns 4 (l - 16) + 2 + s (handed down by lane
    // l - 16), lanes 15 / 31 the goal components 2 gj + s.  Per slot: a primary output row (x, x_next or actions; nullptr: nowhere),
    // x_next as well for a goal component, and the constants of one branch-free clip / normalise expression (an action passes
    // through it unchanged: (v - 0) / 1 between infinite clips), as in k_gather_fused2
    float *p0[2], *p1[2];
    int stride0[2], off[2];
    bool goal[2] = {false, false};
    float mu[2] = {0.f, 0.f};
    double sd[2] = {1.0, 1.0}, clip[2] = {INFINITY, INFINITY}, cobs[2] = {INFINITY, INFINITY};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        p0[s] = p1[s] = nullptr;
        stride0[s] = ldx;
        off[s] = 0;
        if (load == 1) {
            const int c = 2 * gj + s;
            if (c < gd) {
                goal[s] = true;
                p0[s] = A.x; p1[s] = A.xn;                               // g_next := g (ddpg_agent.py:231)
                off[s] = od + c; mu[s] = A.gnz->mean[c]; sd[s] = A.gnz->std[c]; clip[s] = A.clip_g; cobs[s] = A.clip_obs;
            }
        } else if (l != 15 && l != 31) {
            const int col = 4 * (l & 15) + 2 * (l >> 4) + s;          // column of the 64-float pair
            const int oc = col < rw ? col : col - rw;                 // ... inside its row
            if (oc < od) {
                p0[s] = col < rw ? A.x : A.xn;
                off[s] = oc; mu[s] = A.onz->mean[oc]; sd[s] = A.onz->std[oc]; clip[s] = A.clip_o; cobs[s] = A.clip_obs;
            } else if (col < rw && oc < od + ad) {
                p0[s] = A.a;
                stride0[s] = ad;
                off[s] = oc - od;
            }
        }
    }
    const int rc = l < gd ? l : gd - 1;                   // reward: lanes l < gd of each half hold component l
    const int a_lane = (h << 5) + 16 + (rc >> 1), g_lane = (h << 5) + (rc < 2 ? 15 : 31), r_slot = rc & 1;
    const int from = (h << 5) + (l & 15);                 // the row lane a lane 16-30 takes its components from
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef float f4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    union Ld { f4_t f; d2_t d; };
    for (long long base = wave * (2 * FLIGHT); base < A.batch; base += n_waves * (2 * FLIGHT)) {
        PlanRec rec[FLIGHT];
        if constexpr (FAST) {
            fs_fast_recs<FLIGHT>(A, base, lane, h, rec);
        } else {
#pragma unroll
            for (int k = 0; k < FLIGHT; ++k) {
                const long long m = base + 2 * k + h;
                rec[k] = A.plan[m < A.batch ? m : A.batch - 1];
            }
        }
        Ld v[FLIGHT];
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {      // every load of the pass before the first use
            const long long e = rec[k].e;
            const int t = rec[k].t;
            const float *row = P.p_row + (e * (A.T + 1) + t) * rw;
            const double *goal_t = P.p_goal + (e * (A.T + 1) + t) * gw;
            // g': the ag half of goal row future_t, or the g half (doubles gd ..) of goal row t
            const double *gsrc = rec[k].her ? P.p_goal + (e * (A.T + 1) + rec[k].fut) * gw : goal_t + gd;
            const void *p = load == 0 ? (const void *)(row + 4 * l) : load == 1 ? (const void *)(gsrc + 2 * gj)
                          : load == 2 ? (const void *)(goal_t + gw + 2 * (l - 16)) : (const void *)row;
            v[k].f = *reinterpret_cast<const f4_a8 *>(p);
        }
#pragma unroll
        for (int k = 0; k < FLIGHT; ++k) {
            const long long m = base + 2 * k + h;
            const bool live = m < A.batch;
            // components 2, 3 of row lane l go to lane l + 16
            const float z = __shfl(v[k].f[2], from), w = __shfl(v[k].f[3], from);
            const float lo = l < 16 ? v[k].f[0] : z, hi = l < 16 ? v[k].f[1] : w;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float raw = s ? hi : lo;
                double x = goal[s] ? v[k].d[s] : (double)raw;
                x = fmin(fmax(x, -cobs[s]), cobs[s]);                                  // _preproc_og
                x = __ddiv_rn(__dsub_rn(x, (double)mu[s]), sd[s]);                      // normalizer.normalize
                const float o = (float)fmin(fmax(x, -clip[s]), clip[s]);
                if (live && p0[s]) p0[s][m * stride0[s] + off[s]] = o;
                if (live && p1[s]) p1[s][m * ldx + off[s]] = o;
            }
            // reward (her.py:38) on the float64 goals: the same left-to-right sum as every other gather
            const double ax = __shfl(v[k].d[0], a_lane), ay = __shfl(v[k].d[1], a_lane);
            const double gx = __shfl(v[k].d[0], g_lane), gy = __shfl(v[k].d[1], g_lane);
            const double d = __dsub_rn(r_slot ? ay : ax, r_slot ? gy : gx);
            const double sq = __dmul_rn(d, d);
            double sum = 0.0;
            for (int c = 0; c < gd; ++c) {
                const double sc = __shfl(sq, (h << 5) + c);
                sum = (c == 0) ? sc : __dadd_rn(sum, sc);
            }
            if (live && l == 0) {
                if (A.r) A.r[m] = hp_reward(sum, A.sq_threshold);
                if (A.o_e) A.o_e[m] = rec[k].e;
                if (A.o_t) A.o_t[m] = rec[k].t;
                if (A.o_fut) A.o_fut[m] = rec[k].fut;
                if (A.o_her) A.o_her[m] = (unsigned char)rec[k].her;
            }
        }
    }
}

// Batched compute_reward / _is_success of the bmirobot GoalEnvs (bmirobot_env_push_F.py:84-90, :243-245; identical in
// bmirobot_env_pickandplace_v2.py) on device arrays [n][goal_dim] float64.  mode 0: sparse reward -(d > thr) as float32
// (bits 0x80000000 / 0xBF800000); mode 1: dense reward -d as float64; mode 2: success (d < thr) as float32.
// thr_sq = the squared-domain threshold of the predicate (see hp_compute_reward); no square root in modes 0 and 2.
__global__ __launch_bounds__(256) void k_goal_reward(const double *__restrict__ ag, const double *__restrict__ g,
                                                     long long n, int goal_dim, double thr_sq, int mode, float *out32,
                                                     double *out64) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int c = 0; c < goal_dim; ++c) {   // numpy add.reduce over < 8 contiguous elements: left to right
        const double d = __dsub_rn(ag[i * goal_dim + c], g[i * goal_dim + c]);
        const double sq = __dmul_rn(d, d);
        s = (c == 0) ? sq : __dadd_rn(s, sq);
    }
    if (mode == 0) out32[i] = (s >= thr_sq) ? -1.0f : -0.0f;
    else if (mode == 1) out64[i] = -__dsqrt_rn(s);
    else out32[i] = (s < thr_sq) ? 1.0f : 0.0f;
}

// scatter the staged episodes into their slots.  numpy's `buffers[idxs] = mb` lets the LAST
// occurrence of a repeated slot win, so an episode is dropped when a later one has its slot.
#define STORE_PARTS 8
__global__ __launch_bounds__(256) void k_store_scatter(const long long *__restrict__ slots, long long n_new,
                                                       const double *s_obs, const double *s_ag, const double *s_g,
                                                       const double *s_act, double *obs, double *ag, double *g,
                                                       double *act, long long ep_obs, long long ep_ag,
                                                       long long ep_g, long long ep_act) {
    // STORE_PARTS workgroups per episode, each a strided share of its values: one or two copies per thread instead of a
    // loop of fifteen dependent-latency iterations (the copy of 2 episodes was 9.6 us of every cycle)
    store_scatter_share([&](long long j) { return slots[j]; }, blockIdx.x / STORE_PARTS, blockIdx.x % STORE_PARTS, STORE_PARTS,
                        n_new, s_obs, s_ag, s_g, s_act, obs, ag, g, act, ep_obs, ep_ag, ep_g, ep_act);
}

// ------------------------------------------------------------------------------ launchers
int buffer_launch_gather_dict(hp_buffer *b, const PlanRec *d_plan, int64_t batch, double sq_threshold, double *d_out,
                              float *d_r, double *d_r64) {
    const int od = b->obs_dim, gd = b->goal_dim, ad = b->act_dim;
    double *o_obs = d_out;
    double *o_ag = o_obs + batch * od;
    double *o_g = o_ag + batch * gd;
    double *o_act = o_g + batch * gd;
    double *o_obs_next = o_act + batch * ad;
    double *o_ag_next = o_obs_next + batch * od;
    const int waves_per_block = 4;
    dim3 grid((unsigned)((batch + waves_per_block - 1) / waves_per_block));
    hipLaunchKernelGGL(k_gather_dict, grid, dim3(256), 0, b->ctx->stream, b->d_obs, b->d_ag, b->d_g, b->d_act, d_plan,
                       (long long)batch, (int)b->T, od, gd, ad, sq_threshold, o_obs, o_ag, o_g, o_act, o_obs_next,
                       o_ag_next, d_r, d_r64);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// throughput rows of the episodes in st_slots[0 .. n_new) (just scattered) -- or, slots == nullptr, of episodes [first, first + n)
static int launch_pack(hp_buffer *b, const long long *slots, int64_t first, int64_t n) {
    if (!b->p_row || n <= 0) return HP_OK;
    const unsigned per_ep = (unsigned)((b->T + 8) / 8);
    HP_KLOG("k_pack_rows");
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)n * per_ep), dim3(256), 0, b->ctx->stream, slots, (long long)n, (long long)first,
                       b->d_obs, b->d_ag, b->d_g, b->d_act, b->p_row, b->p_goal, (int)b->T, (int)b->obs_dim, (int)b->goal_dim,
                       (int)b->act_dim, (int)b->row_w, (int)b->goal_w);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}
int buffer_launch_pack(hp_buffer *b, int64_t n_new) { return launch_pack(b, b->st_slots.as<long long>(), 0, n_new); }

// stage host episodes on the device (st_*): one pinned copy, one H2D.  copy-in semantics (replay_buffer.py:39-42): the
// caller's arrays are read by the CPU memcpy below and never again; the DMA reads our pinned staging.
static int buffer_stage(hp_buffer *b, const double *obs, const double *ag, const double *g, const double *actions,
                        int64_t n_new) {
    hipStream_t s = b->ctx->stream;
    const size_t n0 = n_new * b->ep_obs() * 8, n1 = n_new * b->ep_ag() * 8, n2 = n_new * b->ep_g() * 8,
                 n3 = n_new * b->ep_act() * 8;
    HP_TRY(b->st_obs.ensure(n0 + n1 + n2 + n3));
    char *dst = b->st_obs.as<char>();
    b->st_ag = reinterpret_cast<double *>(dst + n0);
    b->st_g = reinterpret_cast<double *>(dst + n0 + n1);
    b->st_act = reinterpret_cast<double *>(dst + n0 + n1 + n2);
    HP_TRY(b->pin.ensure(n0 + n1 + n2 + n3));
    char *h = static_cast<char *>(b->pin.p);
    memcpy(h, obs, n0);
    memcpy(h + n0, ag, n1);
    memcpy(h + n0 + n1, g, n2);
    memcpy(h + n0 + n1 + n2, actions, n3);
    HP_CHECK_HIP(hipMemcpyAsync(dst, h, n0 + n1 + n2 + n3, hipMemcpyHostToDevice, s));
    HP_TRY(b->pin.mark(s));
    b->staged_n = n_new;
    return HP_OK;
}

// stage, pick slots, scatter.  Shared by hp_buffer_store and the train-cycle path.
int buffer_stage_and_store(hp_buffer *b, hp_rng *rng, const double *obs, const double *ag, const double *g,
                           const double *actions, int64_t n_new) {
    hipStream_t s = b->ctx->stream;
    HP_TRY(b->st_slots.ensure(n_new * 8));
    HP_TRY(buffer_stage(b, obs, ag, g, actions, n_new));
    HP_TRY(rng_launch_slots(rng, b, n_new, b->st_slots.as<int64_t>()));
    hipLaunchKernelGGL(k_store_scatter, dim3((unsigned)(n_new * STORE_PARTS)), dim3(256), 0, s, b->st_slots.as<long long>(),
                       (long long)n_new, b->st_obs.as<double>(), b->st_ag, b->st_g,
                       b->st_act, b->d_obs, b->d_ag, b->d_g, b->d_act, (long long)b->ep_obs(),
                       (long long)b->ep_ag(), (long long)b->ep_g(), (long long)b->ep_act());
    HP_CHECK_HIP(hipGetLastError());
    HP_TRY(buffer_launch_pack(b, n_new));
    // host mirror of replay_buffer.py:68 and :43
    b->current_size = (b->current_size + n_new < b->size) ? b->current_size + n_new : b->size;
    b->n_transitions_stored += (int64_t)b->T * n_new;
    return HP_OK;
}

// Episodes in a host block registered with the device (hp_host_register: the feeder's shared-memory ring), laid out
// obs | ag | g | actions back to back: asynchronous DMA into the staging, a ticket behind it, and -- store_now -- slots + scatter
// (hp_buffer_store_pinned); without store_now those follow in the cycle's opening launch (hp_agent_train_cycle_pinned).
int buffer_stage_pinned(hp_buffer *b, hp_rng *rng, const double *block, int64_t n_new, uint64_t *ticket, bool store_now) {
    hipStream_t s = b->ctx->stream;
    const size_t n0 = n_new * b->ep_obs() * 8, n1 = n_new * b->ep_ag() * 8, n2 = n_new * b->ep_g() * 8,
                 n3 = n_new * b->ep_act() * 8;
    HP_TRY(b->st_slots.ensure(n_new * 8));
    HP_TRY(b->st_obs.ensure(n0 + n1 + n2 + n3));
    char *dst = b->st_obs.as<char>();
    b->st_ag = reinterpret_cast<double *>(dst + n0);
    b->st_g = reinterpret_cast<double *>(dst + n0 + n1);
    b->st_act = reinterpret_cast<double *>(dst + n0 + n1 + n2);
    HP_CHECK_HIP(hipMemcpyAsync(dst, block, n0 + n1 + n2 + n3, hipMemcpyHostToDevice, s));
    // ticket: an event behind the copy, from a small ring (a ticket older than the ring is done by construction: its
    // event is re-recorded only after this call synchronised on it)
    const uint64_t t = ++b->pin_tickets;
    hipEvent_t &ev = b->pin_events[t % hp_buffer::PIN_RING];
    if (!ev) HP_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else HP_CHECK_HIP(hipEventSynchronize(ev));
    HP_CHECK_HIP(hipEventRecord(ev, s));
    if (ticket) *ticket = t;
    b->staged_n = n_new;
    if (store_now) {
        HP_TRY(rng_launch_slots(rng, b, n_new, b->st_slots.as<int64_t>()));
        hipLaunchKernelGGL(k_store_scatter, dim3((unsigned)(n_new * STORE_PARTS)), dim3(256), 0, s, b->st_slots.as<long long>(),
                           (long long)n_new, b->st_obs.as<double>(), b->st_ag, b->st_g, b->st_act, b->d_obs, b->d_ag, b->d_g,
                           b->d_act, (long long)b->ep_obs(), (long long)b->ep_ag(), (long long)b->ep_g(), (long long)b->ep_act());
        HP_CHECK_HIP(hipGetLastError());
        HP_TRY(buffer_launch_pack(b, n_new));
    }
    b->current_size = (b->current_size + n_new < b->size) ? b->current_size + n_new : b->size;
    b->n_transitions_stored += (int64_t)b->T * n_new;
    return HP_OK;
}

int buffer_stage_for_cycle(hp_buffer *b, const double *obs, const double *ag, const double *g, const double *actions,
                           int64_t n_new) {
    HP_TRY(b->st_slots.ensure(n_new * 8));
    HP_TRY(buffer_stage(b, obs, ag, g, actions, n_new));
    b->current_size = (b->current_size + n_new < b->size) ? b->current_size + n_new : b->size;
    b->n_transitions_stored += (int64_t)b->T * n_new;
    return HP_OK;
}

// --------------------------------------------------------------------------------- C ABI
extern "C" {

int hp_buffer_create(hp_ctx *ctx, int64_t size_episodes, int32_t T, int32_t obs_dim, int32_t goal_dim,
                     int32_t act_dim, hp_buffer **out) {
    HP_REQUIRE(ctx && out, HP_ERR_INVALID, "hp_buffer_create: null argument");
    HP_REQUIRE(size_episodes > 0 && size_episodes < (1ll << 31), HP_ERR_INVALID,
               "hp_buffer_create: size_episodes=%lld must be in [1, 2^31)", (long long)size_episodes);
    HP_REQUIRE(T > 0 && obs_dim > 0 && goal_dim > 0 && act_dim > 0, HP_ERR_INVALID,
               "hp_buffer_create: dimensions must be positive");
    hp_buffer *b = new hp_buffer();
    b->ctx = ctx;
    b->size = size_episodes;
    b->T = T;
    b->obs_dim = obs_dim;
    b->goal_dim = goal_dim;
    b->act_dim = act_dim;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_obs, size_episodes * b->ep_obs() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_ag, size_episodes * b->ep_ag() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_g, size_episodes * b->ep_g() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_act, size_episodes * b->ep_act() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_meta, sizeof(BufMeta));
    if (e == hipSuccess) e = hipMemsetAsync(b->d_meta, 0, sizeof(BufMeta), ctx->stream);
    if (e != hipSuccess) {
        hp_set_error("hp_buffer_create: device allocation failed: %s", hipGetErrorString(e));
        hp_buffer_destroy(b);
        return HP_ERR_HIP;
    }
    *out = b;
    return HP_OK;
}

int hp_buffer_store(hp_buffer *b, hp_rng *rng, const double *obs, const double *ag, const double *g,
                    const double *actions, int64_t n_new) {
    HP_REQUIRE(b && rng && obs && ag && g && actions, HP_ERR_INVALID, "hp_buffer_store: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(n_new > 0, HP_ERR_INVALID, "hp_buffer_store: n_new must be positive");
    // replay_buffer.py:64 with current_size == 0 and inc > size: np.random.randint(0, 0, k) raises
    HP_REQUIRE(!(b->current_size == 0 && n_new > b->size), HP_ERR_INVALID, "high <= 0");
    return buffer_stage_and_store(b, rng, obs, ag, g, actions, n_new);
}

// ---- multi-process feeder support: episodes arrive in a host block the caller registered with the device
// (shared memory that worker processes write), laid out obs | ag | g | actions back to back like the library's own
// staging; the H2D copy reads it directly and asynchronously, the block may be rewritten once the returned ticket is done.
int hp_host_register(hp_ctx *ctx, void *host, size_t bytes) {
    HP_REQUIRE(ctx && host && bytes > 0, HP_ERR_INVALID, "hp_host_register: bad argument");
    CtxGuard guard(ctx);
    HP_CHECK_HIP(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return HP_OK;
}

int hp_host_unregister(hp_ctx *ctx, void *host) {
    HP_REQUIRE(ctx && host, HP_ERR_INVALID, "hp_host_unregister: bad argument");
    CtxGuard guard(ctx);
    HP_CHECK_HIP(hipHostUnregister(host));
    return HP_OK;
}

int hp_buffer_store_pinned(hp_buffer *b, hp_rng *rng, const double *block, int64_t n_new, uint64_t *ticket) {
    HP_REQUIRE(b && rng && block, HP_ERR_INVALID, "hp_buffer_store_pinned: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(n_new > 0, HP_ERR_INVALID, "hp_buffer_store_pinned: n_new must be positive");
    HP_REQUIRE(!(b->current_size == 0 && n_new > b->size), HP_ERR_INVALID, "high <= 0");
    return buffer_stage_pinned(b, rng, block, n_new, ticket, true);
}

int hp_buffer_store_done(hp_buffer *b, uint64_t ticket, int32_t wait, int32_t *done) {
    HP_REQUIRE(b && done, HP_ERR_INVALID, "hp_buffer_store_done: null argument");
    hipEvent_t ev = nullptr;
    {
        HP_SERIALISE(b);
        HP_REQUIRE(ticket >= 1 && ticket <= b->pin_tickets, HP_ERR_INVALID, "hp_buffer_store_done: unknown ticket");
        if (ticket + hp_buffer::PIN_RING <= b->pin_tickets) {   // its event slot has been reused: that store is long done
            *done = 1;
            return HP_OK;
        }
        ev = b->pin_events[ticket % hp_buffer::PIN_RING];
    }
    // outside the context lock: a feeder waiting for its block must not stall the trainer's enqueues.  A concurrent
    // hp_buffer_store_pinned may re-record this very event object for ticket + PIN_RING in the meantime (it synchronises on
    // it first, so OUR copy is done by then): the slot is re-checked afterwards and a recycled slot reads as done, whatever
    // the wait / query observed of the newer copy
    hipError_t e = hipSuccess;
    if (wait) HP_CHECK_HIP(hipEventSynchronize(ev));
    else e = hipEventQuery(ev);
    if (e != hipSuccess && e != hipErrorNotReady) HP_CHECK_HIP(e);
    (void)hipGetLastError();
    *done = (e == hipSuccess) ? 1 : 0;
    if (!*done) {
        HP_SERIALISE(b);
        if (ticket + hp_buffer::PIN_RING <= b->pin_tickets) *done = 1;
    }
    return HP_OK;
}

int hp_buffer_stage(hp_buffer *b, const double *obs, const double *ag, const double *g, const double *actions,
                    int64_t n_new) {
    HP_REQUIRE(b && obs && ag && g && actions, HP_ERR_INVALID, "hp_buffer_stage: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(n_new > 0, HP_ERR_INVALID, "high <= 0");   // sample_her_transitions on an empty dict: randint(0, 0, T)
    return buffer_stage(b, obs, ag, g, actions, n_new);
}

int hp_buffer_info(hp_buffer *b, int64_t *size, int64_t *current_size, int64_t *n_transitions_stored, int32_t *T) {
    HP_REQUIRE(b, HP_ERR_INVALID, "hp_buffer_info: null handle");
    HP_SERIALISE(b);
    if (size) *size = b->size;
    if (current_size) *current_size = b->current_size;
    if (n_transitions_stored) *n_transitions_stored = b->n_transitions_stored;
    if (T) *T = b->T;
    return HP_OK;
}

int hp_buffer_last_slots(hp_buffer *b, int64_t *host_out, int64_t n) {
    HP_REQUIRE(b && host_out, HP_ERR_INVALID, "hp_buffer_last_slots: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(n >= 0 && n <= b->staged_n, HP_ERR_INVALID, "hp_buffer_last_slots: n=%lld exceeds last store (%lld)",
               (long long)n, (long long)b->staged_n);
    HP_CHECK_HIP(hipMemcpyAsync(host_out, b->st_slots.p, n * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    return HP_OK;
}

int hp_buffer_read(hp_buffer *b, int32_t which, int64_t first, int64_t n, double *host_out) {
    HP_REQUIRE(b && host_out, HP_ERR_INVALID, "hp_buffer_read: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(first >= 0 && n >= 0 && first + n <= b->size, HP_ERR_INVALID, "hp_buffer_read: range out of bounds");
    const double *src;
    size_t ep;
    switch (which) {
        case 0: src = b->d_obs; ep = b->ep_obs(); break;
        case 1: src = b->d_ag; ep = b->ep_ag(); break;
        case 2: src = b->d_g; ep = b->ep_g(); break;
        case 3: src = b->d_act; ep = b->ep_act(); break;
        default: hp_set_error("hp_buffer_read: which=%d not in 0..3", which); return HP_ERR_INVALID;
    }
    if (n == 0) return HP_OK;
    HP_CHECK_HIP(hipMemcpyAsync(host_out, src + first * ep, n * ep * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    HP_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    return HP_OK;
}

// diagnostic: device time of the standalone sampler kernels (no host copies): one index draw, then `reps` launches each of
// the index-draw kernel and of the gather / relabel / reward kernel on that plan, bracketed by HIP events.  The stream
// position advances by 1 + reps draws.
int hp_buffer_sample_device_us(hp_buffer *b, hp_rng *rng, int64_t batch, double future_p, double sq_threshold,
                               int32_t reps, double *draw_us, double *gather_us) {
    HP_REQUIRE(b && rng && draw_us && gather_us && batch > 0 && reps > 0, HP_ERR_INVALID, "hp_buffer_sample_device_us: bad argument");
    HP_SERIALISE(b);
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    hipStream_t s = b->ctx->stream;
    const size_t row = (size_t)(2 * b->obs_dim + 3 * b->goal_dim + b->act_dim);
    HP_TRY(b->plan.ensure(batch * sizeof(PlanRec)));
    HP_TRY(b->out.ensure(batch * row * 8 + batch * 4));
    PlanRec *d_plan = b->plan.as<PlanRec>();
    double *d_out = b->out.as<double>();
    float *d_r = reinterpret_cast<float *>(d_out + batch * row);
    struct Events {   // destroyed on every exit path
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < 3; ++i) HP_CHECK_HIP(hipEventCreate(&ev.e[i]));
    HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    HP_TRY(buffer_launch_gather_dict(b, d_plan, batch, sq_threshold, d_out, d_r, nullptr));   // warm
    HP_CHECK_HIP(hipEventRecord(ev.e[0], s));
    for (int i = 0; i < reps; ++i) HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    HP_CHECK_HIP(hipEventRecord(ev.e[1], s));
    for (int i = 0; i < reps; ++i) HP_TRY(buffer_launch_gather_dict(b, d_plan, batch, sq_threshold, d_out, d_r, nullptr));
    HP_CHECK_HIP(hipEventRecord(ev.e[2], s));
    HP_CHECK_HIP(hipEventSynchronize(ev.e[2]));
    float ms01 = 0.f, ms12 = 0.f;
    HP_CHECK_HIP(hipEventElapsedTime(&ms01, ev.e[0], ev.e[1]));
    HP_CHECK_HIP(hipEventElapsedTime(&ms12, ev.e[1], ev.e[2]));
    *draw_us = 1e3 * ms01 / reps;
    *gather_us = 1e3 * ms12 / reps;
    return HP_OK;
}

int hp_buffer_sample(hp_buffer *b, hp_rng *rng, int64_t batch, double future_p, double sq_threshold,
                     const hp_sample_out *o) {
    HP_REQUIRE(b && rng && o, HP_ERR_INVALID, "hp_buffer_sample: null argument");
    HP_SERIALISE(b);
    HP_REQUIRE(batch > 0, HP_ERR_INVALID, "hp_buffer_sample: batch must be positive");
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");  // np.random.randint(0, 0, B), her.py:24
    hipStream_t s = b->ctx->stream;
    const int od = b->obs_dim, gd = b->goal_dim, ad = b->act_dim;
    const size_t row = (size_t)(2 * od + 3 * gd + ad);
    HP_TRY(b->plan.ensure(batch * sizeof(PlanRec)));
    HP_TRY(b->out.ensure(batch * row * 8 + batch * 8 + batch * 4));
    PlanRec *d_plan = b->plan.as<PlanRec>();
    double *d_out = b->out.as<double>();
    double *d_r64 = d_out + batch * row;
    float *d_r = reinterpret_cast<float *>(d_r64 + batch);
    HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    HP_TRY(buffer_launch_gather_dict(b, d_plan, batch, sq_threshold, d_out, d_r, o->r64 ? d_r64 : nullptr));
    double *p = d_out;
    auto pull = [&](double *dst, size_t n) -> hipError_t {
        hipError_t e = dst ? hipMemcpyAsync(dst, p, n * 8, hipMemcpyDeviceToHost, s) : hipSuccess;
        p += n;
        return e;
    };
    HP_CHECK_HIP(pull(o->obs, batch * od));
    HP_CHECK_HIP(pull(o->ag, batch * gd));
    HP_CHECK_HIP(pull(o->g, batch * gd));
    HP_CHECK_HIP(pull(o->actions, batch * ad));
    HP_CHECK_HIP(pull(o->obs_next, batch * od));
    HP_CHECK_HIP(pull(o->ag_next, batch * gd));
    if (o->r) HP_CHECK_HIP(hipMemcpyAsync(o->r, d_r, batch * 4, hipMemcpyDeviceToHost, s));
    if (o->r64) HP_CHECK_HIP(hipMemcpyAsync(o->r64, d_r64, batch * 8, hipMemcpyDeviceToHost, s));
    std::vector<PlanRec> hplan;
    if (o->e || o->t || o->future_t || o->her) {
        hplan.resize(batch);
        HP_CHECK_HIP(hipMemcpyAsync(hplan.data(), d_plan, batch * sizeof(PlanRec), hipMemcpyDeviceToHost, s));
    }
    HP_CHECK_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < hplan.size(); ++i) {
        if (o->e) o->e[i] = hplan[i].e;
        if (o->t) o->t[i] = hplan[i].t;
        if (o->future_t) o->future_t[i] = hplan[i].fut;
        if (o->her) o->her[i] = (uint8_t)hplan[i].her;
    }
    return HP_OK;
}

// replay_buffer.sample + the learner's preprocessing (ddpg_agent.py:227-243) with device outputs: index draw, then the fused
// gather.  Asynchronous on the context's stream; the outputs are caller-owned device memory (e.g. torch tensors).
struct FastDraw {   // hp_buffer_sample_dev_fast: counter-based index draw inside the gather kernel (fs_fast_rec)
    uint64_t seed, call;
    double future_p;
};
static int buffer_launch_gather_fused(hp_buffer *b, const PlanRec *d_plan, hp_norm *on, hp_norm *gn, int64_t batch,
                                      double sq_threshold, double clip_obs, const hp_sample_dev_out *o, const FastDraw *fast = nullptr) {
    FusedSampleArgs A;
    A.fast_seed = fast ? fast->seed : 0ull; A.fast_call = fast ? fast->call : 0ull;
    A.fast_n_eps = (int)b->current_size; A.fast_future_p = fast ? fast->future_p : 0.0;
    A.obs = b->d_obs; A.ag = b->d_ag; A.g = b->d_g; A.act = b->d_act;
    A.plan = d_plan;
    A.onz = on->d; A.gnz = gn->d;
    A.batch = batch;
    A.T = b->T; A.od = b->obs_dim; A.gd = b->goal_dim; A.ad = b->act_dim;
    A.sq_threshold = sq_threshold;
    A.clip_obs = clip_obs;
    A.clip_o = on->clip; A.clip_g = gn->clip;
    A.x = o->x; A.xn = o->x_next; A.a = o->actions; A.r = o->r;
    A.o_e = reinterpret_cast<long long *>(o->e); A.o_t = reinterpret_cast<long long *>(o->t);
    A.o_fut = reinterpret_cast<long long *>(o->future_t);
    A.o_her = o->her;
    const int64_t cap = (int64_t)b->ctx->cu_count * 32;           // grid-stride beyond 32 workgroups per compute unit
    const int ug = (b->goal_dim + 1) / 2, ua = (b->act_dim + 1) / 2;
    if (b->obs_dim + ug + ua <= 32 && b->goal_dim >= 2 && b->act_dim >= 2) {
        // 16-byte loads, two transitions per wavefront instruction (k_gather_fused2): the reference's shapes
        HP_KLOG("k_gather_fused2");
        if (batch >= 16384) {
            const int64_t waves = (batch + 2 * FS2_FLIGHT - 1) / (2 * FS2_FLIGHT), wgs = (waves + 3) / 4;
            const dim3 grid((unsigned)(wgs < cap ? wgs : cap));
            if (fast) hipLaunchKernelGGL((k_gather_fused2<FS2_FLIGHT, true>), grid, dim3(256), 0, b->ctx->stream, A);
            else hipLaunchKernelGGL((k_gather_fused2<FS2_FLIGHT, false>), grid, dim3(256), 0, b->ctx->stream, A);
        } else {
            const int64_t waves = (batch + 1) / 2, wgs = (waves + 3) / 4;
            const dim3 grid((unsigned)(wgs < cap ? wgs : cap));
            if (fast) hipLaunchKernelGGL((k_gather_fused2<1, true>), grid, dim3(256), 0, b->ctx->stream, A);
            else hipLaunchKernelGGL((k_gather_fused2<1, false>), grid, dim3(256), 0, b->ctx->stream, A);
        }
        HP_CHECK_HIP(hipGetLastError());
        return HP_OK;
    }
    HP_REQUIRE(!fast, HP_ERR_INVALID, "hp_buffer_sample_dev_fast: needs obs_dim + ceil(goal_dim / 2) + ceil(act_dim / 2) <= 32 and goal_dim, act_dim >= 2");
    // (us per 262144 transitions of a 5000-episode shard: 4 in flight, grid capped at 8 / 32 workgroups per CU 126.8 / 121.0; 8 in
    // flight 133.6; 2 in flight, cap 16: 130.5; 1 in flight, cap 64: 124.0 -- ~2.1 G transitions/s whatever the shape of the launch)
    const int flight = batch >= 16384 ? 4 : 1;
    const int64_t waves = (batch + flight - 1) / flight, wgs = (waves + 3) / 4;
    HP_KLOG("k_gather_fused");
    if (flight == 4) hipLaunchKernelGGL(k_gather_fused<4>, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(256), 0, b->ctx->stream, A);
    else hipLaunchKernelGGL(k_gather_fused<1>, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(256), 0, b->ctx->stream, A);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static int buffer_launch_gather_packed(hp_buffer *b, const PlanRec *d_plan, hp_norm *on, hp_norm *gn, int64_t batch,
                                       double sq_threshold, double clip_obs, const hp_sample_dev_out *o, const FastDraw *fast = nullptr) {
    PackedSampleArgs P;
    P.p_row = b->p_row; P.p_goal = b->p_goal; P.row_w = b->row_w; P.goal_w = b->goal_w;
    FusedSampleArgs &A = P.f;
    A.fast_seed = fast ? fast->seed : 0ull; A.fast_call = fast ? fast->call : 0ull;
    A.fast_n_eps = (int)b->current_size; A.fast_future_p = fast ? fast->future_p : 0.0;
    A.obs = b->d_obs; A.ag = b->d_ag; A.g = b->d_g; A.act = b->d_act;
    A.plan = d_plan;
    A.onz = on->d; A.gnz = gn->d;
    A.batch = batch;
    A.T = b->T; A.od = b->obs_dim; A.gd = b->goal_dim; A.ad = b->act_dim;
    A.sq_threshold = sq_threshold;
    A.clip_obs = clip_obs;
    A.clip_o = on->clip; A.clip_g = gn->clip;
    A.x = o->x; A.xn = o->x_next; A.a = o->actions; A.r = o->r;
    A.o_e = reinterpret_cast<long long *>(o->e); A.o_t = reinterpret_cast<long long *>(o->t);
    A.o_fut = reinterpret_cast<long long *>(o->future_t);
    A.o_her = o->her;
    const int64_t cap = (int64_t)b->ctx->cu_count * 32;
    HP_KLOG("k_gather_packed");
    if (batch >= 16384) {
        const int64_t waves = (batch + 2 * FS2_FLIGHT - 1) / (2 * FS2_FLIGHT), wgs = (waves + 3) / 4;
        const dim3 grid((unsigned)(wgs < cap ? wgs : cap));
        if (fast) hipLaunchKernelGGL((k_gather_packed<FS2_FLIGHT, true>), grid, dim3(256), 0, b->ctx->stream, P);
        else hipLaunchKernelGGL((k_gather_packed<FS2_FLIGHT, false>), grid, dim3(256), 0, b->ctx->stream, P);
    } else {
        const int64_t waves = (batch + 1) / 2, wgs = (waves + 3) / 4;
        const dim3 grid((unsigned)(wgs < cap ? wgs : cap));
        if (fast) hipLaunchKernelGGL((k_gather_packed<1, true>), grid, dim3(256), 0, b->ctx->stream, P);
        else hipLaunchKernelGGL((k_gather_packed<1, false>), grid, dim3(256), 0, b->ctx->stream, P);
    }
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

static int sample_dev_check(hp_buffer *b, hp_rng *rng, hp_norm *on, hp_norm *gn, int64_t batch, double clip_obs,
                            const hp_sample_dev_out *o, const char *who) {
    HP_REQUIRE(b && rng && on && gn && o, HP_ERR_INVALID, "%s: null argument", who);
    HP_REQUIRE(on->ctx == b->ctx && gn->ctx == b->ctx && rng->ctx == b->ctx, HP_ERR_INVALID, "%s: handles of different contexts", who);
    HP_REQUIRE(on->size == b->obs_dim && gn->size == b->goal_dim, HP_ERR_INVALID,
               "%s: normalizer sizes (%d, %d) do not match the buffer's observation / goal widths (%d, %d)", who, on->size, gn->size,
               b->obs_dim, b->goal_dim);
    HP_REQUIRE(batch > 0, HP_ERR_INVALID, "%s: batch must be positive", who);
    HP_REQUIRE(clip_obs > 0, HP_ERR_INVALID, "%s: clip_obs must be positive (arguments.py:87; pass a huge value for none)", who);
    HP_REQUIRE(b->goal_dim <= 64, HP_ERR_INVALID, "%s: goal_dim > 64", who);
    return HP_OK;
}

extern "C" int hp_buffer_sample_dev(hp_buffer *b, hp_rng *rng, hp_norm *on, hp_norm *gn, int64_t batch, double future_p,
                                    double sq_threshold, double clip_obs, const hp_sample_dev_out *o) {
    HP_TRY(sample_dev_check(b, rng, on, gn, batch, clip_obs, o, "hp_buffer_sample_dev"));
    HP_SERIALISE(b);
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");  // np.random.randint(0, 0, B), her.py:24
    if ((size_t)batch * sizeof(PlanRec) > b->plan.bytes) HP_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));   // an earlier asynchronous call may still read the plan that is about to be freed
    HP_TRY(b->plan.ensure(batch * sizeof(PlanRec)));
    PlanRec *d_plan = b->plan.as<PlanRec>();
    HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    return buffer_launch_gather_fused(b, d_plan, on, gn, batch, sq_threshold, clip_obs, o);
}

// Throughput mode (storage_dtype = fp32).  hp_buffer_enable_f32_rows builds the float32 mirror of observations and
// actions (+ the packed float64 goals) from what the buffer holds and keeps it current behind every later store;
// hp_buffer_sample_dev_f32 is hp_buffer_sample_dev reading it: the same draws from the same stream, the same indices, relabelled
// goals, rewards and goal columns bit for bit; the observation columns are those of float32-rounded observations.
extern "C" int hp_buffer_enable_f32_rows(hp_buffer *b) {
    HP_REQUIRE(b, HP_ERR_INVALID, "hp_buffer_enable_f32_rows: null handle");
    HP_SERIALISE(b);
    if (b->p_row) return HP_OK;
    const int row_w = (b->obs_dim + b->act_dim + 31) / 32 * 32, goal_w = (2 * b->goal_dim + 7) / 8 * 8;
    HP_REQUIRE(row_w == 32 && b->obs_dim <= 28 && b->goal_dim <= 4, HP_ERR_INVALID,
               "hp_buffer_enable_f32_rows: needs obs_dim <= 28, obs_dim + act_dim <= 32 (one 128-byte line of float32 per timestep) and "
               "goal_dim <= 4; got %d, %d, %d", b->obs_dim, b->act_dim, b->goal_dim);
    const size_t rows = (size_t)b->size * (b->T + 1);
    HP_CHECK_HIP(hipMalloc((void **)&b->p_row, rows * row_w * sizeof(float)));
    if (hipMalloc((void **)&b->p_goal, rows * goal_w * sizeof(double)) != hipSuccess) {
        (void)hipFree(b->p_row);
        b->p_row = nullptr;
        hp_set_error("hp_buffer_enable_f32_rows: cannot allocate the goal rows");
        return HP_ERR_HIP;
    }
    b->row_w = row_w;
    b->goal_w = goal_w;
    return launch_pack(b, nullptr, 0, b->size);      // (rows of unfilled slots hold whatever the allocation held: never sampled)
}

extern "C" int hp_buffer_sample_dev_f32(hp_buffer *b, hp_rng *rng, hp_norm *on, hp_norm *gn, int64_t batch, double future_p,
                                        double sq_threshold, double clip_obs, const hp_sample_dev_out *o) {
    HP_TRY(sample_dev_check(b, rng, on, gn, batch, clip_obs, o, "hp_buffer_sample_dev_f32"));
    HP_SERIALISE(b);
    HP_REQUIRE(b->p_row, HP_ERR_STATE, "hp_buffer_sample_dev_f32: hp_buffer_enable_f32_rows first");
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    if ((size_t)batch * sizeof(PlanRec) > b->plan.bytes) HP_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    HP_TRY(b->plan.ensure(batch * sizeof(PlanRec)));
    PlanRec *d_plan = b->plan.as<PlanRec>();
    HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    return buffer_launch_gather_packed(b, d_plan, on, gn, batch, sq_threshold, clip_obs, o);
}

// Fast draw (SURVEY 8b rng_mode = Philox; opt-in, NOT the reference's random stream): hp_buffer_sample_dev / _f32 with the index
// draw inside the gather kernel, keyed by (seed, call, transition) -- no hp_rng, no sequential draw kernel, one launch.  The caller
// owns the counter: the same (seed, call) gives the same minibatch; a training loop passes seed + rank and call = 0, 1, 2, ...
extern "C" int hp_buffer_sample_dev_fast(hp_buffer *b, hp_norm *on, hp_norm *gn, int64_t batch, double future_p, double sq_threshold,
                                         double clip_obs, uint64_t seed, uint64_t call, int32_t f32_rows, const hp_sample_dev_out *o) {
    HP_REQUIRE(b && on && gn && o, HP_ERR_INVALID, "hp_buffer_sample_dev_fast: null argument");
    HP_REQUIRE(on->ctx == b->ctx && gn->ctx == b->ctx, HP_ERR_INVALID, "hp_buffer_sample_dev_fast: handles of different contexts");
    HP_REQUIRE(on->size == b->obs_dim && gn->size == b->goal_dim, HP_ERR_INVALID,
               "hp_buffer_sample_dev_fast: normalizer sizes (%d, %d) do not match the buffer's observation / goal widths (%d, %d)", on->size,
               gn->size, b->obs_dim, b->goal_dim);
    HP_REQUIRE(batch > 0 && clip_obs > 0, HP_ERR_INVALID, "hp_buffer_sample_dev_fast: batch and clip_obs must be positive");
    HP_SERIALISE(b);
    HP_REQUIRE(!f32_rows || b->p_row, HP_ERR_STATE, "hp_buffer_sample_dev_fast: hp_buffer_enable_f32_rows first");
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    const FastDraw fd{seed, call, future_p};
    return f32_rows ? buffer_launch_gather_packed(b, nullptr, on, gn, batch, sq_threshold, clip_obs, o, &fd)
                    : buffer_launch_gather_fused(b, nullptr, on, gn, batch, sq_threshold, clip_obs, o, &fd);
}

// diagnostic: device microseconds per hp_buffer_sample_dev_fast launch (outputs into library scratch), `reps` back to back
extern "C" int hp_buffer_sample_dev_fast_us(hp_buffer *b, hp_norm *on, hp_norm *gn, int64_t batch, double future_p, double sq_threshold,
                                            double clip_obs, int32_t reps, int32_t f32_rows, double *us) {
    HP_REQUIRE(b && on && gn && us && batch > 0 && reps > 0, HP_ERR_INVALID, "hp_buffer_sample_dev_fast_us: bad argument");
    HP_SERIALISE(b);
    hipStream_t s = b->ctx->stream;
    const size_t ldx = (size_t)(b->obs_dim + b->goal_dim);
    if ((size_t)batch * (2 * ldx + b->act_dim + 1) * 4 > b->out.bytes) HP_CHECK_HIP(hipStreamSynchronize(s));
    HP_TRY(b->out.ensure((size_t)batch * (2 * ldx + b->act_dim + 1) * 4));
    hp_sample_dev_out o;
    memset(&o, 0, sizeof(o));
    o.x = b->out.as<float>();
    o.x_next = o.x + batch * ldx;
    o.actions = o.x_next + batch * ldx;
    o.r = o.actions + batch * b->act_dim;
    struct Events {
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < 2; ++i) HP_CHECK_HIP(hipEventCreate(&ev.e[i]));
    HP_TRY(hp_buffer_sample_dev_fast(b, on, gn, batch, future_p, sq_threshold, clip_obs, 1, 0, f32_rows, &o));   // warm
    HP_CHECK_HIP(hipEventRecord(ev.e[0], s));
    for (int i = 0; i < reps; ++i) HP_TRY(hp_buffer_sample_dev_fast(b, on, gn, batch, future_p, sq_threshold, clip_obs, 1, 1 + i, f32_rows, &o));
    HP_CHECK_HIP(hipEventRecord(ev.e[1], s));
    HP_CHECK_HIP(hipEventSynchronize(ev.e[1]));
    float ms = 0.f;
    HP_CHECK_HIP(hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
    *us = 1e3 * ms / reps;
    return HP_OK;
}

// diagnostic twin of hp_buffer_sample_device_us for the fused kernel (outputs into library scratch); f32_rows != 0: the
// throughput-rows kernel (hp_buffer_sample_dev_f32)
extern "C" int hp_buffer_sample_dev_us(hp_buffer *b, hp_rng *rng, hp_norm *on, hp_norm *gn, int64_t batch, double future_p,
                                       double sq_threshold, double clip_obs, int32_t reps, int32_t f32_rows, double *draw_us,
                                       double *gather_us) {
    hp_sample_dev_out o;
    memset(&o, 0, sizeof(o));
    HP_TRY(sample_dev_check(b, rng, on, gn, batch, clip_obs, &o, "hp_buffer_sample_dev_us"));
    HP_REQUIRE(draw_us && gather_us && reps > 0, HP_ERR_INVALID, "hp_buffer_sample_dev_us: bad argument");
    HP_SERIALISE(b);
    HP_REQUIRE(b->current_size > 0, HP_ERR_EMPTY, "high <= 0");
    HP_REQUIRE(!f32_rows || b->p_row, HP_ERR_STATE, "hp_buffer_sample_dev_us: hp_buffer_enable_f32_rows first");
    auto gather = [&](const PlanRec *plan, const hp_sample_dev_out *out) {
        return f32_rows ? buffer_launch_gather_packed(b, plan, on, gn, batch, sq_threshold, clip_obs, out)
                        : buffer_launch_gather_fused(b, plan, on, gn, batch, sq_threshold, clip_obs, out);
    };
    hipStream_t s = b->ctx->stream;
    const size_t ldx = (size_t)(b->obs_dim + b->goal_dim);
    // growing `plan` / `out` frees memory that earlier asynchronous hp_buffer_sample_dev launches may still read: wait for them
    // explicitly instead of relying on hipFree's implicit device synchronisation
    if ((size_t)batch * sizeof(PlanRec) > b->plan.bytes || (size_t)batch * (2 * ldx + b->act_dim + 1) * 4 > b->out.bytes)
        HP_CHECK_HIP(hipStreamSynchronize(s));
    HP_TRY(b->plan.ensure(batch * sizeof(PlanRec)));
    HP_TRY(b->out.ensure((size_t)batch * (2 * ldx + b->act_dim + 1) * 4));
    PlanRec *d_plan = b->plan.as<PlanRec>();
    o.x = b->out.as<float>();
    o.x_next = o.x + batch * ldx;
    o.actions = o.x_next + batch * ldx;
    o.r = o.actions + batch * b->act_dim;
    struct Events {   // destroyed on every exit path
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < 3; ++i) HP_CHECK_HIP(hipEventCreate(&ev.e[i]));
    HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    HP_TRY(gather(d_plan, &o));   // warm
    HP_CHECK_HIP(hipEventRecord(ev.e[0], s));
    for (int i = 0; i < reps; ++i) HP_TRY(rng_launch_plan(rng, b->d_meta, 0, b->T, batch, 1, future_p, d_plan));
    HP_CHECK_HIP(hipEventRecord(ev.e[1], s));
    for (int i = 0; i < reps; ++i) HP_TRY(gather(d_plan, &o));
    HP_CHECK_HIP(hipEventRecord(ev.e[2], s));
    HP_CHECK_HIP(hipEventSynchronize(ev.e[2]));
    float ms01 = 0.f, ms12 = 0.f;
    HP_CHECK_HIP(hipEventElapsedTime(&ms01, ev.e[0], ev.e[1]));
    HP_CHECK_HIP(hipEventElapsedTime(&ms12, ev.e[1], ev.e[2]));
    *draw_us = 1e3 * ms01 / reps;
    *gather_us = 1e3 * ms12 / reps;
    return HP_OK;
}

// ---- compute_reward / _is_success (bmirobot_env_push_F.py:84-90, :243-245) as batched device ops ------------------
static int goal_reward_launch(hp_ctx *ctx, const double *ag, const double *g, int64_t n, int32_t goal_dim, double thr_sq,
                              int mode, float *out32, double *out64) {
    hipLaunchKernelGGL(k_goal_reward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ag, g, (long long)n,
                       (int)goal_dim, thr_sq, mode, out32, out64);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

// smallest double s whose correctly rounded square root is > thr (strict) or >= thr: the predicates d > thr and
// !(d < thr) in the squared domain (sqrt_rn is monotone), so no device square root and no rounding mismatch
static double squared_bound(double thr, bool strict) {
    if (!(thr >= 0.0)) return 0.0;
    double s = thr * thr;
    auto hit = [&](double v) { const double r = std::sqrt(v); return strict ? r > thr : r >= thr; };
    while (s > 0.0 && hit(s)) s = std::nextafter(s, -INFINITY);
    while (!hit(s)) s = std::nextafter(s, INFINITY);
    return s;
}

int hp_compute_reward_dev(hp_ctx *ctx, const double *ag_dev, const double *g_dev, int64_t n, int32_t goal_dim,
                          double distance_threshold, int32_t dense, float *sparse_out_dev, double *dense_out_dev) {
    HP_REQUIRE(ctx && ag_dev && g_dev, HP_ERR_INVALID, "hp_compute_reward_dev: null argument");
    CtxGuard guard(ctx);
    HP_REQUIRE(n >= 0 && goal_dim > 0, HP_ERR_INVALID, "hp_compute_reward_dev: bad shape");
    HP_REQUIRE(dense ? dense_out_dev != nullptr : sparse_out_dev != nullptr, HP_ERR_INVALID,
               "hp_compute_reward_dev: the output of the selected reward type is null");
    if (n == 0) return HP_OK;
    return goal_reward_launch(ctx, ag_dev, g_dev, n, goal_dim, dense ? 0.0 : squared_bound(distance_threshold, true),
                              dense ? 1 : 0, sparse_out_dev, dense_out_dev);
}

int hp_is_success_dev(hp_ctx *ctx, const double *ag_dev, const double *g_dev, int64_t n, int32_t goal_dim,
                      double distance_threshold, float *out_dev) {
    HP_REQUIRE(ctx && ag_dev && g_dev && out_dev, HP_ERR_INVALID, "hp_is_success_dev: null argument");
    CtxGuard guard(ctx);
    HP_REQUIRE(n >= 0 && goal_dim > 0, HP_ERR_INVALID, "hp_is_success_dev: bad shape");
    if (n == 0) return HP_OK;
    return goal_reward_launch(ctx, ag_dev, g_dev, n, goal_dim, squared_bound(distance_threshold, false), 2, out_dev,
                              nullptr);
}

// host-array forms (the GoalEnv API hands numpy arrays over): staged through the device, same kernels
static int goal_reward_host(hp_ctx *ctx, const double *ag, const double *g, int64_t n, int32_t goal_dim, double thr_sq,
                            int mode, void *out_host) {
    DevBuf &ws = ctx->reward_ws;     // guarded by the context lock the callers hold
    const size_t nb = (size_t)n * goal_dim * 8, ob = (size_t)n * (mode == 1 ? 8 : 4);
    HP_TRY(ws.ensure(2 * nb + ob));
    char *d = ws.as<char>();
    hipStream_t s = ctx->stream;
    HP_CHECK_HIP(hipMemcpyAsync(d, ag, nb, hipMemcpyHostToDevice, s));
    HP_CHECK_HIP(hipMemcpyAsync(d + nb, g, nb, hipMemcpyHostToDevice, s));
    HP_TRY(goal_reward_launch(ctx, reinterpret_cast<double *>(d), reinterpret_cast<double *>(d + nb), n, goal_dim, thr_sq,
                              mode, reinterpret_cast<float *>(d + 2 * nb), reinterpret_cast<double *>(d + 2 * nb)));
    HP_CHECK_HIP(hipMemcpyAsync(out_host, d + 2 * nb, ob, hipMemcpyDeviceToHost, s));
    HP_CHECK_HIP(hipStreamSynchronize(s));
    return HP_OK;
}

int hp_compute_reward(hp_ctx *ctx, const double *ag_host, const double *g_host, int64_t n, int32_t goal_dim,
                      double distance_threshold, int32_t dense, float *sparse_out_host, double *dense_out_host) {
    HP_REQUIRE(ctx && ag_host && g_host, HP_ERR_INVALID, "hp_compute_reward: null argument");
    CtxGuard guard(ctx);
    HP_REQUIRE(n >= 0 && goal_dim > 0, HP_ERR_INVALID, "hp_compute_reward: bad shape");
    HP_REQUIRE(dense ? dense_out_host != nullptr : sparse_out_host != nullptr, HP_ERR_INVALID,
               "hp_compute_reward: the output of the selected reward type is null");
    if (n == 0) return HP_OK;
    return goal_reward_host(ctx, ag_host, g_host, n, goal_dim, dense ? 0.0 : squared_bound(distance_threshold, true),
                            dense ? 1 : 0, dense ? (void *)dense_out_host : (void *)sparse_out_host);
}

int hp_is_success(hp_ctx *ctx, const double *ag_host, const double *g_host, int64_t n, int32_t goal_dim,
                  double distance_threshold, float *out_host) {
    HP_REQUIRE(ctx && ag_host && g_host && out_host, HP_ERR_INVALID, "hp_is_success: null argument");
    CtxGuard guard(ctx);
    HP_REQUIRE(n >= 0 && goal_dim > 0, HP_ERR_INVALID, "hp_is_success: bad shape");
    if (n == 0) return HP_OK;
    return goal_reward_host(ctx, ag_host, g_host, n, goal_dim, squared_bound(distance_threshold, false), 2, out_host);
}

void hp_buffer_destroy(hp_buffer *b) {
    if (!b) return;
    if (b->d_obs) (void)hipFree(b->d_obs);
    if (b->d_ag) (void)hipFree(b->d_ag);
    if (b->d_g) (void)hipFree(b->d_g);
    if (b->d_act) (void)hipFree(b->d_act);
    if (b->d_meta) (void)hipFree(b->d_meta);
    if (b->p_row) (void)hipFree(b->p_row);
    if (b->p_goal) (void)hipFree(b->p_goal);
    for (hipEvent_t ev : b->pin_events)
        if (ev) (void)hipEventDestroy(ev);
    b->st_obs.release();
    b->st_slots.release();
    b->pin.release();
    b->plan.release();
    b->out.release();
    delete b;
}

}  // extern "C"
