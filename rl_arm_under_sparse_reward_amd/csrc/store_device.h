// store_device.h -- device side of replay_buffer.store_episode's scatter (replay_buffer.py:39-42), shared by buffer.hip's
// k_store_scatter and the cycle-opening kernel (cycle_open.hip).
#pragma once
#include "internal.h"

// Episode i of the staged batch goes to slot slots[i] unless a LATER staged episode has the same slot (numpy's
// `buffers[idxs] = mb` lets the last occurrence win).  One of `parts` workgroups per episode copies a strided share of its
// values.  Workgroup-uniform control flow; `slot_of(j)` reads slots[j] (plain or agent-scope load, the caller knows).
template <class SlotOf>
__device__ __forceinline__ void store_scatter_share(SlotOf slot_of, long long i, int part, int parts, long long n_new,
                                                    const double *s_obs, const double *s_ag, const double *s_g,
                                                    const double *s_act, double *obs, double *ag, double *g, double *act,
                                                    long long ep_obs, long long ep_ag, long long ep_g, long long ep_act) {
    const long long slot = slot_of(i);
    int dup = 0;
    for (long long j = i + 1 + threadIdx.x; j < n_new; j += blockDim.x) dup |= (slot_of(j) == slot);
    if (__syncthreads_or(dup)) return;
    const long long t0 = (long long)part * blockDim.x + threadIdx.x, step = (long long)parts * blockDim.x;
    for (long long k = t0; k < ep_obs; k += step) obs[slot * ep_obs + k] = s_obs[i * ep_obs + k];
    for (long long k = t0; k < ep_ag; k += step) ag[slot * ep_ag + k] = s_ag[i * ep_ag + k];
    for (long long k = t0; k < ep_g; k += step) g[slot * ep_g + k] = s_g[i * ep_g + k];
    for (long long k = t0; k < ep_act; k += step) act[slot * ep_act + k] = s_act[i * ep_act + k];
}
