// l2_coherence.hip -- what one launch may assume about the per-XCD L2s (MI355X: 8 XCDs, workgroup b runs on XCD b % 8).
// The carry form of the split launch (slab8_split.h) lets workgroups of ONE launch read, with plain loads, parameters that
// workgroups on OTHER XCDs stepped (write-through, sc1) earlier in the same launch.  Three questions, answered by this probe:
//   1. a line cached in XCD 0's L2, then rewritten write-through from XCD 1: does a plain load on XCD 0 (another CU: clean L1)
//      still see the old value (stale hit)?  and after an agent-scope acquire fence (buffer_inv sc1)?
//   2. XCD 0 writes 16 B of a line it does not hold (sc1), XCD 1 then writes the next 16 B (sc1): does a plain load on XCD 0
//      see XCD 1's bytes (no fetch on a partial write) or the old ones (the write allocated the whole line)?
//   3. the same as 2 with a plain (write-back) store from XCD 0.
// build: hipcc --offload-arch=gfx950 -O2 -o l2_coherence.bin l2_coherence.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float ld_plain(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_wt(float *p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_plain(float *p, float v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void sig(unsigned *f) { __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait(unsigned *f) {
    long long n = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++n < (1ll << 24)) __builtin_amdgcn_s_sleep(4);
}

// buf: lines of 32 floats, all 1.0f at entry.  out[]: what the probes saw.
__global__ void k_probe(float *buf, unsigned *flag, float *out) {
    if (threadIdx.x != 0) return;
    float *L = buf, *M = buf + 64, *N = buf + 128;
    const int b = blockIdx.x;
    if (b == 0) {                       // XCD 0, CU a
        out[0] = ld_plain(L);           // line L now sits in XCD 0's L2
        st_wt(M, 5.f);                  // 4 bytes of line M, written through from XCD 0 (M is in nobody's cache)
        st_plain(N, 5.f);               // ... and of line N, write-back
        sig(flag + 0);
    } else if (b == 1) {                // XCD 1
        wait(flag + 0);
        for (int i = 0; i < 32; ++i) st_wt(L + i, 2.f);
        st_wt(M + 4, 6.f);
        st_wt(N + 4, 6.f);
        sig(flag + 1);
    } else if (b == 8) {                // XCD 0, another CU (clean L1)
        wait(flag + 1);
        out[1] = ld_plain(L + 1);       // stale (1) or fresh (2)?
        out[3] = ld_plain(M + 4);       // 6: partial sc1 write did not allocate the rest of the line; 1: it did
        out[4] = ld_plain(N + 4);       // the same after a plain store
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        out[2] = ld_plain(L + 2);       // after buffer_inv sc1
        out[5] = ld_plain(M + 5);
        out[6] = ld_plain(N + 5);
    } else if (b == 9) {                // XCD 1, another CU: sees its own XCD's writes
        wait(flag + 1);
        out[7] = ld_plain(L + 3);
    }
}

int main() {
    float *buf, *out; unsigned *flag;
    CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&flag, 64));
    int stale = 0, part_wt = 0, part_wb = 0, after = 0;
    for (int rep = 0; rep < 200; ++rep) {
        float ones[256]; for (float &x : ones) x = 1.f;
        CK(hipMemcpy(buf, ones, sizeof(ones), hipMemcpyHostToDevice));
        CK(hipMemset(flag, 0, 64)); CK(hipMemset(out, 0, 64));
        hipLaunchKernelGGL(k_probe, dim3(16), dim3(64), 0, 0, buf, flag, out);
        CK(hipDeviceSynchronize());
        float h[16]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        if (rep == 0) printf("first run: L before %.0f | plain reload on XCD 0 %.0f | after acquire fence %.0f | partial sc1 line %.0f -> %.0f | partial plain line %.0f -> %.0f | XCD 1 own %.0f\n",
                             h[0], h[1], h[2], h[3], h[5], h[4], h[6], h[7]);
        stale += h[1] != 2.f; part_wt += h[3] != 6.f; part_wb += h[4] != 6.f; after += (h[2] != 2.f) + (h[5] != 1.f && h[5] != 6.f);
    }
    printf("200 runs: stale plain reload of a cached line %d | partial-line sc1 write hid the peer's bytes %d | partial-line plain write hid them %d | wrong after the fence %d\n",
           stale, part_wt, part_wb, after);
    return 0;
}
