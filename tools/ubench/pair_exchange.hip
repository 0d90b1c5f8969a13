// Microbenchmark for VERDICT r02 item 3: what does it cost two workgroups ON THE SAME XCD (workgroup ids i and i ^ 8: the
// dispatcher deals workgroups round-robin to the 8 XCDs) to exchange a slab of partial sums through their shared L2?
//
//   each workgroup:  store its `bytes` of partial sums (plain or sc1 stores)  ->  s_waitcnt vmcnt(0) + barrier  ->  one lane
//                    stores a flag  ->  one lane polls the partner's flag (L1-bypassing load)  ->  barrier  ->  all lanes
//                    load the partner's slab (L1-bypassing loads)  ->  s_waitcnt vmcnt(0)
//
// That is the hand-off a Megatron-style split of two consecutive 256 x 256 layers over a CU pair would need once per two
// layers (layer l by output columns, layer l + 1 by reduction index; 4 or 8 rows x 256 partial sums = 4 / 8 KB per side).
// It has to beat the 2 x ~1.15 us of weight streaming it saves.  Reported: ns per round trip (100 MHz wall clock over
// `iters` rounds, all pairs running at once like the chains would), the XCD ids of the pair, and a data check.
// Build: hipcc --offload-arch=gfx950 -O3 pair_exchange.hip -o pair_exchange.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// gfx940+ scope bits of a load: none = wavefront (may hit this CU's L1), sc0 = workgroup (still this CU's L1), sc1 = agent
// (misses the L1, served by THIS XCD's L2), sc0 sc1 = system.  Same-XCD exchange therefore needs sc1 loads and nothing
// special on the stores (the L1 is write-through); crossing XCDs needs system scope on both sides (fabric).
template <int SCOPE>   // 0: agent-scope loads, plain stores (shared L2)   1: system scope (sc0 sc1) loads and stores
__device__ __forceinline__ f32x4 load4_bypass(const float *p) {
    f32x4 v;
    if (SCOPE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE>
__device__ __forceinline__ unsigned load1_bypass(const unsigned *p) {
    unsigned v;
    if (SCOPE == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE>
__device__ __forceinline__ void store4(float *p, f32x4 v) {
    if (SCOPE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE>
__device__ __forceinline__ void store1(unsigned *p, unsigned v) {
    if (SCOPE == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// slab: [n_wg][2][floats] (double buffered by round parity), flags: [n_wg] (64 B apart)
template <int SCOPE>
__global__ __launch_bounds__(512) void k_pair(float *slab, unsigned *flags, int floats, int iters, unsigned long long *ticks,
                                              unsigned *xcd, unsigned *bad, int exchange) {
    const int me = blockIdx.x, partner = me ^ 8, tid = threadIdx.x;
    const int n4 = floats / 4;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcd[me] = id & 0xf;
    }
    unsigned wrong = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        float *mine = slab + ((size_t)me * 2 + (it & 1)) * floats;
        const float *theirs = slab + ((size_t)partner * 2 + (it & 1)) * floats;
        for (int i = tid; i < n4; i += 512) {
            const float b = (float)(me * 1000 + it) + acc.x * 0.f;   // depends on the previous round's data
            store4<SCOPE>(mine + 4 * i, f32x4{b, b + 1.f, b + 2.f, (float)i});
        }
        if (exchange) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its stores have been acknowledged by L2
            __syncthreads();
            if (tid == 0) {
                store1<SCOPE>(flags + me * 16, (unsigned)it);
                int spins = 0;   // bounded: a stale-read bug must not hang the box
                while (load1_bypass<SCOPE>(flags + partner * 16) < (unsigned)it && ++spins < 20000) __builtin_amdgcn_s_sleep(1);
                if (spins >= 20000) { atomicAdd(bad, 1000000u); it = iters; }
            }
            __syncthreads();
            for (int i = tid; i < n4; i += 512) {
                const f32x4 v = load4_bypass<SCOPE>(theirs + 4 * i);
                const float b = (float)(partner * 1000 + it);
                wrong += (v.x != b) | (v.y != b + 1.f) | (v.z != b + 2.f) | (v.w != (float)i);
                acc += v;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ticks[me] = t1 - t0;
    if (wrong) atomicAdd(bad, wrong);
    if (acc.x == 1.2345f) slab[0] = acc.y;
}

int main(int argc, char **argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    float *slab; unsigned *flags, *xcd, *bad; unsigned long long *ticks;
    const int max_floats = 4096;
    CK(hipMalloc(&slab, (size_t)n_wg * 2 * max_floats * 4));
    CK(hipMalloc(&flags, n_wg * 64)); CK(hipMalloc(&xcd, n_wg * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&ticks, n_wg * 8));
    std::vector<unsigned long long> ht(n_wg); std::vector<unsigned> hx(n_wg);
    for (int scope = 0; scope < 2; ++scope)
        for (int floats : {1024, 2048, 4096})
            for (int exchange = 0; exchange < 2; ++exchange) {
                CK(hipMemset(flags, 0, n_wg * 64)); CK(hipMemset(bad, 0, 4));
                if (scope == 0) hipLaunchKernelGGL(k_pair<0>, dim3(n_wg), dim3(512), 0, 0, slab, flags, floats, iters, ticks, xcd, bad, exchange);
                else hipLaunchKernelGGL(k_pair<1>, dim3(n_wg), dim3(512), 0, 0, slab, flags, floats, iters, ticks, xcd, bad, exchange);
                CK(hipDeviceSynchronize());
                unsigned hb = 0;
                CK(hipMemcpy(ht.data(), ticks, n_wg * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcd, n_wg * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                double mean = 0, mx = 0; int same = 0;
                for (int i = 0; i < n_wg; ++i) { const double ns = 10.0 * ht[i] / iters; mean += ns / n_wg; mx = ns > mx ? ns : mx; same += hx[i] == hx[i ^ 8]; }
                printf("%s %5d B per side, %s: %7.1f ns per round (max %7.1f), pairs on one XCD %d/%d, wrong words %u\n",
                       scope ? "system scope (fabric)  " : "agent loads (shared L2)", floats * 4, exchange ? "store+flag+poll+load" : "store only (baseline)",
                       mean, mx, same, n_wg, hb);
            }
    return 0;
}
