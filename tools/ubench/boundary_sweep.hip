// Microbenchmark (round 4): what sets the gap between two dependent kernels of a hipGraph?  graph_branches.hip showed 2.5 us
// in front of a 296 x 74 KB launch and 4.0 us in front of a 141 x 119 KB launch with spin kernels that touch no memory --
// 6.5 us of a 40 us update.  Here: X -> Y -> X -> Y ... with Y's shape swept (workgroups, LDS bytes, threads, kernarg bytes),
// gap = first start stamp of a kernel - last end stamp of its predecessor (100 MHz wall clock).
// Build: hipcc --offload-arch=gfx950 -O3 boundary_sweep.hip -o boundary_sweep.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long first, last, first_end; };
struct Pad { char b[1024]; };

__device__ __forceinline__ void body(Stamp *st, int node, unsigned ticks, float *lds) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&st[node].first, t0);
    lds[threadIdx.x] = (float)t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long t1 = wall_clock64(); atomicMax(&st[node].last, t1); atomicMin(&st[node].first_end, t1); }
    if (lds[threadIdx.x ^ 1] == 1.5f) st[node].first = 0;
}
__global__ __launch_bounds__(512) void k_spin(Stamp *st, int node, unsigned ticks) {
    extern __shared__ float lds[];
    body(st, node, ticks, lds);
}
__global__ __launch_bounds__(512) void k_spin_bigarg(Stamp *st, int node, unsigned ticks, const Pad pad) {
    extern __shared__ float lds[];
    body(st, node, ticks + (pad.b[7] == 77), lds);
}
// ends at a FIXED wall-clock offset from the kernel's first start instead of per workgroup: all workgroups end together
__global__ __launch_bounds__(512) void k_spin_together(Stamp *st, int node, unsigned ticks) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&st[node].first, t0);
    lds[threadIdx.x] = (float)t0;
    __syncthreads();
    for (;;) {
        const unsigned long long f = __hip_atomic_load(&st[node].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wall_clock64() - f >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long t1 = wall_clock64(); atomicMax(&st[node].last, t1); atomicMin(&st[node].first_end, t1); }
    if (lds[threadIdx.x ^ 1] == 1.5f) st[node].first = 0;
}

struct Shape { int wgs, threads; size_t lds; int bigarg; const char *name; };

int main() {
    CK(hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_spin_bigarg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_spin_together, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t m;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    const int N = 24;   // kernels per graph (X, Y alternating)
    Stamp *st;
    CK(hipMalloc(&st, sizeof(Stamp) * N));
    std::vector<Stamp> init(N, Stamp{~0ull, 0ull, ~0ull}), h(N);
    Pad pad;
    for (auto &c : pad.b) c = 1;
    const Shape X{141, 512, 119 * 1024, 0, "X"};
    const Shape ys[] = {
        {296, 512, 74 * 1024, 0, "296x512 74K (tiles today)"}, {144, 512, 74 * 1024, 0, "144x512 74K (actor tiles)"},
        {296, 512, 32 * 1024, 0, "296x512 32K"}, {296, 256, 32 * 1024, 0, "296x256 32K"}, {296, 512, 1024, 0, "296x512 1K"},
        {141, 512, 119 * 1024, 0, "141x512 119K (chains today)"}, {141, 512, 64 * 1024, 0, "141x512 64K"}, {141, 512, 1024, 0, "141x512 1K"},
        {141, 256, 119 * 1024, 0, "141x256 119K"}, {64, 512, 119 * 1024, 0, "64x512 119K"}, {256, 512, 119 * 1024, 0, "256x512 119K"},
        {8, 512, 119 * 1024, 0, "8x512 119K"}, {1, 64, 1024, 0, "1x64 1K"}, {141, 512, 119 * 1024, 1, "141x512 119K + 1 KB kernarg"},
    };
    printf("%-34s  gap X->Y   gap Y->X   (us; X = 141 x 512 threads x 119 KB, every kernel spins 10 us)\n", "Y shape");
    for (int together = 0; together < 2; ++together) {
        if (together) printf("-- all workgroups of a kernel end together (fixed offset from its first start) --\n");
        for (const Shape &Y : ys) {
            hipGraph_t g;
            hipGraphExec_t ex;
            CK(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) {
                const Shape &S = (i & 1) ? Y : X;
                if (S.bigarg) hipLaunchKernelGGL(k_spin_bigarg, dim3(S.wgs), dim3(S.threads), S.lds, m, st, i, 1000u, pad);
                else if (together) hipLaunchKernelGGL(k_spin_together, dim3(S.wgs), dim3(S.threads), S.lds, m, st, i, 1000u);
                else hipLaunchKernelGGL(k_spin, dim3(S.wgs), dim3(S.threads), S.lds, m, st, i, 1000u);
            }
            CK(hipStreamEndCapture(m, &g));
            CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            double gxy = 0, gyx = 0, ramp = 0;
            const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipMemcpy(st, init.data(), sizeof(Stamp) * N, hipMemcpyHostToDevice));
                CK(hipGraphLaunch(ex, m));
                CK(hipStreamSynchronize(m));
                if (r < 2) continue;
                CK(hipMemcpy(h.data(), st, sizeof(Stamp) * N, hipMemcpyDeviceToHost));
                double a = 0, b = 0, c = 0;
                int na = 0, nb = 0;
                for (int i = 4; i < N; ++i) {
                    const double gap = (double)(long long)(h[i].first - h[i - 1].last) / 100.0;
                    if (i & 1) { a += gap; ++na; c += (double)(long long)(h[i].last - h[i].first_end) / 100.0; }
                    else { b += gap; ++nb; }
                }
                gxy += a / na; gyx += b / nb; ramp += c / na;
            }
            printf("%-34s  %7.2f   %7.2f    (Y: last end - first end %.2f)\n", Y.name, gxy / reps, gyx / reps, ramp / reps);
            CK(hipGraphExecDestroy(ex));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
