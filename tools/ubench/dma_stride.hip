// Microbenchmark: what limits the operand stream of the weight-gradient tiles (gemm_lds.h ring path / dw64.h) to ~25 GB/s per
// CU when the chain kernels' weight stream (fragment-ordered, contiguous) reaches 139 GB/s per CU?
//
// The tiles read k-major operands [K rows][256 columns] in panels of 32 columns: one LDS-DMA wave instruction = 8 rows x 128 B
// at a row pitch of 1 KiB.  Variants, all with the kernel's own ring (8 waves, 4 slots of 2 KiB, 3 blocks in flight, counted
// vmcnt waits), 256 workgroups = 4 problems x 64 tiles placed like Launch::place_on_xcds:
//   row-major   : the layout of today (pitch 1 KiB, 128-B pieces)
//   panel-major : [panel][row][32 columns] -- the 8 rows of an instruction are 1 KiB contiguous
// and optionally `cold`: a writer kernel rewrites the operands (write-through stores from other XCDs) before every launch, like
// the chain kernel does.  Reported: us per launch and GB/s per CU of operand stream.
// Build: hipcc --offload-arch=gfx950 -O3 dma_stride.hip -o dma_stride.bin ; run: ./dma_stride.bin [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void dma(float *dst, const float *src) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
}

// LAYOUT 0: element (k, col) of an operand at base + k * 256 + col;  1: base + (col / 32) * K * 32 + k * 32 + col % 32
template <int LAYOUT, int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const float *ops, int K, float *sink, unsigned long long *ticks) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 8 * 512];   // up to 8 slots of 2 KiB per wave
    const int bx = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pi = (bx & 7) >> 1, slot = bx >> 3, tm = (bx & 1) * 4 + (slot >> 3), tn = slot & 7;
    const float *A = ops + (size_t)(2 * pi) * K * 256, *B = ops + (size_t)(2 * pi + 1) * K * 256;
    const int rsub = lane >> 3, chunk = lane & 7;
    const float *srcA, *srcB;
    long long step;
    if (LAYOUT == 0) {
        srcA = A + (size_t)(8 * wave + rsub) * 256 + tm * 32 + 4 * chunk;
        srcB = B + (size_t)(8 * wave + rsub) * 256 + tn * 32 + 4 * chunk;
        step = 64LL * 256;
    } else {
        srcA = A + (size_t)tm * K * 32 + (size_t)(8 * wave + rsub) * 32 + 4 * chunk;
        srcB = B + (size_t)tn * K * 32 + (size_t)(8 * wave + rsub) * 32 + 4 * chunk;
        step = 64LL * 32;
    }
    float *ring = lds + wave * (8 * 512);
    const int nblk = (K - 8 * wave + 63) >> 6;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < DEPTH - 1 && i < nblk; ++i) {
        dma(ring + (i % DEPTH) * 512, srcA + i * step);
        dma(ring + (i % DEPTH) * 512 + 256, srcB + i * step);
    }
    for (int i = 0; i < nblk; ++i) {
        const int ahead = i + DEPTH - 1;
        if (ahead < nblk) {
            dma(ring + (ahead % DEPTH) * 512, srcA + ahead * step);
            dma(ring + (ahead % DEPTH) * 512 + 256, srcB + ahead * step);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float *blk = ring + (i % DEPTH) * 512 + lane;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) acc += blk[kp * 64] * blk[256 + kp * 64];   // stands in for 4 MFMAs: one LDS read pair each
    }
    __syncthreads();
    if (tid == 0) { ticks[2 * bx] = t0; ticks[2 * bx + 1] = wall_clock64(); }
    if (acc == 1.2345f) sink[bx] = acc;
}

// rewrite the operands write-through from a different workgroup placement (what the chain kernel does before the tiles run)
__global__ void k_rewrite(float *ops, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(ops + i, v + (float)(i & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int LAYOUT, int DEPTH>
static int run(const char *name, float *ops, int K, float *sink, bool cold) {
    static unsigned long long *ticks = nullptr;
    if (!ticks) CK(hipMalloc(&ticks, 512 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n = (size_t)8 * K * 256;
    const int iters = 200;
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH>), dim3(256), dim3(512), 0, 0, ops, K, sink, ticks);
    float total = 0.f;
    if (!cold) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH>), dim3(256), dim3(512), 0, 0, ops, K, sink, ticks);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&total, e0, e1));
    } else {
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(k_rewrite, dim3(141), dim3(512), 0, 0, ops, n, (float)it);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_stream<LAYOUT, DEPTH>), dim3(256), dim3(512), 0, 0, ops, K, sink, ticks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
    }
    const double us = 1e3 * total / iters;
    std::vector<unsigned long long> ht(512);
    CK(hipMemcpy(ht.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0; double mean = 0;
    for (int b = 0; b < 256; ++b) { lo = ht[2 * b] < lo ? ht[2 * b] : lo; hi = ht[2 * b + 1] > hi ? ht[2 * b + 1] : hi; mean += 0.01 * (ht[2 * b + 1] - ht[2 * b]) / 256; }
    const double span = 0.01 * (hi - lo);
    const double bytes_per_cu = 2.0 * K * 32 * 4;   // one tile per workgroup: A panel + B panel
    printf("%-12s depth %d %s K=%4d: %7.2f us per launch by events; in-kernel: first start -> last end %6.2f us, mean workgroup %6.2f us = %6.1f GB/s per CU\n",
           name, DEPTH, cold ? "cold" : "warm", K, us, span, mean, bytes_per_cu / mean * 1e-3);
    return 0;
}

int main(int argc, char **argv) {
    float *ops, *sink;
    for (int K : {256, 512, 1024, 4096}) {
        if (argc > 1 && atoi(argv[1]) != K) continue;
        const size_t n = (size_t)8 * K * 256;
        CK(hipMalloc(&ops, n * 4)); CK(hipMalloc(&sink, 4096));
        hipLaunchKernelGGL(k_rewrite, dim3(256), dim3(512), 0, 0, ops, n, 1.f);
        for (int cold = 0; cold < 2; ++cold) {
            if (run<0, 4>("row-major", ops, K, sink, cold)) return 1;
            if (run<1, 4>("panel-major", ops, K, sink, cold)) return 1;
            if (run<0, 8>("row-major", ops, K, sink, cold)) return 1;
            if (run<1, 8>("panel-major", ops, K, sink, cold)) return 1;
        }
        CK(hipFree(ops)); CK(hipFree(sink));
    }
    return 0;
}
