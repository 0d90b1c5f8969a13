// Per-CU streaming bandwidth for "many workgroups read the SAME buffer" (the slab engine's weight stream):
// pass 0 is cache-cold (fresh kernel), pass 1 re-reads the same bytes (L2-warm).  512-thread WGs,
// every wave keeps `D` float4 loads in flight.  Prints GB/s per WG for both passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D>
__global__ __launch_bounds__(512) void k_stream(const float4 *__restrict__ src, int n4, unsigned long long *t, float *sink) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        for (int base = 0; base < n4; base += 512 * D) {
            float4 v[D];
#pragma unroll
            for (int j = 0; j < D; ++j) v[j] = src[base + j * 512 + tid];
#pragma unroll
            for (int j = 0; j < D; ++j) acc += v[j].x + v[j].w;
        }
        __syncthreads();
        if (tid == 0) t[blockIdx.x * 2 + pass] = wall_clock64() - t0;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 48;
    const int mb = argc > 2 ? atoi(argv[2]) : 1;
    const int n4 = mb * (1 << 20) / 16;
    float4 *src; unsigned long long *t; float *sink;
    CK(hipMalloc(&src, (size_t)n4 * 16)); CK(hipMemset(src, 0, (size_t)n4 * 16));
    CK(hipMalloc(&t, nwg * 16)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned long long> h(nwg * 2);
    for (int D : {4, 16}) {
        for (int rep = 0; rep < 3; ++rep) {
            // evict: touch a big buffer? a fresh kernel boundary already invalidates L2; MALL may still hold src
            if (D == 4) hipLaunchKernelGGL(k_stream<4>, dim3(nwg), dim3(512), 0, 0, src, n4, t, sink);
            else hipLaunchKernelGGL(k_stream<16>, dim3(nwg), dim3(512), 0, 0, src, n4, t, sink);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), t, nwg * 16, hipMemcpyDeviceToHost));
        double c = 0, w = 0;
        for (int i = 0; i < nwg; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
        c /= nwg; w /= nwg;  // ticks of 10 ns
        printf("nwg=%d size=%dMB inflight/wave=%d: cold %.1f us (%.1f GB/s per WG), warm %.1f us (%.1f GB/s per WG)\n", nwg, mb, D,
               c / 100.0, mb * 1.048576e3 / (c / 100.0) , w / 100.0, mb * 1.048576e3 / (w / 100.0));
    }
    return 0;
}
