// Do FP32 MFMA and weight streaming overlap inside one CU?  512-thread WG, per wave: N rounds of
//   mode 0: 8 MFMA only            mode 1: LDS-DMA of 2 KiB (+ ds_read of it) only
//   mode 2: both in every wave (DMA ring of 12 KiB per wave, counted vmcnt)     mode 3: waves 0-3 MFMA, waves 4-7 DMA
//   mode 4: both in every wave, weights through global_load_dwordx2 -> VGPR
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int VAR>
__global__ __launch_bounds__(512) void k(const float4 *__restrict__ src, int rounds, unsigned long long *t, float *sink) {
    __shared__ __attribute__((aligned(16))) float4 ring[8][12][64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float4 a = make_float4(1.f, 2.f, 3.f, (float)lane), b0 = a, b1 = a;
    const float4 *p = src + (size_t)wave * 64 * 128 + lane;   // 8 waves x 128 KiB = 1 MiB, shared by all WGs (L2-hot)
    const bool do_mma = (MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wave < 4));
    const bool do_dma = (MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4));
    if (do_dma)
        for (int j = 0; j < 12; ++j) __builtin_amdgcn_global_load_lds(p + j * 64, &ring[wave][j][0], 16, 0, 0);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        if (do_dma) {
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            b0 = ring[wave][(2 * r) % 12][lane];
            b1 = ring[wave][(2 * r + 1) % 12][lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(VAR & 2)) {
                const int nb = (2 * r + 12) % 128;
                __builtin_amdgcn_global_load_lds(p + nb * 64, &ring[wave][(2 * r) % 12][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds(p + (nb + 1) * 64, &ring[wave][(2 * r + 1) % 12][0], 16, 0, 0);
            }
        }
        if (MODE == 4) {
            const float2 *p2 = reinterpret_cast<const float2 *>(src) + (size_t)wave * 64 * 256 + lane;
            const float2 x0 = p2[((4 * r) % 256) * 64], x1 = p2[((4 * r + 1) % 256) * 64], x2 = p2[((4 * r + 2) % 256) * 64], x3 = p2[((4 * r + 3) % 256) * 64];
            b0 = make_float4(x0.x, x0.y, x1.x, x1.y);
            b1 = make_float4(x2.x, x2.y, x3.x, x3.y);
        }
        if (do_mma) {
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c1, 0, 0, 0);
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
            if (do_dma && (VAR & 2)) {
                const int nb = (2 * r + 12) % 128;
                __builtin_amdgcn_global_load_lds(p + nb * 64, &ring[wave][(2 * r) % 12][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds(p + (nb + 1) * 64, &ring[wave][(2 * r + 1) % 12][0], 16, 0, 0);
            }
        } else {
            c0[0] += b0.x + b1.y;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) t[blockIdx.x] = wall_clock64() - t0;
    if (c0[0] + c1[1] == 1234.5f) sink[0] = c0[0];
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 48, rounds = 1024;
    float4 *src; unsigned long long *t; float *sink;
    const size_t bytes = (size_t)8 * 64 * 8192 * 16;   // 64 MiB
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 0, bytes));
    CK(hipMalloc(&t, nwg * 8)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned long long> h(nwg);
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL((k<0, 0>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 1: hipLaunchKernelGGL((k<1, 0>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 2: hipLaunchKernelGGL((k<2, 0>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 3: hipLaunchKernelGGL((k<3, 0>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 4: hipLaunchKernelGGL((k<4, 0>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 5: hipLaunchKernelGGL((k<2, 1>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 6: hipLaunchKernelGGL((k<2, 2>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
                case 7: hipLaunchKernelGGL((k<2, 3>), dim3(nwg), dim3(512), 0, 0, src, rounds, t, sink); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), t, nwg * 8, hipMemcpyDeviceToHost));
        double c = 0;
        for (int i = 0; i < nwg; ++i) c += h[i];
        c /= nwg;
        const double us = c / 100.0;
        printf("mode %d: %.1f us for %d rounds: %.1f ns/round/wave", mode, us, rounds, 1e3 * us / rounds);
        if (mode != 0 && mode != 4) printf("  (%.1f GB/s per WG)", 8.0 * rounds * 2048 / (us * 1e3) * ((mode == 3) ? 0.5 : 1.0));
        if (mode != 1) printf("  (%.1f cycles per MFMA per SIMD)", us * 2400.0 / (rounds * 8.0 * ((mode == 3) ? 1 : 2)));
        printf("\n");
    }
    return 0;
}
