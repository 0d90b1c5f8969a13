// Per-CU weight streaming as the slab kernels do it: per-wave LDS-DMA ring (R x 1 KiB), each block read back with one
// ds_read_b128 per lane.  mode 0: DMA only (no read-back); mode 1: DMA + read-back (the kernel's pattern);
// mode 2: 3 of 4 blocks by DMA + read-back, 1 of 4 straight into VGPRs (global_load_dwordx4, prefetched 2 groups ahead);
// mode 3: 2 of 4 by DMA, 2 of 4 direct.  8 waves per workgroup, every workgroup streams the same 1 MiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R 12
typedef float4 Slot[64];
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma8(f32x4 &c0, f32x4 &c1, float a0, float a1, float4 b) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b.x, c0, 4, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b.x, c1, 4, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b.y, c0, 4, 1, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b.y, c1, 4, 1, 0);
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b.z, c0, 4, 2, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b.z, c1, 4, 2, 0);
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b.w, c0, 4, 3, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b.w, c1, 4, 3, 0);
}

__device__ __forceinline__ void dma16(const void *g, void *l) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)l);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "memory", "m0");
}

// blocks of this wave: blk(t) = src + (wave * nper + t) * 64 + lane
template <int MODE>
__global__ __launch_bounds__(512) void k_stream(const float4 *__restrict__ src, int nper, unsigned long long *t, float *sink) {
    __shared__ __attribute__((aligned(16))) Slot ring[8][R];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float4 *w = src + (size_t)wave * nper * 64 + lane;
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (MODE <= 1) {
#pragma unroll
        for (int i = 0; i < R; ++i) dma16(w + (size_t)i * 64, &ring[wave][i][0]);
        for (int b = 0; b < nper; b += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
                if (MODE == 1) {
                    const float4 v = ring[wave][i][lane];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    acc += v.x + v.w;
                }
                const int nb = b + i + R;
                dma16(w + (size_t)(nb < nper ? nb : nper - 1) * 64, &ring[wave][i][0]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // groups of 4 blocks: DMA blocks through the ring, direct blocks through registers two groups ahead
        constexpr int ND = (MODE == 2) ? 1 : 2, NL = 4 - ND;
        float4 d0[ND], d1[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) { d0[j] = w[(size_t)(NL + j) * 64]; d1[j] = w[(size_t)(4 + NL + j) * 64]; }
#pragma unroll
        for (int i = 0; i < R; ++i) dma16(w + (size_t)((i / NL) * 4 + (i % NL)) * 64, &ring[wave][i][0]);
        const int ngroups = nper / 4;
        for (int g = 0; g < ngroups; g += R / NL * 1) {
#pragma unroll
            for (int gg = 0; gg < R / NL; ++gg) {
                const int grp = g + gg;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    // outstanding: up to R DMA + 2 * ND direct; oldest DMA must be done
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1 + 2 * ND) : "memory");
                    const int slot = gg * NL + i;
                    const float4 v = ring[wave][slot][lane];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    acc += v.x + v.w;
                    int ng = grp + R / NL;
                    if (ng >= ngroups) ng = ngroups - 1;
                    dma16(w + (size_t)(ng * 4 + i) * 64, &ring[wave][slot][0]);
                }
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    acc += d0[j].x + d0[j].w;       // compiler waits for d0 here
                    d0[j] = d1[j];
                    int ng = grp + 2;
                    if (ng >= ngroups) ng = ngroups - 1;
                    d1[j] = w[(size_t)(ng * 4 + NL + j) * 64];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) t[blockIdx.x] = wall_clock64() - t0;
    if (acc == 1234.5f) sink[0] = acc;
}

// mode 4: ring + read-back + 8 MFMA 4x4x1 per block on the loaded weights (register-pipelined like s8_ring_step)
// mode 5: ring + read-back, MFMAs on constant operands (no data dependence); mode 6: MFMAs only; mode 7: like 4 but the
// ds_read waits lgkmcnt(0) right away (no register pipelining)
template <int MODE>
__global__ __launch_bounds__(512) void k_mix(const float4 *__restrict__ src, int nper, unsigned long long *t, float *sink) {
    __shared__ __attribute__((aligned(16))) Slot ring[8][R];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float4 *w = src + (size_t)wave * nper * 64 + lane;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float a0 = (float)lane, a1 = (float)(lane + 1);
    float4 bc = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (MODE == 6) {
        for (int b = 0; b < nper; ++b) mma8(c0, c1, a0, a1, bc);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) dma16(w + (size_t)i * 64, &ring[wave][i][0]);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
        float4 bcur = ring[wave][0][lane];
        for (int b = 0; b < nper; b += R) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 2) : "memory");
                const float4 bnext = ring[wave][(i + 1) % R][lane];
                if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                const int nb = b + i + R;
                dma16(w + (size_t)(nb < nper ? nb : nper - 1) * 64, &ring[wave][i][0]);
                if (MODE == 5) { mma8(c0, c1, a0, a1, bc); c0[0] += bcur.x; }
                else if (MODE == 8) {   // half the matrix work per block (4-row slabs): 2 accumulators over even/odd indices
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.x, c0, 4, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.y, c1, 4, 1, 0);
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.z, c0, 4, 2, 0);
                    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.w, c1, 4, 3, 0);
                } else if (MODE == 9) {   // half the matrix work, ONE accumulator (4 dependent MFMAs per block)
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.x, c0, 4, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.y, c0, 4, 1, 0);
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.z, c0, 4, 2, 0);
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, bcur.w, c0, 4, 3, 0);
                } else mma8(c0, c1, a0, a1, bcur);
                bcur = bnext;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) t[blockIdx.x] = wall_clock64() - t0;
    if (c0[0] + c1[1] == 1234.5f) sink[0] = c0[0];
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 96;
    const int nper = 96 * 2;   // blocks per wave: 8 waves x 192 KiB = 1.5 MiB per workgroup
    const size_t bytes = (size_t)8 * nper * 1024;
    float4 *src; unsigned long long *t; float *sink;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 0, bytes));
    CK(hipMalloc(&t, nwg * 8)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned long long> h(nwg);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_stream<0>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 1: hipLaunchKernelGGL(k_stream<1>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 2: hipLaunchKernelGGL(k_stream<2>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 3: hipLaunchKernelGGL(k_stream<3>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), t, nwg * 8, hipMemcpyDeviceToHost));
        double c = 0;
        for (int i = 0; i < nwg; ++i) c += h[i];
        c /= nwg;
        const double us = c / 100.0;
        printf("nwg=%d mode=%d: %.2f us per 256 KiB -> %.1f GB/s per WG\n", nwg, mode, us * 262144.0 / bytes, bytes / us / 1e3);
    }
    for (int mode = 4; mode < 10; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            switch (mode) {
                case 4: hipLaunchKernelGGL(k_mix<4>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 5: hipLaunchKernelGGL(k_mix<5>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 6: hipLaunchKernelGGL(k_mix<6>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 7: hipLaunchKernelGGL(k_mix<7>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 8: hipLaunchKernelGGL(k_mix<8>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
                case 9: hipLaunchKernelGGL(k_mix<9>, dim3(nwg), dim3(512), 0, 0, src, nper, t, sink); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), t, nwg * 8, hipMemcpyDeviceToHost));
        double c = 0;
        for (int i = 0; i < nwg; ++i) c += h[i];
        c /= nwg;
        const double us = c / 100.0;
        printf("nwg=%d mode=%d: %.2f us per 256 KiB (32 blocks per wave)\n", nwg, mode, us * 262144.0 / bytes);
    }
    {   // 4 waves per workgroup, each streaming twice the blocks (same bytes per workgroup), 4-row MFMA work (mode 9)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_mix<9>, dim3(nwg), dim3(256), 0, 0, src, 2 * nper, t, sink);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), t, nwg * 8, hipMemcpyDeviceToHost));
        double c = 0;
        for (int i = 0; i < nwg; ++i) c += h[i];
        c /= nwg;
        printf("nwg=%d mode=9 with 4 waves x 64 blocks: %.2f us per 256 KiB\n", nwg, (c / 100.0) * 262144.0 / bytes);
    }
    return 0;
}
