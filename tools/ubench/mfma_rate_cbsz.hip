#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CB>
__global__ __launch_bounds__(512) void k_rate(unsigned long long *t, float *sink, int rounds) {
    const int lane = threadIdx.x & 63;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float a = (float)lane, b = 1.f, a2 = a + 1.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        // like s8_mma with 2 row groups: two independent accumulators alternate, abid walks 0..7
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 1, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 1, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 2, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 2, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 3, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 3, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 4, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 4, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 5, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 5, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 6, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 6, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 7, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b, c1, CB, 7, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (c0[0] + c1[1] == 1234.5f) sink[0] = c0[0];
}
template <int CB>
__global__ __launch_bounds__(512) void k_rate1(unsigned long long *t, float *sink, int rounds) {   // ONE accumulator (4-row slabs)
    const int lane = threadIdx.x & 63;
    f32x4 c0 = {0, 0, 0, 0};
    float a = (float)lane, b = 1.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 1, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 2, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 3, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 4, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 5, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 6, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, CB, 7, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (c0[0] == 1234.5f) sink[0] = c0[0];
}
int main() {
    unsigned long long *t; float *sink; hipMalloc(&t, 8 * 256); hipMalloc(&sink, 4);
    const int rounds = 2048; unsigned long long h[256];
#define RUN(K, CB, per) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(K<CB>, dim3(256), dim3(512), 0, 0, t, sink, rounds); hipDeviceSynchronize(); } \
    hipMemcpy(h, t, 8 * 256, hipMemcpyDeviceToHost); printf(#K " cbsz=%d: %.2f shader cycles per MFMA per SIMD (2 waves per SIMD)\n", CB, (double)h[0] / (2.0 * rounds * per));
    RUN(k_rate, 4, 16) RUN(k_rate, 3, 16) RUN(k_rate, 0, 16) RUN(k_rate1, 4, 8) RUN(k_rate1, 3, 8) RUN(k_rate1, 0, 8)
    return 0;
}
