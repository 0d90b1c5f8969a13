// v_mfma_f32_4x4x1_16b_f32: layout check + issue rate.  D_b[i][j] += A_b[i] * B_b[j] for 16 blocks b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(float *out) {   // one wave: A lane value = 100*lane, B lane value = lane; out[lane*4+r] = c[r]
    const int lane = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), (float)(1000 * (lane + 1)), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
    f32x4 d = {0, 0, 0, 0};   // cbsz = 4: every block uses the A values held by block `abid` (= 5 here)
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), (float)(1000 * (lane + 1)), d, 4, 5, 0);
    for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = d[r];
}

__global__ __launch_bounds__(512) void k_rate(unsigned long long *t, float *sink, int rounds) {
    const int lane = threadIdx.x & 63;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float a = (float)lane, b = 1.f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
    if (c0[0] + c1[1] == 1234.5f) sink[0] = c0[0];
}

int main() {
    float *out; unsigned long long *t; float *sink;
    CK(hipMalloc(&out, 512 * 4)); CK(hipMalloc(&t, 8 * 48)); CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    std::vector<float> h(512);
    CK(hipMemcpy(h.data(), out, 2048, hipMemcpyDeviceToHost));
    // expectation: block b = lane/4, j = lane%4: c[r] = A_b[r] * B_b[j] = (4b + r + 1) * 1000 * (lane + 1)
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const float want = (float)((lane / 4) * 4 + r + 1) * 1000.f * (lane + 1);
            if (h[lane * 4 + r] != want) { if (bad < 6) printf("lane %d r %d got %g want %g\n", lane, r, h[lane * 4 + r], want); ++bad; }
        }
    printf("layout (c[r] = A_blk[r] * B_lane): %s\n", bad ? "MISMATCH" : "OK");
    bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const float want = (float)(5 * 4 + r + 1) * 1000.f * (lane + 1);
            if (h[256 + lane * 4 + r] != want) { if (bad < 6) printf("bcast lane %d r %d got %g want %g\n", lane, r, h[256 + lane * 4 + r], want); ++bad; }
        }
    printf("broadcast cbsz=4 abid=5 (c[r] = A_blk5[r] * B_lane): %s\n", bad ? "MISMATCH" : "OK");
    const int rounds = 1024;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_rate, dim3(48), dim3(512), 0, 0, t, sink, rounds); CK(hipDeviceSynchronize()); }
    unsigned long long ht[48]; CK(hipMemcpy(ht, t, 8 * 48, hipMemcpyDeviceToHost));
    double us = ht[0] / 100.0;   // 2 waves per SIMD, 16 MFMA per round per wave
    printf("4x4x1: %.1f us for %d rounds -> %.2f cycles per MFMA per SIMD (FLOP/clk/SIMD = %.1f)\n", us, rounds,
           us * 2400.0 / (rounds * 32.0), 512.0 / (us * 2400.0 / (rounds * 32.0)));
    return 0;
}
