// Check of the operand mapping the thin-slab engine's big layers use from round 3 on (slab8.h, s8_big_layer):
// ONE wavefront computes out[4 rows][32 columns] = x[4][256] . W[256][32] with v_mfma_f32_4x4x1 and cbsz = 3:
//   blocks 0-7  (lanes  0-31): columns 0..31, reduction indices   0..127   (A broadcast from block abid)
//   blocks 8-15 (lanes 32-63): columns 0..31, reduction indices 128..255   (A broadcast from block 8 + abid)
// and merges the two halves with v_permlane32_swap (lanes < 32 end up with rows 0,1, lanes >= 32 with rows 2,3).
// Small integers: every partial sum is exact, so the comparison with the CPU is exact whatever the order.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_cbsz3.hip -o mfma_cbsz3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float *x, const float *W, float *out) {   // x [4][256], W [256][32] (k-major), out [4][32]
    const int l = threadIdx.x, i = l & 3, blk = l >> 2, half = blk >> 3;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 16; ++j) {
        // A register j: lane (blk, i) holds x[i][half * 128 + 8 j + (blk & 7)]
        const float a = x[i * 256 + half * 128 + 8 * j + (blk & 7)];
        for (int s = 0; s < 8; ++s) {
            const int t = 8 * j + s;                                 // step: reduction index t (half 0) / 128 + t (half 1)
            const float b = W[(half * 128 + t) * 32 + (l & 31)];    // lane -> column l & 31
            switch (s) {
                case 0: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 0, 0); break;
                case 1: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 1, 0); break;
                case 2: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 2, 0); break;
                case 3: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 3, 0); break;
                case 4: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 4, 0); break;
                case 5: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 5, 0); break;
                case 6: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 6, 0); break;
                default: c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, 7, 0); break;
            }
        }
    }
    // merge: (c[0], c[2]) and (c[1], c[3])
    float v[2];
    for (int p = 0; p < 2; ++p) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[p]), __float_as_uint(c[p + 2]), false, false);
        v[p] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // lanes < 32: lo + hi of row p; lanes >= 32: of row p + 2
    }
    const int row0 = (l >> 5) * 2, col = l & 31;
    out[(row0 + 0) * 32 + col] = v[0];
    out[(row0 + 1) * 32 + col] = v[1];
}

int main() {
    std::vector<float> x(4 * 256), W(256 * 32), ref(4 * 32, 0.f), got(4 * 32);
    srand(3);
    for (auto &v : x) v = (float)(rand() % 7 - 3);
    for (auto &v : W) v = (float)(rand() % 5 - 2);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 32; ++c)
            for (int kk = 0; kk < 256; ++kk) ref[r * 32 + c] += x[r * 256 + kk] * W[kk * 32 + c];
    float *dx, *dW, *dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dout, got.size() * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dW, dout);
    if (hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    int bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
    printf("cbsz = 3 two-half mapping + permlane32_swap merge: %d of %zu outputs differ from the CPU (%s)\n", bad, got.size(), bad ? "WRONG" : "exact");
    return bad != 0;
}
