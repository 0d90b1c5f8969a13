// f32_divsqrt.hip -- is hipcc's plain f32 `/` and sqrtf() correctly rounded on gfx950 (== via-f64 route)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const float *a, const float *b, int n, unsigned long long *bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i];
    float s1 = sqrtf(x), s2 = (float)__dsqrt_rn((double)x);
    float s3 = __fsqrt_rn(x);
    float d1 = x / y, d2 = (float)__ddiv_rn((double)x, (double)y);
    float d3 = __fdiv_rn(x, y);
    if (__float_as_uint(s1) != __float_as_uint(s2)) atomicAdd(&bad[0], 1ull);
    if (__float_as_uint(s3) != __float_as_uint(s2)) atomicAdd(&bad[1], 1ull);
    if (__float_as_uint(d1) != __float_as_uint(d2)) atomicAdd(&bad[2], 1ull);
    if (__float_as_uint(d3) != __float_as_uint(d2)) atomicAdd(&bad[3], 1ull);
}
int main() {
    const int n = 1 << 26;
    float *ha = new float[n], *hb = new float[n];
    uint64_t s = 88172645463325252ull;
    auto nxt = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        // wide exponent range incl. subnormals for sqrt/div operands (positive)
        uint32_t u = (uint32_t)(nxt() >> 33);  // 31 bits: positive floats incl. denormals/inf/nan filtered below
        uint32_t v = (uint32_t)(nxt() >> 33);
        if ((u >> 23) == 255) u &= 0x3fffffff;
        if ((v >> 23) == 255) v &= 0x3fffffff;
        if (i & 1) { u = (u & 0x007fffff) | ((uint32_t)(90 + (nxt() % 60)) << 23); v = (v & 0x007fffff) | ((uint32_t)(100 + (nxt() % 40)) << 23); }
        memcpy(&ha[i], &u, 4); memcpy(&hb[i], &v, 4);
    }
    float *a, *b; unsigned long long *bad, hbad[4] = {0, 0, 0, 0};
    hipMalloc(&a, n * 4ull); hipMalloc(&b, n * 4ull); hipMalloc(&bad, 32);
    hipMemcpy(a, ha, n * 4ull, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4ull, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 32);
    k<<<n / 256, 256>>>(a, b, n, bad);
    hipMemcpy(hbad, bad, 32, hipMemcpyDeviceToHost);
    printf("n=%d mismatches vs f64 route: sqrtf=%llu __fsqrt_rn=%llu div=%llu __fdiv_rn=%llu\n", n, hbad[0], hbad[1], hbad[2], hbad[3]);
    return 0;
}
