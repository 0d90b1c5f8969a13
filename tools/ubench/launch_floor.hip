// Microbenchmark: cost of a dependent kernel boundary on this box, eager vs hipGraph,
// trivial kernel vs a kernel that touches a few MB.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_trivial(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_touch(float *a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0001f + 1.f; }
struct Big { char pad[768]; };
__global__ void k_pingpong(const float *__restrict__ src, float *__restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[(i * 7 + 13) % n] + 1.f; }
__global__ void k_onewg(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c) {
    __shared__ float sh[4];
    float v = a[threadIdx.x * 16] + b[threadIdx.x * 16];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) c[0] = sh[0] + sh[1] + sh[2] + sh[3];
    c[threadIdx.x * 16 + 16] = v;
}
// straight-line code of controllable size: REP dependent FMAs (each ~8 bytes of code), one thread
template <int REP> __global__ void k_codesize(float *p) {
    float v = p[0];
#pragma unroll
    for (int i = 0; i < REP; ++i) v = __builtin_fmaf(v, 1.0001f + i * 1e-7f, 0.5f);
    if (threadIdx.x == 0) p[0] = v;
}
// reads `n4` float4 (clean lines into L2), writes one float per block
__global__ void k_bigread(const float4 *__restrict__ src, int n4, float *out) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.f) out[blockIdx.x] = acc;
}
// writes n floats (dirty lines)
__global__ void k_bigwrite(float *dst, int n, float v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = v;
}
struct Ptrs { const float *in[8]; float *out; };
__global__ void k_manybuf(Ptrs p) {
    float v = 0.f;
    for (int k = 0; k < 8; ++k) v += p.in[k][threadIdx.x];
    p.out[threadIdx.x] = v;
}
__global__ void k_bigarg(const Big b, int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.pad[5]; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    void *pinned = nullptr;
    if (mode & 1) { CK(hipHostMalloc(&pinned, 1 << 20, hipHostMallocDefault)); printf("pinned host buffer allocated\n"); }
    hipEvent_t ev; if (mode & 2) { CK(hipEventCreate(&ev)); printf("timing event created\n"); }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    float *a; int n = 292608; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    Big big; memset(&big, 1, sizeof(big));
    const int N = 2000;
    float *b2; CK(hipMalloc(&b2, n * 4)); CK(hipMemset(b2, 0, n * 4));
    Ptrs sep, sep2, one, one2; float *arena; CK(hipMalloc(&arena, 32 << 20)); CK(hipMemset(arena, 0, 32 << 20));
    for (int k = 0; k < 8; ++k) { float *q; CK(hipMalloc(&q, 65536)); CK(hipMemset(q, 0, 65536)); sep.in[k] = q; sep2.in[k] = q; one.in[k] = arena + k * 16384; one2.in[k] = arena + k * 16384; }
    { float *q; CK(hipMalloc(&q, 65536)); sep.out = q; CK(hipMalloc(&q, 65536)); sep2.out = q; one.out = arena + 9 * 16384; one2.out = arena + 10 * 16384; }
    sep2.in[0] = sep.out; one2.in[0] = one.out;
    float *bigbuf; CK(hipMalloc(&bigbuf, 64 << 20)); CK(hipMemset(bigbuf, 0, 64 << 20));
    for (int variant = 0; variant < 20; ++variant) {
        auto launch = [&](int i) {
            switch (variant) {
                case 0: hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d); break;
                case 1: hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, s, d); break;
                case 2: hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, s, a, n); break;
                case 3: hipLaunchKernelGGL(k_bigarg, dim3(192), dim3(256), 0, s, big, d); break;
                case 4: if (i & 1) hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(256), 0, s, a, b2, 65536);
                        else hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(256), 0, s, b2, a, 65536); break;
                case 5: if (i & 1) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d);
                        else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, a, 65536); break;
                case 6: if (i & 1) hipLaunchKernelGGL(k_onewg, dim3(1), dim3(256), 0, s, a, a + 8192, b2);
                        else hipLaunchKernelGGL(k_onewg, dim3(1), dim3(256), 0, s, b2, b2 + 8192, a); break;
                case 8: if (i & 1) hipLaunchKernelGGL(k_manybuf, dim3(1), dim3(256), 0, s, sep); else hipLaunchKernelGGL(k_manybuf, dim3(1), dim3(256), 0, s, sep2); break;
                case 9: if (i & 1) hipLaunchKernelGGL(k_manybuf, dim3(1), dim3(256), 0, s, one); else hipLaunchKernelGGL(k_manybuf, dim3(1), dim3(256), 0, s, one2); break;
                case 10: hipLaunchKernelGGL(k_codesize<64>, dim3(1), dim3(64), 0, s, a); break;
                case 11: hipLaunchKernelGGL(k_codesize<256>, dim3(1), dim3(64), 0, s, a); break;
                case 12: hipLaunchKernelGGL(k_codesize<1024>, dim3(1), dim3(64), 0, s, a); break;
                case 13: hipLaunchKernelGGL(k_codesize<1024>, dim3(256), dim3(256), 0, s, a); break;
                case 14: hipLaunchKernelGGL(k_bigread, dim3(1024), dim3(256), 0, s, (const float4 *)bigbuf, (4 << 20) / 16, b2); break;              // 4 MB clean
                case 15: if (i & 1) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d); else hipLaunchKernelGGL(k_bigread, dim3(1024), dim3(256), 0, s, (const float4 *)bigbuf, (4 << 20) / 16, b2); break;
                case 16: hipLaunchKernelGGL(k_bigwrite, dim3(256), dim3(256), 0, s, bigbuf, (1 << 20) / 4, 1.f); break;                                     // 1 MB dirty
                case 17: if (i & 1) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d); else hipLaunchKernelGGL(k_bigwrite, dim3(256), dim3(256), 0, s, bigbuf, (1 << 20) / 4, 1.f); break;
                case 18: if (i & 1) hipLaunchKernelGGL(k_onewg, dim3(1), dim3(256), 0, s, bigbuf, bigbuf + 8192, b2); else hipLaunchKernelGGL(k_bigwrite, dim3(256), dim3(256), 0, s, bigbuf, (1 << 20) / 4, 1.f); break;
                case 19: if (i & 1) hipLaunchKernelGGL(k_onewg, dim3(1), dim3(256), 0, s, bigbuf, bigbuf + 8192, b2); else hipLaunchKernelGGL(k_bigread, dim3(1024), dim3(256), 0, s, (const float4 *)bigbuf, (32 << 20) / 16, b2); break;
                case 7: if (i % 3 == 0) hipLaunchKernelGGL(k_pingpong, dim3(1024), dim3(256), 0, s, a, b2, 262144);
                        else if (i % 3 == 1) hipLaunchKernelGGL(k_onewg, dim3(1), dim3(256), 0, s, b2, b2 + 8192, a);
                        else hipLaunchKernelGGL(k_pingpong, dim3(1024), dim3(256), 0, s, b2, a, 262144); break;
            }
        };
        for (int i = 0; i < 200; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        double eager = (now() - t0) / N * 1e6;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 400; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = now();
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double graph = (now() - t0) / (5 * 400) * 1e6;
        printf("variant %d: eager %.2f us/kernel, graph %.2f us/kernel\n", variant, eager, graph);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
