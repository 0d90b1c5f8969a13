// kernarg_preload.hip -- what the FIRST kernel-argument fetch of a workgroup costs, and whether this stack preloads leading
// scalar arguments into SGPRs (gfx942+ "kernarg preload": -mllvm -amdgpu-kernarg-preload-count=N, at most 14 dwords here).
// Every launch has a fresh argument block (cold scalar cache).  Per workgroup: ticks of the 100 MHz wall clock from the first
// instruction until (a) a value selected from the 7 leading scalar arguments is in a register, (b) a field at the END of a 2.6 KB
// by-value struct argument is.  Build twice: with and without the flag.
//   hipcc --offload-arch=gfx950 -O3 kernarg_preload.hip -o kernarg_preload_off.bin
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 kernarg_preload.hip -o kernarg_preload_on.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { unsigned long long pad[320]; unsigned long long tail; };
__global__ void k(unsigned long long a0, unsigned long long a1, unsigned long long a2, unsigned long long a3, unsigned long long a4,
                  unsigned long long a5, unsigned long long a6, const Big B, unsigned long long *out) {
    const unsigned long long t = wall_clock64();
    const int x = blockIdx.x % 7;
    unsigned long long m = x == 0 ? a0 : x == 1 ? a1 : x == 2 ? a2 : x == 3 ? a3 : x == 4 ? a4 : x == 5 ? a5 : a6;
    asm volatile("" : "+s"(m));
    const unsigned long long t1 = wall_clock64();
    unsigned long long v = B.tail;
    asm volatile("" : "+s"(v));
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t; out[3 * blockIdx.x + 1] = t2 - t; out[3 * blockIdx.x + 2] = m + v; }
}
int main() {
    unsigned long long *out; CK(hipMalloc(&out, 256 * 3 * 8));
    Big B{}; B.tail = 7;
    std::vector<unsigned long long> a, b, h(768);
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, 1ull + rep, 2ull, 3ull, 4ull, 5ull, 6ull, 7ull, B, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, 768 * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i) { a.push_back(h[3 * i]); b.push_back(h[3 * i + 1]); }
        if (h[2] != (1ull + rep) + 7ull) { printf("wrong value\n"); return 1; }
    }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("leading scalar argument ready after: median %.2f us (p10 %.2f, p90 %.2f) | field at the end of the struct: median %.2f us (p10 %.2f, p90 %.2f)\n",
           a[a.size() / 2] / 100.0, a[a.size() / 10] / 100.0, a[a.size() * 9 / 10] / 100.0, b[b.size() / 2] / 100.0, b[b.size() / 10] / 100.0,
           b[b.size() * 9 / 10] / 100.0);
    return 0;
}
