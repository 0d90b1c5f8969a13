// Microbenchmark (round 4): what does it cost to run the update as TWO concurrent hipGraph branches?
// Spin kernels with the real kernels' footprints (workgroups x 512 threads, LDS bytes, duration) stand in for
//   A  = actor-side chains + spares   (85 WGs, 119 KB LDS: one per CU, 26.3 us)
//   C  = critic chains + target chains of the next update (128 WGs, 119 KB, 16 us)
//   Tc = critic weight-gradient tiles + optimizer (152 WGs, 74 KB, 7 us)
//   Ta = actor weight-gradient tiles + optimizer  (144 WGs, 74 KB, 7 us)
//   K1 = today's chain kernel (141 WGs, 119 KB, 26.8 us), K2 = today's 296 tiles (74 KB, 7 us)
// Forms, each captured as ONE graph of N updates and replayed:
//   serial   : K1 -> K2                                   (today)
//   serial3  : A+C in one launch (213 WGs) -> Ta+Tc in one launch (what a split WITHOUT overlap would cost)
//   branch   : main  A(u) -> Ta(u);  side  C(u) -> Tc(u);  Tc(u) -> A(u+1) (cross), A(u) -> C(u+1) (cross)
//              (ping-pong critic parameters: no hazard edge)
//   lite     : main  A(u) -> Ta(u);  side  C(u) -> Tc(u);  Ta(u-1) -> C(u) (fork), Tc(u) -> Ta(u) (join)
//              (in-place critic parameters: the critic's optimizer step rides in Ta)
// Every node stamps first start / last end (100 MHz wall clock) so the printed time line of one update in the middle of
// the graph shows where the boundaries go.  Build: hipcc --offload-arch=gfx950 -O3 graph_branches.hip -o graph_branches.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long first, last; };

__global__ __launch_bounds__(512) void k_spin(Stamp *st, int node, unsigned ticks) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&st[node].first, t0);
    lds[threadIdx.x] = (float)t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&st[node].last, wall_clock64());
    if (lds[threadIdx.x ^ 1] == 1.5f) st[node].first = 0;
}

struct Kern { int wgs; size_t lds; double us; };
static const Kern KA{85, 119 * 1024, 26.3}, KC{128, 119 * 1024, 16.0}, KTc{152, 74 * 1024, 7.0}, KTa{144, 74 * 1024, 7.0},
    K1{141, 119 * 1024, 26.8}, K2{296, 74 * 1024, 7.0}, KAC{213, 119 * 1024, 26.3};

static void launch(const Kern &k, hipStream_t s, Stamp *st, int node) {
    hipLaunchKernelGGL(k_spin, dim3(k.wgs), dim3(512), k.lds, s, st, node, (unsigned)(k.us * 100.0));
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 40, reps = 30;
    CK(hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t m, sd;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    Stamp *st;
    const int max_nodes = 4 * N;
    CK(hipMalloc(&st, sizeof(Stamp) * max_nodes));
    std::vector<Stamp> init(max_nodes, Stamp{~0ull, 0ull}), h(max_nodes);
    std::vector<hipEvent_t> ev(4 * N);
    for (size_t i = 0; i < ev.size(); ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const char *names[] = {"serial", "serial3", "branch", "lite"};
    for (int form = 0; form < 4; ++form) {
        hipGraph_t g;
        hipGraphExec_t ex;
        CK(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
        if (form == 0) {
            for (int u = 0; u < N; ++u) { launch(K1, m, st, 4 * u); launch(K2, m, st, 4 * u + 1); }
        } else if (form == 1) {
            for (int u = 0; u < N; ++u) { launch(KAC, m, st, 4 * u); launch(K2, m, st, 4 * u + 1); }
        } else if (form == 2) {
            // side stream joins the capture through an event recorded on main
            CK(hipEventRecord(ev[0], m));
            CK(hipStreamWaitEvent(sd, ev[0], 0));
            for (int u = 0; u < N; ++u) {
                launch(KA, m, st, 4 * u);
                CK(hipEventRecord(ev[4 * u + 1], m));            // A(u) done -> C(u+1) may start (plans / gathered inputs)
                launch(KTa, m, st, 4 * u + 1);
                launch(KC, sd, st, 4 * u + 2);
                launch(KTc, sd, st, 4 * u + 3);
                CK(hipEventRecord(ev[4 * u + 2], sd));           // Tc(u) done -> A(u+1)
                CK(hipStreamWaitEvent(m, ev[4 * u + 2], 0));
                if (u + 1 < N) CK(hipStreamWaitEvent(sd, ev[4 * u + 1], 0));
            }
        } else {
            CK(hipEventRecord(ev[0], m));
            CK(hipStreamWaitEvent(sd, ev[0], 0));
            for (int u = 0; u < N; ++u) {
                launch(KA, m, st, 4 * u);
                launch(KC, sd, st, 4 * u + 2);
                launch(KTc, sd, st, 4 * u + 3);
                CK(hipEventRecord(ev[4 * u + 2], sd));           // Tc(u) -> Ta(u) (join)
                CK(hipStreamWaitEvent(m, ev[4 * u + 2], 0));
                launch(KTa, m, st, 4 * u + 1);
                CK(hipEventRecord(ev[4 * u + 3], m));            // Ta(u) -> C(u+1) (fork)
                CK(hipStreamWaitEvent(sd, ev[4 * u + 3], 0));
            }
            CK(hipEventRecord(ev[1], sd));
            CK(hipStreamWaitEvent(m, ev[1], 0));
        }
        CK(hipStreamEndCapture(m, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ex, m));
        CK(hipStreamSynchronize(m));
        double best = 1e9, sum = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpy(st, init.data(), sizeof(Stamp) * max_nodes, hipMemcpyHostToDevice));
            const double t0 = now();
            CK(hipGraphLaunch(ex, m));
            CK(hipStreamSynchronize(m));
            const double dt = (now() - t0) * 1e6 / N;
            best = dt < best ? dt : best;
            sum += dt;
        }
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * max_nodes, hipMemcpyDeviceToHost));
        // device-side period: first start of update N/2 to first start of update N-2, per update
        const int u0 = N / 2, u1 = N - 2;
        const double period = (double)(h[4 * u1].first - h[4 * u0].first) / 100.0 / (u1 - u0);
        printf("%-8s host us/update best %.2f mean %.2f | device period %.2f us/update\n", names[form], best, sum / reps, period);
        const unsigned long long base = h[4 * u0].first;
        for (int k = 0; k < 8; ++k) {
            const int node = 4 * u0 + k;
            if (h[node].last == 0) continue;
            static const char *nn[2][4] = {{"K1/A+C", "K2", "-", "-"}, {"A", "Ta", "C", "Tc"}};
            printf("    u%+d %-6s start %+7.2f end %+7.2f\n", k / 4, nn[form >= 2][k % 4], (double)(long long)(h[node].first - base) / 100.0,
                   (double)(long long)(h[node].last - base) / 100.0);
        }
        CK(hipGraphExecDestroy(ex));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
