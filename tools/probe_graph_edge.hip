// What does the END (and the START) of a replayed hipGraph cost on the stream it is launched into?  (round 6)
// Every kernel stamps the 100-MHz wall clock at its first and last instruction; the host is kept ahead of the device by a
// 2-ms spinning kernel in front of each sequence, so the gaps are the device's, not the interpreter's.
//   A: spin | k k k k            (all eager)                     gap k3 -> k4
//   B: spin | graph{k k k} | k   (graph, then an eager launch)   gap graph's last -> eager k
//   C: spin | k | graph{k k k}   (eager, then a graph)           gap eager k -> graph's first
//   D: spin | graph{k k k} | graph{k k k}                        gap graph -> graph
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_graph_edge probe_graph_edge.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e__ = (x);                                                                  \
        if (e__ != hipSuccess) {                                                               \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__);    \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__global__ void __launch_bounds__(256) stamp_kernel(unsigned long long* stamps, int slot, double* data, long spin_ticks) {
    const unsigned long long t0 = wall_clock64();
    // a little real work so that the launch has something to write back at its end (20 us of streaming at ~4 MB)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    data[i] = data[i] * 1.000001 + 1.0;
    if (spin_ticks > 0 && blockIdx.x == 0 && threadIdx.x == 0)
        while ((long)(wall_clock64() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[2 * slot] = t0;
        stamps[2 * slot + 1] = wall_clock64();
    }
}

int main() {
    const int blocks = 2048;                       // 512 K doubles = 4 MB read + written per launch
    double* data;
    unsigned long long* stamps;
    CK(hipMalloc(&data, sizeof(double) * blocks * 256));
    CK(hipMemset(data, 0, sizeof(double) * blocks * 256));
    CK(hipHostMalloc(&stamps, sizeof(unsigned long long) * 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto k = [&](int slot, long spin) { hipLaunchKernelGGL(stamp_kernel, dim3(blocks), dim3(256), 0, st, stamps, slot, data, spin); };

    // two graphs of three launches each (slots 1..3 and 5..7)
    hipGraph_t g[2];
    hipGraphExec_t ge[2];
    for (int w = 0; w < 2; ++w) {
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 3; ++i) k(1 + 4 * w + i, 0);
        CK(hipStreamEndCapture(st, &g[w]));
        CK(hipGraphInstantiate(&ge[w], g[w], nullptr, nullptr, 0));
    }
    const long two_ms = 200000;                    // ticks of the 100-MHz clock
    const int reps = 40;
    const char* names[4] = {"A eager -> eager", "B graph -> eager", "C eager -> graph", "D graph -> graph"};
    for (int form = 0; form < 4; ++form) {
        std::vector<double> gaps, inner;
        for (int r = 0; r < reps + 5; ++r) {
            k(0, two_ms);
            int a_end, b_begin;                    // slots whose (end, begin) bracket the edge of interest
            switch (form) {
                case 0: k(1, 0); k(2, 0); k(3, 0); k(4, 0); a_end = 3; b_begin = 4; break;
                case 1: CK(hipGraphLaunch(ge[0], st)); k(4, 0); a_end = 3; b_begin = 4; break;
                case 2: k(4, 0); CK(hipGraphLaunch(ge[1], st)); a_end = 4; b_begin = 5; break;
                default: CK(hipGraphLaunch(ge[0], st)); CK(hipGraphLaunch(ge[1], st)); a_end = 3; b_begin = 5; break;
            }
            CK(hipStreamSynchronize(st));
            if (r < 5) continue;
            gaps.push_back((double)(stamps[2 * b_begin] - stamps[2 * a_end + 1]) / 100.0);
            const int in_a = (form == 2) ? 5 : 1;  // an edge INSIDE a graph / between plain eager launches, for reference
            inner.push_back((double)(stamps[2 * (in_a + 1)] - stamps[2 * in_a + 1]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        std::sort(inner.begin(), inner.end());
        printf("%s: edge gap median %.1f us (min %.1f, max %.1f); an inner edge of the same sequence: median %.1f us\n", names[form],
               gaps[gaps.size() / 2], gaps.front(), gaps.back(), inner[inner.size() / 2]);
    }
    return 0;
}
