// Prices the three ways to run a chain of small dependent tasks on gfx950 (round 6):
//   (1) one kernel per task, replayed as a hipGraph                        -> "boundary"
//   (2) ONE persistent launch: blocks take tickets in task order, a task waits on the done-counter of the task it
//       depends on (agent-scope release / acquire, guide: Guideline 16)    -> "counter"
//   (3) W independent chains: one launch per dependency LEVEL carrying the blocks of all W chains ("union" launch) against
//       W x T separate launches and against the persistent form.
// Task t of chain c: y[t+1][c][i] = y[t][c][i] * 1.000001 + 1 over B blocks x 1024 doubles (8 KB per block).
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_phase probe_phase.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <chrono>
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

constexpr int kPerBlock = 1024;      // doubles per block (4 per thread)

__device__ __forceinline__ void task_body(const double* __restrict__ in, double* __restrict__ out, int vb) {
    const int i = vb * kPerBlock + threadIdx.x * 4;
    const double4 v = *reinterpret_cast<const double4*>(in + i);
    double4 w;
    w.x = v.x * 1.000001 + 1.0;
    w.y = v.y * 1.000001 + 1.0;
    w.z = v.z * 1.000001 + 1.0;
    w.w = v.w * 1.000001 + 1.0;
    *reinterpret_cast<double4*>(out + i) = w;
}

// (1) / (3): `chains` chains side by side in one launch; chain c uses the slab [c * B * kPerBlock, ...)
__global__ void __launch_bounds__(256) step_kernel(const double* in, double* out, int B) {
    task_body(in, out, blockIdx.x);      // the chains are contiguous slabs: block id indexes straight into them
}

// (2): persistent form.  ctl[0] = ticket, ctl[16 + 16 * t] = done counter of task t (own cache line each).
struct Persist {
    double* y;           // [T + 1][chains * B * kPerBlock]
    unsigned* ctl;
    int T, B, chains;    // tasks per chain, blocks per task, chains
    int give_up;         // spin bound
};
__global__ void __launch_bounds__(256) persist_kernel(const Persist p) {
    __shared__ unsigned s_ticket;
    const long slab = (long)p.chains * p.B * kPerBlock;
    const unsigned total = (unsigned)p.T * p.chains * p.B;
    int last_task = -1;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(p.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned tk = s_ticket;
        __syncthreads();
        if (tk >= total) break;
        // ticket order: level-major (all chains of level t before level t + 1), so a waiting block only waits on lower tickets
        const int t = tk / (p.chains * p.B);
        const int r = tk - t * (p.chains * p.B);
        const int c = r / p.B, vb = r - c * p.B;
        const int task = t * p.chains + c;
        if (t > 0 && task != last_task) {
            if (threadIdx.x == 0) {
                const unsigned* done = p.ctl + 16 + 16 * ((t - 1) * p.chains + c);
                int spins = 0;
                while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.B) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > p.give_up) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        last_task = task;
        task_body(p.y + (long)t * slab + (long)c * p.B * kPerBlock, p.y + (long)(t + 1) * slab + (long)c * p.B * kPerBlock, vb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(p.ctl + 16 + 16 * task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int T = 32, R = 50;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int Bs[] = {2, 16, 64, 256, 1024};
    const int Ws[] = {1, 4, 8};
    printf("# T = %d dependent tasks per chain, W chains, B blocks x 8 KB per task; us per LEVEL (T levels), best of 3 x %d replays\n", T, R);
    printf("# %5s %3s | %12s %12s %12s %12s | check\n", "B", "W", "W*T launches", "T union", "persist 256", "persist 512");
    for (int W : Ws)
        for (int B : Bs) {
            const long slab = (long)W * B * kPerBlock;
            double* y;
            CK(hipMalloc(&y, sizeof(double) * slab * (T + 1)));
            std::vector<double> h(slab, 1.0);
            CK(hipMemcpy(y, h.data(), sizeof(double) * slab, hipMemcpyHostToDevice));
            unsigned* ctl;
            const size_t ctl_bytes = sizeof(unsigned) * (16 + 16 * (size_t)T * W);
            CK(hipMalloc(&ctl, ctl_bytes));
            // expected value after T steps
            double ex = 1.0;
            for (int t = 0; t < T; ++t) ex = ex * 1.000001 + 1.0;
            auto check = [&]() {
                std::vector<double> out(slab);
                CK(hipMemcpy(out.data(), y + (long)T * slab, sizeof(double) * slab, hipMemcpyDeviceToHost));
                long bad = 0;
                for (long i = 0; i < slab; ++i) bad += out[i] != ex;
                CK(hipMemset(y + slab, 0, sizeof(double) * slab * T));
                return bad;
            };
            auto time_graph = [&](hipGraphExec_t ge) {
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipGraphLaunch(ge, st));
                    CK(hipStreamSynchronize(st));
                    const double t0 = now_us();
                    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
                    CK(hipStreamSynchronize(st));
                    best = std::min(best, (now_us() - t0) / R / T);
                }
                return best;
            };
            double res[4];
            long bad[4];
            // (a) W*T separate launches (chain-major inside a level), one graph
            {
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int t = 0; t < T; ++t)
                    for (int c = 0; c < W; ++c)
                        hipLaunchKernelGGL(step_kernel, dim3(B), dim3(256), 0, st, y + (long)t * slab + (long)c * B * kPerBlock,
                                           y + (long)(t + 1) * slab + (long)c * B * kPerBlock, B);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                res[0] = time_graph(ge);
                bad[0] = check();
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
            // (b) one union launch per level
            {
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int t = 0; t < T; ++t)
                    hipLaunchKernelGGL(step_kernel, dim3(B * W), dim3(256), 0, st, y + (long)t * slab, y + (long)(t + 1) * slab, B);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                res[1] = time_graph(ge);
                bad[1] = check();
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
            // (c) persistent, 256 / 512 workgroups
            for (int v = 0; v < 2; ++v) {
                Persist p{y, ctl, T, B, W, 1 << 22};
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                CK(hipMemsetAsync(ctl, 0, ctl_bytes, st));
                hipLaunchKernelGGL(persist_kernel, dim3(v ? 512 : 256), dim3(256), 0, st, p);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                res[2 + v] = time_graph(ge);
                bad[2 + v] = check();
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
            printf("  %5d %3d | %12.2f %12.2f %12.2f %12.2f | bad %ld %ld %ld %ld\n", B, W, res[0], res[1], res[2], res[3], bad[0],
                   bad[1], bad[2], bad[3]);
            fflush(stdout);
            CK(hipFree(y));
            CK(hipFree(ctl));
        }
    return 0;
}
