// fp64 MFMA issue rate of ONE wave per SIMD when the MFMAs form few dependent chains (the bra-dressing kernel: 2 chains of
// 13), with vector / scalar instructions in between.  hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int NCH, int NVALU, int NSALU>
__global__ void __launch_bounds__(64) loop(double* out, int iters, unsigned long long* cyc) {
    v4d c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    int v = threadIdx.x, s = iters;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            MFMA(c[m % NCH]);
#pragma unroll
            for (int q = 0; q < NVALU; ++q) asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));
#pragma unroll
            for (int q = 0; q < NSALU; ++q) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4d r = c[0] + c[1] + c[2] + c[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + v + s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH, int NVALU, int NSALU>
void run(const char* label) {
    const int blocks = 1024, iters = 20000;           // one 64-thread block per SIMD
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 64); hipMalloc(&cyc, 8 * blocks);
    loop<NCH, NVALU, NSALU><<<blocks, 64>>>(out, iters / 10, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); loop<NCH, NVALU, NSALU><<<blocks, 64>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks); hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * iters * 16 * 2048.0;
    printf("%-52s %8.3f ms %6.2f TFLOP/s  s_memtime ticks/MFMA %.2f\n", label, ms, flops / ms / 1e9, (double)c[blocks / 2] / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<4, 0, 0>("4 chains, nothing else");
    run<2, 0, 0>("2 chains, nothing else");
    run<1, 0, 0>("1 chain, nothing else");
    run<2, 2, 0>("2 chains + 2 VALU per MFMA");
    run<2, 4, 0>("2 chains + 4 VALU per MFMA");
    run<2, 8, 0>("2 chains + 8 VALU per MFMA");
    run<2, 0, 4>("2 chains + 4 SALU per MFMA");
    run<2, 4, 4>("2 chains + 4 VALU + 4 SALU per MFMA");
    run<4, 4, 4>("4 chains + 4 VALU + 4 SALU per MFMA");
    return 0;
}
