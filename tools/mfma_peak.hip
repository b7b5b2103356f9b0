// Micro-benchmark: attainable v_mfma_f64_16x16x4_f64 rate on this MI355X (register-only loop),
// cycles per MFMA (s_memtime) and the clock the chip holds under that load.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, unsigned long long* cyc, unsigned long long* rt) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;   // non-trivial operands
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int threads, int iters, const char* label) {
    double* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, 8 * blocks); hipMalloc(&rt, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, threads>>>(out, iters / 10, cyc, rt);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, threads>>>(out, iters, cyc, rt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks), r(blocks);
    hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, 8 * blocks, hipMemcpyDeviceToHost);
    double waves = (double)blocks * threads / 64;
    double flops = waves * (double)iters * NACC * 2048.0;
    double clk_ghz = (double)c[blocks / 2] / ((double)r[blocks / 2] * 10.0);   // memrealtime ticks at 100 MHz
    double waves_per_simd = waves / (256.0 * 4.0);
    double cyc_per_mfma_per_simd = (double)c[blocks / 2] / ((double)iters * NACC) / (waves_per_simd < 1 ? 1 : waves_per_simd);
    printf("%-34s %8.3f ms  %7.2f TFLOP/s  clock %.3f GHz  cycles/MFMA/SIMD %.1f\n", label, ms, flops / ms / 1e9, clk_ghz,
           cyc_per_mfma_per_simd);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    run<4>(256, 256, 200000, "1 wave/SIMD, 4 acc");
    run<8>(256, 256, 100000, "1 wave/SIMD, 8 acc");
    run<16>(256, 256, 50000, "1 wave/SIMD, 16 acc");
    run<16>(512, 256, 50000, "2 waves/SIMD, 16 acc");
    run<1>(256, 256, 400000, "1 wave/SIMD, 1 acc (dependent)");
    run<2>(256, 256, 400000, "1 wave/SIMD, 2 acc");
    run<16>(1024, 256, 50000, "4 waves/SIMD, 16 acc");
    return 0;
}
