// Clean fp64 MFMA issue-rate microbenchmark: 16 independent accumulators held in VGPRs by inline asm
// (no compiler-inserted accvgpr moves), optional LDS reads / VALU ops per group of 16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int NLDS, int NVALU>
__global__ void __launch_bounds__(256) loop(double* out, int iters, unsigned long long* cyc) {
    __shared__ double lds[4096];
    v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0, c8 = c0, c9 = c0, c10 = c0, c11 = c0,
        c12 = c0, c13 = c0, c14 = c0, c15 = c0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-6;
    __syncthreads();
    int addr = (threadIdx.x * 8) & 4095;
    double x = 0.0; int v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double l[NLDS > 0 ? NLDS : 1];
#pragma unroll
        for (int q = 0; q < NLDS; ++q) l[q] = lds[(addr + q * 64) & 4095];
#pragma unroll
        for (int q = 0; q < NVALU; ++q) asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));
        MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); MFMA(c4); MFMA(c5); MFMA(c6); MFMA(c7);
        MFMA(c8); MFMA(c9); MFMA(c10); MFMA(c11); MFMA(c12); MFMA(c13); MFMA(c14); MFMA(c15);
#pragma unroll
        for (int q = 0; q < NLDS; ++q) x += l[q];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4d s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x + v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NLDS, int NVALU>
void run(int blocks, int iters, const char* label) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop<NLDS, NVALU><<<blocks, 256>>>(out, iters / 10, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); loop<NLDS, NVALU><<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks); hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double waves_per_simd = blocks * 4 / 1024.0;
    double flops = (double)blocks * 4 * iters * 16 * 2048.0;
    printf("%-44s %8.3f ms %7.2f TFLOP/s (%.1f%%)  cycles/MFMA/SIMD %.2f\n", label, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 78.6,
           (double)c[blocks / 2] / (iters * 16.0) / (waves_per_simd < 1 ? 1 : waves_per_simd));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>(256, 100000, "1 wave/SIMD pure MFMA");
    run<0, 0>(512, 100000, "2 waves/SIMD pure MFMA");
    run<8, 0>(256, 100000, "1 wave/SIMD + 8 ds_read_b64 / 16 MFMA");
    run<8, 0>(512, 100000, "2 waves/SIMD + 8 ds_read_b64 / 16 MFMA");
    run<0, 8>(256, 100000, "1 wave/SIMD + 8 VALU / 16 MFMA");
    run<0, 8>(512, 100000, "2 waves/SIMD + 8 VALU / 16 MFMA");
    run<8, 8>(512, 100000, "2 waves/SIMD + 8 ds_read + 8 VALU / 16 MFMA");
    run<0, 32>(512, 100000, "2 waves/SIMD + 32 VALU / 16 MFMA");
    return 0;
}
