// probe: __builtin_amdgcn_global_load_lds (LDS-DMA) semantics on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ in, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each wave copies 4 chunks of 1 KiB: global chunk g -> LDS row (g), per-lane SOURCE permutation lane^1
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 4 + j;
        const double* src = in + (size_t)blockIdx.x * 2048 + row * 128 + ((lane ^ 1) * 2);
        double* dst = smem + row * 128;            // wave-uniform LDS base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 2048; i += 256) out[(size_t)blockIdx.x * 2048 + i] = smem[i];
}
int main() {
    const int nb = 64, n = nb * 2048;
    std::vector<double> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    double *di, *dout;
    hipMalloc(&di, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<nb, 256, 2048 * 8>>>(di, dout);
    hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const int blk = i / 2048, r = (i % 2048) / 128, c = i % 128, chunk = c / 2, e = c % 2;
        const double expect = blk * 2048 + r * 128 + ((chunk ^ 1) * 2) + e;
        if (o[i] != expect) { if (bad < 5) printf("mismatch at %d: %f vs %f\n", i, o[i], expect); ++bad; }
    }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
