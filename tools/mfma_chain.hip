// Microbenchmark: does v_mfma_f64_16x16x4_f64 run faster when consecutive instructions accumulate into the SAME registers (a chain)
// than when two or more accumulators alternate?  (tools/i8_coexec.hip saw 74 cycles per instruction for one dependent chain per wave,
// tools/peak_fp64.hip 105 cycles for four independent accumulators at twice the waves.)  Issue order pinned with asm volatile; 4 waves
// per SIMD as in K1; operands vary per lane and per iteration like K1's (w^2 in (0.4, 1], table values of mixed sign).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o tools/mfma_chain.bin && tools/mfma_chain.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int PATTERN>   // 0: one chain of 8; 1: two accumulators alternating; 2: 4 on one then 4 on the other; 3: eight accumulators
__global__ __launch_bounds__(256) void k_chain(double *out, long long *clk, int iters, double a, double b) {
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double av = 0.4 + 0.6 * ((threadIdx.x * 37 % 64) / 64.0), bv = b + threadIdx.x * 1e-3 - 0.1;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) MFMA(acc[0], av, bv);
        } else if (PATTERN == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { MFMA(acc[0], av, bv); MFMA(acc[1], av, bv); }
        } else if (PATTERN == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) MFMA(acc[0], av, bv);
#pragma unroll
            for (int k = 0; k < 4; ++k) MFMA(acc[1], av, bv);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) MFMA(acc[k], av, bv);
        }
        av = av * 0.999999 + 1e-7;      // operands change between iterations (two VALU instructions per 8 MFMAs)
        bv = -bv;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int PATTERN>
static void run(const char *what, double *out, long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_chain<PATTERN>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 0.37);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_chain<PATTERN>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 0.37);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    long long h[2];
    (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;
    const double per_simd_ns = ms * 1e6 / iters / 8.0 / (blocks * 4.0 / 1024.0);      // ns per MFMA per SIMD
    printf("%-44s %8.3f ms  %6.1f ns = %5.1f cycles per MFMA per SIMD at %4.0f MHz  %5.1f TFLOP/s\n", what, ms, per_simd_ns, per_simd_ns * mhz * 1e-3, mhz,
           2048.0 * 8 * iters * blocks * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    double *out;
    long long *clk;
    (void)hipMalloc(&out, sizeof(double) * p.multiProcessorCount * 8 * 256);
    (void)hipMalloc(&clk, 16);
    for (int wps : {4, 8}) {
        const int blocks = p.multiProcessorCount * wps;
        printf("-- %d waves per SIMD\n", wps);
        run<0>("one chain of 8", out, clk, blocks, 20000);
        run<1>("two accumulators alternating (K1)", out, clk, blocks, 20000);
        run<2>("4 on one accumulator, then 4 on the other", out, clk, blocks, 20000);
        run<3>("eight accumulators", out, clk, blocks, 20000);
    }
    return 0;
}
