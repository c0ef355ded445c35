// Microbenchmark behind the question "could K1's float64 accumulation be done exactly on the int8 matrix pipe?" (Ozaki
// splitting: w^2 and the table in 8-bit digits, int32 accumulators): how fast is v_mfma_i32_16x16x64_i8 on gfx950, and does it
// co-execute with the vector work K1 needs beside it (the float64 weight chain, integer digit extraction)?
// Per loop iteration and wave: NM independent int8 MFMAs and NF filler instructions of one kind; MFMA only, filler only, both.
// The shader clock during each run is reported (s_memtime against the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/i8_coexec.hip -o tools/i8_coexec && tools/i8_coexec
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef double double4_t __attribute__((ext_vector_type(4)));

enum Filler { NONE, FMA64, ADD_U32, PERM, MFMA64 };

template <int KIND, int NF, int NM>
__global__ __launch_bounds__(256) void k_mix(double *out, long long *clk, int iters, double a, double b) {
    int4_t acc[4] = {int4_t{0, 0, 0, 0}, int4_t{0, 0, 0, 0}, int4_t{0, 0, 0, 0}, int4_t{0, 0, 0, 0}};
    double4_t dacc = {0, 0, 0, 0};
    double xd[8];
    unsigned xi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xd[i] = 1.0 + threadIdx.x * 1e-3 + i; xi[i] = threadIdx.x * 8 + i; }
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    int4_t ia = {(int)threadIdx.x, 3, 5, 7}, ib = {1, (int)threadIdx.x, 2, 4};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int i = k & 7;
            if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(xd[i]) : "v"(av), "v"(bv));
            if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[i]) : "v"(xi[(i + 1) & 7]));
            if (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(xi[i]) : "v"(xi[(i + 1) & 7]), "v"(0x07020500u));
            if (KIND == MFMA64 && (k & 7) == 0) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, dacc, 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = dacc[0] + dacc[1] + dacc[2] + dacc[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += xd[i] + xi[i];
#pragma unroll
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int KIND, int NF, int NM>
static void run(const char *what, double *out, long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_mix<KIND, NF, NM>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.000001, 1e-9);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_mix<KIND, NF, NM>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    long long h[2];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;     // s_memrealtime ticks at 100 MHz
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double ns_per_it = ms * 1e6 / iters / waves_per_simd;                   // per wave-iteration per SIMD
    const double tops = NM ? 2.0 * 16 * 16 * 64 * NM * (double)blocks * 4 * iters / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-44s %8.3f ms  %7.1f ns per wave-iteration per SIMD  shader clock %5.0f MHz  int8 %7.1f TOP/s\n", what, ms, ns_per_it, mhz, tops);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4;      // 4 waves per SIMD
    const int iters = 20000;
    double *out;
    long long *clk;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, 16);
    printf("device %s, %d CUs; %d blocks of 4 waves (4 waves per SIMD), %d iterations\n", p.gcnArchName, p.multiProcessorCount, blocks, iters);
    run<NONE, 0, 8>("8 x mfma_i32_16x16x64_i8", out, clk, blocks, iters);
    run<FMA64, 32, 0>("32 x v_fma_f64", out, clk, blocks, iters);
    run<FMA64, 32, 8>("8 x i8 MFMA + 32 x v_fma_f64", out, clk, blocks, iters);
    run<ADD_U32, 32, 0>("32 x v_add_u32", out, clk, blocks, iters);
    run<ADD_U32, 32, 8>("8 x i8 MFMA + 32 x v_add_u32", out, clk, blocks, iters);
    run<PERM, 32, 0>("32 x v_perm_b32", out, clk, blocks, iters);
    run<PERM, 32, 8>("8 x i8 MFMA + 32 x v_perm_b32", out, clk, blocks, iters);
    run<MFMA64, 32, 0>("4 x mfma_f64_16x16x4", out, clk, blocks, iters);
    run<MFMA64, 32, 8>("8 x i8 MFMA + 4 x mfma_f64_16x16x4", out, clk, blocks, iters);
    run<FMA64, 64, 4>("4 x i8 MFMA + 64 x v_fma_f64", out, clk, blocks, iters);
    run<FMA64, 64, 0>("64 x v_fma_f64", out, clk, blocks, iters);
    return 0;
}
