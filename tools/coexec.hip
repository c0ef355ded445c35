// Microbenchmark: which vector instructions co-execute with v_mfma_f64_16x16x4_f64 on gfx950?
// Each kernel issues, per loop iteration and wave, 2 independent f64 MFMAs and NF independent
// "filler" instructions of one kind.  If the filler runs on hardware the f64 MFMA does not occupy,
// time per iteration stays that of the MFMA-only loop; if it shares the pipe, the times add.
//   hipcc --offload-arch=gfx950 -O3 tools/coexec.hip -o tools/coexec && tools/coexec
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

enum Filler { NONE, FMA64, FMA32, PKFMA32, EXP32, RSQ32, SQRT32, RSQ64, CVT_F32_F64, CVT_F64_F32, ADD_U32, LDS64 };

template <int KIND, int NF, int WITH_MFMA>   // WITH_MFMA: 0 none, 1 = 2 x 16x16x4, 2 = 8 x 4x4x4_4b (same flops)
__global__ __launch_bounds__(256) void k_mix(double *out, int iters, double a, double b) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double4_t acc[2] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
    double xd[8];
    float xf[16];
    unsigned xi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xd[i] = 1.0 + threadIdx.x * 1e-3 + i; xi[i] = threadIdx.x * 8 + i; }
#pragma unroll
    for (int i = 0; i < 16; ++i) xf[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    const float af = (float)a, bf = (float)b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (WITH_MFMA == 1) acc[half] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[half], 0, 0, 0);
            if (WITH_MFMA == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[half][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[half][q], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NF / 2; ++k) {
                const int i = (half * (NF / 2) + k) & 7;
                if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(xd[i]) : "v"(av), "v"(bv));
                if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[i]) : "v"(af), "v"(bf));
                if (KIND == PKFMA32) {
                    typedef float float2_t __attribute__((ext_vector_type(2)));
                    float2_t v = {xf[2 * i], xf[2 * i + 1]};
                    const float2_t c = {af, af}, d = {bf, bf};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(d));
                    xf[2 * i] = v[0]; xf[2 * i + 1] = v[1];
                }
                if (KIND == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(xf[i]));
                if (KIND == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(xf[i]));
                if (KIND == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(xf[i]));
                if (KIND == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(xd[i]));
                if (KIND == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(xf[i]) : "v"(xd[i]));
                if (KIND == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(xd[i]) : "v"(xf[i]));
                if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[i]) : "v"(xi[(i + 1) & 7]));
                if (KIND == LDS64) {
                    double t;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((xi[i] & 1023u) * 8u));
                    asm volatile("s_waitcnt lgkmcnt(3)");
                    xd[i] = t;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += xd[i] + xi[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += xf[i];
    s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static int g_blocks, g_iters;
static double *g_out;

template <int KIND, int NF, int M = 1>
static void run(const char *name) {
    const double both = time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, NF, M>), dim3(g_blocks), dim3(256), 0, 0, g_out, g_iters, 0.5, 0.25); }, 3);
    const double alone = time_ms([&] { hipLaunchKernelGGL((k_mix<KIND, NF, 0>), dim3(g_blocks), dim3(256), 0, 0, g_out, g_iters, 0.5, 0.25); }, 3);
    const double mfma = time_ms([&] { hipLaunchKernelGGL((k_mix<NONE, 0, M>), dim3(g_blocks), dim3(256), 0, 0, g_out, g_iters, 0.5, 0.25); }, 3);
    // per SIMD: 4 waves x iterations; cycles at 2.3 GHz per iteration per wave
    const double cyc = 2.3e6 / (4.0 * g_iters);
    printf("%-14s x%2d per %s: MFMA only %7.3f ms (%4.0f cyc/it), filler only %7.3f ms (%4.0f), both %7.3f ms (%4.0f)  -> overlap %.0f %% of the smaller\n",
           name, NF, M == 1 ? "2 MFMA 16x16x4" : "8 MFMA 4x4x4_4b", mfma, mfma * cyc, alone, alone * cyc, both, both * cyc,
           100.0 * (mfma + alone - both) / (alone < mfma ? alone : mfma));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_blocks = prop.multiProcessorCount * 4;   // 4 waves per SIMD, like K1
    g_iters = 4000;
    hipMalloc(&g_out, sizeof(double) * g_blocks * 256);
    printf("device %s, %d CUs; %d blocks of 4 waves (4 waves per SIMD), %d iterations\n", prop.gcnArchName, prop.multiProcessorCount, g_blocks, g_iters);
    run<FMA64, 16>("v_fma_f64");
    run<FMA32, 16>("v_fma_f32");
    run<FMA32, 32>("v_fma_f32");
    run<PKFMA32, 16>("v_pk_fma_f32");
    run<EXP32, 8>("v_exp_f32");
    run<RSQ32, 8>("v_rsq_f32");
    run<SQRT32, 8>("v_sqrt_f32");
    run<RSQ64, 4>("v_rsq_f64");
    run<CVT_F32_F64, 16>("v_cvt_f32_f64");
    run<CVT_F64_F32, 16>("v_cvt_f64_f32");
    run<ADD_U32, 16>("v_add_u32");
    run<LDS64, 8>("ds_read_b64");
    run<FMA64, 16, 2>("v_fma_f64");
    run<FMA64, 32, 2>("v_fma_f64");
    run<FMA32, 16, 2>("v_fma_f32");
    run<RSQ64, 4, 2>("v_rsq_f64");
    run<ADD_U32, 16, 2>("v_add_u32");
    run<LDS64, 8, 2>("ds_read_b64");
    hipFree(g_out);
    return 0;
}
