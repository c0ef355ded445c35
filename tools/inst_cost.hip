// Microbenchmark: issue cost of the individual instructions K1 is made of, in SIMD
// cycles per wave64 instruction at full occupancy (8 waves per SIMD, independent
// operands), plus the same for a single wave per SIMD.  Used to budget the fp64 pipe
// (DESIGN.md section 3).
//   hipcc --offload-arch=gfx950 -O3 tools/inst_cost.hip -o tools/inst_cost && tools/inst_cost
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// One kernel per instruction: 64 copies per loop iteration on 8 rotating register sets.
#define KERNEL_D(name, INSTR)                                                                   \
    __global__ __launch_bounds__(256) void name(double *out, int iters, double a, double b) {   \
        double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7,  \
               x7 = a * 8;                                                                      \
        int e = 1;                                                                              \
        for (int it = 0; it < iters; ++it) {                                                    \
            REP8(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)           \
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)  \
                              : "v"(b), "v"(e));)                                               \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;     \
    }

#define I_FMA(k) "v_fma_f64 %" #k ", %" #k ", %8, %8\n"
#define I_MUL(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define I_ADD(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define I_MAX(k) "v_max_f64 %" #k ", %" #k ", %8\n"
#define I_RSQ(k) "v_rsq_f64 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f64 %" #k ", %" #k "\n"
#define I_RNDNE(k) "v_rndne_f64 %" #k ", %" #k "\n"
#define I_LDEXP(k) "v_ldexp_f64 %" #k ", %" #k ", %9\n"
KERNEL_D(k_fma, I_FMA)
// dependent chains: every instruction reads the result of the one before it (what K2's tail, one wave per SIMD, is made of)
#define I_FMA_DEP(k) "v_fma_f64 %0, %0, %8, %8\n"
#define I_RCP_DEP(k) "v_rcp_f64 %0, %0\n"
#define I_FMA_DEP2(k) "v_fma_f64 %" #k ", %" #k ", %8, %8\n"  /* used with k & 1: two interleaved chains */
KERNEL_D(k_fma_dep, I_FMA_DEP)
KERNEL_D(k_rcp_dep, I_RCP_DEP)
__global__ __launch_bounds__(256) void k_fma_dep2(double *out, int iters, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2;
    for (int it = 0; it < iters; ++it) {
        REP8(asm volatile("v_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\nv_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\n"
                          "v_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\nv_fma_f64 %0, %0, %2, %2\nv_fma_f64 %1, %1, %2, %2\n"
                          : "+v"(x0), "+v"(x1)
                          : "v"(b));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1;
}
KERNEL_D(k_mul, I_MUL)
KERNEL_D(k_add, I_ADD)
KERNEL_D(k_max, I_MAX)
KERNEL_D(k_rsq, I_RSQ)
KERNEL_D(k_rcp, I_RCP)
KERNEL_D(k_sqrt, I_SQRT)
KERNEL_D(k_rndne, I_RNDNE)
KERNEL_D(k_ldexp, I_LDEXP)

// 32-bit destination / source forms
#define KERNEL_F(name, INSTR)                                                                   \
    __global__ __launch_bounds__(256) void name(double *out, int iters, double a, double b) {   \
        float x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7,   \
              x7 = a * 8;                                                                       \
        float fb = (float)b;                                                                    \
        for (int it = 0; it < iters; ++it) {                                                    \
            REP8(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)           \
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)  \
                              : "v"(fb));)                                                      \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;     \
    }
#define I_FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define I_ADDU32(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_EXP32(k) "v_exp_f32 %" #k ", %" #k "\n"
#define I_RSQ32(k) "v_rsq_f32 %" #k ", %" #k "\n"
KERNEL_F(k_fma32, I_FMA32)
KERNEL_F(k_addu32, I_ADDU32)
KERNEL_F(k_xor, I_XOR)
KERNEL_F(k_exp32, I_EXP32)
KERNEL_F(k_rsq32, I_RSQ32)

// conversions: double -> int/float and back
__global__ __launch_bounds__(256) void k_cvt_i32_f64(double *out, int iters, double a, double b) {
    double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4;
    int y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int it = 0; it < iters; ++it) {
        REP8(asm volatile("v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                          "v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                          : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3)
                          : "v"(x0), "v"(x1), "v"(x2), "v"(x3));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3;
}
__global__ __launch_bounds__(256) void k_cvt_f64_f32(double *out, int iters, double a, double b) {
    float x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4;
    double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int it = 0; it < iters; ++it) {
        REP8(asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                          "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                          : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3)
                          : "v"(x0), "v"(x1), "v"(x2), "v"(x3));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3;
}

typedef void (*kern_t)(double *, int, double, double);

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * cus * 8 * 256);
    struct {
        const char *name;
        kern_t k;
    } list[] = {{"v_fma_f64", k_fma},     {"v_fma_f64 chain", k_fma_dep}, {"v_fma_f64 2 chains", k_fma_dep2}, {"v_rcp_f64 chain", k_rcp_dep},
                {"v_mul_f64", k_mul},         {"v_add_f64", k_add},         {"v_max_f64", k_max},
                {"v_rsq_f64", k_rsq},     {"v_rcp_f64", k_rcp},         {"v_sqrt_f64", k_sqrt},       {"v_rndne_f64", k_rndne},
                {"v_ldexp_f64", k_ldexp}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
                {"v_fma_f32", k_fma32},   {"v_add_u32", k_addu32},      {"v_xor_b32", k_xor},         {"v_exp_f32", k_exp32},
                {"v_rsq_f32", k_rsq32}};
    const int iters = 2000;
    printf("%-20s %14s %14s   (ns per wave-instruction per SIMD; cycles at 2.4 GHz)\n", "instruction", "8 waves/SIMD",
           "1 wave/SIMD");
    for (auto &e : list) {
        double res[2];
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = cus * (mode == 0 ? 8 : 1);  // 4 waves per block -> 8 or 1 waves per SIMD
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.999);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.999);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            const double inst_per_simd = 64.0 * iters * (mode == 0 ? 8 : 1);
            res[mode] = ms * 1e6 / inst_per_simd;
        }
        printf("%-20s %8.2f ns %5.1f cyc %8.2f ns %5.1f cyc\n", e.name, res[0], res[0] * 2.4, res[1], res[1] * 2.4);
    }
    (void)hipFree(out);
    return 0;
}
