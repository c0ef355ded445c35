// Microbenchmark: sustained fp64 rate of the vector unit (v_fma_f64) and of the matrix
// pipe (v_mfma_f64_16x16x4_f64) on this device.  SURVEY.md 8(d) asks for the fp64 peak to
// be confirmed on the box before it is used as a roofline denominator, because the local
// micro-architecture guide lists no fp64 figure.
//   hipcc --offload-arch=gfx950 -O3 tools/peak_fp64.hip -o tools/peak_fp64 && tools/peak_fp64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// Effective shader clock during a kernel: clock64() (s_memtime, shader clock) against
// wall_clock64() (s_memrealtime, constant 100 MHz), taken by one lane around its loop.
__device__ unsigned long long g_probe[2];
struct ClockProbe {
    long long c0, w0;
    __device__ ClockProbe() : c0(clock64()), w0(wall_clock64()) {}
    __device__ void stop() const {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            g_probe[0] = (unsigned long long)(clock64() - c0);
            g_probe[1] = (unsigned long long)(wall_clock64() - w0);
        }
    }
};
static double probe_mhz() {
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h));
    return h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a, double b) {
    double x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const ClockProbe probe;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = fma(x[i], a, b);
    }
    probe.stop();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a, double b) {
    double4_t acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    const ClockProbe probe;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    probe.stop();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the small f64 MFMA: 4 blocks of 4x4x4, one accumulator double per lane (512 flops)
template <int ACC>
__global__ __launch_bounds__(256) void k_mfma4(double *out, int iters, double a, double b) {
    double acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = 0.0;
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    const ClockProbe probe;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
    probe.stop();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one MFMA (1024 lane-FMAs) + 16 independent v_fma_f64 (1024 lane-FMAs) per iteration:
// if the f64 matrix pipe and the f64 vector ALU were separate units the pair would take
// max(), if they share the DP-FMA hardware it takes the sum.
__global__ __launch_bounds__(256) void k_mixed(double *out, int iters, double a, double b) {
    double4_t acc[2] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    const ClockProbe probe;
    for (int it = 0; it < iters; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[1], 0, 0, 0);
#pragma unroll
        for (int i = 8; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    probe.stop();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch, int reps) {
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

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    {
        constexpr int ACC = 8;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999999, 1e-9); }, 5);
        const double flops = 2.0 * ACC * iters * (double)blocks * threads;
        printf("v_fma_f64   (8 chains/lane, %d waves): %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz\n", blocks * 4, ms, flops / ms / 1e9, probe_mhz());
    }
    {
        constexpr int ACC = 4;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5, 0.25); }, 5);
        const double flops = 2.0 * 16 * 16 * 4 * ACC * iters * (double)blocks * (threads / 64);
        printf("v_mfma_f64_16x16x4_f64 (4 acc/wave, %d waves): %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz\n", blocks * 4, ms, flops / ms / 1e9, probe_mhz());
    }
    {
        constexpr int ACC = 1;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5, 0.25); }, 5);
        const double flops = 2.0 * 16 * 16 * 4 * ACC * iters * (double)blocks * (threads / 64);
        printf("v_mfma_f64_16x16x4_f64 (1 dependent acc/wave, 8 waves/SIMD): %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz\n", ms, flops / ms / 1e9, probe_mhz());
    }
    {
        constexpr int ACC = 8;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma4<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5, 0.25); }, 5);
        const double flops = 2.0 * 4 * 4 * 4 * 4 * ACC * iters * (double)blocks * (threads / 64);
        printf("v_mfma_f64_4x4x4_4b_f64 (8 acc/wave, %d waves): %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz\n", blocks * 4, ms,
               flops / ms / 1e9, probe_mhz());
    }
    {
        const int it2 = iters / 2;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(threads), 0, 0, out, it2, 0.999999, 1e-9); }, 5);
        const double f_mfma = 2.0 * 16 * 16 * 4 * 2 * it2 * (double)blocks * (threads / 64);
        const double f_valu = 2.0 * 16 * it2 * (double)blocks * threads;
        printf("mixed 2 MFMA + 16 v_fma_f64 per iteration: %.3f ms  MFMA part %.2f + VALU part %.2f = %.2f TFLOP/s  shader clock %.0f MHz\n", ms,
               f_mfma / ms / 1e9, f_valu / ms / 1e9, (f_mfma + f_valu) / ms / 1e9, probe_mhz());
    }
    {
        // light load: one wave per CU, one dependent chain
        hipLaunchKernelGGL(k_fma<1>, dim3(prop.multiProcessorCount), dim3(64), 0, 0, out, 1 << 18, 0.999999, 1e-9);
        (void)hipDeviceSynchronize();
        printf("light load (1 wave per CU, dependent v_fma_f64 chain): shader clock %.0f MHz\n", probe_mhz());
    }
    hipFree(out);
    return 0;
}
