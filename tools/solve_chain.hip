// Microbenchmark: the inverse-iteration step of K2's tail (27 + 27 FMAs of the two triangular solves, 9 scalings,
// norm, dot, rsq + 2 Newton steps, rescale, max-change) as a hot loop at one wave per SIMD - how many cycles does
// one step take when nothing else is going on?  Compared with the same step inside k_solve_small
// (tools/trace_small.py: ~1200-1300 cycles).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/solve_chain.hip -o tools/solve_chain && tools/solve_chain
#include <hip/hip_runtime.h>

#include <cstdio>

__host__ __device__ constexpr int low(int i, int j) { return i * (i - 1) / 2 + j; }
__host__ __device__ constexpr bool lnz(int i, int k) { return !(i >= 3 && i <= 5 && k <= 2); }

__global__ __launch_bounds__(256) void k_step(const double *lin, double *out, long long *cycles, int iters, int lanes) {
    if ((int)(threadIdx.x & 63) >= lanes) return;
    double l[36], rd[9], v[9];
    for (int k = 0; k < 36; ++k) l[k] = lin[k] * (1.0 + 1e-3 * threadIdx.x);
    for (int k = 0; k < 9; ++k) { rd[k] = lin[36 + k]; v[k] = 1.0 / 3.0; }
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        double y[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double sum = v[i];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k < i && lnz(i, k)) sum = fma(-l[low(i, k)], y[k], sum);
            y[i] = sum;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) y[i] *= rd[i];
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            double sum = y[i];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k > i && lnz(k, i)) sum = fma(-l[low(k, i)], y[k], sum);
            y[i] = sum;
        }
        double nrm2 = 0.0, dot = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            nrm2 = fma(y[i], y[i], nrm2);
            dot = fma(y[i], v[i], dot);
        }
        double r = __builtin_amdgcn_rsq(nrm2);
        const double hx = 0.5 * nrm2;
        r = fma(r, fma(-hx * r, r, 0.5), r);
        r = fma(r, fma(-hx * r, r, 0.5), r);
        const double scale = copysign(r, dot);
        double change = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double yn = y[i] * scale;
            change = fmax(change, fabs(yn - v[i]));
            v[i] = yn;
        }
        if (__all(change < -1.0)) break;   // never: keeps the vote and the branch of the real loop
    }
    const long long t1 = clock64();
    double s = 0.0;
    for (int k = 0; k < 9; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    double h[45];
    for (int k = 0; k < 36; ++k) h[k] = 0.01 * (k % 7) - 0.02;
    for (int k = 0; k < 9; ++k) h[36 + k] = 1.0 + 0.1 * k;
    double *d_l, *d_out;
    long long *d_c, c;
    (void)hipMalloc(&d_l, sizeof(h));
    (void)hipMalloc(&d_out, sizeof(double) * 256 * 1024);
    (void)hipMalloc(&d_c, sizeof(long long));
    (void)hipMemcpy(d_l, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int blocks : {1, 256}) {
        for (int lanes : {16, 64}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, 0, d_l, d_out, d_c, iters, lanes);
            (void)hipMemcpy(&c, d_c, sizeof(c), hipMemcpyDeviceToHost);
            printf("%3d block(s) of 4 waves (one per SIMD), %2d lanes active: %.0f cycles per inverse-iteration step\n", blocks, lanes,
                   (double)c / iters);
        }
    }
    // a short run, as in the kernel: 4 steps only, from a cold start
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_step, dim3(25), dim3(256), 0, 0, d_l, d_out, d_c, 4, 16);
        (void)hipMemcpy(&c, d_c, sizeof(c), hipMemcpyDeviceToHost);
        printf("25 blocks, 16 lanes, 4 steps only: %.0f cycles per step\n", (double)c / 4);
    }
    return 0;
}
