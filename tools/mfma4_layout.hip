// Operand layout of v_mfma_f64_4x4x4_4b_f64 on this device, found by probing: A = 1 in one
// lane, B = 1 in one lane, which lane of D becomes 1?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_layout.hip -o tools/mfma4_layout && tools/mfma4_layout
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void probe(int *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = lane;
        }
    }
}

int main() {
    int *d_out, h[64 * 64];
    (void)hipMalloc(&d_out, sizeof(h));
    (void)hipMemset(d_out, 0xff, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out);
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it pairs with and the D lanes that result
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) printf("  B%-2d->D%-2d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
