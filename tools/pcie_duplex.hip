// PCIe microbenchmark for the host-buffer warp call: 25 MB up (a 4K source image) and 27 MB down (its
// canvas), each ALONE and BOTH AT ONCE on two streams, for (a) hipHostMalloc'ed pinned memory, (b) page-aligned malloc'ed
// memory registered with hipHostRegister, (c) plain pageable memory.  Separates "the link" from "registered pageable memory":
// if the two directions overlap on (a) but not on (b), a pooled pinned staging buffer is what the call needs; if they overlap
// on neither, 52 MB / (both-at-once rate) is the call's floor.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_duplex tools/pcie_duplex.hip && tools/pcie_duplex
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x)                                                                              \
    do {                                                                                   \
        const hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                            \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                               \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t up = (size_t)3840 * 2160 * 3, down = (size_t)4009 * 2242 * 3;
    void *d_up, *d_down;
    CK(hipMalloc(&d_up, up));
    CK(hipMalloc(&d_down, down));
    CK(hipMemset(d_down, 7, down));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char *kinds[3] = {"hipHostMalloc", "hipHostRegister(malloc)", "pageable"};
    for (int kind = 0; kind < 3; ++kind) {
        char *h_up = nullptr, *h_down = nullptr;
        double t_reg = 0;
        if (kind == 0) {
            CK(hipHostMalloc((void **)&h_up, up, hipHostMallocDefault));
            CK(hipHostMalloc((void **)&h_down, down, hipHostMallocDefault));
        } else {
            h_up = (char *)aligned_alloc(4096, (up + 4095) / 4096 * 4096);
            h_down = (char *)aligned_alloc(4096, (down + 4095) / 4096 * 4096);
        }
        memset(h_up, 3, up);
        memset(h_down, 0, down);
        if (kind == 1) {
            const double t0 = now_ms();
            CK(hipHostRegister(h_up, up, hipHostRegisterDefault));
            CK(hipHostRegister(h_down, down, hipHostRegisterDefault));
            t_reg = now_ms() - t0;
        }
        auto run = [&](int mode, int chunks) -> double {      // mode 0: up alone, 1: down alone, 2: both at once
            std::vector<double> ts;
            for (int rep = 0; rep < 9; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now_ms();
                for (int c = 0; c < chunks; ++c) {
                    const size_t ua = up * c / chunks, ub = up * (c + 1) / chunks, da = down * c / chunks, db = down * (c + 1) / chunks;
                    if (mode != 1) CK(hipMemcpyAsync((char *)d_up + ua, h_up + ua, ub - ua, hipMemcpyHostToDevice, s1));
                    if (mode != 0) CK(hipMemcpyAsync(h_down + da, (char *)d_down + da, db - da, hipMemcpyDeviceToHost, s2));
                }
                CK(hipStreamSynchronize(s1));
                CK(hipStreamSynchronize(s2));
                ts.push_back(now_ms() - t0);
            }
            std::sort(ts.begin(), ts.end());
            return ts[ts.size() / 2];
        };
        const double first = run(2, 1);      // includes any first-use cost of the buffers (run()'s median hides it; print the first too)
        for (int chunks : {1, 16}) {
            const double a = run(0, chunks), b = run(1, chunks), c = run(2, chunks);
            printf("%-24s %2d chunk(s): up %6.3f ms (%5.1f GB/s)  down %6.3f ms (%5.1f GB/s)  both at once %6.3f ms (%5.1f GB/s in all; "
                   "sum of the two alone %6.3f, the longer alone %6.3f)\n",
                   kinds[kind], chunks, a, up / a / 1e6, b, down / b / 1e6, c, (up + down) / c / 1e6, a + b, a > b ? a : b);
        }
        printf("%-24s median of the first both-at-once runs %6.3f ms; registration %6.3f ms; down buffer byte 0 = %d\n", kinds[kind], first,
               t_reg, (int)h_down[0]);
        fflush(stdout);
        if (kind == 0) {
            CK(hipHostFree(h_up));
            CK(hipHostFree(h_down));
        } else {
            if (kind == 1) {
                CK(hipHostUnregister(h_up));
                CK(hipHostUnregister(h_down));
            }
            free(h_up);
            free(h_down);
        }
    }
    return 0;
}
