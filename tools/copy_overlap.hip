// Do a big page-locked copy and kernels on another stream run side by side on this stack?  (The resident pipeline's device
// timeline under rocprofv3 showed the solve starting only when the 25 MB image upload - a __amd_rocclr_copyBuffer kernel - had
// finished.)  Every case: wall clock of the whole and, from HIP events, when each part started and ended, microseconds from the
// first enqueue.   hipcc --offload-arch=gfx950 -O2 -o tools/copy_overlap tools/copy_overlap.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define OK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                               \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// a kernel that keeps `blocks` workgroups busy for about `us` microseconds of arithmetic (no memory traffic)
__global__ void k_busy(float *sink, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-7f;
    if (a == 123.456f) sink[0] = a;
}

// upload by a kernel of our own: `blocks` workgroups stream a page-locked host buffer into device memory, 16 B per lane
__global__ void k_pull(const uint4 *__restrict__ host, uint4 *__restrict__ dev, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dev[i] = host[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t big = 24883200, down = 26965161 / 16 * 16, small = 512000;
    char *hA, *hC, *hT2, *dA, *dC, *dT;
    float *sink;
    OK(hipHostMalloc(&hA, big, hipHostMallocDefault));
    OK(hipHostMalloc(&hC, down, hipHostMallocDefault));
    OK(hipHostMalloc(&hT2, small, hipHostMallocDefault));
    char *hT = (char *)malloc(small);
    memset(hA, 1, big);
    memset(hC, 2, down);
    memset(hT, 3, small);
    memset(hT2, 3, small);
    OK(hipMalloc(&dA, big));
    OK(hipMalloc(&dC, down));
    OK(hipMalloc(&dT, small));
    OK(hipMalloc(&sink, 64));
    hipStream_t sa, sb, sc;
    OK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    OK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    OK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t e[12];
    for (auto &x : e) OK(hipEventCreate(&x));
    // calibrate the busy kernel to ~160 us on the whole chip
    int iters = 20000;
    for (int t = 0; t < 4; ++t) {
        OK(hipEventRecord(e[0], sb));
        hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, sb, sink, iters);
        OK(hipEventRecord(e[1], sb));
        OK(hipStreamSynchronize(sb));
        float ms;
        OK(hipEventElapsedTime(&ms, e[0], e[1]));
        iters = (int)(iters * 0.160 / ms);
    }
    struct Case {
        const char *name;
        int up, kernel, small_copy, dn;      // up: 0 none, 1 hipMemcpyAsync, 2 own kernel with 64 blocks, 3 own kernel with 256 blocks
    };                                       // small_copy: 0 none, 1 pageable, 2 page-locked; kernel / dn: 0 / 1
    const Case cases[] = {
        {"upload alone", 1, 0, 0, 0},
        {"kernel alone", 0, 1, 0, 0},
        {"download alone", 0, 0, 0, 1},
        {"upload || kernel", 1, 1, 0, 0},
        {"upload || (pageable 512 KB copy, kernel)", 1, 1, 1, 0},
        {"upload || (page-locked 512 KB copy, kernel)", 1, 1, 2, 0},
        {"download || kernel", 0, 1, 0, 1},
        {"upload || download", 1, 0, 0, 1},
        {"upload || kernel || download", 1, 1, 0, 1},
        {"own pull kernel (64 blocks) alone", 2, 0, 0, 0},
        {"own pull kernel (256 blocks) alone", 3, 0, 0, 0},
        {"own pull kernel (64 blocks) || kernel", 2, 1, 0, 0},
        {"own pull kernel (64 blocks) || (pageable 512 KB copy, kernel) || download", 2, 1, 1, 1},
    };
    for (const Case &c : cases) {
        std::vector<double> wall;
        float t_up[2] = {0, 0}, t_k[2] = {0, 0}, t_dn[2] = {0, 0};
        for (int rep = 0; rep < 9; ++rep) {
            OK(hipDeviceSynchronize());
            const double w0 = now_us();
            OK(hipEventRecord(e[0], sa));
            if (c.up) {
                OK(hipEventRecord(e[1], sa));
                if (c.up == 1)
                    OK(hipMemcpyAsync(dA, hA, big, hipMemcpyHostToDevice, sa));
                else
                    hipLaunchKernelGGL(k_pull, dim3(c.up == 2 ? 64 : 256), dim3(512), 0, sa, (const uint4 *)hA, (uint4 *)dA, big / 16);
                OK(hipEventRecord(e[2], sa));
            }
            if (c.kernel) {
                if (c.small_copy == 1) OK(hipMemcpyAsync(dT, hT, small, hipMemcpyHostToDevice, sb));
                if (c.small_copy == 2) OK(hipMemcpyAsync(dT, hT2, small, hipMemcpyHostToDevice, sb));
                OK(hipEventRecord(e[3], sb));
                hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, sb, sink, iters);
                OK(hipEventRecord(e[4], sb));
            }
            if (c.dn) {
                OK(hipEventRecord(e[5], sc));
                OK(hipMemcpyAsync(hC, dC, down, hipMemcpyDeviceToHost, sc));
                OK(hipEventRecord(e[6], sc));
            }
            OK(hipStreamSynchronize(sa));
            OK(hipStreamSynchronize(sb));
            OK(hipStreamSynchronize(sc));
            wall.push_back(now_us() - w0);
            if (c.up) {
                OK(hipEventElapsedTime(&t_up[0], e[0], e[1]));
                OK(hipEventElapsedTime(&t_up[1], e[0], e[2]));
            }
            if (c.kernel) {
                OK(hipEventElapsedTime(&t_k[0], e[0], e[3]));
                OK(hipEventElapsedTime(&t_k[1], e[0], e[4]));
            }
            if (c.dn) {
                OK(hipEventElapsedTime(&t_dn[0], e[0], e[5]));
                OK(hipEventElapsedTime(&t_dn[1], e[0], e[6]));
            }
        }
        std::sort(wall.begin(), wall.end());
        printf("%-78s wall %7.1f us", c.name, wall[wall.size() / 2]);
        if (c.up) printf("   up %6.1f..%6.1f", t_up[0] * 1e3, t_up[1] * 1e3);
        if (c.kernel) printf("   kernel %6.1f..%6.1f", t_k[0] * 1e3, t_k[1] * 1e3);
        if (c.dn) printf("   down %6.1f..%6.1f", t_dn[0] * 1e3, t_dn[1] * 1e3);
        printf("\n");
        fflush(stdout);
    }
    printf("device byte check: ");
    char probe[4];
    OK(hipMemcpy(probe, dA + big - 4, 4, hipMemcpyDeviceToHost));
    printf("%d (1 expected)\n", probe[3]);
    return 0;
}
