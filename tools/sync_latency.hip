// How long after a kernel's end does the host learn of it?  A ~150 us kernel, then (a) hipStreamSynchronize, (b) a spin on
// hipEventQuery, (c) hipEventSynchronize on an event created with hipEventBlockingSync: wall time from the launch call to the return,
// against the kernel's own duration by HIP events.  (what is left of apap_local_homography's 0.30 ms.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/sync_latency tools/sync_latency.hip && tools/sync_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        const hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                            \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                               \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__global__ void k_spin(long long cycles, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    int *d;
    CK(hipMalloc(&d, 4));
    hipEvent_t e0, e1, eb;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&eb, hipEventBlockingSync));
    const long long cycles = 15000;     // 100 MHz wall clock: 150 us
    for (int i = 0; i < 20; ++i) k_spin<<<256, 64>>>(cycles, d);
    CK(hipDeviceSynchronize());
    const char *names[4] = {"hipStreamSynchronize", "spin on hipEventQuery", "hipEventSynchronize (blocking-sync event)", "hipEventSynchronize (default event)"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> wall, kern;
        for (int rep = 0; rep < 40; ++rep) {
            const double t0 = now_us();
            CK(hipEventRecord(e0, nullptr));
            k_spin<<<256, 64>>>(cycles, d);
            hipEvent_t done = mode == 2 ? eb : e1;
            CK(hipEventRecord(done, nullptr));
            if (mode == 0) CK(hipStreamSynchronize(nullptr));
            else if (mode == 1) { while (hipEventQuery(done) == hipErrorNotReady) { } }
            else CK(hipEventSynchronize(done));
            wall.push_back(now_us() - t0);
            CK(hipDeviceSynchronize());
            if (mode != 2) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); kern.push_back(ms * 1e3); }
        }
        std::sort(wall.begin(), wall.end());
        std::sort(kern.begin(), kern.end());
        printf("%-44s launch -> return %7.1f us (median of 40)   kernel by events %7.1f us\n", names[mode], wall[20], kern.empty() ? 0.0 : kern[kern.size() / 2]);
    }
    return 0;
}
