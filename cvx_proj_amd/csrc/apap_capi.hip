// Host-buffer half of the C ABI (include/apap_hip.h): error reporting, device
// selection, a small grow-only pool of device buffers, and the synchronous entry
// points that copy caller-owned numpy-style buffers to the GPU, enqueue the kernels
// through the "_device" entry points and copy the results back.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "apap_internal.h"

namespace {

thread_local char g_err[512] = "";

#define APAP_HIP_TRY(call)                                                              \
    do {                                                                                \
        const hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return apap::fail(APAP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

using namespace apap;   // DevSlot, S_* slot names

// The device buffers that host-buffer calls reuse live in the caller's context.  Calls made with
// a NULL context share this one pool (a cache: no option or result depends on it) and are
// serialised on its mutex - the reference is single-threaded, ctypes releases the GIL, so guard.
apap_ctx g_shared;

struct PoolLock {
    apap_ctx *pool;
    std::unique_lock<std::mutex> lock;
    explicit PoolLock(apap_ctx *ctx) : pool(ctx ? ctx : &g_shared), lock(pool->mu) {}
};

int slot_get(apap_ctx *pool, int which, size_t bytes, int dev, void **out) {
    DevSlot &s = pool->slots[which];
    if (bytes == 0) bytes = 4;
    if (s.ptr && (s.cap < bytes || s.dev != dev)) {
        (void)hipFree(s.ptr);
        s.ptr = nullptr;
        s.cap = 0;
    }
    if (!s.ptr) {
        // +16: the warp gather reads one dword at a 3-byte pixel and may touch 1 byte
        // past the image; keep that inside the allocation for pooled buffers.
        APAP_HIP_TRY(hipMalloc(&s.ptr, bytes + 16));
        s.cap = bytes;
        s.dev = dev;
    }
    *out = s.ptr;
    return APAP_OK;
}

int select_device(int device, int *chosen) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        (void)hipGetLastError();
        return apap::fail(APAP_ERR_NO_DEVICE,
                          "no HIP device visible (%s); this engine has no CPU fallback",
                          e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device >= count) return apap::fail(APAP_ERR_NO_DEVICE, "device %d requested, %d visible", device, count);
    if (device >= 0) {
        APAP_HIP_TRY(hipSetDevice(device));
        *chosen = device;
    } else {
        APAP_HIP_TRY(hipGetDevice(chosen));
    }
    return APAP_OK;
}

// Host-buffer calls enqueue copies from buffers they own (std::vector, stack arrays) on the null
// stream; whatever way the function is left, the stream is drained before those buffers die.
// Declare it AFTER the buffers it protects.
struct SyncOnExit {
    ~SyncOnExit() { (void)hipStreamSynchronize(nullptr); }
};

int status_to_code(int status, const char *who) {
    if (status & apap::kStatusSingular) return apap::fail(APAP_ERR_SINGULAR, "%s: Singular matrix", who);
    if (status & apap::kStatusIndex)
        return apap::fail(APAP_ERR_INDEX, "%s: index 0 is out of bounds for axis 0 with size 0 (mesh edges do not cover the canvas)", who);
    if (status & apap::kStatusUnprepared)
        return apap::fail(APAP_ERR_INVALID_ARG, "%s: the warp workspace holds no lookup tables for this mesh / canvas (APAP_WARP_GEOMETRY)", who);
    return APAP_OK;
}

}  // namespace

namespace apap {
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace apap

extern "C" {

const char *apap_last_error(void) { return g_err; }
const char *apap_version(void) { return "cvx_proj_amd apap-hip " APAP_ABI_VERSION_STRING " (gfx950)"; }
int apap_abi_version(void) { return APAP_ABI_VERSION; }

int apap_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

// The weight tensor of `cells` cells, streamed through a bounded device buffer (it is 8 n bytes per cell):
// d_table / d_vert are resident, W_out is the host destination.  1 GiB of device staging by default;
// APAP_OPT_WEIGHT_CHUNK_KB lets tests force several chunks.
static int stream_weights(apap_ctx *ctx, apap_ctx *pool, int dev, const void *d_table, int n, const void *d_vert, int cells,
                          double gamma, double sigma, double *W_out) {
    const size_t max_bytes = (size_t)apap::opt(ctx, APAP_OPT_WEIGHT_CHUNK_KB) << 10;
    int chunk = (int)(max_bytes / ((size_t)n * sizeof(double)));
    if (chunk < 1) chunk = 1;
    if (chunk > cells) chunk = cells;
    void *d_W;
    int rc;
    if ((rc = slot_get(pool, S_W, (size_t)chunk * n * sizeof(double), dev, &d_W))) return rc;
    for (int c0 = 0; c0 < cells; c0 += chunk) {
        const int nc = cells - c0 < chunk ? cells - c0 : chunk;
        rc = apap_weights_device(ctx, (const double *)d_table, n, (const double *)d_vert + (size_t)2 * c0, nc, gamma,
                                 sigma, (double *)d_W, nullptr);
        if (rc) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(W_out + (size_t)c0 * n, d_W, (size_t)nc * n * sizeof(double),
                                    hipMemcpyDeviceToHost, nullptr));
        APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return APAP_OK;
}

int apap_local_weights(apap_ctx *ctx, const float *src, int n, const double *vertices, int cells, double gamma,
                       double sigma, double *W_out, int device) {
    return apap_local_weights_pts(ctx, src, 0, n, vertices, cells, gamma, sigma, W_out, device);
}

int apap_local_weights_pts(apap_ctx *ctx, const void *src, int src_f64, int n, const double *vertices, int cells, double gamma,
                           double sigma, double *W_out, int device) {
    if (!src || !vertices || !W_out) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_weights: null argument");
    if (n < 1 || cells < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_weights: n=%d cells=%d", n, cells);
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;
    // the weight kernel reads a keypoint's (x, y) from columns 30, 31 of its table row
    std::vector<double> table((size_t)n * APAP_TABLE_STRIDE, 0.0);
    for (int k = 0; k < n; ++k) {
        table[(size_t)k * APAP_TABLE_STRIDE + 30] = src_f64 ? ((const double *)src)[2 * k] : (double)((const float *)src)[2 * k];
        table[(size_t)k * APAP_TABLE_STRIDE + 31] = src_f64 ? ((const double *)src)[2 * k + 1] : (double)((const float *)src)[2 * k + 1];
    }
    const SyncOnExit drain;   // the async copy below reads `table`
    void *d_table, *d_vert;
    if ((rc = slot_get(pl.pool, S_TABLE, table.size() * sizeof(double), dev, &d_table))) return rc;
    if ((rc = slot_get(pl.pool, S_VERT, (size_t)cells * 2 * sizeof(double), dev, &d_vert))) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_vert, vertices, (size_t)cells * 2 * sizeof(double), hipMemcpyHostToDevice, nullptr));
    return stream_weights(ctx, pl.pool, dev, d_table, n, d_vert, cells, gamma, sigma, W_out);
}

int apap_local_homography(apap_ctx *ctx, const float *src, const float *dst, int n, const double *vertices,
                          int mesh_rows, int mesh_cols, double gamma, double sigma, float *H_out,
                          double *W_out, int device) {
    return apap_local_homography_pts(ctx, src, 0, dst, 0, n, vertices, mesh_rows, mesh_cols, gamma, sigma, H_out, W_out, device);
}

int apap_local_homography_pts(apap_ctx *ctx, const void *src, int src_f64, const void *dst, int dst_f64, int n,
                              const double *vertices, int mesh_rows, int mesh_cols, double gamma, double sigma, float *H_out,
                              double *W_out, int device) {
    if (!src || !dst || !vertices || !H_out) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_homography: null argument");
    if (n < 2 || mesh_rows < 1 || mesh_cols < 1)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_homography: n=%d mesh=%dx%d", n, mesh_rows, mesh_cols);
    if ((long long)mesh_rows * mesh_cols > (1ll << 30)) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_homography: mesh too large");
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;

    // once-per-pair set-up on the host (apap.py:132-145)
    // in the dtype of each keypoint set, as the reference's own functions run (float64 keypoints stay float64 up to the
    // float32 rounding of the matrices and of the DLT rows)
    float N1[9], N2[9], C1[9], C2[9], iC2[9], iN2[9];
    std::vector<double> cf1((size_t)2 * n), cf2((size_t)2 * n), src64((size_t)2 * n);
    std::vector<float> aa((size_t)18 * n);
    rc = apap_host_prepare_pts(src, src_f64, dst, dst_f64, n, N1, N2, C1, C2, iC2, iN2, nullptr, nullptr, cf1.data(), cf2.data());
    if (rc) return rc;
    if ((rc = apap_host_dlt_rows_pts(cf1.data(), cf2.data(), n, src_f64 || dst_f64, aa.data()))) return rc;
    for (size_t i = 0; i < (size_t)2 * n; ++i) src64[i] = src_f64 ? ((const double *)src)[i] : (double)((const float *)src)[i];
    std::vector<double> table((size_t)n * APAP_TABLE_STRIDE);
    double denorm[APAP_DENORM_DOUBLES];
    rc = apap::opt(ctx, APAP_OPT_MOMENTS) == 24 ? apap_host_build_table24(src64.data(), aa.data(), n, table.data())
                                                : apap_host_build_table_rows(src64.data(), aa.data(), n, table.data());
    if (rc) return rc;
    if ((rc = apap_host_build_denorm(iC2, C1, iN2, N1, denorm))) return rc;
    const SyncOnExit drain;   // the async copies below read `table` and `denorm`

    const int cells = mesh_rows * mesh_cols;
    const size_t work_bytes = apap_solve_workspace_bytes(ctx, n, cells);
    void *d_table, *d_vert, *d_denorm, *d_H, *d_work;
    if ((rc = slot_get(pl.pool, S_TABLE, table.size() * sizeof(double), dev, &d_table))) return rc;
    if ((rc = slot_get(pl.pool, S_VERT, (size_t)cells * 2 * sizeof(double), dev, &d_vert))) return rc;
    if ((rc = slot_get(pl.pool, S_DENORM, sizeof(denorm), dev, &d_denorm))) return rc;
    if ((rc = slot_get(pl.pool, S_H, (size_t)cells * 9 * sizeof(float), dev, &d_H))) return rc;
    if ((rc = slot_get(pl.pool, S_WORK, work_bytes, dev, &d_work))) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_vert, vertices, (size_t)cells * 2 * sizeof(double), hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_denorm, denorm, sizeof(denorm), hipMemcpyHostToDevice, nullptr));
    rc = apap_solve_device(ctx, (const double *)d_table, n, (const double *)d_vert, cells, gamma, sigma,
                           (const double *)d_denorm, (float *)d_H, d_work, work_bytes, nullptr);
    if (rc) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(H_out, d_H, (size_t)cells * 9 * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    if (W_out && (rc = stream_weights(ctx, pl.pool, dev, d_table, n, d_vert, cells, gamma, sigma, W_out))) return rc;
    APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    return APAP_OK;
}

// ---- apap_local_warp / apap_local_stitch with PCIe overlapped ---------------------------------------
// The call moves 25 MB up and 27 MB down around a 20 us kernel (4K pair): one after the other that is
// ~1.1 ms, almost all of it PCIe one way at a time.  Here the caller's buffers are pinned for the call, the
// source image goes up in row chunks on one stream, the canvas is warped in row bands on a second - band b
// as soon as the source rows it can read have landed - and every finished band goes down on a third while
// later chunks are still going up.  Measured on the MI355X boxes of this pool (a build with host and event time stamps - git tag r05-hooks -, 4K pair):
// 0.93 ms against 1.14 ms - the two directions do NOT add up here: 25 MB up alone and 27 MB down alone each move
// at ~56 GB/s, both together at ~60 GB/s in all (the same with the canvas written straight into the pinned host
// buffer by the kernel instead of a DMA copy), so what the overlap hides is the kernels, the set-up and the
// per-copy latencies, not half of the bytes.  And pinning is not free: the FIRST call on a buffer (a new virtual range)
// spends ~1 ms registering it and ~7 ms at its first DMA use; only later calls on the same buffers run at 0.93 ms
// (profiles/r03_pcie_overlap.txt).  Hence opt-in (APAP_OPT_OVERLAP_PCIE = 1): right for a caller that streams pairs through buffers it
// keeps, wrong for one warp into a fresh array - the default stays the plain sequence.
// WHICH source rows a band can read is not guessed: the set-up kernel reports, per cell row, an interval that
// contains the source row of every pixel of every cell in it (anchor row -+ the bound of the float32
// estimate's magnitude, apap_kernels.hip fast_record), and the host takes the union over the band's cell
// rows.  Meshes the set-up has no such interval for (irregular edges, cells wider than 254 pixels, a
// perspective denominator that changes sign inside a cell) wait for the whole image: the order of the
// transfers changes, never the bytes of the canvas.
namespace {

struct PinGuard {       // hipHostRegister for the duration of a call
    void *p = nullptr;
    bool pin(const void *ptr, size_t bytes, unsigned flags = hipHostRegisterDefault) {
        // memory the caller already page-locked (hipHostMalloc, torch's pin_memory, its own hipHostRegister): nothing to do,
        // nothing to undo
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeHost) {
            hipPointerAttribute_t last;
            if (hipPointerGetAttributes(&last, (const char *)ptr + bytes - 1) == hipSuccess && last.type == hipMemoryTypeHost) return true;
        }
        (void)hipGetLastError();
        if (hipHostRegister(const_cast<void *>(ptr), bytes, flags) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        p = const_cast<void *>(ptr);
        return true;
    }
    ~PinGuard() {
        if (p) (void)hipHostUnregister(p);
    }
};

int pipe_prepare(apap_ctx *pool, int dev, size_t n_events, size_t pinned_bytes) {
    if (pool->pipe_dev != dev) {     // streams and events belong to a device
        for (void *e : pool->events) (void)hipEventDestroy((hipEvent_t)e);
        pool->events.clear();
        for (void *&st : pool->streams) {
            if (st) (void)hipStreamDestroy((hipStream_t)st);
            st = nullptr;
        }
        pool->pipe_dev = dev;
    }
    for (void *&st : pool->streams)
        if (!st) {
            hipStream_t h;
            APAP_HIP_TRY(hipStreamCreateWithFlags(&h, hipStreamNonBlocking));
            st = h;
        }
    while (pool->events.size() < n_events) {
        hipEvent_t e;
        APAP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pool->events.push_back(e);
    }
    if (pool->pinned_cap < pinned_bytes) {
        if (pool->pinned) (void)hipHostFree(pool->pinned);
        pool->pinned = nullptr;
        pool->pinned_cap = 0;
        APAP_HIP_TRY(hipHostMalloc(&pool->pinned, pinned_bytes, hipHostMallocDefault));
        pool->pinned_cap = pinned_bytes;
    }
    return APAP_OK;
}

struct DrainStreams {    // whatever way the function is left, nothing of it is still running
    apap_ctx *pool;
    ~DrainStreams() {
        for (void *st : pool->streams)
            if (st) (void)hipStreamSynchronize((hipStream_t)st);
    }
};

// rows of `total` in `parts` near-equal pieces: piece i = [cut(i), cut(i + 1))
inline int cut(int total, int parts, int i) { return (int)((long long)total * i / parts); }

// Returns APAP_OK and sets *done = true when the call was served here; *done = false (and APAP_OK) when the
// caller should take the plain sequential path (pinning refused, a mesh without the fast tables).
int warp_overlapped(apap_ctx *ctx, apap_ctx *pool, int dev, const uint8_t *img, int img_h, int img_w, const uint8_t *center,
                    int center_h, int center_w, const float *Hfwd, int mesh_rows, int mesh_cols, const double *mesh_w,
                    int n_w, const double *mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *out,
                    float *Hinv_out, const char *who, bool *done) {
    *done = false;
    const size_t img_bytes = (size_t)img_h * img_w * 3, out_bytes = (size_t)final_w * final_h * 3;
    const size_t cbytes = center ? (size_t)center_h * center_w * 3 : 0;
    if (img_bytes + out_bytes < (8u << 20)) return APAP_OK;      // small pairs: one copy each way is as good
    const int cells = mesh_rows * mesh_cols;
    // (chunks x bands x order were swept in round 4 - profiles/r04_pcie_duplex.txt: a plateau at 0.89-0.93 ms for 4-8 chunks and 4-8
    // bands; grid upload + set-up before the image chunks is worse whenever the inverses come back)
    const int chunks = (int)std::min<size_t>(16, std::max<size_t>(2, img_bytes / (3u << 20)));
    const int bands = (int)std::min<size_t>(16, std::max<size_t>(2, out_bytes / (3u << 20)));
    const size_t range_ints = (size_t)mesh_rows * 2 + 2;
    int rc = pipe_prepare(pool, dev, (size_t)2 * chunks + 2 * bands + 4, range_ints * sizeof(int) + 64);
    if (rc) return rc;
    // Only the three big buffers are pinned, and only when no two of them share a page.  History: with the grid and its
    // inverse pinned as well (numpy's `H.copy()` followed by `np.empty_like(H)`: back to back) one run of round 3 ended in a
    // GPU fault "write access to a read-only page", attributed then to the two registrations sharing a page.  Round 4's
    // stand-alone reproducer of exactly that layout (profiles/r04_hostreg_pages.txt, the program at git tag r05-hooks: both ranges
    // registered, DMA in both directions and at once, a kernel reading one and writing the other through the device
    // pointers, either unregistration order) runs clean on this stack: the shared page was NOT the cause, which stays
    // unknown (no log of the fault survived).  The rule is kept because it costs nothing: pinning 1.4 MB buys nothing (the
    // grid, its inverse and the edges travel as pageable copies on the kernel stream while the image chunks are already going
    // up), and page-disjoint registrations are the case every test exercises (tests/test_gpu_parity.py:
    // test_overlapped_host_warp_on_buffers_that_share_pages runs both layouts).
    auto pages_overlap = [](const void *a, size_t na, const void *b, size_t nb) {
        const uintptr_t pa0 = (uintptr_t)a >> 12, pa1 = ((uintptr_t)a + na - 1) >> 12;
        const uintptr_t pb0 = (uintptr_t)b >> 12, pb1 = ((uintptr_t)b + nb - 1) >> 12;
        return pa0 <= pb1 && pb0 <= pa1;
    };
    if (pages_overlap(img, img_bytes, out, out_bytes) ||
        (center && (pages_overlap(center, cbytes, out, out_bytes) || pages_overlap(center, cbytes, img, img_bytes))))
        return APAP_OK;
    PinGuard pin_img, pin_out, pin_center;
    if (!pin_img.pin(img, img_bytes) || !pin_out.pin(out, out_bytes) || (center && !pin_center.pin(center, cbytes))) return APAP_OK;

    const size_t work_bytes = apap_warp_workspace_bytes(mesh_rows, mesh_cols, final_w, final_h);
    void *d_H, *d_mw, *d_mh, *d_work, *d_status, *d_hinv = nullptr, *d_img, *d_out, *d_center = nullptr;
    if ((rc = slot_get(pool, S_H, (size_t)cells * 9 * sizeof(float), dev, &d_H))) return rc;
    if ((rc = slot_get(pool, S_MESHW, (size_t)n_w * sizeof(double), dev, &d_mw))) return rc;
    if ((rc = slot_get(pool, S_MESHH, (size_t)n_h * sizeof(double), dev, &d_mh))) return rc;
    if ((rc = slot_get(pool, S_WORK, work_bytes, dev, &d_work))) return rc;
    if ((rc = slot_get(pool, S_STATUS, sizeof(int), dev, &d_status))) return rc;
    if (Hinv_out && (rc = slot_get(pool, S_HINV, (size_t)cells * 9 * sizeof(float), dev, &d_hinv))) return rc;
    if ((rc = slot_get(pool, S_IMG, img_bytes, dev, &d_img))) return rc;
    if ((rc = slot_get(pool, S_OUT, out_bytes, dev, &d_out))) return rc;
    if (center && (rc = slot_get(pool, S_AUX, cbytes, dev, &d_center))) return rc;

    hipStream_t s_up = (hipStream_t)pool->streams[0], s_k = (hipStream_t)pool->streams[1], s_dn = (hipStream_t)pool->streams[2];
    hipEvent_t *ev = reinterpret_cast<hipEvent_t *>(pool->events.data());
    hipEvent_t *e_img = ev, *e_cen = ev + chunks, *e_band = ev + 2 * chunks, e_setup = ev[2 * chunks + bands];
    int *h_rng = (int *)pool->pinned;
    int status = 0;
    const DrainStreams drain{pool};     // declared after everything the streams read or write

    // upload stream: source rows (and the centre image's) in chunks, an event after each
    auto enqueue_uploads = [&]() -> int {
        for (int c = 0; c < chunks; ++c) {
            const size_t a = (size_t)cut(img_h, chunks, c) * img_w * 3, b = (size_t)cut(img_h, chunks, c + 1) * img_w * 3;
            if (b > a) APAP_HIP_TRY(hipMemcpyAsync((char *)d_img + a, img + a, b - a, hipMemcpyHostToDevice, s_up));
            APAP_HIP_TRY(hipEventRecord(e_img[c], s_up));
            if (center) {
                const size_t ca = (size_t)cut(center_h, chunks, c) * center_w * 3, cb = (size_t)cut(center_h, chunks, c + 1) * center_w * 3;
                if (cb > ca) APAP_HIP_TRY(hipMemcpyAsync((char *)d_center + ca, center + ca, cb - ca, hipMemcpyHostToDevice, s_up));
                APAP_HIP_TRY(hipEventRecord(e_cen[c], s_up));
            }
        }
        return APAP_OK;
    };
    if ((rc = enqueue_uploads())) return rc;

    // kernels stream: grid and edges up, set-up kernel, its source-row intervals back
    int *d_src_rows = nullptr;
    APAP_HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), s_k));
    APAP_HIP_TRY(hipMemcpyAsync(d_H, Hfwd, (size_t)cells * 9 * sizeof(float), hipMemcpyHostToDevice, s_k));
    APAP_HIP_TRY(hipMemcpyAsync(d_mw, mesh_w, (size_t)n_w * sizeof(double), hipMemcpyHostToDevice, s_k));
    APAP_HIP_TRY(hipMemcpyAsync(d_mh, mesh_h, (size_t)n_h * sizeof(double), hipMemcpyHostToDevice, s_k));
    {
        // where the intervals will live (the same layout the set-up uses); lower bounds start high, upper bounds low
        int *probe = nullptr;
        rc = apap::warp_phase(ctx, (const uint8_t *)d_img, img_h, img_w, (const uint8_t *)d_center, center_h, center_w,
                              (const float *)d_H, mesh_rows, mesh_cols, (const double *)d_mw, n_w, (const double *)d_mh, n_h,
                              final_w, final_h, off_x, off_y, (uint8_t *)d_out, (float *)d_hinv, d_work, work_bytes,
                              (int *)d_status, s_k, 0, 0, 0, &probe);
        if (rc) return rc;
        if (!probe) return APAP_OK;     // a mesh the fast tables do not cover: sequential path
        // one interleaved fill: even words (lower bounds) 0x7f7f7f7f, odd words (upper bounds) 0x80808080
        std::vector<int> init(range_ints);
        for (int r = 0; r < mesh_rows; ++r) {
            init[2 * r] = 0x7f7f7f7f;
            init[2 * r + 1] = (int)0x80808080u;
        }
        init[2 * mesh_rows] = init[2 * mesh_rows + 1] = 0;
        memcpy(h_rng, init.data(), range_ints * sizeof(int));
        // (h_rng is reused for the read-back below: the same stream, so the device has read it by then)
        APAP_HIP_TRY(hipMemcpyAsync(probe, h_rng, range_ints * sizeof(int), hipMemcpyHostToDevice, s_k));
    }
    rc = apap::warp_phase(ctx, (const uint8_t *)d_img, img_h, img_w, (const uint8_t *)d_center, center_h, center_w,
                          (const float *)d_H, mesh_rows, mesh_cols, (const double *)d_mw, n_w, (const double *)d_mh, n_h, final_w,
                          final_h, off_x, off_y, (uint8_t *)d_out, (float *)d_hinv, d_work, work_bytes, (int *)d_status, s_k, 0, 0,
                          apap::kWarpSetup, &d_src_rows);
    if (rc) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(h_rng, d_src_rows, range_ints * sizeof(int), hipMemcpyDeviceToHost, s_k));
    APAP_HIP_TRY(hipEventRecord(e_setup, s_k));
    if (Hinv_out)
        APAP_HIP_TRY(hipMemcpyAsync(Hinv_out, d_hinv, (size_t)cells * 9 * sizeof(float), hipMemcpyDeviceToHost, s_k));

    // the intervals: which chunk must have landed before band b may run
    APAP_HIP_TRY(hipEventSynchronize(e_setup));
    // (the host's cell_row_of below is a binary search: valid on increasing edges only; the device's running-maximum table also
    // serves merely swapped edges, whose pixels all sit in ordinary cells and do not set the flag bit)
    const bool irregular = (h_rng[2 * mesh_rows] & 1) != 0 || !std::is_sorted(mesh_h, mesh_h + n_h);
    auto cell_row_of = [&](int y) {     // "first k with y < mesh_h[k]" - 1 on increasing edges
        const int k = (int)(std::upper_bound(mesh_h, mesh_h + n_h, (double)y) - mesh_h) - 1;
        return std::min(std::max(k, 0), mesh_rows - 1);
    };
    auto chunk_of_row = [&](int total, int row) {      // the chunk that holds `row`
        int c = (int)(((long long)row + 1) * chunks / std::max(total, 1));
        c = std::min(std::max(c, 0), chunks - 1);
        while (c > 0 && cut(total, chunks, c) > row) --c;
        while (c + 1 < chunks && cut(total, chunks, c + 1) <= row) ++c;
        return c;
    };
    for (int b = 0; b < bands; ++b) {
        const int y0 = cut(final_h, bands, b), y1 = cut(final_h, bands, b + 1);
        if (y1 <= y0) {
            APAP_HIP_TRY(hipEventRecord(e_band[b], s_k));
            continue;
        }
        int need = chunks - 1;      // without intervals: the whole image
        if (!irregular) {
            long long hi = -1;
            for (int r = cell_row_of(y0); r <= cell_row_of(y1 - 1); ++r) hi = std::max<long long>(hi, h_rng[2 * r + 1]);
            hi = std::min<long long>(hi + 1, img_h - 1);      // + 1: the gather's dword may reach into the next row
            need = hi < 0 ? 0 : chunk_of_row(img_h, (int)hi);
        }
        APAP_HIP_TRY(hipStreamWaitEvent(s_k, e_img[need], 0));
        if (center) {
            const int c_hi = std::min(std::max(y1 - 1 - off_y, 0), center_h - 1);
            APAP_HIP_TRY(hipStreamWaitEvent(s_k, e_cen[chunk_of_row(center_h, c_hi)], 0));
        }
        uint8_t *d_band = (uint8_t *)d_out + (size_t)y0 * final_w * 3;
        rc = apap::warp_phase(ctx, (const uint8_t *)d_img, img_h, img_w, (const uint8_t *)d_center, center_h, center_w,
                              (const float *)d_H, mesh_rows, mesh_cols, (const double *)d_mw, n_w, (const double *)d_mh, n_h,
                              final_w, final_h, off_x, off_y, d_band, nullptr, d_work, work_bytes, (int *)d_status, s_k, y0,
                              y1 - y0, apap::kWarpRows, nullptr);
        if (rc) return rc;
        APAP_HIP_TRY(hipEventRecord(e_band[b], s_k));
        APAP_HIP_TRY(hipStreamWaitEvent(s_dn, e_band[b], 0));
        APAP_HIP_TRY(hipMemcpyAsync(out + (size_t)y0 * final_w * 3, d_band, (size_t)(y1 - y0) * final_w * 3, hipMemcpyDeviceToHost, s_dn));
    }
    APAP_HIP_TRY(hipMemcpyAsync(h_rng, d_status, sizeof(int), hipMemcpyDeviceToHost, s_k));
    APAP_HIP_TRY(hipStreamSynchronize(s_k));
    status = h_rng[0];
    APAP_HIP_TRY(hipStreamSynchronize(s_dn));
    APAP_HIP_TRY(hipStreamSynchronize(s_up));
    *done = true;
    return status_to_code(status, who);
}

}  // namespace

// `h_bytes` = 4: the grid (and Hinv_out) is float32; 8: float64 (apap_local_warp_f64 only).
static int warp_common(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const uint8_t *center, int center_h,
                       int center_w, const void *Hfwd, size_t h_bytes, int mesh_rows,
                       int mesh_cols, const double *mesh_w, int n_w, const double *mesh_h, int n_h,
                       int final_w, int final_h, int off_x, int off_y, uint8_t *out, void *Hinv_out,
                       double *coords, int device, const char *who) {
    if (!Hfwd || !mesh_w || !mesh_h) return apap::fail(APAP_ERR_INVALID_ARG, "%s: null argument", who);
    if (mesh_rows < 1 || mesh_cols < 1 || n_w < 1 || n_h < 1 || final_w < 1 || final_h < 1)
        return apap::fail(APAP_ERR_INVALID_ARG, "%s: bad size", who);
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;
    if (!coords && h_bytes == sizeof(float) && apap::opt(ctx, APAP_OPT_OVERLAP_PCIE)) {
        bool done = false;
        rc = warp_overlapped(ctx, pl.pool, dev, img, img_h, img_w, center, center_h, center_w, (const float *)Hfwd, mesh_rows,
                             mesh_cols, mesh_w, n_w, mesh_h, n_h, final_w, final_h, off_x, off_y, out, (float *)Hinv_out, who, &done);
        if (rc || done) return rc;
    }
    const int cells = mesh_rows * mesh_cols;
    const size_t work_bytes = apap_warp_workspace_bytes(mesh_rows, mesh_cols, final_w, final_h);
    const size_t pixels = (size_t)final_w * final_h;
    void *d_H, *d_mw, *d_mh, *d_work, *d_status, *d_hinv = nullptr, *d_img = nullptr, *d_out = nullptr;
    if ((rc = slot_get(pl.pool, S_H, (size_t)cells * 9 * h_bytes, dev, &d_H))) return rc;
    if ((rc = slot_get(pl.pool, S_MESHW, (size_t)n_w * sizeof(double), dev, &d_mw))) return rc;
    if ((rc = slot_get(pl.pool, S_MESHH, (size_t)n_h * sizeof(double), dev, &d_mh))) return rc;
    if ((rc = slot_get(pl.pool, S_WORK, work_bytes, dev, &d_work))) return rc;
    if ((rc = slot_get(pl.pool, S_STATUS, sizeof(int), dev, &d_status))) return rc;
    if (Hinv_out && (rc = slot_get(pl.pool, S_HINV, (size_t)cells * 9 * h_bytes, dev, &d_hinv))) return rc;
    const SyncOnExit drain;   // `status` below is a stack variable the last copy writes
    APAP_HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_H, Hfwd, (size_t)cells * 9 * h_bytes, hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_mw, mesh_w, (size_t)n_w * sizeof(double), hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_mh, mesh_h, (size_t)n_h * sizeof(double), hipMemcpyHostToDevice, nullptr));
    if (coords) {
        if ((rc = slot_get(pl.pool, S_OUT, pixels * 2 * sizeof(double), dev, &d_out))) return rc;
        rc = apap_warp_coords_device(ctx, (const float *)d_H, mesh_rows, mesh_cols, (const double *)d_mw, n_w,
                                     (const double *)d_mh, n_h, final_w, final_h, off_x, off_y, (double *)d_out,
                                     d_work, work_bytes, (int *)d_status, nullptr);
        if (rc) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(coords, d_out, pixels * 2 * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    } else {
        const size_t img_bytes = (size_t)img_h * img_w * 3;
        if ((rc = slot_get(pl.pool, S_IMG, img_bytes, dev, &d_img))) return rc;
        if ((rc = slot_get(pl.pool, S_OUT, pixels * 3, dev, &d_out))) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(d_img, img, img_bytes, hipMemcpyHostToDevice, nullptr));
        if (center) {
            void *d_center;
            const size_t cbytes = (size_t)center_h * center_w * 3;
            if ((rc = slot_get(pl.pool, S_AUX, cbytes, dev, &d_center))) return rc;
            APAP_HIP_TRY(hipMemcpyAsync(d_center, center, cbytes, hipMemcpyHostToDevice, nullptr));
            rc = apap_stitch_device(ctx, (const uint8_t *)d_img, img_h, img_w, (const uint8_t *)d_center, center_h, center_w,
                                    (const float *)d_H, mesh_rows, mesh_cols, (const double *)d_mw, n_w,
                                    (const double *)d_mh, n_h, final_w, final_h, off_x, off_y, (uint8_t *)d_out,
                                    (float *)d_hinv, d_work, work_bytes, (int *)d_status, nullptr);
        } else if (h_bytes == sizeof(double)) {
            rc = apap_warp_f64_device(ctx, (const uint8_t *)d_img, img_h, img_w, (const double *)d_H, mesh_rows, mesh_cols,
                                      (const double *)d_mw, n_w, (const double *)d_mh, n_h, final_w, final_h, off_x, off_y,
                                      (uint8_t *)d_out, (double *)d_hinv, d_work, work_bytes, (int *)d_status, nullptr);
        } else {
            rc = apap_warp_device(ctx, (const uint8_t *)d_img, img_h, img_w, (const float *)d_H, mesh_rows, mesh_cols,
                                  (const double *)d_mw, n_w, (const double *)d_mh, n_h, final_w, final_h, off_x, off_y,
                                  (uint8_t *)d_out, (float *)d_hinv, d_work, work_bytes, (int *)d_status, nullptr);
        }
        if (rc) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(out, d_out, pixels * 3, hipMemcpyDeviceToHost, nullptr));
        if (Hinv_out)
            APAP_HIP_TRY(hipMemcpyAsync(Hinv_out, d_hinv, (size_t)cells * 9 * h_bytes, hipMemcpyDeviceToHost, nullptr));
    }
    int status = 0;
    APAP_HIP_TRY(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, nullptr));
    APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    return status_to_code(status, who);
}

int apap_local_warp(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const float *Hfwd, int mesh_rows,
                    int mesh_cols, const double *mesh_w, int n_w, const double *mesh_h, int n_h,
                    int final_w, int final_h, int off_x, int off_y, uint8_t *out, float *Hinv_out,
                    int device) {
    if (!img || !out) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_warp: null image");
    if (img_h < 1 || img_w < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_warp: bad image size");
    return warp_common(ctx, img, img_h, img_w, nullptr, 0, 0, Hfwd, sizeof(float), mesh_rows, mesh_cols, mesh_w, n_w, mesh_h, n_h,
                       final_w, final_h, off_x, off_y, out, Hinv_out, nullptr, device, "apap_local_warp");
}

int apap_local_warp_f64(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const double *Hfwd, int mesh_rows,
                        int mesh_cols, const double *mesh_w, int n_w, const double *mesh_h, int n_h,
                        int final_w, int final_h, int off_x, int off_y, uint8_t *out, double *Hinv_out,
                        int device) {
    if (!img || !out) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_warp_f64: null image");
    if (img_h < 1 || img_w < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_warp_f64: bad image size");
    return warp_common(ctx, img, img_h, img_w, nullptr, 0, 0, Hfwd, sizeof(double), mesh_rows, mesh_cols, mesh_w, n_w, mesh_h,
                       n_h, final_w, final_h, off_x, off_y, out, Hinv_out, nullptr, device, "apap_local_warp_f64");
}

int apap_local_stitch(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const uint8_t *center, int center_h,
                      int center_w, const float *Hfwd, int mesh_rows, int mesh_cols,
                      const double *mesh_w, int n_w, const double *mesh_h, int n_h, int final_w,
                      int final_h, int off_x, int off_y, uint8_t *out, float *Hinv_out, int device) {
    if (!img || !out || !center) return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_stitch: null image");
    if (img_h < 1 || img_w < 1 || center_h < 1 || center_w < 1)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_local_stitch: bad image size");
    return warp_common(ctx, img, img_h, img_w, center, center_h, center_w, Hfwd, sizeof(float), mesh_rows, mesh_cols, mesh_w, n_w,
                       mesh_h, n_h, final_w, final_h, off_x, off_y, out, Hinv_out, nullptr, device, "apap_local_stitch");
}

int apap_warp_coords(apap_ctx *ctx, const float *Hfwd, int mesh_rows, int mesh_cols, const double *mesh_w, int n_w,
                     const double *mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                     double *coords, int device) {
    if (!coords) return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_coords: null output");
    return warp_common(ctx, nullptr, 0, 0, nullptr, 0, 0, Hfwd, sizeof(float), mesh_rows, mesh_cols, mesh_w, n_w, mesh_h, n_h,
                       final_w, final_h, off_x, off_y, nullptr, nullptr, coords, device, "apap_warp_coords");
}

int apap_invert_normalize_flatten(apap_ctx *ctx, const float *H, int cells, double *out, int device) {
    if (!H || !out || cells < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_invert_normalize_flatten: bad argument");
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;
    void *d_H, *d_out, *d_status;
    if ((rc = slot_get(pl.pool, S_H, (size_t)cells * 9 * sizeof(float), dev, &d_H))) return rc;
    if ((rc = slot_get(pl.pool, S_AUX, (size_t)cells * 9 * sizeof(double), dev, &d_out))) return rc;
    if ((rc = slot_get(pl.pool, S_STATUS, sizeof(int), dev, &d_status))) return rc;
    int status = 0;
    const SyncOnExit drain;
    APAP_HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_H, H, (size_t)cells * 9 * sizeof(float), hipMemcpyHostToDevice, nullptr));
    rc = apap_flatten_device(ctx, (const float *)d_H, cells, (double *)d_out, (int *)d_status, nullptr);
    if (rc) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)cells * 9 * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, nullptr));
    APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    return status_to_code(status, "apap_invert_normalize_flatten");
}

int apap_uniform_blend(apap_ctx *ctx, const uint8_t *img1, const uint8_t *img2, int h, int w, uint8_t *out,
                       int device) {
    if (!img1 || !img2 || !out || h < 1 || w < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_uniform_blend: bad argument");
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;
    const size_t bytes = (size_t)h * w * 3;
    void *d_a, *d_b, *d_o;
    if ((rc = slot_get(pl.pool, S_IMG, bytes, dev, &d_a))) return rc;
    if ((rc = slot_get(pl.pool, S_AUX, bytes, dev, &d_b))) return rc;
    if ((rc = slot_get(pl.pool, S_OUT, bytes, dev, &d_o))) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(d_a, img1, bytes, hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemcpyAsync(d_b, img2, bytes, hipMemcpyHostToDevice, nullptr));
    rc = apap_blend_device(ctx, (const uint8_t *)d_a, (const uint8_t *)d_b, h, w, (uint8_t *)d_o, nullptr);
    if (rc) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(out, d_o, bytes, hipMemcpyDeviceToHost, nullptr));
    APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    return APAP_OK;
}

int apap_equalize_hist(apap_ctx *ctx, const uint8_t *img, int h, int w, int channels, uint8_t *out, int device) {
    if (!img || !out || h < 1 || w < 1 || channels < 1 || channels > 4)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_equalize_hist: bad argument");
    PoolLock pl(ctx);
    int dev;
    int rc = select_device(device, &dev);
    if (rc) return rc;
    const size_t bytes = (size_t)h * w * channels;
    const size_t work_bytes = apap_equalize_workspace_bytes(channels);
    void *d_a, *d_o, *d_work;
    if ((rc = slot_get(pl.pool, S_IMG, bytes, dev, &d_a))) return rc;
    if ((rc = slot_get(pl.pool, S_OUT, bytes, dev, &d_o))) return rc;
    if ((rc = slot_get(pl.pool, S_WORK, work_bytes, dev, &d_work))) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(d_a, img, bytes, hipMemcpyHostToDevice, nullptr));
    APAP_HIP_TRY(hipMemsetAsync(d_work, 0, work_bytes, nullptr));  // the pooled buffer is shared: zero it per call
    rc = apap_equalize_hist_device(ctx, (const uint8_t *)d_a, h, w, channels, (uint8_t *)d_o, d_work, work_bytes, nullptr);
    if (rc) return rc;
    APAP_HIP_TRY(hipMemcpyAsync(out, d_o, bytes, hipMemcpyDeviceToHost, nullptr));
    APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    return APAP_OK;
}

int apap_find_homography_ransac(apap_ctx *ctx, const float *src, const float *dst, int n, double thresh, int iterations,
                                unsigned long long seed, double *H_out, uint8_t *mask_out, int *inliers_out,
                                int device) {
    if (!src || !dst || !H_out || !mask_out || !inliers_out)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_find_homography_ransac: null argument");
    if (n < 4) return apap::fail(APAP_ERR_INVALID_ARG, "apap_find_homography_ransac: n=%d, a homography needs 4 points", n);
    int result[2] = {0, 0};
    {
        PoolLock pl(ctx);
        int dev;
        int rc = select_device(device, &dev);
        if (rc) return rc;
        const size_t pt_bytes = (size_t)n * 2 * sizeof(float);
        const size_t work_bytes = apap_ransac_workspace_bytes(n, iterations);
        void *d_src, *d_dst, *d_work, *d_H, *d_mask, *d_result;
        if ((rc = slot_get(pl.pool, S_IMG, pt_bytes, dev, &d_src))) return rc;
        if ((rc = slot_get(pl.pool, S_AUX, pt_bytes, dev, &d_dst))) return rc;
        if ((rc = slot_get(pl.pool, S_WORK, work_bytes, dev, &d_work))) return rc;
        if ((rc = slot_get(pl.pool, S_DENORM, 9 * sizeof(double), dev, &d_H))) return rc;
        if ((rc = slot_get(pl.pool, S_OUT, (size_t)n, dev, &d_mask))) return rc;
        if ((rc = slot_get(pl.pool, S_STATUS, 2 * sizeof(int), dev, &d_result))) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(d_src, src, pt_bytes, hipMemcpyHostToDevice, nullptr));
        APAP_HIP_TRY(hipMemcpyAsync(d_dst, dst, pt_bytes, hipMemcpyHostToDevice, nullptr));
        rc = apap_ransac_device(ctx, (const float *)d_src, (const float *)d_dst, n, thresh, iterations, seed, (double *)d_H,
                                (uint8_t *)d_mask, (int *)d_result, d_work, work_bytes, nullptr);
        if (rc) return rc;
        APAP_HIP_TRY(hipMemcpyAsync(mask_out, d_mask, (size_t)n, hipMemcpyDeviceToHost, nullptr));
        APAP_HIP_TRY(hipMemcpyAsync(result, d_result, sizeof(result), hipMemcpyDeviceToHost, nullptr));
        APAP_HIP_TRY(hipStreamSynchronize(nullptr));
    }
    *inliers_out = result[1];
    if (result[1] < 4) {  // cv.findHomography returns no model
        memset(mask_out, 0, (size_t)n);
        return APAP_OK;
    }
    // re-fit to the inliers with the hot path's own normalised DLT: one cell, every weight 1
    std::vector<float> s_in, d_in;
    s_in.reserve((size_t)2 * result[1]);
    d_in.reserve((size_t)2 * result[1]);
    for (int k = 0; k < n; ++k) {
        if (!mask_out[k]) continue;
        s_in.push_back(src[2 * k]); s_in.push_back(src[2 * k + 1]);
        d_in.push_back(dst[2 * k]); d_in.push_back(dst[2 * k + 1]);
    }
    const double vertex[2] = {0.0, 0.0};
    float H32[9];
    const int rc = apap_local_homography(ctx, s_in.data(), d_in.data(), (int)(s_in.size() / 2), vertex, 1, 1, 1.0, 1.0, H32,
                                         nullptr, device);
    if (rc) return rc;
    for (int i = 0; i < 9; ++i) H_out[i] = (double)H32[i];
    return APAP_OK;
}

}  // extern "C"
