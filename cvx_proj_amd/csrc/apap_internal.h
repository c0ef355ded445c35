// Internal declarations shared by the host set-up, the kernel launchers and the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>

#include <mutex>
#include <vector>

#include "../../include/apap_hip.h"

namespace apap {

struct ProfSpan {   // one bracketed kernel: two HIP events on the launch stream
    int slot;
    void *a, *b;
};

struct DevSlot {    // one pooled device buffer of the host-buffer entry points
    void *ptr = nullptr;
    size_t cap = 0;
    int dev = -1;
};
enum { S_TABLE, S_VERT, S_DENORM, S_H, S_WORK, S_W, S_IMG, S_OUT, S_MESHW, S_MESHH, S_HINV, S_STATUS, S_AUX, S_COUNT };

}  // namespace apap

// The context of include/apap_hip.h: options, profiling events, device-buffer pool.  Nothing else
// in the library is mutable after load.
struct apap_ctx {
    int opt[APAP_OPT_COUNT] = {APAP_VARIANT_AUTO, APAP_EIGEN_AUTO, 1, 0, 4096, 1, 1 << 20, 4096, 1, 0, 0, 30, 0};
    std::vector<apap::ProfSpan> spans;
    apap::DevSlot slots[apap::S_COUNT];
    std::mutex mu;   // serialises the host-buffer entry points that share this context's pool
    // overlap of PCIe and kernels in apap_local_warp / apap_local_stitch: three streams (upload, kernels,
    // download), their events, a small pinned buffer for what the host reads back mid-call; made on first use
    void *streams[3] = {nullptr, nullptr, nullptr};
    std::vector<void *> events;
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    int pipe_dev = -1;
};

namespace apap {

// Option `which` of `ctx`, or its built-in default when ctx is NULL.
int opt(const apap_ctx *ctx, int which);

// Records a thread-local message for apap_last_error() and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Records "<what>: <hipGetErrorString>" and returns APAP_ERR_HIP.
int hip_fail(int hip_error, const char *what);

// Optional per-kernel timing (APAP_OPT_PROFILE / apap_ctx_profile_read): brackets the kernels
// launched in its scope with HIP events on `stream`, kept in the context.
struct ProfScope {
    ProfScope(apap_ctx *ctx, int slot, void *stream);
    ~ProfScope();
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;

  private:
    apap_ctx *ctx_;
    void *stream_, *a_, *b_;
    int slot_;
    bool on_;
};

bool inv3_f64(const double in[9], double out[9]);
bool inv3_f32(const float in[9], float out[9]);

// ---- geometry of the solve launch (shared by launcher and workspace sizing) ----
struct SolvePlan {
    int variant;       // APAP_VARIANT_VALU or APAP_VARIANT_MFMA
    int cells_pad;     // cells rounded up to the cell tile of the variant
    int cell_tiles;    // grid.x
    int splits;        // grid.y: independent slices of the keypoint list
    int pts_per_split; // keypoints per slice (a multiple of 4)
    size_t moment_bytes;
};
SolvePlan plan_solve(int n, int cells, int variant, int batch, int want_waves, int plan_cells = 0, int moments = 30);

// ---- the warp in two phases on one workspace (the host-buffer entry points overlap PCIe with it) ----
constexpr int kWarpSetup = APAP_WARP_GEOMETRY | APAP_WARP_CELLS;   // lookup tables + cell inverses and fast records (and, if asked, source-row intervals)
constexpr int kWarpRows = APAP_WARP_GATHER;   // the gather kernel over canvas rows [row_begin, row_begin + row_count)
// apap_warp_rows_device / apap_stitch_device with a choice of phases.  With `d_src_rows` non-null the set-up also
// fills, per cell row, the interval of source rows its pixels can read (device ints [rows][2], then one flag word:
// bit 0 = irregular mesh, intervals void); the caller pre-sets the lower bounds to a large and the upper bounds to a
// small value.  *d_src_rows is set to the device address, or to NULL when this mesh has no such table; phase 0 only
// reports that address and launches nothing.
int warp_phase(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const uint8_t *d_center, int center_h, int center_w,
               const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h,
               int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *d_out_band, float *d_Hinv_out, void *d_work,
               size_t work_bytes, int *d_status, void *stream, int row_begin, int row_count, int phase, int **d_src_rows);

constexpr int kMoments = 30;       // distinct sums of A^T W^2 A
constexpr int kStatusSingular = 1; // bit 0 of the device status word
constexpr int kStatusIndex = 2;    // bit 1
constexpr int kStatusUnprepared = 4;   // bit 2: a gather on a workspace whose lookup tables were not built for this mesh / canvas

}  // namespace apap
