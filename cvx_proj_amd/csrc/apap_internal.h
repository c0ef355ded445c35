// Internal declarations shared by the host set-up, the kernel launchers and the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/apap_hip.h"

namespace apap {

// Records a thread-local message for apap_last_error() and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Records "<what>: <hipGetErrorString>" and returns APAP_ERR_HIP.
int hip_fail(int hip_error, const char *what);

// Optional per-kernel timing (apap_profile_enable / apap_profile_read): brackets the kernels
// launched in its scope with HIP events on `stream`.
struct ProfScope {
    ProfScope(int slot, void *stream);
    ~ProfScope();
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;

  private:
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
SolvePlan plan_solve(int n, int cells, int variant, int batch = 1);

constexpr int kMoments = 30;       // distinct sums of A^T W^2 A
constexpr int kStatusSingular = 1; // bit 0 of the device status word
constexpr int kStatusIndex = 2;    // bit 1

}  // namespace apap
