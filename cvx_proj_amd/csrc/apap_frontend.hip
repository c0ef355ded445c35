// Callers on either side of the APAP hot path, device half:
//   * per-channel histogram equalisation, the pre-processing of apap.py:236-237
//     (utils.py:85-91: np.stack([cv.equalizeHist(img[..., i]) ...])),
//   * the seed homography of baseline_stitch_test.py:42 (cv.findHomography, RANSAC, 5 px).
// Byte / integer work bound by HBM (equalisation) or by launch latency (RANSAC: a few
// thousand 8 x 8 solves and a few million point tests).  No MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "apap_internal.h"

namespace {

inline int hip_fail(hipError_t e, const char *what) { return apap::hip_fail((int)e, what); }

// --------------------------------------------------------------------------------
// Histogram equalisation of an interleaved uint8 image, all channels in one pass.
// Both kernels stream the image in wave-sized chunks of 3 KiB: three global_load_dwordx4 per
// lane, each instruction covering 1 KiB of consecutive bytes.  The channel of byte j of the
// vector a lane loads is (head + 16 v + j) % C; 3 KiB is a multiple of every channel count
// (1..4), so that phase is a per-lane constant and the channel of each of the lane's 48 bytes
// selects one of C per-lane base addresses at compile time.
// E1 k_eq_hist: per-wave private histograms in LDS (ds_add_u32), summed per block and added to
//    one of kEqReplicas global histograms (same-address L2 atomics limit a single one).
// E1b k_eq_lut: one block sums the replicas, zeroes them again (the workspace contract: zero on
//    entry, zero on return - no memset per call) and builds the C x 256 lookup table (ballots
//    and wave scans, all channels at once).
// E2 k_eq_apply: maps every byte through the table (768 bytes in LDS, address formed by one
//    v_perm_b32) and stores 1 KiB per instruction.
// Algorithmic HBM traffic: 3 bytes per byte of image (read, read, write).
// --------------------------------------------------------------------------------
constexpr int kEqThreads = 256;
constexpr int kEqWaves = kEqThreads / 64;
constexpr int kEqMaxChannels = 4;
constexpr int kEqReplicas = 16;
constexpr int kEqChunkVecs = 192;  // 16-byte vectors per wave per step (3 KiB)

// workspace: replicas x C x 256 counters | 16 spare bytes | C x 256 table bytes
__host__ __device__ constexpr size_t eq_ticket_offset(int c) { return (size_t)kEqReplicas * c * 256 * sizeof(unsigned int); }
__host__ __device__ constexpr size_t eq_lut_offset(int c) { return eq_ticket_offset(c) + 16; }
__host__ __device__ constexpr size_t eq_workspace(int c) { return eq_lut_offset(c) + (size_t)c * 256; }

struct EqSplit {
    size_t head;    // bytes before the first 16-byte boundary
    size_t chunks;  // 3 KiB chunks of the aligned body
    size_t tail;    // first byte after the body
};

__device__ __forceinline__ EqSplit eq_split(const uint8_t *img, size_t bytes) {
    EqSplit s;
    s.head = min(bytes, (size_t)((16 - ((uintptr_t)img & 15)) & 15));
    s.chunks = (bytes - s.head) / (kEqChunkVecs * 16);
    s.tail = s.head + s.chunks * (kEqChunkVecs * 16);
    return s;
}

// LDS byte address of a __shared__ object, and a byte load through such an address
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
__device__ __forceinline__ unsigned int lds_addr(const uint8_t *p) { return (unsigned int)(uintptr_t)(const lds_u8_t *)p; }
__device__ __forceinline__ unsigned int lds_load_u8(unsigned int addr) { return *(const lds_u8_t *)(uintptr_t)addr; }

// phase step between the three vectors a lane loads (they are 64 vectors = 1 KiB apart)
template <int C>
constexpr int kEqStep = 1024 % C;

// cv::equalizeHist's tables for all C channels at once, 256 threads: thread t owns bin t of
// every channel and holds its count in `mine`.  Wave-level ballots and shuffles, three barriers.
// lut[i] for i below the first non-empty bin is 0 (no pixel has such a value).
template <int C>
__device__ void eq_build_luts(const unsigned int (&mine)[C], uint8_t *lut /* C x 256 */, int total) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ int s_first[C][kEqWaves];          // first non-empty bin of each wave's 64 bins (or 256)
    __shared__ unsigned int s_count[C][kEqWaves];  // that bin's count
    __shared__ unsigned int s_sum[C][kEqWaves];    // sum of the wave's bins after the channel's first bin
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const unsigned long long m = __ballot(mine[c] != 0u);
        const int f = m ? __builtin_ctzll(m) : 64;
        const unsigned int cnt = __shfl(mine[c], f & 63);
        if (lane == 0) {
            s_first[c][wave] = m ? wave * 64 + f : 256;
            s_count[c][wave] = m ? cnt : 0u;
        }
    }
    __syncthreads();
    int i0[C];
    unsigned int first_count[C], scan[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        i0[c] = 256;
        first_count[c] = 0u;
#pragma unroll
        for (int w = kEqWaves - 1; w >= 0; --w) {
            if (s_first[c][w] != 256) {
                i0[c] = s_first[c][w];
                first_count[c] = s_count[c][w];
            }
        }
        // inclusive scan over the wave of the bins after i0
        unsigned int v = t > i0[c] ? mine[c] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int up = __shfl_up(v, d);
            if (lane >= d) v += up;
        }
        scan[c] = v;
        if (lane == 63) s_sum[c][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int w = 0; w < kEqWaves - 1; ++w) scan[c] += w < wave ? s_sum[c][w] : 0u;
        uint8_t v;
        if (i0[c] == 256 || (int)first_count[c] == total) {
            v = (uint8_t)t;  // constant plane: the output is the input
        } else if (t <= i0[c]) {
            v = 0;
        } else {
            const float scale = 255.0f / (float)(total - (int)first_count[c]);  // correctly rounded (hipcc default)
            const int r = __float2int_rn((float)(int)scan[c] * scale);          // cvRound: half to even
            v = (uint8_t)min(max(r, 0), 255);
        }
        lut[c * 256 + t] = v;
    }
}

template <int C>
__global__ __launch_bounds__(kEqThreads) void k_eq_hist(const uint8_t *__restrict__ img, size_t bytes,
                                                        unsigned char *__restrict__ work) {
    __shared__ unsigned int h[kEqWaves][C * 256];
    unsigned int *hist = reinterpret_cast<unsigned int *>(work);
    for (int i = threadIdx.x; i < kEqWaves * C * 256; i += kEqThreads) (&h[0][0])[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned int *mine = h[threadIdx.x >> 6];
    const EqSplit sp = eq_split(img, bytes);
    if (blockIdx.x == 0) {  // the ends that do not fill a chunk, one byte per thread
        for (size_t i = threadIdx.x; i < sp.head; i += kEqThreads) atomicAdd(&mine[(i % C) * 256 + img[i]], 1u);
        for (size_t i = sp.tail + threadIdx.x; i < bytes; i += kEqThreads) atomicAdd(&mine[(i % C) * 256 + img[i]], 1u);
    }
    {
        // per-lane histogram bases: off[m] is where the lane counts a byte whose position in its
        // vector is m modulo C
        const int phi = (int)((sp.head + 16u * (unsigned)lane) % C);
        int off[C];  // indices, not pointers: the compiler must keep seeing LDS (ds_add_u32, not flat atomics)
#pragma unroll
        for (int m = 0; m < C; ++m) off[m] = ((phi + m) % C) * 256;
        const uint4 *vec = reinterpret_cast<const uint4 *>(img + sp.head);
        const size_t wave = (size_t)blockIdx.x * kEqWaves + (threadIdx.x >> 6);
        const size_t nwaves = (size_t)gridDim.x * kEqWaves;
        size_t c = wave;
        if (c < sp.chunks) {
            // register double buffer: the next chunk's loads fly during this chunk's 48 LDS atomics
            const uint4 *p = vec + c * kEqChunkVecs + lane;
            uint4 q0 = p[0], q1 = p[64], q2 = p[128];
            while (true) {
                const size_t cn = c + nwaves;
                uint4 n0 = q0, n1 = q1, n2 = q2;
                if (cn < sp.chunks) {
                    const uint4 *pn = vec + cn * kEqChunkVecs + lane;
                    n0 = pn[0]; n1 = pn[64]; n2 = pn[128];
                }
                const unsigned int w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                for (int k = 0; k < 12; ++k) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int m = ((k / 4) * kEqStep<C> + 4 * (k % 4) + b) % C;  // compile-time
                        atomicAdd(&mine[off[m] + (int)((w[k] >> (8 * b)) & 0xffu)], 1u);
                    }
                }
                if (cn >= sp.chunks) break;
                q0 = n0; q1 = n1; q2 = n2;
                c = cn;
            }
        }
    }
    __syncthreads();
    unsigned int *replica = hist + (size_t)(blockIdx.x % kEqReplicas) * C * 256;
    for (int i = threadIdx.x; i < C * 256; i += kEqThreads) {
        unsigned int s = 0;
#pragma unroll
        for (int w = 0; w < kEqWaves; ++w) s += h[w][i];
        if (s) atomicAdd(&replica[i], s);
    }
}

// E1b: one block sums the replicas, leaves them zeroed for the next call, and builds the table.
// (Doing this in the last block of k_eq_hist to arrive needs a device-scope fence in every
// block; on this 8-L2 part that fence took the kernel from 9 to 76 us.)
template <int C>
__global__ __launch_bounds__(kEqThreads) void k_eq_lut(int total, unsigned char *__restrict__ work) {
    unsigned int *hist = reinterpret_cast<unsigned int *>(work);
    unsigned int v[C][kEqReplicas];  // all C x kEqReplicas loads in flight together
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int r = 0; r < kEqReplicas; ++r) v[c][r] = hist[(size_t)r * C * 256 + c * 256 + threadIdx.x];
    }
    unsigned int mine[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        unsigned int s = 0;
#pragma unroll
        for (int r = 0; r < kEqReplicas; ++r) {
            s += v[c][r];
            hist[(size_t)r * C * 256 + c * 256 + threadIdx.x] = 0u;
        }
        mine[c] = s;
    }
    eq_build_luts<C>(mine, work + eq_lut_offset(C), total);
}

template <int C>
__global__ __launch_bounds__(kEqThreads) void k_eq_apply(const uint8_t *__restrict__ img, size_t bytes,
                                                         const unsigned char *__restrict__ work,
                                                         uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(256))) uint8_t s_lut[C * 256];
    const int lane = threadIdx.x & 63;
    const EqSplit sp = eq_split(img, bytes);
    const uint4 *vec = reinterpret_cast<const uint4 *>(img + sp.head);
    const size_t wave = (size_t)blockIdx.x * kEqWaves + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * kEqWaves;
    // issue the first chunk's loads before the table is fetched
    size_t c = wave;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0;
    if (c < sp.chunks) {
        const uint4 *p = vec + c * kEqChunkVecs + lane;
        q0 = p[0]; q1 = p[64]; q2 = p[128];
    }
    for (int i = threadIdx.x; i < C * 64; i += kEqThreads)
        reinterpret_cast<unsigned int *>(s_lut)[i] = reinterpret_cast<const unsigned int *>(work + eq_lut_offset(C))[i];
    __syncthreads();
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < sp.head; i += kEqThreads) out[i] = s_lut[(i % C) * 256 + img[i]];
        for (size_t i = sp.tail + threadIdx.x; i < bytes; i += kEqThreads) out[i] = s_lut[(i % C) * 256 + img[i]];
    }
    // per-lane table bases as LDS byte addresses (256-aligned: the byte value replaces the low
    // byte with one v_perm_b32)
    const int phi = (int)((sp.head + 16u * (unsigned)lane) % C);
    unsigned int off[C];
#pragma unroll
    for (int m = 0; m < C; ++m)
        off[m] = lds_addr(s_lut + ((phi + m) % C) * 256);
    uint8_t *obody = out + sp.head;
    const bool out_aligned = (((uintptr_t)obody) & 15) == 0;
    while (c < sp.chunks) {
        const size_t cn = c + nwaves;
        uint4 n0 = q0, n1 = q1, n2 = q2;
        if (cn < sp.chunks) {
            const uint4 *pn = vec + cn * kEqChunkVecs + lane;
            n0 = pn[0]; n1 = pn[64]; n2 = pn[128];
        }
        const unsigned int w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        unsigned int r[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            unsigned int acc = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = ((k / 4) * kEqStep<C> + 4 * (k % 4) + b) % C;  // compile-time
                // LDS address = table base with its low byte replaced by byte b of w[k]
                const unsigned int addr = __builtin_amdgcn_perm(off[m], w[k], 0x07060500u | (unsigned)b);
                const unsigned int v = lds_load_u8(addr);
                acc |= v << (8 * b);
            }
            r[k] = acc;
        }
        uint8_t *o = obody + (c * kEqChunkVecs + lane) * 16;
        if (out_aligned) {
            // non-temporal: written once, never read back here (a flat 16-byte copy of 50 MB runs 8.5 -> 6.3 us with the
            // hint on its stores, tools/k3_access.hip)
            typedef unsigned Dwords4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(Dwords4{r[0], r[1], r[2], r[3]}, reinterpret_cast<Dwords4 *>(o));
            __builtin_nontemporal_store(Dwords4{r[4], r[5], r[6], r[7]}, reinterpret_cast<Dwords4 *>(o + 1024));
            __builtin_nontemporal_store(Dwords4{r[8], r[9], r[10], r[11]}, reinterpret_cast<Dwords4 *>(o + 2048));
        } else {
            // `out` has another alignment than `img`: unaligned dword stores
#pragma unroll
            for (int k = 0; k < 12; ++k) __builtin_memcpy(o + 1024 * (k / 4) + 4 * (k % 4), &r[k], 4);
        }
        q0 = n0; q1 = n1; q2 = n2;
        c = cn;
    }
}

template <int C>
int launch_equalize(apap_ctx *ctx, const uint8_t *d_img, size_t bytes, int total, uint8_t *d_out, unsigned char *work, hipStream_t s) {
    // 4 blocks per CU when the image is large: enough loads in flight, few enough blocks that
    // the per-block histogram flush stays small next to the streaming
    const size_t chunks = bytes / (kEqChunkVecs * 16) + 1;
    const unsigned blocks = (unsigned)min((size_t)1024, (chunks + kEqWaves - 1) / kEqWaves);
    {
        apap::ProfScope prof(ctx, APAP_PROF_EQ_HIST, s);
        hipLaunchKernelGGL(k_eq_hist<C>, dim3(blocks), dim3(kEqThreads), 0, s, d_img, bytes, work);
        hipLaunchKernelGGL(k_eq_lut<C>, dim3(1), dim3(kEqThreads), 0, s, total, work);
    }
    {
        apap::ProfScope prof(ctx, APAP_PROF_EQ_APPLY, s);
        hipLaunchKernelGGL(k_eq_apply<C>, dim3(blocks), dim3(kEqThreads), 0, s, d_img, bytes, (const unsigned char *)work,
                           d_out);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_equalize_hist_device launch");
    return APAP_OK;
}

// --------------------------------------------------------------------------------
// Seed homography by RANSAC (contract of cv.findHomography(src, dst, cv.RANSAC, thresh),
// baseline_stitch_test.py:42; specification: oracle/frontend_oracle.py, bit for bit).
// R1 k_ransac_hyp: one lane per hypothesis - counter-based sampler (splitmix64), 8 x 8 Gaussian
//    elimination with partial pivoting on a lane-private 8 x 9 system kept in LDS (dynamic row
//    indices; index-major layout, so the 64 lanes never conflict).  Products and differences are
//    rounded separately (the library is built with -ffp-contract=off), as in the oracle.
// R2 k_ransac_score: one block per hypothesis counts the points within the threshold.
// R3 k_ransac_select: first hypothesis with the most inliers, its matrix and its mask.
// A few thousand tiny solves and a few million point tests: launch-latency-bound, no roofline.
// --------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr int kHypLanes = 64;

__global__ __launch_bounds__(kHypLanes) void k_ransac_hyp(const float *__restrict__ src, const float *__restrict__ dst,
                                                          int n, int iterations, unsigned long long seed,
                                                          double *__restrict__ H /* iterations x 9 */) {
    __shared__ double sys[72][kHypLanes];
    const int t = threadIdx.x;
    const int h = blockIdx.x * kHypLanes + t;
    if (h >= iterations) return;  // lane-private work below: no barriers
#define A(r, c) sys[(r) * 9 + (c)][t]
    // four distinct indices: draw j is taken modulo n - j, then stepped over the earlier picks
    // in ascending order
    int pick[4];
    for (int j = 0; j < 4; ++j) {
        const unsigned int r = (unsigned int)(splitmix64(seed + 4ull * (unsigned long long)h + (unsigned long long)j) >> 32);
        int p = (int)(r % (unsigned int)(n - j));
        int prev[3];
        for (int k = 0; k < j; ++k) prev[k] = pick[k];
        for (int a = 1; a < j; ++a) {  // insertion sort of at most three earlier picks
            const int v = prev[a];
            int b = a - 1;
            while (b >= 0 && prev[b] > v) { prev[b + 1] = prev[b]; --b; }
            prev[b + 1] = v;
        }
        for (int k = 0; k < j; ++k) p += (p >= prev[k]) ? 1 : 0;
        pick[j] = p;
    }
    for (int j = 0; j < 4; ++j) {
        const double x = (double)src[2 * pick[j]], y = (double)src[2 * pick[j] + 1];
        const double u = (double)dst[2 * pick[j]], v = (double)dst[2 * pick[j] + 1];
        const int r0 = 2 * j, r1 = 2 * j + 1;
        A(r0, 0) = x; A(r0, 1) = y; A(r0, 2) = 1.0; A(r0, 3) = 0.0; A(r0, 4) = 0.0; A(r0, 5) = 0.0;
        A(r0, 6) = -(u * x); A(r0, 7) = -(u * y); A(r0, 8) = u;
        A(r1, 0) = 0.0; A(r1, 1) = 0.0; A(r1, 2) = 0.0; A(r1, 3) = x; A(r1, 4) = y; A(r1, 5) = 1.0;
        A(r1, 6) = -(v * x); A(r1, 7) = -(v * y); A(r1, 8) = v;
    }
    bool ok = true;
    for (int c = 0; c < 8; ++c) {
        int p = c;
        double best = fabs(A(c, c));
        for (int r = c + 1; r < 8; ++r) {
            const double v = fabs(A(r, c));
            if (v > best) { best = v; p = r; }  // first largest
        }
        if (p != c) {
            for (int cc = 0; cc < 9; ++cc) {
                const double tmp = A(p, cc);
                A(p, cc) = A(c, cc);
                A(c, cc) = tmp;
            }
        }
        const double piv = A(c, c);
        ok = ok && piv != 0.0 && isfinite(piv);
        for (int r = c + 1; r < 8; ++r) {
            const double f = A(r, c) / piv;
            for (int cc = c; cc < 9; ++cc) {
                const double prod = f * A(c, cc);
                A(r, cc) = A(r, cc) - prod;
            }
        }
    }
    double sol[8];
    for (int i = 7; i >= 0; --i) {
        double s = A(i, 8);
        for (int j = i + 1; j < 8; ++j) {
            const double prod = A(i, j) * sol[j];
            s = s - prod;
        }
        sol[i] = s / A(i, i);
    }
#undef A
    const double nan = __builtin_nan("");
    for (int i = 0; i < 8; ++i) H[(size_t)h * 9 + i] = ok ? sol[i] : nan;
    H[(size_t)h * 9 + 8] = ok ? 1.0 : nan;
}

// squared forward reprojection error, operations in the oracle's order
__device__ __forceinline__ bool ransac_inlier(const double (&h)[9], double x, double y, double u, double v, double thr2) {
    const double w = (h[6] * x + h[7] * y) + h[8];
    const double px = ((h[0] * x + h[1] * y) + h[2]) / w;
    const double py = ((h[3] * x + h[4] * y) + h[5]) / w;
    const double dx = px - u, dy = py - v;
    return dx * dx + dy * dy <= thr2;  // NaN compares false
}

__global__ __launch_bounds__(256) void k_ransac_score(const float *__restrict__ src, const float *__restrict__ dst, int n,
                                                      const double *__restrict__ H, double thr2,
                                                      int *__restrict__ counts) {
    __shared__ int s_part[4];
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = H[(size_t)blockIdx.x * 9 + i];
    int c = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float2 s = reinterpret_cast<const float2 *>(src)[k];
        const float2 d = reinterpret_cast<const float2 *>(dst)[k];
        c += ransac_inlier(h, (double)s.x, (double)s.y, (double)d.x, (double)d.y, thr2) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void k_ransac_select(const float *__restrict__ src, const float *__restrict__ dst, int n,
                                                       const double *__restrict__ H, const int *__restrict__ counts,
                                                       int iterations, double thr2, double *__restrict__ H_best,
                                                       uint8_t *__restrict__ mask, int *__restrict__ result) {
    __shared__ int s_cnt[256], s_idx[256];
    int bc = -1, bi = 0x7fffffff;
    for (int i = threadIdx.x; i < iterations; i += 256) {  // ascending: strictly greater keeps the first
        const int c = counts[i];
        if (c > bc) { bc = c; bi = i; }
    }
    s_cnt[threadIdx.x] = bc;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            const int oc = s_cnt[threadIdx.x + d], oi = s_idx[threadIdx.x + d];
            if (oc > s_cnt[threadIdx.x] || (oc == s_cnt[threadIdx.x] && oi < s_idx[threadIdx.x])) {
                s_cnt[threadIdx.x] = oc;
                s_idx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int best = s_idx[0];
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = H[(size_t)best * 9 + i];
    if (threadIdx.x < 9) H_best[threadIdx.x] = h[threadIdx.x];
    if (threadIdx.x == 0) {
        result[0] = best;
        result[1] = s_cnt[0];
    }
    for (int k = threadIdx.x; k < n; k += 256) {
        const float2 s = reinterpret_cast<const float2 *>(src)[k];
        const float2 d = reinterpret_cast<const float2 *>(dst)[k];
        mask[k] = ransac_inlier(h, (double)s.x, (double)s.y, (double)d.x, (double)d.y, thr2) ? 1 : 0;
    }
}

}  // namespace

extern "C" {

size_t apap_ransac_workspace_bytes(int n, int iterations) {
    if (n < 4 || iterations < 1) return 0;
    return (size_t)iterations * 9 * sizeof(double) + (size_t)iterations * sizeof(int);
}

int apap_ransac_device(apap_ctx *ctx, const float *d_src, const float *d_dst, int n, double thresh, int iterations,
                       unsigned long long seed, double *d_H_best, uint8_t *d_mask, int *d_result, void *d_work,
                       size_t work_bytes, void *stream) {
    if (!d_src || !d_dst || !d_H_best || !d_mask || !d_result || !d_work)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_ransac_device: null device pointer");
    if (n < 4 || iterations < 1 || iterations > (1 << 24) || !(thresh >= 0.0))
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_ransac_device: n=%d (need >= 4) iterations=%d thresh=%g", n,
                          iterations, thresh);
    if (work_bytes < apap_ransac_workspace_bytes(n, iterations))
        return apap::fail(APAP_ERR_WORKSPACE, "apap_ransac_device: workspace %zu < %zu bytes", work_bytes,
                          apap_ransac_workspace_bytes(n, iterations));
    if (((uintptr_t)d_work & 7) != 0 || ((uintptr_t)d_src & 7) != 0 || ((uintptr_t)d_dst & 7) != 0)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_ransac_device: points and workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double *H = (double *)d_work;
    int *counts = (int *)(H + (size_t)iterations * 9);
    const double thr2 = thresh * thresh;
    {
        apap::ProfScope prof(ctx, APAP_PROF_RANSAC, s);
        hipLaunchKernelGGL(k_ransac_hyp, dim3((iterations + kHypLanes - 1) / kHypLanes), dim3(kHypLanes), 0, s, d_src,
                           d_dst, n, iterations, seed, H);
        hipLaunchKernelGGL(k_ransac_score, dim3(iterations), dim3(256), 0, s, d_src, d_dst, n, (const double *)H, thr2,
                           counts);
        hipLaunchKernelGGL(k_ransac_select, dim3(1), dim3(256), 0, s, d_src, d_dst, n, (const double *)H,
                           (const int *)counts, iterations, thr2, d_H_best, d_mask, d_result);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_ransac_device launch");
    return APAP_OK;
}

size_t apap_equalize_workspace_bytes(int channels) {
    if (channels < 1 || channels > kEqMaxChannels) return 0;
    return eq_workspace(channels);
}

int apap_equalize_hist_device(apap_ctx *ctx, const uint8_t *d_img, int h, int w, int channels, uint8_t *d_out, void *d_work,
                              size_t work_bytes, void *stream) {
    if (!d_img || !d_out || !d_work) return apap::fail(APAP_ERR_INVALID_ARG, "apap_equalize_hist_device: null device pointer");
    if (h < 1 || w < 1 || channels < 1 || channels > kEqMaxChannels)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_equalize_hist_device: h=%d w=%d channels=%d (1..4 channels)", h, w,
                          channels);
    if ((unsigned long long)h * (unsigned long long)w >= (1ull << 31))
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_equalize_hist_device: plane of 2^31 pixels or more");
    if (work_bytes < apap_equalize_workspace_bytes(channels))
        return apap::fail(APAP_ERR_WORKSPACE, "apap_equalize_hist_device: workspace %zu < %zu bytes", work_bytes,
                          apap_equalize_workspace_bytes(channels));
    if (((uintptr_t)d_work & 15) != 0) return apap::fail(APAP_ERR_INVALID_ARG, "apap_equalize_hist_device: workspace not 16-byte aligned");
    const int total = h * w;
    const size_t bytes = (size_t)total * channels;
    unsigned char *hist = (unsigned char *)d_work;
    hipStream_t s = (hipStream_t)stream;
    switch (channels) {
        case 1: return launch_equalize<1>(ctx, d_img, bytes, total, d_out, hist, s);
        case 2: return launch_equalize<2>(ctx, d_img, bytes, total, d_out, hist, s);
        case 3: return launch_equalize<3>(ctx, d_img, bytes, total, d_out, hist, s);
        default: return launch_equalize<4>(ctx, d_img, bytes, total, d_out, hist, s);
    }
}

}  // extern "C"
