// Access-pattern microbenchmark for K3: the strip kernel's unit of work - a lane owns 4 consecutive
// pixels of a canvas row and 4 rows of them, a wave 256 pixels x 4 rows - with NO coordinate arithmetic (a pure translation
// of the C3 source into the C3 canvas), so that what is timed is the access pattern alone.  Load forms x store forms:
//
//   loads    G4   four dword gathers per lane and row at the pixels' byte offsets (today's kernel)
//            W16  one unaligned 16-byte window per lane and row
//            D3   one unaligned 12-byte load per lane and row
//            G4A  four ALIGNED 8-byte loads per lane and row (the two dwords around each pixel), pixel cut out with v_alignbyte
//   stores   U12  one unaligned 12-byte store per lane and row (today's kernel)
//            A16  the row's 768 bytes re-laid through LDS: 48 lanes store one ALIGNED 16-byte piece, head / tail bytes of
//                 the row segment by one byte store
//            D12  the row's bytes shifted to the next dword boundary in registers (one wave shift + three v_alignbyte):
//                 62-63 lanes store a dword-ALIGNED 12-byte piece, head / tail bytes by byte stores
//            N    none (the loads' own time)
//   and a flat 16-byte streaming copy of the same byte counts (the floor of any form).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/k3_access tools/k3_access.hip && tools/k3_access
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        const hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                            \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                               \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef u3 u3u __attribute__((aligned(1)));
typedef u3 u3d __attribute__((aligned(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#ifndef NT
#define NT 0
#endif
struct __attribute__((packed, aligned(1))) B12 { unsigned a, b, c; };
struct __attribute__((packed, aligned(4))) D12 { unsigned a, b, c; };
struct __attribute__((packed, aligned(1))) B16 { unsigned a, b, c, d; };
struct __attribute__((packed, aligned(4))) W5 { unsigned a, b, c, d, e; };

enum { G4 = 0, W16 = 1, D3 = 2, ZERO = 3, G4A = 4 };
enum { U12 = 0, A16 = 1, SD12 = 2, NONE = 3 };

constexpr int kRows = 4;
constexpr int kRowBytes = 768 + 16;      // one wave-row in LDS (+ the fifth dword a piece's read may touch)

template <int L, int S>
__global__ __launch_bounds__(256) void k_copy(const uint8_t *__restrict__ img, int img_h, int img_w, int final_w, int final_h,
                                              int off_x, int off_y, uint8_t *__restrict__ out, int remap) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[S == A16 ? 4 * kRows * kRowBytes : 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // remap (1-D launch): workgroup L runs on XCD L % 8 - give the blocks of one strip-row (x-adjacent: they share the
    // cache lines at their common edges) to ONE XCD, consecutive strip-rows to consecutive XCDs
    int bx = blockIdx.x, by = blockIdx.y;
    if (remap) {
        const int nbx = (final_w + 255) / 256, wg = blockIdx.x, j = wg >> 3;
        bx = j % nbx;
        by = (j / nbx) * 8 + (wg & 7);
    }
    const int j0 = (bx * 64 + lane) * 4;
    const int y_first = (by * 4 + wave) * kRows;
    const int y_end = min(y_first + kRows, final_h);
    if (y_first >= y_end) return;
    if (S != A16 && S != SD12 && j0 >= final_w) return;
    const unsigned last = (unsigned)img_h * (unsigned)img_w * 3u - 4u;
    const int npx = max(0, min(4, final_w - j0));
    unsigned px[kRows][3];
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        const int y = min(y_first + t, y_end - 1), sy = y - off_y;
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sx = j0 + k - off_x;
            const bool ok = ((unsigned)sx < (unsigned)img_w) & ((unsigned)sy < (unsigned)img_h);
            o[k] = ok ? (__umul24((unsigned)sy, (unsigned)img_w) + (unsigned)sx) * 3u : 0xffffffffu;
        }
        if (L == ZERO) {     // no loads at all: the store pattern's own time
            px[t][0] = o[0]; px[t][1] = o[1] ^ o[2]; px[t][2] = o[3];
        } else if (L == G4A) {
            unsigned p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned oc = o[k] < last ? o[k] : last;
                const uint2 v2 = *reinterpret_cast<const uint2 *>(img + (oc & ~3u));       // dword-aligned, 8 bytes
                const unsigned v = __builtin_amdgcn_alignbyte(v2.y, v2.x, oc & 3u);
                p[k] = (unsigned)__builtin_amdgcn_bitop3_b32((int)v, 0x00ffffff, __builtin_amdgcn_sbfe((int)o[k], 31u, 1u), 0x40);
            }
            px[t][0] = p[0] | (p[1] << 24);
            px[t][1] = __builtin_amdgcn_perm(p[2], p[1], 0x05040201u);
            px[t][2] = __builtin_amdgcn_perm(p[3], p[2], 0x06050402u);
        } else if (L == G4) {
            unsigned p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned oc = o[k] < last ? o[k] : last;
                unsigned v;
                __builtin_memcpy(&v, img + oc, 4);
                v = __builtin_amdgcn_alignbyte(0u, v, o[k] - oc);
                p[k] = (unsigned)__builtin_amdgcn_bitop3_b32((int)v, 0x00ffffff, __builtin_amdgcn_sbfe((int)o[k], 31u, 1u), 0x40);
            }
            px[t][0] = p[0] | (p[1] << 24);
            px[t][1] = __builtin_amdgcn_perm(p[2], p[1], 0x05040201u);
            px[t][2] = __builtin_amdgcn_perm(p[3], p[2], 0x06050402u);
        } else {
            // the window form is exact only where the four pixels are consecutive source pixels: here all of them or none
            // apart from the two lanes that straddle the source's left / right edge (their pixels are masked per pixel)
            unsigned base = min(min(o[0], o[1]), min(o[2], o[3]));
            const bool any = base != 0xffffffffu;
            const unsigned lim = 0xfffffff0u;   // (the benchmark allocates slack behind the image)
            // lanes whose window would leave the image: step back (the real kernel keeps the dword gathers for them)
            const unsigned oc = base < lim ? base : lim;
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (any) {
                if (L == W16) {
                    B16 v;
                    __builtin_memcpy(&v, img + oc, 16);
                    w[0] = v.a; w[1] = v.b; w[2] = v.c; w[3] = v.d;
                } else {
                    B12 v;
                    __builtin_memcpy(&v, img + oc, 12);
                    w[0] = v.a; w[1] = v.b; w[2] = v.c;
                }
            }
            // (pixels left of the first valid one would need a shift: in this translation only the lane at the source's
            // left edge; the benchmark's canvas is compared against the G4 form with those lanes' pixels excluded)
            const unsigned m0 = o[0] != 0xffffffffu ? 0xffffffffu : 0u;
            const unsigned m3 = o[3] != 0xffffffffu ? 0xffffffffu : 0u;
            px[t][0] = w[0] & m0;
            px[t][1] = w[1] & m0 & m3;
            px[t][2] = w[2] & m3;
        }
    }
    if (S == NONE) {
        unsigned x = 0;
#pragma unroll
        for (int t = 0; t < kRows; ++t) x ^= px[t][0] ^ px[t][1] ^ px[t][2];
        if (x == 0x12345678u) out[0] = 1;
        return;
    }
    if (S == U12) {
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            const int y = y_first + t;
            if (y >= y_end) break;
            uint8_t *o = out + ((size_t)y * (size_t)final_w) * 3 + (unsigned)j0 * 3u;
            if (npx == 4) {
#if NT
                u3 v = {px[t][0], px[t][1], px[t][2]};
                __builtin_nontemporal_store(v, (u3u *)o);
#else
                B12 v = {px[t][0], px[t][1], px[t][2]};
                __builtin_memcpy(o, &v, 12);
#endif
            } else {
                for (int b = 0; b < 3 * npx; ++b) o[b] = (uint8_t)(px[t][b >> 2] >> (8 * (b & 3)));
            }
        }
        return;
    }
    // the wave's segment of a canvas row: `nb` bytes from x = jw
    const int jw = bx * 256;
    const int nb = 3 * min(256, final_w - jw);
    if (S == A16) {
        uint8_t *mine = lds + wave * (kRows * kRowBytes);
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            D12 v = {px[t][0], px[t][1], px[t][2]};
            __builtin_memcpy(mine + t * kRowBytes + 12 * lane, &v, 12);
        }
        // (wave-private region: no barrier, the compiler's lgkmcnt wait orders the write before the read)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            const int y = y_first + t;
            if (y >= y_end) break;
            uint8_t *row = out + ((size_t)y * (size_t)final_w + (size_t)jw) * 3;
            const unsigned a = (unsigned)(uintptr_t)row;
            const int head = __builtin_amdgcn_readfirstlane(min((int)((0u - a) & 15u), nb));
            const int np = (nb - head) >> 4, tail = nb - head - 16 * np;
            const uint8_t *src = mine + t * kRowBytes;
            if (lane < np) {
                W5 v;
                __builtin_memcpy(&v, src + ((head >> 2) + 4 * lane) * 4, 20);
                const unsigned s = (unsigned)head & 3u;
                uint4 q;
                q.x = __builtin_amdgcn_alignbyte(v.b, v.a, s);
                q.y = __builtin_amdgcn_alignbyte(v.c, v.b, s);
                q.z = __builtin_amdgcn_alignbyte(v.d, v.c, s);
                q.w = __builtin_amdgcn_alignbyte(v.e, v.d, s);
#if NT
                u4 qq = {q.x, q.y, q.z, q.w};
                __builtin_nontemporal_store(qq, (u4 *)(row + head + 16 * lane));
#else
                *reinterpret_cast<uint4 *>(row + head + 16 * lane) = q;
#endif
            }
            // head bytes by lanes 0..15, tail bytes by lanes 16..31
            const int idx = lane < 16 ? lane : nb - tail + (lane - 16);
            const bool on = lane < 16 ? lane < head : (lane - 16) < tail;
#if NT
            if (lane < 32 && on) __builtin_nontemporal_store(src[idx], row + idx);
#else
            if (lane < 32 && on) row[idx] = src[idx];
#endif
        }
        return;
    }
    if (S == SD12) {
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            const int y = y_first + t;
            if (y >= y_end) break;
            uint8_t *row = out + ((size_t)y * (size_t)final_w + (size_t)jw) * 3;
            const unsigned a = (unsigned)(uintptr_t)row;
            const unsigned s = __builtin_amdgcn_readfirstlane((0u - a) & 3u);      // bytes to the next dword boundary
            // the next lane's first dword (wave shift left by one: DPP wave_shl:1)
            const unsigned na = (unsigned)__builtin_amdgcn_update_dpp(0, (int)px[t][0], 0x130, 0xf, 0xf, false);
            const unsigned fa = __builtin_amdgcn_alignbyte(px[t][1], px[t][0], s);
            const unsigned fb = __builtin_amdgcn_alignbyte(px[t][2], px[t][1], s);
            const unsigned fc = __builtin_amdgcn_alignbyte(na, px[t][2], s);
            // lane l's shifted piece = row bytes [s + 12 l, s + 12 l + 12); valid while it ends inside the segment
            const int endb = (int)s + 12 * lane + 12;
            D12 v = {fa, fb, fc};
            if (endb <= nb) {
                __builtin_memcpy(row + s + 12 * lane, &v, 12);
            } else if (endb - 12 < nb) {     // the last lane with bytes: what is left, byte by byte
                for (int b = 0; b < nb - (endb - 12); ++b) row[s + 12 * lane + b] = (uint8_t)((b < 4 ? fa : b < 8 ? fb : fc) >> (8 * (b & 3)));
            }
            // head: the first s bytes of the segment are lane 0's first bytes
            const unsigned h0 = __builtin_amdgcn_readfirstlane(px[t][0]);
            if (lane < (int)s && lane < nb) row[lane] = (uint8_t)(h0 >> (8 * lane));
        }
        return;
    }
}

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ a, size_t na, uint4 *__restrict__ b, size_t nbv) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
    uint4 acc = {0, 0, 0, 0};
    for (size_t k = i; k < na; k += n) {
        const uint4 v = a[k];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
#if NT
    for (size_t k = i; k < nbv; k += n) { u4 qq = {acc.x, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(qq, (u4 *)(b + k)); }
#else
    for (size_t k = i; k < nbv; k += n) b[k] = acc;
#endif
}

template <int L, int S>
static int run(const char *name, const uint8_t *d_img, int ih, int iw, int fw, int fh, int ox, int oy, uint8_t *d_out,
               std::vector<uint8_t> *ref, const uint8_t *d_img2, uint8_t *d_out2, int remap = 0) {
    const dim3 grid = remap ? dim3((unsigned)(((fw + 255) / 256) * (((fh + 15) / 16 + 7) / 8 * 8)), 1) : dim3((fw + 255) / 256, (fh + 15) / 16);
    const dim3 block(256);
    const size_t ob = (size_t)fw * fh * 3;
    CK(hipMemset(d_out, 0xee, ob));
    k_copy<L, S><<<grid, block>>>(d_img, ih, iw, fw, fh, ox, oy, d_out, remap);
    CK(hipDeviceSynchronize());
    const char *verdict = "";
    if (S != NONE && L != ZERO) {
        std::vector<uint8_t> got(ob);
        CK(hipMemcpy(got.data(), d_out, ob, hipMemcpyDeviceToHost));
        if (ref->empty()) {
            *ref = got;
            verdict = "(reference)";
        } else {
            // G4 masks per pixel, the window forms per lane: exclude lanes that straddle the source's edge columns
            size_t bad = 0;
            for (int y = 0; y < fh; ++y)
                for (int x = 0; x < fw; ++x) {
                    const int g = x & ~3;
                    const bool straddle = (g - ox < 0 && g + 3 - ox >= 0) || (g - ox < iw && g + 3 - ox >= iw);
                    if (straddle) continue;
                    const size_t o = ((size_t)y * fw + x) * 3;
                    bad += got[o] != (*ref)[o] || got[o + 1] != (*ref)[o + 1] || got[o + 2] != (*ref)[o + 2];
                }
            verdict = bad == 0 ? "same canvas" : "CANVAS DIFFERS";
            if (bad) printf("   %zu pixels differ\n", bad);
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, ms_cold = 0;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 300; ++i) k_copy<L, S><<<grid, block>>>(d_img, ih, iw, fw, fh, ox, oy, d_out, remap);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) k_copy<L, S><<<grid, block>>>(d_img, ih, iw, fw, fh, ox, oy, d_out, remap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    {   // cold: 16 image / canvas sets (832 MB) in rotation
        const size_t ib = (size_t)ih * iw * 3, istep = (ib + 255) / 256 * 256, ostep = (ob + 255) / 256 * 256;
        for (int i = 0; i < 64; ++i) k_copy<L, S><<<grid, block>>>(d_img2 + (i % 16) * istep, ih, iw, fw, fh, ox, oy, d_out2 + (i % 16) * ostep, remap);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 160; ++i) k_copy<L, S><<<grid, block>>>(d_img2 + (i % 16) * istep, ih, iw, fw, fh, ox, oy, d_out2 + (i % 16) * ostep, remap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_cold, e0, e1));
        ms_cold /= 160;
    }
    printf("%-12s%s warm %6.2f us   cold %6.2f us   %s\n", name, remap ? " XCD" : "    ", best / 200 * 1e3, ms_cold * 1e3, verdict);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const int iw = 3840, ih = 2160, fw = argc > 1 ? atoi(argv[1]) : 4009, fh = argc > 2 ? atoi(argv[2]) : 2242, ox = 84, oy = 41;
    const size_t ib = (size_t)ih * iw * 3, ob = (size_t)fw * fh * 3;
    const size_t istep = (ib + 255) / 256 * 256, ostep = (ob + 255) / 256 * 256;
    uint8_t *d_img, *d_out, *d_img2, *d_out2;
    CK(hipMalloc(&d_img, ib + 256));
    CK(hipMalloc(&d_out, ob + 256));
    CK(hipMalloc(&d_img2, istep * 16 + 256));
    CK(hipMalloc(&d_out2, ostep * 16));
    std::vector<uint8_t> h(ib);
    uint32_t s = 12345;
    for (size_t i = 0; i < ib; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (uint8_t)(s >> 24);
    }
    CK(hipMemcpy(d_img, h.data(), ib, hipMemcpyHostToDevice));
    for (int i = 0; i < 16; ++i) CK(hipMemcpy(d_img2 + i * istep, h.data(), ib, hipMemcpyHostToDevice));
    printf("NT = %d (1: non-temporal stores)\n", NT);
    printf("canvas %d x %d, source %d x %d; in-range pixels read 3 B and write 3 B, blank ones write 3 B\n", fw, fh, iw, ih);
    std::vector<uint8_t> ref;
#define RUN(L, S) if (run<L, S>(#L "+" #S, d_img, ih, iw, fw, fh, ox, oy, d_out, &ref, d_img2, d_out2)) return 1
    RUN(G4, U12);
    RUN(G4, A16);
    RUN(G4, SD12);
    RUN(G4, NONE);
    RUN(W16, U12);
    RUN(W16, A16);
    RUN(W16, SD12);
    RUN(W16, NONE);
    RUN(D3, U12);
    RUN(D3, A16);
    RUN(D3, SD12);
    RUN(D3, NONE);
    RUN(G4A, U12);
    RUN(G4A, NONE);
    RUN(ZERO, U12);
    RUN(ZERO, A16);
    RUN(ZERO, SD12);
#define RUNX(L, S) if (run<L, S>(#L "+" #S, d_img, ih, iw, fw, fh, ox, oy, d_out, &ref, d_img2, d_out2, 1)) return 1
    RUNX(G4, U12);
    RUNX(G4, A16);
    RUNX(G4, NONE);
    RUNX(W16, U12);
    RUNX(W16, A16);
    RUNX(ZERO, U12);
    RUNX(ZERO, A16);
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 300; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, ib / 16, (uint4 *)d_out, ob / 16);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, ib / 16, (uint4 *)d_out, ob / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s warm %6.2f us   (flat 16-byte loads of the source, flat 16-byte stores of the canvas)\n", "stream", ms / 200 * 1e3);
        for (int i = 0; i < 300; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, ib / 16, (uint4 *)d_out, 0);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, ib / 16, (uint4 *)d_out, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s warm %6.2f us   (the loads alone)\n", "stream-read", ms / 200 * 1e3);
        for (int i = 0; i < 300; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, 0, (uint4 *)d_out, ob / 16);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, 0, (uint4 *)d_out, ob / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s warm %6.2f us   (the stores alone)\n", "stream-fill", ms / 200 * 1e3);
        for (int i = 0; i < 300; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, 0, (uint4 *)d_out, 0);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) k_stream<<<2048, 256>>>((const uint4 *)d_img, 0, (uint4 *)d_out, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s warm %6.2f us   (an empty kernel of the same grid: the launch cadence)\n", "stream-none", ms / 200 * 1e3);
    }
    return 0;
}
