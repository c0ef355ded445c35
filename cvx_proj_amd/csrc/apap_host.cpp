// Host-side set-up of the APAP engine: the once-per-pair arithmetic that the
// reference's APAP.local_homography performs before its cell loop (apap.py:132-145),
// restated in C++ with the float32 / float64 roundings numpy applies, so that the
// numbers handed to the GPU kernels are the ones the reference feeds its own loop.
//
// Build note: this file must be compiled with -ffp-contract=off.  Several lines rely
// on a float32 product being rounded BEFORE the following add (numpy evaluates
// "a * b + c" as two ufunc calls); a fused multiply-add would change the last bit.
#include <cmath>
#include <cstring>
#include <vector>

#include "apap_internal.h"

namespace {

// The reference runs its set-up in the dtype of the keypoints it is handed (apap.py:35-100: nothing there casts its
// argument): float32 keypoints - what utils.get_features returns, apap.py:241-243 - stay float32 throughout, float64
// keypoints stay float64 until the entries of the 3 x 3 matrices and of the DLT rows are rounded into float32 arrays
// (apap.py:53-55, 85-87, 104).  One template serves both; T is the dtype of ONE point set (the two sets need not agree).

// numpy's add.reduce over a contiguous 1-D array of T (pairwise summation with an 8-way unrolled leaf of at most
// 128 elements).  np.mean of a 1-D array uses it (apap.py:49).
template <typename T>
T pairwise_sum(const T *a, long n) {
    if (n < 8) {
        T res = 0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        T r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// numpy's add.reduce over n contiguous values: the reduction iterator hands the inner loop at most 8192 elements
// (the ufunc buffer size) at a time, each piece is summed pairwise and the pieces are accumulated left to right.
// Verified against numpy for n up to 50 000, float32 and float64, for 1-D arrays and for the column-major view of
// getNormalize2DPts' output.
template <typename T>
T numpy_sum(const T *a, long n) {
    const long kBuf = 8192;
    if (n <= kBuf) return pairwise_sum(a, n);
    T res = pairwise_sum(a, kBuf);
    for (long lo = kBuf; lo < n; lo += kBuf) res += pairwise_sum(a + lo, n - lo < kBuf ? n - lo : kBuf);
    return res;
}

// add.reduce along axis 0 of an (n, 2) C-contiguous array: numpy walks the rows and adds each into the 2-element
// output, i.e. plain sequential sums.
template <typename T>
void column_sums(const T *p, int n, T &s0, T &s1) {
    s0 = 0;
    s1 = 0;
    for (int i = 0; i < n; ++i) {
        s0 += p[2 * i];
        s1 += p[2 * i + 1];
    }
}

// numpy divides a float32 sum by the intp count in float64 and casts back; a float64 sum in float64.
template <typename T>
inline T div_count(T s, int n) { return (T)((double)s / (double)n); }

// APAP.getNormalize2DPts, apap.py:35-59.  `out`: the normalised points in T (a float32 set stays float32: sgemm;
// a float64 set is multiplied by the float32 matrix widened: dgemm).
template <typename T>
void normalize_2d_pts(const T *pts, int n, float t[9], T *out) {
    T s0, s1;
    column_sums(pts, n, s0, s1);
    const T c0 = div_count(s0, n), c1 = div_count(s1, n);
    std::vector<T> dist((size_t)n);
    for (int i = 0; i < n; ++i) {
        const T dx = pts[2 * i] - c0, dy = pts[2 * i + 1] - c1;
        const T sx = dx * dx, sy = dy * dy;
        dist[i] = std::sqrt(sx + sy);
    }
    const T mean_dist = div_count(numpy_sum(dist.data(), n), n);
    // np.float32 + python float stays float32 (NEP 50); np.sqrt(2) is a float64.
    const T denom = mean_dist + (T)1e-8;
    const double scale = std::sqrt(2.0) / (double)denom;
    for (int k = 0; k < 9; ++k) t[k] = 0.0f;
    t[0] = (float)scale;
    t[2] = (float)(-scale * (double)c0);
    t[4] = (float)scale;
    t[5] = (float)(-scale * (double)c1);
    t[8] = 1.0f;
    if (out) {
        // t.dot([x, y, 1]^T) in T (gemm; the zero and the one are exact): round(round(t00 x) + t02).
        const T t0 = (T)t[0], t2 = (T)t[2], t4 = (T)t[4], t5 = (T)t[5];
        for (int i = 0; i < n; ++i) {
            const T px = t0 * pts[2 * i];
            const T py = t4 * pts[2 * i + 1];
            out[2 * i] = px + t2;
            out[2 * i + 1] = py + t5;
        }
    }
}

// APAP.getConditionerFromPts, apap.py:63-89.  Its argument is the normalised point set
// of getNormalize2DPts, which numpy hands over as a column-major VIEW
// (``t.dot(p.T).T[:, :2]``, apap.py:57-58): reductions along axis 0 then run over
// contiguous memory and use pairwise summation, unlike the row-major keypoint input
// of getNormalize2DPts.
template <typename T>
void conditioner_from_pts(const T *pts, int n, float Tm[9]) {
    std::vector<T> col0((size_t)n), col1((size_t)n);
    for (int i = 0; i < n; ++i) {
        col0[i] = pts[2 * i];
        col1[i] = pts[2 * i + 1];
    }
    const T m0 = div_count(numpy_sum(col0.data(), n), n);
    const T m1 = div_count(numpy_sum(col1.data(), n), n);
    for (int i = 0; i < n; ++i) {
        const T d0 = col0[i] - m0, d1 = col1[i] - m1;
        col0[i] = d0 * d0;
        col1[i] = d1 * d1;
    }
    const T q0 = numpy_sum(col0.data(), n), q1 = numpy_sum(col1.data(), n);
    T sd0 = std::sqrt(div_count(q0, n)), sd1 = std::sqrt(div_count(q1, n));
    // std * std * n / (n - 1), all in T
    const T fn = (T)n, fn1 = (T)(n - 1);
    T v0 = sd0 * sd0;
    v0 = v0 * fn;
    v0 = v0 / fn1;
    T v1 = sd1 * sd1;
    v1 = v1 * fn;
    v1 = v1 / fn1;
    T std_x = std::sqrt(v0), std_y = std::sqrt(v1);
    if (std_x == (T)0) std_x += (T)1;
    if (std_y == (T)0) std_y += (T)1;
    const double norm_x = std::sqrt(2.0) / (double)std_x;
    const double norm_y = std::sqrt(2.0) / (double)std_y;
    for (int k = 0; k < 9; ++k) Tm[k] = 0.0f;
    Tm[0] = (float)norm_x;
    Tm[2] = (float)(-norm_x * (double)m0);
    Tm[4] = (float)norm_y;
    Tm[5] = (float)(-norm_y * (double)m1);
    Tm[8] = 1.0f;
}

// APAP.point_normalize, apap.py:92-100 (a float64 point times a float32 matrix entry is a float64 product).
template <typename T>
void point_normalize(const T *nf, int n, const float c[9], T *cf) {
    const T c0 = (T)c[0], c2 = (T)c[2], c4 = (T)c[4], c5 = (T)c[5];
    for (int i = 0; i < n; ++i) {
        const T a = nf[2 * i] * c0;
        const T b = nf[2 * i + 1] * c4;
        cf[2 * i] = a + c2;
        cf[2 * i + 1] = b + c5;
    }
}

// one point set through apap.py:133-140 in its own dtype; nf / cf (optional) come back widened to float64 (exact)
template <typename T>
void prepare_set(const T *pts, int n, float N[9], float Cm[9], double *nf, double *cf) {
    std::vector<T> a((size_t)2 * n), b((size_t)2 * n);
    normalize_2d_pts(pts, n, N, a.data());
    conditioner_from_pts(a.data(), n, Cm);
    point_normalize(a.data(), n, Cm, b.data());
    for (size_t i = 0; i < (size_t)2 * n; ++i) {
        if (nf) nf[i] = (double)a[i];
        if (cf) cf[i] = (double)b[i];
    }
}

}  // namespace

namespace apap {

// numpy.linalg.inv on a float32 3x3 computes in float64 (dgesv: LU with partial
// pivoting, then two triangular solves against the identity) and casts the result to
// float32.  Returns false on an exactly-zero pivot (numpy raises LinAlgError).
bool inv3_f64(const double a_in[9], double out[9]) {
    double a[3][3], b[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            a[i][j] = a_in[3 * i + j];
            b[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int k = 0; k < 3; ++k) {
        int p = k;
        double best = std::fabs(a[k][k]);
        for (int i = k + 1; i < 3; ++i)
            if (std::fabs(a[i][k]) > best) {
                best = std::fabs(a[i][k]);
                p = i;
            }
        if (a[p][k] == 0.0) return false;
        if (p != k)
            for (int j = 0; j < 3; ++j) {
                std::swap(a[k][j], a[p][j]);
                std::swap(b[k][j], b[p][j]);
            }
        for (int i = k + 1; i < 3; ++i) {
            const double l = a[i][k] / a[k][k];
            a[i][k] = 0.0;
            for (int j = k + 1; j < 3; ++j) a[i][j] -= l * a[k][j];
            for (int j = 0; j < 3; ++j) b[i][j] -= l * b[k][j];
        }
    }
    for (int j = 0; j < 3; ++j)
        for (int i = 2; i >= 0; --i) {
            double s = b[i][j];
            for (int k = i + 1; k < 3; ++k) s -= a[i][k] * out[3 * k + j];
            out[3 * i + j] = s / a[i][i];
        }
    return true;
}

bool inv3_f32(const float in[9], float out[9]) {
    double a[9], r[9];
    for (int k = 0; k < 9; ++k) a[k] = (double)in[k];
    if (!inv3_f64(a, r)) return false;
    for (int k = 0; k < 9; ++k) out[k] = (float)r[k];
    return true;
}

}  // namespace apap

extern "C" {

int apap_host_prepare(const float *src, const float *dst, int n, float *N1, float *N2, float *C1,
                      float *C2, float *iC2, float *iN2, float *nf1, float *nf2, float *cf1,
                      float *cf2) {
    if (!src || !dst) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_prepare: null keypoint array");
    if (n < 2) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_prepare: need at least 2 keypoints, got %d", n);
    float tN1[9], tN2[9], tC1[9], tC2[9], tiC2[9], tiN2[9];
    std::vector<float> a1((size_t)2 * n), a2((size_t)2 * n), b1((size_t)2 * n), b2((size_t)2 * n);
    normalize_2d_pts<float>(src, n, tN1, a1.data());
    normalize_2d_pts<float>(dst, n, tN2, a2.data());
    conditioner_from_pts<float>(a1.data(), n, tC1);
    conditioner_from_pts<float>(a2.data(), n, tC2);
    point_normalize<float>(a1.data(), n, tC1, b1.data());
    point_normalize<float>(a2.data(), n, tC2, b2.data());
    if (!apap::inv3_f32(tC2, tiC2)) return apap::fail(APAP_ERR_SINGULAR, "Singular matrix (conditioner C2)");
    if (!apap::inv3_f32(tN2, tiN2)) return apap::fail(APAP_ERR_SINGULAR, "Singular matrix (normaliser N2)");
    const size_t m9 = 9 * sizeof(float), pn = (size_t)2 * n * sizeof(float);
    if (N1) std::memcpy(N1, tN1, m9);
    if (N2) std::memcpy(N2, tN2, m9);
    if (C1) std::memcpy(C1, tC1, m9);
    if (C2) std::memcpy(C2, tC2, m9);
    if (iC2) std::memcpy(iC2, tiC2, m9);
    if (iN2) std::memcpy(iN2, tiN2, m9);
    if (nf1) std::memcpy(nf1, a1.data(), pn);
    if (nf2) std::memcpy(nf2, a2.data(), pn);
    if (cf1) std::memcpy(cf1, b1.data(), pn);
    if (cf2) std::memcpy(cf2, b2.data(), pn);
    return APAP_OK;
}

int apap_host_dlt_rows(const float *cf1, const float *cf2, int n, float *aa) {
    if (!cf1 || !cf2 || !aa || n < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_dlt_rows: bad argument");
    std::memset(aa, 0, (size_t)n * 18 * sizeof(float));
    for (int k = 0; k < n; ++k) {
        const float x = cf1[2 * k], y = cf1[2 * k + 1];
        const float nxp = -cf2[2 * k], nyp = -cf2[2 * k + 1];
        float *r1 = aa + (size_t)18 * k, *r2 = r1 + 9;
        r1[0] = x;
        r1[1] = y;
        r1[2] = 1.0f;
        r1[6] = nxp * x;
        r1[7] = nxp * y;
        r1[8] = nxp;
        r2[3] = x;
        r2[4] = y;
        r2[5] = 1.0f;
        r2[6] = nyp * x;
        r2[7] = nyp * y;
        r2[8] = nyp;
    }
    return APAP_OK;
}

int apap_host_build_table(const float *src, const float *cf1, const float *cf2, int n,
                          double *table) {
    if (!src || !cf1 || !cf2 || !table || n < 1)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_build_table: bad argument");
    for (int k = 0; k < n; ++k) {
        const float xf = cf1[2 * k], yf = cf1[2 * k + 1];
        const float nxp = -cf2[2 * k], nyp = -cf2[2 * k + 1];
        // the float32-rounded DLT entries of apap.py:109-118, widened
        const double x = xf, y = yf;
        const double a = (double)(nxp * xf), b = (double)(nxp * yf), c = nxp;
        const double d = (double)(nyp * xf), e = (double)(nyp * yf), f = nyp;
        double *t = table + (size_t)APAP_TABLE_STRIDE * k;
        t[0] = x * x; t[1] = x * y; t[2] = x; t[3] = y * y; t[4] = y; t[5] = 1.0;
        t[6] = x * a; t[7] = x * b; t[8] = x * c;
        t[9] = y * a; t[10] = y * b; t[11] = y * c;
        t[12] = a; t[13] = b; t[14] = c;
        t[15] = x * d; t[16] = x * e; t[17] = x * f;
        t[18] = y * d; t[19] = y * e; t[20] = y * f;
        t[21] = d; t[22] = e; t[23] = f;
        t[24] = a * a + d * d; t[25] = a * b + d * e; t[26] = a * c + d * f;
        t[27] = b * b + e * e; t[28] = b * c + e * f; t[29] = c * c + f * f;
        t[30] = (double)src[2 * k];
        t[31] = (double)src[2 * k + 1];
    }
    return APAP_OK;
}

int apap_host_prepare_pts(const void *src, int src_f64, const void *dst, int dst_f64, int n, float *N1, float *N2, float *C1,
                          float *C2, float *iC2, float *iN2, double *nf1, double *nf2, double *cf1, double *cf2) {
    if (!src || !dst) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_prepare_pts: null keypoint array");
    if (n < 2) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_prepare_pts: need at least 2 keypoints, got %d", n);
    float tN1[9], tN2[9], tC1[9], tC2[9], tiC2[9], tiN2[9];
    if (src_f64) prepare_set<double>((const double *)src, n, tN1, tC1, nf1, cf1);
    else prepare_set<float>((const float *)src, n, tN1, tC1, nf1, cf1);
    if (dst_f64) prepare_set<double>((const double *)dst, n, tN2, tC2, nf2, cf2);
    else prepare_set<float>((const float *)dst, n, tN2, tC2, nf2, cf2);
    if (!apap::inv3_f32(tC2, tiC2)) return apap::fail(APAP_ERR_SINGULAR, "Singular matrix (conditioner C2)");
    if (!apap::inv3_f32(tN2, tiN2)) return apap::fail(APAP_ERR_SINGULAR, "Singular matrix (normaliser N2)");
    const size_t m9 = 9 * sizeof(float);
    if (N1) std::memcpy(N1, tN1, m9);
    if (N2) std::memcpy(N2, tN2, m9);
    if (C1) std::memcpy(C1, tC1, m9);
    if (C2) std::memcpy(C2, tC2, m9);
    if (iC2) std::memcpy(iC2, tiC2, m9);
    if (iN2) std::memcpy(iN2, tiN2, m9);
    return APAP_OK;
}

int apap_host_dlt_rows_pts(const double *cf1, const double *cf2, int n, int product_f64, float *aa) {
    if (!cf1 || !cf2 || !aa || n < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_dlt_rows_pts: bad argument");
    std::memset(aa, 0, (size_t)n * 18 * sizeof(float));
    for (int k = 0; k < n; ++k) {
        float *r1 = aa + (size_t)18 * k, *r2 = r1 + 9;
        const double x = cf1[2 * k], y = cf1[2 * k + 1], nxp = -cf2[2 * k], nyp = -cf2[2 * k + 1];
        r1[0] = r2[3] = (float)x;
        r1[1] = r2[4] = (float)y;
        r1[2] = r2[5] = 1.0f;
        r1[8] = (float)nxp;
        r2[8] = (float)nyp;
        if (product_f64) {      // a float64 operand: the product is a float64, rounded once when it is stored (apap.py:109-118)
            r1[6] = (float)(nxp * x);
            r1[7] = (float)(nxp * y);
            r2[6] = (float)(nyp * x);
            r2[7] = (float)(nyp * y);
        } else {                // both sets float32 (the values ARE float32): a float32 product
            r1[6] = (float)nxp * (float)x;
            r1[7] = (float)nxp * (float)y;
            r2[6] = (float)nyp * (float)x;
            r2[7] = (float)nyp * (float)y;
        }
    }
    return APAP_OK;
}

int apap_host_build_table_rows(const double *src, const float *aa, int n, double *table) {
    if (!src || !aa || !table || n < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_build_table_rows: bad argument");
    for (int k = 0; k < n; ++k) {
        const float *r1 = aa + (size_t)18 * k, *r2 = r1 + 9;
        const double x = r1[0], y = r1[1], a = r1[6], b = r1[7], c = r1[8], d = r2[6], e = r2[7], f = r2[8];
        double *t = table + (size_t)APAP_TABLE_STRIDE * k;
        t[0] = x * x; t[1] = x * y; t[2] = x; t[3] = y * y; t[4] = y; t[5] = 1.0;
        t[6] = x * a; t[7] = x * b; t[8] = x * c;
        t[9] = y * a; t[10] = y * b; t[11] = y * c;
        t[12] = a; t[13] = b; t[14] = c;
        t[15] = x * d; t[16] = x * e; t[17] = x * f;
        t[18] = y * d; t[19] = y * e; t[20] = y * f;
        t[21] = d; t[22] = e; t[23] = f;
        t[24] = a * a + d * d; t[25] = a * b + d * e; t[26] = a * c + d * f;
        t[27] = b * b + e * e; t[28] = b * c + e * f; t[29] = c * c + f * f;
        t[30] = src[2 * k];
        t[31] = src[2 * k + 1];
    }
    return APAP_OK;
}

int apap_host_build_table24(const double *src, const float *aa, int n, double *table) {
    if (!src || !aa || !table || n < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_build_table24: bad argument");
    const long long marker_bits = 0x7ff8242424242424ll;     // kTable24Marker of apap_kernels.hip
    double marker;
    std::memcpy(&marker, &marker_bits, sizeof marker);
    for (int k = 0; k < n; ++k) {
        const float *r1 = aa + (size_t)18 * k, *r2 = r1 + 9;
        const double x = r1[0], y = r1[1], c = r1[8], f = r2[8];
        const double pp[6] = {x * x, x * y, x, y * y, y, 1.0};
        const double r = c * c + f * f;
        double *t = table + (size_t)APAP_TABLE_STRIDE * k;
        for (int j = 0; j < 6; ++j) {
            t[j] = pp[j];
            t[6 + j] = c * pp[j];
            t[12 + j] = f * pp[j];
            t[18 + j] = r * pp[j];
        }
        t[24] = r1[6]; t[25] = r1[7]; t[26] = r2[6]; t[27] = r2[7];
        t[28] = marker;
        const float sf[2] = {(float)src[2 * k], (float)src[2 * k + 1]};
        std::memcpy(&t[29], sf, sizeof(double));
        t[30] = src[2 * k];
        t[31] = src[2 * k + 1];
    }
    return APAP_OK;
}

int apap_host_build_denorm(const float *iC2, const float *C1, const float *iN2, const float *N1,
                           double *denorm) {
    if (!iC2 || !C1 || !iN2 || !N1 || !denorm)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_host_build_denorm: null argument");
    for (int k = 0; k < 9; ++k) {
        denorm[k] = iC2[k];
        denorm[9 + k] = C1[k];
        denorm[18 + k] = iN2[k];
        denorm[27 + k] = N1[k];
    }
    return APAP_OK;
}

}  // extern "C"
