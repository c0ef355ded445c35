/* A host in plain C driving the engine through the C ABI alone (no Python, no torch):
 *   gcc -O2 examples/c_host.c -Iinclude -Lcvx_proj_amd -lapap_hip -Wl,-rpath,$PWD/cvx_proj_amd -lm -o c_host && ./c_host
 * Builds a small synthetic pair, runs local_homography, local_warp, the output stage and the
 * two callers of the path (equalisation, RANSAC seed homography) with the NULL context (= the
 * defaults), prints checksums and - given a file name - dumps its inputs and outputs raw, so that
 * tests/test_gpu_parity.py can feed the same bytes to the Python binding and compare exactly. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "apap_hip.h"

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char **argv) {
    enum { N = 200, ROWS = 12, COLS = 15, W = 320, Hh = 240 };
    if (apap_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    unsigned seed = 7;
    float *src = malloc(sizeof(float) * 2 * N), *dst = malloc(sizeof(float) * 2 * N);
    for (int i = 0; i < N; ++i) {
        const float x = (float)(lcg(&seed) % (W * 16)) / 16.0f, y = (float)(lcg(&seed) % (Hh * 16)) / 16.0f;
        src[2 * i] = x; src[2 * i + 1] = y;
        dst[2 * i] = 1.01f * x + 0.02f * y + 4.0f + (float)(lcg(&seed) % 64) / 64.0f;
        dst[2 * i + 1] = -0.015f * x + 0.99f * y + 3.0f + (float)(lcg(&seed) % 64) / 64.0f;
    }
    const int fw = W + 8, fh = Hh + 6, ox = 0, oy = 0;
    double *vert = malloc(sizeof(double) * ROWS * COLS * 2), mesh_w[COLS + 1], mesh_h[ROWS + 1];
    for (int r = 0; r < ROWS; ++r)
        for (int c = 0; c < COLS; ++c) {
            vert[(r * COLS + c) * 2] = (c + 0.5) * fw / COLS;
            vert[(r * COLS + c) * 2 + 1] = (r + 0.5) * fh / ROWS;
        }
    for (int c = 0; c <= COLS; ++c) mesh_w[c] = (double)c * fw / COLS;
    for (int r = 0; r <= ROWS; ++r) mesh_h[r] = (double)r * fh / ROWS;
    float *H = malloc(sizeof(float) * ROWS * COLS * 9);
    int rc = apap_local_homography(NULL, src, dst, N, vert, ROWS, COLS, 0.5, 100.0, H, NULL, -1);
    if (rc) { fprintf(stderr, "local_homography: %s\n", apap_last_error()); return 1; }
    unsigned char *img = malloc((size_t)W * Hh * 3), *out = malloc((size_t)fw * fh * 3);
    for (size_t i = 0; i < (size_t)W * Hh * 3; ++i) img[i] = (unsigned char)(lcg(&seed) & 0xff);
    rc = apap_local_warp(NULL, img, Hh, W, H, ROWS, COLS, mesh_w, COLS + 1, mesh_h, ROWS + 1, fw, fh, ox, oy, out, NULL, -1);
    if (rc) { fprintf(stderr, "local_warp: %s\n", apap_last_error()); return 1; }
    double *flat = malloc(sizeof(double) * ROWS * COLS * 9);
    rc = apap_invert_normalize_flatten(NULL, H, ROWS * COLS, flat, -1);
    if (rc) { fprintf(stderr, "flatten: %s\n", apap_last_error()); return 1; }
    double hs = 0.0, fs = 0.0;
    unsigned long long ps = 0;
    for (int i = 0; i < ROWS * COLS * 9; ++i) { hs += fabs((double)H[i]); fs += fabs(flat[i]); }
    for (size_t i = 0; i < (size_t)fw * fh * 3; ++i) ps += out[i];
    printf("%s\nH_abs_sum %.9e\nflat_abs_sum %.9e\npixel_sum %llu\n", apap_version(), hs, fs, ps);
    if (argc > 1) { /* raw dump: src, dst (float32), vertices, mesh_w, mesh_h (float64), H (float32), image, canvas (uint8), flat (float64) */
        FILE *f = fopen(argv[1], "wb");
        if (!f) { perror(argv[1]); return 1; }
        fwrite(src, sizeof(float), 2 * N, f); fwrite(dst, sizeof(float), 2 * N, f);
        fwrite(vert, sizeof(double), ROWS * COLS * 2, f); fwrite(mesh_w, sizeof(double), COLS + 1, f); fwrite(mesh_h, sizeof(double), ROWS + 1, f);
        fwrite(H, sizeof(float), ROWS * COLS * 9, f); fwrite(img, 1, (size_t)W * Hh * 3, f); fwrite(out, 1, (size_t)fw * fh * 3, f);
        fwrite(flat, sizeof(double), ROWS * COLS * 9, f);
        fclose(f);
    }
    /* the callers of the path: pre-processing (utils.py:88) and seed homography
     * (baseline_stitch_test.py:42) */
    unsigned char *eq = malloc((size_t)W * Hh * 3), *mask = malloc(N);
    rc = apap_equalize_hist(NULL, img, Hh, W, 3, eq, -1);
    if (rc) { fprintf(stderr, "equalize_hist: %s\n", apap_last_error()); return 1; }
    unsigned long long es = 0;
    for (size_t i = 0; i < (size_t)W * Hh * 3; ++i) es += eq[i];
    double Hs[9];
    int inliers = 0;
    rc = apap_find_homography_ransac(NULL, src, dst, N, 5.0, APAP_RANSAC_ITERATIONS, APAP_RANSAC_SEED, Hs, mask, &inliers, -1);
    if (rc) { fprintf(stderr, "find_homography_ransac: %s\n", apap_last_error()); return 1; }
    double ss = 0.0;
    for (int i = 0; i < 9; ++i) ss += fabs(Hs[i]);
    printf("equalized_sum %llu\nransac_inliers %d\nseed_H_abs_sum %.9e\n", es, inliers, ss);
    free(eq); free(mask);
    /* error path: a mesh that does not cover the canvas is an index error, not a crash */
    mesh_h[ROWS] = fh - 10.0;
    rc = apap_local_warp(NULL, img, Hh, W, H, ROWS, COLS, mesh_w, COLS + 1, mesh_h, ROWS + 1, fw, fh, ox, oy, out, NULL, -1);
    printf("uncovered_mesh_rc %d (%s)\n", rc, rc == APAP_ERR_INDEX ? "APAP_ERR_INDEX" : "unexpected");
    free(src); free(dst); free(vert); free(H); free(img); free(out); free(flat);
    return rc == APAP_ERR_INDEX ? 0 : 1;
}
