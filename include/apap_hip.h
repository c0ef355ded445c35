/*
 * apap_hip.h - C ABI of libapap_hip.so, the MI355X (gfx950) engine for the APAP
 * moving-DLT path of Enigmatisms/cvx_proj.
 *
 * The reference has no FFI layer: its boundary for this path is the Python surface
 * of class APAP in pyviz/apap.py and the helpers of pyviz/apap_utils.py, consumed only
 * by apap.py's __main__ (apap.py:238-265).  Every entry point below names the
 * reference interface it stands in for.  A maintainer binds them with ctypes; the
 * binding is shown in INTEGRATION.md and shipped as cvx_proj_amd/_native.py.
 *
 * Conventions
 *  - Plain C types only.  All arrays are row-major and caller-owned; the library
 *    never keeps a pointer after a call returns.
 *  - Every function that returns int returns APAP_OK (0) or one of the APAP_ERR_*
 *    codes; apap_last_error() then gives a thread-local human-readable message.
 *    The reference signals the same conditions with Python exceptions
 *    (numpy.linalg.LinAlgError from apap.py:165-166,203; IndexError from
 *    apap.py:207,209-210; ValueError from the shape unpacking at apap.py:129-130).
 *  - "Host" entry points take host pointers and are synchronous: inputs are copied
 *    to the selected GPU, the kernels run, outputs are copied back before return.
 *  - "_device" entry points take DEVICE pointers (hipMalloc'ed by the caller, or a
 *    torch tensor's data_ptr()) and a hipStream_t passed as void* (NULL = the
 *    default stream).  They only enqueue work; the caller synchronises.  They are
 *    what bench.py and the multi-GPU driver use to keep data resident in HBM.
 *  - Every compute entry point takes an apap_ctx* first (NULL = defaults): options, profiling
 *    and the device-buffer pool live there, not in the process (section "context" below).
 *  - There is no CPU fallback.  Without a usable gfx950 device every compute entry
 *    point fails with APAP_ERR_NO_DEVICE.
 */
#ifndef APAP_HIP_H
#define APAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APAP_OK 0
#define APAP_ERR_INVALID_ARG 1 /* bad pointer / size (reference: ValueError on unpack)       */
#define APAP_ERR_NO_DEVICE 2   /* no HIP device, or device index out of range                */
#define APAP_ERR_HIP 3         /* a HIP runtime call failed; message has hipGetErrorString   */
#define APAP_ERR_SINGULAR 4    /* exact-zero pivot in a 3x3 inverse (reference: LinAlgError) */
#define APAP_ERR_INDEX 5       /* mesh edges do not cover the canvas (reference: IndexError) */
#define APAP_ERR_WORKSPACE 6   /* caller-provided workspace too small                        */

/* Doubles per keypoint in the device point table (see apap_host_build_table). */
#define APAP_TABLE_STRIDE 32
/* Doubles in the de-normalisation block: inv(C2), C1, inv(N2), N1 (3x3 row-major each). */
#define APAP_DENORM_DOUBLES 36
/* Doubles per cell in the padded inverse-homography buffer the warp kernel reads. */
#define APAP_HINV_STRIDE 10

/* Solver variants (APAP_OPT_SOLVER_VARIANT).  All produce the same float32 grids on the golden
 * vectors; they differ in the summation order of the 30 moment sums.  With APAP_OPT_MOMENTS = 24: AUTO / MFMA = one
 * 16x16x4 + two 4x4x4_4b instructions per step, MFMA4 / MFMA4X2 = six 4x4x4_4b; VALU is refused. */
#define APAP_VARIANT_AUTO 0
#define APAP_VARIANT_VALU 1 /* one lane per cell, fp64 FMA accumulation            */
#define APAP_VARIANT_MFMA 2 /* v_mfma_f64_16x16x4_f64 accumulation, table via LDS  */
#define APAP_VARIANT_MFMA4 3   /* v_mfma_f64_4x4x4_4b_f64, 16 cells per wave          */
#define APAP_VARIANT_MFMA4X2 4 /* v_mfma_f64_4x4x4_4b_f64, 32 cells per wave share B  */

/* Eigen-solvers of K2 (APAP_OPT_EIGEN_SOLVER). */
#define APAP_EIGEN_AUTO 0              /* = inverse iteration with Jacobi fallback          */
#define APAP_EIGEN_JACOBI 1            /* cyclic Jacobi sweeps only                         */
#define APAP_EIGEN_INVERSE_ITERATION 2 /* LDL^T inverse iteration; Jacobi for cells without a
                                          spectral gap and rank-deficient systems            */

/* Kernel slots of apap_ctx_profile_read. */
#define APAP_PROF_ASSEMBLE 0 /* K1: weighted moment sums A^T W^2 A      */
#define APAP_PROF_EIGEN 1    /* K2: eigen-solve + de-normalise          */
#define APAP_PROF_INVERT 2   /* per-cell 3x3 inverse (+ lookup table)   */
#define APAP_PROF_LUT 3      /* canvas row/column -> cell lookup table  */
#define APAP_PROF_WARP 4     /* K3: backward warp gather                */
#define APAP_PROF_EQ_HIST 5  /* E1: per-channel histogram               */
#define APAP_PROF_EQ_APPLY 6 /* E2: table rebuild + mapping             */
#define APAP_PROF_RANSAC 7   /* R1-R3: hypotheses, scoring, selection   */
#define APAP_PROF_SLOTS 8

/* ABI generation.  Generation 2 put `apap_ctx *` first in every compute entry point while keeping the
 * symbol names of generation 1: a caller built against the old header would still link and then pass a
 * float* where the context goes.  Such a caller must refuse to run: check apap_abi_version() ==
 * APAP_ABI_VERSION once after loading (cvx_proj_amd/_native.py does); a generation-1 library does not
 * export the symbol at all.  Bump on any change of an existing signature. */
#define APAP_ABI_VERSION 6
#define APAP_ABI_VERSION_STRING "0.6"

/* ---------------------------------------------------------------- diagnostics --- */
const char *apap_last_error(void);
const char *apap_version(void);
int apap_abi_version(void);
/* Number of visible HIP devices; 0 when there is none (never an error). */
int apap_device_count(void);

/* -------------------------------------------------------------------- context --- */
/* The library keeps NO process-wide mutable state (the reference's class is re-entrant: all of
 * its state is on `self`, apap.py:22-32).  Every compute entry point takes an `apap_ctx *` as its
 * first argument; NULL means "the built-in defaults, no profiling" and is always valid.  A context
 * carries the options below, the HIP events of its profiling, and the pool of device buffers its
 * host-buffer calls reuse.  One context must not be used from two threads at once; different
 * contexts (and NULL) are independent - NULL-context host-buffer calls share one pool and are
 * serialised on it. */
typedef struct apap_ctx apap_ctx;
#define APAP_OPT_SOLVER_VARIANT 0 /* APAP_VARIANT_*; default AUTO (= MFMA)                              */
#define APAP_OPT_EIGEN_SOLVER 1   /* APAP_EIGEN_*; default AUTO                                         */
#define APAP_OPT_CAREFUL 2        /* 1 (default): cells whose eigen-gap is below 1e-3 of the trace, and every
                                     cell when n < 5, are re-solved from the weighted 2n x 9 rows (Givens QR +
                                     one-sided Jacobi) as apap.py:159-161 does; 0: normal equations only    */
#define APAP_OPT_PROFILE 3        /* 1: bracket every kernel with HIP events (apap_ctx_profile_read)     */
#define APAP_OPT_WANT_WAVES 4     /* tuning: waves K1 aims at before it stops splitting the keypoints     */
#define APAP_OPT_WARP_ROWS 5      /* canvas rows per wave of K3: 1 (default) = chosen from the size of the launch (4 for one
                                     4K canvas and for the fused stitch, 6 for an 8K canvas, 8 for batches of many
                                     canvases); 2, 4, 5, 6, 8 = that many; 0 = the flat-order kernel.  Same canvas bytes in
                                     every form                                                                             */
#define APAP_OPT_WEIGHT_CHUNK_KB 6 /* device staging of the optional weight tensor, KiB (default 1 GiB)   */
#define APAP_OPT_FUSED_MAX_CELLS 7 /* tuning: meshes of up to this many cells (x batch) take the fused K1 + K2
                                      launch when the variant is AUTO (default 4096; 0 = never)             */
#define APAP_OPT_WARP_FAST 8       /* 1 (default): K3 decides a pixel from a float32 estimate of its source coordinate
                                      and takes the exact float64 sequence only where the estimate is within its error
                                      bound of an integer (same canvas, byte for byte); 0: float64 for every pixel    */
#define APAP_OPT_OVERLAP_PCIE 9    /* 0 (default): apap_local_warp / apap_local_stitch make one copy up, one kernel, one copy
                                      down.  1: they pin the caller's buffers for the call (hipHostRegister; buffers the caller already page-locked -
                                      hipHostMalloc, its own registration - are taken as they are) and overlap the
                                      image upload, the warp (in row bands) and the canvas download on three streams.  For
                                      callers that REUSE their image / canvas buffers: the first uses of a buffer pay 8-26 ms of
                                      pinning and mapping (4K pair), every later call saves ~15 %                              */
#define APAP_OPT_PLAN_CELLS 10      /* 0 (default): the solve picks its kernel (fused small-mesh launch or K1 + K2) and its
                                      keypoint splits from THIS call's cells x batch.  c > 0: as for ONE pair of c cells,
                                      whatever the call holds - a shard of a mesh (cvx_proj_amd/dist.py) then sums every
                                      cell's keypoints in the order the whole mesh would on one GPU: the same bits for any
                                      number of ranks, at the price of fewer, larger blocks per GPU                      */
#define APAP_OPT_MOMENTS 11         /* 30 (default): K1 sums the 30 distinct entries of A^T W^2 A with the reference's
                                      float32-ROUNDED DLT products kept verbatim (apap.py:103-119): grids bit-identical to the
                                      reference's.  24 (opt-in): the 24 sums of the EXACT products (SURVEY.md section 8a:
                                      [[S0,0,Sx],[0,S0,Sy],[Sx,Sy,Sr]]) - a quarter less matrix-pipe work in K1, a fifth less
                                      slab traffic; NOT bit-identical: one float32 ulp in a few per cent of a grid's entries
                                      (reprojection-RMSE delta < 1e-4 px on BASELINE's configurations, tests/test_gpu_moments24.py).
                                      The device entry points then expect the table of apap_host_build_table24 (a table of the
                                      other layout gives NaN grids); the host-buffer entry points build the right one themselves.
                                      No fused small-mesh launch, no VALU variant in this mode                             */
#define APAP_OPT_WEIGHTS_F32 12     /* 0 (default).  1 (opt-in, honoured with APAP_OPT_MOMENTS = 24 only): K1 evaluates w^2 in
                                      float32 (v_sqrt_f32, v_exp_f32: ~2e-7 relative) instead of float64; the sums stay
                                      float64.  Same class of result as MOMENTS = 24 alone (one float32 ulp here and there)  */
#define APAP_OPT_COUNT 13
apap_ctx *apap_ctx_create(void);
void apap_ctx_destroy(apap_ctx *ctx); /* frees the pooled device buffers and pending events; NULL is a no-op */
int apap_ctx_set_option(apap_ctx *ctx, int option, int value);
int apap_ctx_get_option(const apap_ctx *ctx, int option, int *value); /* ctx may be NULL: the defaults */
/* Waits for the events recorded since the previous read and returns, per slot, the summed
 * milliseconds and the number of launches. */
int apap_ctx_profile_read(apap_ctx *ctx, float *ms, int *launches);

/* ------------------------------------------------------------ host-only helpers --- */
/* No GPU needed.  They restate, in C and in float32 exactly as numpy evaluates the
 * reference, the once-per-pair set-up that APAP.local_homography performs before its
 * cell loop. */

/* APAP.getNormalize2DPts x2, getConditionerFromPts x2, point_normalize x2
 * (apap.py:35-100,133-140) and the two float32 inverses of apap.py:165-166.
 * src, dst: n x 2 float32.  Outputs (any may be NULL): N1,N2,C1,C2,iC2,iN2 are 3x3
 * float32 row-major; nf1,nf2,cf1,cf2 are n x 2 float32. */
int apap_host_prepare(const float *src, const float *dst, int n, float *N1, float *N2, float *C1,
                      float *C2, float *iC2, float *iN2, float *nf1, float *nf2, float *cf1,
                      float *cf2);

/* APAP.matrix_generate (apap.py:103-119): the 2n x 9 float32 DLT matrix. */
int apap_host_dlt_rows(const float *cf1, const float *cf2, int n, float *aa);

/* The same set-up in the dtype of the keypoints, as the reference's functions run when they are handed float64 points
 * (nothing in apap.py:35-100 casts its argument: float64 keypoints stay float64 until the 3 x 3 matrices and the DLT rows
 * are rounded into float32 arrays, apap.py:53-55,85-87,104-118).  src / dst: n x 2 float32 or float64, each with its own flag
 * (numpy promotes per set).  nf*, cf* come back as float64 (a float32 set's values widened, exactly).  product_f64 of the
 * DLT rows = "one of the two sets was float64": the products -cf2 * cf1 are then float64 products rounded once on the store;
 * with both sets float32 the three *_pts functions give the bits of the float32 functions above. */
int apap_host_prepare_pts(const void *src, int src_f64, const void *dst, int dst_f64, int n, float *N1, float *N2, float *C1,
                          float *C2, float *iC2, float *iN2, double *nf1, double *nf2, double *cf1, double *cf2);
int apap_host_dlt_rows_pts(const double *cf1, const double *cf2, int n, int product_f64, float *aa);
/* the device point table from the DLT rows themselves (`aa`: 2n x 9 float32) and the float64 (or widened) source keypoints */
int apap_host_build_table_rows(const double *src, const float *aa, int n, double *table);

/* Device point table: per keypoint APAP_TABLE_STRIDE doubles -
 *   [0..29]  the 30 distinct entries of r1 r1^T + r2 r2^T, r1/r2 being the point's two
 *            float32 DLT rows (products of float32 values are exact in float64),
 *   [30,31]  the source keypoint (x, y) widened to float64.
 * and the 36-double de-normalisation block. */
int apap_host_build_table(const float *src, const float *cf1, const float *cf2, int n,
                          double *table);
/* The table of APAP_OPT_MOMENTS = 24, from the same DLT rows `aa` (2n x 9 float32) and source keypoints: with p = (x, y, 1)
 * and c = -x', f = -y' as `aa` holds them (aa[k][0..1], aa[2k][8], aa[2k+1][8]), and pp = (xx, xy, x, yy, y, 1) -
 *   [0..5] pp   [6..11] c pp   [12..17] f pp   [18..23] (c^2 + f^2) pp     (float64 products of the float32 values)
 *   [24..27] the rows' own float32 products aa[2k][6], aa[2k][7], aa[2k+1][6], aa[2k+1][7] (the careful path re-solves from
 *            the reference's rows), [28] the layout marker (a quiet NaN), [29] the source keypoint as two float32
 *            (APAP_OPT_WEIGHTS_F32), [30,31] the source keypoint (x, y) as float64. */
int apap_host_build_table24(const double *src, const float *aa, int n, double *table);
int apap_host_build_denorm(const float *iC2, const float *C1, const float *iN2, const float *N1,
                           double *denorm);

/* ------------------------------------------------------ host-buffer entry points --- */

/* APAP.local_homography (apap.py:121-169).
 *   src, dst   n x 2 float32 keypoints (src -> dst)
 *   vertices   mesh_rows x mesh_cols x 2 float64 cell sample points
 *   gamma,sigma  the two scalars of APAP.__init__ (apap.py:22-32)
 *   H_out      mesh_rows x mesh_cols x 9 float32
 *   W_out      NULL, or mesh_rows x mesh_cols x n float64 (the reference's second
 *              return value; 8*n bytes per cell of extra HBM + PCIe traffic)
 *   device     HIP device index, or -1 for the current device */
int apap_local_homography(apap_ctx *ctx, const float *src, const float *dst, int n, const double *vertices,
                          int mesh_rows, int mesh_cols, double gamma, double sigma, float *H_out,
                          double *W_out, int device);
/* The same for keypoints of either dtype (src_f64 / dst_f64: the array holds float64): the set-up then runs as the
 * reference's does on such arrays (apap_host_prepare_pts), and the weights use the float64 source keypoints as they are
 * (apap.py:150: `vertices - src_point` in float64).  apap_local_homography is this call with both flags 0. */
int apap_local_homography_pts(apap_ctx *ctx, const void *src, int src_f64, const void *dst, int dst_f64, int n,
                              const double *vertices, int mesh_rows, int mesh_cols, double gamma, double sigma, float *H_out,
                              double *W_out, int device);

/* The second return value of APAP.local_homography alone (apap.py:144,150-153,169): W_out[c][k] =
 * max(exp(-|vertices[c] - src[k]| / sigma^2), gamma), cells x n float64, for ANY list of `cells` sample points
 * (the whole mesh, or the cells a caller indexes: cvx_proj_amd.apap.LazyWeights).  The reference's own caller
 * never reads this tensor (apap.py:242); computing it on demand keeps its 8 n bytes per cell off the default path. */
int apap_local_weights(apap_ctx *ctx, const float *src, int n, const double *vertices, int cells, double gamma,
                       double sigma, double *W_out, int device);
int apap_local_weights_pts(apap_ctx *ctx, const void *src, int src_f64, int n, const double *vertices, int cells, double gamma,
                           double sigma, double *W_out, int device);

/* APAP.local_warp (apap.py:186-217).
 *   img        img_h x img_w x 3 uint8
 *   Hfwd       mesh_rows x mesh_cols x 9 float32; inverted per cell inside, like
 *              apap.py:201-203 does in place
 *   mesh_w/h   cell edges along x / y (the two rows of get_mesh), n_w / n_h entries
 *   out        final_h x final_w x 3 uint8
 *   Hinv_out   NULL, or mesh_rows x mesh_cols x 9 float32 receiving the inverses (what
 *              the reference leaves in its mutated argument) */
int apap_local_warp(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const float *Hfwd, int mesh_rows,
                    int mesh_cols, const double *mesh_w, int n_w, const double *mesh_h, int n_h,
                    int final_w, int final_h, int off_x, int off_y, uint8_t *out, float *Hinv_out,
                    int device);

/* APAP.local_warp for a float64 grid.  The reference inverts the cells in the grid's own dtype
 * (apap.py:201-203: numpy.linalg.inv keeps float64) and multiplies in float64, so a float64 grid is
 * NOT rounded to float32 on the way: Hfwd and Hinv_out are float64 here, everything else is as
 * apap_local_warp. */
int apap_local_warp_f64(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const double *Hfwd, int mesh_rows,
                        int mesh_cols, const double *mesh_w, int n_w, const double *mesh_h, int n_h,
                        int final_w, int final_h, int off_x, int off_y, uint8_t *out, double *Hinv_out,
                        int device);

/* The stitch the reference's __main__ keeps commented out (apap.py:258-262), fused into
 * one pass: warp `img` like apap_local_warp, paste `center` (center_h x center_w x 3) at
 * (off_x, off_y) on an empty canvas, uniform_blend (apap_utils.py:75-88) the two.  The
 * centre image must fit the canvas at the offsets (the reference's slice assignment raises
 * otherwise): APAP_ERR_INVALID_ARG. */
int apap_local_stitch(apap_ctx *ctx, const uint8_t *img, int img_h, int img_w, const uint8_t *center, int center_h,
                      int center_w, const float *Hfwd, int mesh_rows, int mesh_cols,
                      const double *mesh_w, int n_w, const double *mesh_h, int n_h, int final_w,
                      int final_h, int off_x, int off_y, uint8_t *out, float *Hinv_out, int device);

/* Same inputs as apap_local_warp; writes the float64 target coordinates (tx, ty) of
 * every canvas pixel (apap.py:211-213) instead of gathering.  coords: final_h x
 * final_w x 2 float64.  For parity tests of the coordinate arithmetic. */
int apap_warp_coords(apap_ctx *ctx, const float *Hfwd, int mesh_rows, int mesh_cols, const double *mesh_w, int n_w,
                     const double *mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                     double *coords, int device);

/* Output stage of apap.py:250-264: per cell H <- inv(H), H /= H[2,2] (float32), then
 * the transposed 3x3 flattened to 9 float64.  H: cells x 9 float32; out: cells x 9. */
int apap_invert_normalize_flatten(apap_ctx *ctx, const float *H, int cells, double *out, int device);

/* uniform_blend (apap_utils.py:75-88) over two h x w x 3 uint8 canvases. */
int apap_uniform_blend(apap_ctx *ctx, const uint8_t *img1, const uint8_t *img2, int h, int w, uint8_t *out,
                       int device);

/* -------------------------------------------------- resident (device) entry points --- */

/* Bytes of scratch apap_solve_device needs for this problem size. */
size_t apap_solve_workspace_bytes(apap_ctx *ctx, int n, int cells);

/* Per-cell weighted DLT + eigen-solve + de-normalisation on resident data.
 *   d_table    n x APAP_TABLE_STRIDE doubles   (apap_host_build_table)
 *   d_vertices cells x 2 doubles
 *   d_denorm   APAP_DENORM_DOUBLES doubles     (apap_host_build_denorm)
 *   d_H        cells x 9 floats (output)
 *   d_work     scratch of apap_solve_workspace_bytes(ctx, n, cells) bytes */
int apap_solve_device(apap_ctx *ctx, const double *d_table, int n, const double *d_vertices, int cells,
                      double gamma, double sigma, const double *d_denorm, float *d_H, void *d_work,
                      size_t work_bytes, void *stream);

/* The same for a BATCH of image pairs with equal n and cell count in one launch
 * (blockIdx.z = pair): d_tables batch x n x 32, d_denorms batch x 36, d_H batch x cells x 9;
 * d_vertices + k * vertices_stride is pair k's mesh (stride in doubles; 0 = one mesh shared
 * by all pairs).  Fills the chip with fewer keypoint splits than `batch` separate calls. */
size_t apap_solve_batch_workspace_bytes(apap_ctx *ctx, int n, int cells, int batch);
int apap_solve_batch_device(apap_ctx *ctx, const double *d_tables, int n, const double *d_vertices, long long vertices_stride,
                            int cells, double gamma, double sigma, const double *d_denorms, float *d_H,
                            int batch, void *d_work, size_t work_bytes, void *stream);

/* The solve of a caller that WARPS NEXT (apap.py:240-243 followed by :186-217 - the reference's own sequence): the same
 * kernels as apap_solve_batch_device with the warp's per-cell set-up riding in the eigen-solve kernel's tail, where it costs
 * a fraction of a launch of its own: every cell leaves, beside its float32 H, its inverse, its float32-estimate record and its
 * exact-path floats in the warp workspace - what APAP_WARP_CELLS would compute from the stored grid, bit for bit (one
 * device function serves both).  The warp that follows runs apap_warp_batch_device(... phases = APAP_WARP_GATHER ...) on that
 * workspace (APAP_WARP_GEOMETRY once per mesh / canvas geometry, before or after).  cells = mesh_rows * mesh_cols; the mesh
 * edges, canvas size and offsets are the warp's.  d_status: the warp's status word (bit 0: a singular cell).  Meshes beyond
 * 4096 edges per axis: APAP_ERR_INVALID_ARG (solve and warp them with the separate entry points). */
int apap_solve_warp_batch_device(apap_ctx *ctx, const double *d_tables, int n, const double *d_vertices, long long vertices_stride,
                                 double gamma, double sigma, const double *d_denorms, float *d_H, int batch, void *d_work,
                                 size_t work_bytes, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w,
                                 const double *d_mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                                 void *d_warp_work, size_t warp_work_bytes, int *d_status, void *stream);

/* The weights tensor alone: d_W cells x n doubles. */
int apap_weights_device(apap_ctx *ctx, const double *d_table, int n, const double *d_vertices, int cells,
                        double gamma, double sigma, double *d_W, void *stream);

size_t apap_warp_workspace_bytes(int mesh_rows, int mesh_cols, int final_w, int final_h);

/* Backward warp on resident data.  d_status: one int the kernels OR error bits into
 * (bit 0: singular cell, bit 1: index error); zero it before the call.  d_Hinv_out
 * may be NULL. */
int apap_warp_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const float *d_Hfwd, int mesh_rows,
                     int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h,
                     int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *d_out,
                     float *d_Hinv_out, void *d_work, size_t work_bytes, int *d_status,
                     void *stream);

/* apap_warp_device for a float64 grid (see apap_local_warp_f64). */
int apap_warp_f64_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const double *d_Hfwd, int mesh_rows,
                         int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h, int final_w,
                         int final_h, int off_x, int off_y, uint8_t *d_out, double *d_Hinv_out, void *d_work,
                         size_t work_bytes, int *d_status, void *stream);

/* apap_warp_device restricted to canvas rows [row_begin, row_begin + row_count): what one
 * rank computes when the warp of ONE pair is sharded over GPUs (cvx_proj_amd/dist.py).
 * d_out_band receives row_count x final_w x 3 bytes. */
int apap_warp_rows_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const float *d_Hfwd, int mesh_rows,
                          int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h,
                          int final_w, int final_h, int off_x, int off_y, int row_begin, int row_count,
                          uint8_t *d_out_band, void *d_work, size_t work_bytes, int *d_status, void *stream);

/* The warp of a BATCH of independent pairs in one set of launches (BASELINE.json config 5: 64 pairs of one size; the
 * reference runs apap.py:186-217 once per pair), and the general form of every warp entry point above.
 * All pairs share the mesh edges, the canvas size and the offsets (pairs of one configuration do); pair k has its own
 *   image   d_imgs + k * img_stride bytes (img_h x img_w x 3; stride 0 = one image for all),
 *   centre  d_centers + k * center_stride bytes, or d_centers = NULL for the plain warp (non-NULL: the fused stitch),
 *   grid    d_Hfwd + k * mesh_rows * mesh_cols * 9 floats (what apap_solve_batch_device writes),
 *   canvas  d_outs + k * out_stride bytes, receiving rows [row_begin, row_begin + row_count) of its canvas,
 *   inverse d_Hinv_out + k * mesh_rows * mesh_cols * 9 floats, when d_Hinv_out is not NULL.
 * grid.z of the kernels is the pair: one set-up launch covers every pair's cells, one gather launch every canvas -
 * 64 pairs fill the chip where one pair's ~9000 waves are 1.45 generations with idle set-up, ramp and tail.
 * `phases`: which steps run on the workspace, any combination of
 *   APAP_WARP_GEOMETRY  canvas row / column -> cell tables: depend on the edges, the canvas size and the offsets only;
 *   APAP_WARP_CELLS     per-cell inverses and the float32 estimate's records: depend on the H grids (and the edges);
 *   APAP_WARP_GATHER    K3, reading what the other two left in the workspace.
 * A caller that warps many grids over one geometry runs GEOMETRY once and CELLS | GATHER per grid; APAP_WARP_ALL is the
 * one-call form.  A GATHER on a workspace whose tables were never built, or were built for another mesh shape / canvas
 * size, touches nothing and sets bit 2 (value 4) of *d_status (the tables carry a stamp of the sizes they were built for;
 * bits 0 and 1 are the singular cell and the uncovered canvas of the reference's LinAlgError / IndexError).  d_work: apap_warp_batch_workspace_bytes(...) bytes; the layout is private but stable between calls
 * with equal (mesh_rows, mesh_cols, final_w, final_h, batch). */
#define APAP_WARP_GEOMETRY 1
#define APAP_WARP_CELLS 2
#define APAP_WARP_GATHER 4
#define APAP_WARP_ALL 7
size_t apap_warp_batch_workspace_bytes(int mesh_rows, int mesh_cols, int final_w, int final_h, int batch);
int apap_warp_batch_device(apap_ctx *ctx, const uint8_t *d_imgs, long long img_stride, int img_h, int img_w,
                           const uint8_t *d_centers, long long center_stride, int center_h, int center_w,
                           const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w,
                           const double *d_mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                           int row_begin, int row_count, uint8_t *d_outs, long long out_stride, float *d_Hinv_out,
                           int batch, int phases, void *d_work, size_t work_bytes, int *d_status, void *stream);

/* Resident-data twin of apap_local_stitch. */
int apap_stitch_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const uint8_t *d_center, int center_h,
                       int center_w, const float *d_Hfwd, int mesh_rows, int mesh_cols,
                       const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h, int final_w,
                       int final_h, int off_x, int off_y, uint8_t *d_out, float *d_Hinv_out, void *d_work,
                       size_t work_bytes, int *d_status, void *stream);

/* Coordinates-only twin of apap_warp_device (d_coords: final_h x final_w x 2 doubles). */
int apap_warp_coords_device(apap_ctx *ctx, const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w,
                            int n_w, const double *d_mesh_h, int n_h, int final_w, int final_h,
                            int off_x, int off_y, double *d_coords, void *d_work, size_t work_bytes,
                            int *d_status, void *stream);

int apap_flatten_device(apap_ctx *ctx, const float *d_H, int cells, double *d_out, int *d_status, void *stream);

int apap_blend_device(apap_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int h, int w, uint8_t *d_out, void *stream);

/* ------------------------------------------- callers of the path --- */
/* Pre-processing of apap.py:236-237 = utils.py:85-91 visualize_equalized_hist:
 *   np.stack([cv.equalizeHist(img[..., i]) for i in range(3)], axis=-1)
 * on an interleaved uint8 image of 1..4 channels (h x w x channels); out may alias nothing.
 * cv::equalizeHist (opencv-python 4.6.0.66, absent from this image) is restated from its
 * published algorithm: see oracle/frontend_oracle.py. */
int apap_equalize_hist(apap_ctx *ctx, const uint8_t *img, int h, int w, int channels, uint8_t *out, int device);
size_t apap_equalize_workspace_bytes(int channels);
/* Resident-data form: three kernels on `stream` (histogram; table; mapping).  WORKSPACE
 * CONTRACT: d_work (16-byte aligned, apap_equalize_workspace_bytes(channels) bytes) must be all
 * zero on entry - zero it once after allocating it - and is all zero again, apart from the table
 * at its end, when the call's kernels have run; so back-to-back calls need no memset. */
int apap_equalize_hist_device(apap_ctx *ctx, const uint8_t *d_img, int h, int w, int channels, uint8_t *d_out, void *d_work,
                              size_t work_bytes, void *stream);

/* Seed homography, the contract of baseline_stitch_test.py:42
 *     H, mask = cv.findHomography(src_pts, dst_pts, cv.RANSAC, thresh)
 * src, dst: n x 2 float32.  `iterations` 4-point hypotheses drawn by a counter-based sampler
 * (`seed`), forward reprojection error against thresh, the first hypothesis with the most
 * inliers wins; H_out (9 doubles, row-major, H[8] = 1) is the normalised DLT of the hot path
 * (apap.py:35-119,160-168, all weights 1) re-fitted to its inliers; mask_out: n bytes 0/1.
 * *inliers_out < 4 means no model (cv returns None): H_out is then left untouched.  This is this
 * repository's estimator, not OpenCV's (sampler, adaptive stopping and LM polish differ):
 * oracle/frontend_oracle.py is its specification. */
#define APAP_RANSAC_ITERATIONS 2048
#define APAP_RANSAC_SEED 0x5EEDC0DE5EEDC0DEull
int apap_find_homography_ransac(apap_ctx *ctx, const float *src, const float *dst, int n, double thresh, int iterations,
                                unsigned long long seed, double *H_out, uint8_t *mask_out, int *inliers_out,
                                int device);
size_t apap_ransac_workspace_bytes(int n, int iterations);
/* Device half (no re-fit): d_H_best 9 doubles = the winning 4-point model, d_mask n bytes,
 * d_result 2 ints = {winning hypothesis, its inlier count}.  Points 8-byte aligned. */
int apap_ransac_device(apap_ctx *ctx, const float *d_src, const float *d_dst, int n, double thresh, int iterations,
                       unsigned long long seed, double *d_H_best, uint8_t *d_mask, int *d_result, void *d_work,
                       size_t work_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* APAP_HIP_H */
