/* gsplat_amd — C ABI of the MI355X (gfx950) Gaussian-rasterization hot path.
 *
 * This is the drop-in boundary: one entry point per kernel-level stage of
 * gsplat.rasterization() / rasterization_2dgs(). Every function
 *   - takes raw DEVICE pointers, extents and a hipStream_t (as void*) — no torch types;
 *   - never allocates, frees or synchronises: outputs and workspaces are caller-owned
 *     (query sizes with the *_workspace_bytes functions);
 *   - is re-entrant (no global mutable state; safe from autograd worker threads);
 *   - returns 0 on success, a negative GSX_ERR_* code otherwise; gsx_last_error() gives the
 *     message (thread local). The torch shim (gsplat_amd/_ops.py) turns codes into
 *     RuntimeError / ValueError exactly where the reference raises them.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference checkout, nerfstudio-project/gsplat 1.6.0). The torch-side binding a gsplat
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Layout conventions (identical to the reference ops): all float tensors fp32, contiguous,
 * row-major; "rows" R means [I*N] (dense, I = B*C images) or [nnz] (packed); bool tensors are
 * 1 byte per element (torch.bool).
 */
#ifndef GSPLAT_AMD_H_
#define GSPLAT_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_ABI_VERSION 1

#define GSX_OK 0
#define GSX_ERR_ARG (-1)       /* bad argument: maps to ValueError/RuntimeError in the shim */
#define GSX_ERR_LAUNCH (-2)    /* HIP launch/runtime failure */
#define GSX_ERR_WORKSPACE (-3) /* workspace too small */
#define GSX_ERR_OVERFLOW (-4)  /* key width / index overflow (Intersect.cpp:219-228) */

/* camera models — values of _C.CameraModelType (gsplat/cuda/include/Common.h:75-82) */
#define GSX_CAMERA_PINHOLE 0
#define GSX_CAMERA_ORTHO 1
#define GSX_CAMERA_FISHEYE 2

const char *gsx_last_error(void);
int gsx_version(void);
const char *gsx_arch(void); /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * quat_scale_to_covar_preci{,_bwd}
 * replaces torch ops gsplat::quat_scale_to_covar_preci{,_bwd} (gsplat/cuda/ext.cpp:984-991;
 * host gsplat/cuda/csrc/QuatScaleToCovar.cpp; device math include/Utils.cuh:228-347).
 * quats [n,4] (wxyz, normalised inside), scales [n,3]. Outputs [n,3,3] or, if triu, [n,6] in
 * order (00,01,02,11,12,22). Any of covars/precis may be NULL.
 * bwd: v_covars / v_precis same layout as the forward output (either may be NULL).
 * ------------------------------------------------------------------------------------------- */
int gsx_quat_scale_to_covar_fwd(const float *quats, const float *scales, int64_t n, int triu,
                                float *covars, float *precis, void *stream);
int gsx_quat_scale_to_covar_bwd(const float *quats, const float *scales, int64_t n, int triu,
                                const float *v_covars, const float *v_precis,
                                float *v_quats, float *v_scales, void *stream);
/* The same op in double precision (the reference instantiates it for float and double: QuatScaleToCovarCUDA.cu:145,
 * AT_DISPATCH_FLOATING_TYPES; its tests pass float64): identical layouts, every operation in IEEE double. */
int gsx_quat_scale_to_covar_fwd_f64(const double *quats, const double *scales, int64_t n, int triu,
                                    double *covars, double *precis, void *stream);
int gsx_quat_scale_to_covar_bwd_f64(const double *quats, const double *scales, int64_t n, int triu,
                                    const double *v_covars, const double *v_precis,
                                    double *v_quats, double *v_scales, void *stream);

/* ---------------------------------------------------------------------------------------------
 * fully_fused_projection (dense): gsplat::projection_ewa_3dgs_fused{,_bwd}
 * (ext.cpp:1052-1063; host Projection.cpp:366-440, 579-694; kernels
 * ProjectionEWA3DGSFused.cu:38-219, 378-638).
 * means [B,N,3]; covars [B,N,6] XOR (quats [B,N,4] + scales [B,N,3]); opacities [B,N] or NULL;
 * viewmats [B,C,4,4]; Ks [B,C,3,3]. Outputs: radii int32 [B,C,N,2]; means2d [B,C,N,2];
 * depths [B,C,N]; conics [B,C,N,3]; compensations [B,C,N] or NULL. Culled rows get radii=(0,0)
 * and ZEROS in the other outputs (the reference leaves them uninitialised).
 * bwd outputs are fully written (no pre-zeroing needed) except v_viewmats which must be
 * zero-initialised when non-NULL: v_means [B,N,3], v_covars [B,N,6] | (v_quats [B,N,4],
 * v_scales [B,N,3]), v_viewmats [B,C,4,4] or NULL.
 * ------------------------------------------------------------------------------------------- */
int gsx_project_ewa_fwd(const float *means, const float *covars, const float *quats, const float *scales,
                        const float *opacities, const float *viewmats, const float *Ks,
                        uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                        float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                        int32_t *radii, float *means2d, float *depths, float *conics, float *compensations,
                        void *stream);
int gsx_project_ewa_bwd(const float *means, const float *covars, const float *quats, const float *scales,
                        const float *viewmats, const float *Ks,
                        uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                        float eps2d, int camera_model,
                        const int32_t *radii, const float *conics, const float *compensations,
                        const float *v_means2d, uint32_t v_means2d_stride /* floats per row: 2 if contiguous */,
                        const float *v_depths, const float *v_conics, uint32_t v_conics_stride /* 3 if contiguous */,
                        const float *v_compensations,
                        float *v_means, float *v_covars, float *v_quats, float *v_scales, float *v_viewmats,
                        void *stream);
/* gsx_project_ewa_bwd that also reduces the cotangent of the per-view opacities: v_view_opacities[(b C + c) N + g], elements
 * v_view_opacities_stride floats apart (1 = contiguous; the row stride of gsx_raster3d_bwd's gradient rows when it is their
 * opacity column, read in place) -> v_opacities[b N + g] = sum over c. The kernel reads those rows anyway; autograd would
 * otherwise copy the strided column out with a kernel of its own. No counterpart in the reference (the per-view opacities are a
 * broadcast view there: gsplat/rendering.py:511-520). */
int gsx_project_ewa_bwd_opac(const float *means, const float *covars, const float *quats, const float *scales,
                             const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                             uint32_t width, uint32_t height, float eps2d, int camera_model,
                             const int32_t *radii, const float *conics, const float *compensations,
                             const float *v_means2d, uint32_t v_means2d_stride, const float *v_depths,
                             const float *v_conics, uint32_t v_conics_stride, const float *v_compensations,
                             const float *v_view_opacities, uint32_t v_view_opacities_stride, float *v_means,
                             float *v_covars, float *v_quats, float *v_scales, float *v_viewmats,
                             float *v_opacities, void *stream);

/* ---------------------------------------------------------------------------------------------
 * fully_fused_projection (packed): gsplat::projection_ewa_3dgs_packed{,_bwd}
 * (ext.cpp:1065-1077; host Projection.cpp:858-1215; kernels ProjectionEWA3DGSPacked.cu).
 * Two-step protocol, allocation stays with the caller:
 *   1. gsx_project_ewa_packed_count: visible[B*C*N] int32 (1/0) for every (image, gaussian);
 *      caller runs gsx_scan_i32 over it (inclusive cumsum) and reads the total nnz = cum[last];
 *   2. gsx_project_ewa_packed_write: compacts rows in (batch, camera, gaussian) order into
 *      batch_ids/camera_ids/gaussian_ids int64 [nnz], indptr int32 [B*C+1], radii int32 [nnz,2],
 *      means2d [nnz,2], depths [nnz], conics [nnz,3], compensations [nnz] or NULL.
 * bwd accumulates with atomics: v_means/v_covars/v_quats/v_scales/v_viewmats must be zeroed.
 * ------------------------------------------------------------------------------------------- */
int gsx_project_ewa_packed_count(const float *means, const float *covars, const float *quats, const float *scales,
                                 const float *opacities, const float *viewmats, const float *Ks,
                                 uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                 float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int camera_model, int calc_compensations, int32_t *visible, void *stream);
int gsx_project_ewa_packed_write(const float *means, const float *covars, const float *quats, const float *scales,
                                 const float *opacities, const float *viewmats, const float *Ks,
                                 uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                 float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int camera_model, const int64_t *row_offsets /* INCLUSIVE cumsum of visible (gsx_scan_i32) */,
                                 int64_t nnz,
                                 int64_t *batch_ids, int64_t *camera_ids, int64_t *gaussian_ids, int32_t *indptr,
                                 int32_t *radii, float *means2d, float *depths, float *conics,
                                 float *compensations, void *stream);

/* The same two passes with the rows placed from BLOCK counts (one int32 per 256 (image, Gaussian) pairs, scanned by one
 * workgroup that also publishes the row count) instead of one flag per pair scanned by the caller: identical outputs, no
 * [B C N] temporaries, three launches instead of five. block_counts / block_offsets: [gsx_project_packed_blocks(B C N)]
 * int32. nnz_device / nnz_host may be NULL; nnz_host must be pinned host memory (one 8-byte system-scope store: a caller
 * that set it to -1 beforehand can poll it instead of synchronising the stream). */
int64_t gsx_project_packed_blocks(int64_t pairs);
int gsx_project_ewa_packed_count_blocks(const float *means, const float *covars, const float *quats, const float *scales,
                                        const float *opacities, const float *viewmats, const float *Ks, uint32_t B,
                                        uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                        float near_plane, float far_plane, float radius_clip, int camera_model,
                                        int calc_compensations, int32_t *block_counts, int32_t *block_offsets,
                                        int64_t *nnz_device, int64_t *nnz_host, void *stream);
int gsx_project_ewa_packed_write_blocks(const float *means, const float *covars, const float *quats, const float *scales,
                                        const float *opacities, const float *viewmats, const float *Ks, uint32_t B,
                                        uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                        float near_plane, float far_plane, float radius_clip, int camera_model,
                                        const int32_t *block_offsets, int64_t *batch_ids, int64_t *camera_ids,
                                        int64_t *gaussian_ids, int32_t *indptr, int32_t *radii, float *means2d,
                                        float *depths, float *conics, float *compensations, void *stream);
int gsx_project_ewa_packed_bwd(const float *means, const float *covars, const float *quats, const float *scales,
                               const float *viewmats, const float *Ks,
                               uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                               float eps2d, int camera_model, int64_t nnz,
                               const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                               const float *conics, const float *compensations,
                               const float *v_means2d, uint32_t v_means2d_stride, const float *v_depths,
                               const float *v_conics, uint32_t v_conics_stride,
                               const float *v_compensations,
                               const int32_t *row_map /* NULL, or gsx_packed_row_map output: Gaussian-major walk, every output
                                                         row written once (no atomics, outputs need no zero-fill) */,
                               float *v_means, float *v_covars, float *v_quats, float *v_scales,
                               float *v_viewmats, void *stream);
/* gsx_project_ewa_packed_bwd (with a row map: Gaussian-major, every output written once; without: row-major into ZERO-FILLED
 * outputs, v_opacities included) that also reduces the cotangent of the packed rows' opacities: v_view_opacities[row], v_view_opacities_stride floats apart (1 = contiguous; the row stride of gsx_raster3d_bwd's
 * gradient rows when it is their opacity column) -> v_opacities[b N + g] = sum over the Gaussian's rows (0 without rows).
 * Replaces the index_add (+ zero fill) autograd runs for opacities[gaussian_ids] (reference gsplat/rendering.py:507-510). */
int gsx_project_ewa_packed_bwd_opac(const float *means, const float *covars, const float *quats, const float *scales,
                                    const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                    uint32_t width, uint32_t height, float eps2d, int camera_model, int64_t nnz,
                                    const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                    const float *conics, const float *compensations, const float *v_means2d,
                                    uint32_t v_means2d_stride, const float *v_depths, const float *v_conics,
                                    uint32_t v_conics_stride, const float *v_compensations,
                                    const float *v_view_opacities, uint32_t v_view_opacities_stride,
                                    const int32_t *row_map, float *v_means, float *v_covars, float *v_quats,
                                    float *v_scales, float *v_viewmats, float *v_opacities, void *stream);
/* sparse_grad=True (reference host fn Projection.cpp:1125-1200: `at::zeros({nnz, .})` + make_sparse_coo_grad; kernel
 * ProjectionEWA3DGSPacked.cu:385-684 with sparse_grad): the per-Gaussian gradients are [nnz, 3] / [nnz, 6] / [nnz, 4] /
 * [nnz, 3] ROWS, one per packed row, each written exactly once (no zero-fill needed, no dense [N, .] tensor anywhere). The
 * caller wraps them as COO over gaussian_ids. v_viewmats [B,C,4,4] (zero-filled by the caller) or NULL. */
int gsx_project_ewa_packed_bwd_rows(const float *means, const float *covars, const float *quats, const float *scales,
                                    const float *viewmats, const float *Ks,
                                    uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                    float eps2d, int camera_model, int64_t nnz,
                                    const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                    const float *conics, const float *compensations,
                                    const float *v_means2d, uint32_t v_means2d_stride, const float *v_depths,
                                    const float *v_conics, uint32_t v_conics_stride,
                                    const float *v_compensations,
                                    float *v_means_rows, float *v_covars_rows, float *v_quats_rows, float *v_scales_rows,
                                    float *v_viewmats, void *stream);
/* row_map int32 [B*C*N]: index of the packed row of (batch, camera, gaussian), or -1 when that pair is not stored.
 * Lets the packed backward kernels (projection, SH with D = 3) run one thread per Gaussian over its rows instead of
 * one thread per row with atomics (10 resp. 3*K fp32 atomics per row when a Gaussian is seen by several cameras). */
int gsx_packed_row_map(const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids, int64_t nnz,
                       uint32_t B, uint32_t C, uint32_t N, int32_t *row_map, void *stream);

/* ---------------------------------------------------------------------------------------------
 * spherical_harmonics{,_bwd}: gsplat::spherical_harmonics{,_bwd} (ext.cpp:994-1002; host
 * SphericalHarmonics.cpp; kernels SphericalHarmonicsCUDA.cu:444-569, 786-890).
 * dir = mean - camera centre (centre = -R^T t from viewmats), normalised; Sloan basis, deg <= 4.
 * Dense (nnz < 0): means [B,N,3], viewmats [B,C,4,4], coeffs [N,K,D] shared by all images,
 *        masks bool [B,C,N] or NULL, colors [B,C,N,D].
 * Packed (nnz >= 0): batch/camera/gaussian ids int64 [nnz], masks bool [nnz] or NULL,
 *        colors [nnz,D]; coeffs_gathered=1: coeffs is [nnz,K,D] (pre-gathered by the caller, the
 *        reference contract); coeffs_gathered=0: coeffs is [N,K,D] and is indexed through
 *        gaussian_ids inside the kernel (saves the 4*K*D B/row gather copy; used by our
 *        rasterization() orchestrator).
 * Masked rows are written as zeros. bwd: v_coeffs has the shape of coeffs; it is fully written
 * except in packed+ungathered mode, where it must be ZEROED by the caller. v_means [B,N,3] or
 * NULL, must be ZEROED by the caller (accumulated with atomics).
 * ------------------------------------------------------------------------------------------- */
int gsx_sh_fwd(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
               const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
               const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz /* <0: dense */,
               int coeffs_gathered, uint32_t K, uint32_t D,
               const int32_t *radii /* NULL, or [rows,2]: like masks, a row is live iff both radii > 0 */,
               int post /* 1: colors = max(sh + 0.5, 0), the rasterization() post-op (Rendering.cpp:1160) */,
               float *colors, void *stream);
/* gsx_sh_fwd for D == 3 that ALSO writes, for every live row, the compositing kernels' array-of-structures row
 *   splat_rows [rows][12 floats] = (x, y, conic a, conic b | conic c, opacity, colour 0, colour 1 | colour 2, 0, 0, 0)
 * from the projection's outputs of the same rows (means2d [rows,2], conics [rows,3], opacities [rows]) and the colours it has
 * just computed - so that gsx_raster3d_fwd_rows / gsx_raster3d_bwd_fill_rows stage a list entry with three 16-byte loads from
 * ONE row instead of four gathers from four arrays. The reference has no counterpart: it is a layout of the intermediates of
 * gsplat::rasterization_3dgs (Rendering.cpp:1146-1160 -> :1353-1435), invisible outside rasterization(). 16-byte aligned. */
int gsx_sh_fwd_rows(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                    const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                    const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                    int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, int post, float *colors,
                    const float *means2d, const float *conics, const float *opacities, float *splat_rows, void *stream);
int gsx_sh_bwd(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
               const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
               const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
               int coeffs_gathered, uint32_t K, uint32_t D,
               const int32_t *radii /* as in gsx_sh_fwd */,
               const float *post_colors /* NULL, or the forward output computed with post=1: cuts the gradient where 0 */,
               const float *v_colors, uint32_t v_colors_stride /* floats per row; 0 = D (contiguous) */,
               const int32_t *row_map /* NULL, or gsx_packed_row_map output (packed rows, coeffs_gathered = 0, D = 3):
                                         Gaussian-major walk, v_coeffs / v_means fully written, no atomics */,
               float *v_coeffs, float *v_means,
               float *v_dirs /* NULL, or zero-initialised [rows,3]: per-row d(loss)/d(view direction), from which the
                                caller forms v_viewmats (= t (x) sum v_dir for R, R sum v_dir for t) */,
               void *stream);

/* ---------------------------------------------------------------------------------------------
 * Unscented-Transform projection (3DGUT): gsplat::projection_ut_3dgs_fused (ext.cpp:1230-1239; kernel
 * ProjectionUT3DGSFused.cu; algorithm as gsplat/cuda/_torch_impl_ut.py:69-644). Forward only - the op carries no gradient.
 * Dense rows [B,C,N]. camera_model 0 = pinhole (optionally OpenCV-distorted: radial [B,C,6] (k1..k6; pad 4 -> 6 with
 * zeros), tangential [B,C,2], thin_prism [B,C,4]; NULL = absent), 1 = orthographic (no coefficients), 2 = OpenCV fisheye
 * (k1..k4 in radial[..., 0:4]; fisheye_max_angle [B,C] = largest ray angle the model projects, i.e. where
 * d/dtheta of theta (1 + k1 theta^2 + .. + k4 theta^8) first vanishes, capped at the image corner - computed by the
 * caller, _torch_cameras.py:1344-1521). Global shutter.
 * Seven sigma points mean, mean +- sqrt(3 + lambda) scale_i R[:, i] (lambda = alpha^2 (3 + kappa) - 3) go through the
 * camera model; mean2d / cov2d are the UT-weighted moments of their pixels (sums stop at the first invalid point when
 * require_all_sigma_points_valid). Then blur (+eps2d I) and compensation, conic = inverse, opacity-aware extent (opacities
 * NULL = none), eigenvalue-bounded radii, radius clip, image cull. Rows that fail a check are written as zeros.
 * compensations may be NULL. f-theta cameras: gsx_project_ut_ftheta_fwd below; rolling shutter: gsx_project_ut_rs_fwd. Lidar is not built: -1.
 * ------------------------------------------------------------------------------------------- */
int gsx_project_ut_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                       const float *viewmats, const float *Ks, const float *radial, const float *tangential,
                       const float *thin_prism, const float *fisheye_max_angle, uint32_t B, uint32_t C, uint32_t N,
                       uint32_t width, uint32_t height,
                       float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                       float ut_alpha, float ut_beta, float ut_kappa, float in_image_margin_factor,
                       int require_all_sigma_points_valid, int32_t *radii, float *means2d, float *depths,
                       float *conics, float *compensations, void *stream);
/* The same op for camera_model 3, f-theta (gsplat/cuda/_torch_cameras.py FThetaCamera; parameter record
 * FThetaCameraDistortionParameters, Cameras.h:103-117 / ext.cpp:165-226): the pixel distance from the principal point is a
 * polynomial of the ray angle - `angle_to_pixeldist_poly` when reference_poly = 1, three Newton steps on
 * `pixeldist_to_angle_poly` from that start when reference_poly = 0 - the angle is clamped at max_angle (rays beyond it are
 * invalid; there is no in-front test), pixel = (c, d; e, 1) offset + principal point + 0.5. `ftheta` is a HOST array of 17
 * floats: reference_poly (0 / 1), pixeldist_to_angle_poly[6], angle_to_pixeldist_poly[6] (lowest degree first), max_angle,
 * linear_cde[3]. One record per call (as in the reference); everything else as gsx_project_ut_fwd. */
int gsx_project_ut_ftheta_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                              const float *viewmats, const float *Ks, const float *ftheta, uint32_t B, uint32_t C, uint32_t N,
                              uint32_t width, uint32_t height, float eps2d, float near_plane, float far_plane, float radius_clip,
                              float ut_alpha, float ut_beta, float ut_kappa, float in_image_margin_factor,
                              int require_all_sigma_points_valid, int32_t *radii, float *means2d, float *depths,
                              float *conics, float *compensations, void *stream);
/* The same op with a ROLLING SHUTTER and / or the Euclidean sort depth, for every camera model above (ProjectionUT3DGSFused.cu:
 * 121-127, 239-240, 411; Cameras.cuh:76-135, 362-429, 549-660): `viewmats1` [B,C,4,4] = pose at the END of the frame (NULL with
 * rs_type 4), `rs_type` 0 top-to-bottom, 1 left-to-right, 2 bottom-to-top, 3 right-to-left, 4 global (Cameras.h:38-45). Every
 * sigma point is projected with the start pose (else the end pose), then up to ten rounds of "read-out time of its pixel -> pose at
 * that time (translation lerp, rotation slerp) -> project again"; the Gaussian's own camera-frame position (near / far, depth)
 * uses the pose at mid-frame. `global_z_order` 0: depths = |mean_c| instead of mean_c.z, and f-theta cameras cull near / far on
 * |mean_c|. `ftheta`: the 17-float HOST record of gsx_project_ut_ftheta_fwd (camera_model 3) or NULL. */
int gsx_project_ut_rs_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                          const float *viewmats0, const float *viewmats1, const float *Ks, const float *radial,
                          const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                          const float *ftheta, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                          float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                          int rs_type, int global_z_order, float ut_alpha, float ut_beta, float ut_kappa,
                          float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii,
                          float *means2d, float *depths, float *conics, float *compensations, void *stream);

/* ---------------------------------------------------------------------------------------------
 * From-world ("eval3d") compositing of 3DGUT, FORWARD: the compositing half of gsplat::rasterize_to_pixels_from_world_3dgs
 * (ext.cpp:1241-1252; kernels RasterizeToPixelsFromWorld3DGS*Fwd.cu; math as gsplat/cuda/_torch_impl_eval3d.py:135-495).
 * A sample is the response of the 3D Gaussian along the pixel's ray: M = S^-1 R^T, o' = M (ray_o - mean),
 * d' = M ray_d / |M ray_d|, alpha = min(opacity * exp(-|d' x o'|^2 / 2), 0.99), nothing when -d' . o' < 0; then the
 * classic front-to-back rule (skip alpha < 1/255, stop before transmittance <= 1e-4).
 * means [B,N,3], quats [B,N,4], scales [B,N,3]; colors [I,N,D], opacities [I,N] with I = B * cameras_per_batch and list rows
 * = image * N + gaussian (flatten_ids, isect_offsets as for gsx_raster3d_fwd); rays [I,H,W,6] = world-space origin | unit
 * direction per pixel (the caller generates them from its camera model). Outputs as gsx_raster3d_fwd, except
 * last_ids = -1 where no sample contributed. The backward pass is not built yet.
 * ------------------------------------------------------------------------------------------- */
int gsx_raster_world_fwd(const float *means, const float *quats, const float *scales, const float *colors,
                         const float *opacities, const float *rays, const float *backgrounds, const uint8_t *masks,
                         const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                         uint32_t cameras_per_batch, uint32_t n_gaussians, uint32_t n_isects, uint32_t cdim, uint32_t width,
                         uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *render_colors,
                         float *render_alphas, int32_t *last_ids, void *stream);
/* The same launch + sample_counts int32 [I,H,W] (may be NULL): the number of samples each pixel blended - the op's
 * `return_sample_counts` output (Rasterization.cpp:2404; `n_accumulated` in RasterizeToPixelsFromWorld3DGS.cuh:785). */
int gsx_raster_world_fwd_counts(const float *means, const float *quats, const float *scales, const float *colors,
                                const float *opacities, const float *rays, const float *backgrounds, const uint8_t *masks,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                uint32_t cameras_per_batch, uint32_t n_gaussians, uint32_t n_isects, uint32_t cdim,
                                uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                                float *render_colors, float *render_alphas, int32_t *last_ids, int32_t *sample_counts,
                                void *stream);
/* ... + `use_hit_distance` (the LAST colour channel of every sample is the distance along the ray to its closest approach in
 * world units, |scale * (d' hit_t)|, instead of colors[row][cdim - 1]: RasterizeToPixelsFromWorld3DGS.cuh:715-720, 745-751) and
 * render_normals float [I,H,W,3] (may be NULL): sum of vis * the Gaussian's third axis R[:, 2], unit length, turned to face the
 * ray (:762-768) - the op's `use_hit_distance` / `return_normals` (Rasterization.cpp:2404, 2656-2702). */
int gsx_raster_world_fwd_ex(const float *means, const float *quats, const float *scales, const float *colors,
                            const float *opacities, const float *rays, const float *backgrounds, const uint8_t *masks,
                            const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                            uint32_t cameras_per_batch, uint32_t n_gaussians, uint32_t n_isects, uint32_t cdim,
                            uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                            int use_hit_distance, float *render_colors, float *render_alphas, int32_t *last_ids,
                            int32_t *sample_counts, float *render_normals, void *stream);

/* Backward of gsx_raster_world_fwd (reference host fn rasterize_to_pixels_from_world_3dgs_bwd, Rasterization.cpp:2920;
 * kernel RasterizeToPixelsFromWorld3DGSParallelBatchBwd.cu:300-930). Gradient rows v_rows [I * N][row_stride >= 10 + cdim],
 * ZEROED by the caller: v_mean (3) | torque (3) | v_scale (3) | v_opacity | v_colors[cdim]. `torque` = dL/dw for a world-frame
 * rotation vector w applied to the Gaussian's rotation (R -> exp([w]x) R); the caller sums the rows of one batch's cameras and
 * maps it to the raw quaternion: v_quat = (2 / |q|) (0, torque) (x) q / |q| (Hamilton product, w first). The reference forms
 * v_quats per sample from v_M (:840-890); the torque is the same derivative without the nine-term intermediate whose
 * stretch part cancels. render_alphas / last_ids are the forward outputs; v_render_alphas may be NULL. */
int gsx_raster_world_bwd(const float *means, const float *quats, const float *scales, const float *colors,
                         const float *opacities, const float *rays, const float *backgrounds, const uint8_t *masks,
                         const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
                         const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                         uint32_t n_images, uint32_t cameras_per_batch, uint32_t n_gaussians, uint32_t n_isects,
                         uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                         uint32_t tile_h, float *v_rows, uint32_t row_stride, void *stream);
/* Backward of gsx_raster_world_fwd_ex (RasterizeToPixelsFromWorld3DGSParallelBatchBwd.cu:806-880): the rows of
 * gsx_raster_world_bwd - the hit distance's direct dependence on the scale goes into the v_scale columns, the normals' share
 * (n0 x v_n0, n0 = the Gaussian's unit third axis) into the torque columns. With use_hit_distance the last colour channel
 * receives no colour gradient. v_render_normals float [I,H,W,3] may be NULL.
 * v_rays float [I,H,W,6] (may be NULL; zero-initialised by the caller): the cotangent of the rays, origin | direction - the
 * reference's v_rays output (Rasterization.cpp:2920-3100; its tests differentiate with respect to the rays). */
int gsx_raster_world_bwd_ex(const float *means, const float *quats, const float *scales, const float *colors,
                            const float *opacities, const float *rays, const float *backgrounds, const uint8_t *masks,
                            const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
                            const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                            const float *v_render_normals, uint32_t n_images, uint32_t cameras_per_batch,
                            uint32_t n_gaussians, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
                            uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int use_hit_distance, float *v_rows,
                            uint32_t row_stride, float *v_rays, void *stream);

/* Rays of every pixel of I images, [I,H,W,6] = world-space origin | unit direction, the zero ray where the camera model cannot
 * invert the pixel (the from-world rasterizer gives such a pixel no samples): what the reference's kernels derive per thread when
 * no `rays` tensor is passed (compute_world_ray, RasterizeToPixelsFromWorld3DGS.cuh:349-529; element_to_world_ray_shutter_pose,
 * Cameras.cuh:503-546; the models' image_point_to_camera_ray: Cameras.cuh:717, 866, 1062-1290, 1472-1520, 1672-1760).
 * viewmats_rs NULL / rs_type 4 = global shutter. Coefficients as in gsx_project_ut_rs_fwd, one record per image: radial [I,6],
 * tangential [I,2], thin_prism [I,4], fisheye_max_angle [I]; `ftheta` = the 17-float HOST record of gsx_project_ut_ftheta_fwd. */
int gsx_camera_rays(const float *viewmats, const float *viewmats_rs, const float *Ks, const float *radial,
                    const float *tangential, const float *thin_prism, const float *fisheye_max_angle, const float *ftheta,
                    uint32_t n_images, uint32_t width, uint32_t height, int camera_model, int rs_type, float *rays,
                    void *stream);

/* External (windshield) distortion, the reference's BivariateWindshieldModel (ExternalDistortion.h / .cuh; Python statement
 * gsplat/cuda/_torch_external_distortion.py). `ext` = HOST array of 84 floats: horizontal | vertical | horizontal inverse |
 * vertical inverse polynomial, each in the order-5 triangular layout of 21 coefficients (lower orders zero-padded the way
 * pad_coefficients_to_max_order does, ExternalDistortion.cuh:104-124).
 * gsx_camera_rays_ext = gsx_camera_rays behind the windshield (the camera model's ray goes through the inverse polynomials,
 * Cameras.cuh:473-484; orthographic :862-886); gsx_project_ut_ext_fwd = gsx_project_ut_rs_fwd behind it (every sigma point's
 * ray goes through the forward polynomials before the camera model, Cameras.cuh:462-470; orthographic :795-829). */
int gsx_camera_rays_ext(const float *viewmats, const float *viewmats_rs, const float *Ks, const float *radial,
                        const float *tangential, const float *thin_prism, const float *fisheye_max_angle, const float *ftheta,
                        const float *ext, uint32_t n_images, uint32_t width, uint32_t height, int camera_model, int rs_type,
                        float *rays, void *stream);
int gsx_project_ut_ext_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                           const float *viewmats0, const float *viewmats1, const float *Ks, const float *radial,
                           const float *tangential, const float *thin_prism, const float *fisheye_max_angle, const float *ftheta,
                           const float *ext, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                           float near_plane, float far_plane, float radius_clip, int camera_model, int rs_type,
                           int global_z_order, float ut_alpha, float ut_beta, float ut_kappa, float in_image_margin_factor,
                           int require_all_sigma_points_valid, int32_t *radii, float *means2d, float *depths, float *conics,
                           float *compensations, void *stream);
/* gsplat::distort_camera_rays / gsplat::eval_bivariate_poly (ExternalDistortionWrappers.cu:30-160): the model on n rays [n,3]
 * with ONE pair of polynomials (HOST arrays of 21 floats; the inverse pair undistorts), one bivariate polynomial at n points. */
int gsx_distort_camera_rays(const float *rays, int64_t n, const float *horizontal_poly, const float *vertical_poly, float *out,
                            void *stream);
int gsx_eval_bivariate_poly(const float *x, const float *y, int64_t n, const float *poly, float *out, void *stream);

/* Tile intersection of spinning-lidar cameras: gsplat::intersect_tile_lidar (ext.cpp:1037-1040; Intersect.cpp:388-520;
 * IntersectTileLidar.cu:136-415; torch statement gsplat/cuda/_torch_impl_lidar.py:34-394). means2d / radii are azimuth |
 * elevation in ANGULAR PIXELS (angle * 1024); radii as int32 (what the projection writes) or float (one of the two pointers);
 * the tiling of RowOffsetStructuredSpinningLidarModelParametersExt: fields of view (radians; python doubles), spinning direction
 * (1 = counter-clockwise), n_bins_{azimuth,elevation}, cdf_elevation int32 [res_el + 1], cdf_dense_ray_mask int32
 * [res_el + 1][res_az + 1] (summed-area table of the rays). count -> tiles_per_gauss int32 [rows]; the caller takes its INCLUSIVE
 * prefix sum (int64) and calls emit, which writes (image | tile | depth bits) keys and row ids in the order of the torch
 * statement (elevation-major, region A before region B); the keys then go through gsx_isect_tile_sort / gsx_sort_pairs with
 * tile_width = n_bins_azimuth, tile_height = n_bins_elevation. */
int gsx_isect_lidar_count(const float *means2d, const int32_t *radii_i32, const float *radii_f32, int64_t rows,
                          int64_t n_per_image, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                          double fov_vert_span, int spinning_ccw, uint32_t n_bins_azimuth, uint32_t n_bins_elevation,
                          uint32_t cdf_resolution_azimuth, uint32_t cdf_resolution_elevation, const int32_t *cdf_elevation,
                          const int32_t *cdf_dense_ray_mask, int32_t *tiles_per_gauss, void *stream);
int gsx_isect_lidar_emit(const float *means2d, const int32_t *radii_i32, const float *radii_f32, const float *depths,
                         const int64_t *image_ids, const int64_t *cum_tiles, int64_t rows, int64_t n_per_image,
                         uint32_t n_images, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                         double fov_vert_span, int spinning_ccw, uint32_t n_bins_azimuth, uint32_t n_bins_elevation,
                         uint32_t cdf_resolution_azimuth, uint32_t cdf_resolution_elevation, const int32_t *cdf_elevation,
                         const int32_t *cdf_dense_ray_mask, int64_t *isect_ids, int32_t *flatten_ids, void *stream);

/* Spinning-lidar cameras in the two 3DGUT kernels (camera_model 4; Lidars.cuh:40-330; torch statement
 * gsplat/cuda/_torch_lidars.py:214-374). gsx_project_ut_lidar_fwd = gsplat::projection_ut_3dgs_fused with `lidar_coeffs`: the
 * image point of a sigma point is (azimuth, elevation) * 1024, valid inside the fields of view (+ the UT margin), no
 * image-bounds culling, radial near / far culling and depth with global_z_order = 0; a rolling shutter reads the time of an
 * angle off angles_to_columns_map (int32 [map_h][map_w], device; NULL with rs_type 4). Ks is not used (any [B,C,3,3]).
 * gsx_lidar_rays: the world ray of every element [I, n_rows, n_columns, 6] (origin | unit direction; the zero ray outside the
 * fields of view) - what the from-world kernels derive per thread for lidar elements (element_to_world_ray_shutter_pose). */
int gsx_project_ut_lidar_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                             const float *viewmats0, const float *viewmats1, const float *Ks, double fov_horiz_start,
                             double fov_horiz_span, double fov_vert_start, double fov_vert_span, int spinning_ccw,
                             const int32_t *angles_to_columns_map, uint32_t map_h, uint32_t map_w, uint32_t n_columns, uint32_t B,
                             uint32_t C, uint32_t N, float eps2d, float near_plane, float far_plane, float radius_clip,
                             int rs_type, int global_z_order, float ut_alpha, float ut_beta, float ut_kappa,
                             float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii, float *means2d,
                             float *depths, float *conics, float *compensations, void *stream);
int gsx_lidar_rays(const float *viewmats, const float *viewmats_rs, const float *row_elevations, const float *column_azimuths,
                   const float *row_azimuth_offsets, uint32_t n_images, uint32_t n_rows, uint32_t n_columns,
                   double fov_horiz_start, double fov_horiz_span, double fov_vert_start, double fov_vert_span, double fov_eps,
                   int spinning_ccw, const int32_t *angles_to_columns_map, uint32_t map_h, uint32_t map_w, int rs_type,
                   float *rays, void *stream);

/* assemble_proj_features_unpacked_fwd: gsplat::assemble_proj_features_unpacked_fwd (ext.cpp:1015-1020; host
 * SphericalHarmonics.cpp:572-676; kernel SphericalHarmonicsCUDA.cu:1100-1250). Dense rows only. Writes
 * out [B,C,N, Dc + E + has_depth] = [ post(SH colours of coeffs [N,K,Dc]) | extra (+0.5 when extra_post == 1) | depth ]
 * in one pass. color_post / extra_post: 0 none, 1 x + 0.5, 2 max(x + 0.5, 0). extra is [B,C,N,E] (extra_has_c) or
 * [B,N,E]; depths [B,C,N], or NULL with has_depth for a zero column; masks bool [B,C,N] or NULL: masked rows get zero
 * colours (their extra / depth columns are still written). relu_mask bool [B,C,N,Dc] or NULL (only with color_post = 2):
 * out > 0 on the unmasked rows, untouched elsewhere (the caller zero-fills it). */
int gsx_assemble_features_fwd(int degrees_to_use, uint32_t B, uint32_t C, uint32_t N, uint32_t K, uint32_t Dc, uint32_t E,
                              int color_post, int extra_post, int has_depth, int extra_has_c,
                              const float *means, const float *viewmats, const float *coeffs, const float *extra,
                              const float *depths, const uint8_t *masks, float *out, uint8_t *relu_mask, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Spherical harmonics over a window of bands, fp32 or fp16 coefficients, D = 3 (csrc/sh_band.hip):
 *   first_band = 1: gsplat::spherical_harmonics_l1_plus{,_bwd} (ext.cpp:1005-1014; SphericalHarmonicsL1PlusCUDA.cu:441
 *                   fwd, :648 bwd): coeffs = shN [N, K-1, 3], bands 1 .. K-1, read IN PLACE (no concatenation with sh0);
 *   first_band = 0, coeff_dtype = 1: gsplat::spherical_harmonics{,_bwd} with at::kHalf coefficients
 *                   (SphericalHarmonicsCUDA.cu:609-638, 1306-1328): coefficients half, arithmetic and colours float.
 * K counts bases INCLUDING band 0 (K >= (degrees_to_use + 1)^2); coefficient rows in memory hold K - first_band bases and are
 * indexed by Gaussian (packed rows through gaussian_ids). coeff_dtype: 0 float32, 1 float16 (v_coeffs has the same type).
 * Forward: colours [rows, 3] (masked rows 0). Backward: one thread per Gaussian walks the images - v_coeffs
 * [N, K - first_band, 3] and v_means [B, N, 3] are written once per Gaussian (no atomics, no zero fill needed), v_dirs
 * [rows, 3] (optional) receives d(loss)/d(view direction) per row; packed rows need row_map = gsx_packed_row_map.
 * ------------------------------------------------------------------------------------------- */
int gsx_sh_band_fwd(int degrees_to_use, int first_band, int coeff_dtype, const float *means, const float *viewmats,
                    const void *coeffs, const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                    const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz, uint32_t K,
                    float *colors, void *stream);
int gsx_sh_band_bwd(int degrees_to_use, int first_band, int coeff_dtype, const float *means, const float *viewmats,
                    const void *coeffs, const uint8_t *masks, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                    uint32_t K, const float *v_colors, const int32_t *row_map, void *v_coeffs, float *v_means,
                    float *v_dirs, void *stream);

/* ---------------------------------------------------------------------------------------------
 * isect_tiles: gsplat::intersect_tile (ext.cpp:1022-1026; host Intersect.cpp:170-329; kernel
 * IntersectTile.cu:214-464) split into its stages so that allocation stays with the caller:
 *   gsx_isect_count  -> tiles_per_gauss int32 [R]
 *   gsx_scan_i32     -> cum int64 [R] (INCLUSIVE prefix sum, like the reference's cumsum);
 *                       total = cum[R-1]
 *   gsx_isect_emit   -> isect_ids int64 [M], flatten_ids int32 [M] (unsorted)
 *   gsx_sort_pairs   -> stable LSD radix sort on key bits [0, end_bit)
 *   gsx_isect_offsets-> gsplat::intersect_offset (ext.cpp:1027; IntersectTile.cu:925-988)
 * conics+opacities non-NULL selects the exact ellipse/tile test (AccuTile/SNUGBOX), else AABB.
 * image_ids (int64 [R]) non-NULL = packed rows. Keys: image << (32+tile_bits) | tile << 32 |
 * bits(float depth).
 * ------------------------------------------------------------------------------------------- */
int gsx_isect_count(const float *means2d, const int32_t *radii, const float *conics, const float *opacities,
                    const int64_t *image_ids, int64_t rows, uint32_t n_per_image, uint32_t n_images,
                    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *tiles_per_gauss, void *stream);
int gsx_isect_emit(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                   const float *opacities, const int64_t *image_ids, const int64_t *cum_tiles_per_gauss,
                   int64_t rows, uint32_t n_per_image, uint32_t n_images,
                   uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                   int64_t *isect_ids, int32_t *flatten_ids, void *stream);
/* float64 rows (the reference dispatches intersect_tile over float and double, IntersectTile.cu AT_DISPATCH_FLOATING_TYPES;
 * tests/test_basic.py:1268-1316): radius-box enumeration in double, the depth in the key narrowed to float32. The exact
 * ellipse test (conics + opacities) is fp32 only. */
int gsx_isect_count_f64(const double *means2d, const int32_t *radii, const int64_t *image_ids, int64_t rows,
                        uint32_t n_per_image, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                        int32_t *tiles_per_gauss, void *stream);
int gsx_isect_emit_f64(const double *means2d, const int32_t *radii, const double *depths, const int64_t *image_ids,
                       const int64_t *cum_tiles_per_gauss, int64_t rows, uint32_t n_per_image, uint32_t n_images,
                       uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int64_t *isect_ids, int32_t *flatten_ids,
                       void *stream);
int64_t gsx_scan_workspace_bytes(int64_t n);
int gsx_scan_i32(const int32_t *in, int64_t n, int64_t *out_inclusive, void *workspace, int64_t workspace_bytes,
                 void *stream);
int64_t gsx_sort_pairs_workspace_bytes(int64_t n);
/* Sorts (keys, vals) by key bits [0,end_bit). keys_alt/vals_alt are ping-pong buffers of the same size.
 * On return *result_in_alt tells whether the sorted data is in the alt buffers (1) or the primary (0). */
int gsx_sort_pairs(int64_t *keys, int32_t *vals, int64_t *keys_alt, int32_t *vals_alt, int64_t n, int end_bit,
                   void *workspace, int64_t workspace_bytes, int *result_in_alt, void *stream);
/* Structured sort of the emitted pairs = gsx_sort_pairs on key bits [0, 32+tile_bits+image_bits) with identical output,
 * done as one bucketing pass by (image, tile) + a per-tile depth sort in LDS (csrc/tile_sort.hip; 40 B instead of 144 B of
 * HBM traffic per pair). Supported while n_images*tile_w*tile_h fits the LDS histogram (gsx_isect_tile_sort_supported);
 * inputs are not modified. */
int gsx_isect_tile_sort_supported(uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int64_t gsx_isect_tile_sort_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int gsx_isect_tile_sort(const int64_t *isect_ids, const int32_t *flatten_ids, int64_t n_isects, uint32_t n_images,
                        uint32_t tile_w, uint32_t tile_h, int64_t *isect_ids_sorted, int32_t *flatten_ids_sorted,
                        void *workspace, int64_t workspace_bytes, void *stream);
int gsx_isect_offsets(const int64_t *isect_ids_sorted, int64_t n_isects, uint32_t n_images, uint32_t tile_w,
                      uint32_t tile_h, int32_t *offsets /* [I,tile_h,tile_w] */, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused isect_tiles(sort=True) + isect_offset_encode (same outputs as gsx_isect_count/emit + gsx_isect_tile_sort +
 * gsx_isect_offsets, bit for bit): the per-(chunk, tile) histogram is taken while counting and the emission writes
 * each (depth, row) pair directly into its tile's segment, so the unsorted key/value arrays and the histogram / scatter
 * passes never touch HBM (8 B instead of 40 B per intersection ahead of the per-tile sort; csrc/isect_fused.hip).
 * Dense rows [n_images * N] (any n_images while n_images * tiles fits the LDS histogram), or packed rows of ONE image.
 *   1. gsx_isect_fused_count: tiles_per_gauss int32 [rows], isect_offsets int32 [n_images * tiles] (= intersect_offset),
 *      *n_isects (int64 in device memory OR in pinned, device-mapped host memory: 8 bytes need no copy kernel). The caller
 *      reads n_isects, allocates the exact-length outputs, then
 *   2. gsx_isect_fused_emit_sort with the SAME count workspace: isect_ids int64 [n_isects], flatten_ids int32 [n_isects].
 * ------------------------------------------------------------------------------------------- */
int gsx_isect_fused_supported(uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed);
int64_t gsx_isect_fused_count_workspace_bytes(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int64_t gsx_isect_fused_emit_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int gsx_isect_fused_count(const float *means2d, const int32_t *radii, const float *conics, const float *opacities,
                          const uint8_t *tile_mask /* NULL, or [n_images * tiles] flags: only flagged tiles receive
                                                      intersections (gsplat::intersect_tile_sparse); tiles_per_gauss may then be NULL */,
                          int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                          int32_t *tiles_per_gauss, int32_t *isect_offsets, int64_t *n_isects,
                          int64_t *max_tile_len /* optional (NULL): length of the longest tile list, written before n_isects */,
                          void *count_workspace, int64_t count_workspace_bytes, void *stream);
int gsx_isect_fused_emit_sort(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                              const float *opacities, const uint8_t *tile_mask, int64_t rows, uint32_t n_images, uint32_t tile_size,
                              uint32_t tile_w, uint32_t tile_h, void *count_workspace, int64_t count_workspace_bytes,
                              const int32_t *isect_offsets, int64_t n_isects, int64_t *isect_ids_sorted,
                              int32_t *flatten_ids_sorted, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Binned isect_tiles(sort=True) + isect_offset_encode (csrc/isect_binned.hip): the same outputs as the fused pair above,
 * bit for bit (= gsplat::intersect_tile with sort + gsplat::intersect_offset: ext.cpp:1022-1027; Intersect.cpp:170-329;
 * IntersectTile.cu:214-464, 925-988, 1078-1121), built tile-owner-major: the screen is cut into bins of 4 x 2 tiles, every
 * row is entered once per bin its tile rectangle overlaps ((depth, row) + a tile mask from the exact walk clipped to the
 * bin), and ONE workgroup per bin deals its entries into per-tile runs in LDS, one wave sorts each run by (depth, row) and
 * writes keys and row ids once, contiguously. No per-intersection scattered stores, no per-chunk tile table. Dense rows
 * [n_images * N] or packed rows of ONE image. gsx_isect_binned_supported() says whether an input is in the range where this
 * beats the fused pair (mid-sized inputs with tile lists of up to ~512 entries: DESIGN.md section 4).
 *   1. gsx_isect_binned_count (needs depths already): tiles_per_gauss int32 [rows] (or NULL), isect_offsets int32
 *      [n_images * tiles], *n_isects (device or pinned host memory). *n_isects == GSX_ISECT_RETRY (-2): the entries did
 *      not fit the workspace (a scene of very large Gaussians) or one bin is too crowded for the sort's LDS arena (a real
 *      scene's dense region; not when GSX_ISECT_PATH=binned forces the path) - run gsx_isect_fused_* instead.
 *   2. gsx_isect_binned_emit_sort with the SAME count workspace (untouched in between).
 * ------------------------------------------------------------------------------------------- */
#define GSX_ISECT_RETRY (-2)
int gsx_isect_binned_supported(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed);
/* A binding that got GSX_ISECT_RETRY says so (gsx_isect_binned_note_retry): the next 63 intersections of that shape (row count
 * to 64 k, images, tile grid) then skip the attempt - a trainer renders the same clustered scene every step - and the 64th
 * probes again. gsx_isect_binned_supported() is a pure query (range + notes; asking changes nothing);
 * gsx_isect_binned_should_try() is the same answer AND counts one skipped intersection: a binding calls it exactly once per
 * intersection, where it chooses the path, and carries that choice to the second half. Same reference op as
 * gsx_isect_binned_count (gsplat::intersect_tile, Intersect.cpp:170-329). */
int gsx_isect_binned_should_try(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed);
int gsx_isect_binned_note_retry(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h); /* returns 0 */
/* WHOSE retry notes: never the process'. They live in a path-memory object; the three functions above read and write the one
 * that is CURRENT for the calling thread - a private per-thread default until the caller installs its own. A binding that
 * serves several independent callers (two trainers, a trainer and a viewer) creates one object per caller and brackets that
 * caller's intersections with gsx_isect_path_memory_use(mem) ... gsx_isect_path_memory_use(previous); which kernel runs then
 * depends on the history of that caller alone. _use returns the previously current object (NULL = the thread's default),
 * _destroy of the current object falls back to the default. No reference counterpart: the reference has one intersection path
 * (Intersect.cpp:170-329); this is bookkeeping of the path choice inside gsplat::intersect_tile. */
void *gsx_isect_path_memory_create(void);
void gsx_isect_path_memory_destroy(void *mem);
void *gsx_isect_path_memory_use(void *mem);
int64_t gsx_isect_binned_count_workspace_bytes(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int64_t gsx_isect_binned_emit_workspace_bytes(int64_t n_isects);
int gsx_isect_binned_count(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                           const float *opacities, const uint8_t *tile_mask, int64_t rows, uint32_t n_images,
                           uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *tiles_per_gauss,
                           int32_t *isect_offsets, int64_t *n_isects, int64_t *max_tile_len /* optional, as above */,
                           void *count_workspace, int64_t count_workspace_bytes, void *stream);
int gsx_isect_binned_emit_sort(int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                               void *count_workspace, int64_t count_workspace_bytes, const int32_t *isect_offsets,
                               int64_t n_isects, int64_t longest_list /* from the count half; 0 = unknown */,
                               int64_t *isect_ids_sorted, int32_t *flatten_ids_sorted, void *workspace,
                               int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * rasterize_to_pixels (3DGS): gsplat::rasterize_to_pixels_3dgs{,_bwd} (ext.cpp:1079-1089; host
 * Rasterization.cpp:275-365, 484-587; kernels RasterizeToPixels3DGSSerialBatch{Fwd,Bwd}.cu).
 * Any channel count >= 1 (chunked by 32 internally); tile_size in [1,16].
 * fwd outputs: render_colors [I,H,W,cdim], render_alphas [I,H,W,1], last_ids int32 [I,H,W].
 * bwd: ONE zero-initialised array-of-structures gradient buffer v_rows [R][row_stride] (R = rows of means2d):
 *   row = (v_means2d.x, v_means2d.y, v_conics.a, .b, .c, v_opacities, [v_means2d_abs.x, .y if has_abs], v_colors[cdim])
 * so the reference's v_means2d / v_conics / v_colors / v_opacities (/ absgrad) tensors are COLUMN VIEWS of it
 * (row_stride >= 6 + 2*has_abs + cdim). One Gaussian's gradients share a cache line and the kernel adds consecutive
 * floats from consecutive lanes: ~6 L2 atomic transactions per wave instruction instead of 64 with one tensor per
 * quantity (measured: 160 us of a 707 us launch at 1M Gaussians / 1080p). gsx_project_ewa*_bwd read v_means2d /
 * v_conics through a row stride for the same reason. v_render_alphas may be NULL (no gradient reaches the alphas:
 * treated as zeros). Likewise v_depths may be NULL in gsx_project_ewa*_bwd.
 * v_backgrounds is a torch-side reduction in the reference (Rasterization.cpp:567-577) and in the shim.
 * ------------------------------------------------------------------------------------------- */
int gsx_raster3d_fwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                     uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                     float *render_colors, float *render_alphas, int32_t *last_ids, void *stream);
/* gsx_raster3d_fwd with the 48-byte array-of-structures rows of gsx_sh_fwd_rows beside the four arrays (cdim == 3; NULL = the
 * plain call). Same op (gsplat::rasterize_to_pixels_3dgs), same results bit for bit: the rows hold the same values. */
int gsx_raster3d_fwd_rows(const float *means2d, const float *conics, const float *colors, const float *opacities,
                          const float *splat_rows, const float *backgrounds, const uint8_t *masks,
                          const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects,
                          uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                          float *render_colors, float *render_alphas, int32_t *last_ids, void *stream);
/* The longest-list length above which a caller should take gsx_raster3d_{fwd,bwd}_seg: max(2 seg_len, 3 x the mean list,
 * n_isects / 1024 = a workgroup slot's share of the launch); inside them, lists longer than max(2 seg_len, 3 x the mean) are cut.
 * A caller that knows the longest list (the intersection reports it) takes the _seg entries only when it exceeds this. */
int64_t gsx_raster3d_seg_cut(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, uint32_t seg_len);

/* Forward with long tile lists cut into segments of seg_len entries that separate workgroups composite and a per-pixel
 * combine (csrc/raster3d_seg.hip): same outputs as gsx_raster3d_fwd up to fp32 rounding of the transmittance products. For
 * scenes with skewed lists (the reference's garden profile: mean 386, longest 8822 entries) the per-tile launch is as long
 * as its longest list; tiles on which the reference's early termination would fire are walked the ordinary way. The caller
 * decides when it pays (gsx_isect_*_count report the longest list) and provides the workspace. */
int64_t gsx_raster3d_seg_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, uint32_t cdim,
                                         uint32_t seg_len);
int gsx_raster3d_fwd_seg(const float *means2d, const float *conics, const float *colors, const float *opacities,
                         const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                         const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                         uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                         float *render_colors, float *render_alphas, int32_t *last_ids, uint32_t seg_len, void *workspace,
                         int64_t workspace_bytes, void *stream);
/* Backward counterpart of gsx_raster3d_fwd_seg (<= 4 channels, 16 x 16 tiles, no absgrad - where the backward runs its
 * variants T / W): a pre-pass gives every slice its own transmittance and colour-cotangent sum, a per-pixel prefix turns them
 * into the transmittance and the "behind" sum at the END of every slice, and the slices are then walked back to front
 * independently, together with the short tiles. `seg_len` is the forward's slice length; the one-wave-per-tile kernel cuts
 * its slices half as long (one instruction stream per slice), so the workspace has its own size function. Replaces the
 * same reference op as gsx_raster3d_bwd (gsplat::rasterize_to_pixels_3dgs_bwd, Rasterization.cpp:484-587). */
int64_t gsx_raster3d_bwd_seg_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, uint32_t cdim,
                                             uint32_t seg_len);
int gsx_raster3d_bwd_seg(const float *means2d, const float *conics, const float *colors, const float *opacities,
                         const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                         const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                         const float *v_render_colors, const float *v_render_alphas, uint32_t n_images, uint32_t n_isects,
                         uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                         uint32_t tile_h, float *v_rows, uint32_t row_stride, uint32_t seg_len, void *workspace,
                         int64_t workspace_bytes, void *stream);
/* gsx_raster3d_bwd_seg for a caller that still holds the workspace of the gsx_raster3d_fwd_seg call over the SAME lists (same
 * n_isects, cdim <= 4, seg_len; not written since): the forward's compositing pass left every slice's colour sums and end
 * transmittance per pixel, which give the backward the state at the end of every slice directly - the pre-pass (a second
 * evaluation of every entry of the long lists, ~0.2 of the segmented backward on the reference's garden profile) is not run.
 * fwd_workspace is only read (a retained graph may run the backward again). NULL, or a backward whose slices are shorter than
 * the forward's (GSX_RASTER3D_BWD_SEG=w): exactly gsx_raster3d_bwd_seg. Results equal gsx_raster3d_bwd_seg up to the association
 * order of the "behind" sums. Replaces the same reference op (gsplat::rasterize_to_pixels_3dgs_bwd, Rasterization.cpp:484-587). */
int gsx_raster3d_bwd_seg_reuse(const float *means2d, const float *conics, const float *colors, const float *opacities,
                               const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                               const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                               const float *v_render_colors, const float *v_render_alphas, uint32_t n_images,
                               uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                               uint32_t tile_w, uint32_t tile_h, float *v_rows, uint32_t row_stride, uint32_t seg_len,
                               const void *fwd_workspace, int64_t fwd_workspace_bytes, void *workspace,
                               int64_t workspace_bytes, void *stream);
int gsx_raster3d_bwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                     const float *v_render_colors, const float *v_render_alphas,
                     uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                     uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                     int has_abs, float *v_rows, uint32_t row_stride, void *stream);
/* The same launch with the tiles taken LONGEST-FIRST: what a tile costs the backward is its list up to its last contributor
 * (49 .. 486 entries on the c3 scene around a mean of 160), and in launch order the kernel lasts as long as its unluckiest
 * workgroup slot. With a workspace of gsx_raster3d_bwd_workspace_bytes() two small launches sort the tiles by that cost
 * (per XCD, so that neighbouring tiles keep sharing an L2) before the compositing launch; results are the same. NULL / too
 * small a workspace, sparse layouts, absgrad, more than four channels: launch order, exactly gsx_raster3d_bwd.
 * v_colors_pixel_stride >= 0: v_render_colors is read in place from a layout that is linear in the pixel index
 * p = (image * height + y) * width + x - element (p, k) at p * v_colors_pixel_stride + k * v_colors_channel_stride floats
 * (autograd hands cotangents over as views: the gradient of sum() is one float with both strides 0); -1 = contiguous. */
int64_t gsx_raster3d_bwd_workspace_bytes(uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
int gsx_raster3d_bwd_ws(const float *means2d, const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                     const float *v_render_colors, const float *v_render_alphas,
                     uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                     uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                     int has_abs, float *v_rows, uint32_t row_stride, int64_t v_colors_pixel_stride,
                     int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream);
/* gsx_raster3d_bwd_ws for gradient rows that are NOT zero-filled yet: the call fills v_rows_to_fill rows of row_stride floats
 * itself - inside the tile-order cost kernel when one is launched (a kernel short of memory work), with a memset otherwise.
 * v_rows_to_fill == 0 is exactly gsx_raster3d_bwd_ws (the caller filled the rows). */
int gsx_raster3d_bwd_fill(const float *means2d, const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                     const float *v_render_colors, const float *v_render_alphas,
                     uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                     uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                     int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill, int64_t v_colors_pixel_stride,
                     int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream);
/* gsx_raster3d_bwd_fill with the same rows (cdim == 3; read by the one-wave-per-tile kernel, ignored by the others). */
int gsx_raster3d_bwd_fill_rows(const float *means2d, const float *conics, const float *colors, const float *opacities,
                               const float *splat_rows, const float *backgrounds, const uint8_t *masks,
                               const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
                               const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                               uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
                               uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows,
                               uint32_t row_stride, int64_t v_rows_to_fill, int64_t v_colors_pixel_stride,
                               int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream);

/* Sparse pixel sets: gsplat::rasterize_to_pixels_sparse{,_bwd} (ext.cpp:1090-1104; RasterizeToPixelsSparse{Fwd,Bwd}.cu,
 * RasterizeSparseAddressing.cuh). One workgroup per ACTIVE tile (active_tiles int32 [AT], ascending dense tile ids;
 * tile_offsets int32 [AT+1] from the masked intersection); tile_pixel_mask uint64 [AT, words] = raster-order bitmask of the
 * requested pixels, tile_pixel_cumsum int64 [AT] inclusive, pixel_map int64 [P]: position in (tile, in-tile) order ->
 * caller's pixel index = the ROW of the [P, ...] outputs / cotangents. Everything else as gsx_raster3d_{fwd,bwd}. */
int gsx_raster3d_sparse_fwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                            const float *backgrounds, const uint8_t *masks, const int32_t *active_tiles,
                            const int32_t *tile_offsets, const int32_t *flatten_ids, const uint64_t *tile_pixel_mask,
                            const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active,
                            uint32_t words_per_tile, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                            uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *render_colors,
                            float *render_alphas, int32_t *last_ids, void *stream);
int gsx_raster3d_sparse_bwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                            const float *backgrounds, const uint8_t *masks, const int32_t *active_tiles,
                            const int32_t *tile_offsets, const int32_t *flatten_ids, const uint64_t *tile_pixel_mask,
                            const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active,
                            uint32_t words_per_tile, const float *render_alphas, const int32_t *last_ids,
                            const float *v_render_colors, const float *v_render_alphas, uint32_t n_images,
                            uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                            uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride,
                            void *stream);

/* ---------------------------------------------------------------------------------------------
 * proj(): gsplat::projection_ewa_simple{,_bwd} (ext.cpp:1043-1050; ProjectionEWASimple.cu). Camera-space means
 * [rows,3] and covariances [rows,3,3] (rows = B*C*N, camera of a row = row / n_per_camera, Ks [B*C,3,3]) -> means2d
 * [rows,2], covars2d [rows,2,2]; no blur, no culling. bwd fully writes v_means [rows,3] and v_covars [rows,3,3].
 * ------------------------------------------------------------------------------------------- */
int gsx_project_simple_fwd(const float *means, const float *covars, const float *Ks, int64_t rows,
                           uint32_t n_per_camera, uint32_t width, uint32_t height, int camera_model, float *means2d,
                           float *covars2d, void *stream);
int gsx_project_simple_bwd(const float *means, const float *covars, const float *Ks, int64_t rows,
                           uint32_t n_per_camera, uint32_t width, uint32_t height, int camera_model,
                           const float *v_means2d, const float *v_covars2d, float *v_means, float *v_covars,
                           void *stream);

/* ---------------------------------------------------------------------------------------------
 * rasterize_to_indices: gsplat::rasterize_to_indices_3dgs / _2dgs (ext.cpp:1105-1109, 1200-1204; kernels
 * RasterizeToIndices{3DGS,2DGS}SerialBatch.cu). mode 0: geom = conics [R,3]; mode 1: geom = ray_transforms [R,9].
 * range_start/range_end count batches of tile_size^2 entries of each tile list; transmittances [I,H,W] is the state at
 * range_start. Pass 1 (chunk_starts NULL): chunk_cnts int32 [I,H,W]; pass 2: gaussian_ids / pixel_ids int64 [n_elems]
 * at chunk_starts (exclusive cumsum of the counts); pixel_ids hold pixel + image*H*W.
 * ------------------------------------------------------------------------------------------- */
int gsx_raster_indices(int mode, uint32_t range_start, uint32_t range_end, const float *transmittances,
                       const float *means2d, const float *geom, const float *opacities, const int32_t *isect_offsets,
                       const int32_t *flatten_ids, uint32_t n_images, uint32_t n_per_image, uint32_t n_isects,
                       uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                       const int32_t *chunk_starts, int32_t *chunk_cnts, int64_t *gaussian_ids, int64_t *pixel_ids,
                       void *stream);

/* ---------------------------------------------------------------------------------------------
 * 2DGS projection: gsplat::projection_2dgs_fused{,_bwd} / projection_2dgs_packed{,_bwd}
 * (ext.cpp:1163-1184; kernels Projection2DGSFused.cu:39-339, 341-505, Projection2DGSPacked.cu; VJP
 * Projection2DGS.cuh:29-115). means [B,N,3], quats [B,N,4] (wxyz, normalised inside), scales [B,N,3] (the
 * third scale is ignored), viewmats [B,C,4,4], Ks [B,C,3,3]. Outputs: radii int32 [B,C,N,2], means2d [B,C,N,2],
 * depths [B,C,N], ray_transforms [B,C,N,3,3] (rows of K [RS0 RS1 mean_c]), normals [B,C,N,3] (camera space,
 * facing the camera). Culled rows: radii = 0 and zeros elsewhere. Packed = same count / scan / write protocol as
 * gsx_project_ewa_packed_*. bwd (dense) fully writes v_means/v_quats/v_scales; v_viewmats (may be NULL) must be
 * zeroed. bwd (packed) accumulates with atomics: all outputs must be zeroed.
 * ------------------------------------------------------------------------------------------- */
int gsx_project_2dgs_fwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                         const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                         float near_plane, float far_plane, float radius_clip, int32_t *radii, float *means2d,
                         float *depths, float *ray_transforms, float *normals, void *stream);
int gsx_project_2dgs_packed_count(const float *means, const float *quats, const float *scales, const float *viewmats,
                                  const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                  float near_plane, float far_plane, float radius_clip, int32_t *visible, void *stream);
int gsx_project_2dgs_packed_write(const float *means, const float *quats, const float *scales, const float *viewmats,
                                  const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                  float near_plane, float far_plane, float radius_clip, const int64_t *row_offsets,
                                  int64_t nnz, int64_t *batch_ids, int64_t *camera_ids, int64_t *gaussian_ids,
                                  int32_t *indptr, int32_t *radii, float *means2d, float *depths,
                                  float *ray_transforms, float *normals, void *stream);
int gsx_project_2dgs_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                         const float *Ks, uint32_t B, uint32_t C, uint32_t N, const int32_t *radii,
                         const float *ray_transforms, const float *v_means2d, const float *v_depths /* may be NULL */,
                         const float *v_ray_transforms, const float *v_normals,
                         uint32_t v_row_stride /* 0: the three gradients are contiguous [rows,2] / [rows,9] / [rows,3];
                                                  else they are column views of gsx_raster2d_bwd's v_rows with this stride */,
                         uint32_t v_depths_stride /* floats between consecutive rows' depth cotangents: 0 / 1 = contiguous, or
                                                     the row stride of v_rows when it is their depth-channel column */,
                         float *v_means, float *v_quats, float *v_scales, float *v_viewmats, void *stream);
/* gsx_project_2dgs_bwd that also reduces the cotangent of the per-view opacities (as gsx_project_ewa_bwd_opac):
 * v_view_opacities[(b C + c) N + g], v_view_opacities_stride floats apart (1 = contiguous; the row stride of gsx_raster2d_bwd's
 * gradient rows when it is their opacity column) -> v_opacities[b N + g] = sum over c. */
int gsx_project_2dgs_bwd_opac(const float *means, const float *quats, const float *scales, const float *viewmats,
                              const float *Ks, uint32_t B, uint32_t C, uint32_t N, const int32_t *radii,
                              const float *ray_transforms, const float *v_means2d, const float *v_depths,
                              const float *v_ray_transforms, const float *v_normals, uint32_t v_row_stride,
                              uint32_t v_depths_stride, const float *v_view_opacities, uint32_t v_view_opacities_stride, float *v_means,
                              float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities, void *stream);
int gsx_project_2dgs_packed_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                                const float *Ks, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                                const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                const float *ray_transforms, const float *v_means2d, const float *v_depths,
                                const float *v_ray_transforms, const float *v_normals, uint32_t v_row_stride,
                                float *v_means, float *v_quats, float *v_scales, float *v_viewmats, void *stream);
/* sparse_grad=True (reference Projection.cpp:1780-1863): v_means / v_quats / v_scales are [nnz, 3] / [nnz, 4] / [nnz, 3]
 * rows, one per packed row, each written once; the caller wraps them as COO over gaussian_ids. */
int gsx_project_2dgs_packed_bwd_rows(const float *means, const float *quats, const float *scales, const float *viewmats,
                                     const float *Ks, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                                     const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                     const float *ray_transforms, const float *v_means2d, const float *v_depths,
                                     const float *v_ray_transforms, const float *v_normals, uint32_t v_row_stride,
                                     float *v_means, float *v_quats, float *v_scales, float *v_viewmats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * rasterize_to_pixels (2DGS): gsplat::rasterize_to_pixels_2dgs{,_bwd} (ext.cpp:1186-1199; kernels
 * RasterizeToPixels2DGSSerialBatch{Fwd,Bwd}.cu). 1 <= cdim <= 32 in one launch (the reference does not chunk
 * channels on this path either); the LAST channel is the depth used by the distortion / median outputs.
 * fwd outputs: render_colors [I,H,W,cdim], render_alphas [I,H,W,1], render_normals [I,H,W,3], render_distort
 * [I,H,W,1] (zeros unless distloss), render_median [I,H,W,1], last_ids / median_ids int32 [I,H,W].
 * bwd: v_render_distort NULL = distloss off. ONE zero-initialised array-of-structures gradient buffer v_rows
 * [R][row_stride], row = (v_means2d 2 | v_opacities 1 | v_densify 2 | v_normals 3 | v_ray_transforms 9 |
 * [v_means2d_abs 2 if has_abs] | v_colors cdim), row_stride >= 17 + 2*has_abs + cdim: the reference's gradient tensors
 * are column views of it (same reason and measurement as gsx_raster3d_bwd).
 * ------------------------------------------------------------------------------------------- */
int gsx_raster2d_fwd(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                     const float *normals, const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                     uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int distloss,
                     float *render_colors, float *render_alphas, float *render_normals, float *render_distort,
                     float *render_median, int32_t *last_ids, int32_t *median_ids, void *stream);
int gsx_raster2d_bwd(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                     const float *normals, const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_colors, const float *render_alphas,
                     const int32_t *last_ids, const int32_t *median_ids, const float *v_render_colors,
                     const float *v_render_alphas, const float *v_render_normals, const float *v_render_distort,
                     const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                     uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows,
                     uint32_t row_stride, void *stream);
/* The same launch with the tiles taken longest-first (see gsx_raster3d_bwd_ws; workspace of
 * gsx_raster3d_bwd_workspace_bytes(n_images, tile_w, tile_h); NULL = launch order). */
int gsx_raster2d_bwd_ws(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                     const float *normals, const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_colors, const float *render_alphas,
                     const int32_t *last_ids, const int32_t *median_ids, const float *v_render_colors,
                     const float *v_render_alphas, const float *v_render_normals, const float *v_render_distort,
                     const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                     uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows,
                     uint32_t row_stride, void *workspace, int64_t workspace_bytes, void *stream);
/* gsx_raster2d_bwd_ws for gradient rows that are NOT zero-filled yet (v_rows_to_fill rows; 0 = the caller filled them): the
 * call fills them itself, inside the tile-order cost kernel when one is launched. In both entries v_render_alphas,
 * v_render_normals, v_render_distort and v_render_median may be NULL (= zeros: an output the loss does not use). */
int gsx_raster2d_bwd_fill(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                     const float *normals, const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const float *render_colors, const float *render_alphas,
                     const int32_t *last_ids, const int32_t *median_ids, const float *v_render_colors,
                     const float *v_render_alphas, const float *v_render_normals, const float *v_render_distort,
                     const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                     uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows,
                     uint32_t row_stride, int64_t v_rows_to_fill, void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * 2DGS per-pixel post-processing (reference: gsplat/rendering.py:1519-1552; C++ orchestrator
 * gsplat/cuda/csrc/Rendering.cpp:1653-1702 depth_to_points_2dgs / depth_to_normal_2dgs and :1905-1935, composed there from
 * ATen ops). One launch per direction:
 *   expected_depth != 0: colors_out = colors with the last channel divided by max(alpha, 1e-10)   (else colors_out = NULL)
 *   normals_world      = inv(viewmat)[:3,:3] . normals                                            (always)
 *   depth_source 1 | 2 : surf_normals = normalize(cross(P(y+1,x) - P(y-1,x), P(y,x+1) - P(y,x-1))) of the points
 *                        unprojected from the depth map (1: the last colour channel AFTER the normalisation above,
 *                        2: the median depth map), zero on the one-pixel border                   (0: surf_normals = NULL)
 * colors [I,H,W,cdim], alphas / median [I,H,W], normals [I,H,W,3], viewmats [I,4,4] (world-to-camera, last row 0 0 0 1;
 * inverted in the kernel), Ks [I,3,3]. The backward takes the output gradients (v_colors_out NULL iff !expected_depth,
 * v_surf_normals NULL when that output got no gradient) and WRITES v_colors [I,H,W,cdim], v_alphas (may be NULL unless
 * expected_depth), v_normals, v_median (may be NULL). No gradient reaches viewmats / Ks: callers that differentiate the
 * cameras compose the same maths from tensor ops instead.
 * ------------------------------------------------------------------------------------------- */
int gsx_surfel_post_fwd(const float *colors, const float *alphas, const float *normals, const float *median,
                        const float *viewmats, const float *Ks, uint32_t n_images, uint32_t width, uint32_t height,
                        uint32_t cdim, int expected_depth, int depth_source, float *colors_out, float *normals_world,
                        float *surf_normals, void *stream);
int gsx_surfel_post_bwd(const float *colors, const float *alphas, const float *normals, const float *median,
                        const float *viewmats, const float *Ks, uint32_t n_images, uint32_t width, uint32_t height,
                        uint32_t cdim, int expected_depth, int depth_source, const float *v_colors_out,
                        const float *v_normals_world, const float *v_surf_normals, float *v_colors, float *v_alphas,
                        float *v_normals, float *v_median, void *stream);

/* Instrumentation (no reference counterpart): work counters of the compositing pass, for the vector-ALU roofline of
 * bench.py (pairs x (14 + 2 D) flop against the fp32 peak). Replays the forward walk of
 * gsx_raster3d_fwd and ADDS to stats[0..3] (caller zeroes them):
 *   [0] (pixel, Gaussian) pairs a per-pixel serial walk evaluates (the reference kernel's work: every list entry up to
 *       and including the one that saturates the pixel), [1] lane evaluations of this backend (64 x (wave, Gaussian) pairs
 *       surviving the wave-level culling), [2] of those the lanes whose pixel was still open, [3] contributing pairs,
 *   [4] lane evaluations of (wave, Gaussian) pairs in which no lane passes the alpha test. stats has 8 words. */
int gsx_raster3d_pair_stats(const float *means2d, const float *conics, const float *opacities,
                            const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                            uint32_t n_isects, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                            uint32_t tile_h, uint64_t *stats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Query rasterizers (dense tile layout): gsplat::rasterize_num_contributing_gaussians,
 * rasterize_contributing_gaussian_ids, rasterize_top_contributing_gaussian_ids (ext.cpp:1111-1134; kernels
 * RasterizeContributingCommon.cuh:28-198 + the three accumulators). Same walk and thresholds as gsx_raster3d_fwd.
 * n_per_image = N for dense rows [I*N] (returned ids are row % N), 0 for packed rows (ids are the rows).
 *   num_contributing: counts int32 [I,H,W] and alphas float [I,H,W].
 *   contributing_ids: ids int32 / weights float [I,H,W,max_contributing], PRE-FILLED by the caller with -1 / 0; the first
 *     min(count, max_contributing) slots of a pixel receive (id, alpha*T) front to back.
 *   top_contributing: the num_depth_samples strongest contributors by alpha*T (a new sample replaces the weakest kept one
 *     only if strictly stronger; first weakest on ties), re-sorted front to back, padded with (-1, 0).
 * ------------------------------------------------------------------------------------------- */
int gsx_raster3d_num_contributing(const float *means2d, const float *conics, const float *opacities,
                                  const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                  uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                  uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *counts, float *alphas,
                                  void *stream);
int gsx_raster3d_contributing_ids(const float *means2d, const float *conics, const float *opacities,
                                  const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                  uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                  uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t max_contributing,
                                  int32_t *ids, float *weights, void *stream);
int gsx_raster3d_top_contributing(const float *means2d, const float *conics, const float *opacities,
                                  const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                  uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                  uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t num_depth_samples,
                                  int32_t *ids, float *weights, void *stream);

/* Sparse pixel sets: gsplat::rasterize_num_contributing_gaussians_sparse / rasterize_contributing_gaussian_ids_sparse /
 * rasterize_top_contributing_gaussian_ids_sparse (ext.cpp:1115-1140; Rasterization.cpp:1196-1250, 1393-...). Layout as
 * gsx_raster3d_sparse_fwd; outputs are rows [P] / [P, K] in the caller's pixel order. */
int gsx_raster3d_sparse_num_contributing(const float *means2d, const float *conics, const float *opacities,
        const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
        const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map,
        uint32_t n_active, uint32_t words_per_tile, uint32_t n_images, uint32_t n_isects, uint32_t n_per_image,
        uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *counts, float *alphas, void *stream);
int gsx_raster3d_sparse_contributing_ids(const float *means2d, const float *conics, const float *opacities,
        const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
        const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map,
        uint32_t n_active, uint32_t words_per_tile, uint32_t n_images, uint32_t n_isects, uint32_t n_per_image,
        uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t max_contributing, int32_t *ids, float *weights,
        void *stream);
int gsx_raster3d_sparse_top_contributing(const float *means2d, const float *conics, const float *opacities,
        const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
        const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map,
        uint32_t n_active, uint32_t words_per_tile, uint32_t n_images, uint32_t n_isects, uint32_t n_per_image,
        uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t num_depth_samples, int32_t *ids, float *weights,
        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Pack / unpack around the personalised row exchange of the Gaussian-sharded multi-GPU path (seam B; the reference
 * builds its send buffers and splits the received ones with at::cat / index / contiguous:
 * gsplat/cuda/csrc/DistributedCollectives.cpp:368-453). For k < n_groups (<= 8), row < rows, j < widths[k], on 32-bit words:
 *     dst[k][row * dst_strides[k] + j] = src[k][row * src_strides[k] + j]
 * src / dst / strides / widths are HOST tables of n_groups entries; the pointers in src / dst are device pointers.
 * Pack: dst[k] = message + column offset of group k, dst_strides[k] = message row width; unpack: the other way round. */
int gsx_copy_column_groups(uint32_t n_groups, const void *const *src, const uint32_t *src_strides, void *const *dst,
                           const uint32_t *dst_strides, const uint32_t *widths, int64_t rows, void *stream);
/* The same copy with a ROW MAP for ranks that own several cameras: one side holds the rows source rank by source rank, each
 * source's block camera-major ([C_local][N_k] rows for source k: seg_cameras_times_n[k] = C_local * N_k rows, seg_n[k] = N_k),
 * the other ("mapped") side holds them as [C_local][sum N_k]: row (c, n) of source k <-> mapped row c * sum N + offset_k + n.
 * map_dst != 0: the destination is the mapped side (unpack after the exchange), else the source is (pack gradients for the
 * reverse exchange). Replaces the reference's at::cat of per-source pieces + per-field contiguous copies
 * (DistributedCollectives.cpp:420-453) by one pass; segment tables are HOST arrays, n_segments <= 64. */
int gsx_copy_column_groups_mapped(uint32_t n_groups, const void *const *src, const uint32_t *src_strides, void *const *dst,
                                  const uint32_t *dst_strides, const uint32_t *widths, uint32_t n_segments,
                                  const int64_t *seg_cameras_times_n, const int64_t *seg_n, int map_dst, void *stream);

/* Message <-> field arrays, both sides coalesced: `message` is a contiguous [rows][message_stride] array of 32-bit words (what
 * the exchange sends / receives; message_stride <= 16); group k is its columns [columns[k], columns[k] + widths[k]) and, on the
 * other side, the field array fields[k] with row stride field_strides[k] (>= widths[k]: contiguous arrays or column views).
 * to_message != 0: fields -> message (the groups must cover every column), else message -> fields. A workgroup moves 256
 * message rows through LDS. n_segments > 0: the FIELD side is in [C_local][sum N] order (row map as in
 * gsx_copy_column_groups_mapped), the message in exchange order. */
int gsx_copy_message_columns(void *message, uint32_t message_stride, int64_t rows, uint32_t n_groups, const uint32_t *columns,
                             const uint32_t *widths, void *const *fields, const uint32_t *field_strides, int to_message,
                             uint32_t n_segments, const int64_t *seg_cameras_times_n, const int64_t *seg_n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer-side ops of the training step around the rasterizer.
 * gsx_adam: gsplat::adam (ext.cpp:1217; csrc/AdamCUDA.cu:34-75). In-place fused Adam step without bias correction on
 *   the rows g of [n_rows, row_width] tensors with valid[g] != 0 (valid NULL = all rows); masked rows keep parameter
 *   and moments.
 * gsx_relocation: gsplat::relocation (csrc/RelocationCUDA.cu:34-80; gsplat/relocation.py:23-67), MCMC Eq. 9.
 *   ratios int32 [n] in [1, n_max], binoms float [n_max, n_max] (binoms[i][k] = C(i, k)).
 * gsx_mcmc_perturb: gsplat::mcmc_perturb_positions (ext.cpp:1256; csrc/MCMCPerturbCUDA.cu:24-58). In place:
 *   positions += Sigma (noise * sigmoid(-k (sigmoid(opacities_logit) - t)) * noise_scale), Sigma from quats (wxyz,
 *   normalised inside) and LOG scales.
 * ------------------------------------------------------------------------------------------- */
int gsx_adam(float *param, const float *param_grad, float *exp_avg, float *exp_avg_sq, const uint8_t *valid,
             int64_t n_rows, uint32_t row_width, float lr, float b1, float b2, float eps, void *stream);
/* DefaultStrategy's per-step statistics for dense rows (reference gsplat/strategy/default.py:226-285, _update_state - there a
 * torch.where + three boolean-mask gathers with host reads + index_add_): per Gaussian g over the C views it is visible in
 * (radii[(c N + g)] > 0 on both axes): grad2d[g] += |(grad.x half_w, grad.y half_h)|, count[g] += 1,
 * radii_state[g] = max(radii_state[g], max(radii) * inv_max_dim) (radii_state may be NULL). grad element (c, g, k) at
 * grad[(c N + g) * grad_stride + k]: the retained gradient of means2d is a column view of the compositing backward's rows. */
int gsx_strategy_accumulate(const float *grad, uint32_t grad_stride, const int32_t *radii, uint32_t C, uint32_t N,
                            float half_w, float half_h, float inv_max_dim, float *grad2d, float *count, float *radii_state,
                            void *stream);
int gsx_relocation(const float *opacities, const float *scales, const int32_t *ratios, const float *binoms, int64_t n,
                   int n_max, float min_opacity, float *new_opacities, float *new_scales, void *stream);
int gsx_mcmc_perturb(float *positions, const float *quats, const float *scales_log, const float *opacities_logit,
                     const float *noise, int64_t n, float noise_scale, float t, float k, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused SSIM loss of the training step. Replaces what gsplat/losses.py:150-200 (ssim_loss)
 * evaluates - the third-party `fused_ssim` CUDA extension when installed, else torch_ssim_loss: five depthwise 11 x 11
 * convolutions - with one kernel per direction: 11-tap Gaussian window (sigma 1.5), zero padding, C1 = 0.01^2, C2 = 0.03^2.
 * Images are [B, C, H, W] addressed through element strides (b, c, h, w), so channels-last renders are read in place.
 * fwd: partial_sums [gsx_ssim_blocks(B, C, H, W)] = per-workgroup sums of the SSIM map (the caller adds them up and divides
 * by B C H W); dmaps [B, C, H, W, 3] (or NULL) = d ssim / d (mu1, E[x x], E[x y]) for the backward.
 * bwd: v_img1 (strides_v) = g * d sum(ssim map) / d img1 with g = dL / d map (a constant for a mean) = grad_scale, times the
 * device scalar *grad_scale_device when that is given (the incoming gradient of the mean: no host read). */
int64_t gsx_ssim_blocks(uint32_t B, uint32_t C, uint32_t H, uint32_t W);
int gsx_ssim_fwd(const float *img1, const int64_t *strides1, const float *img2, const int64_t *strides2, uint32_t B, uint32_t C,
                 uint32_t H, uint32_t W, float *partial_sums, float *dmaps, void *stream);
int gsx_ssim_bwd(const float *img1, const int64_t *strides1, const float *img2, const int64_t *strides2, uint32_t B, uint32_t C,
                 uint32_t H, uint32_t W, const float *dmaps, float grad_scale, const float *grad_scale_device, float *v_img1,
                 const int64_t *strides_v,
                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_AMD_H_ */
