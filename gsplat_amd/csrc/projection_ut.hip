// Unscented-Transform projection of 3D Gaussians (3DGUT), forward only (the op carries no gradient), gfx950.
// C-ABI entry: gsx_project_ut_fwd (replaces torch op gsplat::projection_ut_3dgs_fused, gsplat/cuda/ext.cpp:1230-1239;
// host fn gsplat/cuda/csrc/Projection.cpp; kernel gsplat/cuda/csrc/ProjectionUT3DGSFused.cu; the reference's torch
// statement of the algorithm: gsplat/cuda/_torch_impl_ut.py:69-644 with the camera models of
// gsplat/cuda/_torch_cameras.py:696-757 (perfect pinhole), :927-1086 (OpenCV pinhole), :793-848 (orthographic)).
//
// One thread per (camera, Gaussian): a streaming kernel, 44 B read + 32 B written per row like the EWA projection.
//   1. seven sigma points: mean, mean +- sqrt(3 + lambda) scale_i R[:, i], lambda = alpha^2 (3 + kappa) - 3;
//   2. each one goes to the camera frame and through the camera model (valid = in front, distortion factor > 0.8,
//      inside the image grown by margin);
//   3. mean2d = sum w_m p_i, cov2d = sum w_c (p_i - mean2d)(p_i - mean2d)^T; with require_all_valid the sums stop at the
//      first invalid point;
//   4. blur + compensation, determinant / diagonal checks, conic = inverse (of cov + 1e-6 I, like the reference),
//      opacity-aware extent, eigenvalue-bounded radii, radius clip, image-bounds cull. Invalid rows are zero.
// Built so far: camera_model 0 (pinhole, with optional radial[6] / tangential[2] / thin-prism[4] coefficients),
// 1 (orthographic), 2 (OpenCV fisheye, _torch_cameras.py:1335-1697: k1..k4 in radial[0..3]; the largest angle the
// model projects is a per-camera quantity computed by the caller) and 3 (f-theta, _torch_cameras.py FThetaCamera:
// pixel distance = polynomial of the ray angle, either polynomial direction as the calibrated one, linear (c, d, e) sensor map,
// principal point shifted by half a pixel; gsx_project_ut_ftheta_fwd), global shutter. Lidar, rolling shutter and the
// windshield model are rejected.
#include "projmath.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

struct ProjUtArgs {
    const float *means, *quats, *scales, *opacities; // [B,N,3] [B,N,4] [B,N,3] [B,N] or null
    const float *viewmats, *Ks;                      // [B,C,4,4] [B,C,3,3]
    const float *radial, *tangential, *thin_prism;   // [B,C,6] [B,C,2] [B,C,4] or null
    const float *max_angle;                          // [B,C]: fisheye only (largest ray angle the model projects)
    uint32_t B, C, N, width, height;
    float eps2d, near_plane, far_plane, radius_clip;
    int camera_model, require_all_valid, distorted;
    // rolling shutter (ProjectionUT3DGSFused.cu:121-127, Cameras.cuh:549-660): pose at the END of the frame, or null = global
    const float *viewmats1; // [B,C,4,4]
    int rs_type;            // 0 top-to-bottom, 1 left-to-right, 2 bottom-to-top, 3 right-to-left, 4 global (Cameras.h:38-45)
    int depth_is_distance;  // global_z_order = false: the sort depth is |mean_c| instead of mean_c.z (ProjectionUT3DGSFused.cu:240)
    int radial_cull;        // ... and f-theta cameras then cull near / far on |mean_c| (:411)
    float w_m0, w_c0, w_i, spread, margin;
    // f-theta (camera_model 3; Cameras.h FThetaCameraDistortionParameters): one record per call
    int ft_reference_poly;        // 1: angle -> pixel distance is the calibrated polynomial; 0: its inverse is (Newton on it)
    float ft_p2a[6], ft_a2p[6];   // pixeldist_to_angle_poly, angle_to_pixeldist_poly (lowest degree first)
    float ft_max_angle, ft_c, ft_d, ft_e;
    // external (windshield) distortion, one record per call (ExternalDistortion.h / .cuh: BivariateWindshieldModel): bivariate
    // polynomials of the ray's two angles, order 5 layout (21 coefficients, lower orders zero-padded by the caller)
    // spinning lidar (camera_model 4; Lidars.cuh, gsplat/cuda/_torch_lidars.py): the "image" is (azimuth, elevation) * 1024
    float ld_h0, ld_hs, ld_v0, ld_vs; // fields of view: start and span, radians
    int ld_ccw;                       // spinning direction: 1 = counter-clockwise
    const int32_t *ld_map;            // angles_to_columns_map [ld_map_h][ld_map_w] (rolling shutter: angle -> column -> time)
    int ld_map_h, ld_map_w, ld_columns;
    int ext;                      // 0: none
    float ext_h[21], ext_v[21];   // forward (distort): horizontal, vertical
    float ext_hi[21], ext_vi[21]; // inverse (undistort)
    int32_t *radii;       // [B,C,N,2]
    float *means2d;       // [B,C,N,2]
    float *depths;        // [B,C,N]
    float *conics;        // [B,C,N,3]
    float *compensations; // [B,C,N] or null
};

struct UtDistortion {
    float k[6], p[2], s[4];
    float max_angle;
};

// torch.remainder on floats (the result takes the sign of the divisor): the lidar's relative azimuth
__device__ __forceinline__ float lidar_mod(float x, float period)
{
    float m = fmodf(x, period);
    if (m != 0.0f && ((period < 0.0f) != (m < 0.0f))) m += period;
    return m;
}

// ---- external distortion: the bivariate windshield model (restated from ExternalDistortion.cuh:64-92, 168-205 and its Python
// statement gsplat/cuda/_torch_external_distortion.py:38-77). A ray is described by the two angles phi = asin(x / |r|),
// theta = asin(y / |r|); two polynomials of (phi, theta) give the angles of the distorted ray, whose z keeps the sign of the
// input's. The polynomial is stored by descending inner order: block k holds the 6 - k coefficients (in x) of y^k.
__device__ __forceinline__ float bivariate_poly21(const float *c, float x, float y)
{
    float outer[6];
    int start = 0;
#pragma unroll
    for (int inner = 5; inner >= 0; --inner) {
        float r = 0.0f;
#pragma unroll
        for (int i = start + inner; i >= start; --i) r = r * x + c[i];
        outer[5 - inner] = r;
        start += inner + 1;
    }
    float r = 0.0f;
#pragma unroll
    for (int i = 5; i >= 0; --i) r = r * y + outer[i];
    return r;
}
__device__ __forceinline__ void windshield_ray(const float *hp, const float *vp, const float *in, float *out)
{
    const float len = sqrtf(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]);
    if (len < 1e-6f) { out[0] = in[0]; out[1] = in[1]; out[2] = in[2]; return; }
    const float phi   = asinf(fminf(1.0f, fmaxf(-1.0f, in[0] / len)));
    const float theta = asinf(fminf(1.0f, fmaxf(-1.0f, in[1] / len)));
    const float x = sinf(bivariate_poly21(hp, phi, theta)), y = sinf(bivariate_poly21(vp, phi, theta));
    out[0] = x; out[1] = y;
    out[2] = sqrtf(1.0f - fminf(x * x + y * y, 1.0f)) * (in[2] < 0.0f ? -1.0f : 1.0f);
}

// camera-frame point -> pixel; returns validity
__device__ __forceinline__ bool ut_project_point_model(const ProjUtArgs &a, const Cam &c, const UtDistortion &d, const float *p,
                                                       float &px, float &py);
// ... behind the windshield: the ray is distorted BEFORE the camera model sees it (BaseCameraModel::camera_ray_to_image_point,
// Cameras.cuh:462-470); the orthographic model, whose input is a point, goes through the ray (x, y, 1) and back (:795-829)
__device__ __forceinline__ bool ut_project_point(const ProjUtArgs &a, const Cam &c, const UtDistortion &d, const float *p,
                                                 float &px, float &py)
{
    if (!a.ext) return ut_project_point_model(a, c, d, p, px, py);
    float q[3];
    if (a.camera_model == 1) {
        if (!(p[2] > 0.0f)) { px = py = 0.0f; return false; }
        const float r[3] = {p[0], p[1], 1.0f};
        float o[3];
        windshield_ray(a.ext_h, a.ext_v, r, o);
        const float x = o[0] / o[2], y = o[1] / o[2];
        if (!(o[2] > 0.0f) || !isfinite(x) || !isfinite(y)) { px = py = 0.0f; return false; }
        q[0] = x; q[1] = y; q[2] = p[2];
    } else {
        windshield_ray(a.ext_h, a.ext_v, p, q);
    }
    return ut_project_point_model(a, c, d, q, px, py);
}
__device__ __forceinline__ bool ut_project_point_model(const ProjUtArgs &a, const Cam &c, const UtDistortion &d, const float *p,
                                                 float &px, float &py)
{
    if (a.camera_model == 4) { // spinning lidar: image point = (azimuth, elevation) in angular pixels; valid inside the fields of view
        const float n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        const float inv = n2 > 0.0f ? 1.0f / sqrtf(n2) : 0.0f;
        const float az = atan2f(p[1] * inv, p[0] * inv), el = asinf(p[2] * inv);
        px = az * 1024.0f;
        py = el * 1024.0f;
        const float rel_az = lidar_mod(a.ld_ccw ? az - a.ld_h0 : a.ld_h0 - az, 6.2831855f), rel_el = a.ld_v0 - el;
        const float m_el = a.margin * a.ld_vs, m_az = a.margin * a.ld_hs;
        return (rel_el <= a.ld_vs + m_el) && (rel_az <= a.ld_hs + m_az) && (rel_el >= -m_el) && (rel_az >= -m_az);
    }
    const bool front = p[2] > 0.0f;
    bool ok          = front;
    float u, v;
    bool zero_behind = true;
    if (a.camera_model == 2) { // OpenCV fisheye: r = theta (1 + k1 theta^2 + ... + k4 theta^8), theta clamped at max_angle
        const float ax = fabsf(p[0]), ay = fabsf(p[1]);
        const float big = fmaxf(ax, ay), small = fminf(ax, ay);
        const float ratio = big > 0.0f ? small / big : 0.0f;
        float rxy = big > 0.0f ? big * sqrtf(1.0f + ratio * ratio) : 0.0f; // overflow-safe hypot, like the reference
        if (!(rxy > 0.0f)) rxy = 1.1920929e-07f;
        const float th_full = atan2f(rxy, p[2]);
        const float th = fminf(th_full, d.max_angle), t2 = th * th;
        const float poly  = th * (1.0f + t2 * (d.k[0] + t2 * (d.k[1] + t2 * (d.k[2] + t2 * d.k[3]))));
        const float delta = poly / rxy;
        px = delta * p[0] * c.fx + c.cx;
        py = delta * p[1] * c.fy + c.cy;
        const float mx = (float)a.width * a.margin, my = (float)a.height * a.margin;
        const bool inb = (px >= -mx) && (px < (float)a.width + mx) && (py >= -my) && (py < (float)a.height + my);
        return front && (delta > 0.0f) && (th_full < d.max_angle) && inb;
    }
    if (a.camera_model == 3) { // f-theta: pixel distance from the principal point = polynomial of the ray angle
        const float ax = fabsf(p[0]), ay = fabsf(p[1]);
        const float big = fmaxf(ax, ay), small = fminf(ax, ay);
        const float ratio = big > 0.0f ? small / big : 0.0f;
        float rxy = big > 0.0f ? big * sqrtf(1.0f + ratio * ratio) : 0.0f; // overflow-safe hypot, like the reference
        if (!(rxy > 0.0f)) rxy = 1.1920929e-07f;
        const float th_full = atan2f(rxy, p[2]);
        const float th = fminf(th_full, a.ft_max_angle);
        auto poly6 = [](const float *k, float x) { return k[0] + x * (k[1] + x * (k[2] + x * (k[3] + x * (k[4] + x * k[5])))); };
        float dist = poly6(a.ft_a2p, th);
        if (a.ft_reference_poly == 0) { // the angle-of-distance polynomial is the calibrated one: invert it (three Newton steps)
            bool done = false;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const float slope = a.ft_p2a[1] + dist * (2.0f * a.ft_p2a[2] + dist * (3.0f * a.ft_p2a[3]
                                    + dist * (4.0f * a.ft_p2a[4] + dist * (5.0f * a.ft_p2a[5]))));
                const float step = (poly6(a.ft_p2a, dist) - th) / slope;
                dist = done ? dist : dist - step;
                done = done || (fabsf(step) < 1e-6f);
            }
        }
        const float ix = dist * p[0] / rxy, iy = dist * p[1] / rxy;
        px = a.ft_c * ix + a.ft_d * iy + (c.cx + 0.5f);
        py = a.ft_e * ix + iy + (c.cy + 0.5f);
        const float mx = (float)a.width * a.margin, my = (float)a.height * a.margin;
        const bool inb = (px >= -mx) && (px < (float)a.width + mx) && (py >= -my) && (py < (float)a.height + my);
        return (th_full < a.ft_max_angle) && inb; // no "in front" test for this model
    }
    if (a.camera_model == 1) { // orthographic
        u = p[0];
        v = p[1];
    } else {
        u = p[0] / p[2];
        v = p[1] / p[2];
        if (a.distorted) {
            const float uu = u * u, vv = v * v, r2 = uu + vv;
            const float a1 = 2.0f * u * v, a2 = r2 + 2.0f * uu, a3 = r2 + 2.0f * vv;
            const float icd = (1.0f + r2 * (d.k[0] + r2 * (d.k[1] + r2 * d.k[2])))
                              / (1.0f + r2 * (d.k[3] + r2 * (d.k[4] + r2 * d.k[5])));
            const float du = d.p[0] * a1 + d.p[1] * a2 + r2 * (d.s[0] + r2 * d.s[1]);
            const float dv = d.p[0] * a3 + d.p[1] * a1 + r2 * (d.s[2] + r2 * d.s[3]);
            u  = icd * u + du;
            v  = icd * v + dv;
            ok = ok && (icd > 0.8f);
            zero_behind = false; // the distorted model keeps the coordinates of points behind the camera
        }
    }
    px = u * c.fx + c.cx;
    py = v * c.fy + c.cy;
    if (zero_behind && !front) px = py = 0.0f;
    const float mx = (float)a.width * a.margin, my = (float)a.height * a.margin;
    const bool inb = (px >= -mx) && (px < (float)a.width + mx) && (py >= -my) && (py < (float)a.height + my);
    return ok && inb;
}

// ---- rolling shutter: the camera pose is interpolated between the start and the end of the frame at the time the sensor reads
// the pixel a point lands on (restated from Cameras.cuh:76-135, 362-429, 503-660 and gsplat/cuda/_math.py:381-458, 513-560,
// 587-645) ----------------------------------------------------------------------------------------------------------------------
struct UtPose {
    float q[4]; // w, x, y, z
    float t[3];
};
// glm::quat_cast of a row-major rotation matrix
__device__ __forceinline__ void ut_rotmat_to_quat(const float *R, float *q)
{
    const float fx = R[0] - R[4] - R[8], fy = R[4] - R[0] - R[8], fz = R[8] - R[0] - R[4], fw = R[0] + R[4] + R[8];
    int big = 0;
    float best = fw;
    if (fx > best) { best = fx; big = 1; }
    if (fy > best) { best = fy; big = 2; }
    if (fz > best) { best = fz; big = 3; }
    const float val = sqrtf(best + 1.0f) * 0.5f, mult = 0.25f / val;
    if (big == 0) { q[0] = val; q[1] = (R[7] - R[5]) * mult; q[2] = (R[2] - R[6]) * mult; q[3] = (R[3] - R[1]) * mult; }
    else if (big == 1) { q[0] = (R[7] - R[5]) * mult; q[1] = val; q[2] = (R[3] + R[1]) * mult; q[3] = (R[2] + R[6]) * mult; }
    else if (big == 2) { q[0] = (R[2] - R[6]) * mult; q[1] = (R[3] + R[1]) * mult; q[2] = val; q[3] = (R[7] + R[5]) * mult; }
    else { q[0] = (R[3] - R[1]) * mult; q[1] = (R[2] + R[6]) * mult; q[2] = (R[7] + R[5]) * mult; q[3] = val; }
}
// glm::rotate(q, v) = v + 2 (w (u x v) + u x (u x v)), u = q.xyz
__device__ __forceinline__ void ut_quat_rotate(const float *q, const float *v, float *o)
{
    const float ux = q[2] * v[2] - q[3] * v[1], uy = q[3] * v[0] - q[1] * v[2], uz = q[1] * v[1] - q[2] * v[0];
    const float wx = q[2] * uz - q[3] * uy, wy = q[3] * ux - q[1] * uz, wz = q[1] * uy - q[2] * ux;
    o[0] = v[0] + 2.0f * (q[0] * ux + wx);
    o[1] = v[1] + 2.0f * (q[0] * uy + wy);
    o[2] = v[2] + 2.0f * (q[0] * uz + wz);
}
// pose at relative frame time tm: translation lerp, rotation glm::slerp (short arc, component-wise mix when nearly parallel) +
// normalisation
__device__ __forceinline__ UtPose ut_interpolate_pose(const UtPose &p0, const UtPose &p1, float tm)
{
    UtPose r;
#pragma unroll
    for (int i = 0; i < 3; ++i) r.t[i] = (1.0f - tm) * p0.t[i] + tm * p1.t[i];
    float cos_theta = p0.q[0] * p1.q[0] + p0.q[1] * p1.q[1] + p0.q[2] * p1.q[2] + p0.q[3] * p1.q[3];
    const float sgn = cos_theta < 0.0f ? -1.0f : 1.0f;
    cos_theta *= sgn;
    float wa, wb;
    if (cos_theta > 1.0f - 1.1920929e-07f) {
        wa = 1.0f - tm; wb = tm;
    } else {
        const float angle = acosf(fminf(cos_theta, 1.0f)), inv = 1.0f / sinf(angle);
        wa = sinf((1.0f - tm) * angle) * inv; wb = sinf(tm * angle) * inv;
    }
    float n2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.q[i] = wa * p0.q[i] + wb * sgn * p1.q[i]; n2 += r.q[i] * r.q[i]; }
    const float inv_n = n2 > 0.0f ? 1.0f / sqrtf(n2) : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.q[i] *= inv_n;
    return r;
}
__device__ __forceinline__ float ut_relative_frame_time(const ProjUtArgs &a, float px, float py)
{
    if (a.camera_model == 4) { // lidar: the column that fires at this angle (looked up in the angle map) / (columns - 1)
        if (a.rs_type == 4 || !a.ld_map) return 0.0f;
        const float kToAngle = 1.0f / 1024.0f;
        const float az = px * kToAngle, el = py * kToAngle;
        const float rel_az = lidar_mod(a.ld_ccw ? az - a.ld_h0 : a.ld_h0 - az, 6.2831855f), rel_el = a.ld_v0 - el;
        const float res_h = a.ld_hs / (float)(a.ld_map_w - 1), res_v = a.ld_vs / (float)(a.ld_map_h - 1);
        const int iv = (int)fminf(fmaxf(rel_el / res_v + 0.5f, 0.0f), (float)(a.ld_map_h - 1));
        const int ih = (int)fminf(fmaxf(rel_az / res_h + 0.5f, 0.0f), (float)(a.ld_map_w - 1));
        return (float)a.ld_map[(size_t)iv * a.ld_map_w + ih] / (float)(a.ld_columns - 1);
    }
    const float W = (float)a.width, H = (float)a.height;
    switch (a.rs_type) {
    case 0: return a.height > 1 ? floorf(py) / (H - 1.0f) : 0.5f;
    case 1: return a.width > 1 ? floorf(px) / (W - 1.0f) : 0.5f;
    case 2: return a.height > 1 ? (H - ceilf(py)) / (H - 1.0f) : 0.5f;
    case 3: return a.width > 1 ? (W - ceilf(px)) / (W - 1.0f) : 0.5f;
    default: return 0.0f;
    }
}
// world point -> pixel under a rolling shutter: project with the start pose (else the end pose), then up to ten rounds of
// "frame time of the pixel -> pose at that time -> project again" (fixed point of the read-out time)
__device__ __forceinline__ bool ut_project_point_rs(const ProjUtArgs &a, const Cam &c, const UtDistortion &d, const UtPose &p0,
                                                    const UtPose &p1, const float *wp, float &px, float &py)
{
    float cp[3], sx, sy, ex, ey;
    ut_quat_rotate(p0.q, wp, cp);
#pragma unroll
    for (int i = 0; i < 3; ++i) cp[i] += p0.t[i];
    const bool ok_s = ut_project_point(a, c, d, cp, sx, sy);
    ut_quat_rotate(p1.q, wp, cp);
#pragma unroll
    for (int i = 0; i < 3; ++i) cp[i] += p1.t[i];
    const bool ok_e = ut_project_point(a, c, d, cp, ex, ey);
    if (!ok_s && !ok_e) { // no valid projection at either end of the frame: the end-of-frame result, invalid
        px = ex; py = ey;
        return false;
    }
    px = ok_s ? sx : ex;
    py = ok_s ? sy : ey;
    bool valid = true;
    float prev_time = -3.402823466e+38f;
    for (int it = 0; it < 10; ++it) {
        const float tm = ut_relative_frame_time(a, px, py);
        if (tm == prev_time) break; // same time = same pose = same pixel from here on
        prev_time = tm;
        const UtPose pr = ut_interpolate_pose(p0, p1, tm);
        ut_quat_rotate(pr.q, wp, cp);
#pragma unroll
        for (int i = 0; i < 3; ++i) cp[i] += pr.t[i];
        valid = ut_project_point(a, c, d, cp, px, py);
    }
    return valid;
}

__global__ void __launch_bounds__(256) project_ut_kernel(const ProjUtArgs a)
{
    const int64_t rows = (int64_t)a.B * a.C * a.N;
    const int64_t row  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const uint32_t g = (uint32_t)(row % a.N), bc = (uint32_t)(row / a.N), b = bc / a.C;
    const size_t gi  = (size_t)b * a.N + g;

    auto write_invalid = [&]() {
        a.radii[2 * row] = a.radii[2 * row + 1] = 0;
        a.means2d[2 * row] = a.means2d[2 * row + 1] = 0.0f;
        a.depths[row] = 0.0f;
        a.conics[3 * row] = a.conics[3 * row + 1] = a.conics[3 * row + 2] = 0.0f;
        if (a.compensations) a.compensations[row] = 0.0f;
    };

    const Cam c = load_cam(a.viewmats + (size_t)bc * 16, a.Ks + (size_t)bc * 9);
    UtDistortion d{};
    d.max_angle = a.max_angle ? a.max_angle[bc] : 0.0f;
    if (a.distorted) {
#pragma unroll
        for (int i = 0; i < 6; ++i) d.k[i] = a.radial ? a.radial[(size_t)bc * 6 + i] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) d.p[i] = a.tangential ? a.tangential[(size_t)bc * 2 + i] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) d.s[i] = a.thin_prism ? a.thin_prism[(size_t)bc * 4 + i] : 0.0f;
    }

    const float *m = a.means + gi * 3, *q = a.quats + gi * 4, *s = a.scales + gi * 3;
    const float eps = 1.1920929e-07f; // float32 machine epsilon (torch.finfo(float32).eps)
    const float qn2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const bool alive = (qn2 > eps) && (s[0] > eps) && (s[1] > eps) && (s[2] > eps);
    if (!alive) {
        write_invalid();
        return;
    }
    float qn[4], Rq[9];
    quat_normalize(q, qn);
    quat_to_rotmat(qn, Rq); // columns = principal axes

    // sigma points in the camera frame: centre + the three scaled axes (camera-frame axis = R_cam * axis)
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = c.R[3 * i] * m[0] + c.R[3 * i + 1] * m[1] + c.R[3 * i + 2] * m[2] + c.t[i];

    float px[7], py[7];
    bool okp[7];
    const bool rolling = a.viewmats1 != nullptr && a.rs_type != 4;
    if (rolling) {
        // the Gaussian's own camera-frame position (culling, depth) uses the pose at the MIDDLE of the frame; every sigma
        // point is projected with the pose of the moment its pixel is read
        UtPose p0, p1;
        ut_rotmat_to_quat(c.R, p0.q);
#pragma unroll
        for (int i = 0; i < 3; ++i) p0.t[i] = c.t[i];
        const float *v1 = a.viewmats1 + (size_t)bc * 16;
        const float R1[9] = {v1[0], v1[1], v1[2], v1[4], v1[5], v1[6], v1[8], v1[9], v1[10]};
        ut_rotmat_to_quat(R1, p1.q);
        p1.t[0] = v1[3]; p1.t[1] = v1[7]; p1.t[2] = v1[11];
        const UtPose mid = ut_interpolate_pose(p0, p1, 0.5f);
        ut_quat_rotate(mid.q, m, pc);
#pragma unroll
        for (int i = 0; i < 3; ++i) pc[i] += mid.t[i];
        okp[0] = ut_project_point_rs(a, c, d, p0, p1, m, px[0], py[0]);
        for (int k = 0; k < 3; ++k) {
            const float ox = a.spread * s[k] * Rq[k], oy = a.spread * s[k] * Rq[3 + k], oz = a.spread * s[k] * Rq[6 + k];
            const float wpp[3] = {m[0] + ox, m[1] + oy, m[2] + oz}, wpm[3] = {m[0] - ox, m[1] - oy, m[2] - oz};
            okp[1 + k] = ut_project_point_rs(a, c, d, p0, p1, wpp, px[1 + k], py[1 + k]);
            okp[4 + k] = ut_project_point_rs(a, c, d, p0, p1, wpm, px[4 + k], py[4 + k]);
        }
    } else {
    okp[0] = ut_project_point(a, c, d, pc, px[0], py[0]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // world-space sigma points are mean +- offset_k; they are moved to the camera frame as points (not as
        // centre + R offset) so that the rounding matches the reference's statement point by point
        const float ox = a.spread * s[k] * Rq[k], oy = a.spread * s[k] * Rq[3 + k], oz = a.spread * s[k] * Rq[6 + k];
        float wp[3], cp[3];
        wp[0] = m[0] + ox; wp[1] = m[1] + oy; wp[2] = m[2] + oz;
#pragma unroll
        for (int i = 0; i < 3; ++i) cp[i] = c.R[3 * i] * wp[0] + c.R[3 * i + 1] * wp[1] + c.R[3 * i + 2] * wp[2] + c.t[i];
        okp[1 + k] = ut_project_point(a, c, d, cp, px[1 + k], py[1 + k]);
        wp[0] = m[0] - ox; wp[1] = m[1] - oy; wp[2] = m[2] - oz;
#pragma unroll
        for (int i = 0; i < 3; ++i) cp[i] = c.R[3 * i] * wp[0] + c.R[3 * i + 1] * wp[1] + c.R[3 * i + 2] * wp[2] + c.t[i];
        okp[4 + k] = ut_project_point(a, c, d, cp, px[4 + k], py[4 + k]);
    }
    }
    const float z = pc[2], dist = sqrtf(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    const float cull_depth = a.radial_cull ? dist : z;

    // UT weights; with require_all_valid the sums stop at the first invalid point (weights of the rest are zero)
    float wm[7], wc[7];
    bool valid;
    {
        bool run = true, any = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            any = any || okp[i];
            run = run && okp[i];
            const bool use = a.require_all_valid ? run : true;
            wm[i] = use ? (i == 0 ? a.w_m0 : a.w_i) : 0.0f;
            wc[i] = use ? (i == 0 ? a.w_c0 : a.w_i) : 0.0f;
        }
        valid = a.require_all_valid ? run : any;
    }
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        mx += wm[i] * px[i];
        my += wm[i] * py[i];
    }
    float cxx = 0.0f, cxy = 0.0f, cyy = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float dx = px[i] - mx, dy = py[i] - my;
        cxx += wc[i] * dx * dx;
        cxy += wc[i] * dx * dy;
        cyy += wc[i] * dy * dy;
    }
    valid = valid && (cull_depth >= a.near_plane) && (cull_depth <= a.far_plane);

    const float det0 = cxx * cyy - cxy * cxy;
    cxx += a.eps2d;
    cyy += a.eps2d;
    const float det  = cxx * cyy - cxy * cxy;
    const float comp = sqrtf(fmaxf(det0 / det, kMinCompensation * kMinCompensation));
    valid = valid && (det > 0.0f) && (cxx > 0.0f) && (cyy > 0.0f);

    float extend = kGaussianExtend;
    if (a.opacities) {
        const float op = a.opacities[gi] * comp;
        valid = valid && (op >= kAlphaThreshold);
        extend = fminf(extend, sqrtf(2.0f * logf(fmaxf(op / kAlphaThreshold, 1.0f))));
    }
    const float hb = 0.5f * (cxx + cyy);
    const float lam_max = hb + sqrtf(fmaxf(hb * hb - det, 0.01f));
    const float r_eig = extend * sqrtf(fmaxf(lam_max, 0.0f));
    const float rx = ceilf(fminf(extend * sqrtf(fmaxf(cxx, 0.0f)), r_eig));
    const float ry = ceilf(fminf(extend * sqrtf(fmaxf(cyy, 0.0f)), r_eig));
    valid = valid && (fmaxf(rx, ry) > a.radius_clip);
    if (a.camera_model != 4) // (a lidar's sigma points were tested against its fields of view)
        valid = valid && (mx + rx > 0.0f) && (mx - rx < (float)a.width) && (my + ry > 0.0f) && (my - ry < (float)a.height);
    if (!valid) {
        write_invalid();
        return;
    }
    const float ixx = cxx + 1e-6f, iyy = cyy + 1e-6f;
    const float idet = ixx * iyy - cxy * cxy;
    a.radii[2 * row]       = (int32_t)rx;
    a.radii[2 * row + 1]   = (int32_t)ry;
    a.means2d[2 * row]     = mx;
    a.means2d[2 * row + 1] = my;
    a.depths[row]          = a.depth_is_distance ? dist : z;
    a.conics[3 * row]      = iyy / idet;
    a.conics[3 * row + 1]  = -cxy / idet;
    a.conics[3 * row + 2]  = ixx / idet;
    if (a.compensations) a.compensations[row] = comp;
}

// ---- pixel -> world ray (the from-world rasterizer's ray generation when no `rays` are given) ---------------------------------
// One thread per pixel: the camera model's INVERSE at the pixel centre (x + 0.5, y + 0.5), then the shutter pose of that pixel
// (relative frame time of its row / column -> translation lerp + rotation slerp; a global shutter is time 0 of equal poses) takes
// the camera ray to the world: origin = q^-1 (-t) (orthographic: q^-1 (uv, 0) - t), direction = q^-1 d. An image point the
// model cannot invert yields the ZERO ray, which the compositing kernels treat as "no samples" (the reference marks the pixel
// done: RasterizeToPixelsFromWorld3DGS.cuh:345-347, 512-526). Restated from Cameras.cuh:717 (perfect pinhole), :866 (orthographic),
// :1062-1290 (OpenCV pinhole: five Newton steps on the 2 x 2 system of the distortion), :1472-1520 (OpenCV fisheye: Newton on the
// odd polynomial from a linear first guess), :1672-1760 (f-theta: the backward polynomial, or Newton on the forward one) and
// their torch statements in gsplat/cuda/_torch_cameras.py.
struct RayGenArgs {
    ProjUtArgs cam;  // viewmats (start of frame), viewmats1 (end, or null), Ks, coefficients, width, height, camera_model, rs_type, ft_*
    uint32_t n_images;
    float *rays;     // [I,H,W,6]
};

__device__ __forceinline__ bool pixel_to_camera_ray(const ProjUtArgs &a, const Cam &c, const UtDistortion &d, float ipx, float ipy,
                                                    float *ray /*[3] unit direction*/, float *org /*[3] camera-frame origin*/)
{
    org[0] = org[1] = org[2] = 0.0f;
    auto unit = [&](float x, float y, float z) { // _safe_normalize: the zero vector stays zero
        const float n2 = x * x + y * y + z * z;
        const float inv = n2 > 0.0f ? 1.0f / sqrtf(n2) : 0.0f;
        ray[0] = x * inv; ray[1] = y * inv; ray[2] = z * inv;
    };
    if (a.camera_model == 3) { // f-theta: undo the affine sensor map, then angle = backward polynomial of the pixel distance
        const float qx = ipx - (c.cx + 0.5f), qy = ipy - (c.cy + 0.5f);
        const float det = a.ft_c - a.ft_e * a.ft_d;
        if (fabsf(det) < 1e-8f) { ray[0] = ray[1] = 0.0f; ray[2] = 1.0f; return false; }
        const float u = (qx - a.ft_d * qy) / det, v = (-a.ft_e * qx + a.ft_c * qy) / det;
        const float delta = sqrtf(u * u + v * v);
        auto poly6 = [](const float *k, float x) { return k[0] + x * (k[1] + x * (k[2] + x * (k[3] + x * (k[4] + x * k[5])))); };
        float theta = poly6(a.ft_p2a, delta);
        bool converged = true;
        if (a.ft_reference_poly != 0) { // the forward polynomial is the calibrated one: three Newton steps from the backward fit
            converged = false;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const float slope = a.ft_a2p[1] + theta * (2.0f * a.ft_a2p[2] + theta * (3.0f * a.ft_a2p[3]
                                    + theta * (4.0f * a.ft_a2p[4] + theta * (5.0f * a.ft_a2p[5]))));
                const float step = (poly6(a.ft_a2p, theta) - delta) / slope;
                theta     = converged ? theta : theta - step;
                converged = converged || (fabsf(step) < 1e-6f);
            }
        }
        if (!converged) { ray[0] = ray[1] = 0.0f; ray[2] = 1.0f; return false; }
        if (delta >= 1e-6f) {
            const float sc = sinf(theta) / delta;
            unit(sc * u, sc * v, cosf(theta));
        } else {
            ray[0] = ray[1] = 0.0f; ray[2] = 1.0f;
        }
        return true;
    }
    const float u0 = (ipx - c.cx) / c.fx, v0 = (ipy - c.cy) / c.fy;
    if (a.camera_model == 1) { // orthographic: parallel rays, the pixel moves the ORIGIN
        org[0] = u0; org[1] = v0;
        ray[0] = ray[1] = 0.0f; ray[2] = 1.0f;
        return true;
    }
    if (a.camera_model == 2) { // OpenCV fisheye: solve theta (1 + k1 theta^2 + .. + k4 theta^8) = |uv| by Newton
        const float delta = sqrtf(u0 * u0 + v0 * v0);
        const float max_norm = fmaxf((float)a.width * 0.5f / c.fx, (float)a.height * 0.5f / c.fy);
        float theta = (d.max_angle / max_norm) * delta; // linear first guess (approx_backward_poly)
        bool converged = false;
        for (int it = 0; it < 20; ++it) {
            const float t2 = theta * theta;
            const float f  = theta * (1.0f + t2 * (d.k[0] + t2 * (d.k[1] + t2 * (d.k[2] + t2 * d.k[3]))));
            const float df = 1.0f + t2 * (3.0f * d.k[0] + t2 * (5.0f * d.k[1] + t2 * (7.0f * d.k[2] + t2 * 9.0f * d.k[3])));
            const float step = (f - delta) / df;
            theta     = converged ? theta : theta - step;
            converged = converged || (fabsf(step) < 1e-6f);
            if (converged) break;
        }
        if (theta < 0.0f || !(theta < d.max_angle) || !converged) { ray[0] = ray[1] = 0.0f; ray[2] = 1.0f; return false; }
        if (delta >= 1e-6f) {
            const float sc = sinf(theta) / delta;
            ray[0] = sc * u0; ray[1] = sc * v0; ray[2] = cosf(theta);
        } else {
            ray[0] = ray[1] = 0.0f; ray[2] = 1.0f;
        }
        return true;
    }
    if (!a.distorted) { // perfect pinhole
        unit(u0, v0, 1.0f);
        return true;
    }
    // OpenCV pinhole: find (x, y) whose distorted image is (u0, v0) - Newton on the residual of the distortion model
    float x = u0, y = v0;
    bool converged = false, ok = true;
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
        const float r = x * x + y * y, r2 = r * r;
        const float alpha = 1.0f + r * (d.k[0] + r * (d.k[1] + r * d.k[2]));
        const float beta  = 1.0f + r * (d.k[3] + r * (d.k[4] + r * d.k[5]));
        const float dd = alpha / beta;
        const bool vj  = dd > 0.0f;
        float fx = dd * x + 2.0f * d.p[0] * x * y + d.p[1] * (r + 2.0f * x * x) + d.s[0] * r + d.s[1] * r2 - u0;
        float fy = dd * y + 2.0f * d.p[1] * x * y + d.p[0] * (r + 2.0f * y * y) + d.s[2] * r + d.s[3] * r2 - v0;
        const float alpha_r = d.k[0] + r * (2.0f * d.k[1] + r * (3.0f * d.k[2]));
        const float beta_r  = d.k[3] + r * (2.0f * d.k[4] + r * (3.0f * d.k[5]));
        const float d_r = (alpha_r * beta - alpha * beta_r) / (beta * beta);
        const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
        float fx_x = dd + d_x * x + 2.0f * d.p[0] * y + 6.0f * d.p[1] * x + 2.0f * x * (d.s[0] + 2.0f * d.s[1] * r);
        float fx_y = d_y * x + 2.0f * d.p[0] * x + 2.0f * d.p[1] * y + 2.0f * y * (d.s[0] + 2.0f * d.s[1] * r);
        float fy_x = d_x * y + 2.0f * d.p[1] * y + 2.0f * d.p[0] * x + 2.0f * x * (d.s[2] + 2.0f * d.s[3] * r);
        float fy_y = dd + d_y * y + 2.0f * d.p[1] * x + 6.0f * d.p[0] * y + 2.0f * y * (d.s[2] + 2.0f * d.s[3] * r);
        if (!vj) { fx = fy = fx_x = fx_y = fy_x = fy_y = 0.0f; }
        ok = ok && vj;
        const float det = fx_x * fy_y - fx_y * fy_x;
        ok = ok && (fabsf(det) >= 1e-6f);
        const float dx = -(fx * fy_y - fy * fx_y) / det, dy = -(fy * fx_x - fx * fy_x) / det;
        if (!(converged || !ok)) { x += dx; y += dy; }
        converged = converged || (ok && fabsf(dx) < 1e-6f && fabsf(dy) < 1e-6f);
    }
    unit(x, y, 1.0f);
    return converged;
}

__global__ void __launch_bounds__(256) camera_rays_kernel(const RayGenArgs g)
{
    const ProjUtArgs &a = g.cam;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)a.width * a.height;
    if (idx >= per * g.n_images) return;
    const uint32_t img = (uint32_t)(idx / per);
    const uint32_t rem = (uint32_t)(idx % per), py = rem / a.width, px = rem % a.width;
    const Cam c = load_cam(a.viewmats + (size_t)img * 16, a.Ks + (size_t)img * 9);
    UtDistortion d{};
    if (a.radial)
        for (int i = 0; i < 6; ++i) d.k[i] = a.radial[(size_t)img * 6 + i];
    if (a.tangential) { d.p[0] = a.tangential[(size_t)img * 2]; d.p[1] = a.tangential[(size_t)img * 2 + 1]; }
    if (a.thin_prism)
        for (int i = 0; i < 4; ++i) d.s[i] = a.thin_prism[(size_t)img * 4 + i];
    d.max_angle = a.max_angle ? a.max_angle[img] : 0.0f;
    const float ipx = (float)px + 0.5f, ipy = (float)py + 0.5f;
    float ray[3], org[3];
    bool valid = pixel_to_camera_ray(a, c, d, ipx, ipy, ray, org);
    if (a.ext && valid) { // the camera model's ray is the one BEHIND the windshield: undo it (Cameras.cuh:473-484; ortho :862-886)
        if (a.camera_model == 1) {
            const float r[3] = {org[0], org[1], 1.0f};
            float o[3];
            windshield_ray(a.ext_hi, a.ext_vi, r, o);
            const float x = o[0] / o[2], y = o[1] / o[2];
            valid  = (o[2] > 0.0f) && isfinite(x) && isfinite(y);
            org[0] = x; org[1] = y;
        } else {
            float o[3];
            windshield_ray(a.ext_hi, a.ext_vi, ray, o);
            ray[0] = o[0]; ray[1] = o[1]; ray[2] = o[2];
        }
    }
    // the pose of this pixel
    UtPose p0, p1;
    ut_rotmat_to_quat(c.R, p0.q);
    p0.t[0] = c.t[0]; p0.t[1] = c.t[1]; p0.t[2] = c.t[2];
    p1 = p0;
    if (a.viewmats1) {
        const Cam c1 = load_cam(a.viewmats1 + (size_t)img * 16, a.Ks + (size_t)img * 9);
        ut_rotmat_to_quat(c1.R, p1.q);
        p1.t[0] = c1.t[0]; p1.t[1] = c1.t[1]; p1.t[2] = c1.t[2];
    }
    const UtPose pose = ut_interpolate_pose(p0, p1, ut_relative_frame_time(a, ipx, ipy));
    const float qi[4] = {pose.q[0], -pose.q[1], -pose.q[2], -pose.q[3]};
    const float rel[3] = {org[0] - pose.t[0], org[1] - pose.t[1], org[2] - pose.t[2]};
    float o[3], dw[3];
    ut_quat_rotate(qi, rel, o);
    ut_quat_rotate(qi, ray, dw);
    float *out = g.rays + (size_t)idx * 6;
    const float keep = valid ? 1.0f : 0.0f; // invalid rays are all zero
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[i]     = o[i] * keep;
        out[3 + i] = dw[i] * keep;
    }
}

} // namespace gsx

// `ext`: HOST array of 84 floats = horizontal | vertical | horizontal inverse | vertical inverse polynomial, each in the order-5
// layout (21 coefficients); NULL = no external distortion
static void set_external_distortion(gsx::ProjUtArgs &a, const float *ext)
{
    a.ext = ext ? 1 : 0;
    if (!ext) return;
    for (int i = 0; i < 21; ++i) {
        a.ext_h[i] = ext[i]; a.ext_v[i] = ext[21 + i]; a.ext_hi[i] = ext[42 + i]; a.ext_vi[i] = ext[63 + i];
    }
}

struct LidarHost { // the lidar record of one call (fields of view in radians; the angle -> column map lives on the device)
    double h0, hs, v0, vs;
    int ccw;
    const int32_t *map;
    int map_h, map_w, n_columns;
};
static void set_lidar(gsx::ProjUtArgs &a, const LidarHost *l)
{
    if (!l) return;
    a.ld_h0 = (float)l->h0; a.ld_hs = (float)l->hs; a.ld_v0 = (float)l->v0; a.ld_vs = (float)l->vs;
    a.ld_ccw = l->ccw ? 1 : 0; a.ld_map = l->map; a.ld_map_h = l->map_h; a.ld_map_w = l->map_w; a.ld_columns = l->n_columns;
}

static int project_ut_launch(const float *means, const float *quats, const float *scales, const float *opacities,
                             const float *viewmats, const float *Ks, const float *radial, const float *tangential,
                             const float *thin_prism, const float *fisheye_max_angle, const float *ftheta, uint32_t B, uint32_t C,
                             uint32_t N, uint32_t width, uint32_t height, float eps2d, float near_plane, float far_plane,
                             float radius_clip, int camera_model, float ut_alpha, float ut_beta, float ut_kappa,
                             float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii, float *means2d,
                             float *depths, float *conics, float *compensations, void *stream, const float *viewmats1 = nullptr,
                             int rs_type = 4, int global_z_order = 1, const float *ext = nullptr, const LidarHost *lidar = nullptr)
{
    using namespace gsx;
    GSX_REQUIRE(rs_type >= 0 && rs_type <= 4, "gsx_project_ut_rs_fwd: rolling shutter type %d (0 .. 3 rolling, 4 global)", rs_type);
    GSX_REQUIRE(rs_type == 4 || viewmats1 != nullptr, "gsx_project_ut_rs_fwd: a rolling shutter needs the end-of-frame poses");
    const int64_t rows = (int64_t)B * C * N;
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means && quats && scales && viewmats && Ks, "gsx_project_ut_fwd: null input");
    GSX_REQUIRE(radii && means2d && depths && conics, "gsx_project_ut_fwd: null output");
    GSX_REQUIRE(camera_model >= 0 && camera_model <= 4,
                "gsx_project_ut_fwd: camera model %d is not built (pinhole = 0, orthographic = 1, fisheye = 2, f-theta = 3, lidar = 4 are)",
                camera_model);
    GSX_REQUIRE((camera_model == 4) == (lidar != nullptr), "gsx_project_ut_fwd: the lidar model (4) goes through gsx_project_ut_lidar_fwd");
    GSX_REQUIRE(camera_model != 4 || (!radial && !tangential && !thin_prism && !ext), "gsx_project_ut_lidar_fwd: a lidar takes no distortion coefficients");
    GSX_REQUIRE(camera_model != 4 || rs_type == 4 || (lidar->map && lidar->map_h > 1 && lidar->map_w > 1 && lidar->n_columns > 1),
                "gsx_project_ut_lidar_fwd: a rolling shutter needs the angles_to_columns_map");
    GSX_REQUIRE(camera_model != 1 || (!radial && !tangential && !thin_prism),
                "gsx_project_ut_fwd: the orthographic model takes no distortion coefficients");
    GSX_REQUIRE(camera_model != 2 || (fisheye_max_angle && !tangential && !thin_prism),
                "gsx_project_ut_fwd: the fisheye model needs fisheye_max_angle and takes radial coefficients only");
    GSX_REQUIRE(camera_model != 3 || (ftheta && !radial && !tangential && !thin_prism),
                "gsx_project_ut_ftheta_fwd: the f-theta model needs its parameter record and takes no other coefficients");
    const double lam = (double)ut_alpha * ut_alpha * (3.0 + ut_kappa) - 3.0;
    GSX_REQUIRE(3.0 + lam > 0.0, "gsx_project_ut_fwd: alpha^2 (3 + kappa) must be positive");
    ProjUtArgs a{};
    a.means = means; a.quats = quats; a.scales = scales; a.opacities = opacities; a.viewmats = viewmats; a.Ks = Ks;
    a.radial = radial; a.tangential = tangential; a.thin_prism = thin_prism; a.max_angle = fisheye_max_angle;
    a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.require_all_valid = require_all_sigma_points_valid;
    a.distorted = (radial || tangential || thin_prism) ? 1 : 0;
    if (camera_model == 3) {
        a.ft_reference_poly = ftheta[0] != 0.0f ? 1 : 0;
        for (int i = 0; i < 6; ++i) { a.ft_p2a[i] = ftheta[1 + i]; a.ft_a2p[i] = ftheta[7 + i]; }
        a.ft_max_angle = ftheta[13]; a.ft_c = ftheta[14]; a.ft_d = ftheta[15]; a.ft_e = ftheta[16];
    }
    a.w_m0 = (float)(lam / (3.0 + lam));
    a.w_c0 = (float)(lam / (3.0 + lam) + (1.0 - (double)ut_alpha * ut_alpha + ut_beta));
    a.w_i  = (float)(1.0 / (2.0 * (3.0 + lam)));
    a.spread = (float)sqrt(3.0 + lam);
    a.margin = in_image_margin_factor;
    a.viewmats1 = rs_type == 4 ? nullptr : viewmats1; a.rs_type = rs_type;
    set_external_distortion(a, ext);
    set_lidar(a, lidar);
    a.depth_is_distance = global_z_order ? 0 : 1;
    a.radial_cull = (!global_z_order && (camera_model == 3 || camera_model == 4)) ? 1 : 0; // models that accept rays with z <= 0
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.conics = conics; a.compensations = compensations;
    project_ut_kernel<<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ut_fwd");
}

extern "C" int gsx_project_ut_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                                  const float *viewmats, const float *Ks, const float *radial, const float *tangential,
                                  const float *thin_prism, const float *fisheye_max_angle, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                  uint32_t height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                  int camera_model, float ut_alpha, float ut_beta, float ut_kappa,
                                  float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii,
                                  float *means2d, float *depths, float *conics, float *compensations, void *stream)
{
    GSX_REQUIRE(camera_model != 3, "gsx_project_ut_fwd: the f-theta model takes its parameters through gsx_project_ut_ftheta_fwd");
    return project_ut_launch(means, quats, scales, opacities, viewmats, Ks, radial, tangential, thin_prism, fisheye_max_angle, nullptr,
                             B, C, N, width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, ut_alpha, ut_beta,
                             ut_kappa, in_image_margin_factor, require_all_sigma_points_valid, radii, means2d, depths, conics,
                             compensations, stream);
}

// f-theta cameras: `ftheta` is a HOST array of 17 floats - reference_poly (0 / 1), pixeldist_to_angle_poly[6],
// angle_to_pixeldist_poly[6], max_angle, linear_cde[3] (the fields of FThetaCameraDistortionParameters, Cameras.h:103-117)
extern "C" int gsx_project_ut_ftheta_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                                         const float *viewmats, const float *Ks, const float *ftheta, uint32_t B, uint32_t C,
                                         uint32_t N, uint32_t width, uint32_t height, float eps2d, float near_plane, float far_plane,
                                         float radius_clip, float ut_alpha, float ut_beta, float ut_kappa,
                                         float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii,
                                         float *means2d, float *depths, float *conics, float *compensations, void *stream)
{
    GSX_REQUIRE(ftheta != nullptr, "gsx_project_ut_ftheta_fwd: null parameter record");
    return project_ut_launch(means, quats, scales, opacities, viewmats, Ks, nullptr, nullptr, nullptr, nullptr, ftheta, B, C, N,
                             width, height, eps2d, near_plane, far_plane, radius_clip, 3, ut_alpha, ut_beta, ut_kappa,
                             in_image_margin_factor, require_all_sigma_points_valid, radii, means2d, depths, conics, compensations,
                             stream);
}

// Every camera model of the two entries above + rolling shutter + the Euclidean sort depth, one entry: `viewmats1` = pose at the
// end of the frame (NULL with rs_type 4 = global), `rs_type` as Cameras.h:38-45, `global_z_order` 0 = depths are |mean_c| (and
// f-theta cameras cull near / far radially). `ftheta` = the 17-float host record of gsx_project_ut_ftheta_fwd or NULL.
extern "C" int gsx_project_ut_rs_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                                     const float *viewmats0, const float *viewmats1, const float *Ks, const float *radial,
                                     const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                                     const float *ftheta, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                     float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                                     int rs_type, int global_z_order, float ut_alpha, float ut_beta, float ut_kappa,
                                     float in_image_margin_factor, int require_all_sigma_points_valid, int32_t *radii,
                                     float *means2d, float *depths, float *conics, float *compensations, void *stream)
{
    return project_ut_launch(means, quats, scales, opacities, viewmats0, Ks, radial, tangential, thin_prism, fisheye_max_angle, ftheta,
                             B, C, N, width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, ut_alpha, ut_beta,
                             ut_kappa, in_image_margin_factor, require_all_sigma_points_valid, radii, means2d, depths, conics,
                             compensations, stream, viewmats1, rs_type, global_z_order);
}

// Rays of every pixel of I images, [I,H,W,6] = world-space origin | unit direction (the zero ray where the camera model cannot
// invert the pixel): what the reference's from-world rasterizer derives per thread when no `rays` tensor is passed
// (compute_world_ray, RasterizeToPixelsFromWorld3DGS.cuh:349-529; BaseCameraModel::element_to_world_ray_shutter_pose,
// Cameras.cuh:503-546). viewmats_rs NULL / rs_type 4 = global shutter; coefficients as in gsx_project_ut_rs_fwd, one record per
// image (radial [I,6], tangential [I,2], thin_prism [I,4], fisheye_max_angle [I]); `ftheta` the 17-float HOST record.
static int camera_rays_impl(const float *viewmats, const float *viewmats_rs, const float *Ks, const float *radial,
                            const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                            const float *ftheta, const float *ext, uint32_t n_images, uint32_t width, uint32_t height,
                            int camera_model, int rs_type, float *rays, void *stream)
{
    using namespace gsx;
    const int64_t n = (int64_t)n_images * width * height;
    if (n == 0) return GSX_OK;
    GSX_REQUIRE(viewmats && Ks && rays, "gsx_camera_rays: null pointer");
    GSX_REQUIRE(camera_model >= 0 && camera_model <= 3, "gsx_camera_rays: camera model %d is not built (pinhole 0, ortho 1, fisheye 2, f-theta 3)", camera_model);
    GSX_REQUIRE(rs_type >= 0 && rs_type <= 4, "gsx_camera_rays: rolling shutter type %d (0 .. 3 rolling, 4 global)", rs_type);
    GSX_REQUIRE(rs_type == 4 || viewmats_rs != nullptr, "gsx_camera_rays: a rolling shutter needs the end-of-frame poses");
    GSX_REQUIRE(camera_model != 1 || (!radial && !tangential && !thin_prism), "gsx_camera_rays: the orthographic model takes no distortion coefficients");
    GSX_REQUIRE(camera_model != 2 || (fisheye_max_angle && !tangential && !thin_prism), "gsx_camera_rays: the fisheye model needs fisheye_max_angle and takes radial coefficients only");
    GSX_REQUIRE(camera_model != 3 || (ftheta && !radial && !tangential && !thin_prism), "gsx_camera_rays: the f-theta model needs its parameter record and takes no other coefficients");
    RayGenArgs g{};
    ProjUtArgs &a = g.cam;
    a.viewmats = viewmats; a.viewmats1 = rs_type == 4 ? nullptr : viewmats_rs; a.Ks = Ks;
    a.radial = radial; a.tangential = tangential; a.thin_prism = thin_prism; a.max_angle = fisheye_max_angle;
    a.width = width; a.height = height; a.camera_model = camera_model; a.rs_type = rs_type;
    a.distorted = (radial || tangential || thin_prism) ? 1 : 0;
    if (camera_model == 3) {
        a.ft_reference_poly = ftheta[0] != 0.0f ? 1 : 0;
        for (int i = 0; i < 6; ++i) { a.ft_p2a[i] = ftheta[1 + i]; a.ft_a2p[i] = ftheta[7 + i]; }
        a.ft_max_angle = ftheta[13]; a.ft_c = ftheta[14]; a.ft_d = ftheta[15]; a.ft_e = ftheta[16];
    }
    set_external_distortion(a, ext);
    g.n_images = n_images; g.rays = rays;
    camera_rays_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(g);
    return check_launch("camera_rays");
}

extern "C" int gsx_camera_rays(const float *viewmats, const float *viewmats_rs, const float *Ks, const float *radial,
                               const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                               const float *ftheta, uint32_t n_images, uint32_t width, uint32_t height, int camera_model,
                               int rs_type, float *rays, void *stream)
{
    return camera_rays_impl(viewmats, viewmats_rs, Ks, radial, tangential, thin_prism, fisheye_max_angle, ftheta, nullptr, n_images,
                            width, height, camera_model, rs_type, rays, stream);
}
// ... behind a windshield: `ext` = HOST array of 84 floats (horizontal | vertical | horizontal inverse | vertical inverse
// polynomial, order-5 layout of 21 coefficients each: BivariateWindshieldModelParameters, ExternalDistortion.h); the camera
// model's ray is taken through the INVERSE polynomials (BaseCameraModel::image_point_to_camera_ray, Cameras.cuh:473-484)
extern "C" int gsx_camera_rays_ext(const float *viewmats, const float *viewmats_rs, const float *Ks, const float *radial,
                                   const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                                   const float *ftheta, const float *ext, uint32_t n_images, uint32_t width, uint32_t height,
                                   int camera_model, int rs_type, float *rays, void *stream)
{
    return camera_rays_impl(viewmats, viewmats_rs, Ks, radial, tangential, thin_prism, fisheye_max_angle, ftheta, ext, n_images,
                            width, height, camera_model, rs_type, rays, stream);
}
// gsx_project_ut_rs_fwd behind a windshield (`ext` as above): every sigma point's ray is taken through the FORWARD polynomials
// before the camera model projects it (BaseCameraModel::camera_ray_to_image_point, Cameras.cuh:462-470)
extern "C" int gsx_project_ut_ext_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                                      const float *viewmats0, const float *viewmats1, const float *Ks, const float *radial,
                                      const float *tangential, const float *thin_prism, const float *fisheye_max_angle,
                                      const float *ftheta, const float *ext, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                      uint32_t height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                      int camera_model, int rs_type, int global_z_order, float ut_alpha, float ut_beta,
                                      float ut_kappa, float in_image_margin_factor, int require_all_sigma_points_valid,
                                      int32_t *radii, float *means2d, float *depths, float *conics, float *compensations,
                                      void *stream)
{
    return project_ut_launch(means, quats, scales, opacities, viewmats0, Ks, radial, tangential, thin_prism, fisheye_max_angle, ftheta,
                             B, C, N, width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, ut_alpha, ut_beta,
                             ut_kappa, in_image_margin_factor, require_all_sigma_points_valid, radii, means2d, depths, conics,
                             compensations, stream, viewmats1, rs_type, global_z_order, ext);
}

// gsplat::distort_camera_rays / gsplat::eval_bivariate_poly (ExternalDistortionWrappers.cu:30-160): the windshield model on a
// batch of rays [n,3] with ONE pair of polynomials (host arrays of 21 floats, order-5 layout; pass the inverse pair to undistort),
// and one bivariate polynomial at n points.
namespace gsx {
struct Poly21 { float c[21]; };
__global__ void __launch_bounds__(256) distort_rays_kernel(const float *rays, Poly21 h, Poly21 v, float *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float in[3] = {rays[3 * i], rays[3 * i + 1], rays[3 * i + 2]};
    float o[3];
    windshield_ray(h.c, v.c, in, o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
}
__global__ void __launch_bounds__(256) bivariate_poly_kernel(const float *x, const float *y, Poly21 p, float *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bivariate_poly21(p.c, x[i], y[i]);
}
} // namespace gsx
extern "C" int gsx_distort_camera_rays(const float *rays, int64_t n, const float *horizontal_poly, const float *vertical_poly,
                                       float *out, void *stream)
{
    using namespace gsx;
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(rays && horizontal_poly && vertical_poly && out, "gsx_distort_camera_rays: null pointer");
    Poly21 h, v;
    for (int i = 0; i < 21; ++i) { h.c[i] = horizontal_poly[i]; v.c[i] = vertical_poly[i]; }
    distort_rays_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(rays, h, v, out, n);
    return check_launch("distort_camera_rays");
}
extern "C" int gsx_eval_bivariate_poly(const float *x, const float *y, int64_t n, const float *poly, float *out, void *stream)
{
    using namespace gsx;
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(x && y && poly && out, "gsx_eval_bivariate_poly: null pointer");
    Poly21 p;
    for (int i = 0; i < 21; ++i) p.c[i] = poly[i];
    bivariate_poly_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(x, y, p, out, n);
    return check_launch("eval_bivariate_poly");
}

// Spinning-lidar cameras (camera_model 4; Lidars.cuh:40-330, gsplat/cuda/_torch_lidars.py:214-374): a sigma point's image point is
// (azimuth, elevation) * 1024, valid inside the fields of view (+ the UT margin); no image-bounds culling; radial near / far
// culling and depth with global_z_order = 0; a rolling shutter reads the time of an angle off `angles_to_columns_map`
// (int32 [map_h][map_w] on the device; NULL with rs_type 4). Ks is not used by this model (pass any [B,C,3,3]).
extern "C" int gsx_project_ut_lidar_fwd(const float *means, const float *quats, const float *scales, const float *opacities,
                                        const float *viewmats0, const float *viewmats1, const float *Ks, double fov_horiz_start,
                                        double fov_horiz_span, double fov_vert_start, double fov_vert_span, int spinning_ccw,
                                        const int32_t *angles_to_columns_map, uint32_t map_h, uint32_t map_w, uint32_t n_columns,
                                        uint32_t B, uint32_t C, uint32_t N, float eps2d, float near_plane, float far_plane,
                                        float radius_clip, int rs_type, int global_z_order, float ut_alpha, float ut_beta,
                                        float ut_kappa, float in_image_margin_factor, int require_all_sigma_points_valid,
                                        int32_t *radii, float *means2d, float *depths, float *conics, float *compensations,
                                        void *stream)
{
    LidarHost l{fov_horiz_start, fov_horiz_span, fov_vert_start, fov_vert_span, spinning_ccw, angles_to_columns_map,
                (int)map_h, (int)map_w, (int)n_columns};
    return project_ut_launch(means, quats, scales, opacities, viewmats0, Ks, nullptr, nullptr, nullptr, nullptr, nullptr, B, C, N,
                             n_columns, map_h, eps2d, near_plane, far_plane, radius_clip, 4, ut_alpha, ut_beta, ut_kappa,
                             in_image_margin_factor, require_all_sigma_points_valid, radii, means2d, depths, conics, compensations,
                             stream, viewmats1, rs_type, global_z_order, nullptr, &l);
}

// World rays of a spinning lidar's elements, [I, n_rows, n_columns, 6] (origin | unit direction; the zero ray for an element
// outside the fields of view): element (row, column) looks along azimuth = column_azimuths[column] + row_azimuth_offsets[row]
// (wrapped into (-pi, pi]) and elevation = row_elevations[row] (Lidars.cuh element_to_image_point / image_point_to_camera_ray;
// gsplat/cuda/_torch_lidars.py:266-324); its pose is the one at the time its column fires (rolling shutter) or the frame's.
namespace gsx {
struct LidarRayArgs {
    ProjUtArgs cam; // viewmats, viewmats1, rs_type, ld_*
    const float *row_el, *col_az, *row_off;
    uint32_t n_images, n_rows, n_cols;
    float eps;      // fov_eps_rad
    float *rays;
};
__global__ void __launch_bounds__(256) lidar_rays_kernel(const LidarRayArgs g)
{
    const ProjUtArgs &a = g.cam;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)g.n_rows * g.n_cols;
    if (idx >= per * g.n_images) return;
    const uint32_t img = (uint32_t)(idx / per), rem = (uint32_t)(idx % per), row = rem / g.n_cols, col = rem % g.n_cols;
    const float kPi = 3.1415927f;
    const float el = g.row_el[row];
    float az = g.col_az[col] + g.row_off[row];
    az = az > kPi ? az - 2.0f * kPi : az;
    az = az <= -kPi ? az + 2.0f * kPi : az;
    const float ipx = az * 1024.0f, ipy = el * 1024.0f; // the element's image point
    // image point -> camera ray (through the scaled angles, like the reference)
    const float kToAngle = 1.0f / 1024.0f;
    const float aa = ipx * kToAngle, ee = ipy * kToAngle, ce = cosf(ee);
    float ray[3] = {cosf(aa) * ce, sinf(aa) * ce, sinf(ee)};
    const float n2 = ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2], inv = n2 > 0.0f ? 1.0f / sqrtf(n2) : 0.0f;
    ray[0] *= inv; ray[1] *= inv; ray[2] *= inv;
    // valid_sensor_angles: inside the fields of view widened by eps on both sides
    const float v0 = a.ld_v0 + g.eps, h0 = a.ld_ccw ? a.ld_h0 - g.eps : a.ld_h0 + g.eps;
    const float rel_el = v0 - ee, rel_az = lidar_mod(a.ld_ccw ? aa - h0 : h0 - aa, 6.2831855f);
    const bool valid = (rel_el <= a.ld_vs + g.eps * 2.0f) && (rel_az <= a.ld_hs + g.eps * 2.0f);
    const Cam c = load_cam(a.viewmats + (size_t)img * 16, a.viewmats + (size_t)img * 16); // (intrinsics are not used)
    UtPose p0, p1;
    ut_rotmat_to_quat(c.R, p0.q);
    p0.t[0] = c.t[0]; p0.t[1] = c.t[1]; p0.t[2] = c.t[2];
    p1 = p0;
    if (a.viewmats1) {
        const Cam c1 = load_cam(a.viewmats1 + (size_t)img * 16, a.viewmats1 + (size_t)img * 16);
        ut_rotmat_to_quat(c1.R, p1.q);
        p1.t[0] = c1.t[0]; p1.t[1] = c1.t[1]; p1.t[2] = c1.t[2];
    }
    const UtPose pose = ut_interpolate_pose(p0, p1, ut_relative_frame_time(a, ipx, ipy));
    const float qi[4] = {pose.q[0], -pose.q[1], -pose.q[2], -pose.q[3]};
    const float rel[3] = {-pose.t[0], -pose.t[1], -pose.t[2]};
    float o[3], dw[3];
    ut_quat_rotate(qi, rel, o);
    ut_quat_rotate(qi, ray, dw);
    float *out = g.rays + (size_t)idx * 6;
    const float keep = valid ? 1.0f : 0.0f;
    for (int i = 0; i < 3; ++i) {
        out[i]     = o[i] * keep;
        out[3 + i] = dw[i] * keep;
    }
}
} // namespace gsx
extern "C" int gsx_lidar_rays(const float *viewmats, const float *viewmats_rs, const float *row_elevations,
                              const float *column_azimuths, const float *row_azimuth_offsets, uint32_t n_images, uint32_t n_rows,
                              uint32_t n_columns, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                              double fov_vert_span, double fov_eps, int spinning_ccw, const int32_t *angles_to_columns_map,
                              uint32_t map_h, uint32_t map_w, int rs_type, float *rays, void *stream)
{
    using namespace gsx;
    const int64_t n = (int64_t)n_images * n_rows * n_columns;
    if (n == 0) return GSX_OK;
    GSX_REQUIRE(viewmats && row_elevations && column_azimuths && row_azimuth_offsets && rays, "gsx_lidar_rays: null pointer");
    GSX_REQUIRE(rs_type >= 0 && rs_type <= 4, "gsx_lidar_rays: rolling shutter type %d (0 .. 3 rolling, 4 global)", rs_type);
    GSX_REQUIRE(rs_type == 4 || (viewmats_rs && angles_to_columns_map && map_h > 1 && map_w > 1 && n_columns > 1),
                "gsx_lidar_rays: a rolling shutter needs the end-of-frame poses and the angles_to_columns_map");
    LidarRayArgs g{};
    LidarHost l{fov_horiz_start, fov_horiz_span, fov_vert_start, fov_vert_span, spinning_ccw, angles_to_columns_map, (int)map_h,
                (int)map_w, (int)n_columns};
    set_lidar(g.cam, &l);
    g.cam.camera_model = 4; g.cam.rs_type = rs_type;
    g.cam.viewmats = viewmats; g.cam.viewmats1 = rs_type == 4 ? nullptr : viewmats_rs;
    g.row_el = row_elevations; g.col_az = column_azimuths; g.row_off = row_azimuth_offsets;
    g.n_images = n_images; g.n_rows = n_rows; g.n_cols = n_columns; g.eps = (float)fov_eps; g.rays = rays;
    lidar_rays_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(g);
    return check_launch("lidar_rays");
}
