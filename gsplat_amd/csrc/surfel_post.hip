// Per-pixel post-processing of the 2DGS orchestrator, fused into one kernel per direction.
//
// After compositing, rasterization_2dgs turns the accumulated depth channel into the expected depth (divide by the pixel's
// alpha), rotates the rendered normals from camera space to world space with inv(viewmat), and derives a second normal map
// from the depth map by central differences of the unprojected points (reference: gsplat/rendering.py:1519-1552 and the
// C++ orchestrator gsplat/cuda/csrc/Rendering.cpp:1653-1702, 1905-1935, which composes it from ~40 ATen ops per direction:
// three of them batched GEMMs with a 3x3 operand, 125-135 us each at 1080p, the rest elementwise passes over [H, W, 3]).
// Here: one thread per pixel, every input read once (depths of the four neighbours come from L2), every output written once.
// HBM-bound: (D + 1 + 3 + 1) floats in, (D + 3 + 3) floats out per pixel.
//
// The camera-to-world transform is inverted in the kernel (adjugate of the upper-left 3x3 block of the view matrix; the
// translation is not needed: the ray origin cancels in the central differences).
#include "common.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

struct SurfelPostArgs {
    const float *colors, *alphas, *normals, *median, *viewmats, *Ks;
    uint32_t width, height, cdim;
    int expected_depth, depth_source; // depth_source: 0 none, 1 last colour channel (after normalisation), 2 median
    // forward outputs
    float *colors_out, *normals_world, *surf_normals;
    // backward
    const float *v_colors_out, *v_normals_world, *v_surf_normals;
    float *v_colors, *v_alphas, *v_normals, *v_median;
};

struct Mat3 {
    float m[9];
    __device__ __forceinline__ void mul(float x, float y, float z, float &ox, float &oy, float &oz) const
    {
        ox = fmaf(m[0], x, fmaf(m[1], y, m[2] * z));
        oy = fmaf(m[3], x, fmaf(m[4], y, m[5] * z));
        oz = fmaf(m[6], x, fmaf(m[7], y, m[8] * z));
    }
    __device__ __forceinline__ void mul_t(float x, float y, float z, float &ox, float &oy, float &oz) const
    {
        ox = fmaf(m[0], x, fmaf(m[3], y, m[6] * z));
        oy = fmaf(m[1], x, fmaf(m[4], y, m[7] * z));
        oz = fmaf(m[2], x, fmaf(m[5], y, m[8] * z));
    }
};

// inverse of the upper-left 3x3 block of a row-major 4x4 view matrix (uniform per image: scalar loads)
__device__ __forceinline__ Mat3 cam_to_world_rotation(const float *__restrict__ V)
{
    const float a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[6], g = V[8], h = V[9], i = V[10];
    const float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const float inv = 1.0f / (a * A + b * B + c * C);
    Mat3 r;
    r.m[0] = A * inv; r.m[1] = (c * h - b * i) * inv; r.m[2] = (b * f - c * e) * inv;
    r.m[3] = B * inv; r.m[4] = (a * i - c * g) * inv; r.m[5] = (c * d - a * f) * inv;
    r.m[6] = C * inv; r.m[7] = (b * g - a * h) * inv; r.m[8] = (a * e - b * d) * inv;
    return r;
}

struct Intrinsics {
    float inv_fx, inv_fy, cx, cy;
    __device__ __forceinline__ void dir(uint32_t x, uint32_t y, float &dx, float &dy) const
    {
        dx = ((float)x - cx + 0.5f) * inv_fx;
        dy = ((float)y - cy + 0.5f) * inv_fy;
    }
};

constexpr float kAlphaFloor = 1e-10f, kNormFloor = 1e-12f;

__device__ __forceinline__ float depth_at(const SurfelPostArgs &a, size_t pix)
{
    if (a.depth_source == 2) return a.median[pix];
    const float acc = a.colors[pix * a.cdim + (a.cdim - 1)];
    return a.expected_depth ? acc / fmaxf(a.alphas[pix], kAlphaFloor) : acc;
}

// central differences of the unprojected points around interior pixel (x, y), in world space, from its four neighbours' depths
__device__ __forceinline__ void point_differences(const Mat3 &R, const Intrinsics &K, uint32_t x, uint32_t y, float d_up, float d_dn,
                                                  float d_lf, float d_rt, float (&du)[3], float (&dv)[3])
{
    float rx, ry, rx1, ry1, rx0, ry0;
    K.dir(x, y, rx, ry);
    K.dir(x + 1, y + 1, rx1, ry1);
    K.dir(x - 1, y - 1, rx0, ry0);
    // camera-space points are depth * (dir.x, dir.y, 1); the first difference is along the image rows (y), as the reference's
    R.mul(rx * (d_dn - d_up), fmaf(d_dn, ry1, -d_up * ry0), d_dn - d_up, du[0], du[1], du[2]);
    R.mul(fmaf(d_rt, rx1, -d_lf * rx0), ry * (d_rt - d_lf), d_rt - d_lf, dv[0], dv[1], dv[2]);
}

__device__ __forceinline__ void cross3(const float (&p)[3], const float (&q)[3], float (&o)[3])
{
    o[0] = fmaf(p[1], q[2], -p[2] * q[1]);
    o[1] = fmaf(p[2], q[0], -p[0] * q[2]);
    o[2] = fmaf(p[0], q[1], -p[1] * q[0]);
}

__device__ __forceinline__ Intrinsics load_intrinsics(const float *__restrict__ K)
{
    return Intrinsics{1.0f / K[0], 1.0f / K[4], K[2], K[5]};
}

// Workgroup = 32 x 8 pixels. The depth map of the block and a halo is formed ONCE per pixel in LDS (with expected depths that is
// two strided loads and a division): a pixel's central differences read its four neighbours there instead of evaluating
// depth_at() four times from global memory (round 6: forward 45 -> see profiles/r11_ab.md; same values, same arithmetic).
constexpr int kPBX = 32, kPBY = 8;

__device__ __forceinline__ float depth_or_zero(const SurfelPostArgs &a, size_t img, int gx, int gy)
{
    return (gx >= 0 && gy >= 0 && gx < (int)a.width && gy < (int)a.height) ? depth_at(a, img + (size_t)gy * a.width + gx) : 0.0f;
}

__global__ void __launch_bounds__(256) surfel_post_fwd_kernel(const SurfelPostArgs a)
{
    __shared__ float s_d[kPBY + 2][kPBX + 2]; // depths, halo 1
    const int bx0 = (int)blockIdx.x * kPBX, by0 = (int)blockIdx.y * kPBY;
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5, im = blockIdx.z;
    const uint32_t x = (uint32_t)bx0 + tx, y = (uint32_t)by0 + ty;
    const size_t img = (size_t)im * a.width * a.height;
    if (a.surf_normals) {
        for (int i = (int)threadIdx.x; i < (kPBY + 2) * (kPBX + 2); i += 256) {
            const int ry = i / (kPBX + 2), rx = i % (kPBX + 2);
            s_d[ry][rx] = depth_or_zero(a, img, bx0 + rx - 1, by0 + ry - 1);
        }
        __syncthreads();
    }
    if (x >= a.width || y >= a.height) return;
    const Mat3 R     = cam_to_world_rotation(a.viewmats + 16 * (size_t)im);
    const size_t pix = img + (size_t)y * a.width + x;
    if (a.colors_out) {
        const float inv = 1.0f / fmaxf(a.alphas[pix], kAlphaFloor);
        if (a.cdim == 4 && ((reinterpret_cast<uintptr_t>(a.colors) | reinterpret_cast<uintptr_t>(a.colors_out)) & 15u) == 0) { // RGB + depth: 16-byte accesses
            float4 c = reinterpret_cast<const float4 *>(a.colors)[pix];
            c.w *= inv;
            reinterpret_cast<float4 *>(a.colors_out)[pix] = c;
        } else {
            for (uint32_t k = 0; k + 1 < a.cdim; ++k) a.colors_out[pix * a.cdim + k] = a.colors[pix * a.cdim + k];
            a.colors_out[pix * a.cdim + a.cdim - 1] = a.colors[pix * a.cdim + a.cdim - 1] * inv;
        }
    }
    float wx, wy, wz;
    R.mul(a.normals[3 * pix], a.normals[3 * pix + 1], a.normals[3 * pix + 2], wx, wy, wz);
    a.normals_world[3 * pix] = wx; a.normals_world[3 * pix + 1] = wy; a.normals_world[3 * pix + 2] = wz;
    if (a.surf_normals) {
        float n[3] = {0.0f, 0.0f, 0.0f};
        if (x >= 1 && y >= 1 && x + 1 < a.width && y + 1 < a.height) {
            const Intrinsics K = load_intrinsics(a.Ks + 9 * (size_t)im);
            float du[3], dv[3];
            point_differences(R, K, x, y, s_d[ty][tx + 1], s_d[ty + 2][tx + 1], s_d[ty + 1][tx], s_d[ty + 1][tx + 2], du, dv);
            cross3(du, dv, n);
            const float s = 1.0f / fmaxf(sqrtf(fmaf(n[0], n[0], fmaf(n[1], n[1], n[2] * n[2]))), kNormFloor);
            n[0] *= s; n[1] *= s; n[2] *= s;
        }
        a.surf_normals[3 * pix] = n[0]; a.surf_normals[3 * pix + 1] = n[1]; a.surf_normals[3 * pix + 2] = n[2];
    }
}

// gradient of the loss w.r.t. the two point differences of interior pixel (x, y), given the gradient of its unit normal
__device__ __forceinline__ void difference_grads(const SurfelPostArgs &a, const Mat3 &R, const Intrinsics &K, size_t img,
                                                 uint32_t x, uint32_t y, float d_up, float d_dn, float d_lf, float d_rt,
                                                 float (&v_du)[3], float (&v_dv)[3])
{
    const size_t pix = img + (size_t)y * a.width + x;
    float du[3], dv[3], n[3];
    point_differences(R, K, x, y, d_up, d_dn, d_lf, d_rt, du, dv);
    cross3(du, dv, n);
    const float g[3]  = {a.v_surf_normals[3 * pix], a.v_surf_normals[3 * pix + 1], a.v_surf_normals[3 * pix + 2]};
    const float len   = sqrtf(fmaf(n[0], n[0], fmaf(n[1], n[1], n[2] * n[2])));
    float v_n[3];
    if (len > kNormFloor) { // n / |n|: the gradient is the part of g orthogonal to the unit normal, over |n|
        const float s = 1.0f / len, dot = (n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) * s * s;
        for (int k = 0; k < 3; ++k) v_n[k] = (g[k] - n[k] * dot) * s;
    } else { // n / floor (torch.nn.functional.normalize: the norm's subgradient at the clamp is zero)
        for (int k = 0; k < 3; ++k) v_n[k] = g[k] * (1.0f / kNormFloor);
    }
    cross3(dv, v_n, v_du); // d((du x dv) . v_n) / d du = dv x v_n
    cross3(v_n, du, v_dv); // d((du x dv) . v_n) / d dv = v_n x du
}

// Same blocks as the forward. A pixel's depth gradient collects the difference gradients of its four neighbours; each of those
// used to be evaluated by every pixel that needed it (four times, from sixteen depth_at() calls per pixel): now the block forms
// the depth map (halo 2) and every site's difference gradients (halo 1) once, in LDS. Sites that are not interior pixels of the
// image hold zeros, so the four terms are added unconditionally (x + 0 is exact: the same sums as the guarded form).
__global__ void __launch_bounds__(256) surfel_post_bwd_kernel(const SurfelPostArgs a)
{
    __shared__ float s_d[kPBY + 4][kPBX + 4];    // depths, halo 2
    __shared__ float s_g[6][kPBY + 2][kPBX + 2]; // v_du[3] | v_dv[3] of every site, halo 1
    const int bx0 = (int)blockIdx.x * kPBX, by0 = (int)blockIdx.y * kPBY;
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5, im = blockIdx.z;
    const uint32_t x = (uint32_t)bx0 + tx, y = (uint32_t)by0 + ty;
    const Mat3 R     = cam_to_world_rotation(a.viewmats + 16 * (size_t)im);
    const size_t img = (size_t)im * a.width * a.height;
    const bool want_depth_grad = a.v_surf_normals && a.depth_source;
    if (want_depth_grad) {
        const Intrinsics K = load_intrinsics(a.Ks + 9 * (size_t)im);
        for (int i = (int)threadIdx.x; i < (kPBY + 4) * (kPBX + 4); i += 256) {
            const int ry = i / (kPBX + 4), rx = i % (kPBX + 4);
            s_d[ry][rx] = depth_or_zero(a, img, bx0 + rx - 2, by0 + ry - 2);
        }
        __syncthreads();
        for (int i = (int)threadIdx.x; i < (kPBY + 2) * (kPBX + 2); i += 256) {
            const int ry = i / (kPBX + 2), rx = i % (kPBX + 2);
            const int gx = bx0 + rx - 1, gy = by0 + ry - 1;
            float v_du[3] = {0.0f, 0.0f, 0.0f}, v_dv[3] = {0.0f, 0.0f, 0.0f};
            if (gx >= 1 && gy >= 1 && gx + 1 < (int)a.width && gy + 1 < (int)a.height) // site (gx, gy) = s_d[ry + 1][rx + 1]
                difference_grads(a, R, K, img, (uint32_t)gx, (uint32_t)gy, s_d[ry][rx + 1], s_d[ry + 2][rx + 1], s_d[ry + 1][rx],
                                 s_d[ry + 1][rx + 2], v_du, v_dv);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_g[k][ry][rx]     = v_du[k];
                s_g[3 + k][ry][rx] = v_dv[k];
            }
        }
        __syncthreads();
    }
    if (x >= a.width || y >= a.height) return;
    const size_t pix = img + (size_t)y * a.width + x;

    float cx, cy, cz;
    R.mul_t(a.v_normals_world[3 * pix], a.v_normals_world[3 * pix + 1], a.v_normals_world[3 * pix + 2], cx, cy, cz);
    a.v_normals[3 * pix] = cx; a.v_normals[3 * pix + 1] = cy; a.v_normals[3 * pix + 2] = cz;

    // gradient of this pixel's depth from the (up to four) interior neighbours whose differences read it: pixel (x, y) is site
    // [ty + 1][tx + 1]; it is the lower neighbour of (x, y-1): +v_du, the upper one of (x, y+1): -v_du, the right one of
    // (x-1, y): +v_dv, the left one of (x+1, y): -v_dv
    float v_depth = 0.0f;
    if (want_depth_grad) {
        const Intrinsics K = load_intrinsics(a.Ks + 9 * (size_t)im);
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = 0.0f;
            v += s_g[k][ty][tx + 1];
            v -= s_g[k][ty + 2][tx + 1];
            v += s_g[3 + k][ty + 1][tx];
            v -= s_g[3 + k][ty + 1][tx + 2];
            vp[k] = v;
        }
        float rx, ry, wx, wy, wz;
        K.dir(x, y, rx, ry);
        R.mul(rx, ry, 1.0f, wx, wy, wz); // d point / d depth = world-space ray direction
        v_depth = vp[0] * wx + vp[1] * wy + vp[2] * wz;
    }
    if (a.v_median) a.v_median[pix] = a.depth_source == 2 ? v_depth : 0.0f;

    const uint32_t D = a.cdim, last = D - 1;
    float v_last     = a.depth_source == 1 ? v_depth : 0.0f; // gradient of the (normalised) depth channel
    const bool quad  = D == 4 && a.v_colors_out && (reinterpret_cast<uintptr_t>(a.v_colors_out) & 15u) == 0
                      && (reinterpret_cast<uintptr_t>(a.v_colors) & 15u) == 0; // RGB + depth: 16-byte accesses
    float4 vq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quad) {
        vq = reinterpret_cast<const float4 *>(a.v_colors_out)[pix];
        v_last += vq.w;
    } else if (a.v_colors_out) {
        for (uint32_t k = 0; k < last; ++k) a.v_colors[pix * D + k] = a.v_colors_out[pix * D + k];
        v_last += a.v_colors_out[pix * D + last];
    } else {
        for (uint32_t k = 0; k < last; ++k) a.v_colors[pix * D + k] = 0.0f;
    }
    float v_alpha = 0.0f;
    if (a.expected_depth) {
        const float alpha = a.alphas[pix], inv = 1.0f / fmaxf(alpha, kAlphaFloor);
        if (alpha >= kAlphaFloor) v_alpha = -v_last * a.colors[pix * D + last] * inv * inv;
        v_last *= inv;
    }
    if (quad) {
        vq.w = v_last;
        reinterpret_cast<float4 *>(a.v_colors)[pix] = vq;
    } else {
        a.v_colors[pix * D + last] = v_last;
    }
    if (a.v_alphas) a.v_alphas[pix] = v_alpha;
}

} // namespace gsx

using namespace gsx;

static int check_common(const char *fn, const float *colors, const float *alphas, const float *normals, const float *median,
                        const float *viewmats, const float *Ks, uint32_t n_images, uint32_t width, uint32_t height,
                        uint32_t cdim, int expected_depth, int depth_source)
{
    GSX_REQUIRE(depth_source >= 0 && depth_source <= 2, "%s: depth_source must be 0, 1 or 2, got %d", fn, depth_source);
    GSX_REQUIRE(!(expected_depth || depth_source == 1) || cdim >= 1, "%s: the depth channel needs cdim >= 1", fn);
    GSX_REQUIRE(n_images == 0 || width == 0 || height == 0 || (normals && viewmats), "%s: null normals / viewmats", fn);
    GSX_REQUIRE(!(expected_depth || depth_source == 1) || n_images * width * height == 0 || colors,
                "%s: null colors with a depth channel in use", fn);
    GSX_REQUIRE(!expected_depth || n_images * width * height == 0 || alphas, "%s: expected_depth needs alphas", fn);
    GSX_REQUIRE(depth_source != 2 || median, "%s: depth_source 2 needs the median depth map", fn);
    GSX_REQUIRE(depth_source == 0 || Ks, "%s: a depth source needs the intrinsics", fn);
    GSX_REQUIRE(n_images <= 65535 && height <= 4u * 65535u, "%s: %u images x %u rows exceed the launch grid", fn, n_images,
                height);
    return GSX_OK;
}

extern "C" int gsx_surfel_post_fwd(const float *colors, const float *alphas, const float *normals, const float *median,
                                   const float *viewmats, const float *Ks, uint32_t n_images, uint32_t width,
                                   uint32_t height, uint32_t cdim, int expected_depth, int depth_source, float *colors_out,
                                   float *normals_world, float *surf_normals, void *stream)
{
    if (int rc = check_common("gsx_surfel_post_fwd", colors, alphas, normals, median, viewmats, Ks, n_images, width, height,
                              cdim, expected_depth, depth_source))
        return rc;
    GSX_REQUIRE(!expected_depth == !colors_out, "gsx_surfel_post_fwd: colors_out is written iff expected_depth is set");
    GSX_REQUIRE(!depth_source == !surf_normals, "gsx_surfel_post_fwd: surf_normals is written iff a depth source is named");
    if (n_images == 0 || width == 0 || height == 0) return GSX_OK;
    GSX_REQUIRE(normals_world, "gsx_surfel_post_fwd: null normals_world");
    SurfelPostArgs a{};
    a.colors = colors; a.alphas = alphas; a.normals = normals; a.median = median; a.viewmats = viewmats; a.Ks = Ks;
    a.width = width; a.height = height; a.cdim = cdim; a.expected_depth = expected_depth; a.depth_source = depth_source;
    a.colors_out = colors_out; a.normals_world = normals_world; a.surf_normals = surf_normals;
    surfel_post_fwd_kernel<<<dim3((uint32_t)ceil_div(width, kPBX), (uint32_t)ceil_div(height, kPBY), n_images), dim3(256), 0,
                             (hipStream_t)stream>>>(a);
    return check_launch("surfel_post_fwd");
}

extern "C" int gsx_surfel_post_bwd(const float *colors, const float *alphas, const float *normals, const float *median,
                                   const float *viewmats, const float *Ks, uint32_t n_images, uint32_t width,
                                   uint32_t height, uint32_t cdim, int expected_depth, int depth_source,
                                   const float *v_colors_out, const float *v_normals_world, const float *v_surf_normals,
                                   float *v_colors, float *v_alphas, float *v_normals, float *v_median, void *stream)
{
    if (int rc = check_common("gsx_surfel_post_bwd", colors, alphas, normals, median, viewmats, Ks, n_images, width, height,
                              cdim, expected_depth, depth_source))
        return rc;
    if (n_images == 0 || width == 0 || height == 0) return GSX_OK;
    GSX_REQUIRE(v_normals_world && v_normals, "gsx_surfel_post_bwd: null normal gradients");
    GSX_REQUIRE(cdim == 0 || v_colors, "gsx_surfel_post_bwd: null v_colors");
    GSX_REQUIRE(!expected_depth || v_alphas, "gsx_surfel_post_bwd: expected_depth needs v_alphas");
    SurfelPostArgs a{};
    a.colors = colors; a.alphas = alphas; a.normals = normals; a.median = median; a.viewmats = viewmats; a.Ks = Ks;
    a.width = width; a.height = height; a.cdim = cdim; a.expected_depth = expected_depth; a.depth_source = depth_source;
    a.v_colors_out = v_colors_out; a.v_normals_world = v_normals_world; a.v_surf_normals = v_surf_normals;
    a.v_colors = v_colors; a.v_alphas = v_alphas; a.v_normals = v_normals; a.v_median = v_median;
    surfel_post_bwd_kernel<<<dim3((uint32_t)ceil_div(width, kPBX), (uint32_t)ceil_div(height, kPBY), n_images), dim3(256), 0,
                             (hipStream_t)stream>>>(a);
    return check_launch("surfel_post_bwd");
}
