// Tile intersection for spinning-lidar "cameras" (3DGUT): a Gaussian's footprint is a box in ANGLE space - azimuth x elevation,
// in angular pixels (angle * 1024) - and the tiles are the cells of the lidar's tiling (n_bins_azimuth uniform cells over the
// horizontal field of view; n_bins_elevation cells that a CDF maps onto the rows' elevations). C-ABI entries:
// gsx_isect_lidar_count / gsx_isect_lidar_emit; replace gsplat::intersect_tile_lidar (ext.cpp:1037-1040; host fn Intersect.cpp:
// 388-520; kernel IntersectTileLidar.cu:136-415; torch statement gsplat/cuda/_torch_impl_lidar.py:34-394, which this file follows
// operation for operation in float32 - the counts and keys are compared for EQUALITY with it by the reference's tests).
//
// Per row (one thread): the mean's angles relative to the start of the field of view (azimuth in the spinning direction, modulo a
// full circle; elevation downwards), +- the extent. Azimuth wraps: the box becomes region A and, when it crosses the 0 / 2 pi seam,
// region B. Each region's ends are sampled into the DENSE grid (floor / ceil) and the tiling; a summed-area table of the rays
// (cdf_dense_ray_mask) says whether the region holds any ray at all - a region without rays yields no tile. A periodic field of
// view (span >= 2 pi) merges B into A across the seam, tile indices taken modulo n_bins_azimuth.
// Compiled with -ffp-contract=off: every float operation is one IEEE operation, as in the torch statement.
#include "common.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

struct LidarTileArgs {
    const float *means2d;      // [R,2] azimuth | elevation, angular pixels
    const int32_t *radii;      // [R,2] extent in angular pixels (int32, as the projection writes them) ...
    const float *radii_f;      // ... or float (one of the two)
    const float *depths;       // [R] (emit)
    const int64_t *image_ids;  // [R] or null (dense: row / n_per_image)
    int64_t rows, n_per_image;
    float ref_az, ref_el;      // start of the field of view, angular pixels
    float span_az, span_el;    // its spans
    float full_circle;         // 2 pi * 1024
    int ccw;                   // spinning direction: 1 = counter-clockwise
    int n_bins_az, n_bins_el, res_az, res_el, periodic;
    const int32_t *cdf_el;     // [res_el + 1] dense elevation bin -> tile row
    const int32_t *raycdf;     // [res_el + 1][res_az + 1] summed-area table of the dense ray mask
    uint32_t tile_bits;
    int32_t *tiles_per_gauss;  // [R] (count)
    const int64_t *cum;        // [R] inclusive prefix sum of tiles_per_gauss (emit)
    int64_t *isect_ids;        // [M] (emit)
    int32_t *flatten_ids;      // [M] (emit)
};

struct LidarRanges {
    int el0, el1, a0, a1, b0, b1;
};

// torch.remainder on floats: the result takes the sign of the divisor
__device__ __forceinline__ float py_mod(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

struct LidarSample {
    int tile_az, tile_el, dense_az, dense_el;
};
template <bool CEIL>
__device__ __forceinline__ LidarSample lidar_sample(const LidarTileArgs &a, float rel_az, float rel_el)
{
    LidarSample s;
    const float norm_az = rel_az / a.span_az, norm_el = rel_el / a.span_el;
    const float da = norm_az * (float)a.res_az, de = norm_el * (float)a.res_el, ta = norm_az * (float)a.n_bins_az;
    s.dense_az = (int)(CEIL ? ceilf(da) : floorf(da));
    s.dense_el = (int)(CEIL ? ceilf(de) : floorf(de));
    const int de_c = min(max(s.dense_el, 0), a.res_el); // (in range by construction; the clamp only guards the loads)
    if (CEIL) // [min, max) is half-open: the tile row AFTER the one the last covered dense bin belongs to
        s.tile_el = s.dense_el >= 1 ? min(a.cdf_el[max(de_c - 1, 0)] + 1, a.n_bins_el) : a.cdf_el[de_c];
    else
        s.tile_el = a.cdf_el[de_c];
    s.tile_az = (int)(CEIL ? ceilf(ta) : floorf(ta));
    return s;
}
__device__ __forceinline__ bool lidar_has_rays(const LidarTileArgs &a, const LidarSample &b, const LidarSample &e)
{
    const int w = a.res_az + 1;
    auto at = [&](int el, int az) { return a.raycdf[(size_t)min(max(el, 0), a.res_el) * w + min(max(az, 0), a.res_az)]; };
    const int n = at(e.dense_el, e.dense_az) - at(b.dense_el, e.dense_az) - at(e.dense_el, b.dense_az) + at(b.dense_el, b.dense_az);
    return n > 0 || (b.dense_az <= 0 && e.dense_az >= a.res_az); // the full-cover case is taken without reading the table
}

__device__ __forceinline__ LidarRanges lidar_ranges(const LidarTileArgs &a, int64_t r)
{
    LidarRanges o{0, 0, 0, 0, 0, 0};
    const float az = a.means2d[2 * r], el = a.means2d[2 * r + 1];
    const float ex = a.radii ? (float)a.radii[2 * r] : a.radii_f[2 * r], ey = a.radii ? (float)a.radii[2 * r + 1] : a.radii_f[2 * r + 1];
    const bool nonzero = ex > 0.0f && ey > 0.0f;
    // 1. relative angles of the mean, then +- the extent
    const float rel_az = py_mod(a.ccw ? az - a.ref_az : a.ref_az - az, a.full_circle);
    const float rel_el = a.ref_el - el;
    float beg_az = rel_az - ex, end_az = rel_az + ex;
    const float beg_el = rel_el - ey, end_el = rel_el + ey;
    // 2. regions A and B
    const bool full_cover = beg_az <= 0.0f && end_az >= a.span_az;
    end_az = fminf(end_az, beg_az + a.full_circle);
    const bool overflows = end_az > a.full_circle, underflows = beg_az < 0.0f;
    float begA = (full_cover || underflows) ? 0.0f : beg_az;
    float endA = full_cover ? a.span_az : (overflows ? a.full_circle : end_az);
    float begB = (underflows && !full_cover) ? beg_az + a.full_circle : 0.0f;
    float endB = (overflows && !full_cover) ? end_az - a.full_circle : ((underflows && !full_cover) ? a.full_circle : 0.0f);
    // 3. clamp to the field of view, sample the ends
    auto caz = [&](float v) { return fminf(fmaxf(v, 0.0f), a.span_az); };
    auto cel = [&](float v) { return fminf(fmaxf(v, 0.0f), a.span_el); };
    const float be = cel(beg_el), ee = cel(end_el);
    const LidarSample sbA = lidar_sample<false>(a, caz(begA), be), seA = lidar_sample<true>(a, caz(endA), ee);
    const LidarSample sbB = lidar_sample<false>(a, caz(begB), be), seB = lidar_sample<true>(a, caz(endB), ee);
    // 4. any ray in the regions?
    const bool raysA = nonzero && lidar_has_rays(a, sbA, seA), raysB = nonzero && lidar_has_rays(a, sbB, seB);
    const bool rays  = raysA || raysB;
    // 5. tile ranges
    o.el0 = rays ? sbA.tile_el : 0;
    o.el1 = rays ? seA.tile_el : 0;
    int a0 = raysA ? sbA.tile_az : 0, a1 = raysA ? seA.tile_az : 0;
    int b0 = raysB ? sbB.tile_az : 0, b1 = raysB ? seB.tile_az : 0;
    const int nb = a.n_bins_az;
    if (a.periodic) { // tiles wrap: B joins A across the seam, at most n_bins wide
        if (raysB && underflows) a0 = b0 - nb;
        if (raysB && overflows) a1 = b1 + nb;
        a0 = max(a0, a1 - nb);
        a1 = min(a1, a0 + nb);
        b0 = b1 = 0;
    } else if (raysA && raysB && b0 < a1 && a0 < b1) { // the two ranges meet: the whole field of view
        a0 = 0; a1 = nb; b0 = b1 = 0;
    }
    o.a0 = a0; o.a1 = a1; o.b0 = b0; o.b1 = b1;
    return o;
}

__global__ void __launch_bounds__(256) lidar_count_kernel(const LidarTileArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    const LidarRanges g = lidar_ranges(a, r);
    a.tiles_per_gauss[r] = (g.el1 - g.el0) * ((g.a1 - g.a0) + (g.b1 - g.b0));
}

__global__ void __launch_bounds__(256) lidar_emit_kernel(const LidarTileArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    const LidarRanges g = lidar_ranges(a, r);
    const int64_t n = (int64_t)(g.el1 - g.el0) * ((g.a1 - g.a0) + (g.b1 - g.b0));
    if (n <= 0) return;
    int64_t out = a.cum[r] - n; // inclusive prefix -> this row's first slot
    const uint64_t image = a.image_ids ? (uint64_t)a.image_ids[r] : (uint64_t)(r / a.n_per_image);
    const uint64_t lo    = (uint64_t)__float_as_uint(a.depths[r]);
    for (int el = g.el0; el < g.el1; ++el) { // elevation-major, region A before region B (the order of the torch statement)
        for (int pass = 0; pass < 2; ++pass) {
            const int s = pass ? g.b0 : g.a0, e = pass ? g.b1 : g.a1;
            for (int az = s; az < e; ++az) {
                int t = az;
                if (a.periodic) { t %= a.n_bins_az; if (t < 0) t += a.n_bins_az; }
                const uint64_t tile = (uint64_t)el * (uint64_t)a.n_bins_az + (uint64_t)t;
                a.isect_ids[out]   = (int64_t)((((image << a.tile_bits) | tile) << 32) | lo);
                a.flatten_ids[out] = (int32_t)r;
                ++out;
            }
        }
    }
}

} // namespace gsx

using namespace gsx;

static int lidar_fill(LidarTileArgs &a, const char *fn, const float *means2d, const int32_t *radii_i32, const float *radii_f32,
                      int64_t rows, int64_t n_per_image, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                      double fov_vert_span, int spinning_ccw, uint32_t n_bins_azimuth, uint32_t n_bins_elevation,
                      uint32_t cdf_resolution_azimuth, uint32_t cdf_resolution_elevation, const int32_t *cdf_elevation,
                      const int32_t *cdf_dense_ray_mask)
{
    GSX_REQUIRE(means2d && (radii_i32 || radii_f32) && cdf_elevation && cdf_dense_ray_mask, "%s: null pointer", fn);
    GSX_REQUIRE(n_bins_azimuth > 0 && n_bins_elevation > 0 && n_per_image > 0, "%s: empty tiling", fn);
    constexpr double kScale = 1024.0; // ANGLE_TO_PIXEL_SCALING_FACTOR
    a.means2d = means2d; a.radii = radii_i32; a.radii_f = radii_f32; a.rows = rows; a.n_per_image = n_per_image;
    // python floats (double) meet float32 tensors: the scalar is rounded to float once
    a.ref_az = (float)(fov_horiz_start * kScale); a.ref_el = (float)(fov_vert_start * kScale);
    a.span_az = (float)(kScale * fov_horiz_span); a.span_el = (float)(kScale * fov_vert_span);
    a.full_circle = (float)(2.0 * 3.141592653589793 * kScale);
    a.periodic    = (kScale * fov_horiz_span >= 2.0 * 3.141592653589793 * kScale) ? 1 : 0;
    a.ccw = spinning_ccw ? 1 : 0;
    a.n_bins_az = (int)n_bins_azimuth; a.n_bins_el = (int)n_bins_elevation;
    a.res_az = (int)cdf_resolution_azimuth; a.res_el = (int)cdf_resolution_elevation;
    a.cdf_el = cdf_elevation; a.raycdf = cdf_dense_ray_mask;
    return GSX_OK;
}

extern "C" int gsx_isect_lidar_count(const float *means2d, const int32_t *radii_i32, const float *radii_f32, int64_t rows,
                                     int64_t n_per_image, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                                     double fov_vert_span, int spinning_ccw, uint32_t n_bins_azimuth, uint32_t n_bins_elevation,
                                     uint32_t cdf_resolution_azimuth, uint32_t cdf_resolution_elevation,
                                     const int32_t *cdf_elevation, const int32_t *cdf_dense_ray_mask, int32_t *tiles_per_gauss,
                                     void *stream)
{
    if (rows <= 0) return GSX_OK;
    LidarTileArgs a{};
    const int rc = lidar_fill(a, "gsx_isect_lidar_count", means2d, radii_i32, radii_f32, rows, n_per_image, fov_horiz_start,
                              fov_horiz_span, fov_vert_start, fov_vert_span, spinning_ccw, n_bins_azimuth, n_bins_elevation,
                              cdf_resolution_azimuth, cdf_resolution_elevation, cdf_elevation, cdf_dense_ray_mask);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(tiles_per_gauss, "gsx_isect_lidar_count: null output");
    a.tiles_per_gauss = tiles_per_gauss;
    lidar_count_kernel<<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("isect_lidar_count");
}

extern "C" int gsx_isect_lidar_emit(const float *means2d, const int32_t *radii_i32, const float *radii_f32, const float *depths,
                                    const int64_t *image_ids, const int64_t *cum_tiles, int64_t rows, int64_t n_per_image,
                                    uint32_t n_images, double fov_horiz_start, double fov_horiz_span, double fov_vert_start,
                                    double fov_vert_span, int spinning_ccw, uint32_t n_bins_azimuth, uint32_t n_bins_elevation,
                                    uint32_t cdf_resolution_azimuth, uint32_t cdf_resolution_elevation,
                                    const int32_t *cdf_elevation, const int32_t *cdf_dense_ray_mask, int64_t *isect_ids,
                                    int32_t *flatten_ids, void *stream)
{
    if (rows <= 0) return GSX_OK;
    LidarTileArgs a{};
    const int rc = lidar_fill(a, "gsx_isect_lidar_emit", means2d, radii_i32, radii_f32, rows, n_per_image, fov_horiz_start,
                              fov_horiz_span, fov_vert_start, fov_vert_span, spinning_ccw, n_bins_azimuth, n_bins_elevation,
                              cdf_resolution_azimuth, cdf_resolution_elevation, cdf_elevation, cdf_dense_ray_mask);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(depths && cum_tiles && isect_ids && flatten_ids, "gsx_isect_lidar_emit: null pointer");
    (void)n_images;
    a.depths = depths; a.image_ids = image_ids; a.cum = cum_tiles; a.isect_ids = isect_ids; a.flatten_ids = flatten_ids;
    uint32_t bits = 0;
    for (uint64_t v = (uint64_t)n_bins_azimuth * n_bins_elevation; v > 1 && ((v - 1) >> bits) != 0; ++bits) {}
    a.tile_bits = bits;
    lidar_emit_kernel<<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("isect_lidar_emit");
}
