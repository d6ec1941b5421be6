// Gaussian -> tile walk shared by the intersection kernels (intersect.hip, isect_fused.hip).
//
// Restates the behaviour of the reference's tile test (gsplat/cuda/csrc/IntersectTile.cu:83-207 ellipse helpers,
// :288-387 count/emit bodies). Every translation unit that includes this header is compiled with -ffp-contract=off:
// each float operation below is a single IEEE operation (HIP's / and sqrtf are correctly rounded, det_logf uses
// explicit fmaf) and the CPU oracle (oracle/gsplat_oracle.c) performs the same operations in the same order, so the
// INTEGER outputs (tile counts, keys, offsets) are bit-identical between the two.
#pragma once
#include "common.hpp"

namespace gsx {

__host__ __device__ __forceinline__ int f2i_trunc_sat(float x)
{
    // float -> int truncation, saturating (NaN -> 0), identical on host and device
    if (!(x == x)) return 0;
    if (x >= 2.0e9f) return 2000000000;
    if (x <= -2.0e9f) return -2000000000;
    return (int)x;
}
__host__ __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Where does the line u = coord cut the level-t ellipse of conic (A,B,C) centred at p?
// Returns the two roots in v. (quadratic: coeff*(v-pv)^2 + 2*B*h*(v-pv) + other*h^2 - t = 0)
__host__ __device__ __forceinline__ void ellipse_cut(float B, float coeff, float disc, float t, float pu, float pv,
                                                     float coord, float &lo, float &hi)
{
    const float h    = coord - pu;
    const float arg  = disc * h * h + t * coeff;
    const float root = sqrtf(arg > 0.0f ? arg : 0.0f);
    const float mbh  = -B * h;
    lo               = (mbh - root) / coeff + pv;
    hi               = (mbh + root) / coeff + pv;
}

// The walk, slab by slab: span(alongY, u, tv0, tv1) = the tiles (x, y) = (v, u) [alongY] or (u, v) for v in [tv0, tv1), in
// the order walk_tiles emits them (an empty span, tv1 <= tv0, may be reported). A row's spans are what the Gaussian-major
// path keeps from its counting pass so that the emission pass does not repeat the arithmetic (isect_fused.hip).
template <typename Span>
__host__ __device__ __forceinline__ void walk_spans(
    float mx, float my, float rx, float ry, bool has_conic, float A, float B, float C, float opacity,
    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, Span &&span)
{
    const float ts = (float)tile_size;
    // x / ts, bit for bit: for a power-of-two tile size the product with the (exact) reciprocal IS the correctly
    // rounded quotient, and it replaces a ~12-instruction IEEE division on the hot path of the slab loop
    const bool ts_pow2 = (tile_size & (tile_size - 1u)) == 0u;
    const float ts_inv = 1.0f / ts;
    auto div_ts        = [&](float x) { return ts_pow2 ? x * ts_inv : x / ts; };
    if (has_conic) {
        // exact ellipse-vs-tile walk (SNUGBOX bbox + per-slab extents), opacity-aware level set
        const float disc = B * B - A * C;
        float t          = 2.0f * det_logf(opacity * 255.0f);
        const float tmax = kGaussianExtend * kGaussianExtend;
        if (t > tmax) t = tmax;
        if (!(t > 0.0f) || !(disc < 0.0f)) return;
        const float s  = -t / disc;
        const float ex = sqrtf(s * C), ey = sqrtf(s * A);
        const float bminx = mx - ex, bmaxx = mx + ex, bminy = my - ey, bmaxy = my + ey;
        const float bx_c = B * ex / C, by_a = B * ey / A;
        // coordinate (on the other axis) at which each bbox side touches the ellipse
        const float y_at_xmin = my + bx_c, y_at_xmax = my - bx_c;
        const float x_at_ymin = mx + by_a, x_at_ymax = mx - by_a;

        const int rminx = clampi(f2i_trunc_sat(div_ts(bminx)), 0, (int)tile_w);
        const int rminy = clampi(f2i_trunc_sat(div_ts(bminy)), 0, (int)tile_h);
        const int rmaxx = clampi(f2i_trunc_sat(div_ts(bmaxx) + 1.0f), 0, (int)tile_w);
        const int rmaxy = clampi(f2i_trunc_sat(div_ts(bmaxy) + 1.0f), 0, (int)tile_h);
        const int yspan = rmaxy - rminy, xspan = rmaxx - rminx;
        if (yspan <= 0 || xspan <= 0) return;

        // iterate slabs along the SHORTER span (u), solve the covered range on the other axis (v)
        const bool alongY = yspan < xspan;
        const int u0 = alongY ? rminy : rminx, u1 = alongY ? rmaxy : rmaxx;
        const int v0 = alongY ? rminx : rminy, v1 = alongY ? rmaxx : rmaxy;
        const float pu = alongY ? my : mx, pv = alongY ? mx : my;
        const float bmin_u = alongY ? bminy : bminx, bmax_u = alongY ? bmaxy : bmaxx;
        const float bmin_v = alongY ? bminx : bminy, bmax_v = alongY ? bmaxx : bmaxy;
        const float u_at_vmin = alongY ? y_at_xmin : x_at_ymin; // u where v is minimal
        const float u_at_vmax = alongY ? y_at_xmax : x_at_ymax;
        const float coeff     = alongY ? A : C;

        float hi_lo = bmax_v, hi_hi = bmin_v; // "empty" interval: neutral under min/max below
        float lo_lo, lo_hi;
        float line_lo = (float)u0 * ts;
        if (bmin_u <= line_lo) ellipse_cut(B, coeff, disc, t, pu, pv, line_lo, lo_lo, lo_hi);
        else { lo_lo = hi_lo; lo_hi = hi_hi; }

        for (int u = u0; u < u1; ++u) {
            const float line_hi = line_lo + ts;
            if (line_hi <= bmax_u) ellipse_cut(B, coeff, disc, t, pu, pv, line_hi, hi_lo, hi_hi);
            const float vmin = (line_lo <= u_at_vmin && u_at_vmin < line_hi) ? bmin_v : fminf(lo_lo, hi_lo);
            const float vmax = (line_lo <= u_at_vmax && u_at_vmax < line_hi) ? bmax_v : fmaxf(lo_hi, hi_hi);
            const int tv0    = clampi(f2i_trunc_sat(div_ts(vmin)), v0, v1);
            const int tv1    = clampi(f2i_trunc_sat(div_ts(vmax) + 1.0f), v0, v1);
            span(alongY, u, tv0, tv1);
            lo_lo   = hi_lo;
            lo_hi   = hi_hi;
            line_lo = line_hi;
        }
        return;
    }
    // axis-aligned bounding box of (mean +- radius): min inclusive (floor), max exclusive (ceil)
    const float tx = div_ts(mx), ty = div_ts(my), trx = div_ts(rx), try_ = div_ts(ry);
    const int x0 = clampi(f2i_trunc_sat(floorf(tx - trx)), 0, (int)tile_w);
    const int y0 = clampi(f2i_trunc_sat(floorf(ty - try_)), 0, (int)tile_h);
    const int x1 = clampi(f2i_trunc_sat(ceilf(tx + trx)), 0, (int)tile_w);
    const int y1 = clampi(f2i_trunc_sat(ceilf(ty + try_)), 0, (int)tile_h);
    for (int y = y0; y < y1; ++y) span(true, y, x0, x1);
}

// Visits every tile touched by the Gaussian; calls emit(tile_id). Returns the tile count.
template <typename Emit>
__host__ __device__ __forceinline__ int32_t walk_tiles(
    float mx, float my, float rx, float ry, bool has_conic, float A, float B, float C, float opacity,
    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, Emit &&emit)
{
    int32_t count = 0;
    walk_spans(mx, my, rx, ry, has_conic, A, B, C, opacity, tile_size, tile_w, tile_h, [&](bool alongY, int u, int tv0, int tv1) {
        for (int v = tv0; v < tv1; ++v) {
            emit(alongY ? (int64_t)u * tile_w + v : (int64_t)v * tile_w + u);
            ++count;
        }
    });
    return count;
}

// float64 rows (the reference instantiates its intersection kernel for double as well, AT_DISPATCH_FLOATING_TYPES in
// IntersectTile.cu, and tests it: tests/test_basic.py:1268-1316): the radius-box enumeration of the branch above in double -
// quotient, floor / ceil, clamp: every step a single IEEE operation, identical on host and device.
__host__ __device__ __forceinline__ int f2i_trunc_sat(double x)
{
    if (!(x == x)) return 0;
    if (x >= 2.0e9) return 2000000000;
    if (x <= -2.0e9) return -2000000000;
    return (int)x;
}
template <typename Emit>
__host__ __device__ __forceinline__ int32_t walk_tiles_aabb_f64(double mx, double my, double rx, double ry, uint32_t tile_size,
                                                                uint32_t tile_w, uint32_t tile_h, Emit &&emit)
{
    const double ts = (double)tile_size;
    const double tx = mx / ts, ty = my / ts, trx = rx / ts, try_ = ry / ts;
    const int x0 = clampi(f2i_trunc_sat(floor(tx - trx)), 0, (int)tile_w);
    const int y0 = clampi(f2i_trunc_sat(floor(ty - try_)), 0, (int)tile_h);
    const int x1 = clampi(f2i_trunc_sat(ceil(tx + trx)), 0, (int)tile_w);
    const int y1 = clampi(f2i_trunc_sat(ceil(ty + try_)), 0, (int)tile_h);
    int32_t count = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            emit((int64_t)y * tile_w + x);
            ++count;
        }
    return count;
}

} // namespace gsx
