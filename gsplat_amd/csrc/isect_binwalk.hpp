// Tile walk CLIPPED to a rectangle of tiles, for the binned intersection (isect_binned.hip).
//
// walk_tiles (isect_walk.hpp) visits every tile a Gaussian touches; the binned path needs, for one (Gaussian, bin)
// pair, exactly those of them that lie inside the bin - as a bit mask - without paying for the slabs outside it.
// walk_prepare() is the first half of walk_tiles (level set, SNUGBOX, tile rectangle: the same IEEE operations in the
// same order), walk_clipped() is its slab loop started at an arbitrary slab:
//   * slab boundaries are (float)u * ts, exact for u * ts < 2^24, so the accumulated `line_lo += ts` of walk_tiles and a
//     direct product agree bit for bit;
//   * the state carried from slab to slab is (lo_lo, lo_hi) = the last cut that was taken at a line <= bmax_u (or the
//     "empty" interval if none was): it is reconstructed by walking back from the first wanted slab to the nearest
//     such line, normally one step.
// The union of the clipped walks over a partition of the tile grid is therefore the unclipped walk, tile for tile
// (tools/check_binwalk.cpp verifies this on the host over random and adversarial inputs).
// Every translation unit that includes this header is compiled with -ffp-contract=off.
#pragma once
#include "isect_walk.hpp"

namespace gsx {

struct WalkPrep {
    int x0, y0, x1, y1; // tile rectangle [x0,x1) x [y0,y1), clamped to the grid; the walk never leaves it
    bool any;           // false: the Gaussian touches no tile
    bool ellipse, alongY;
    float B, coeff, disc, t, pu, pv, bmin_u, bmax_u, bmin_v, bmax_v, u_at_vmin, u_at_vmax;
};

__host__ __device__ __forceinline__ WalkPrep walk_prepare(float mx, float my, float rx, float ry, bool has_conic, float A,
                                                          float B, float C, float opacity, uint32_t tile_size,
                                                          uint32_t tile_w, uint32_t tile_h)
{
    WalkPrep p;
    p.any = false; p.ellipse = has_conic; p.alongY = false;
    p.x0 = p.y0 = p.x1 = p.y1 = 0;
    p.B = p.coeff = p.disc = p.t = p.pu = p.pv = p.bmin_u = p.bmax_u = p.bmin_v = p.bmax_v = p.u_at_vmin = p.u_at_vmax = 0.0f;
    const float ts     = (float)tile_size;
    const bool ts_pow2 = (tile_size & (tile_size - 1u)) == 0u;
    const float ts_inv = 1.0f / ts;
    auto div_ts        = [&](float x) { return ts_pow2 ? x * ts_inv : x / ts; };
    if (has_conic) {
        const float disc = B * B - A * C;
        float t          = 2.0f * det_logf(opacity * 255.0f);
        const float tmax = kGaussianExtend * kGaussianExtend;
        if (t > tmax) t = tmax;
        if (!(t > 0.0f) || !(disc < 0.0f)) return p;
        const float s  = -t / disc;
        const float ex = sqrtf(s * C), ey = sqrtf(s * A);
        const float bminx = mx - ex, bmaxx = mx + ex, bminy = my - ey, bmaxy = my + ey;
        const float bx_c = B * ex / C, by_a = B * ey / A;
        const float y_at_xmin = my + bx_c, y_at_xmax = my - bx_c;
        const float x_at_ymin = mx + by_a, x_at_ymax = mx - by_a;
        p.x0 = clampi(f2i_trunc_sat(div_ts(bminx)), 0, (int)tile_w);
        p.y0 = clampi(f2i_trunc_sat(div_ts(bminy)), 0, (int)tile_h);
        p.x1 = clampi(f2i_trunc_sat(div_ts(bmaxx) + 1.0f), 0, (int)tile_w);
        p.y1 = clampi(f2i_trunc_sat(div_ts(bmaxy) + 1.0f), 0, (int)tile_h);
        const int yspan = p.y1 - p.y0, xspan = p.x1 - p.x0;
        if (yspan <= 0 || xspan <= 0) return p;
        p.any    = true;
        p.alongY = yspan < xspan;
        p.B = B; p.disc = disc; p.t = t;
        p.coeff = p.alongY ? A : C;
        p.pu = p.alongY ? my : mx; p.pv = p.alongY ? mx : my;
        p.bmin_u = p.alongY ? bminy : bminx; p.bmax_u = p.alongY ? bmaxy : bmaxx;
        p.bmin_v = p.alongY ? bminx : bminy; p.bmax_v = p.alongY ? bmaxx : bmaxy;
        p.u_at_vmin = p.alongY ? y_at_xmin : x_at_ymin;
        p.u_at_vmax = p.alongY ? y_at_xmax : x_at_ymax;
        return p;
    }
    const float tx = div_ts(mx), ty = div_ts(my), trx = div_ts(rx), try_ = div_ts(ry);
    p.x0  = clampi(f2i_trunc_sat(floorf(tx - trx)), 0, (int)tile_w);
    p.y0  = clampi(f2i_trunc_sat(floorf(ty - try_)), 0, (int)tile_h);
    p.x1  = clampi(f2i_trunc_sat(ceilf(tx + trx)), 0, (int)tile_w);
    p.y1  = clampi(f2i_trunc_sat(ceilf(ty + try_)), 0, (int)tile_h);
    p.any = p.x1 > p.x0 && p.y1 > p.y0;
    return p;
}

// Visits the tiles of the walk that lie in [cx0,cx1) x [cy0,cy1): emit(tile_x, tile_y). Returns their number.
template <typename Emit>
__host__ __device__ __forceinline__ int32_t walk_clipped(const WalkPrep &p, uint32_t tile_size, int cx0, int cy0, int cx1,
                                                         int cy1, Emit &&emit)
{
    if (!p.any) return 0;
    int32_t count = 0;
    if (!p.ellipse) {
        const int xa = p.x0 > cx0 ? p.x0 : cx0, xb = p.x1 < cx1 ? p.x1 : cx1;
        const int ya = p.y0 > cy0 ? p.y0 : cy0, yb = p.y1 < cy1 ? p.y1 : cy1;
        for (int y = ya; y < yb; ++y)
            for (int x = xa; x < xb; ++x) {
                emit(x, y);
                ++count;
            }
        return count;
    }
    const float ts     = (float)tile_size;
    const bool ts_pow2 = (tile_size & (tile_size - 1u)) == 0u;
    const float ts_inv = 1.0f / ts;
    auto div_ts        = [&](float x) { return ts_pow2 ? x * ts_inv : x / ts; };
    const int u0 = p.alongY ? p.y0 : p.x0, u1 = p.alongY ? p.y1 : p.x1;
    const int v0 = p.alongY ? p.x0 : p.y0, v1 = p.alongY ? p.x1 : p.y1;
    const int cu0 = p.alongY ? cy0 : cx0, cu1 = p.alongY ? cy1 : cx1;
    const int cv0 = p.alongY ? cx0 : cy0, cv1 = p.alongY ? cx1 : cy1;
    const int us = u0 > cu0 ? u0 : cu0, ue = u1 < cu1 ? u1 : cu1;
    if (us >= ue) return 0;
    if ((v0 > cv0 ? v0 : cv0) >= (v1 < cv1 ? v1 : cv1)) return 0;

    float hi_lo = p.bmax_v, hi_hi = p.bmin_v; // "empty" interval
    float lo_lo, lo_hi;
    float line_lo = (float)us * ts;
    if (us == u0) {
        if (p.bmin_u <= line_lo) ellipse_cut(p.B, p.coeff, p.disc, p.t, p.pu, p.pv, line_lo, lo_lo, lo_hi);
        else { lo_lo = hi_lo; lo_hi = hi_hi; }
    } else {
        // state after slab us - 1: the cut at the nearest line k * ts <= bmax_u with u0 < k <= us (walk_tiles takes a cut
        // at the upper line of a slab only then), or the empty interval when no such line exists
        for (int k = us; k > u0; --k) {
            const float line = (float)k * ts;
            if (line <= p.bmax_u) {
                ellipse_cut(p.B, p.coeff, p.disc, p.t, p.pu, p.pv, line, hi_lo, hi_hi);
                break;
            }
        }
        lo_lo = hi_lo; lo_hi = hi_hi;
    }
    for (int u = us; u < ue; ++u) {
        const float line_hi = line_lo + ts;
        if (line_hi <= p.bmax_u) ellipse_cut(p.B, p.coeff, p.disc, p.t, p.pu, p.pv, line_hi, hi_lo, hi_hi);
        const float vmin = (line_lo <= p.u_at_vmin && p.u_at_vmin < line_hi) ? p.bmin_v : fminf(lo_lo, hi_lo);
        const float vmax = (line_lo <= p.u_at_vmax && p.u_at_vmax < line_hi) ? p.bmax_v : fmaxf(lo_hi, hi_hi);
        int tv0          = clampi(f2i_trunc_sat(div_ts(vmin)), v0, v1);
        int tv1          = clampi(f2i_trunc_sat(div_ts(vmax) + 1.0f), v0, v1);
        if (tv0 < cv0) tv0 = cv0;
        if (tv1 > cv1) tv1 = cv1;
        for (int v = tv0; v < tv1; ++v) {
            if (p.alongY) emit(v, u);
            else emit(u, v);
            ++count;
        }
        lo_lo   = hi_lo;
        lo_hi   = hi_hi;
        line_lo = line_hi;
    }
    return count;
}

} // namespace gsx
