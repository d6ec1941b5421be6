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

// The part of the walk inside [cx0,cx1) x [cy0,cy1), slab by slab: span(alongY, u, tv0, tv1) = the tiles (x, y) = (v, u)
// [alongY] or (u, v) for v in [tv0, tv1) (an empty span, tv1 <= tv0, may be reported).
template <typename Span>
__host__ __device__ __forceinline__ void walk_clipped_spans(const WalkPrep &p, uint32_t tile_size, int cx0, int cy0, int cx1,
                                                            int cy1, Span &&span)
{
    if (!p.any) return;
    if (!p.ellipse) {
        const int xa = p.x0 > cx0 ? p.x0 : cx0, xb = p.x1 < cx1 ? p.x1 : cx1;
        const int ya = p.y0 > cy0 ? p.y0 : cy0, yb = p.y1 < cy1 ? p.y1 : cy1;
        for (int y = ya; y < yb; ++y) span(true, y, xa, xb);
        return;
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
    if (us >= ue) return;
    if ((v0 > cv0 ? v0 : cv0) >= (v1 < cv1 ? v1 : cv1)) return;

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
        span(p.alongY, u, tv0, tv1);
        lo_lo   = hi_lo;
        lo_hi   = hi_hi;
        line_lo = line_hi;
    }
}

// Visits the tiles of the walk that lie in [cx0,cx1) x [cy0,cy1): emit(tile_x, tile_y). Returns their number.
template <typename Emit>
__host__ __device__ __forceinline__ int32_t walk_clipped(const WalkPrep &p, uint32_t tile_size, int cx0, int cy0, int cx1,
                                                         int cy1, Emit &&emit)
{
    int32_t count = 0;
    walk_clipped_spans(p, tile_size, cx0, cy0, cx1, cy1, [&](bool alongY, int u, int tv0, int tv1) {
        for (int v = tv0; v < tv1; ++v) {
            if (alongY) emit(v, u);
            else emit(u, v);
            ++count;
        }
    });
    return count;
}

// ---- a block of tiles as one 64-bit word (isect_binned.hip, round 5) ---------------------------------------------------------
// A block is kx x ky bins of bw x bh tiles, W = kx * bw tiles wide, at most 64 tiles: bit dy * W + dx.
__host__ __device__ __forceinline__ uint64_t low_bits64(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

// The tiles of the walk inside the block whose first tile is (cx0, cy0), clipped to [cx0,cx1) x [cy0,cy1). A slab's span
// enters the word as a run of bits (a row of the block) or as a run of `col` = one bit per row of the block, in column 0.
__host__ __device__ __forceinline__ uint64_t block_mask(const WalkPrep &p, uint32_t tile_size, uint32_t tile_w, int cx0, int cy0,
                                                        int cx1, int cy1, uint32_t W, uint64_t col, const uint8_t *tmask)
{
    uint64_t M = 0;
    walk_clipped_spans(p, tile_size, cx0, cy0, cx1, cy1, [&](bool alongY, int uu, int tv0, int tv1) {
        if (tv1 <= tv0) return;
        if (alongY) // tiles x in [tv0, tv1) at y = uu
            M |= (low_bits64((uint32_t)(tv1 - cx0)) ^ low_bits64((uint32_t)(tv0 - cx0))) << ((uint32_t)(uu - cy0) * W);
        else // tiles y in [tv0, tv1) at x = uu
            M |= (col & (low_bits64((uint32_t)(tv1 - cy0) * W) ^ low_bits64((uint32_t)(tv0 - cy0) * W))) << (uu - cx0);
    });
    if (tmask) // (rare) tiles the caller masked out
        for (uint64_t left = M; left;) {
            const int b = __builtin_ctzll(left);
            left &= left - 1ull;
            if (!tmask[(size_t)(cy0 + b / (int)W) * tile_w + (size_t)(cx0 + b % (int)W)]) M &= ~(1ull << b);
        }
    return M;
}

// tiles of bin (i, j) of the block, as the 16-bit mask the sort kernel deals from: bit y' * bw + x'
__host__ __device__ __forceinline__ uint32_t block_bin_mask(uint64_t M, uint32_t bw, uint32_t bh, uint32_t kx, uint32_t i, uint32_t j)
{
    const uint32_t W = kx * bw, row_bits = (1u << bw) - 1u;
    uint32_t m = 0;
    for (uint32_t y = 0; y < bh; ++y) m |= ((uint32_t)(M >> ((j * bh + y) * W + i * bw)) & row_bits) << (y * bw);
    return m;
}

} // namespace gsx
