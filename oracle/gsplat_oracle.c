/* gsplat_oracle.c — CPU restatement of the reference algorithm for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (gsplat_amd/) may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the timed CPU baseline — never as the thing shipped.
 *
 * Plain C (gcc, OpenMP). Each function cites the reference lines it follows. The tile-intersection
 * functions are compiled with -ffp-contract=off so that their float operations are single IEEE
 * operations — the HIP kernels do the same, which makes the INTEGER outputs bit-comparable.
 *
 * Pinning: oracle/pin_against_reference.py (run where /root/reference exists) checks these
 * functions against the reference's own Python implementation (gsplat/cuda/_torch_impl.py) and
 * writes tests/golden/ fixtures; tests/test_oracle_golden.py re-checks them anywhere.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* number of OpenMP threads used by the parallel loops below (bench.py's cpu_baseline bounds it) */
void gso_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

#define ALPHA_THRESHOLD (1.0f / 255.0f) /* gsplat/cuda/include/Common.h:97 */
#define GAUSSIAN_EXTEND 3.33f           /* Common.h:99 */
#define MAX_ALPHA 0.99f                 /* Common.h:105 */
#define TRANSMITTANCE_THRESHOLD 1e-4f   /* Common.h:106 */
#define MIN_ONE_MINUS_ALPHA 1e-6f       /* Common.h:114 */

/* ------------------------------------------------------------------------------------------
 * deterministic natural log: the same operation sequence as gsx::det_logf (csrc/common.hpp).
 * The reference uses the fast-math __logf at these places (IntersectTile.cu:303,
 * ProjectionEWA3DGSFused.cu:180); any ~1 ulp log is within its behaviour.
 * ---------------------------------------------------------------------------------------- */
float gso_det_logf(float x)
{
    union { float f; uint32_t u; } v;
    v.f   = x;
    int e = (int)((v.u >> 23) & 0xFF) - 127;
    v.u   = (v.u & 0x007FFFFFu) | 0x3F800000u;
    float m = v.f;
    if (m > 1.41421356f) { m *= 0.5f; e += 1; }
    const float s  = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = 0.2222222222f;
    p = fmaf(p, s2, 0.2857142857f);
    p = fmaf(p, s2, 0.4f);
    p = fmaf(p, s2, 0.6666666667f);
    p = fmaf(p, s2, 2.0f);
    const float lm = p * s;
    return fmaf((float)e, 0.69314718056f, lm);
}

void gso_det_logf_array(const float *in, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = gso_det_logf(in[i]);
}

static int f2i_trunc_sat(float x)
{
    if (!(x == x)) return 0;
    if (x >= 2.0e9f) return 2000000000;
    if (x <= -2.0e9f) return -2000000000;
    return (int)x;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void ellipse_cut(float B, float coeff, float disc, float t, float pu, float pv, float coord, float *lo, float *hi)
{
    const float h    = coord - pu;
    const float arg  = disc * h * h + t * coeff;
    const float root = sqrtf(arg > 0.0f ? arg : 0.0f);
    const float mbh  = -B * h;
    *lo = (mbh - root) / coeff + pv;
    *hi = (mbh + root) / coeff + pv;
}

/* Tiles touched by one Gaussian. Follows IntersectTile.cu:288-373 + helpers :83-207 (ellipse /
 * AccuTile-SNUGBOX mode, when has_conic) and :374-463 (AABB mode). Writes tile ids (row-major
 * tile index) into out (may be NULL) in the reference's emission order; returns the count. */
static int32_t walk_tiles(float mx, float my, float rx, float ry, int has_conic, float A, float B, float C,
                          float opacity, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int64_t *out)
{
    const float ts = (float)tile_size;
    int32_t count  = 0;
    if (has_conic) {
        const float disc = B * B - A * C;
        float t          = 2.0f * gso_det_logf(opacity * 255.0f);
        const float tmax = GAUSSIAN_EXTEND * GAUSSIAN_EXTEND;
        if (t > tmax) t = tmax;
        if (!(t > 0.0f) || !(disc < 0.0f)) return 0;
        const float s  = -t / disc;
        const float ex = sqrtf(s * C), ey = sqrtf(s * A);
        const float bminx = mx - ex, bmaxx = mx + ex, bminy = my - ey, bmaxy = my + ey;
        const float bx_c = B * ex / C, by_a = B * ey / A;
        const float y_at_xmin = my + bx_c, y_at_xmax = my - bx_c;
        const float x_at_ymin = mx + by_a, x_at_ymax = mx - by_a;
        const int rminx = clampi(f2i_trunc_sat(bminx / ts), 0, (int)tile_w);
        const int rminy = clampi(f2i_trunc_sat(bminy / ts), 0, (int)tile_h);
        const int rmaxx = clampi(f2i_trunc_sat(bmaxx / ts + 1.0f), 0, (int)tile_w);
        const int rmaxy = clampi(f2i_trunc_sat(bmaxy / ts + 1.0f), 0, (int)tile_h);
        const int yspan = rmaxy - rminy, xspan = rmaxx - rminx;
        if (yspan <= 0 || xspan <= 0) return 0;
        const int alongY = yspan < xspan;
        const int u0 = alongY ? rminy : rminx, u1 = alongY ? rmaxy : rmaxx;
        const int v0 = alongY ? rminx : rminy, v1 = alongY ? rmaxx : rmaxy;
        const float pu = alongY ? my : mx, pv = alongY ? mx : my;
        const float bmin_u = alongY ? bminy : bminx, bmax_u = alongY ? bmaxy : bmaxx;
        const float bmin_v = alongY ? bminx : bminy, bmax_v = alongY ? bmaxx : bmaxy;
        const float u_at_vmin = alongY ? y_at_xmin : x_at_ymin;
        const float u_at_vmax = alongY ? y_at_xmax : x_at_ymax;
        const float coeff     = alongY ? A : C;
        float hi_lo = bmax_v, hi_hi = bmin_v, lo_lo, lo_hi;
        float line_lo = (float)u0 * ts;
        if (bmin_u <= line_lo) ellipse_cut(B, coeff, disc, t, pu, pv, line_lo, &lo_lo, &lo_hi);
        else { lo_lo = hi_lo; lo_hi = hi_hi; }
        for (int u = u0; u < u1; ++u) {
            const float line_hi = line_lo + ts;
            if (line_hi <= bmax_u) ellipse_cut(B, coeff, disc, t, pu, pv, line_hi, &hi_lo, &hi_hi);
            const float vmin = (line_lo <= u_at_vmin && u_at_vmin < line_hi) ? bmin_v : fminf(lo_lo, hi_lo);
            const float vmax = (line_lo <= u_at_vmax && u_at_vmax < line_hi) ? bmax_v : fmaxf(lo_hi, hi_hi);
            const int tv0    = clampi(f2i_trunc_sat(vmin / ts), v0, v1);
            const int tv1    = clampi(f2i_trunc_sat(vmax / ts + 1.0f), v0, v1);
            for (int v = tv0; v < tv1; ++v) {
                if (out) out[count] = alongY ? (int64_t)u * tile_w + v : (int64_t)v * tile_w + u;
                ++count;
            }
            lo_lo = hi_lo; lo_hi = hi_hi; line_lo = line_hi;
        }
        return count;
    }
    const float tx = mx / ts, ty = my / ts, trx = rx / ts, try_ = ry / ts;
    const int x0 = clampi(f2i_trunc_sat(floorf(tx - trx)), 0, (int)tile_w);
    const int y0 = clampi(f2i_trunc_sat(floorf(ty - try_)), 0, (int)tile_h);
    const int x1 = clampi(f2i_trunc_sat(ceilf(tx + trx)), 0, (int)tile_w);
    const int y1 = clampi(f2i_trunc_sat(ceilf(ty + try_)), 0, (int)tile_h);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            if (out) out[count] = (int64_t)y * tile_w + x;
            ++count;
        }
    return count;
}

static uint32_t bits_for_count(uint64_t count)
{ /* MathUtils.h:25-35 */
    uint32_t b = 0;
    if (count <= 1) return 0;
    uint64_t v = count - 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

/* isect_tiles pass 1 (IntersectTile.cu:214-464 with cum_tiles_per_gauss == nullptr). */
void gso_isect_count(const float *means2d, const int32_t *radii, const float *conics, const float *opacities,
                     int64_t rows, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *tiles_per_gauss)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i) {
        const float rx = (float)radii[2 * i], ry = (float)radii[2 * i + 1];
        if (rx <= 0.0f || ry <= 0.0f) { tiles_per_gauss[i] = 0; continue; }
        const int hc = conics && opacities;
        tiles_per_gauss[i] = walk_tiles(means2d[2 * i], means2d[2 * i + 1], rx, ry, hc, hc ? conics[3 * i] : 0.f,
                                        hc ? conics[3 * i + 1] : 0.f, hc ? conics[3 * i + 2] : 0.f,
                                        hc ? opacities[i] : 0.f, tile_size, tile_w, tile_h, NULL);
    }
}

/* isect_tiles pass 2: emits (key, flatten id); cum = INCLUSIVE cumsum of tiles_per_gauss.
 * key = image << (32 + tile_bits) | tile << 32 | bits(float depth)   (IntersectTile.cu:266-286). */
int gso_isect_emit(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                   const float *opacities, const int64_t *image_ids, const int64_t *cum, int64_t rows,
                   uint32_t n_per_image, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                   int64_t *isect_ids, int32_t *flatten_ids)
{
    const uint32_t tile_bits = bits_for_count((uint64_t)tile_w * tile_h), image_bits = bits_for_count(n_images);
    if (tile_bits + image_bits > 32) return -4;
    int64_t max_tiles = (int64_t)tile_w * tile_h;
#pragma omp parallel
    {
        int64_t *buf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(max_tiles > 0 ? max_tiles : 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < rows; ++i) {
            const float rx = (float)radii[2 * i], ry = (float)radii[2 * i + 1];
            if (rx <= 0.0f || ry <= 0.0f) continue;
            const int hc = conics && opacities;
            const int32_t n = walk_tiles(means2d[2 * i], means2d[2 * i + 1], rx, ry, hc, hc ? conics[3 * i] : 0.f,
                                         hc ? conics[3 * i + 1] : 0.f, hc ? conics[3 * i + 2] : 0.f,
                                         hc ? opacities[i] : 0.f, tile_size, tile_w, tile_h, buf);
            const int64_t iid = image_ids ? image_ids[i] : i / (n_per_image ? n_per_image : 1);
            union { float f; uint32_t u; } d;
            d.f = depths[i];
            int64_t cur = i == 0 ? 0 : cum[i - 1];
            for (int32_t k = 0; k < n; ++k) {
                isect_ids[cur] = (int64_t)(((uint64_t)iid << (32 + tile_bits)) | ((uint64_t)buf[k] << 32) | (uint64_t)d.u);
                flatten_ids[cur] = (int32_t)i;
                ++cur;
            }
        }
        free(buf);
    }
    return 0;
}

/* isect_offset_encode (IntersectTile.cu:925-988 / _torch_impl.py:455-481). */
void gso_isect_offsets(const int64_t *sorted_ids, int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h,
                       int32_t *offsets)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    const uint32_t tile_bits = bits_for_count((uint64_t)n_tiles);
    const uint64_t tmask = tile_bits >= 32 ? 0xFFFFFFFFull : ((1ull << tile_bits) - 1ull);
    int64_t k = 0;
    for (int64_t i = 0; i < n_isects; ++i) {
        const uint64_t key = (uint64_t)sorted_ids[i] >> 32;
        const int64_t lin  = (int64_t)(key >> tile_bits) * n_tiles + (int64_t)(key & tmask);
        for (; k <= lin && k < total; ++k) offsets[k] = (int32_t)i;
    }
    for (; k < total; ++k) offsets[k] = (int32_t)n_isects;
}

/* ------------------------------------------------------------------------------------------
 * rasterize_to_pixels (3DGS) forward. Follows RasterizeToPixels3DGSDevice.cuh:44-100 and
 * RasterizeToPixels3DGSSerialBatchFwd.cu:128-295; equals _torch_impl.py:713-924 (accumulate /
 * _rasterize_to_pixels) on the contributing set defined by RasterizeToIndices3DGSSerialBatch.cu.
 * ---------------------------------------------------------------------------------------- */
void gso_raster3d_fwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                      const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                      const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width,
                      uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *render_colors,
                      float *render_alphas, int32_t *last_ids)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int masked = masks && !masks[blk];
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        float *acc = (float *)malloc(sizeof(float) * cdim);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                if (masked) {
                    for (uint32_t k = 0; k < cdim; ++k) render_colors[pix * cdim + k] = bg ? bg[k] : 0.0f;
                    render_alphas[pix] = 0.0f;
                    last_ids[pix]      = 0;
                    continue;
                }
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                float T = 1.0f;
                int32_t cur = 0;
                for (uint32_t k = 0; k < cdim; ++k) acc[k] = 0.0f;
                for (int32_t idx = start; idx < end; ++idx) {
                    const int32_t g = flatten_ids[idx];
                    const float dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
                    const float a = conics[3 * (size_t)g], b = conics[3 * (size_t)g + 1], c = conics[3 * (size_t)g + 2];
                    const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    const float vis   = expf(-sigma);
                    const float alpha = fminf(MAX_ALPHA, opacities[g] * vis);
                    if (sigma < 0.0f || alpha < ALPHA_THRESHOLD) continue;
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= TRANSMITTANCE_THRESHOLD) break; /* exclusive stop */
                    const float w = alpha * T;
                    for (uint32_t k = 0; k < cdim; ++k) acc[k] += colors[(size_t)g * cdim + k] * w;
                    cur = idx;
                    T   = next_T;
                }
                for (uint32_t k = 0; k < cdim; ++k) render_colors[pix * cdim + k] = bg ? acc[k] + T * bg[k] : acc[k];
                render_alphas[pix] = 1.0f - T;
                last_ids[pix]      = cur;
            }
        free(acc);
    }
}

/* rasterize_to_pixels (3DGS) backward. Follows RasterizeToPixels3DGSDevice.cuh:105-173 and
 * RasterizeToPixels3DGSSerialBatchBwd.cu:130-318. Per-sample math in fp32 (as the reference),
 * the sums over pixels in fp64 (the reference's atomics are fp32 in unspecified order; fp64 sums
 * are the order-independent value they approximate). Outputs are fp64 arrays, zero-filled here. */
void gso_raster3d_bwd(const float *means2d, const float *conics, const float *colors, const float *opacities,
                      const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                      const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                      const float *v_render_colors, const float *v_render_alphas, uint32_t n_images,
                      uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                      uint32_t tile_w, uint32_t tile_h, int64_t n_rows, double *v_means2d_abs, double *v_means2d,
                      double *v_conics, double *v_colors, double *v_opacities)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    if (v_means2d_abs) memset(v_means2d_abs, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_means2d, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_conics, 0, sizeof(double) * 3 * (size_t)n_rows);
    memset(v_colors, 0, sizeof(double) * (size_t)cdim * (size_t)n_rows);
    memset(v_opacities, 0, sizeof(double) * (size_t)n_rows);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        if (masks && !masks[blk]) continue;
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        if (end <= start) continue;
        float *buffer = (float *)malloc(sizeof(float) * cdim);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                const float T_final = 1.0f - render_alphas[pix];
                float T = T_final;
                const int32_t bin_final = last_ids[pix];
                const float *v_c = v_render_colors + pix * cdim;
                const float v_a  = v_render_alphas[pix];
                for (uint32_t k = 0; k < cdim; ++k) buffer[k] = 0.0f;
                int32_t hi = bin_final < end - 1 ? bin_final : end - 1;
                for (int32_t idx = hi; idx >= start; --idx) {
                    const int32_t g = flatten_ids[idx];
                    const float dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
                    const float a = conics[3 * (size_t)g], b = conics[3 * (size_t)g + 1], c = conics[3 * (size_t)g + 2];
                    const float opac  = opacities[g];
                    const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    const float vis   = expf(-sigma);
                    const float alpha = fminf(MAX_ALPHA, opac * vis);
                    if (sigma < 0.0f || alpha < ALPHA_THRESHOLD) continue;
                    const float ra = 1.0f / fmaxf(MIN_ONE_MINUS_ALPHA, 1.0f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    float v_alpha = 0.0f;
                    const float *col = colors + (size_t)g * cdim;
                    for (uint32_t k = 0; k < cdim; ++k) {
#pragma omp atomic
                        v_colors[(size_t)g * cdim + k] += (double)(fac * v_c[k]);
                        v_alpha += (col[k] * T - buffer[k] * ra) * v_c[k];
                    }
                    v_alpha += T_final * ra * v_a;
                    if (bg) {
                        float accum = 0.0f;
                        for (uint32_t k = 0; k < cdim; ++k) accum += bg[k] * v_c[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= MAX_ALPHA) {
                        const float v_sigma = -opac * vis * v_alpha;
                        const float vx = v_sigma * (a * dx + b * dy), vy = v_sigma * (b * dx + c * dy);
#pragma omp atomic
                        v_conics[3 * (size_t)g + 0] += (double)(0.5f * v_sigma * dx * dx);
#pragma omp atomic
                        v_conics[3 * (size_t)g + 1] += (double)(v_sigma * dx * dy);
#pragma omp atomic
                        v_conics[3 * (size_t)g + 2] += (double)(0.5f * v_sigma * dy * dy);
#pragma omp atomic
                        v_means2d[2 * (size_t)g + 0] += (double)vx;
#pragma omp atomic
                        v_means2d[2 * (size_t)g + 1] += (double)vy;
                        if (v_means2d_abs) {
#pragma omp atomic
                            v_means2d_abs[2 * (size_t)g + 0] += (double)fabsf(vx);
#pragma omp atomic
                            v_means2d_abs[2 * (size_t)g + 1] += (double)fabsf(vy);
                        }
#pragma omp atomic
                        v_opacities[g] += (double)(vis * v_alpha);
                    }
                    for (uint32_t k = 0; k < cdim; ++k) buffer[k] += col[k] * fac;
                }
            }
        free(buffer);
    }
}

/* The same backward with the PER-SAMPLE math in fp64 as well (inputs and the forward state it starts from are the fp32
 * values): what every fp32 evaluation of these sums approximates. tests/test_gpu_pipeline.py measures, per element, how far
 * the fp32 oracle above AND the GPU kernels are from it - two fp32 evaluations held to the same band. The contributor set is
 * the fp32 one (`last_ids`, and the alpha test on the fp64 alpha: a pair exactly on the 1/255 edge may differ). */
/* optional second output of gso_raster3d_bwd_f64: [n_rows][6 + cdim] sums of the ABSOLUTE values of the terms of every
 * gradient component (v_mean2d 2 | v_conic 3 | v_opacity 1 | v_color cdim) - the conditioning of each sum: an evaluation
 * whose samples carry a relative error eps is off by at most eps times this. Set before the call, cleared by it. */
static double *g_abs_terms = NULL;
void gso_set_abs_terms(double *p) { g_abs_terms = p; }

void gso_raster3d_bwd_f64(const float *means2d, const float *conics, const float *colors, const float *opacities,
                          const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                          const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                          const float *v_render_colors, const float *v_render_alphas, uint32_t n_images,
                          uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                          uint32_t tile_w, uint32_t tile_h, int64_t n_rows, double *v_means2d_abs, double *v_means2d,
                          double *v_conics, double *v_colors, double *v_opacities)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    if (v_means2d_abs) memset(v_means2d_abs, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_means2d, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_conics, 0, sizeof(double) * 3 * (size_t)n_rows);
    memset(v_colors, 0, sizeof(double) * (size_t)cdim * (size_t)n_rows);
    memset(v_opacities, 0, sizeof(double) * (size_t)n_rows);
    double *abs_terms = g_abs_terms;
    g_abs_terms       = NULL;
    const uint32_t KA = 6 + cdim;
    if (abs_terms) memset(abs_terms, 0, sizeof(double) * (size_t)KA * (size_t)n_rows);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        if (masks && !masks[blk]) continue;
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        if (end <= start) continue;
        double *buffer = (double *)malloc(sizeof(double) * cdim);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                const double px = (double)ox + 0.5, py = (double)oy + 0.5;
                const double T_final = 1.0 - (double)render_alphas[pix];
                double T = T_final;
                const int32_t bin_final = last_ids[pix];
                const float *v_c = v_render_colors + pix * cdim;
                const double v_a = v_render_alphas[pix];
                for (uint32_t k = 0; k < cdim; ++k) buffer[k] = 0.0;
                int32_t hi = bin_final < end - 1 ? bin_final : end - 1;
                for (int32_t idx = hi; idx >= start; --idx) {
                    const int32_t g = flatten_ids[idx];
                    const double dx = (double)means2d[2 * (size_t)g] - px, dy = (double)means2d[2 * (size_t)g + 1] - py;
                    const double a = conics[3 * (size_t)g], b = conics[3 * (size_t)g + 1], c = conics[3 * (size_t)g + 2];
                    const double opac  = opacities[g];
                    const double sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    const double vis   = exp(-sigma);
                    const double alpha = fmin((double)MAX_ALPHA, opac * vis);
                    if (sigma < 0.0 || alpha < (double)ALPHA_THRESHOLD) continue;
                    const double ra = 1.0 / fmax((double)MIN_ONE_MINUS_ALPHA, 1.0 - alpha);
                    T *= ra;
                    const double fac = alpha * T;
                    double v_alpha = 0.0;
                    const float *col = colors + (size_t)g * cdim;
                    for (uint32_t k = 0; k < cdim; ++k) {
#pragma omp atomic
                        v_colors[(size_t)g * cdim + k] += fac * (double)v_c[k];
                        if (abs_terms) {
#pragma omp atomic
                            abs_terms[(size_t)g * KA + 6 + k] += fabs(fac * (double)v_c[k]);
                        }
                        v_alpha += ((double)col[k] * T - buffer[k] * ra) * (double)v_c[k];
                    }
                    v_alpha += T_final * ra * v_a;
                    if (bg) {
                        double accum = 0.0;
                        for (uint32_t k = 0; k < cdim; ++k) accum += (double)bg[k] * (double)v_c[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= (double)MAX_ALPHA) {
                        const double v_sigma = -opac * vis * v_alpha;
                        const double vx = v_sigma * (a * dx + b * dy), vy = v_sigma * (b * dx + c * dy);
#pragma omp atomic
                        v_conics[3 * (size_t)g + 0] += 0.5 * v_sigma * dx * dx;
#pragma omp atomic
                        v_conics[3 * (size_t)g + 1] += v_sigma * dx * dy;
#pragma omp atomic
                        v_conics[3 * (size_t)g + 2] += 0.5 * v_sigma * dy * dy;
#pragma omp atomic
                        v_means2d[2 * (size_t)g + 0] += vx;
#pragma omp atomic
                        v_means2d[2 * (size_t)g + 1] += vy;
                        if (v_means2d_abs) {
#pragma omp atomic
                            v_means2d_abs[2 * (size_t)g + 0] += fabs(vx);
#pragma omp atomic
                            v_means2d_abs[2 * (size_t)g + 1] += fabs(vy);
                        }
#pragma omp atomic
                        v_opacities[g] += vis * v_alpha;
                        if (abs_terms) {
                            double *at = abs_terms + (size_t)g * KA;
#pragma omp atomic
                            at[0] += fabs(vx);
#pragma omp atomic
                            at[1] += fabs(vy);
#pragma omp atomic
                            at[2] += fabs(0.5 * v_sigma * dx * dx);
#pragma omp atomic
                            at[3] += fabs(v_sigma * dx * dy);
#pragma omp atomic
                            at[4] += fabs(0.5 * v_sigma * dy * dy);
#pragma omp atomic
                            at[5] += fabs(vis * v_alpha);
                        }
                    }
                    for (uint32_t k = 0; k < cdim; ++k) buffer[k] += (double)col[k] * fac;
                }
            }
        free(buffer);
    }
}

/* The same backward with the SUMS in fp32 too: per (tile, Gaussian) a float running sum over the tile's pixels in raster
 * order, added to float totals - one of the many orders in which the reference's fp32 atomics may combine the same terms
 * (its kernel reduces a warp with a tree and adds the result to global memory with atomicAdd, Bwd.cu:270-318). This is the
 * "other fp32 evaluation" whose distance from gso_raster3d_bwd_f64 is the envelope of tests/test_gpu_pipeline.py's per-element
 * band. Outputs fp64 arrays holding the float totals. No absgrad. */
void gso_raster3d_bwd_f32sum(const float *means2d, const float *conics, const float *colors, const float *opacities,
                             const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                             const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                             const float *v_render_colors, const float *v_render_alphas, uint32_t n_images,
                             uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                             uint32_t tile_w, uint32_t tile_h, int64_t n_rows, double *v_means2d_abs, double *v_means2d,
                             double *v_conics, double *v_colors, double *v_opacities)
{
    (void)v_means2d_abs;
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    const uint32_t K = 6 + cdim; /* per Gaussian: v_mean2d 2 | v_conic 3 | v_opacity 1 | v_color cdim */
    float *tot = (float *)calloc((size_t)n_rows * K, sizeof(float));
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        if (masks && !masks[blk]) continue;
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        if (end <= start) continue;
        float *buffer = (float *)malloc(sizeof(float) * cdim);
        float *loc    = (float *)calloc((size_t)(end - start) * K, sizeof(float));
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                const float T_final = 1.0f - render_alphas[pix];
                float T = T_final;
                const int32_t bin_final = last_ids[pix];
                const float *v_c = v_render_colors + pix * cdim;
                const float v_a  = v_render_alphas[pix];
                for (uint32_t k = 0; k < cdim; ++k) buffer[k] = 0.0f;
                int32_t hi = bin_final < end - 1 ? bin_final : end - 1;
                for (int32_t idx = hi; idx >= start; --idx) {
                    const int32_t g = flatten_ids[idx];
                    float *acc = loc + (size_t)(idx - start) * K;
                    const float dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
                    const float a = conics[3 * (size_t)g], b = conics[3 * (size_t)g + 1], c = conics[3 * (size_t)g + 2];
                    const float opac  = opacities[g];
                    const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    const float vis   = expf(-sigma);
                    const float alpha = fminf(MAX_ALPHA, opac * vis);
                    if (sigma < 0.0f || alpha < ALPHA_THRESHOLD) continue;
                    const float ra = 1.0f / fmaxf(MIN_ONE_MINUS_ALPHA, 1.0f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    float v_alpha = 0.0f;
                    const float *col = colors + (size_t)g * cdim;
                    for (uint32_t k = 0; k < cdim; ++k) {
                        acc[6 + k] += fac * v_c[k];
                        v_alpha += (col[k] * T - buffer[k] * ra) * v_c[k];
                    }
                    v_alpha += T_final * ra * v_a;
                    if (bg) {
                        float accum = 0.0f;
                        for (uint32_t k = 0; k < cdim; ++k) accum += bg[k] * v_c[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= MAX_ALPHA) {
                        const float v_sigma = -opac * vis * v_alpha;
                        acc[0] += v_sigma * (a * dx + b * dy);
                        acc[1] += v_sigma * (b * dx + c * dy);
                        acc[2] += 0.5f * v_sigma * dx * dx;
                        acc[3] += v_sigma * dx * dy;
                        acc[4] += 0.5f * v_sigma * dy * dy;
                        acc[5] += vis * v_alpha;
                    }
                    for (uint32_t k = 0; k < cdim; ++k) buffer[k] += col[k] * fac;
                }
            }
        for (int32_t idx = start; idx < end; ++idx) {
            const float *acc = loc + (size_t)(idx - start) * K;
            float *t         = tot + (size_t)flatten_ids[idx] * K;
            for (uint32_t k = 0; k < K; ++k)
                if (acc[k] != 0.0f) {
#pragma omp atomic
                    t[k] += acc[k];
                }
        }
        free(loc);
        free(buffer);
    }
    for (int64_t g = 0; g < n_rows; ++g) {
        const float *t = tot + (size_t)g * K;
        v_means2d[2 * g] = t[0]; v_means2d[2 * g + 1] = t[1];
        v_conics[3 * g] = t[2]; v_conics[3 * g + 1] = t[3]; v_conics[3 * g + 2] = t[4];
        v_opacities[g] = t[5];
        for (uint32_t k = 0; k < cdim; ++k) v_colors[(size_t)g * cdim + k] = t[6 + k];
    }
    free(tot);
}

/* rasterize_to_indices (RasterizeToIndices3DGSSerialBatch.cu:128-192): the (gaussian, pixel, image)
 * triples that contribute, in the order the reference's torch rasterizer consumes them. Used only
 * to drive the reference's own `accumulate` when pinning this oracle. Two-pass: call with out
 * pointers NULL to get the count. */
int64_t gso_raster3d_indices(const float *means2d, const float *conics, const float *opacities,
                             const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                             uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                             uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int64_t *gaussian_ids,
                             int64_t *pixel_ids, int64_t *image_ids)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    int64_t count = 0;
    for (int64_t blk = 0; blk < total; ++blk) {
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                float T = 1.0f;
                for (int32_t idx = start; idx < end; ++idx) {
                    const int32_t g = flatten_ids[idx];
                    const float dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
                    const float a = conics[3 * (size_t)g], b = conics[3 * (size_t)g + 1], c = conics[3 * (size_t)g + 2];
                    const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    const float alpha = fminf(MAX_ALPHA, opacities[g] * expf(-sigma));
                    if (sigma < 0.0f || alpha < ALPHA_THRESHOLD) continue;
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= TRANSMITTANCE_THRESHOLD) break;
                    if (gaussian_ids) {
                        gaussian_ids[count] = g % (int64_t)n_per_image;
                        pixel_ids[count]    = (int64_t)oy * width + ox;
                        image_ids[count]    = img;
                    }
                    ++count;
                    T = next_T;
                }
            }
    }
    return count;
}

/* ==========================================================================================
 * 2DGS (surfel) compositing. Follows RasterizeToPixels2DGSSerialBatchFwd.cu:43-465 and
 * RasterizeToPixels2DGSSerialBatchBwd.cu:41-700 (the distortion and median outputs exist only in
 * the CUDA kernels; colours / alphas / normals also equal _torch_impl_2dgs.py:111-334).
 * ray_transforms rows are (u_M, v_M, w_M) = rows of K*[RS0 RS1 mean_c].
 * ========================================================================================== */
#define FILTER_INV_SQUARE_2DGS 2.0f /* gsplat/cuda/include/Rasterization.h:40 */

typedef struct {
    int valid;
    float alpha, vis, opac, gw3, gw2, sx, sy, dx, dy;
    float hu[3], hv[3], rc[3], wM[3];
} S2Sample;

static S2Sample s2_eval(const float *means2d, const float *rt, const float *opacities, int32_t g, float px, float py)
{
    S2Sample s;
    memset(&s, 0, sizeof(s));
    const float *M = rt + 9 * (size_t)g;
    for (int k = 0; k < 3; ++k) {
        s.wM[k] = M[6 + k];
        s.hu[k] = px * M[6 + k] - M[k];
        s.hv[k] = py * M[6 + k] - M[3 + k];
    }
    s.rc[0] = s.hu[1] * s.hv[2] - s.hu[2] * s.hv[1];
    s.rc[1] = s.hu[2] * s.hv[0] - s.hu[0] * s.hv[2];
    s.rc[2] = s.hu[0] * s.hv[1] - s.hu[1] * s.hv[0];
    if (s.rc[2] == 0.0f) return s; /* Fwd.cu: `if (ray_cross.z == 0.0) continue;` */
    s.sx = s.rc[0] / s.rc[2];
    s.sy = s.rc[1] / s.rc[2];
    s.gw3 = s.sx * s.sx + s.sy * s.sy;
    s.dx  = means2d[2 * (size_t)g] - px;
    s.dy  = means2d[2 * (size_t)g + 1] - py;
    s.gw2 = FILTER_INV_SQUARE_2DGS * (s.dx * s.dx + s.dy * s.dy);
    const float sigma = 0.5f * fminf(s.gw3, s.gw2);
    s.opac  = opacities[g];
    s.vis   = expf(-sigma);
    s.alpha = fminf(MAX_ALPHA, s.opac * s.vis);
    s.valid = !(sigma < 0.0f || s.alpha < ALPHA_THRESHOLD);
    return s;
}

void gso_raster2d_fwd(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                      const float *normals, const float *backgrounds, const uint8_t *masks,
                      const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects,
                      uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                      uint32_t tile_h, int distloss, float *render_colors, float *render_alphas, float *render_normals,
                      float *render_distort, float *render_median, int32_t *last_ids, int32_t *median_ids)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int masked = masks && !masks[blk];
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        float *acc = (float *)malloc(sizeof(float) * cdim);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                if (masked) {
                    for (uint32_t k = 0; k < cdim; ++k) render_colors[pix * cdim + k] = bg ? bg[k] : 0.0f;
                    render_alphas[pix] = 0.0f;
                    for (int k = 0; k < 3; ++k) render_normals[pix * 3 + k] = 0.0f;
                    render_distort[pix] = 0.0f;
                    render_median[pix]  = 0.0f;
                    last_ids[pix] = 0;
                    median_ids[pix] = 0;
                    continue;
                }
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                float T = 1.0f, distort = 0.0f, accum_vis_depth = 0.0f, median_depth = 0.0f;
                float nrm[3] = {0.f, 0.f, 0.f};
                int32_t cur = 0, median_idx = 0;
                for (uint32_t k = 0; k < cdim; ++k) acc[k] = 0.0f;
                for (int32_t idx = start; idx < end; ++idx) {
                    const int32_t g = flatten_ids[idx];
                    const S2Sample s = s2_eval(means2d, ray_transforms, opacities, g, px, py);
                    if (!s.valid) continue;
                    const float next_T = T * (1.0f - s.alpha);
                    if (next_T <= TRANSMITTANCE_THRESHOLD) break;
                    const float vis = s.alpha * T;
                    const float *c = colors + (size_t)g * cdim;
                    for (uint32_t k = 0; k < cdim; ++k) acc[k] += c[k] * vis;
                    for (int k = 0; k < 3; ++k) nrm[k] += normals[3 * (size_t)g + k] * vis;
                    if (distloss) {
                        const float depth = c[cdim - 1];
                        const float d0 = vis * depth * (1.0f - T), d1 = vis * accum_vis_depth;
                        distort += 2.0f * (d0 - d1);
                        accum_vis_depth += vis * depth;
                    }
                    if (T > 0.5f) {
                        median_depth = c[cdim - 1];
                        median_idx   = idx;
                    }
                    cur = idx;
                    T   = next_T;
                }
                for (uint32_t k = 0; k < cdim; ++k) render_colors[pix * cdim + k] = bg ? acc[k] + T * bg[k] : acc[k];
                render_alphas[pix] = 1.0f - T;
                for (int k = 0; k < 3; ++k) render_normals[pix * 3 + k] = nrm[k];
                render_distort[pix] = distloss ? distort : 0.0f;
                render_median[pix]  = median_depth;
                last_ids[pix]   = cur;
                median_ids[pix] = median_idx;
            }
        free(acc);
    }
}

#define ATOMIC_ADD(dst, val) _Pragma("omp atomic") dst += (double)(val)

/* Backward; per-sample math in fp32 like the reference, sums over pixels in fp64 (see gso_raster3d_bwd).
 * v_render_distort may be NULL (distloss off). */
void gso_raster2d_bwd(const float *means2d, const float *ray_transforms, const float *colors, const float *opacities,
                      const float *normals, const float *backgrounds, const uint8_t *masks,
                      const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_colors,
                      const float *render_alphas, const int32_t *last_ids, const int32_t *median_ids,
                      const float *v_render_colors, const float *v_render_alphas, const float *v_render_normals,
                      const float *v_render_distort, const float *v_render_median, uint32_t n_images,
                      uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
                      uint32_t tile_w, uint32_t tile_h, int64_t n_rows, double *v_means2d_abs, double *v_means2d,
                      double *v_ray_transforms, double *v_colors, double *v_opacities, double *v_normals,
                      double *v_densify)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    if (v_means2d_abs) memset(v_means2d_abs, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_means2d, 0, sizeof(double) * 2 * (size_t)n_rows);
    memset(v_ray_transforms, 0, sizeof(double) * 9 * (size_t)n_rows);
    memset(v_colors, 0, sizeof(double) * (size_t)cdim * (size_t)n_rows);
    memset(v_opacities, 0, sizeof(double) * (size_t)n_rows);
    memset(v_normals, 0, sizeof(double) * 3 * (size_t)n_rows);
    memset(v_densify, 0, sizeof(double) * 2 * (size_t)n_rows);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < total; ++blk) {
        if (masks && !masks[blk]) continue;
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const float *bg = backgrounds ? backgrounds + (size_t)img * cdim : NULL;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        if (end <= start) continue;
        float *buffer = (float *)malloc(sizeof(float) * cdim);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const size_t pix = ((size_t)img * height + oy) * width + ox;
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                const float T_final = 1.0f - render_alphas[pix];
                float T = T_final;
                const int32_t bin_final = last_ids[pix], median_idx = median_ids[pix];
                const float *v_c = v_render_colors + pix * cdim, *v_n = v_render_normals + pix * 3;
                const float v_a = v_render_alphas[pix], v_median = v_render_median[pix];
                float buffer_n[3] = {0.f, 0.f, 0.f};
                for (uint32_t k = 0; k < cdim; ++k) buffer[k] = 0.0f;
                float v_distort = 0.f, accum_d = 0.f, accum_w = 0.f, accum_d_buffer = 0.f, accum_w_buffer = 0.f,
                      distort_buffer = 0.f;
                if (v_render_distort) {
                    v_distort      = v_render_distort[pix];
                    accum_d_buffer = render_colors[pix * cdim + cdim - 1];
                    accum_d        = accum_d_buffer;
                    accum_w_buffer = render_alphas[pix];
                    accum_w        = accum_w_buffer;
                }
                int32_t hi = bin_final < end - 1 ? bin_final : end - 1;
                for (int32_t idx = hi; idx >= start; --idx) {
                    const int32_t g = flatten_ids[idx];
                    const S2Sample s = s2_eval(means2d, ray_transforms, opacities, g, px, py);
                    if (!s.valid) continue;
                    const float *col = colors + (size_t)g * cdim, *nr = normals + 3 * (size_t)g;
                    if (idx == median_idx) ATOMIC_ADD(v_colors[(size_t)g * cdim + cdim - 1], v_median);
                    const float ra = 1.0f / fmaxf(MIN_ONE_MINUS_ALPHA, 1.0f - s.alpha);
                    T *= ra;
                    const float fac = s.alpha * T;
                    float v_alpha = 0.0f;
                    for (uint32_t k = 0; k < cdim; ++k) {
                        ATOMIC_ADD(v_colors[(size_t)g * cdim + k], fac * v_c[k]);
                        v_alpha += (col[k] * T - buffer[k] * ra) * v_c[k];
                    }
                    for (int k = 0; k < 3; ++k) {
                        ATOMIC_ADD(v_normals[3 * (size_t)g + k], fac * v_n[k]);
                        v_alpha += (nr[k] * T - buffer_n[k] * ra) * v_n[k];
                    }
                    v_alpha += T_final * ra * v_a;
                    if (bg) {
                        float accum = 0.0f;
                        for (uint32_t k = 0; k < cdim; ++k) accum += bg[k] * v_c[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (v_render_distort) {
                        const float depth = col[cdim - 1];
                        const float dl_dw =
                            2.0f * (2.0f * (depth * accum_w_buffer - accum_d_buffer) + (accum_d - depth * accum_w));
                        v_alpha += (dl_dw * T - distort_buffer * ra) * v_distort;
                        accum_d_buffer -= fac * depth;
                        accum_w_buffer -= fac;
                        distort_buffer += dl_dw * fac;
                        ATOMIC_ADD(v_colors[(size_t)g * cdim + cdim - 1],
                                   2.0f * fac * (2.0f - 2.0f * T - accum_w + fac) * v_distort);
                    }
                    if (s.opac * s.vis <= MAX_ALPHA) {
                        const float v_G = s.opac * v_alpha;
                        if (s.gw3 <= s.gw2) {
                            const float vsx = v_G * -s.vis * s.sx, vsy = v_G * -s.vis * s.sy;
                            const float a = vsx / s.rc[2], b = vsy / s.rc[2];
                            const float vrc[3] = {a, b, -(a * s.sx + b * s.sy)};
                            /* v_h_u = h_v x v_rc ; v_h_v = v_rc x h_u */
                            const float vhu[3] = {s.hv[1] * vrc[2] - s.hv[2] * vrc[1], s.hv[2] * vrc[0] - s.hv[0] * vrc[2],
                                                  s.hv[0] * vrc[1] - s.hv[1] * vrc[0]};
                            const float vhv[3] = {vrc[1] * s.hu[2] - vrc[2] * s.hu[1], vrc[2] * s.hu[0] - vrc[0] * s.hu[2],
                                                  vrc[0] * s.hu[1] - vrc[1] * s.hu[0]};
                            for (int k = 0; k < 3; ++k) {
                                ATOMIC_ADD(v_ray_transforms[9 * (size_t)g + k], -vhu[k]);
                                ATOMIC_ADD(v_ray_transforms[9 * (size_t)g + 3 + k], -vhv[k]);
                                ATOMIC_ADD(v_ray_transforms[9 * (size_t)g + 6 + k], px * vhu[k] + py * vhv[k]);
                            }
                            ATOMIC_ADD(v_densify[2 * (size_t)g + 0], -vhu[2] * s.wM[2]);
                            ATOMIC_ADD(v_densify[2 * (size_t)g + 1], -vhv[2] * s.wM[2]);
                        } else {
                            const float vx = v_G * (-s.vis * FILTER_INV_SQUARE_2DGS * s.dx);
                            const float vy = v_G * (-s.vis * FILTER_INV_SQUARE_2DGS * s.dy);
                            ATOMIC_ADD(v_means2d[2 * (size_t)g + 0], vx);
                            ATOMIC_ADD(v_means2d[2 * (size_t)g + 1], vy);
                            if (v_means2d_abs) {
                                ATOMIC_ADD(v_means2d_abs[2 * (size_t)g + 0], fabsf(vx));
                                ATOMIC_ADD(v_means2d_abs[2 * (size_t)g + 1], fabsf(vy));
                            }
                        }
                        ATOMIC_ADD(v_opacities[g], s.vis * v_alpha);
                    }
                    for (uint32_t k = 0; k < cdim; ++k) buffer[k] += col[k] * fac;
                    for (int k = 0; k < 3; ++k) buffer_n[k] += nr[k] * fac;
                }
            }
        free(buffer);
    }
}

/* (gaussian, pixel, image) triples that contribute (RasterizeToIndices2DGSSerialBatch.cu): drives the reference's
 * accumulate_2dgs when pinning. Call with NULL outputs for the count. */
int64_t gso_raster2d_indices(const float *means2d, const float *ray_transforms, const float *opacities,
                             const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                             uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                             uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int64_t *gaussian_ids,
                             int64_t *pixel_ids, int64_t *image_ids)
{
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    int64_t count = 0;
    for (int64_t blk = 0; blk < total; ++blk) {
        const uint32_t img = (uint32_t)(blk / n_tiles), tile = (uint32_t)(blk % n_tiles);
        const uint32_t tx = tile % tile_w, ty = tile / tile_w;
        const int32_t start = isect_offsets[blk];
        const int32_t end   = (blk == total - 1) ? (int32_t)n_isects : isect_offsets[blk + 1];
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                const uint32_t ox = tx * tile_size + lx, oy = ty * tile_size + ly;
                if (ox >= width || oy >= height) continue;
                const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
                float T = 1.0f;
                for (int32_t idx = start; idx < end; ++idx) {
                    const int32_t g = flatten_ids[idx];
                    const S2Sample s = s2_eval(means2d, ray_transforms, opacities, g, px, py);
                    if (!s.valid) continue;
                    const float next_T = T * (1.0f - s.alpha);
                    if (next_T <= TRANSMITTANCE_THRESHOLD) break;
                    if (gaussian_ids) {
                        gaussian_ids[count] = g % (int64_t)n_per_image;
                        pixel_ids[count]    = (int64_t)oy * width + ox;
                        image_ids[count]    = img;
                    }
                    ++count;
                    T = next_T;
                }
            }
    }
    return count;
}
