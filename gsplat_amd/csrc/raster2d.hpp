// Shared pieces of the 2DGS (surfel) compositing kernels: argument block, per-sample evaluation, wave-level culling.
// Used by raster2d.hip (forward, reduction / one-wave backward) and raster2d_bwd_m.hip (backward on the matrix cores).
// Semantics restated from gsplat/cuda/csrc/RasterizeToPixels2DGSSerialBatchFwd.cu:43-465 and
// RasterizeToPixels2DGSSerialBatchBwd.cu:41-700 (see raster2d.hip).
#pragma once
#include "raster3d.hpp"

namespace gsx {

struct Raster2DArgs {
    uint32_t n_images, n_isects, width, height, tile_size, tile_w, tile_h, cdim;
    int distloss;
    const float *means2d;        // [R, 2]
    const float *ray_transforms; // [R, 9] rows u_M, v_M, w_M
    const float *colors;         // [R, cdim]
    const float *opacities;      // [R]
    const float *normals;        // [R, 3]
    const float *backgrounds;    // [I, cdim] or null
    const uint8_t *masks;        // [I, th, tw] or null
    const int32_t *isect_offsets;
    const int32_t *flatten_ids;
    // forward outputs / backward inputs
    float *render_colors, *render_alphas, *render_normals, *render_distort, *render_median;
    int32_t *last_ids, *median_ids;
    // backward inputs
    const float *v_render_colors, *v_render_alphas, *v_render_normals, *v_render_distort, *v_render_median;
    // backward output (zero-initialised): ONE array-of-structures buffer [R][row_stride]; row =
    // (v_means2d 2 | v_opacities 1 | v_densify 2 | v_normals 3 | v_ray_transforms 9 | [v_means2d_abs 2] | v_colors cdim)
    float *v_rows;
    uint32_t row_stride;
    const int32_t *tile_order; // backward: workgroup -> tile map, longest first (csrc/tile_order.hip), or null = launch order
};

// Bounding box (centre, half extents) of the pixels where a surfel can reach alpha >= 1/255, i.e. where
// min(G3, G2) <= 2 L with L = ln(255 opac): the union of
//   * the projection of the uv-disc |s| <= k, k^2 = 2 L. With rows u, v, w of the ray transform its exact screen
//     AABB is  c = f (k^2 (u0 w0 + u1 w1) - u2 w2),  h^2 = c^2 - f (k^2 (u0^2 + u1^2) - u2^2),
//     f = 1 / (k^2 (w0^2 + w1^2) - w2^2)  (the 2DGS bounding-box formula with (k^2, k^2, -1) in place of (1, 1, -1);
//     it is what Projection2DGSFused.cu evaluates at k = 1), valid while the denominator is negative;
//   * the low-pass disc |pixel - mean2d| <= sqrt(L)  (G2 = 2 |d|^2).
// Anything degenerate (disc reaching the camera plane, NaN) returns an infinite box = never culled; opac <= 1/255
// returns an empty one. A small margin dwarfs the rounding of the fast intrinsics.
__device__ __forceinline__ float4 surfel_cull_box(const float *M, float mx, float my, float opac)
{
    const float L = __logf(255.0f * opac) + 0.01f;
    if (!(L > 0.0f)) return make_float4(0.0f, 0.0f, -1.0f, -1.0f);
    const float k2  = 2.0f * L;
    const float den = k2 * (M[6] * M[6] + M[7] * M[7]) - M[8] * M[8];
    const float4 never = make_float4(0.0f, 0.0f, INFINITY, INFINITY);
    if (!(den < 0.0f)) return never;
    const float f   = 1.0f / den;
    const float cx  = f * (k2 * (M[0] * M[6] + M[1] * M[7]) - M[2] * M[8]);
    const float cy  = f * (k2 * (M[3] * M[6] + M[4] * M[7]) - M[5] * M[8]);
    const float hx2 = cx * cx - f * (k2 * (M[0] * M[0] + M[1] * M[1]) - M[2] * M[2]);
    const float hy2 = cy * cy - f * (k2 * (M[3] * M[3] + M[4] * M[4]) - M[5] * M[5]);
    if (!(hx2 >= 0.0f) || !(hy2 >= 0.0f)) return never;
    const float hx = sqrtf(hx2), hy = sqrtf(hy2), r2 = sqrtf(L);
    const float x0 = fminf(cx - hx, mx - r2), x1 = fmaxf(cx + hx, mx + r2);
    const float y0 = fminf(cy - hy, my - r2), y1 = fmaxf(cy + hy, my + r2);
    if (!(x1 - x0 < INFINITY) || !(y1 - y0 < INFINITY)) return never;
    return make_float4(0.5f * (x0 + x1), 0.5f * (y0 + y1), 0.5f * (x1 - x0) * 1.001f + 0.02f,
                       0.5f * (y1 - y0) * 1.001f + 0.02f);
}

struct Surfel { // one pixel x one surfel
    bool valid;
    float alpha, vis, gw3, gw2, sx, sy, dx, dy, rcz_inv;
};

// zeta = h_u x h_v with h_u = px w - u, h_v = py w - v is AFFINE in the pixel: (px w - u) x (py w - v) = u x v + px (v x w) +
// py (w x u) (the px py term is w x w = 0). So per (tile, surfel) the staging thread forms zeta at the TILE CENTRE with the
// well-conditioned product of the two small vectors h_u(c), h_v(c), plus the two gradients Z1 = v x w and Z2 = w x u, and a
// pixel at q = pixel - centre (|q| <= 7.5) costs 6 FMAs for zeta instead of 6 + 9 for the two rays and their cross product.
// Staged as three float4: (zeta_c, mean.x - c.x), (Z1, mean.y - c.y), (Z2, opacity).
__device__ __forceinline__ void stage_surfel(const float *M, float mx, float my, float opac, float cx, float cy, float4 &a,
                                             float4 &b, float4 &c)
{
    const float hu[3] = {cx * M[6] - M[0], cx * M[7] - M[1], cx * M[8] - M[2]};
    const float hv[3] = {cy * M[6] - M[3], cy * M[7] - M[4], cy * M[8] - M[5]};
    a = make_float4(hu[1] * hv[2] - hu[2] * hv[1], hu[2] * hv[0] - hu[0] * hv[2], hu[0] * hv[1] - hu[1] * hv[0], mx - cx);
    b = make_float4(M[4] * M[8] - M[5] * M[7], M[5] * M[6] - M[3] * M[8], M[3] * M[7] - M[4] * M[6], my - cy); // v x w
    c = make_float4(M[7] * M[2] - M[8] * M[1], M[8] * M[0] - M[6] * M[2], M[6] * M[1] - M[7] * M[0], opac);    // w x u
}

__device__ __forceinline__ Surfel eval_surfel(const float4 A /*zeta_c, x'*/, const float4 B /*Z1, y'*/, const float4 C /*Z2, opac*/,
                                              float qx, float qy, float thr = kAlphaThreshold)
{
    Surfel s;
    const float rx = fmaf(qy, C.x, fmaf(qx, B.x, A.x));
    const float ry = fmaf(qy, C.y, fmaf(qx, B.y, A.y));
    const float rz = fmaf(qy, C.z, fmaf(qx, B.z, A.z));
    s.rcz_inv = __builtin_amdgcn_rcpf(rz);
    s.sx  = rx * s.rcz_inv;
    s.sy  = ry * s.rcz_inv;
    s.gw3 = s.sx * s.sx + s.sy * s.sy;
    s.dx  = A.w - qx;
    s.dy  = B.w - qy;
    s.gw2 = kFilterInvSquare2DGS * (s.dx * s.dx + s.dy * s.dy);
    const float sigma = 0.5f * fminf(s.gw3, s.gw2);
    s.vis   = __expf(-sigma);
    s.alpha = fminf(kMaxAlpha, C.w * s.vis);
    s.valid = (rz != 0.0f) && !(sigma < 0.0f) && !(s.alpha < thr);
    return s;
}

// Second, exact stage of the wave-level culling. The box of surfel_cull_box() lets through every (wave, surfel) pair whose
// footprint misses the wave's 8 x 8 pixel rectangle diagonally: 31 % of the pairs the c5 backward evaluated had no lane that
// passes the alpha test. A pixel passes iff min(G3, G2) <= 2 L (L = ln(255 opac)), i.e. iff it lies
//   * in the low-pass disc |pixel - mean| <= sqrt(L)   -> distance from the mean to the rectangle, or
//   * where G3 = |zeta.xy|^2 / zeta.z^2 <= 2 L. zeta is affine in the pixel (stage_surfel), so that is F(q) <= 0 for the
//     quadratic F = zeta^T diag(1, 1, -2L) zeta in q = pixel - tile centre: its minimum over the rectangle, exactly - at the
//     unconstrained minimiser if that lies inside, else on the edges facing it (one clamped 1-D minimisation per edge).
// Conservative: the level carries the same +0.01 as the box, the comparison a margin of 1e-4 of the magnitude of F's terms;
// a quadratic that is not convex (a surfel seen nearly edge-on: the footprint is a hyperbola branch) is never culled.
__device__ __forceinline__ bool surfel_reaches_rect(const float4 A /*zeta_c, mean.x'*/, const float4 B /*Z1, mean.y'*/,
                                                    const float4 C /*Z2, opacity*/, float rcx, float rcy, float hw, float hh)
{
    const float L = __logf(255.0f * C.w) + 0.01f;
    if (!(L > 0.0f)) return false;
    const float x0 = rcx - hw, x1 = rcx + hw, y0 = rcy - hh, y1 = rcy + hh;
    {   // low-pass disc
        const float dx = fmaxf(fmaxf(x0 - A.w, A.w - x1), 0.0f), dy = fmaxf(fmaxf(y0 - B.w, B.w - y1), 0.0f);
        if (dx * dx + dy * dy <= L) return true;
    }
    const float k2 = 2.0f * L;
    const float a = fmaf(B.x, B.x, fmaf(B.y, B.y, -k2 * B.z * B.z));
    const float c = fmaf(C.x, C.x, fmaf(C.y, C.y, -k2 * C.z * C.z));
    const float b = 2.0f * fmaf(B.x, C.x, fmaf(B.y, C.y, -k2 * B.z * C.z));
    const float d = 2.0f * fmaf(A.x, B.x, fmaf(A.y, B.y, -k2 * A.z * B.z));
    const float e = 2.0f * fmaf(A.x, C.x, fmaf(A.y, C.y, -k2 * A.z * C.z));
    const float f = fmaf(A.x, A.x, fmaf(A.y, A.y, -k2 * A.z * A.z));
    const float det = 4.0f * a * c - b * b;
    if (!(a > 0.0f && c > 0.0f && det > 0.0f)) return true; // not an ellipse: keep
    const float inv = __builtin_amdgcn_rcpf(det);
    const float xm = (b * e - 2.0f * c * d) * inv, ym = (b * d - 2.0f * a * e) * inv; // unconstrained minimiser
    const float xe = fminf(fmaxf(xm, x0), x1), ye = fminf(fmaxf(ym, y0), y1);
    const float ys = fminf(fmaxf(-0.5f * fmaf(b, xe, e) * __builtin_amdgcn_rcpf(c), y0), y1); // along x = xe
    const float xs = fminf(fmaxf(-0.5f * fmaf(b, ye, d) * __builtin_amdgcn_rcpf(a), x0), x1); // along y = ye
    auto F = [&](float x, float y) { return fmaf(x, fmaf(a, x, fmaf(b, y, d)), fmaf(y, fmaf(c, y, e), f)); };
    const float X = fmaxf(fabsf(x0), fabsf(x1)), Y = fmaxf(fabsf(y0), fabsf(y1));
    const float mag = fmaf(X, fmaf(a, X, fmaf(fabsf(b), Y, fabsf(d))), fmaf(Y, fmaf(c, Y, fabsf(e)), fabsf(f)));
    return !(fminf(F(xe, ys), F(xs, ye)) > 1e-4f * mag); // NaN: keep
}

// Matrix-core backward (raster2d_bwd_m.hip): <= 4 channels, 16 x 16 tiles, no absgrad. GSX_RASTER2D_BWD=r|w|m at run time.
#ifndef GSX_RASTER2D_BWD_DEFAULT
#define GSX_RASTER2D_BWD_DEFAULT 'r'
#endif
bool raster2d_bwd_m_applies(const Raster2DArgs &a, bool has_abs);
int raster2d_bwd_m_launch(const Raster2DArgs &a, hipStream_t stream);

} // namespace gsx
