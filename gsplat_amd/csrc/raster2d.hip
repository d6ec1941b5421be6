// 2DGS (surfel) front-to-back compositing, forward + backward, gfx950.
// C-ABI entries: gsx_raster2d_fwd / gsx_raster2d_bwd (replace torch ops gsplat::rasterize_to_pixels_2dgs{,_bwd};
// reference kernels gsplat/cuda/csrc/RasterizeToPixels2DGSSerialBatchFwd.cu:43-465 and
// RasterizeToPixels2DGSSerialBatchBwd.cu:41-700, constants gsplat/cuda/include/Rasterization.h:40).
//
// Same MI355X work decomposition as the 3DGS kernels (raster3d.hpp): one workgroup of four wave64s per 16x16 tile, a
// wave owns an 8x8 pixel quadrant, the sorted list is walked in LDS-staged batches. The backward reduces the
// per-Gaussian sums with the wave64 reduce-scatter (permlane swaps + DPP), accumulates them in an LDS row per Gaussian
// (ds_add_f32) and flushes ONE global atomic per (tile, Gaussian, component) into array-of-structures gradient rows,
// transposed (consecutive lanes -> consecutive floats of one surfel's row; see raster3d_bwd.hip for the measurement that
// motivates it). v_densify is not reduced at all: it is (v_uM.z, v_vM.z) * w_M.z, a per-Gaussian factor applied to
// already-reduced sums at flush time. Forward and backward cull (wave, surfel) pairs with the staged bounding box of
// the surfel's alpha >= 1/255 region (surfel_cull_box below).
//
// Per-sample math (Fwd.cu:356-428): h_u = px w_M - u_M, h_v = py w_M - v_M, zeta = h_u x h_v, s = zeta.xy / zeta.z,
// G3 = |s|^2, G2 = 2 |mean2d - p|^2, sigma = min(G3, G2) / 2, alpha = min(0.99, opac exp(-sigma)); skip if zeta.z == 0,
// sigma < 0 or alpha < 1/255; stop (exclusive) if T (1 - alpha) <= 1e-4.
#include <cstdlib>

#include "raster2d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

constexpr int kBatch2 = 256;

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int CH>
__global__ void __launch_bounds__(256) raster2d_fwd_kernel(const Raster2DArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_A  = reinterpret_cast<float4 *>(smem_raw);
    float4 *s_B  = s_A + kBatch2;
    float4 *s_C  = s_B + kBatch2;
    float4 *s_N  = s_C + kBatch2;                             // normal xyz, pad
    float4 *s_cull = s_N + kBatch2;                           // surfel_cull_box
    float *s_col = reinterpret_cast<float *>(s_cull + kBatch2); // [kBatch2][CH]

    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t n_blocks        = tiles_per_image * a.n_images;
    const uint32_t blk             = xcd_remap(blockIdx.x, n_blocks);
    if (blk >= n_blocks) return;
    const uint32_t image_id = blk / tiles_per_image, tile_id = blk % tiles_per_image;
    const uint32_t tile_x = tile_id % a.tile_w, tile_y = tile_id / a.tile_w;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const uint32_t ox = tile_x * a.tile_size + lx, oy = tile_y * a.tile_size + ly;
    const bool inside = (lx < a.tile_size) && (ly < a.tile_size) && (ox < a.width) && (oy < a.height);
    const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
    const float half = 0.5f * (float)a.tile_size; // tile centre, and this lane's pixel centre relative to it (exact)
    const float tcx = (float)(tile_x * a.tile_size) + half, tcy = (float)(tile_y * a.tile_size) + half;
    const float qx = (float)lx + 0.5f - half, qy = (float)ly + 0.5f - half;
    const size_t pix = ((size_t)image_id * a.height + oy) * a.width + ox;
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)image_id * a.cdim : nullptr;
    const int nch   = (int)a.cdim;

    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) {
        if (inside) {
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < nch) a.render_colors[pix * a.cdim + k] = bg ? bg[k] : 0.0f;
            a.render_alphas[pix] = 0.0f;
            for (int k = 0; k < 3; ++k) a.render_normals[pix * 3 + k] = 0.0f;
            a.render_distort[pix] = 0.0f;
            a.render_median[pix]  = 0.0f;
            a.last_ids[pix]       = 0;
            a.median_ids[pix]     = 0;
        }
        return;
    }

    const int32_t range_start = a.isect_offsets[(size_t)image_id * tiles_per_image + tile_id];
    const int32_t range_end   = (blk == n_blocks - 1) ? (int32_t)a.n_isects
                                                      : a.isect_offsets[(size_t)image_id * tiles_per_image + tile_id + 1];
    const int32_t n_batches   = (range_end - range_start + kBatch2 - 1) / kBatch2;

    float T = 1.0f, distort = 0.0f, accum_vis_depth = 0.0f, median_depth = 0.0f;
    uint32_t cur_idx = 0, median_idx = 0;
    float acc[CH], nrm[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = 0.0f;
    float thr = inside ? kAlphaThreshold : INFINITY; // alpha threshold of this pixel; +inf = done (or not rendered)
    const uint32_t lane = tid & 63u;

    for (int32_t b = 0; b < n_batches; ++b) {
        if (__syncthreads_count(!(thr < INFINITY)) == (int)blockDim.x) break;
        const int32_t batch_start = range_start + kBatch2 * b;
        for (int s = (int)tid; s < kBatch2; s += (int)blockDim.x) {
            const int32_t idx = batch_start + s;
            if (idx < range_end) {
                const int32_t g = a.flatten_ids[idx];
                const float *M  = a.ray_transforms + 9 * (size_t)g;
                const float2 xy = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                stage_surfel(M, xy.x, xy.y, opac, tcx, tcy, s_A[s], s_B[s], s_C[s]);
                s_cull[s] = surfel_cull_box(M, xy.x, xy.y, opac);
                const float *n = a.normals + 3 * (size_t)g;
                s_N[s] = make_float4(n[0], n[1], n[2], 0.0f);
                const float *c = a.colors + (size_t)g * a.cdim;
#pragma unroll
                for (int k = 0; k < CH; ++k) s_col[s * CH + k] = (k < nch) ? c[k] : 0.0f;
            }
        }
        __syncthreads();
        const int32_t batch_size = min(kBatch2, range_end - batch_start);
        // the rectangle shrinks with the pixels that are still open (refreshed once per batch: ~50 instructions)
        const WaveRect rect = wave_pixel_rect(thr < INFINITY, px, py);
        // 64 staged surfels are tested per instruction against this wave's pixel rectangle; only the ballot survivors
        // are evaluated, front to back (a culled pair has no pixel that could pass the alpha test)
        for (int32_t j = 0; j < batch_size; j += 64) {
          if (__builtin_amdgcn_ballot_w64(thr < INFINITY) == 0ull) break;
          const int32_t tl = j + (int32_t)lane;
          bool hit         = false;
          if (tl < batch_size) {
              const float4 cu = s_cull[tl];
              hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
              if (hit) hit = surfel_reaches_rect(s_A[tl], s_B[tl], s_C[tl], rect.cx - tcx, rect.cy - tcy, rect.hw, rect.hh);
          }
          uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) {
            const int32_t bit = (int32_t)__builtin_ctzll(todo);
            const int32_t t   = j + bit;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit)); // todo &= todo - 1 in one scalar instruction
            const Surfel s = eval_surfel(s_A[t], s_B[t], s_C[t], qx, qy, thr);
            // Branch-free body (raster3d_fwd.hip: the scalar unit, not the vector ALU, is the busy pipe when the pixel's
            // state lives in exec-style masks): `thr` is this pixel's alpha threshold, +inf once it is done.
            if (__builtin_amdgcn_ballot_w64(s.valid) == 0ull) continue; // wave-uniform
            float a_m          = s.valid ? s.alpha : 0.0f;
            const float next_T = fmaf(-T, a_m, T);               // == T when a_m == 0, and T > 1e-4 while not done
            const bool sat     = next_T <= kTransmittanceThresh; // saturated: this surfel is excluded
            thr                = sat ? INFINITY : thr;
            a_m                = sat ? 0.0f : a_m;
            const float w      = a_m * T;
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[k] += s_col[t * CH + k] * w;
            const float4 n = s_N[t];
            nrm[0] += n.x * w; nrm[1] += n.y * w; nrm[2] += n.z * w;
            const float depth = s_col[t * CH + nch - 1];
            if (a.distloss) {
                distort += 2.0f * (w * depth * (1.0f - T) - w * accum_vis_depth);
                accum_vis_depth += w * depth;
            }
            const bool contrib = a_m > 0.0f;
            const bool med     = contrib && T > 0.5f;
            median_depth = med ? depth : median_depth;
            median_idx   = med ? (uint32_t)(batch_start + t) : median_idx;
            cur_idx      = contrib ? (uint32_t)(batch_start + t) : cur_idx;
            T            = sat ? T : next_T;
          }
        }
    }
    if (inside) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < nch) a.render_colors[pix * a.cdim + k] = bg ? (acc[k] + T * bg[k]) : acc[k];
        a.render_alphas[pix] = 1.0f - T;
        for (int k = 0; k < 3; ++k) a.render_normals[pix * 3 + k] = nrm[k];
        a.render_distort[pix] = a.distloss ? distort : 0.0f;
        a.render_median[pix]  = median_depth;
        a.last_ids[pix]       = (int32_t)cur_idx;
        a.median_ids[pix]     = (int32_t)median_idx;
    }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// accumulator row layout: [0,CH) colours | CH..CH+2 normals | CH+3..CH+11 ray_transforms | CH+12,13 means2d |
//                         CH+14 opacity | (CH+15,16 |means2d| when ABS)
template <int CH, bool ABS>
struct Bwd2Cfg {
    static constexpr int K     = CH + 15 + (ABS ? 2 : 0);
    static constexpr int KQ    = (K + 3) / 4;
    static constexpr int KP    = (K | 1);
    static constexpr int BATCH = (CH >= 8) ? 64 : 128;
    static constexpr size_t smem =
        (size_t)BATCH * (8 * sizeof(float4) + 2 * sizeof(int32_t) + sizeof(float) * (CH + KP));
};

template <int CH, bool ABS>
__global__ void __launch_bounds__(256) raster2d_bwd_kernel(const Raster2DArgs a)
{
    using Cfg           = Bwd2Cfg<CH, ABS>;
    constexpr int K     = Cfg::K;
    constexpr int KQ    = Cfg::KQ;
    constexpr int KP    = Cfg::KP;
    constexpr int BATCH = Cfg::BATCH;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_A      = reinterpret_cast<float4 *>(smem_raw);
    float4 *s_B      = s_A + BATCH;
    float4 *s_C      = s_B + BATCH;
    float4 *s_N      = s_C + BATCH;
    float4 *s_cull   = s_N + BATCH;
    float4 *s_Za     = s_cull + BATCH; // evaluation form (stage_surfel): zeta at the tile centre | Z1 | Z2
    float4 *s_Zb     = s_Za + BATCH;
    float4 *s_Zc     = s_Zb + BATCH;
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_Zc + BATCH);
    int32_t *s_touch = s_id + BATCH;
    float *s_col     = reinterpret_cast<float *>(s_touch + BATCH);
    float *s_acc     = s_col + BATCH * CH;

    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t n_blocks        = tiles_per_image * a.n_images;
    const uint32_t slot            = xcd_remap(blockIdx.x, n_blocks);
    if (slot >= n_blocks) return;
    const uint32_t blk = a.tile_order ? (uint32_t)a.tile_order[slot] : slot;
    const uint32_t image_id = blk / tiles_per_image, tile_id = blk % tiles_per_image;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;
    const uint32_t tile_x = tile_id % a.tile_w, tile_y = tile_id / a.tile_w;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const uint32_t ox = tile_x * a.tile_size + lx, oy = tile_y * a.tile_size + ly;
    const bool inside = (lx < a.tile_size) && (ly < a.tile_size) && (ox < a.width) && (oy < a.height);
    const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
    const float half = 0.5f * (float)a.tile_size;
    const float tcx = (float)(tile_x * a.tile_size) + half, tcy = (float)(tile_y * a.tile_size) + half;
    const float qx = (float)lx + 0.5f - half, qy = (float)ly + 0.5f - half;
    const size_t pix = inside ? ((size_t)image_id * a.height + oy) * a.width + ox : 0;
    const int nch    = (int)a.cdim;
    const float X0 = (float)(tile_x * a.tile_size), Y0 = (float)(tile_y * a.tile_size); // tile origin
    const float plx = px - X0, ply = py - Y0;                                           // tile-local pixel centre

    const int32_t range_start = a.isect_offsets[(size_t)image_id * tiles_per_image + tile_id];
    int32_t range_end         = (blk == n_blocks - 1) ? (int32_t)a.n_isects
                                                      : a.isect_offsets[(size_t)image_id * tiles_per_image + tile_id + 1];
    if (range_end <= range_start) return;

    const float T_final      = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    float T                  = T_final;
    const int32_t bin_final  = inside ? a.last_ids[pix] : -1;
    // nothing behind the last contributor of the whole tile is needed (see raster3d_bwd.hip): those entries are not staged
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    {
        int32_t *s_m = reinterpret_cast<int32_t *>(smem_raw); // staging area, not in use yet
        if (lane == 0) s_m[tid >> 6] = wave_bin_final;
        __syncthreads();
        int32_t m = s_m[0];
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) m = max(m, s_m[w]);
        __syncthreads();
        range_end = min(range_end, m + 1);
    }
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return;
    float v_c[CH], v_n[3];
#pragma unroll
    for (int k = 0; k < CH; ++k) v_c[k] = (inside && k < nch) ? a.v_render_colors[pix * a.cdim + k] : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v_n[k] = (inside && a.v_render_normals) ? a.v_render_normals[pix * 3 + k] : 0.0f;
    const float v_a      = (inside && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
    // The cotangent of the median depth goes to the depth channel of ONE surfel per pixel, the one the forward pass recorded
    // (median_ids; a pixel with any contributor has one: the first contributor sees T = 1 > 0.5). It is added here, once per
    // pixel, instead of being tested for on every (pixel, surfel) pair of the walk (two instructions and two registers less
    // per pair; reference Bwd.cu adds it inside the walk).
    if (inside && T_final < 1.0f) {
        const float v_median = a.v_render_median ? a.v_render_median[pix] : 0.0f;
        if (v_median != 0.0f) {
            constexpr int GEO0 = 17 + (ABS ? 2 : 0);
            atomic_add_f32(a.v_rows + (size_t)a.flatten_ids[a.median_ids[pix]] * a.row_stride + GEO0 + nch - 1, v_median);
        }
    }
    float bg_dot = 0.0f;
    if (a.backgrounds) {
        const float *bg = a.backgrounds + (size_t)image_id * a.cdim;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < nch) bg_dot += bg[k] * v_c[k];
    }
    const float tail_term = T_final * (v_a - bg_dot); // what lies behind the whole list
    float behind          = 0.0f;                     // B (see the pixel loop)
    // Distortion loss, restated (reference Bwd.cu:560-600 keeps accum_d, accum_w, their two back-buffers and a distortion
    // buffer per pixel): with Wb / Db the sums of fac and fac * depth over the surfels BEHIND,
    //   dl_dw = 2 (2 (depth (accum_w - Wb) - (accum_d - Db)) + accum_d - depth accum_w) = 2 (depth P - Q),
    //   P = accum_w - 2 Wb,  Q = accum_d - 2 Db   (two running values; accum_d itself drops out),
    // and the distortion buffer only ever enters v_alpha next to `behind` with the same factors, so it is carried inside it:
    //   v_alpha = T (cv + dl_dw v_distort) + (tail - behind') / (1 - alpha),  behind' += fac (cv + dl_dw v_distort).
    // Four values per pixel instead of six, nine instructions per pair instead of seventeen; same sums in another order.
    const bool dist = a.v_render_distort != nullptr;
    float vd2 = 0.f, c2aw = 0.f, dP = 0.f, dQ = 0.f; // 2 v_distort | 2 - accum_w | P | Q
    if (dist && inside) {
        vd2  = 2.0f * a.v_render_distort[pix];
        dQ   = a.render_colors[pix * a.cdim + nch - 1];
        dP   = a.render_alphas[pix];
        c2aw = 2.0f - dP;
    }

    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int k = 0; k < KP; ++k) s_acc[s * KP + k] = 0.0f;
        s_touch[s] = 0;
    }

    for (int32_t b = 0; b < n_batches; ++b) {
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);
        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            const int32_t idx = batch_end - s;
            if (idx >= range_start) {
                const int32_t g = a.flatten_ids[idx];
                const float *M  = a.ray_transforms + 9 * (size_t)g;
                const float2 xy = reinterpret_cast<const float2 *>(a.means2d)[g];
                s_id[s] = g;
                s_A[s]  = make_float4(M[0], M[1], M[2], xy.x);
                s_B[s]  = make_float4(M[3], M[4], M[5], xy.y);
                const float opac = a.opacities[g];
                s_C[s]  = make_float4(M[6], M[7], M[8], opac);
                stage_surfel(M, xy.x, xy.y, opac, tcx, tcy, s_Za[s], s_Zb[s], s_Zc[s]);
                s_cull[s] = surfel_cull_box(M, xy.x, xy.y, opac);
                const float *n = a.normals + 3 * (size_t)g;
                s_N[s]  = make_float4(n[0], n[1], n[2], 0.0f);
                const float *c = a.colors + (size_t)g * a.cdim;
#pragma unroll
                for (int k = 0; k < CH; ++k) s_col[s * CH + k] = (k < nch) ? c[k] : 0.0f;
            }
        }
        __syncthreads();

        const int32_t t_first = max(0, batch_end - wave_bin_final); // surfels behind every last contributor: skipped
        // pixels whose last contributor lies in front of this whole batch take no part in it: the rectangle is the box of
        // the others (it grows from batch to batch as the walk moves towards the front)
        const WaveRect rect = wave_pixel_rect(inside && bin_final >= batch_end - (batch_size - 1), px, py);
        for (int32_t j = (t_first & ~63); j < batch_size; j += 64) {
          const int32_t tl = j + (int32_t)lane;
          bool hit         = false;
          if (tl >= t_first && tl < batch_size) {
              const float4 cu = s_cull[tl];
              hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
              if (hit) hit = surfel_reaches_rect(s_Za[tl], s_Zb[tl], s_Zc[tl], rect.cx - tcx, rect.cy - tcy, rect.hw, rect.hh);
          }
          uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) { // scalar loop over the survivors, back to front
            const int32_t bit = (int32_t)__builtin_ctzll(todo);
            const int32_t t   = j + bit;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit)); // todo &= todo - 1 in one scalar instruction
            const float4 C = s_Zc[t];
            const Surfel s = eval_surfel(s_Za[t], s_Zb[t], C, qx, qy);
            const bool valid = inside && (batch_end - t <= bin_final) && s.valid;
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;

            // branch-free: invalid lanes run with alpha = vis = 0 (every contribution becomes exactly 0)
            const float alpha = valid ? s.alpha : 0.0f;
            const float vis   = valid ? s.vis : 0.0f;
            const float opac  = C.w;
            float loc[KQ * 4];
#pragma unroll
            for (int k = 0; k < KQ * 4; ++k) loc[k] = 0.0f;

            const float ra  = __builtin_amdgcn_rcpf(fmaxf(kMinOneMinusAlpha, 1.0f - alpha));
            T              *= ra;
            const float fac = alpha * T;
            // The back-buffers of the colour and normal channels only enter through their dot product with the pixel's
            // cotangents, B = sum_k buffer_k v_k, and B += fac (c . v_c + n . v_n): one scalar per pixel (raster3d_bwd.hip).
            float cv = 0.0f;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                loc[k] = fac * v_c[k];
                cv     = fmaf(s_col[t * CH + k], v_c[k], cv);
            }
            const float4 nr = s_N[t];
            const float nrv[3] = {nr.x, nr.y, nr.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                loc[CH + k] = fac * v_n[k];
                cv          = fmaf(nrv[k], v_n[k], cv);
            }
            if (dist) {
                const float depth = s_col[t * CH + nch - 1];
                cv                = fmaf(fmaf(depth, dP, -dQ), vd2, cv);                    // cv + dl_dw v_distort
                const float v_depth_ch = fac * vd2 * (fmaf(-2.0f, T, c2aw) + fac);        // 2 fac (2 - 2 T - accum_w + fac) v_distort
                dP = fmaf(-2.0f, fac, dP);
                dQ = fmaf(-2.0f * depth, fac, dQ);
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (k == nch - 1) loc[k] += v_depth_ch;
            }
            const float v_alpha = fmaf(ra, tail_term - behind, cv * T);
            behind              = fmaf(fac, cv, behind);

            const float ov       = opac * vis;
            const bool unclamped = valid && (ov <= kMaxAlpha);
            const float v_G      = unclamped ? opac * v_alpha : 0.0f;
            const bool use3d     = s.gw3 <= s.gw2;
            {
                // 3D branch: through s = zeta.xy / zeta.z
                const float g3   = use3d ? v_G * -vis : 0.0f;
                const float a_   = g3 * s.sx * s.rcz_inv, b_ = g3 * s.sy * s.rcz_inv;
                const float vrc[3] = {a_, b_, -(a_ * s.sx + b_ * s.sy)};
                // The ray-transform gradient is LINEAR in three moments of vrc over the pixels (hu = px w - u,
                // hv = py w - v):  v_uM = sum -(hv x vrc) = v x S0 - w x Sy,  v_vM = sum -(vrc x hu) = S0 x u - Sx x w,
                // v_wM = sum px (hv x vrc) + py (vrc x hu) = Sx x v + u x Sy, with S0 = sum vrc, Sx = sum px vrc,
                // Sy = sum py vrc. Only the moments are reduced (tile-local pixel coordinates keep them small); the
                // cross products are taken once per (tile, surfel) in the flush instead of once per pixel.
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // g3 == 0 can still meet inf/NaN geometry on invalid lanes: select, do not multiply
                    const float vr  = (use3d && unclamped) ? vrc[k] : 0.0f;
                    loc[CH + 3 + k] = vr;
                    loc[CH + 6 + k] = plx * vr;
                    loc[CH + 9 + k] = ply * vr;
                }
                // 2D (low-pass) branch
                const float g2 = (!use3d && unclamped) ? v_G * (-vis * kFilterInvSquare2DGS) : 0.0f;
                const float vx = g2 * s.dx, vy = g2 * s.dy;
                loc[CH + 12] = vx;
                loc[CH + 13] = vy;
                loc[CH + 14] = unclamped ? vis * v_alpha : 0.0f;
                if constexpr (ABS) {
                    loc[CH + 15] = fabsf(vx);
                    loc[CH + 16] = fabsf(vy);
                }
            }
            float mine = 0.0f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float r = wave_sum4_scatter(loc[4 * j], loc[4 * j + 1], loc[4 * j + 2], loc[4 * j + 3]);
                if ((int)(lane & 15u) == j) mine = r;
            }
            const int vidx = 4 * (int)(lane & 15u) + (int)(lane >> 4);
            if ((int)(lane & 15u) < KQ && vidx < K) atomicAdd(&s_acc[t * KP + vidx], mine);
            if (lane == 0) s_touch[t] = 1;
          }
        }
        __syncthreads();

        // transposed flush into the AoS gradient rows: element e -> (surfel s, output column c); accumulator slots:
        // [0,CH) colours | CH..CH+2 normals | CH+3..5 S0, CH+6..8 S_lx, CH+9..11 S_ly (moments of vrc, tile-local) |
        // CH+12,13 means2d | CH+14 opacity | CH+15,16 abs
        constexpr int GEO  = 17 + (ABS ? 2 : 0);
        constexpr int NCOL = GEO + CH;
        for (int e = (int)tid; e < batch_size * NCOL; e += (int)blockDim.x) {
            const int s = e / NCOL, c = e - s * NCOL;
            if (!s_touch[s]) continue;
            const float *row = s_acc + s * KP;
            float val;
            int col = c;
            if (c < 2) val = row[CH + 12 + c];                               // v_means2d
            else if (c == 2) val = row[CH + 14];                             // v_opacities
            else if (c < 5 || (c >= 8 && c < 17)) {
                // component k of row r (0 = u_M, 1 = v_M, 2 = w_M) of the ray-transform gradient from the moments;
                // v_densify (columns 3, 4) = (v_uM.z, v_vM.z) * w_M.z
                const int r = c < 5 ? c - 3 : (c - 8) / 3, k = c < 5 ? 2 : (c - 8) % 3;
                const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                const float4 A4 = s_A[s], B4 = s_B[s], C4 = s_C[s];
                auto pick = [](const float4 &q, int i) { return i == 0 ? q.x : (i == 1 ? q.y : q.z); }; // no scratch
                const float u1 = pick(A4, k1), u2 = pick(A4, k2), v1 = pick(B4, k1), v2 = pick(B4, k2);
                const float w1 = pick(C4, k1), w2 = pick(C4, k2);
                const float s01 = row[CH + 3 + k1], s02 = row[CH + 3 + k2];
                const float sx1 = X0 * s01 + row[CH + 6 + k1], sx2 = X0 * s02 + row[CH + 6 + k2];
                const float sy1 = Y0 * s01 + row[CH + 9 + k1], sy2 = Y0 * s02 + row[CH + 9 + k2];
                // (p x q)_k = p_k1 q_k2 - p_k2 q_k1
                if (r == 0) val = (v1 * s02 - v2 * s01) - (w1 * sy2 - w2 * sy1);      // v_uM = v x S0 - w x Sy
                else if (r == 1) val = (s01 * u2 - s02 * u1) - (sx1 * w2 - sx2 * w1); // v_vM = S0 x u - Sx x w
                else val = (sx1 * v2 - sx2 * v1) + (u1 * sy2 - u2 * sy1);             // v_wM = Sx x v + u x Sy
                if (c < 5) val *= C4.z;
            } else if (c < 8) val = row[CH + (c - 5)];               // v_normals
            else if (c < GEO) val = row[CH + 15 + (c - 17)];                 // |v_means2d|
            else {
                const int k = c - GEO;
                if (k >= nch) continue;
                val = row[k];
                col = GEO + k;
            }
            atomic_add_f32(a.v_rows + (size_t)s_id[s] * a.row_stride + col, val);
        }
        __syncthreads();
        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            if (s < batch_size && s_touch[s]) {
                float *row = s_acc + s * KP;
#pragma unroll
                for (int k = 0; k < KP; ++k) row[k] = 0.0f;
                s_touch[s] = 0;
            }
        }
    }
}

// ---- backward, ONE WAVE PER TILE (the decomposition of variant W in raster3d_bwd.hip) -------------------------------------
// The kernel above reduces K = 15 + D values per (wave, surfel) pair over the 64 pixels of one 8 x 8 quadrant - 62 of its
// ~160 VALU instructions per pair, 2.8 pairs per staged surfel on c5 - and combines the four waves of a tile with LDS float
// atomics between workgroup barriers. Here a tile is one wave64 and a lane owns the same pixel of each quadrant: the K values
// are first summed over the lane's (up to) four pixels with plain adds and reduced ONCE per (tile, surfel); the totals pass
// through a 2-slot LDS scratch, and lane c of slot g turns them into column c of the gradient row (the cross products of the
// ray-transform gradient are linear in nine moments, as above) and adds it to HBM. Wave-private staging (64 surfels per
// batch, one per lane, rows requested one batch ahead), a 4-bit quadrant mask per staged surfel from the two-stage cull,
// no __syncthreads, no LDS atomics, no accumulator table; tiles longest-first (tile_order.hip). D <= 4, 16 x 16 tiles, no
// absgrad. Parity-green (tests/test_gpu_variants.py) and selectable (GSX_RASTER2D_BWD=w); not the default, see launch2_bwd.
#ifndef GSX_BWD2_H_WAVES // the half-tile variant (NQ = 2)
#define GSX_BWD2_H_WAVES 4
#endif
#ifndef GSX_BWD2_W_WAVES // 168 VGPRs (three waves per SIMD) spill 51 registers at four channels: two
#define GSX_BWD2_W_WAVES 2
#endif
#ifndef GSX_BWD2_W_STAGE_AHEAD
#define GSX_BWD2_W_STAGE_AHEAD 1
#endif
template <int CH>
struct Bwd2WCfg {
    static constexpr int K     = CH + 15;
    static constexpr int KQ    = (K + 3) / 4;
    static constexpr int NCOL  = 17 + CH;
    static constexpr int BATCH = 64;
    static constexpr int SLOTS = 2;  // surfels per flush: 2 slots x 32 lanes
    static constexpr int TP    = 24; // floats per slot of the totals scratch (K <= 19)
    static constexpr size_t smem = (size_t)BATCH * (8 * sizeof(float4)) + sizeof(float) * SLOTS * TP;
};

// NQ = 4: one wave per tile (four pixels per lane). NQ = 2 (round 5): one wave per HALF tile - the quadrants above or below
// the tile's middle, two pixels per lane: half the per-pixel state (three waves per SIMD instead of two), one reduction per
// (half tile, surfel) instead of one per (quadrant, surfel) as in the four-wave kernel; the halves of a tile are workgroups b
// and b + 8 (the same XCD, dispatched next to each other: the staged surfels are still in its L2).
template <int CH, bool DIST, int NQ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NQ == 2 ? GSX_BWD2_H_WAVES : (DIST ? 2 : GSX_BWD2_W_WAVES)))) raster2d_bwd_w_kernel(const Raster2DArgs a)
{
    using Cfg           = Bwd2WCfg<CH>;
    constexpr int K     = Cfg::K;
    constexpr int KQ    = Cfg::KQ;
    constexpr int NCOL  = Cfg::NCOL;
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int TP    = Cfg::TP;
    static_assert(CH <= 4, "colours are staged as one float4");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_A   = reinterpret_cast<float4 *>(smem_raw); // u_M, mean.x
    float4 *s_B   = s_A + BATCH;                          // v_M, mean.y
    float4 *s_C   = s_B + BATCH;                          // w_M, opacity
    float4 *s_N   = s_C + BATCH;                          // normal, flatten id (bits)
    float4 *s_Za  = s_N + BATCH;                          // evaluation form (stage_surfel)
    float4 *s_Zb  = s_Za + BATCH;
    float4 *s_Zc  = s_Zb + BATCH;
    float4 *s_col = s_Zc + BATCH;                         // colours (zero padded)
    float *s_tot  = reinterpret_cast<float *>(s_col + BATCH); // [SLOTS][TP] totals of the open slots

    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t n_blocks        = tiles_per_image * a.n_images;
    // NQ == 2: workgroups 16 i + j and 16 i + 8 + j (j < 8) are the upper and the lower half of tile slot 8 i + j
    const uint32_t unit     = NQ == 4 ? blockIdx.x : (blockIdx.x >> 4) * 8u + (blockIdx.x & 7u);
    const uint32_t q_first  = NQ == 4 ? 0u : ((blockIdx.x >> 3) & 1u) * 2u; // first quadrant of this wave
    const uint32_t slot_idx = xcd_remap(unit, n_blocks);
    if (slot_idx >= n_blocks) return;
    const uint32_t blk      = a.tile_order ? (uint32_t)a.tile_order[slot_idx] : slot_idx;
    const uint32_t image_id = blk / tiles_per_image, tile_id = blk % tiles_per_image;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;
    const uint32_t tile_x = tile_id % a.tile_w, tile_y = tile_id / a.tile_w;
    const int32_t range_start = a.isect_offsets[blk];
    const int32_t list_end    = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    if (list_end <= range_start) return;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lqx = lane & 7u, lqy = lane >> 3;
    const int nch = (int)a.cdim;
    const float tcx = (float)(tile_x * 16u) + 8.0f, tcy = (float)(tile_y * 16u) + 8.0f; // tile centre
    const float X0 = (float)(tile_x * 16u), Y0 = (float)(tile_y * 16u);                   // tile origin
    constexpr bool dist = DIST; // the distortion loss adds six per-pixel values: its own instantiation (two waves per SIMD)

    // per-pixel state, pixel q = this lane's pixel of quadrant q
    float T[NQ], behind[NQ], tail_term[NQ], v_c[NQ][CH], v_n[NQ][3];
    float vd2[NQ], c2aw[NQ], dP[NQ], dQ[NQ]; // distortion loss, see raster2d_bwd_kernel: 2 v_distort | 2 - accum_w | P | Q
    int32_t bin_final[NQ], qmax[NQ];
    int32_t tile_last = -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t gq = q_first + (uint32_t)q; // quadrant of the tile
        const uint32_t lx = ((gq & 1u) << 3) | lqx, ly = ((gq >> 1) << 3) | lqy;
        const uint32_t ox = tile_x * 16u + lx, oy = tile_y * 16u + ly;
        const bool inside = ox < a.width && oy < a.height;
        const size_t pix  = inside ? ((size_t)image_id * a.height + oy) * a.width + ox : 0;
        const float T_fin = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
        T[q] = T_fin; behind[q] = 0.0f;
        bin_final[q]  = inside ? a.last_ids[pix] : -1;
        if (inside && T_fin < 1.0f) { // the median depth's cotangent: once per pixel (see raster2d_bwd_kernel)
            const float v_median = a.v_render_median ? a.v_render_median[pix] : 0.0f;
            if (v_median != 0.0f)
                atomic_add_f32(a.v_rows + (size_t)a.flatten_ids[a.median_ids[pix]] * a.row_stride + 17 + nch - 1, v_median);
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) v_c[q][k] = (inside && k < nch) ? a.v_render_colors[pix * a.cdim + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) v_n[q][k] = (inside && a.v_render_normals) ? a.v_render_normals[pix * 3 + k] : 0.0f;
        const float v_a = (inside && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
        float bg_dot    = 0.0f;
        if (a.backgrounds) {
            const float *bg = a.backgrounds + (size_t)image_id * a.cdim;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < nch) bg_dot += bg[k] * v_c[q][k];
        }
        tail_term[q] = T_fin * (v_a - bg_dot);
        vd2[q] = c2aw[q] = dP[q] = dQ[q] = 0.0f;
        if (dist && inside) {
            vd2[q]  = 2.0f * a.v_render_distort[pix];
            dQ[q]   = a.render_colors[pix * a.cdim + nch - 1];
            dP[q]   = a.render_alphas[pix];
            c2aw[q] = 2.0f - dP[q];
        }
        qmax[q]   = wave_max_i32(bin_final[q]);
        tile_last = max(tile_last, qmax[q]);
    }
    const int32_t range_end = min(list_end, tile_last + 1); // nothing behind the tile's last contributor is needed
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return;

    int slot      = 0; // wave-uniform: open slots
    int slot_t[2] = {0, 0}; // staged index of the surfel in each slot (wave-uniform)

    // totals of the open slots -> gradient rows: lane (slot g = lane / 32, column c = lane % 32)
    auto flush = [&](int n_slots) {
        wave_lds_sync();
        const int g = (int)(lane >> 5), c = (int)(lane & 31u);
        if (g < n_slots && c < NCOL) {
            const int s      = g == 0 ? slot_t[0] : slot_t[1];
            const float *row = s_tot + g * TP;
            constexpr int GEO = 17;
            float val;
            int col = c;
            bool put = true;
            if (c < 2) val = row[CH + 12 + c];                               // v_means2d
            else if (c == 2) val = row[CH + 14];                             // v_opacities
            else if (c < 5 || (c >= 8 && c < 17)) {
                // component k of row r (0 = u_M, 1 = v_M, 2 = w_M) of the ray-transform gradient from the moments;
                // v_densify (columns 3, 4) = (v_uM.z, v_vM.z) * w_M.z  (see the flush of the kernel above)
                const int r = c < 5 ? c - 3 : (c - 8) / 3, k = c < 5 ? 2 : (c - 8) % 3;
                const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                const float4 A4 = s_A[s], B4 = s_B[s], C4 = s_C[s];
                auto pick = [](const float4 &q4, int i) { return i == 0 ? q4.x : (i == 1 ? q4.y : q4.z); };
                const float u1 = pick(A4, k1), u2 = pick(A4, k2), v1 = pick(B4, k1), v2 = pick(B4, k2);
                const float w1 = pick(C4, k1), w2 = pick(C4, k2);
                const float s01 = row[CH + 3 + k1], s02 = row[CH + 3 + k2];
                const float sx1 = X0 * s01 + row[CH + 6 + k1], sx2 = X0 * s02 + row[CH + 6 + k2];
                const float sy1 = Y0 * s01 + row[CH + 9 + k1], sy2 = Y0 * s02 + row[CH + 9 + k2];
                if (r == 0) val = (v1 * s02 - v2 * s01) - (w1 * sy2 - w2 * sy1);      // v_uM = v x S0 - w x Sy
                else if (r == 1) val = (s01 * u2 - s02 * u1) - (sx1 * w2 - sx2 * w1); // v_vM = S0 x u - Sx x w
                else val = (sx1 * v2 - sx2 * v1) + (u1 * sy2 - u2 * sy1);             // v_wM = Sx x v + u x Sy
                if (c < 5) val *= C4.z;
            } else if (c < 8) val = row[CH + (c - 5)];                       // v_normals
            else {
                const int k = c - GEO;
                put = k < nch;
                val = row[k < CH ? k : 0];
                col = GEO + k;
            }
            if (put) atomic_add_f32(a.v_rows + (size_t)__float_as_int(s_N[s].w) * a.row_stride + col, val);
        }
        wave_lds_sync();
    };

    struct Fetched { float M[9]; float2 xy; float opac, n[3], cv[4]; };
    auto entry_of = [&](int32_t b) -> int32_t {
        const int32_t idx = range_end - 1 - BATCH * b - (int32_t)lane;
        return (b < n_batches && idx >= range_start) ? a.flatten_ids[idx] : -1;
    };
    auto fetch = [&](int32_t gi, Fetched &f) {
        if (gi < 0) return;
        const float *M = a.ray_transforms + 9 * (size_t)gi;
#pragma unroll
        for (int i = 0; i < 9; ++i) f.M[i] = M[i];
        f.xy   = reinterpret_cast<const float2 *>(a.means2d)[gi];
        f.opac = a.opacities[gi];
        const float *n = a.normals + 3 * (size_t)gi;
        f.n[0] = n[0]; f.n[1] = n[1]; f.n[2] = n[2];
        const float *cp = a.colors + (size_t)gi * a.cdim;
#pragma unroll
        for (int k = 0; k < 4; ++k) f.cv[k] = (k < CH && k < nch) ? cp[k] : 0.0f;
    };
    int32_t g_cur = entry_of(0), g_nxt = entry_of(1);
    Fetched f_cur{};
    fetch(g_cur, f_cur);

    for (int32_t b = 0; b < n_batches; ++b) {
        const int32_t batch_end = range_end - 1 - BATCH * b;
        int hitmask             = 0;
        {
            const int32_t idx = batch_end - (int32_t)lane;
            const int32_t gi  = g_cur;
            const Fetched f   = f_cur;
            if (gi >= 0) {
                s_A[lane] = make_float4(f.M[0], f.M[1], f.M[2], f.xy.x);
                s_B[lane] = make_float4(f.M[3], f.M[4], f.M[5], f.xy.y);
                s_C[lane] = make_float4(f.M[6], f.M[7], f.M[8], f.opac);
                s_N[lane] = make_float4(f.n[0], f.n[1], f.n[2], __int_as_float(gi));
                float4 za, zb, zc;
                stage_surfel(f.M, f.xy.x, f.xy.y, f.opac, tcx, tcy, za, zb, zc);
                s_Za[lane] = za; s_Zb[lane] = zb; s_Zc[lane] = zc;
                s_col[lane] = make_float4(f.cv[0], f.cv[1], f.cv[2], f.cv[3]);
                const float4 cu = surfel_cull_box(f.M, f.xy.x, f.xy.y, f.opac);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint32_t gq = q_first + (uint32_t)q;
                    const float rcx = (gq & 1u) ? 4.0f : -4.0f, rcy = (gq >> 1) ? 4.0f : -4.0f; // quadrant centre - tile centre
                    bool hit = idx <= qmax[q] && (fabsf(cu.x - (tcx + rcx)) - 3.5f <= cu.z)
                            && (fabsf(cu.y - (tcy + rcy)) - 3.5f <= cu.w);
                    if (hit) hit = surfel_reaches_rect(za, zb, zc, rcx, rcy, 3.5f, 3.5f);
                    hitmask |= hit ? (1 << q) : 0;
                }
            }
        }
        g_cur = g_nxt;
        fetch(g_cur, f_cur); // rows of batch b + 1: in flight while batch b is walked
        g_nxt = entry_of(b + 2);
        wave_lds_sync();

        const int32_t behind_s = __builtin_amdgcn_readfirstlane(batch_end);
        uint64_t todo          = __builtin_amdgcn_ballot_w64(hitmask != 0);
        while (todo) {
            const int32_t t = (int32_t)__builtin_ctzll(todo);
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t));
            const int qm       = __builtin_amdgcn_readlane(hitmask, t);
            const float4 Za = s_Za[t], Zb = s_Zb[t], Zc = s_Zc[t];
            const float4 nr    = s_N[t];
            const float4 c4    = s_col[t];
            const float colv[4] = {c4.x, c4.y, c4.z, c4.w};
            const float nrv[3]  = {nr.x, nr.y, nr.z};
            const float opac    = Zc.w;
            const int32_t list_idx = behind_s - t;
            float sum[KQ * 4];
#pragma unroll
            for (int k = 0; k < KQ * 4; ++k) sum[k] = 0.0f;
            bool contributed = false; // wave-uniform
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(qm & (1 << q))) continue; // scalar
                const int gq   = (int)q_first + q;
                const float qx = (float)(((gq & 1) << 3) | (int)lqx) - 7.5f, qy = (float)(((gq >> 1) << 3) | (int)lqy) - 7.5f;
                const Surfel sf  = eval_surfel(Za, Zb, Zc, qx, qy);
                const bool valid = (list_idx <= bin_final[q]) && sf.valid; // outside pixels: bin_final = -1
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
                contributed = true;
                // branch-free: invalid lanes run with alpha = vis = 0 (every contribution becomes exactly 0)
                const float alpha = valid ? sf.alpha : 0.0f;
                const float vis   = valid ? sf.vis : 0.0f;
                const float ra    = __builtin_amdgcn_rcpf(fmaxf(kMinOneMinusAlpha, 1.0f - alpha));
                T[q]             *= ra;
                const float fac   = alpha * T[q];
                float cv = 0.0f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    sum[k] = fmaf(fac, v_c[q][k], sum[k]);
                    cv     = fmaf(colv[k], v_c[q][k], cv);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sum[CH + k] = fmaf(fac, v_n[q][k], sum[CH + k]);
                    cv          = fmaf(nrv[k], v_n[q][k], cv);
                }
                if (dist) {
                    float depth = colv[0];
#pragma unroll
                    for (int k = 1; k < CH; ++k)
                        if (k == nch - 1) depth = colv[k];
                    cv = fmaf(fmaf(depth, dP[q], -dQ[q]), vd2[q], cv);                          // cv + dl_dw v_distort
                    const float v_depth_ch = fac * vd2[q] * (fmaf(-2.0f, T[q], c2aw[q]) + fac); // extra grad of the last channel
                    dP[q] = fmaf(-2.0f, fac, dP[q]);
                    dQ[q] = fmaf(-2.0f * depth, fac, dQ[q]);
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (k == nch - 1) sum[k] += v_depth_ch;
                }
                const float v_alpha = fmaf(ra, tail_term[q] - behind[q], cv * T[q]);
                behind[q]           = fmaf(fac, cv, behind[q]);

                const float ov       = opac * vis;
                const bool unclamped = valid && (ov <= kMaxAlpha);
                const float v_G      = unclamped ? opac * v_alpha : 0.0f;
                const bool use3d     = sf.gw3 <= sf.gw2;
                const float g3   = use3d ? v_G * -vis : 0.0f;
                const float a_   = g3 * sf.sx * sf.rcz_inv, b_ = g3 * sf.sy * sf.rcz_inv;
                const float vrc[3] = {a_, b_, -(a_ * sf.sx + b_ * sf.sy)};
                const float plx = (float)(((gq & 1) << 3) | (int)lqx) + 0.5f, ply = (float)(((gq >> 1) << 3) | (int)lqy) + 0.5f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // g3 == 0 can still meet inf/NaN geometry on invalid lanes: select, do not multiply
                    const float vr   = (use3d && unclamped) ? vrc[k] : 0.0f;
                    sum[CH + 3 + k] += vr;
                    sum[CH + 6 + k]  = fmaf(plx, vr, sum[CH + 6 + k]);
                    sum[CH + 9 + k]  = fmaf(ply, vr, sum[CH + 9 + k]);
                }
                const float g2 = (!use3d && unclamped) ? v_G * (-vis * kFilterInvSquare2DGS) : 0.0f;
                sum[CH + 12] = fmaf(g2, sf.dx, sum[CH + 12]);
                sum[CH + 13] = fmaf(g2, sf.dy, sum[CH + 13]);
                sum[CH + 14] += unclamped ? vis * v_alpha : 0.0f;
            }
            if (!contributed) continue;
            // ONE reduction per (tile, surfel): row r of group j ends up with the total of value 4 j + r
            float mine = 0.0f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float r = wave_sum4_scatter(sum[4 * j], sum[4 * j + 1], sum[4 * j + 2], sum[4 * j + 3]);
                if ((int)(lane & 15u) == j) mine = r;
            }
            const int vidx = 4 * (int)(lane & 15u) + (int)(lane >> 4);
            if ((int)(lane & 15u) < KQ && vidx < K) s_tot[slot * TP + vidx] = mine;
            if (slot == 0) slot_t[0] = t;
            else slot_t[1] = t;
            if (++slot == SLOTS) {
                flush(SLOTS);
                slot = 0;
            }
        }
        if (slot) { // the staged rows the open slots point into are overwritten by the next batch
            flush(slot);
            slot = 0;
        }
    }
}

template <int CH>
static int launch2_fwd(const Raster2DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid  = ((n_blocks + 7u) / 8u) * 8u;
    const uint32_t block = a.tile_size <= 8 ? 64u : 256u;
    const size_t smem    = kBatch2 * (5 * sizeof(float4) + sizeof(float) * CH);
    raster2d_fwd_kernel<CH><<<dim3(grid), dim3(block), smem, stream>>>(a);
    return check_launch("raster2d_fwd");
}

template <int CH, bool ABS>
static int launch2_bwd(const Raster2DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0 || a.n_isects == 0) return GSX_OK;
    const uint32_t grid  = ((n_blocks + 7u) / 8u) * 8u;
    const uint32_t block = a.tile_size <= 8 ? 64u : 256u;
    if constexpr (!ABS && CH <= 4) {
        if (raster2d_bwd_m_applies(a, ABS)) return raster2d_bwd_m_launch(a, stream); // the sums as one matrix product
        // w: one wave per tile: on c5 it issues 19 % fewer VALU and half the LDS instructions than the reduction kernel but
        // needs 246 VGPRs - two waves per SIMD, 75 % VALU issue where the reduction kernel (five waves) runs at 97 % - and
        // takes the same time (profiles/r08_ab.md #19, #20). h keeps half of that saving at 152 VGPRs, three waves per SIMD
        // GSX_RASTER2D_BWD = h (default since round 5: one wave per HALF tile, 1.75 ms on c5) | r (the four-wave reduction
        // kernel below, 1.83) | w (one wave per tile, 1.83) | m (raster2d_bwd_m.hip, 2.74); read once per process
        static const char use = [] {
            const char *e = getenv("GSX_RASTER2D_BWD");
            if (e && (e[0] == 'w' || e[0] == 'W')) return 'w';
            if (e && (e[0] == 'r' || e[0] == 'R')) return 'r';
            if (e && (e[0] == 'm' || e[0] == 'M')) return 'm';
            return 'h';
        }();
        if (a.tile_size == 16 && use == 'w') {
            if (a.v_render_distort) raster2d_bwd_w_kernel<CH, true, 4><<<dim3(grid), dim3(64), Bwd2WCfg<CH>::smem, stream>>>(a);
            else raster2d_bwd_w_kernel<CH, false, 4><<<dim3(grid), dim3(64), Bwd2WCfg<CH>::smem, stream>>>(a);
            return check_launch("raster2d_bwd_w");
        }
        if (a.tile_size == 16 && use == 'h') { // one wave per half tile
            if (a.v_render_distort) raster2d_bwd_w_kernel<CH, true, 2><<<dim3(2u * grid), dim3(64), Bwd2WCfg<CH>::smem, stream>>>(a);
            else raster2d_bwd_w_kernel<CH, false, 2><<<dim3(2u * grid), dim3(64), Bwd2WCfg<CH>::smem, stream>>>(a);
            return check_launch("raster2d_bwd_h");
        }
    }
    raster2d_bwd_kernel<CH, ABS><<<dim3(grid), dim3(block), Bwd2Cfg<CH, ABS>::smem, stream>>>(a);
    return check_launch("raster2d_bwd");
}

template <bool ABS>
static int dispatch2_bwd(const Raster2DArgs &a, hipStream_t s)
{
    const uint32_t n = a.cdim;
    if (n <= 1) return launch2_bwd<1, ABS>(a, s);
    if (n <= 2) return launch2_bwd<2, ABS>(a, s);
    if (n <= 3) return launch2_bwd<3, ABS>(a, s);
    if (n <= 4) return launch2_bwd<4, ABS>(a, s);
    if (n <= 8) return launch2_bwd<8, ABS>(a, s);
    if (n <= 16) return launch2_bwd<16, ABS>(a, s);
    return launch2_bwd<32, ABS>(a, s);
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_raster2d_fwd(const float *means2d, const float *ray_transforms, const float *colors,
                                const float *opacities, const float *normals, const float *backgrounds,
                                const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
                                uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int distloss, float *render_colors,
                                float *render_alphas, float *render_normals, float *render_distort, float *render_median,
                                int32_t *last_ids, int32_t *median_ids, void *stream)
{
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster2d_fwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1 && cdim <= 32, "gsx_raster2d_fwd: unsupported number of channels %u (1..32)", cdim);
    GSX_REQUIRE(render_colors && render_alphas && render_normals && render_distort && render_median && last_ids
                && median_ids, "gsx_raster2d_fwd: null output");
    GSX_REQUIRE(n_isects == 0 || (means2d && ray_transforms && colors && opacities && normals && flatten_ids),
                "gsx_raster2d_fwd: null input");
    GSX_REQUIRE(isect_offsets != nullptr || n_images * tile_w * tile_h == 0, "gsx_raster2d_fwd: null isect_offsets");
    Raster2DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height; a.tile_size = tile_size;
    a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim; a.distloss = distloss;
    a.means2d = means2d; a.ray_transforms = ray_transforms; a.colors = colors; a.opacities = opacities;
    a.normals = normals; a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets;
    a.flatten_ids = flatten_ids;
    a.render_colors = render_colors; a.render_alphas = render_alphas; a.render_normals = render_normals;
    a.render_distort = render_distort; a.render_median = render_median; a.last_ids = last_ids; a.median_ids = median_ids;
    hipStream_t s = (hipStream_t)stream;
    if (cdim <= 1) return launch2_fwd<1>(a, s);
    if (cdim <= 2) return launch2_fwd<2>(a, s);
    if (cdim <= 3) return launch2_fwd<3>(a, s);
    if (cdim <= 4) return launch2_fwd<4>(a, s);
    if (cdim <= 8) return launch2_fwd<8>(a, s);
    if (cdim <= 16) return launch2_fwd<16>(a, s);
    return launch2_fwd<32>(a, s);
}

extern "C" int gsx_raster2d_bwd(const float *means2d, const float *ray_transforms, const float *colors,
                                const float *opacities, const float *normals, const float *backgrounds,
                                const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                const float *render_colors, const float *render_alphas, const int32_t *last_ids,
                                const int32_t *median_ids, const float *v_render_colors, const float *v_render_alphas,
                                const float *v_render_normals, const float *v_render_distort,
                                const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                                uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                                int has_abs, float *v_rows, uint32_t row_stride, void *stream)
{
    return gsx_raster2d_bwd_ws(means2d, ray_transforms, colors, opacities, normals, backgrounds, masks, isect_offsets, flatten_ids,
                               render_colors, render_alphas, last_ids, median_ids, v_render_colors, v_render_alphas,
                               v_render_normals, v_render_distort, v_render_median, n_images, n_isects, cdim, width, height,
                               tile_size, tile_w, tile_h, has_abs, v_rows, row_stride, nullptr, 0, stream);
}

// gsx_raster2d_bwd with a workspace of gsx_raster3d_bwd_workspace_bytes(n_images, tile_w, tile_h): tiles longest-first
// (csrc/tile_order.hip; same rule and same results as gsx_raster3d_bwd_ws)
extern "C" int gsx_raster2d_bwd_ws(const float *means2d, const float *ray_transforms, const float *colors,
                                   const float *opacities, const float *normals, const float *backgrounds,
                                   const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                   const float *render_colors, const float *render_alphas, const int32_t *last_ids,
                                   const int32_t *median_ids, const float *v_render_colors, const float *v_render_alphas,
                                   const float *v_render_normals, const float *v_render_distort,
                                   const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                                   uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                                   int has_abs, float *v_rows, uint32_t row_stride, void *workspace, int64_t workspace_bytes,
                                   void *stream)
{
    return gsx_raster2d_bwd_fill(means2d, ray_transforms, colors, opacities, normals, backgrounds, masks, isect_offsets, flatten_ids,
                                 render_colors, render_alphas, last_ids, median_ids, v_render_colors, v_render_alphas,
                                 v_render_normals, v_render_distort, v_render_median, n_images, n_isects, cdim, width, height,
                                 tile_size, tile_w, tile_h, has_abs, v_rows, row_stride, 0, workspace, workspace_bytes, stream);
}

// gsx_raster2d_bwd_ws for gradient rows that are NOT zero-filled yet: the call fills v_rows_to_fill rows of row_stride floats
// itself - inside the tile-order cost kernel when one is launched, with a memset otherwise (as gsx_raster3d_bwd_fill).
extern "C" int gsx_raster2d_bwd_fill(const float *means2d, const float *ray_transforms, const float *colors,
                                   const float *opacities, const float *normals, const float *backgrounds,
                                   const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                   const float *render_colors, const float *render_alphas, const int32_t *last_ids,
                                   const int32_t *median_ids, const float *v_render_colors, const float *v_render_alphas,
                                   const float *v_render_normals, const float *v_render_distort,
                                   const float *v_render_median, uint32_t n_images, uint32_t n_isects, uint32_t cdim,
                                   uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                                   int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill, void *workspace,
                                   int64_t workspace_bytes,
                                   void *stream)
{
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster2d_bwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1 && cdim <= 32, "gsx_raster2d_bwd: unsupported number of channels %u (1..32)", cdim);
    int64_t fill_bytes = v_rows_to_fill > 0 ? v_rows_to_fill * (int64_t)row_stride * 4 : 0;
    auto fill_now      = [&]() -> int { // rows not filled by a kernel of this call
        if (fill_bytes > 0 && v_rows && hipMemsetAsync(v_rows, 0, (size_t)fill_bytes, (hipStream_t)stream) != hipSuccess) {
            set_last_error("gsx_raster2d_bwd_fill: memset failed");
            return GSX_ERR_LAUNCH;
        }
        fill_bytes = 0;
        return GSX_OK;
    };
    if (n_isects == 0) return fill_now(); // no intersections: nothing to add to the (zero-filled, possibly empty) gradient rows
    GSX_REQUIRE(v_rows, "gsx_raster2d_bwd: null gradient output");
    GSX_REQUIRE(row_stride >= 17u + (has_abs ? 2u : 0u) + cdim, "gsx_raster2d_bwd: row_stride %u too small", row_stride);
    GSX_REQUIRE(n_isects == 0 || (means2d && ray_transforms && colors && opacities && normals && flatten_ids
                                  && render_colors && render_alphas && last_ids && median_ids && v_render_colors
                                  && isect_offsets), // v_render_alphas / _normals / _distort / _median: NULL = zeros
                "gsx_raster2d_bwd: null input");
    Raster2DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height; a.tile_size = tile_size;
    a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.ray_transforms = ray_transforms; a.colors = colors; a.opacities = opacities;
    a.normals = normals; a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets;
    a.flatten_ids = flatten_ids;
    a.render_colors = const_cast<float *>(render_colors); a.render_alphas = const_cast<float *>(render_alphas);
    a.last_ids = const_cast<int32_t *>(last_ids); a.median_ids = const_cast<int32_t *>(median_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas; a.v_render_normals = v_render_normals;
    a.v_render_distort = v_render_distort; a.v_render_median = v_render_median;
    a.v_rows = v_rows; a.row_stride = row_stride;
    hipStream_t s = (hipStream_t)stream;
    {
        int rc               = GSX_OK;
        const bool in_kernel = fill_bytes > 0 && (reinterpret_cast<uintptr_t>(v_rows) & 15u) == 0;
        a.tile_order = build_tile_order(isect_offsets, last_ids, n_images, tile_size, tile_w, tile_h, width, height, n_isects,
                                        workspace, workspace_bytes, s, &rc, in_kernel ? v_rows : nullptr, fill_bytes);
        if (rc != GSX_OK) return rc;
        if (a.tile_order && in_kernel) fill_bytes = 0; // done by the cost kernel
    }
    if (int rc = fill_now(); rc != GSX_OK) return rc;
    return has_abs ? dispatch2_bwd<true>(a, s) : dispatch2_bwd<false>(a, s);
}
