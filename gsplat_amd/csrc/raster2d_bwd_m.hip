// 2DGS (surfel) compositing, backward, with the per-(tile, surfel) sums on the matrix cores (gfx950).
// Launched by gsx_raster2d_bwd_ws (raster2d.hip) for <= 4 channels, 16 x 16 tiles, no absgrad; replaces the same reference
// kernel, RasterizeToPixels2DGSSerialBatchBwd.cu:41-700.
//
// The reduction kernel (raster2d_bwd_kernel) spends 62 of its ~150 vector instructions per (wave, surfel) pair on wave
// reductions of K = 15 + D per-pixel values, and the kernel runs at 0.97 of VALU issue: 1.83 ms on c5, the one target that
// every round missed. All of those sums are (one scalar per pixel and surfel) x (a quantity of the pixel alone):
//     colours, normals        sum_p fac(p, s) * [v_c(p) | v_n(p)]                                   D + 3 sums
//     opacity, depth addend   sum_p w(p, s),  sum_p vdc(p, s)                                       2
//     ray-transform gradient  sum_p (a, b, c)(p, s) * (1, x_p, y_p)   (three moments of vrc, raster2d.hip)   9
//     low-pass branch         sum_p g2(p, s) * (1, x_p, y_p)          (v_mean2d = mean' G0 - (Gx, Gy))        3
// i.e. ONE matrix product per group of surfels:  [rows: v_c | v_n | 1 | x | y] x [pixels] . [pixels] x [columns: (surfel,
// scalar)]  with seven scalars per (pixel, surfel): fac, w, vdc, a, b, c, g2. The pixel loop parks them in a wave-private LDS
// matrix (row = (slot, scalar), column = pixel); every FOUR surviving surfels the wave multiplies with
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: exact fp32; M = the ten pixel-side rows, N = 28 (slot, scalar) columns
// in two blocks, K = 4 pixels per instruction, 16 k-steps over the wave's 64 pixels = 32 instructions per four surfels on
// the MFMA pipe, beside the vector ALU). The useful entries of the result (21 per surfel) are added to the per-tile LDS
// accumulator row of the surfel (the reduction kernel's layout + the three low-pass moments) and flushed per batch as there.
// No cross-lane reduction in the pixel loop. Four waves per tile (wave = 8 x 8 quadrant, lane = pixel), 64 staged surfels
// per batch. The pixel-side rows use TILE-LOCAL pixel centres (0.5 .. 15.5), like the reduction kernel's moments.
//
// MEASURED AND NOT THE DEFAULT (round 5, profiles/r09_ab.md): parity-green on its first run (38 / 38 of tests/test_gpu_2dgs.py
// under GSX_RASTER2D_BWD=m) and SLOWER - 2.74 ms on c5 where the reduction kernel takes 1.83. What made the same scheme win
// at 32 colour channels (raster3d_bwd_m.hip) is missing here: the product is narrow - 21 useful entries of the 10 x 7 block
// a surfel gets, ten of sixteen rows, four surfels per multiplication because seven parked scalars per (pixel, surfel) are
// 7.6 KB of LDS per wave already - so a pair costs eight MFMAs (256 cycles of the matrix pipe), seven LDS stores and two
// slow LDS float atomics, at three waves per SIMD. Kept selectable and covered by tests/test_gpu_variants.py.
#include <cstdlib>

#include "raster2d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef GSX_BWD2_M_WAVES
#define GSX_BWD2_M_WAVES 3
#endif

template <int CH>
struct Bwd2MCfg {
    static constexpr int BATCH = 64;           // staged surfels per batch: one per lane of the culling test
    static constexpr int SLOTS = 4;            // surfels per multiplication
    static constexpr int NSC   = 7;            // parked scalars per (pixel, surfel): fac, w, vdc, a, b, c, g2
    static constexpr int WP    = 68;           // floats per parked row: 64 pixels + 4
    // accumulator row: [0,CH) colours | CH..CH+2 normals | CH+3..5 S0 | CH+6..8 S_lx | CH+9..11 S_ly | CH+12..14 G0 Gx Gy (low-pass
    // moments) | CH+15 opacity
    static constexpr int K     = CH + 16;
    static constexpr int KP    = (K | 1);
    static constexpr size_t smem = (size_t)BATCH * (8 * sizeof(float4) + 2 * sizeof(int32_t) + sizeof(float) * (CH + KP))
                                   + sizeof(float) * 4 * SLOTS * NSC * WP;
};

template <int CH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSX_BWD2_M_WAVES)))
raster2d_bwd_m_kernel(const Raster2DArgs a)
{
    using Cfg           = Bwd2MCfg<CH>;
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int NSC   = Cfg::NSC;
    constexpr int WP    = Cfg::WP;
    constexpr int KP    = Cfg::KP;
    static_assert(CH <= 4, "rows 0 .. 3 of the pixel-side operand are the colour cotangents");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_A      = reinterpret_cast<float4 *>(smem_raw); // u_M, mean.x
    float4 *s_B      = s_A + BATCH;                          // v_M, mean.y
    float4 *s_C      = s_B + BATCH;                          // w_M, opacity
    float4 *s_N      = s_C + BATCH;                          // normal
    float4 *s_cull   = s_N + BATCH;                          // surfel_cull_box
    float4 *s_Za     = s_cull + BATCH;                       // evaluation form (stage_surfel): zeta at the tile centre | Z1 | Z2
    float4 *s_Zb     = s_Za + BATCH;
    float4 *s_Zc     = s_Zb + BATCH;
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_Zc + BATCH);
    int32_t *s_touch = s_id + BATCH;
    float *s_col     = reinterpret_cast<float *>(s_touch + BATCH); // [BATCH][CH]
    float *s_acc     = s_col + BATCH * CH;                         // [BATCH][KP]
    float *s_w       = s_acc + BATCH * KP;                         // [4 waves][SLOTS * NSC][WP]

    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t n_blocks        = tiles_per_image * a.n_images;
    const uint32_t slot_idx        = xcd_remap(blockIdx.x, n_blocks);
    if (slot_idx >= n_blocks) return;
    const uint32_t blk = a.tile_order ? (uint32_t)a.tile_order[slot_idx] : slot_idx;
    const uint32_t image_id = blk / tiles_per_image, tile_id = blk % tiles_per_image;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;
    const uint32_t tile_x = tile_id % a.tile_w, tile_y = tile_id / a.tile_w;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lx, ly;
    tile_pixel(tid, 16u, lx, ly);
    const uint32_t ox = tile_x * 16u + lx, oy = tile_y * 16u + ly;
    const bool inside = (ox < a.width) && (oy < a.height);
    const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
    const float tcx = (float)(tile_x * 16u) + 8.0f, tcy = (float)(tile_y * 16u) + 8.0f;
    const float qx = (float)lx - 7.5f, qy = (float)ly - 7.5f; // pixel centre relative to the tile centre (exact)
    const size_t pix = inside ? ((size_t)image_id * a.height + oy) * a.width + ox : 0;
    const int nch    = (int)a.cdim;
    const float X0 = (float)(tile_x * 16u), Y0 = (float)(tile_y * 16u); // tile origin

    const int32_t range_start = a.isect_offsets[blk];
    int32_t range_end         = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    if (range_end <= range_start) return;

    const float T_final     = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    float T                 = T_final;
    const int32_t bin_final = inside ? a.last_ids[pix] : -1;
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    { // nothing behind the last contributor of the whole tile is staged
        int32_t *s_m = reinterpret_cast<int32_t *>(smem_raw);
        if (lane == 0) s_m[wave] = wave_bin_final;
        __syncthreads();
        const int32_t m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
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
    const float v_a = (inside && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
    // the median depth's cotangent: once per pixel, to the recorded surfel's depth channel (see raster2d_bwd_kernel)
    if (inside && T_final < 1.0f) {
        const float v_median = a.v_render_median ? a.v_render_median[pix] : 0.0f;
        if (v_median != 0.0f)
            atomic_add_f32(a.v_rows + (size_t)a.flatten_ids[a.median_ids[pix]] * a.row_stride + 17 + nch - 1, v_median);
    }
    float bg_dot = 0.0f;
    if (a.backgrounds) {
        const float *bg = a.backgrounds + (size_t)image_id * a.cdim;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < nch) bg_dot += bg[k] * v_c[k];
    }
    const float tail_term = T_final * (v_a - bg_dot); // what lies behind the whole list
    float behind          = 0.0f;                     // B (see raster2d_bwd_kernel), with the distortion buffer inside
    const bool dist = a.v_render_distort != nullptr;
    float vd2 = 0.f, c2aw = 0.f, dP = 0.f, dQ = 0.f; // distortion loss (raster2d_bwd_kernel): 2 v_distort | 2 - accum_w | P | Q
    if (dist && inside) {
        vd2  = 2.0f * a.v_render_distort[pix];
        dQ   = a.render_colors[pix * a.cdim + nch - 1];
        dP   = a.render_alphas[pix];
        c2aw = 2.0f - dP;
    }

    float *s_ww = s_w + wave * (SLOTS * NSC * WP); // this wave's parked matrix: row (slot * NSC + scalar), column = pixel
    const int bj = (int)(lane & 15u), bk = (int)(lane >> 4);
    // ---- the pixel-side operand (A of the MFMA: lane l supplies row l & 15, k = l >> 4), in registers for the whole tile:
    // row r of pixel 16 bk + s for k-step s; rows 0 .. 3 colour cotangents, 4 .. 6 normal cotangents, 7: 1, 8: x, 9: y (tile-local
    // pixel centre), 10 .. 15: 0
    float areg[16];
    {
        f32x4 *tmp = reinterpret_cast<f32x4 *>(s_ww); // [2][64]: (v_c 0..3) | (v_n 0..2, -) per pixel, through the still unused parked region
        float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CH; ++k) c4[k] = v_c[k];
        tmp[lane]      = f32x4{c4[0], c4[1], c4[2], c4[3]};
        tmp[64 + lane] = f32x4{v_n[0], v_n[1], v_n[2], 0.0f};
        wave_lds_sync();
        const float *tf = reinterpret_cast<const float *>(tmp);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int pp = 16 * bk + s; // pixel pp of this wave = its lane pp: (pp & 7, pp >> 3) inside the quadrant
            const float plx = (float)(((wave & 1u) << 3) | (uint32_t)(pp & 7)) + 0.5f;
            const float ply = (float)(((wave >> 1) << 3) | (uint32_t)(pp >> 3)) + 0.5f;
            float val = 0.0f;
            if (bj < 4) val = tf[pp * 4 + bj];
            else if (bj < 7) val = tf[(64 + pp) * 4 + (bj - 4)];
            else if (bj == 7) val = 1.0f;
            else if (bj == 8) val = plx;
            else if (bj == 9) val = ply;
            areg[s] = val;
        }
        wave_lds_sync();
    }
    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int k = 0; k < KP; ++k) s_acc[s * KP + k] = 0.0f;
        s_touch[s] = 0;
    }

    for (int32_t b = 0; b < n_batches; ++b) {
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);
        { // staging: threads 0 .. 63 one surfel each (the rows are short: 9 + 2 + 1 + 3 + CH floats)
            const int s = (int)tid;
            const int32_t idx = batch_end - s;
            if (s < BATCH && idx >= range_start) {
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
        // pixels whose last contributor lies in front of this whole batch take no part in it: the rectangle is the box of the others
        const WaveRect rect = wave_pixel_rect(inside && bin_final >= batch_end - (batch_size - 1), px, py);
        bool hit = false;
        if ((int32_t)lane >= t_first && (int32_t)lane < batch_size) {
            const float4 cu = s_cull[lane];
            hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
            if (hit) hit = surfel_reaches_rect(s_Za[lane], s_Zb[lane], s_Zc[lane], rect.cx - tcx, rect.cy - tcy, rect.hw, rect.hh);
        }
        uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
        while (todo) { // the survivors FOUR at a time, back to front
            int32_t ts[SLOTS];
            int n = 0;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                ts[k] = 0;
                if (todo) {
                    ts[k] = (int32_t)__builtin_ctzll(todo);
                    asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(ts[k])); // todo &= todo - 1 in one scalar instruction
                    n = k + 1;
                }
            }
            uint32_t live = 0; // slots some pixel of the wave took (wave-uniform)
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                float sc[NSC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // fac, w, vdc, a, b, c, g2
                if (k < n) { // wave-uniform
                    const int32_t t = ts[k];
                    const float4 C  = s_Zc[t];
                    const Surfel s  = eval_surfel(s_Za[t], s_Zb[t], C, qx, qy);
                    const bool valid = inside && (batch_end - t <= bin_final) && s.valid;
                    if (__builtin_amdgcn_ballot_w64(valid) != 0ull) { // wave-uniform
                        live |= 1u << k;
                        // branch-free: invalid lanes run with alpha = vis = 0 (every contribution becomes exactly 0)
                        const float alpha = valid ? s.alpha : 0.0f;
                        const float vis   = valid ? s.vis : 0.0f;
                        const float opac  = C.w;
                        const float ra  = __builtin_amdgcn_rcpf(fmaxf(kMinOneMinusAlpha, 1.0f - alpha));
                        T              *= ra;
                        const float fac = alpha * T;
                        float cv = 0.0f;
#pragma unroll
                        for (int c = 0; c < CH; ++c) cv = fmaf(s_col[t * CH + c], v_c[c], cv);
                        const float4 nr = s_N[t];
                        cv = fmaf(nr.x, v_n[0], fmaf(nr.y, v_n[1], fmaf(nr.z, v_n[2], cv)));
                        float vdc = 0.0f;
                        if (dist) {
                            const float depth = s_col[t * CH + nch - 1];
                            cv  = fmaf(fmaf(depth, dP, -dQ), vd2, cv);               // cv + dl_dw v_distort
                            vdc = fac * vd2 * (fmaf(-2.0f, T, c2aw) + fac);          // 2 fac (2 - 2 T - accum_w + fac) v_distort
                            dP  = fmaf(-2.0f, fac, dP);
                            dQ  = fmaf(-2.0f * depth, fac, dQ);
                        }
                        const float v_alpha = fmaf(ra, tail_term - behind, cv * T);
                        behind              = fmaf(fac, cv, behind);
                        const float ov       = opac * vis;
                        const bool unclamped = valid && (ov <= kMaxAlpha);
                        const float v_G      = unclamped ? opac * v_alpha : 0.0f;
                        const bool use3d     = s.gw3 <= s.gw2;
                        // 3D branch: through s = zeta.xy / zeta.z; g3 == 0 can still meet inf / NaN geometry on invalid lanes: select
                        const float g3 = use3d ? v_G * -vis : 0.0f;
                        const float a_ = g3 * s.sx * s.rcz_inv, b_ = g3 * s.sy * s.rcz_inv;
                        const bool on3 = use3d && unclamped;
                        sc[0] = fac;
                        sc[1] = unclamped ? vis * v_alpha : 0.0f;
                        sc[2] = vdc;
                        sc[3] = on3 ? a_ : 0.0f;
                        sc[4] = on3 ? b_ : 0.0f;
                        sc[5] = on3 ? -(a_ * s.sx + b_ * s.sy) : 0.0f;
                        sc[6] = (!use3d && unclamped) ? v_G * (-vis * kFilterInvSquare2DGS) : 0.0f; // low-pass branch
                    }
                }
                // rows of a slot that was not taken (or does not exist) are zero: they meet finite pixel-side rows
#pragma unroll
                for (int j = 0; j < NSC; ++j) s_ww[(k * NSC + j) * WP + (int)lane] = sc[j];
            }
            wave_lds_sync();
            if (live) {
                // D[row r][column (slot, scalar)] = sum over the wave's pixels of row_r(pixel) * scalar(pixel, slot): two column
                // blocks (28 columns), k-step s = pixels {s, 16 + s, 32 + s, 48 + s}; B of lane l: column l & 15 (+ 16), pixel 16 bk + s
                f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int nbk = 0; nbk < 2; ++nbk) {
                    const int col = 16 * nbk + bj;
                    float bv[16];
                    const f32x4 *pb = reinterpret_cast<const f32x4 *>(s_ww + (col < SLOTS * NSC ? col : 0) * WP + 16 * bk);
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f32x4 x = pb[h];
                        bv[4 * h] = x.x; bv[4 * h + 1] = x.y; bv[4 * h + 2] = x.z; bv[4 * h + 3] = x.w;
                    }
#pragma unroll
                    for (int s = 0; s < 16; ++s) d[nbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s], bv[s], d[nbk], 0, 0, 0);
                }
                // results: column bj (+ 16), rows 4 bk + i. The useful ones go to the surfel's accumulator row (ds_add_f32)
#pragma unroll
                for (int nbk = 0; nbk < 2; ++nbk) {
                    const int col = 16 * nbk + bj;
                    const int slot = col / NSC, scalar = col - slot * NSC;
                    const bool col_ok = col < SLOTS * NSC && ((live >> slot) & 1u);
                    const int32_t t_g = slot == 0 ? ts[0] : (slot == 1 ? ts[1] : (slot == 2 ? ts[2] : ts[3]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * bk + i;
                        int idx = -1; // where (row, scalar) lives in the accumulator row
                        if (scalar == 0) idx = row < 4 ? (row < nch ? row : -1) : (row < 7 ? CH + (row - 4) : -1);
                        else if (scalar == 1) idx = row == 7 ? CH + 15 : -1;
                        else if (scalar == 2) idx = row == 7 ? nch - 1 : -1; // the depth channel's extra gradient
                        else if (scalar < 6) idx = (row >= 7 && row < 10) ? CH + 3 + 3 * (row - 7) + (scalar - 3) : -1;
                        else idx = (row >= 7 && row < 10) ? CH + 12 + (row - 7) : -1;
                        if (col_ok && idx >= 0) atomicAdd(&s_acc[t_g * KP + idx], d[nbk][i]);
                    }
                    if (col_ok && scalar == 0 && bk == 0) s_touch[t_g] = 1;
                }
            }
            wave_lds_sync(); // the next group overwrites the parked rows
        }
        __syncthreads();

        // transposed flush into the AoS gradient rows (raster2d_bwd_kernel): element e -> (surfel s, output column c)
        constexpr int GEO  = 17;
        constexpr int NCOL = GEO + CH;
        for (int e = (int)tid; e < batch_size * NCOL; e += (int)blockDim.x) {
            const int s = e / NCOL, c = e - s * NCOL;
            if (!s_touch[s]) continue;
            const float *row = s_acc + s * KP;
            float val;
            int col = c;
            if (c < 2) {
                // low-pass branch: v_mean2d = sum g2 (mean - pixel) = mean' G0 - (Gx, Gy) with mean' and the pixel centres tile-local
                const float G0 = row[CH + 12];
                val = (c == 0) ? fmaf(s_A[s].w - X0, G0, -row[CH + 13]) : fmaf(s_B[s].w - Y0, G0, -row[CH + 14]);
            } else if (c == 2) val = row[CH + 15]; // v_opacities
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
                if (r == 0) val = (v1 * s02 - v2 * s01) - (w1 * sy2 - w2 * sy1);      // v_uM = v x S0 - w x Sy
                else if (r == 1) val = (s01 * u2 - s02 * u1) - (sx1 * w2 - sx2 * w1); // v_vM = S0 x u - Sx x w
                else val = (sx1 * v2 - sx2 * v1) + (u1 * sy2 - u2 * sy1);             // v_wM = Sx x v + u x Sy
                if (c < 5) val *= C4.z;
            } else if (c < 8) val = row[CH + (c - 5)]; // v_normals
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

// GSX_RASTER2D_BWD = r (reduction kernel) | w (one wave per tile) | m (this kernel); read once per process
static char bwd2_variant()
{
    static const char v = [] {
        const char *e = getenv("GSX_RASTER2D_BWD");
        if (e && (e[0] == 'r' || e[0] == 'R')) return 'r';
        if (e && (e[0] == 'w' || e[0] == 'W')) return 'w';
        if (e && (e[0] == 'm' || e[0] == 'M')) return 'm';
        return GSX_RASTER2D_BWD_DEFAULT;
    }();
    return v;
}
bool raster2d_bwd_m_applies(const Raster2DArgs &a, bool has_abs)
{
    return bwd2_variant() == 'm' && !has_abs && a.tile_size == 16 && a.cdim <= 4;
}
int raster2d_bwd_m_launch(const Raster2DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0 || a.n_isects == 0) return GSX_OK;
    const uint32_t grid = ((n_blocks + 7u) / 8u) * 8u;
    if (a.cdim <= 1) raster2d_bwd_m_kernel<1><<<dim3(grid), dim3(256), Bwd2MCfg<1>::smem, stream>>>(a);
    else if (a.cdim <= 2) raster2d_bwd_m_kernel<2><<<dim3(grid), dim3(256), Bwd2MCfg<2>::smem, stream>>>(a);
    else if (a.cdim <= 3) raster2d_bwd_m_kernel<3><<<dim3(grid), dim3(256), Bwd2MCfg<3>::smem, stream>>>(a);
    else raster2d_bwd_m_kernel<4><<<dim3(grid), dim3(256), Bwd2MCfg<4>::smem, stream>>>(a);
    return check_launch("raster2d_bwd_m");
}

} // namespace gsx
