// 3DGS alpha compositing, backward (gfx950).
// C-ABI entry: gsx_raster3d_bwd  (replaces torch op gsplat::rasterize_to_pixels_3dgs_bwd,
// reference host fn gsplat/cuda/csrc/Rasterization.cpp:484-587, kernel
// RasterizeToPixels3DGSSerialBatchBwd.cu:41-320, math RasterizeToPixels3DGSDevice.cuh:105-173).
//
// Gradient accumulation is re-designed for wave64 + 160 KiB LDS instead of the reference's
// "32-lane reduce, then 9+D global atomics per warp per Gaussian":
//   1. each lane computes its pixel's contribution to one Gaussian as K = D+6 [+2 absgrad] values:
//      D colour terms and the six MOMENTS of w = v_sigma over the pixel offsets d = mean - pixel
//      (sum w dx^2, w dx dy, w dy^2, w dx, w dy, w); v_conics, v_means2d and v_opacities are linear in
//      those moments and are formed once per (tile, Gaussian) at flush time instead of once per pixel;
//   2. the wave reduces FOUR values at a time with permlane16/32 swaps + DPP (common.hpp), a single
//      left-over value with 6 DPP adds (row_bcast);
//   3. the K totals land in K different lanes, and ONE ds_add_f32 adds them into a per-tile LDS
//      accumulator row for that Gaussian (4 waves -> 4-way LDS contention at most);
//   4. after the batch the accumulator is flushed with global_atomic_add_f32 — one atomic per (tile, Gaussian,
//      component) instead of one per (warp, Gaussian, component) — into an ARRAY-OF-STRUCTURES gradient buffer
//      [R][6 (+2) + D], TRANSPOSED: consecutive lanes add consecutive floats of one row, so a wave-level atomic
//      touches ~6 cache lines instead of 64. Measured on c3: the per-tensor layout (64 lines per instruction)
//      cost 160 us of exposed L2-atomic time out of 707 us; the AoS flush costs < 5 us.
#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

template <int CH, bool ABS>
struct BwdCfg {
    static constexpr int K     = CH + 6 + (ABS ? 2 : 0);     // values reduced per Gaussian
    static constexpr bool TAIL1 = (K % 4) == 1;              // one left-over value: cheaper single reduction
    static constexpr int KQ    = TAIL1 ? K / 4 : (K + 3) / 4; // groups of four
    static constexpr int KP    = (K | 1);                    // odd row stride -> conflict-free flush
    static constexpr int BATCH = (CH >= 16) ? 128 : 256;     // LDS budget for wide channel chunks
    static constexpr size_t smem =
        (size_t)BATCH * (2 * sizeof(float4) + sizeof(float2) + sizeof(int32_t) * 2 + sizeof(float) * (CH + KP));
};

template <int CH, bool ABS>
__global__ void __launch_bounds__(256) raster3d_bwd_kernel(const Raster3DArgs a)
{
    using Cfg           = BwdCfg<CH, ABS>;
    constexpr int K     = Cfg::K;
    constexpr int KQ    = Cfg::KQ;
    constexpr int KP    = Cfg::KP;
    constexpr int BATCH = Cfg::BATCH;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ga     = reinterpret_cast<float4 *>(smem_raw);       // x, y, log2(opac), A   (stage_gaussian)
    float2 *s_gb     = reinterpret_cast<float2 *>(s_ga + BATCH);   // B, C
    float4 *s_cull   = reinterpret_cast<float4 *>(s_gb + BATCH);   // x, y, half extents of alpha>=1/255
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_cull + BATCH); // flatten id of the row
    int32_t *s_touch = s_id + BATCH;                               // any lane contributed?
    float *s_col     = reinterpret_cast<float *>(s_touch + BATCH); // [BATCH][CH]
    float *s_acc     = s_col + BATCH * CH;                         // [BATCH][KP]

    TileCtx tc;
    if (!tile_context(a, blockIdx.x, tc)) return;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image_id = tc.image_id, tile_id = tc.tile_id;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;

    const uint32_t tid  = threadIdx.x;
    const uint32_t lane = tid & 63u;

    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
    const bool inside  = prow >= 0;
    const float px     = (float)(tc.tile_x * a.tile_size + lx) + 0.5f;
    const float py     = (float)(tc.tile_y * a.tile_size + ly) + 0.5f;
    const size_t pix   = inside ? (size_t)prow : 0;

    const int32_t range_start = tc.range_start, range_end = tc.range_end;
    const int32_t n_batches   = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return;

    // per-pixel state
    const float T_final     = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    float T                 = T_final;
    const int32_t bin_final = inside ? a.last_ids[pix] : -1;
    float v_c[CH], buffer[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        v_c[k]    = (inside && k < (int)a.nch) ? a.v_render_colors[pix * a.cdim + a.ch_off + k] : 0.0f;
        buffer[k] = 0.0f;
    }
    // alpha-gradient term and background term belong to exactly one channel chunk / all chunks resp.
    const float v_a = (inside && a.first_chunk && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f; // null = zeros
    float bg_dot    = 0.0f; // sum_k bg_k * v_c_k (this chunk)
    if (a.backgrounds) {
        const float *bg = a.backgrounds + (size_t)image_id * a.cdim + a.ch_off;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch) bg_dot += bg[k] * v_c[k];
    }
    const float va_minus_bg      = v_a - bg_dot;
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    const WaveRect rect          = wave_pixel_rect(inside, px, py);

    // zero the accumulator rows this thread owns
    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int k = 0; k < KP; ++k) s_acc[s * KP + k] = 0.0f;
        s_touch[s] = 0;
    }

    for (int32_t b = 0; b < n_batches; ++b) {
        // back-to-front: slot 0 is the farthest-back Gaussian of this batch
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);

        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            const int32_t idx = batch_end - s;
            if (idx >= range_start) {
                const int32_t g  = a.flatten_ids[idx];
                const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                s_id[s]        = g;
                float4 ga;
                float2 gb;
                stage_gaussian(xy.x, xy.y, opac, ca, cb, cc, ga, gb);
                s_ga[s]        = ga;
                s_gb[s]        = gb;
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]      = make_float4(xy.x, xy.y, he.x, he.y);
                const float *c = a.colors + (size_t)g * a.cdim + a.ch_off;
#pragma unroll
                for (int k = 0; k < CH; ++k) s_col[s * CH + k] = (k < (int)a.nch) ? c[k] : 0.0f;
            }
        }
        __syncthreads();

        // Gaussians behind every pixel's last contributor in this wave are skipped wholesale.
        const int32_t t_first = max(0, batch_end - wave_bin_final);
        for (int32_t j = (t_first & ~63); j < batch_size; j += 64) {
          // cull 64 staged Gaussians at once against this wave's pixel rectangle (raster3d.hpp)
          const int32_t tl = j + (int32_t)lane;
          bool hit         = false;
          if (tl >= t_first && tl < batch_size) {
              const float4 cu = s_cull[tl];
              hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
          }
          uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) { // scalar loop over the survivors, back to front
            const int32_t t = j + (int32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            // Branch-free body: invalid lanes run the same arithmetic with alpha = vis = 0, which makes
            // every contribution exactly 0 and leaves T / buffer unchanged (1/(1-0) == 1 exactly).
            const float4 ga = s_ga[t];
            const float2 gb = s_gb[t];
            const float dx = ga.x - px;
            const float dy = ga.y - py;
            const float q  = staged_q(ga, gb, dx, dy);
            const float ov_r = staged_alpha_raw(ga, q); // opac * exp(-sigma), unclamped
            // lanes outside the image have bin_final = -1 and can never be valid
            const bool valid = (batch_end - t <= bin_final) && !(q < 0.0f) && !(fminf(kMaxAlpha, ov_r) < kAlphaThreshold);
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue; // wave-uniform

            const float ov    = valid ? ov_r : 0.0f;
            const float alpha = fminf(kMaxAlpha, ov);
            float loc[(K + 3) / 4 * 4];
            {
                const float ra  = __builtin_amdgcn_rcpf(fmaxf(kMinOneMinusAlpha, 1.0f - alpha));
                T              *= ra;
                const float fac = alpha * T;
                float v_alpha   = 0.0f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const float c = s_col[t * CH + k];
                    loc[k]        = fac * v_c[k];
                    v_alpha      += (c * T - buffer[k] * ra) * v_c[k];
                    buffer[k]    += c * fac;
                }
                v_alpha += T_final * ra * va_minus_bg;
                const bool unclamped = ov <= kMaxAlpha; // alpha-clamp branch: geometry/opacity grads vanish
                const float v_sigma  = unclamped ? -ov * v_alpha : 0.0f;
                const float wdx = v_sigma * dx, wdy = v_sigma * dy;
                loc[CH + 0]     = wdx * dx; // moments; see flush
                loc[CH + 1]     = wdx * dy;
                loc[CH + 2]     = wdy * dy;
                loc[CH + 3]     = wdx;
                loc[CH + 4]     = wdy;
                loc[CH + 5]     = v_sigma;
                if constexpr (ABS) {
                    // |conic . (wdx, wdy)| with conic = (2A, B; B, 2C) / log2(e); the 1/log2(e) is applied at flush
                    loc[CH + 6] = fabsf(2.0f * ga.w * wdx + gb.x * wdy);
                    loc[CH + 7] = fabsf(gb.x * wdx + 2.0f * gb.y * wdy);
                }
#pragma unroll
                for (int k = K; k < (K + 3) / 4 * 4; ++k) loc[k] = 0.0f;
            }
            // reduce four values per step; row r of group j ends up with total of value 4j + r
            float mine = 0.0f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float r = wave_sum4_scatter(loc[4 * j], loc[4 * j + 1], loc[4 * j + 2], loc[4 * j + 3]);
                if ((int)(lane & 15u) == j) mine = r;
            }
            int vidx    = 4 * (int)(lane & 15u) + (int)(lane >> 4);
            bool writer = (int)(lane & 15u) < KQ && vidx < K;
            if constexpr (Cfg::TAIL1) { // value K-1 alone: total lands in the last row; lane 63 adds it
                const float r = wave_sum_last_row(loc[K - 1]);
                if (lane == 63u) {
                    mine   = r;
                    vidx   = K - 1;
                    writer = true;
                }
            }
            if (writer) atomicAdd(&s_acc[t * KP + vidx], mine); // ds_add_f32
            if (lane == 0) s_touch[t] = 1;
          }
        }
        __syncthreads();

        // flush, transposed: element e of the batch's [batch_size][NCOL] gradient block -> (Gaussian s, column c);
        // consecutive lanes -> consecutive floats of one AoS row. Moments become gradients here (linear maps):
        //   v_xy = Q (S_x, S_y), v_conic = (S_xx/2, S_xy, S_yy/2), v_opacity = sum vis*v_alpha = -S_w / opacity
        // (a touched Gaussian has opacity >= 1/255), with Q = (2A, B; B, 2C) / log2(e) and opacity = exp2(lo) from
        // the staged form.
        constexpr float kInvLog2e = 1.0f / kLog2e;
        constexpr int GEO  = 6 + (ABS ? 2 : 0);
        constexpr int NCOL = GEO + CH;
        for (int e = (int)tid; e < batch_size * NCOL; e += (int)blockDim.x) {
            const int s = e / NCOL, c = e - s * NCOL;
            if (!s_touch[s]) continue;
            const float *row = s_acc + s * KP;
            float val;
            int col = c;
            if (c < 2) {
                const float4 ga = s_ga[s];
                const float2 gb = s_gb[s];
                const float sx = row[CH + 3], sy = row[CH + 4];
                val = kInvLog2e * ((c == 0) ? (2.0f * ga.w * sx + gb.x * sy) : (gb.x * sx + 2.0f * gb.y * sy));
            } else if (c < 5) {
                const float m = row[CH + c - 2];
                val           = (c == 3) ? m : 0.5f * m;
            } else if (c == 5) {
                val = -row[CH + 5] * __builtin_amdgcn_exp2f(-s_ga[s].z);
            } else if (c < GEO) {
                val = kInvLog2e * row[CH + c]; // |v_xy| sums (ABS): accumulator slots CH+6, CH+7 <- columns 6, 7
            } else {
                const int k = c - GEO;
                if (k >= (int)a.nch) continue;
                val = row[k];
                col = GEO + (int)a.ch_off + k;
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
        // The next iteration's staging writes s_ga/s_gb/s_col/s_id (read above only by the owner
        // thread or before the barrier); its barrier orders the zeroed rows before new ds_adds.
    }
}

template <int CH, bool ABS>
static int launch_bwd(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0 || a.n_isects == 0) return GSX_OK;
    const uint32_t grid  = ((n_blocks + 7u) / 8u) * 8u;
    const uint32_t block = a.tile_size <= 8 ? 64u : 256u;
    using Cfg = BwdCfg<CH, ABS>;
    const size_t smem = Cfg::smem;
    raster3d_bwd_kernel<CH, ABS><<<dim3(grid), dim3(block), smem, stream>>>(a);
    return check_launch("raster3d_bwd");
}

template <bool ABS>
static int bwd_dispatch(Raster3DArgs a, hipStream_t stream)
{
    uint32_t off = 0;
    bool first   = true;
    do {
        const uint32_t rem = a.cdim - off;
        const uint32_t n   = rem > 32 ? 32 : rem;
        a.ch_off           = off;
        a.nch              = n;
        a.first_chunk      = first ? 1u : 0u;
        int rc;
        if (n <= 1) rc = launch_bwd<1, ABS>(a, stream);
        else if (n <= 2) rc = launch_bwd<2, ABS>(a, stream);
        else if (n <= 3) rc = launch_bwd<3, ABS>(a, stream);
        else if (n <= 4) rc = launch_bwd<4, ABS>(a, stream);
        else if (n <= 8) rc = launch_bwd<8, ABS>(a, stream);
        else if (n <= 16) rc = launch_bwd<16, ABS>(a, stream);
        else rc = launch_bwd<32, ABS>(a, stream);
        if (rc != GSX_OK) return rc;
        off += n;
        first = false;
    } while (off < a.cdim);
    return GSX_OK;
}

} // namespace gsx

extern "C" int gsx_raster3d_bwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_bwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_bwd: channels must be >= 1");
    GSX_REQUIRE(v_rows, "gsx_raster3d_bwd: null gradient output");
    GSX_REQUIRE(row_stride >= 6u + (has_abs ? 2u : 0u) + cdim,
                "gsx_raster3d_bwd: row_stride %u too small for 6%s + %u channels", row_stride, has_abs ? " + 2" : "", cdim);
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids && render_alphas && last_ids
                                  && v_render_colors && isect_offsets),
                "gsx_raster3d_bwd: null input");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets; a.flatten_ids = flatten_ids;
    a.render_alphas = const_cast<float *>(render_alphas); a.last_ids = const_cast<int32_t *>(last_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas;
    a.v_rows = v_rows; a.row_stride = row_stride;
    return has_abs ? bwd_dispatch<true>(a, (hipStream_t)stream) : bwd_dispatch<false>(a, (hipStream_t)stream);
}

// Sparse pixel set (gsplat::rasterize_to_pixels_sparse_bwd, reference RasterizeToPixelsSparseBwd.cu): render_alphas,
// last_ids and the cotangents are rows [P, ...]; gradient rows v_rows as in gsx_raster3d_bwd.
extern "C" int gsx_raster3d_sparse_bwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
    const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active,
    uint32_t words_per_tile, const float *render_alphas, const int32_t *last_ids, const float *v_render_colors,
    const float *v_render_alphas, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_sparse_bwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_sparse_bwd: channels must be >= 1");
    GSX_REQUIRE(v_rows, "gsx_raster3d_sparse_bwd: null gradient output");
    GSX_REQUIRE(row_stride >= 6u + (has_abs ? 2u : 0u) + cdim, "gsx_raster3d_sparse_bwd: row_stride %u too small", row_stride);
    if (n_active == 0 || n_isects == 0) return GSX_OK;
    GSX_REQUIRE(words_per_tile * 64u >= tile_size * tile_size, "gsx_raster3d_sparse_bwd: pixel mask too narrow");
    GSX_REQUIRE(active_tiles && tile_offsets && tile_pixel_mask && tile_pixel_cumsum && pixel_map && means2d && conics
                && colors && opacities && flatten_ids && render_alphas && last_ids && v_render_colors,
                "gsx_raster3d_sparse_bwd: null input");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = tile_offsets; a.flatten_ids = flatten_ids;
    a.render_alphas = const_cast<float *>(render_alphas); a.last_ids = const_cast<int32_t *>(last_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas;
    a.v_rows = v_rows; a.row_stride = row_stride;
    a.sp_active_tiles = active_tiles; a.sp_pixel_mask = tile_pixel_mask; a.sp_pixel_cumsum = tile_pixel_cumsum;
    a.sp_pixel_map = pixel_map; a.n_active = n_active; a.sp_words = words_per_tile;
    return has_abs ? bwd_dispatch<true>(a, (hipStream_t)stream) : bwd_dispatch<false>(a, (hipStream_t)stream);
}
