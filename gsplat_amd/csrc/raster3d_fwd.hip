// 3DGS front-to-back alpha compositing, forward (gfx950).
// C-ABI entry: gsx_raster3d_fwd  (replaces torch op gsplat::rasterize_to_pixels_3dgs,
// reference host fn gsplat/cuda/csrc/Rasterization.cpp:275-365, kernel
// RasterizeToPixels3DGSSerialBatchFwd.cu:41-297).
#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

constexpr int kBatch = 256;
// Staged layout: (x, y, log2 opac, A) | (B, C, colour 0, colour 1) | colours 2.. - b128 + b128 + b32 per surviving Gaussian
// at 3 channels instead of b128 + b64 + 3 x b32 (r05 A/B on c3: 0.304 -> 0.280 ms per launch, profiles/r05_ab.md).

template <int CH>
__global__ void __launch_bounds__(256) raster3d_fwd_kernel(const Raster3DArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ga   = reinterpret_cast<float4 *>(smem_raw);                 // x, y, log2(opac), A   (stage_gaussian)
    float4 *s_cull = s_ga + kBatch;                                        // x, y, half extents of alpha >= 1/255
    constexpr int CX = CH > 2 ? CH - 2 : 0;
    float4 *s_gbc  = s_cull + kBatch;                                      // B, C, colour 0, colour 1
    float *s_col   = reinterpret_cast<float *>(s_gbc + kBatch);            // [kBatch][CX]: colours 2..

    TileCtx tc;
    if (!tile_context(a, blockIdx.x, tc)) return; // uniform for the whole workgroup
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image_id = tc.image_id, tile_id = tc.tile_id;
    const uint32_t tid      = threadIdx.x;

    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly); // output row, -1 = this lane renders nothing
    const bool inside  = prow >= 0;
    const float px     = (float)(tc.tile_x * a.tile_size + lx) + 0.5f;
    const float py     = (float)(tc.tile_y * a.tile_size + ly) + 0.5f;
    const size_t pix   = inside ? (size_t)prow : 0;

    const float *bg = a.backgrounds ? a.backgrounds + (size_t)image_id * a.cdim + a.ch_off : nullptr;

    // masked-off tile: background colour, zero alpha, last_id 0 (reference Fwd.cu:141-159)
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) {
        if (inside) {
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < (int)a.nch) a.render_colors[pix * a.cdim + a.ch_off + k] = bg ? bg[k] : 0.0f;
            if (a.first_chunk) {
                a.render_alphas[pix] = 0.0f;
                a.last_ids[pix]      = 0;
            }
        }
        return;
    }

    const int32_t range_start = tc.range_start, range_end = tc.range_end;
    const int32_t n_batches   = (range_end - range_start + kBatch - 1) / kBatch;

    float T          = 1.0f;
    uint32_t cur_idx = 0;
    float acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = 0.0f;
    bool done = !inside;
    const uint32_t lane = tid & 63u;
    const WaveRect rect = wave_pixel_rect(inside, px, py);

    for (int32_t b = 0; b < n_batches; ++b) {
        // block-wide early out: every pixel of the tile finished. Also fences LDS reuse.
        if (__syncthreads_count(done) == (int)blockDim.x) break;

        const int32_t batch_start = range_start + kBatch * b;
        for (int s = (int)tid; s < kBatch; s += (int)blockDim.x) {
            const int32_t idx = batch_start + s;
            if (idx < range_end) {
                const int32_t g  = a.flatten_ids[idx];
                const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                float4 ga;
                float2 gb;
                stage_gaussian(xy.x, xy.y, opac, ca, cb, cc, ga, gb);
                s_ga[s]         = ga;
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]       = make_float4(xy.x, xy.y, he.x, he.y);
                const float *c  = a.colors + (size_t)g * a.cdim + a.ch_off;
                float cv[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) cv[k] = (k < (int)a.nch) ? c[k] : 0.0f;
                s_gbc[s] = make_float4(gb.x, gb.y, cv[0], CH > 1 ? cv[CH > 1 ? 1 : 0] : 0.0f);
#pragma unroll
                for (int k = 2; k < CH; ++k) s_col[s * CX + k - 2] = cv[k];
            }
        }
        __syncthreads();

        const int32_t batch_size = min(kBatch, range_end - batch_start);
        // Each wave tests 64 staged Gaussians at a time (one per lane) against the rectangle of ITS 8x8 pixel
        // centres (raster3d.hpp) and walks only the survivors, front to back, with a scalar loop over the ballot.
        // A culled (wave, Gaussian) pair has no lane that would pass the alpha test, so results are unchanged.
        for (int32_t j = 0; j < batch_size; j += 64) {
            // wave-level early termination: a finished quadrant stops evaluating (it still takes part in
            // staging and in the barriers above).
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            const int32_t tl = j + (int32_t)lane;
            bool hit         = false;
            if (tl < batch_size) {
                const float4 cu = s_cull[tl];
                hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
            }
            uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
            while (todo) {
                const int32_t t = j + (int32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const float4 ga = s_ga[t];
                const float4 gbc = s_gbc[t];
                const float2 gb  = make_float2(gbc.x, gbc.y);
                const float dx    = ga.x - px;
                const float dy    = ga.y - py;
                const float q     = staged_q(ga, gb, dx, dy);
                const float alpha = fminf(kMaxAlpha, staged_alpha_raw(ga, q));
                if (done || q < 0.0f || alpha < kAlphaThreshold) continue;
                const float next_T = T * (1.0f - alpha);
                if (next_T <= kTransmittanceThresh) { // saturated: this Gaussian is excluded
                    done = true;
                    continue;
                }
                const float w = alpha * T;
                acc[0] += gbc.z * w;
                if constexpr (CH > 1) acc[1] += gbc.w * w;
#pragma unroll
                for (int k = 2; k < CH; ++k) acc[k] += s_col[t * CX + k - 2] * w;
                cur_idx = (uint32_t)(batch_start + t);
                T       = next_T;
            }
        }
    }

    if (inside) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch)
                a.render_colors[pix * a.cdim + a.ch_off + k] = bg ? (acc[k] + T * bg[k]) : acc[k];
        if (a.first_chunk) {
            a.render_alphas[pix] = 1.0f - T;
            a.last_ids[pix]      = (int32_t)cur_idx;
        }
    }
}

template <int CH>
static int launch_fwd(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid   = ((n_blocks + 7u) / 8u) * 8u; // xcd_remap needs a multiple of 8
    const uint32_t block  = a.tile_size <= 8 ? 64u : 256u;
    const size_t smem     = kBatch * (3 * sizeof(float4) + sizeof(float) * (CH > 2 ? CH - 2 : 0));
    hipLaunchKernelGGL(raster3d_fwd_kernel<CH>, dim3(grid), dim3(block), smem, stream, a);
    return check_launch("raster3d_fwd");
}

int raster3d_fwd_dispatch(Raster3DArgs a, hipStream_t stream)
{
    // walk the channel dimension in chunks of <= 32 (register budget); alphas, last_ids come
    // from the first chunk (same rule as the reference orchestrator, Rendering.cpp:1353-1435).
    uint32_t off = 0;
    bool first   = true;
    do {
        const uint32_t rem = a.cdim - off;
        const uint32_t n   = rem > 32 ? 32 : rem;
        a.ch_off           = off;
        a.nch              = n;
        a.first_chunk      = first ? 1u : 0u;
        int rc;
        if (n <= 1) rc = launch_fwd<1>(a, stream);
        else if (n <= 2) rc = launch_fwd<2>(a, stream);
        else if (n <= 3) rc = launch_fwd<3>(a, stream);
        else if (n <= 4) rc = launch_fwd<4>(a, stream);
        else if (n <= 8) rc = launch_fwd<8>(a, stream);
        else if (n <= 16) rc = launch_fwd<16>(a, stream);
        else rc = launch_fwd<32>(a, stream);
        if (rc != GSX_OK) return rc;
        off += n;
        first = false;
    } while (off < a.cdim);
    return GSX_OK;
}

} // namespace gsx

extern "C" int gsx_raster3d_fwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, float *render_colors, float *render_alphas, int32_t *last_ids, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_fwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_fwd: channels must be >= 1");
    GSX_REQUIRE(render_colors && render_alphas && last_ids, "gsx_raster3d_fwd: null output");
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids),
                "gsx_raster3d_fwd: null input");
    GSX_REQUIRE(isect_offsets != nullptr || n_images * tile_w * tile_h == 0, "gsx_raster3d_fwd: null isect_offsets");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets; a.flatten_ids = flatten_ids;
    a.render_colors = render_colors; a.render_alphas = render_alphas; a.last_ids = last_ids;
    return raster3d_fwd_dispatch(a, (hipStream_t)stream);
}

// Sparse pixel set (gsplat::rasterize_to_pixels_sparse, reference RasterizeToPixelsSparseFwd.cu): same kernel, one
// workgroup per ACTIVE tile; outputs are rows [P, ...] in the caller's pixel order (raster3d.hpp: TileCtx / pixel_row).
extern "C" int gsx_raster3d_sparse_fwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
    const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active,
    uint32_t words_per_tile, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *render_colors, float *render_alphas, int32_t *last_ids,
    void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_sparse_fwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_sparse_fwd: channels must be >= 1");
    if (n_active == 0) return GSX_OK;
    GSX_REQUIRE(words_per_tile * 64u >= tile_size * tile_size, "gsx_raster3d_sparse_fwd: pixel mask too narrow");
    GSX_REQUIRE(active_tiles && tile_offsets && tile_pixel_mask && tile_pixel_cumsum && pixel_map && render_colors
                && render_alphas && last_ids, "gsx_raster3d_sparse_fwd: null layout / output");
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids), "gsx_raster3d_sparse_fwd: null input");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = tile_offsets; a.flatten_ids = flatten_ids;
    a.render_colors = render_colors; a.render_alphas = render_alphas; a.last_ids = last_ids;
    a.sp_active_tiles = active_tiles; a.sp_pixel_mask = tile_pixel_mask; a.sp_pixel_cumsum = tile_pixel_cumsum;
    a.sp_pixel_map = pixel_map; a.n_active = n_active; a.sp_words = words_per_tile;
    return raster3d_fwd_dispatch(a, (hipStream_t)stream);
}
