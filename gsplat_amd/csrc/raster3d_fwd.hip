// 3DGS front-to-back alpha compositing, forward (gfx950).
// C-ABI entry: gsx_raster3d_fwd  (replaces torch op gsplat::rasterize_to_pixels_3dgs,
// reference host fn gsplat/cuda/csrc/Rasterization.cpp:275-365, kernel
// RasterizeToPixels3DGSSerialBatchFwd.cu:41-297).
#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

#ifndef GSX_FWD_AT // 1: blended alpha through one select (alpha or 0), then plain arithmetic; 0: selects on w and T (A/B)
#define GSX_FWD_AT 1
#endif
constexpr int kBatch = 256;

// Staged layout: one 48-byte row per Gaussian (raster3d.hpp StagedRow: the tile-centre polynomial of the exponent + up to
// four colours) read with b128 + b128 + b64 from one address register; colours 4.. in a separate table.
// r05 A/B on c3 (profiles/r05_ab.md): float4 + float2 + 3 floats 0.304 ms -> packed float4s 0.280 -> e-form see there.

// PRE = true: the pre-pass of the SEGMENTED BACKWARD (raster3d_seg.hip) - one workgroup per segment item walks its slice
// front to back from transmittance 1 with the contributors the forward pass fixed (alpha test, list index <= last_ids; no
// early-termination rule) and stores, per pixel, the slice's transmittance and S = sum_i alpha_i T_i (c_i . v_colour).
template <int CH, bool PRE = false>
__global__ void __launch_bounds__(256) raster3d_fwd_kernel(const Raster3DArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int CX = CH > 4 ? CH - 4 : 0;
    StagedRow *s_st = reinterpret_cast<StagedRow *>(smem_raw);              // [kBatch]
    float4 *s_cull  = reinterpret_cast<float4 *>(s_st + kBatch);            // mean - tile centre, half extents of alpha >= 1/255
    float *s_col    = reinterpret_cast<float *>(s_cull + kBatch);           // [kBatch][CX]: colours 4..

    TileCtx tc;
    uint32_t seg_item;
    if (!tile_context_seg(a, blockIdx.x, tc, seg_item)) return; // uniform for the whole workgroup
    // what this workgroup is: a whole tile (also the short tiles of a compositing-pass launch), or a segment in the
    // transmittance / compositing pass
    const uint32_t seg_mode = seg_item == 0xFFFFFFFFu ? 0u : a.seg_mode;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image_id = tc.image_id, tile_id = tc.tile_id;
    const uint32_t tid      = threadIdx.x;

    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly); // output row, -1 = this lane renders nothing
    const bool inside  = prow >= 0;
    // tile centre in pixel coordinates, and this lane's pixel centre relative to it (multiples of 0.5: exact)
    const float half = 0.5f * (float)a.tile_size;
    const float cx   = (float)(tc.tile_x * a.tile_size) + half, cy = (float)(tc.tile_y * a.tile_size) + half;
    const float u    = (float)lx + 0.5f - half, v = (float)ly + 0.5f - half;
    const size_t pix = inside ? (size_t)prow : 0;

    const float *bg = a.backgrounds ? a.backgrounds + (size_t)image_id * a.cdim + a.ch_off : nullptr;

    // masked-off tile: background colour, zero alpha, last_id 0 (reference Fwd.cu:141-159)
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) {
        if constexpr (PRE) {
            a.seg_T[(size_t)seg_item * 256 + tid]   = 1.0f;
            a.seg_out[(size_t)seg_item * 256 + tid] = 0.0f;
            return;
        }
        if (seg_mode == 1u) { // a masked tile's segments contribute nothing: transmittance 1 everywhere
            a.seg_T[(size_t)seg_item * 256 + tid] = 1.0f;
            return;
        }
        if (seg_mode == 2u) {
            for (uint32_t k = 0; k < a.nch; ++k) a.seg_out[((size_t)seg_item * (a.nch + 1) + k) * 256 + tid] = 0.0f;
            a.seg_out[((size_t)seg_item * (a.nch + 1) + a.nch) * 256 + tid] = 1.0f;
            a.seg_last[(size_t)seg_item * 256 + tid] = -1;
            return;
        }
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

    const int32_t range_start = tc.range_start;
    int32_t range_end = tc.range_end;
    int32_t last_id   = -1; // PRE: this pixel's last contributor (list index); entries behind it belong to other pixels
    if constexpr (PRE) {
        last_id = inside ? a.last_ids[pix] : -1;
        __shared__ int32_t s_m[4];
        const int32_t wm = wave_max_i32(last_id);
        if ((tid & 63u) == 0) s_m[tid >> 6] = wm;
        __syncthreads();
        range_end = min(range_end, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])) + 1); // nothing behind the tile's last one
    }
    const int32_t n_batches = range_end > range_start ? (range_end - range_start + kBatch - 1) / kBatch : 0;

    float T          = 1.0f;
    uint32_t cur_idx = seg_mode ? 0xFFFFFFFFu : 0u; // a segment reports "no contributor" as -1
    float acc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[k] = 0.0f;
    float thr = inside ? kAlphaThreshold : INFINITY; // alpha threshold of this pixel; +inf = done (or not rendered)
    if (seg_mode == 2u) { // compositing pass of a segment: the transmittance in front of it (raster3d_seg.hip: seg_prefix)
        T = a.seg_T[(size_t)seg_item * 256 + tid];
        if (!(T > kTransmittanceThresh)) thr = INFINITY; // the pixel stopped in an earlier segment
    }
    const uint32_t lane = tid & 63u;

    for (int32_t b = 0; b < n_batches; ++b) {
        // block-wide early out: every pixel of the tile finished. Also fences LDS reuse.
        if (__syncthreads_count(!(thr < INFINITY)) == (int)blockDim.x) break;

        const int32_t batch_start = __builtin_amdgcn_readfirstlane(range_start + kBatch * b); // wave-uniform: keep it scalar
        for (int s = (int)tid; s < kBatch; s += (int)blockDim.x) {
            const int32_t idx = batch_start + s;
            if (idx < range_end) {
                const int32_t g  = a.flatten_ids[idx];
                float2 xy;
                float opac, ca, cb, cc;
                v4f row2 = v4f{0.f, 0.f, 0.f, 0.f}, row1 = row2;
                if (a.splat_rows) { // one 48-byte row (raster3d.hpp): three 16-byte loads from one address
                    const v4f *rw = reinterpret_cast<const v4f *>(a.splat_rows) + 3 * (size_t)g;
                    const v4f row0 = rw[0];
                    row1 = rw[1]; row2 = rw[2];
                    xy = make_float2(row0.x, row0.y); ca = row0.z; cb = row0.w; cc = row1.x; opac = row1.y;
                } else {
                    xy   = reinterpret_cast<const float2 *>(a.means2d)[g];
                    opac = a.opacities[g];
                    ca = a.conics[3 * (size_t)g]; cb = a.conics[3 * (size_t)g + 1]; cc = a.conics[3 * (size_t)g + 2];
                }
                const float ax = xy.x - cx, ay = xy.y - cy;
                v4f p0;
                float nA, nB, nC;
                stage_gaussian_f(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]       = make_float4(ax, ay, he.x, he.y);
                float cv[CH > 4 ? CH : 4];
                if (a.splat_rows) { // cdim == 3 (checked by the entry point): the colours sit in the row
#pragma unroll
                    for (int k = 0; k < (CH > 4 ? CH : 4); ++k) cv[k] = 0.0f;
                    cv[0] = row1.z;
                    if (CH > 1) cv[1] = row1.w;
                    if (CH > 2) cv[2] = row2.x;
                } else {
                    const float *c = a.colors + (size_t)g * a.cdim + a.ch_off;
#pragma unroll
                    for (int k = 0; k < (CH > 4 ? CH : 4); ++k) cv[k] = (k < CH && k < (int)a.nch) ? c[k] : 0.0f;
                }
                s_st[s].p0 = p0;
                s_st[s].p1 = v4f{nA, nB, nC, cv[2]};
                s_st[s].p2 = v4f{cv[0], cv[1], cv[3], 0.0f};
#pragma unroll
                for (int k = 4; k < CH; ++k) s_col[s * CX + k - 4] = cv[k];
            }
        }
        __syncthreads();

        const int32_t batch_size = min(kBatch, range_end - batch_start);
        // The rectangle of the wave's pixels that are still open, in tile-centre coordinates like s_cull. It shrinks as
        // pixels saturate; refreshed once per batch (~50 instructions per wave): 0.175 -> 0.169 ms on c3 (r05_ab #36).
        const WaveRect rect = wave_pixel_rect(thr < INFINITY, u, v);
        // Each wave tests 64 staged Gaussians at a time (one per lane) against the rectangle of ITS 8x8 pixel
        // centres (raster3d.hpp) and walks only the survivors, front to back, with a scalar loop over the ballot.
        // A culled (wave, Gaussian) pair has no lane that would pass the alpha test, so results are unchanged.
        for (int32_t j = 0; j < batch_size; j += 64) {
            // wave-level early termination: a finished quadrant stops evaluating (it still takes part in
            // staging and in the barriers above).
            if (__builtin_amdgcn_ballot_w64(thr < INFINITY) == 0ull) break;
            const int32_t tl = j + (int32_t)lane;
            bool hit         = false;
            if (tl < batch_size) {
                const float4 cu = s_cull[tl];
                hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
                if (hit) hit = rect_reaches_level(s_st[tl].p0, s_st[tl].p1, cu.x, cu.y, rect); // exact second stage
            }
            uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
            while (todo) {
                const int32_t bit = (int32_t)__builtin_ctzll(todo);
                const int32_t t   = j + bit;
                asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit)); // todo &= todo - 1 in one scalar instruction

                const v4f p0 = s_st[t].p0;
                const v4f p1 = s_st[t].p1;
                const v4f p2 = s_st[t].p2; // one address register for the three reads (b128 + b128 + b64 / b128)
                const float e     = staged_f(p0, p1.x, p1.y, p1.z, u, v);
                const float alpha = fminf(kMaxAlpha, __builtin_amdgcn_exp2f(e));
                // Branch-free body. The scalar unit, not the vector ALU, was the busiest pipe of this kernel when "pixel is
                // done" lived in an EXEC-style mask (25 scalar instructions per surviving Gaussian, r05 PMC): the state now
                // lives in vector registers - `thr` is the pixel's alpha threshold and becomes +inf once it is done - and the
                // three decisions (passes / saturates / is blended) are lane masks combined on the scalar side.
                bool ok = !(e > p0.w) && !(alpha < thr); // e > lo <=> sigma < 0
                if constexpr (PRE) ok = ok && (batch_start + t <= last_id);
                if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue; // wave-uniform
                const float next_T = fmaf(-T, alpha, T);
                const bool low     = PRE ? false : next_T <= kTransmittanceThresh; // saturated: this Gaussian is excluded
                const bool sat = ok && low, take = ok && !sat;
#if GSX_FWD_AT
                const float at = take ? alpha : 0.0f;
                const float w  = at * T;
#else
                const float w  = take ? alpha * T : 0.0f;
#endif
                acc[0] += p2.x * w;
                if constexpr (CH > 1) acc[1] += p2.y * w;
                if constexpr (CH > 2) acc[2] += p1.w * w;
                if constexpr (CH > 3) acc[3] += p2.z * w;
#pragma unroll
                for (int k = 4; k < CH; ++k) acc[k] += s_col[t * CX + k - 4] * w;
                cur_idx = take ? (uint32_t)(batch_start + t) : cur_idx;
#if GSX_FWD_AT
                T       = fmaf(-T, at, T); // == next_T where the Gaussian is blended, T exactly where not: a full-rate fma for a select
#else
                T       = take ? next_T : T;
#endif
                thr     = sat ? INFINITY : thr;
            }
        }
    }

    if constexpr (PRE) {
        float S = 0.0f;
        if (inside) {
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < (int)a.nch) S = fmaf(acc[k], a.v_render_colors[pix * a.cdim + a.ch_off + k], S);
        }
        a.seg_T[(size_t)seg_item * 256 + tid]   = T;
        a.seg_out[(size_t)seg_item * 256 + tid] = S;
        return;
    }
    if (seg_mode == 1u) { // transmittance of this slice from 1; 0 = the early-termination rule fired inside it
        a.seg_T[(size_t)seg_item * 256 + tid] = (inside && !(thr < INFINITY)) ? 0.0f : T;
        return;
    }
    if (seg_mode == 2u) { // partial result of this segment, pixel-major planes (raster3d_seg.hip combines them)
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch) a.seg_out[((size_t)seg_item * (a.nch + 1) + k) * 256 + tid] = acc[k];
        a.seg_out[((size_t)seg_item * (a.nch + 1) + a.nch) * 256 + tid] = T;
        a.seg_last[(size_t)seg_item * 256 + tid] = (int32_t)cur_idx;
        return;
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
    const uint32_t n_blocks = a.seg_mode ? a.seg_grid : (a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images);
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid   = ((n_blocks + 7u) / 8u) * 8u; // xcd_remap needs a multiple of 8
    const uint32_t block  = a.tile_size <= 8 ? 64u : 256u;
    const size_t smem     = kBatch * (sizeof(StagedRow) + sizeof(float4) + sizeof(float) * (CH > 4 ? CH - 4 : 0));
    hipLaunchKernelGGL(raster3d_fwd_kernel<CH>, dim3(grid), dim3(block), smem, stream, a);
    return check_launch("raster3d_fwd");
}

// pre-pass of the segmented backward (<= 4 channels: the backward's variant T): a.seg_mode = 1 item list, grid a.seg_grid
int raster3d_bwd_prepass_launch(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t grid = ((a.seg_grid + 7u) / 8u) * 8u;
    if (grid == 0) return GSX_OK;
    const size_t smem = kBatch * (sizeof(StagedRow) + sizeof(float4));
    if (a.nch <= 1) hipLaunchKernelGGL((raster3d_fwd_kernel<1, true>), dim3(grid), dim3(256), smem, stream, a);
    else if (a.nch <= 2) hipLaunchKernelGGL((raster3d_fwd_kernel<2, true>), dim3(grid), dim3(256), smem, stream, a);
    else if (a.nch <= 3) hipLaunchKernelGGL((raster3d_fwd_kernel<3, true>), dim3(grid), dim3(256), smem, stream, a);
    else hipLaunchKernelGGL((raster3d_fwd_kernel<4, true>), dim3(grid), dim3(256), smem, stream, a);
    return check_launch("raster3d_bwd_prepass");
}

// one launch for the channel chunk described by a.ch_off / a.nch / a.first_chunk
int raster3d_fwd_launch_chunk(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n = a.nch;
    if (raster3d_fwd_w_applies(a)) return raster3d_fwd_w_launch(a, stream); // one wave per tile (raster3d_fwd_w.hip)
    if (raster3d_fwd_m_applies(a)) return raster3d_fwd_m_launch(a, stream); // 5 .. 32 channels: the render as a matrix product
    if (n <= 1) return launch_fwd<1>(a, stream);
    if (n <= 2) return launch_fwd<2>(a, stream);
    if (n <= 3) return launch_fwd<3>(a, stream);
    if (n <= 4) return launch_fwd<4>(a, stream);
    if (n <= 8) return launch_fwd<8>(a, stream);
    if (n <= 16) return launch_fwd<16>(a, stream);
    return launch_fwd<32>(a, stream);
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
        const int rc = raster3d_fwd_launch_chunk(a, stream);
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

// gsx_raster3d_fwd with the Gaussians' 48-byte array-of-structures rows beside the four arrays (cdim == 3; raster3d.hpp)
extern "C" int gsx_raster3d_fwd_rows(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *splat_rows,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, float *render_colors, float *render_alphas, int32_t *last_ids, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_fwd_rows: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(!splat_rows || cdim == 3, "gsx_raster3d_fwd_rows: the rows hold three colours; cdim is %u", cdim);
    GSX_REQUIRE(!splat_rows || (reinterpret_cast<uintptr_t>(splat_rows) & 15u) == 0, "gsx_raster3d_fwd_rows: rows must be 16-byte aligned");
    GSX_REQUIRE(render_colors && render_alphas && last_ids, "gsx_raster3d_fwd_rows: null output");
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids), "gsx_raster3d_fwd_rows: null input");
    GSX_REQUIRE(isect_offsets != nullptr || n_images * tile_w * tile_h == 0, "gsx_raster3d_fwd_rows: null isect_offsets");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities; a.splat_rows = splat_rows;
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
