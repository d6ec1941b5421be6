// Shared pieces of the 3DGS alpha-compositing kernels (forward + backward).
//
// Work decomposition (MI355X-first, not the reference's 32-lane warp layout):
//   * one 256-thread workgroup per 16x16 tile = 4 wave64s;
//   * wave w owns the 8x8 pixel QUADRANT (w&1, w>>1) of the tile, lane l the pixel
//     (l&7, l>>3) inside it. An 8x8 footprint per wave makes wave-level culling and
//     wave-level early termination far more effective than 16x4 strips;
//   * the tile's depth-sorted Gaussian list is walked in batches of 256 staged in LDS
//     (coalesced gather through flatten_ids, then conflict-free broadcast reads).
//
// Semantics restated from the reference (behaviour, not code):
//   gsplat/cuda/csrc/RasterizeToPixels3DGSDevice.cuh:44-173 (per-sample math)
//   gsplat/cuda/csrc/RasterizeToPixels3DGSSerialBatchFwd.cu:41-297
//   gsplat/cuda/csrc/RasterizeToPixels3DGSSerialBatchBwd.cu:41-320
#pragma once
#include "common.hpp"

namespace gsx {

struct Raster3DArgs {
    // geometry of the launch
    uint32_t n_images;    // I
    uint32_t n_isects;    // M
    uint32_t width;
    uint32_t height;
    uint32_t tile_size;   // <= 16
    uint32_t tile_w;
    uint32_t tile_h;
    // channel chunking: this launch handles channels [ch_off, ch_off + nch) of cdim
    uint32_t cdim;
    uint32_t ch_off;
    uint32_t nch;
    uint32_t first_chunk; // 1 => this launch also owns alpha / last_ids / geometry-from-alpha terms
    // forward inputs (rows indexed by flatten_ids: [I*N] dense or [nnz] packed)
    const float *means2d;     // [R, 2]
    const float *conics;      // [R, 3]
    const float *colors;      // [R, cdim]
    const float *opacities;   // [R]
    const float *backgrounds; // [I, cdim] or null
    const uint8_t *masks;     // [I, tile_h, tile_w] or null (torch bool)
    const int32_t *isect_offsets; // [I, tile_h, tile_w]
    const int32_t *flatten_ids;   // [M]
    // optional (cdim == 3 only): ONE 48-byte array-of-structures row per Gaussian row,
    //   (x, y, conic a, conic b | conic c, opacity, colour 0, colour 1 | colour 2, -, -, -),
    // the same values as means2d / conics / opacities / colors. A staging thread then needs three 16-byte loads from ONE row
    // instead of four gathers from four arrays (c3 forward 0.203 -> 0.181 ms in the kernel harness, profiles/r10_ab.md): the
    // SH forward of rasterization() writes the rows while it has the row's colours in registers (gsx_sh_fwd_rows)
    const float *splat_rows;
    // forward outputs
    float *render_colors; // [I, H, W, cdim]
    float *render_alphas; // [I, H, W, 1]
    int32_t *last_ids;    // [I, H, W]
    // backward inputs
    const float *v_render_colors; // [I, H, W, cdim]; vrc_strided: element (pixel p, channel k) at p * vrc_ps + k * vrc_cs
    uint32_t vrc_strided;
    int64_t vrc_ps, vrc_cs;
    const float *v_render_alphas; // [I, H, W, 1]
    // backward output (zero-initialised by the caller; accumulated with atomics): ONE array-of-structures buffer
    // [R][row_stride]; row = (v_mean2d.x, v_mean2d.y, v_conic.a, v_conic.b, v_conic.c, v_opacity,
    // [|v_mean2d.x|, |v_mean2d.y| when has_abs], v_color[0..cdim)). A Gaussian's gradients share a cache line,
    // and the flush writes consecutive floats from consecutive lanes (see raster3d_bwd.hip).
    float *v_rows;
    uint32_t row_stride;
    // sparse pixel layout of gsplat::rasterize_to_pixels_sparse (all null / 0 for dense images): one workgroup per
    // ACTIVE tile, only the pixels whose bit is set are rendered, and pixel outputs / cotangents are rows of packed
    // [P, ...] tensors in the caller's pixel order (reference RasterizeSparseAddressing.cuh, SparseTileLayout.cu)
    const int32_t *sp_active_tiles; // [AT] dense tile ids (image * tiles_per_image + tile), ascending
    const uint64_t *sp_pixel_mask;  // [AT, sp_words] raster-order bitmask (bit = row_in_tile * tile_size + col_in_tile)
    const int64_t *sp_pixel_cumsum; // [AT] inclusive number of requested pixels per active tile
    const int64_t *sp_pixel_map;    // [P] position in (tile, in-tile) order -> caller's pixel index
    uint32_t n_active, sp_words;
    // long tile lists cut into SEGMENTS that separate workgroups composite (raster3d_seg.hip; dense layouts only). All
    // zero / null: one workgroup per tile walks its whole list.
    //   seg_mode 0  per-tile launch; with seg_len > 0 it leaves the tiles longer than seg_len to the launches below
    //   seg_mode 1  one workgroup per SEGMENT item (tile block, first list index), list slice [first, first + seg_len):
    //               transmittance pass - the product of (1 - alpha) over the slice, from 1 (0 = the pixel stops inside)
    //   seg_mode 2  same items: compositing pass, every pixel starts at the transmittance in front of its segment
    uint32_t seg_mode, seg_len, seg_grid;
    uint32_t seg_cut; // lists LONGER than this are cut into slices of seg_len (raster3d_seg.hip: seg_cut_for)
    const int32_t *seg_items; // [n][2]
    const int32_t *seg_count; // number of items, on the device
    float *seg_T;             // [item][256]  mode 1 out: transmittance of the slice; then (prefix kernel) in front of it: mode 2 in
    float *seg_out;           // [item][nch + 1][256]  mode 2 out: partial colours, transmittance at the end of the slice
    int32_t *seg_last;        // [item][256]  mode 2 out: last contributing list index, -1 = none
    // backward, dense layouts: workgroup -> tile map sorted by work (raster3d_bwd.hip: tile_order_*), or null = launch order
    const int32_t *tile_order;
    // forward, dense layouts (one-wave kernel, raster3d_fwd_w.hip): [I * tile_h * tile_w] what each tile costs the backward - its
    // list up to the last contributor, the quantity tile_order.hip sorts by - or null
    int32_t *tile_cost;
};

// Block index -> (image, tile) with an XCD-aware remap: hardware places workgroup b on
// XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch"); neighbouring tiles share Gaussians,
// so give each XCD a contiguous run of tiles to keep those rows in ITS private L2.
// Pure performance remap: any placement is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n_blocks)
{
    constexpr uint32_t kXcds = 8;
    const uint32_t per_xcd   = (n_blocks + kXcds - 1) / kXcds;
    return (b % kXcds) * per_xcd + (b / kXcds);
}

// Which tile does this workgroup own, which slice of the sorted intersection list, and which output row does this lane
// write? Dense: tile = block, row = pixel index in the [I, H, W] image. Sparse: tile = active_tiles[block], list slice from
// the per-active-tile offsets, row = pixel_map[first row of the tile + rank of the lane's pixel among the set bits].
// index of cotangent (pixel row `pix`, channel k) in v_render_colors: contiguous rows of cdim floats, or any layout that is
// linear in the pixel index (an expanded scalar - the gradient of sum() - has both strides 0)
template <typename Args>
__device__ __forceinline__ size_t vrc_index(const Args &a, size_t pix, uint32_t k)
{
    return a.vrc_strided ? (size_t)((int64_t)pix * a.vrc_ps + (int64_t)k * a.vrc_cs) : pix * a.cdim + k;
}

struct TileCtx {
    uint32_t image_id, tile_id, tile_x, tile_y;
    int32_t range_start, range_end;
};
template <class Args> // Raster3DArgs or QueryArgs (raster_query.hip): same grid / layout field names
__device__ __forceinline__ bool tile_context(const Args &a, uint32_t block, TileCtx &t)
{
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    if (a.sp_active_tiles) {
        const uint32_t at = xcd_remap(block, a.n_active);
        if (at >= a.n_active) return false;
        const uint32_t gt = (uint32_t)a.sp_active_tiles[at];
        t.image_id = gt / tiles_per_image;
        t.tile_id  = gt % tiles_per_image;
        t.range_start = a.isect_offsets[at];
        t.range_end   = a.isect_offsets[at + 1]; // [AT + 1] with the n_isects sentinel
    } else {
        const uint32_t n_blocks = tiles_per_image * a.n_images;
        const uint32_t blk      = xcd_remap(block, n_blocks);
        if (blk >= n_blocks) return false;
        t.image_id = blk / tiles_per_image;
        t.tile_id  = blk % tiles_per_image;
        t.range_start = a.isect_offsets[blk];
        t.range_end   = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    }
    t.tile_x = t.tile_id % a.tile_w;
    t.tile_y = t.tile_id / a.tile_w;
    return true;
}
// The compositing launches' version: segment items (seg_mode 1 / 2), else tile_context minus the tiles that are left to
// the segment launches. `item` = index of the workgroup's item (modes 1, 2).
__device__ __forceinline__ bool tile_context_seg(const Raster3DArgs &a, uint32_t block, TileCtx &t, uint32_t &item)
{
    item = 0;
    if (a.seg_mode == 0u) {
        if (!tile_context(a, block, t)) return false;
        return !(a.seg_len && !a.sp_active_tiles && (uint32_t)(t.range_end - t.range_start) > a.seg_cut);
    }
    const int32_t n_items = *a.seg_count;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h, n_blocks = tiles_per_image * a.n_images;
    if ((int32_t)block >= n_items) {
        // compositing pass: the workgroups behind the segment items take the SHORT tiles, whole list, straight into the
        // image (item = ~0) - one launch for everything, the long segments first
        if (a.seg_mode != 2u) return false;
        uint32_t blk = block - (uint32_t)n_items;
        if (a.tile_order) { // backward, one wave per unit: the short tiles longest-first too (tile_order.hip; per XCD range)
            if (blk >= ((n_blocks + 7u) / 8u) * 8u) return false;
            const uint32_t idx = xcd_remap(blk, n_blocks);
            if (idx >= n_blocks) return false;
            blk = (uint32_t)a.tile_order[idx];
        }
        if (blk >= n_blocks) return false;
        item       = 0xFFFFFFFFu;
        t.image_id = blk / tiles_per_image;
        t.tile_id  = blk % tiles_per_image;
        t.tile_x   = t.tile_id % a.tile_w;
        t.tile_y   = t.tile_id / a.tile_w;
        t.range_start = a.isect_offsets[blk];
        t.range_end   = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
        return (uint32_t)(t.range_end - t.range_start) <= a.seg_cut;
    }
    item = block;
    const uint32_t blk = (uint32_t)a.seg_items[2 * block];
    t.image_id = blk / tiles_per_image;
    t.tile_id  = blk % tiles_per_image;
    t.tile_x   = t.tile_id % a.tile_w;
    t.tile_y   = t.tile_id / a.tile_w;
    const int32_t tile_end = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    t.range_start = a.seg_items[2 * block + 1];
    t.range_end   = min(t.range_start + (int32_t)a.seg_len, tile_end);
    return true;
}

// output row of pixel (lx, ly) of the tile, or -1 when the pixel is not rendered
template <class Args>
__device__ __forceinline__ int64_t pixel_row(const Args &a, const TileCtx &t, uint32_t block, uint32_t lx, uint32_t ly)
{
    if (!(lx < a.tile_size && ly < a.tile_size)) return -1;
    const uint32_t ox = t.tile_x * a.tile_size + lx, oy = t.tile_y * a.tile_size + ly;
    if (!(ox < a.width && oy < a.height)) return -1;
    if (!a.sp_active_tiles) return ((int64_t)t.image_id * a.height + oy) * a.width + ox;
    const uint32_t at      = xcd_remap(block, a.n_active);
    const uint64_t *words  = a.sp_pixel_mask + (size_t)at * a.sp_words;
    const uint32_t in_tile = ly * a.tile_size + lx, word = in_tile >> 6, bit = in_tile & 63u;
    if (!((words[word] >> bit) & 1ull)) return -1;
    uint32_t rank = __popcll(words[word] & ((1ull << bit) - 1ull));
    for (uint32_t w = 0; w < word; ++w) rank += __popcll(words[w]);
    const int64_t first = at == 0 ? 0 : a.sp_pixel_cumsum[at - 1];
    return a.sp_pixel_map[first + rank];
}

// Longest tiles first (csrc/tile_order.hip): builds, in `ws`, the workgroup -> tile map of a backward compositing launch
// sorted by the tile's cost (its list up to its last contributor), per XCD. Returns the map (indexed by xcd_remap()), or null
// when the launch keeps its launch order (no / too small a workspace, tiles other than 16 x 16, fewer tiles than a round of
// workgroup slots, GSX_RASTER3D_BWD_ORDER=0); *rc is a GSX_* code.
int64_t tile_order_workspace_bytes(uint32_t n_images, uint32_t tile_w, uint32_t tile_h);
const int32_t *build_tile_order(const int32_t *isect_offsets, const int32_t *last_ids, uint32_t n_images, uint32_t tile_size,
                                uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height, uint32_t n_isects, void *ws,
                                int64_t ws_bytes, hipStream_t stream, int *rc, void *zero_ptr = nullptr, int64_t zero_bytes = 0);

// One wave per tile (raster3d_fwd_w.hip): <= 4 channels per launch, 16 x 16 tiles, no segments. GSX_RASTER3D_FWD=q|w at run time.
// NOT the default: measured SLOWER than the four-waves-per-tile kernel (0.290 against 0.199 ms at c3, see raster3d_fwd_w.hip).
#ifndef GSX_RASTER3D_FWD_DEFAULT
#define GSX_RASTER3D_FWD_DEFAULT 'q'
#endif
// Wide colour rows (5 .. 32 channels per launch, 16 x 16 tiles, no segments) on the matrix cores: raster3d_fwd_m.hip.
// GSX_RASTER3D_FWD_WIDE=q keeps the four-wave kernel.
bool raster3d_fwd_m_applies(const Raster3DArgs &a);
int raster3d_fwd_m_launch(const Raster3DArgs &a, hipStream_t stream);
bool raster3d_fwd_w_applies(const Raster3DArgs &a);
int raster3d_fwd_w_launch(const Raster3DArgs &a, hipStream_t stream);

// ---- longest tiles first: csrc/tile_order.hip builds the order, the backward kernels read it through Raster3DArgs::tile_order ----
// tile_context() through the order (dense layouts)
__device__ __forceinline__ bool tile_context_ordered(const Raster3DArgs &a, uint32_t block, TileCtx &t)
{
    const uint32_t tiles_per_image = a.tile_w * a.tile_h, n_blocks = tiles_per_image * a.n_images;
    const uint32_t idx = xcd_remap(block, n_blocks);
    if (idx >= n_blocks) return false;
    const uint32_t blk = (uint32_t)a.tile_order[idx];
    t.image_id = blk / tiles_per_image;
    t.tile_id  = blk % tiles_per_image;
    t.tile_x   = t.tile_id % a.tile_w;
    t.tile_y   = t.tile_id / a.tile_w;
    t.range_start = a.isect_offsets[blk];
    t.range_end   = (blk == n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    return true;
}

// Wide colour rows (5 .. 32 channels per launch, 16 x 16 tiles, no absgrad) on the matrix cores: raster3d_bwd_m.hip.
// GSX_RASTER3D_BWD_WIDE=r keeps the reduction kernel.
bool raster3d_bwd_m_applies(const Raster3DArgs &a, bool has_abs);
int raster3d_bwd_m_launch(const Raster3DArgs &a, hipStream_t stream);

// thread -> pixel inside the tile.
__device__ __forceinline__ void tile_pixel(uint32_t tid, uint32_t tile_size, uint32_t &lx, uint32_t &ly)
{
    if (tile_size == 16) {
        const uint32_t w = tid >> 6, l = tid & 63u;
        lx = ((w & 1u) << 3) | (l & 7u);
        ly = ((w >> 1) << 3) | (l >> 3);
    } else {
        lx = tid % tile_size;
        ly = tid / tile_size; // >= tile_size for surplus threads -> treated as outside
    }
}

// ---- staged form of one Gaussian in LDS -------------------------------------------------------------
// The compositing loops evaluate alpha in base 2 with the constants folded in at staging time (once per tile and
// Gaussian instead of once per pixel and Gaussian):
//   opac * exp(-sigma) = exp2(lo - q),   lo = log2(opac),   q = A dx^2 + B dx dy + C dy^2,
//   A = log2(e)/2 * conic.a,  B = log2(e) * conic.b,  C = log2(e)/2 * conic.c          (sigma < 0  <=>  q < 0)
// which is one v_exp_f32 and no multiply by log2(e) / opacity per pixel. Same quantity as the reference's
// `opac * __expf(-sigma)` (RasterizeToPixels3DGSDevice.cuh:44-56) to ~1e-6 relative.
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ void stage_gaussian(float x, float y, float opac, float ca, float cb, float cc, float4 &ga,
                                               float2 &gb)
{
    const float lo = opac > 0.0f ? __log2f(opac) : -INFINITY; // opac <= 0 (or NaN) can never pass the alpha test
    ga = make_float4(x, y, lo, 0.5f * kLog2e * ca);
    gb = make_float2(kLog2e * cb, 0.5f * kLog2e * cc);
}
// q and the unclamped alpha of the staged Gaussian at offset (dx, dy) = mean - pixel
__device__ __forceinline__ float staged_q(const float4 &ga, const float2 &gb, float dx, float dy)
{
    return fmaf(dx, fmaf(ga.w, dx, gb.x * dy), gb.y * dy * dy);
}
__device__ __forceinline__ float staged_alpha_raw(const float4 &ga, float q)
{
    return __builtin_amdgcn_exp2f(ga.z - q);
}
// The exponent on a (x', y', lo, A | B, C) row with (x', y') = mean - TILE CENTRE and (dx, dy) = (x', y') - (pixel - tile centre):
// instruction for instruction what staged_f() evaluates on its (ax, ay, lo | nA, nB, nC) row (negation is exact), so that the
// reduction backward and the query kernels take the alpha test exactly where the forward took it. sigma < 0  <=>  e > ga.z.
__device__ __forceinline__ float staged_e_offset(const float4 &ga, const float2 &gb, float dx, float dy)
{
    return fmaf(dx, fmaf(-gb.x, dy, -ga.w * dx), fmaf(-gb.y * dy, dy, ga.z));
}

// ---- staged form, tile-centre polynomial ("e-form"; raster3d_fwd.hip and variant T of raster3d_bwd.hip) -------------------
// With a = mean - c (c = centre of the tile) and a lane's pixel at u = pixel - c (|u|, |v| <= 7.5, exact in fp32):
//   q(u, v) = A (a.x - u)^2 + B (a.x - u)(a.y - v) + C (a.y - v)^2
//           = q0 - u gu - v gv + A u^2 + B u v + C v^2,   q0 = q(0, 0),  gu = 2 A a.x + B a.y,  gv = B a.x + 2 C a.y
// so the exponent of alpha = exp2(lo - q) is
//   e(u, v) = e0 + u (gu + nA u + nB v) + v (gv + nC v),   e0 = lo - q0,  (nA, nB, nC) = -(A, B, C)
// i.e. FIVE fmas per (pixel, Gaussian) instead of 2 subtractions + 5 for q + 1 for lo - q; e0 / gu / gv are formed once
// per (tile, Gaussian) by the staging thread. sigma < 0  <=>  q < 0  <=>  e > lo. Conditioning: the terms that cancel are
// bounded by |u| (|gu| + ...) with |u| <= 7.5 and, for the Gaussians that can pass the alpha test at all, |a| <= 7.5 +
// extent: a few hundred at most for the tightest footprint (eps2d = 0.3) -> ~2e-5 absolute on e, ~1.5e-5 relative on alpha
// (the tolerance of the parity tests is 1e-4 .. 1e-3; variant T's moments about the tile centre are conditioned alike).
//
// The sigma < 0 test. Mathematically e <= lo wherever the conic is positive definite, with equality at the mean; in fp32 the
// cancelling terms leave e a few 1e-5 ABOVE lo for a pixel centre that sits (almost) exactly on the mean - and `e > lo` then
// dropped the Gaussian at the one pixel where it weighs most (found by the reference's own
// test_rasterize_num_contributing_gaussians, whose means are pixel centres). The staged reject level is therefore
// lo + kLoMargin: a pair is rejected only where sigma < -kLoMargin / log2(e) ~ -8.5e-5, which a valid conic never reaches and
// which costs an invalid one nothing that matters (alpha within 1.0001 of the opacity). Readers that need lo itself (the
// backward's 1 / opacity) subtract the margin again.
#ifndef GSX_DFORM
#define GSX_DFORM 1 // 1: every kernel evaluates the exponent from the offset (stage_gaussian_f / staged_f below): e == lo at the mean
#endif
constexpr float kLoMargin = GSX_DFORM ? 0.0f : 1.220703125e-4f; // 2^-13 for the polynomial; the offset form needs none
struct StagedRow { // one staged Gaussian in LDS: 48 bytes, read as b128 + b128 + b64 from ONE address register
    v4f p0;        // e0, gu, gv, lo + kLoMargin
    v4f p1;        // nA, nB, nC, colour 2
    v4f p2;        // colour 0, colour 1, colour 3, -
};
__device__ __forceinline__ void stage_gaussian_e(float ax, float ay, float opac, float ca, float cb, float cc, v4f &p0,
                                                 float &nA, float &nB, float &nC)
{
    const float lo = opac > 0.0f ? __log2f(opac) : -INFINITY; // opac <= 0 (or NaN) can never pass the alpha test
    const float A = 0.5f * kLog2e * ca, B = kLog2e * cb, C = 0.5f * kLog2e * cc;
    const float q0 = fmaf(ax, fmaf(A, ax, B * ay), C * ay * ay);
    p0 = v4f{lo - q0, fmaf(2.0f * A, ax, B * ay), fmaf(B, ax, 2.0f * C * ay), lo + kLoMargin};
    nA = -A; nB = -B; nC = -C;
}
__device__ __forceinline__ float staged_e(const v4f &p0, float nA, float nB, float nC, float u, float v)
{
    const float t1 = fmaf(nB, v, fmaf(nA, u, p0.y));
    const float t2 = fmaf(nC, v, p0.z);
    return fmaf(v, t2, fmaf(u, t1, p0.x));
}

// ---- staged form of the FORWARD-side kernels ("d-form": raster3d_fwd*.hip, raster_indices.hip) ---------------------------------
// The tile-centre polynomial costs five fmas per (pixel, Gaussian) but loses digits where its terms cancel: |e - exact| reaches
// ~2e-5 (p99.99) on log2(alpha), 1.5e-5 relative on alpha - which the REFERENCE'S OWN forward test does not allow: it compares
// the rasterizer with its PyTorch restatement at atol 1e-5 (tests/test_basic.py:2639, torch.testing.assert_close defaults), and
// over the shim 23 of 2.4 M colour values missed that by up to 0.41e-5 (profiles/r10_reference_suite.txt). The forward kernels
// therefore evaluate the exponent the way the reference does, from the offset d = mean - pixel (two subtractions more per pair):
//   e = lo - (A dx^2 + B dx dy + C dy^2) = fma(dx, fma(nB, dy, nA dx), fma(nC dy, dy, lo))
// whose error is a few ulp of the RESULT (~3e-6 at p99.99). e == lo exactly at the mean, so the sigma < 0 test needs no margin.
// The BACKWARD kernels evaluate the same function through the same instructions (stage_gaussian_f / staged_f): a pair is blended
// by the backward iff the forward blended it - with the polynomial in the backward and the offset form in the forward a pair
// within 1.5e-5 of the 1/255 test was taken by one and not by the other (6 of 369 k rows of v_means2d off by up to 6e-3 in the
// reference's test_rasterize_to_pixels). Their moments stay in the tile-centre frame (u, v); one wave per tile (variant W)
// shares dx / dy between the four pixels of a lane, so the offset form costs it 18 instructions where the polynomial took 20.
// GSX_DFORM=0 builds the polynomial into every kernel again (A/B: no measurable difference at c3, profiles/r10_ab.md).
__device__ __forceinline__ void stage_gaussian_f(float ax, float ay, float opac, float ca, float cb, float cc, v4f &p0,
                                                 float &nA, float &nB, float &nC)
{
#if GSX_DFORM
    const float lo = opac > 0.0f ? __log2f(opac) : -INFINITY; // opac <= 0 (or NaN) can never pass the alpha test
    p0 = v4f{ax, ay, lo, lo}; // mean - tile centre | base of the exponent | reject level (e > lo <=> sigma < 0)
    nA = -0.5f * kLog2e * ca; nB = -kLog2e * cb; nC = -0.5f * kLog2e * cc;
#else
    stage_gaussian_e(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
#endif
}
// (u, v) = pixel centre - tile centre
__device__ __forceinline__ float staged_f(const v4f &p0, float nA, float nB, float nC, float u, float v)
{
#if GSX_DFORM
    const float dx = p0.x - u, dy = p0.y - v;
    return fmaf(dx, fmaf(nB, dy, nA * dx), fmaf(nC * dy, dy, p0.z));
#else
    return staged_e(p0, nA, nB, nC, u, v);
#endif
}

// ---- wave-level culling -------------------------------------------------------------------------
// A Gaussian can only pass the reference's `alpha >= 1/255` test (Device.cuh:52-55) at offsets d with
// sigma(d) = 1/2 d^T Q d <= L = ln(255 * opacity). The staging thread of each Gaussian computes the
// axis-aligned half extents of that ellipse ONCE per tile (with a safety margin that dwarfs the
// rounding of __expf/__logf), and every wave tests 64 staged Gaussians at a time (one per lane)
// against the rectangle of ITS pixel centres; a 64-bit ballot then drives a scalar loop over the
// survivors only. This never changes a result: a culled (wave, Gaussian) pair has no lane that
// would have passed the alpha test.
__device__ __forceinline__ float2 cull_half_extent(float opac, float ca, float cb, float cc)
{
    const float L   = __logf(255.0f * opac) + 0.01f; // NaN / -inf for opac <= 0 -> never hits
    const float det = ca * cc - cb * cb;
    if (!(det > 0.0f)) return make_float2(INFINITY, INFINITY); // not positive definite: do not cull
    if (!(L > 0.0f)) return make_float2(-1.0f, -1.0f);
    const float k = 2.0f * L / det;
    // negative-definite conics give sqrt(negative) = NaN -> compares false -> culled (sigma < 0 everywhere)
    return make_float2(sqrtf(k * cc) * 1.0001f + 1e-3f, sqrtf(k * ca) * 1.0001f + 1e-3f);
}

// Rectangle (centre, half size) of the pixel centres owned by the active lanes of this wave.
struct WaveRect {
    float cx, cy, hw, hh;
    bool any;
};
__device__ __forceinline__ WaveRect wave_pixel_rect(bool inside, float px, float py)
{
    float xmin = inside ? px : INFINITY, xmax = inside ? px : -INFINITY;
    float ymin = inside ? py : INFINITY, ymax = inside ? py : -INFINITY;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o));
        xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        ymin = fminf(ymin, __shfl_xor(ymin, o));
        ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    }
    WaveRect r;
    r.any = xmin <= xmax;
    r.cx = 0.5f * (xmin + xmax); r.hw = 0.5f * (xmax - xmin);
    r.cy = 0.5f * (ymin + ymax); r.hh = 0.5f * (ymax - ymin);
    return r;
}

// Second, exact stage of the wave-level culling (for the kernels that stage the e-form, StagedRow above): the minimum of the
// exponent's quadratic q over the wave's pixel rectangle, against the level where alpha = exp2(lo - q) drops below 1/255.
// The axis-aligned box of cull_half_extent() lets through (wave, Gaussian) pairs whose ellipse misses the rectangle
// diagonally - 13 % of the surviving pairs on c3 had no lane that passes the alpha test (r05 counters). q is convex, so its
// minimum over a box lies on the edges that face the mean: one 1-D minimisation (with clamping) along the nearer vertical
// edge, one along the nearer horizontal edge. Conservative: a margin of 0.01 in log2 units (0.7 % in alpha) dwarfs the
// rounding of the staged form; conics that are not positive definite are never culled (like cull_half_extent).
__device__ __forceinline__ bool rect_reaches_level(const v4f &p0, const v4f &p1, float ax, float ay, const WaveRect &r)
{
    const float A = -p1.x, B = -p1.y, C = -p1.z; // log2(e)/2 a, log2(e) b, log2(e)/2 c
    if (!(A > 0.0f && C > 0.0f && 4.0f * A * C > B * B)) return true;
    // offsets d = mean - pixel over the rectangle
    const float dxl = ax - r.cx - r.hw, dxh = ax - r.cx + r.hw, dyl = ay - r.cy - r.hh, dyh = ay - r.cy + r.hh;
    const float dxe = fminf(fmaxf(0.0f, dxl), dxh), dye = fminf(fmaxf(0.0f, dyl), dyh); // nearest edge (0 if the mean is inside)
    const float dys = fminf(fmaxf(-0.5f * B * dxe * __builtin_amdgcn_rcpf(C), dyl), dyh);
    const float dxs = fminf(fmaxf(-0.5f * B * dye * __builtin_amdgcn_rcpf(A), dxl), dxh);
    const float q1 = fmaf(dxe, fmaf(A, dxe, B * dys), C * dys * dys);
    const float q2 = fmaf(dxs, fmaf(A, dxs, B * dye), C * dye * dye);
    return fminf(q1, q2) <= p0.w + (7.99435344f + 0.01f); // log2(255)
}

} // namespace gsx
