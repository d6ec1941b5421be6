// Long tile lists cut into segments that separate workgroups composite (forward), gfx950.
// C-ABI entries: gsx_raster3d_seg_workspace_bytes, gsx_raster3d_fwd_seg.
//
// One workgroup per tile (the reference's decomposition too: RasterizeToPixels3DGSSerialBatchFwd.cu:41) makes the launch as
// long as its LONGEST list. Real scenes have skewed lists - the reference's garden profile: mean 386 entries, 99th percentile
// 4107, longest 8822 on 8160 tiles, 2.7x the ideal per-slot load whatever the launch order - so a tile whose list exceeds
// `seg_len` is cut into segments of seg_len entries, one workgroup each:
//   1. seg_plan       one thread per tile: tiles longer than seg_len -> segment items (tile, first index) + a long-tile record
//   3. raster3d_fwd   seg_mode 1: TRANSMITTANCE pass - per segment and pixel the product of (1 - alpha) over the slice
//                     (0 if the reference's early termination fires inside the slice even when entered at T = 1)
//   4. seg_prefix     one workgroup per long tile: running product over its segments -> the transmittance IN FRONT of every
//                     segment, per pixel
//   5. raster3d_fwd   seg_mode 2: compositing pass - ONE launch for the segments (first) and the short tiles (behind them,
//                     whole list, straight into the image); every pixel of a segment starts at that transmittance, so colours,
//                     the early-termination rule (T' <= 1e-4 stops the pixel and excludes the Gaussian: compared on the true
//                     running transmittance) and last_ids are those of the sequential walk; a pixel whose prefix is already
//                     <= 1e-4 stopped in an earlier segment and does nothing (transmittance only decreases)
//   6. seg_combine    one workgroup per long tile: colours add up, the final transmittance / last contributor come from the
//                     last segment the pixel was alive in
// The only difference to the sequential walk is the association order of the transmittance products (segment products
// first): ~1e-7 relative, which can move a threshold decision on a pixel that sits exactly on it - the class of difference
// the reference's own CUDA-vs-torch tolerances cover. The price is pass 3: the alphas of a long tile are evaluated twice
// (~0.6 of a forward); in exchange its critical path is one segment instead of the whole list.
#include <cstdlib>

#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

int raster3d_fwd_launch_chunk(const Raster3DArgs &a, hipStream_t stream);   // raster3d_fwd.hip
int raster3d_bwd_prepass_launch(const Raster3DArgs &a, hipStream_t stream); // raster3d_fwd.hip (forward kernel, PRE = true)
int raster3d_bwd_t_launch_items(const Raster3DArgs &a, hipStream_t stream); // raster3d_bwd.hip
int raster3d_bwd_w_launch_items(const Raster3DArgs &a, hipStream_t stream); // raster3d_bwd.hip
bool raster3d_bwd_uses_variant_t();
bool raster3d_bwd_uses_variant_w();

// Slice length of the BACKWARD for a forward slice length: the one-wave-per-tile kernel (variant W) walks a slice as ONE
// instruction stream, so its slices are half as long as those of the four-wave kernels (the critical path of a launch is
// its longest unit of work); the pre-pass costs the same either way (it evaluates every entry of the long lists once).
// Garden x25, backward in ms (profiles/r09_ab.md): slices of 1024 0.863, 512 0.634 - 0.653, 256 0.667 - 0.698; variant T on
// slices of 1024: 0.647 - 0.656. GSX_BWD_SEG_DIV overrides the divisor (A/B).
// Which kernel walks the slices: variant T (four waves per unit) unless GSX_RASTER3D_BWD_SEG=w. Variant W over half-length
// slices ties it on the garden x25 scene (0.645 against 0.646 ms) and LOSES on the 49 M-Gaussian scene of the reference's
// profiling table (7.1 M intersections: 1.21 against 0.89 ms; slices of 1024 / 256: 1.12 / 1.27) - profiles/r09_ab.md #40.
static bool seg_bwd_on_variant_w()
{
    static const bool w = [] {
        const char *e = getenv("GSX_RASTER3D_BWD_SEG");
        return e && e[0] == 'w';
    }();
    return w && raster3d_bwd_uses_variant_w();
}

static uint32_t bwd_slice_len(uint32_t seg_len)
{
    if (!seg_bwd_on_variant_w() || seg_len == 0) return seg_len;
    static const uint32_t div = [] {
        const char *e = getenv("GSX_BWD_SEG_DIV");
        const int v   = e ? atoi(e) : 2;
        return (uint32_t)(v >= 1 ? v : 1);
    }();
    const uint32_t l = seg_len / div;
    return l < 256u ? 256u : l;
}

struct SegHeader { // device memory, zeroed before every use
    int32_t n_items, n_long, pad[2];
};

struct SegPlan {
    SegHeader *hdr;
    int32_t *items; // [max_items][2] (tile block, first list index)
    int32_t *longs; // [max_long][3]  (tile block, first item, number of segments)
    int32_t *last;  // [max_items][256]
    float *T;       // [max_items][256]
    float *out;     // [max_items][nch_max + 1][256]
    uint32_t max_items, max_long;
};

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// Which lists are cut: those longer than max(2 slices, 3 x the mean list). Segments pay off for OUTLIERS - a few crowded tiles
// that would otherwise bound the launch (garden: mean 386, longest 8822). When every list is long (c4: 4 M Gaussians per
// image, mean 1800, longest ~2100) the per-tile walk with its early termination is the faster one: cutting everything
// there took the forward from 0.64 to 3.4 ms.
static uint32_t seg_cut_for(int64_t n_isects, uint32_t n_tiles_total, uint32_t seg_len)
{
    const int64_t mean3 = n_tiles_total ? 3 * (n_isects / (int64_t)n_tiles_total) : 0;
    const int64_t cut   = mean3 > 2 * (int64_t)seg_len ? mean3 : 2 * (int64_t)seg_len;
    return (uint32_t)(cut > 0x7FFFFFFF ? 0x7FFFFFFF : cut);
}

extern "C" int64_t gsx_raster3d_seg_cut(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, uint32_t seg_len)
{
    if (seg_len == 0) return INT64_MAX;
    // WHETHER to take the segment entries at all: a launch is only bound by its longest list while that list is longer than
    // a workgroup slot's share of all the work (256 CUs x ~4 resident workgroups). garden x25 at batch 1: 3.15 M
    // intersections (3080 per slot), lists of up to 8822 - segments take the forward from 0.59 to 0.37 ms; the same scene at
    // batch 4 (12.6 M, 12.3 k per slot) runs its forward in 0.72 ms per tile and in 1.12 ms in segments.
    const int64_t cut = (int64_t)seg_cut_for(n_isects, n_images * tile_w * tile_h, seg_len), share = n_isects / 1024;
    return share > cut ? share : cut;
}

static void seg_bounds(int64_t n_isects, uint32_t n_tiles_total, uint32_t seg_len, uint32_t &max_items, uint32_t &max_long)
{
    const int64_t by_len = n_isects / seg_len; // a long tile has more than seg_cut >= 2 seg_len entries
    max_long  = (uint32_t)(by_len < (int64_t)n_tiles_total ? by_len : (int64_t)n_tiles_total);
    max_items = (uint32_t)(by_len + max_long); // ceil(len / seg_len) <= len / seg_len + 1 per long tile
}

static int64_t seg_layout(int64_t n_isects, uint32_t n_tiles_total, uint32_t nch_max, uint32_t seg_len, unsigned char *base,
                          SegPlan *p)
{
    SegPlan t{};
    seg_bounds(n_isects, n_tiles_total, seg_len, t.max_items, t.max_long);
    unsigned char *q = base;
    auto take = [&](int64_t bytes) {
        unsigned char *r = q;
        q += align256(bytes);
        return r;
    };
    t.hdr   = reinterpret_cast<SegHeader *>(take(sizeof(SegHeader)));
    t.items = reinterpret_cast<int32_t *>(take((int64_t)t.max_items * 8));
    t.longs = reinterpret_cast<int32_t *>(take((int64_t)t.max_long * 12));
    t.last  = reinterpret_cast<int32_t *>(take((int64_t)t.max_items * 256 * 4));
    t.T     = reinterpret_cast<float *>(take((int64_t)t.max_items * 256 * 4));
    t.out   = reinterpret_cast<float *>(take((int64_t)t.max_items * (nch_max + 1) * 256 * 4));
    if (p) *p = t;
    return (int64_t)(q - base);
}

__global__ void __launch_bounds__(256) seg_plan_kernel(const int32_t *offsets, uint32_t n_blocks, uint32_t n_isects,
                                                       uint32_t seg_len, uint32_t seg_cut, SegPlan p)
{
    const uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks) return;
    const int32_t start = offsets[blk], end = (blk == n_blocks - 1) ? (int32_t)n_isects : offsets[blk + 1];
    const uint32_t len  = (uint32_t)(end - start);
    if (len <= seg_cut) return;
    const uint32_t n_seg = (len + seg_len - 1) / seg_len;
    const int32_t li = atomicAdd(&p.hdr->n_long, 1);
    const int32_t s0 = atomicAdd(&p.hdr->n_items, (int32_t)n_seg);
    p.longs[3 * li] = (int32_t)blk; p.longs[3 * li + 1] = s0; p.longs[3 * li + 2] = (int32_t)n_seg;
    for (uint32_t k = 0; k < n_seg; ++k) {
        p.items[2 * (s0 + (int32_t)k)]     = (int32_t)blk;
        p.items[2 * (s0 + (int32_t)k) + 1] = start + (int32_t)(k * seg_len);
    }
}

// one workgroup per long tile, thread = pixel: slice transmittances -> transmittance in front of every slice (in place)
__global__ void __launch_bounds__(256) seg_prefix_kernel(SegPlan p)
{
    const int32_t li = (int32_t)blockIdx.x;
    if (li >= p.hdr->n_long) return;
    const int32_t s0 = p.longs[3 * li + 1], n_seg = p.longs[3 * li + 2];
    float P = 1.0f;
    for (int32_t k = 0; k < n_seg; ++k) {
        float *slot   = p.T + (size_t)(s0 + k) * 256 + threadIdx.x;
        const float t = *slot;
        *slot         = P;
        P *= t;
    }
}

// BACKWARD. The sequential backward walks a list back to front carrying, per pixel, the transmittance T (divided back Gaussian
// by Gaussian) and B = sum over the Gaussians BEHIND of alpha_i T_i (c_i . v_colour). A slice can start on its own once it
// knows both at its end: the pre-pass (forward kernel, PRE) gives every slice's own transmittance T_k and
// S_k = sum_i alpha_i T_i^(from 1) (c_i . v_colour); with P_k = T_0 ... T_(k-1) the transmittance in front of slice k,
//   T at the END of slice k = P_k T_k,      B at the end of slice k = sum_(j > k) P_j S_j.
// One workgroup per long tile, thread = pixel: T[item] <- end transmittance, S[item] <- B at the end (both in place).
__global__ void __launch_bounds__(256) seg_bwd_prefix_kernel(SegPlan p)
{
    const int32_t li = (int32_t)blockIdx.x;
    if (li >= p.hdr->n_long) return;
    const int32_t s0 = p.longs[3 * li + 1], n_seg = p.longs[3 * li + 2];
    float P = 1.0f;
    for (int32_t k = 0; k < n_seg; ++k) {
        const size_t at = (size_t)(s0 + k) * 256 + threadIdx.x;
        const float t   = p.T[at];
        p.out[at]       = P * p.out[at]; // P_k S_k
        P *= t;
        p.T[at] = P;                     // transmittance at the end of slice k
    }
    float behind = 0.0f;
    for (int32_t k = n_seg - 1; k >= 0; --k) {
        const size_t at = (size_t)(s0 + k) * 256 + threadIdx.x;
        const float ps  = p.out[at];
        p.out[at]       = behind;
        behind += ps;
    }
}

// The same two per-slice values WITHOUT a pre-pass, for a caller that kept the workspace of the forward call over the same lists
// (gsx_raster3d_bwd_seg_reuse): the forward's compositing pass left, per slice k and pixel, the colour sums C_k,c = sum_i alpha_i T_i
// c_i,c taken with the TRUE running transmittance (plane c of `out`) and that transmittance at the slice's end (plane nch), so
//   T at the end of slice k = out[k][nch],      B at the end of slice k = sum_(j > k) sum_c v_colour,c C_j,c.
// A pixel that stopped inside slice k keeps its final transmittance there and has zero sums behind it - exactly what the
// backward of its last contributor starts from. One workgroup per long tile of the FORWARD's plan, thread = pixel (tile_pixel).
__global__ void __launch_bounds__(256) seg_bwd_from_fwd_kernel(const Raster3DArgs a, SegPlan pf, float *T_end, float *B_end)
{
    const int32_t li = (int32_t)blockIdx.x;
    if (li >= pf.hdr->n_long) return;
    const uint32_t blk = (uint32_t)pf.longs[3 * li];
    const int32_t s0 = pf.longs[3 * li + 1], n_seg = pf.longs[3 * li + 2];
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    TileCtx tc;
    tc.image_id = blk / tiles_per_image; tc.tile_id = blk % tiles_per_image;
    tc.tile_x = tc.tile_id % a.tile_w; tc.tile_y = tc.tile_id / a.tile_w;
    tc.range_start = tc.range_end = 0;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, 0u, lx, ly);
    float v_c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (prow >= 0)
        for (uint32_t c = 0; c < a.nch; ++c) v_c[c] = a.v_render_colors[vrc_index(a, (size_t)prow, a.ch_off + c)];
    const uint32_t planes = a.nch + 1;
    const float *__restrict__ src = pf.out;
    float *__restrict__ t_out = T_end, *__restrict__ b_out = B_end;
    float behind = 0.0f;
    // four slices per round: their 4 (nch + 1) reads are independent of each other and of the stores (one slice per round was a
    // chain of dependent round trips to memory per workgroup: 24 us for ~300 workgroups)
    for (int32_t k1 = n_seg; k1 > 0; k1 -= 4) {
        float te[4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t k = k1 - 1 - j;
            te[j] = 0.0f; d[j] = 0.0f;
            if (k >= 0) {
                const size_t it = (size_t)(s0 + k);
                te[j] = src[(it * planes + a.nch) * 256 + tid];
                for (uint32_t c = 0; c < a.nch; ++c) d[j] = fmaf(v_c[c], src[(it * planes + c) * 256 + tid], d[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t k = k1 - 1 - j;
            if (k >= 0) {
                const size_t it = (size_t)(s0 + k);
                t_out[it * 256 + tid] = te[j];
                b_out[it * 256 + tid] = behind;
                behind += d[j];
            }
        }
    }
}

// one workgroup per long tile; thread = pixel in the per-tile launch's order (tile_pixel)
__global__ void __launch_bounds__(256) seg_combine_kernel(const Raster3DArgs a, SegPlan p)
{
    const int32_t li = (int32_t)blockIdx.x;
    if (li >= p.hdr->n_long) return;
    const uint32_t blk = (uint32_t)p.longs[3 * li];
    const int32_t s0 = p.longs[3 * li + 1], n_seg = p.longs[3 * li + 2];
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    TileCtx tc;
    tc.image_id = blk / tiles_per_image; tc.tile_id = blk % tiles_per_image;
    tc.tile_x = tc.tile_id % a.tile_w; tc.tile_y = tc.tile_id / a.tile_w;
    tc.range_start = tc.range_end = 0;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, 0u, lx, ly);
    if (prow < 0) return;
    const size_t pix      = (size_t)prow;
    const uint32_t planes = a.nch + 1;
    // the last segment the pixel was alive in gives the final transmittance; contributors only exist in such segments
    float T_final = 1.0f;
    int32_t last  = -1;
    for (int32_t k = 0; k < n_seg; ++k) {
        const size_t it = (size_t)(s0 + k);
        if (!(p.T[it * 256 + tid] > kTransmittanceThresh)) break; // stopped before this segment (and all later ones)
        T_final         = p.out[(it * planes + a.nch) * 256 + tid];
        const int32_t l = p.last[it * 256 + tid];
        last            = l >= 0 ? l : last;
    }
    const float *bg   = a.backgrounds ? a.backgrounds + (size_t)tc.image_id * a.cdim + a.ch_off : nullptr;
    const bool masked = a.masks && !a.masks[(size_t)tc.image_id * tiles_per_image + tc.tile_id];
    for (uint32_t c = 0; c < a.nch; ++c) {
        float acc = 0.0f;
        for (int32_t k = 0; k < n_seg; ++k) acc += p.out[((size_t)(s0 + k) * planes + c) * 256 + tid]; // front to back
        a.render_colors[pix * a.cdim + a.ch_off + c] = masked ? (bg ? bg[c] : 0.0f) : (bg ? acc + T_final * bg[c] : acc);
    }
    if (a.first_chunk) {
        a.render_alphas[pix] = masked ? 0.0f : 1.0f - T_final;
        a.last_ids[pix]      = masked ? 0 : (last >= 0 ? last : 0);
    }
}

} // namespace gsx

using namespace gsx;

extern "C" int64_t gsx_raster3d_seg_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h,
                                                    uint32_t cdim, uint32_t seg_len)
{
    if (seg_len == 0 || n_isects <= 0) return 512;
    const uint32_t nch_max = cdim > 32 ? 32 : cdim;
    return seg_layout(n_isects, n_images * tile_w * tile_h, nch_max, seg_len, nullptr, nullptr) + 512;
}

// workspace of gsx_raster3d_bwd_seg (its slices may be shorter than the forward's: bwd_slice_len)
extern "C" int64_t gsx_raster3d_bwd_seg_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h,
                                                        uint32_t cdim, uint32_t seg_len)
{
    // + the longest-first order of the short tiles (variant W: one wave per unit of work, tile_order.hip)
    return gsx_raster3d_seg_workspace_bytes(n_isects, n_images, tile_w, tile_h, cdim, bwd_slice_len(seg_len))
           + tile_order_workspace_bytes(n_images, tile_w, tile_h) + 256;
}

extern "C" int gsx_raster3d_fwd_seg(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects,
    uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *render_colors,
    float *render_alphas, int32_t *last_ids, uint32_t seg_len, void *workspace, int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_fwd_seg: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_fwd_seg: channels must be >= 1");
    GSX_REQUIRE(seg_len >= 256, "gsx_raster3d_fwd_seg: seg_len must be >= 256 (one staged batch), got %u", seg_len);
    GSX_REQUIRE(render_colors && render_alphas && last_ids, "gsx_raster3d_fwd_seg: null output");
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids), "gsx_raster3d_fwd_seg: null input");
    GSX_REQUIRE(isect_offsets != nullptr || n_images * tile_w * tile_h == 0, "gsx_raster3d_fwd_seg: null isect_offsets");
    hipStream_t s = (hipStream_t)stream;
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets; a.flatten_ids = flatten_ids;
    a.render_colors = render_colors; a.render_alphas = render_alphas; a.last_ids = last_ids;
    const uint32_t n_blocks = n_images * tile_w * tile_h;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t nch_max = cdim > 32 ? 32 : cdim;
    const uint32_t seg_cut = seg_cut_for(n_isects, n_blocks, seg_len);
    SegPlan p{};
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    if (workspace == nullptr
        || (base - reinterpret_cast<unsigned char *>(workspace)) + seg_layout(n_isects, n_blocks, nch_max, seg_len, base, &p) > workspace_bytes) {
        set_last_error("gsx_raster3d_fwd_seg: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    if (hipMemsetAsync(p.hdr, 0, sizeof(SegHeader), s) != hipSuccess) return check_launch("raster3d_fwd_seg memset");
    seg_plan_kernel<<<dim3((n_blocks + 255) / 256), dim3(256), 0, s>>>(isect_offsets, n_blocks, n_isects, seg_len, seg_cut, p);
    uint32_t off = 0;
    bool first   = true;
    do {
        const uint32_t rem = cdim - off;
        a.ch_off = off; a.nch = rem > 32 ? 32 : rem; a.first_chunk = first ? 1u : 0u;
        a.seg_len = seg_len; a.seg_cut = seg_cut;
        int rc;
        if (p.max_items > 0) {
            a.seg_items = p.items; a.seg_count = &p.hdr->n_items;
            a.seg_T = p.T; a.seg_out = p.out; a.seg_last = p.last;
            if (first) { // the transmittances do not depend on the channel chunk
                Raster3DArgs t = a;
                t.seg_mode = 1; t.seg_grid = p.max_items;
                t.nch = 1; // one channel is the cheapest instantiation; its colour sum is not stored
                rc = raster3d_fwd_launch_chunk(t, s);
                if (rc != GSX_OK) return rc;
                seg_prefix_kernel<<<dim3(p.max_long), dim3(256), 0, s>>>(p);
            }
            // ONE compositing launch: the segment items first (the longest units of work), the short tiles behind them
            // (in launch order: taking the short tiles longest-first like the one-wave backward does COSTS the forward 0.39 ->
            // 0.45 ms on the garden x25 scene - neighbouring tiles stop sharing an L2 - profiles/r11_ab.md #3)
            a.seg_mode = 2; a.seg_grid = p.max_items + n_blocks;
            rc = raster3d_fwd_launch_chunk(a, s);
            if (rc != GSX_OK) return rc;
            a.seg_mode = 0;
            seg_combine_kernel<<<dim3(p.max_long), dim3(256), 0, s>>>(a, p);
        } else {
            a.seg_mode = 0;
            rc = raster3d_fwd_launch_chunk(a, s);
            if (rc != GSX_OK) return rc;
        }
        off += a.nch;
        first = false;
    } while (off < cdim);
    return check_launch("raster3d_fwd_seg");
}

// Backward with long tile lists cut into segments (see seg_bwd_prefix_kernel). Applies where the backward runs its variant T
// (<= 4 channels, 16 x 16 tiles, no absgrad); anything else is the caller's to send to gsx_raster3d_bwd.
static int raster3d_bwd_seg_impl(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas, uint32_t n_images, uint32_t n_isects,
    uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *v_rows,
    uint32_t row_stride, uint32_t seg_len, const void *fwd_workspace, int64_t fwd_workspace_bytes, void *workspace,
    int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(tile_size == 16 && cdim >= 1 && cdim <= 4,
                "gsx_raster3d_bwd_seg: needs 16 x 16 tiles and <= 4 channels (got tile %u, %u channels): use gsx_raster3d_bwd",
                tile_size, cdim);
    if (!raster3d_bwd_uses_variant_t()) // GSX_RASTER3D_BWD=r (A/B switch): the reduction kernel has no segment support
        return gsx_raster3d_bwd(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
                                last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height, tile_size,
                                tile_w, tile_h, 0, v_rows, row_stride, stream);
    GSX_REQUIRE(seg_len >= 256, "gsx_raster3d_bwd_seg: seg_len must be >= 256, got %u", seg_len);
    if (n_isects == 0) return GSX_OK;
    const uint32_t fwd_seg_len = seg_len;
    seg_len = bwd_slice_len(seg_len); // workspace: gsx_raster3d_bwd_seg_workspace_bytes
    GSX_REQUIRE(v_rows && row_stride >= 6u + cdim, "gsx_raster3d_bwd_seg: gradient rows missing / too narrow");
    GSX_REQUIRE(means2d && conics && colors && opacities && flatten_ids && render_alphas && last_ids && v_render_colors
                && isect_offsets, "gsx_raster3d_bwd_seg: null input");
    hipStream_t s = (hipStream_t)stream;
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets; a.flatten_ids = flatten_ids;
    a.render_alphas = const_cast<float *>(render_alphas); a.last_ids = const_cast<int32_t *>(last_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas;
    a.v_rows = v_rows; a.row_stride = row_stride;
    a.ch_off = 0; a.nch = cdim; a.first_chunk = 1;
    const uint32_t n_blocks = n_images * tile_w * tile_h;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t seg_cut = seg_cut_for(n_isects, n_blocks, seg_len);
    SegPlan p{};
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    if (workspace == nullptr
        || (base - reinterpret_cast<unsigned char *>(workspace)) + seg_layout(n_isects, n_blocks, cdim, seg_len, base, &p) > workspace_bytes) {
        set_last_error("gsx_raster3d_bwd_seg: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    // The forward's workspace over the same lists (gsx_raster3d_bwd_seg_reuse): its plan IS this launch's plan and its
    // per-slice sums replace the pre-pass - when the backward cuts its slices as long as the forward did (variant T)
    SegPlan pf{};
    bool reuse = false;
    if (fwd_workspace && seg_len == fwd_seg_len) {
        unsigned char *fbase = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(fwd_workspace) + 255) & ~(uintptr_t)255);
        const int64_t need = (fbase - reinterpret_cast<const unsigned char *>(fwd_workspace))
                             + seg_layout(n_isects, n_blocks, cdim, fwd_seg_len, fbase, &pf);
        if (need > fwd_workspace_bytes) {
            set_last_error("gsx_raster3d_bwd_seg_reuse: forward workspace too small for these lists (not the forward call's?)");
            return GSX_ERR_WORKSPACE;
        }
        reuse = true;
    }
    a.seg_len = seg_len; a.seg_cut = seg_cut;
    a.seg_T = p.T; a.seg_out = p.out; a.seg_last = p.last;
    int rc;
    if (reuse) {
        a.seg_items = pf.items; a.seg_count = &pf.hdr->n_items;
        if (pf.max_items > 0) seg_bwd_from_fwd_kernel<<<dim3(pf.max_long), dim3(256), 0, s>>>(a, pf, p.T, p.out);
    } else {
        if (hipMemsetAsync(p.hdr, 0, sizeof(SegHeader), s) != hipSuccess) return check_launch("raster3d_bwd_seg memset");
        seg_plan_kernel<<<dim3((n_blocks + 255) / 256), dim3(256), 0, s>>>(isect_offsets, n_blocks, n_isects, seg_len, seg_cut, p);
        a.seg_items = p.items; a.seg_count = &p.hdr->n_items;
        if (p.max_items > 0) {
            a.seg_mode = 1; a.seg_grid = p.max_items;
            rc = raster3d_bwd_prepass_launch(a, s);
            if (rc != GSX_OK) return rc;
            seg_bwd_prefix_kernel<<<dim3(p.max_long), dim3(256), 0, s>>>(p);
        }
    }
    // the slices first, the short tiles behind them. The short-tile range must hold round8(n_blocks) workgroups whatever the
    // device-side item count is (xcd_remap is a bijection over round8(n_blocks) slots only)
    a.seg_mode = 2; a.seg_grid = p.max_items + ((n_blocks + 7u) / 8u) * 8u;
    if (seg_bwd_on_variant_w()) {
        // one wave per unit: a short tile of up to seg_cut entries started late is the launch's tail - take them longest-first
        // (the order lives behind the segment plan in the workspace, when the caller sized it with the backward's own function)
        unsigned char *ord = base + align256(seg_layout(n_isects, n_blocks, cdim, seg_len, base, nullptr));
        const int64_t left = workspace_bytes - (ord - reinterpret_cast<unsigned char *>(workspace));
        int orc = GSX_OK;
        a.tile_order = build_tile_order(isect_offsets, last_ids, n_images, tile_size, tile_w, tile_h, width, height, n_isects, ord,
                                        left, s, &orc);
        if (orc != GSX_OK) return orc;
    }
    rc = seg_bwd_on_variant_w() ? raster3d_bwd_w_launch_items(a, s) : raster3d_bwd_t_launch_items(a, s);
    if (rc != GSX_OK) return rc;
    return check_launch("raster3d_bwd_seg");
}

extern "C" int gsx_raster3d_bwd_seg(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas, uint32_t n_images, uint32_t n_isects,
    uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *v_rows,
    uint32_t row_stride, uint32_t seg_len, void *workspace, int64_t workspace_bytes, void *stream)
{
    return raster3d_bwd_seg_impl(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
                                 last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height, tile_size,
                                 tile_w, tile_h, v_rows, row_stride, seg_len, nullptr, 0, workspace, workspace_bytes, stream);
}

// gsx_raster3d_bwd_seg for a caller that still holds the workspace of the gsx_raster3d_fwd_seg call over the same lists (same
// seg_len, <= 4 channels = one channel chunk): no pre-pass (seg_bwd_from_fwd_kernel). NULL = gsx_raster3d_bwd_seg.
extern "C" int gsx_raster3d_bwd_seg_reuse(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas, uint32_t n_images, uint32_t n_isects,
    uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, float *v_rows,
    uint32_t row_stride, uint32_t seg_len, const void *fwd_workspace, int64_t fwd_workspace_bytes, void *workspace,
    int64_t workspace_bytes, void *stream)
{
    return raster3d_bwd_seg_impl(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
                                 last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height, tile_size,
                                 tile_w, tile_h, v_rows, row_stride, seg_len, fwd_workspace, fwd_workspace_bytes, workspace,
                                 workspace_bytes, stream);
}
