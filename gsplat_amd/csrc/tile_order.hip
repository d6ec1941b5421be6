// Longest tiles first: the workgroup -> tile order of the backward compositing launches (3DGS variants T / W, 2DGS).
// Not in the reference (its backward takes the tiles in launch order, RasterizeToPixels3DGSSerialBatchBwd.cu:41-110); a
// scheduling change only - the same tiles run, the gradients are accumulated with float atomics in either order.
#include <cstdlib>

#include "raster3d.hpp"

namespace gsx {

// What a tile costs the backward is the length of its list UP TO ITS LAST CONTRIBUTOR (early termination in the forward
// pass): on c3 that is 160 entries on average, 49 .. 486 per tile (the full lists: 466 +- 30). Workgroups are dispatched in
// index order onto a few slots per CU, so in launch order the kernel lasts as long as its unluckiest slot: list scheduling
// of the measured costs gives 1.23 x the ideal sum / slots at 5 workgroups per CU (variant T) and 1.66 x at 12 one-wave
// workgroups (variant W), against 1.06 x / 1.09 x longest-first (tools/tile_balance.py). Two small launches build the
// order: (1) one wave per tile takes the maximum of last_ids over its pixels -> cost; (2) one workgroup per XCD sorts the
// tiles of its range by cost / 4, descending, entirely in LDS (histogram, scan from the top, one slot per tile). The
// XCD-aware map is kept: workgroup b still runs on XCD b % 8 and XCD x still owns the contiguous tile range x, only the order
// INSIDE the range changes (neighbouring tiles share Gaussians: they stay in one XCD's L2). Ties inside a bucket are ordered by
// LDS atomics: scheduling only, the gradients are accumulated with float atomics in either case.
constexpr uint32_t kOrderBuckets = 1024; // cost / 4, saturating: lists of up to 4096 staged entries are told apart
struct TileOrderArgs {
    const int32_t *isect_offsets, *last_ids;
    uint32_t n_images, tile_w, tile_h, width, height, n_isects, n_blocks, per_xcd;
    int32_t *cost;  // [n_blocks]
    int32_t *order; // [n_blocks] in xcd_remap() index space
    void *zero_ptr; // (optional) the launch's gradient rows, zero-filled by the cost kernel: 16-byte aligned, zero_bytes % 4 == 0
    int64_t zero_bytes;
};
__global__ void __launch_bounds__(256) tile_order_cost_kernel(const TileOrderArgs a)
{
    if (a.zero_ptr) { // the gradient rows the compositing launch accumulates into: this kernel is short of memory work
        const int64_t n16 = a.zero_bytes >> 4;
        uint4 *z          = reinterpret_cast<uint4 *>(a.zero_ptr);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0 && (int64_t)threadIdx.x < ((a.zero_bytes & 15) >> 2))
            reinterpret_cast<float *>(a.zero_ptr)[n16 * 4 + threadIdx.x] = 0.0f;
    }
    const uint32_t lane = threadIdx.x & 63u, blk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (blk >= a.n_blocks) return;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image = blk / tiles_per_image, tile = blk % tiles_per_image;
    const uint32_t x0 = (tile % a.tile_w) * 16u, y0 = (tile / a.tile_w) * 16u;
    int32_t m = -1;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t p = lane + 64u * i, ox = x0 + (p & 15u), oy = y0 + (p >> 4);
        if (ox < a.width && oy < a.height) m = max(m, a.last_ids[((size_t)image * a.height + oy) * a.width + ox]);
    }
    m = wave_max_i32(m);
    if (lane == 0) {
        const int32_t start = a.isect_offsets[blk];
        const int32_t end   = (blk == a.n_blocks - 1) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
        a.cost[blk]         = max(0, min(end, m + 1) - start);
    }
}
// one workgroup per XCD range: bucket sort by cost / 4, most expensive first
__global__ void __launch_bounds__(1024) tile_order_sort_kernel(const TileOrderArgs a)
{
    __shared__ int32_t s_hist[kOrderBuckets];
    __shared__ int32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t first = blockIdx.x * a.per_xcd, last = min(first + a.per_xcd, a.n_blocks);
    s_hist[tid] = 0; // kOrderBuckets == blockDim.x
    __syncthreads();
    for (uint32_t blk = first + tid; blk < last; blk += 1024u)
        atomicAdd(&s_hist[min((uint32_t)a.cost[blk] >> 2, kOrderBuckets - 1u)], 1);
    __syncthreads();
    // exclusive scan from the top: thread t owns bucket 1023 - t
    const int32_t mine = s_hist[kOrderBuckets - 1u - tid];
    int32_t inc        = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t y = __shfl_up(inc, o);
        if ((int)lane >= o) inc += y;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    int32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    s_hist[kOrderBuckets - 1u - tid] = before + inc - mine; // first slot of the bucket
    __syncthreads();
    for (uint32_t blk = first + tid; blk < last; blk += 1024u) {
        const int32_t slot = atomicAdd(&s_hist[min((uint32_t)a.cost[blk] >> 2, kOrderBuckets - 1u)], 1);
        a.order[first + (uint32_t)slot] = (int32_t)blk;
    }
}

} // namespace gsx

namespace gsx {

// GSX_RASTER3D_BWD_ORDER: "0" / "launch" = workgroups in launch order (A/B); "force" = sort however few tiles there are (tests:
// small images then take the ordered path too); default: sort when there are more tiles than a round of workgroup slots
static int bwd_lpt_mode()
{
    static const int mode = [] {
        const char *e = getenv("GSX_RASTER3D_BWD_ORDER");
        if (e && (e[0] == '0' || e[0] == 'l')) return 0;
        if (e && e[0] == 'f') return 2;
        return 1;
    }();
    return mode;
}
// Builds the longest-first order in `ws` (>= gsx_raster3d_bwd_workspace_bytes) and returns the pointer for Raster3DArgs, or
// null when the launch keeps its launch order (no workspace, sparse layout, fewer tiles than workgroup slots, switched off).
const int32_t *build_tile_order(const int32_t *isect_offsets, const int32_t *last_ids, uint32_t n_images, uint32_t tile_size,
                                uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height, uint32_t n_isects, void *ws,
                                int64_t ws_bytes, hipStream_t stream, int *rc, void *zero_ptr, int64_t zero_bytes)
{
    *rc = GSX_OK;
    const uint32_t n_blocks = tile_w * tile_h * n_images;
    const int mode = bwd_lpt_mode();
    if (!ws || tile_size != 16 || mode == 0 || (mode == 1 && n_blocks < 2048u)) return nullptr;
    if (ws_bytes < tile_order_workspace_bytes(n_images, tile_w, tile_h)) return nullptr;
    TileOrderArgs o{};
    o.isect_offsets = isect_offsets; o.last_ids = last_ids; o.n_images = n_images; o.tile_w = tile_w; o.tile_h = tile_h;
    o.width = width; o.height = height; o.n_isects = n_isects; o.n_blocks = n_blocks; o.per_xcd = (n_blocks + 7u) / 8u;
    o.cost  = reinterpret_cast<int32_t *>(ws);
    o.order = o.cost + n_blocks;
    o.zero_ptr = zero_ptr; o.zero_bytes = zero_bytes; // the caller fills the rows itself when no order is built (null return)
    tile_order_cost_kernel<<<dim3((n_blocks + 3u) / 4u), dim3(256), 0, stream>>>(o);
    tile_order_sort_kernel<<<dim3((n_blocks + o.per_xcd - 1u) / o.per_xcd), dim3(1024), 0, stream>>>(o);
    *rc = check_launch("tile order");
    return *rc == GSX_OK ? o.order : nullptr;
}


int64_t tile_order_workspace_bytes(uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return (int64_t)sizeof(int32_t) * 2ll * n_images * tile_w * tile_h;
}

} // namespace gsx
