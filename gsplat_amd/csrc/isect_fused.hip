// Fused tile intersection: the per-(chunk, tile) histogram is taken WHILE counting, and the emission pass writes every
// (depth, row) pair straight into its tile's segment — the unsorted (key, value) arrays of gsx_isect_emit, the
// histogram pass and the scatter pass of gsx_isect_tile_sort are never materialised.
//
//   count+hist   gsx_isect_fused_count      walk the tiles of every row (isect_walk.hpp) -> tiles_per_gauss[row] and an
//                                           LDS histogram of tile ids per CHUNK of rows -> table[chunk][tile]
//   column scan  (tile_sort.hip)            per (image, tile): running sum over that image's chunks (exclusive, in place)
//                                           + tile totals -> exclusive scan -> isect_offsets (the reference's
//                                           intersect_offset output, for free) and n_isects
//   emit+scatter gsx_isect_fused_emit_sort  same walk; slot = LDS cursor[tile]++ (cursor = offset + chunk prefix);
//                                           bucketed[slot] = (depth bits, row)
//   tile sort    (tile_sort.hip)            per-tile LDS sort by (depth, row), unchanged
//
// Result: identical to gsx_isect_count/emit + gsx_isect_tile_sort (= a stable sort on the full key, ties in ascending
// row order), because the per-tile sort does not depend on the arrival order inside a segment. HBM traffic per
// intersection drops from 12 (emit) + 8 (hist) + 12 + 8 (scatter) to 8 bytes before the per-tile sort.
//
// Chunks are image-aligned (dense rows: N per image; packed rows are supported for a single image), so a chunk's
// histogram only needs the tiles of ONE image: the table is [n_chunks][tiles per image].
// Compiled with -ffp-contract=off (the walk must be bit-exact with the oracle; see isect_walk.hpp).
#include "isect_walk.hpp"
#include "isect_fused.hpp"

namespace gsx {

constexpr int kFusedThreads = 1024;

__device__ __forceinline__ void chunk_rows(const FusedGeom &g, uint32_t chunk, int64_t &lo, int64_t &hi)
{
    const uint32_t img = chunk / g.cpi, sub = chunk % g.cpi;
    lo = (int64_t)img * g.rows_per_image + (int64_t)sub * g.rpc;
    hi = min(lo + (int64_t)g.rpc, (int64_t)(img + 1) * g.rows_per_image);
}

struct RowGeom {
    bool live;
    float mx, my, rx, ry, A, B, C, op;
};
__device__ __forceinline__ RowGeom load_row_geom(const FusedArgs &a, int64_t r, bool has_conic)
{
    RowGeom q;
    q.rx = (float)a.radii[2 * r];
    q.ry = (float)a.radii[2 * r + 1];
    q.live = q.rx > 0.0f && q.ry > 0.0f;
    q.mx = q.my = q.A = q.B = q.C = q.op = 0.0f;
    if (q.live) {
        q.mx = a.means2d[2 * r];
        q.my = a.means2d[2 * r + 1];
        if (has_conic) {
            q.A  = a.conics[3 * r];
            q.B  = a.conics[3 * r + 1];
            q.C  = a.conics[3 * r + 2];
            q.op = a.opacities[r];
        }
    }
    return q;
}

__global__ void __launch_bounds__(kFusedThreads) fused_count_hist_kernel(const FusedArgs a)
{
    extern __shared__ int32_t s_hist[];
    const FusedGeom &g = a.geom;
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) s_hist[t] = 0;
    __syncthreads();
    int64_t lo, hi;
    chunk_rows(g, blockIdx.x, lo, hi);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)(blockIdx.x / g.cpi) * g.n_tiles : nullptr;
    for (int64_t r = lo + threadIdx.x; r < hi; r += kFusedThreads) {
        const RowGeom q = load_row_geom(a, r, has_conic);
        int32_t n       = 0;
        if (q.live)
            n = walk_tiles(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h,
                           [&](int64_t tile) {
                               if (!tmask || tmask[tile]) atomicAdd(&s_hist[tile], 1);
                           });
        if (a.tiles_per_gauss) a.tiles_per_gauss[r] = n;
    }
    __syncthreads();
    int32_t *out = a.table + (int64_t)blockIdx.x * g.n_tiles;
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) out[t] = s_hist[t];
}

__global__ void __launch_bounds__(kFusedThreads) fused_emit_scatter_kernel(const FusedArgs a)
{
    extern __shared__ int32_t s_cur[];
    const FusedGeom &g = a.geom;
    const uint32_t img = blockIdx.x / g.cpi;
    const int32_t *pre = a.table + (int64_t)blockIdx.x * g.n_tiles;      // exclusive prefix over this image's chunks
    const int32_t *off = a.isect_offsets + (int64_t)img * g.n_tiles;     // start of every (image, tile) segment
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) s_cur[t] = off[t] + pre[t];
    __syncthreads();
    int64_t lo, hi;
    chunk_rows(g, blockIdx.x, lo, hi);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)img * g.n_tiles : nullptr;
    for (int64_t r = lo + threadIdx.x; r < hi; r += kFusedThreads) {
        const RowGeom q = load_row_geom(a, r, has_conic);
        if (!q.live) continue;
        const uint32_t dbits = __float_as_uint(a.depths[r]);
        walk_tiles(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h,
                   [&](int64_t tile) {
                       if (tmask && !tmask[tile]) return;
                       const int32_t slot = atomicAdd(&s_cur[tile], 1);
                       a.bucketed[slot]   = make_uint2(dbits, (uint32_t)r);
                   });
    }
}

static void set_lds_limit_once()
{
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void *)fused_count_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute((const void *)fused_emit_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    }
}

int launch_fused_count_hist(const FusedArgs &a, hipStream_t s)
{
    set_lds_limit_once();
    fused_count_hist_kernel<<<dim3(a.geom.n_chunks), dim3(kFusedThreads), (size_t)a.geom.n_tiles * sizeof(int32_t), s>>>(a);
    return check_launch("isect_fused_count");
}

int launch_fused_emit_scatter(const FusedArgs &a, hipStream_t s)
{
    set_lds_limit_once();
    fused_emit_scatter_kernel<<<dim3(a.geom.n_chunks), dim3(kFusedThreads), (size_t)a.geom.n_tiles * sizeof(int32_t), s>>>(a);
    return check_launch("isect_fused_emit");
}

} // namespace gsx
