// Shared declarations of the fused tile-intersection path (isect_fused.hip: walk kernels, compiled with
// -ffp-contract=off; tile_sort.hip: column scan, per-tile sort and the C-ABI entries).
#pragma once
#include "common.hpp"
#include <cstdio>
#include <cstdlib>

namespace gsx {

struct SpanRecord;                     // isect_spans.hpp: 16 bytes per row
constexpr int64_t kSpanRecordBytes = 16;

struct FusedGeom {
    int64_t rows;           // all rows
    int64_t rows_per_image; // dense: N; packed (single image): rows
    uint32_t n_images, cpi /* chunks per image */, rpc /* rows per chunk */, n_chunks;
    uint32_t threads;       // workgroup size of the two walk kernels: 1024 (one per CU) or 512 (two per CU)
    uint32_t tile_size, tile_w, tile_h, n_tiles /* per image */;
};

struct FusedArgs {
    FusedGeom geom;
    const float *means2d;         // [R,2]
    const int32_t *radii;         // [R,2]
    const float *depths;          // [R]   (emit)
    const float *conics;          // [R,3] or null
    const float *opacities;       // [R]   or null
    const uint8_t *tile_mask;     // [n_images * n_tiles] or null: only tiles with a non-zero flag receive intersections
    int32_t *tiles_per_gauss;     // [R]   (count)
    int32_t *table;               // [n_chunks][n_tiles]: histogram, then exclusive prefix over an image's chunks
    SpanRecord *spans;            // [R] or null: what the counting pass's walk found, row by row (isect_spans.hpp: SpanPacker)
    const int32_t *isect_offsets; // [n_images * n_tiles] (emit)
    uint2 *bucketed;              // [M] (emit)
};

// Does the counting pass record the rows' spans for the emission (isect_fused.hip)? It pays on very large inputs only;
// GSX_FUSED_SPANS=1 / 0 forces it (tests, A/B).
inline bool fused_records_spans(int64_t rows)
{
    const char *e = getenv("GSX_FUSED_SPANS");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return rows > 6000000;
}

// workgroups of 1024 threads, one chunk of rows each: at least 4096 rows per chunk, at most 768 (256 for huge inputs) chunks
inline FusedGeom fused_geometry(int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h)
{
    FusedGeom g{};
    g.rows = rows; g.n_images = n_images ? n_images : 1;
    g.rows_per_image = rows / g.n_images;
    // measured at c3 (1M rows, 8160 tiles; count / column scan / emit+scatter in us): 256 threads x 512 rows: 56 / 153 /
    // 105; 512 x 1024: 46 / 75 / 91; 1024 x 2048: 48 / 29 / 95; 1024 x 4096: 50 / 16 / 102 -> the table (chunks x tiles)
    // must stay small, the walk does not care: at least 4096 rows per chunk
    // Chunks are what the 256 CUs schedule dynamically: a real scene's rows come spatially clustered (garden x25: chunks over
    // crowded regions run several times longer), so up to 768 chunks of >= 4096 rows while the [chunk][tile] table stays
    // small; very large inputs (c4: 16 M random rows) keep one chunk per CU - more only lengthens the table.
    // measured (count + emit+sort, ms): garden x25 256 chunks 0.423, 384 0.344, 768 0.310, 2048 0.310; c4 256 2.44, 768 2.60
    // A/B: GSX_FUSED_WG = "<threads>x<chunks>" (e.g. 512x512); 512-thread workgroups (two per CU) gained nothing
    int64_t kMaxChunks = rows <= 6000000 ? 768 : 256;
    // "the table must stay small": 768 chunks were measured at 8160 tiles (25 MB of table). A larger tile grid keeps the table
    // below that budget instead of growing it threefold (4K image, 32 k tiles, 3 M rows: 95 MB and a 3x longer column scan).
    {
        const int64_t n_tiles   = (int64_t)tile_w * tile_h > 0 ? (int64_t)tile_w * tile_h : 1;
        const int64_t by_budget = 768ll * 8192 / n_tiles; // [chunk][tile] cells of the measured configuration
        if (kMaxChunks > 256 && by_budget < kMaxChunks) kMaxChunks = by_budget < 256 ? 256 : by_budget;
    }
    g.threads = 1024;
    static const char *const wg_env = getenv("GSX_FUSED_WG");
    if (wg_env) {
        unsigned t = 0, c = 0;
        if (sscanf(wg_env, "%ux%u", &t, &c) == 2 && (t == 512 || t == 1024) && c >= 64 && c <= 2048) { g.threads = t; kMaxChunks = c; }
    }
    constexpr int64_t kMinRows = 4096;
    const int64_t max_cpi = kMaxChunks / g.n_images > 0 ? kMaxChunks / g.n_images : 1;
    int64_t cpi = (g.rows_per_image + kMinRows - 1) / kMinRows;
    // ... but a small input still wants a workgroup per CU: 100 k packed rows (a camera that sees a tenth of a 1 M scene) made
    // 25 chunks - 25 of 256 CUs walking, and the near, large Gaussians that survive such a cut are the expensive rows (packed
    // step 0.849 ms against 0.747 dense). Down to 256 rows per chunk the table stays at the c3 size or below.
    {
        const int64_t per_image = (256 + (int64_t)g.n_images - 1) / (int64_t)g.n_images;
        const int64_t floor_cpi = g.rows_per_image / 256 < per_image ? g.rows_per_image / 256 : per_image;
        if (cpi < floor_cpi) cpi = floor_cpi;
    }
    if (cpi < 1) cpi = 1;
    if (cpi > max_cpi) cpi = max_cpi;
    g.cpi = (uint32_t)cpi;
    g.rpc = (uint32_t)((g.rows_per_image + cpi - 1) / cpi);
    if (g.rpc == 0) g.rpc = 1;
    g.n_chunks = g.cpi * g.n_images;
    g.tile_size = tile_size; g.tile_w = tile_w; g.tile_h = tile_h; g.n_tiles = tile_w * tile_h;
    return g;
}

struct TileSortArgs {
    const uint64_t *keys_in;
    const int32_t *vals_in;
    int64_t n;
    uint32_t n_tiles, tile_bits, n_bins, n_chunks;
    int64_t chunk_len;
    int32_t *table;        // [n_bins][n_chunks] histogram, then exclusive scan (in place via table_scanned)
    int32_t *table_scanned;
    uint2 *bucketed;       // [n] (depth bits, flatten id) grouped by bin
    uint64_t *keys_out;
    int32_t *vals_out;
    uint2 *scratch;        // [n] ping-pong for oversized tiles
    int32_t *big_count;    // number of tiles longer than kCapSmall (filled by the MODE 0 launch)
    int32_t *big_list;     // [n_bins] their bin ids
    int32_t sort_int;      // GSX_ISECT_SORT=int: every list on the integer network (bitonic64.hpp)
};

int launch_fused_count_hist(const FusedArgs &a, hipStream_t s);
int launch_fused_emit_scatter(const FusedArgs &a, hipStream_t s);
int launch_colscan(int32_t *table, int32_t *totals, uint32_t n_cols, uint32_t cpi, uint32_t n_images, hipStream_t s);
int launch_big_tile_sort(const TileSortArgs &a, hipStream_t s); // tiles listed in a.big_list[0 .. *a.big_count)

} // namespace gsx
