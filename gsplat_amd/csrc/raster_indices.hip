// rasterize_to_indices (3DGS and 2DGS): the (gaussian, pixel) pairs that contribute to each pixel inside a window of
// the tile lists — the reference's debugging / PyTorch-rasterizer support op.
// C-ABI entry: gsx_raster_indices. Replaces gsplat::rasterize_to_indices_3dgs / _2dgs (ext.cpp:1105-1109, 1200-1204;
// kernels RasterizeToIndices3DGSSerialBatch.cu:38-193, RasterizeToIndices2DGSSerialBatch.cu).
// Two-pass protocol driven by the caller: pass 1 (chunk_starts == NULL) writes the per-pixel counts, the caller takes an
// exclusive cumsum, pass 2 writes gaussian ids and the combined pixel id (pixel + image * H * W) at those offsets.
// `range_start` / `range_end` count batches of tile_size^2 list entries, as in the reference. Not a hot path: one thread
// per pixel, parameters read straight from global memory.
//
// The alpha of a pair is evaluated with the SAME staged forms as the compositing kernels (raster3d.hpp: stage_gaussian_e /
// staged_e about the tile centre; raster2d.hpp: stage_surfel / eval_surfel) and the transmittance advances with the same
// fma, so that the set of pairs this op reports IS the set the rasterizer blends - the contract of the reference, whose
// two kernels share one device function (RasterizeToPixels3DGSDevice.cuh). The reference's own PyTorch rasterizer
// (_torch_impl.py:_rasterize_to_pixels) takes its pairs from this op and is compared with the rasterizer at 1e-5: an
// alpha test decided by two different roundings flips a pair per ~10^5 pixels (reference suite, round 6).
#include "raster2d.hpp"

namespace gsx {

struct IndicesArgs {
    int mode; // 0: 3DGS (geom = conics [R,3]), 1: 2DGS (geom = ray_transforms [R,9])
    uint32_t range_start, range_end;
    const float *transmittances, *means2d, *geom, *opacities;
    const int32_t *isect_offsets, *flatten_ids;
    uint32_t n_images, n_per_image, n_isects, width, height, tile_size, tile_w, tile_h;
    const int32_t *chunk_starts;
    int32_t *chunk_cnts;
    int64_t *gaussian_ids, *pixel_ids;
};

__global__ void __launch_bounds__(256) raster_indices_kernel(const IndicesArgs a)
{
    const int64_t pix_per_image = (int64_t)a.width * a.height;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pix_per_image * a.n_images) return;
    const uint32_t image_id = (uint32_t)(gid / pix_per_image);
    const int64_t pix_id    = gid % pix_per_image;
    const uint32_t oy = (uint32_t)(pix_id / a.width), ox = (uint32_t)(pix_id % a.width);
    const uint32_t tile_id = (oy / a.tile_size) * a.tile_w + (ox / a.tile_size);
    const size_t blk = (size_t)image_id * a.tile_w * a.tile_h + tile_id;
    const int32_t start = a.isect_offsets[blk];
    const int32_t end   = (blk + 1 == (size_t)a.n_images * a.tile_w * a.tile_h) ? (int32_t)a.n_isects : a.isect_offsets[blk + 1];
    const int64_t bs    = (int64_t)a.tile_size * a.tile_size;
    const int64_t lo    = start + bs * a.range_start;
    const int64_t hi    = min((int64_t)end, start + bs * (int64_t)a.range_end);
    const bool first    = a.chunk_starts == nullptr;
    const uint32_t tile_x = ox / a.tile_size, tile_y = oy / a.tile_size;
    const float half = 0.5f * (float)a.tile_size; // tile centre and this pixel's centre relative to it, as the compositing kernels form them
    const float tcx = (float)(tile_x * a.tile_size) + half, tcy = (float)(tile_y * a.tile_size) + half;
    const float u = (float)(ox - tile_x * a.tile_size) + 0.5f - half, v = (float)(oy - tile_y * a.tile_size) + 0.5f - half;
    float T = a.transmittances[gid];
    int32_t cnt = 0;
    const int64_t base = first ? 0 : a.chunk_starts[gid];
    for (int64_t idx = lo; idx < hi; ++idx) {
        const int32_t g = a.flatten_ids[idx];
        float alpha;
        bool valid;
        const float mx = a.means2d[2 * (size_t)g], my = a.means2d[2 * (size_t)g + 1];
        if (a.mode == 0) {
            const float *c = a.geom + 3 * (size_t)g;
            v4f p0;
            float nA, nB, nC;
            stage_gaussian_f(mx - tcx, my - tcy, a.opacities[g], c[0], c[1], c[2], p0, nA, nB, nC);
            const float e = staged_f(p0, nA, nB, nC, u, v);
            alpha = fminf(kMaxAlpha, __builtin_amdgcn_exp2f(e));
            valid = !(e > p0.w) && !(alpha < kAlphaThreshold); // e > lo (+ margin) <=> sigma < 0
        } else {
            float4 A, B, C;
            stage_surfel(a.geom + 9 * (size_t)g, mx, my, a.opacities[g], tcx, tcy, A, B, C);
            const Surfel sf = eval_surfel(A, B, C, u, v);
            alpha = sf.alpha;
            valid = sf.valid;
        }
        if (!valid) continue;
        const float next_T = fmaf(-T, alpha, T); // the compositing kernels' update
        if (next_T <= kTransmittanceThresh) break; // exclusive stop
        if (!first) {
            a.gaussian_ids[base + cnt] = (int64_t)(g % (int32_t)a.n_per_image);
            a.pixel_ids[base + cnt]    = gid;
        }
        ++cnt;
        T = next_T;
    }
    if (first) a.chunk_cnts[gid] = cnt;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_raster_indices(int mode, uint32_t range_start, uint32_t range_end, const float *transmittances,
                                  const float *means2d, const float *geom, const float *opacities,
                                  const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                  uint32_t n_per_image, uint32_t n_isects, uint32_t width, uint32_t height,
                                  uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, const int32_t *chunk_starts,
                                  int32_t *chunk_cnts, int64_t *gaussian_ids, int64_t *pixel_ids, void *stream)
{
    const int64_t total = (int64_t)n_images * width * height;
    if (total == 0) return GSX_OK;
    GSX_REQUIRE(mode == 0 || mode == 1, "gsx_raster_indices: mode must be 0 (3DGS) or 1 (2DGS)");
    GSX_REQUIRE(tile_size > 0 && n_per_image > 0, "gsx_raster_indices: bad tile_size / n_per_image");
    GSX_REQUIRE(transmittances && isect_offsets, "gsx_raster_indices: null input");
    GSX_REQUIRE(n_isects == 0 || (means2d && geom && opacities && flatten_ids), "gsx_raster_indices: null input");
    GSX_REQUIRE(chunk_starts ? (gaussian_ids && pixel_ids) || true : chunk_cnts != nullptr, "gsx_raster_indices: null output");
    IndicesArgs a{};
    a.mode = mode; a.range_start = range_start; a.range_end = range_end; a.transmittances = transmittances;
    a.means2d = means2d; a.geom = geom; a.opacities = opacities; a.isect_offsets = isect_offsets;
    a.flatten_ids = flatten_ids; a.n_images = n_images; a.n_per_image = n_per_image; a.n_isects = n_isects;
    a.width = width; a.height = height; a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h;
    a.chunk_starts = chunk_starts; a.chunk_cnts = chunk_cnts; a.gaussian_ids = gaussian_ids; a.pixel_ids = pixel_ids;
    raster_indices_kernel<<<dim3((uint32_t)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("raster_indices");
}
