// Gaussian -> tile intersection: count / emit / offsets (gfx950).
//
// Restates the behaviour of gsplat::intersect_tile + gsplat::intersect_offset
// (reference gsplat/cuda/csrc/IntersectTile.cu:83-207 ellipse helpers, :214-464 count+emit kernel,
// :925-988 offsets kernel; host gsplat/cuda/csrc/Intersect.cpp:170-329, :547).
//
// This file is compiled with -ffp-contract=off: every float operation below is a single IEEE
// operation (HIP's / and sqrtf are correctly rounded, det_logf uses explicit fmaf), and the CPU
// oracle (oracle/gsplat_oracle.c) performs the same operations in the same order, so the INTEGER
// outputs (tiles_per_gauss, isect_ids, flatten_ids, offsets) are bit-identical between the two.
#include "isect_walk.hpp"

namespace gsx {

struct IsectArgs {
    const float *means2d;      // [R,2]
    const int32_t *radii;      // [R,2]
    const float *depths;       // [R] (emit only)
    const float *conics;       // [R,3] or null
    const float *opacities;    // [R] or null
    const int64_t *image_ids;  // [R] or null (packed)
    const int64_t *cum;        // [R] inclusive cumsum (emit only)
    int64_t rows;
    uint32_t n_per_image;      // N (dense)
    uint32_t tile_size, tile_w, tile_h;
    uint32_t tile_n_bits;
    int32_t *tiles_per_gauss;  // count only
    int64_t *isect_ids;        // emit only
    int32_t *flatten_ids;      // emit only
};

template <bool EMIT>
__global__ void __launch_bounds__(256) isect_kernel(const IsectArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.rows) return;
    const float rx = (float)a.radii[2 * idx], ry = (float)a.radii[2 * idx + 1];
    if (rx <= 0.0f || ry <= 0.0f) {
        if (!EMIT) a.tiles_per_gauss[idx] = 0;
        return;
    }
    const float mx = a.means2d[2 * idx], my = a.means2d[2 * idx + 1];
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    float A = 0.f, B = 0.f, C = 0.f, op = 0.f;
    if (has_conic) {
        A  = a.conics[3 * idx];
        B  = a.conics[3 * idx + 1];
        C  = a.conics[3 * idx + 2];
        op = a.opacities[idx];
    }
    if (!EMIT) {
        a.tiles_per_gauss[idx] =
            walk_tiles(mx, my, rx, ry, has_conic, A, B, C, op, a.tile_size, a.tile_w, a.tile_h, [](int64_t) {});
    } else {
        const int64_t iid   = a.image_ids ? a.image_ids[idx] : idx / a.n_per_image;
        const uint64_t hi   = (uint64_t)iid << (32 + a.tile_n_bits);
        const uint64_t dkey = (uint64_t)__float_as_uint(a.depths[idx]);
        int64_t cur         = idx == 0 ? 0 : a.cum[idx - 1];
        walk_tiles(mx, my, rx, ry, has_conic, A, B, C, op, a.tile_size, a.tile_w, a.tile_h, [&](int64_t tile) {
            a.isect_ids[cur]   = (int64_t)(hi | ((uint64_t)tile << 32) | dkey);
            a.flatten_ids[cur] = (int32_t)idx;
            ++cur;
        });
    }
}

// float64 rows: radius boxes only (the exact ellipse test stays fp32), keys carry the depth narrowed to float32 - the 32 key
// bits must stay a monotonic function of the depth (a bare reinterpretation of half a double is not; test_basic.py:1282-1287)
template <bool EMIT>
__global__ void __launch_bounds__(256) isect_f64_kernel(const double *means2d, const int32_t *radii, const double *depths,
                                                        const int64_t *image_ids, const int64_t *cum, int64_t rows,
                                                        uint32_t n_per_image, uint32_t tile_size, uint32_t tile_w,
                                                        uint32_t tile_h, uint32_t tile_n_bits, int32_t *tiles_per_gauss,
                                                        int64_t *isect_ids, int32_t *flatten_ids)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows) return;
    const double rx = (double)radii[2 * idx], ry = (double)radii[2 * idx + 1];
    if (rx <= 0.0 || ry <= 0.0) {
        if (!EMIT) tiles_per_gauss[idx] = 0;
        return;
    }
    const double mx = means2d[2 * idx], my = means2d[2 * idx + 1];
    if (!EMIT) {
        tiles_per_gauss[idx] = walk_tiles_aabb_f64(mx, my, rx, ry, tile_size, tile_w, tile_h, [](int64_t) {});
    } else {
        const int64_t iid   = image_ids ? image_ids[idx] : idx / n_per_image;
        const uint64_t hi   = (uint64_t)iid << (32 + tile_n_bits);
        const uint64_t dkey = (uint64_t)__float_as_uint((float)depths[idx]);
        int64_t cur         = idx == 0 ? 0 : cum[idx - 1];
        walk_tiles_aabb_f64(mx, my, rx, ry, tile_size, tile_w, tile_h, [&](int64_t tile) {
            isect_ids[cur]   = (int64_t)(hi | ((uint64_t)tile << 32) | dkey);
            flatten_ids[cur] = (int32_t)idx;
            ++cur;
        });
    }
}

// offsets[k] = first index of the sorted list whose (image, tile) >= k.
__global__ void __launch_bounds__(256) isect_offsets_kernel(
    const int64_t *sorted_ids, int64_t n_isects, uint32_t n_tiles, uint32_t tile_n_bits, int64_t total_tiles,
    int32_t *offsets)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_isects) return;
    const uint64_t tmask = (tile_n_bits >= 32) ? 0xFFFFFFFFull : ((1ull << tile_n_bits) - 1ull);
    auto linear = [&](int64_t key) -> int64_t {
        const uint64_t k = (uint64_t)key >> 32;
        return (int64_t)(k >> tile_n_bits) * n_tiles + (int64_t)(k & tmask);
    };
    const int64_t cur  = linear(sorted_ids[i]);
    const int64_t prev = i == 0 ? -1 : linear(sorted_ids[i - 1]);
    for (int64_t k = prev + 1; k <= cur; ++k) offsets[k] = (int32_t)i;
    if (i == n_isects - 1)
        for (int64_t k = cur + 1; k < total_tiles; ++k) offsets[k] = (int32_t)n_isects;
}

static uint32_t bits_for_count(uint64_t count)
{ // bits to index 0..count-1 (0 when count <= 1)  — reference MathUtils.h:25-35
    uint32_t b = 0;
    if (count <= 1) return 0;
    uint64_t v = count - 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

} // namespace gsx

extern "C" int gsx_isect_count(const float *means2d, const int32_t *radii, const float *conics, const float *opacities,
                               const int64_t *image_ids, int64_t rows, uint32_t n_per_image, uint32_t n_images,
                               uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *tiles_per_gauss,
                               void *stream)
{
    using namespace gsx;
    (void)n_images;
    GSX_REQUIRE(rows >= 0, "gsx_isect_count: negative rows");
    GSX_REQUIRE(tile_size > 0, "gsx_isect_count: tile_size must be positive");
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means2d && radii && tiles_per_gauss, "gsx_isect_count: null pointer");
    IsectArgs a{};
    a.means2d = means2d; a.radii = radii; a.conics = conics; a.opacities = opacities; a.image_ids = image_ids;
    a.rows = rows; a.n_per_image = n_per_image ? n_per_image : 1; a.tile_size = tile_size; a.tile_w = tile_w;
    a.tile_h = tile_h; a.tiles_per_gauss = tiles_per_gauss;
    isect_kernel<false><<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("isect_count");
}

extern "C" int gsx_isect_count_f64(const double *means2d, const int32_t *radii, const int64_t *image_ids, int64_t rows,
                                   uint32_t n_per_image, uint32_t n_images, uint32_t tile_size, uint32_t tile_w,
                                   uint32_t tile_h, int32_t *tiles_per_gauss, void *stream)
{
    using namespace gsx;
    (void)n_images;
    GSX_REQUIRE(rows >= 0, "gsx_isect_count_f64: negative rows");
    GSX_REQUIRE(tile_size > 0, "gsx_isect_count_f64: tile_size must be positive");
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means2d && radii && tiles_per_gauss, "gsx_isect_count_f64: null pointer");
    isect_f64_kernel<false><<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        means2d, radii, nullptr, image_ids, nullptr, rows, n_per_image ? n_per_image : 1, tile_size, tile_w, tile_h, 0,
        tiles_per_gauss, nullptr, nullptr);
    return check_launch("isect_count_f64");
}

extern "C" int gsx_isect_emit_f64(const double *means2d, const int32_t *radii, const double *depths, const int64_t *image_ids,
                                  const int64_t *cum_tiles_per_gauss, int64_t rows, uint32_t n_per_image, uint32_t n_images,
                                  uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int64_t *isect_ids,
                                  int32_t *flatten_ids, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(rows >= 0, "gsx_isect_emit_f64: negative rows");
    GSX_REQUIRE(tile_size > 0, "gsx_isect_emit_f64: tile_size must be positive");
    const uint32_t tile_bits = bits_for_count((uint64_t)tile_w * tile_h), image_bits = bits_for_count(n_images);
    if (tile_bits + image_bits > 32) {
        set_last_error("gsx_isect_emit_f64: tile bits (%u) + image bits (%u) exceed the 32 key bits above the depth",
                       tile_bits, image_bits);
        return GSX_ERR_OVERFLOW;
    }
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means2d && radii && depths && cum_tiles_per_gauss, "gsx_isect_emit_f64: null pointer");
    isect_f64_kernel<true><<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        means2d, radii, depths, image_ids, cum_tiles_per_gauss, rows, n_per_image ? n_per_image : 1, tile_size, tile_w, tile_h,
        tile_bits, nullptr, isect_ids, flatten_ids);
    return check_launch("isect_emit_f64");
}

extern "C" int gsx_isect_emit(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                              const float *opacities, const int64_t *image_ids, const int64_t *cum_tiles_per_gauss,
                              int64_t rows, uint32_t n_per_image, uint32_t n_images, uint32_t tile_size,
                              uint32_t tile_w, uint32_t tile_h, int64_t *isect_ids, int32_t *flatten_ids, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(rows >= 0, "gsx_isect_emit: negative rows");
    GSX_REQUIRE(tile_size > 0, "gsx_isect_emit: tile_size must be positive");
    const uint32_t tile_bits = bits_for_count((uint64_t)tile_w * tile_h), image_bits = bits_for_count(n_images);
    if (tile_bits + image_bits > 32) {
        set_last_error("gsx_isect_emit: tile bits (%u) + image bits (%u) exceed the 32 key bits above the depth",
                       tile_bits, image_bits);
        return GSX_ERR_OVERFLOW;
    }
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means2d && radii && depths && cum_tiles_per_gauss, "gsx_isect_emit: null pointer");
    IsectArgs a{};
    a.means2d = means2d; a.radii = radii; a.depths = depths; a.conics = conics; a.opacities = opacities;
    a.image_ids = image_ids; a.cum = cum_tiles_per_gauss; a.rows = rows; a.n_per_image = n_per_image ? n_per_image : 1;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.tile_n_bits = tile_bits;
    a.isect_ids = isect_ids; a.flatten_ids = flatten_ids;
    isect_kernel<true><<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("isect_emit");
}

extern "C" int gsx_isect_offsets(const int64_t *isect_ids_sorted, int64_t n_isects, uint32_t n_images, uint32_t tile_w,
                                 uint32_t tile_h, int32_t *offsets, void *stream)
{
    using namespace gsx;
    const int64_t n_tiles = (int64_t)tile_w * tile_h, total = n_tiles * n_images;
    if (total == 0) return GSX_OK;
    GSX_REQUIRE(offsets != nullptr, "gsx_isect_offsets: null offsets");
    GSX_REQUIRE(n_isects >= 0 && n_isects < (1ll << 31), "gsx_isect_offsets: n_isects out of range");
    hipStream_t s = (hipStream_t)stream;
    if (n_isects == 0) {
        if (hipMemsetAsync(offsets, 0, total * sizeof(int32_t), s) != hipSuccess) {
            set_last_error("gsx_isect_offsets: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    GSX_REQUIRE(isect_ids_sorted != nullptr, "gsx_isect_offsets: null ids");
    isect_offsets_kernel<<<dim3((uint32_t)ceil_div(n_isects, 256)), dim3(256), 0, s>>>(
        isect_ids_sorted, n_isects, (uint32_t)n_tiles, bits_for_count((uint64_t)n_tiles), total, offsets);
    return check_launch("isect_offsets");
}
