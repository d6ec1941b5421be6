// Query rasterizers of the 3DGS compositing pass (dense tile layout): which Gaussians contribute
// to a pixel, and with which radiance weight alpha_i * T_i. No colours, no gradients.
//
// C-ABI entries (one kernel, three accumulators):
//   gsx_raster3d_num_contributing  gsplat::rasterize_num_contributing_gaussians  (ext.cpp:1111; RasterizeNumContributingGaussians.cu)
//   gsx_raster3d_contributing_ids  gsplat::rasterize_contributing_gaussian_ids   (ext.cpp:1120; RasterizeContributingGaussianIds.cu:40-98)
//   gsx_raster3d_top_contributing  gsplat::rasterize_top_contributing_gaussian_ids (ext.cpp:1131; RasterizeTopContributingGaussianIds.cu:22-170)
// Shared walk = reference RasterizeContributingCommon.cuh:28-198: front to back over the tile's sorted list; a Gaussian
// contributes to a pixel when sigma >= 0 and alpha = min(0.99, opac exp(-sigma)) >= 1/255; the pixel stops
// (exclusive) when T (1 - alpha) <= 1e-4. Ids are local Gaussian ids (row % N for dense rows, the row itself when
// packed). Work decomposition and wave-level culling are those of raster3d_fwd.hip.
#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

struct QueryArgs {
    uint32_t n_images, n_isects, width, height, tile_size, tile_w, tile_h;
    uint32_t n_per_image; // 0 = packed rows (ids are the rows themselves)
    uint32_t K;           // ids: slots per pixel (max count); top: number of samples
    const float *means2d, *conics, *opacities;
    const int32_t *isect_offsets, *flatten_ids;
    // sparse pixel set (raster3d.hpp TileCtx / pixel_row); all null / 0 for the dense layout. Outputs are then rows [P, ...]
    const int32_t *sp_active_tiles;
    const uint64_t *sp_pixel_mask;
    const int64_t *sp_pixel_cumsum, *sp_pixel_map;
    uint32_t n_active, sp_words;
    int32_t *counts; // mode 0: [I,H,W]
    float *alphas;   // mode 0: [I,H,W]
    int32_t *ids;    // mode 1/2: [I,H,W,K]  (mode 1: pre-filled with -1 by the caller)
    float *weights;  // mode 1/2: [I,H,W,K]  (mode 1: pre-filled with 0)
    unsigned long long *stats; // mode 3: [4] work counters, added to with atomics (see gsx_raster3d_pair_stats)
};

constexpr int kQBatch = 256;
enum { kQCount = 0, kQIds = 1, kQTop = 2, kQStats = 3 };

template <int MODE>
__global__ void __launch_bounds__(256) raster3d_query_kernel(const QueryArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ga   = reinterpret_cast<float4 *>(smem_raw);        // x, y, log2(opac), A
    float4 *s_cull = s_ga + kQBatch;
    float2 *s_gb   = reinterpret_cast<float2 *>(s_cull + kQBatch); // B, C
    int32_t *s_id  = reinterpret_cast<int32_t *>(s_gb + kQBatch);  // local Gaussian id
    // MODE == kQTop: per-pixel top-K scratch, [K][blockDim] so that a lane's slots sit in its own bank column
    float *s_w     = reinterpret_cast<float *>(s_id + kQBatch);
    uint32_t *s_d  = reinterpret_cast<uint32_t *>(s_w + (MODE == kQTop ? a.K * blockDim.x : 0));
    int32_t *s_i   = reinterpret_cast<int32_t *>(s_d + (MODE == kQTop ? a.K * blockDim.x : 0));

    TileCtx tc;
    if (!tile_context(a, blockIdx.x, tc)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, nthr = blockDim.x;
    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
    const bool inside  = prow >= 0;
    // the tile-centre frame of the compositing kernels: the alpha test is theirs, bit for bit (raster3d.hpp: staged_e_offset)
    const float half_q = 0.5f * (float)a.tile_size;
    const float tcx = (float)(tc.tile_x * a.tile_size) + half_q, tcy = (float)(tc.tile_y * a.tile_size) + half_q;
    const float px = (float)lx + 0.5f - half_q, py = (float)ly + 0.5f - half_q; // pixel centre - tile centre
    const size_t pix = inside ? (size_t)prow : 0;

    const int32_t range_start = tc.range_start, range_end = tc.range_end;
    const int32_t n_batches   = (range_end - range_start + kQBatch - 1) / kQBatch;

    float T        = 1.0f;
    uint32_t count = 0; // contributions so far (= depth index of the next one)
    uint32_t walked = 0, wave_evals = 0, lane_evals = 0, empty_evals = 0; // kQStats: see gsx_raster3d_pair_stats
    bool done      = !inside;
    if constexpr (MODE == kQTop) {
        for (uint32_t k = 0; k < a.K; ++k) {
            s_w[k * nthr + tid] = 0.0f;
            s_d[k * nthr + tid] = 0xFFFFFFFFu;
            s_i[k * nthr + tid] = -1;
        }
    }
    const WaveRect rect = wave_pixel_rect(inside, px, py);

    for (int32_t b = 0; b < n_batches; ++b) {
        if (__syncthreads_count(done) == (int)nthr) break;
        const int32_t batch_start = range_start + kQBatch * b;
        for (int s = (int)tid; s < kQBatch; s += (int)nthr) {
            const int32_t idx = batch_start + s;
            if (idx < range_end) {
                const int32_t g  = a.flatten_ids[idx];
                const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                float4 ga;
                float2 gb;
                stage_gaussian(xy.x - tcx, xy.y - tcy, opac, ca, cb, cc, ga, gb);
                s_ga[s] = ga;
                s_gb[s] = gb;
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]       = make_float4(ga.x, ga.y, he.x, he.y);
                s_id[s]         = a.n_per_image ? (int32_t)((uint32_t)g % a.n_per_image) : g;
            }
        }
        __syncthreads();
        const int32_t batch_size = min(kQBatch, range_end - batch_start);
        for (int32_t j = 0; j < batch_size; j += 64) {
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
                const float2 gb = s_gb[t];
                const float dx = ga.x - px, dy = ga.y - py;
                const float e     = staged_e_offset(ga, gb, dx, dy);
                const float alpha = fminf(kMaxAlpha, __builtin_amdgcn_exp2f(e));
                const bool neg    = e > ga.z; // sigma < 0
                if constexpr (MODE == kQStats) {
                    ++wave_evals;
                    lane_evals += done ? 0u : 1u;
                    empty_evals += __builtin_amdgcn_ballot_w64(inside && !neg && !(alpha < kAlphaThreshold)) == 0ull ? 1u : 0u;
                }
                if (done || neg || alpha < kAlphaThreshold) continue;
                const float next_T = fmaf(-T, alpha, T); // the compositing kernels' update
                if (next_T <= kTransmittanceThresh) {
                    done = true;
                    if constexpr (MODE == kQStats) walked = (uint32_t)(batch_start + t - range_start + 1);
                    continue;
                }
                const float w = alpha * T;
                if constexpr (MODE == kQIds) {
                    if (count < a.K) {
                        a.ids[pix * a.K + count]     = s_id[t];
                        a.weights[pix * a.K + count] = w;
                    }
                } else if constexpr (MODE == kQTop) {
                    // replace the currently weakest sample (first one on ties) if this one is strictly stronger
                    uint32_t kmin = 0;
                    float wmin    = s_w[tid];
                    for (uint32_t k = 1; k < a.K; ++k) {
                        const float wk = s_w[k * nthr + tid];
                        if (wk < wmin) {
                            wmin = wk;
                            kmin = k;
                        }
                    }
                    if (w > wmin) {
                        s_w[kmin * nthr + tid] = w;
                        s_d[kmin * nthr + tid] = count;
                        s_i[kmin * nthr + tid] = s_id[t];
                    }
                }
                ++count;
                T = next_T;
            }
        }
    }
    if constexpr (MODE == kQStats) {
        // [0] (pixel, Gaussian) pairs a per-pixel serial walk of the tile lists evaluates (every list entry up to and including
        //     the one that saturates the pixel; no culling) - the reference kernel's work, SURVEY.md 8(d) "pairs"
        // [1] lane evaluations of this backend: 64 x (wave, Gaussian) pairs that survive the wave-level culling
        // [2] of those, lanes whose pixel was still open   [3] contributing pairs (alpha >= 1/255, before saturation)
        if (inside && !done) walked = (uint32_t)(range_end - range_start);
        unsigned long long v0 = inside ? walked : 0u, v2 = lane_evals, v3 = inside ? count : 0u;
        for (int o = 32; o >= 1; o >>= 1) {
            v0 += __shfl_xor(v0, o);
            v2 += __shfl_xor(v2, o);
            v3 += __shfl_xor(v3, o);
        }
        if (lane == 0) {
            atomicAdd(&a.stats[0], v0);
            atomicAdd(&a.stats[1], 64ull * wave_evals);
            atomicAdd(&a.stats[2], v2);
            atomicAdd(&a.stats[3], v3);
            atomicAdd(&a.stats[4], 64ull * empty_evals);
        }
        return;
    }
    if (!inside) return;
    if constexpr (MODE == kQCount) {
        a.counts[pix] = (int32_t)count;
        a.alphas[pix] = 1.0f - T;
    } else if constexpr (MODE == kQTop) {
        // back into front-to-back order: insertion sort by depth index (unused slots carry 0xFFFFFFFF and stay last)
        for (uint32_t i = 1; i < a.K; ++i) {
            const uint32_t kd = s_d[i * nthr + tid];
            const float kw    = s_w[i * nthr + tid];
            const int32_t ki  = s_i[i * nthr + tid];
            int32_t jj        = (int32_t)i - 1;
            while (jj >= 0 && s_d[jj * nthr + tid] > kd) {
                s_d[(jj + 1) * nthr + tid] = s_d[jj * nthr + tid];
                s_w[(jj + 1) * nthr + tid] = s_w[jj * nthr + tid];
                s_i[(jj + 1) * nthr + tid] = s_i[jj * nthr + tid];
                --jj;
            }
            s_d[(jj + 1) * nthr + tid] = kd;
            s_w[(jj + 1) * nthr + tid] = kw;
            s_i[(jj + 1) * nthr + tid] = ki;
        }
        for (uint32_t k = 0; k < a.K; ++k) {
            a.ids[pix * a.K + k]     = s_i[k * nthr + tid];
            a.weights[pix * a.K + k] = s_w[k * nthr + tid];
        }
    }
}

template <int MODE>
static int launch_query(const QueryArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid  = ((n_blocks + 7u) / 8u) * 8u;
    const uint32_t block = a.tile_size <= 8 ? 64u : 256u;
    size_t smem = kQBatch * (2 * sizeof(float4) + sizeof(float2) + sizeof(int32_t));
    if (MODE == kQTop) {
        smem += (size_t)a.K * block * 12;
        if (smem > 160 * 1024 - 4096) {
            set_last_error("gsx_raster3d_top_contributing: num_depth_samples %u needs %zu bytes of LDS per tile", a.K, smem);
            return GSX_ERR_ARG;
        }
        static PerDeviceOnce once;
        if (once.first()) {
            (void)hipFuncSetAttribute((const void *)raster3d_query_kernel<kQTop>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 4096);
        }
    }
    raster3d_query_kernel<MODE><<<dim3(grid), dim3(block), smem, stream>>>(a);
    return check_launch("raster3d_query");
}

static int fill_query(const char *fn, QueryArgs &a, const float *means2d, const float *conics, const float *opacities,
                      const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images, uint32_t n_isects,
                      uint32_t n_per_image, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                      uint32_t tile_h)
{
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "%s: tile_size must be in [1,16], got %u", fn, tile_size);
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && opacities && flatten_ids), "%s: null input", fn);
    GSX_REQUIRE(isect_offsets != nullptr || n_images * tile_w * tile_h == 0, "%s: null isect_offsets", fn);
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height; a.tile_size = tile_size;
    a.tile_w = tile_w; a.tile_h = tile_h; a.n_per_image = n_per_image;
    a.means2d = means2d; a.conics = conics; a.opacities = opacities; a.isect_offsets = isect_offsets;
    a.flatten_ids = flatten_ids;
    return GSX_OK;
}

static int fill_sparse(const char *fn, QueryArgs &a, const int32_t *active_tiles, const uint64_t *tile_pixel_mask,
                       const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active, uint32_t words)
{
    GSX_REQUIRE(n_active == 0 || (active_tiles && tile_pixel_mask && tile_pixel_cumsum && pixel_map), "%s: null layout", fn);
    GSX_REQUIRE(n_active == 0 || words * 64u >= a.tile_size * a.tile_size, "%s: pixel mask too narrow", fn);
    a.sp_active_tiles = active_tiles; a.sp_pixel_mask = tile_pixel_mask; a.sp_pixel_cumsum = tile_pixel_cumsum;
    a.sp_pixel_map = pixel_map; a.n_active = n_active; a.sp_words = words;
    return GSX_OK;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_raster3d_num_contributing(const float *means2d, const float *conics, const float *opacities,
                                             const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                             uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                             uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *counts,
                                             float *alphas, void *stream)
{
    QueryArgs a{};
    int rc = fill_query("gsx_raster3d_num_contributing", a, means2d, conics, opacities, isect_offsets, flatten_ids, n_images,
                        n_isects, n_per_image, width, height, tile_size, tile_w, tile_h);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(counts && alphas, "gsx_raster3d_num_contributing: null output");
    a.counts = counts; a.alphas = alphas;
    return launch_query<kQCount>(a, (hipStream_t)stream);
}

extern "C" int gsx_raster3d_contributing_ids(const float *means2d, const float *conics, const float *opacities,
                                             const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                             uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                             uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t max_contributing,
                                             int32_t *ids, float *weights, void *stream)
{
    if (max_contributing == 0) return GSX_OK;
    QueryArgs a{};
    int rc = fill_query("gsx_raster3d_contributing_ids", a, means2d, conics, opacities, isect_offsets, flatten_ids, n_images,
                        n_isects, n_per_image, width, height, tile_size, tile_w, tile_h);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(ids && weights, "gsx_raster3d_contributing_ids: null output");
    a.K = max_contributing; a.ids = ids; a.weights = weights;
    return launch_query<kQIds>(a, (hipStream_t)stream);
}

extern "C" int gsx_raster3d_top_contributing(const float *means2d, const float *conics, const float *opacities,
                                             const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                             uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height,
                                             uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, uint32_t num_depth_samples,
                                             int32_t *ids, float *weights, void *stream)
{
    if (num_depth_samples == 0) return GSX_OK;
    QueryArgs a{};
    int rc = fill_query("gsx_raster3d_top_contributing", a, means2d, conics, opacities, isect_offsets, flatten_ids, n_images,
                        n_isects, n_per_image, width, height, tile_size, tile_w, tile_h);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(ids && weights, "gsx_raster3d_top_contributing: null output");
    a.K = num_depth_samples; a.ids = ids; a.weights = weights;
    return launch_query<kQTop>(a, (hipStream_t)stream);
}

// Instrumentation (not a reference op): how much work the compositing pass of this scene is. Replays the forward walk and
// adds four counters to stats[0..3] (zeroed by the caller; see the kernel): bench.py prices the vector ALU with them
// (pairs x (14 + 2 D) flop against the fp32 peak).
extern "C" int gsx_raster3d_pair_stats(const float *means2d, const float *conics, const float *opacities,
                                       const int32_t *isect_offsets, const int32_t *flatten_ids, uint32_t n_images,
                                       uint32_t n_isects, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t tile_w,
                                       uint32_t tile_h, uint64_t *stats, void *stream)
{
    QueryArgs a{};
    int rc = fill_query("gsx_raster3d_pair_stats", a, means2d, conics, opacities, isect_offsets, flatten_ids, n_images, n_isects,
                        0, width, height, tile_size, tile_w, tile_h);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(stats, "gsx_raster3d_pair_stats: null output");
    a.stats = reinterpret_cast<unsigned long long *>(stats);
    return launch_query<kQStats>(a, (hipStream_t)stream);
}

// ---- sparse pixel sets: gsplat::rasterize_*_sparse (ext.cpp:1115-1140); outputs are rows [P, ...] in the caller's order ----
#define GSX_SPARSE_QUERY_PARAMS                                                                                              \
    const float *means2d, const float *conics, const float *opacities, const int32_t *active_tiles,                          \
        const int32_t *tile_offsets, const int32_t *flatten_ids, const uint64_t *tile_pixel_mask,                            \
        const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active, uint32_t words_per_tile,              \
        uint32_t n_images, uint32_t n_isects, uint32_t n_per_image, uint32_t width, uint32_t height, uint32_t tile_size,     \
        uint32_t tile_w, uint32_t tile_h
#define GSX_SPARSE_QUERY_FILL(fn)                                                                                            \
    if (n_active == 0) return GSX_OK;                                                                                        \
    QueryArgs a{};                                                                                                           \
    int rc = fill_query(fn, a, means2d, conics, opacities, tile_offsets, flatten_ids, n_images, n_isects, n_per_image,       \
                        width, height, tile_size, tile_w, tile_h);                                                           \
    if (rc == GSX_OK) rc = fill_sparse(fn, a, active_tiles, tile_pixel_mask, tile_pixel_cumsum, pixel_map, n_active,         \
                                       words_per_tile);                                                                      \
    if (rc != GSX_OK) return rc;

extern "C" int gsx_raster3d_sparse_num_contributing(GSX_SPARSE_QUERY_PARAMS, int32_t *counts, float *alphas, void *stream)
{
    GSX_SPARSE_QUERY_FILL("gsx_raster3d_sparse_num_contributing")
    GSX_REQUIRE(counts && alphas, "gsx_raster3d_sparse_num_contributing: null output");
    a.counts = counts; a.alphas = alphas;
    return launch_query<kQCount>(a, (hipStream_t)stream);
}

extern "C" int gsx_raster3d_sparse_contributing_ids(GSX_SPARSE_QUERY_PARAMS, uint32_t max_contributing, int32_t *ids,
                                                    float *weights, void *stream)
{
    if (max_contributing == 0) return GSX_OK;
    GSX_SPARSE_QUERY_FILL("gsx_raster3d_sparse_contributing_ids")
    GSX_REQUIRE(ids && weights, "gsx_raster3d_sparse_contributing_ids: null output");
    a.K = max_contributing; a.ids = ids; a.weights = weights;
    return launch_query<kQIds>(a, (hipStream_t)stream);
}

extern "C" int gsx_raster3d_sparse_top_contributing(GSX_SPARSE_QUERY_PARAMS, uint32_t num_depth_samples, int32_t *ids,
                                                    float *weights, void *stream)
{
    if (num_depth_samples == 0) return GSX_OK;
    GSX_SPARSE_QUERY_FILL("gsx_raster3d_sparse_top_contributing")
    GSX_REQUIRE(ids && weights, "gsx_raster3d_sparse_top_contributing: null output");
    a.K = num_depth_samples; a.ids = ids; a.weights = weights;
    return launch_query<kQTop>(a, (hipStream_t)stream);
}
