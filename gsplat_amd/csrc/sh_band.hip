// Spherical harmonics over a WINDOW of bands, coefficients in fp32 or fp16, gfx950.
// C-ABI entries: gsx_sh_band_fwd, gsx_sh_band_bwd. They are the kernels behind
//   gsplat::spherical_harmonics_l1_plus{,_bwd}  (first_band = 1: shN [N, K-1, 3] holds bands 1 .. K-1 and is read in place -
//       a trainer that stores sh0 and shN apart never materialises their concatenation; reference
//       gsplat/cuda/csrc/SphericalHarmonicsL1PlusCUDA.cu:441 fwd, :648 bwd), and
//   gsplat::spherical_harmonics{,_bwd} with at::kHalf coefficients (first_band = 0; reference
//       SphericalHarmonicsCUDA.cu:609-638 fwd dispatch, :1306-1328 bwd: coefficients half, arithmetic and colours float).
// (fp32 coefficients with first_band = 0 stay on the tuned kernels of sh.hip.)
//
// Layout of the work, chosen for the memory system (this op is an HBM stream: ~200 B of coefficients per row at degree 3):
//   forward   one thread per (image, Gaussian) row. When the 64 rows of a wave are consecutive Gaussians, their coefficient
//             rows are ONE contiguous block (64 x 45 floats at degree 3): the wave copies it global -> LDS in 16-byte pieces
//             (1 KiB per instruction instead of 4 bytes of 64 different lines), then every lane reads its own row (row stride
//             45 words: odd, conflict-free).
//   backward  one thread per GAUSSIAN walks the images: the gradient row is accumulated in registers and written once,
//             through the same tile, with coalesced 16-byte stores - no atomics, no zero fill; rows of Gaussians that no
//             image sees are written as zeros. Packed rows are reached through the [B*C*N] row map (gsx_packed_row_map).
#include "sh_math.hpp"
#include <hip/hip_fp16.h>

namespace gsx {

struct BandArgs {
    const float *means, *viewmats;
    const void *coeffs;
    const uint8_t *masks;
    const int64_t *batch_ids, *camera_ids, *gaussian_ids;
    uint32_t B, C, N, KM; // KM: bases per coefficient row IN MEMORY (K - first_band)
    int64_t nnz;          // < 0: dense rows [B, C, N]
    float *colors;        // fwd [rows, 3]
    const float *v_colors; // bwd [rows, 3]
    const int32_t *row_map; // bwd, packed: [B*C*N] -> packed row or -1
    void *v_coeffs;       // bwd [N, KM, 3]
    float *v_means;       // bwd [B, N, 3] or null
    float *v_dirs;        // bwd [rows, 3] or null: d(loss)/d(view direction) per row (-> v_viewmats on the host side)
    int tile_ok;          // coefficient / gradient blocks may move in 16-byte pieces (base pointers 16-byte aligned)
};

template <typename T> __device__ __forceinline__ float cf_load(const T *p);
template <> __device__ __forceinline__ float cf_load<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float cf_load<__half>(const __half *p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void cf_store(T *p, float v);
template <> __device__ __forceinline__ void cf_store<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void cf_store<__half>(__half *p, float v) { *p = __float2half(v); }

__device__ __forceinline__ void band_view_dir(const BandArgs &a, uint32_t b, uint32_t c, uint32_t g, float *d)
{
    const float *m = a.means + ((size_t)b * a.N + g) * 3;
    const float *V = a.viewmats + ((size_t)b * a.C + c) * 16;
    const float tx = V[3], ty = V[7], tz = V[11];
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j] = m[j] + V[j] * tx + V[4 + j] * ty + V[8 + j] * tz;
}

// global <-> LDS copy of the wave's block of `n_rows` consecutive coefficient rows (row_elems elements of T each), in 16-byte
// pieces where the block allows it (its start is 16-byte aligned when the first row index is a multiple of 8) and element by
// element for the ragged end.
// All the 16-byte pieces a lane moves are issued BEFORE the first one is consumed (up to 16 in flight per lane: a full
// 64-row block of degree-3 rows is 11.25 pieces per lane); the straightforward load -> store loop exposed one memory round
// trip per piece (forward 68 us against the 41 us of the full-band kernel for 6 % fewer bytes).
template <typename T, bool TO_LDS>
__device__ __forceinline__ void band_block_copy(T *gmem, T *lds, uint32_t n_elems, uint32_t lane, bool wide = true)
{
    constexpr uint32_t per16 = 16u / sizeof(T);
    const uint32_t n16 = wide ? n_elems / per16 : 0u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 *g4 = reinterpret_cast<u4 *>(gmem);
    u4 *l4 = reinterpret_cast<u4 *>(lds);
    constexpr int kFlight = 16;
    for (uint32_t base = 0; base < n16; base += 64u * kFlight) {
        u4 v[kFlight];
#pragma unroll
        for (int it = 0; it < kFlight; ++it) {
            const uint32_t i = base + (uint32_t)it * 64u + lane;
            if (i < n16) v[it] = TO_LDS ? __builtin_nontemporal_load(g4 + i) : l4[i];
        }
#pragma unroll
        for (int it = 0; it < kFlight; ++it) {
            const uint32_t i = base + (uint32_t)it * 64u + lane;
            if (i < n16) {
                if (TO_LDS) l4[i] = v[it];
                else __builtin_nontemporal_store(v[it], g4 + i);
            }
        }
    }
    for (uint32_t i = n16 * per16 + lane; i < n_elems; i += 64u) {
        if (TO_LDS) lds[i] = gmem[i];
        else gmem[i] = lds[i];
    }
}

// ---- forward: one thread per row ---------------------------------------------------------------------------------------
template <int DEG, int K0, typename T>
__global__ void __launch_bounds__(256) sh_band_fwd_kernel(const BandArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NBM = NB - K0; // bases evaluated, of them in memory
    extern __shared__ __attribute__((aligned(16))) unsigned char band_smem[];
    const int64_t rows  = a.nnz >= 0 ? a.nnz : (int64_t)a.B * a.C * a.N;
    const int64_t row   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool have = row < rows;
    uint32_t b = 0, c = 0, g = 0;
    if (have) {
        if (a.nnz >= 0) { b = (uint32_t)a.batch_ids[row]; c = (uint32_t)a.camera_ids[row]; g = (uint32_t)a.gaussian_ids[row]; }
        else { g = (uint32_t)(row % a.N); c = (uint32_t)((row / a.N) % a.C); b = (uint32_t)(row / ((int64_t)a.N * a.C)); }
    }
    const bool live = have && !(a.masks && !a.masks[row]);
    const uint64_t live_mask = __builtin_amdgcn_ballot_w64(live);
    const uint32_t row_elems = a.KM * 3u;
    const T *co_g = reinterpret_cast<const T *>(a.coeffs);
    const T *my_row = co_g + (size_t)g * row_elems;
    bool from_tile  = false; // wave-uniform
    if (NBM > 0 && live_mask != 0ull) {
        // the wave's coefficient rows are one block when its Gaussians are g0, g0 + 1, ... (dense rows inside one image)
        const uint32_t g0 = (uint32_t)__shfl((int)g, 0);
        const bool consecutive = a.tile_ok && __builtin_amdgcn_ballot_w64(have && g != g0 + lane) == 0ull && (g0 & 7u) == 0u;
        if (consecutive) {
            const uint32_t n_rows = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(have));
            T *tile = reinterpret_cast<T *>(band_smem) + (size_t)wave * 64u * row_elems;
            band_block_copy<T, true>(const_cast<T *>(co_g) + (size_t)g0 * row_elems, tile, n_rows * row_elems, lane);
            wave_lds_sync();
            from_tile = true;
        }
    }
    if (!have) return;
    float *out = a.colors + row * 3;
    if (!live || NBM <= 0) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    float d[3];
    band_view_dir(a, b, c, g, d);
    const float n2  = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float inv = n2 > 0.0f ? rsqrtf(n2) : 0.0f;
    float Y[NB];
    sh_bases<false>(DEG, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    // two copies of the loop: one pointer that is either LDS or global makes every read a FLAT load (68 us instead of 45)
    auto accumulate = [&](const T *src) {
#pragma unroll
        for (int k = K0; k < NB; ++k) {
            const T *q = src + 3 * (k - K0);
            r0 += Y[k] * cf_load<T>(q);
            r1 += Y[k] * cf_load<T>(q + 1);
            r2 += Y[k] * cf_load<T>(q + 2);
        }
    };
    if (from_tile) accumulate(reinterpret_cast<const T *>(band_smem) + ((size_t)wave * 64u + lane) * row_elems);
    else accumulate(my_row);
    out[0] = r0; out[1] = r1; out[2] = r2;
}

// ---- backward: one thread per Gaussian ----------------------------------------------------------------------------------
template <int DEG, int K0, typename T, bool WANT_DIR>
__global__ void __launch_bounds__(256) sh_band_bwd_kernel(const BandArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NBM = NB - K0, NF = (NBM > 0 ? NBM : 1) * 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char band_smem[];
    const uint32_t g    = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool have = g < a.N;
    const uint32_t row_elems = a.KM * 3u;
    T *tile = reinterpret_cast<T *>(band_smem) + (size_t)wave * 64u * row_elems;
    const uint32_t g0     = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); // first Gaussian of the wave: a multiple of 64
    const uint32_t n_rows = g0 < a.N ? min(64u, a.N - g0) : 0u;
    if (n_rows == 0u) return; // wave-uniform
    T *co_g = const_cast<T *>(reinterpret_cast<const T *>(a.coeffs));
    // the coefficient rows are only needed for the gradient of the direction; they stay in the tile (LDS) while the images
    // are walked - holding them in registers next to the gradient row costs 45 more VGPRs (2 waves per SIMD instead of 3)
    if (WANT_DIR && NBM > 0) {
        band_block_copy<T, true>(co_g + (size_t)g0 * row_elems, tile, n_rows * row_elems, lane, a.tile_ok != 0);
        wave_lds_sync();
    }
    const T *co = tile + (size_t)lane * row_elems;
    float vco[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) vco[i] = 0.0f;
    if (have) {
        for (uint32_t b = 0; b < a.B; ++b) {
            float vm[3] = {0.f, 0.f, 0.f};
            for (uint32_t c = 0; c < a.C; ++c) {
                int64_t row = ((int64_t)b * a.C + c) * a.N + g;
                if (a.row_map) row = a.row_map[row];
                if (row < 0) continue;
                if (a.masks && !a.masks[row]) {
                    if (WANT_DIR && a.v_dirs) a.v_dirs[row * 3] = a.v_dirs[row * 3 + 1] = a.v_dirs[row * 3 + 2] = 0.0f;
                    continue;
                }
                const float vc0 = a.v_colors[row * 3], vc1 = a.v_colors[row * 3 + 1], vc2 = a.v_colors[row * 3 + 2];
                float d[3];
                band_view_dir(a, b, c, g, d);
                const float n2  = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                const float inv = n2 > 0.0f ? rsqrtf(n2) : 0.0f;
                const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
                if constexpr (WANT_DIR) {
                    float Y[NB], Yx[NB], Yy[NB], Yz[NB];
                    sh_bases<true>(DEG, x, y, z, Y, Yx, Yy, Yz);
                    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                    for (int k = K0; k < NB; ++k) {
                        const int i = 3 * (k - K0);
                        vco[i] += Y[k] * vc0; vco[i + 1] += Y[k] * vc1; vco[i + 2] += Y[k] * vc2;
                        const float w = cf_load<T>(co + i) * vc0 + cf_load<T>(co + i + 1) * vc1 + cf_load<T>(co + i + 2) * vc2;
                        gx += Yx[k] * w; gy += Yy[k] * w; gz += Yz[k] * w;
                    }
                    const float dot = gx * x + gy * y + gz * z; // through the normalisation: (g - (g.n) n) / |d|
                    const float vx = (gx - dot * x) * inv, vy = (gy - dot * y) * inv, vz = (gz - dot * z) * inv;
                    vm[0] += vx; vm[1] += vy; vm[2] += vz;
                    if (a.v_dirs) { a.v_dirs[row * 3] = vx; a.v_dirs[row * 3 + 1] = vy; a.v_dirs[row * 3 + 2] = vz; }
                } else {
                    float Y[NB];
                    sh_bases<false>(DEG, x, y, z, Y, nullptr, nullptr, nullptr);
#pragma unroll
                    for (int k = K0; k < NB; ++k) {
                        const int i = 3 * (k - K0);
                        vco[i] += Y[k] * vc0; vco[i + 1] += Y[k] * vc1; vco[i + 2] += Y[k] * vc2;
                    }
                }
            }
            if (WANT_DIR && a.v_means) {
                float *o = a.v_means + ((size_t)b * a.N + g) * 3;
                o[0] = vm[0]; o[1] = vm[1]; o[2] = vm[2];
            }
        }
    }
    wave_lds_sync(); // every lane is done reading coefficient rows from the tile
    if (have) {
        // this lane's gradient row -> tile (bands past the evaluated degree are zero)
        T *mine = tile + (size_t)lane * row_elems;
        for (uint32_t i = 0; i < row_elems; ++i) cf_store<T>(mine + i, 0.0f);
        if (NBM > 0) {
#pragma unroll
            for (int i = 0; i < NBM * 3; ++i) cf_store<T>(mine + i, vco[i]);
        }
    }
    wave_lds_sync();
    band_block_copy<T, false>(reinterpret_cast<T *>(a.v_coeffs) + (size_t)g0 * row_elems, tile, n_rows * row_elems, lane, a.tile_ok != 0);
}

template <int K0, typename T>
static void launch_band_fwd(int deg, const BandArgs &a, int64_t rows, hipStream_t s)
{
    const uint32_t threads = (size_t)256 * a.KM * 3 * sizeof(T) > 48 * 1024 ? 64u : 256u; // the 4 waves' tiles must fit 64 KiB
    const dim3 grid((uint32_t)ceil_div(rows, threads)), block(threads);
    const size_t lds = (size_t)threads * a.KM * 3 * sizeof(T);
    switch (deg) {
    case 0: sh_band_fwd_kernel<0, K0, T><<<grid, block, lds, s>>>(a); break;
    case 1: sh_band_fwd_kernel<1, K0, T><<<grid, block, lds, s>>>(a); break;
    case 2: sh_band_fwd_kernel<2, K0, T><<<grid, block, lds, s>>>(a); break;
    case 3: sh_band_fwd_kernel<3, K0, T><<<grid, block, lds, s>>>(a); break;
    default: sh_band_fwd_kernel<4, K0, T><<<grid, block, lds, s>>>(a); break;
    }
}

template <int K0, typename T, bool WANT_DIR>
static void launch_band_bwd(int deg, const BandArgs &a, hipStream_t s)
{
    const uint32_t threads = (size_t)256 * a.KM * 3 * sizeof(T) > 48 * 1024 ? 64u : 256u;
    const dim3 grid((uint32_t)ceil_div((int64_t)a.N, threads)), block(threads);
    const size_t lds = (size_t)threads * a.KM * 3 * sizeof(T);
    switch (deg) {
    case 0: sh_band_bwd_kernel<0, K0, T, WANT_DIR><<<grid, block, lds, s>>>(a); break;
    case 1: sh_band_bwd_kernel<1, K0, T, WANT_DIR><<<grid, block, lds, s>>>(a); break;
    case 2: sh_band_bwd_kernel<2, K0, T, WANT_DIR><<<grid, block, lds, s>>>(a); break;
    case 3: sh_band_bwd_kernel<3, K0, T, WANT_DIR><<<grid, block, lds, s>>>(a); break;
    default: sh_band_bwd_kernel<4, K0, T, WANT_DIR><<<grid, block, lds, s>>>(a); break;
    }
}

static int check_band(const char *fn, int degree, int first_band, int dtype, uint32_t K, const void *means, const void *viewmats,
                      const void *coeffs, int64_t nnz, const void *bi, const void *ci, const void *gi)
{
    GSX_REQUIRE(degree >= 0 && degree <= 4, "%s: degrees_to_use must be in [0,4], got %d", fn, degree);
    GSX_REQUIRE(first_band == 0 || first_band == 1, "%s: first_band must be 0 or 1", fn);
    GSX_REQUIRE(dtype == 0 || dtype == 1, "%s: coefficient dtype must be 0 (float32) or 1 (float16)", fn);
    GSX_REQUIRE((uint32_t)((degree + 1) * (degree + 1)) <= K, "%s: K=%u bases too few for degree %d", fn, K, degree);
    GSX_REQUIRE(K > (uint32_t)first_band || degree == 0, "%s: empty coefficient rows", fn);
    GSX_REQUIRE(K <= 64, "%s: K=%u exceeds the LDS tile (64 bases)", fn, K);
    GSX_REQUIRE(means && viewmats && (coeffs || K == (uint32_t)first_band), "%s: null input", fn);
    GSX_REQUIRE(nnz < 0 || nnz == 0 || (bi && ci && gi), "%s: packed mode needs batch/camera/gaussian ids", fn);
    return GSX_OK;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_sh_band_fwd(int degrees_to_use, int first_band, int coeff_dtype, const float *means, const float *viewmats,
                               const void *coeffs, const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                               const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz, uint32_t K,
                               float *colors, void *stream)
{
    const int64_t rows = nnz >= 0 ? nnz : (int64_t)B * C * N;
    if (rows == 0) return GSX_OK;
    int rc = check_band("gsx_sh_band_fwd", degrees_to_use, first_band, coeff_dtype, K, means, viewmats, coeffs, nnz, batch_ids,
                        camera_ids, gaussian_ids);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(colors, "gsx_sh_band_fwd: null output");
    BandArgs a{};
    a.means = means; a.viewmats = viewmats; a.coeffs = coeffs; a.masks = masks;
    a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.B = B; a.C = C; a.N = N; a.KM = K - (uint32_t)first_band; a.nnz = nnz; a.colors = colors;
    a.tile_ok = (reinterpret_cast<uintptr_t>(coeffs) & 15u) == 0u;
    hipStream_t s = (hipStream_t)stream;
    if (first_band == 0) {
        if (coeff_dtype == 0) launch_band_fwd<0, float>(degrees_to_use, a, rows, s);
        else launch_band_fwd<0, __half>(degrees_to_use, a, rows, s);
    } else {
        if (coeff_dtype == 0) launch_band_fwd<1, float>(degrees_to_use, a, rows, s);
        else launch_band_fwd<1, __half>(degrees_to_use, a, rows, s);
    }
    return check_launch("sh_band_fwd");
}

extern "C" int gsx_sh_band_bwd(int degrees_to_use, int first_band, int coeff_dtype, const float *means, const float *viewmats,
                               const void *coeffs, const uint8_t *masks, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                               uint32_t K, const float *v_colors, const int32_t *row_map, void *v_coeffs, float *v_means,
                               float *v_dirs, void *stream)
{
    if (N == 0) return GSX_OK;
    int rc = check_band("gsx_sh_band_bwd", degrees_to_use, first_band, coeff_dtype, K, means, viewmats, coeffs, -1, nullptr,
                        nullptr, nullptr);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(v_coeffs, "gsx_sh_band_bwd: null v_coeffs");
    GSX_REQUIRE(nnz < 0 || row_map, "gsx_sh_band_bwd: packed rows need the [B*C*N] row map (gsx_packed_row_map)");
    GSX_REQUIRE(v_colors || (int64_t)B * C == 0 || nnz == 0, "gsx_sh_band_bwd: null v_colors");
    BandArgs a{};
    a.means = means; a.viewmats = viewmats; a.coeffs = coeffs; a.masks = masks;
    a.B = B; a.C = C; a.N = N; a.KM = K - (uint32_t)first_band; a.nnz = nnz;
    a.v_colors = v_colors; a.row_map = nnz >= 0 ? row_map : nullptr; a.v_coeffs = v_coeffs; a.v_means = v_means; a.v_dirs = v_dirs;
    a.tile_ok = ((reinterpret_cast<uintptr_t>(coeffs) | reinterpret_cast<uintptr_t>(v_coeffs)) & 15u) == 0u;
    if (a.KM == 0) return GSX_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool want_dir = v_means != nullptr || v_dirs != nullptr;
#define GSX_BAND_BWD(K0, T)                                                     \
    do {                                                                        \
        if (want_dir) launch_band_bwd<K0, T, true>(degrees_to_use, a, s);       \
        else launch_band_bwd<K0, T, false>(degrees_to_use, a, s);               \
    } while (0)
    if (first_band == 0) {
        if (coeff_dtype == 0) GSX_BAND_BWD(0, float);
        else GSX_BAND_BWD(0, __half);
    } else {
        if (coeff_dtype == 0) GSX_BAND_BWD(1, float);
        else GSX_BAND_BWD(1, __half);
    }
#undef GSX_BAND_BWD
    return check_launch("sh_band_bwd");
}
