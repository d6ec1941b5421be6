// Spherical-harmonics colour evaluation (degree <= 4) forward / backward, gfx950.
// C-ABI entries: gsx_sh_fwd, gsx_sh_bwd. Replaces gsplat::spherical_harmonics{,_bwd}
// (reference gsplat/cuda/csrc/SphericalHarmonicsCUDA.cu:48-146 basis, :444-569 fwd, :786-890 bwd;
//  torch restatement gsplat/cuda/_torch_impl.py:968-1067).
//
// Basis: Sloan, "Efficient Spherical Harmonic Evaluation" (JCGT 2013) polynomial forms; the
// constants are the published ones. Direction = mean - camera centre, centre = -R^T t.
#include "sh_math.hpp"

namespace gsx {

struct ShArgs {
    int degree;
    const float *means, *viewmats, *coeffs;
    const uint8_t *masks;
    const int64_t *batch_ids, *camera_ids, *gaussian_ids;
    uint32_t B, C, N, K, D;
    int64_t nnz; // < 0: dense
    int coeffs_gathered; // packed: 1 = coeffs is [nnz,K,D]; 0 = coeffs is [N,K,D] indexed by gaussian_ids
    // fused pieces of the rasterization() orchestrator (all optional):
    const int32_t *radii;       // [rows,2]: alternative to masks, row is live iff both radii > 0 (Rendering.cpp:1146)
    int post;                   // forward: colors = max(sh + 0.5, 0)  (Rendering.cpp:1160 / rendering.py:714-718)
    const float *post_colors;   // backward: that forward output; the gradient is cut where it is 0 (clamp_min VJP)
    uint32_t vc_stride;         // row stride (floats) of v_colors: D when contiguous (AoS gradient rows otherwise)
    float *colors;
    const float *v_colors;
    float *v_coeffs, *v_means;
    const int32_t *row_map; // optional, packed rows only: [B*C*N] -> packed row or -1 (Gaussian-major backward)
    float *v_dirs; // optional [rows,3], zero-initialised: per-row d(loss)/d(view direction) (-> v_viewmats on the host)
    int atomic_coeffs; // packed + !gathered + more than one image: rows of one Gaussian collide
    // optional, D == 3: the compositing kernels' 48-byte array-of-structures row of every LIVE row (csrc/raster3d.hpp:
    // Raster3DArgs::splat_rows) = (x, y, conic a, b | conic c, opacity, colour 0, 1 | colour 2, 0, 0, 0), written while the row's
    // colours are in registers; r_* are the projection's outputs for the same rows
    const float *r_means2d, *r_conics, *r_opacities; // [rows,2] [rows,3] [rows]
    float *splat_rows;                               // [rows,12]
};
__device__ __forceinline__ void write_splat_row(const ShArgs &a, int64_t row, float c0, float c1, float c2)
{
    if (!a.splat_rows) return;
    const float2 xy = reinterpret_cast<const float2 *>(a.r_means2d)[row];
    const float *cn = a.r_conics + 3 * row;
    v4f *dst = reinterpret_cast<v4f *>(a.splat_rows) + 3 * row;
    dst[0] = v4f{xy.x, xy.y, cn[0], cn[1]};
    dst[1] = v4f{cn[2], a.r_opacities[row], c0, c1};
    dst[2] = v4f{c2, 0.0f, 0.0f, 0.0f};
}

__device__ __forceinline__ bool row_dead(const ShArgs &a, int64_t row)
{
    if (a.masks && !a.masks[row]) return true;
    if (a.radii && !(a.radii[2 * row] > 0 && a.radii[2 * row + 1] > 0)) return true;
    return false;
}
__device__ __forceinline__ float post_color(const ShArgs &a, float x) { return a.post ? fmaxf(x + 0.5f, 0.0f) : x; }
// incoming gradient of channel ch of `row` (through the fused clamp when the forward applied it)
__device__ __forceinline__ float load_vc(const ShArgs &a, int64_t row, uint32_t ch)
{
    const float v = a.v_colors[row * a.vc_stride + ch];
    return (a.post_colors && !(a.post_colors[row * a.D + ch] > 0.0f)) ? 0.0f : v;
}

// generic kernels: accumulate one (row, channel) contribution to d(loss)/d(dir) into v_means[b,g] and / or v_dirs[row]
__device__ __forceinline__ void add_v_dir(const ShArgs &a, int64_t row, uint32_t b, uint32_t g, float vx, float vy, float vz)
{
    if (a.v_means) {
        float *vm = a.v_means + ((size_t)b * a.N + g) * 3;
        atomic_add_f32(vm + 0, vx); atomic_add_f32(vm + 1, vy); atomic_add_f32(vm + 2, vz);
    }
    if (a.v_dirs) {
        float *vd = a.v_dirs + row * 3;
        atomic_add_f32(vd + 0, vx); atomic_add_f32(vd + 1, vy); atomic_add_f32(vd + 2, vz);
    }
}

// unnormalised view direction of gaussian (b,g) seen from camera (b,c): mean + R^T t
__device__ __forceinline__ void view_dir(const ShArgs &a, uint32_t b, uint32_t c, uint32_t g, float *d)
{
    const float *m = a.means + ((size_t)b * a.N + g) * 3;
    const float *V = a.viewmats + ((size_t)b * a.C + c) * 16;
    const float tx = V[3], ty = V[7], tz = V[11];
#pragma unroll
    for (int j = 0; j < 3; ++j) d[j] = m[j] + V[j] * tx + V[4 + j] * ty + V[8 + j] * tz;
}

__device__ __forceinline__ float safe_inv_norm(const float *d)
{
    const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    return n2 > 0.0f ? rsqrtf(n2) : 0.0f;
}

// one thread per (row, channel)
__global__ void __launch_bounds__(256) sh_fwd_kernel(const ShArgs a)
{
    const int64_t rows = a.nnz >= 0 ? a.nnz : (int64_t)a.B * a.C * a.N;
    const int64_t idx  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * a.D) return;
    const int64_t row = idx / a.D;
    const uint32_t ch = (uint32_t)(idx % a.D);
    if (row_dead(a, row)) {
        a.colors[idx] = 0.0f;
        return;
    }
    uint32_t b, c, g;
    int64_t crow;
    if (a.nnz >= 0) {
        b = (uint32_t)a.batch_ids[row]; c = (uint32_t)a.camera_ids[row]; g = (uint32_t)a.gaussian_ids[row];
        crow = a.coeffs_gathered ? row : (int64_t)g;
    } else {
        g = (uint32_t)(row % a.N); c = (uint32_t)((row / a.N) % a.C); b = (uint32_t)(row / ((int64_t)a.N * a.C));
        crow = g;
    }
    float d[3];
    view_dir(a, b, c, g, d);
    const float inv = safe_inv_norm(d);
    float Y[kMaxBases];
    sh_bases<false>(a.degree, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
    const int nb    = (a.degree + 1) * (a.degree + 1);
    const float *co = a.coeffs + ((size_t)crow * a.K) * a.D + ch;
    float acc       = 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxBases; ++k)
        if (k < nb) acc += Y[k] * co[(size_t)k * a.D];
    a.colors[idx] = post_color(a, acc);
}

// dense backward: one thread per (gaussian, channel); loops over all B*C images in registers,
// writes v_coeffs once (no atomics). v_means (optional, zero-initialised) via atomics.
__global__ void __launch_bounds__(256) sh_bwd_dense_kernel(const ShArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.N * a.D) return;
    const uint32_t g = (uint32_t)(idx / a.D), ch = (uint32_t)(idx % a.D);
    const int nb     = (a.degree + 1) * (a.degree + 1);
    float vco[kMaxBases];
#pragma unroll
    for (int k = 0; k < kMaxBases; ++k) vco[k] = 0.0f;
    const float *co = a.coeffs + ((size_t)g * a.K) * a.D + ch;

    for (uint32_t b = 0; b < a.B; ++b)
        for (uint32_t c = 0; c < a.C; ++c) {
            const int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            if (row_dead(a, row)) continue;
            const float vc = load_vc(a, row, ch);
            float d[3];
            view_dir(a, b, c, g, d);
            const float inv = safe_inv_norm(d);
            const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
            if (a.v_means || a.v_dirs) {
                float Y[kMaxBases], Yx[kMaxBases], Yy[kMaxBases], Yz[kMaxBases];
                sh_bases<true>(a.degree, x, y, z, Y, Yx, Yy, Yz);
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                for (int k = 0; k < kMaxBases; ++k)
                    if (k < nb) {
                        vco[k] += Y[k] * vc;
                        const float w = co[(size_t)k * a.D] * vc;
                        gx += Yx[k] * w; gy += Yy[k] * w; gz += Yz[k] * w;
                    }
                // through the normalisation: v_d = (g - (g.n) n) / |d|
                const float dot = gx * x + gy * y + gz * z;
                add_v_dir(a, row, b, g, (gx - dot * x) * inv, (gy - dot * y) * inv, (gz - dot * z) * inv);
            } else {
                float Y[kMaxBases];
                sh_bases<false>(a.degree, x, y, z, Y, nullptr, nullptr, nullptr);
#pragma unroll
                for (int k = 0; k < kMaxBases; ++k)
                    if (k < nb) vco[k] += Y[k] * vc;
            }
        }
    float *out = a.v_coeffs + ((size_t)g * a.K) * a.D + ch;
    for (uint32_t k = 0; k < a.K; ++k) out[(size_t)k * a.D] = ((int)k < nb) ? vco[k < kMaxBases ? k : 0] : 0.0f;
}

// packed backward: one thread per (row, channel).
__global__ void __launch_bounds__(256) sh_bwd_packed_kernel(const ShArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.nnz * a.D) return;
    const int64_t row = idx / a.D;
    const uint32_t ch = (uint32_t)(idx % a.D);
    const int nb      = (a.degree + 1) * (a.degree + 1);
    const uint32_t b = (uint32_t)a.batch_ids[row], c = (uint32_t)a.camera_ids[row], g = (uint32_t)a.gaussian_ids[row];
    const int64_t crow = a.coeffs_gathered ? row : (int64_t)g;
    float *out         = a.v_coeffs + ((size_t)crow * a.K) * a.D + ch;
    const bool masked  = row_dead(a, row);
    if (masked) {
        if (a.coeffs_gathered)
            for (uint32_t k = 0; k < a.K; ++k) out[(size_t)k * a.D] = 0.0f;
        return;
    }
    const float vc = load_vc(a, row, ch);
    float d[3];
    view_dir(a, b, c, g, d);
    const float inv = safe_inv_norm(d);
    const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
    float Y[kMaxBases], Yx[kMaxBases], Yy[kMaxBases], Yz[kMaxBases];
    const bool want_dir = a.v_means || a.v_dirs;
    if (want_dir) sh_bases<true>(a.degree, x, y, z, Y, Yx, Yy, Yz);
    else sh_bases<false>(a.degree, x, y, z, Y, nullptr, nullptr, nullptr);
    if (a.coeffs_gathered) {
        for (uint32_t k = 0; k < a.K; ++k) out[(size_t)k * a.D] = ((int)k < nb) ? Y[k < kMaxBases ? k : 0] * vc : 0.0f;
    } else {
        // v_coeffs [N,K,D] zero-initialised by the caller
#pragma unroll
        for (int k = 0; k < kMaxBases; ++k)
            if (k < nb) {
                if (a.atomic_coeffs) atomic_add_f32(out + (size_t)k * a.D, Y[k] * vc);
                else out[(size_t)k * a.D] = Y[k] * vc;
            }
    }
    if (want_dir) {
        const float *co = a.coeffs + ((size_t)crow * a.K) * a.D + ch;
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxBases; ++k)
            if (k < nb) {
                const float w = co[(size_t)k * a.D] * vc;
                gx += Yx[k] * w; gy += Yy[k] * w; gz += Yz[k] * w;
            }
        const float dot = gx * x + gy * y + gz * z;
        add_v_dir(a, row, b, g, (gx - dot * x) * inv, (gy - dot * y) * inv, (gz - dot * z) * inv);
    }
}

// ------------------------------------------------------------------------------------------
// D == 3 fast path: ONE THREAD PER ROW. The basis is evaluated once per row (not once per channel), the coefficient
// row ([K][3] floats, contiguous) is streamed with 16-byte loads when its size allows, and the backward writes the
// whole v_coeffs row from one thread (16-byte stores). Degree is a template parameter so the basis unrolls.
// ------------------------------------------------------------------------------------------
template <int NF>
__device__ __forceinline__ void load_row(const float *src, bool vec, float *dst)
{
    if (vec) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
        for (int i = 0; i < (NF + 3) / 4; ++i) {
            const float4 v = s4[i];
            if (4 * i + 0 < NF) dst[4 * i + 0] = v.x;
            if (4 * i + 1 < NF) dst[4 * i + 1] = v.y;
            if (4 * i + 2 < NF) dst[4 * i + 2] = v.z;
            if (4 * i + 3 < NF) dst[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NF; ++i) dst[i] = src[i];
    }
}

__device__ __forceinline__ void row_ids(const ShArgs &a, int64_t row, uint32_t &b, uint32_t &c, uint32_t &g, int64_t &crow)
{
    if (a.nnz >= 0) {
        b = (uint32_t)a.batch_ids[row]; c = (uint32_t)a.camera_ids[row]; g = (uint32_t)a.gaussian_ids[row];
        crow = a.coeffs_gathered ? row : (int64_t)g;
    } else {
        g = (uint32_t)(row % a.N); c = (uint32_t)((row / a.N) % a.C); b = (uint32_t)(row / ((int64_t)a.N * a.C));
        crow = g;
    }
}

// ---- wave-cooperative coefficient tiles ---------------------------------------------------------------------------------
// A thread that streams its own [K][3] row touches 16 bytes of 64 different cache lines per load / store instruction. When
// the 64 rows of a wave are consecutive in memory (dense rows, or packed rows with gathered coefficients) the wave moves them
// as ONE contiguous block instead: lane l moves float4 number (it * 64 + l) of the block, 1 KiB per instruction, through a
// wave-private LDS tile in which every lane then reads (or has written) its own row. Row stride Q + 1 float4: a lane's 16-byte
// accesses at stride 13 (degree 3) fall on 16 distinct bank quads per group of 16 lanes.
template <int NF>
struct ShTile {
    static constexpr int Q = (NF + 3) / 4, STRIDE = Q + 1;
    static constexpr size_t kBytesPerWave = 64 * STRIDE * sizeof(v4f);
};

// true iff the wave's coefficient rows are crow(lane 0) + lane (rows past the end count as consecutive)
__device__ __forceinline__ bool wave_rows_consecutive(int64_t crow, bool have, uint32_t lane)
{
    const int64_t first = __shfl(crow, 0) ;
    return __builtin_amdgcn_ballot_w64(have && crow != first + (int64_t)lane) == 0ull;
}

// rows whose bit in `mask` is set: global -> tile (first Q float4 of each row; row_q = float4 per row in memory)
template <int NF>
__device__ __forceinline__ void tile_load(const float *block, uint32_t row_q, uint64_t mask, v4f *tile, uint32_t lane)
{
    constexpr int Q = ShTile<NF>::Q, STRIDE = ShTile<NF>::STRIDE;
    const v4f *src = reinterpret_cast<const v4f *>(block);
    v4f v[Q];
    bool on[Q];
#pragma unroll
    for (int it = 0; it < Q; ++it) {
        const uint32_t idx = it * 64 + lane, r = idx / Q, q = idx - r * Q;
        on[it] = (mask >> r) & 1ull;
        if (on[it]) v[it] = __builtin_nontemporal_load(src + (size_t)r * row_q + q);
    }
#pragma unroll
    for (int it = 0; it < Q; ++it) {
        const uint32_t idx = it * 64 + lane, r = idx / Q, q = idx - r * Q;
        if (on[it]) tile[r * STRIDE + q] = v[it];
    }
}

template <int NF>
__device__ __forceinline__ void tile_read_row(const v4f *tile, uint32_t lane, float *dst)
{
    constexpr int Q = ShTile<NF>::Q, STRIDE = ShTile<NF>::STRIDE;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const v4f v = tile[lane * STRIDE + i];
        if (4 * i + 0 < NF) dst[4 * i + 0] = v.x;
        if (4 * i + 1 < NF) dst[4 * i + 1] = v.y;
        if (4 * i + 2 < NF) dst[4 * i + 2] = v.z;
        if (4 * i + 3 < NF) dst[4 * i + 3] = v.w;
    }
}

// the tile rows (each written by its own lane) -> the wave's rows in memory, zero-filled up to row_q float4 per row; rows
// whose bit in `mask` is clear (past the end) are not written
template <int NF>
__device__ __forceinline__ void tile_flush_rows(float *block, uint32_t row_q, uint64_t mask, const v4f *tile, uint32_t lane)
{
    constexpr int Q = ShTile<NF>::Q, STRIDE = ShTile<NF>::STRIDE;
    wave_lds_sync();
    v4f *dst = reinterpret_cast<v4f *>(block);
    if (row_q == (uint32_t)Q) {
#pragma unroll
        for (int it = 0; it < Q; ++it) {
            const uint32_t idx = it * 64 + lane, r = idx / Q, q = idx - r * Q;
            if ((mask >> r) & 1ull) __builtin_nontemporal_store(tile[r * STRIDE + q], dst + idx);
        }
    } else {
        for (uint32_t idx = lane; idx < 64u * row_q; idx += 64u) {
            const uint32_t r = idx / row_q, q = idx - r * row_q;
            if ((mask >> r) & 1ull) dst[idx] = q < (uint32_t)Q ? tile[r * STRIDE + q] : v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int DEG>
__global__ void __launch_bounds__(256) sh3_fwd_kernel(const ShArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_smem[];
    const int64_t rows = a.nnz >= 0 ? a.nnz : (int64_t)a.B * a.C * a.N;
    const int64_t row  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool have = row < rows;
    const bool live = have && !row_dead(a, row);
    uint32_t b = 0, c = 0, g = 0;
    int64_t crow = 0;
    if (have) row_ids(a, row, b, c, g, crow);
    const bool vec = ((a.K * 3u) & 3u) == 0u; // rows are 16-byte aligned iff K*3 floats is a multiple of 4
    float co[NF];
    const uint64_t live_mask = __builtin_amdgcn_ballot_w64(live);
    if (live_mask == 0ull) { // nothing to evaluate in this wave
        if (have) a.colors[row * 3] = a.colors[row * 3 + 1] = a.colors[row * 3 + 2] = 0.0f;
        return;
    }
    const bool tiled = DEG < 4 && vec && wave_rows_consecutive(crow, have, lane); // wave-uniform
    if (tiled) {
        v4f *tile = reinterpret_cast<v4f *>(sh_smem + (threadIdx.x >> 6) * ShTile<NF>::kBytesPerWave);
        const int64_t crow0 = __shfl(crow, 0);
        tile_load<NF>(a.coeffs + (size_t)crow0 * a.K * 3, a.K * 3u / 4u, live_mask, tile, lane);
        wave_lds_sync();
        if (live) tile_read_row<NF>(tile, lane, co);
    }
    if (!have) return;
    float *out = a.colors + row * 3;
    if (!live) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    float d[3];
    view_dir(a, b, c, g, d);
    const float inv = safe_inv_norm(d);
    float Y[NB];
    sh_bases<false>(DEG, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
    if (!tiled) load_row<NF>(a.coeffs + (size_t)crow * a.K * 3, vec, co);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        r0 += Y[k] * co[3 * k];
        r1 += Y[k] * co[3 * k + 1];
        r2 += Y[k] * co[3 * k + 2];
    }
    r0 = post_color(a, r0); r1 = post_color(a, r1); r2 = post_color(a, r2);
    out[0] = r0; out[1] = r1; out[2] = r2;
    write_splat_row(a, row, r0, r1, r2);
}

// Dense rows of SEVERAL images: one thread per Gaussian walks the images, so that the [K][3] coefficient row (192 of the
// ~220 bytes a row touches at degree 3) is fetched once per Gaussian instead of once per (image, Gaussian) - with 4 M
// Gaussians the images' rows are 768 MB apart, no cache holds them in between (c4: 550 -> 2xx us per launch). The 64
// coefficient rows of a wave move as one tile (see ShTile); colours are written row by row, consecutive lanes to
// consecutive rows of the same image.
template <int DEG>
__global__ void __launch_bounds__(256) sh3_fwd_gaussian_major_kernel(const ShArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_smem[];
    const int64_t gi    = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool have     = gi < (int64_t)a.N;
    const uint32_t g    = (uint32_t)gi;
    const uint32_t n_img = a.B * a.C;
    bool any = false;
    if (have)
        for (uint32_t i = 0; i < n_img && !any; ++i) any = !row_dead(a, (int64_t)i * a.N + g);
    const uint64_t any_mask = __builtin_amdgcn_ballot_w64(any);
    float co[NF];
    if (any_mask) {
        v4f *tile = reinterpret_cast<v4f *>(sh_smem + (threadIdx.x >> 6) * ShTile<NF>::kBytesPerWave);
        tile_load<NF>(a.coeffs + (size_t)(gi - lane) * a.K * 3, a.K * 3u / 4u, any_mask, tile, lane);
        wave_lds_sync();
        if (any) tile_read_row<NF>(tile, lane, co);
    }
    if (!have) return;
    for (uint32_t b = 0; b < a.B; ++b)
        for (uint32_t c = 0; c < a.C; ++c) {
            const int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            float *out        = a.colors + row * 3;
            if (row_dead(a, row)) {
                out[0] = out[1] = out[2] = 0.0f;
                continue;
            }
            float d[3];
            view_dir(a, b, c, g, d);
            const float inv = safe_inv_norm(d);
            float Y[NB];
            sh_bases<false>(DEG, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                r0 += Y[k] * co[3 * k];
                r1 += Y[k] * co[3 * k + 1];
                r2 += Y[k] * co[3 * k + 2];
            }
            r0 = post_color(a, r0); r1 = post_color(a, r1); r2 = post_color(a, r2);
            out[0] = r0; out[1] = r1; out[2] = r2;
            write_splat_row(a, row, r0, r1, r2);
        }
}

// store a v_coeffs row: values for the first NF floats, zeros up to K*3 (only when `fill_tail`)
template <int NF>
__device__ __forceinline__ void store_row(float *dst, bool vec, const float *val, uint32_t row_floats, bool fill_tail)
{
    if (vec) {
        float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
        for (int i = 0; i < (NF + 3) / 4; ++i) {
            float4 v;
            v.x = (4 * i + 0 < NF) ? val[4 * i + 0] : 0.f;
            v.y = (4 * i + 1 < NF) ? val[4 * i + 1] : 0.f;
            v.z = (4 * i + 2 < NF) ? val[4 * i + 2] : 0.f;
            v.w = (4 * i + 3 < NF) ? val[4 * i + 3] : 0.f;
            // a partially filled last vector only happens when NF % 4 != 0; its tail floats belong to k >= NB
            if (4 * i + 3 < NF || fill_tail) d4[i] = v;
            else {
                if (4 * i + 0 < NF) dst[4 * i + 0] = v.x;
                if (4 * i + 1 < NF) dst[4 * i + 1] = v.y;
                if (4 * i + 2 < NF) dst[4 * i + 2] = v.z;
            }
        }
        if (fill_tail)
            for (uint32_t i = (uint32_t)((NF + 3) / 4); i < row_floats / 4; ++i) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int i = 0; i < NF; ++i) dst[i] = val[i];
        if (fill_tail)
            for (uint32_t i = NF; i < row_floats; ++i) dst[i] = 0.0f;
    }
}

// per-row VJP pieces shared by the packed and dense backward
template <int DEG, bool WANT_MEANS>
__device__ __forceinline__ void sh3_row_vjp(const ShArgs &a, uint32_t b, uint32_t c, uint32_t g, int64_t crow,
                                            const float *vc, bool vec, float *vco /*[NB*3], accumulated*/,
                                            float *v_dir /*[3], accumulated (d/dmean)*/)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
    float d[3];
    view_dir(a, b, c, g, d);
    const float inv = safe_inv_norm(d);
    const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
    if constexpr (WANT_MEANS) {
        float Y[NB], Yx[NB], Yy[NB], Yz[NB];
        sh_bases<true>(DEG, x, y, z, Y, Yx, Yy, Yz);
        float co[NF];
        load_row<NF>(a.coeffs + (size_t)crow * a.K * 3, vec, co);
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            vco[3 * k] += Y[k] * vc[0]; vco[3 * k + 1] += Y[k] * vc[1]; vco[3 * k + 2] += Y[k] * vc[2];
            const float w = co[3 * k] * vc[0] + co[3 * k + 1] * vc[1] + co[3 * k + 2] * vc[2];
            gx += Yx[k] * w; gy += Yy[k] * w; gz += Yz[k] * w;
        }
        const float dot = gx * x + gy * y + gz * z; // through the normalisation: (g - (g.n) n) / |d|
        v_dir[0] += (gx - dot * x) * inv;
        v_dir[1] += (gy - dot * y) * inv;
        v_dir[2] += (gz - dot * z) * inv;
    } else {
        float Y[NB];
        sh_bases<false>(DEG, x, y, z, Y, nullptr, nullptr, nullptr);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            vco[3 * k] += Y[k] * vc[0]; vco[3 * k + 1] += Y[k] * vc[1]; vco[3 * k + 2] += Y[k] * vc[2];
        }
    }
}

template <int DEG, bool WANT_MEANS>
__global__ void __launch_bounds__(256) sh3_bwd_packed_kernel(const ShArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nnz) return;
    uint32_t b, c, g;
    int64_t crow;
    row_ids(a, row, b, c, g, crow);
    const bool vec = ((a.K * 3u) & 3u) == 0u;
    float *out     = a.v_coeffs + (size_t)crow * a.K * 3;
    float vco[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) vco[i] = 0.0f;
    if (row_dead(a, row)) {
        if (a.coeffs_gathered) store_row<NF>(out, vec, vco, a.K * 3, true);
        return;
    }
    const float vc[3] = {load_vc(a, row, 0), load_vc(a, row, 1), load_vc(a, row, 2)};
    float v_dir[3] = {0.f, 0.f, 0.f};
    sh3_row_vjp<DEG, WANT_MEANS>(a, b, c, g, crow, vc, vec, vco, v_dir);
    if (a.coeffs_gathered) store_row<NF>(out, vec, vco, a.K * 3, true);
    else if (!a.atomic_coeffs) store_row<NF>(out, vec, vco, a.K * 3, false); // zero-initialised by the caller
    else {
#pragma unroll
        for (int i = 0; i < NF; ++i) atomic_add_f32(out + i, vco[i]);
    }
    if constexpr (WANT_MEANS) {
        if (a.v_means) {
            float *vm = a.v_means + ((size_t)b * a.N + g) * 3;
            atomic_add_f32(vm + 0, v_dir[0]);
            atomic_add_f32(vm + 1, v_dir[1]);
            atomic_add_f32(vm + 2, v_dir[2]);
        }
        if (a.v_dirs) {
            float *vd = a.v_dirs + row * 3;
            vd[0] = v_dir[0]; vd[1] = v_dir[1]; vd[2] = v_dir[2];
        }
    }
}

// dense (or packed rows addressed through row_map): one thread per Gaussian loops over the images; v_coeffs and
// v_means are written once per Gaussian — no atomics, no zero-initialised outputs. This version streams every row from its
// own thread (rows of 75 floats, degree 4); sh3_bwd_tiled_kernel below is the one that normally runs.
template <int DEG, bool WANT_MEANS>
__global__ void __launch_bounds__(256) sh3_bwd_dense_kernel(const ShArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= (int64_t)a.N) return;
    const uint32_t g = (uint32_t)gi;
    const bool vec   = ((a.K * 3u) & 3u) == 0u;
    float vco[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) vco[i] = 0.0f;
    for (uint32_t b = 0; b < a.B; ++b) {
        float v_dir[3] = {0.f, 0.f, 0.f};
        for (uint32_t c = 0; c < a.C; ++c) {
            int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            if (a.row_map) {
                row = a.row_map[row];
                if (row < 0) continue;
            }
            if (row_dead(a, row)) continue;
            const float vc[3] = {load_vc(a, row, 0), load_vc(a, row, 1), load_vc(a, row, 2)};
            float vd[3] = {0.f, 0.f, 0.f};
            sh3_row_vjp<DEG, WANT_MEANS>(a, b, c, g, (int64_t)g, vc, vec, vco, vd);
            v_dir[0] += vd[0]; v_dir[1] += vd[1]; v_dir[2] += vd[2];
            if constexpr (WANT_MEANS) {
                if (a.v_dirs) {
                    float *o = a.v_dirs + row * 3;
                    o[0] = vd[0]; o[1] = vd[1]; o[2] = vd[2];
                }
            }
        }
        if constexpr (WANT_MEANS) {
            if (a.v_means) {
                float *vm = a.v_means + ((size_t)b * a.N + g) * 3; // [B,N,3]: one thread per (b, g) -> plain store
                vm[0] = v_dir[0]; vm[1] = v_dir[1]; vm[2] = v_dir[2];
            }
        }
    }
    store_row<NF>(a.v_coeffs + (size_t)g * a.K * 3, vec, vco, a.K * 3, true);
}

// The same walk with the coefficient rows of a wave (Gaussians g0 .. g0 + 63: one contiguous block of [N, K, 3]) moved as
// wave-cooperative tiles. The coefficient row is never held in registers: the 16 sums w_k = coeffs[k] . v_colour that the
// mean gradient needs are formed straight from the lane's LDS row. MULTI = more than one image: the gradient row accumulates
// in registers over the images; otherwise it is the outer product Y (x) v_colour of the single live row, written straight
// into the lane's tile row once the coefficients in it have been consumed.
// WANT_COEFFS = false: only the mean / direction gradients (the several-image case runs as two launches, see launch_sh3_bwd).
// waves_per_eu(3): the tile (13 KiB per wave at degree 3) allows three workgroups per CU; without the hint the several-image
// variant takes 172 VGPRs, four more than three waves per SIMD leave.
template <int DEG, bool WANT_MEANS, bool MULTI, bool WANT_COEFFS = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
sh3_bwd_tiled_kernel(const ShArgs a)
{
    constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3, Q = ShTile<NF>::Q, STRIDE = ShTile<NF>::STRIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_smem[];
    const int64_t gi    = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool have     = gi < (int64_t)a.N;
    const uint32_t g    = (uint32_t)gi;
    v4f *tile           = reinterpret_cast<v4f *>(sh_smem + (threadIdx.x >> 6) * ShTile<NF>::kBytesPerWave);
    v4f *mine           = tile + lane * STRIDE;
    const uint64_t have_mask = __builtin_amdgcn_ballot_w64(have);
    const size_t block0 = (size_t)(gi - lane) * a.K * 3;
    const uint32_t row_q = a.K * 3u / 4u;
    const uint32_t n_img = MULTI ? a.B * a.C : 1u;

    if constexpr (WANT_MEANS) {
        // the coefficient row is needed (for d/d mean) iff the Gaussian is live in some image
        bool any = false;
        if (have)
            for (uint32_t i = 0; i < n_img && !any; ++i) {
                int64_t row = (int64_t)i * a.N + g;
                if (a.row_map) row = a.row_map[row];
                any = row >= 0 && !row_dead(a, row);
            }
        const uint64_t any_mask = __builtin_amdgcn_ballot_w64(any);
        if (any_mask) {
            tile_load<NF>(a.coeffs + block0, row_q, any_mask, tile, lane);
            wave_lds_sync();
        }
    }
    float vco[MULTI ? NF : 1];
#pragma unroll
    for (int i = 0; i < (MULTI ? NF : 1); ++i) vco[i] = 0.0f;
    float Y[NB], vc[3] = {0.f, 0.f, 0.f}; // !MULTI: the live row's basis and cotangent, for the outer product at the end
#pragma unroll
    for (int k = 0; k < NB; ++k) Y[k] = 0.0f;
    for (uint32_t b = 0; have && b < (MULTI ? a.B : 1u); ++b) {
        float v_dir[3] = {0.f, 0.f, 0.f};
        for (uint32_t c = 0; c < (MULTI ? a.C : 1u); ++c) {
            int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            if (a.row_map) {
                row = a.row_map[row];
                if (row < 0) continue;
            }
            if (row_dead(a, row)) continue;
            vc[0] = load_vc(a, row, 0); vc[1] = load_vc(a, row, 1); vc[2] = load_vc(a, row, 2);
            float d[3];
            view_dir(a, b, c, g, d);
            const float inv = safe_inv_norm(d);
            const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
            if constexpr (WANT_MEANS) {
                float w[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) w[k] = 0.0f;
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const v4f v = mine[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * i + j < NF) w[(4 * i + j) / 3] = fmaf(v[j], vc[(4 * i + j) % 3], w[(4 * i + j) / 3]);
                }
                float Yx[NB], Yy[NB], Yz[NB];
                sh_bases<true>(DEG, x, y, z, Y, Yx, Yy, Yz);
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    gx = fmaf(Yx[k], w[k], gx); gy = fmaf(Yy[k], w[k], gy); gz = fmaf(Yz[k], w[k], gz);
                }
                const float dot = gx * x + gy * y + gz * z; // through the normalisation: (g - (g.n) n) / |d|
                const float vd[3] = {(gx - dot * x) * inv, (gy - dot * y) * inv, (gz - dot * z) * inv};
                v_dir[0] += vd[0]; v_dir[1] += vd[1]; v_dir[2] += vd[2];
                if (a.v_dirs) {
                    float *o = a.v_dirs + row * 3;
                    o[0] = vd[0]; o[1] = vd[1]; o[2] = vd[2];
                }
            } else {
                sh_bases<false>(DEG, x, y, z, Y, nullptr, nullptr, nullptr);
            }
            if constexpr (MULTI && WANT_COEFFS) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    vco[3 * k] = fmaf(Y[k], vc[0], vco[3 * k]);
                    vco[3 * k + 1] = fmaf(Y[k], vc[1], vco[3 * k + 1]);
                    vco[3 * k + 2] = fmaf(Y[k], vc[2], vco[3 * k + 2]);
                }
            }
        }
        if constexpr (WANT_MEANS) {
            if (a.v_means) {
                float *vm = a.v_means + ((size_t)b * a.N + g) * 3; // [B,N,3]: one thread per (b, g) -> plain store
                vm[0] = v_dir[0]; vm[1] = v_dir[1]; vm[2] = v_dir[2];
            }
        }
    }
    if constexpr (!WANT_COEFFS) return;
    // the lane's gradient row -> its tile row (which only this lane has read) -> memory, as one block per wave
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        v4f v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 4 * i + j;
            if constexpr (MULTI) v[j] = f < NF ? vco[f < NF ? f : 0] : 0.0f;
            else v[j] = f < NF ? Y[(f < NF ? f : 0) / 3] * vc[f % 3] : 0.0f; // no live row: vc = 0
        }
        mine[i] = v;
    }
    tile_flush_rows<NF>(a.v_coeffs + block0, row_q, have_mask, tile, lane);
}

template <int DEG>
static void launch_sh3_bwd(const ShArgs &a, hipStream_t s)
{
    constexpr int NF = (DEG + 1) * (DEG + 1) * 3;
    const bool want_means = a.v_means || a.v_dirs;
    if (a.nnz < 0 || a.row_map) {
        const dim3 grid((uint32_t)ceil_div((int64_t)a.N, 256));
        if (DEG < 4 && ((a.K * 3u) & 3u) == 0u) { // 16-byte rows: wave-cooperative tiles
            const size_t smem = 4 * ShTile<NF>::kBytesPerWave;
            const bool multi  = a.B * a.C > 1;
            if (want_means && multi) {
                // One launch although it needs 172 VGPRs (the gradient row accumulates in 48 registers across the images):
                // split into a mean-gradient launch and a coefficient-gradient launch (80 / 96 VGPRs) it was slower, 0.77
                // instead of 0.62 ms on c4 - both halves read the cotangents out of the 36-byte gradient rows again, and
                // that traffic (576 MB at 4 M Gaussians x 4 images), not occupancy, is what bounds the kernel.
                sh3_bwd_tiled_kernel<DEG < 4 ? DEG : 0, true, true><<<grid, dim3(256), smem, s>>>(a);
            } else if (want_means) sh3_bwd_tiled_kernel<DEG < 4 ? DEG : 0, true, false><<<grid, dim3(256), smem, s>>>(a);
            else if (multi) sh3_bwd_tiled_kernel<DEG < 4 ? DEG : 0, false, true><<<grid, dim3(256), smem, s>>>(a);
            else sh3_bwd_tiled_kernel<DEG < 4 ? DEG : 0, false, false><<<grid, dim3(256), smem, s>>>(a);
        } else if (want_means) sh3_bwd_dense_kernel<DEG, true><<<grid, dim3(256), 0, s>>>(a);
        else sh3_bwd_dense_kernel<DEG, false><<<grid, dim3(256), 0, s>>>(a);
    } else {
        const dim3 grid((uint32_t)ceil_div(a.nnz, 256));
        if (want_means) sh3_bwd_packed_kernel<DEG, true><<<grid, dim3(256), 0, s>>>(a);
        else sh3_bwd_packed_kernel<DEG, false><<<grid, dim3(256), 0, s>>>(a);
    }
}

// ---- assemble_proj_features_unpacked_fwd: [SH colours | extra signals | depth] rows in one pass -----------------------
// (reference SphericalHarmonicsCUDA.cu:1100-1250). One thread per dense row (b, c, g); DEG >= 0 selects the D = 3
// row loader, DEG < 0 the generic channel loop. Rows whose mask is clear get zero colours (extra / depth columns are
// still written) and leave relu_mask untouched, like the reference kernel.
struct AssembleArgs {
    uint32_t Dc, E, width;
    int color_post, extra_post, has_depth, extra_has_c; // post: 0 none, 1 x + 0.5, 2 max(x + 0.5, 0)
    const float *extra, *depths;                          // depths NULL with has_depth: the column is zero
    float *out;
    uint8_t *relu_mask;
};

__device__ __forceinline__ float assemble_post(float v, int post)
{
    if (post == 1) return v + 0.5f;
    if (post == 2) return fmaxf(v + 0.5f, 0.0f);
    return v;
}

template <int DEG>
__global__ void __launch_bounds__(256) assemble_features_kernel(const ShArgs a, const AssembleArgs f)
{
    const int64_t rows = (int64_t)a.B * a.C * a.N;
    const int64_t row  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const uint32_t g = (uint32_t)(row % a.N), c = (uint32_t)((row / a.N) % a.C), b = (uint32_t)(row / ((int64_t)a.N * a.C));
    float *dst = f.out + row * f.width;
    if (a.masks && !a.masks[row]) {
        for (uint32_t ch = 0; ch < f.Dc; ++ch) dst[ch] = 0.0f;
    } else {
        float d[3];
        view_dir(a, b, c, g, d);
        const float inv = safe_inv_norm(d);
        uint8_t *rm = f.relu_mask ? f.relu_mask + row * f.Dc : nullptr;
        if constexpr (DEG >= 0) {
            constexpr int NB = (DEG + 1) * (DEG + 1), NF = NB * 3;
            float Y[NB];
            sh_bases<false>(DEG, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
            float co[NF];
            load_row<NF>(a.coeffs + (size_t)g * a.K * 3, ((a.K * 3u) & 3u) == 0u, co);
            float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                r[0] += Y[k] * co[3 * k];
                r[1] += Y[k] * co[3 * k + 1];
                r[2] += Y[k] * co[3 * k + 2];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float v = assemble_post(r[ch], f.color_post);
                dst[ch] = v;
                if (rm) rm[ch] = v > 0.0f;
            }
        } else {
            float Y[kMaxBases];
            sh_bases<false>(a.degree, d[0] * inv, d[1] * inv, d[2] * inv, Y, nullptr, nullptr, nullptr);
            const int nb = (a.degree + 1) * (a.degree + 1);
            for (uint32_t ch = 0; ch < f.Dc; ++ch) {
                const float *co = a.coeffs + ((size_t)g * a.K) * f.Dc + ch;
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < kMaxBases; ++k)
                    if (k < nb) acc += Y[k] * co[(size_t)k * f.Dc];
                const float v = assemble_post(acc, f.color_post);
                dst[ch] = v;
                if (rm) rm[ch] = v > 0.0f;
            }
        }
    }
    if (f.E) {
        const float shift = f.extra_post == 1 ? 0.5f : 0.0f;
        const float *src  = f.extra + (f.extra_has_c ? (size_t)row : (size_t)b * a.N + g) * f.E;
        for (uint32_t e = 0; e < f.E; ++e) dst[f.Dc + e] = src[e] + shift;
    }
    if (f.has_depth) dst[f.Dc + f.E] = f.depths ? f.depths[row] : 0.0f;
}

static int check_sh(const char *fn, int degree, uint32_t K, uint32_t D, const float *means, const float *viewmats,
                    const float *coeffs, int64_t nnz, const int64_t *bi, const int64_t *ci, const int64_t *gi)
{
    GSX_REQUIRE(degree >= 0 && degree <= 4, "%s: degrees_to_use must be in [0,4], got %d", fn, degree);
    GSX_REQUIRE((uint32_t)((degree + 1) * (degree + 1)) <= K, "%s: coeffs K=%u too small for degree %d", fn, K, degree);
    GSX_REQUIRE(D >= 1, "%s: D must be >= 1", fn);
    GSX_REQUIRE(means && viewmats && coeffs, "%s: null input", fn);
    GSX_REQUIRE(nnz < 0 || nnz == 0 || (bi && ci && gi), "%s: packed mode needs batch/camera/gaussian ids", fn);
    return GSX_OK;
}

} // namespace gsx

using namespace gsx;

static int sh_fwd_impl(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                       const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                       const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                       int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, int post, float *colors,
                       const float *r_means2d, const float *r_conics, const float *r_opacities, float *splat_rows, void *stream);
extern "C" int gsx_sh_fwd(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                          const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                          const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                          int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, int post, float *colors,
                          void *stream)
{
    return sh_fwd_impl(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, B, C, N, nnz,
                       coeffs_gathered, K, D, radii, post, colors, nullptr, nullptr, nullptr, nullptr, stream);
}
// gsx_sh_fwd (D == 3) that also writes the compositing kernels' 48-byte array-of-structures row of every live row
// (include/gsplat_amd.h): means2d [rows,2], conics [rows,3], opacities [rows] are the projection's outputs for the same rows.
extern "C" int gsx_sh_fwd_rows(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                               const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                               const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                               int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, int post, float *colors,
                               const float *means2d, const float *conics, const float *opacities, float *splat_rows, void *stream)
{
    GSX_REQUIRE(D == 3, "gsx_sh_fwd_rows: the rows hold three colours; D is %u", D);
    GSX_REQUIRE(means2d && conics && opacities && splat_rows, "gsx_sh_fwd_rows: null row input / output");
    GSX_REQUIRE((reinterpret_cast<uintptr_t>(splat_rows) & 15u) == 0, "gsx_sh_fwd_rows: rows must be 16-byte aligned");
    return sh_fwd_impl(degrees_to_use, means, viewmats, coeffs, masks, batch_ids, camera_ids, gaussian_ids, B, C, N, nnz,
                       coeffs_gathered, K, D, radii, post, colors, means2d, conics, opacities, splat_rows, stream);
}
static int sh_fwd_impl(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                       const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                       const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                       int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, int post, float *colors,
                       const float *r_means2d, const float *r_conics, const float *r_opacities, float *splat_rows, void *stream)
{
    const int64_t rows = nnz >= 0 ? nnz : (int64_t)B * C * N;
    if (rows == 0) return GSX_OK;
    int rc = check_sh("gsx_sh_fwd", degrees_to_use, K, D, means, viewmats, coeffs, nnz, batch_ids, camera_ids, gaussian_ids);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(colors, "gsx_sh_fwd: null output");
    ShArgs a{};
    a.degree = degrees_to_use; a.means = means; a.viewmats = viewmats; a.coeffs = coeffs; a.masks = masks;
    a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.B = B; a.C = C; a.N = N; a.K = K; a.D = D; a.nnz = nnz; a.coeffs_gathered = coeffs_gathered; a.colors = colors;
    a.radii = radii; a.post = post;
    a.r_means2d = r_means2d; a.r_conics = r_conics; a.r_opacities = r_opacities; a.splat_rows = splat_rows;
    hipStream_t s = (hipStream_t)stream;
    if (D == 3 && nnz < 0 && (uint64_t)B * C > 1 && degrees_to_use < 4 && ((K * 3u) & 3u) == 0u) {
        // several images over the same Gaussians: Gaussian-major walk, one coefficient fetch per Gaussian
        const dim3 grid((uint32_t)ceil_div((int64_t)N, 256));
        switch (degrees_to_use) {
        case 0: sh3_fwd_gaussian_major_kernel<0><<<grid, dim3(256), 4 * ShTile<3>::kBytesPerWave, s>>>(a); break;
        case 1: sh3_fwd_gaussian_major_kernel<1><<<grid, dim3(256), 4 * ShTile<12>::kBytesPerWave, s>>>(a); break;
        case 2: sh3_fwd_gaussian_major_kernel<2><<<grid, dim3(256), 4 * ShTile<27>::kBytesPerWave, s>>>(a); break;
        default: sh3_fwd_gaussian_major_kernel<3><<<grid, dim3(256), 4 * ShTile<48>::kBytesPerWave, s>>>(a); break;
        }
    } else if (D == 3) {
        const dim3 grid((uint32_t)ceil_div(rows, 256));
        switch (degrees_to_use) {
        case 0: sh3_fwd_kernel<0><<<grid, dim3(256), 4 * ShTile<3>::kBytesPerWave, s>>>(a); break;
        case 1: sh3_fwd_kernel<1><<<grid, dim3(256), 4 * ShTile<12>::kBytesPerWave, s>>>(a); break;
        case 2: sh3_fwd_kernel<2><<<grid, dim3(256), 4 * ShTile<27>::kBytesPerWave, s>>>(a); break;
        case 3: sh3_fwd_kernel<3><<<grid, dim3(256), 4 * ShTile<48>::kBytesPerWave, s>>>(a); break;
        default: sh3_fwd_kernel<4><<<grid, dim3(256), 0, s>>>(a); break;
        }
    } else {
        sh_fwd_kernel<<<dim3((uint32_t)ceil_div(rows * D, 256)), dim3(256), 0, s>>>(a);
    }
    return check_launch("sh_fwd");
}

extern "C" int gsx_sh_bwd(int degrees_to_use, const float *means, const float *viewmats, const float *coeffs,
                          const uint8_t *masks, const int64_t *batch_ids, const int64_t *camera_ids,
                          const int64_t *gaussian_ids, uint32_t B, uint32_t C, uint32_t N, int64_t nnz,
                          int coeffs_gathered, uint32_t K, uint32_t D, const int32_t *radii, const float *post_colors,
                          const float *v_colors, uint32_t v_colors_stride, const int32_t *row_map, float *v_coeffs,
                          float *v_means, float *v_dirs, void *stream)
{
    int rc = check_sh("gsx_sh_bwd", degrees_to_use, K, D, means, viewmats, coeffs, nnz, batch_ids, camera_ids, gaussian_ids);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(v_coeffs, "gsx_sh_bwd: null v_coeffs");
    ShArgs a{};
    a.degree = degrees_to_use; a.means = means; a.viewmats = viewmats; a.coeffs = coeffs; a.masks = masks;
    a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.B = B; a.C = C; a.N = N; a.K = K; a.D = D; a.nnz = nnz; a.coeffs_gathered = coeffs_gathered;
    a.v_colors = v_colors; a.v_coeffs = v_coeffs; a.v_means = v_means; a.v_dirs = v_dirs;
    a.radii = radii; a.post_colors = post_colors; a.vc_stride = v_colors_stride ? v_colors_stride : D;
    a.row_map = (nnz >= 0 && D == 3 && !coeffs_gathered) ? row_map : nullptr; // Gaussian-major walk of packed rows
    a.atomic_coeffs = (B * C) > 1;
    if (D == 3 && (nnz < 0 ? (int64_t)N > 0 : nnz > 0)) {
        GSX_REQUIRE(v_colors || (nnz < 0 && (int64_t)B * C == 0), "gsx_sh_bwd: null v_colors");
        hipStream_t s = (hipStream_t)stream;
        switch (degrees_to_use) {
        case 0: launch_sh3_bwd<0>(a, s); break;
        case 1: launch_sh3_bwd<1>(a, s); break;
        case 2: launch_sh3_bwd<2>(a, s); break;
        case 3: launch_sh3_bwd<3>(a, s); break;
        default: launch_sh3_bwd<4>(a, s); break;
        }
        return check_launch("sh_bwd");
    }
    if (nnz < 0) {
        if ((int64_t)N * D == 0) return GSX_OK;
        GSX_REQUIRE(v_colors || (int64_t)B * C == 0, "gsx_sh_bwd: null v_colors");
        sh_bwd_dense_kernel<<<dim3((uint32_t)ceil_div((int64_t)N * D, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    } else {
        if (nnz == 0) return GSX_OK;
        GSX_REQUIRE(v_colors, "gsx_sh_bwd: null v_colors");
        sh_bwd_packed_kernel<<<dim3((uint32_t)ceil_div(nnz * D, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    }
    return check_launch("sh_bwd");
}

extern "C" int gsx_assemble_features_fwd(int degrees_to_use, uint32_t B, uint32_t C, uint32_t N, uint32_t K, uint32_t Dc,
                                         uint32_t E, int color_post, int extra_post, int has_depth, int extra_has_c,
                                         const float *means, const float *viewmats, const float *coeffs,
                                         const float *extra, const float *depths, const uint8_t *masks, float *out,
                                         uint8_t *relu_mask, void *stream)
{
    const int64_t rows = (int64_t)B * C * N;
    if (rows == 0) return GSX_OK;
    int rc = check_sh("gsx_assemble_features_fwd", degrees_to_use, K, Dc, means, viewmats, coeffs, -1, nullptr, nullptr, nullptr);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(out, "gsx_assemble_features_fwd: null output");
    GSX_REQUIRE(color_post >= 0 && color_post <= 2 && extra_post >= 0 && extra_post <= 2,
                "gsx_assemble_features_fwd: post ops must be 0 (none), 1 (shift) or 2 (shift + relu)");
    GSX_REQUIRE(E == 0 || extra, "gsx_assemble_features_fwd: extra is required when E > 0");
    GSX_REQUIRE(!relu_mask || color_post == 2, "gsx_assemble_features_fwd: relu_mask needs color_post = 2");
    ShArgs a{};
    a.degree = degrees_to_use; a.means = means; a.viewmats = viewmats; a.coeffs = coeffs; a.masks = masks;
    a.B = B; a.C = C; a.N = N; a.K = K; a.D = Dc; a.nnz = -1; a.coeffs_gathered = 1;
    AssembleArgs f{};
    f.Dc = Dc; f.E = E; f.width = Dc + E + (has_depth ? 1u : 0u);
    f.color_post = color_post; f.extra_post = extra_post; f.has_depth = has_depth; f.extra_has_c = extra_has_c;
    f.extra = extra; f.depths = depths; f.out = out; f.relu_mask = relu_mask;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((uint32_t)ceil_div(rows, 256)), block(256);
    if (Dc == 3) {
        switch (degrees_to_use) {
        case 0: assemble_features_kernel<0><<<grid, block, 0, s>>>(a, f); break;
        case 1: assemble_features_kernel<1><<<grid, block, 0, s>>>(a, f); break;
        case 2: assemble_features_kernel<2><<<grid, block, 0, s>>>(a, f); break;
        case 3: assemble_features_kernel<3><<<grid, block, 0, s>>>(a, f); break;
        default: assemble_features_kernel<4><<<grid, block, 0, s>>>(a, f); break;
        }
    } else {
        assemble_features_kernel<-1><<<grid, block, 0, s>>>(a, f);
    }
    return check_launch("assemble_features_fwd");
}
