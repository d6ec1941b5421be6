// 3DGS front-to-back alpha compositing, forward, WIDE colour rows (5 .. 32 channels per launch) on the matrix cores (gfx950):
// the forward sibling of raster3d_bwd_m.hip. Launched by gsx_raster3d_fwd (raster3d_fwd.hip) for 16 x 16 tiles; replaces the
// same reference kernel, RasterizeToPixels3DGSSerialBatchFwd.cu:41-297 (per-sample math RasterizeToPixels3DGSDevice.cuh:44-103).
//
// With D colour channels the render of a pixel is  out[p][k] = sum over Gaussians g of  wgt(p, g) * c[g][k],  wgt = alpha T:
// a MATRIX PRODUCT  [pixels x Gaussians] . [Gaussians x D]  whose right-hand side is the staged colour table. The four-wave
// kernel spends D FMAs and D / 4 broadcast LDS reads per (pixel, Gaussian) pair on it (0.64 ms at 32 channels on c3 against
// 0.17 at three). Here the pixel walk only produces the scalar wgt and parks it in a wave-private LDS matrix; every SIXTEEN
// surviving Gaussians the wave multiplies with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact fp32): M = 16 pixels
// (four blocks cover the wave's 8 x 8 quadrant), N = 16 channels (one or two blocks), K = 4 Gaussians per instruction - 32
// instructions per 16 Gaussians at 32 channels, on the MFMA pipe, which runs beside the vector ALU. The accumulators ARE the
// render: they stay in registers for the whole tile (pixel 16 pb + 4 (l >> 4) + i, channel 16 nb + (l & 15)) and are written
// once at the end. The pixel walk is the branch-free body of the four-wave kernel (`thr` = the pixel's alpha threshold, +inf
// once it is done). Groups are sixteen CONSECUTIVE list entries, so the order of the partial sums depends on the list alone
// (see the walk): the reference's front-to-back order up to the association inside a group.
#include <cstdlib>

#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NB> // 16-column blocks of colour channels per launch: 1 (<= 16 channels) or 2 (<= 32)
struct FwdMCfg {
    static constexpr int CHP   = 16 * NB;
    static constexpr int CP    = CHP + 4; // floats per staged colour row (+ 4: a wave's b128 stores spread over the banks)
    static constexpr int BATCH = 64;      // staged Gaussians per batch: one per lane of the culling test
    static constexpr int SLOTS = 16;      // Gaussians per multiplication (the K of four MFMA steps)
    static constexpr int WP    = 68;      // floats per parked row: 64 pixels + 4
    static constexpr size_t smem = (size_t)BATCH * (sizeof(StagedRow) + sizeof(float4) + sizeof(float) * CP) + sizeof(float) * 4 * SLOTS * WP;
};

template <int NB>
__global__ void __launch_bounds__(256) raster3d_fwd_m_kernel(const Raster3DArgs a)
{
    using Cfg           = FwdMCfg<NB>;
    constexpr int CHP   = Cfg::CHP;
    constexpr int CP    = Cfg::CP;
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int WP    = Cfg::WP;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    StagedRow *s_st = reinterpret_cast<StagedRow *>(smem_raw);         // e-form of the exponent (raster3d.hpp); its colour fields are unused
    float4 *s_cull  = reinterpret_cast<float4 *>(s_st + BATCH);        // mean - tile centre, half extents of alpha >= 1/255
    float *s_col    = reinterpret_cast<float *>(s_cull + BATCH);       // [BATCH][CP] colours, zero padded to CHP
    float *s_w      = s_col + BATCH * CP;                              // [4 waves][SLOTS][WP]: wgt per (slot, pixel)

    TileCtx tc;
    if (!tile_context(a, blockIdx.x, tc)) return; // uniform for the whole workgroup
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lx, ly;
    tile_pixel(tid, 16u, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly); // output row, -1 = this lane renders nothing
    const bool inside  = prow >= 0;
    const size_t pix   = inside ? (size_t)prow : 0;
    const float tile_cx = (float)(tc.tile_x * 16u) + 8.0f, tile_cy = (float)(tc.tile_y * 16u) + 8.0f;
    const float pu = (float)lx - 7.5f, pv = (float)ly - 7.5f; // this lane's pixel centre relative to the tile centre (exact)
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tc.image_id * a.cdim + a.ch_off : nullptr;

    // masked-off tile: background colour, zero alpha, last_id 0 (reference Fwd.cu:141-159)
    if (a.masks && !a.masks[(size_t)tc.image_id * tiles_per_image + tc.tile_id]) {
        if (inside) {
            for (uint32_t k = 0; k < a.nch; ++k) a.render_colors[pix * a.cdim + a.ch_off + k] = bg ? bg[k] : 0.0f;
            if (a.first_chunk) {
                a.render_alphas[pix] = 0.0f;
                a.last_ids[pix]      = 0;
            }
        }
        return;
    }

    const int32_t range_start = tc.range_start, range_end = tc.range_end;
    const int32_t n_batches   = range_end > range_start ? (range_end - range_start + BATCH - 1) / BATCH : 0;

    float T          = 1.0f;
    uint32_t cur_idx = 0u;
    float thr        = inside ? kAlphaThreshold : INFINITY; // alpha threshold of this pixel; +inf = done (or not rendered)
    f32x4 acc[4][NB]; // the render, in the MFMA result layout: pixel 16 pb + 4 bk + i of this wave, channel 16 nb + bj
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[pb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float *s_ww  = s_w + wave * (SLOTS * WP); // this wave's parked matrix: wgt at [slot][pixel]
    const int bj = (int)(lane & 15u), bk = (int)(lane >> 4);

    for (int32_t b = 0; b < n_batches; ++b) {
        // block-wide early out: every pixel of the tile finished. Also fences the reuse of the staged tables.
        if (__syncthreads_count(!(thr < INFINITY)) == (int)blockDim.x) break;
        const int32_t batch_start = range_start + BATCH * b;
        const int32_t batch_size  = min(BATCH, range_end - batch_start);
        { // staging by the whole workgroup: thread (entry s, part) takes a quarter of entry s's colour row; part 0 the geometry
            const int s = (int)(tid & 63u), part = (int)(tid >> 6);
            const int32_t idx = batch_start + s;
            if (idx < range_end) {
                const int32_t g = a.flatten_ids[idx];
                if (part == 0) {
                    const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                    const float opac = a.opacities[g];
                    const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                    const float ax = xy.x - tile_cx, ay = xy.y - tile_cy;
                    v4f p0;
                    float nA, nB, nC;
                    stage_gaussian_f(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
                    const float2 he = cull_half_extent(opac, ca, cb, cc);
                    s_cull[s]  = make_float4(ax, ay, he.x, he.y);
                    s_st[s].p0 = p0;
                    s_st[s].p1 = v4f{nA, nB, nC, 0.0f};
                }
                constexpr int Q = CHP / 4; // channels per part
                const float *c  = a.colors + (size_t)g * a.cdim + a.ch_off + Q * part;
                float cv[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) cv[k] = (Q * part + k < (int)a.nch) ? c[k] : 0.0f;
                f32x4 *dst = reinterpret_cast<f32x4 *>(s_col + s * CP + Q * part);
#pragma unroll
                for (int h = 0; h < Q / 4; ++h) dst[h] = f32x4{cv[4 * h], cv[4 * h + 1], cv[4 * h + 2], cv[4 * h + 3]};
            } else if (b == 0) { // rows behind a short first batch were never written: zero them once (0 x NaN is NaN)
                constexpr int Q = CHP / 4;
                f32x4 *dst = reinterpret_cast<f32x4 *>(s_col + s * CP + Q * part);
#pragma unroll
                for (int h = 0; h < Q / 4; ++h) dst[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();

        // the rectangle of the wave's pixels that are still open (it shrinks as pixels saturate; refreshed per batch)
        const WaveRect rect = wave_pixel_rect(thr < INFINITY, pu, pv);
        bool hit = false;
        if ((int32_t)lane < batch_size) {
            const float4 cu = s_cull[lane];
            hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
            if (hit) hit = rect_reaches_level(s_st[lane].p0, s_st[lane].p1, cu.x, cu.y, rect); // exact second stage
        }
        const uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
        // The staged entries are taken in FIXED groups of sixteen consecutive list entries (a group without a survivor is
        // skipped; inside a group the culled entries park zeros): which partial sums are formed, and in which order, then
        // depends on the list alone - not on what the cull let through, i.e. not on the state of the wave's other pixels. A
        // pixel's value is the same whatever its neighbours are (a sparse pixel set renders bit for bit what the dense image
        // shows there: tests/test_gpu_sparse.py), at the price of multiplying a few all-zero rows.
        for (int grp = 0; grp < BATCH / SLOTS; ++grp) {
            const uint32_t gm = (uint32_t)(todo >> (SLOTS * grp)) & 0xFFFFu; // survivors of this group (wave-uniform)
            if (gm == 0u) continue;
            // wave-level early termination: a finished quadrant stops evaluating (it still takes part in staging and barriers)
            if (__builtin_amdgcn_ballot_w64(thr < INFINITY) == 0ull) break;
            // the pixel walk: branch-free body of raster3d_fwd.hip; parks wgt = alpha T (0 where the Gaussian is not blended)
#pragma unroll 4
            for (int k = 0; k < SLOTS; ++k) {
                float wgt = 0.0f;
                if ((gm >> k) & 1u) { // wave-uniform
                    const int32_t t = SLOTS * grp + k;
                    const v4f p0 = s_st[t].p0;
                    const v4f p1 = s_st[t].p1;
                    const float e     = staged_f(p0, p1.x, p1.y, p1.z, pu, pv);
                    const float alpha = fminf(kMaxAlpha, __builtin_amdgcn_exp2f(e));
                    const bool ok     = !(e > p0.w) && !(alpha < thr); // e > lo <=> sigma < 0
                    const float next_T = fmaf(-T, alpha, T);
                    const bool low     = next_T <= kTransmittanceThresh; // saturated: this Gaussian is excluded
                    const bool sat = ok && low, take = ok && !sat;
                    const float at = take ? alpha : 0.0f;
                    wgt     = at * T;
                    cur_idx = take ? (uint32_t)(batch_start + t) : cur_idx;
                    T       = fmaf(-T, at, T); // == next_T where the Gaussian is blended, T exactly where not
                    thr     = sat ? INFINITY : thr;
                }
                s_ww[k * WP + (int)lane] = wgt;
            }
            wave_lds_sync();
            // out[pixel][channel] += wgt[pixel][slot] . colour[slot][channel]: k-step s contracts slots {s, 4 + s, 8 + s, 12 + s};
            // lane l supplies k = l >> 4 (slot 4 bk + s): A = wgt of pixel 16 pb + bj, B = colour of channel 16 nb + bj. Entries
            // behind the end of the list have zero weights and meet whatever the table holds there (finite: staged earlier or
            // zero-filled below)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int slot = 4 * bk + s;
                const int t_b  = SLOTS * grp + slot;
                float bcol[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bcol[nb] = s_col[t_b * CP + 16 * nb + bj];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const float av = s_ww[slot * WP + 16 * pb + bj];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[pb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bcol[nb], acc[pb][nb], 0, 0, 0);
                }
            }
            wave_lds_sync(); // the next group overwrites the parked rows
        }
    }

    // epilogue. alpha / last contributor: the pixel's own lane. Colours: lane (bj, bk) holds channels 16 nb + bj of the pixels
    // 16 pb + 4 bk + i of this wave; their final transmittance (for the background term) comes through the parked region.
    if (inside && a.first_chunk) {
        a.render_alphas[pix] = 1.0f - T;
        a.last_ids[pix]      = (int32_t)cur_idx;
    }
    s_ww[lane] = T;
    wave_lds_sync();
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 16 * pb + 4 * bk + i; // pixel p of this wave = its lane p
            const uint32_t plx = ((wave & 1u) << 3) | (uint32_t)(p & 7), ply = ((wave >> 1) << 3) | (uint32_t)(p >> 3);
            const int64_t pr = pixel_row(a, tc, blockIdx.x, plx, ply);
            if (pr < 0) continue;
            const float Tp = s_ww[p];
            float *dst     = a.render_colors + (size_t)pr * a.cdim + a.ch_off;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int ch = 16 * nb + bj;
                if (ch < (int)a.nch) dst[ch] = bg ? (acc[pb][nb][i] + Tp * bg[ch]) : acc[pb][nb][i];
            }
        }
}

// GSX_RASTER3D_FWD_WIDE=q keeps the four-wave kernel for wide colour rows (A/B; read once per process)
static bool fwd_m_enabled()
{
    static const bool on = [] {
        const char *e = getenv("GSX_RASTER3D_FWD_WIDE");
        return !(e && (e[0] == 'q' || e[0] == 'Q' || e[0] == '0'));
    }();
    return on;
}
static uint32_t fwd_m_min_channels() // GSX_RASTER3D_FWD_WIDE_MIN overrides (A/B)
{
    static const uint32_t v = [] {
        const char *e = getenv("GSX_RASTER3D_FWD_WIDE_MIN");
        const int x   = e ? atoi(e) : 17;
        return (uint32_t)(x > 5 ? x : 5);
    }();
    return v;
}
bool raster3d_fwd_m_applies(const Raster3DArgs &a)
{
    // from 17 channels on (two column blocks): at 5 / 8 / 12 channels the four-wave kernel is the faster one (0.25 / 0.28 / 0.33
    // against 0.34 / 0.35 / 0.38 ms at c3), at 16 they tie (0.38), at 32 this one wins (0.445 against 0.653; profiles/r09_ab.md)
    return fwd_m_enabled() && a.tile_size == 16 && a.nch >= fwd_m_min_channels() && a.nch <= 32 && a.seg_mode == 0 && a.seg_len == 0;
}
int raster3d_fwd_m_launch(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid = ((n_blocks + 7u) / 8u) * 8u; // xcd_remap needs a multiple of 8
    if (a.nch <= 16) raster3d_fwd_m_kernel<1><<<dim3(grid), dim3(256), FwdMCfg<1>::smem, stream>>>(a);
    else raster3d_fwd_m_kernel<2><<<dim3(grid), dim3(256), FwdMCfg<2>::smem, stream>>>(a);
    return check_launch("raster3d_fwd_m");
}

} // namespace gsx
