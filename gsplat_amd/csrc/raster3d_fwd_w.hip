// 3DGS front-to-back alpha compositing, forward, ONE WAVE PER TILE (gfx950): the sibling of the backward's variant W
// (raster3d_bwd.hip). Launched by gsx_raster3d_fwd (raster3d_fwd.hip) for <= 4 channels per launch and 16 x 16 tiles;
// replaces the same reference kernel, RasterizeToPixels3DGSSerialBatchFwd.cu:41-297 (per-sample math
// RasterizeToPixels3DGSDevice.cuh:44-103).
//
// The four-waves-per-tile kernel (raster3d_fwd.hip) pays for the fact that a tile belongs to four waves: the staged list is
// shared through workgroup barriers (a finished quadrant keeps staging and waiting), and the loop control of the walk -
// scalar instructions and three broadcast LDS reads - runs once per (wave, Gaussian) pair, 2.8 times per staged Gaussian on
// c3. Here a tile is ONE wave64 and a lane owns FOUR pixels, the same position in each 8 x 8 quadrant:
//   * staging is wave-private: 64 Gaussians per batch, one per lane, no __syncthreads anywhere; the rows of batch b + 1 and
//     the list entries of batch b + 2 are requested before batch b is walked (nothing else of the wave could hide the two
//     dependent global reads);
//   * the staging lane runs the two-stage cull of the wave-level test (raster3d.hpp) against the rectangles of the pixels
//     that are STILL OPEN in each quadrant - refreshed per batch from four ballots with scalar bit arithmetic (lane =
//     (qx, qy), so the row range is ctz / clz of the ballot and the column range those of its fold) - and keeps the 4-bit
//     answer in a register; the walk reads the staged 48-byte row ONCE per Gaussian and branches over the quadrants with
//     scalar bit tests;
//   * the pixel body is the branch-free one of the four-wave kernel: `thr` is the pixel's alpha threshold, +inf once it is
//     done; a quadrant whose pixels are all done leaves the scalar mask of open quadrants, the tile stops when it is empty;
//   * the tile's cost for the backward (its list up to the last contributor, tile_order.hip) is a by-product: one wave
//     maximum of the last-contributor indices the lanes hold anyway.
// LDS: 64 x 48 B per wave; registers are what bounds residency (state of four pixels + one prefetched row: 86 VGPRs).
//
// MEASURED AND NOT THE DEFAULT (round 5, profiles/r09_ab.md #1): parity-green on its first run, and 0.290 ms at c3 where the
// four-wave kernel takes 0.199 in the same harness - at 4, 5 and 6 waves per SIMD alike, with or without the shrinking
// rectangles. What the backward's variant W could absorb the forward cannot: a tile is now ONE instruction stream, a wave
// issues at most one instruction every ~5 cycles however idle its SIMD is (tools/issue_rate.hip), and per staged Gaussian the
// stream is ~100 issue slots plus an LDS round trip and up to nine branches - so the longest tile of c3 (486 entries to its
// last contributor against a mean of 160) needs 0.1 - 0.2 ms on its own, as long as the whole launch should take. The
// backward hides the same tail by starting the long tiles first (tile_order.hip), but the forward only learns a tile's cost by
// running it (the list lengths, 466 +- 30, say nothing); four waves per tile cut the critical path by the number of quadrants.
// Kept selectable (GSX_RASTER3D_FWD=w) and covered by tests/test_gpu_variants.py.
#include <cstdlib>

#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

#ifndef GSX_FWD_W_WAVES // waves per SIMD the register allocation aims at
#define GSX_FWD_W_WAVES 5
#endif
#ifndef GSX_FWD_W_RECT // 1: cull against the rectangle of the quadrant's OPEN pixels; 0: against the whole quadrant
#define GSX_FWD_W_RECT 1
#endif

// column / row range of the set bits of a quadrant ballot (lane = qy * 8 + qx); m != 0
__device__ __forceinline__ void open_range(uint64_t m, int &xmin, int &xmax, int &ymin, int &ymax)
{
    ymin = (int)(__builtin_ctzll(m) >> 3);
    ymax = (int)((63 - __builtin_clzll(m)) >> 3);
    uint32_t c = (uint32_t)m | (uint32_t)(m >> 32);
    c |= c >> 16;
    c = (c | (c >> 8)) & 0xFFu;
    xmin = (int)__builtin_ctz(c);
    xmax = 31 - (int)__builtin_clz(c);
}

// NQ = 4: one wave per tile. NQ = 2 (GSX_RASTER3D_FWD=h): one wave per HALF tile, two pixels per lane - workgroups 16 i + j and
// 16 i + 8 + j (j < 8: the same XCD, dispatched next to each other) are the upper and the lower half of tile slot 8 i + j.
template <int CH, int NQ>
__device__ __forceinline__ void raster3d_fwd_w_body(const Raster3DArgs &a)
{
    constexpr int BATCH = 64;
    static_assert(CH <= 4, "one staged row carries four colours");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    StagedRow *s_st = reinterpret_cast<StagedRow *>(smem_raw); // [BATCH]: e-form of the exponent + colours (raster3d.hpp)

    constexpr uint32_t kParts = 4u / (uint32_t)NQ; // waves per tile
    const uint32_t unit    = (blockIdx.x / (8u * kParts)) * 8u + (blockIdx.x & 7u);
    const uint32_t q_first = ((blockIdx.x >> 3) % kParts) * (uint32_t)NQ; // first quadrant of this wave
    TileCtx tc;
    if (!tile_context(a, unit, tc)) return;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t lane = threadIdx.x & 63u, qx = lane & 7u, qy = lane >> 3;
    // this lane's four pixels: (qx, qy) inside each quadrant; centres relative to the tile centre (multiples of 0.5: exact)
    const float pu[2] = {(float)qx - 7.5f, (float)qx + 0.5f}, pv[2] = {(float)qy - 7.5f, (float)qy + 0.5f};
    const float tile_cx = (float)(tc.tile_x * 16u) + 8.0f, tile_cy = (float)(tc.tile_y * 16u) + 8.0f;
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tc.image_id * a.cdim + a.ch_off : nullptr;
    auto row_of = [&](int q) -> int64_t { // output row of the lane's pixel in this wave's quadrant q, -1 = not rendered
        const uint32_t gq = q_first + (uint32_t)q;
        return pixel_row(a, tc, unit, ((gq & 1u) << 3) | qx, ((gq >> 1) << 3) | qy);
    };
    int32_t *cost_out = (NQ == 4 && a.tile_cost && !a.sp_active_tiles) ? a.tile_cost + (size_t)tc.image_id * tiles_per_image + tc.tile_id : nullptr;

    // masked-off tile: background colour, zero alpha, last_id 0 (reference Fwd.cu:141-159)
    if (a.masks && !a.masks[(size_t)tc.image_id * tiles_per_image + tc.tile_id]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t prow = row_of(q);
            if (prow < 0) continue;
            const size_t pix = (size_t)prow;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < (int)a.nch) a.render_colors[pix * a.cdim + a.ch_off + k] = bg ? bg[k] : 0.0f;
            if (a.first_chunk) {
                a.render_alphas[pix] = 0.0f;
                a.last_ids[pix]      = 0;
            }
        }
        if (cost_out && lane == 0) *cost_out = 0;
        return;
    }

    float T[NQ], thr[NQ], acc[NQ][CH];
    uint32_t cur_idx[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        T[q]       = 1.0f;
        thr[q]     = row_of(q) >= 0 ? kAlphaThreshold : INFINITY; // alpha threshold of the pixel; +inf = done (or not rendered)
        cur_idx[q] = 0u;
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[q][k] = 0.0f;
    }

    const int32_t range_start = tc.range_start, range_end = tc.range_end;
    const int32_t n_batches   = range_end > range_start ? (range_end - range_start + BATCH - 1) / BATCH : 0;

    struct Fetched { float2 xy; float opac, ca, cb, cc, cv[4]; };
    auto entry_of = [&](int32_t b) -> int32_t { // flatten id of this lane's entry of batch b, -1 = none
        const int32_t idx = range_start + BATCH * b + (int32_t)lane;
        return (b < n_batches && idx < range_end) ? a.flatten_ids[idx] : -1;
    };
    auto fetch = [&](int32_t g, Fetched &f) {
        if (g < 0) return;
        f.xy   = reinterpret_cast<const float2 *>(a.means2d)[g];
        f.opac = a.opacities[g];
        f.ca = a.conics[3 * (size_t)g]; f.cb = a.conics[3 * (size_t)g + 1]; f.cc = a.conics[3 * (size_t)g + 2];
        const float *cp = a.colors + (size_t)g * a.cdim + a.ch_off;
#pragma unroll
        for (int k = 0; k < 4; ++k) f.cv[k] = (k < CH && k < (int)a.nch) ? cp[k] : 0.0f;
    };
    int32_t g_cur = entry_of(0), g_nxt = entry_of(1);
    Fetched f_cur{};
    fetch(g_cur, f_cur);

    uint32_t open = (1u << NQ) - 1u; // quadrants with a pixel that is not done (wave-uniform)
    for (int32_t b = 0; b < n_batches; ++b) {
        // the rectangle of each quadrant's open pixels, tile-centre coordinates (it shrinks as pixels saturate)
        float rcx[NQ], rcy[NQ], rhw[NQ], rhh[NQ];
        open = 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(thr[q] < INFINITY);
            rcx[q] = rcy[q] = rhw[q] = rhh[q] = 0.0f;
            if (m) {
                open |= 1u << q;
                int xmin = 0, xmax = 7, ymin = 0, ymax = 7;
#if GSX_FWD_W_RECT
                open_range(m, xmin, xmax, ymin, ymax);
#endif
                const uint32_t gq = q_first + (uint32_t)q;
                rcx[q] = 0.5f * (float)(xmin + xmax) + ((gq & 1u) ? 0.5f : -7.5f);
                rcy[q] = 0.5f * (float)(ymin + ymax) + ((gq >> 1) ? 0.5f : -7.5f);
                rhw[q] = 0.5f * (float)(xmax - xmin);
                rhh[q] = 0.5f * (float)(ymax - ymin);
            }
        }
        if (!open) break; // every pixel of the tile finished

        int hitmask = 0; // this lane's staged Gaussian: quadrants whose open pixels it can reach
        {
            const int32_t g = g_cur;
            const Fetched f = f_cur;
            if (g >= 0) {
                const float opac = f.opac, ca = f.ca, cb = f.cb, cc = f.cc;
                const float ax = f.xy.x - tile_cx, ay = f.xy.y - tile_cy;
                v4f p0;
                float nA, nB, nC;
                stage_gaussian_f(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
                const v4f p1  = v4f{nA, nB, nC, f.cv[2]};
                s_st[lane].p0 = p0;
                s_st[lane].p1 = p1;
                s_st[lane].p2 = v4f{f.cv[0], f.cv[1], f.cv[3], 0.0f};
                const float2 he = cull_half_extent(opac, ca, cb, cc);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!(open & (1u << q))) continue; // scalar
                    WaveRect r;
                    r.cx = rcx[q]; r.cy = rcy[q]; r.hw = rhw[q]; r.hh = rhh[q]; r.any = true;
                    bool hit = (fabsf(ax - r.cx) - r.hw <= he.x) && (fabsf(ay - r.cy) - r.hh <= he.y);
                    if (hit) hit = rect_reaches_level(p0, p1, ax, ay, r); // exact second stage
                    hitmask |= hit ? (1 << q) : 0;
                }
            }
        }
        g_cur = g_nxt;
        fetch(g_cur, f_cur); // rows of batch b + 1: in flight while batch b is walked
        g_nxt = entry_of(b + 2);
        wave_lds_sync();

        const int32_t batch_start = __builtin_amdgcn_readfirstlane(range_start + BATCH * b);
        uint64_t todo             = __builtin_amdgcn_ballot_w64(hitmask != 0);
        while (todo) {
            const int32_t t = (int32_t)__builtin_ctzll(todo);
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t)); // todo &= todo - 1 in one scalar instruction
            const uint32_t qm = (uint32_t)__builtin_amdgcn_readlane(hitmask, t) & open;
            if (!qm) continue; // its quadrants closed since it was staged
            const v4f p0 = s_st[t].p0;
            const v4f p1 = s_st[t].p1;
            const v4f p2 = s_st[t].p2; // one address register for the three reads
            // the list index of this Gaussian in a vector register, once per Gaussian (the compiler re-materialises a v_mov per
            // quadrant otherwise: a select takes its "true" operand from a VGPR)
            uint32_t idx;
            asm("v_mov_b32 %0, %1" : "=v"(idx) : "s"(batch_start + t));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(qm & (1u << q))) continue; // scalar
                const uint32_t gq = q_first + (uint32_t)q;
                const float e     = staged_f(p0, p1.x, p1.y, p1.z, pu[gq & 1u], pv[gq >> 1]);
                const float alpha = fminf(kMaxAlpha, __builtin_amdgcn_exp2f(e));
                // branch-free body (raster3d_fwd.hip): passes / saturates / is blended are lane masks combined on the scalar side
                const bool ok = !(e > p0.w) && !(alpha < thr[q]); // e > lo <=> sigma < 0
                if (__builtin_amdgcn_ballot_w64(ok) == 0ull) { // wave-uniform: no pixel of the quadrant takes this Gaussian
                    // ... maybe because none is left: a quadrant that finished is found out HERE, by the first Gaussian that
                    // reaches it afterwards (a ballot per blended pair would cost two vector instructions each)
                    if (__builtin_amdgcn_ballot_w64(thr[q] < INFINITY) == 0ull) open &= ~(1u << q);
                    continue;
                }
                const float next_T = fmaf(-T[q], alpha, T[q]);
                const bool low     = next_T <= kTransmittanceThresh; // saturated: this Gaussian is excluded
                const bool sat = ok && low, take = ok && !sat;
                const float at = take ? alpha : 0.0f;
                const float w  = at * T[q];
                acc[q][0] += p2.x * w;
                if constexpr (CH > 1) acc[q][1] += p2.y * w;
                if constexpr (CH > 2) acc[q][2] += p1.w * w;
                if constexpr (CH > 3) acc[q][3] += p2.z * w;
                cur_idx[q] = take ? idx : cur_idx[q];
                T[q]       = fmaf(-T[q], at, T[q]); // == next_T where the Gaussian is blended, T exactly where not (a full-rate fma for a select)
                thr[q]     = sat ? INFINITY : thr[q];
            }
            if (!open) break;
        }
        if (!open) break;
        // the next batch's staging overwrites s_st: LDS operations of one wave execute in order, and the reads above were
        // issued before those writes
    }

    int32_t last = -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int64_t prow = row_of(q);
        if (prow < 0) continue;
        const size_t pix = (size_t)prow;
        last             = max(last, (int32_t)cur_idx[q]);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch) a.render_colors[pix * a.cdim + a.ch_off + k] = bg ? (acc[q][k] + T[q] * bg[k]) : acc[q][k];
        if (a.first_chunk) {
            a.render_alphas[pix] = 1.0f - T[q];
            a.last_ids[pix]      = (int32_t)cur_idx[q];
        }
    }
    if (cost_out) { // what this tile costs the backward: its list up to the last contributor (tile_order.hip)
        last = wave_max_i32(last);
        if (lane == 0) *cost_out = max(0, min(range_end, last + 1) - range_start);
    }
}

template <int CH, int NQ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSX_FWD_W_WAVES)))
raster3d_fwd_w_kernel(const Raster3DArgs a)
{
    raster3d_fwd_w_body<CH, NQ>(a);
}

// GSX_RASTER3D_FWD=q selects the four-waves-per-tile kernel (A/B; read once per process)
static char fwd_variant()
{
    static const char v = [] {
        const char *e = getenv("GSX_RASTER3D_FWD");
        if (e && (e[0] == 'q' || e[0] == 'Q')) return 'q';
        if (e && (e[0] == 'w' || e[0] == 'W')) return 'w';
        if (e && (e[0] == 'h' || e[0] == 'H')) return 'h';
        return GSX_RASTER3D_FWD_DEFAULT;
    }();
    return v;
}
bool raster3d_fwd_w_applies(const Raster3DArgs &a)
{
    return (fwd_variant() == 'w' || fwd_variant() == 'h') && a.tile_size == 16 && a.nch <= 4 && a.seg_mode == 0 && a.seg_len == 0;
}
int raster3d_fwd_w_launch(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0) return GSX_OK;
    const uint32_t grid = ((n_blocks + 7u) / 8u) * 8u; // xcd_remap needs a multiple of 8
    const size_t smem   = 64 * sizeof(StagedRow);
    if (fwd_variant() == 'h') { // one wave per half tile
        if (a.nch <= 1) raster3d_fwd_w_kernel<1, 2><<<dim3(2u * grid), dim3(64), smem, stream>>>(a);
        else if (a.nch <= 2) raster3d_fwd_w_kernel<2, 2><<<dim3(2u * grid), dim3(64), smem, stream>>>(a);
        else if (a.nch <= 3) raster3d_fwd_w_kernel<3, 2><<<dim3(2u * grid), dim3(64), smem, stream>>>(a);
        else raster3d_fwd_w_kernel<4, 2><<<dim3(2u * grid), dim3(64), smem, stream>>>(a);
        return check_launch("raster3d_fwd_h");
    }
    if (a.nch <= 1) raster3d_fwd_w_kernel<1, 4><<<dim3(grid), dim3(64), smem, stream>>>(a);
    else if (a.nch <= 2) raster3d_fwd_w_kernel<2, 4><<<dim3(grid), dim3(64), smem, stream>>>(a);
    else if (a.nch <= 3) raster3d_fwd_w_kernel<3, 4><<<dim3(grid), dim3(64), smem, stream>>>(a);
    else raster3d_fwd_w_kernel<4, 4><<<dim3(grid), dim3(64), smem, stream>>>(a);
    return check_launch("raster3d_fwd_w");
}

} // namespace gsx
