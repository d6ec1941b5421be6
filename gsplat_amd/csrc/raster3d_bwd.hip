// 3DGS alpha compositing, backward (gfx950).
// C-ABI entry: gsx_raster3d_bwd  (replaces torch op gsplat::rasterize_to_pixels_3dgs_bwd,
// reference host fn gsplat/cuda/csrc/Rasterization.cpp:484-587, kernel
// RasterizeToPixels3DGSSerialBatchBwd.cu:41-320, math RasterizeToPixels3DGSDevice.cuh:105-173).
//
// Gradient accumulation is re-designed for wave64 + 160 KiB LDS instead of the reference's
// "32-lane reduce, then 9+D global atomics per warp per Gaussian":
//   1. each lane computes its pixel's contribution to one Gaussian as K = D+6 [+2 absgrad] values:
//      D colour terms and the six MOMENTS of w = v_sigma over the pixel offsets d = mean - pixel
//      (sum w dx^2, w dx dy, w dy^2, w dx, w dy, w); v_conics, v_means2d and v_opacities are linear in
//      those moments and are formed once per (tile, Gaussian) at flush time instead of once per pixel;
//   2. the wave reduces FOUR values at a time with permlane16/32 swaps + DPP (common.hpp), a single
//      left-over value with 6 DPP adds (row_bcast);
//   3. the K totals land in K different lanes, and ONE ds_add_f32 adds them into a per-tile LDS
//      accumulator row for that Gaussian (4 waves -> 4-way LDS contention at most);
//   4. after the batch the accumulator is flushed with global_atomic_add_f32 — one atomic per (tile, Gaussian,
//      component) instead of one per (warp, Gaussian, component) — into an ARRAY-OF-STRUCTURES gradient buffer
//      [R][6 (+2) + D], TRANSPOSED: consecutive lanes add consecutive floats of one row, so a wave-level atomic
//      touches ~6 cache lines instead of 64. Measured on c3: the per-tensor layout (64 lines per instruction)
//      cost 160 us of exposed L2-atomic time out of 707 us; the AoS flush costs < 5 us.
#include <cstdlib>
#include <cstring>

#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

template <int CH, bool ABS>
struct BwdCfg {
    static constexpr int K     = CH + 6 + (ABS ? 2 : 0);     // values reduced per Gaussian
    static constexpr bool TAIL1 = (K % 4) == 1;              // one left-over value: cheaper single reduction
    static constexpr int KQ    = TAIL1 ? K / 4 : (K + 3) / 4; // groups of four
    static constexpr int KP    = (K | 1);                    // odd row stride -> conflict-free flush
    static constexpr int BATCH = (CH >= 16) ? 128 : 256;     // LDS budget for wide channel chunks
    static constexpr size_t smem =
        (size_t)BATCH * (2 * sizeof(float4) + sizeof(float2) + sizeof(int32_t) * 2 + sizeof(float) * (CH + KP));
};

template <int CH, bool ABS>
__global__ void __launch_bounds__(256) raster3d_bwd_kernel(const Raster3DArgs a)
{
    using Cfg           = BwdCfg<CH, ABS>;
    constexpr int K     = Cfg::K;
    constexpr int KQ    = Cfg::KQ;
    constexpr int KP    = Cfg::KP;
    constexpr int BATCH = Cfg::BATCH;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *s_ga     = reinterpret_cast<float4 *>(smem_raw);       // x, y, log2(opac), A   (stage_gaussian)
    float2 *s_gb     = reinterpret_cast<float2 *>(s_ga + BATCH);   // B, C
    float4 *s_cull   = reinterpret_cast<float4 *>(s_gb + BATCH);   // x, y, half extents of alpha>=1/255
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_cull + BATCH); // flatten id of the row
    int32_t *s_touch = s_id + BATCH;                               // any lane contributed?
    float *s_col     = reinterpret_cast<float *>(s_touch + BATCH); // [BATCH][CH]
    float *s_acc     = s_col + BATCH * CH;                         // [BATCH][KP]

    TileCtx tc;
    if (!tile_context(a, blockIdx.x, tc)) return;
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image_id = tc.image_id, tile_id = tc.tile_id;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;

    const uint32_t tid  = threadIdx.x;
    const uint32_t lane = tid & 63u;

    uint32_t lx, ly;
    tile_pixel(tid, a.tile_size, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
    const bool inside  = prow >= 0;
    // everything in the tile-centre frame of the forward kernels (the alpha test must be the forward's, bit for bit)
    const float half_r = 0.5f * (float)a.tile_size;
    const float tcx    = (float)(tc.tile_x * a.tile_size) + half_r, tcy = (float)(tc.tile_y * a.tile_size) + half_r;
    const float px     = (float)lx + 0.5f - half_r; // pixel centre - tile centre
    const float py     = (float)ly + 0.5f - half_r;
    const size_t pix   = inside ? (size_t)prow : 0;

    const int32_t range_start = tc.range_start;
    if (tc.range_end <= range_start) return;

    // per-pixel state
    const float T_final     = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    float T                 = T_final;
    const int32_t bin_final = inside ? a.last_ids[pix] : -1;
    float v_c[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k)
        v_c[k] = (inside && k < (int)a.nch) ? a.v_render_colors[vrc_index(a, pix, a.ch_off + (uint32_t)k)] : 0.0f;
    // alpha-gradient term and background term belong to exactly one channel chunk / all chunks resp.
    const float v_a = (inside && a.first_chunk && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f; // null = zeros
    float bg_dot    = 0.0f; // sum_k bg_k * v_c_k (this chunk)
    if (a.backgrounds) {
        const float *bg = a.backgrounds + (size_t)image_id * a.cdim + a.ch_off;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch) bg_dot += bg[k] * v_c[k];
    }
    // v_alpha = sum_k (c_k T - buffer_k / (1 - alpha)) v_c,k + T_final / (1 - alpha) (v_a - bg . v_c)  (Device.cuh:105-173) with
    // buffer_k = sum over the Gaussians behind of c_k fac. Only B = sum_k buffer_k v_c,k is ever used, and B += fac (c . v_c):
    // ONE scalar per pixel instead of CH back-buffers (as variants T / W) - two instead of five instructions per channel
    // and pair, CH registers less: at 32 channels that is a third of this kernel's vector instructions.
    const float tail_term        = T_final * (v_a - bg_dot); // what lies behind the whole list
    float behind                 = 0.0f;
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    const WaveRect rect          = wave_pixel_rect(inside, px, py);
    // nothing behind the LAST contributor of the whole tile is needed (the forward's early termination cuts the lists of a
    // dense scene to a fraction): those entries are not even staged (as variants T / W)
    int32_t range_end = tc.range_end;
    {
        int32_t *s_m = reinterpret_cast<int32_t *>(smem_raw); // staging area, not in use yet
        if (lane == 0) s_m[tid >> 6] = wave_bin_final;
        __syncthreads();
        int32_t m = s_m[0];
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) m = max(m, s_m[w]);
        __syncthreads();
        range_end = min(range_end, m + 1);
    }
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return; // uniform: no pixel of the tile has a contributor

    // zero the accumulator rows this thread owns
    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int k = 0; k < KP; ++k) s_acc[s * KP + k] = 0.0f;
        s_touch[s] = 0;
    }

    for (int32_t b = 0; b < n_batches; ++b) {
        // back-to-front: slot 0 is the farthest-back Gaussian of this batch
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);

        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            const int32_t idx = batch_end - s;
            if (idx >= range_start) {
                const int32_t g  = a.flatten_ids[idx];
                const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                s_id[s]        = g;
                float4 ga;
                float2 gb;
                stage_gaussian(xy.x - tcx, xy.y - tcy, opac, ca, cb, cc, ga, gb);
                s_ga[s]        = ga;
                s_gb[s]        = gb;
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]      = make_float4(ga.x, ga.y, he.x, he.y);
                const float *c = a.colors + (size_t)g * a.cdim + a.ch_off;
#pragma unroll
                for (int k = 0; k < CH; ++k) s_col[s * CH + k] = (k < (int)a.nch) ? c[k] : 0.0f;
            }
        }
        __syncthreads();

        // Gaussians behind every pixel's last contributor in this wave are skipped wholesale.
        const int32_t t_first = max(0, batch_end - wave_bin_final);
        for (int32_t j = (t_first & ~63); j < batch_size; j += 64) {
          // cull 64 staged Gaussians at once against this wave's pixel rectangle (raster3d.hpp)
          const int32_t tl = j + (int32_t)lane;
          bool hit         = false;
          if (tl >= t_first && tl < batch_size) {
              const float4 cu = s_cull[tl];
              hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
          }
          uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) { // scalar loop over the survivors, back to front
            const int32_t t = j + (int32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            // Branch-free body: invalid lanes run the same arithmetic with alpha = vis = 0, which makes
            // every contribution exactly 0 and leaves T / buffer unchanged (1/(1-0) == 1 exactly).
            const float4 ga = s_ga[t];
            const float2 gb = s_gb[t];
            const float dx = ga.x - px;
            const float dy = ga.y - py;
            const float e    = staged_e_offset(ga, gb, dx, dy);
            const float ov_r = __builtin_amdgcn_exp2f(e); // opac * exp(-sigma), unclamped
            // lanes outside the image have bin_final = -1 and can never be valid; e > lo <=> sigma < 0
            const bool valid = (batch_end - t <= bin_final) && !(e > ga.z) && !(fminf(kMaxAlpha, ov_r) < kAlphaThreshold);
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue; // wave-uniform

            const float ov    = valid ? ov_r : 0.0f;
            const float alpha = fminf(kMaxAlpha, ov);
            float loc[(K + 3) / 4 * 4];
            {
                const float ra  = __builtin_amdgcn_rcpf(fmaxf(kMinOneMinusAlpha, 1.0f - alpha));
                T              *= ra;
                const float fac = alpha * T;
                float cv        = 0.0f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    loc[k] = fac * v_c[k];
                    cv     = fmaf(s_col[t * CH + k], v_c[k], cv);
                }
                const float v_alpha = fmaf(ra, tail_term - behind, cv * T);
                behind              = fmaf(fac, cv, behind);
                const bool unclamped = ov <= kMaxAlpha; // alpha-clamp branch: geometry/opacity grads vanish
                const float v_sigma  = unclamped ? -ov * v_alpha : 0.0f;
                const float wdx = v_sigma * dx, wdy = v_sigma * dy;
                loc[CH + 0]     = wdx * dx; // moments; see flush
                loc[CH + 1]     = wdx * dy;
                loc[CH + 2]     = wdy * dy;
                loc[CH + 3]     = wdx;
                loc[CH + 4]     = wdy;
                loc[CH + 5]     = v_sigma;
                if constexpr (ABS) {
                    // |conic . (wdx, wdy)| with conic = (2A, B; B, 2C) / log2(e); the 1/log2(e) is applied at flush
                    loc[CH + 6] = fabsf(2.0f * ga.w * wdx + gb.x * wdy);
                    loc[CH + 7] = fabsf(gb.x * wdx + 2.0f * gb.y * wdy);
                }
#pragma unroll
                for (int k = K; k < (K + 3) / 4 * 4; ++k) loc[k] = 0.0f;
            }
            // reduce four values per step; row r of group j ends up with total of value 4j + r
            float mine = 0.0f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float r = wave_sum4_scatter(loc[4 * j], loc[4 * j + 1], loc[4 * j + 2], loc[4 * j + 3]);
                if ((int)(lane & 15u) == j) mine = r;
            }
            int vidx    = 4 * (int)(lane & 15u) + (int)(lane >> 4);
            bool writer = (int)(lane & 15u) < KQ && vidx < K;
            if constexpr (Cfg::TAIL1) { // value K-1 alone: total lands in the last row; lane 63 adds it
                const float r = wave_sum_last_row(loc[K - 1]);
                if (lane == 63u) {
                    mine   = r;
                    vidx   = K - 1;
                    writer = true;
                }
            }
            if (writer) atomicAdd(&s_acc[t * KP + vidx], mine); // ds_add_f32
            if (lane == 0) s_touch[t] = 1;
          }
        }
        __syncthreads();

        // flush, transposed: element e of the batch's [batch_size][NCOL] gradient block -> (Gaussian s, column c);
        // consecutive lanes -> consecutive floats of one AoS row. Moments become gradients here (linear maps):
        //   v_xy = Q (S_x, S_y), v_conic = (S_xx/2, S_xy, S_yy/2), v_opacity = sum vis*v_alpha = -S_w / opacity
        // (a touched Gaussian has opacity >= 1/255), with Q = (2A, B; B, 2C) / log2(e) and opacity = exp2(lo) from
        // the staged form.
        constexpr float kInvLog2e = 1.0f / kLog2e;
        constexpr int GEO  = 6 + (ABS ? 2 : 0);
        constexpr int NCOL = GEO + CH;
        for (int e = (int)tid; e < batch_size * NCOL; e += (int)blockDim.x) {
            const int s = e / NCOL, c = e - s * NCOL;
            if (!s_touch[s]) continue;
            const float *row = s_acc + s * KP;
            float val;
            int col = c;
            if (c < 2) {
                const float4 ga = s_ga[s];
                const float2 gb = s_gb[s];
                const float sx = row[CH + 3], sy = row[CH + 4];
                val = kInvLog2e * ((c == 0) ? (2.0f * ga.w * sx + gb.x * sy) : (gb.x * sx + 2.0f * gb.y * sy));
            } else if (c < 5) {
                const float m = row[CH + c - 2];
                val           = (c == 3) ? m : 0.5f * m;
            } else if (c == 5) {
                val = -row[CH + 5] * __builtin_amdgcn_exp2f(-s_ga[s].z);
            } else if (c < GEO) {
                val = kInvLog2e * row[CH + c]; // |v_xy| sums (ABS): accumulator slots CH+6, CH+7 <- columns 6, 7
            } else {
                const int k = c - GEO;
                if (k >= (int)a.nch) continue;
                val = row[k];
                col = GEO + (int)a.ch_off + k;
            }
            atomic_add_f32(a.v_rows + (size_t)s_id[s] * a.row_stride + col, val);
        }
        __syncthreads();
        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            if (s < batch_size && s_touch[s]) {
                float *row = s_acc + s * KP;
#pragma unroll
                for (int k = 0; k < KP; ++k) row[k] = 0.0f;
                s_touch[s] = 0;
            }
        }
        // The next iteration's staging writes s_ga/s_gb/s_col/s_id (read above only by the owner
        // thread or before the barrier); its barrier orders the zeroed rows before new ds_adds.
    }
}

// ---- variant T: per-Gaussian sums by a TRANSPOSED walk instead of cross-lane reductions ---------------------------------
// The kernel above spends about half of its VALU instructions turning 64 per-pixel values into one per-Gaussian value
// (K = CH + 6 wave reductions per (wave, Gaussian)). Every one of those sums has the form
//     sum over pixels p of  fac(p, g) * v_c[k](p)        (colour columns)
//     sum over pixels p of  w(p, g)   * phi_m(p)         (moments: phi = 1, u, v, u^2, u v, v^2 of the pixel position)
// i.e. (one scalar per pixel and Gaussian) x (a quantity of the pixel alone). So the pixel loop only produces the two
// scalars fac = alpha * T and w = v_sigma and parks them in a wave-private LDS matrix W[slot][pixel] (one ds_write_b64 per
// Gaussian). Every 8 Gaussians the wave turns round: lane (g, v) owns Gaussian slot g and the 8 pixels of quadrant row v,
// streams its 8 (fac, w) pairs back with four b128 reads and accumulates the K sums with plain FMAs - the pixel columns
// u are compile-time constants, the row's cotangents sit in registers for the whole kernel - then the eight rows are
// folded (one DPP add + 3 permlane swaps per four values) and added to the per-tile accumulator. Moments are taken about
// the TILE origin (the same for the four waves) and moved to the Gaussian's mean once per (tile, Gaussian) in the flush:
// d = mean - pixel = a - (u, v) with a = mean - origin. LDS traffic is what bounds this variant (MI355X_MICROARCH.md, LDS:
// ds_write_b32 costs 4 cycles, a single-lane store as much as a full one, ds_read_b96 8), hence: one b64 store per
// Gaussian, no per-Gaussian single-lane stores (the slot -> Gaussian map lives in a VGPR, lane s = slot s), no cotangent
// table in LDS, and a W layout (row pitch 176, row-of-pixels pitch 20 floats) whose b128 reads are conflict-free.
// Without absgrad (|.| per pixel is not a product of that form), CH <= 4, 16 x 16 tiles.
// Tunables (A/B builds: make SUFFIX=_x EXTRA=-DGSX_BWD_T_...=..; measured on c3, MI355X, profiles/r04_ab_variant_t.md).
// The defaults keep a workgroup at 30.4 KiB of LDS and <= 102 VGPRs, i.e. 5 workgroups per CU: occupancy and batch length
// pull in opposite directions (BATCH 128 / 144 / 160 -> 533 / 526 / 572 us: 160 drops to 4 workgroups per CU).
#ifndef GSX_RASTER3D_BWD_DEFAULT // 't': variant T, 'w': variant W (one wave per tile, below); GSX_RASTER3D_BWD=r|t|w at run time
#define GSX_RASTER3D_BWD_DEFAULT 'w'
#endif
#ifndef GSX_BWD_T_BATCH
#define GSX_BWD_T_BATCH 128
#endif
#ifndef GSX_BWD_T_WROW // 176 / 20: conflict-free b128 reads; 132 / 16: 2-way conflicts but 5.6 KiB less LDS per workgroup
#define GSX_BWD_T_WROW 132
#define GSX_BWD_T_WGRP 16
#endif
// GSX_BWD_T_PRIVATE=1: one accumulator table PER WAVE, added to with plain LDS read-modify-write, instead of one per workgroup
// hit with ds_add_f32: on gfx950 a float LDS atomic retires ~0.75 lanes per cycle and CU - 28x slower than an integer one
// or a plain write (tools/issue_rate.hip) - and variant T issues 72 of them per turn of 8 Gaussians.
#ifndef GSX_BWD_T_PRIVATE
#define GSX_BWD_T_PRIVATE 0
#endif
#ifndef GSX_BWD_T_WAVES // waves per SIMD the register allocation aims at
#define GSX_BWD_T_WAVES 5
#endif
template <int CH>
struct BwdTCfg {
    static constexpr int K     = CH + 6;
    static constexpr int KP    = (K | 1);
    static constexpr int KG    = (K + 3) / 4;      // groups of four for the fold
    static constexpr int BATCH = CH <= 3 ? GSX_BWD_T_BATCH : GSX_BWD_T_BATCH - 16; // <= 30.7 KiB of LDS per workgroup for every CH
    static constexpr int SLOTS = 8;                // Gaussians per turn
    static constexpr int WROW  = GSX_BWD_T_WROW;   // floats per slot: 8 pixel rows x WGRP
    static constexpr int WGRP  = GSX_BWD_T_WGRP;   // floats per row of 8 pixels: 8 x (fac, w) + 4 (bank spread)
    static constexpr int NACC  = GSX_BWD_T_PRIVATE ? 4 : 1; // accumulator tables
    static constexpr size_t stage_bytes =
        (size_t)BATCH * (sizeof(StagedRow) + sizeof(float4) + sizeof(int32_t) * 2 + sizeof(float) * KP * NACC);
    static constexpr size_t smem = stage_bytes + sizeof(float) * (4 * SLOTS * WROW);
};

template <int CH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSX_BWD_T_WAVES)))
raster3d_bwd_t_kernel(Raster3DArgs a)
{
    using Cfg           = BwdTCfg<CH>;
    constexpr int K     = Cfg::K;
    constexpr int KP    = Cfg::KP;
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int WROW  = Cfg::WROW;
    constexpr int WGRP  = Cfg::WGRP;
    static_assert(CH <= 4, "cotangent rows are staged as float4");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    StagedRow *s_st  = reinterpret_cast<StagedRow *>(smem_raw);    // tile-centre polynomial of the exponent + colours (raster3d.hpp)
    float4 *s_cull   = reinterpret_cast<float4 *>(s_st + BATCH);   // mean - tile centre, half extents of alpha >= 1/255
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_cull + BATCH);
    int32_t *s_touch = s_id + BATCH;
    float *s_acc     = reinterpret_cast<float *>(s_touch + BATCH); // [NACC][BATCH][KP]: colours | S0 Su Sv Suu Suv Svv
    float *s_w       = s_acc + Cfg::NACC * BATCH * KP;             // [4 waves][SLOTS][WROW]

    TileCtx tc;
    uint32_t seg_item = 0;
    if (a.tile_order) {
        if (!tile_context_ordered(a, blockIdx.x, tc)) return;
    } else if (!tile_context_seg(a, blockIdx.x, tc, seg_item)) return;
    const bool in_segment = a.seg_mode != 0u && seg_item != 0xFFFFFFFFu; // a slice of a long tile list (raster3d_seg.hip)
    const uint32_t tiles_per_image = a.tile_w * a.tile_h;
    const uint32_t image_id = tc.image_id, tile_id = tc.tile_id;
    if (a.masks && !a.masks[(size_t)image_id * tiles_per_image + tile_id]) return;

    const uint32_t tid  = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    uint32_t lx, ly;
    tile_pixel(tid, 16u, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
    const bool inside  = prow >= 0;
    // tile centre in pixel coordinates; this lane's pixel centre relative to it (multiples of 0.5 in [-7.5, 7.5]: exact)
    const float tile_cx = (float)(tc.tile_x * 16u) + 8.0f, tile_cy = (float)(tc.tile_y * 16u) + 8.0f;
    const float pu      = (float)lx - 7.5f, pv = (float)ly - 7.5f;
    const size_t pix    = inside ? (size_t)prow : 0;

    const int32_t range_start = tc.range_start;
    if (tc.range_end <= range_start) return;

    const float T_final     = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    // a slice starts (back to front) from the transmittance at ITS end and from what lies behind it (raster3d_seg.hip)
    float T                 = in_segment ? a.seg_T[(size_t)seg_item * 256 + tid] : T_final;
    const int32_t bin_final = inside ? a.last_ids[pix] : -1;
    // The forward pass stopped every pixel at its last contributor (early termination at T <= 1e-4 cuts the lists of a
    // dense scene to a fraction of their length): nothing behind the LAST contributor of the whole tile is ever needed, so
    // those list entries are not even staged.
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    {
        int32_t *s_m = reinterpret_cast<int32_t *>(smem_raw); // staging area, not in use yet
        if (lane == 0) s_m[wave] = wave_bin_final;
        __syncthreads();
        const int32_t m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        __syncthreads();
        tc.range_end = min(tc.range_end, m + 1);
    }
    const int32_t range_end = tc.range_end;
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return; // uniform: no pixel of the tile has a contributor
    float v_c[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) v_c[k] = (inside && k < (int)a.nch) ? a.v_render_colors[vrc_index(a, pix, a.ch_off + (uint32_t)k)] : 0.0f;
    const float v_a = (inside && a.first_chunk && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
    float bg_dot    = 0.0f;
    if (a.backgrounds) {
        const float *bg = a.backgrounds + (size_t)image_id * a.cdim + a.ch_off;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < (int)a.nch) bg_dot += bg[k] * v_c[k];
    }
    const float tail_term = T_final * (v_a - bg_dot); // T_final (v_a - bg . v_c): what lies behind the whole list
    float behind          = in_segment ? a.seg_out[(size_t)seg_item * 256 + tid] : 0.0f; // B = sum_k buffer_k v_c,k (pixel loop)
    const WaveRect rect          = wave_pixel_rect(inside, pu, pv); // tile-centre coordinates, like s_cull

    // roles in a turn: this lane owns slot bg and quadrant row bv (pixels 8 bv .. 8 bv + 7 of the wave, u = 0..7)
    const int bg = (int)(lane & 7u), bv = (int)(lane >> 3);
    float *s_ww = s_w + wave * (SLOTS * WROW); // this wave's W: element (slot, pixel p) = float2 at slot * WROW + (p / 8) * WGRP + 2 (p % 8)
    const int w_off = (int)(lane >> 3) * WGRP + 2 * (int)(lane & 7u); // where this lane's pixel lives inside a slot row

    // the cotangents of this lane's turn pixels, in registers for the whole kernel (handed over through the still unused W)
    float vcr[8][CH];
    {
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
        float *r   = reinterpret_cast<float *>(&row);
#pragma unroll
        for (int k = 0; k < CH; ++k) r[k] = v_c[k];
        float4 *tmp = reinterpret_cast<float4 *>(s_ww);
        tmp[lane]   = row;
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 o   = tmp[8 * bv + u];
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int k = 0; k < CH; ++k) vcr[u][k] = ov[k];
        }
        wave_lds_sync();
    }
    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int w = 0; w < Cfg::NACC; ++w)
#pragma unroll
            for (int k = 0; k < KP; ++k) s_acc[(w * BATCH + s) * KP + k] = 0.0f;
        s_touch[s] = 0;
    }

    // quadrant origin in tile-CENTRE coordinates: the moments are taken about the tile centre (same for the four waves)
    const float u0 = (float)((wave & 1u) << 3) - 7.5f, v0 = (float)((wave >> 1) << 3) - 7.5f;
    int slot   = 0; // wave-uniform: slots filled since the last turn
    float *const w_ptr0 = s_ww + w_off; // this lane's (fac, w) cell in slot 0 ...
    float *w_ptr        = w_ptr0;       // ... and in the next free slot (a vector add per Gaussian instead of scalar address math)
    int slot_t = 0; // LANE s (s < 8) holds the staged index of the Gaussian in slot s (v_writelane); the turn broadcasts it

    // one turn: sums of the filled slots -> s_acc (all lanes of the wave take part)
    auto turn = [&](int n_slots) {
        wave_lds_sync();
        float acc[Cfg::KG * 4];
#pragma unroll
        for (int k = 0; k < Cfg::KG * 4; ++k) acc[k] = 0.0f;
        const float4 *wr = reinterpret_cast<const float4 *>(s_ww + bg * WROW + bv * WGRP);
        // Packed fp32 (v_pk_fma_f32: two FMAs per issue slot): the colour sums in pairs, (sum w u, sum w u^2) as a pair whose
        // multiplier (u, u^2) is a compile-time constant. 4 (D = 3) instead of 6 instructions per pixel.
        v2f cpair[(CH + 1) / 2], s12 = v2f{0.0f, 0.0f}; // colour sums 2k, 2k+1 | sum w u, sum w u^2 over the row
#pragma unroll
        for (int k = 0; k < (CH + 1) / 2; ++k) cpair[k] = v2f{0.0f, 0.0f};
        float s0 = 0.0f; // sum w
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float4 x = wr[h]; // (fac, w) of u = 2h, 2h + 1
            const float ff[2] = {x.x, x.z}, ww[2] = {x.y, x.w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ui  = 2 * h + e;
                const float u = (float)ui;
#pragma unroll
                for (int k = 0; k < (CH + 1) / 2; ++k) {
                    const v2f vc = v2f{vcr[ui][2 * k], 2 * k + 1 < CH ? vcr[ui][2 * k + 1 < CH ? 2 * k + 1 : 0] : 0.0f};
                    cpair[k]     = v2f{ff[e], ff[e]} * vc + cpair[k];
                }
                s0 += ww[e];
                s12 = v2f{ww[e], ww[e]} * v2f{u, u * u} + s12;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = (k & 1) ? cpair[k / 2].y : cpair[k / 2].x;
        const float s1 = s12.x, s2 = s12.y;
        // quadrant coordinates -> tile coordinates: u' = u + u0, v' = bv + v0 (one row per lane, so v' is a constant here)
        {
            const float vt = (float)bv + v0;
            const float Su = fmaf(u0, s0, s1);
            acc[CH + 0]    = s0;
            acc[CH + 1]    = Su;
            acc[CH + 2]    = vt * s0;
            acc[CH + 3]    = s2 + u0 * (2.0f * s1 + u0 * s0);
            acc[CH + 4]    = vt * Su;
            acc[CH + 5]    = vt * vt * s0;
        }
        // fold the eight rows: lanes c and c ^ 8 share a 16-lane row, then the four rows; afterwards lane (row r, column c)
        // holds sum number 4 j + r of slot c % 8
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += dpp_ror8(acc[k]); // row_ror:8
        const int frow = (int)(lane >> 4), fcol = (int)(lane & 15u);
        // the writer lanes (lane % 16 < 8, in all four rows) fetch the staged index of THEIR slot from lane (lane % 8): one
        // ds_bpermute per turn instead of a move + compare + select per Gaussian in the pixel loop
        const int t_g = __builtin_amdgcn_ds_bpermute((int)((lane & 7u) << 2), slot_t);
        const bool wr_lane = fcol < SLOTS && fcol < n_slots;
#pragma unroll
        for (int j = 0; j < Cfg::KG; ++j) {
            const float tot = rows_sum4_scatter(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
            const int idx   = 4 * j + frow;
            if (idx < K && wr_lane) {
                if constexpr (GSX_BWD_T_PRIVATE) s_acc[((int)wave * BATCH + t_g) * KP + idx] += tot; // this wave's own table
                else atomicAdd(&s_acc[t_g * KP + idx], tot);                                        // ds_add_f32
            }
        }
        if (frow == 0 && wr_lane) s_touch[t_g] = 1;
        wave_lds_sync();
    };

    for (int32_t b = 0; b < n_batches; ++b) {
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);

        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            const int32_t idx = batch_end - s;
            if (idx >= range_start) {
                const int32_t g  = a.flatten_ids[idx];
                const float2 xy  = reinterpret_cast<const float2 *>(a.means2d)[g];
                const float opac = a.opacities[g];
                const float ca = a.conics[3 * (size_t)g], cb = a.conics[3 * (size_t)g + 1], cc = a.conics[3 * (size_t)g + 2];
                s_id[s]        = g;
                const float ax = xy.x - tile_cx, ay = xy.y - tile_cy;
                v4f p0;
                float nA, nB, nC;
                stage_gaussian_f(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
                const float2 he = cull_half_extent(opac, ca, cb, cc);
                s_cull[s]      = make_float4(ax, ay, he.x, he.y);
                const float *c = a.colors + (size_t)g * a.cdim + a.ch_off;
                float cv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) cv[k] = (k < CH && k < (int)a.nch) ? c[k] : 0.0f;
                s_st[s].p0 = p0;
                s_st[s].p1 = v4f{nA, nB, nC, cv[2]};
                s_st[s].p2 = v4f{cv[0], cv[1], cv[3], 0.0f};
            }
        }
        __syncthreads();

        const int32_t t_first = __builtin_amdgcn_readfirstlane(max(0, batch_end - wave_bin_final)); // keeps j, t and the LDS addresses in SGPRs
        const int32_t behind_s = __builtin_amdgcn_readfirstlane(batch_end); // list index of staged slot t = behind_s - t
        for (int32_t j = (t_first & ~63); j < batch_size; j += 64) {
          const int32_t tl = j + (int32_t)lane;
          bool hit         = false;
          if (tl >= t_first && tl < batch_size) {
              const float4 cu = s_cull[tl];
              hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
              if (hit) hit = rect_reaches_level(s_st[tl].p0, s_st[tl].p1, cu.x, cu.y, rect); // exact second stage
          }
          uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) {
            const int32_t bit = (int32_t)__builtin_ctzll(todo);
            const int32_t t   = j + bit;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit)); // todo &= todo - 1 in ONE scalar instruction (the compiler emits three)
            const v4f p0 = s_st[t].p0;
            const v4f p1 = s_st[t].p1;
            // the colours ride along with the two reads above (one address register, their latency under the exponent's
            // chain) although a pair in which no lane passes does not need them
            const v2f c01 = *reinterpret_cast<const v2f *>(&s_st[t].p2);
            [[maybe_unused]] const v4f p2 = CH > 3 ? s_st[t].p2 : v4f{0.f, 0.f, 0.f, 0.f};
            asm volatile("" ::"v"(c01.x), "v"(c01.y)); // keeps the read HERE (the compiler would sink it below the branch)
            const float e     = staged_f(p0, p1.x, p1.y, p1.z, pu, pv);
            const float ov_r  = __builtin_amdgcn_exp2f(e); // opac * exp(-sigma), unclamped
            const float al_r  = fminf(kMaxAlpha, ov_r);
            // lanes outside the image have bin_final = -1 and can never be valid; e > lo <=> sigma < 0
            const bool valid = (bin_final >= behind_s - t) && !(e > p0.w) && !(al_r < kAlphaThreshold);
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue; // wave-uniform

            float col[CH];
            if constexpr (CH <= 3) {
                col[0] = c01.x;
                if constexpr (CH > 1) col[1] = c01.y;
                if constexpr (CH > 2) col[2] = p1.w;
            } else {
                col[0] = p2.x; col[1] = p2.y; col[2] = p1.w; col[3] = p2.z;
            }
            // invalid lanes: alpha = 0 -> fac = 0, w = 0, T and buffer unchanged (1 / (1 - 0) == 1 exactly)
            const float alpha = valid ? al_r : 0.0f;
            const float ra    = __builtin_amdgcn_rcpf(1.0f - alpha); // alpha <= kMaxAlpha = 0.99: no guard needed
            T                *= ra;
            const float fac   = alpha * T;
            // v_alpha = sum_k (c_k T - buffer_k / (1 - alpha)) v_c,k + T_final / (1 - alpha) (v_a - bg . v_c)  (Device.cuh:105-173)
            // with buffer_k = sum over the Gaussians behind of c_k fac. Only the dot product B = sum_k buffer_k v_c,k is ever
            // used, and it obeys B += fac (c . v_c): one scalar per pixel instead of D, 7 instructions instead of 14.
            float cv = col[0] * v_c[0];
#pragma unroll
            for (int k = 1; k < CH; ++k) cv = fmaf(col[k], v_c[k], cv);
            const float v_alpha = fmaf(ra, tail_term - behind, cv * T);
            behind              = fmaf(fac, cv, behind);
            // alpha-clamp branch (opac exp(-sigma) > 0.99): no geometry gradient; invalid lanes: none either
            const float v_sigma = (valid && ov_r <= kMaxAlpha) ? -ov_r * v_alpha : 0.0f;
            *reinterpret_cast<float2 *>(w_ptr) = make_float2(fac, v_sigma); // ds_write_b64 into slot `slot`
            w_ptr += WROW;
            // t and slot are wave-uniform; gfx9 allows one SGPR per VALU instruction, so the lane select travels in M0
            // (saved and restored: the compiler treats M0 as reserved)
            {
                uint32_t m0_saved;
                asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                    : "+v"(slot_t), "=&s"(m0_saved)
                    : "s"(t), "s"(slot));
            }
            if (++slot == SLOTS) {
                turn(SLOTS);
                slot  = 0;
                w_ptr = w_ptr0;
            }
          }
        }
        if (slot) { // the accumulator rows of this batch are flushed below: finish the open turn first
            turn(slot);
            slot  = 0;
            w_ptr = w_ptr0;
        }
        __syncthreads();

        // flush, transposed (see the kernel above). Raw tile-origin moments -> moments of d = mean - pixel = a - (u, v):
        //   S_w = S0, S_x = ax S0 - Su, S_xx = ax^2 S0 - 2 ax Su + Suu, S_xy = ax ay S0 - ax Sv - ay Su + Suv, ...
        constexpr float kInvLog2e = 1.0f / kLog2e;
        constexpr int NCOL = 6 + CH;
        for (int e = (int)tid; e < batch_size * NCOL; e += (int)blockDim.x) {
            const int s = e / NCOL, c = e - s * NCOL;
            if (!s_touch[s]) continue;
            float rowv[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float v = s_acc[s * KP + k];
#pragma unroll
                for (int w = 1; w < Cfg::NACC; ++w) v += s_acc[(w * BATCH + s) * KP + k];
                rowv[k] = v;
            }
            const float *row = rowv;
            const float4 cu  = s_cull[s];
            const float ax = cu.x, ay = cu.y; // mean - tile centre: the moments are about the tile centre too
            const float S0 = row[CH], Su = row[CH + 1], Sv = row[CH + 2];
            float val;
            int col = c;
            if (c < 2) {
                const v4f p1 = s_st[s].p1; // (-A, -B, -C) of the staged form: Q = (2A, B; B, 2C) / log2(e)
                const float sx = fmaf(ax, S0, -Su), sy = fmaf(ay, S0, -Sv);
                val = -kInvLog2e * ((c == 0) ? (2.0f * p1.x * sx + p1.y * sy) : (p1.y * sx + 2.0f * p1.z * sy));
            } else if (c == 2) {
                val = 0.5f * (ax * (ax * S0 - 2.0f * Su) + row[CH + 3]);
            } else if (c == 3) {
                val = ax * (ay * S0 - Sv) - ay * Su + row[CH + 4];
            } else if (c == 4) {
                val = 0.5f * (ay * (ay * S0 - 2.0f * Sv) + row[CH + 5]);
            } else if (c == 5) {
                val = -S0 * __builtin_amdgcn_exp2f(kLoMargin - s_st[s].p0.w); // 1 / opacity
            } else {
                const int k = c - 6;
                if (k >= (int)a.nch) continue;
                val = row[k];
                col = 6 + (int)a.ch_off + k;
            }
            atomic_add_f32(a.v_rows + (size_t)s_id[s] * a.row_stride + col, val);
        }
        __syncthreads();
        for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
            if (s < batch_size && s_touch[s]) {
#pragma unroll
                for (int w = 0; w < Cfg::NACC; ++w)
#pragma unroll
                    for (int k = 0; k < KP; ++k) s_acc[(w * BATCH + s) * KP + k] = 0.0f;
                s_touch[s] = 0;
            }
        }
    }
}

// ---- variant W: ONE WAVE PER TILE ----------------------------------------------------------------------------------------
// Variant T still pays for the fact that a tile belongs to FOUR waves: every per-(tile, Gaussian) sum is combined across them
// with LDS float atomics (ds_add_f32 retires ~0.75 lanes per cycle and CU: 0.067 of the 0.43 ms launch), the staged list is
// shared through workgroup barriers (staging + barriers: a quarter of the launch), and the loop control of the pixel walk -
// scalar instructions and broadcast LDS reads - is paid once per (wave, Gaussian) pair, 2.8 times per staged Gaussian on c3.
// Here a tile is ONE wave64 and a lane owns FOUR pixels, the same position in each 8 x 8 quadrant:
//   * staging is wave-private (64 Gaussians per batch, one per lane): no __syncthreads anywhere in the kernel;
//   * the staging lane tests its Gaussian against the four quadrant rectangles (the same two-stage test as the wave-level
//     culling of variant T, vectorised over 64 Gaussians) and keeps the 4-bit answer in a register; the walk reads the staged
//     row ONCE per Gaussian and branches over the quadrants that can be reached (scalar bit tests);
//   * (fac, w) of the 256 pixels are parked per slot and quadrant; every FOUR contributing Gaussians the wave turns round:
//     lane (slot, quadrant, row pair) sums 16 pixels with plain FMAs (cotangents in registers), lanes of quadrants the
//     Gaussian did not reach skip their reads, the 16 lanes of a slot are one DPP row: a row all-reduce gives every lane
//     the K totals, and lane c of the row turns them into column c of the gradient row and adds it to HBM - consecutive
//     lanes, consecutive floats of one row. No accumulator table, no LDS atomics, no re-zeroing, no separate flush pass.
// LDS per wave: 64 x (48 + 16) B staged + 4 slots x 2304 B of W = 13.3 KB -> 12 waves per CU, each with up to 168 VGPRs;
// the four independent pixel chains per lane give the instruction-level parallelism that variant T gets from occupancy.
// CH <= 4, 16 x 16 tiles, no absgrad (as variant T).
#ifndef GSX_BWD_W_WAVES
#define GSX_BWD_W_WAVES 3
#endif
#ifndef GSX_BWD_W_ABS_WAVES
#define GSX_BWD_W_ABS_WAVES 2
#endif
#ifndef GSX_BWD_W_PREFETCH // the next survivor's staged row is read while the current one is composited
#define GSX_BWD_W_PREFETCH 0
#endif
#ifndef GSX_BWD_W_EAGER // exponent / alpha / validity of the four quadrants before any branch
#define GSX_BWD_W_EAGER 0
#endif
#ifndef GSX_BWD_W_STAGE_AHEAD // the next batch's rows are requested before the current batch is walked
#define GSX_BWD_W_STAGE_AHEAD 1
#endif
template <int CH, bool ABS = false>
struct BwdWCfg {
    static constexpr int GEO   = 6 + (ABS ? 2 : 0); // geometry columns of the gradient row (+ |v_mean2d| with absgrad)
    static constexpr int K     = CH + GEO;
    static constexpr int NCOL  = GEO + CH;
    static constexpr int BATCH = 64;     // one staged Gaussian per lane
    static constexpr int SLOTS = 4;      // Gaussians per turn: 4 slots x 4 quadrants x 4 row pairs = 64 lanes
    static constexpr int GP    = 36;     // floats per group of 16 pixels (two rows of a quadrant): 16 x (fac, w) + 4 (bank spread)
    static constexpr int QP    = 4 * GP; // per quadrant
    static constexpr int SP    = 4 * QP; // per slot
    static constexpr size_t smem = (size_t)BATCH * (sizeof(StagedRow) + sizeof(float4)) + sizeof(float) * SLOTS * SP;
};

template <int CH, bool ABS>
__device__ __forceinline__ void raster3d_bwd_w_body(const Raster3DArgs &a)
{
    using Cfg           = BwdWCfg<CH, ABS>;
    constexpr int K     = Cfg::K;
    constexpr int NCOL  = Cfg::NCOL;
    constexpr int GEO   = Cfg::GEO;
    static_assert(NCOL <= 16, "lane c of a 16-lane row owns column c of the gradient row");
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int GP = Cfg::GP, QP = Cfg::QP, SP = Cfg::SP;
    static_assert(CH <= 4, "cotangent rows are exchanged as float4");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    StagedRow *s_st = reinterpret_cast<StagedRow *>(smem_raw);  // e-form of the exponent + colours (raster3d.hpp)
    float4 *s_aux   = reinterpret_cast<float4 *>(s_st + BATCH); // mean - tile centre (x, y), flatten id (bits), -
    float *s_w      = reinterpret_cast<float *>(s_aux + BATCH); // [SLOTS][4 quadrants][4 groups][GP]: (fac, w) per pixel

    TileCtx tc;
    uint32_t seg_item = 0;
    if (a.seg_mode != 0u) { // a launch over slices of long tile lists + the short tiles behind them (raster3d_seg.hip)
        if (!tile_context_seg(a, blockIdx.x, tc, seg_item)) return;
    } else if (a.tile_order ? !tile_context_ordered(a, blockIdx.x, tc) : !tile_context(a, blockIdx.x, tc)) return;
    const bool in_segment = a.seg_mode != 0u && seg_item != 0xFFFFFFFFu;
    if (a.masks && !a.masks[(size_t)tc.image_id * (a.tile_w * a.tile_h) + tc.tile_id]) return;
    const int32_t range_start = tc.range_start;
    if (tc.range_end <= range_start) return;

    const uint32_t lane = threadIdx.x & 63u;
    // this lane's four pixels: (qx, qy) inside each quadrant; centres relative to the tile centre (multiples of 0.5: exact)
    const uint32_t qx = lane & 7u, qy = lane >> 3;
    const float pu[2] = {(float)qx - 7.5f, (float)qx + 0.5f}, pv[2] = {(float)qy - 7.5f, (float)qy + 0.5f};
    const float tile_cx = (float)(tc.tile_x * 16u) + 8.0f, tile_cy = (float)(tc.tile_y * 16u) + 8.0f;

    float T[4], behind[4], tail_term[4], v_c[4][CH];
    int32_t bin_final[4];
    int32_t tile_last = -1;
    int32_t qmax[4]; // last contributor of each quadrant (wave-uniform)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t lx = ((uint32_t)(q & 1) << 3) | qx, ly = ((uint32_t)(q >> 1) << 3) | qy;
        const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
        const bool inside  = prow >= 0;
        const size_t pix   = inside ? (size_t)prow : 0;
        const float T_fin  = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
        // a slice starts (back to front) from the transmittance at ITS end and from what lies behind it; the pre-pass stores
        // both per pixel in the four-wave kernels' thread order: quadrant q, lane (raster3d_seg.hip: seg_bwd_prefix_kernel)
        T[q]               = in_segment ? a.seg_T[(size_t)seg_item * 256 + (size_t)q * 64 + lane] : T_fin;
        behind[q]          = in_segment ? a.seg_out[(size_t)seg_item * 256 + (size_t)q * 64 + lane] : 0.0f;
        bin_final[q]       = inside ? a.last_ids[pix] : -1;
#pragma unroll
        for (int k = 0; k < CH; ++k) v_c[q][k] = (inside && k < (int)a.nch) ? a.v_render_colors[vrc_index(a, pix, a.ch_off + (uint32_t)k)] : 0.0f;
        const float v_a = (inside && a.first_chunk && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
        float bg_dot    = 0.0f;
        if (a.backgrounds) {
            const float *bg = a.backgrounds + (size_t)tc.image_id * a.cdim + a.ch_off;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (k < (int)a.nch) bg_dot += bg[k] * v_c[q][k];
        }
        tail_term[q] = T_fin * (v_a - bg_dot); // T_final (v_a - bg . v_c): what lies behind the whole list
        qmax[q]      = wave_max_i32(bin_final[q]);
        tile_last    = max(tile_last, qmax[q]);
    }
    // nothing behind the tile's last contributor is ever needed (early termination cut the lists in the forward pass)
    const int32_t range_end = min(tc.range_end, tile_last + 1);
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return;

    // roles in a turn: lane = (slot tg, quadrant tq, row pair tj) sums the 16 pixels of rows 2 tj, 2 tj + 1 of quadrant tq
    const int tg = (int)(lane >> 4), tq = (int)((lane >> 2) & 3u), tj = (int)(lane & 3u);
    // their cotangents, in registers for the whole kernel (handed over through the still unused W region)
    float vcr[16][CH];
    {
        float4 *tmp = reinterpret_cast<float4 *>(s_w); // [4 quadrants][64 pixels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
            float *r   = reinterpret_cast<float *>(&row);
#pragma unroll
            for (int k = 0; k < CH; ++k) r[k] = v_c[q][k];
            tmp[q * 64 + (int)lane] = row;
        }
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float4 o    = tmp[tq * 64 + 16 * tj + p];
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int k = 0; k < CH; ++k) vcr[p][k] = ov[k];
        }
        wave_lds_sync();
    }

    // where this lane's (fac, w) of quadrant q lives inside a slot: in-quadrant pixel index = lane
    float *const w_ptr0 = s_w + (int)(lane >> 4) * GP + 2 * (int)(lane & 15u);
    float *w_ptr        = w_ptr0; // ... of the next free slot
    int slot            = 0;      // wave-uniform: slots filled since the last turn
    int slot_info       = 0;      // LANE s (s < SLOTS): staged index << 4 | quadrants that contributed, of the Gaussian in slot s

    // one turn: sums of the filled slots -> gradient rows in HBM (all lanes take part)
    auto turn = [&](int n_slots) {
        wave_lds_sync();
        const int info  = __builtin_amdgcn_ds_bpermute(tg << 2, slot_info);
        const bool live = tg < n_slots;
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.0f;
        if (live && ((info >> tq) & 1)) { // quadrants the Gaussian did not reach were never written: their lanes add nothing
            const v4f *rd = reinterpret_cast<const v4f *>(s_w + tg * SP + tq * QP + tj * GP);
            float rs[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}; // per row: sum w, sum w ul, sum w ul^2 (ul = 0..7: constants)
            // local (ul, vl in {0, 1}) -> tile-centre coordinates: u = ul + u0, v = vl + v0
            const float u0 = (float)((tq & 1) << 3) - 7.5f, v0 = (float)(((tq >> 1) << 3) + 2 * tj) - 7.5f;
            // absgrad (reference Device.cuh: v_xy_abs = |v_sigma * conic . d|, summed per PIXEL): not a moment of w, but
            // conic . d * log2(e) is affine in the pixel - the gradient of the staged exponent, (gu + 2 nA u + nB v,
            // gv + nB u + 2 nC v) - so the turn forms it per pixel from two row constants: two instructions per pixel and axis
            [[maybe_unused]] float gxr[2] = {0.f, 0.f}, gyr[2] = {0.f, 0.f}, ax2 = 0.f, bx1 = 0.f, sabs[2] = {0.f, 0.f};
            if constexpr (ABS) {
                const v4f q0 = s_st[info >> 4].p0, q1 = s_st[info >> 4].p1; // staged row | nA, nB, nC, -
#if GSX_DFORM // row = (ax, ay, lo, lo): the gradient of the exponent at the tile centre from the mean's offset
                const float gu = -fmaf(2.0f * q1.x, q0.x, q1.y * q0.y), gv = -fmaf(q1.y, q0.x, 2.0f * q1.z * q0.y);
#else         // row = (e0, gu, gv, lo)
                const float gu = q0.y, gv = q0.z;
#endif
                ax2 = 2.0f * q1.x; bx1 = q1.y;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float vr = v0 + (float)r;
                    gxr[r] = fmaf(ax2, u0, fmaf(q1.y, vr, gu));
                    gyr[r] = fmaf(q1.y, u0, fmaf(2.0f * q1.z, vr, gv));
                }
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const v4f x       = rd[h]; // (fac, w) of pixels 2h, 2h + 1
                const float ff[2] = {x.x, x.z}, ww[2] = {x.y, x.w};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int pp   = 2 * h + e;
                    const float ul = (float)(pp & 7);
#pragma unroll
                    for (int k = 0; k < CH; ++k) acc[k] = fmaf(ff[e], vcr[pp][k], acc[k]);
                    rs[pp >> 3][0] += ww[e];
                    rs[pp >> 3][1] = fmaf(ww[e], ul, rs[pp >> 3][1]);
                    rs[pp >> 3][2] = fmaf(ww[e], ul * ul, rs[pp >> 3][2]);
                    if constexpr (ABS) {
                        const float gx = fmaf(ax2, ul, gxr[pp >> 3]), gy = fmaf(bx1, ul, gyr[pp >> 3]);
                        sabs[0] = fmaf(fabsf(ww[e]), fabsf(gx), sabs[0]);
                        sabs[1] = fmaf(fabsf(ww[e]), fabsf(gy), sabs[1]);
                    }
                }
            }
            if constexpr (ABS) {
                acc[CH + 6] = sabs[0];
                acc[CH + 7] = sabs[1];
            }
            const float s0 = rs[0][0] + rs[1][0], s1 = rs[0][1] + rs[1][1], s2 = rs[0][2] + rs[1][2];
            const float t1 = rs[1][0], m11 = rs[1][1]; // sum w vl (= sum w vl^2), sum w ul vl
            const float Su = fmaf(u0, s0, s1);
            acc[CH + 0]    = s0;
            acc[CH + 1]    = Su;
            acc[CH + 2]    = fmaf(v0, s0, t1);
            acc[CH + 3]    = s2 + u0 * (2.0f * s1 + u0 * s0);
            acc[CH + 4]    = m11 + u0 * t1 + v0 * Su;
            acc[CH + 5]    = t1 + v0 * (2.0f * t1 + v0 * s0);
        }
        // the 16 lanes of a slot are one DPP row: every lane of the row gets the K totals
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = row16_sum(acc[k]);
        // lane c of the row -> column c of the Gaussian's gradient row. Raw tile-centre moments -> moments of d = mean - pixel
        // = a - (u, v):  S_x = ax S0 - Su, S_xx = ax^2 S0 - 2 ax Su + Suu, S_xy = ax ay S0 - ax Sv - ay Su + Suv, ...
        const int c = (int)(lane & 15u);
        if (live && c < NCOL) {
            constexpr float kInvLog2e = 1.0f / kLog2e;
            const int t_g    = info >> 4;
            const float4 aux = s_aux[t_g];
            const v4f p1     = s_st[t_g].p1; // (-A, -B, -C) of the staged form: Q = (2A, B; B, 2C) / log2(e)
            const float lo   = s_st[t_g].p0.w - kLoMargin;
            const float ax = aux.x, ay = aux.y;
            const int32_t id = __float_as_int(aux.z);
            const float S0 = acc[CH], Su = acc[CH + 1], Sv = acc[CH + 2];
            const float sx = fmaf(ax, S0, -Su), sy = fmaf(ay, S0, -Sv);
            float val;
            int col  = c;
            bool put = true;
            if (c == 0) val = -kInvLog2e * (2.0f * p1.x * sx + p1.y * sy);
            else if (c == 1) val = -kInvLog2e * (p1.y * sx + 2.0f * p1.z * sy);
            else if (c == 2) val = 0.5f * (ax * (ax * S0 - 2.0f * Su) + acc[CH + 3]);
            else if (c == 3) val = ax * (ay * S0 - Sv) - ay * Su + acc[CH + 4];
            else if (c == 4) val = 0.5f * (ay * (ay * S0 - 2.0f * Sv) + acc[CH + 5]);
            else if (c == 5) val = -S0 * __builtin_amdgcn_exp2f(-lo); // v_opacity = sum vis v_alpha = -S_w / opacity
            else if (ABS && c < GEO) val = kInvLog2e * (c == 6 ? acc[CH + (ABS ? 6 : 0)] : acc[CH + (ABS ? 7 : 0)]); // sum |v_mean2d|
            else {
                const int k = c - GEO;
                val         = 0.0f;
#pragma unroll
                for (int kk = 0; kk < CH; ++kk) val = (k == kk) ? acc[kk] : val;
                put = k < (int)a.nch;
                col = GEO + (int)a.ch_off + k;
            }
            if (put) atomic_add_f32(a.v_rows + (size_t)id * a.row_stride + col, val);
        }
        wave_lds_sync();
    };

    // Staging is wave-private, so its two dependent global reads (list entry -> the Gaussian's rows) would sit in front of every
    // batch with nothing of this wave to hide them: the rows of batch b + 1 and the list entries of batch b + 2 are requested
    // before batch b is walked (GSX_BWD_W_STAGE_AHEAD=0: the plain order, for A/B).
    struct Fetched { float2 xy; float opac, ca, cb, cc, cv[4]; };
    auto entry_of = [&](int32_t b) -> int32_t { // flatten id of this lane's entry of batch b, -1 = none
        const int32_t idx = range_end - 1 - BATCH * b - (int32_t)lane;
        return (b < n_batches && idx >= range_start) ? a.flatten_ids[idx] : -1;
    };
    auto fetch = [&](int32_t g, Fetched &f) {
        if (g < 0) return;
        if (a.splat_rows) { // one 48-byte array-of-structures row (raster3d.hpp; cdim == 3): three 16-byte loads, one address
            const v4f *rw = reinterpret_cast<const v4f *>(a.splat_rows) + 3 * (size_t)g;
            const v4f r0 = rw[0], r1 = rw[1], r2 = rw[2];
            f.xy = make_float2(r0.x, r0.y); f.ca = r0.z; f.cb = r0.w; f.cc = r1.x; f.opac = r1.y;
            f.cv[0] = r1.z; f.cv[1] = CH > 1 ? r1.w : 0.0f; f.cv[2] = CH > 2 ? r2.x : 0.0f; f.cv[3] = 0.0f;
            return;
        }
        f.xy   = reinterpret_cast<const float2 *>(a.means2d)[g];
        f.opac = a.opacities[g];
        f.ca = a.conics[3 * (size_t)g]; f.cb = a.conics[3 * (size_t)g + 1]; f.cc = a.conics[3 * (size_t)g + 2];
        const float *cp = a.colors + (size_t)g * a.cdim + a.ch_off;
#pragma unroll
        for (int k = 0; k < 4; ++k) f.cv[k] = (k < CH && k < (int)a.nch) ? cp[k] : 0.0f;
    };
#if GSX_BWD_W_STAGE_AHEAD
    int32_t g_cur = entry_of(0), g_nxt = entry_of(1);
    Fetched f_cur{};
    fetch(g_cur, f_cur);
#endif
    for (int32_t b = 0; b < n_batches; ++b) {
        // back to front: staged slot s is list entry batch_end - s
        const int32_t batch_end = range_end - 1 - BATCH * b;
        int hitmask             = 0; // this lane's staged Gaussian: quadrants whose pixels it can reach
        {
            const int32_t idx = batch_end - (int32_t)lane;
#if GSX_BWD_W_STAGE_AHEAD
            const int32_t g = g_cur;
            const Fetched f = f_cur;
#else
            const int32_t g = entry_of(b);
            Fetched f{};
            fetch(g, f);
#endif
            if (g >= 0) {
                const float opac = f.opac, ca = f.ca, cb = f.cb, cc = f.cc;
                const float ax = f.xy.x - tile_cx, ay = f.xy.y - tile_cy;
                v4f p0;
                float nA, nB, nC;
                stage_gaussian_f(ax, ay, opac, ca, cb, cc, p0, nA, nB, nC);
                const v4f p1 = v4f{nA, nB, nC, f.cv[2]};
                s_st[lane].p0 = p0;
                s_st[lane].p1 = p1;
                s_st[lane].p2 = v4f{f.cv[0], f.cv[1], f.cv[3], 0.0f};
                s_aux[lane]   = make_float4(ax, ay, __int_as_float(g), 0.0f);
                // the two-stage test of the wave-level culling (raster3d.hpp), once per quadrant; a quadrant whose pixels all
                // stopped in front of this entry cannot be reached either
                const float2 he = cull_half_extent(opac, ca, cb, cc);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    WaveRect r;
                    r.cx = (q & 1) ? 4.0f : -4.0f; r.cy = (q >> 1) ? 4.0f : -4.0f; r.hw = 3.5f; r.hh = 3.5f; r.any = true;
                    bool hit = idx <= qmax[q] && (fabsf(ax - r.cx) - r.hw <= he.x) && (fabsf(ay - r.cy) - r.hh <= he.y);
                    if (hit) hit = rect_reaches_level(p0, p1, ax, ay, r);
                    hitmask |= hit ? (1 << q) : 0;
                }
            }
        }
#if GSX_BWD_W_STAGE_AHEAD
        g_cur = g_nxt;
        fetch(g_cur, f_cur);   // rows of batch b + 1: in flight while batch b is walked
        g_nxt = entry_of(b + 2);
#endif
        wave_lds_sync();

        const int32_t behind_s = __builtin_amdgcn_readfirstlane(batch_end); // list index of staged slot t = behind_s - t
        uint64_t todo          = __builtin_amdgcn_ballot_w64(hitmask != 0);
#if GSX_BWD_W_PREFETCH
        // software pipeline: the staged row of the NEXT survivor is requested while the current one is composited
        int32_t t_next = todo ? (int32_t)__builtin_ctzll(todo) : 0;
        v4f n0 = s_st[t_next].p0, n1 = s_st[t_next].p1, n2 = s_st[t_next].p2;
#endif
        while (todo) {
#if GSX_BWD_W_PREFETCH
            const int32_t t = t_next;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t));
            const v4f p0 = n0, p1 = n1, p2 = n2;
            t_next = todo ? (int32_t)__builtin_ctzll(todo) : t;
            n0 = s_st[t_next].p0; n1 = s_st[t_next].p1; n2 = s_st[t_next].p2;
            const int qm = __builtin_amdgcn_readlane(hitmask, t);
#else
            const int32_t t = (int32_t)__builtin_ctzll(todo);
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t)); // todo &= todo - 1 in one scalar instruction
            const int qm = __builtin_amdgcn_readlane(hitmask, t);
            const v4f p0 = s_st[t].p0;
            const v4f p1 = s_st[t].p1;
            const v4f p2 = s_st[t].p2;
#endif
            float col[CH];
            col[0] = p2.x;
            if constexpr (CH > 1) col[1] = p2.y;
            if constexpr (CH > 2) col[2] = p1.w;
            if constexpr (CH > 3) col[3] = p2.z;
            const int32_t list_idx = behind_s - t;
            int contributed        = 0; // wave-uniform
#if GSX_BWD_W_EAGER
            // the four quadrants' exponent / alpha / validity first, branch-free (four independent chains in flight), ...
            float ov_q[4], al_q[4];
            bool valid_q[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float e = staged_f(p0, p1.x, p1.y, p1.z, pu[q & 1], pv[q >> 1]);
                ov_q[q]       = __builtin_amdgcn_exp2f(e);
                al_q[q]       = fminf(kMaxAlpha, ov_q[q]);
                valid_q[q]    = (qm & (1 << q)) && (bin_final[q] >= list_idx) && !(e > p0.w) && !(al_q[q] < kAlphaThreshold);
            }
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if GSX_BWD_W_EAGER
                const float ov_r = ov_q[q], al_r = al_q[q];
                const bool valid = valid_q[q];
#else
                if (!(qm & (1 << q))) continue; // scalar
                const float e    = staged_f(p0, p1.x, p1.y, p1.z, pu[q & 1], pv[q >> 1]);
                const float ov_r = __builtin_amdgcn_exp2f(e); // opac * exp(-sigma), unclamped
                const float al_r = fminf(kMaxAlpha, ov_r);
                // pixels outside the image have bin_final = -1 and can never be valid; e > lo <=> sigma < 0
                const bool valid = (bin_final[q] >= list_idx) && !(e > p0.w) && !(al_r < kAlphaThreshold);
#endif
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue; // wave-uniform
                // invalid lanes: alpha = 0 -> fac = 0, w = 0, T and `behind` unchanged (1 / (1 - 0) == 1 exactly)
                const float alpha = valid ? al_r : 0.0f;
                const float ra    = __builtin_amdgcn_rcpf(1.0f - alpha); // alpha <= kMaxAlpha = 0.99: no guard needed
                T[q]             *= ra;
                const float fac   = alpha * T[q];
                // v_alpha = sum_k (c_k T - buffer_k / (1 - alpha)) v_c,k + T_final / (1 - alpha) (v_a - bg . v_c)  (Device.cuh:105-173);
                // only B = sum_k buffer_k v_c,k is ever used and it obeys B += fac (c . v_c)  (as variant T)
                float cvd = col[0] * v_c[q][0];
#pragma unroll
                for (int k = 1; k < CH; ++k) cvd = fmaf(col[k], v_c[q][k], cvd);
                const float v_alpha = fmaf(ra, tail_term[q] - behind[q], cvd * T[q]);
                behind[q]           = fmaf(fac, cvd, behind[q]);
                // alpha-clamp branch (opac exp(-sigma) > 0.99): no geometry gradient; invalid lanes: none either
                const float v_sigma = (valid && ov_r <= kMaxAlpha) ? -ov_r * v_alpha : 0.0f;
                *reinterpret_cast<float2 *>(w_ptr + q * QP) = make_float2(fac, v_sigma); // ds_write_b64 into (slot, quadrant q)
                contributed |= 1 << q;
            }
            if (contributed) {
                const int info = (t << 4) | contributed;
                uint32_t m0_saved; // the lane select of v_writelane travels in M0 (saved and restored: the compiler reserves it)
                asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                    : "+v"(slot_info), "=&s"(m0_saved)
                    : "s"(info), "s"(slot));
                w_ptr += SP;
                if (++slot == SLOTS) {
                    turn(SLOTS);
                    slot  = 0;
                    w_ptr = w_ptr0;
                }
            }
        }
        if (slot) { // the staged rows the open slots point into are overwritten by the next batch: finish the turn first
            turn(slot);
            slot  = 0;
            w_ptr = w_ptr0;
        }
    }
}

// three waves per SIMD (<= 168 VGPRs) hold the state of one to three channels; four channels (64 cotangent registers per lane)
// take two waves per SIMD rather than spill
template <int CH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSX_BWD_W_WAVES)))
raster3d_bwd_w_kernel(Raster3DArgs a)
{
    raster3d_bwd_w_body<CH, false>(a);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) raster3d_bwd_w4_kernel(Raster3DArgs a)
{
    raster3d_bwd_w_body<4, false>(a);
}
// absgrad: two more sums per (tile, Gaussian), formed per pixel in the turn (see there); two waves per SIMD
template <int CH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSX_BWD_W_ABS_WAVES)))
raster3d_bwd_w_abs_kernel(Raster3DArgs a)
{
    raster3d_bwd_w_body<CH, true>(a);
}

// Variant T is the default where it applies; GSX_RASTER3D_BWD=r selects the reduction kernel (read once per process).
static char bwd_variant()
{
    static const char v = [] {
        const char *e = getenv("GSX_RASTER3D_BWD");
        if (e && (e[0] == 'r' || e[0] == 'R')) return 'r';
        if (e && (e[0] == 't' || e[0] == 'T')) return 't';
        if (e && (e[0] == 'w' || e[0] == 'W')) return 'w';
        return GSX_RASTER3D_BWD_DEFAULT;
    }();
    return v;
}
static bool use_variant_t() { return bwd_variant() != 'r'; }

template <int CH, bool ABS>
static int launch_bwd(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0 || a.n_isects == 0) return GSX_OK;
    const uint32_t grid  = ((n_blocks + 7u) / 8u) * 8u;
    if constexpr (CH <= 4) {
        if (a.tile_size == 16 && bwd_variant() == 'w') {
            if constexpr (ABS) raster3d_bwd_w_abs_kernel<CH><<<dim3(grid), dim3(64), BwdWCfg<CH, true>::smem, stream>>>(a);
            else if constexpr (CH == 4) raster3d_bwd_w4_kernel<<<dim3(grid), dim3(64), BwdWCfg<4>::smem, stream>>>(a);
            else raster3d_bwd_w_kernel<CH><<<dim3(grid), dim3(64), BwdWCfg<CH>::smem, stream>>>(a);
            return check_launch("raster3d_bwd_w");
        }
        if constexpr (!ABS) {
            if (a.tile_size == 16 && use_variant_t()) {
                raster3d_bwd_t_kernel<CH><<<dim3(grid), dim3(256), BwdTCfg<CH>::smem, stream>>>(a);
                return check_launch("raster3d_bwd_t");
            }
        }
    }
    const uint32_t block = a.tile_size <= 8 ? 64u : 256u;
    using Cfg = BwdCfg<CH, ABS>;
    const size_t smem = Cfg::smem;
    raster3d_bwd_kernel<CH, ABS><<<dim3(grid), dim3(block), smem, stream>>>(a);
    return check_launch("raster3d_bwd");
}

// variant T over a segment item list + the short tiles behind it (raster3d_seg.hip); a.nch <= 4, tile size 16, no absgrad
int raster3d_bwd_t_launch_items(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t grid = ((a.seg_grid + 7u) / 8u) * 8u;
    if (grid == 0) return GSX_OK;
    if (a.nch <= 1) raster3d_bwd_t_kernel<1><<<dim3(grid), dim3(256), BwdTCfg<1>::smem, stream>>>(a);
    else if (a.nch <= 2) raster3d_bwd_t_kernel<2><<<dim3(grid), dim3(256), BwdTCfg<2>::smem, stream>>>(a);
    else if (a.nch <= 3) raster3d_bwd_t_kernel<3><<<dim3(grid), dim3(256), BwdTCfg<3>::smem, stream>>>(a);
    else raster3d_bwd_t_kernel<4><<<dim3(grid), dim3(256), BwdTCfg<4>::smem, stream>>>(a);
    return check_launch("raster3d_bwd_t(segments)");
}
bool raster3d_bwd_uses_variant_t() { return use_variant_t(); }
bool raster3d_bwd_uses_variant_w() { return bwd_variant() == 'w'; }
// variant W over the same item list (one wave per slice / short tile); a.nch <= 4, tile size 16, no absgrad
int raster3d_bwd_w_launch_items(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t grid = ((a.seg_grid + 7u) / 8u) * 8u;
    if (grid == 0) return GSX_OK;
    if (a.nch <= 1) raster3d_bwd_w_kernel<1><<<dim3(grid), dim3(64), BwdWCfg<1>::smem, stream>>>(a);
    else if (a.nch <= 2) raster3d_bwd_w_kernel<2><<<dim3(grid), dim3(64), BwdWCfg<2>::smem, stream>>>(a);
    else if (a.nch <= 3) raster3d_bwd_w_kernel<3><<<dim3(grid), dim3(64), BwdWCfg<3>::smem, stream>>>(a);
    else raster3d_bwd_w4_kernel<<<dim3(grid), dim3(64), BwdWCfg<4>::smem, stream>>>(a);
    return check_launch("raster3d_bwd_w(segments)");
}

// Five to eight channels: variant W takes them FOUR AT A TIME (the gradient of alpha is linear in the channels, so every launch
// adds its share of the geometry gradients; first_chunk carries the alpha cotangent). Measured at c3 (profiles/r09_ab.md):
// 8 channels 0.856 ms in two launches of W against 0.890 for one of the reduction kernel; 16 channels 1.70 against 1.33, 32
// channels 3.45 against 2.48 - every launch walks the pixels again, and from three launches on that outweighs the cheaper sums.
// Not with absgrad: sum |v_sigma g| over the pixels is not linear in the channels. GSX_BWD_W_WIDE=lo,hi overrides the range
// (0,0 = never).
static void bwd_w_wide_range(uint32_t &lo, uint32_t &hi)
{
    static const uint64_t v = [] {
        uint32_t l = 0, h = 0; // off since the matrix-core kernel (raster3d_bwd_m.hip) takes 5 .. 32 channels
        if (const char *e = getenv("GSX_BWD_W_WIDE")) {
            l = (uint32_t)atoi(e);
            const char *c = strchr(e, ',');
            h = c ? (uint32_t)atoi(c + 1) : l;
        }
        return ((uint64_t)l << 32) | h;
    }();
    lo = (uint32_t)(v >> 32);
    hi = (uint32_t)v;
}
static bool bwd_w_wide(const Raster3DArgs &a, bool has_abs)
{
    uint32_t lo, hi;
    bwd_w_wide_range(lo, hi);
    return !has_abs && a.cdim > 4 && a.tile_size == 16 && bwd_variant() == 'w' && a.cdim >= lo && a.cdim <= hi;
}

template <bool ABS>
static int bwd_dispatch(Raster3DArgs a, hipStream_t stream)
{
    const uint32_t width = bwd_w_wide(a, ABS) ? 4u : 32u; // channels per launch
    uint32_t off = 0;
    bool first   = true;
    do {
        const uint32_t rem = a.cdim - off;
        const uint32_t n   = rem > width ? width : rem;
        a.ch_off           = off;
        a.nch              = n;
        a.first_chunk      = first ? 1u : 0u;
        int rc;
        if (bwd_variant() != 'r' && raster3d_bwd_m_applies(a, ABS)) rc = raster3d_bwd_m_launch(a, stream); // 5 .. 32 channels: raster3d_bwd_m.hip
        else if (n <= 1) rc = launch_bwd<1, ABS>(a, stream);
        else if (n <= 2) rc = launch_bwd<2, ABS>(a, stream);
        else if (n <= 3) rc = launch_bwd<3, ABS>(a, stream);
        else if (n <= 4) rc = launch_bwd<4, ABS>(a, stream);
        else if (n <= 8) rc = launch_bwd<8, ABS>(a, stream);
        else if (n <= 16) rc = launch_bwd<16, ABS>(a, stream);
        else rc = launch_bwd<32, ABS>(a, stream);
        if (rc != GSX_OK) return rc;
        off += n;
        first = false;
    } while (off < a.cdim);
    return GSX_OK;
}

} // namespace gsx

extern "C" int64_t gsx_raster3d_bwd_workspace_bytes(uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return gsx::tile_order_workspace_bytes(n_images, tile_w, tile_h);
}

extern "C" int gsx_raster3d_bwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, void *stream)
{
    return gsx_raster3d_bwd_ws(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
                               last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height, tile_size,
                               tile_w, tile_h, has_abs, v_rows, row_stride, -1, 1, nullptr, 0, stream);
}

// gsx_raster3d_bwd with a workspace (gsx_raster3d_bwd_workspace_bytes): the launch then takes the tiles longest-first (see
// "longest tiles first" above). Same results; without a workspace the tiles run in launch order.
// v_colors_pixel_stride >= 0: v_render_colors is not [I, H, W, cdim]-contiguous but linear in the pixel index p = (i H + y) W + x:
// element (p, k) at p * pixel_stride + k * channel_stride floats (autograd hands over expanded or sliced cotangents as
// views: the gradient of sum() has both strides 0); -1: contiguous.
extern "C" int gsx_raster3d_bwd_ws(
    const float *means2d, const float *conics, const float *colors, const float *opacities,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, int64_t v_colors_pixel_stride,
    int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream)
{
    return gsx_raster3d_bwd_fill(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
                                 last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height, tile_size,
                                 tile_w, tile_h, has_abs, v_rows, row_stride, 0, v_colors_pixel_stride, v_colors_channel_stride,
                                 workspace, workspace_bytes, stream);
}

// gsx_raster3d_bwd_ws for gradient rows that are NOT zero-filled yet: the call fills v_rows_to_fill rows of row_stride floats
// itself - inside the tile-order cost kernel when one is launched (a kernel short of memory work: 36 MB of zeros cost it ~2 us
// where a fill kernel of its own takes 7.5), with a memset otherwise. v_rows_to_fill == 0: the caller filled them.
static int raster3d_bwd_fill_impl(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *splat_rows,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill,
    int64_t v_colors_pixel_stride, int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream);
extern "C" int gsx_raster3d_bwd_fill(
    const float *means2d, const float *conics, const float *colors, const float *opacities,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill,
    int64_t v_colors_pixel_stride, int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream)
{
    return raster3d_bwd_fill_impl(means2d, conics, colors, opacities, nullptr, backgrounds, masks, isect_offsets, flatten_ids,
                                  render_alphas, last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height,
                                  tile_size, tile_w, tile_h, has_abs, v_rows, row_stride, v_rows_to_fill, v_colors_pixel_stride,
                                  v_colors_channel_stride, workspace, workspace_bytes, stream);
}
// gsx_raster3d_bwd_fill with the Gaussians' 48-byte array-of-structures rows beside the four arrays (cdim == 3, no absgrad;
// raster3d.hpp: Raster3DArgs::splat_rows). Only the one-wave-per-tile kernel reads them; every other kernel ignores the pointer.
extern "C" int gsx_raster3d_bwd_fill_rows(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *splat_rows,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill,
    int64_t v_colors_pixel_stride, int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(!splat_rows || cdim == 3, "gsx_raster3d_bwd_fill_rows: the rows hold three colours; cdim is %u", cdim);
    GSX_REQUIRE(!splat_rows || (reinterpret_cast<uintptr_t>(splat_rows) & 15u) == 0, "gsx_raster3d_bwd_fill_rows: rows must be 16-byte aligned");
    return raster3d_bwd_fill_impl(means2d, conics, colors, opacities, splat_rows, backgrounds, masks, isect_offsets, flatten_ids,
                                  render_alphas, last_ids, v_render_colors, v_render_alphas, n_images, n_isects, cdim, width, height,
                                  tile_size, tile_w, tile_h, has_abs, v_rows, row_stride, v_rows_to_fill, v_colors_pixel_stride,
                                  v_colors_channel_stride, workspace, workspace_bytes, stream);
}
static int raster3d_bwd_fill_impl(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *splat_rows,
    const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
    const float *render_alphas, const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height, uint32_t tile_size,
    uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, int64_t v_rows_to_fill,
    int64_t v_colors_pixel_stride, int64_t v_colors_channel_stride, void *workspace, int64_t workspace_bytes, void *stream)
{
    using namespace gsx;
    int64_t fill_bytes = v_rows_to_fill > 0 ? v_rows_to_fill * (int64_t)row_stride * 4 : 0;
    auto fill_now      = [&]() -> int { // rows not filled by a kernel of this call
        if (fill_bytes > 0 && v_rows && hipMemsetAsync(v_rows, 0, (size_t)fill_bytes, (hipStream_t)stream) != hipSuccess) {
            set_last_error("gsx_raster3d_bwd_fill: memset failed");
            return GSX_ERR_LAUNCH;
        }
        fill_bytes = 0;
        return GSX_OK;
    };
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_bwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_bwd: channels must be >= 1");
    if (n_isects == 0) return fill_now(); // no intersections: nothing to add to the (zero-filled, possibly empty) gradient rows
    GSX_REQUIRE(v_rows, "gsx_raster3d_bwd: null gradient output");
    GSX_REQUIRE(row_stride >= 6u + (has_abs ? 2u : 0u) + cdim,
                "gsx_raster3d_bwd: row_stride %u too small for 6%s + %u channels", row_stride, has_abs ? " + 2" : "", cdim);
    GSX_REQUIRE(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids && render_alphas && last_ids
                                  && v_render_colors && isect_offsets),
                "gsx_raster3d_bwd: null input");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities; a.splat_rows = splat_rows;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = isect_offsets; a.flatten_ids = flatten_ids;
    a.render_alphas = const_cast<float *>(render_alphas); a.last_ids = const_cast<int32_t *>(last_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas;
    a.v_rows = v_rows; a.row_stride = row_stride;
    if (v_colors_pixel_stride >= 0) { // cotangents read in place from a layout that is linear in the pixel index
        GSX_REQUIRE(v_colors_channel_stride >= 0, "gsx_raster3d_bwd_ws: negative channel stride");
        a.vrc_strided = 1u; a.vrc_ps = v_colors_pixel_stride; a.vrc_cs = v_colors_channel_stride;
    }
    // the launches that read the order: variant W (also with absgrad, and four channels at a time), variant T
    a.nch = cdim > 32 ? 32 : cdim; // what the first launch will see (bwd_dispatch sets it per chunk)
    const bool wide_m = cdim > 4 && bwd_variant() != 'r' && raster3d_bwd_m_applies(a, has_abs != 0);
    if (wide_m || (bwd_variant() == 'w' ? (tile_size == 16 && (cdim <= 4 || bwd_w_wide(a, has_abs != 0)))
                                        : (!has_abs && cdim <= 4 && bwd_variant() != 'r'))) {
        int rc       = GSX_OK;
        const bool in_kernel = fill_bytes > 0 && (reinterpret_cast<uintptr_t>(v_rows) & 15u) == 0;
        a.tile_order = a.sp_active_tiles ? nullptr
                                         : build_tile_order(a.isect_offsets, a.last_ids, a.n_images, a.tile_size, a.tile_w, a.tile_h,
                                                            a.width, a.height, a.n_isects, workspace, workspace_bytes,
                                                            (hipStream_t)stream, &rc, in_kernel ? v_rows : nullptr, fill_bytes);
        if (rc != GSX_OK) return rc;
        if (a.tile_order && in_kernel) fill_bytes = 0; // done by the cost kernel
    }
    if (int rc = fill_now(); rc != GSX_OK) return rc;
    return has_abs ? bwd_dispatch<true>(a, (hipStream_t)stream) : bwd_dispatch<false>(a, (hipStream_t)stream);
}

// Sparse pixel set (gsplat::rasterize_to_pixels_sparse_bwd, reference RasterizeToPixelsSparseBwd.cu): render_alphas,
// last_ids and the cotangents are rows [P, ...]; gradient rows v_rows as in gsx_raster3d_bwd.
extern "C" int gsx_raster3d_sparse_bwd(
    const float *means2d, const float *conics, const float *colors, const float *opacities, const float *backgrounds,
    const uint8_t *masks, const int32_t *active_tiles, const int32_t *tile_offsets, const int32_t *flatten_ids,
    const uint64_t *tile_pixel_mask, const int64_t *tile_pixel_cumsum, const int64_t *pixel_map, uint32_t n_active,
    uint32_t words_per_tile, const float *render_alphas, const int32_t *last_ids, const float *v_render_colors,
    const float *v_render_alphas, uint32_t n_images, uint32_t n_isects, uint32_t cdim, uint32_t width, uint32_t height,
    uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int has_abs, float *v_rows, uint32_t row_stride, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(tile_size >= 1 && tile_size <= 16, "gsx_raster3d_sparse_bwd: tile_size must be in [1,16], got %u", tile_size);
    GSX_REQUIRE(cdim >= 1, "gsx_raster3d_sparse_bwd: channels must be >= 1");
    if (n_isects == 0) return GSX_OK; // no intersections: nothing to add to the (zero-filled, possibly empty) gradient rows
    GSX_REQUIRE(v_rows, "gsx_raster3d_sparse_bwd: null gradient output");
    GSX_REQUIRE(row_stride >= 6u + (has_abs ? 2u : 0u) + cdim, "gsx_raster3d_sparse_bwd: row_stride %u too small", row_stride);
    if (n_active == 0 || n_isects == 0) return GSX_OK;
    GSX_REQUIRE(words_per_tile * 64u >= tile_size * tile_size, "gsx_raster3d_sparse_bwd: pixel mask too narrow");
    GSX_REQUIRE(active_tiles && tile_offsets && tile_pixel_mask && tile_pixel_cumsum && pixel_map && means2d && conics
                && colors && opacities && flatten_ids && render_alphas && last_ids && v_render_colors,
                "gsx_raster3d_sparse_bwd: null input");
    Raster3DArgs a{};
    a.n_images = n_images; a.n_isects = n_isects; a.width = width; a.height = height;
    a.tile_size = tile_size; a.tile_w = tile_w; a.tile_h = tile_h; a.cdim = cdim;
    a.means2d = means2d; a.conics = conics; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks; a.isect_offsets = tile_offsets; a.flatten_ids = flatten_ids;
    a.render_alphas = const_cast<float *>(render_alphas); a.last_ids = const_cast<int32_t *>(last_ids);
    a.v_render_colors = v_render_colors; a.v_render_alphas = v_render_alphas;
    a.v_rows = v_rows; a.row_stride = row_stride;
    a.sp_active_tiles = active_tiles; a.sp_pixel_mask = tile_pixel_mask; a.sp_pixel_cumsum = tile_pixel_cumsum;
    a.sp_pixel_map = pixel_map; a.n_active = n_active; a.sp_words = words_per_tile;
    return has_abs ? bwd_dispatch<true>(a, (hipStream_t)stream) : bwd_dispatch<false>(a, (hipStream_t)stream);
}
