// 3DGS alpha compositing, backward, WIDE colour rows (5 .. 32 channels per launch) on the matrix cores (gfx950).
// Launched by gsx_raster3d_bwd_ws (raster3d_bwd.hip) for 16 x 16 tiles without absgrad; replaces the same reference kernel,
// RasterizeToPixels3DGSSerialBatchBwd.cu:41-320 (its channel list goes up to 512: Config.h:71-72; the trainer's feature
// renders and the reference's published "32 ch" profile rows run here).
//
// With D colour channels the per-(tile, Gaussian) sums of the backward are
//     v_colour[g][k] = sum over pixels p of  fac(p, g) * v_c(p, k)            D columns
//     M_m[g]         = sum over pixels p of  w(p, g)   * phi_m(p)             6 moments (phi = 1, u, v, u^2, u v, v^2)
// i.e. TWO MATRIX PRODUCTS  [Gaussians x pixels] . [pixels x (D | 6)]  whose right-hand sides depend on the pixel alone. The
// reduction kernel (raster3d_bwd_kernel) forms them with D + 6 wave reductions per (wave, Gaussian) pair - at 32 channels 100
// half-rate cross-lane instructions per pair, 0.026 of the HBM roofline on the reference's 32-channel profile - and the
// one-wave kernels (variants T / W) would need 16 x D cotangent registers per lane for their turn. Here the pixel loop only
// produces the two scalars per (pixel, Gaussian), fac = alpha T and w = v_sigma, and parks them in a wave-private LDS matrix;
// every SIXTEEN surviving Gaussians the wave multiplies:
//   * v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact fp32, the same fmaf chain a scalar loop would run;
//     MI355X_MICROARCH.md "FP32-input MFMA"): M = 16 Gaussians, N = 16 columns, K = 4 pixels per instruction, 16 k-steps
//     over the wave's 64 pixels; two N-blocks of colour columns + one block holding the six moment columns = 48 instructions
//     per 16 Gaussians on the MFMA pipe, which runs beside the vector ALU;
//   * the A operand (lane l: row l & 15, k = l >> 4) is the parked matrix read back with four ds_read_b128 per lane - the
//     k-step s contracts pixels {s, 16 + s, 32 + s, 48 + s}, so a lane's sixteen A values are contiguous; the B operands
//     (cotangents of those pixels, the moment polynomials) sit in registers for the whole tile;
//   * the result rows land as (column = l & 15, rows 4 (l >> 4) .. + 3): colour columns are added to the gradient rows in HBM
//     straight from the accumulators (consecutive lanes, consecutive floats of one row), the six moments go to a per-tile LDS
//     row per Gaussian and become v_mean2d / v_conic / v_opacity in the flush (moments about the tile centre, as variant T).
// The D-term dot product c . v_c that the gradient of alpha needs per (pixel, Gaussian) is a THIRD product of the same shape,
// [Gaussians x channels] . [channels x pixels], taken for the sixteen Gaussians of a group before their pixel walk and handed
// to the pixels through the same parked rows (the pixel's lane reads its cell, then overwrites it with fac).
// No cross-lane reduction anywhere; the pixel loop is the branch-free body of variant T with one LDS read for the dot product.
// Four waves per tile (wave = 8 x 8 quadrant, lane = pixel), batches of 64 staged Gaussians shared through LDS.
// This is the one place of the backend that uses MFMA: the 3-channel path has no dense contraction.
#include <cstdlib>

#include "raster3d.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef GSX_BWD_M_WAVES
#define GSX_BWD_M_WAVES 3
#endif
#ifndef GSX_BWD_M_DBG // timing ablations (A/B builds only; results are wrong when set): 1 no colour atomics, 2 no gradient
#define GSX_BWD_M_DBG 0 // MFMAs, 4 no dot-product MFMAs, 8 no pixel walk, 16 no geometry flush, 32 no colour staging
#endif
#ifndef GSX_BWD_M_ILP // survivors whose state-independent half is formed together (see the walk)
#define GSX_BWD_M_ILP 1
#endif

template <int NB> // 16-column blocks of colour channels per launch: 1 (<= 16 channels) or 2 (<= 32)
struct BwdMCfg {
    static constexpr int CHP   = 16 * NB; // padded channel count
    static constexpr int KC    = CHP / 4; // k-steps of the dot-product multiplication: channels {s, KC + s, 2 KC + s, 3 KC + s}
    static constexpr int BATCH = 64;      // staged Gaussians per batch: one per lane of the culling test
    static constexpr int SLOTS = 16;      // Gaussians per multiplication (the M of the MFMA)
    static constexpr int WP    = 68;      // floats per parked row: 64 pixels + 4 (the b128 A-operand reads of 16 rows spread over all banks)
    static constexpr int KA    = 8;       // floats per accumulator row: S0 Su Sv Suu Suv Svv + 2 pad
    static constexpr int CP    = CHP + 4; // floats per staged colour row: + 4 so that the rows of a wave's b128 stores / reads spread over the banks
    static constexpr size_t smem = (size_t)BATCH * (sizeof(StagedRow) + sizeof(float4) + 2 * sizeof(int32_t) + sizeof(float) * (CP + KA))
                                   + sizeof(float) * 4 * 2 * SLOTS * WP;
};

template <int NB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSX_BWD_M_WAVES)))
raster3d_bwd_m_kernel(const Raster3DArgs a)
{
    using Cfg           = BwdMCfg<NB>;
    constexpr int CHP   = Cfg::CHP;
    constexpr int KC    = Cfg::KC;
    constexpr int BATCH = Cfg::BATCH;
    constexpr int SLOTS = Cfg::SLOTS;
    constexpr int WP    = Cfg::WP;
    constexpr int KA    = Cfg::KA;
    constexpr int CP    = Cfg::CP;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    StagedRow *s_st  = reinterpret_cast<StagedRow *>(smem_raw);      // e-form of the exponent (raster3d.hpp); its colour fields are unused
    float4 *s_cull   = reinterpret_cast<float4 *>(s_st + BATCH);     // mean - tile centre, half extents of alpha >= 1/255
    int32_t *s_id    = reinterpret_cast<int32_t *>(s_cull + BATCH);  // flatten id of the row
    int32_t *s_touch = s_id + BATCH;
    float *s_col     = reinterpret_cast<float *>(s_touch + BATCH);   // [BATCH][CP] colours, zero padded to CHP
    float *s_acc     = s_col + BATCH * CP;                          // [BATCH][KA] tile-centre moments
    float *s_w       = s_acc + BATCH * KA;                           // [4 waves][2][SLOTS][WP]: (c . v_c, then fac) | w per (slot, pixel)

    TileCtx tc;
    if (a.tile_order ? !tile_context_ordered(a, blockIdx.x, tc) : !tile_context(a, blockIdx.x, tc)) return;
    if (a.masks && !a.masks[(size_t)tc.image_id * (a.tile_w * a.tile_h) + tc.tile_id]) return;
    const int32_t range_start = tc.range_start;
    if (tc.range_end <= range_start) return;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lx, ly;
    tile_pixel(tid, 16u, lx, ly);
    const int64_t prow = pixel_row(a, tc, blockIdx.x, lx, ly);
    const bool inside  = prow >= 0;
    const size_t pix   = inside ? (size_t)prow : 0;
    const float tile_cx = (float)(tc.tile_x * 16u) + 8.0f, tile_cy = (float)(tc.tile_y * 16u) + 8.0f;
    const float pu = (float)lx - 7.5f, pv = (float)ly - 7.5f; // this lane's pixel centre relative to the tile centre (exact)

    const float T_final     = inside ? 1.0f - a.render_alphas[pix] : 1.0f;
    float T                 = T_final;
    const int32_t bin_final = inside ? a.last_ids[pix] : -1;
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    int32_t range_end = tc.range_end;
    { // nothing behind the tile's last contributor is staged (as the other variants)
        int32_t *s_m = reinterpret_cast<int32_t *>(smem_raw);
        if (lane == 0) s_m[wave] = wave_bin_final;
        __syncthreads();
        const int32_t m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        __syncthreads();
        range_end = min(range_end, m + 1);
    }
    const int32_t n_batches = (range_end - range_start + BATCH - 1) / BATCH;
    if (n_batches <= 0) return; // uniform: no pixel of the tile has a contributor

    float *s_ww = s_w + wave * (2 * SLOTS * WP); // this wave's parked matrices: rows [0, SLOTS) dot product then fac, rows [SLOTS, 2 SLOTS) w
    const int bj = (int)(lane & 15u), bk = (int)(lane >> 4); // this lane's column / k-group in every MFMA operand
    // ---- B operands of the three multiplications, in registers for the whole tile ---------------------------------------------
    //  gradient products (contraction over the wave's 64 pixels, k-step s = pixels {s, 16 + s, 32 + s, 48 + s}):
    //     bcol[nb][s] = cotangent of pixel 16 bk + s, channel 16 nb + bj;   bphi[s] = moment polynomial bj of that pixel
    //  dot-product product c . v_c (contraction over the channels, k-step s = channels {s, KC + s, 2 KC + s, 3 KC + s}):
    //     bdot[pb][s] = cotangent of pixel 16 pb + bj, channel KC bk + s
    float bcol[NB][16], bphi[16], bdot[4][KC];
    float tail_term;
    {
        float v_c[CHP]; // this pixel's cotangents, zero padded
        // contiguous rows of whole float4s: 16-byte loads (a lane's row is 4 * cdim bytes from its neighbour's, so every scalar
        // load instruction touches 64 cache lines)
        const bool vec4 = !a.vrc_strided && ((a.cdim | a.ch_off) & 3u) == 0u && (reinterpret_cast<uintptr_t>(a.v_render_colors) & 15u) == 0u;
        if (vec4) {
            const f32x4 *vp = reinterpret_cast<const f32x4 *>(a.v_render_colors + pix * a.cdim + a.ch_off);
#pragma unroll
            for (int q4 = 0; q4 < CHP / 4; ++q4) {
                f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
                if (inside && 4 * q4 < (int)a.nch) x = vp[q4]; // nch is a multiple of 4 here or the row's tail belongs to the next chunk: masked below
#pragma unroll
                for (int j = 0; j < 4; ++j) v_c[4 * q4 + j] = (4 * q4 + j < (int)a.nch) ? x[j] : 0.0f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CHP; ++k) v_c[k] = (inside && k < (int)a.nch) ? a.v_render_colors[vrc_index(a, pix, a.ch_off + (uint32_t)k)] : 0.0f;
        }
        const float v_a = (inside && a.first_chunk && a.v_render_alphas) ? a.v_render_alphas[pix] : 0.0f;
        float bg_dot    = 0.0f;
        if (a.backgrounds) {
            const float *bg = a.backgrounds + (size_t)tc.image_id * a.cdim + a.ch_off;
#pragma unroll
            for (int k = 0; k < CHP; ++k)
                if (k < (int)a.nch) bg_dot += bg[k] * v_c[k];
        }
        tail_term = T_final * (v_a - bg_dot); // T_final (v_a - bg . v_c): what lies behind the whole list
        // the cotangents of the wave's 64 pixels change hands through the (still unused) parked-matrix region:
        // [channel quad][pixel][4] so that a pixel's row goes out as b128 stores
        f32x4 *tmp = reinterpret_cast<f32x4 *>(s_ww);
#pragma unroll
        for (int q4 = 0; q4 < CHP / 4; ++q4) tmp[q4 * 64 + (int)lane] = f32x4{v_c[4 * q4], v_c[4 * q4 + 1], v_c[4 * q4 + 2], v_c[4 * q4 + 3]};
        wave_lds_sync();
        const float *tf = reinterpret_cast<const float *>(tmp);
        auto cot = [&](int pp, int ch) { return tf[((ch >> 2) * 64 + pp) * 4 + (ch & 3)]; };
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int pp = 16 * bk + s;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bcol[nb][s] = cot(pp, 16 * nb + bj);
            // pixel pp of this wave = lane pp: (qx, qy) = (pp & 7, pp >> 3) inside the wave's quadrant
            const float u = (float)(((wave & 1u) << 3) | (uint32_t)(pp & 7)) - 7.5f;
            const float v = (float)(((wave >> 1) << 3) | (uint32_t)(pp >> 3)) - 7.5f;
            const float ph[6] = {1.0f, u, v, u * u, u * v, v * v};
            float val = 0.0f;
#pragma unroll
            for (int m = 0; m < 6; ++m) val = (bj == m) ? ph[m] : val;
            bphi[s] = val;
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int s = 0; s < KC; ++s) bdot[pb][s] = cot(16 * pb + bj, KC * bk + s);
        wave_lds_sync();
    }
    float behind        = 0.0f;                              // B = sum_k buffer_k v_c,k (raster3d_bwd.hip, variant T)
    const WaveRect rect = wave_pixel_rect(inside, pu, pv);   // tile-centre coordinates, like s_cull

    for (int s = (int)tid; s < BATCH; s += (int)blockDim.x) {
#pragma unroll
        for (int k = 0; k < KA; ++k) s_acc[s * KA + k] = 0.0f;
        s_touch[s] = 0;
    }

    for (int32_t b = 0; b < n_batches; ++b) {
        // back to front: staged slot s is list entry batch_end - s
        const int32_t batch_end  = range_end - 1 - BATCH * b;
        const int32_t batch_size = min(BATCH, batch_end + 1 - range_start);
        // staging by the whole workgroup: thread (entry s = tid & 63, part = tid >> 6) takes a quarter of entry s's colour row
        // (b128 stores into rows of CP floats: one wave's 64 rows spread over the banks - a [BATCH][32] table written by one wave
        // with a 128-byte lane stride cost 2048 conflict cycles per batch); part 0 also stages the geometry
        {
            const int s = (int)(tid & 63u), part = (int)(tid >> 6);
            const int32_t idx = batch_end - s;
            if (idx >= range_start) {
                const int32_t g = a.flatten_ids[idx];
                if (part == 0) {
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
                    s_st[s].p0     = p0;
                    s_st[s].p1     = v4f{nA, nB, nC, 0.0f};
                }
                constexpr int Q = CHP / 4; // channels per part
                const float *c  = a.colors + (size_t)g * a.cdim + a.ch_off + Q * part;
                float cv[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) cv[k] = (!(GSX_BWD_M_DBG & 32) && Q * part + k < (int)a.nch) ? c[k] : 0.0f;
                f32x4 *dst = reinterpret_cast<f32x4 *>(s_col + s * CP + Q * part);
#pragma unroll
                for (int h = 0; h < Q / 4; ++h) dst[h] = f32x4{cv[4 * h], cv[4 * h + 1], cv[4 * h + 2], cv[4 * h + 3]};
            }
        }
        __syncthreads();

        const int32_t t_first  = __builtin_amdgcn_readfirstlane(max(0, batch_end - wave_bin_final));
        const int32_t behind_s = __builtin_amdgcn_readfirstlane(batch_end); // list index of staged slot t = behind_s - t
        bool hit = false;
        if ((int32_t)lane >= t_first && (int32_t)lane < batch_size) {
            const float4 cu = s_cull[lane];
            hit = (fabsf(cu.x - rect.cx) - rect.hw <= cu.z) && (fabsf(cu.y - rect.cy) - rect.hh <= cu.w);
            if (hit) hit = rect_reaches_level(s_st[lane].p0, s_st[lane].p1, cu.x, cu.y, rect); // exact second stage
        }
        uint64_t todo = __builtin_amdgcn_ballot_w64(hit);
        // The survivors are taken SIXTEEN at a time (the rows of one multiplication), back to front:
        while (todo) {
            // (1) slots: lane k of slot_t <- staged index of the k-th survivor of the group. A survivor IS a lane of the culling
            // test (staged index = lane): it counts the survivors below it (v_mbcnt) and, if that rank is < 16, sends its index to
            // lane `rank` (ds_permute: a push through the LDS crossbar, no memory) - one step for the group, not a scalar loop of 16.
            const int n_todo = (int)__builtin_popcountll(todo);
            const int n      = n_todo < SLOTS ? n_todo : SLOTS;
            const int rank   = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
            const bool take  = ((todo >> lane) & 1ull) && rank < SLOTS;
            // lanes that are not taken push to lanes >= 32, which nobody reads
            const int slot_t = __builtin_amdgcn_ds_permute((take ? rank : 32 + (int)(lane & 31u)) << 2, (int)lane);
            if (n_todo <= SLOTS) todo = 0ull;
            else {
                const int32_t t_last = __builtin_amdgcn_readlane(slot_t, SLOTS - 1); // the 16th survivor: everything up to it is taken
                todo &= ~((2ull << t_last) - 1ull);
            }
            // (2) c . v_c for the group's 16 x 64 (Gaussian, pixel) pairs on the matrix cores: rows = slots, columns = pixels
            // (four blocks of 16), contraction over the channels. A = the staged colour row of slot bj, channels KC bk .. + KC - 1.
            {
                int t_a = __builtin_amdgcn_ds_bpermute(bj << 2, slot_t); // every lane runs the permute; lane k < n of slot_t is set
                t_a     = bj < n ? t_a : 0;
                float ac[KC];
                const f32x4 *pc = reinterpret_cast<const f32x4 *>(s_col + t_a * CP + KC * bk);
#pragma unroll
                for (int h = 0; h < KC / 4; ++h) {
                    const f32x4 x = pc[h];
                    ac[4 * h] = x.x; ac[4 * h + 1] = x.y; ac[4 * h + 2] = x.z; ac[4 * h + 3] = x.w;
                }
                f32x4 d[4];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) d[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !(GSX_BWD_M_DBG & 4)
#pragma unroll
                for (int s = 0; s < KC; ++s)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) d[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s], bdot[pb][s], d[pb], 0, 0, 0);
#endif
                // result (slot 4 bk + i, pixel 16 pb + bj) -> the cell the pixel's lane reads in (3) and overwrites with fac
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_ww[(4 * bk + i) * WP + 16 * pb + bj] = d[pb][i];
                wave_lds_sync();
            }
            // (3) the pixel walk over the group's Gaussians, GSX_BWD_M_ILP at a time: what does not depend on the pixel's running
            // state (staged row, exponent, alpha, the parked dot product) is formed for the small group first - independent
            // instruction streams: at three waves per SIMD the kernel waits on exactly these round trips - then the
            // transmittance recurrence and the parking stores run Gaussian by Gaussian. Branch-free: a Gaussian that no pixel of
            // the wave takes parks zeros (its rows of the products are zero; `dead` keeps it out of the atomics).
            uint32_t dead = 0; // wave-uniform: slots no pixel of this wave took
            struct Front { float ov_r, al_r, cv; bool valid; };
            for (int k0 = 0; k0 < ((GSX_BWD_M_DBG & 8) ? 0 : n); k0 += GSX_BWD_M_ILP) {
                Front fr[GSX_BWD_M_ILP];
#pragma unroll
                for (int i = 0; i < GSX_BWD_M_ILP; ++i) {
                    const int k     = min(k0 + i, n - 1); // the tail repeats the last slot: its results are dropped below
                    const int32_t t = __builtin_amdgcn_readlane(slot_t, k);
                    const v4f p0 = s_st[t].p0;
                    const v4f p1 = s_st[t].p1;
                    const float e = staged_f(p0, p1.x, p1.y, p1.z, pu, pv);
                    fr[i].ov_r = __builtin_amdgcn_exp2f(e); // opac * exp(-sigma), unclamped
                    fr[i].al_r = fminf(kMaxAlpha, fr[i].ov_r);
                    // lanes outside the image have bin_final = -1 and can never be valid; e > lo <=> sigma < 0
                    fr[i].valid = (bin_final >= behind_s - t) && !(e > p0.w) && !(fr[i].al_r < kAlphaThreshold);
                    fr[i].cv    = s_ww[k * WP + (int)lane];
                }
#pragma unroll
                for (int i = 0; i < GSX_BWD_M_ILP; ++i) {
                    const int k = k0 + i;
                    if (k >= n) break; // wave-uniform
                    const Front &f = fr[i];
                    if (__builtin_amdgcn_ballot_w64(f.valid) == 0ull) dead |= 1u << k;
                    // invalid lanes: alpha = 0 -> fac = 0, w = 0, T and `behind` unchanged (1 / (1 - 0) == 1 exactly)
                    const float alpha = f.valid ? f.al_r : 0.0f;
                    const float ra    = __builtin_amdgcn_rcpf(1.0f - alpha); // alpha <= kMaxAlpha = 0.99: no guard needed
                    T                *= ra;
                    const float fac   = alpha * T;
                    // v_alpha = sum_k (c_k T - buffer_k / (1 - alpha)) v_c,k + T_final / (1 - alpha) (v_a - bg . v_c); only
                    // B = sum_k buffer_k v_c,k is ever used and it obeys B += fac (c . v_c)  (raster3d_bwd.hip, variant T)
                    const float v_alpha = fmaf(ra, tail_term - behind, f.cv * T);
                    behind              = fmaf(fac, f.cv, behind);
                    // alpha-clamp branch (opac exp(-sigma) > 0.99): no geometry gradient; invalid lanes: none either
                    const float v_sigma = (f.valid && f.ov_r <= kMaxAlpha) ? -f.ov_r * v_alpha : 0.0f;
                    s_ww[k * WP + (int)lane]           = fac;
                    s_ww[(SLOTS + k) * WP + (int)lane] = v_sigma;
                }
            }
            wave_lds_sync();
            // (4) the gradient products: colour columns -> HBM, moments -> s_acc
            {
                float af[16], aw[16]; // A operands: row bj (slot), pixels 16 bk .. + 15 - contiguous in the parked row
                const f32x4 *pf = reinterpret_cast<const f32x4 *>(s_ww + bj * WP + 16 * bk);
                const f32x4 *pw = reinterpret_cast<const f32x4 *>(s_ww + (SLOTS + bj) * WP + 16 * bk);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f32x4 x = pf[h], y = pw[h];
                    af[4 * h] = x.x; af[4 * h + 1] = x.y; af[4 * h + 2] = x.z; af[4 * h + 3] = x.w;
                    aw[4 * h] = y.x; aw[4 * h + 1] = y.y; aw[4 * h + 2] = y.z; aw[4 * h + 3] = y.w;
                }
                f32x4 acc_c[NB], acc_m = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc_c[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !(GSX_BWD_M_DBG & 2)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc_c[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bcol[nb][s], acc_c[nb], 0, 0, 0);
                    acc_m = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], bphi[s], acc_m, 0, 0, 0);
                }
#endif
                // results: column bj, rows 4 bk + i (C/D map of the 16x16 forms: col = lane & 15, row = 4 (lane >> 4) + i)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * bk + i;
                    const int t_g = __builtin_amdgcn_ds_bpermute(row << 2, slot_t); // staged index of the Gaussian in slot `row`
                    if (row < n && !((dead >> row) & 1u)) {
                        float *grow = a.v_rows + (size_t)s_id[t_g] * a.row_stride + 6 + a.ch_off;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int ch = 16 * nb + bj;
#if !(GSX_BWD_M_DBG & 1)
                            if (ch < (int)a.nch) atomic_add_f32(grow + ch, acc_c[nb][i]);
#else
                            if (ch < (int)a.nch && acc_c[nb][i] == 1234.5f) atomic_add_f32(grow + ch, acc_c[nb][i]);
#endif
                        }
                        if (bj < 6) atomicAdd(&s_acc[t_g * KA + bj], acc_m[i]); // ds_add_f32: six lanes per Gaussian
                        if (bj == 6) s_touch[t_g] = 1;
                    }
                }
                wave_lds_sync(); // the next group's dot products overwrite the parked rows
            }
        }
        __syncthreads();

        // flush of the geometry columns, transposed (raster3d_bwd.hip, variant T): raw tile-centre moments -> moments of
        // d = mean - pixel = a - (u, v):  S_x = ax S0 - Su, S_xx = ax^2 S0 - 2 ax Su + Suu, S_xy = ax ay S0 - ax Sv - ay Su + Suv, ...
        // Eight consecutive lanes own one accumulator row (six of them a gradient column each): consecutive lanes add consecutive
        // floats of one row in HBM, and because a row's readers sit in ONE wave they can zero it themselves right after (LDS
        // operations of a wave execute in order) - no barrier between the flush and the re-zeroing, no separate zero pass.
        constexpr float kInvLog2e = 1.0f / kLog2e;
        for (int e = (int)tid; e < ((GSX_BWD_M_DBG & 16) ? 0 : BATCH * KA); e += (int)blockDim.x) {
            const int s = e >> 3, c = e & 7;
            const bool live = s < batch_size && s_touch[s] != 0; // the same for the eight lanes of a row
            float val = 0.0f;
            if (live && c < 6) {
                const float *row = s_acc + s * KA;
                const float4 cu  = s_cull[s];
                const float ax = cu.x, ay = cu.y;
                const float S0 = row[0], Su = row[1], Sv = row[2];
                if (c < 2) {
                    const v4f p1 = s_st[s].p1; // (-A, -B, -C) of the staged form: Q = (2A, B; B, 2C) / log2(e)
                    const float sx = fmaf(ax, S0, -Su), sy = fmaf(ay, S0, -Sv);
                    val = -kInvLog2e * ((c == 0) ? (2.0f * p1.x * sx + p1.y * sy) : (p1.y * sx + 2.0f * p1.z * sy));
                } else if (c == 2) {
                    val = 0.5f * (ax * (ax * S0 - 2.0f * Su) + row[3]);
                } else if (c == 3) {
                    val = ax * (ay * S0 - Sv) - ay * Su + row[4];
                } else if (c == 4) {
                    val = 0.5f * (ay * (ay * S0 - 2.0f * Sv) + row[5]);
                } else {
                    val = -S0 * __builtin_amdgcn_exp2f(kLoMargin - s_st[s].p0.w); // v_opacity = sum vis v_alpha = -S_w / opacity
                }
                atomic_add_f32(a.v_rows + (size_t)s_id[s] * a.row_stride + c, val);
            }
            wave_lds_sync(); // every lane of the row has read it
            if (live) {
                s_acc[s * KA + c] = 0.0f;
                if (c == 7) s_touch[s] = 0;
            }
        }
        // the next batch's staging writes s_st / s_cull / s_col / s_id, which the flush above reads: one barrier in between; the
        // staging barrier then orders the zeroed rows before the next ds_adds
        __syncthreads();
    }
}

// GSX_RASTER3D_BWD_WIDE=r keeps the reduction kernel for wide colour rows (A/B; read once per process)
static bool bwd_m_enabled()
{
    static const bool on = [] {
        const char *e = getenv("GSX_RASTER3D_BWD_WIDE");
        return !(e && (e[0] == 'r' || e[0] == 'R' || e[0] == '0'));
    }();
    return on;
}
bool raster3d_bwd_m_applies(const Raster3DArgs &a, bool has_abs)
{
    return bwd_m_enabled() && !has_abs && a.tile_size == 16 && a.nch > 4 && a.nch <= 32;
}
int raster3d_bwd_m_launch(const Raster3DArgs &a, hipStream_t stream)
{
    const uint32_t n_blocks = a.sp_active_tiles ? a.n_active : a.tile_w * a.tile_h * a.n_images;
    if (n_blocks == 0 || a.n_isects == 0) return GSX_OK;
    const uint32_t grid = ((n_blocks + 7u) / 8u) * 8u;
    if (a.nch <= 16) raster3d_bwd_m_kernel<1><<<dim3(grid), dim3(256), BwdMCfg<1>::smem, stream>>>(a);
    else raster3d_bwd_m_kernel<2><<<dim3(grid), dim3(256), BwdMCfg<2>::smem, stream>>>(a);
    return check_launch("raster3d_bwd_m");
}

} // namespace gsx
