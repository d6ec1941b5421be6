// Binned tile intersection: isect_tiles(sort=True) + isect_offset_encode with every tile list assembled, sorted and written
// by the ONE workgroup that owns the tile - no scattered 8-byte stores, no [chunk][tile] table, no intermediate pair array.
//
// Replaces, for the same outputs bit for bit, the reference's count -> cumsum -> emit -> 6-pass device radix sort -> offsets
// (gsplat/cuda/csrc/IntersectTile.cu:214-464, 1078-1121, 925-988; host Intersect.cpp:170-329) and this backend's earlier
// Gaussian-major fused path (isect_fused.hip, kept as the fallback).
//
// The screen is cut into BINS of bw x bh tiles (bw * bh <= 16: a bin's tiles are the bits of a 16-bit mask).
//
//   before the host learns n_isects
//   A  bin_rect      per row: the walk's tile rectangle (walk_prepare) -> rectangle of bins, a 32-byte record
//                    (mean, conic | radii, opacity, depth) and an LDS histogram of bins per chunk of rows -> table[chunk][bin]
//   -  colscan       running sum over an image's chunks per bin, bin totals (tile_sort.hip: launch_colscan)
//   B  bin_plan      one workgroup: scan of the bin totals -> bin_start; blocks of 1024 entries per bin; zero fills
//   C  bin_scatter   per row and overlapped bin: entry = row id, slot from an LDS cursor (bin_start + chunk prefix)
//   D  bin_mask      per ENTRY (row, bin): the walk clipped to the bin (walk_clipped: only the slabs inside it) -> 16-bit
//                    mask of the bin's tiles the Gaussian touches; depth copied next to it; per-tile counts by wave
//                    ballots -> 16 LDS counters per block -> 16 global adds per block; tiles_per_gauss
//   E  tile_plan     one workgroup: scan of the tile counts -> isect_offsets, n_isects (pinned host word); cuts every
//                    bin into ITEMS = runs of its tiles whose lists fit the LDS sort together
//   after the host allocated the exact-length outputs
//   F  bin_emit      one workgroup per item: gathers the bin's entries that touch the item's tiles, sorts them ONCE by
//                    (depth, row) in LDS (64-bit keys + the mask as payload), then splits the sorted run into the tiles'
//                    lists with ballots + prefix sums (a stable split of a sorted sequence is sorted) and writes each
//                    list contiguously: (image|tile|depth) keys and row ids
//   -  tiles longer than the LDS sort go through the work-list sort of tile_sort.hip (launch_big_tile_sort)
//
// Sorting entries instead of intersections means sorting V * dup (~1.5 V) elements instead of M (~4 V), and the per-tile
// sort, the scatter and the offsets of the earlier path collapse into kernel F.
// Compiled with -ffp-contract=off (the walk must be bit-exact with the oracle; isect_walk.hpp / isect_binwalk.hpp).
#include "isect_binwalk.hpp"
#include "isect_fused.hpp"
#include <cstdlib>

namespace gsx {

constexpr int kBnThreads  = 256;
constexpr int kBnBlock    = 1024;  // entries per block of kernel D
constexpr int kBnMaxBins  = 16384; // bins in total (images x bins per image)
constexpr uint32_t kBnMaxTiles = 36864;

struct BinHeader { // device memory, written by B / E / F
    int32_t overflow, n_entries, n_blocks, n_items, big_count, pad[3];
};

struct BinGeom {
    int64_t rows, rows_per_image, cap_entries;
    uint32_t n_images, cpi, rpc, n_chunks;
    uint32_t tile_size, tile_w, tile_h, n_tiles;
    uint32_t bw, bh, bins_x, bins_y, n_bins, n_bins_total, max_blocks, tile_bits, item_cap;
};

struct BinBuffers {
    BinHeader *hdr;
    v4f *records;        // [rows][2]
    uint32_t *brect;     // [rows] x0 | y0 << 8 | x1 << 16 | y1 << 24 (bins)
    int32_t *table;      // [n_chunks][n_bins]
    int32_t *bin_count;  // [n_bins_total]
    int32_t *bin_start;  // [n_bins_total + 1]
    int32_t *blk_start;  // [n_bins_total + 1]
    int32_t *blk_bin;    // [max_blocks]
    uint32_t *e_row;     // [cap] row | multi << 31
    uint32_t *e_depth;   // [cap]
    uint16_t *e_mask;    // [cap]
    int32_t *tile_count; // [n_images * n_tiles]
    int2 *items;         // [n_bins_total * 16]
    int32_t *big_list;   // [n_images * n_tiles] tiles longer than the LDS sort (kernel F -> work-list sort)
};

struct BinArgs {
    BinGeom g;
    BinBuffers b;
    const float *means2d;
    const int32_t *radii;
    const float *depths;
    const float *conics;
    const float *opacities;
    const uint8_t *tile_mask;
    int32_t *tiles_per_gauss;
    int32_t *isect_offsets;
    int64_t *n_isects;
    // emit
    uint64_t *keys_out;
    int32_t *vals_out;
    uint2 *bucketed;
    int32_t *big_count, *big_list;
};

__device__ __forceinline__ void bn_chunk_rows(const BinGeom &g, uint32_t chunk, int64_t &lo, int64_t &hi, uint32_t &img)
{
    img                = chunk / g.cpi;
    const uint32_t sub = chunk % g.cpi;
    lo                 = (int64_t)img * g.rows_per_image + (int64_t)sub * g.rpc;
    hi                 = min(lo + (int64_t)g.rpc, (int64_t)(img + 1) * g.rows_per_image);
}

// ---- A: rectangle of bins per row, record, histogram ------------------------------------------------------------------
__global__ void __launch_bounds__(kBnThreads) bin_rect_kernel(const BinArgs a)
{
    extern __shared__ int32_t s_hist[];
    const BinGeom &g = a.g;
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kBnThreads) s_hist[i] = 0;
    __syncthreads();
    int64_t lo, hi;
    uint32_t img;
    bn_chunk_rows(g, blockIdx.x, lo, hi, img);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    constexpr int kU = 4;
    for (int64_t base = lo; base < hi; base += kBnThreads * kU) {
        float mx[kU], my[kU], rx[kU], ry[kU], dp[kU], A[kU], B[kU], C[kU], op[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) { // every load of the thread's rows is in flight before any is used
            const int64_t r = base + q * kBnThreads + threadIdx.x;
            rx[q] = ry[q] = 0.0f;
            mx[q] = my[q] = dp[q] = A[q] = B[q] = C[q] = op[q] = 0.0f;
            if (r < hi) {
                rx[q] = (float)a.radii[2 * r];
                ry[q] = (float)a.radii[2 * r + 1];
                mx[q] = a.means2d[2 * r];
                my[q] = a.means2d[2 * r + 1];
                dp[q] = a.depths[r];
                if (has_conic) {
                    A[q]  = a.conics[3 * r];
                    B[q]  = a.conics[3 * r + 1];
                    C[q]  = a.conics[3 * r + 2];
                    op[q] = a.opacities[r];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const int64_t r = base + q * kBnThreads + threadIdx.x;
            if (r >= hi) continue;
            uint32_t rect = 0;
            if (rx[q] > 0.0f && ry[q] > 0.0f) {
                const WalkPrep p = walk_prepare(mx[q], my[q], rx[q], ry[q], has_conic, A[q], B[q], C[q], op[q], g.tile_size,
                                                g.tile_w, g.tile_h);
                if (p.any) {
                    const uint32_t bx0 = (uint32_t)p.x0 / g.bw, bx1 = ((uint32_t)p.x1 + g.bw - 1) / g.bw;
                    const uint32_t by0 = (uint32_t)p.y0 / g.bh, by1 = ((uint32_t)p.y1 + g.bh - 1) / g.bh;
                    rect = bx0 | (by0 << 8) | (bx1 << 16) | (by1 << 24);
                    for (uint32_t by = by0; by < by1; ++by)
                        for (uint32_t bx = bx0; bx < bx1; ++bx) atomicAdd(&s_hist[by * g.bins_x + bx], 1);
                    v4f r0, r1;
                    r0.x = mx[q]; r0.y = my[q];
                    if (has_conic) { r0.z = A[q]; r0.w = B[q]; r1.x = C[q]; r1.y = op[q]; }
                    else { r0.z = rx[q]; r0.w = ry[q]; r1.x = 0.0f; r1.y = 0.0f; }
                    r1.z = dp[q]; r1.w = 0.0f;
                    a.b.records[2 * r]     = r0;
                    a.b.records[2 * r + 1] = r1;
                }
            }
            a.b.brect[r] = rect;
            if (a.tiles_per_gauss) a.tiles_per_gauss[r] = 0;
        }
    }
    __syncthreads();
    int32_t *out = a.b.table + (int64_t)blockIdx.x * g.n_bins;
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kBnThreads) out[i] = s_hist[i];
}

// ---- one-workgroup exclusive scan helper (1024 threads, thread-contiguous runs) ----------------------------------------
// in[0..n) -> out[0..n) exclusive; returns the grand total in every thread. f(i) maps the element before it is summed.
template <typename F>
__device__ __forceinline__ int64_t block_scan_1024(const int32_t *in, int32_t *out, uint32_t n, int64_t *s_part, F &&f)
{
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n);
    int64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += f(in[i]);
    // wave-level inclusive scan, then the 16 wave totals
    int64_t inc    = sum;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    int64_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int64_t v = s_part[w];
        if (w < wave) base += v;
        total += v;
    }
    int64_t run = base + inc - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        const int32_t v = f(in[i]);
        out[i]          = (int32_t)run;
        run += v;
    }
    __syncthreads(); // s_part may be reused
    return total;
}

// ---- B: bin starts, blocks, zero fills -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) bin_plan_kernel(const BinArgs a)
{
    __shared__ int64_t s_part[16];
    const BinGeom &g = a.g;
    const uint32_t nb = g.n_bins_total;
    const int64_t n_entries = block_scan_1024(a.b.bin_count, a.b.bin_start, nb, s_part, [](int32_t v) { return v; });
    const bool overflow     = n_entries > g.cap_entries;
    const int64_t n_blocks =
        block_scan_1024(a.b.bin_count, a.b.blk_start, nb, s_part, [](int32_t v) { return (v + kBnBlock - 1) / kBnBlock; });
    if (threadIdx.x == 0) {
        a.b.bin_start[nb] = (int32_t)(overflow ? 0 : n_entries);
        a.b.blk_start[nb] = (int32_t)n_blocks;
        a.b.hdr->overflow  = overflow ? 1 : 0;
        a.b.hdr->n_entries = overflow ? 0 : (int32_t)n_entries;
        a.b.hdr->n_blocks  = overflow ? 0 : (int32_t)n_blocks;
        a.b.hdr->n_items   = 0;
        a.b.hdr->big_count = 0;
    }
    __syncthreads(); // blk_start written by this workgroup is read back below
    __threadfence_block();
    if (!overflow)
        for (uint32_t b = threadIdx.x; b < nb; b += 1024) {
            const int32_t s = a.b.blk_start[b], n = (a.b.bin_count[b] + kBnBlock - 1) / kBnBlock;
            for (int32_t k = 0; k < n; ++k) a.b.blk_bin[s + k] = (int32_t)b;
        }
    const uint32_t nt = g.n_images * g.n_tiles;
    for (uint32_t t = threadIdx.x; t < nt; t += 1024) a.b.tile_count[t] = 0;
}

// ---- C: entries (row ids) grouped by bin -------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBnThreads) bin_scatter_kernel(const BinArgs a)
{
    extern __shared__ int32_t s_cur[];
    const BinGeom &g = a.g;
    if (a.b.hdr->overflow) return;
    int64_t lo, hi;
    uint32_t img;
    bn_chunk_rows(g, blockIdx.x, lo, hi, img);
    const int32_t *pre   = a.b.table + (int64_t)blockIdx.x * g.n_bins; // exclusive prefix over this image's chunks
    const int32_t *start = a.b.bin_start + (int64_t)img * g.n_bins;
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kBnThreads) s_cur[i] = start[i] + pre[i];
    __syncthreads();
    constexpr int kU = 4;
    for (int64_t base = lo; base < hi; base += kBnThreads * kU) {
        uint32_t rect[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const int64_t r = base + q * kBnThreads + threadIdx.x;
            rect[q]         = r < hi ? a.b.brect[r] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const int64_t r    = base + q * kBnThreads + threadIdx.x;
            const uint32_t bx0 = rect[q] & 255u, by0 = (rect[q] >> 8) & 255u, bx1 = (rect[q] >> 16) & 255u, by1 = rect[q] >> 24;
            const uint32_t multi = ((bx1 - bx0) * (by1 - by0) > 1u) ? 0x80000000u : 0u;
            for (uint32_t by = by0; by < by1; ++by)
                for (uint32_t bx = bx0; bx < bx1; ++bx) {
                    const int32_t slot = atomicAdd(&s_cur[by * g.bins_x + bx], 1);
                    a.b.e_row[slot]    = (uint32_t)r | multi;
                }
        }
    }
}

// ---- D: per entry, the mask of the bin's tiles; tile counts ----------------------------------------------------------
__global__ void __launch_bounds__(kBnThreads) bin_mask_kernel(const BinArgs a)
{
    __shared__ int32_t s_cnt[16];
    const BinGeom &g = a.g;
    const int32_t blk = (int32_t)blockIdx.x;
    if (blk >= a.b.hdr->n_blocks) return;
    const int32_t bin = a.b.blk_bin[blk];
    const int32_t e0  = a.b.bin_start[bin] + (blk - a.b.blk_start[bin]) * kBnBlock;
    const int32_t e1  = min(e0 + kBnBlock, a.b.bin_start[bin + 1]);
    const uint32_t img = (uint32_t)bin / g.n_bins, lb = (uint32_t)bin % g.n_bins;
    const int cx0 = (int)((lb % g.bins_x) * g.bw), cy0 = (int)((lb / g.bins_x) * g.bh);
    const int cx1 = min(cx0 + (int)g.bw, (int)g.tile_w), cy1 = min(cy0 + (int)g.bh, (int)g.tile_h);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)img * g.n_tiles : nullptr;
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    constexpr int kU = kBnBlock / kBnThreads;
    uint32_t row[kU];
    v4f r0[kU], r1[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
        const int32_t e = e0 + q * kBnThreads + (int32_t)threadIdx.x;
        row[q]          = e < e1 ? a.b.e_row[e] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < kU; ++q)
        if (row[q] != 0xFFFFFFFFu) {
            const int64_t r = (int64_t)(row[q] & 0x7FFFFFFFu);
            r0[q]           = a.b.records[2 * r];
            r1[q]           = a.b.records[2 * r + 1];
        }
    const int lane     = (int)(threadIdx.x & 63u);
    const int n_bits   = (int)(g.bw * g.bh);
    int32_t my_cnt     = 0; // lane t < n_bits: entries of this wave that touch tile t of the bin
#pragma unroll
    for (int q = 0; q < kU; ++q) {
        const int32_t e = e0 + q * kBnThreads + (int32_t)threadIdx.x;
        uint32_t mask   = 0;
        if (row[q] != 0xFFFFFFFFu) {
            const WalkPrep p = has_conic ? walk_prepare(r0[q].x, r0[q].y, 1.0f, 1.0f, true, r0[q].z, r0[q].w, r1[q].x, r1[q].y,
                                                        g.tile_size, g.tile_w, g.tile_h)
                                         : walk_prepare(r0[q].x, r0[q].y, r0[q].z, r0[q].w, false, 0.f, 0.f, 0.f, 0.f,
                                                        g.tile_size, g.tile_w, g.tile_h);
            walk_clipped(p, g.tile_size, cx0, cy0, cx1, cy1, [&](int x, int y) {
                if (tmask && !tmask[(size_t)y * g.tile_w + x]) return;
                mask |= 1u << ((y - cy0) * (int)g.bw + (x - cx0));
            });
            a.b.e_mask[e]  = (uint16_t)mask;
            a.b.e_depth[e] = __float_as_uint(r1[q].z);
            if (a.tiles_per_gauss && mask) {
                const int64_t r = (int64_t)(row[q] & 0x7FFFFFFFu);
                if (row[q] & 0x80000000u) atomicAdd(&a.tiles_per_gauss[r], (int32_t)__popc(mask));
                else a.tiles_per_gauss[r] = (int32_t)__popc(mask);
            }
        }
        for (int t = 0; t < n_bits; ++t) {
            const uint64_t b = __builtin_amdgcn_ballot_w64((mask >> t) & 1u);
            if (lane == t) my_cnt += (int32_t)__popcll(b);
        }
    }
    if (lane < n_bits && my_cnt) atomicAdd(&s_cnt[lane], my_cnt);
    __syncthreads();
    if ((int)threadIdx.x < n_bits && s_cnt[threadIdx.x]) {
        const int tx = cx0 + (int)threadIdx.x % (int)g.bw, ty = cy0 + (int)threadIdx.x / (int)g.bw;
        // a set bit is always a tile inside the grid (the walk is clamped to it)
        atomicAdd(&a.b.tile_count[(size_t)img * g.n_tiles + (size_t)ty * g.tile_w + tx], s_cnt[threadIdx.x]);
    }
}

// ---- E: offsets, n_isects, items -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_plan_kernel(const BinArgs a)
{
    __shared__ int64_t s_part[16];
    const BinGeom &g = a.g;
    if (a.b.hdr->overflow) {
        if (threadIdx.x == 0) *a.n_isects = -2; // the caller reruns the Gaussian-major path (gsx_isect_binned_count's contract)
        return;
    }
    const uint32_t nt   = g.n_images * g.n_tiles;
    const int64_t total = block_scan_1024(a.b.tile_count, a.isect_offsets, nt, s_part, [](int32_t v) { return v; });
    const int32_t cap   = (int32_t)g.item_cap;
    for (uint32_t bin = threadIdx.x; bin < g.n_bins_total; bin += 1024) {
        if (a.b.bin_count[bin] == 0) continue;
        const uint32_t img = bin / g.n_bins, lb = bin % g.n_bins;
        const uint32_t tx0 = (lb % g.bins_x) * g.bw, ty0 = (lb / g.bins_x) * g.bh;
        const int nbits    = (int)(g.bw * g.bh);
        int t0 = 0, acc = 0;
        auto flush = [&](int from, int to, int big) {
            const int32_t idx = atomicAdd(&a.b.hdr->n_items, 1);
            a.b.items[idx]    = make_int2((int)bin, from | (to << 8) | (big << 16));
        };
        for (int t = 0; t < nbits; ++t) {
            const uint32_t tx = tx0 + (uint32_t)t % g.bw, ty = ty0 + (uint32_t)t / g.bw;
            const int32_t c   = (tx < g.tile_w && ty < g.tile_h) ? a.b.tile_count[(size_t)img * g.n_tiles + (size_t)ty * g.tile_w + tx] : 0;
            if (c > cap) {
                if (acc > 0) flush(t0, t, 0);
                flush(t, t + 1, 1);
                t0 = t + 1; acc = 0;
            } else if (acc + c > cap) {
                flush(t0, t, 0);
                t0 = t; acc = c;
            } else acc += c;
        }
        if (acc > 0) flush(t0, nbits, 0);
    }
    if (threadIdx.x == 0) {
        __threadfence_system();
        *a.n_isects = total;
    }
}

// ---- F: per item: gather, sort once, split, write -------------------------------------------------------------------------
__device__ __forceinline__ int bn_phys(int i) { return i + (i >> 3); }
__device__ __forceinline__ void bn_cmpx(uint64_t &x, uint64_t &y, uint32_t &mx, uint32_t &my, bool up)
{
    const bool sw     = (x > y) == up;
    const uint64_t lo = sw ? y : x, hi = sw ? x : y;
    const uint32_t ml = sw ? my : mx, mh = sw ? mx : my;
    x = lo; y = hi; mx = ml; my = mh;
}

template <int G>
__device__ __forceinline__ void bn_bitonic_group(uint64_t *s, uint16_t *sm, int P, int k, int lj)
{
    constexpr int R = 1 << G;
    for (int t = threadIdx.x; t < (P >> G); t += kBnThreads) {
        const int i   = ((t >> lj) << (lj + G)) | (t & ((1 << lj) - 1));
        const bool up = (i & k) == 0;
        uint64_t e[R];
        uint32_t m[R];
#pragma unroll
        for (int b = 0; b < R; ++b) {
            e[b] = s[bn_phys(i | (b << lj))];
            m[b] = sm[bn_phys(i | (b << lj))];
        }
#pragma unroll
        for (int q = G - 1; q >= 0; --q)
#pragma unroll
            for (int b = 0; b < R; ++b)
                if (!(b & (1 << q))) bn_cmpx(e[b], e[b | (1 << q)], m[b], m[b | (1 << q)], up);
#pragma unroll
        for (int b = 0; b < R; ++b) {
            s[bn_phys(i | (b << lj))]  = e[b];
            sm[bn_phys(i | (b << lj))] = (uint16_t)m[b];
        }
    }
    __syncthreads();
}

template <int CAP>
__global__ void __launch_bounds__(kBnThreads) bin_emit_kernel(const BinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int kWords = CAP + CAP / 8;
    uint64_t *s_key = reinterpret_cast<uint64_t *>(smem_raw);
    uint16_t *s_msk = reinterpret_cast<uint16_t *>(s_key + kWords);
    uint16_t *s_out = s_msk + kWords;
    uint16_t *s_cnt = s_out + CAP; // [CAP / 64][16]
    __shared__ int32_t s_n;
    __shared__ int32_t s_tcnt[16], s_loc[17], s_goff[16];
    __shared__ uint64_t s_hi[16];
    const BinGeom &g = a.g;
    const int n_items = a.b.hdr->n_items;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int it = (int)blockIdx.x; it < n_items; it += (int)gridDim.x) {
        const int2 item = a.b.items[it];
        const int bin = item.x, t0 = item.y & 255, t1 = (item.y >> 8) & 255, big = item.y >> 16;
        const int32_t e0 = a.b.bin_start[bin], e1 = a.b.bin_start[bin + 1];
        const uint32_t img = (uint32_t)bin / g.n_bins, lb = (uint32_t)bin % g.n_bins;
        const uint32_t tx0 = (lb % g.bins_x) * g.bw, ty0 = (lb / g.bins_x) * g.bh;
        const uint32_t range = ((1u << t1) - 1u) & ~((1u << t0) - 1u);
        if (threadIdx.x == 0) s_n = 0;
        if (threadIdx.x < 16) {
            const int t       = (int)threadIdx.x;
            const uint32_t tx = tx0 + (uint32_t)t % g.bw, ty = ty0 + (uint32_t)t / g.bw;
            const bool in     = t >= t0 && t < t1 && tx < g.tile_w && ty < g.tile_h;
            const size_t gid  = (size_t)img * g.n_tiles + (size_t)ty * g.tile_w + tx;
            s_tcnt[t]         = in ? a.b.tile_count[gid] : 0;
            s_goff[t]         = in ? a.isect_offsets[gid] : 0;
            s_hi[t]           = in ? ((((uint64_t)img << g.tile_bits) | ((uint64_t)ty * g.tile_w + tx)) << 32) : 0ull;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t run = 0;
            for (int t = 0; t < 16; ++t) {
                s_loc[t] = run;
                run += s_tcnt[t];
            }
            s_loc[16] = run;
        }
        // gather the entries that touch [t0, t1)
        for (int32_t base = e0; base < e1; base += kBnThreads) {
            const int32_t e    = base + (int32_t)threadIdx.x;
            const uint32_t m   = e < e1 ? ((uint32_t)a.b.e_mask[e] & range) : 0u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0u);
            if (bal == 0ull) continue;
            int32_t wbase = 0;
            if (lane == (int)__builtin_ctzll(bal)) wbase = atomicAdd(&s_n, (int32_t)__popcll(bal));
            wbase = __builtin_amdgcn_readlane(wbase, (int)__builtin_ctzll(bal));
            if (m) {
                const int32_t j      = wbase + (int32_t)__popcll(bal & lt_mask);
                const uint32_t row   = a.b.e_row[e] & 0x7FFFFFFFu;
                const uint32_t depth = a.b.e_depth[e];
                if (big) a.bucketed[(int64_t)s_goff[t0] + j] = make_uint2(depth, row);
                else {
                    s_key[bn_phys(j)] = ((uint64_t)depth << 32) | row;
                    s_msk[bn_phys(j)] = (uint16_t)m;
                }
            }
        }
        __syncthreads();
        if (big) { // one tile longer than the LDS sort: its unsorted segment goes through the work-list sort
            if (threadIdx.x == 0) {
                const uint32_t tx = tx0 + (uint32_t)t0 % g.bw, ty = ty0 + (uint32_t)t0 / g.bw;
                a.big_list[atomicAdd(a.big_count, 1)] = (int32_t)(img * g.n_tiles + ty * g.tile_w + tx);
            }
            __syncthreads();
            continue;
        }
        const int n_sel = s_n;
        int lp = 7;
        while ((1 << lp) < n_sel) ++lp;
        const int P = 1 << lp;
        for (int i = n_sel + (int)threadIdx.x; i < P; i += kBnThreads) {
            s_key[bn_phys(i)] = ~0ull;
            s_msk[bn_phys(i)] = 0;
        }
        __syncthreads();
        // bitonic sort of (depth, row) with the mask as payload; phases k = 2, 4, 8 in registers on 8 consecutive words
        for (int t = threadIdx.x; t < (P >> 3); t += kBnThreads) {
            uint64_t e[8];
            uint32_t m[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                e[b] = s_key[bn_phys(8 * t + b)];
                m[b] = s_msk[bn_phys(8 * t + b)];
            }
#pragma unroll
            for (int lk = 1; lk <= 3; ++lk)
#pragma unroll
                for (int q = lk - 1; q >= 0; --q)
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (!(b & (1 << q))) {
                            const bool up = lk < 3 ? ((b & (1 << lk)) == 0) : ((t & 1) == 0);
                            bn_cmpx(e[b], e[b | (1 << q)], m[b], m[b | (1 << q)], up);
                        }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                s_key[bn_phys(8 * t + b)] = e[b];
                s_msk[bn_phys(8 * t + b)] = (uint16_t)m[b];
            }
        }
        __syncthreads();
        for (int lk = 4; lk <= lp; ++lk) {
            const int k = 1 << lk;
            for (int top = lk - 1; top >= 0;) {
                const int gsz = top + 1 < 3 ? top + 1 : 3;
                const int lj  = top - gsz + 1;
                if (gsz == 3) bn_bitonic_group<3>(s_key, s_msk, P, k, lj);
                else if (gsz == 2) bn_bitonic_group<2>(s_key, s_msk, P, k, lj);
                else bn_bitonic_group<1>(s_key, s_msk, P, k, lj);
                top -= gsz;
            }
        }
        // split: chunk c of 64 sorted entries, tile t -> count; prefix over chunks; positions
        const int n_chunks = (n_sel + 63) >> 6;
        for (int c = wave; c < n_chunks; c += kBnThreads / 64) {
            const int i      = c * 64 + lane;
            const uint32_t m = i < n_sel ? (uint32_t)s_msk[bn_phys(i)] : 0u;
            int32_t mine     = 0;
            for (int t = t0; t < t1; ++t) {
                const uint64_t b = __builtin_amdgcn_ballot_w64((m >> t) & 1u);
                if (lane == t) mine = (int32_t)__popcll(b);
            }
            if (lane >= t0 && lane < t1) s_cnt[c * 16 + lane] = (uint16_t)mine;
        }
        __syncthreads();
        if ((int)threadIdx.x >= t0 && (int)threadIdx.x < t1) {
            int32_t run = 0;
            for (int c = 0; c < n_chunks; ++c) {
                const int32_t v             = s_cnt[c * 16 + threadIdx.x];
                s_cnt[c * 16 + threadIdx.x] = (uint16_t)run;
                run += v;
            }
        }
        __syncthreads();
        for (int c = wave; c < n_chunks; c += kBnThreads / 64) {
            const int i      = c * 64 + lane;
            const uint32_t m = i < n_sel ? (uint32_t)s_msk[bn_phys(i)] : 0u;
            for (int t = t0; t < t1; ++t) {
                const uint64_t b = __builtin_amdgcn_ballot_w64((m >> t) & 1u);
                if ((m >> t) & 1u) s_out[s_loc[t] + (int32_t)s_cnt[c * 16 + t] + (int32_t)__popcll(b & lt_mask)] = (uint16_t)i;
            }
        }
        __syncthreads();
        const int n_out = s_loc[16];
        for (int p = threadIdx.x; p < n_out; p += kBnThreads) {
            int t = t0;
            while (t + 1 < t1 && p >= s_loc[t + 1]) ++t;
            const uint64_t key = s_key[bn_phys((int)s_out[p])];
            const int64_t dst  = (int64_t)s_goff[t] + (p - s_loc[t]);
            a.keys_out[dst]    = s_hi[t] | (key >> 32);
            a.vals_out[dst]    = (int32_t)(uint32_t)key;
        }
        __syncthreads(); // LDS is reused by the next item
    }
}

// ------------------------------------------------------------------------------------------------------------------------
static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static uint32_t bits_for(uint64_t count)
{
    uint32_t b = 0;
    if (count <= 1) return 0;
    uint64_t v = count - 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

static void bin_dims(uint32_t &bw, uint32_t &bh, uint32_t &cap)
{
    bw = 4; bh = 4; cap = 4096;
    if (const char *e = getenv("GSX_ISECT_BIN")) { // "WxH" (tiles), W * H <= 16: A/B switch
        unsigned w = 0, h = 0;
        if (sscanf(e, "%ux%u", &w, &h) == 2 && w >= 1 && h >= 1 && w * h <= 16) { bw = w; bh = h; }
    }
    if (const char *e = getenv("GSX_ISECT_CAP")) {
        const int c = atoi(e);
        if (c == 2048 || c == 4096) cap = (uint32_t)c;
    }
}

static bool bin_geometry(BinGeom &g, int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                         int64_t cap_entries)
{
    g = BinGeom{};
    g.rows = rows; g.n_images = n_images ? n_images : 1;
    g.rows_per_image = rows / g.n_images;
    bin_dims(g.bw, g.bh, g.item_cap);
    g.tile_size = tile_size; g.tile_w = tile_w; g.tile_h = tile_h; g.n_tiles = tile_w * tile_h;
    g.bins_x = (tile_w + g.bw - 1) / g.bw; g.bins_y = (tile_h + g.bh - 1) / g.bh;
    g.n_bins = g.bins_x * g.bins_y; g.n_bins_total = g.n_bins * g.n_images;
    g.tile_bits = bits_for(g.n_tiles);
    // chunks of >= 2048 rows, image-aligned, at most ~1024 in total (the [chunk][bin] table stays small)
    const int64_t max_cpi = 1024 / g.n_images > 0 ? 1024 / g.n_images : 1;
    int64_t cpi = (g.rows_per_image + 2047) / 2048;
    if (cpi < 1) cpi = 1;
    if (cpi > max_cpi) cpi = max_cpi;
    g.cpi = (uint32_t)cpi;
    g.rpc = (uint32_t)((g.rows_per_image + cpi - 1) / cpi);
    if (g.rpc == 0) g.rpc = 1;
    g.n_chunks    = g.cpi * g.n_images;
    g.cap_entries = cap_entries;
    g.max_blocks  = (uint32_t)(cap_entries / kBnBlock + g.n_bins_total + 1);
    return g.bins_x <= 255 && g.bins_y <= 255 && g.n_bins_total <= (uint32_t)kBnMaxBins
           && (uint64_t)g.n_images * g.n_tiles <= kBnMaxTiles && rows < (1ll << 28);
}

// entries the workspace is sized for: a row costs one entry per bin its tile rectangle overlaps
static int64_t bin_cap_entries(int64_t rows) { return 4 * rows + 262144; }

static int64_t bin_layout(const BinGeom &g, unsigned char *base, BinBuffers *b)
{
    unsigned char *p = base;
    auto take = [&](int64_t bytes) {
        unsigned char *q = p;
        p += align256(bytes);
        return q;
    };
    BinBuffers t{};
    t.hdr        = reinterpret_cast<BinHeader *>(take(sizeof(BinHeader)));
    t.records    = reinterpret_cast<v4f *>(take(g.rows * 32));
    t.brect      = reinterpret_cast<uint32_t *>(take(g.rows * 4));
    t.table      = reinterpret_cast<int32_t *>(take((int64_t)g.n_chunks * g.n_bins * 4));
    t.bin_count  = reinterpret_cast<int32_t *>(take((int64_t)g.n_bins_total * 4));
    t.bin_start  = reinterpret_cast<int32_t *>(take(((int64_t)g.n_bins_total + 1) * 4));
    t.blk_start  = reinterpret_cast<int32_t *>(take(((int64_t)g.n_bins_total + 1) * 4));
    t.blk_bin    = reinterpret_cast<int32_t *>(take((int64_t)g.max_blocks * 4));
    t.e_row      = reinterpret_cast<uint32_t *>(take(g.cap_entries * 4));
    t.e_depth    = reinterpret_cast<uint32_t *>(take(g.cap_entries * 4));
    t.e_mask     = reinterpret_cast<uint16_t *>(take(g.cap_entries * 2));
    t.tile_count = reinterpret_cast<int32_t *>(take((int64_t)g.n_images * g.n_tiles * 4));
    t.items      = reinterpret_cast<int2 *>(take((int64_t)g.n_bins_total * 16 * 8));
    t.big_list   = reinterpret_cast<int32_t *>(take((int64_t)g.n_images * g.n_tiles * 4));
    if (b) *b = t;
    return (int64_t)(p - base);
}

template <int CAP>
static constexpr size_t emit_lds_bytes() { return (size_t)(CAP + CAP / 8) * 10 + (size_t)CAP * 2 + (size_t)(CAP / 64) * 32; }

} // namespace gsx

using namespace gsx;

extern "C" int gsx_isect_binned_supported(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed)
{
    if (getenv("GSX_ISECT_LEGACY")) return 0;
    if (packed && n_images != 1) return 0; // packed rows: per-image row counts live on the device
    if (n_images < 1 || tile_w < 1 || tile_h < 1 || rows < 0) return 0;
    if (rows % n_images != 0) return 0;
    BinGeom g;
    return bin_geometry(g, rows > 0 ? rows : 1, n_images, 16, tile_w, tile_h, 1) ? 1 : 0;
}

extern "C" int64_t gsx_isect_binned_count_workspace_bytes(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    BinGeom g;
    if (rows < 1) rows = 1;
    bin_geometry(g, rows, n_images, 16, tile_w, tile_h, bin_cap_entries(rows));
    return bin_layout(g, nullptr, nullptr) + 512;
}

extern "C" int64_t gsx_isect_binned_emit_workspace_bytes(int64_t n_isects)
{
    if (n_isects < 1) n_isects = 1;
    return 2 * align256(n_isects * 8) + 512; // unsorted segments of oversized tiles + their ping-pong buffer
}

static int binned_setup(const char *fn, BinArgs &a, int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w,
                        uint32_t tile_h, void *ws, int64_t ws_bytes)
{
    GSX_REQUIRE(rows >= 1 && tile_size > 0, "%s: bad rows / tile_size", fn);
    GSX_REQUIRE(rows % (n_images ? n_images : 1) == 0, "%s: rows (%lld) must be a multiple of n_images (%u)", fn,
                (long long)rows, n_images);
    GSX_REQUIRE(bin_geometry(a.g, rows, n_images, tile_size, tile_w, tile_h, bin_cap_entries(rows)),
                "%s: %u images x %u x %u tiles not supported", fn, n_images, tile_w, tile_h);
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    if (ws == nullptr || (base - reinterpret_cast<unsigned char *>(ws)) + bin_layout(a.g, base, &a.b) > ws_bytes) {
        set_last_error("%s: count workspace too small", fn);
        return GSX_ERR_WORKSPACE;
    }
    return GSX_OK;
}

extern "C" int gsx_isect_binned_count(const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                                      const float *opacities, const uint8_t *tile_mask, int64_t rows, uint32_t n_images,
                                      uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, int32_t *tiles_per_gauss,
                                      int32_t *isect_offsets, int64_t *n_isects, void *count_workspace,
                                      int64_t count_workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_bins = n_images * tile_w * tile_h;
    GSX_REQUIRE(isect_offsets && n_isects, "gsx_isect_binned_count: null output");
    if (rows == 0) {
        if (hipMemsetAsync(isect_offsets, 0, (size_t)n_bins * 4, s) != hipSuccess
            || hipMemsetAsync(n_isects, 0, 8, s) != hipSuccess) {
            set_last_error("gsx_isect_binned_count: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    GSX_REQUIRE(means2d && radii && depths, "gsx_isect_binned_count: null pointer");
    BinArgs a{};
    int rc = binned_setup("gsx_isect_binned_count", a, rows, n_images, tile_size, tile_w, tile_h, count_workspace,
                          count_workspace_bytes);
    if (rc != GSX_OK) return rc;
    a.means2d = means2d; a.radii = radii; a.depths = depths; a.conics = conics; a.opacities = opacities;
    a.tile_mask = tile_mask; a.tiles_per_gauss = tiles_per_gauss; a.isect_offsets = isect_offsets; a.n_isects = n_isects;
    const size_t bins_lds = (size_t)a.g.n_bins * sizeof(int32_t);
    bin_rect_kernel<<<dim3(a.g.n_chunks), dim3(kBnThreads), bins_lds, s>>>(a);
    rc = launch_colscan(a.b.table, a.b.bin_count, a.g.n_bins, a.g.cpi, a.g.n_images, s);
    if (rc != GSX_OK) return rc;
    bin_plan_kernel<<<dim3(1), dim3(1024), 0, s>>>(a);
    bin_scatter_kernel<<<dim3(a.g.n_chunks), dim3(kBnThreads), bins_lds, s>>>(a);
    bin_mask_kernel<<<dim3(a.g.max_blocks), dim3(kBnThreads), 0, s>>>(a);
    tile_plan_kernel<<<dim3(1), dim3(1024), 0, s>>>(a);
    return check_launch("isect_binned_count");
}

extern "C" int gsx_isect_binned_emit_sort(int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w,
                                          uint32_t tile_h, void *count_workspace, int64_t count_workspace_bytes,
                                          const int32_t *isect_offsets, int64_t n_isects, int64_t *isect_ids_sorted,
                                          int32_t *flatten_ids_sorted, void *workspace, int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(n_isects >= 0 && n_isects < (1ll << 31), "gsx_isect_binned_emit_sort: n_isects out of range");
    if (n_isects == 0 || rows == 0) return GSX_OK;
    GSX_REQUIRE(isect_offsets && isect_ids_sorted && flatten_ids_sorted, "gsx_isect_binned_emit_sort: null pointer");
    hipStream_t s = (hipStream_t)stream;
    BinArgs a{};
    int rc = binned_setup("gsx_isect_binned_emit_sort", a, rows, n_images, tile_size, tile_w, tile_h, count_workspace,
                          count_workspace_bytes);
    if (rc != GSX_OK) return rc;
    if (workspace == nullptr || workspace_bytes < gsx_isect_binned_emit_workspace_bytes(n_isects)) {
        set_last_error("gsx_isect_binned_emit_sort: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    unsigned char *p = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.bucketed = reinterpret_cast<uint2 *>(p); p += align256(n_isects * 8);
    uint2 *scratch = reinterpret_cast<uint2 *>(p);
    a.isect_offsets = const_cast<int32_t *>(isect_offsets);
    a.keys_out = reinterpret_cast<uint64_t *>(isect_ids_sorted);
    a.vals_out = flatten_ids_sorted;
    a.big_count = &a.b.hdr->big_count;
    a.big_list  = a.b.big_list;
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void *)bin_emit_kernel<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds_bytes<4096>());
        (void)hipFuncSetAttribute((const void *)bin_emit_kernel<2048>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds_bytes<2048>());
    }
    // persistent grid over the items (their number lives on the device): about one workgroup per expected item
    int64_t grid = n_isects / (a.g.item_cap / 2) + a.g.n_bins_total;
    if (grid > (int64_t)a.g.n_bins_total * 16) grid = (int64_t)a.g.n_bins_total * 16;
    if (grid < 1) grid = 1;
    if (a.g.item_cap == 2048) bin_emit_kernel<2048><<<dim3((uint32_t)grid), dim3(kBnThreads), emit_lds_bytes<2048>(), s>>>(a);
    else bin_emit_kernel<4096><<<dim3((uint32_t)grid), dim3(kBnThreads), emit_lds_bytes<4096>(), s>>>(a);
    rc = check_launch("isect_binned_emit");
    if (rc != GSX_OK) return rc;
    TileSortArgs t{};
    t.n = n_isects; t.n_tiles = a.g.n_tiles; t.tile_bits = a.g.tile_bits; t.n_bins = a.g.n_images * a.g.n_tiles;
    t.n_chunks = 1;
    t.table_scanned = const_cast<int32_t *>(isect_offsets);
    t.bucketed = a.bucketed; t.scratch = scratch;
    t.big_count = a.big_count; t.big_list = a.big_list;
    t.keys_out = a.keys_out; t.vals_out = a.vals_out;
    return launch_big_tile_sort(t, s);
}
