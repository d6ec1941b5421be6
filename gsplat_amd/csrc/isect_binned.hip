// Binned tile intersection: isect_tiles(sort=True) + isect_offset_encode with every tile list assembled, sorted and written
// from LDS by the workgroup that owns the tile - no scattered 8-byte stores into tile segments in HBM, no [chunk][tile]
// table, no intermediate pair array.
//
// Replaces, for the same outputs bit for bit, the reference's count -> cumsum -> emit -> 6-pass device radix sort -> offsets
// (gsplat/cuda/csrc/IntersectTile.cu:214-464, 1078-1121, 925-988; host Intersect.cpp:170-329) and this backend's earlier
// Gaussian-major fused path (isect_fused.hip, kept as the fallback).
//
// The screen is cut into BINS of bw x bh tiles (bw * bh <= 16: a bin's tiles are the bits of a 16-bit mask).
//
//   before the host learns n_isects
//   A  bin_rect      per row: the WALK, once (round 5): a row inside one block of 2 x 4 bins (8 x 8 tiles) leaves a 16-byte
//                    record - the block's 64 tiles as one bit each, the block's origin, the bins in use, the depth bits; a row
//                    of many slabs or over several blocks goes through an LDS queue of (row, block) units that the workgroup
//                    drains together, the units of a row over several blocks become unit records in the chunk's region. LDS
//                    histogram of the bins in use per chunk of rows -> table[chunk][bin]; tiles_per_gauss.
//   B  bin_colscan   running sum over an image's chunks per bin (in place) + bin totals; the workgroup that finishes last
//                    scans the totals into bin_start and decides whether the call goes back (capacity, crowded bins)
//   D  bin_scatter   deals the records out: entry = (depth, row) + the bin's 16-bit tile mask (a few shifts of the record's
//                    word) at an LDS cursor (bin_start + chunk prefix). No floating point. Consecutive chunks run on one XCD.
//   E  bin_tiles     one workgroup per bin: per-tile counts from the masks (wave ballots) -> tile_count
//   F  tile_plan     one workgroup: scan of the tile counts -> isect_offsets, n_isects (pinned host word)
//   after the host allocated the exact-length outputs
//   G  bin_sort      one workgroup per bin: every entry is dealt to the LDS lists of the tiles in its mask (LDS cursors:
//                    integer LDS atomics run at the rate of LDS writes on gfx950), then ONE WAVE PER TILE sorts its list
//                    with a bitonic network that needs no workgroup barrier and writes keys (image|tile|depth) and row ids
//                    contiguously. Tiles that do not fit the LDS arena together go in further batches; a tile longer than
//                    the arena goes through the work-list sort of tile_sort.hip (launch_big_tile_sort).
//
// Measured on MI355X (tools/issue_rate.hip): ONE wave issues an instruction every ~5 cycles whatever the dependencies, a SIMD
// reaches ~0.6 / 0.77 instructions per cycle only with 4 / 8 waves. These kernels are chains of dependent, divergent work, so
// they are laid out for WAVES, not for work per thread: 1024-thread workgroups, two rows per thread, two workgroups per CU.
//
// The 64-bit sort word is (depth bits << 32 | row), so equal depths come out in ascending row order = the emission order
// of the reference (stable sort on the key).
// Compiled with -ffp-contract=off (the walk must be bit-exact with the oracle; isect_walk.hpp / isect_binwalk.hpp).
#include "isect_binwalk.hpp"
#include "bitonic64.hpp"
#include "isect_fused.hpp"
#include <cstdlib>
#include <mutex>
#include <new>

namespace gsx {

constexpr int kRowThreads     = 1024;  // kernels A, D (row-major)
constexpr int kBnThreads      = 256;   // kernel E
constexpr int kBnMaxBins      = 16384; // bins in total (images x bins per image)
constexpr uint32_t kBnMaxTiles = 36864;

struct BinHeader { // device memory; the first four words are zeroed by the first workgroup of kernel A
    int32_t overflow;  // kernel B's verdict: the call goes back to the Gaussian-major path
    int32_t big_count; // tiles handed to the work-list sort (kernel G)
    int32_t b_done;    // workgroups of kernel B that have finished (the last one plans)
    int32_t pad[5];
};

struct BinGeom {
    int64_t rows, rows_per_image, cap_entries;
    uint32_t n_images, cpi, rpc, n_chunks;
    uint32_t tile_size, tile_w, tile_h, n_tiles;
    uint32_t bw, bh, bins_x, bins_y, n_bins, n_bins_total, tile_bits;
    uint32_t kx, ky; // a row record's BLOCK: kx x ky bins = (kx bw) x (ky bh) <= 64 tiles, one bit each
    int32_t skew_cap; // > 0: a bin with more entries than this sends the call back to the Gaussian-major path (bn_overflow)
    int32_t skew_ratio; // > 0: ... and so does a largest bin of more than this many times the mean bin
};

struct BinBuffers {
    BinHeader *hdr;
    int32_t *table;      // [n_chunks][n_bins]
    int32_t *bin_count;  // [n_bins_total]
    int32_t *unit_count; // [n_chunks] unit records of a chunk (may exceed kUnitCap: the call is sent back then)
    uint4 *unit_rec;     // [n_chunks][kUnitCap] one block of a row over several blocks (layout of row_rec)
    uint32_t *unit_row;  // [n_chunks][kUnitCap] its row
    int32_t *bin_start;  // [n_bins_total + 1]
    uint4 *row_rec;      // [rows] kernel A's verdict on a row: (tile mask of its block: 64 bits, origin | kind | bins, depth bits)
    uint2 *e_pair;       // [cap] (depth bits, row)
    uint16_t *e_mask;    // [cap]
    int32_t *tile_count; // [n_images * n_tiles]
    int32_t *big_list;   // [n_images * n_tiles] tiles longer than the LDS arena (kernel G -> work-list sort)
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
    int64_t *max_tile_len; // optional: longest tile list (lets the caller cut long lists into segments, raster3d_seg.hip)
    // emit
    uint64_t *keys_out;
    int32_t *vals_out;
    uint2 *bucketed;
    uint32_t dbg; // GSX_ISECT_DBG ablation bits (timing experiments only; outputs are wrong when set)
    uint32_t sort_int; // GSX_ISECT_SORT=int: every list on the integer network
};

__device__ __forceinline__ void bn_chunk_rows(const BinGeom &g, uint32_t chunk, int64_t &lo, int64_t &hi, uint32_t &img)
{
    img                = chunk / g.cpi;
    const uint32_t sub = chunk % g.cpi;
    lo                 = (int64_t)img * g.rows_per_image + (int64_t)sub * g.rpc;
    hi                 = min(lo + (int64_t)g.rpc, (int64_t)(img + 1) * g.rows_per_image);
}

struct BnRow {
    float mx, my, rx, ry, A, B, C, op;
};
__device__ __forceinline__ BnRow bn_load_row(const BinArgs &a, int64_t r, bool has_conic)
{
    BnRow q;
    const int2 rad = reinterpret_cast<const int2 *>(a.radii)[r];
    const v2f m    = reinterpret_cast<const v2f *>(a.means2d)[r];
    q.rx = (float)rad.x; q.ry = (float)rad.y; q.mx = m.x; q.my = m.y;
    q.A = q.B = q.C = q.op = 0.0f;
    if (has_conic) {
        q.A  = a.conics[3 * r];
        q.B  = a.conics[3 * r + 1];
        q.C  = a.conics[3 * r + 2];
        q.op = a.opacities[r];
    }
    return q;
}
__device__ __forceinline__ WalkPrep bn_prepare(const BnRow &q, bool has_conic, const BinGeom &g)
{
    if (!(q.rx > 0.0f && q.ry > 0.0f)) {
        WalkPrep p{};
        p.any = false;
        return p;
    }
    return walk_prepare(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h);
}

// ---- A: the walk of every row, once ----------------------------------------------------------------------------------------
// Round 5. Until round 4 this kernel took a row's rectangle of bins only, and kernel D prepared the row again and walked every
// (row, bin) pair clipped to the bin (1900 instructions per 64 rows, 78 us at c3). Now the walk happens HERE, once:
//   * a row whose tile rectangle lies inside a BLOCK of kx x ky bins (2 x 4 bins of 4 x 2 tiles = 8 x 8 tiles: 98.9 % of c3's
//     visible rows) is walked over that block and its record keeps the block's 64 tiles as one bit each - a slab's span
//     [tv0, tv1) enters the word as a run of bits (a row of the block) or a run of a strided column pattern, never tile by
//     tile, and the mask of one bin is a few shifts of the word;
//   * a wave pays for its most expensive lane (c3: 1.6 slabs per row on average, but 4.7 % of the rows have four or more), so
//     a lane walks its row itself only up to kInlineSlabs slabs. Longer rows, and every row over more than one block (as one
//     UNIT per block of its rectangle), park their prepared walk in LDS and push units on a queue that the whole workgroup
//     drains: a unit = the walk clipped to one block -> one mask word;
//   * a unit of a row over several blocks becomes a UNIT RECORD (the row record's layout + the row id) in the chunk's region of
//     kUnitCap records. A chunk that needs more sends the call back to the Gaussian-major path (kernel B's verdict).
// The bin counts are exact: a bin the walk leaves empty gets no entry (the bin rectangle of round 4 counted it). Kernel D
// deals records out and does no floating point.
constexpr uint32_t kRecNone = 0u, kRecSmall = 1u, kRecBig = 2u; // bits 16, 17 of the record's origin word; bits 18.. = bins in use
constexpr int kInlineSlabs  = 2;    // slabs a lane walks itself
constexpr int kAPark        = 512;  // rows parked per round
constexpr int kAQueue       = 2048; // units per round
constexpr int kATake        = 128;  // units one row may push per round (bounds the serial work of the pushing thread)
constexpr int kParkWords    = 17;   // rectangle (2), slot | flags, bins, depth bits, the prepared walk (12)
constexpr uint32_t kUnitCap = 1024; // unit records per chunk

// The bin shape as the row kernels see it: the default (4 x 2 tiles, blocks of 2 x 4 bins) as compile-time constants - the
// quotients and remainders by it are shifts -, any other shape (GSX_ISECT_BIN, an A/B switch) from the geometry record.
template <bool FIXED>
struct BinShape {
    uint32_t bw, bh, kx, ky;
    __device__ __forceinline__ explicit BinShape(const BinGeom &g)
        : bw(FIXED ? 4u : g.bw), bh(FIXED ? 2u : g.bh), kx(FIXED ? 2u : g.kx), ky(FIXED ? 4u : g.ky) {}
};
// tiles of bin (i, j) of the block, as the 16-bit mask the sort kernel deals from: bit y' * bw + x'
template <bool FIXED>
__device__ __forceinline__ uint32_t rec_bin_mask(uint64_t M, const BinShape<FIXED> &sh, uint32_t i, uint32_t j)
{
    return block_bin_mask(M, sh.bw, sh.bh, sh.kx, i, j);
}
// record of a block's mask; counts the bins it uses in the chunk's histogram
template <bool FIXED>
__device__ __forceinline__ uint4 make_record(uint64_t M, uint32_t bbx, uint32_t bby, uint32_t dbits, const BinGeom &g,
                                             const BinShape<FIXED> &sh, int32_t *s_hist)
{
    uint32_t used = 0;
    for (uint32_t j = 0; j < sh.ky; ++j)
        for (uint32_t i = 0; i < sh.kx; ++i)
            if (rec_bin_mask(M, sh, i, j)) {
                used |= 1u << (j * sh.kx + i);
                if (s_hist) atomicAdd(&s_hist[(bby + j) * g.bins_x + bbx + i], 1);
            }
    return make_uint4((uint32_t)M, (uint32_t)(M >> 32), bbx | (bby << 8) | (kRecSmall << 16) | (used << 18), dbits);
}

__device__ __forceinline__ void park_walk(uint32_t *rw, const WalkPrep &p, uint32_t slot, bool one, uint32_t org, uint32_t dbits)
{
    rw[0] = (uint32_t)p.x0 | ((uint32_t)p.y0 << 16); rw[1] = (uint32_t)p.x1 | ((uint32_t)p.y1 << 16);
    rw[2] = slot | (p.ellipse ? 1u << 16 : 0u) | (p.alongY ? 1u << 17 : 0u) | (one ? 1u << 18 : 0u);
    rw[3] = org;
    rw[4] = dbits;
    const float pf[12] = {p.B, p.coeff, p.disc, p.t, p.pu, p.pv, p.bmin_u, p.bmax_u, p.bmin_v, p.bmax_v, p.u_at_vmin, p.u_at_vmax};
#pragma unroll
    for (int k = 0; k < 12; ++k) rw[5 + k] = __float_as_uint(pf[k]);
}
__device__ __forceinline__ WalkPrep parked_walk(const uint32_t *rw)
{
    WalkPrep p;
    p.x0 = (int)(rw[0] & 0xFFFFu); p.y0 = (int)(rw[0] >> 16); p.x1 = (int)(rw[1] & 0xFFFFu); p.y1 = (int)(rw[1] >> 16);
    p.any = true; p.ellipse = (rw[2] >> 16) & 1u; p.alongY = (rw[2] >> 17) & 1u;
    p.B = __uint_as_float(rw[5]); p.coeff = __uint_as_float(rw[6]); p.disc = __uint_as_float(rw[7]);
    p.t = __uint_as_float(rw[8]); p.pu = __uint_as_float(rw[9]); p.pv = __uint_as_float(rw[10]);
    p.bmin_u = __uint_as_float(rw[11]); p.bmax_u = __uint_as_float(rw[12]); p.bmin_v = __uint_as_float(rw[13]);
    p.bmax_v = __uint_as_float(rw[14]); p.u_at_vmin = __uint_as_float(rw[15]); p.u_at_vmax = __uint_as_float(rw[16]);
    return p;
}

template <bool FIXED>
__global__ void __launch_bounds__(kRowThreads) bin_rect_kernel(const BinArgs a)
{
    extern __shared__ int32_t s_hist[];
    __shared__ uint32_t s_q[kAQueue];
    __shared__ uint32_t s_row[kAPark][kParkWords];
    __shared__ int32_t s_tpg[2 * kRowThreads];
    __shared__ int32_t s_qn, s_qlim, s_mn, s_units, s_anybig;
    const BinGeom &g = a.g;
    const BinShape<FIXED> sh(g);
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kRowThreads) s_hist[i] = 0;
    if (threadIdx.x == 0) {
        s_units = 0; s_mn = 0; s_qn = 0; s_qlim = kAQueue; s_anybig = 0;
        if (blockIdx.x == 0) { a.b.hdr->overflow = 0; a.b.hdr->big_count = 0; a.b.hdr->b_done = 0; }
    }
    int64_t lo, hi;
    uint32_t img;
    bn_chunk_rows(g, blockIdx.x, lo, hi, img);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)img * g.n_tiles : nullptr;
    const uint32_t W = sh.kx * sh.bw, Hh = sh.ky * sh.bh;
    uint64_t col = 0; // one bit per row of the block, in column 0
    for (uint32_t y = 0; y < Hh; ++y) col |= 1ull << (y * W);
    uint4 *unit_rec    = a.b.unit_rec + (size_t)blockIdx.x * kUnitCap;
    uint32_t *unit_row = a.b.unit_row + (size_t)blockIdx.x * kUnitCap;
    constexpr int kU = 2;
    for (int64_t base = lo; base < hi; base += kRowThreads * kU) {
        BnRow q[kU];
        uint32_t dbits[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { // every load of the thread's rows is in flight before any is used
            const int64_t r = base + u * kRowThreads + threadIdx.x;
            q[u].rx = q[u].ry = 0.0f;
            dbits[u] = 0u;
            if (r < hi) {
                q[u]     = bn_load_row(a, r, has_conic);
                dbits[u] = __float_as_uint(a.depths[r]);
            }
        }
        if (base != lo) { // (a chunk of more than 2048 rows) everyone has left the previous iteration's queue
            __syncthreads();
            if (threadIdx.x == 0) { s_mn = 0; s_qn = 0; s_qlim = kAQueue; s_anybig = 0; }
        }
        __syncthreads(); // the queue's counters, s_hist
        // a row that waits for the queue: its units, how many were pushed, its parking slot (-1: none yet)
        uint32_t units[kU], done[kU];
        int32_t ms[kU];
        bool left = false; // units that found no room in the first round
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = base + u * kRowThreads + threadIdx.x;
            units[u] = done[u] = 0u;
            ms[u]    = -1;
            if (r >= hi) continue;
            uint4 rec        = make_uint4(0u, 0u, kRecNone << 16, dbits[u]);
            const WalkPrep p = bn_prepare(q[u], has_conic, g);
            bool now         = true;
            if (p.any) {
                const uint32_t bx0 = (uint32_t)p.x0 / sh.bw, bx1 = ((uint32_t)p.x1 + sh.bw - 1) / sh.bw;
                const uint32_t by0 = (uint32_t)p.y0 / sh.bh, by1 = ((uint32_t)p.y1 + sh.bh - 1) / sh.bh;
                const uint32_t w = bx1 - bx0, h = by1 - by0;
                const bool one   = w <= sh.kx && h <= sh.ky;
                const int slabs  = p.ellipse ? min(p.x1 - p.x0, p.y1 - p.y0) : 1;
                if (one && slabs <= (int)(((a.dbg >> 16) & 15u) ? ((a.dbg >> 16) & 15u) : (uint32_t)kInlineSlabs) && !(a.dbg & 16u)) {
                    const int cx0 = (int)(bx0 * sh.bw), cy0 = (int)(by0 * sh.bh);
                    rec = make_record(block_mask(p, g.tile_size, g.tile_w, cx0, cy0, cx0 + (int)W, cy0 + (int)Hh, W, col, tmask), bx0, by0, dbits[u], g, sh, s_hist);
                } else if (!(a.dbg & 64u)) {
                    // park the prepared walk and push its units at once: in the usual case one barrier separates this from the drain
                    units[u] = one ? 1u : ((w + sh.kx - 1) / sh.kx) * ((h + sh.ky - 1) / sh.ky);
                    const int32_t m = atomicAdd(&s_mn, 1);
                    if (m < kAPark) {
                        ms[u] = m;
                        park_walk(s_row[m], p, (uint32_t)(u * kRowThreads) + threadIdx.x, one,
                                  bx0 | (by0 << 8) | ((w - 1u) << 16) | ((h - 1u) << 24), dbits[u]);
                        const int32_t take = (int32_t)min(units[u], (uint32_t)kATake);
                        const int32_t qpos = atomicAdd(&s_qn, take);
                        if (qpos + take > kAQueue) atomicMin(&s_qlim, qpos); // units are valid below the first reservation that did not fit
                        else {
                            for (int32_t k = 0; k < take; ++k) s_q[qpos + k] = (uint32_t)m | ((uint32_t)k << 9);
                            done[u] = (uint32_t)take;
                        }
                    }
                    left |= done[u] != units[u];
                    if (one) now = false; // its record is written by the lane that takes the unit
                    else {
                        rec.z = kRecBig << 16;
                        s_tpg[u * kRowThreads + threadIdx.x] = 0;
                        s_anybig = 1;
                    }
                }
            }
            if (now && !(a.dbg & 32u)) {
                a.b.row_rec[r] = rec;
                if (a.tiles_per_gauss && ((rec.z >> 16) & 3u) != kRecBig) a.tiles_per_gauss[r] = __popc(rec.x) + __popc(rec.y);
            }
        }
        // rounds of { drain the queue together } / (rare) { park and push what found no room }
        for (bool more = __syncthreads_or(left);;) {
            const int32_t n_q = (a.dbg & 256u) ? 0 : min(s_qn, s_qlim);
            for (int32_t k = (int32_t)threadIdx.x; k < n_q; k += kRowThreads) {
                const uint32_t pr  = s_q[k];
                const uint32_t *rw = s_row[pr & 511u];
                const WalkPrep p   = parked_walk(rw);
                const int slot     = (int)(rw[2] & 0xFFFFu);
                const bool one     = (rw[2] >> 18) & 1u;
                const uint32_t o = rw[3], bx0 = o & 255u, by0 = (o >> 8) & 255u, w = ((o >> 16) & 255u) + 1u;
                const uint32_t wb = (w + sh.kx - 1) / sh.kx, idx = pr >> 9;
                const uint32_t bbx = bx0 + (idx % wb) * sh.kx, bby = by0 + (idx / wb) * sh.ky;
                const int cx0 = (int)(bbx * sh.bw), cy0 = (int)(bby * sh.bh);
                const uint64_t M = (a.dbg & 512u) ? 1ull : block_mask(p, g.tile_size, g.tile_w, cx0, cy0, cx0 + (int)W, cy0 + (int)Hh, W, col, tmask);
                const int64_t r  = base + slot;
                if (one) {
                    a.b.row_rec[r] = make_record(M, bbx, bby, rw[4], g, sh, s_hist);
                    if (a.tiles_per_gauss) a.tiles_per_gauss[r] = __popcll(M);
                } else if (M) {
                    const uint32_t at = (uint32_t)atomicAdd(&s_units, 1);
                    const uint4 rec   = make_record(M, bbx, bby, rw[4], g, sh, s_hist); // counted even when the region is full:
                    if (at < kUnitCap) {                                               // the call is sent back anyway
                        unit_rec[at] = rec;
                        unit_row[at] = (uint32_t)r;
                    }
                    atomicAdd(&s_tpg[slot], __popcll(M));
                }
            }
            if (!more) break;
            // (rare: more waiting rows than parking slots, a row of more than kATake units, a full queue)
            __syncthreads(); // this round's reads of s_q / s_row
            if (threadIdx.x == 0) { s_mn = 0; s_qn = 0; s_qlim = kAQueue; }
            __syncthreads();
            left = false;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (units[u] == done[u]) continue;
                const int32_t m = atomicAdd(&s_mn, 1);
                if (m < kAPark) {
                    const int64_t r  = base + u * kRowThreads + threadIdx.x;
                    const WalkPrep p = bn_prepare(bn_load_row(a, r, has_conic), has_conic, g);
                    const uint32_t bx0 = (uint32_t)p.x0 / sh.bw, bx1 = ((uint32_t)p.x1 + sh.bw - 1) / sh.bw;
                    const uint32_t by0 = (uint32_t)p.y0 / sh.bh, by1 = ((uint32_t)p.y1 + sh.bh - 1) / sh.bh;
                    park_walk(s_row[m], p, (uint32_t)(u * kRowThreads) + threadIdx.x, units[u] == 1u,
                              bx0 | (by0 << 8) | ((bx1 - bx0 - 1u) << 16) | ((by1 - by0 - 1u) << 24), dbits[u]);
                    const int32_t take = (int32_t)min(units[u] - done[u], (uint32_t)kATake);
                    const int32_t qpos = atomicAdd(&s_qn, take);
                    if (qpos + take > kAQueue) atomicMin(&s_qlim, qpos);
                    else {
                        for (int32_t k = 0; k < take; ++k) s_q[qpos + k] = (uint32_t)m | ((done[u] + (uint32_t)k) << 9);
                        done[u] += (uint32_t)take;
                    }
                }
                left |= done[u] != units[u];
            }
            more = __syncthreads_or(left);
        }
        if (s_anybig) { // uniform: written before the first barrier of the rounds
            __syncthreads();
            if (a.tiles_per_gauss) {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t r = base + u * kRowThreads + threadIdx.x;
                    if (r < hi && units[u] > 1u) a.tiles_per_gauss[r] = s_tpg[u * kRowThreads + threadIdx.x];
                }
            }
        }
    }
    __syncthreads();
    int32_t *out = a.b.table + (int64_t)blockIdx.x * g.n_bins;
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kRowThreads) out[i] = s_hist[i];
    if (threadIdx.x == 0) a.b.unit_count[blockIdx.x] = s_units;
}

// ---- B: exclusive running sum over an image's chunks for every (image, bin), in place; totals[bin]; the plan ---------------
// The table is short and wide in the wrong direction (hundreds of chunks x a few hundred bins): 32 bins x 32 chunk segments
// per workgroup, two passes over a segment of ~cpi / 32 entries. The workgroup that finishes LAST scans the bin totals into
// bin_start (kernel D's cursors are bin_start + the chunk's prefix: two loads) and decides whether the call goes back to the
// Gaussian-major path:
//   * more entries than the workspace holds, a chunk with more unit records than its region;
//   * a crowded bin - a real scene's dense region (garden x25 has bins of > 10 k entries next to a mean of ~2 k) does not fit
//     the sort kernel's LDS arena and goes through its slow paths: emit + sort 0.80 ms where the Gaussian-major path takes 0.20;
//   * clustered but small (garden x1: 112 k rows, no bin near the arena's size): the crowded bins' workgroups are the launch's
//     tail - binned 0.218 ms against 0.128 Gaussian-major - while a uniform scene's largest bin is ~1.5 x its mean.
constexpr int kCsCols = 32, kCsSegs2 = 32, kCsPer = 32; // kCsPer >= ceil(max chunks per image / kCsSegs2)
__global__ void __launch_bounds__(kCsCols *kCsSegs2) bin_colscan_kernel(const BinArgs a, uint32_t col_groups)
{
    __shared__ int32_t s_seg[kCsSegs2][kCsCols + 1];
    __shared__ int64_t s_part[16];
    __shared__ int32_t s_max, s_last, s_umax[16];
    const BinGeom &g      = a.g;
    const uint32_t n_cols = g.n_bins, cpi = g.cpi;
    const uint32_t img = blockIdx.x / col_groups, cg = blockIdx.x % col_groups;
    const uint32_t lane_c = threadIdx.x % kCsCols, seg = threadIdx.x / kCsCols;
    const uint32_t c      = cg * kCsCols + lane_c;
    const bool live       = c < n_cols;
    const uint32_t per    = (cpi + kCsSegs2 - 1) / kCsSegs2;
    const uint32_t r0 = seg * per, r1 = min(r0 + per, cpi);
    int32_t *col = a.b.table + (int64_t)img * cpi * n_cols + c;
    // the segment lives in registers: ONE round trip to memory for its up to kCsPer loads (cpi <= 1024: bin_geometry), not one
    // per entry in two dependent loops
    int32_t v[kCsPer];
    int32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kCsPer; ++i) {
        const uint32_t r = r0 + (uint32_t)i;
        v[i]             = (live && r < r1) ? col[(int64_t)r * n_cols] : 0;
    }
#pragma unroll
    for (int i = 0; i < kCsPer; ++i) sum += v[i];
    s_seg[seg][lane_c] = sum;
    __syncthreads();
    int32_t run = 0;
    for (uint32_t k = 0; k < seg; ++k) run += s_seg[k][lane_c];
    if (live) {
#pragma unroll
        for (int i = 0; i < kCsPer; ++i) {
            const uint32_t r = r0 + (uint32_t)i;
            if (r < r1) col[(int64_t)r * n_cols] = run;
            run += v[i];
        }
        if (seg == kCsSegs2 - 1) a.b.bin_count[(int64_t)img * n_cols + c] = run;
    }
    // the last workgroup to get here plans the rest of the call (every ticket is a release at device scope, the last one acquires)
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&a.b.hdr->b_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int64_t n = block_scan_i32_1024(a.b.bin_count, a.b.bin_start, g.n_bins_total, s_part, &s_max);
    int32_t um = 0;
    for (uint32_t i = threadIdx.x; i < g.n_chunks; i += kCsCols * kCsSegs2) um = max(um, a.b.unit_count[i]);
    um = wave_max_i32(um);
    if ((threadIdx.x & 63u) == 0) s_umax[threadIdx.x >> 6] = um;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) um = max(um, s_umax[w]);
        const int64_t mx    = s_max;
        const bool lopsided = g.skew_ratio > 0 && mx * (int64_t)g.n_bins_total > (int64_t)g.skew_ratio * n && mx > 256;
        a.b.bin_start[g.n_bins_total] = (int32_t)min(n, (int64_t)INT32_MAX);
        a.b.hdr->overflow = (n > g.cap_entries || um > (int32_t)kUnitCap || (g.skew_cap > 0 && mx > g.skew_cap) || lopsided) ? 1 : 0;
    }
}

// ---- D: entries (depth, row | mask) grouped by bin ----------------------------------------------------------------------
// Row records and the chunk's unit records are dealt out: one entry per bin in use, at the bin's LDS cursor = bin_start + the
// chunk's prefix inside the bin. Workgroups go to the eight XCDs round robin; consecutive CHUNKS write neighbouring slots of
// every bin (their prefixes follow each other), so a run of consecutive chunks is given to one XCD: a line of the entry
// arrays fills up in one L2 (c3: the scattered stores cost 24 us with chunk = workgroup index, 13 us with this map).
template <bool FIXED>
__device__ __forceinline__ void bn_deal(const BinArgs &a, const BinShape<FIXED> &sh, const uint4 &rec, uint32_t row, int32_t *s_cur)
{
    const uint32_t bx0 = rec.z & 255u, by0 = (rec.z >> 8) & 255u;
    const uint64_t M   = ((uint64_t)rec.y << 32) | rec.x;
    for (uint32_t used = rec.z >> 18; used;) {
        const uint32_t k = (uint32_t)__builtin_ctz(used);
        used &= used - 1u;
        const uint32_t i = k % sh.kx, j = k / sh.kx;
        const int32_t slot = atomicAdd(&s_cur[(by0 + j) * a.g.bins_x + bx0 + i], 1);
        if (a.dbg & 8u) continue;
        a.b.e_pair[slot] = make_uint2(rec.w, row);
        a.b.e_mask[slot] = (uint16_t)rec_bin_mask(M, sh, i, j);
    }
}

template <bool FIXED>
__global__ void __launch_bounds__(kRowThreads) bin_scatter_kernel(const BinArgs a)
{
    extern __shared__ int32_t s_cur[];
    const BinGeom &g = a.g;
    const BinShape<FIXED> sh(g);
    const uint32_t per_xcd = gridDim.x / 8u;
    const uint32_t chunk   = (a.dbg & 4096u) ? blockIdx.x : (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
    if (chunk >= g.n_chunks) return;
    int64_t lo, hi;
    uint32_t img;
    bn_chunk_rows(g, chunk, lo, hi, img);
    constexpr int kU = 2;
    uint4 rec[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { // the first records travel while the cursors are made
        const int64_t r = lo + u * kRowThreads + threadIdx.x;
        rec[u]          = r < hi ? a.b.row_rec[r] : make_uint4(0u, 0u, 0u, 0u);
    }
    const int32_t *start = a.b.bin_start + (int64_t)img * g.n_bins;
    const int32_t *pre   = a.b.table + (int64_t)chunk * g.n_bins; // exclusive prefix over this image's chunks
    for (uint32_t i = threadIdx.x; i < g.n_bins; i += kRowThreads) s_cur[i] = start[i] + pre[i];
    const int32_t n_units = min(a.b.unit_count[chunk], (int32_t)kUnitCap);
    if (a.b.hdr->overflow) return; // the same answer in every workgroup
    __syncthreads();
    for (int64_t base = lo; base < hi; base += kU * kRowThreads) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = base + u * kRowThreads + threadIdx.x;
            if (base != lo) rec[u] = r < hi ? a.b.row_rec[r] : make_uint4(0u, 0u, 0u, 0u);
            if (((rec[u].z >> 16) & 3u) == kRecSmall) bn_deal(a, sh, rec[u], (uint32_t)r, s_cur);
        }
    }
    for (int32_t k = (int32_t)threadIdx.x; k < n_units; k += kRowThreads)
        bn_deal(a, sh, a.b.unit_rec[(size_t)chunk * kUnitCap + k], a.b.unit_row[(size_t)chunk * kUnitCap + k], s_cur);
}

// ---- E: per-tile counts of one bin --------------------------------------------------------------------------------------
// (Round 5 tried the scan of kernel F in the last workgroup of this one: a release at device scope per workgroup - 1020 L2
// write-backs - costs more than the launch it saves: 27 us against 6.4 + 5.7.)
__global__ void __launch_bounds__(kBnThreads) bin_tiles_kernel(const BinArgs a)
{
    __shared__ int32_t s_cnt[16];
    const BinGeom &g = a.g;
    if (a.b.hdr->overflow) return;
    const uint32_t bin = blockIdx.x;
    const int32_t e0 = a.b.bin_start[bin], e1 = a.b.bin_start[bin + 1];
    const int lane = (int)(threadIdx.x & 63u);
    const int n_bits = (int)(g.bw * g.bh);
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int32_t mine = 0; // lane t < n_bits: entries seen by this wave that touch tile t
    constexpr int kU = 8;
    for (int32_t base = e0; base < e1; base += kBnThreads * kU) {
        uint32_t m[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int32_t e = base + u * kBnThreads + (int32_t)threadIdx.x;
            m[u]            = e < e1 ? (uint32_t)a.b.e_mask[e] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (base + u * kBnThreads >= e1) break; // wave-uniform
            for (int t = 0; t < n_bits; ++t) {
                const uint64_t b = __builtin_amdgcn_ballot_w64((m[u] >> t) & 1u);
                if (lane == t) mine += (int32_t)__popcll(b);
            }
        }
    }
    if (lane < n_bits && mine) atomicAdd(&s_cnt[lane], mine);
    __syncthreads();
    if ((int)threadIdx.x < n_bits) {
        const uint32_t img = bin / g.n_bins, lb = bin % g.n_bins;
        const uint32_t tx = (lb % g.bins_x) * g.bw + threadIdx.x % g.bw, ty = (lb / g.bins_x) * g.bh + threadIdx.x / g.bw;
        if (tx < g.tile_w && ty < g.tile_h) a.b.tile_count[(size_t)img * g.n_tiles + (size_t)ty * g.tile_w + tx] = s_cnt[threadIdx.x];
    }
}

// ---- F: offsets, n_isects -----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_plan_kernel(const BinArgs a)
{
    __shared__ int64_t s_part[16];
    __shared__ int32_t s_max;
    const BinGeom &g = a.g;
    if (a.b.hdr->overflow) {
        if (threadIdx.x == 0) {
            __threadfence_system();
            *a.n_isects = GSX_ISECT_RETRY; // the caller reruns the Gaussian-major path (gsx_isect_binned_count's contract)
        }
        return;
    }
    const uint32_t nt   = g.n_images * g.n_tiles;
    const int64_t total = block_scan_i32_1024(a.b.tile_count, a.isect_offsets, nt, s_part, a.max_tile_len ? &s_max : nullptr);
    if (threadIdx.x == 0) {
        if (a.max_tile_len) *a.max_tile_len = (int64_t)s_max; // written BEFORE n_isects: the host reads it once that arrived
        __threadfence_system();
        *a.n_isects = total;
    }
}

// ---- G: per bin: deal the entries to the tiles' LDS lists, one wave sorts each list, write ---------------------------------
// The network is bitonic64.hpp's: on the keys as doubles (v_min_f64 / v_max_f64) unless a list holds a depth that does not
// order like a positive normal double (the deal loop flags it) - that list takes the integer network.
constexpr int kArenaPerWave = 512; // arena sort words per tile-wave: 16 waves -> 8192 words (+ 1/8 padding) = 72 KiB

__global__ void __launch_bounds__(1024) bin_sort_kernel(const BinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const BinGeom &g = a.g;
    const int n_bits = (int)(g.bw * g.bh), n_thr = n_bits * 64;
    const int arena_words = kArenaPerWave * n_bits;
    uint64_t *s_arena = reinterpret_cast<uint64_t *>(smem_raw); // [arena_words + arena_words / 8]
    __shared__ int32_t s_tcnt[16], s_goff[16], s_base[16], s_batch[16], s_cur[16], s_nbatch, s_n;
    __shared__ uint32_t s_odd; // bit t: tile t's list holds a key the f64 network cannot order (bt_key_is_odd)
    __shared__ uint64_t s_hi[16];
    const uint32_t bin = blockIdx.x;
    const int32_t e0 = a.b.bin_start[bin], e1 = a.b.bin_start[bin + 1];
    if (e0 == e1) return;
    const uint32_t img = bin / g.n_bins, lb = bin % g.n_bins;
    const uint32_t tx0 = (lb % g.bins_x) * g.bw, ty0 = (lb / g.bins_x) * g.bh;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if ((int)threadIdx.x < 16) {
        const int t       = (int)threadIdx.x;
        const uint32_t tx = tx0 + (uint32_t)t % g.bw, ty = ty0 + (uint32_t)t / g.bw;
        const bool in     = t < n_bits && tx < g.tile_w && ty < g.tile_h;
        const size_t gid  = (size_t)img * g.n_tiles + (size_t)ty * g.tile_w + tx;
        const int n       = in ? a.b.tile_count[gid] : 0;
        s_tcnt[t]         = n;
        s_goff[t]         = in ? a.isect_offsets[gid] : 0;
        s_hi[t]           = in ? ((((uint64_t)img << g.tile_bits) | ((uint64_t)ty * g.tile_w + tx)) << 32) : 0ull;
        // cut the arena: tile t of batch s_batch[t] sorts at physical word s_base[t] (16 lanes of the first wave)
        int P = 0;
        if (n > 0) {
            P = 64;
            while (P < n) P <<= 1;
        }
        const bool big = P > arena_words;
        if (big) P = 0;
        int batch = 0, used = 0, my_batch = n > 0 ? 0 : -1, my_base = 0;
        for (int k = 0; k < 16; ++k) { // every lane replays the greedy cut over the 16 sizes
            const int Pk = __shfl(P, k, 16);
            if (Pk == 0) continue;
            if (used + Pk > arena_words) { ++batch; used = 0; }
            if (k == t) { my_batch = batch; my_base = used + (used >> 3); }
            used += Pk;
        }
        s_batch[t] = big ? -2 : my_batch;
        s_base[t]  = my_base;
        if (t == 15) s_nbatch = batch + 1;
    }
    __syncthreads();
    const int n_batch = s_nbatch;
    for (int b = 0; b < n_batch; ++b) {
        if (threadIdx.x < 16) s_cur[threadIdx.x] = 0;
        if (threadIdx.x == 16) s_odd = a.sort_int ? 0xFFFFu : 0u;
        uint32_t bmask = 0;
        for (int t = 0; t < n_bits; ++t) bmask |= (s_batch[t] == b) ? (1u << t) : 0u;
        __syncthreads();
        // deal every entry to the lists of the tiles in its mask (LDS cursor per tile). (Round 5: requesting several trips'
        // entries together, or a group ahead, changes nothing - 70.6 .. 71.9 us for 1 / 2 / 4 / 8 trips -: with eight waves per
        // SIMD the other waves cover a wave's round trips; the group held ahead costs registers and a workgroup per CU, 77 us.)
        for (int32_t e = e0 + (int32_t)threadIdx.x; e < e1 && !(a.dbg & 2u); e += n_thr) {
            uint32_t m = (uint32_t)a.b.e_mask[e] & bmask;
            if (m == 0u) continue;
            const uint2 pr     = a.b.e_pair[e];
            const uint64_t key = ((uint64_t)pr.x << 32) | pr.y;
            if (bt_key_is_odd(pr.x)) atomicOr(&s_odd, m);
            while (m) {
                const int t = __builtin_ctz(m);
                m &= m - 1u;
                const int32_t pos = atomicAdd(&s_cur[t], 1);
                s_arena[s_base[t] + bt_phys(pos)] = key;
            }
        }
        __syncthreads();
        if (wave < n_bits && s_batch[wave] == b) { // one wave per tile: pad, sort, write
            uint64_t *sw = s_arena + s_base[wave];
            const int n  = s_tcnt[wave];
            int lp = 6;
            while ((1 << lp) < n) ++lp;
            const bool as_int = (s_odd >> wave) & 1u; // wave-uniform
            const uint64_t pad = as_int ? kBtPadInt : kBtPadF64;
            for (int i = n + lane; i < (1 << lp); i += 64) sw[bt_phys(i)] = pad;
            wave_lds_sync();
            if (a.dbg & 1u) {
            } else if (as_int) bt_sort_int<64>(sw, lp, lane, [] { wave_lds_sync(); });
            else bt_sort_f64<64>(sw, lp, lane, [] { wave_lds_sync(); });
            const int64_t off = s_goff[wave];
            const uint64_t hi = s_hi[wave];
            for (int i = lane; i < n; i += 64) {
                const uint64_t w    = sw[bt_phys(i)];
                a.keys_out[off + i] = hi | (w >> 32);
                a.vals_out[off + i] = (int32_t)(uint32_t)w;
            }
        }
        __syncthreads(); // the arena is cut anew for the next batch
    }
    // tiles longer than the arena: their unsorted segment goes through the work-list sort
    for (int t = 0; t < n_bits; ++t) {
        if (s_batch[t] != -2) continue;
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int64_t off = s_goff[t];
        for (int32_t base = e0; base < e1; base += n_thr) {
            const int32_t e    = base + (int32_t)threadIdx.x;
            const bool has     = e < e1 && (((uint32_t)a.b.e_mask[e] >> t) & 1u);
            const uint64_t bal = __builtin_amdgcn_ballot_w64(has);
            if (bal == 0ull) continue;
            const int first = (int)__builtin_ctzll(bal);
            int32_t wbase   = 0;
            if (lane == first) wbase = atomicAdd(&s_n, (int32_t)__popcll(bal));
            wbase = __builtin_amdgcn_readlane(wbase, first);
            if (has) a.bucketed[off + wbase + (int32_t)__popcll(bal & lt_mask)] = a.b.e_pair[e];
        }
        if (threadIdx.x == 0) {
            const uint32_t tx = tx0 + (uint32_t)t % g.bw, ty = ty0 + (uint32_t)t / g.bw;
            a.b.big_list[atomicAdd(&a.b.hdr->big_count, 1)] = (int32_t)(img * g.n_tiles + ty * g.tile_w + tx);
        }
        __syncthreads();
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

static void bin_dims(uint32_t &bw, uint32_t &bh)
{
    bw = 4; bh = 2; // measured on c3 (r3f / r3g): 4x4 0.233, 4x2 0.206, 2x4 0.209, 2x2 0.225, 8x2 0.235 ms; round 5 (the other
                    // shapes on the row kernels' generic instantiation): 0.205, 0.171, 0.198, 0.189, 0.203
    if (const char *e = getenv("GSX_ISECT_BIN")) { // "WxH" (tiles), W * H <= 16: A/B switch
        unsigned w = 0, h = 0;
        if (sscanf(e, "%ux%u", &w, &h) == 2 && w >= 1 && h >= 1 && w * h <= 16) { bw = w; bh = h; }
    }
}

static bool bin_geometry(BinGeom &g, int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w, uint32_t tile_h,
                         int64_t cap_entries)
{
    g = BinGeom{};
    g.rows = rows; g.n_images = n_images ? n_images : 1;
    g.rows_per_image = rows / g.n_images;
    bin_dims(g.bw, g.bh);
    g.tile_size = tile_size; g.tile_w = tile_w; g.tile_h = tile_h; g.n_tiles = tile_w * tile_h;
    g.bins_x = (tile_w + g.bw - 1) / g.bw; g.bins_y = (tile_h + g.bh - 1) / g.bh;
    g.kx = 2; g.ky = g.bw * g.bh <= 8 ? 4 : 2; // 4 x 2 tiles per bin: a block of 8 x 8 tiles
    g.n_bins = g.bins_x * g.bins_y; g.n_bins_total = g.n_bins * g.n_images;
    g.tile_bits = bits_for(g.n_tiles);
    // chunks of >= 2048 rows, image-aligned, at most ~1024 in total (the [chunk][bin] table stays small)
    const int64_t max_cpi = 1024 / g.n_images > 0 ? 1024 / g.n_images : 1;
    int64_t cpi = (g.rows_per_image + 2047) / 2048;
    // fewer rows than 256 chunks of 2048: halve the chunks rather than leave CUs without a workgroup (250 k packed rows: 122
    // workgroups of the 256 the chip runs at once)
    {
        const int64_t per_image = (256 + (int64_t)g.n_images - 1) / (int64_t)g.n_images;
        const int64_t floor_cpi = g.rows_per_image / 1024 < per_image ? g.rows_per_image / 1024 : per_image;
        if (cpi < floor_cpi) cpi = floor_cpi;
    }
    if (cpi < 1) cpi = 1;
    if (cpi > max_cpi) cpi = max_cpi;
    g.cpi = (uint32_t)cpi;
    g.rpc = (uint32_t)((g.rows_per_image + cpi - 1) / cpi);
    if (g.rpc == 0) g.rpc = 1;
    g.n_chunks    = g.cpi * g.n_images;
    g.cap_entries = cap_entries;
    return g.n_bins_total <= (uint32_t)kBnMaxBins && (uint64_t)g.n_images * g.n_tiles <= kBnMaxTiles && rows < (1ll << 28)
           && g.bins_x <= 256 && g.bins_y <= 256; // (parked row, bin x, bin y) pairs of kernel D: 10 + 8 + 8 bits
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
    t.table      = reinterpret_cast<int32_t *>(take((int64_t)g.n_chunks * g.n_bins * 4));
    t.bin_count  = reinterpret_cast<int32_t *>(take((int64_t)g.n_bins_total * 4));
    t.unit_count = reinterpret_cast<int32_t *>(take((int64_t)g.n_chunks * 4));
    t.unit_rec   = reinterpret_cast<uint4 *>(take((int64_t)g.n_chunks * kUnitCap * 16));
    t.unit_row   = reinterpret_cast<uint32_t *>(take((int64_t)g.n_chunks * kUnitCap * 4));
    t.bin_start  = reinterpret_cast<int32_t *>(take(((int64_t)g.n_bins_total + 1) * 4));
    t.row_rec    = reinterpret_cast<uint4 *>(take(g.rows * 16));
    t.e_pair     = reinterpret_cast<uint2 *>(take(g.cap_entries * 8));
    t.e_mask     = reinterpret_cast<uint16_t *>(take(g.cap_entries * 2));
    t.tile_count = reinterpret_cast<int32_t *>(take((int64_t)g.n_images * g.n_tiles * 4));
    t.big_list   = reinterpret_cast<int32_t *>(take((int64_t)g.n_images * g.n_tiles * 4));
    if (b) *b = t;
    return (int64_t)(p - base);
}

} // namespace gsx

using namespace gsx;

// ---- inputs that sent the binned path back recently ---------------------------------------------------------------------------
// A clustered scene fails the skew test 25 us into the count half and the call starts over Gaussian-major (garden x25, packed:
// forward 1258 fps where the Gaussian-major path alone gives 1340). A trainer renders the same scene every step: after a retry
// the next 63 calls of that shape (row count to 64 k, images, tile grid) skip the attempt, the 64th probes again.
//
// WHOSE memory: never the process'. The notes live in a PathMemory object. A caller that wants its own (a trainer, a renderer
// serving one scene) creates one with gsx_isect_path_memory_create() and makes it current for its calls with
// gsx_isect_path_memory_use(); every thread that never did owns a private default. Which kernel runs therefore depends on the
// history of THAT caller only - two scenes of one shape rendered alternately by two owners do not flip each other's path.
namespace {
struct RetryNote { uint64_t key; int left; };
struct PathMemory {
    std::mutex mu;
    RetryNote notes[16] = {};
    uint32_t next = 0;
};
thread_local PathMemory t_default_memory;
thread_local PathMemory *t_current_memory = nullptr; // null: the thread's own default
PathMemory &current_memory() { return t_current_memory ? *t_current_memory : t_default_memory; }
uint64_t retry_key(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return ((uint64_t)(rows >> 16) << 40) ^ ((uint64_t)n_images << 32) ^ ((uint64_t)tile_w << 16) ^ (uint64_t)tile_h ^ (1ull << 63);
}
// consume = true: this call IS one of the 63 that skip the attempt (gsx_isect_binned_should_try, once per intersection)
bool retried_recently(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, bool consume)
{
    const uint64_t key = retry_key(rows, n_images, tile_w, tile_h);
    PathMemory &m = current_memory();
    std::lock_guard<std::mutex> lock(m.mu);
    for (auto &n : m.notes)
        if (n.key == key && n.left > 0) {
            if (consume) --n.left;
            return true;
        }
    return false;
}
} // namespace

extern "C" void *gsx_isect_path_memory_create(void) { return new (std::nothrow) PathMemory(); }
extern "C" void gsx_isect_path_memory_destroy(void *mem)
{
    PathMemory *m = static_cast<PathMemory *>(mem);
    if (m && t_current_memory == m) t_current_memory = nullptr;
    delete m;
}
extern "C" void *gsx_isect_path_memory_use(void *mem)
{
    void *prev = t_current_memory;
    t_current_memory = static_cast<PathMemory *>(mem);
    return prev;
}

extern "C" int gsx_isect_binned_note_retry(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    const uint64_t key = retry_key(rows, n_images, tile_w, tile_h);
    PathMemory &m = current_memory();
    std::lock_guard<std::mutex> lock(m.mu);
    for (auto &n : m.notes)
        if (n.key == key) {
            n.left = 63;
            return GSX_OK;
        }
    m.notes[m.next++ % 16u] = RetryNote{key, 63};
    return GSX_OK;
}

// A pure function of its arguments, the environment switches and the retry notes: querying it changes nothing (a binding
// may size a workspace with it, a test may probe it). The decision that COUNTS one of the 63 skipped calls is
// gsx_isect_binned_should_try, made once per intersection.
static int binned_decide(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed, bool consume);
extern "C" int gsx_isect_binned_supported(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed)
{
    return binned_decide(rows, n_images, tile_w, tile_h, packed, false);
}
extern "C" int gsx_isect_binned_should_try(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed)
{
    return binned_decide(rows, n_images, tile_w, tile_h, packed, true);
}
static int binned_decide(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed, bool consume)
{
    // GSX_ISECT_PATH = binned | legacy forces the choice (tests, A/B); GSX_ISECT_LEGACY is the older spelling of "legacy"
    const char *force = getenv("GSX_ISECT_PATH");
    if (getenv("GSX_ISECT_LEGACY") || (force && force[0] == 'l')) return 0;
    if (packed && n_images != 1) return 0; // packed rows: per-image row counts live on the device
    if (n_images < 1 || tile_w < 1 || tile_h < 1 || rows < 0) return 0;
    if (rows % n_images != 0) return 0;
    BinGeom g;
    if (!bin_geometry(g, rows > 0 ? rows : 1, n_images, 16, tile_w, tile_h, 1)) return 0;
    if (force && force[0] == 'b') return 1;
    // measured (MI355X, profiles/r07_ab.md): 1 M rows x 8160 tiles (c3) 0.214 vs 0.249 ms; 4 M rows per image (c4:
    // ~1800 entries per tile, one wave sorting 2048 words) 4.3 vs 2.5 ms -> dense images and small calls stay Gaussian-major
    // What decides is the typical tile list: one wave sorts a list padded to a power of two, and the eight lists of a bin share
    // a 4096-word arena - lists of up to 512 entries (c3: 122 rows per tile, 466 per list) fit in one go. Uniform scene,
    // fused vs binned in ms (tools/gpu_isect_sizes.py): 250 k rows 0.141 / 0.103, 600 k 0.189 / 0.156, 1 M 0.245 / 0.209,
    // 2 M (245 rows per tile, lists of ~930) 0.399 / 0.470; four cameras: 4 x 250 k 0.265 / 0.222, 4 x 1 M 0.757 / 0.584.
    // Clustered scenes are caught by the skew test (bn_overflow); very small calls (garden x1, 112 k clustered rows: 0.17 / 0.29)
    // stay Gaussian-major - per IMAGE: four cameras over the same 112 k rows (1020 nearly empty bins each) 0.28 / 0.41.
    // Round 4 (f64 sort network, a workgroup per CU on small inputs: profiles/r08_ab.md #29): the uniform scene is faster binned
    // from 60 k rows on (0.068 / 0.114 ms; 120 k 0.071 / 0.117; the 100 k near rows a far plane leaves of c3 0.107 / 0.167), the
    // clustered garden stays faster Gaussian-major at every size (x1 0.207 / 0.128 forced) - it fails the skew test, and an
    // input that did is not tried again for a while (retried_recently).
    if (!(g.rows_per_image >= 49152 && g.rows_per_image <= 144ll * (int64_t)g.n_tiles)) return 0;
    return retried_recently(rows, n_images, tile_w, tile_h, consume) ? 0 : 1;
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
    if (const char *e = getenv("GSX_ISECT_DBG")) a.dbg = (uint32_t)atoi(e);
    a.sort_int = bitonic_f64_enabled() ? 0u : 1u;
    // skew abort: 3/4 of the sort arena's words (a c3 bin of 4x2 tiles holds ~1400 entries, its longest ~1700); not when the
    // path was forced (tests drive the big-list code with it); GSX_ISECT_BIN_SKEW overrides (0 = never)
    {
        const char *force = getenv("GSX_ISECT_PATH"), *sk = getenv("GSX_ISECT_BIN_SKEW");
        a.g.skew_cap = (force && force[0] == 'b') ? 0 : (int32_t)(kArenaPerWave * (int)(a.g.bw * a.g.bh) * 3 / 4);
        a.g.skew_ratio = (force && force[0] == 'b') ? 0 : 4;
        if (sk) { a.g.skew_cap = atoi(sk); a.g.skew_ratio = a.g.skew_cap > 0 ? 4 : 0; }
    }
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
                                      int32_t *isect_offsets, int64_t *n_isects, int64_t *max_tile_len, void *count_workspace,
                                      int64_t count_workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_bins = n_images * tile_w * tile_h;
    GSX_REQUIRE(isect_offsets && n_isects, "gsx_isect_binned_count: null output");
    if (rows == 0) {
        if (max_tile_len && hipMemsetAsync(max_tile_len, 0, 8, s) != hipSuccess) return check_launch("isect_binned_count memset");
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
    a.max_tile_len = max_tile_len;
    const size_t bins_lds = (size_t)a.g.n_bins * sizeof(int32_t);
    const bool fixed      = a.g.bw == 4 && a.g.bh == 2 && a.g.kx == 2 && a.g.ky == 4;
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void *)bin_rect_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void *)bin_rect_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    }
    if (fixed) bin_rect_kernel<true><<<dim3(a.g.n_chunks), dim3(kRowThreads), bins_lds, s>>>(a);
    else bin_rect_kernel<false><<<dim3(a.g.n_chunks), dim3(kRowThreads), bins_lds, s>>>(a);
    const uint32_t col_groups = (a.g.n_bins + kCsCols - 1) / kCsCols;
    bin_colscan_kernel<<<dim3(col_groups * a.g.n_images), dim3(kCsCols * kCsSegs2), 0, s>>>(a, col_groups);
    const uint32_t d_grid = (a.g.n_chunks + 7u) / 8u * 8u; // a whole number of workgroups per XCD (chunk map of kernel D)
    if (fixed) bin_scatter_kernel<true><<<dim3(d_grid), dim3(kRowThreads), bins_lds, s>>>(a);
    else bin_scatter_kernel<false><<<dim3(d_grid), dim3(kRowThreads), bins_lds, s>>>(a);
    bin_tiles_kernel<<<dim3(a.g.n_bins_total), dim3(kBnThreads), 0, s>>>(a);
    tile_plan_kernel<<<dim3(1), dim3(1024), 0, s>>>(a);
    return check_launch("isect_binned_count");
}

extern "C" int gsx_isect_binned_emit_sort(int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w,
                                          uint32_t tile_h, void *count_workspace, int64_t count_workspace_bytes,
                                          const int32_t *isect_offsets, int64_t n_isects, int64_t longest_list,
                                          int64_t *isect_ids_sorted, int32_t *flatten_ids_sorted, void *workspace,
                                          int64_t workspace_bytes, void *stream)
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
    const int n_bits = (int)(a.g.bw * a.g.bh);
    const size_t sort_lds = (size_t)(kArenaPerWave * n_bits) * 9;
    static PerDeviceOnce once;
    if (once.first())
        (void)hipFuncSetAttribute((const void *)bin_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    bin_sort_kernel<<<dim3(a.g.n_bins_total), dim3(64 * n_bits), sort_lds, s>>>(a);
    rc = check_launch("isect_binned_emit");
    if (rc != GSX_OK) return rc;
    TileSortArgs t{};
    t.n = n_isects; t.n_tiles = a.g.n_tiles; t.tile_bits = a.g.tile_bits; t.n_bins = a.g.n_images * a.g.n_tiles;
    t.n_chunks = 1;
    t.table_scanned = const_cast<int32_t *>(isect_offsets);
    t.bucketed = a.bucketed; t.scratch = scratch;
    t.big_count = &a.b.hdr->big_count; t.big_list = a.b.big_list;
    t.keys_out = a.keys_out; t.vals_out = a.vals_out;
    // every list sorts inside bin_sort's arena when its power-of-two padding fits: a caller that knows the longest list (the
    // count half reported it) spares the launch of the (then empty) work-list sort - 4 us of a 147 KiB-LDS grid spinning up
    if (longest_list > 0 && longest_list <= (int64_t)(kArenaPerWave * n_bits) / 2) return GSX_OK;
    return launch_big_tile_sort(t, s);
}
