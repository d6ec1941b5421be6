// Fused tile intersection: the per-(chunk, tile) histogram is taken WHILE counting, and the emission pass writes every
// (depth, row) pair straight into its tile's segment — the unsorted (key, value) arrays of gsx_isect_emit, the
// histogram pass and the scatter pass of gsx_isect_tile_sort are never materialised.
//
//   count+hist   gsx_isect_fused_count      walk the tiles of every row (isect_walk.hpp) -> tiles_per_gauss[row] and an
//                                           LDS histogram of tile ids per CHUNK of rows -> table[chunk][tile]
//   column scan  (tile_sort.hip)            per (image, tile): running sum over that image's chunks (exclusive, in place)
//                                           + tile totals -> exclusive scan -> isect_offsets (the reference's
//                                           intersect_offset output, for free) and n_isects
//                                           very large inputs (fused_records_spans): + the row's SPANS (one 16-byte record:
//                                           first slab, up to six (start, length) pairs) - what the walk found, so that the
//                                           emission does not repeat its divisions and square roots. c4 (16 M rows): count
//                                           0.61 -> 0.77 ms, emit + sort 1.79 -> 1.50; garden x25 (2.8 M rows): 0.108 -> 0.131,
//                                           0.198 -> 0.198 - the emission of a small input is bound by its scattered stores
//   emit+scatter gsx_isect_fused_emit_sort  the same walk, or the recorded spans (rows that did not fit a record walk again);
//                                           slot = LDS cursor[tile]++ (cursor = offset + chunk prefix);
//                                           bucketed[slot] = (depth bits, row)
//   tile sort    (tile_sort.hip)            per-tile LDS sort by (depth, row), unchanged
//
// Result: identical to gsx_isect_count/emit + gsx_isect_tile_sort (= a stable sort on the full key, ties in ascending
// row order), because the per-tile sort does not depend on the arrival order inside a segment. HBM traffic per
// intersection drops from 12 (emit) + 8 (hist) + 12 + 8 (scatter) to 8 bytes before the per-tile sort.
//
// Both walk kernels visit the rows of a chunk in a load-balanced order (for_rows_balanced below).
// Chunks are image-aligned (dense rows: N per image; packed rows are supported for a single image), so a chunk's
// histogram only needs the tiles of ONE image: the table is [n_chunks][tiles per image].
// Compiled with -ffp-contract=off (the walk must be bit-exact with the oracle; see isect_walk.hpp).
#include "isect_spans.hpp"
#include "isect_fused.hpp"

namespace gsx {


__device__ __forceinline__ void chunk_rows(const FusedGeom &g, uint32_t chunk, int64_t &lo, int64_t &hi)
{
    const uint32_t img = chunk / g.cpi, sub = chunk % g.cpi;
    lo = (int64_t)img * g.rows_per_image + (int64_t)sub * g.rpc;
    hi = min(lo + (int64_t)g.rpc, (int64_t)(img + 1) * g.rows_per_image);
}

struct RowGeom {
    bool live;
    float mx, my, rx, ry, A, B, C, op;
};
__device__ __forceinline__ RowGeom load_row_geom(const FusedArgs &a, int64_t r, bool has_conic)
{
    // every field is read unconditionally (dead rows included): no read depends on another one's result
    RowGeom q;
    q.rx = (float)a.radii[2 * r];
    q.ry = (float)a.radii[2 * r + 1];
    q.mx = a.means2d[2 * r];
    q.my = a.means2d[2 * r + 1];
    q.A = q.B = q.C = q.op = 0.0f;
    if (has_conic) {
        q.A  = a.conics[3 * r];
        q.B  = a.conics[3 * r + 1];
        q.C  = a.conics[3 * r + 2];
        q.op = a.opacities[r];
    }
    q.live = q.rx > 0.0f && q.ry > 0.0f;
    return q;
}

// ---- load balancing of the walk ------------------------------------------------------------------------------------
// The walk of one row costs (slabs + tiles) of that row, and a wave pays for its LARGEST row: with rows in storage order
// (near, large Gaussians scattered among thousands of 1-4 tile ones) most lanes idle most of the time. So every
// kFusedSub rows of a chunk are first ordered by a size class — the tile area of the radius box (counting pass) or the
// recorded tile count (emission), 0 = largest — with an
// LDS counting sort, and thread t walks the t-th row of that order: waves get rows of similar cost, the big ones first.
// Which thread walks a row changes nothing in the outputs (counts per row; slots inside a tile segment are sorted later).
constexpr int kFusedPer = 4; // rows per thread and round: a sub-chunk is kFusedPer x THREADS rows
// dynamic LDS = one int32 per tile of an image (<= kMaxBins = 36864 of tile_sort.hip) next to ~8.3 KiB of static LDS below
constexpr int kFusedMaxDynLds = 36864 * 4;

__device__ __forceinline__ int size_class(const FusedArgs &a, int64_t r)
{
    const FusedGeom &g = a.geom;
    const float rx = (float)a.radii[2 * r], ry = (float)a.radii[2 * r + 1];
    if (!(rx > 0.0f && ry > 0.0f)) return 63; // dead rows last
    const float mx = a.means2d[2 * r], my = a.means2d[2 * r + 1], ts = (float)g.tile_size;
    const int x0 = clampi(f2i_trunc_sat((mx - rx) / ts), 0, (int)g.tile_w), x1 = clampi(f2i_trunc_sat((mx + rx) / ts) + 1, 0, (int)g.tile_w);
    const int y0 = clampi(f2i_trunc_sat((my - ry) / ts), 0, (int)g.tile_h), y1 = clampi(f2i_trunc_sat((my + ry) / ts) + 1, 0, (int)g.tile_h);
    const int area = max(x1 - x0, 0) * max(y1 - y0, 0);
    return 62 - min(area, 62); // 0 = the largest boxes, 62 = nothing on screen
}

// visits the rows [lo, hi) of a chunk in balanced order: `load(row)` for ALL rows of this thread first, then
// `body(row, data)` once per row. The walk of a row starts with two dependent global reads (radii, then the geometry of a
// live row); issued row by row they cost two full memory latencies per row at 16 waves per CU - 46 of the 62 us of the
// counting kernel on c3 (r05 ablation). Loading the kPer rows of a thread up front turns eight serial latencies into one.
template <int kFusedThreads, typename Classify, typename Load, typename Body>
__device__ __forceinline__ void for_rows_balanced(const FusedArgs &a, int64_t lo, int64_t hi, uint16_t *s_order, int32_t *s_cnt,
                                                  Classify &&classify, Load &&load, Body &&body)
{
    constexpr int kFusedSub = kFusedPer * kFusedThreads; // s_order: kFusedSub entries, s_cnt: 64 (static LDS of the caller:
                                                         // a kernel that calls this twice must not pay for two copies)
    static_assert(kFusedSub <= 65536, "order entries are uint16");
    constexpr int kPer = kFusedPer;
    for (int64_t sub = lo; sub < hi; sub += kFusedSub) {
        const int n = (int)min((int64_t)kFusedSub, hi - sub);
        if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        int cls[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = (int)threadIdx.x + q * kFusedThreads;
            cls[q]      = i < n ? classify(sub + i) : -1;
            if (cls[q] >= 0) atomicAdd(&s_cnt[cls[q]], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) { // exclusive scan of the 64 class counts by the first wave
            const int c = s_cnt[threadIdx.x];
            int incl    = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if ((int)threadIdx.x >= o) incl += up;
            }
            s_cnt[threadIdx.x] = incl - c;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if (cls[q] >= 0) s_order[atomicAdd(&s_cnt[cls[q]], 1)] = (uint16_t)((int)threadIdx.x + q * kFusedThreads);
        __syncthreads();
        int64_t rows[kPer];
        decltype(load((int64_t)0)) data[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = (int)threadIdx.x + q * kFusedThreads;
            rows[q]     = i < n ? sub + (int64_t)s_order[i] : -1;
            if (rows[q] >= 0) data[q] = load(rows[q]);
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if (rows[q] >= 0) body(rows[q], data[q]);
        __syncthreads(); // s_order / s_cnt are reused by the next sub-chunk
    }
}

template <int kFusedThreads>
__global__ void __launch_bounds__(kFusedThreads) fused_count_hist_kernel(const FusedArgs a)
{
    extern __shared__ int32_t s_hist[];
    __shared__ uint16_t s_order[kFusedPer * kFusedThreads];
    __shared__ int32_t s_cnt[64];
    const FusedGeom &g = a.geom;
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) s_hist[t] = 0;
    __syncthreads();
    int64_t lo, hi;
    chunk_rows(g, blockIdx.x, lo, hi);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)(blockIdx.x / g.cpi) * g.n_tiles : nullptr;
    for_rows_balanced<kFusedThreads>(
        a, lo, hi, s_order, s_cnt, [&](int64_t r) { return size_class(a, r); },
        [&](int64_t r) { return load_row_geom(a, r, has_conic); },
        [&](int64_t r, const RowGeom &q) {
            int32_t n = 0;
            SpanPacker sp;
            if (q.live)
                walk_spans(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h,
                           [&](bool alongY, int u, int tv0, int tv1) {
                               if (a.spans) sp.add(alongY, u, tv0, tv1);
                               for (int v = tv0; v < tv1; ++v) {
                                   const int64_t tile = alongY ? (int64_t)u * g.tile_w + v : (int64_t)v * g.tile_w + u;
                                   if (!tmask || tmask[tile]) atomicAdd(&s_hist[tile], 1);
                                   ++n;
                               }
                           });
            if (a.tiles_per_gauss) a.tiles_per_gauss[r] = n;
            if (a.spans) a.spans[r] = sp.record();
        });
    __syncthreads();
    int32_t *out = a.table + (int64_t)blockIdx.x * g.n_tiles;
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) out[t] = s_hist[t];
}

template <int kFusedThreads>
__global__ void __launch_bounds__(kFusedThreads) fused_emit_scatter_kernel(const FusedArgs a)
{
    extern __shared__ int32_t s_cur[];
    __shared__ uint16_t s_order[kFusedPer * kFusedThreads];
    __shared__ int32_t s_cnt[64];
    const FusedGeom &g = a.geom;
    const uint32_t img = blockIdx.x / g.cpi;
    const int32_t *pre = a.table + (int64_t)blockIdx.x * g.n_tiles;      // exclusive prefix over this image's chunks
    const int32_t *off = a.isect_offsets + (int64_t)img * g.n_tiles;     // start of every (image, tile) segment
    for (uint32_t t = threadIdx.x; t < g.n_tiles; t += kFusedThreads) s_cur[t] = off[t] + pre[t];
    __syncthreads();
    int64_t lo, hi;
    chunk_rows(g, blockIdx.x, lo, hi);
    const bool has_conic = (a.conics != nullptr) && (a.opacities != nullptr);
    const uint8_t *tmask = a.tile_mask ? a.tile_mask + (size_t)img * g.n_tiles : nullptr;
    if (a.spans) { // the counting pass recorded what its walk found
        struct RowEmit {
            SpanRecord rec;
            uint32_t dbits;
        };
        for_rows_balanced<kFusedThreads>(
            a, lo, hi, s_order, s_cnt,
            [&](int64_t r) { // most tiles first; rows without tiles last
                const int n = span_tiles(a.spans[r]);
                return n == 0 ? 63 : 62 - min(n, 62);
            },
            [&](int64_t r) {
                RowEmit e;
                e.rec   = a.spans[r];
                e.dbits = __float_as_uint(a.depths[r]);
                return e;
            },
            [&](int64_t r, const RowEmit &e) {
                const uint32_t dbits = e.dbits;
                auto put = [&](int64_t tile) {
                    if (tmask && !tmask[tile]) return;
                    const int32_t slot = atomicAdd(&s_cur[tile], 1);
                    a.bucketed[slot]   = make_uint2(dbits, (uint32_t)r);
                };
                if (span_slabs(e.rec) != kSpanWalk) {
                    span_tiles_visit(e.rec, g.tile_w, put);
                    return;
                }
                const RowGeom q = load_row_geom(a, r, has_conic); // more slabs / longer spans than a record holds: walk again
                if (!q.live) return;
                walk_tiles(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h, put);
            });
        return;
    }
    struct RowEmit {
        RowGeom q;
        uint32_t dbits;
    };
    for_rows_balanced<kFusedThreads>(
        a, lo, hi, s_order, s_cnt, [&](int64_t r) { return size_class(a, r); },
        [&](int64_t r) {
            RowEmit e;
            e.q     = load_row_geom(a, r, has_conic);
            e.dbits = __float_as_uint(a.depths[r]);
            return e;
        },
        [&](int64_t r, const RowEmit &e) {
            const RowGeom &q = e.q;
            if (!q.live) return;
            const uint32_t dbits = e.dbits;
            walk_tiles(q.mx, q.my, q.rx, q.ry, has_conic, q.A, q.B, q.C, q.op, g.tile_size, g.tile_w, g.tile_h,
                       [&](int64_t tile) {
                           if (tmask && !tmask[tile]) return;
                           const int32_t slot = atomicAdd(&s_cur[tile], 1);
                           a.bucketed[slot]   = make_uint2(dbits, (uint32_t)r);
                       });
        });
}

template <int T>
static void set_lds_limit_once()
{
    static PerDeviceOnce once;
    if (once.first()) {
        // a refused limit would otherwise surface as the NEXT launch's "invalid argument" (hipGetLastError is sticky)
        if (hipFuncSetAttribute((const void *)fused_count_hist_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedMaxDynLds) != hipSuccess
            || hipFuncSetAttribute((const void *)fused_emit_scatter_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedMaxDynLds) != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "gsplat_amd: isect_fused: dynamic LDS limit of %d bytes refused (static LDS grew?)\n", kFusedMaxDynLds);
        }
    }
}

int launch_fused_count_hist(const FusedArgs &a, hipStream_t s)
{
    const size_t lds = (size_t)a.geom.n_tiles * sizeof(int32_t);
    if (a.geom.threads == 512) {
        set_lds_limit_once<512>();
        fused_count_hist_kernel<512><<<dim3(a.geom.n_chunks), dim3(512), lds, s>>>(a);
    } else {
        set_lds_limit_once<1024>();
        fused_count_hist_kernel<1024><<<dim3(a.geom.n_chunks), dim3(1024), lds, s>>>(a);
    }
    return check_launch("isect_fused_count");
}

int launch_fused_emit_scatter(const FusedArgs &a, hipStream_t s)
{
    const size_t lds = (size_t)a.geom.n_tiles * sizeof(int32_t);
    if (a.geom.threads == 512) {
        set_lds_limit_once<512>();
        fused_emit_scatter_kernel<512><<<dim3(a.geom.n_chunks), dim3(512), lds, s>>>(a);
    } else {
        set_lds_limit_once<1024>();
        fused_emit_scatter_kernel<1024><<<dim3(a.geom.n_chunks), dim3(1024), lds, s>>>(a);
    }
    return check_launch("isect_fused_emit");
}

} // namespace gsx
