// Sort of the (image | tile | depth) intersection keys, MI355X-native: ONE bucketing pass by (image, tile) + a per-tile
// depth sort that never leaves LDS.
//
// C-ABI entry: gsx_isect_tile_sort (+ gsx_isect_tile_sort_workspace_bytes). Replaces the 6-pass device-wide
// cub::DeviceRadixSort::SortPairs of the reference (gsplat/cuda/csrc/IntersectTile.cu:1078-1121) for the keys produced by
// gsx_isect_emit; the result is IDENTICAL to a stable ascending sort on the full key (ties keep ascending flatten id,
// which is the emission order), so isect_ids / flatten_ids stay bit-exact with the reference and the oracle.
//
// Why this shape on MI355X: a generic LSD radix sort moves every 12-byte pair through HBM once per 8 key bits
// (6 round trips at 1080p). The key is structured — the high bits name one of I*th*tw tiles and a tile's list
// (hundreds to a few thousand entries) fits in the 160 KiB LDS of a CU — so:
//   A  bucket_hist    LDS-privatised histogram of tile ids per contiguous chunk of the unsorted list
//   B  scan           exclusive scan of the [tile][chunk] table            (gsx scan kernels, scan_sort.hip)
//   C  bucket_scatter each pair goes to its tile's segment (LDS cursor atomics give the slot; arrival order inside a
//                     tile is arbitrary) as one 8-byte (depth bits, flatten id) store
//   D  tile_sort      one workgroup per tile. n <= 2048: bitonic sort of the 64-bit (depth, flatten id) words in LDS (ties
//                     come out in ascending id order). Longer tiles (work list, persistent grid): stable 4 x 8-bit LSD
//                     radix sort of the 32 depth bits in LDS (wave64 ballot ranking, as scan_sort.hip) followed by
//                     ordering the runs of EQUAL depth by flatten id; tiles beyond the LDS capacity run the same passes
//                     through global ping-pong buffers (rare; correct, slower). Either way the output does not depend
//                     on the arrival order produced by C.
// HBM traffic: read 12 B + write 8 B (C) + read 8 B + write 12 B (D) per pair = 40 B instead of 144 B.
#include "bitonic64.hpp"
#include "isect_fused.hpp"
#include <cstdlib>
#include <cstring>

namespace gsx {

int run_scan_i32_exclusive(const int32_t *in, int64_t n, int32_t *out, void *ws, int64_t ws_bytes, hipStream_t stream);
int64_t scan_workspace_bytes_for(int64_t n);

constexpr int kTsThreads   = 256;
constexpr uint32_t kMaxBins = 36864; // LDS histogram: 144 KiB of the 160 KiB
constexpr int kCapSmall    = 2048;  // entries sorted in 32 KiB-class LDS
constexpr int kCapLarge    = 9152;  // entries sorted with (almost) the whole LDS: 2 x 9152 x 8 B + the 16 KiB of per-wave digit bases < 160 KiB

// struct TileSortArgs: isect_fused.hpp (shared with isect_binned.hip)

__device__ __forceinline__ uint32_t key_bin(uint64_t key, uint32_t n_tiles, uint32_t tile_bits)
{
    const uint64_t hi = key >> 32;
    const uint64_t tmask = tile_bits >= 32 ? 0xFFFFFFFFull : ((1ull << tile_bits) - 1ull);
    return (uint32_t)((hi >> tile_bits) * n_tiles + (hi & tmask));
}

// A: per-chunk histogram of bins, LDS-privatised. Chunks are LARGE (>= 32 Ki pairs): the [bin][chunk] table is written
// with a stride of n_chunks ints (one cache line per entry), so its size — not the key stream — sets the cost of A, B
// and the cursor load of C; large chunks also put several pairs of one bin next to each other in C's output.
constexpr int kBkThreads = 1024;
constexpr int kBkUnroll  = 4;

__global__ void __launch_bounds__(kBkThreads) bucket_hist_kernel(const TileSortArgs a)
{
    extern __shared__ int32_t s_hist[];
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += kBkThreads) s_hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * a.chunk_len;
    const int64_t hi = min(a.n, lo + a.chunk_len);
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)kBkThreads * kBkUnroll) {
        uint64_t k[kBkUnroll];
#pragma unroll
        for (int u = 0; u < kBkUnroll; ++u) {
            const int64_t i = i0 + (int64_t)u * kBkThreads;
            k[u] = i < hi ? a.keys_in[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kBkUnroll; ++u)
            if (i0 + (int64_t)u * kBkThreads < hi) atomicAdd(&s_hist[key_bin(k[u], a.n_tiles, a.tile_bits)], 1);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += kBkThreads) a.table[(int64_t)i * a.n_chunks + blockIdx.x] = s_hist[i];
}

// C: scatter into tile segments
__global__ void __launch_bounds__(kBkThreads) bucket_scatter_kernel(const TileSortArgs a)
{
    extern __shared__ int32_t s_cur[];
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += kBkThreads)
        s_cur[i] = a.table_scanned[(int64_t)i * a.n_chunks + blockIdx.x];
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * a.chunk_len;
    const int64_t hi = min(a.n, lo + a.chunk_len);
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)kBkThreads * kBkUnroll) {
        uint64_t k[kBkUnroll];
        int32_t v[kBkUnroll];
#pragma unroll
        for (int u = 0; u < kBkUnroll; ++u) {
            const int64_t i = i0 + (int64_t)u * kBkThreads;
            k[u] = i < hi ? a.keys_in[i] : 0ull;
            v[u] = i < hi ? a.vals_in[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < kBkUnroll; ++u)
            if (i0 + (int64_t)u * kBkThreads < hi) {
                const int32_t dst = atomicAdd(&s_cur[key_bin(k[u], a.n_tiles, a.tile_bits)], 1);
                a.bucketed[dst]   = make_uint2((uint32_t)k[u], (uint32_t)v[u]);
            }
    }
}

// ------------------------------------------------------------------------------------------
// D: per-tile stable LSD radix sort on the 32 depth bits (+ tie ordering by id)
// ------------------------------------------------------------------------------------------
// One stable 8-bit pass over n elements src -> dst (both LDS or both global) by a workgroup of NW waves; wave w owns the
// contiguous run [w*per_wave, (w+1)*per_wave). s_base [NW][256]: per-wave digit counts, turned in place into per-wave bases.
template <int NW, typename Ptr>
__device__ __forceinline__ void radix_pass(Ptr src, Ptr dst, int n, int shift, int32_t (*s_base)[256], int64_t *s_scan)
{
    static_assert(NW >= 4, "the digit scan runs on the first four waves");
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int per_wave = ((n + 64 * NW - 1) / (64 * NW)) * 64; // multiple of 64
    const int w_lo = wave * per_wave;
    for (int i = threadIdx.x; i < NW * 256; i += NW * 64) (&s_base[0][0])[i] = 0;
    __syncthreads();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // count
    for (int r = 0; r < per_wave; r += 64) {
        const int idx   = w_lo + r + lane;
        const bool live = idx < n;
        const int d     = live ? (int)((src[idx].x >> shift) & 0xFFu) : 0;
        if (live) atomicAdd(&s_base[wave][d], 1);
    }
    __syncthreads();
    // thread d < 256: digit totals -> exclusive scan over the digits, then per-wave bases (in place)
    int32_t tot = 0, inc = 0;
    if (threadIdx.x < 256u) {
        const int d = (int)threadIdx.x;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += s_base[w][d];
        inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(inc, o);
            if (lane >= o) inc += y;
        }
        if (lane == 63) s_scan[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < 256u) {
        const int d = (int)threadIdx.x;
        int32_t base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < wave) base += (int32_t)s_scan[w];
        int32_t run = base + inc - tot;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int32_t c = s_base[w][d];
            s_base[w][d]    = run;
            run += c;
        }
    }
    __syncthreads();
    // rank + scatter, round by round (s_base[wave][d] is advanced as the wave's running counter)
    for (int r = 0; r < per_wave; r += 64) {
        const int idx   = w_lo + r + lane;
        const bool live = idx < n;
        const uint2 e   = live ? src[idx] : make_uint2(0u, 0u);
        const int d     = (int)((e.x >> shift) & 0xFFu);
        uint64_t peers  = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool b     = (d >> bit) & 1;
            const uint64_t m = __builtin_amdgcn_ballot_w64(b);
            peers &= b ? m : ~m;
        }
        volatile int32_t *cnt = &s_base[wave][0];
        const int before = live ? cnt[d] : 0;
        const int pos    = __popcll(peers & lt_mask);
        if (live) dst[before + pos] = e;
        if (live && pos == 0) cnt[d] = before + __popcll(peers);
    }
    __syncthreads();
}

// runs of equal depth: order by flatten id (ascending) — one thread per run start; runs are short (exact depth ties)
template <typename Ptr>
__device__ __forceinline__ void fix_ties(Ptr a, int n)
{
    for (int i = threadIdx.x; i < n; i += (int)blockDim.x) {
        const uint32_t k = a[i].x;
        if ((i == 0 || a[i - 1].x != k) && i + 1 < n && a[i + 1].x == k) {
            int j = i + 1;
            while (j < n && a[j].x == k) ++j;
            for (int p = i + 1; p < j; ++p) { // insertion sort of ids in [i, j)
                const uint32_t v = a[p].y;
                int q = p - 1;
                while (q >= i && a[q].y > v) {
                    a[q + 1].y = a[q].y;
                    --q;
                }
                a[q + 1].y = v;
            }
        }
    }
    __syncthreads();
}

// Small tiles (n <= kCapSmall): bitonic sort of the 64-bit (depth bits, flatten id) words in LDS (bitonic64.hpp). The id in
// the low half makes exact depth ties come out in ascending id order, so no tie pass is needed. Longer tiles are appended to
// the work list for the radix kernel below.
constexpr int kTsSmallWords = kCapSmall + kCapSmall / 8;

__global__ void __launch_bounds__(kTsThreads) tile_sort_small_kernel(const TileSortArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *s = reinterpret_cast<uint64_t *>(smem_raw);
    const uint32_t bin  = blockIdx.x;
    const int64_t start = a.table_scanned[(int64_t)bin * a.n_chunks];
    const int64_t end   = (bin + 1 == a.n_bins) ? a.n : a.table_scanned[(int64_t)(bin + 1) * a.n_chunks];
    const int n         = (int)(end - start);
    if (n <= 0) return;
    if (n > kCapSmall) {
        if (threadIdx.x == 0) a.big_list[atomicAdd(a.big_count, 1)] = (int32_t)bin;
        return;
    }
    int lp = 7; // at least 128 words
    while ((1 << lp) < n) ++lp;
    const int P = 1 << lp;
    const uint2 *g_in = a.bucketed + start;
    // the thread's words wait in registers until the workgroup knows which network sorts them (the pad differs)
    constexpr int kPer = kCapSmall / kTsThreads;
    uint64_t w[kPer];
    bool odd = a.sort_int != 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int i = (int)threadIdx.x + u * kTsThreads;
        w[u]        = 0ull;
        if (i < n) {
            const uint2 e = g_in[i];
            w[u]          = ((uint64_t)e.x << 32) | e.y;
            odd |= bt_key_is_odd(e.x);
        }
    }
    const bool as_int  = __syncthreads_or(odd);
    const uint64_t pad = as_int ? kBtPadInt : kBtPadF64;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int i = (int)threadIdx.x + u * kTsThreads;
        if (i < P) s[bt_phys(i)] = i < n ? w[u] : pad;
    }
    __syncthreads();
    if (as_int) bt_sort_int<kTsThreads>(s, lp, (int)threadIdx.x, [] { __syncthreads(); });
    else bt_sort_f64<kTsThreads>(s, lp, (int)threadIdx.x, [] { __syncthreads(); });
    const uint64_t tile = bin % a.n_tiles, img = bin / a.n_tiles;
    const uint64_t hi   = ((img << a.tile_bits) | tile) << 32;
    for (int i = threadIdx.x; i < n; i += kTsThreads) {
        const uint64_t v      = s[bt_phys(i)];
        a.keys_out[start + i] = hi | (v >> 32);
        a.vals_out[start + i] = (int32_t)(uint32_t)v;
    }
}

// A small persistent grid walks the work list (tile_sort_small_kernel appends the lists longer than kCapSmall): n <= CAP
// sorts in (nearly all of) the LDS, longer tiles sort
//         through global memory. With no oversized tile the launch costs a few microseconds.
constexpr int kWlWaves = 16; // work-list sort: 1024 threads (four waves per workgroup left 3/4 of the CU's issue slots empty)
template <int CAP, int MODE>
__global__ void __launch_bounds__(kWlWaves * 64) tile_sort_kernel(const TileSortArgs a)
{
    static_assert(MODE == 1, "only the work-list mode is instantiated");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int32_t s_base[kWlWaves][256];
    __shared__ int64_t s_scan[4];
    uint2 *s_a = reinterpret_cast<uint2 *>(smem_raw);
    uint2 *s_b = s_a + CAP;
    constexpr int NT = kWlWaves * 64;

  const int32_t n_work = *a.big_count;
  for (int32_t wi = (int32_t)blockIdx.x; wi < n_work; wi += (int32_t)gridDim.x) {
    const uint32_t bin  = (uint32_t)a.big_list[wi];
    const int64_t start = a.table_scanned[(int64_t)bin * a.n_chunks];
    const int64_t end   = (bin + 1 == a.n_bins) ? a.n : a.table_scanned[(int64_t)(bin + 1) * a.n_chunks];
    const int n         = (int)(end - start);
    // high 32 bits of every key of this bin
    const uint64_t tile = bin % a.n_tiles, img = bin / a.n_tiles;
    const uint64_t hi   = ((img << a.tile_bits) | tile) << 32;
    uint2 *g_in         = a.bucketed + start;

    if (n <= CAP) {
        for (int i = threadIdx.x; i < n; i += NT) s_a[i] = g_in[i];
        __syncthreads();
        radix_pass<kWlWaves>(s_a, s_b, n, 0, s_base, s_scan);
        radix_pass<kWlWaves>(s_b, s_a, n, 8, s_base, s_scan);
        radix_pass<kWlWaves>(s_a, s_b, n, 16, s_base, s_scan);
        radix_pass<kWlWaves>(s_b, s_a, n, 24, s_base, s_scan);
        fix_ties(s_a, n);
        for (int i = threadIdx.x; i < n; i += NT) {
            const uint2 e         = s_a[i];
            a.keys_out[start + i] = hi | e.x;
            a.vals_out[start + i] = (int32_t)e.y;
        }
    } else {
        // oversized tile: same passes through global memory (segment is private to this workgroup; __syncthreads
        // orders global accesses within a workgroup)
        uint2 *g_tmp = a.scratch + start;
        radix_pass<kWlWaves>(g_in, g_tmp, n, 0, s_base, s_scan);
        radix_pass<kWlWaves>(g_tmp, g_in, n, 8, s_base, s_scan);
        radix_pass<kWlWaves>(g_in, g_tmp, n, 16, s_base, s_scan);
        radix_pass<kWlWaves>(g_tmp, g_in, n, 24, s_base, s_scan);
        fix_ties(g_in, n);
        for (int i = threadIdx.x; i < n; i += NT) {
            const uint2 e         = g_in[i];
            a.keys_out[start + i] = hi | e.x;
            a.vals_out[start + i] = (int32_t)e.y;
        }
    }
    __syncthreads(); // LDS / counters are reused by the next work item
  }
}

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

bool bitonic_f64_enabled() // read per call: the tests switch it inside one process
{
    const char *e = getenv("GSX_ISECT_SORT");
    return !(e && strcmp(e, "int") == 0);
}

static uint32_t ts_chunks(int64_t n)
{
    int64_t c = ceil_div(n, 32768);
    if (c < 1) c = 1;
    if (c > 512) c = 512;
    return (uint32_t)c;
}

static int64_t tile_sort_ws_bytes(int64_t n, uint32_t n_bins)
{
    const int64_t table = (int64_t)n_bins * ts_chunks(n) * (int64_t)sizeof(int32_t);
    return 2 * align256(table) + 2 * align256(n * (int64_t)sizeof(uint2))
           + align256(scan_workspace_bytes_for((int64_t)n_bins * ts_chunks(n))) + align256(((int64_t)n_bins + 1) * 4) + 512;
}

// ------------------------------------------------------------------------------------------
// fused path (isect_fused.hip): column scan of the [chunk][tile] table and scan of the tile totals
// ------------------------------------------------------------------------------------------
// Exclusive running sum over an image's chunks for every (image, tile), in place; totals[bin] = tile count.
// A workgroup owns 32 consecutive tiles of one image (lanes along the tile axis: 128-byte coalesced rows) x 8 segments
// of the chunk axis: pass 1 sums each segment, an LDS step turns the 8 segment sums into segment bases, pass 2 rewrites
// the segment with the running prefix. Two reads of the table, but 8x32 independent columns per workgroup instead of
// one serial walk per tile (which was latency-bound: 139 us for a 489 x 8160 table).
constexpr int kCsTiles = 32, kCsSegs = 8;
__global__ void __launch_bounds__(kCsTiles *kCsSegs) fused_colscan_kernel(int32_t *table, int32_t *totals, uint32_t n_tiles,
                                                                          uint32_t cpi, uint32_t tile_groups)
{
    __shared__ int32_t s_seg[kCsSegs][kCsTiles];
    const uint32_t img = blockIdx.x / tile_groups, tg = blockIdx.x % tile_groups;
    const uint32_t lane_t = threadIdx.x % kCsTiles, seg = threadIdx.x / kCsTiles;
    const uint32_t t      = tg * kCsTiles + lane_t;
    const bool live       = t < n_tiles;
    const uint32_t per    = (cpi + kCsSegs - 1) / kCsSegs;
    const uint32_t c0 = seg * per, c1 = min(c0 + per, cpi);
    int32_t *col = table + (int64_t)img * cpi * n_tiles + t;
    int32_t sum  = 0;
    if (live)
        for (uint32_t c = c0; c < c1; ++c) sum += col[(int64_t)c * n_tiles];
    s_seg[seg][lane_t] = sum;
    __syncthreads();
    int32_t run = 0;
#pragma unroll
    for (int k = 0; k < kCsSegs; ++k)
        if (k < (int)seg) run += s_seg[k][lane_t];
    if (live) {
        for (uint32_t c = c0; c < c1; ++c) {
            const int32_t v           = col[(int64_t)c * n_tiles];
            col[(int64_t)c * n_tiles] = run;
            run += v;
        }
        if (seg == kCsSegs - 1) totals[(int64_t)img * n_tiles + t] = run;
    }
}

// Exclusive scan of the (image, tile) totals by ONE workgroup (n_bins <= kMaxBins): offsets[b] = start of segment b,
// *n_isects = grand total (int64; the caller rejects >= 2^31 before any int32 offset is used).
__global__ void __launch_bounds__(1024) fused_totals_scan_kernel(const int32_t *totals, uint32_t n_bins, int32_t *offsets,
                                                                 int64_t *n_isects, int64_t *max_tile_len)
{
    __shared__ int64_t s_part[16];
    __shared__ int32_t s_max;
    const int64_t total = block_scan_i32_1024(totals, offsets, n_bins, s_part, max_tile_len ? &s_max : nullptr);
    if (threadIdx.x == 0) {
        // the longest tile list is written before n_isects (the host reads it once n_isects has arrived); n_isects may live
        // in pinned host memory that the host polls: one 8-byte system-scope store behind a system-scope fence
        if (max_tile_len) *max_tile_len = (int64_t)s_max;
        __threadfence_system();
        *n_isects = total;
    }
}

// shared with isect_binned.hip: the same column scan over a [chunk][bin] table, and the work-list sort of oversized tiles
int launch_colscan(int32_t *table, int32_t *totals, uint32_t n_cols, uint32_t cpi, uint32_t n_images, hipStream_t s)
{
    const uint32_t groups = (n_cols + kCsTiles - 1) / kCsTiles;
    fused_colscan_kernel<<<dim3(groups * n_images), dim3(kCsTiles * kCsSegs), 0, s>>>(table, totals, n_cols, cpi, groups);
    return check_launch("isect colscan");
}
// The work list (lists longer than kCapSmall) on a persistent grid of one 147 KiB workgroup per CU. Two size classes (a
// second instantiation with half the LDS for lists up to 4608 entries, two workgroups per CU) were measured and dropped:
// c4 110 vs 114 us, garden x25 +45 us - the launches run one after the other and each has its own tail.
static int launch_work_list_sorts(TileSortArgs a, hipStream_t s)
{
    static PerDeviceOnce once;
    if (once.first())
        (void)hipFuncSetAttribute((const void *)tile_sort_kernel<kCapLarge, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(2 * kCapLarge * sizeof(uint2)));
    tile_sort_kernel<kCapLarge, 1><<<dim3(256), dim3(kWlWaves * 64), 2 * kCapLarge * sizeof(uint2), s>>>(a);
    return check_launch("isect work-list tile sort");
}

int launch_big_tile_sort(const TileSortArgs &a, hipStream_t s) { return launch_work_list_sorts(a, s); }

static int64_t fused_count_ws_bytes(const FusedGeom &g)
{
    return align256((int64_t)g.n_chunks * g.n_tiles * 4) + align256((int64_t)g.n_images * g.n_tiles * 4)
           + (fused_records_spans(g.rows) ? align256((g.rows > 0 ? g.rows : 1) * kSpanRecordBytes) : 0) + 512;
}
static int64_t fused_emit_ws_bytes(int64_t n, uint32_t n_bins)
{
    return 2 * align256(n * (int64_t)sizeof(uint2)) + align256(((int64_t)n_bins + 1) * 4) + 512;
}

static uint32_t bits_for(uint64_t count)
{
    uint32_t b = 0;
    if (count <= 1) return 0;
    uint64_t v = count - 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_isect_tile_sort_supported(uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    const uint64_t bins = (uint64_t)n_images * tile_w * tile_h;
    return bins >= 1 && bins <= kMaxBins;
}

extern "C" int64_t gsx_isect_tile_sort_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return tile_sort_ws_bytes(n_isects > 0 ? n_isects : 1, n_images * tile_w * tile_h);
}

extern "C" int gsx_isect_tile_sort(const int64_t *isect_ids, const int32_t *flatten_ids, int64_t n_isects,
                                   uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int64_t *isect_ids_sorted,
                                   int32_t *flatten_ids_sorted, void *workspace, int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(n_isects >= 0 && n_isects < (1ll << 31), "gsx_isect_tile_sort: n_isects out of range");
    if (n_isects == 0) return GSX_OK;
    GSX_REQUIRE(gsx_isect_tile_sort_supported(n_images, tile_w, tile_h),
                "gsx_isect_tile_sort: %u images x %u x %u tiles exceed the LDS histogram (use gsx_sort_pairs)", n_images,
                tile_w, tile_h);
    GSX_REQUIRE(isect_ids && flatten_ids && isect_ids_sorted && flatten_ids_sorted, "gsx_isect_tile_sort: null buffer");
    const uint32_t n_tiles = tile_w * tile_h, n_bins = n_images * n_tiles;
    if (workspace == nullptr || workspace_bytes < tile_sort_ws_bytes(n_isects, n_bins)) {
        set_last_error("gsx_isect_tile_sort: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    TileSortArgs a{};
    a.sort_int = bitonic_f64_enabled() ? 0 : 1;
    a.keys_in = reinterpret_cast<const uint64_t *>(isect_ids);
    a.vals_in = flatten_ids;
    a.n = n_isects; a.n_tiles = n_tiles; a.tile_bits = bits_for(n_tiles); a.n_bins = n_bins;
    a.n_chunks  = ts_chunks(n_isects);
    a.chunk_len = ceil_div(n_isects, (int64_t)a.n_chunks);
    const int64_t table_elems = (int64_t)n_bins * a.n_chunks;
    unsigned char *p = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.table         = reinterpret_cast<int32_t *>(p); p += align256(table_elems * 4);
    a.table_scanned = reinterpret_cast<int32_t *>(p); p += align256(table_elems * 4);
    a.bucketed      = reinterpret_cast<uint2 *>(p);   p += align256(n_isects * 8);
    a.scratch       = reinterpret_cast<uint2 *>(p);   p += align256(n_isects * 8);
    void *scan_ws   = p; p += align256(scan_workspace_bytes_for(table_elems));
    a.big_count     = reinterpret_cast<int32_t *>(p);
    a.big_list      = a.big_count + 1;
    a.keys_out = reinterpret_cast<uint64_t *>(isect_ids_sorted);
    a.vals_out = flatten_ids_sorted;

    const size_t hist_lds = (size_t)n_bins * sizeof(int32_t);
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void *)bucket_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute((const void *)bucket_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    }
    bucket_hist_kernel<<<dim3(a.n_chunks), dim3(kBkThreads), hist_lds, s>>>(a);
    int rc = run_scan_i32_exclusive(a.table, table_elems, a.table_scanned, scan_ws, scan_workspace_bytes_for(table_elems), s);
    if (rc != GSX_OK) return rc;
    bucket_scatter_kernel<<<dim3(a.n_chunks), dim3(kBkThreads), hist_lds, s>>>(a);
    if (hipMemsetAsync(a.big_count, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("isect_tile_sort memset");
    tile_sort_small_kernel<<<dim3(n_bins), dim3(kTsThreads), kTsSmallWords * sizeof(uint64_t), s>>>(a);
    return launch_work_list_sorts(a, s);
}

// ---------------------------------------------------------------------------------------------------
// fused intersection (see isect_fused.hip)
// ---------------------------------------------------------------------------------------------------
extern "C" int gsx_isect_fused_supported(uint32_t n_images, uint32_t tile_w, uint32_t tile_h, int packed)
{
    const uint64_t tiles = (uint64_t)tile_w * tile_h, bins = tiles * n_images;
    if (packed && n_images != 1) return 0; // packed rows: per-image row counts live on the device
    return n_images >= 1 && tiles >= 1 && bins <= kMaxBins;
}

extern "C" int64_t gsx_isect_fused_count_workspace_bytes(int64_t rows, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return fused_count_ws_bytes(fused_geometry(rows > 0 ? rows : 1, n_images, 16, tile_w, tile_h));
}

extern "C" int64_t gsx_isect_fused_emit_workspace_bytes(int64_t n_isects, uint32_t n_images, uint32_t tile_w, uint32_t tile_h)
{
    return fused_emit_ws_bytes(n_isects > 0 ? n_isects : 1, n_images * tile_w * tile_h);
}

static int fused_setup(const char *fn, FusedArgs &a, int64_t rows, uint32_t n_images, uint32_t tile_size, uint32_t tile_w,
                       uint32_t tile_h, void *count_ws, int64_t count_ws_bytes, int32_t **totals)
{
    GSX_REQUIRE(rows >= 0 && tile_size > 0, "%s: bad rows / tile_size", fn);
    GSX_REQUIRE(gsx_isect_fused_supported(n_images, tile_w, tile_h, 0), "%s: %u images x %u x %u tiles not supported", fn,
                n_images, tile_w, tile_h);
    GSX_REQUIRE(rows % n_images == 0, "%s: rows (%lld) must be a multiple of n_images (%u)", fn, (long long)rows, n_images);
    a.geom = fused_geometry(rows, n_images, tile_size, tile_w, tile_h);
    if (count_ws == nullptr || count_ws_bytes < fused_count_ws_bytes(a.geom)) {
        set_last_error("%s: count workspace too small", fn);
        return GSX_ERR_WORKSPACE;
    }
    unsigned char *p = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(count_ws) + 255) & ~(uintptr_t)255);
    a.table = reinterpret_cast<int32_t *>(p); p += align256((int64_t)a.geom.n_chunks * a.geom.n_tiles * 4);
    *totals = reinterpret_cast<int32_t *>(p); p += align256((int64_t)a.geom.n_images * a.geom.n_tiles * 4);
    a.spans = fused_records_spans(a.geom.rows) ? reinterpret_cast<SpanRecord *>(p) : nullptr;
    return GSX_OK;
}

extern "C" int gsx_isect_fused_count(const float *means2d, const int32_t *radii, const float *conics, const float *opacities,
                                     const uint8_t *tile_mask, int64_t rows, uint32_t n_images, uint32_t tile_size,
                                     uint32_t tile_w, uint32_t tile_h,
                                     int32_t *tiles_per_gauss, int32_t *isect_offsets, int64_t *n_isects, int64_t *max_tile_len,
                                     void *count_workspace, int64_t count_workspace_bytes, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const uint32_t n_bins = n_images * tile_w * tile_h;
    GSX_REQUIRE(isect_offsets && n_isects, "gsx_isect_fused_count: null output");
    if (rows == 0) {
        if (hipMemsetAsync(isect_offsets, 0, (size_t)n_bins * 4, s) != hipSuccess
            || hipMemsetAsync(n_isects, 0, 8, s) != hipSuccess
            || (max_tile_len && hipMemsetAsync(max_tile_len, 0, 8, s) != hipSuccess)) {
            set_last_error("gsx_isect_fused_count: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    GSX_REQUIRE(means2d && radii, "gsx_isect_fused_count: null pointer");
    FusedArgs a{};
    int32_t *totals = nullptr;
    int rc = fused_setup("gsx_isect_fused_count", a, rows, n_images, tile_size, tile_w, tile_h, count_workspace,
                         count_workspace_bytes, &totals);
    if (rc != GSX_OK) return rc;
    a.means2d = means2d; a.radii = radii; a.conics = conics; a.opacities = opacities; a.tiles_per_gauss = tiles_per_gauss;
    a.tile_mask = tile_mask;
    rc = launch_fused_count_hist(a, s);
    if (rc != GSX_OK) return rc;
    const uint32_t tile_groups = (a.geom.n_tiles + kCsTiles - 1) / kCsTiles;
    fused_colscan_kernel<<<dim3(tile_groups * n_images), dim3(kCsTiles * kCsSegs), 0, s>>>(a.table, totals, a.geom.n_tiles,
                                                                                         a.geom.cpi, tile_groups);
    fused_totals_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(totals, n_bins, isect_offsets, n_isects, max_tile_len);
    return check_launch("isect_fused_count scans");
}

extern "C" int gsx_isect_fused_emit_sort(const float *means2d, const int32_t *radii, const float *depths,
                                         const float *conics, const float *opacities, const uint8_t *tile_mask,
                                         int64_t rows, uint32_t n_images,
                                         uint32_t tile_size, uint32_t tile_w, uint32_t tile_h, void *count_workspace,
                                         int64_t count_workspace_bytes, const int32_t *isect_offsets, int64_t n_isects,
                                         int64_t *isect_ids_sorted, int32_t *flatten_ids_sorted, void *workspace,
                                         int64_t workspace_bytes, void *stream)
{
    GSX_REQUIRE(n_isects >= 0 && n_isects < (1ll << 31), "gsx_isect_fused_emit_sort: n_isects out of range");
    if (n_isects == 0 || rows == 0) return GSX_OK;
    GSX_REQUIRE(means2d && radii && depths && isect_offsets && isect_ids_sorted && flatten_ids_sorted,
                "gsx_isect_fused_emit_sort: null pointer");
    hipStream_t s = (hipStream_t)stream;
    FusedArgs f{};
    int32_t *totals = nullptr;
    int rc = fused_setup("gsx_isect_fused_emit_sort", f, rows, n_images, tile_size, tile_w, tile_h, count_workspace,
                         count_workspace_bytes, &totals);
    if (rc != GSX_OK) return rc;
    const uint32_t n_tiles = tile_w * tile_h, n_bins = n_images * n_tiles;
    if (workspace == nullptr || workspace_bytes < fused_emit_ws_bytes(n_isects, n_bins)) {
        set_last_error("gsx_isect_fused_emit_sort: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    TileSortArgs a{};
    a.sort_int = bitonic_f64_enabled() ? 0 : 1;
    a.n = n_isects; a.n_tiles = n_tiles; a.tile_bits = bits_for(n_tiles); a.n_bins = n_bins;
    a.n_chunks = 1; // table_scanned[bin * 1] = start of segment `bin`
    a.table_scanned = const_cast<int32_t *>(isect_offsets);
    unsigned char *p = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.bucketed  = reinterpret_cast<uint2 *>(p); p += align256(n_isects * 8);
    a.scratch   = reinterpret_cast<uint2 *>(p); p += align256(n_isects * 8);
    a.big_count = reinterpret_cast<int32_t *>(p);
    a.big_list  = a.big_count + 1;
    a.keys_out = reinterpret_cast<uint64_t *>(isect_ids_sorted);
    a.vals_out = flatten_ids_sorted;
    f.means2d = means2d; f.radii = radii; f.depths = depths; f.conics = conics; f.opacities = opacities;
    f.isect_offsets = isect_offsets; f.bucketed = a.bucketed; f.tile_mask = tile_mask;
    rc = launch_fused_emit_scatter(f, s);
    if (rc != GSX_OK) return rc;
    if (hipMemsetAsync(a.big_count, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("isect_fused memset");
    tile_sort_small_kernel<<<dim3(n_bins), dim3(kTsThreads), kTsSmallWords * sizeof(uint64_t), s>>>(a);
    return launch_work_list_sorts(a, s);
}
