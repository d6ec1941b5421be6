// Device-wide prefix sum and stable LSD radix sort for the tile-intersection stage (gfx950).
//
// Replaces the CUB calls of the reference (cub::DeviceScan / at::cumsum in Intersect.cpp:258,
// cub::DeviceRadixSort::SortPairs in IntersectTile.cu:1078-1121). Written for wave64:
// digit ranking inside a wave uses 64-bit ballots (8 ballots resolve an 8-bit digit for 64 keys),
// every wave owns a CONTIGUOUS run of keys so ranking needs no workgroup barrier per round, and
// the sort is stable (equal keys keep their emission order = ascending flatten id).
#include "common.hpp"

namespace gsx {

// ------------------------------------------------------------------------------------------
// scan: int32 in -> int64/int32 out, inclusive or exclusive. Three kernels:
//   reduce chunks -> scan the chunk sums (one workgroup) -> rescan chunks with their base.
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems   = 16;
constexpr int kScanChunk   = kScanThreads * kScanItems; // 4096 elements per workgroup

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t x)
{
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    return x;
}

// inclusive scan across the 256 threads of the workgroup; returns the inclusive value, total via *total
__device__ __forceinline__ int64_t block_incl_scan_i64(int64_t x, int64_t *s_wave /*[4]*/, int64_t *total)
{
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int64_t inc = wave_incl_scan_i64(x);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const int64_t v = s_wave[w];
        if (w < wave) base += v;
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return inc + base;
}

__global__ void __launch_bounds__(kScanThreads) scan_reduce_kernel(const int32_t *in, int64_t n, int64_t *chunk_sums)
{
    __shared__ int64_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanChunk;
    int64_t acc        = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + (int64_t)i * kScanThreads + threadIdx.x;
        if (idx < n) acc += in[idx];
    }
    int64_t total;
    block_incl_scan_i64(acc, s_wave, &total);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of chunk_sums in place
__global__ void __launch_bounds__(kScanThreads) scan_chunks_kernel(int64_t *chunk_sums, int64_t n_chunks)
{
    __shared__ int64_t s_wave[4];
    int64_t carry = 0;
    for (int64_t base = 0; base < n_chunks; base += kScanThreads) {
        const int64_t idx = base + threadIdx.x;
        const int64_t v   = idx < n_chunks ? chunk_sums[idx] : 0;
        int64_t total;
        const int64_t inc = block_incl_scan_i64(v, s_wave, &total);
        if (idx < n_chunks) chunk_sums[idx] = carry + inc - v;
        carry += total;
    }
}

template <typename OutT, bool INCLUSIVE>
__global__ void __launch_bounds__(kScanThreads)
scan_apply_kernel(const int32_t *in, int64_t n, const int64_t *chunk_bases, OutT *out)
{
    __shared__ int64_t s_wave[4];
    // thread owns kScanItems CONSECUTIVE elements so the scan order is the memory order
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanItems;
    int32_t v[kScanItems];
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        acc += v[i];
    }
    int64_t total;
    const int64_t inc = block_incl_scan_i64(acc, s_wave, &total);
    int64_t run       = chunk_bases[blockIdx.x] + inc - acc;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) {
            if (INCLUSIVE) out[base + i] = (OutT)(run + v[i]);
            else out[base + i] = (OutT)run;
        }
        run += v[i];
    }
}

static int64_t scan_ws_bytes(int64_t n) { return (ceil_div(n > 0 ? n : 1, kScanChunk) + 1) * (int64_t)sizeof(int64_t); }

template <typename OutT, bool INCLUSIVE>
static int run_scan(const int32_t *in, int64_t n, OutT *out, void *ws, int64_t ws_bytes, hipStream_t stream)
{
    if (n <= 0) return GSX_OK;
    if (ws_bytes < scan_ws_bytes(n) || ws == nullptr) {
        set_last_error("scan: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)scan_ws_bytes(n));
        return GSX_ERR_WORKSPACE;
    }
    const int64_t n_chunks = ceil_div(n, kScanChunk);
    int64_t *chunk_sums    = reinterpret_cast<int64_t *>(ws);
    scan_reduce_kernel<<<dim3((uint32_t)n_chunks), dim3(kScanThreads), 0, stream>>>(in, n, chunk_sums);
    scan_chunks_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(chunk_sums, n_chunks);
    scan_apply_kernel<OutT, INCLUSIVE><<<dim3((uint32_t)n_chunks), dim3(kScanThreads), 0, stream>>>(in, n, chunk_sums, out);
    return check_launch("scan");
}

// entry points for the other translation units (tile_sort.hip)
int run_scan_i32_exclusive(const int32_t *in, int64_t n, int32_t *out, void *ws, int64_t ws_bytes, hipStream_t stream)
{
    return run_scan<int32_t, false>(in, n, out, ws, ws_bytes, stream);
}
int64_t scan_workspace_bytes_for(int64_t n) { return scan_ws_bytes(n); }

// ------------------------------------------------------------------------------------------
// radix sort of (int64 key, int32 value) pairs, 8 bits per pass.
// ------------------------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortRounds  = 16;                          // keys per lane
constexpr int kSortWaveRun = 64 * kSortRounds;            // 1024 consecutive keys per wave
constexpr int kSortChunk   = kSortThreads * kSortRounds;  // 4096 keys per workgroup
constexpr int kRadix       = 256;

// per-workgroup digit histogram, stored digit-major: hist[d * n_chunks + chunk]
__global__ void __launch_bounds__(kSortThreads)
sort_hist_kernel(const uint64_t *keys, int64_t n, int shift, int64_t n_chunks, int32_t *hist)
{
    __shared__ int32_t s_hist[kRadix];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortChunk;
#pragma unroll
    for (int i = 0; i < kSortRounds; ++i) {
        const int64_t idx = base + (int64_t)i * kSortThreads + threadIdx.x;
        if (idx < n) atomicAdd(&s_hist[(int)((keys[idx] >> shift) & 0xFFu)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_chunks + blockIdx.x] = s_hist[threadIdx.x];
}

__global__ void __launch_bounds__(kSortThreads)
sort_scatter_kernel(const uint64_t *keys_in, const int32_t *vals_in, uint64_t *keys_out, int32_t *vals_out,
                    int64_t n, int shift, int64_t n_chunks, const int32_t *hist_scanned /* exclusive, digit-major */)
{
    __shared__ int32_t s_cnt[4][kRadix];  // per-wave running digit counts
    __shared__ int32_t s_base[4][kRadix]; // global base + exclusive prefix over waves

    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    for (int i = threadIdx.x; i < 4 * kRadix; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    const int64_t wave_base = (int64_t)blockIdx.x * kSortChunk + (int64_t)wave * kSortWaveRun;
    uint64_t key[kSortRounds];
    uint16_t rank[kSortRounds];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t idx = wave_base + (int64_t)r * 64 + lane;
        const bool live   = idx < n;
        key[r]            = live ? keys_in[idx] : ~0ull;
        const int d       = (int)((key[r] >> shift) & 0xFFu);
        // lanes of this wave holding the same digit (dead lanes match nobody that is live)
        uint64_t peers = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool b       = (d >> bit) & 1;
            const uint64_t m   = __builtin_amdgcn_ballot_w64(b);
            peers &= b ? m : ~m;
        }
        // wave-synchronous LDS counter: every lane reads, then the first peer updates. volatile keeps
        // the compiler from forwarding a stale value between rounds (another LANE may have written it).
        volatile int32_t *cnt = &s_cnt[wave][0];
        const int before = live ? cnt[d] : 0;
        const int pos    = __popcll(peers & lt_mask);
        rank[r]          = (uint16_t)(before + pos);
        if (live && pos == 0) cnt[d] = before + __popcll(peers);
    }
    __syncthreads();

    // thread d: exclusive prefix of digit d over the four waves, on top of the global base
    {
        const int d  = (int)threadIdx.x;
        int32_t run  = hist_scanned[(int64_t)d * n_chunks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s_base[w][d] = run;
            run += s_cnt[w][d];
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t idx = wave_base + (int64_t)r * 64 + lane;
        if (idx < n) {
            const int d       = (int)((key[r] >> shift) & 0xFFu);
            const int64_t dst = (int64_t)s_base[wave][d] + rank[r];
            keys_out[dst]     = key[r];
            vals_out[dst]     = vals_in[idx];
        }
    }
}

static int64_t sort_ws_bytes(int64_t n)
{
    const int64_t n_chunks = ceil_div(n > 0 ? n : 1, kSortChunk);
    const int64_t hist     = kRadix * n_chunks * (int64_t)sizeof(int32_t);
    return 2 * hist + scan_ws_bytes(kRadix * n_chunks) + 256;
}

} // namespace gsx

extern "C" int64_t gsx_scan_workspace_bytes(int64_t n) { return gsx::scan_ws_bytes(n); }

extern "C" int gsx_scan_i32(const int32_t *in, int64_t n, int64_t *out_inclusive, void *workspace,
                            int64_t workspace_bytes, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(n >= 0, "gsx_scan_i32: negative length");
    GSX_REQUIRE(n == 0 || (in && out_inclusive), "gsx_scan_i32: null pointer");
    return run_scan<int64_t, true>(in, n, out_inclusive, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int64_t gsx_sort_pairs_workspace_bytes(int64_t n) { return gsx::sort_ws_bytes(n); }

extern "C" int gsx_sort_pairs(int64_t *keys, int32_t *vals, int64_t *keys_alt, int32_t *vals_alt, int64_t n,
                              int end_bit, void *workspace, int64_t workspace_bytes, int *result_in_alt, void *stream)
{
    using namespace gsx;
    GSX_REQUIRE(n >= 0 && n < (1ll << 31), "gsx_sort_pairs: n out of range");
    GSX_REQUIRE(end_bit >= 0 && end_bit <= 64, "gsx_sort_pairs: end_bit out of range");
    GSX_REQUIRE(result_in_alt != nullptr, "gsx_sort_pairs: null result_in_alt");
    *result_in_alt = 0;
    if (n == 0 || end_bit == 0) return GSX_OK;
    GSX_REQUIRE(keys && vals && keys_alt && vals_alt, "gsx_sort_pairs: null buffer");
    if (workspace == nullptr || workspace_bytes < sort_ws_bytes(n)) {
        set_last_error("gsx_sort_pairs: workspace too small");
        return GSX_ERR_WORKSPACE;
    }
    hipStream_t s          = (hipStream_t)stream;
    const int64_t n_chunks = ceil_div(n, kSortChunk);
    const int64_t n_hist   = kRadix * n_chunks;
    int32_t *hist          = reinterpret_cast<int32_t *>(workspace);
    int32_t *hist_scanned  = hist + n_hist;
    void *scan_ws          = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(hist_scanned + n_hist) + 255) & ~(uintptr_t)255);
    const int64_t scan_ws_b = scan_ws_bytes(n_hist);

    uint64_t *k_in = reinterpret_cast<uint64_t *>(keys), *k_out = reinterpret_cast<uint64_t *>(keys_alt);
    int32_t *v_in = vals, *v_out = vals_alt;
    const int passes = (end_bit + 7) / 8;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        sort_hist_kernel<<<dim3((uint32_t)n_chunks), dim3(kSortThreads), 0, s>>>(k_in, n, shift, n_chunks, hist);
        int rc = run_scan<int32_t, false>(hist, n_hist, hist_scanned, scan_ws, scan_ws_b, s);
        if (rc != GSX_OK) return rc;
        sort_scatter_kernel<<<dim3((uint32_t)n_chunks), dim3(kSortThreads), 0, s>>>(k_in, v_in, k_out, v_out, n, shift,
                                                                                  n_chunks, hist_scanned);
        rc = check_launch("sort_scatter");
        if (rc != GSX_OK) return rc;
        uint64_t *tk = k_in; k_in = k_out; k_out = tk;
        int32_t *tv = v_in; v_in = v_out; v_out = tv;
    }
    *result_in_alt = passes & 1;
    return GSX_OK;
}
