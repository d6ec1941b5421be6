// gsplat_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Everything here is written for wave64 / CDNA4 only. There is no CUDA path,
// no hipify output and no multi-backend dispatch.
//
// Constants restate the reference's contract (values only):
//   gsplat/cuda/include/Common.h:97-114, gsplat/cuda/_constants.py:16-27
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gsplat_amd.h"

namespace gsx {

typedef float v4f __attribute__((ext_vector_type(4))); // true vector types: one ds_read_b128 / b64 per load (a HIP float4 is
typedef float v2f __attribute__((ext_vector_type(2))); // a struct of scalars that the backend re-merges as it sees fit)

// ---- constants of the rasterization contract --------------------------------
constexpr float kAlphaThreshold      = 1.0f / 255.0f; // skip if alpha < this
constexpr float kGaussianExtend      = 3.33f;         // truncation in std-devs
constexpr float kMaxAlpha            = 0.99f;         // alpha clamp
constexpr float kTransmittanceThresh = 1e-4f;         // pixel stops (exclusive) when T' <= this
constexpr float kMinCompensation     = 0.005f;        // floor of sqrt(det/det_blur)
constexpr float kMinOneMinusAlpha    = 1e-6f;         // floor of (1-alpha) in backward
constexpr float kFilterInvSquare2DGS = 2.0f;          // 2DGS low-pass: min(3D kernel, 2*|d|^2)

constexpr int kWave = 64;

// ---- error plumbing for the C-ABI --------------------------------------------
// Every gsx_* entry point returns 0 on success or a negative code; the message is
// retrievable with gsx_last_error() (thread local).
// (codes GSX_OK / GSX_ERR_* come from include/gsplat_amd.h)

void set_last_error(const char *fmt, ...);
int check_launch(const char *what);

// hipFuncSetAttribute applies to the CURRENT device, so "already raised the dynamic-LDS limit" is tracked per device
// (one process per GPU is the normal deployment, but a process that drives several devices must still work).
// Idempotent; racing threads set the same value.
struct PerDeviceOnce {
    bool done[64] = {};
    bool first()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) return true;
        const bool f = !done[dev];
        done[dev]    = true;
        return f;
    }
};

#define GSX_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::gsx::set_last_error(__VA_ARGS__); \
            return GSX_ERR_ARG;         \
        }                                      \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave64 cross-lane primitives (DPP; no LDS traffic) -----------------------
// DPP control words (GCN3/CDNA ISA): quad_perm = 0x00..0xFF, row_shr:n = 0x110+n,
// row_mirror = 0x140, row_half_mirror = 0x141, row_bcast15 = 0x142, row_bcast31 = 0x143.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float x)
{
    // old = 0 so lanes that are masked off / read out of bounds contribute 0.
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false));
}

// x of the lane eight places further on in the same 16-lane row (row_ror:8). Every lane has a source, so `old` never shows:
// passing x itself (instead of dpp_f32's 0) with bound_ctrl set lets the compiler fold the move into v_add_f32_dpp without
// first materialising a zero (one instruction instead of three per value).
__device__ __forceinline__ float dpp_ror8(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x128, 0xF, 0xF, true));
}

// Sum over each row of 16 lanes; every lane of the row ends up with the row sum.
__device__ __forceinline__ float row16_sum(float x)
{
#if defined(GSX_SAFE_REDUCE) && GSX_SAFE_REDUCE
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
#endif
    x += dpp_f32<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);  // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x); // row_half_mirror
    x += dpp_f32<0x140>(x); // row_mirror
    return x;
}

// Sum over the 64 lanes of the wave. The total is returned in every lane (via
// v_readlane of the four row sums, which also makes the value wave-uniform).
__device__ __forceinline__ float wave_sum(float x)
{
    x = row16_sum(x);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return (r0 + r1) + (r2 + r3);
}

// Sum over the 64 lanes; the total is valid in the LAST row only (lanes 48..63). 6 DPP adds, no readlane:
// row sums, then row_bcast15 (lane 15 of the previous row -> rows 1 and 3), then row_bcast31 (lane 31 -> rows 2, 3).
__device__ __forceinline__ float wave_sum_last_row(float x)
{
#if defined(GSX_SAFE_REDUCE) && GSX_SAFE_REDUCE
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    return x;
#else
    x = row16_sum(x);
    // all rows enabled (lets the compiler fuse v_add_f32_dpp): rows without a source read 0 (old = 0); rows 1 and
    // 2 end up with partial sums nobody reads, row 3 = (r2 + r3) + (r0 + r1).
    x += dpp_f32<0x142>(x); // row_bcast15: lane 15 of the previous row
    x += dpp_f32<0x143>(x); // row_bcast31: lane 31 -> rows 2 and 3
    return x;
#endif
}

// Reduce FOUR per-lane values over the wave in one go ("reduce-scatter"): on return every lane of
// 16-lane row r (= lane >> 4) holds the wave-wide sum of value r of (a, b, c, d).
// gfx950 v_permlane16_swap / v_permlane32_swap fold two registers into one per step, so four
// 64-lane reductions cost 3 swaps + 3 adds + 4 DPP adds instead of 4 x 6 DPP adds.
//   permlane16_swap(x, y): odd rows of x <-> even rows of y      (ISA V_PERMLANE16_SWAP_B32)
//   permlane32_swap(x, y): lanes 32-63 of x <-> lanes 0-31 of y  (ISA V_PERMLANE32_SWAP_B32)
// Build with -DGSX_SAFE_REDUCE=1 to get a plain __shfl_xor version (debug / A-B reference).
__device__ __forceinline__ float wave_sum4_scatter(float a, float b, float c, float d)
{
#if defined(GSX_SAFE_REDUCE) && GSX_SAFE_REDUCE
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
        d += __shfl_xor(d, o);
    }
    const int row = (int)(threadIdx.x & 63u) >> 4;
    return row == 0 ? a : (row == 1 ? b : (row == 2 ? c : d));
#else
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float P = __uint_as_float(p[0]) + __uint_as_float(p[1]); // rows: a01, b01, a23, b23
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(d), false, false);
    const float Q = __uint_as_float(q[0]) + __uint_as_float(q[1]); // rows: c01, d01, c23, d23
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
    const float R = __uint_as_float(r[0]) + __uint_as_float(r[1]); // rows: a, b, c, d (per column)
    return row16_sum(R);
#endif
}

// The cross-row half of wave_sum4_scatter: lane (row r, column c) returns the sum over the four 16-lane rows, taken at
// column c, of value r of (a, b, c, d) - i.e. four independent 4-way sums per column in 3 swaps + 3 adds.
__device__ __forceinline__ float rows_sum4_scatter(float a, float b, float c, float d)
{
#if defined(GSX_SAFE_REDUCE) && GSX_SAFE_REDUCE
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    c += __shfl_xor(c, 16); c += __shfl_xor(c, 32);
    d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
    const int row = (int)(threadIdx.x & 63u) >> 4;
    return row == 0 ? a : (row == 1 ? b : (row == 2 ? c : d));
#else
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float P = __uint_as_float(p[0]) + __uint_as_float(p[1]); // rows: a01, b01, a23, b23
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(d), false, false);
    const float Q = __uint_as_float(q[0]) + __uint_as_float(q[1]); // rows: c01, d01, c23, d23
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);          // rows: a, b, c, d (per column)
#endif
}

// Order LDS traffic between the lanes of ONE wave (data handed from lane to lane through a wave-private LDS region).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_max_i32(int x)
{
    // butterfly with DPP inside rows, then readlane across rows
    auto step = [](int v, int o) { return v > o ? v : o; };
#if defined(GSX_SAFE_REDUCE) && GSX_SAFE_REDUCE
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = step(x, __shfl_xor(x, o));
    return x;
#endif
    x = step(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    x = step(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    x = step(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
    x = step(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
    const int r0 = __builtin_amdgcn_readlane(x, 0);
    const int r1 = __builtin_amdgcn_readlane(x, 16);
    const int r2 = __builtin_amdgcn_readlane(x, 32);
    const int r3 = __builtin_amdgcn_readlane(x, 48);
    return step(step(r0, r1), step(r2, r3));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Hardware fp32 atomic add (global_atomic_add_f32, no CAS loop, no return).
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// ---- deterministic natural log -------------------------------------------------
// Used where a float decision feeds INTEGER outputs (tile counts / radii): the
// CPU oracle restates the same sequence of IEEE operations (explicit fmaf, no
// contraction), so the integer results are bit-identical between the two.
// Accuracy ~2 ulp on normal positive inputs, which is at least as tight as the
// fast-math __logf the reference uses at the same places
// (gsplat/cuda/csrc/ProjectionEWA3DGSFused.cu:180, IntersectTile.cu:303).
__host__ __device__ __forceinline__ float det_logf(float x)
{
    // x = m * 2^e, m in [sqrt(1/2), sqrt(2))
    union { float f; uint32_t u; } v;
    v.f        = x;
    int e      = (int)((v.u >> 23) & 0xFF) - 127;
    v.u        = (v.u & 0x007FFFFFu) | 0x3F800000u; // m in [1,2)
    float m    = v.f;
    if (m > 1.41421356f) {
        m *= 0.5f;
        e += 1;
    }
    // log(m) = 2*atanh(s), s = (m-1)/(m+1); |s| <= 0.1716
    const float s  = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p        = 0.2222222222f;              // 2/9
    p              = fmaf(p, s2, 0.2857142857f); // 2/7
    p              = fmaf(p, s2, 0.4f);          // 2/5
    p              = fmaf(p, s2, 0.6666666667f); // 2/3
    p              = fmaf(p, s2, 2.0f);
    const float lm = p * s;
    return fmaf((float)e, 0.69314718056f, lm);
}

// Exclusive scan of n int32 counts by ONE workgroup of 1024 threads, 4096 elements per trip: 16-byte coalesced loads and
// stores (thread t owns elements 4 t .. 4 t + 3 of the trip), a wave scan, 16 wave totals, a running carry. Returns the
// grand total (int64: the callers reject >= 2^31 before any int32 offset is used) and, when `max_out` is set, the largest
// element in the same pass. s_part: 16 int64 of LDS. Replaces thread-contiguous runs read with a 128-byte lane stride
// (c4, 32640 tiles: 71 -> ~12 us; c3: 16 -> ~4 us).
__device__ __forceinline__ int64_t block_scan_i32_1024(const int32_t *in, int32_t *out, uint32_t n, int64_t *s_part,
                                                       int32_t *max_out)
{
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    int64_t carry  = 0;
    int32_t mx     = 0;
    for (uint32_t base = 0; base < n; base += 4096u) {
        const uint32_t i = base + 4u * threadIdx.x;
        int32_t v[4]     = {0, 0, 0, 0};
        if (vec && i + 3u < n) {
            const int4 q = *reinterpret_cast<const int4 *>(in + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (i + k < n) v[k] = in[i + k];
        }
        mx              = max(max(mx, max(v[0], v[1])), max(v[2], v[3]));
        const int64_t s = (int64_t)v[0] + v[1] + v[2] + v[3];
        int64_t inc     = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(inc, o);
            if (lane >= o) inc += y;
        }
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        int64_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int64_t x = s_part[w];
            if (w < wave) before += x;
            all += x;
        }
        int64_t run = carry + before + inc - s;
        if (vec && i + 3u < n) {
            int4 q;
            q.x = (int32_t)run; q.y = (int32_t)(run + v[0]); q.z = (int32_t)(run + v[0] + v[1]);
            q.w = (int32_t)(run + v[0] + v[1] + v[2]);
            *reinterpret_cast<int4 *>(out + i) = q;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (i + k < n) {
                    out[i + k] = (int32_t)run;
                    run += v[k];
                }
        }
        carry += all;
        __syncthreads(); // s_part is rewritten by the next trip
    }
    if (max_out) {
        mx = wave_max_i32(mx);
        if (lane == 0) s_part[wave] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t m = 0;
            for (int w = 0; w < 16; ++w) m = max(m, s_part[w]);
            *max_out = (int32_t)m;
        }
        __syncthreads();
    }
    return carry;
}

} // namespace gsx
