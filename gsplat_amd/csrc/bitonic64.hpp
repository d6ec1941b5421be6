// Bitonic sort of 64-bit (depth bits << 32 | row) words in LDS: the per-tile depth sort of the intersection paths
// (isect_binned.hip: one wave per tile; tile_sort.hip: one workgroup per tile). Result = ascending unsigned order of the
// words, which is the reference's stable radix sort on (image | tile | depth) with ties in emission order
// (gsplat/cuda/csrc/IntersectTile.cu:1078-1121).
//
// Two networks over the same layout:
//   *_int   compare-exchange on the words as unsigned integers. On gfx950 that is v_cmp_lt_u64 + v_cmp_gt_u64 (the compiler
//           turns the two selects into umin / umax and expands each) + 4 v_cndmask per exchange: six instructions of the
//           3.6-cycle class (profiles/issue_rate_f64.json: v_cmp_gt_u64 0.56 / ns / SIMD, like every compare).
//   *_f64   the same words read as IEEE doubles: v_min_f64 + v_max_f64 per exchange - two instructions of the same class
//           (v_min_f64 0.55 / ns / SIMD). A positive normal double
//           orders exactly like its bit pattern, min / max return one operand unchanged, and a descending block is the
//           ascending network on the NEGATED words (one v_xor on the high dword when a word is loaded and stored, where the
//           integer network complements both dwords). Valid when every high dword (the depth's float bits) lies in
//           [0x00100000, 0x7FF00000): a positive float >= 1.5e-39 that is not a NaN. The callers look at the keys while they
//           stage them (bt_key_is_odd) and send a list with any other depth - negative, zero, denormal, NaN - through *_int.
//           Pads: +inf (kBtPadF64), which sorts behind every such key; ~0 (kBtPadInt) for the integer network.
//
// Layout: word i lives at bt_phys(i) = i + i / 8 (one pad word per eight spreads the 8-word-strided accesses over the banks).
// Phases k = 2, 4, 8 run in registers on 8 consecutive words; every later phase is cut into groups of three strides for which
// a thread owns all 8 words: 512 words in 16 LDS round trips instead of 45. NT threads with rank `tid` share the array; SYNC()
// orders the LDS traffic between round trips (a workgroup barrier, or a compiler fence when one wave owns the array).
#pragma once
#include "common.hpp"

namespace gsx {

constexpr uint64_t kBtPadInt = ~0ull;
constexpr uint64_t kBtPadF64 = 0x7FF0000000000000ull;

__device__ __forceinline__ int bt_phys(int i) { return i + (i >> 3); }
// word offset of element (i | b << lj) from element i when bits [lj, lj + 3) of i are zero: uniform over the wave
__device__ __forceinline__ int bt_off(int b, int lj)
{
    const int sb = b << lj;
    return sb + (sb >> 3);
}
__device__ __forceinline__ bool bt_key_is_odd(uint32_t depth_bits)
{
    return (depth_bits - 0x00100000u) >= (0x7FF00000u - 0x00100000u);
}

// ---- integer network -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bt_cmpx_int(uint64_t &x, uint64_t &y, bool up)
{
    const bool sw     = (x > y) == up;
    const uint64_t lo = sw ? y : x, hi = sw ? x : y;
    x = lo;
    y = hi;
}
// ascending compare-exchange: the select masks come straight from the vector compare (a mask that passes through the scalar
// unit first - e.g. xor-ed with a per-lane direction flag - stalls every dependent v_cndmask on gfx950)
__device__ __forceinline__ void bt_cmpx_int_up(uint64_t &x, uint64_t &y)
{
    const bool sw    = x > y;
    const uint64_t t = sw ? y : x;
    y                = sw ? x : y;
    x                = t;
}

template <int G, int NT>
__device__ __forceinline__ void bt_group_int(uint64_t *s, int P, int lk, int lj, int tid)
{
    constexpr int R = 1 << G;
    for (int t = tid; t < (P >> G); t += NT) {
        const int i = ((t >> lj) << (lj + G)) | (t & ((1 << lj) - 1));
        // a descending block is an ascending one on the complemented words: m = all ones where (i & k) != 0
        const uint32_t m32 = 0u - (((uint32_t)i >> lk) & 1u);
        const uint64_t m   = ((uint64_t)m32 << 32) | m32;
        uint64_t *p        = s + bt_phys(i);
        uint64_t e[R];
#pragma unroll
        for (int b = 0; b < R; ++b) e[b] = p[bt_off(b, lj)] ^ m;
#pragma unroll
        for (int q = G - 1; q >= 0; --q)
#pragma unroll
            for (int b = 0; b < R; ++b)
                if (!(b & (1 << q))) bt_cmpx_int_up(e[b], e[b | (1 << q)]);
#pragma unroll
        for (int b = 0; b < R; ++b) p[bt_off(b, lj)] = e[b] ^ m;
    }
}

template <int NT, typename Sync>
__device__ __forceinline__ void bt_sort_int(uint64_t *s, int lp, int tid, Sync &&SYNC)
{
    const int P = 1 << lp;
    for (int t = tid; t < (P >> 3); t += NT) {
        uint64_t *p = s + 9 * t; // bt_phys(8 t + b) = 9 t + b
        uint64_t e[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) e[b] = p[b];
#pragma unroll
        for (int lk = 1; lk <= 3; ++lk)
#pragma unroll
            for (int q = lk - 1; q >= 0; --q)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (!(b & (1 << q))) {
                        const bool up = lk < 3 ? ((b & (1 << lk)) == 0) : ((t & 1) == 0); // ((8 t + b) & k) == 0
                        bt_cmpx_int(e[b], e[b | (1 << q)], up);
                    }
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = e[b];
    }
    SYNC();
    for (int lk = 4; lk <= lp; ++lk) {
        for (int top = lk - 1; top >= 0;) { // log2 of the largest stride still to do in this phase
            const int gsz = top + 1 < 3 ? top + 1 : 3;
            const int lj  = top - gsz + 1;
            if (gsz == 3) bt_group_int<3, NT>(s, P, lk, lj, tid);
            else if (gsz == 2) bt_group_int<2, NT>(s, P, lk, lj, tid);
            else bt_group_int<1, NT>(s, P, lk, lj, tid);
            SYNC();
            top -= gsz;
        }
    }
}

// ---- the same network on the words as doubles ---------------------------------------------------------------------------------
// x <- min, y <- max. Inline asm: the builtins go through llvm.minnum / maxnum, which in IEEE mode canonicalise every operand
// the compiler cannot prove quiet (a v_max_f64 x, x per loaded word).
__device__ __forceinline__ void bt_minmax(double &x, double &y)
{
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    x = lo;
    y = hi;
}
__device__ __forceinline__ double bt_as_f64(uint64_t w) { return __builtin_bit_cast(double, w); }
__device__ __forceinline__ uint64_t bt_as_u64(double d) { return __builtin_bit_cast(uint64_t, d); }

template <int G, int NT>
__device__ __forceinline__ void bt_group_f64(uint64_t *s, int P, int lk, int lj, int tid)
{
    constexpr int R = 1 << G;
    for (int t = tid; t < (P >> G); t += NT) {
        const int i = ((t >> lj) << (lj + G)) | (t & ((1 << lj) - 1));
        // a descending block is an ascending one on the negated words
        const uint64_t neg = (uint64_t)(((uint32_t)i >> lk) & 1u) << 63;
        uint64_t *p        = s + bt_phys(i);
        double e[R];
#pragma unroll
        for (int b = 0; b < R; ++b) e[b] = bt_as_f64(p[bt_off(b, lj)] ^ neg);
#pragma unroll
        for (int q = G - 1; q >= 0; --q)
#pragma unroll
            for (int b = 0; b < R; ++b)
                if (!(b & (1 << q))) bt_minmax(e[b], e[b | (1 << q)]);
#pragma unroll
        for (int b = 0; b < R; ++b) p[bt_off(b, lj)] = bt_as_u64(e[b]) ^ neg;
    }
}

template <int NT, typename Sync>
__device__ __forceinline__ void bt_sort_f64(uint64_t *s, int lp, int tid, Sync &&SYNC)
{
    const int P = 1 << lp;
    for (int t = tid; t < (P >> 3); t += NT) {
        uint64_t *p = s + 9 * t;
        double e[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) e[b] = bt_as_f64(p[b]);
        // k = 2, 4: the direction of element 8 t + b depends on b alone - the operands swap places at compile time
#pragma unroll
        for (int lk = 1; lk <= 2; ++lk)
#pragma unroll
            for (int q = lk - 1; q >= 0; --q)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (!(b & (1 << q))) {
                        if ((b & (1 << lk)) == 0) bt_minmax(e[b], e[b | (1 << q)]);
                        else bt_minmax(e[b | (1 << q)], e[b]);
                    }
        // k = 8: ascending for even t, descending (= ascending on the negated words) for odd t
        const uint64_t neg = (uint64_t)((uint32_t)t & 1u) << 63;
#pragma unroll
        for (int b = 0; b < 8; ++b) e[b] = bt_as_f64(bt_as_u64(e[b]) ^ neg);
#pragma unroll
        for (int q = 2; q >= 0; --q)
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (!(b & (1 << q))) bt_minmax(e[b], e[b | (1 << q)]);
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = bt_as_u64(e[b]) ^ neg;
    }
    SYNC();
    for (int lk = 4; lk <= lp; ++lk) {
        for (int top = lk - 1; top >= 0;) {
            const int gsz = top + 1 < 3 ? top + 1 : 3;
            const int lj  = top - gsz + 1;
            if (gsz == 3) bt_group_f64<3, NT>(s, P, lk, lj, tid);
            else if (gsz == 2) bt_group_f64<2, NT>(s, P, lk, lj, tid);
            else bt_group_f64<1, NT>(s, P, lk, lj, tid);
            SYNC();
            top -= gsz;
        }
    }
}

// GSX_ISECT_SORT=int keeps every list on the integer network (A/B, tests of the fallback)
bool bitonic_f64_enabled();

} // namespace gsx
