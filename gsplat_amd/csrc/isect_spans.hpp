// A row's tile spans as ONE 16-byte record: what the counting pass of the Gaussian-major intersection keeps of its walk so
// that the emission pass does not repeat the arithmetic (isect_fused.hip; host check: tools/check_spans.cpp).
#pragma once
#include "isect_walk.hpp"

namespace gsx {

struct alignas(16) SpanRecord { // one 16-byte load / store on the device
    uint32_t x, y, z, w;
};
static_assert(sizeof(SpanRecord) == 16, "the workspace is sized for 16-byte records (tile_sort.hip: kSpanRecordBytes)");

// x: [11:0] first slab u0, [23:12] first span's start v_first, [24] alongY, [31:28] slabs n (0 = no tile, kSpanWalk = the row
// does not fit: the emission walks it again); y, z, w: six 16-bit (int8 start - v_first | uint8 length << 8), slab k is u0 + k.
constexpr uint32_t kSpanSlabs = 6, kSpanWalk = 15;
struct SpanPacker {
    uint32_t hdr = 0, w[3] = {0u, 0u, 0u};
    int n = 0, v_first = 0;
    bool walk = false;
    __host__ __device__ __forceinline__ void add(bool alongY, int u, int tv0, int tv1)
    {
        const int len = tv1 > tv0 ? tv1 - tv0 : 0;
        if (n == 0) {
            v_first = tv0;
            hdr     = (uint32_t)u | ((uint32_t)tv0 << 12) | (alongY ? 1u << 24 : 0u);
            walk |= u > 4095 || tv0 > 4095;
        }
        const int rel = len ? tv0 - v_first : 0;
        walk |= n >= (int)kSpanSlabs || rel < -128 || rel > 127 || len > 255;
        if (!walk) w[n >> 1] |= (((uint32_t)rel & 0xFFu) | ((uint32_t)len << 8)) << (16 * (n & 1));
        ++n;
    }
    __host__ __device__ __forceinline__ SpanRecord record() const
    {
        return SpanRecord{hdr | ((walk ? kSpanWalk : (uint32_t)n) << 28), w[0], w[1], w[2]};
    }
};
__host__ __device__ __forceinline__ uint32_t span_slabs(const SpanRecord &rec) { return rec.x >> 28; }
__host__ __device__ __forceinline__ int span_tiles(const SpanRecord &rec) // tiles of a recorded row (kSpanWalk rows: unknown -> "many")
{
    const uint32_t n = span_slabs(rec);
    if (n == kSpanWalk) return 1 << 20;
    return (int)(((rec.y >> 8) & 0xFFu) + (rec.y >> 24) + ((rec.z >> 8) & 0xFFu) + (rec.z >> 24) + ((rec.w >> 8) & 0xFFu) + (rec.w >> 24));
}
// visits the tiles of a recorded row in the order of the walk
template <typename Emit>
__host__ __device__ __forceinline__ void span_tiles_visit(const SpanRecord &rec, uint32_t tile_w, Emit &&emit)
{
    const uint32_t n = span_slabs(rec);
    const int u0 = (int)(rec.x & 0xFFFu), v_first = (int)((rec.x >> 12) & 0xFFFu);
    const bool alongY = (rec.x >> 24) & 1u;
    const uint64_t lo64 = ((uint64_t)rec.z << 32) | rec.y;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t e = k < 4 ? (uint32_t)(lo64 >> (16 * k)) & 0xFFFFu : (rec.w >> (16 * (k - 4))) & 0xFFFFu;
        const int v0 = v_first + (int)(int8_t)(e & 0xFFu), len = (int)(e >> 8), u = u0 + (int)k;
        for (int v = v0; v < v0 + len; ++v) emit(alongY ? (int64_t)u * tile_w + v : (int64_t)v * tile_w + u);
    }
}


} // namespace gsx
