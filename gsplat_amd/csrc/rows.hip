// Column-group copies between row-major buffers: the pack / unpack step around the personalised row exchange of the
// Gaussian-sharded multi-GPU path (seam B; the reference assembles its send buffers and splits the received ones with
// at::cat / index / contiguous ops: gsplat/cuda/csrc/DistributedCollectives.cpp:368-453). One launch replaces the
// ~5 torch copy kernels on either side of the all-to-all: pack = several per-row tensors (any row stride, e.g. column
// views of the AoS gradient rows) -> one array-of-structures message; unpack = a received message -> contiguous tensors.
// Pure HBM-bound word copy: every 32-bit word is read once and written once.
#include "common.hpp"
#include "../../include/gsplat_amd.h"

namespace gsx {

constexpr uint32_t kMaxGroups = 8;

struct ColumnGroups {
    const uint32_t *src[kMaxGroups];
    uint32_t *dst[kMaxGroups];
    uint32_t src_stride[kMaxGroups], dst_stride[kMaxGroups], width[kMaxGroups];
    uint32_t first[kMaxGroups + 1]; // prefix sum of the widths: word w of a row belongs to group k iff first[k] <= w < first[k+1]
    uint32_t n_groups;
    int64_t rows;
};

// 2^LOG lanes per row (>= words per row): lane -> (row in block, word), so a thread's group / column are fixed for its
// whole grid-stride loop and no division is needed; neighbouring lanes touch neighbouring words of the AoS side.
template <int LOG>
__global__ void __launch_bounds__(256) copy_column_groups_kernel(const ColumnGroups a)
{
    constexpr uint32_t kLanes = 1u << LOG, kRowsPerBlock = 256u >> LOG;
    const uint32_t w = threadIdx.x & (kLanes - 1u);
    if (w >= a.first[a.n_groups]) return;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t g = 1; g < kMaxGroups; ++g)
        if (g < a.n_groups && w >= a.first[g]) k = g;
    const uint32_t j          = w - a.first[k];
    const uint32_t *src       = a.src[k] + j;
    uint32_t *dst             = a.dst[k] + j;
    const int64_t ss = a.src_stride[k], ds = a.dst_stride[k];
    for (int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> LOG); row < a.rows;
         row += (int64_t)gridDim.x * kRowsPerBlock)
        dst[row * ds] = src[row * ss];
}

template <int LOG>
static void launch_copy(const ColumnGroups &a, hipStream_t s)
{
    const int64_t blocks = ceil_div(a.rows, (int64_t)(256 >> LOG));
    const uint32_t grid  = (uint32_t)(blocks < 256 * 16 ? blocks : 256 * 16);
    copy_column_groups_kernel<LOG><<<dim3(grid), dim3(256), 0, s>>>(a);
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_copy_column_groups(uint32_t n_groups, const void *const *src, const uint32_t *src_strides,
                                      void *const *dst, const uint32_t *dst_strides, const uint32_t *widths, int64_t rows,
                                      void *stream)
{
    GSX_REQUIRE(n_groups >= 1 && n_groups <= kMaxGroups, "gsx_copy_column_groups: n_groups must be in [1,%u], got %u",
                kMaxGroups, n_groups);
    GSX_REQUIRE(src && src_strides && dst && dst_strides && widths, "gsx_copy_column_groups: null table");
    GSX_REQUIRE(rows >= 0, "gsx_copy_column_groups: negative row count");
    ColumnGroups a{};
    a.n_groups = n_groups; a.rows = rows;
    for (uint32_t k = 0; k < n_groups; ++k) {
        GSX_REQUIRE(widths[k] >= 1 && src_strides[k] >= widths[k] && dst_strides[k] >= widths[k],
                    "gsx_copy_column_groups: group %u: width %u, strides %u -> %u", k, widths[k], src_strides[k], dst_strides[k]);
        GSX_REQUIRE(rows == 0 || (src[k] && dst[k]), "gsx_copy_column_groups: group %u: null pointer", k);
        a.src[k] = static_cast<const uint32_t *>(src[k]); a.dst[k] = static_cast<uint32_t *>(dst[k]);
        a.src_stride[k] = src_strides[k]; a.dst_stride[k] = dst_strides[k]; a.width[k] = widths[k];
        a.first[k + 1] = a.first[k] + widths[k];
    }
    if (rows == 0) return GSX_OK;
    const uint32_t total = a.first[n_groups];
    GSX_REQUIRE(total <= 256, "gsx_copy_column_groups: %u words per row (at most 256)", total);
    hipStream_t s = (hipStream_t)stream;
    if (total <= 4) launch_copy<2>(a, s);
    else if (total <= 8) launch_copy<3>(a, s);
    else if (total <= 16) launch_copy<4>(a, s);
    else if (total <= 32) launch_copy<5>(a, s);
    else if (total <= 64) launch_copy<6>(a, s);
    else if (total <= 128) launch_copy<7>(a, s);
    else launch_copy<8>(a, s);
    return check_launch("copy_column_groups");
}
