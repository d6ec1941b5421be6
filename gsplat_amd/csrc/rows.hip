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
constexpr uint32_t kMaxSegments = 64; // source ranks of one exchange

struct ColumnGroups {
    const uint32_t *src[kMaxGroups];
    uint32_t *dst[kMaxGroups];
    uint32_t src_stride[kMaxGroups], dst_stride[kMaxGroups], width[kMaxGroups];
    uint32_t first[kMaxGroups + 1]; // prefix sum of the widths: word w of a row belongs to group k iff first[k] <= w < first[k+1]
    uint32_t n_groups;
    int64_t rows;
    // optional row map (n_seg > 0): the rows of ONE side are [camera][all Gaussians] while the other side holds them source
    // rank by source rank, each source's block camera-major: row r of segment k (seg_start[k] <= r < seg_start[k+1]) is
    // camera c = (r - seg_start[k]) / seg_n[k], Gaussian n = (r - seg_start[k]) % seg_n[k] of that source and sits at row
    // c * total_n + seg_off[k] + n on the mapped side (map_dst: the destination is the mapped side, else the source).
    uint32_t n_seg, map_dst;
    int64_t total_n;
    int64_t seg_start[kMaxSegments + 1];
    int64_t seg_n[kMaxSegments], seg_off[kMaxSegments];
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
    if (a.n_seg == 0) {
        for (int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> LOG); row < a.rows;
             row += (int64_t)gridDim.x * kRowsPerBlock)
            dst[row * ds] = src[row * ss];
        return;
    }
    uint32_t sk = 0; // segments are visited in order by a grid-stride walk: the search resumes where it stopped
    for (int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> LOG); row < a.rows;
         row += (int64_t)gridDim.x * kRowsPerBlock) {
        while (sk + 1 < a.n_seg && row >= a.seg_start[sk + 1]) ++sk;
        const int64_t local = row - a.seg_start[sk], n = a.seg_n[sk];
        const int64_t c = local / n, mapped = c * a.total_n + a.seg_off[sk] + (local - c * n);
        if (a.map_dst) dst[mapped * ds] = src[row * ss];
        else dst[row * ds] = src[mapped * ss];
    }
}

template <int LOG>
static void launch_copy(const ColumnGroups &a, hipStream_t s)
{
    const int64_t blocks = ceil_div(a.rows, (int64_t)(256 >> LOG));
    const uint32_t grid  = (uint32_t)(blocks < 256 * 16 ? blocks : 256 * 16);
    copy_column_groups_kernel<LOG><<<dim3(grid), dim3(256), 0, s>>>(a);
}

// ---- message <-> field arrays through LDS --------------------------------------------------------------------------------
// The kernel above gives every word of a row its own lane: the array-of-structures side is read in full cache lines, but the
// field arrays receive 8- to 48-byte pieces (4 rows per wave). When one side is a MESSAGE - a contiguous [rows][stride] array,
// which is what the exchange sends and receives - a workgroup moves 256 message rows as ONE contiguous block through LDS and
// touches every field array with consecutive lanes on consecutive words: both sides coalesced (c4, 16 M rows x 9 words:
// 640 -> ~250 us per launch). The optional row map applies to the field side (see ColumnGroups).
constexpr int kMsgTileRows = 256;
constexpr uint32_t kMsgMaxStride = 16; // words per message row (9 for geometry + radii, 7 for its gradient, D for features)

struct MessageCopy {
    uint32_t *msg;
    uint32_t *field[kMaxGroups];
    uint32_t msg_stride, n_groups, to_msg;
    uint32_t col[kMaxGroups], width[kMaxGroups], field_stride[kMaxGroups];
    int64_t rows;
    uint32_t n_seg;
    int64_t total_n;
    int64_t seg_start[kMaxSegments + 1];
    int64_t seg_n[kMaxSegments], seg_off[kMaxSegments];
};

__device__ __forceinline__ int64_t mapped_row(const MessageCopy &a, int64_t row)
{
    uint32_t sk = 0;
    while (sk + 1 < a.n_seg && row >= a.seg_start[sk + 1]) ++sk;
    const int64_t local = row - a.seg_start[sk], n = a.seg_n[sk];
    const int64_t c = local / n;
    return c * a.total_n + a.seg_off[sk] + (local - c * n);
}

__global__ void __launch_bounds__(256) copy_message_kernel(const MessageCopy a)
{
    __shared__ uint32_t s_tile[kMsgTileRows * kMsgMaxStride];
    __shared__ int64_t s_row[kMsgTileRows]; // field-side row of every message row of the tile (one 64-bit division per ROW)
    const int64_t row0 = (int64_t)blockIdx.x * kMsgTileRows;
    const int n_rows   = (int)min((int64_t)kMsgTileRows, a.rows - row0);
    const int n_words  = n_rows * (int)a.msg_stride;
    uint32_t *tile_g   = a.msg + row0 * a.msg_stride;
    if ((int)threadIdx.x < n_rows) s_row[threadIdx.x] = a.n_seg ? mapped_row(a, row0 + threadIdx.x) : row0 + threadIdx.x;
    if (!a.to_msg)
        for (int i = (int)threadIdx.x; i < n_words; i += 256) s_tile[i] = tile_g[i];
    __syncthreads();
    for (uint32_t k = 0; k < a.n_groups; ++k) {
        const uint32_t w = a.width[k], col = a.col[k];
        // i / w for i < 4096, w <= 16 by one multiplication: the error of ceil(2^20 / w) stays below 0.004 < 1 / w
        const uint32_t recip = ((1u << 20) + w - 1u) / w;
        const int64_t fs = a.field_stride[k];
        uint32_t *f = a.field[k];
        for (uint32_t i = threadIdx.x; i < (uint32_t)n_rows * w; i += 256) {
            const uint32_t r = (i * recip) >> 20, j = i - r * w;
            const int64_t m = s_row[r];
            if (a.to_msg) s_tile[r * a.msg_stride + col + j] = f[m * fs + j];
            else f[m * fs + j] = s_tile[r * a.msg_stride + col + j];
        }
    }
    if (a.to_msg) {
        __syncthreads();
        for (int i = (int)threadIdx.x; i < n_words; i += 256) tile_g[i] = s_tile[i];
    }
}

} // namespace gsx

using namespace gsx;

static int copy_groups(uint32_t n_groups, const void *const *src, const uint32_t *src_strides, void *const *dst,
                       const uint32_t *dst_strides, const uint32_t *widths, int64_t rows, uint32_t n_seg,
                       const int64_t *seg_rows, const int64_t *seg_n, int map_dst, void *stream);

extern "C" int gsx_copy_column_groups(uint32_t n_groups, const void *const *src, const uint32_t *src_strides,
                                      void *const *dst, const uint32_t *dst_strides, const uint32_t *widths, int64_t rows,
                                      void *stream)
{
    return copy_groups(n_groups, src, src_strides, dst, dst_strides, widths, rows, 0, nullptr, nullptr, 0, stream);
}

extern "C" int gsx_copy_column_groups_mapped(uint32_t n_groups, const void *const *src, const uint32_t *src_strides,
                                             void *const *dst, const uint32_t *dst_strides, const uint32_t *widths,
                                             uint32_t n_segments, const int64_t *seg_cameras_times_n, const int64_t *seg_n,
                                             int map_dst, void *stream)
{
    GSX_REQUIRE(n_segments >= 1 && n_segments <= kMaxSegments, "gsx_copy_column_groups_mapped: n_segments must be in [1,%u], got %u",
                kMaxSegments, n_segments);
    GSX_REQUIRE(seg_cameras_times_n && seg_n, "gsx_copy_column_groups_mapped: null segment table");
    int64_t rows = 0;
    for (uint32_t k = 0; k < n_segments; ++k) {
        GSX_REQUIRE(seg_n[k] >= 0 && seg_cameras_times_n[k] >= 0 && (seg_n[k] == 0 ? seg_cameras_times_n[k] == 0
                                                                                   : seg_cameras_times_n[k] % seg_n[k] == 0),
                    "gsx_copy_column_groups_mapped: segment %u: %lld rows are not a whole number of blocks of %lld", k,
                    (long long)seg_cameras_times_n[k], (long long)seg_n[k]);
        rows += seg_cameras_times_n[k];
    }
    return copy_groups(n_groups, src, src_strides, dst, dst_strides, widths, rows, n_segments, seg_cameras_times_n, seg_n, map_dst,
                       stream);
}

static int copy_groups(uint32_t n_groups, const void *const *src, const uint32_t *src_strides, void *const *dst,
                       const uint32_t *dst_strides, const uint32_t *widths, int64_t rows, uint32_t n_seg,
                       const int64_t *seg_rows, const int64_t *seg_n, int map_dst, void *stream)
{
    GSX_REQUIRE(n_groups >= 1 && n_groups <= kMaxGroups, "gsx_copy_column_groups: n_groups must be in [1,%u], got %u",
                kMaxGroups, n_groups);
    GSX_REQUIRE(src && src_strides && dst && dst_strides && widths, "gsx_copy_column_groups: null table");
    GSX_REQUIRE(rows >= 0, "gsx_copy_column_groups: negative row count");
    ColumnGroups a{};
    a.n_groups = n_groups; a.rows = rows;
    a.n_seg = 0; a.map_dst = map_dst ? 1u : 0u;
    for (uint32_t k = 0, kk = 0; k < n_seg; ++k) { // empty segments (a rank without Gaussians) are dropped
        a.total_n += seg_n[k];
        if (seg_rows[k] == 0) continue;
        a.seg_start[kk + 1] = a.seg_start[kk] + seg_rows[k];
        a.seg_n[kk] = seg_n[k];
        a.seg_off[kk] = a.total_n - seg_n[k];
        a.n_seg = ++kk;
    }
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

extern "C" int gsx_copy_message_columns(void *message, uint32_t message_stride, int64_t rows, uint32_t n_groups,
                                        const uint32_t *columns, const uint32_t *widths, void *const *fields,
                                        const uint32_t *field_strides, int to_message, uint32_t n_segments,
                                        const int64_t *seg_cameras_times_n, const int64_t *seg_n, void *stream)
{
    GSX_REQUIRE(n_groups >= 1 && n_groups <= kMaxGroups, "gsx_copy_message_columns: n_groups must be in [1,%u], got %u", kMaxGroups,
                n_groups);
    GSX_REQUIRE(message_stride >= 1 && message_stride <= kMsgMaxStride, "gsx_copy_message_columns: %u words per message row (1..%u)",
                message_stride, kMsgMaxStride);
    GSX_REQUIRE(columns && widths && fields && field_strides, "gsx_copy_message_columns: null table");
    GSX_REQUIRE(rows >= 0, "gsx_copy_message_columns: negative row count");
    GSX_REQUIRE(n_segments <= kMaxSegments, "gsx_copy_message_columns: at most %u segments, got %u", kMaxSegments, n_segments);
    GSX_REQUIRE(n_segments == 0 || (seg_cameras_times_n && seg_n), "gsx_copy_message_columns: null segment table");
    MessageCopy a{};
    a.msg = static_cast<uint32_t *>(message); a.msg_stride = message_stride; a.n_groups = n_groups; a.to_msg = to_message ? 1u : 0u;
    a.rows = rows;
    bool covered[kMsgMaxStride] = {};
    for (uint32_t k = 0; k < n_groups; ++k) {
        GSX_REQUIRE(widths[k] >= 1 && columns[k] + widths[k] <= message_stride && field_strides[k] >= widths[k],
                    "gsx_copy_message_columns: group %u: columns [%u, %u) of %u, field stride %u", k, columns[k],
                    columns[k] + widths[k], message_stride, field_strides[k]);
        GSX_REQUIRE(rows == 0 || fields[k], "gsx_copy_message_columns: group %u: null field", k);
        a.col[k] = columns[k]; a.width[k] = widths[k]; a.field[k] = static_cast<uint32_t *>(fields[k]);
        a.field_stride[k] = field_strides[k];
        for (uint32_t j = 0; j < widths[k]; ++j) covered[columns[k] + j] = true;
    }
    if (to_message) // every word of the message tile is written from LDS: the groups must fill it
        for (uint32_t j = 0; j < message_stride; ++j)
            GSX_REQUIRE(covered[j], "gsx_copy_message_columns: column %u of the message is not covered by any group", j);
    int64_t mapped_rows = 0;
    for (uint32_t k = 0, kk = 0; k < n_segments; ++k) {
        GSX_REQUIRE(seg_n[k] >= 0 && seg_cameras_times_n[k] >= 0
                        && (seg_n[k] == 0 ? seg_cameras_times_n[k] == 0 : seg_cameras_times_n[k] % seg_n[k] == 0),
                    "gsx_copy_message_columns: segment %u: %lld rows are not a whole number of blocks of %lld", k,
                    (long long)seg_cameras_times_n[k], (long long)seg_n[k]);
        a.total_n += seg_n[k];
        mapped_rows += seg_cameras_times_n[k];
        if (seg_cameras_times_n[k] == 0) continue;
        a.seg_start[kk + 1] = a.seg_start[kk] + seg_cameras_times_n[k];
        a.seg_n[kk] = seg_n[k];
        a.seg_off[kk] = a.total_n - seg_n[k];
        a.n_seg = ++kk;
    }
    GSX_REQUIRE(n_segments == 0 || mapped_rows == rows, "gsx_copy_message_columns: the segments hold %lld rows, the message %lld",
                (long long)mapped_rows, (long long)rows);
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(message, "gsx_copy_message_columns: null message");
    const int64_t blocks = ceil_div(rows, (int64_t)kMsgTileRows);
    GSX_REQUIRE(blocks <= 0x7fffffffll, "gsx_copy_message_columns: too many rows");
    copy_message_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("copy_message_columns");
}
