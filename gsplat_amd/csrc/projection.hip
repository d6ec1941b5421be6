// EWA projection of 3D Gaussians (fused quat/scale -> covariance -> camera -> image), gfx950.
//
// C-ABI entries: gsx_project_ewa_{fwd,bwd}, gsx_project_ewa_packed_{count,write,bwd},
//                gsx_quat_scale_to_covar_{fwd,bwd}.
// Replaces gsplat::projection_ewa_3dgs_fused{,_bwd} / _packed{,_bwd} / quat_scale_to_covar_preci{,_bwd}
// (reference kernels gsplat/cuda/csrc/ProjectionEWA3DGSFused.cu:38-219, 378-638,
//  ProjectionEWA3DGSPacked.cu:39-284, 385-684, QuatScaleToCovarCUDA.cu).
//
// Design differences from the reference (MI355X-first, same results):
//   * backward runs ONE THREAD PER GAUSSIAN and loops over the cameras in registers, so
//     v_means / v_quats / v_scales / v_covars are written once, without atomics and
//     deterministically (the reference launches one thread per (camera, gaussian) and merges
//     with warp reductions + atomics). Only v_viewmats needs a cross-Gaussian reduction: wave64
//     reduce-scatter (permlane swaps + DPP) then 12 atomics per wave per camera.
//   * the packed forward is count -> (caller scans) -> write, recomputing the cheap projection
//     instead of staging block counts through the host.
#include "projmath.hpp"

namespace gsx {

struct ProjArgs {
    const float *means, *covars, *quats, *scales, *opacities, *viewmats, *Ks;
    uint32_t B, C, N, width, height;
    float eps2d, near_plane, far_plane, radius_clip;
    int camera_model;
    int calc_compensations;
    // dense outputs / packed outputs
    int32_t *radii;
    float *means2d, *depths, *conics, *compensations;
    // packed
    int32_t *visible;
    const int64_t *row_offsets;
    int64_t nnz;
    int64_t *batch_ids, *camera_ids, *gaussian_ids;
    int32_t *indptr;
};

struct ProjOut {
    bool ok;
    int rx, ry;
    float mx, my, depth, ca, cb, cc, comp;
};

// world covariance (3x3 full) of Gaussian (b, g)
__device__ __forceinline__ void load_world_covar(const ProjArgs &a, uint32_t b, uint32_t g, float *S)
{
    if (a.covars) {
        const float *c = a.covars + ((size_t)b * a.N + g) * 6;
        S[0] = c[0]; S[1] = c[1]; S[2] = c[2];
        S[3] = c[1]; S[4] = c[3]; S[5] = c[4];
        S[6] = c[2]; S[7] = c[4]; S[8] = c[5];
    } else {
        const float *q = a.quats + ((size_t)b * a.N + g) * 4;
        const float *s = a.scales + ((size_t)b * a.N + g) * 3;
        float qn[4], Rq[9];
        quat_normalize(q, qn);
        quat_to_rotmat(qn, Rq);
        quat_scale_to_covar(Rq, s, false, S);
    }
}

// Dense, count and write kernels each inline this body. The file is compiled with -ffp-contract=off (Makefile), so
// every inlined copy performs the same sequence of IEEE operations: packed rows are bit-identical to dense rows and
// the count / write passes always agree on visibility. (A shared __noinline__ body gave the same guarantee but cost
// 208 B of scratch per thread and 3x the run time.)
// `world`: the caller's cache of the Gaussian's world covariance (valid once *have_world is set): a thread that projects one
// Gaussian into several cameras forms it once. One camera per thread: pass a fresh cache.
__device__ __forceinline__ ProjOut project_one(const ProjArgs &a, uint32_t b, uint32_t c, uint32_t g, const float *p,
                                               float *world, bool *have_world)
{
    ProjOut o;
    o.ok = false;
    o.rx = o.ry = 0;
    o.mx = o.my = o.depth = o.ca = o.cb = o.cc = o.comp = 0.0f;
    const Cam cam  = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = cam.R[3 * i] * p[0] + cam.R[3 * i + 1] * p[1] + cam.R[3 * i + 2] * p[2] + cam.t[i];
    if (pc[2] < a.near_plane || pc[2] > a.far_plane) return o;

    float RS[9], Sc[9];
    if (!*have_world) {
        load_world_covar(a, b, g, world);
        *have_world = true;
    }
    const float *S = world;
    mm3(cam.R, S, RS);
    mm3_nt(RS, cam.R, Sc);

    const Proj2D pr = project_camera(a.camera_model, cam, a.width, a.height, pc, Sc);
    const float det_orig = pr.a * pr.d - pr.b * pr.b;
    const float ca = pr.a + a.eps2d, cd = pr.d + a.eps2d;
    const float det_blur = ca * cd - pr.b * pr.b;
    if (det_blur <= 0.0f) return o;
    const float comp = sqrtf(fmaxf(kMinCompensation * kMinCompensation, det_orig / det_blur));

    float extend = kGaussianExtend;
    if (a.opacities) {
        float opacity = a.opacities[(size_t)b * a.N + g];
        if (a.calc_compensations) opacity *= comp;
        if (opacity < kAlphaThreshold) return o;
        extend = fminf(kGaussianExtend, sqrtf(2.0f * det_logf(opacity * 255.0f)));
    }
    const float rx = ceilf(extend * sqrtf(ca)), ry = ceilf(extend * sqrtf(cd));
    if (rx <= a.radius_clip && ry <= a.radius_clip) return o;
    if (pr.mx + rx <= 0.0f || pr.mx - rx >= (float)a.width || pr.my + ry <= 0.0f || pr.my - ry >= (float)a.height) return o;

    const float inv_det = 1.0f / det_blur;
    o.ok = true;
    o.rx = (int)rx; o.ry = (int)ry;
    o.mx = pr.mx; o.my = pr.my; o.depth = pc[2];
    o.ca = cd * inv_det; o.cb = -pr.b * inv_det; o.cc = ca * inv_det;
    o.comp = comp;
    return o;
}

__device__ __forceinline__ ProjOut project_one(const ProjArgs &a, uint32_t b, uint32_t c, uint32_t g)
{
    float world[9];
    bool have_world = false;
    return project_one(a, b, c, g, a.means + ((size_t)b * a.N + g) * 3, world, &have_world);
}

__global__ void __launch_bounds__(256) project_fwd_kernel(const ProjArgs a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx >= count) return;
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    const ProjOut o  = project_one(a, b, c, g);
    a.radii[2 * idx]       = o.rx;
    a.radii[2 * idx + 1]   = o.ry;
    a.means2d[2 * idx]     = o.mx;
    a.means2d[2 * idx + 1] = o.my;
    a.depths[idx]          = o.depth;
    a.conics[3 * idx]      = o.ca;
    a.conics[3 * idx + 1]  = o.cb;
    a.conics[3 * idx + 2]  = o.cc;
    if (a.compensations) a.compensations[idx] = o.comp;
}

// Several cameras over the same Gaussians: one thread per Gaussian walks the cameras - mean, rotation and scale are read,
// and the world covariance formed, once per Gaussian instead of once per (camera, Gaussian). Same arithmetic per row as the
// kernel above (bit-identical outputs); rows of one camera are written by consecutive lanes.
__global__ void __launch_bounds__(256) project_fwd_gaussian_major_kernel(const ProjArgs a)
{
    const int64_t bg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // over B * N
    if (bg >= (int64_t)a.B * a.N) return;
    const uint32_t b = (uint32_t)(bg / a.N), g = (uint32_t)(bg % a.N);
    const float *pm = a.means + (size_t)bg * 3;
    const float p[3] = {pm[0], pm[1], pm[2]};
    float world[9];
    bool have_world = false;
    for (uint32_t c = 0; c < a.C; ++c) {
        const ProjOut o   = project_one(a, b, c, g, p, world, &have_world);
        const int64_t idx = ((int64_t)b * a.C + c) * a.N + g;
        a.radii[2 * idx]       = o.rx;
        a.radii[2 * idx + 1]   = o.ry;
        a.means2d[2 * idx]     = o.mx;
        a.means2d[2 * idx + 1] = o.my;
        a.depths[idx]          = o.depth;
        a.conics[3 * idx]      = o.ca;
        a.conics[3 * idx + 1]  = o.cb;
        a.conics[3 * idx + 2]  = o.cc;
        if (a.compensations) a.compensations[idx] = o.comp;
    }
}

__global__ void __launch_bounds__(256) project_count_kernel(const ProjArgs a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx >= count) return;
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    a.visible[idx]   = project_one(a, b, c, g).ok ? 1 : 0;
}

__global__ void __launch_bounds__(256) project_write_kernel(const ProjArgs a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx > count) return;
    // CSR pointer over images: indptr[i] = #visible rows before image i
    if (idx % a.N == 0) {
        const int64_t img = idx / a.N; // 0..B*C
        a.indptr[img]     = (int32_t)(idx == 0 ? 0 : a.row_offsets[idx - 1]);
    }
    if (idx == count) return;
    const int64_t row  = idx == 0 ? 0 : a.row_offsets[idx - 1]; // row_offsets = INCLUSIVE cumsum of visible
    const int64_t next = a.row_offsets[idx];
    if (next == row) return; // not visible
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    const ProjOut o  = project_one(a, b, c, g);
    a.batch_ids[row]       = b;
    a.camera_ids[row]      = c;
    a.gaussian_ids[row]    = g;
    a.radii[2 * row]       = o.rx;
    a.radii[2 * row + 1]   = o.ry;
    a.means2d[2 * row]     = o.mx;
    a.means2d[2 * row + 1] = o.my;
    a.depths[row]          = o.depth;
    a.conics[3 * row]      = o.ca;
    a.conics[3 * row + 1]  = o.cb;
    a.conics[3 * row + 2]  = o.cc;
    if (a.compensations) a.compensations[row] = o.comp;
}

// ---- packed rows from BLOCK counts -----------------------------------------------------------------------------------------
// The count pass above writes one flag per (image, Gaussian) pair and the caller scans all of them (three launches over
// 4 B x B C N) before the write pass can place a row. A row's place only needs (a) the number of visible pairs in the
// 256-pair blocks before its own - one int32 per block, scanned by ONE workgroup that also publishes the total - and
// (b) its rank inside its block, which the write pass re-derives with a ballot from the visibility it recomputes anyway.
// Blocks cover pairs [256 k, 256 k + 256) of the flattened (b, c, g) order; there is one block more than pairs need when
// B C N is a multiple of 256, so that the one-past-the-end position (indptr's last entry) has a block too.
struct PackedBlocks {
    int32_t *block_counts;        // [n_blocks]      (count pass)
    const int32_t *block_offsets; // [n_blocks]      (write pass: exclusive scan of the counts)
};

__global__ void __launch_bounds__(256) project_count_blocks_kernel(const ProjArgs a, PackedBlocks pb)
{
    __shared__ int32_t s_n[4];
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    bool ok = false;
    if (idx < count) {
        const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
        ok = project_one(a, b, c, g).ok;
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
    if ((threadIdx.x & 63u) == 0) s_n[threadIdx.x >> 6] = (int32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) pb.block_counts[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}

// one workgroup: exclusive scan of the block counts; the total goes to device memory and (optionally) straight into a
// pinned host word the caller polls - no copy kernel, no stream synchronisation
__global__ void __launch_bounds__(1024) packed_block_scan_kernel(const int32_t *counts, uint32_t n_blocks, int32_t *offsets,
                                                                int64_t *nnz_device, int64_t *nnz_host)
{
    __shared__ int64_t s_part[16];
    const int64_t total = block_scan_i32_1024(counts, offsets, n_blocks, s_part, nullptr);
    if (threadIdx.x == 0) {
        if (nnz_device) *nnz_device = total;
        if (nnz_host) {
            __threadfence_system();
            *nnz_host = total;
        }
    }
}

__global__ void __launch_bounds__(256) project_write_blocks_kernel(const ProjArgs a, PackedBlocks pb)
{
    __shared__ int32_t s_n[4];
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t b = 0, c = 0, g = 0;
    ProjOut o;
    o.ok = false;
    if (idx < count) {
        g = (uint32_t)(idx % a.N); c = (uint32_t)((idx / a.N) % a.C); b = (uint32_t)(idx / ((int64_t)a.N * a.C));
        o = project_one(a, b, c, g);
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(o.ok);
    if (lane == 0) s_n[wave] = (int32_t)__popcll(m);
    __syncthreads();
    int32_t before = pb.block_offsets[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) before += s_n[w];
    const int64_t row = (int64_t)before + (int64_t)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    // CSR pointer over images: indptr[i] = #visible rows before image i (the one-past-the-end pair closes it)
    if (idx <= count && idx % a.N == 0) a.indptr[idx / a.N] = (int32_t)row;
    if (!o.ok) return;
    a.batch_ids[row]       = b;
    a.camera_ids[row]      = c;
    a.gaussian_ids[row]    = g;
    a.radii[2 * row]       = o.rx;
    a.radii[2 * row + 1]   = o.ry;
    a.means2d[2 * row]     = o.mx;
    a.means2d[2 * row + 1] = o.my;
    a.depths[row]          = o.depth;
    a.conics[3 * row]      = o.ca;
    a.conics[3 * row + 1]  = o.cb;
    a.conics[3 * row + 2]  = o.cc;
    if (a.compensations) a.compensations[row] = o.comp;
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
struct ProjBwdArgs {
    const float *means, *covars, *quats, *scales, *viewmats, *Ks;
    uint32_t B, C, N, width, height;
    float eps2d;
    int camera_model;
    // dense: per (b,c,g); packed: per row
    const int32_t *radii;
    const float *conics, *compensations;
    const float *v_means2d, *v_depths, *v_conics, *v_compensations;
    uint32_t m2_stride, con_stride; // row strides (floats) of v_means2d / v_conics: 2 / 3 when contiguous, or the
                                    // stride of gsx_raster3d_bwd's AoS gradient rows when they are column views of it
    int64_t nnz;
    const int64_t *batch_ids, *camera_ids, *gaussian_ids;
    const int32_t *row_map; // packed rows walked Gaussian-major: [B*C*N] -> packed row or -1
    int rows_out;           // sparse_grad: the per-Gaussian outputs are [nnz, .] rows (one per packed row), not [B, N, .]
    float *v_means, *v_covars, *v_quats, *v_scales, *v_viewmats;
    // (optional) the cotangent of the per-view opacities [B, C, N] - a column of the compositing backward's gradient rows,
    // opac_stride floats apart - summed over the views into v_opacities [B, N] by the Gaussian-major kernel: it reads those
    // rows anyway, and autograd would otherwise copy the strided column out on its own (36 MB read for 4 MB at c3)
    const float *v_view_opacities;
    uint32_t opac_stride;
    float *v_opacities;
};

// Gradient of one (camera, gaussian) pair wrt world mean (3), world covariance (3x3) and the camera
// pose (v_R 9, v_t 3). `row` indexes the per-pair tensors.
__device__ __forceinline__ void pair_vjp(const ProjBwdArgs &a, const Cam &cam, const float *p, const float *S,
                                         int64_t row, float *v_p, float *v_S, float *v_R, float *v_t, bool want_pose)
{
    const float ca = a.conics[3 * row], cb = a.conics[3 * row + 1], cc = a.conics[3 * row + 2];
    const float *vcon = a.v_conics + (size_t)a.con_stride * row;
    const float va = vcon[0], vb = 0.5f * vcon[1], vc = vcon[2];
    // v_cov2d = -P V P with P = [[ca,cb],[cb,cc]], V = [[va,vb],[vb,vc]]
    const float t00 = va * ca + vb * cb, t01 = va * cb + vb * cc;
    const float t10 = vb * ca + vc * cb, t11 = vb * cb + vc * cc;
    float g00 = -(ca * t00 + cb * t10);
    float g01 = -(ca * t01 + cb * t11);
    float g11 = -(cb * t01 + cc * t11);
    if (a.v_compensations) {
        const float comp = a.compensations[row], v_comp = a.v_compensations[row];
        const float detP = ca * cc - cb * cb;
        const float vsq  = v_comp * 0.5f / (comp + 1e-6f);
        const float omc  = 1.0f - comp * comp;
        g00 += vsq * (omc * ca - a.eps2d * detP);
        g01 += vsq * (omc * cb);
        g11 += vsq * (omc * cc - a.eps2d * detP);
    }
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = cam.R[3 * i] * p[0] + cam.R[3 * i + 1] * p[1] + cam.R[3 * i + 2] * p[2] + cam.t[i];
    float RS[9], Sc[9];
    mm3(cam.R, S, RS);
    mm3_nt(RS, cam.R, Sc);

    float v_pc[3] = {0.f, 0.f, 0.f}, v_Sc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v_Sc[i] = 0.0f;
    const float *vm2 = a.v_means2d + (size_t)a.m2_stride * row;
    project_camera_vjp(a.camera_model, cam, a.width, a.height, pc, Sc, g00, g01, g11, vm2[0], vm2[1], v_pc, v_Sc);
    if (a.v_depths) v_pc[2] += a.v_depths[row]; // null = no gradient reaches the depths

    // world mean: v_p += R^T v_pc
#pragma unroll
    for (int j = 0; j < 3; ++j) v_p[j] += cam.R[j] * v_pc[0] + cam.R[3 + j] * v_pc[1] + cam.R[6 + j] * v_pc[2];
    // world covariance: v_S += R^T v_Sc R
    float tmp[9], add[9];
    mm3_tn(cam.R, v_Sc, tmp);
    mm3(tmp, cam.R, add);
#pragma unroll
    for (int i = 0; i < 9; ++i) v_S[i] += add[i];
    if (want_pose) {
        // v_R += v_pc p^T + v_Sc R S^T + v_Sc^T R S ; v_t += v_pc
        float A1[9], A2[9];
        mm3(v_Sc, RS, A1);                 // v_Sc (R S)   [S symmetric: R S^T = R S]
        float v_ScT[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v_ScT[3 * i + j] = v_Sc[3 * j + i];
        mm3(v_ScT, RS, A2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) v_R[3 * i + j] += v_pc[i] * p[j] + A1[3 * i + j] + A2[3 * i + j];
            v_t[i] += v_pc[i];
        }
    }
}

// write per-Gaussian gradients from (v_p, v_S); `out` = output row (the Gaussian's own row bg, or the packed row when the
// caller asked for [nnz, .] gradient rows)
template <bool ATOMIC>
__device__ __forceinline__ void store_gaussian_grads(const ProjBwdArgs &a, uint32_t b, uint32_t g, size_t out,
                                                     const float *v_p, const float *v_S)
{
    const size_t bg = (size_t)b * a.N + g;
    auto put = [](float *dst, float v) {
        if (ATOMIC) atomic_add_f32(dst, v);
        else *dst = v;
    };
    if (a.v_means)
        for (int i = 0; i < 3; ++i) put(a.v_means + out * 3 + i, v_p[i]);
    if (a.covars) {
        if (a.v_covars) {
            put(a.v_covars + out * 6 + 0, v_S[0]);
            put(a.v_covars + out * 6 + 1, v_S[1] + v_S[3]);
            put(a.v_covars + out * 6 + 2, v_S[2] + v_S[6]);
            put(a.v_covars + out * 6 + 3, v_S[4]);
            put(a.v_covars + out * 6 + 4, v_S[5] + v_S[7]);
            put(a.v_covars + out * 6 + 5, v_S[8]);
        }
    } else {
        const float *q = a.quats + bg * 4;
        const float *s = a.scales + bg * 3;
        float qn[4], Rq[9], v_q[4] = {0.f, 0.f, 0.f, 0.f}, v_s[3] = {0.f, 0.f, 0.f};
        const float inv = quat_normalize(q, qn);
        quat_to_rotmat(qn, Rq);
        quat_scale_to_covar_vjp(qn, inv, Rq, s, v_S, v_q, v_s);
        if (a.v_quats)
            for (int i = 0; i < 4; ++i) put(a.v_quats + out * 4 + i, v_q[i]);
        if (a.v_scales)
            for (int i = 0; i < 3; ++i) put(a.v_scales + out * 3 + i, v_s[i]);
    }
}

// wave-reduce the 12 pose-gradient values and add them to v_viewmats[b,c]
__device__ __forceinline__ void reduce_pose_grads(float *v_viewmat /*[16]*/, const float *v_R, const float *v_t)
{
    const uint32_t lane = threadIdx.x & 63u;
    float mine = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // group j = row j of [R | t]: (R[j][0], R[j][1], R[j][2], t[j])
        const float r = wave_sum4_scatter(v_R[3 * j], v_R[3 * j + 1], v_R[3 * j + 2], v_t[j]);
        if ((int)(lane & 15u) == j) mine = r;
    }
    const int row = (int)(lane & 15u), col = (int)(lane >> 4);
    if (row < 3) atomic_add_f32(v_viewmat + 4 * row + col, mine);
}

// Register budget of the dense backward: left alone the allocator takes 153 VGPRs (3 waves per SIMD); capped at 128 (4 waves,
// 72 bytes of scratch per lane) the kernel is faster - c4 (16 M rows): 380 -> 336 us, c3: 41 -> 38 us (profiles/r05_ab.md #22)
#ifndef GSX_PROJ_BWD_WAVES
#define GSX_PROJ_BWD_WAVES 4
#endif
template <bool POSE>
__global__ void __launch_bounds__(256)
#if GSX_PROJ_BWD_WAVES
__attribute__((amdgpu_waves_per_eu(GSX_PROJ_BWD_WAVES)))
#endif
project_bwd_kernel(const ProjBwdArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // over B*N
    const bool live   = idx < (int64_t)a.B * a.N;
    const uint32_t b = live ? (uint32_t)(idx / a.N) : 0, g = live ? (uint32_t)(idx % a.N) : 0;
    float p[3] = {0.f, 0.f, 0.f}, S[9];
    ProjArgs fa{};
    fa.covars = a.covars; fa.quats = a.quats; fa.scales = a.scales; fa.N = a.N;
    if (live) {
        const float *pm = a.means + ((size_t)b * a.N + g) * 3;
        p[0] = pm[0]; p[1] = pm[1]; p[2] = pm[2];
        load_world_covar(fa, b, g, S);
    }
    float v_p[3] = {0.f, 0.f, 0.f}, v_S[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v_S[i] = 0.0f;

    // block-uniform batch id is not guaranteed (a block may straddle two batches), so pose grads
    // are reduced per wave and the wave handles the (rare) straddle by looping over its batch ids.
    for (uint32_t c = 0; c < a.C; ++c) {
        float v_R[9], v_t[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) v_R[i] = 0.0f;
        if (live) {
            int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            bool visible;
            if (a.row_map) { // packed rows: every stored row is visible
                row     = a.row_map[row];
                visible = row >= 0;
            } else {
                visible = a.radii[2 * row] > 0 && a.radii[2 * row + 1] > 0;
            }
            if (visible) {
                const Cam cam = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
                pair_vjp(a, cam, p, S, row, v_p, v_S, v_R, v_t, POSE);
            }
        }
        if (POSE) {
            // all lanes of a wave usually share b; handle straddling waves batch by batch
            const uint32_t b_lo = __builtin_amdgcn_readfirstlane(b);
            uint32_t b_hi       = b_lo;
            {
                int m = wave_max_i32(live ? (int)b : (int)b_lo);
                b_hi  = (uint32_t)m;
            }
            for (uint32_t bb = b_lo; bb <= b_hi; ++bb) {
                float r[9], t[3];
                const bool sel = live && (b == bb);
#pragma unroll
                for (int i = 0; i < 9; ++i) r[i] = sel ? v_R[i] : 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) t[i] = sel ? v_t[i] : 0.0f;
                reduce_pose_grads(a.v_viewmats + ((size_t)bb * a.C + c) * 16, r, t);
            }
        }
    }
    if (live) store_gaussian_grads<false>(a, b, g, (size_t)b * a.N + g, v_p, v_S);
    // the cotangent of the per-view opacities, summed over the views: a loop of its own AFTER the geometry (inside the loop above
    // the accumulator and the column's address cost the capped register file 28 more bytes of scratch per lane: 36 -> 51 us at
    // c3); the rows' cache lines were read a moment ago. Dense rows: an invisible pair's entry is the exact zero the
    // compositing backward never touched.
    if (live && a.v_opacities) {
        float v_o = 0.0f;
        for (uint32_t c = 0; c < a.C; ++c) {
            int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            if (a.row_map) {
                row = a.row_map[row];
                if (row < 0) continue;
            }
            v_o += a.v_view_opacities[(size_t)row * a.opac_stride];
        }
        a.v_opacities[(size_t)b * a.N + g] = v_o;
    }
}

// UNIQUE: every Gaussian appears in at most one row (a single image, B*C == 1): plain stores into the zero-filled
// outputs instead of atomics (124 -> ~45 us at 1M rows: ten scattered fp32 atomics per row otherwise)
template <bool POSE, bool UNIQUE>
__global__ void __launch_bounds__(256) project_packed_bwd_kernel(const ProjBwdArgs a)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // over nnz
    const bool live   = row < a.nnz;
    uint32_t b = 0, c = 0, g = 0;
    float v_p[3] = {0.f, 0.f, 0.f}, v_S[9], v_R[9], v_t[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) v_S[i] = v_R[i] = 0.0f;
    if (live) {
        b = (uint32_t)a.batch_ids[row]; c = (uint32_t)a.camera_ids[row]; g = (uint32_t)a.gaussian_ids[row];
        ProjArgs fa{};
        fa.covars = a.covars; fa.quats = a.quats; fa.scales = a.scales; fa.N = a.N;
        float S[9];
        load_world_covar(fa, b, g, S);
        const float *pm = a.means + ((size_t)b * a.N + g) * 3;
        const float p[3] = {pm[0], pm[1], pm[2]};
        const Cam cam = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
        pair_vjp(a, cam, p, S, row, v_p, v_S, v_R, v_t, POSE);
        store_gaussian_grads<!UNIQUE>(a, b, g, a.rows_out ? (size_t)row : (size_t)b * a.N + g, v_p, v_S);
        if (a.v_opacities) { // the row's opacity cotangent -> its Gaussian (zero-filled output, like the others of this route)
            const float v_o = a.v_view_opacities[(size_t)row * a.opac_stride];
            if (UNIQUE) a.v_opacities[(size_t)b * a.N + g] = v_o;
            else atomic_add_f32(a.v_opacities + (size_t)b * a.N + g, v_o);
        }
    }
    if (POSE) {
        // rows are sorted by image; a wave spans a small contiguous range of images
        const int img  = live ? (int)(b * a.C + c) : -1;
        const int i_hi = wave_max_i32(img);
        int i_lo       = wave_max_i32(live ? -img : -2147483647);
        i_lo           = -i_lo;
        if (i_hi >= 0) {
            for (int im = i_lo; im <= i_hi; ++im) {
                float r[9], t[3];
                const bool sel = live && img == im;
                if (__builtin_amdgcn_ballot_w64(sel) == 0ull) continue;
#pragma unroll
                for (int i = 0; i < 9; ++i) r[i] = sel ? v_R[i] : 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) t[i] = sel ? v_t[i] : 0.0f;
                reduce_pose_grads(a.v_viewmats + (size_t)im * 16, r, t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// standalone quat_scale_to_covar_preci
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_sym(float *dst, const float *M, bool triu)
{
    if (triu) {
        dst[0] = M[0]; dst[1] = M[1]; dst[2] = M[2]; dst[3] = M[4]; dst[4] = M[5]; dst[5] = M[8];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = M[i];
    }
}

__global__ void __launch_bounds__(256) qs2c_fwd_kernel(const float *quats, const float *scales, int64_t n, int triu,
                                                       float *covars, float *precis)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qn[4], Rq[9], M[9];
    quat_normalize(quats + 4 * i, qn);
    quat_to_rotmat(qn, Rq);
    const int stride = triu ? 6 : 9;
    if (covars) {
        quat_scale_to_covar(Rq, scales + 3 * i, false, M);
        store_sym(covars + stride * i, M, triu);
    }
    if (precis) {
        quat_scale_to_covar(Rq, scales + 3 * i, true, M);
        store_sym(precis + stride * i, M, triu);
    }
}

__device__ __forceinline__ void load_grad_sym(const float *src, bool triu, float *G)
{
    if (triu) {
        // gradient wrt the 6-vector: off-diagonals are shared by two matrix entries
        G[0] = src[0]; G[1] = 0.5f * src[1]; G[2] = 0.5f * src[2];
        G[3] = 0.5f * src[1]; G[4] = src[3]; G[5] = 0.5f * src[4];
        G[6] = 0.5f * src[2]; G[7] = 0.5f * src[4]; G[8] = src[5];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) G[i] = src[i];
    }
}

__global__ void __launch_bounds__(256) qs2c_bwd_kernel(const float *quats, const float *scales, int64_t n, int triu,
                                                       const float *v_covars, const float *v_precis, float *v_quats,
                                                       float *v_scales)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qn[4], Rq[9], G[9], v_q[4] = {0.f, 0.f, 0.f, 0.f}, v_s[3] = {0.f, 0.f, 0.f};
    const float inv = quat_normalize(quats + 4 * i, qn);
    quat_to_rotmat(qn, Rq);
    const int stride = triu ? 6 : 9;
    if (v_covars) {
        load_grad_sym(v_covars + stride * i, triu, G);
        quat_scale_to_covar_vjp(qn, inv, Rq, scales + 3 * i, G, v_q, v_s);
    }
    if (v_precis) {
        load_grad_sym(v_precis + stride * i, triu, G);
        quat_scale_to_preci_vjp(qn, inv, Rq, scales + 3 * i, G, v_q, v_s);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = v_q[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) v_scales[3 * i + k] = v_s[k];
}

// ------------------------------------------------------------------------------------------
// projection_ewa_simple: camera-space mean/covariance -> image-space mean / 2x2 covariance (no blur, no culling)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) proj_simple_fwd_kernel(const float *means, const float *covars, const float *Ks,
                                                              int64_t rows, uint32_t N, uint32_t W, uint32_t H, int model,
                                                              float *means2d, float *covars2d)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows) return;
    const float *K = Ks + (idx / N) * 9;
    Cam cam{};
    cam.fx = K[0]; cam.cx = K[2]; cam.fy = K[4]; cam.cy = K[5];
    const Proj2D pr = project_camera(model, cam, W, H, means + idx * 3, covars + idx * 9);
    means2d[2 * idx] = pr.mx; means2d[2 * idx + 1] = pr.my;
    covars2d[4 * idx] = pr.a; covars2d[4 * idx + 1] = pr.b; covars2d[4 * idx + 2] = pr.b; covars2d[4 * idx + 3] = pr.d;
}

__global__ void __launch_bounds__(256) proj_simple_bwd_kernel(const float *means, const float *covars, const float *Ks,
                                                              int64_t rows, uint32_t N, uint32_t W, uint32_t H, int model,
                                                              const float *v_means2d, const float *v_covars2d,
                                                              float *v_means, float *v_covars)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows) return;
    const float *K = Ks + (idx / N) * 9;
    Cam cam{};
    cam.fx = K[0]; cam.cx = K[2]; cam.fy = K[4]; cam.cy = K[5];
    const float *p = means + idx * 3, *Sc = covars + idx * 9, *V = v_covars2d + idx * 4;
    float v_p[3] = {0.f, 0.f, 0.f}, v_Sc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v_Sc[i] = 0.0f;
    // symmetric part through the shared VJP; the antisymmetric part of V only reaches the covariance: J^T A J
    const float g01 = 0.5f * (V[1] + V[2]), an = 0.5f * (V[1] - V[2]);
    project_camera_vjp(model, cam, W, H, p, Sc, V[0], g01, V[3], v_means2d[2 * idx], v_means2d[2 * idx + 1], v_p, v_Sc);
    const Proj2D pr = project_camera(model, cam, W, H, p, Sc);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v_Sc[3 * i + j] += an * (pr.J[i] * pr.J[3 + j] - pr.J[3 + i] * pr.J[j]);
#pragma unroll
    for (int i = 0; i < 3; ++i) v_means[idx * 3 + i] = v_p[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) v_covars[idx * 9 + i] = v_Sc[i];
}

// packed rows -> dense (b, c, g) index: row_map[(b*C + c)*N + g] = row (the caller pre-fills the map with -1)
__global__ void __launch_bounds__(256) packed_row_map_kernel(const int64_t *batch_ids, const int64_t *camera_ids,
                                                             const int64_t *gaussian_ids, int64_t nnz, uint32_t C, uint32_t N,
                                                             int32_t *row_map)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nnz) return;
    row_map[(batch_ids[r] * C + camera_ids[r]) * (int64_t)N + gaussian_ids[r]] = (int32_t)r;
}

static int check_proj_common(const char *fn, const float *means, const float *covars, const float *quats,
                             const float *scales, const float *viewmats, const float *Ks, int camera_model)
{
    GSX_REQUIRE(means && viewmats && Ks, "%s: null means/viewmats/Ks", fn);
    GSX_REQUIRE(covars || (quats && scales), "%s: need covars or (quats and scales)", fn);
    GSX_REQUIRE(camera_model == GSX_CAMERA_PINHOLE || camera_model == GSX_CAMERA_ORTHO || camera_model == GSX_CAMERA_FISHEYE,
                "%s: unsupported camera model %d (pinhole/ortho/fisheye only)", fn, camera_model);
    return GSX_OK;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_project_ewa_fwd(const float *means, const float *covars, const float *quats, const float *scales,
                                   const float *opacities, const float *viewmats, const float *Ks, uint32_t B,
                                   uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                   float near_plane, float far_plane, float radius_clip, int camera_model,
                                   int32_t *radii, float *means2d, float *depths, float *conics, float *compensations,
                                   void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    if (count == 0) return GSX_OK;
    int rc = check_proj_common("gsx_project_ewa_fwd", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(radii && means2d && depths && conics, "gsx_project_ewa_fwd: null output");
    ProjArgs a{};
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.opacities = opacities;
    a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.calc_compensations = compensations != nullptr;
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.conics = conics; a.compensations = compensations;
    if (C > 1)
        project_fwd_gaussian_major_kernel<<<dim3((uint32_t)ceil_div((int64_t)B * N, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    else
        project_fwd_kernel<<<dim3((uint32_t)ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_fwd");
}

extern "C" int gsx_project_ewa_packed_count(const float *means, const float *covars, const float *quats,
                                            const float *scales, const float *opacities, const float *viewmats,
                                            const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                            uint32_t height, float eps2d, float near_plane, float far_plane,
                                            float radius_clip, int camera_model, int calc_compensations,
                                            int32_t *visible, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    if (count == 0) return GSX_OK;
    int rc = check_proj_common("gsx_project_ewa_packed_count", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(visible, "gsx_project_ewa_packed_count: null output");
    ProjArgs a{};
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.opacities = opacities;
    a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.calc_compensations = calc_compensations; a.visible = visible;
    project_count_kernel<<<dim3((uint32_t)ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_packed_count");
}

extern "C" int gsx_project_ewa_packed_write(const float *means, const float *covars, const float *quats,
                                            const float *scales, const float *opacities, const float *viewmats,
                                            const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                            uint32_t height, float eps2d, float near_plane, float far_plane,
                                            float radius_clip, int camera_model, const int64_t *row_offsets,
                                            int64_t nnz, int64_t *batch_ids, int64_t *camera_ids,
                                            int64_t *gaussian_ids, int32_t *indptr, int32_t *radii, float *means2d,
                                            float *depths, float *conics, float *compensations, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    GSX_REQUIRE(indptr, "gsx_project_ewa_packed_write: null indptr");
    if (count == 0) {
        // indptr [B*C+1] all zeros
        if (hipMemsetAsync(indptr, 0, ((size_t)B * C + 1) * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
            set_last_error("gsx_project_ewa_packed_write: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    int rc = check_proj_common("gsx_project_ewa_packed_write", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(row_offsets, "gsx_project_ewa_packed_write: null row_offsets");
    GSX_REQUIRE(nnz == 0 || (batch_ids && camera_ids && gaussian_ids && radii && means2d && depths && conics),
                "gsx_project_ewa_packed_write: null output");
    ProjArgs a{};
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.opacities = opacities;
    a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.calc_compensations = compensations != nullptr;
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.conics = conics; a.compensations = compensations;
    a.row_offsets = row_offsets; a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids;
    a.gaussian_ids = gaussian_ids; a.indptr = indptr;
    project_write_kernel<<<dim3((uint32_t)ceil_div(count + 1, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_packed_write");
}

extern "C" int64_t gsx_project_packed_blocks(int64_t pairs) { return (pairs + 1 + 255) / 256; }

// gsx_project_ewa_packed_count + scan in block form (see PackedBlocks): block_counts / block_offsets are
// [gsx_project_packed_blocks(B C N)] int32; nnz_device / nnz_host (either may be null) receive the row count, the host word
// (pinned memory) as one 8-byte system-scope store that a polling caller sees as soon as the scan has run.
extern "C" int gsx_project_ewa_packed_count_blocks(const float *means, const float *covars, const float *quats,
                                                   const float *scales, const float *opacities, const float *viewmats,
                                                   const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                                   uint32_t height, float eps2d, float near_plane, float far_plane,
                                                   float radius_clip, int camera_model, int calc_compensations,
                                                   int32_t *block_counts, int32_t *block_offsets, int64_t *nnz_device,
                                                   int64_t *nnz_host, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    GSX_REQUIRE(count < (1ll << 31) - 256, "gsx_project_ewa_packed_count_blocks: %lld (image, Gaussian) pairs exceed int32 offsets",
                (long long)count);
    GSX_REQUIRE(block_counts && block_offsets, "gsx_project_ewa_packed_count_blocks: null block buffer");
    if (count > 0) {
        int rc = check_proj_common("gsx_project_ewa_packed_count_blocks", means, covars, quats, scales, viewmats, Ks, camera_model);
        if (rc != GSX_OK) return rc;
    }
    ProjArgs a{};
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.opacities = opacities;
    a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.calc_compensations = calc_compensations;
    PackedBlocks pb{block_counts, block_offsets};
    const uint32_t n_blocks = (uint32_t)gsx_project_packed_blocks(count);
    project_count_blocks_kernel<<<dim3(n_blocks), dim3(256), 0, (hipStream_t)stream>>>(a, pb);
    packed_block_scan_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(block_counts, n_blocks, block_offsets, nnz_device,
                                                                            nnz_host);
    return check_launch("project_ewa_packed_count_blocks");
}

// gsx_project_ewa_packed_write with the rows placed from the block offsets above (same outputs, bit for bit)
extern "C" int gsx_project_ewa_packed_write_blocks(const float *means, const float *covars, const float *quats,
                                                   const float *scales, const float *opacities, const float *viewmats,
                                                   const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                                                   uint32_t height, float eps2d, float near_plane, float far_plane,
                                                   float radius_clip, int camera_model, const int32_t *block_offsets,
                                                   int64_t *batch_ids, int64_t *camera_ids, int64_t *gaussian_ids,
                                                   int32_t *indptr, int32_t *radii, float *means2d, float *depths,
                                                   float *conics, float *compensations, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    GSX_REQUIRE(indptr && block_offsets, "gsx_project_ewa_packed_write_blocks: null indptr / block offsets");
    GSX_REQUIRE(count < (1ll << 31) - 256, "gsx_project_ewa_packed_write_blocks: too many (image, Gaussian) pairs");
    if (count == 0) { // indptr [B*C+1] all zeros
        if (hipMemsetAsync(indptr, 0, ((size_t)B * C + 1) * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
            set_last_error("gsx_project_ewa_packed_write_blocks: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    if (count > 0) {
        int rc = check_proj_common("gsx_project_ewa_packed_write_blocks", means, covars, quats, scales, viewmats, Ks, camera_model);
        if (rc != GSX_OK) return rc;
        GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && radii && means2d && depths && conics,
                    "gsx_project_ewa_packed_write_blocks: null output");
    }
    ProjArgs a{};
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.opacities = opacities;
    a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
    a.camera_model = camera_model; a.calc_compensations = compensations != nullptr;
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.conics = conics; a.compensations = compensations;
    a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids; a.indptr = indptr;
    PackedBlocks pb{nullptr, block_offsets};
    project_write_blocks_kernel<<<dim3((uint32_t)gsx_project_packed_blocks(count)), dim3(256), 0, (hipStream_t)stream>>>(a, pb);
    return check_launch("project_ewa_packed_write_blocks");
}

static void fill_bwd(ProjBwdArgs &a, const float *means, const float *covars, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width,
                     uint32_t height, float eps2d, int camera_model, const float *conics, const float *compensations,
                     const float *v_means2d, uint32_t m2_stride, const float *v_depths, const float *v_conics,
                     uint32_t con_stride, const float *v_compensations, float *v_means, float *v_covars,
                     float *v_quats, float *v_scales, float *v_viewmats)
{
    a.m2_stride = m2_stride; a.con_stride = con_stride;
    a.means = means; a.covars = covars; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks;
    a.B = B; a.C = C; a.N = N; a.width = width; a.height = height; a.eps2d = eps2d; a.camera_model = camera_model;
    a.conics = conics; a.compensations = compensations; a.v_means2d = v_means2d; a.v_depths = v_depths;
    a.v_conics = v_conics; a.v_compensations = (compensations && v_compensations) ? v_compensations : nullptr;
    a.v_means = v_means; a.v_covars = v_covars; a.v_quats = v_quats; a.v_scales = v_scales; a.v_viewmats = v_viewmats;
}

extern "C" int gsx_project_ewa_bwd(const float *means, const float *covars, const float *quats, const float *scales,
                                   const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                   uint32_t width, uint32_t height, float eps2d, int camera_model,
                                   const int32_t *radii, const float *conics, const float *compensations,
                                   const float *v_means2d, uint32_t v_means2d_stride, const float *v_depths,
                                   const float *v_conics, uint32_t v_conics_stride,
                                   const float *v_compensations, float *v_means, float *v_covars, float *v_quats,
                                   float *v_scales, float *v_viewmats, void *stream)
{
    const int64_t count = (int64_t)B * N;
    if (count == 0 || C == 0) return GSX_OK;
    int rc = check_proj_common("gsx_project_ewa_bwd", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(radii && conics && v_means2d && v_conics, "gsx_project_ewa_bwd: null input");
    GSX_REQUIRE(v_means2d_stride >= 2 && v_conics_stride >= 3, "gsx_project_ewa_bwd: row strides must be >= 2 / >= 3");
    ProjBwdArgs a{};
    fill_bwd(a, means, covars, quats, scales, viewmats, Ks, B, C, N, width, height, eps2d, camera_model, conics,
             compensations, v_means2d, v_means2d_stride, v_depths, v_conics, v_conics_stride, v_compensations, v_means,
             v_covars, v_quats, v_scales, v_viewmats);
    a.radii = radii;
    const dim3 grid((uint32_t)ceil_div(count, 256)), block(256);
    if (v_viewmats) project_bwd_kernel<true><<<grid, block, 0, (hipStream_t)stream>>>(a);
    else project_bwd_kernel<false><<<grid, block, 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_bwd");
}

// gsx_project_ewa_bwd that also reduces the cotangent of the per-view opacities: v_view_opacities[(b C + c) N + g] at
// v_view_opacities_stride floats per element (1 = contiguous; the row stride of gsx_raster3d_bwd's gradient rows when it is
// their opacity column) -> v_opacities[b N + g] = sum over c. No counterpart in the reference, where the per-view opacities are a
// broadcast view and autograd reduces their gradient with its own kernels (gsplat/rendering.py:511-520).
extern "C" int gsx_project_ewa_bwd_opac(const float *means, const float *covars, const float *quats, const float *scales,
                                        const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                        uint32_t width, uint32_t height, float eps2d, int camera_model,
                                        const int32_t *radii, const float *conics, const float *compensations,
                                        const float *v_means2d, uint32_t v_means2d_stride, const float *v_depths,
                                        const float *v_conics, uint32_t v_conics_stride, const float *v_compensations,
                                        const float *v_view_opacities, uint32_t v_view_opacities_stride, float *v_means,
                                        float *v_covars, float *v_quats, float *v_scales, float *v_viewmats,
                                        float *v_opacities, void *stream)
{
    const int64_t count = (int64_t)B * N;
    if (count == 0) return GSX_OK;
    GSX_REQUIRE(v_opacities, "gsx_project_ewa_bwd_opac: null v_opacities");
    if (C == 0) { // no views: the cotangent is empty, every Gaussian's gradient is zero
        if (hipMemsetAsync(v_opacities, 0, (size_t)count * sizeof(float), (hipStream_t)stream) != hipSuccess) {
            set_last_error("gsx_project_ewa_bwd_opac: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    GSX_REQUIRE(v_view_opacities && v_view_opacities_stride >= 1, "gsx_project_ewa_bwd_opac: null opacity cotangent");
    int rc = check_proj_common("gsx_project_ewa_bwd_opac", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(radii && conics && v_means2d && v_conics, "gsx_project_ewa_bwd_opac: null input");
    GSX_REQUIRE(v_means2d_stride >= 2 && v_conics_stride >= 3, "gsx_project_ewa_bwd_opac: row strides must be >= 2 / >= 3");
    ProjBwdArgs a{};
    fill_bwd(a, means, covars, quats, scales, viewmats, Ks, B, C, N, width, height, eps2d, camera_model, conics,
             compensations, v_means2d, v_means2d_stride, v_depths, v_conics, v_conics_stride, v_compensations, v_means,
             v_covars, v_quats, v_scales, v_viewmats);
    a.radii = radii;
    a.v_view_opacities = v_view_opacities; a.opac_stride = v_view_opacities_stride; a.v_opacities = v_opacities;
    const dim3 grid((uint32_t)ceil_div(count, 256)), block(256);
    if (v_viewmats) project_bwd_kernel<true><<<grid, block, 0, (hipStream_t)stream>>>(a);
    else project_bwd_kernel<false><<<grid, block, 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_bwd_opac");
}

extern "C" int gsx_project_ewa_packed_bwd(const float *means, const float *covars, const float *quats,
                                          const float *scales, const float *viewmats, const float *Ks, uint32_t B,
                                          uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                          int camera_model, int64_t nnz, const int64_t *batch_ids,
                                          const int64_t *camera_ids, const int64_t *gaussian_ids, const float *conics,
                                          const float *compensations, const float *v_means2d,
                                          uint32_t v_means2d_stride, const float *v_depths, const float *v_conics,
                                          uint32_t v_conics_stride, const float *v_compensations,
                                          const int32_t *row_map, float *v_means, float *v_covars, float *v_quats,
                                          float *v_scales, float *v_viewmats, void *stream)
{
    if (nnz <= 0) return GSX_OK;
    int rc = check_proj_common("gsx_project_ewa_packed_bwd", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && conics && v_means2d && v_conics,
                "gsx_project_ewa_packed_bwd: null input");
    GSX_REQUIRE(v_means2d_stride >= 2 && v_conics_stride >= 3,
                "gsx_project_ewa_packed_bwd: row strides must be >= 2 / >= 3");
    ProjBwdArgs a{};
    fill_bwd(a, means, covars, quats, scales, viewmats, Ks, B, C, N, width, height, eps2d, camera_model, conics,
             compensations, v_means2d, v_means2d_stride, v_depths, v_conics, v_conics_stride, v_compensations, v_means,
             v_covars, v_quats, v_scales, v_viewmats);
    a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    hipStream_t s = (hipStream_t)stream;
    if (row_map) {
        // Gaussian-major walk through the row map: one thread per Gaussian loops over the images, every output row is
        // written exactly once (no atomics, no zero-initialised outputs needed) — the dense kernel on packed rows
        a.row_map = row_map;
        const dim3 g2((uint32_t)ceil_div((int64_t)B * N, 256));
        if (v_viewmats) project_bwd_kernel<true><<<g2, dim3(256), 0, s>>>(a);
        else project_bwd_kernel<false><<<g2, dim3(256), 0, s>>>(a);
        return check_launch("project_ewa_packed_bwd");
    }
    const dim3 grid((uint32_t)ceil_div(nnz, 256)), block(256);
    const bool unique = (uint64_t)B * C == 1;
    if (v_viewmats) {
        if (unique) project_packed_bwd_kernel<true, true><<<grid, block, 0, s>>>(a);
        else project_packed_bwd_kernel<true, false><<<grid, block, 0, s>>>(a);
    } else {
        if (unique) project_packed_bwd_kernel<false, true><<<grid, block, 0, s>>>(a);
        else project_packed_bwd_kernel<false, false><<<grid, block, 0, s>>>(a);
    }
    return check_launch("project_ewa_packed_bwd");
}

// gsx_project_ewa_packed_bwd (either route: with a row map Gaussian-major, without one row-major into zero-filled outputs,
// v_opacities included) that also reduces the cotangent of the packed rows' opacities: v_view_opacities[row] at v_view_opacities_stride floats per row (the opacity column of gsx_raster3d_bwd's gradient
// rows, read in place) -> v_opacities[b N + g] = sum over the Gaussian's rows, 0 for a Gaussian without rows. Replaces the
// index_add (+ zero fill) autograd runs for `opacities[gaussian_ids]` (reference gsplat/rendering.py:507-510).
extern "C" int gsx_project_ewa_packed_bwd_opac(const float *means, const float *covars, const float *quats,
                                               const float *scales, const float *viewmats, const float *Ks, uint32_t B,
                                               uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                               int camera_model, int64_t nnz, const int64_t *batch_ids,
                                               const int64_t *camera_ids, const int64_t *gaussian_ids, const float *conics,
                                               const float *compensations, const float *v_means2d,
                                               uint32_t v_means2d_stride, const float *v_depths, const float *v_conics,
                                               uint32_t v_conics_stride, const float *v_compensations,
                                               const float *v_view_opacities, uint32_t v_view_opacities_stride,
                                               const int32_t *row_map, float *v_means, float *v_covars, float *v_quats,
                                               float *v_scales, float *v_viewmats, float *v_opacities, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((int64_t)B * N == 0) return GSX_OK; // no Gaussians: nothing to write
    GSX_REQUIRE(v_opacities, "gsx_project_ewa_packed_bwd_opac: null v_opacities");
    if (nnz <= 0) { // no rows: the cotangent is empty (its pointer may be null), every Gaussian's gradient is zero
        if (hipMemsetAsync(v_opacities, 0, (size_t)B * N * sizeof(float), s) != hipSuccess) {
            set_last_error("gsx_project_ewa_packed_bwd_opac: memset failed");
            return GSX_ERR_LAUNCH;
        }
        return GSX_OK;
    }
    GSX_REQUIRE(v_view_opacities && v_view_opacities_stride >= 1, "gsx_project_ewa_packed_bwd_opac: null opacity cotangent");
    int rc = check_proj_common("gsx_project_ewa_packed_bwd_opac", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && conics && v_means2d && v_conics,
                "gsx_project_ewa_packed_bwd_opac: null input");
    GSX_REQUIRE(v_means2d_stride >= 2 && v_conics_stride >= 3,
                "gsx_project_ewa_packed_bwd_opac: row strides must be >= 2 / >= 3");
    ProjBwdArgs a{};
    fill_bwd(a, means, covars, quats, scales, viewmats, Ks, B, C, N, width, height, eps2d, camera_model, conics,
             compensations, v_means2d, v_means2d_stride, v_depths, v_conics, v_conics_stride, v_compensations, v_means,
             v_covars, v_quats, v_scales, v_viewmats);
    a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.row_map = row_map;
    a.v_view_opacities = v_view_opacities; a.opac_stride = v_view_opacities_stride; a.v_opacities = v_opacities;
    if (row_map) { // Gaussian-major: every output row written once
        const dim3 g2((uint32_t)ceil_div((int64_t)B * N, 256));
        if (v_viewmats) project_bwd_kernel<true><<<g2, dim3(256), 0, s>>>(a);
        else project_bwd_kernel<false><<<g2, dim3(256), 0, s>>>(a);
        return check_launch("project_ewa_packed_bwd_opac");
    }
    // row-major: into ZERO-FILLED outputs (v_opacities as well), plain stores for a single image, atomics otherwise
    const dim3 grid((uint32_t)ceil_div(nnz, 256)), block(256);
    const bool unique = (uint64_t)B * C == 1;
    if (v_viewmats) {
        if (unique) project_packed_bwd_kernel<true, true><<<grid, block, 0, s>>>(a);
        else project_packed_bwd_kernel<true, false><<<grid, block, 0, s>>>(a);
    } else {
        if (unique) project_packed_bwd_kernel<false, true><<<grid, block, 0, s>>>(a);
        else project_packed_bwd_kernel<false, false><<<grid, block, 0, s>>>(a);
    }
    return check_launch("project_ewa_packed_bwd_opac");
}

// sparse_grad=True (reference Projection.cpp:1125-1200, kernel ProjectionEWA3DGSPacked.cu:385-684): the per-Gaussian
// gradients are [nnz, .] ROWS, one per packed row, written once each with plain stores (the caller wraps them as COO over
// gaussian_ids; no dense [N, .] tensor exists anywhere). v_viewmats as in gsx_project_ewa_packed_bwd.
extern "C" int gsx_project_ewa_packed_bwd_rows(const float *means, const float *covars, const float *quats,
                                               const float *scales, const float *viewmats, const float *Ks, uint32_t B,
                                               uint32_t C, uint32_t N, uint32_t width, uint32_t height, float eps2d,
                                               int camera_model, int64_t nnz, const int64_t *batch_ids,
                                               const int64_t *camera_ids, const int64_t *gaussian_ids,
                                               const float *conics, const float *compensations, const float *v_means2d,
                                               uint32_t v_means2d_stride, const float *v_depths, const float *v_conics,
                                               uint32_t v_conics_stride, const float *v_compensations,
                                               float *v_means_rows, float *v_covars_rows, float *v_quats_rows,
                                               float *v_scales_rows, float *v_viewmats, void *stream)
{
    if (nnz <= 0) return GSX_OK;
    int rc = check_proj_common("gsx_project_ewa_packed_bwd_rows", means, covars, quats, scales, viewmats, Ks, camera_model);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && conics && v_means2d && v_conics,
                "gsx_project_ewa_packed_bwd_rows: null input");
    GSX_REQUIRE(v_means2d_stride >= 2 && v_conics_stride >= 3,
                "gsx_project_ewa_packed_bwd_rows: row strides must be >= 2 / >= 3");
    ProjBwdArgs a{};
    fill_bwd(a, means, covars, quats, scales, viewmats, Ks, B, C, N, width, height, eps2d, camera_model, conics,
             compensations, v_means2d, v_means2d_stride, v_depths, v_conics, v_conics_stride, v_compensations,
             v_means_rows, v_covars_rows, v_quats_rows, v_scales_rows, v_viewmats);
    a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.rows_out = 1;
    const dim3 grid((uint32_t)ceil_div(nnz, 256)), block(256);
    if (v_viewmats) project_packed_bwd_kernel<true, true><<<grid, block, 0, (hipStream_t)stream>>>(a);
    else project_packed_bwd_kernel<false, true><<<grid, block, 0, (hipStream_t)stream>>>(a);
    return check_launch("project_ewa_packed_bwd_rows");
}

extern "C" int gsx_quat_scale_to_covar_fwd(const float *quats, const float *scales, int64_t n, int triu,
                                           float *covars, float *precis, void *stream)
{
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(quats && scales, "gsx_quat_scale_to_covar_fwd: null input");
    qs2c_fwd_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(quats, scales, n, triu,
                                                                                           covars, precis);
    return check_launch("quat_scale_to_covar_fwd");
}

extern "C" int gsx_quat_scale_to_covar_bwd(const float *quats, const float *scales, int64_t n, int triu,
                                           const float *v_covars, const float *v_precis, float *v_quats,
                                           float *v_scales, void *stream)
{
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(quats && scales && v_quats && v_scales, "gsx_quat_scale_to_covar_bwd: null pointer");
    qs2c_bwd_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        quats, scales, n, triu, v_covars, v_precis, v_quats, v_scales);
    return check_launch("quat_scale_to_covar_bwd");
}

extern "C" int gsx_project_simple_fwd(const float *means, const float *covars, const float *Ks, int64_t rows,
                                      uint32_t n_per_camera, uint32_t width, uint32_t height, int camera_model,
                                      float *means2d, float *covars2d, void *stream)
{
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means && covars && Ks && means2d && covars2d && n_per_camera > 0, "gsx_project_simple_fwd: bad argument");
    proj_simple_fwd_kernel<<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        means, covars, Ks, rows, n_per_camera, width, height, camera_model, means2d, covars2d);
    return check_launch("project_simple_fwd");
}

extern "C" int gsx_project_simple_bwd(const float *means, const float *covars, const float *Ks, int64_t rows,
                                      uint32_t n_per_camera, uint32_t width, uint32_t height, int camera_model,
                                      const float *v_means2d, const float *v_covars2d, float *v_means, float *v_covars,
                                      void *stream)
{
    if (rows == 0) return GSX_OK;
    GSX_REQUIRE(means && covars && Ks && v_means2d && v_covars2d && v_means && v_covars && n_per_camera > 0,
                "gsx_project_simple_bwd: bad argument");
    proj_simple_bwd_kernel<<<dim3((uint32_t)ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        means, covars, Ks, rows, n_per_camera, width, height, camera_model, v_means2d, v_covars2d, v_means, v_covars);
    return check_launch("project_simple_bwd");
}

extern "C" int gsx_packed_row_map(const int64_t *batch_ids, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                  int64_t nnz, uint32_t B, uint32_t C, uint32_t N, int32_t *row_map, void *stream)
{
    const int64_t total = (int64_t)B * C * N;
    if (total == 0) return GSX_OK;
    GSX_REQUIRE(row_map, "gsx_packed_row_map: null output");
    GSX_REQUIRE(nnz >= 0 && nnz < (1ll << 31), "gsx_packed_row_map: nnz out of range");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(row_map, 0xFF, (size_t)total * sizeof(int32_t), s) != hipSuccess) { // all bytes 0xFF = int32 -1
        set_last_error("gsx_packed_row_map: memset failed");
        return GSX_ERR_LAUNCH;
    }
    if (nnz == 0) return GSX_OK;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids, "gsx_packed_row_map: null ids");
    packed_row_map_kernel<<<dim3((uint32_t)ceil_div(nnz, 256)), dim3(256), 0, s>>>(batch_ids, camera_ids, gaussian_ids, nnz, C,
                                                                                  N, row_map);
    return check_launch("packed_row_map");
}
