// 2DGS (surfel) projection: ray-splat transform K [R Rq diag(sx,sy) | mean_c], screen-space AABB and normal, gfx950.
//
// C-ABI entries: gsx_project_2dgs_{fwd,bwd}, gsx_project_2dgs_packed_{count,write,bwd}.
// Replaces gsplat::projection_2dgs_fused{,_bwd} / projection_2dgs_packed{,_bwd}
// (reference kernels gsplat/cuda/csrc/Projection2DGSFused.cu:39-339, 341-505, Projection2DGSPacked.cu,
//  VJP gsplat/cuda/csrc/Projection2DGS.cuh:29-115; torch restatement _torch_impl_2dgs.py:27-108).
//
// Same design as projection.hip: the dense backward runs one thread per Gaussian and loops over the cameras in
// registers (no atomics, deterministic), pose gradients are wave-reduced; the packed forward is count -> scan -> write.
// Formulas are re-derived in row-major form: with u = R Rq[:,0] sx, v = R Rq[:,1] sy, n = R Rq[:,2],
// W = [u v mean_c] (columns), the stored ray transform has rows
//   M0 = fx W[0,:] + cx W[2,:],  M1 = fy W[1,:] + cy W[2,:],  M2 = W[2,:].
#include "projmath.hpp"

namespace gsx {

struct Proj2Args {
    const float *means, *quats, *scales, *viewmats, *Ks;
    uint32_t B, C, N, width, height;
    float near_plane, far_plane, radius_clip;
    int32_t *radii;
    float *means2d, *depths, *ray_transforms, *normals;
    int32_t *visible;
    const int64_t *row_offsets;
    int64_t nnz;
    int64_t *batch_ids, *camera_ids, *gaussian_ids;
    int32_t *indptr;
};

struct Proj2Out {
    bool ok;
    int rx, ry;
    float mx, my, depth;
    float M[9];
    float nrm[3];
};

// Inlined into the dense, count and write kernels; the file is compiled with -ffp-contract=off (Makefile) so all
// copies perform the same IEEE operations (bit-identical rows, count/write always agree on visibility).
__device__ __forceinline__ Proj2Out project2_one(const Proj2Args &a, uint32_t b, uint32_t c, uint32_t g)
{
    Proj2Out o;
    o.ok = false;
    o.rx = o.ry = 0;
    o.mx = o.my = o.depth = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.M[i] = 0.0f;
    o.nrm[0] = o.nrm[1] = o.nrm[2] = 0.0f;

    const Cam cam  = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
    const float *p = a.means + ((size_t)b * a.N + g) * 3;
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = cam.R[3 * i] * p[0] + cam.R[3 * i + 1] * p[1] + cam.R[3 * i + 2] * p[2] + cam.t[i];
    if (pc[2] <= a.near_plane || pc[2] >= a.far_plane) return o; // reference: <= / >= here (Fused.cu:155)

    const float *q = a.quats + ((size_t)b * a.N + g) * 4;
    const float *s = a.scales + ((size_t)b * a.N + g) * 3;
    float qn[4], Rq[9], RR[9];
    quat_normalize(q, qn);
    quat_to_rotmat(qn, Rq);
    mm3(cam.R, Rq, RR); // camera-space axes of the surfel (columns)
    // W rows: (u_i, v_i, pc_i)
    float W[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        W[3 * i + 0] = RR[3 * i + 0] * s[0];
        W[3 * i + 1] = RR[3 * i + 1] * s[1];
        W[3 * i + 2] = pc[i];
    }
    float M0[3], M1[3], M2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        M0[j] = cam.fx * W[j] + cam.cx * W[6 + j];
        M1[j] = cam.fy * W[3 + j] + cam.cy * W[6 + j];
        M2[j] = W[6 + j];
    }
    const float distance = M2[0] * M2[0] + M2[1] * M2[1] - M2[2] * M2[2];
    if (distance == 0.0f) return o;
    const float f  = 1.0f / distance;
    const float mx = f * (M0[0] * M2[0] + M0[1] * M2[1] - M0[2] * M2[2]);
    const float my = f * (M1[0] * M2[0] + M1[1] * M2[1] - M1[2] * M2[2]);
    const float tx = f * (M0[0] * M0[0] + M0[1] * M0[1] - M0[2] * M0[2]);
    const float ty = f * (M1[0] * M1[0] + M1[1] * M1[1] - M1[2] * M1[2]);
    const float rx = ceilf(kGaussianExtend * sqrtf(fmaxf(1e-4f, mx * mx - tx)));
    const float ry = ceilf(kGaussianExtend * sqrtf(fmaxf(1e-4f, my * my - ty)));
    if (rx <= a.radius_clip && ry <= a.radius_clip) return o;
    if (mx + rx <= 0.0f || mx - rx >= (float)a.width || my + ry <= 0.0f || my - ry >= (float)a.height) return o;

    // normal = third camera-space axis, flipped to face the camera
    const float n0 = RR[2], n1 = RR[5], n2 = RR[8];
    const float mult = (-(n0 * pc[0] + n1 * pc[1] + n2 * pc[2])) > 0.0f ? 1.0f : -1.0f;
    o.ok = true;
    o.rx = (int)rx; o.ry = (int)ry;
    o.mx = mx; o.my = my; o.depth = pc[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o.M[j] = M0[j]; o.M[3 + j] = M1[j]; o.M[6 + j] = M2[j];
    }
    o.nrm[0] = n0 * mult; o.nrm[1] = n1 * mult; o.nrm[2] = n2 * mult;
    return o;
}

__device__ __forceinline__ void store2(const Proj2Args &a, int64_t row, const Proj2Out &o)
{
    a.radii[2 * row]       = o.rx;
    a.radii[2 * row + 1]   = o.ry;
    a.means2d[2 * row]     = o.mx;
    a.means2d[2 * row + 1] = o.my;
    a.depths[row]          = o.depth;
#pragma unroll
    for (int i = 0; i < 9; ++i) a.ray_transforms[9 * row + i] = o.M[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.normals[3 * row + i] = o.nrm[i];
}

__global__ void __launch_bounds__(256) project2_fwd_kernel(const Proj2Args a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx >= count) return;
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    store2(a, idx, project2_one(a, b, c, g)); // culled rows: radii 0 and zeros elsewhere
}

__global__ void __launch_bounds__(256) project2_count_kernel(const Proj2Args a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx >= count) return;
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    a.visible[idx]   = project2_one(a, b, c, g).ok ? 1 : 0;
}

__global__ void __launch_bounds__(256) project2_write_kernel(const Proj2Args a)
{
    const int64_t idx   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t count = (int64_t)a.B * a.C * a.N;
    if (idx > count) return;
    if (idx % a.N == 0) a.indptr[idx / a.N] = (int32_t)(idx == 0 ? 0 : a.row_offsets[idx - 1]);
    if (idx == count) return;
    const int64_t row  = idx == 0 ? 0 : a.row_offsets[idx - 1];
    const int64_t next = a.row_offsets[idx];
    if (next == row) return;
    const uint32_t g = (uint32_t)(idx % a.N), c = (uint32_t)((idx / a.N) % a.C), b = (uint32_t)(idx / ((int64_t)a.N * a.C));
    a.batch_ids[row]    = b;
    a.camera_ids[row]   = c;
    a.gaussian_ids[row] = g;
    store2(a, row, project2_one(a, b, c, g));
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
struct Proj2BwdArgs {
    const float *means, *quats, *scales, *viewmats, *Ks;
    uint32_t B, C, N;
    const int32_t *radii;          // dense only
    const float *ray_transforms;   // per row
    const float *v_means2d, *v_depths, *v_ray_transforms, *v_normals;
    uint32_t depth_stride; // floats between the depth cotangents of consecutive rows (1 = contiguous)
    uint32_t m2_stride, rt_stride, n_stride; // row strides (floats): 2 / 9 / 3, or the stride of the AoS gradient rows
    int64_t nnz;
    const int64_t *batch_ids, *camera_ids, *gaussian_ids;
    int rows_out; // packed, sparse_grad: outputs are [nnz, .] rows (one per packed row) instead of [B, N, .]
    float *v_means, *v_quats, *v_scales, *v_viewmats;
    // (optional, dense) the cotangent of the per-view opacities [B, C, N], opac_stride floats apart (a column of the compositing
    // backward's gradient rows), summed over the views into v_opacities [B, N] - as in projection.hip
    const float *v_view_opacities;
    uint32_t opac_stride;
    float *v_opacities;
};

// VJP of one (camera, surfel) pair. Accumulates v_p (world mean), v_Rq (rotation matrix of the surfel, 3x3),
// v_s (sx, sy) and, if want_pose, v_R / v_t of the camera.
__device__ __forceinline__ void pair2_vjp(const Proj2BwdArgs &a, const Cam &cam, const float *p, const float *Rq,
                                          const float *s, int64_t row, float *v_p, float *v_Rq, float *v_s,
                                          float *v_R, float *v_t, bool want_pose)
{
    const float *M = a.ray_transforms + 9 * row;
    float vM[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) vM[i] = a.v_ray_transforms[(size_t)a.rt_stride * row + i];
    if (a.v_depths) vM[8] += a.v_depths[(size_t)a.depth_stride * row]; // depth = M2.z (null = no gradient reaches the depths)
    const float vmx = a.v_means2d[(size_t)a.m2_stride * row], vmy = a.v_means2d[(size_t)a.m2_stride * row + 1];
    if (vmx != 0.0f || vmy != 0.0f) {
        // mean2d_x = sum(sgn M0 M2) / d, d = sum(sgn M2 M2), sgn = (1,1,-1)
        const float sg[3] = {1.0f, 1.0f, -1.0f};
        const float d = M[6] * M[6] + M[7] * M[7] - M[8] * M[8];
        const float f = 1.0f / d;
        const float px = f * (M[0] * M[6] + M[1] * M[7] - M[2] * M[8]);
        const float py = f * (M[3] * M[6] + M[4] * M[7] - M[5] * M[8]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float sm2 = sg[j] * M[6 + j] * f;
            vM[j]     += vmx * sm2;
            vM[3 + j] += vmy * sm2;
            vM[6 + j] += vmx * (sg[j] * M[j] * f - 2.0f * px * sm2) + vmy * (sg[j] * M[3 + j] * f - 2.0f * py * sm2);
        }
    }
    // W rows from M rows: M0 = fx W0 + cx W2, M1 = fy W1 + cy W2, M2 = W2
    float vW[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vW[j]     = cam.fx * vM[j];
        vW[3 + j] = cam.fy * vM[3 + j];
        vW[6 + j] = cam.cx * vM[j] + cam.cy * vM[3 + j] + vM[6 + j];
    }
    // columns of vW: v_u, v_v, v_pc
    const float v_u[3]  = {vW[0], vW[3], vW[6]};
    const float v_vv[3] = {vW[1], vW[4], vW[7]};
    const float v_pc[3] = {vW[2], vW[5], vW[8]};
    // normal (camera space) = mult * R Rq[:,2]
    float pc[3], RR2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pc[i]  = cam.R[3 * i] * p[0] + cam.R[3 * i + 1] * p[1] + cam.R[3 * i + 2] * p[2] + cam.t[i];
        RR2[i] = cam.R[3 * i] * Rq[2] + cam.R[3 * i + 1] * Rq[5] + cam.R[3 * i + 2] * Rq[8];
    }
    const float mult = (-(RR2[0] * pc[0] + RR2[1] * pc[1] + RR2[2] * pc[2])) > 0.0f ? 1.0f : -1.0f;
    const float *vnp  = a.v_normals + (size_t)a.n_stride * row;
    const float v_n[3] = {mult * vnp[0], mult * vnp[1], mult * vnp[2]};

    // back through the camera rotation: x_cam = R x_world
    float tu[3], tv[3], tn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tu[j] = cam.R[j] * v_u[0] + cam.R[3 + j] * v_u[1] + cam.R[6 + j] * v_u[2];
        tv[j] = cam.R[j] * v_vv[0] + cam.R[3 + j] * v_vv[1] + cam.R[6 + j] * v_vv[2];
        tn[j] = cam.R[j] * v_n[0] + cam.R[3 + j] * v_n[1] + cam.R[6 + j] * v_n[2];
        v_p[j] += cam.R[j] * v_pc[0] + cam.R[3 + j] * v_pc[1] + cam.R[6 + j] * v_pc[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v_Rq[3 * i + 0] += tu[i] * s[0];
        v_Rq[3 * i + 1] += tv[i] * s[1];
        v_Rq[3 * i + 2] += tn[i];
        v_s[0] += tu[i] * Rq[3 * i + 0];
        v_s[1] += tv[i] * Rq[3 * i + 1];
    }
    if (want_pose) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                v_R[3 * i + j] += v_u[i] * Rq[3 * j] * s[0] + v_vv[i] * Rq[3 * j + 1] * s[1] + v_n[i] * Rq[3 * j + 2]
                                + v_pc[i] * p[j];
            v_t[i] += v_pc[i];
        }
    }
}

// wave-reduce the 12 pose-gradient values and add them to v_viewmats[b,c] (same scheme as projection.hip)
__device__ __forceinline__ void reduce_pose_grads2(float *v_viewmat, const float *v_R, const float *v_t)
{
    const uint32_t lane = threadIdx.x & 63u;
    float mine = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float r = wave_sum4_scatter(v_R[3 * j], v_R[3 * j + 1], v_R[3 * j + 2], v_t[j]);
        if ((int)(lane & 15u) == j) mine = r;
    }
    const int row = (int)(lane & 15u), col = (int)(lane >> 4);
    if (row < 3) atomic_add_f32(v_viewmat + 4 * row + col, mine);
}

template <bool POSE>
__global__ void __launch_bounds__(256) project2_bwd_kernel(const Proj2BwdArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // over B*N
    const bool live   = idx < (int64_t)a.B * a.N;
    const uint32_t b = live ? (uint32_t)(idx / a.N) : 0, g = live ? (uint32_t)(idx % a.N) : 0;
    float p[3] = {0.f, 0.f, 0.f}, s[2] = {1.f, 1.f}, qn[4] = {1.f, 0.f, 0.f, 0.f}, Rq[9], inv = 1.0f;
    if (live) {
        const float *pm = a.means + ((size_t)b * a.N + g) * 3;
        p[0] = pm[0]; p[1] = pm[1]; p[2] = pm[2];
        s[0] = a.scales[((size_t)b * a.N + g) * 3];
        s[1] = a.scales[((size_t)b * a.N + g) * 3 + 1];
        inv  = quat_normalize(a.quats + ((size_t)b * a.N + g) * 4, qn);
    }
    quat_to_rotmat(qn, Rq);
    float v_p[3] = {0.f, 0.f, 0.f}, v_s[2] = {0.f, 0.f}, v_Rq[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v_Rq[i] = 0.0f;

    for (uint32_t c = 0; c < a.C; ++c) {
        float v_R[9], v_t[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) v_R[i] = 0.0f;
        if (live) {
            const int64_t row = ((int64_t)b * a.C + c) * a.N + g;
            if (a.radii[2 * row] > 0 && a.radii[2 * row + 1] > 0) {
                const Cam cam = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
                pair2_vjp(a, cam, p, Rq, s, row, v_p, v_Rq, v_s, v_R, v_t, POSE);
            }
        }
        if (POSE) {
            const uint32_t b_lo = __builtin_amdgcn_readfirstlane(b);
            const uint32_t b_hi = (uint32_t)wave_max_i32(live ? (int)b : (int)b_lo);
            for (uint32_t bb = b_lo; bb <= b_hi; ++bb) {
                float r[9], t[3];
                const bool sel = live && (b == bb);
#pragma unroll
                for (int i = 0; i < 9; ++i) r[i] = sel ? v_R[i] : 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) t[i] = sel ? v_t[i] : 0.0f;
                reduce_pose_grads2(a.v_viewmats + ((size_t)bb * a.C + c) * 16, r, t);
            }
        }
    }
    if (live) {
        const size_t bg = (size_t)b * a.N + g;
        float v_q[4] = {0.f, 0.f, 0.f, 0.f};
        quat_to_rotmat_vjp(qn, inv, v_Rq, v_q);
#pragma unroll
        for (int i = 0; i < 3; ++i) a.v_means[bg * 3 + i] = v_p[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) a.v_quats[bg * 4 + i] = v_q[i];
        a.v_scales[bg * 3 + 0] = v_s[0];
        a.v_scales[bg * 3 + 1] = v_s[1];
        a.v_scales[bg * 3 + 2] = 0.0f;
        if (a.v_opacities) { // an invisible pair's entry is the exact zero the compositing backward never touched
            float v_o = 0.0f;
            for (uint32_t c = 0; c < a.C; ++c) v_o += a.v_view_opacities[(size_t)(((int64_t)b * a.C + c) * a.N + g) * a.opac_stride];
            a.v_opacities[bg] = v_o;
        }
    }
}

// packed: one thread per row, atomics into zero-initialised outputs
template <bool POSE>
__global__ void __launch_bounds__(256) project2_packed_bwd_kernel(const Proj2BwdArgs a)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live   = row < a.nnz;
    uint32_t b = 0, c = 0, g = 0;
    if (live) {
        b = (uint32_t)a.batch_ids[row]; c = (uint32_t)a.camera_ids[row]; g = (uint32_t)a.gaussian_ids[row];
    }
    float v_R[9], v_t[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) v_R[i] = 0.0f;
    if (live) {
        const size_t bg = (size_t)b * a.N + g;
        const float *pm = a.means + bg * 3;
        const float p[3] = {pm[0], pm[1], pm[2]};
        const float s[2] = {a.scales[bg * 3], a.scales[bg * 3 + 1]};
        float qn[4], Rq[9];
        const float inv = quat_normalize(a.quats + bg * 4, qn);
        quat_to_rotmat(qn, Rq);
        float v_p[3] = {0.f, 0.f, 0.f}, v_s[2] = {0.f, 0.f}, v_Rq[9], v_q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) v_Rq[i] = 0.0f;
        const Cam cam = load_cam(a.viewmats + ((size_t)b * a.C + c) * 16, a.Ks + ((size_t)b * a.C + c) * 9);
        pair2_vjp(a, cam, p, Rq, s, row, v_p, v_Rq, v_s, v_R, v_t, POSE);
        quat_to_rotmat_vjp(qn, inv, v_Rq, v_q);
        if (a.rows_out) { // sparse_grad: [nnz, .] rows, each written once
#pragma unroll
            for (int i = 0; i < 3; ++i) a.v_means[row * 3 + i] = v_p[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) a.v_quats[row * 4 + i] = v_q[i];
            a.v_scales[row * 3 + 0] = v_s[0];
            a.v_scales[row * 3 + 1] = v_s[1];
            a.v_scales[row * 3 + 2] = 0.0f;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) atomic_add_f32(a.v_means + bg * 3 + i, v_p[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) atomic_add_f32(a.v_quats + bg * 4 + i, v_q[i]);
            atomic_add_f32(a.v_scales + bg * 3 + 0, v_s[0]);
            atomic_add_f32(a.v_scales + bg * 3 + 1, v_s[1]);
        }
    }
    if (POSE) {
        // rows are sorted by image: most waves hold a single (b,c); loop over the images present in the wave
        const int img    = live ? (int)(b * a.C + c) : -1;
        const int img_hi = wave_max_i32(img);
        int img_lo       = -wave_max_i32(live ? -img : -0x7fffffff);
        if (img_hi < 0) return;
        for (int im = img_lo; im <= img_hi; ++im) {
            float r[9], t[3];
            const bool sel = live && (img == im);
            if (__builtin_amdgcn_ballot_w64(sel) == 0ull) continue;
#pragma unroll
            for (int i = 0; i < 9; ++i) r[i] = sel ? v_R[i] : 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) t[i] = sel ? v_t[i] : 0.0f;
            reduce_pose_grads2(a.v_viewmats + (size_t)im * 16, r, t);
        }
    }
}

static int check2(const char *fn, const float *means, const float *quats, const float *scales, const float *viewmats,
                  const float *Ks)
{
    GSX_REQUIRE(means && quats && scales && viewmats && Ks, "%s: null input", fn);
    return GSX_OK;
}

static void fill2(Proj2Args &a, const float *means, const float *quats, const float *scales, const float *viewmats,
                  const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height, float near_plane,
                  float far_plane, float radius_clip)
{
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks;
    a.B = B; a.C = C; a.N = N; a.width = width; a.height = height;
    a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_project_2dgs_fwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                                    const float *Ks, uint32_t B, uint32_t C, uint32_t N, uint32_t width, uint32_t height,
                                    float near_plane, float far_plane, float radius_clip, int32_t *radii, float *means2d,
                                    float *depths, float *ray_transforms, float *normals, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    if (count == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_fwd", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(radii && means2d && depths && ray_transforms && normals, "gsx_project_2dgs_fwd: null output");
    Proj2Args a{};
    fill2(a, means, quats, scales, viewmats, Ks, B, C, N, width, height, near_plane, far_plane, radius_clip);
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.ray_transforms = ray_transforms; a.normals = normals;
    project2_fwd_kernel<<<dim3((uint32_t)ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_fwd");
}

extern "C" int gsx_project_2dgs_packed_count(const float *means, const float *quats, const float *scales,
                                             const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                             uint32_t width, uint32_t height, float near_plane, float far_plane,
                                             float radius_clip, int32_t *visible, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    if (count == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_packed_count", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(visible, "gsx_project_2dgs_packed_count: null output");
    Proj2Args a{};
    fill2(a, means, quats, scales, viewmats, Ks, B, C, N, width, height, near_plane, far_plane, radius_clip);
    a.visible = visible;
    project2_count_kernel<<<dim3((uint32_t)ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_packed_count");
}

extern "C" int gsx_project_2dgs_packed_write(const float *means, const float *quats, const float *scales,
                                             const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                             uint32_t width, uint32_t height, float near_plane, float far_plane,
                                             float radius_clip, const int64_t *row_offsets, int64_t nnz,
                                             int64_t *batch_ids, int64_t *camera_ids, int64_t *gaussian_ids,
                                             int32_t *indptr, int32_t *radii, float *means2d, float *depths,
                                             float *ray_transforms, float *normals, void *stream)
{
    const int64_t count = (int64_t)B * C * N;
    if (count == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_packed_write", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(row_offsets && indptr, "gsx_project_2dgs_packed_write: null row_offsets/indptr");
    GSX_REQUIRE(nnz == 0 || (batch_ids && camera_ids && gaussian_ids && radii && means2d && depths && ray_transforms
                             && normals), "gsx_project_2dgs_packed_write: null output");
    Proj2Args a{};
    fill2(a, means, quats, scales, viewmats, Ks, B, C, N, width, height, near_plane, far_plane, radius_clip);
    a.row_offsets = row_offsets; a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids;
    a.gaussian_ids = gaussian_ids; a.indptr = indptr;
    a.radii = radii; a.means2d = means2d; a.depths = depths; a.ray_transforms = ray_transforms; a.normals = normals;
    project2_write_kernel<<<dim3((uint32_t)ceil_div(count + 1, 256)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_packed_write");
}

extern "C" int gsx_project_2dgs_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                                    const float *Ks, uint32_t B, uint32_t C, uint32_t N, const int32_t *radii,
                                    const float *ray_transforms, const float *v_means2d, const float *v_depths,
                                    const float *v_ray_transforms, const float *v_normals, uint32_t v_row_stride,
                                    uint32_t v_depths_stride, float *v_means, float *v_quats,
                                    float *v_scales, float *v_viewmats, void *stream)
{
    if ((int64_t)B * N == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_bwd", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(C == 0 || (radii && ray_transforms && v_means2d && v_ray_transforms && v_normals),
                "gsx_project_2dgs_bwd: null input");
    GSX_REQUIRE(v_means && v_quats && v_scales, "gsx_project_2dgs_bwd: null output");
    Proj2BwdArgs a{};
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N;
    a.radii = radii; a.ray_transforms = ray_transforms; a.v_means2d = v_means2d; a.v_depths = v_depths;
    a.v_ray_transforms = v_ray_transforms; a.v_normals = v_normals;
    a.m2_stride = v_row_stride ? v_row_stride : 2u; a.rt_stride = v_row_stride ? v_row_stride : 9u;
    a.n_stride = v_row_stride ? v_row_stride : 3u;
    a.depth_stride = v_depths_stride ? v_depths_stride : 1u;
    a.v_means = v_means; a.v_quats = v_quats; a.v_scales = v_scales; a.v_viewmats = v_viewmats;
    const dim3 grid((uint32_t)ceil_div((int64_t)B * N, 256));
    if (v_viewmats) project2_bwd_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    else project2_bwd_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_bwd");
}

// gsx_project_2dgs_bwd that also reduces the cotangent of the per-view opacities (see gsx_project_ewa_bwd_opac):
// v_view_opacities[(b C + c) N + g] at v_view_opacities_stride floats per element -> v_opacities[b N + g] = sum over c.
extern "C" int gsx_project_2dgs_bwd_opac(const float *means, const float *quats, const float *scales, const float *viewmats,
                                         const float *Ks, uint32_t B, uint32_t C, uint32_t N, const int32_t *radii,
                                         const float *ray_transforms, const float *v_means2d, const float *v_depths,
                                         const float *v_ray_transforms, const float *v_normals, uint32_t v_row_stride,
                                         uint32_t v_depths_stride, const float *v_view_opacities,
                                         uint32_t v_view_opacities_stride, float *v_means,
                                         float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities, void *stream)
{
    if ((int64_t)B * N == 0) return GSX_OK;
    GSX_REQUIRE(v_opacities, "gsx_project_2dgs_bwd_opac: null v_opacities");
    int rc = check2("gsx_project_2dgs_bwd_opac", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(C == 0 || (radii && ray_transforms && v_means2d && v_ray_transforms && v_normals && v_view_opacities
                           && v_view_opacities_stride >= 1), "gsx_project_2dgs_bwd_opac: null input");
    GSX_REQUIRE(v_means && v_quats && v_scales, "gsx_project_2dgs_bwd_opac: null output");
    Proj2BwdArgs a{};
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N;
    a.radii = radii; a.ray_transforms = ray_transforms; a.v_means2d = v_means2d; a.v_depths = v_depths;
    a.v_ray_transforms = v_ray_transforms; a.v_normals = v_normals;
    a.m2_stride = v_row_stride ? v_row_stride : 2u; a.rt_stride = v_row_stride ? v_row_stride : 9u;
    a.n_stride = v_row_stride ? v_row_stride : 3u;
    a.depth_stride = v_depths_stride ? v_depths_stride : 1u;
    a.v_means = v_means; a.v_quats = v_quats; a.v_scales = v_scales; a.v_viewmats = v_viewmats;
    a.v_view_opacities = v_view_opacities; a.opac_stride = v_view_opacities_stride; a.v_opacities = v_opacities;
    const dim3 grid((uint32_t)ceil_div((int64_t)B * N, 256));
    if (v_viewmats) project2_bwd_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    else project2_bwd_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_bwd_opac");
}

extern "C" int gsx_project_2dgs_packed_bwd(const float *means, const float *quats, const float *scales,
                                           const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                           int64_t nnz, const int64_t *batch_ids, const int64_t *camera_ids,
                                           const int64_t *gaussian_ids, const float *ray_transforms,
                                           const float *v_means2d, const float *v_depths, const float *v_ray_transforms,
                                           const float *v_normals, uint32_t v_row_stride, float *v_means,
                                           float *v_quats, float *v_scales,
                                           float *v_viewmats, void *stream)
{
    if (nnz == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_packed_bwd", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && ray_transforms && v_means2d && v_ray_transforms
                && v_normals, "gsx_project_2dgs_packed_bwd: null input");
    GSX_REQUIRE(v_means && v_quats && v_scales, "gsx_project_2dgs_packed_bwd: null output");
    Proj2BwdArgs a{};
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N;
    a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.ray_transforms = ray_transforms; a.v_means2d = v_means2d; a.v_depths = v_depths;
    a.v_ray_transforms = v_ray_transforms; a.v_normals = v_normals;
    a.m2_stride = v_row_stride ? v_row_stride : 2u; a.rt_stride = v_row_stride ? v_row_stride : 9u;
    a.n_stride = v_row_stride ? v_row_stride : 3u;
    a.depth_stride = 1u;
    a.v_means = v_means; a.v_quats = v_quats; a.v_scales = v_scales; a.v_viewmats = v_viewmats;
    const dim3 grid((uint32_t)ceil_div(nnz, 256));
    if (v_viewmats) project2_packed_bwd_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    else project2_packed_bwd_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_packed_bwd");
}

// sparse_grad=True (reference Projection.cpp:1780-1863): v_means / v_quats / v_scales are [nnz, .] rows, one per packed row,
// written once each with plain stores; the caller wraps them as COO over gaussian_ids.
extern "C" int gsx_project_2dgs_packed_bwd_rows(const float *means, const float *quats, const float *scales,
                                           const float *viewmats, const float *Ks, uint32_t B, uint32_t C, uint32_t N,
                                           int64_t nnz, const int64_t *batch_ids, const int64_t *camera_ids,
                                           const int64_t *gaussian_ids, const float *ray_transforms,
                                           const float *v_means2d, const float *v_depths, const float *v_ray_transforms,
                                           const float *v_normals, uint32_t v_row_stride, float *v_means,
                                           float *v_quats, float *v_scales,
                                           float *v_viewmats, void *stream)
{
    if (nnz == 0) return GSX_OK;
    int rc = check2("gsx_project_2dgs_packed_bwd_rows", means, quats, scales, viewmats, Ks);
    if (rc != GSX_OK) return rc;
    GSX_REQUIRE(batch_ids && camera_ids && gaussian_ids && ray_transforms && v_means2d && v_ray_transforms
                && v_normals, "gsx_project_2dgs_packed_bwd_rows: null input");
    GSX_REQUIRE(v_means && v_quats && v_scales, "gsx_project_2dgs_packed_bwd_rows: null output");
    Proj2BwdArgs a{};
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks; a.B = B; a.C = C; a.N = N;
    a.nnz = nnz; a.batch_ids = batch_ids; a.camera_ids = camera_ids; a.gaussian_ids = gaussian_ids;
    a.ray_transforms = ray_transforms; a.v_means2d = v_means2d; a.v_depths = v_depths;
    a.v_ray_transforms = v_ray_transforms; a.v_normals = v_normals;
    a.m2_stride = v_row_stride ? v_row_stride : 2u; a.rt_stride = v_row_stride ? v_row_stride : 9u;
    a.n_stride = v_row_stride ? v_row_stride : 3u;
    a.depth_stride = 1u;
    a.v_means = v_means; a.v_quats = v_quats; a.v_scales = v_scales; a.v_viewmats = v_viewmats;
    a.rows_out = 1;
    const dim3 grid((uint32_t)ceil_div(nnz, 256));
    if (v_viewmats) project2_packed_bwd_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    else project2_packed_bwd_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
    return check_launch("project_2dgs_packed_bwd_rows");
}
