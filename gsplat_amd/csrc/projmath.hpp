// Per-Gaussian projection math (EWA splatting) and its vector-Jacobian products, gfx950.
//
// Restates the behaviour of the reference device functions (formulas re-derived, row-major here):
//   quaternion -> rotation / covariance:  gsplat/cuda/include/Utils.cuh:228-347
//   world -> camera:                      Utils.cuh:81-148
//   blur + compensation:                  Utils.cuh:448-496
//   pinhole / ortho / fisheye projection: Utils.cuh:498-855
// All 3x3 matrices are row-major float[9]; symmetric 3x3 are stored full for clarity.
#pragma once
#include "common.hpp"

namespace gsx {

struct Cam {
    float R[9]; // world->camera rotation, row-major
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float *viewmat /*[16]*/, const float *K /*[9]*/)
{
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = viewmat[4 * i + j];
        c.t[i] = viewmat[4 * i + 3];
    }
    c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
    return c;
}

// C = A * B (3x3)
__device__ __forceinline__ void mm3(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A * B^T
__device__ __forceinline__ void mm3_nt(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
// C = A^T * B
__device__ __forceinline__ void mm3_tn(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// unit quaternion (w,x,y,z) normalised from q; returns inv_norm
__device__ __forceinline__ float quat_normalize(const float *q, float *n)
{
    const float inv = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    n[0] = q[0] * inv; n[1] = q[1] * inv; n[2] = q[2] * inv; n[3] = q[3] * inv;
    return inv;
}

__device__ __forceinline__ void quat_to_rotmat(const float *n /*unit wxyz*/, float *R)
{
    const float w = n[0], x = n[1], y = n[2], z = n[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// dL/dq (raw, un-normalised quaternion) from dL/dR
__device__ __forceinline__ void quat_to_rotmat_vjp(const float *n, float inv_norm, const float *v, float *v_q)
{
    const float w = n[0], x = n[1], y = n[2], z = n[3];
    float g[4];
    g[0] = 2.f * (-z * v[1] + y * v[2] + z * v[3] - x * v[5] - y * v[6] + x * v[7]);
    g[1] = 2.f * (y * v[1] + z * v[2] + y * v[3] - 2.f * x * v[4] - w * v[5] + z * v[6] + w * v[7] - 2.f * x * v[8]);
    g[2] = 2.f * (-2.f * y * v[0] + x * v[1] + w * v[2] + x * v[3] + z * v[5] - w * v[6] + z * v[7] - 2.f * y * v[8]);
    g[3] = 2.f * (-2.f * z * v[0] - w * v[1] + x * v[2] + w * v[3] - 2.f * z * v[4] + y * v[5] + x * v[6] + y * v[7]);
    const float d = g[0] * n[0] + g[1] * n[1] + g[2] * n[2] + g[3] * n[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) v_q[i] += (g[i] - d * n[i]) * inv_norm;
}

// Sigma = (Rq diag(s)) (Rq diag(s))^T ; if inverse: uses 1/s (precision matrix)
__device__ __forceinline__ void quat_scale_to_covar(const float *Rq, const float *s, bool inverse, float *Sig)
{
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[3 * i + j] = Rq[3 * i + j] * (inverse ? 1.0f / s[j] : s[j]);
    mm3_nt(M, M, Sig);
}

// VJP of Sigma = M M^T with M = Rq diag(s): accumulates v_q (raw quaternion) and v_s.
__device__ __forceinline__ void quat_scale_to_covar_vjp(const float *qn, float inv_norm, const float *Rq,
                                                        const float *s, const float *v_Sig, float *v_q, float *v_s)
{
    float M[9], G[9], v_M[9], v_R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            M[3 * i + j] = Rq[3 * i + j] * s[j];
            G[3 * i + j] = v_Sig[3 * i + j] + v_Sig[3 * j + i];
        }
    mm3(G, M, v_M);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v_R[3 * i + j] = v_M[3 * i + j] * s[j];
    quat_to_rotmat_vjp(qn, inv_norm, v_R, v_q);
#pragma unroll
    for (int j = 0; j < 3; ++j) v_s[j] += Rq[j] * v_M[j] + Rq[3 + j] * v_M[3 + j] + Rq[6 + j] * v_M[6 + j];
}

// VJP of P = N N^T with N = Rq diag(1/s) (precision matrix)
__device__ __forceinline__ void quat_scale_to_preci_vjp(const float *qn, float inv_norm, const float *Rq,
                                                        const float *s, const float *v_P, float *v_q, float *v_s)
{
    float N[9], G[9], v_N[9], v_R[9];
    const float is[3] = {1.0f / s[0], 1.0f / s[1], 1.0f / s[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            N[3 * i + j] = Rq[3 * i + j] * is[j];
            G[3 * i + j] = v_P[3 * i + j] + v_P[3 * j + i];
        }
    mm3(G, N, v_N);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v_R[3 * i + j] = v_N[3 * i + j] * is[j];
    quat_to_rotmat_vjp(qn, inv_norm, v_R, v_q);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        v_s[j] += -is[j] * is[j] * (Rq[j] * v_N[j] + Rq[3 + j] * v_N[3 + j] + Rq[6 + j] * v_N[6 + j]);
}

// ---- camera projection of a camera-space Gaussian ---------------------------------------
// J is 2x3 row-major; returns cov2d (a=00, b=01, d=11) and mean2d.
struct Proj2D {
    float mx, my;   // mean2d
    float a, b, d;  // cov2d (before blur)
    float J[6];
};

__device__ __forceinline__ void pinhole_limits(const Cam &c, uint32_t W, uint32_t H, float &lxp, float &lxn,
                                               float &lyp, float &lyn)
{
    const float tan_fovx = 0.5f * (float)W / c.fx, tan_fovy = 0.5f * (float)H / c.fy;
    lxp = ((float)W - c.cx) / c.fx + 0.3f * tan_fovx;
    lxn = c.cx / c.fx + 0.3f * tan_fovx;
    lyp = ((float)H - c.cy) / c.fy + 0.3f * tan_fovy;
    lyn = c.cy / c.fy + 0.3f * tan_fovy;
}

__device__ __forceinline__ void cov2d_from_J(const float *J, const float *S /*3x3 sym*/, float &a, float &b, float &d)
{
    // T = J S (2x3); cov = T J^T
    float T[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[3 * i + j] = J[3 * i] * S[j] + J[3 * i + 1] * S[3 + j] + J[3 * i + 2] * S[6 + j];
    a = T[0] * J[0] + T[1] * J[1] + T[2] * J[2];
    b = T[0] * J[3] + T[1] * J[4] + T[2] * J[5];
    d = T[3] * J[3] + T[4] * J[4] + T[5] * J[5];
}

__device__ __forceinline__ Proj2D project_camera(int model, const Cam &c, uint32_t W, uint32_t H, const float *p /*cam*/,
                                                 const float *Sc /*cam covar 3x3*/)
{
    Proj2D o;
    const float x = p[0], y = p[1], z = p[2];
    if (model == GSX_CAMERA_ORTHO) {
        o.J[0] = c.fx; o.J[1] = 0.f; o.J[2] = 0.f; o.J[3] = 0.f; o.J[4] = c.fy; o.J[5] = 0.f;
        o.mx = c.fx * x + c.cx;
        o.my = c.fy * y + c.cy;
    } else if (model == GSX_CAMERA_FISHEYE) {
        const float eps = 1e-7f;
        const float r   = sqrtf(x * x + y * y) + eps;
        const float th  = atan2f(r, z + eps);
        o.mx = x * c.fx * th / r + c.cx;
        o.my = y * c.fy * th / r + c.cy;
        const float x2 = x * x + eps, y2 = y * y, xy = x * y, r2 = x2 + y2;
        const float il2 = 1.0f / (r2 + z * z);
        const float bb  = atan2f(r, z) / r / r2;
        const float aa  = z * il2 / r2;
        o.J[0] = c.fx * (x2 * aa + y2 * bb); o.J[1] = c.fx * xy * (aa - bb); o.J[2] = -c.fx * x * il2;
        o.J[3] = c.fy * xy * (aa - bb);      o.J[4] = c.fy * (y2 * aa + x2 * bb); o.J[5] = -c.fy * y * il2;
    } else {
        float lxp, lxn, lyp, lyn;
        pinhole_limits(c, W, H, lxp, lxn, lyp, lyn);
        const float rz = 1.0f / z, rz2 = rz * rz;
        const float tx = z * fminf(lxp, fmaxf(-lxn, x * rz));
        const float ty = z * fminf(lyp, fmaxf(-lyn, y * rz));
        o.J[0] = c.fx * rz; o.J[1] = 0.f; o.J[2] = -c.fx * tx * rz2;
        o.J[3] = 0.f; o.J[4] = c.fy * rz; o.J[5] = -c.fy * ty * rz2;
        o.mx = c.fx * x * rz + c.cx;
        o.my = c.fy * y * rz + c.cy;
    }
    cov2d_from_J(o.J, Sc, o.a, o.b, o.d);
    return o;
}

// VJP of project_camera: given dL/dcov2d (symmetric g00,g01,g11 with g01 the FULL off-diagonal entry of the
// symmetric matrix, i.e. matrix [[g00,g01],[g01,g11]]) and dL/dmean2d, accumulate dL/dp (cam) and dL/dSc (3x3).
__device__ __forceinline__ void project_camera_vjp(int model, const Cam &c, uint32_t W, uint32_t H, const float *p,
                                                   const float *Sc, float g00, float g01, float g11, float vmx,
                                                   float vmy, float *v_p, float *v_Sc)
{
    const float x = p[0], y = p[1], z = p[2];
    Proj2D f = project_camera(model, c, W, H, p, Sc);
    const float *J = f.J;
    // v_Sc += J^T G J
    float GJ[6]; // G J (2x3)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        GJ[j]     = g00 * J[j] + g01 * J[3 + j];
        GJ[3 + j] = g01 * J[j] + g11 * J[3 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v_Sc[3 * i + j] += J[i] * GJ[j] + J[3 + i] * GJ[3 + j];
    // v_J = 2 G J Sc   (G and Sc symmetric)
    float vJ[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            vJ[3 * i + j] = 2.0f * (GJ[3 * i] * Sc[j] + GJ[3 * i + 1] * Sc[3 + j] + GJ[3 * i + 2] * Sc[6 + j]);

    if (model == GSX_CAMERA_ORTHO) {
        v_p[0] += c.fx * vmx;
        v_p[1] += c.fy * vmy;
    } else if (model == GSX_CAMERA_FISHEYE) {
        const float eps = 1e-7f;
        const float x2 = x * x + eps, y2 = y * y, xy = x * y, r2 = x2 + y2;
        const float r   = sqrtf(x * x + y * y) + eps;
        const float l2  = r2 + z * z, il2 = 1.0f / l2;
        const float th  = atan2f(r, z);
        const float bb  = th / r / r2;
        const float aa  = z * il2 / r2;
        // mean2d = (fx x th/r + cx, fy y th/r + cy): d(th/r)/d(x,y,z) expressed through aa, bb
        v_p[0] += c.fx * (x2 * aa + y2 * bb) * vmx + c.fy * xy * (aa - bb) * vmy;
        v_p[1] += c.fx * xy * (aa - bb) * vmx + c.fy * (y2 * aa + x2 * bb) * vmy;
        v_p[2] += -c.fx * x * il2 * vmx - c.fy * y * il2 * vmy;
        // derivatives of aa = z/(l2 r2), bb = th/(r r2), il2
        const float ir2 = 1.0f / r2, ir = 1.0f / r;
        const float daa_dx = -2.0f * x * z * (il2 * il2 * ir2 + il2 * ir2 * ir2);
        const float daa_dy = -2.0f * y * z * (il2 * il2 * ir2 + il2 * ir2 * ir2);
        const float daa_dz = il2 * ir2 - 2.0f * z * z * il2 * il2 * ir2;
        const float ir3 = ir * ir2, ir5 = ir3 * ir2;
        const float dbb_dx = z * x * il2 * ir2 * ir2 - 3.0f * th * x * ir5;
        const float dbb_dy = z * y * il2 * ir2 * ir2 - 3.0f * th * y * ir5;
        const float dbb_dz = -il2 * ir2;
        const float dil2_dx = -2.0f * x * il2 * il2, dil2_dy = -2.0f * y * il2 * il2, dil2_dz = -2.0f * z * il2 * il2;
        const float amb = aa - bb;
        // J00 = fx (x2 aa + y2 bb); J01 = fx xy amb; J02 = -fx x il2
        // J10 = fy xy amb;          J11 = fy (y2 aa + x2 bb); J12 = -fy y il2
        const float dJ00[3] = {c.fx * (2.0f * x * aa + x2 * daa_dx + y2 * dbb_dx),
                               c.fx * (x2 * daa_dy + 2.0f * y * bb + y2 * dbb_dy),
                               c.fx * (x2 * daa_dz + y2 * dbb_dz)};
        const float dM[3]   = {y * amb + xy * (daa_dx - dbb_dx), x * amb + xy * (daa_dy - dbb_dy),
                               xy * (daa_dz - dbb_dz)}; // d(xy amb)
        const float dJ02[3] = {-c.fx * (il2 + x * dil2_dx), -c.fx * x * dil2_dy, -c.fx * x * dil2_dz};
        const float dJ11[3] = {c.fy * (y2 * daa_dx + 2.0f * x * bb + x2 * dbb_dx),
                               c.fy * (2.0f * y * aa + y2 * daa_dy + x2 * dbb_dy),
                               c.fy * (y2 * daa_dz + x2 * dbb_dz)};
        const float dJ12[3] = {-c.fy * y * dil2_dx, -c.fy * (il2 + y * dil2_dy), -c.fy * y * dil2_dz};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            v_p[k] += dJ00[k] * vJ[0] + c.fx * dM[k] * vJ[1] + dJ02[k] * vJ[2] + c.fy * dM[k] * vJ[3]
                    + dJ11[k] * vJ[4] + dJ12[k] * vJ[5];
    } else {
        float lxp, lxn, lyp, lyn;
        pinhole_limits(c, W, H, lxp, lxn, lyp, lyn);
        const float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const float tx = z * fminf(lxp, fmaxf(-lxn, xr));
        const float ty = z * fminf(lyp, fmaxf(-lyn, yr));
        v_p[0] += c.fx * rz * vmx;
        v_p[1] += c.fy * rz * vmy;
        v_p[2] += -(c.fx * x * vmx + c.fy * y * vmy) * rz2;
        // J02 = -fx tx / z^2 ; J12 = -fy ty / z^2 ; J00 = fx / z ; J11 = fy / z
        if (xr <= lxp && xr >= -lxn) v_p[0] += -c.fx * rz2 * vJ[2];
        else v_p[2] += -c.fx * rz3 * vJ[2] * tx;
        if (yr <= lyp && yr >= -lyn) v_p[1] += -c.fy * rz2 * vJ[5];
        else v_p[2] += -c.fy * rz3 * vJ[5] * ty;
        v_p[2] += -c.fx * rz2 * vJ[0] - c.fy * rz2 * vJ[4] + 2.0f * c.fx * tx * rz3 * vJ[2] + 2.0f * c.fy * ty * rz3 * vJ[5];
    }
}

} // namespace gsx
