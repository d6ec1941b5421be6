// quat_scale_to_covar_preci in double precision: the reference instantiates this op for float AND double
// (gsplat/cuda/csrc/QuatScaleToCovarCUDA.cu:145, AT_DISPATCH_FLOATING_TYPES) and its tests call it with float64 inputs. A
// translation unit of its own - the float kernels of projection.hip and their register allocation are not touched: the same
// formulas (projmath.hpp: unit quaternion -> rotation, Sigma = (R diag s)(R diag s)^T, its inverse with 1 / s, and their
// vector-Jacobian products), every operation in IEEE double.
// C-ABI entries: gsx_quat_scale_to_covar_{fwd,bwd}_f64.
#include "common.hpp"

namespace gsx {
namespace {

__device__ __forceinline__ double qnorm(const double *q, double *n)
{
    const double inv = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    n[0] = q[0] * inv; n[1] = q[1] * inv; n[2] = q[2] * inv; n[3] = q[3] * inv;
    return inv;
}
__device__ __forceinline__ void rotmat(const double *n /* unit wxyz */, double *R)
{
    const double w = n[0], x = n[1], y = n[2], z = n[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}
// dL/dq (raw, un-normalised quaternion) += from dL/dR
__device__ __forceinline__ void rotmat_vjp(const double *n, double inv_norm, const double *v, double *v_q)
{
    const double w = n[0], x = n[1], y = n[2], z = n[3];
    double g[4];
    g[0] = 2.0 * (-z * v[1] + y * v[2] + z * v[3] - x * v[5] - y * v[6] + x * v[7]);
    g[1] = 2.0 * (y * v[1] + z * v[2] + y * v[3] - 2.0 * x * v[4] - w * v[5] + z * v[6] + w * v[7] - 2.0 * x * v[8]);
    g[2] = 2.0 * (-2.0 * y * v[0] + x * v[1] + w * v[2] + x * v[3] + z * v[5] - w * v[6] + z * v[7] - 2.0 * y * v[8]);
    g[3] = 2.0 * (-2.0 * z * v[0] - w * v[1] + x * v[2] + w * v[3] - 2.0 * z * v[4] + y * v[5] + x * v[6] + y * v[7]);
    const double d = g[0] * n[0] + g[1] * n[1] + g[2] * n[2] + g[3] * n[3];
    for (int i = 0; i < 4; ++i) v_q[i] += (g[i] - d * n[i]) * inv_norm;
}
// M M^T with M = R diag(t), t = s or 1 / s
__device__ __forceinline__ void outer(const double *R, const double *t, double *S)
{
    double M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[3 * i + j] = R[3 * i + j] * t[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[3 * i + j] = M[3 * i] * M[3 * j] + M[3 * i + 1] * M[3 * j + 1] + M[3 * i + 2] * M[3 * j + 2];
}
// VJP of S = M M^T, M = R diag(t): v_R accumulated through rotmat_vjp, v_t[j] returned (the caller maps it to v_s)
__device__ __forceinline__ void outer_vjp(const double *qn, double inv_norm, const double *R, const double *t, const double *v_S,
                                          double *v_q, double *v_t)
{
    double M[9], G[9], v_M[9], v_R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            M[3 * i + j] = R[3 * i + j] * t[j];
            G[3 * i + j] = v_S[3 * i + j] + v_S[3 * j + i];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v_M[3 * i + j] = G[3 * i] * M[j] + G[3 * i + 1] * M[3 + j] + G[3 * i + 2] * M[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v_R[3 * i + j] = v_M[3 * i + j] * t[j];
    rotmat_vjp(qn, inv_norm, v_R, v_q);
    for (int j = 0; j < 3; ++j) v_t[j] = R[j] * v_M[j] + R[3 + j] * v_M[3 + j] + R[6 + j] * v_M[6 + j];
}
__device__ __forceinline__ void store_sym(double *dst, const double *M, bool triu)
{
    if (triu) {
        dst[0] = M[0]; dst[1] = M[1]; dst[2] = M[2]; dst[3] = M[4]; dst[4] = M[5]; dst[5] = M[8];
    } else {
        for (int i = 0; i < 9; ++i) dst[i] = M[i];
    }
}
__device__ __forceinline__ void load_grad_sym(const double *src, bool triu, double *G)
{
    if (triu) { // gradient wrt the 6-vector: an off-diagonal is shared by two matrix entries
        G[0] = src[0]; G[1] = 0.5 * src[1]; G[2] = 0.5 * src[2];
        G[3] = 0.5 * src[1]; G[4] = src[3]; G[5] = 0.5 * src[4];
        G[6] = 0.5 * src[2]; G[7] = 0.5 * src[4]; G[8] = src[5];
    } else {
        for (int i = 0; i < 9; ++i) G[i] = src[i];
    }
}

__global__ void __launch_bounds__(256) qs2c_fwd_f64_kernel(const double *quats, const double *scales, int64_t n, int triu,
                                                           double *covars, double *precis)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qn[4], R[9], M[9];
    qnorm(quats + 4 * i, qn);
    rotmat(qn, R);
    const double *s   = scales + 3 * i;
    const int stride  = triu ? 6 : 9;
    if (covars) {
        outer(R, s, M);
        store_sym(covars + stride * i, M, triu);
    }
    if (precis) {
        const double is[3] = {1.0 / s[0], 1.0 / s[1], 1.0 / s[2]};
        outer(R, is, M);
        store_sym(precis + stride * i, M, triu);
    }
}

__global__ void __launch_bounds__(256) qs2c_bwd_f64_kernel(const double *quats, const double *scales, int64_t n, int triu,
                                                           const double *v_covars, const double *v_precis, double *v_quats,
                                                           double *v_scales)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qn[4], R[9], G[9], v_q[4] = {0.0, 0.0, 0.0, 0.0}, v_s[3] = {0.0, 0.0, 0.0}, v_t[3];
    const double inv = qnorm(quats + 4 * i, qn);
    rotmat(qn, R);
    const double *s  = scales + 3 * i;
    const int stride = triu ? 6 : 9;
    if (v_covars) {
        load_grad_sym(v_covars + stride * i, triu, G);
        outer_vjp(qn, inv, R, s, G, v_q, v_t);
        for (int j = 0; j < 3; ++j) v_s[j] += v_t[j];
    }
    if (v_precis) {
        const double is[3] = {1.0 / s[0], 1.0 / s[1], 1.0 / s[2]};
        load_grad_sym(v_precis + stride * i, triu, G);
        outer_vjp(qn, inv, R, is, G, v_q, v_t);
        for (int j = 0; j < 3; ++j) v_s[j] += -is[j] * is[j] * v_t[j]; // d(1 / s) / ds
    }
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = v_q[k];
    for (int k = 0; k < 3; ++k) v_scales[3 * i + k] = v_s[k];
}

} // namespace
} // namespace gsx

using namespace gsx;

extern "C" int gsx_quat_scale_to_covar_fwd_f64(const double *quats, const double *scales, int64_t n, int triu, double *covars,
                                               double *precis, void *stream)
{
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(quats && scales, "gsx_quat_scale_to_covar_fwd_f64: null input");
    qs2c_fwd_f64_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(quats, scales, n, triu, covars, precis);
    return check_launch("quat_scale_to_covar_fwd_f64");
}

extern "C" int gsx_quat_scale_to_covar_bwd_f64(const double *quats, const double *scales, int64_t n, int triu,
                                               const double *v_covars, const double *v_precis, double *v_quats,
                                               double *v_scales, void *stream)
{
    if (n <= 0) return GSX_OK;
    GSX_REQUIRE(quats && scales && v_quats && v_scales, "gsx_quat_scale_to_covar_bwd_f64: null pointer");
    qs2c_bwd_f64_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(quats, scales, n, triu, v_covars,
                                                                                             v_precis, v_quats, v_scales);
    return check_launch("quat_scale_to_covar_bwd_f64");
}
