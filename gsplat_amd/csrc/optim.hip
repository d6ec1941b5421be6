// Optimizer-side kernels of the training step that calls the rasterizer, gfx950.
//
// gsx_adam        replaces gsplat::adam (reference gsplat/cuda/ext.cpp:1217, kernel csrc/AdamCUDA.cu:34-75): fused,
//                 row-masked Adam step WITHOUT bias correction (the reference kernel has none): for every element p of a
//                 row g with valid[g]:   m = b1 m + (1-b1) grad;  v = b2 v + (1-b2) grad^2;  p -= lr m / (sqrt(v) + eps).
//                 Rows with valid[g] == false are left untouched (parameter AND moments): "selective Adam" of Taming-3DGS.
// gsx_relocation  replaces gsplat::relocation (reference csrc/RelocationCUDA.cu:34-80, python gsplat/relocation.py:23-67):
//                 3DGS-as-MCMC (arXiv 2404.09591, Eq. 9): new opacity 1 - (1 - o)^(1/n) clamped to [min_opacity, 1 - eps],
//                 new scale = o / sum_{i=1..n} sum_{k<i} C(i-1, k) (-1)^k o_new^(k+1) / sqrt(k+1)  x  old scale.
//
// gsx_mcmc_perturb replaces gsplat::mcmc_perturb_positions (reference ext.cpp:1256; csrc/MCMCPerturbCUDA.cu:24-58): the
//                 SGLD noise step of the MCMC strategy, means += Sigma (noise * sigmoid(-k (sigmoid(o_logit) - t)) *
//                 noise_scale), Sigma = R diag(exp(log_scales))^2 R^T from the (un-normalised) quaternion.
//
// All three are pure HBM streams (Adam: 16 B read + 12 B written per element). Adam is vectorised 4 floats per lane when the
// row width allows it (rows of one Gaussian are contiguous, so a float4 never straddles two rows when D % 4 == 0 ... and
// a per-element row lookup is used otherwise).
#include "projmath.hpp"
#include <float.h>

namespace gsx {

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float lr, float b1, float b2, float eps)
{
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    p += -lr * m / (sqrtf(v) + eps);
}

__global__ void __launch_bounds__(256) adam_kernel(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                                   const uint8_t *valid, int64_t n_rows, uint32_t D, float lr, float b1,
                                                   float b2, float eps)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * (int64_t)D) return;
    if (valid && !valid[i / D]) return;
    float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    adam_one(p, grad[i], m, v, lr, b1, b2, eps);
    param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
}

// D % 4 == 0 and 16-byte aligned bases: one float4 per lane, the four elements belong to one row
__global__ void __launch_bounds__(256) adam_vec4_kernel(float4 *param, const float4 *grad, float4 *exp_avg,
                                                        float4 *exp_avg_sq, const uint8_t *valid, int64_t n_vec,
                                                        uint32_t vec_per_row, float lr, float b1, float b2, float eps)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vec) return;
    if (valid && !valid[i / vec_per_row]) return;
    float4 p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    const float4 g = grad[i];
    adam_one(p.x, g.x, m.x, v.x, lr, b1, b2, eps);
    adam_one(p.y, g.y, m.y, v.y, lr, b1, b2, eps);
    adam_one(p.z, g.z, m.z, v.z, lr, b1, b2, eps);
    adam_one(p.w, g.w, m.w, v.w, lr, b1, b2, eps);
    param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
}

__global__ void __launch_bounds__(256) relocation_kernel(const float *opacities, const float *scales, const int32_t *ratios,
                                                         const float *binoms, int64_t n, int n_max, float min_opacity,
                                                         float *new_opacities, float *new_scales)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int n_idx = ratios[i];
    const float o   = opacities[i];
    float o_new     = 1.0f - powf(1.0f - o, 1.0f / (float)n_idx);
    o_new           = fminf(fmaxf(o_new, min_opacity), 1.0f - FLT_EPSILON); // clamp BEFORE the scale (as the reference)
    new_opacities[i] = o_new;
    float denom = 0.0f;
    for (int r = 1; r <= n_idx; ++r) {
        float pw = o_new, sign = 1.0f; // o_new^(k+1), (-1)^k
        for (int k = 0; k < r; ++k) {
            denom += binoms[(r - 1) * n_max + k] * (sign / sqrtf((float)(k + 1))) * pw;
            pw *= o_new;
            sign = -sign;
        }
    }
    const float coeff = o / denom;
#pragma unroll
    for (int c = 0; c < 3; ++c) new_scales[3 * i + c] = coeff * scales[3 * i + c];
}

__global__ void __launch_bounds__(256) mcmc_perturb_kernel(float *positions, const float *quats, const float *scales_log,
                                                           const float *opacities_logit, const float *noise, int64_t n,
                                                           float noise_scale, float t, float k)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qn[4], Rq[9], S[9];
    quat_normalize(quats + 4 * i, qn);
    quat_to_rotmat(qn, Rq);
    const float sc[3] = {expf(scales_log[3 * i]), expf(scales_log[3 * i + 1]), expf(scales_log[3 * i + 2])};
    quat_scale_to_covar(Rq, sc, false, S);
    const float density = 1.0f / (1.0f + expf(-opacities_logit[i]));
    const float w       = noise_scale / (1.0f + expf(k * (density - t))); // sigmoid(-k (density - t)) * noise_scale
    const float nz[3]   = {noise[3 * i] * w, noise[3 * i + 1] * w, noise[3 * i + 2] * w};
#pragma unroll
    for (int r = 0; r < 3; ++r) positions[3 * i + r] += S[3 * r] * nz[0] + S[3 * r + 1] * nz[1] + S[3 * r + 2] * nz[2];
}

} // namespace gsx

using namespace gsx;

extern "C" int gsx_mcmc_perturb(float *positions, const float *quats, const float *scales_log, const float *opacities_logit,
                                const float *noise, int64_t n, float noise_scale, float t, float k, void *stream)
{
    GSX_REQUIRE(n >= 0, "gsx_mcmc_perturb: negative count");
    if (n == 0) return GSX_OK;
    GSX_REQUIRE(positions && quats && scales_log && opacities_logit && noise, "gsx_mcmc_perturb: null pointer");
    mcmc_perturb_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        positions, quats, scales_log, opacities_logit, noise, n, noise_scale, t, k);
    return check_launch("mcmc_perturb");
}

extern "C" int gsx_adam(float *param, const float *param_grad, float *exp_avg, float *exp_avg_sq, const uint8_t *valid,
                        int64_t n_rows, uint32_t row_width, float lr, float b1, float b2, float eps, void *stream)
{
    GSX_REQUIRE(n_rows >= 0, "gsx_adam: negative row count");
    if (n_rows == 0 || row_width == 0) return GSX_OK;
    GSX_REQUIRE(param && param_grad && exp_avg && exp_avg_sq, "gsx_adam: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = n_rows * (int64_t)row_width;
    const bool aligned = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(param_grad)
                           | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u) == 0;
    if ((row_width & 3u) == 0 && aligned) {
        const int64_t n_vec = n / 4;
        adam_vec4_kernel<<<dim3((uint32_t)ceil_div(n_vec, 256)), dim3(256), 0, s>>>(
            reinterpret_cast<float4 *>(param), reinterpret_cast<const float4 *>(param_grad),
            reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq), valid, n_vec, row_width / 4, lr, b1,
            b2, eps);
    } else {
        adam_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, s>>>(param, param_grad, exp_avg, exp_avg_sq, valid,
                                                                            n_rows, row_width, lr, b1, b2, eps);
    }
    return check_launch("adam");
}

// DefaultStrategy's per-step statistics for dense rows [C, N] (reference gsplat/strategy/default.py:226-285, _update_state): per
// Gaussian, over the views it is visible in (both radii > 0): grad2d += |(g.x half_w, g.y half_h)|, count += 1,
// radii = max(radii, max(r.x, r.y) / max_dim). The reference gathers the visible pairs (torch.where + three mask indexings, each
// with a host read of its size) and scatters with index_add_; the tensor-op form this replaces took ~10 launches per step.
// One thread per Gaussian, the views in ascending order; a row that is not visible is never read for its gradient's VALUE
// (a non-finite gradient there must not reach the sum: selects, not products).
__global__ void __launch_bounds__(256) strategy_accumulate_kernel(const float *grad, uint32_t grad_stride, const int32_t *radii,
                                                                  uint32_t C, uint32_t N, float half_w, float half_h,
                                                                  float inv_max_dim, float *grad2d, float *count, float *radii_state)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= N) return;
    float sum = 0.0f, cnt = 0.0f, rmax = 0.0f;
    for (uint32_t c = 0; c < C; ++c) {
        const size_t row = (size_t)c * N + g;
        const int2 r     = reinterpret_cast<const int2 *>(radii)[row];
        if (r.x > 0 && r.y > 0) {
            const float gx = grad[row * grad_stride] * half_w, gy = grad[row * grad_stride + 1] * half_h;
            sum += sqrtf(gx * gx + gy * gy);
            cnt += 1.0f;
            rmax = fmaxf(rmax, (float)max(r.x, r.y) * inv_max_dim);
        }
    }
    grad2d[g] += sum;
    count[g] += cnt;
    if (radii_state) radii_state[g] = fmaxf(radii_state[g], rmax);
}

extern "C" int gsx_relocation(const float *opacities, const float *scales, const int32_t *ratios, const float *binoms,
                              int64_t n, int n_max, float min_opacity, float *new_opacities, float *new_scales,
                              void *stream)
{
    GSX_REQUIRE(n >= 0 && n_max >= 1, "gsx_relocation: bad sizes");
    if (n == 0) return GSX_OK;
    GSX_REQUIRE(opacities && scales && ratios && binoms && new_opacities && new_scales, "gsx_relocation: null pointer");
    relocation_kernel<<<dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        opacities, scales, ratios, binoms, n, n_max, min_opacity, new_opacities, new_scales);
    return check_launch("relocation");
}

extern "C" int gsx_strategy_accumulate(const float *grad, uint32_t grad_stride, const int32_t *radii, uint32_t C, uint32_t N,
                                       float half_w, float half_h, float inv_max_dim, float *grad2d, float *count,
                                       float *radii_state, void *stream)
{
    if (C == 0 || N == 0) return GSX_OK;
    GSX_REQUIRE(grad && radii && grad2d && count, "gsx_strategy_accumulate: null pointer");
    GSX_REQUIRE(grad_stride >= 2, "gsx_strategy_accumulate: gradient rows are at least two floats apart, got %u", grad_stride);
    strategy_accumulate_kernel<<<dim3((uint32_t)ceil_div((int64_t)N, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        grad, grad_stride, radii, C, N, half_w, half_h, inv_max_dim, grad2d, count, radii_state);
    return check_launch("strategy_accumulate");
}
