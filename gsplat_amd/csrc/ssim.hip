// Fused SSIM loss of the training step (examples/simple_trainer.py:951-961 blends
// `1 - SSIM` with L1; gsplat/losses.py:150-200 evaluates it with the third-party `fused_ssim` CUDA extension when that is
// installed, else with five depthwise 11 x 11 torch convolutions - 4.7 ms per 1080p step through MIOpen on this GPU).
// C-ABI: gsx_ssim_fwd / gsx_ssim_bwd. Semantics restated from gsplat/losses.py: torch_ssim_loss (Wang et al. 2004: 11-tap
// Gaussian window of sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over batch, channels and pixels).
//
// One workgroup = one 32 x 16 tile of one (batch, channel) plane. The window is an outer product, so the five local moments
// (x, y, xx, yy, xy) are taken as a horizontal 11-tap pass over the tile + 5 halo rows each side (LDS), then a vertical pass
// per pixel. The forward also stores the three partial derivatives of the SSIM value with respect to the local moments of
// image 1 (mu1, sigma1^2, sigma12); the backward is the same two passes over those maps:
//   dL/dx(p) = sum_q w(q - p) [ g(q) (dmu1(q) + 2 x(p) dsig1(q) + y(p) dsig12(q)) ],  g = dL/dmap (a constant for the mean).
// Images are read through strides, so the rasterizer's [B, H, W, C] output is consumed in place as [B, C, H, W].
#include "common.hpp"

namespace gsx {

constexpr int kSsimTw = 32, kSsimTh = 16, kSsimPad = 5, kSsimTaps = 11;
__constant__ float kSsimW[kSsimTaps] = {0.0010283803567290306f, 0.0075987582094967365f, 0.036000773310661316f,
                                        0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                        0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                        0.0075987582094967365f, 0.0010283803567290306f};

struct SsimArgs {
    const float *x, *y;        // image 1 (differentiated), image 2
    int64_t sx[4], sy[4];      // strides (elements) of (b, c, h, w)
    int32_t B, C, H, W;
    float *partial;            // fwd: [n_blocks] sums of the SSIM map
    float *dmaps;              // [B, C, H, W, 3] contiguous: d ssim / d (mu1, sigma1^2, sigma12); null = no derivative maps
    float grad_scale;          // bwd: dL/dmap = grad_scale * (*grad_dev, when given)
    const float *grad_dev;     // bwd: device scalar (the incoming gradient of the mean), or null
    float *gx;                 // bwd: gradient of image 1, strides sgx
    int64_t sgx[4];
};

template <bool BWD>
__global__ void __launch_bounds__(kSsimTw *kSsimTh) ssim_kernel(const SsimArgs a)
{
    constexpr int IW = kSsimTw + 2 * kSsimPad, IH = kSsimTh + 2 * kSsimPad; // 42 x 26 inputs
    constexpr int NM = BWD ? 3 : 5;                                          // maps convolved
    __shared__ float s_in[NM][IH][IW + 1];
    __shared__ float s_h[NM][IH][kSsimTw + 1];
    const int tx = threadIdx.x % kSsimTw, ty = threadIdx.x / kSsimTw;
    const int plane = blockIdx.z, b = plane / a.C, c = plane % a.C;
    const int x0 = blockIdx.x * kSsimTw, y0 = blockIdx.y * kSsimTh;
    const float *px = a.x + b * a.sx[0] + c * a.sx[1];
    const float *py = a.y + b * a.sy[0] + c * a.sy[1];
    const float *pd = a.dmaps + ((size_t)plane * a.H * a.W) * 3;
    // tile + halo (zeros outside the image = zero padding)
    for (int i = threadIdx.x; i < IH * IW; i += kSsimTw * kSsimTh) {
        const int r = i / IW, q = i % IW, gy = y0 + r - kSsimPad, gx = x0 + q - kSsimPad;
        const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if constexpr (BWD) {
            float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
            if (in) { // the address is only formed for pixels of the image (gy, gx may be negative in the halo)
                const float *d = pd + ((size_t)gy * a.W + (size_t)gx) * 3;
                d0 = d[0]; d1 = d[1]; d2 = d[2];
            }
            s_in[0][r][q] = d0;
            s_in[1][r][q] = d1;
            s_in[2][r][q] = d2;
        } else {
            const float vx = in ? px[gy * a.sx[2] + gx * a.sx[3]] : 0.0f, vy = in ? py[gy * a.sy[2] + gx * a.sy[3]] : 0.0f;
            s_in[0][r][q] = vx;
            s_in[1][r][q] = vy;
            s_in[2][r][q] = vx * vx;
            s_in[3][r][q] = vy * vy;
            s_in[4][r][q] = vx * vy;
        }
    }
    __syncthreads();
    // horizontal pass: IH rows x kSsimTw columns
    for (int i = threadIdx.x; i < IH * kSsimTw; i += kSsimTw * kSsimTh) {
        const int r = i / kSsimTw, q = i % kSsimTw;
        float acc[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = 0.0f;
#pragma unroll
        for (int t = 0; t < kSsimTaps; ++t) {
            const float w = kSsimW[t];
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m] = fmaf(w, s_in[m][r][q + t], acc[m]);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) s_h[m][r][q] = acc[m];
    }
    __syncthreads();
    // vertical pass: this thread's pixel
    float v[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) v[m] = 0.0f;
#pragma unroll
    for (int t = 0; t < kSsimTaps; ++t) {
        const float w = kSsimW[t];
#pragma unroll
        for (int m = 0; m < NM; ++m) v[m] = fmaf(w, s_h[m][ty + t][tx], v[m]);
    }
    const int gx = x0 + tx, gy = y0 + ty;
    const bool inside = gx < a.W && gy < a.H;
    if constexpr (BWD) {
        if (inside) {
            const float xv = px[gy * a.sx[2] + gx * a.sx[3]], yv = py[gy * a.sy[2] + gx * a.sy[3]];
            const float g = a.grad_dev ? a.grad_scale * *a.grad_dev : a.grad_scale;
            a.gx[b * a.sgx[0] + c * a.sgx[1] + gy * a.sgx[2] + gx * a.sgx[3]] = g * (v[0] + 2.0f * xv * v[1] + yv * v[2]);
        }
    } else {
        constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1 = v[0], mu2 = v[1];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = v[2] - mu1_sq, s2 = v[3] - mu2_sq, s12 = v[4] - mu12;
        const float A = 2.0f * mu12 + C1, Bq = 2.0f * s12 + C2, Cq = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
        const float inv = 1.0f / (Cq * D);
        const float ssim = A * Bq * inv;
        if (a.dmaps && inside) {
            // ssim = A B / (C D) with A(mu1), B(s12), C(mu1), D(s1); the moments themselves depend on mu1 through
            // s1 = E[xx] - mu1^2 and s12 = E[xy] - mu1 mu2: folded in here so that the backward convolves three maps only
            const float d_mu1 = (2.0f * mu2 * Bq * inv - ssim * 2.0f * mu1 / Cq)   // through A and C
                              + (-2.0f * mu1) * (-ssim / D)                          // through s1 = E[xx] - mu1^2
                              + (-mu2) * (2.0f * A * inv);                           // through s12 = E[xy] - mu1 mu2
            float *d = a.dmaps + (((size_t)plane * a.H + gy) * a.W + gx) * 3;
            d[0] = d_mu1;
            d[1] = -ssim / D;        // d ssim / d E[xx]
            d[2] = 2.0f * A * inv;   // d ssim / d E[xy]
        }
        // sum of the map over the tile (image pixels only)
        float sum = inside ? ssim : 0.0f;
        sum       = wave_sum(sum);
        __shared__ float s_red[kSsimTw * kSsimTh / 64];
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int w = 0; w < kSsimTw * kSsimTh / 64; ++w) t += s_red[w];
            a.partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
        }
    }
}

} // namespace gsx

using namespace gsx;

extern "C" int64_t gsx_ssim_blocks(uint32_t B, uint32_t C, uint32_t H, uint32_t W)
{
    return (int64_t)B * C * ((H + kSsimTh - 1) / kSsimTh) * ((W + kSsimTw - 1) / kSsimTw);
}

extern "C" int gsx_ssim_fwd(const float *img1, const int64_t *strides1, const float *img2, const int64_t *strides2, uint32_t B,
                            uint32_t C, uint32_t H, uint32_t W, float *partial_sums, float *dmaps, void *stream)
{
    if ((int64_t)B * C * H * W == 0) return GSX_OK;
    GSX_REQUIRE(img1 && img2 && strides1 && strides2 && partial_sums, "gsx_ssim_fwd: null argument");
    GSX_REQUIRE((int64_t)B * C <= 65535, "gsx_ssim_fwd: more than 65535 (batch, channel) planes");
    SsimArgs a{};
    a.x = img1; a.y = img2; a.B = (int32_t)B; a.C = (int32_t)C; a.H = (int32_t)H; a.W = (int32_t)W; a.partial = partial_sums; a.dmaps = dmaps;
    for (int i = 0; i < 4; ++i) { a.sx[i] = strides1[i]; a.sy[i] = strides2[i]; }
    const dim3 grid((W + kSsimTw - 1) / kSsimTw, (H + kSsimTh - 1) / kSsimTh, B * C);
    ssim_kernel<false><<<grid, dim3(kSsimTw * kSsimTh), 0, (hipStream_t)stream>>>(a);
    return check_launch("ssim_fwd");
}

extern "C" int gsx_ssim_bwd(const float *img1, const int64_t *strides1, const float *img2, const int64_t *strides2, uint32_t B,
                            uint32_t C, uint32_t H, uint32_t W, const float *dmaps, float grad_scale,
                            const float *grad_scale_device, float *v_img1,
                            const int64_t *strides_v, void *stream)
{
    if ((int64_t)B * C * H * W == 0) return GSX_OK;
    GSX_REQUIRE(img1 && img2 && strides1 && strides2 && dmaps && v_img1 && strides_v, "gsx_ssim_bwd: null argument");
    GSX_REQUIRE((int64_t)B * C <= 65535, "gsx_ssim_bwd: more than 65535 (batch, channel) planes");
    SsimArgs a{};
    a.x = img1; a.y = img2; a.B = (int32_t)B; a.C = (int32_t)C; a.H = (int32_t)H; a.W = (int32_t)W; a.dmaps = const_cast<float *>(dmaps);
    a.grad_scale = grad_scale; a.grad_dev = grad_scale_device; a.gx = v_img1;
    for (int i = 0; i < 4; ++i) { a.sx[i] = strides1[i]; a.sy[i] = strides2[i]; a.sgx[i] = strides_v[i]; }
    const dim3 grid((W + kSsimTw - 1) / kSsimTw, (H + kSsimTh - 1) / kSsimTh, B * C);
    ssim_kernel<true><<<grid, dim3(kSsimTw * kSsimTh), 0, (hipStream_t)stream>>>(a);
    return check_launch("ssim_bwd");
}
