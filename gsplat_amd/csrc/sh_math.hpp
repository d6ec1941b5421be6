// Real spherical-harmonics basis (degree <= 4) shared by the SH kernels (sh.hip, sh_band.hip).
// Basis: Sloan, "Efficient Spherical Harmonic Evaluation" (JCGT 2013) polynomial forms; the constants are the published
// ones (reference gsplat/cuda/csrc/SphericalHarmonicsCUDA.cu:48-146; torch restatement gsplat/cuda/_torch_impl.py:968-1067).
#pragma once
#include "common.hpp"

namespace gsx {

constexpr int kMaxBases = 25;

// Y[0..nb) for unit direction (x,y,z); if GRAD also dY/dx, dY/dy, dY/dz (x,y,z treated as free).
template <bool GRAD>
__device__ __forceinline__ void sh_bases(int degree, float x, float y, float z, float *Y, float *Yx, float *Yy, float *Yz)
{
    Y[0] = 0.2820947917738781f;
    if (GRAD) { Yx[0] = Yy[0] = Yz[0] = 0.0f; }
    if (degree < 1) return;
    const float c1 = 0.48860251190292f;
    Y[1] = -c1 * y; Y[2] = c1 * z; Y[3] = -c1 * x;
    if (GRAD) {
        Yx[1] = 0.f; Yy[1] = -c1; Yz[1] = 0.f;
        Yx[2] = 0.f; Yy[2] = 0.f; Yz[2] = c1;
        Yx[3] = -c1; Yy[3] = 0.f; Yz[3] = 0.f;
    }
    if (degree < 2) return;
    const float z2 = z * z;
    const float C1 = x * x - y * y, S1 = 2.0f * x * y; // cos/sin(1*phi) * r^1... (Sloan's fC1,fS1)
    const float C1x = 2.0f * x, C1y = -2.0f * y, S1x = 2.0f * y, S1y = 2.0f * x;
    {
        const float b = -1.092548430592079f * z, bz = -1.092548430592079f;
        const float a = 0.5462742152960395f;
        Y[4] = a * S1; Y[5] = b * y; Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f; Y[7] = b * x; Y[8] = a * C1;
        if (GRAD) {
            Yx[4] = a * S1x; Yy[4] = a * S1y; Yz[4] = 0.f;
            Yx[5] = 0.f; Yy[5] = b; Yz[5] = bz * y;
            Yx[6] = 0.f; Yy[6] = 0.f; Yz[6] = 2.0f * 0.9461746957575601f * z;
            Yx[7] = b; Yy[7] = 0.f; Yz[7] = bz * x;
            Yx[8] = a * C1x; Yy[8] = a * C1y; Yz[8] = 0.f;
        }
    }
    if (degree < 3) return;
    const float C2 = x * C1 - y * S1, S2 = x * S1 + y * C1;
    const float C2x = 3.0f * C1, C2y = -3.0f * S1, S2x = 3.0f * S1, S2y = 3.0f * C1;
    {
        const float c = -2.285228997322329f * z2 + 0.4570457994644658f, cz = -2.0f * 2.285228997322329f * z;
        const float b = 1.445305721320277f * z, bz = 1.445305721320277f;
        const float a = -0.5900435899266435f;
        Y[9] = a * S2; Y[10] = b * S1; Y[11] = c * y;
        Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
        Y[13] = c * x; Y[14] = b * C1; Y[15] = a * C2;
        if (GRAD) {
            Yx[9] = a * S2x; Yy[9] = a * S2y; Yz[9] = 0.f;
            Yx[10] = b * S1x; Yy[10] = b * S1y; Yz[10] = bz * S1;
            Yx[11] = 0.f; Yy[11] = c; Yz[11] = cz * y;
            Yx[12] = 0.f; Yy[12] = 0.f; Yz[12] = 3.0f * 1.865881662950577f * z2 - 1.119528997770346f;
            Yx[13] = c; Yy[13] = 0.f; Yz[13] = cz * x;
            Yx[14] = b * C1x; Yy[14] = b * C1y; Yz[14] = bz * C1;
            Yx[15] = a * C2x; Yy[15] = a * C2y; Yz[15] = 0.f;
        }
    }
    if (degree < 4) return;
    const float C3 = x * C2 - y * S2, S3 = x * S2 + y * C2;
    const float C3x = 4.0f * C2, C3y = -4.0f * S2, S3x = 4.0f * S2, S3y = 4.0f * C2;
    {
        const float d = z * (-4.683325804901025f * z2 + 2.007139630671868f);
        const float dz = -3.0f * 4.683325804901025f * z2 + 2.007139630671868f;
        const float c = 3.31161143515146f * z2 - 0.47308734787878f, cz = 2.0f * 3.31161143515146f * z;
        const float b = -1.770130769779931f * z, bz = -1.770130769779931f;
        const float a = 0.6258357354491763f;
        const float p12 = 1.865881662950577f * z2 - 1.119528997770346f;     // Y12 / z
        const float p6  = 0.9461746957575601f * z2 - 0.3153915652525201f;   // Y6
        Y[16] = a * S3; Y[17] = b * S2; Y[18] = c * S1; Y[19] = d * y;
        Y[20] = 1.984313483298443f * z2 * p12 - 1.006230589874905f * p6;
        Y[21] = d * x; Y[22] = c * C1; Y[23] = b * C2; Y[24] = a * C3;
        if (GRAD) {
            Yx[16] = a * S3x; Yy[16] = a * S3y; Yz[16] = 0.f;
            Yx[17] = b * S2x; Yy[17] = b * S2y; Yz[17] = bz * S2;
            Yx[18] = c * S1x; Yy[18] = c * S1y; Yz[18] = cz * S1;
            Yx[19] = 0.f; Yy[19] = d; Yz[19] = dz * y;
            Yx[20] = 0.f; Yy[20] = 0.f;
            Yz[20] = 1.984313483298443f * (2.0f * z * p12 + z2 * 2.0f * 1.865881662950577f * z)
                   - 1.006230589874905f * 2.0f * 0.9461746957575601f * z;
            Yx[21] = d; Yy[21] = 0.f; Yz[21] = dz * x;
            Yx[22] = c * C1x; Yy[22] = c * C1y; Yz[22] = cz * C1;
            Yx[23] = b * C2x; Yy[23] = b * C2y; Yz[23] = bz * C2;
            Yx[24] = a * C3x; Yy[24] = a * C3y; Yz[24] = 0.f;
        }
    }
}

} // namespace gsx
