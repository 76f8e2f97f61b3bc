// Spherical-harmonics colours of the rasterizer (degree 0..3), forward and backward, one Gaussian per thread.
// Replaces computeColorFromSH of the reference (cuda_rasterizer/forward.cu:20-71, backward.cu:20-139; constants
// auxiliary.h:22-39).  AnimatableGaussians itself always renders with colors_precomp (sh_degree = 0, shs = None);
// this path exists for the API surface of diff_gaussian_rasterization_depth_alpha.
//
//   colour_ch = max(0, sum_i coef_i(dir) * sh[i][ch] + 0.5),  dir = (mean - campos) / |mean - campos|
// accumulated in coefficient order (the order the reference's vec3 expressions evaluate in); coef_i is also
// d colour / d sh_i.  Included by the two preprocess translation units, which are compiled without FMA contraction.
#pragma once

namespace ag {

__device__ constexpr float kSh0 = 0.28209479177387814f;
__device__ constexpr float kSh1 = 0.4886025119029199f;
__device__ constexpr float kSh2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                       0.5462742152960396f };
__device__ constexpr float kSh3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                       -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

struct ShDir {
    float ox, oy, oz;   // mean - campos
    float x, y, z;      // normalised
};

__device__ __forceinline__ ShDir sh_direction(float mx, float my, float mz, const float* __restrict__ campos)
{
    ShDir d;
    d.ox = mx - campos[0]; d.oy = my - campos[1]; d.oz = mz - campos[2];
    const float len = sqrtf(d.ox * d.ox + d.oy * d.oy + d.oz * d.oz);
    d.x = d.ox / len; d.y = d.oy / len; d.z = d.oz / len;
    return d;
}

// basis values for the unit direction; c[i] valid for i < (deg + 1)^2
__device__ __forceinline__ void sh_coefficients(int deg, float x, float y, float z, float (&c)[16])
{
    c[0] = kSh0;
    if (deg > 0) {
        c[1] = -(kSh1 * y);
        c[2] = kSh1 * z;
        c[3] = -(kSh1 * x);
    }
    if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        c[4] = kSh2[0] * xy;
        c[5] = kSh2[1] * yz;
        c[6] = kSh2[2] * (2.0f * zz - xx - yy);
        c[7] = kSh2[3] * xz;
        c[8] = kSh2[4] * (xx - yy);
        if (deg > 2) {
            c[9] = kSh3[0] * y * (3.0f * xx - yy);
            c[10] = kSh3[1] * xy * z;
            c[11] = kSh3[2] * y * (4.0f * zz - xx - yy);
            c[12] = kSh3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            c[13] = kSh3[4] * x * (4.0f * zz - xx - yy);
            c[14] = kSh3[5] * z * (xx - yy);
            c[15] = kSh3[6] * x * (xx - 3.0f * yy);
        }
    }
}

// rgb[3] (clamped at 0) and the three clamp flags of one Gaussian; sh points at its [M][3] coefficients
__device__ __forceinline__ void sh_colour(int deg, const ShDir& d, const float* __restrict__ sh, float (&rgb)[3], uint8_t (&clamped)[3])
{
    float c[16];
    sh_coefficients(deg, d.x, d.y, d.z, c);
    const int n = (deg + 1) * (deg + 1);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float r = c[0] * sh[ch];
        for (int i = 1; i < n; i++) r = r + c[i] * sh[3 * i + ch];
        r += 0.5f;
        clamped[ch] = r < 0.f;
        rgb[ch] = r > 0.0f ? r : 0.0f;
    }
}

// Backward of one Gaussian: writes dL_dsh[i][ch] for i < n (the caller zeroes the rest) and returns the gradient that
// reaches the mean through the view direction in gm[3].  g[] = dL/dcolour with clamped channels already zeroed.
__device__ __forceinline__ void sh_backward(int deg, const ShDir& d, const float* __restrict__ sh, const float (&g)[3],
                                            float* __restrict__ dL_dsh, float (&gm)[3])
{
    float c[16];
    sh_coefficients(deg, d.x, d.y, d.z, c);
    const int n = (deg + 1) * (deg + 1);
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dL_dsh[3 * i + ch] = c[i] * g[ch];

    const float x = d.x, y = d.y, z = d.z;
    float ddx[3] = { 0.f, 0.f, 0.f }, ddy[3] = { 0.f, 0.f, 0.f }, ddz[3] = { 0.f, 0.f, 0.f };
#define AG_SH(i) sh[3 * (i) + ch]
    if (deg > 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { ddx[ch] = -kSh1 * AG_SH(3); ddy[ch] = -kSh1 * AG_SH(1); ddz[ch] = kSh1 * AG_SH(2); }
    }
    if (deg > 1) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            ddx[ch] += kSh2[0] * y * AG_SH(4) + kSh2[2] * 2.f * -x * AG_SH(6) + kSh2[3] * z * AG_SH(7) + kSh2[4] * 2.f * x * AG_SH(8);
            ddy[ch] += kSh2[0] * x * AG_SH(4) + kSh2[1] * z * AG_SH(5) + kSh2[2] * 2.f * -y * AG_SH(6) + kSh2[4] * 2.f * -y * AG_SH(8);
            ddz[ch] += kSh2[1] * y * AG_SH(5) + kSh2[2] * 2.f * 2.f * z * AG_SH(6) + kSh2[3] * x * AG_SH(7);
        }
    }
    if (deg > 2) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            ddx[ch] += (kSh3[0] * AG_SH(9) * 3.f * 2.f * xy + kSh3[1] * AG_SH(10) * yz + kSh3[2] * AG_SH(11) * -2.f * xy +
                        kSh3[3] * AG_SH(12) * -3.f * 2.f * xz + kSh3[4] * AG_SH(13) * (-3.f * xx + 4.f * zz - yy) +
                        kSh3[5] * AG_SH(14) * 2.f * xz + kSh3[6] * AG_SH(15) * 3.f * (xx - yy));
            ddy[ch] += (kSh3[0] * AG_SH(9) * 3.f * (xx - yy) + kSh3[1] * AG_SH(10) * xz +
                        kSh3[2] * AG_SH(11) * (-3.f * yy + 4.f * zz - xx) + kSh3[3] * AG_SH(12) * -3.f * 2.f * yz +
                        kSh3[4] * AG_SH(13) * -2.f * xy + kSh3[5] * AG_SH(14) * -2.f * yz + kSh3[6] * AG_SH(15) * -3.f * 2.f * xy);
            ddz[ch] += (kSh3[1] * AG_SH(10) * xy + kSh3[2] * AG_SH(11) * 4.f * 2.f * yz +
                        kSh3[3] * AG_SH(12) * 3.f * (2.f * zz - xx - yy) + kSh3[4] * AG_SH(13) * 4.f * 2.f * xz +
                        kSh3[5] * AG_SH(14) * (xx - yy));
        }
    }
#undef AG_SH
    const float gdx = ddx[0] * g[0] + ddx[1] * g[1] + ddx[2] * g[2];
    const float gdy = ddy[0] * g[0] + ddy[1] * g[1] + ddy[2] * g[2];
    const float gdz = ddz[0] * g[0] + ddz[1] * g[1] + ddz[2] * g[2];
    // through dir = v / |v| (auxiliary.h:107-117)
    const float s2 = d.ox * d.ox + d.oy * d.oy + d.oz * d.oz;
    const float inv = 1.0f / sqrtf(s2 * s2 * s2);
    gm[0] = ((+s2 - d.ox * d.ox) * gdx - d.oy * d.ox * gdy - d.oz * d.ox * gdz) * inv;
    gm[1] = (-d.ox * d.oy * gdx + (s2 - d.oy * d.oy) * gdy - d.oz * d.oy * gdz) * inv;
    gm[2] = (-d.ox * d.oz * gdx - d.oy * d.oz * gdy + (s2 - d.oz * d.oz) * gdz) * inv;
}

}  // namespace ag
