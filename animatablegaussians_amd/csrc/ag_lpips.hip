// LPIPS-VGG16 helpers (gfx950): 2x2 max-pool and the fused per-level perceptual distance.  See include/ag_lpips.h.
// Both are streaming, HBM/L2-bound: one thread per output pixel, lanes along pixels (coalesced), channel loops strided by
// the plane size.  The level kernel walks the channels twice (norms, then the weighted squared difference) -- the second
// pass hits L2 -- and reduces over pixels with a wave shuffle + one atomic per workgroup.
#include "ag_common.h"
#include "../../include/ag_lpips.h"

namespace ag {

__global__ void __launch_bounds__(256) maxpool2x2_forward_kernel(float* __restrict__ y, uint8_t* __restrict__ arg,
                                                                const float* __restrict__ x, int C, int H, int W)
{
    const int oh = H / 2, ow = W / 2;
    const long long total = (long long)C * oh * ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const long long r = i / ow;
        const int oy = (int)(r % oh), c = (int)(r / oh);
        const float* src = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
        const float v0 = src[0], v1 = src[1], v2 = src[W], v3 = src[W + 1];
        float m = v0;
        int a = 0;
        // strict > keeps the first maximum; NaN propagates like torch (a NaN candidate replaces the running value)
        if (v1 > m || v1 != v1) { m = v1; a = 1; }
        if (v2 > m || v2 != v2) { m = v2; a = 2; }
        if (v3 > m || v3 != v3) { m = v3; a = 3; }
        y[i] = m;
        if (arg) arg[i] = (uint8_t)a;
    }
}

__global__ void __launch_bounds__(256) maxpool2x2_backward_kernel(float* __restrict__ gx, const float* __restrict__ gy,
                                                                 const uint8_t* __restrict__ arg, int C, int H, int W)
{
    const int oh = H / 2, ow = W / 2;
    // one thread per INPUT pixel so that every element of gx (odd trailing row / column included) is written exactly once
    const long long total = (long long)C * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ix = (int)(i % W);
        const long long r = i / W;
        const int iy = (int)(r % H), c = (int)(r / H);
        const int ox = ix >> 1, oy = iy >> 1;
        float v = 0.f;
        if (ox < ow && oy < oh) {
            const size_t o = ((size_t)c * oh + oy) * ow + ox;
            if (arg[o] == (uint8_t)(2 * (iy & 1) + (ix & 1))) v = gy[o];
        }
        gx[i] = v;
    }
}

constexpr float kLpipsEps = 1e-10f;

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) lpips_level_forward_kernel(float* __restrict__ out, const float* __restrict__ f0,
                                                                 const float* __restrict__ f1, const float* __restrict__ lin, int C,
                                                                 int HW)
{
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p], b = f1[(size_t)c * HW + p];
            s0 += a * a;
            s1 += b * b;
        }
        const float i0 = 1.0f / (sqrtf(s0 + kLpipsEps) + kLpipsEps), i1 = 1.0f / (sqrtf(s1 + kLpipsEps) + kLpipsEps);
        float v = 0.f;
        for (int c = 0; c < C; c++) {
            const float d = f0[(size_t)c * HW + p] * i0 - f1[(size_t)c * HW + p] * i1;
            v += lin[c] * (d * d);
        }
        acc += v;
    }
    acc = wave_sum64(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)HW);
}

// a_c = f0_c * i0 with i0 = 1 / (n0 + eps), n0 = sqrt(S + eps), S = sum f0^2:
//   d a_c / d f0_k = delta_ck * i0 - f0_c * f0_k * i0^2 / n0
//   dL/df0_k = i0 * (G_k - f0_k * (i0 / n0) * sum_c G_c f0_c),   G_c = g * 2 lin_c (a_c - b_c) / HW
__global__ void __launch_bounds__(256) lpips_level_backward_kernel(float* __restrict__ gf0, const float* __restrict__ gout,
                                                                  const float* __restrict__ f0, const float* __restrict__ f1,
                                                                  const float* __restrict__ lin, int C, int HW)
{
    const float g = gout[0] * 2.0f / (float)HW;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p], b = f1[(size_t)c * HW + p];
            s0 += a * a;
            s1 += b * b;
        }
        const float n0 = sqrtf(s0 + kLpipsEps);
        const float i0 = 1.0f / (n0 + kLpipsEps), i1 = 1.0f / (sqrtf(s1 + kLpipsEps) + kLpipsEps);
        float t = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p];
            t += lin[c] * (a * i0 - f1[(size_t)c * HW + p] * i1) * a;
        }
        const float k = t * i0 / n0;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p];
            const float G = lin[c] * (a * i0 - f1[(size_t)c * HW + p] * i1);
            gf0[(size_t)c * HW + p] = g * i0 * (G - a * k);
        }
    }
}

static int grid_for(long long total, int cap)
{
    long long b = (total + 255) / 256;
    if (b > cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

}  // namespace ag

using namespace ag;

extern "C" {

int ag_maxpool2x2_forward(float* y, uint8_t* arg, const float* x, int32_t C, int32_t H, int32_t W, void* stream)
{
    if (C < 0 || H < 2 || W < 2) { set_error("maxpool2x2: need H, W >= 2"); return AG_ERR_INVALID_ARGUMENT; }
    if (C == 0) return AG_OK;
    if (!y || !x) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)C * (H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool2x2_forward_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, arg,
                       x, C, H, W);
    return check_hip(hipGetLastError(), "maxpool2x2_forward_kernel");
}

int ag_maxpool2x2_backward(float* gx, const float* gy, const uint8_t* arg, int32_t C, int32_t H, int32_t W, void* stream)
{
    if (C < 0 || H < 2 || W < 2) { set_error("maxpool2x2: need H, W >= 2"); return AG_ERR_INVALID_ARGUMENT; }
    if (C == 0) return AG_OK;
    if (!gx || !gy || !arg) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)C * H * W;
    hipLaunchKernelGGL(maxpool2x2_backward_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gx,
                       gy, arg, C, H, W);
    return check_hip(hipGetLastError(), "maxpool2x2_backward_kernel");
}

int ag_lpips_level_forward(float* out, const float* f0, const float* f1, const float* lin, int32_t C, int32_t HW, void* stream)
{
    if (C <= 0 || HW <= 0 || !out || !f0 || !f1 || !lin) { set_error("bad lpips_level arguments"); return AG_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(lpips_level_forward_kernel, dim3(grid_for(HW, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, f0,
                       f1, lin, C, HW);
    return check_hip(hipGetLastError(), "lpips_level_forward_kernel");
}

int ag_lpips_level_backward(float* gf0, const float* gout, const float* f0, const float* f1, const float* lin, int32_t C, int32_t HW,
                            void* stream)
{
    if (C <= 0 || HW <= 0 || !gf0 || !gout || !f0 || !f1 || !lin) { set_error("bad lpips_level arguments"); return AG_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(lpips_level_backward_kernel, dim3(grid_for(HW, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gf0,
                       gout, f0, f1, lin, C, HW);
    return check_hip(hipGetLastError(), "lpips_level_backward_kernel");
}

}  // extern "C"
