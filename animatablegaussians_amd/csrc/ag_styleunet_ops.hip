// fused bias + activation and upfirdn2d (FIR resampling) for the StyleUNet (gfx950).
//
// Replaces network/styleunet/fused_bias_act_kernel.cu:18-104 and upfirdn2d_kernel.cu:49-368 (see
// include/ag_styleunet.h).  Both are pure streaming operators, HBM-bound:
//   fused_bias_act: 4 B in (+4 B ref) + 4 B out per element; float4-vectorised grid-stride loop;
//   upfirdn2d     : one output element per thread, x-fastest so a wave writes 256 contiguous bytes and its <= 16 taps
//                   read overlapping contiguous spans that stay in L1/L2 (each input element feeds <= kh*kw/(up^2)
//                   outputs); the <= 4x4 FIR taps sit in LDS.
#include <cstring>
#include "ag_common.h"
#include "../../include/ag_styleunet.h"

namespace ag {

__device__ __forceinline__ float bias_act_one(float x, float ref, int mode, float alpha)
{
    switch (mode) {
    case 12: case 32: return 0.0f;
    case 30: return (x > 0.0f) ? x : x * alpha;
    case 31: return (ref > 0.0f) ? x : x * alpha;
    default: return x;   // 10, 11 and the reference's `default:`
    }
}

__global__ void __launch_bounds__(256) fused_bias_act_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                            const float* __restrict__ bias, const float* __restrict__ ref,
                                                            int mode, float alpha, float scale, long long size_x,
                                                            long long step_b, int size_b, int vec_ok)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec_ok) {
        // step_b % 4 == 0 and 16-byte aligned pointers: the four lanes of a float4 share one bias element
        const long long n4 = size_x >> 2;
        for (; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const float b = bias ? bias[((i << 2) / step_b) % size_b] : 0.0f;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4*>(ref)[i];
            float4 o;
            o.x = bias_act_one(v.x + b, r.x, mode, alpha) * scale;
            o.y = bias_act_one(v.y + b, r.y, mode, alpha) * scale;
            o.z = bias_act_one(v.z + b, r.z, mode, alpha) * scale;
            o.w = bias_act_one(v.w + b, r.w, mode, alpha) * scale;
            reinterpret_cast<float4*>(out)[i] = o;
        }
        // tail (size_x % 4) handled by the scalar loop below on the remaining elements
        i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x;
    }
    for (; i < size_x; i += stride) {
        const float b = bias ? bias[(i / step_b) % size_b] : 0.0f;
        out[i] = bias_act_one(x[i] + b, ref ? ref[i] : 0.0f, mode, alpha) * scale;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// StyledConv tail in one pass (dual_styleunet.py:598-604): y = lrelu(x + nw * noise[pix] + bias[c], slope) * scale
// and its backward with the two reductions folded in: gx = gy * (y > 0 ? 1 : slope) * scale,
// gbias[c] += sum_pix gx, gnw += sum_{c,pix} gx * noise[pix].  One workgroup owns a run of pixels of ONE channel, so
// the bias is a scalar and both sums are a workgroup reduction + one atomic each.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kNbaChunk = 4096;   // pixels per workgroup (16 per thread)

__global__ void __launch_bounds__(256) noise_bias_act_forward_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                                    const float* __restrict__ noise, const float* __restrict__ nw,
                                                                    const float* __restrict__ bias, int C, int HW, int chunks,
                                                                    float slope, float scale)
{
    const int c = blockIdx.x / chunks, p0 = (blockIdx.x - c * chunks) * kNbaChunk;
    const float b = bias ? bias[c] : 0.f, w = noise ? nw[0] : 0.f;
    const float* xr = x + (size_t)c * HW;
    float* yr = y + (size_t)c * HW;
    const int pend = min(HW, p0 + kNbaChunk);
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < pend; p += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xr + p);
            float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) n = *reinterpret_cast<const float4*>(noise + p);
            float4 o;
            float t;
            t = v.x + w * n.x + b; o.x = (t > 0.f ? t : t * slope) * scale;
            t = v.y + w * n.y + b; o.y = (t > 0.f ? t : t * slope) * scale;
            t = v.z + w * n.z + b; o.z = (t > 0.f ? t : t * slope) * scale;
            t = v.w + w * n.w + b; o.w = (t > 0.f ? t : t * slope) * scale;
            *reinterpret_cast<float4*>(yr + p) = o;
        }
    } else {
        for (int p = p0 + threadIdx.x; p < pend; p += 256) {
            const float t = xr[p] + (noise ? w * noise[p] : 0.f) + b;
            yr[p] = (t > 0.f ? t : t * slope) * scale;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) noise_bias_act_backward_kernel(float* __restrict__ gx, const float* __restrict__ gy,
                                                                     const float* __restrict__ y, const float* __restrict__ noise,
                                                                     float* __restrict__ gbias, float* __restrict__ gnw, int C,
                                                                     int HW, int chunks, float slope, float scale)
{
    __shared__ float s_red[2][4];
    const int c = blockIdx.x / chunks, p0 = (blockIdx.x - c * chunks) * kNbaChunk;
    const float* gyr = gy + (size_t)c * HW;
    const float* yr = y + (size_t)c * HW;
    float* gxr = gx + (size_t)c * HW;
    const int pend = min(HW, p0 + kNbaChunk);
    float sb = 0.f, sn = 0.f;
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < pend; p += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(gyr + p);
            const float4 o = *reinterpret_cast<const float4*>(yr + p);
            float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) n = *reinterpret_cast<const float4*>(noise + p);
            float4 r;
            r.x = g.x * (o.x > 0.f ? 1.f : slope) * scale;
            r.y = g.y * (o.y > 0.f ? 1.f : slope) * scale;
            r.z = g.z * (o.z > 0.f ? 1.f : slope) * scale;
            r.w = g.w * (o.w > 0.f ? 1.f : slope) * scale;
            *reinterpret_cast<float4*>(gxr + p) = r;
            sb += (r.x + r.y) + (r.z + r.w);
            sn += (r.x * n.x + r.y * n.y) + (r.z * n.z + r.w * n.w);
        }
    } else {
        for (int p = p0 + threadIdx.x; p < pend; p += 256) {
            const float r = gyr[p] * (yr[p] > 0.f ? 1.f : slope) * scale;
            gxr[p] = r;
            sb += r;
            if (noise) sn += r * noise[p];
        }
    }
    sb = wave_sum(sb);
    sn = wave_sum(sn);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[0][wave] = sb; s_red[1][wave] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gbias) atomicAdd(gbias + c, (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
        if (gnw) atomicAdd(gnw, (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight modulation + demodulation of ModulatedConv2d's fused branch (dual_styleunet.py:254-259), one workgroup per
// output channel:  w'[co][ci][k] = (scale * W[co][ci][k]) * style[ci];  d[co] = rsqrt(sum w'^2 + 1e-8);  out = w' * d.
// `transposed` writes out[ci][co][k] (the layout conv_transpose2d takes, :268-272).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* s_red)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) modulate_weight_forward_kernel(float* __restrict__ out, float* __restrict__ dcoef,
                                                                     const float* __restrict__ W, const float* __restrict__ style,
                                                                     float scale, int demod, int Co, int Ci, int K2, int transposed)
{
    __shared__ float s_red[4];
    const int co = blockIdx.x, n = Ci * K2;
    const float* w = W + (size_t)co * n;
    float q = 0.f;
    if (demod) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float v = (scale * w[i]) * style[i / K2];
            q += v * v;
        }
        q = block_sum_256(q, s_red);
    }
    const float d = demod ? rsqrtf(q + 1e-8f) : 1.0f;
    if (threadIdx.x == 0 && dcoef) dcoef[co] = d;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / K2, k = i - ci * K2;
        const float v = (scale * w[i]) * style[ci];
        const size_t o = transposed ? ((size_t)ci * Co + co) * K2 + k : (size_t)co * n + i;
        out[o] = demod ? v * d : v;
    }
}

// g = dL/dout.  dL/dw' = g * d - d^3 * (sum g w') * w'   (g when there is no demodulation);
// dW = dL/dw' * scale * style[ci];  dstyle[ci] += sum_{co, k} dL/dw' * scale * W     (float atomics over co)
__global__ void __launch_bounds__(256) modulate_weight_backward_kernel(float* __restrict__ dW, float* __restrict__ dstyle,
                                                                      const float* __restrict__ g, const float* __restrict__ W,
                                                                      const float* __restrict__ style, const float* __restrict__ dcoef,
                                                                      float scale, int demod, int Co, int Ci, int K2, int transposed)
{
    __shared__ float s_red[4];
    const int co = blockIdx.x, n = Ci * K2;
    const float* w = W + (size_t)co * n;
    float gw = 0.f;
    const float d = demod ? dcoef[co] : 1.0f;
    if (demod) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int ci = i / K2, k = i - ci * K2;
            const size_t o = transposed ? ((size_t)ci * Co + co) * K2 + k : (size_t)co * n + i;
            gw += g[o] * ((scale * w[i]) * style[ci]);
        }
        gw = block_sum_256(gw, s_red);
    }
    const float c3 = d * d * d * gw;
    // one thread per input channel keeps the K2 taps of that channel together: a single atomic per (co, ci)
    for (int ci = threadIdx.x; ci < Ci; ci += 256) {
        const float sc = scale * style[ci];
        float ds = 0.f;
        for (int k = 0; k < K2; k++) {
            const int i = ci * K2 + k;
            const size_t o = transposed ? ((size_t)ci * Co + co) * K2 + k : (size_t)co * n + i;
            const float sw = scale * w[i];
            const float gp = demod ? g[o] * d - c3 * (sw * style[ci]) : g[o];
            dW[(size_t)co * n + i] = gp * sc;
            ds += gp * sw;
        }
        atomicAdd(dstyle + ci, ds);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 2x2 block transforms: the Haar analysis / synthesis of the wavelet skip path (dual_styleunet.py:374-425) as ONE pass
// instead of 4 upfirdn2d calls + 3 additions / a concatenation.
//   split: y[b][c][i][j] = sum_p M[b][p] * x[c][2i + p/2][2j + p%2]          x [C][2h][2w] -> y [4][C][h][w]
//   merge: x[c][2i + p/2][2j + p%2] = sum_b M[p][b] * y[b][c][i][j]          y [4][C][h][w] -> x [C][2h][2w]
// ------------------------------------------------------------------------------------------------------------------
struct Block2x2 { float m[16]; };

__global__ void __launch_bounds__(256) block2x2_split_kernel(float* __restrict__ y, const float* __restrict__ x, Block2x2 M, int C,
                                                            int h, int w)
{
    const long long total = (long long)C * h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int jx = (int)(i % w);
        const long long r = i / w;
        const int iy = (int)(r % h), c = (int)(r / h);
        const float* src = x + ((size_t)c * 2 * h + 2 * iy) * (2 * w) + 2 * jx;
        const float2 t = *reinterpret_cast<const float2*>(src), b = *reinterpret_cast<const float2*>(src + 2 * w);
        const float v[4] = { t.x, t.y, b.x, b.y };
#pragma unroll
        for (int band = 0; band < 4; band++)
            y[(size_t)band * total + i] = M.m[4 * band + 0] * v[0] + M.m[4 * band + 1] * v[1] + M.m[4 * band + 2] * v[2] + M.m[4 * band + 3] * v[3];
    }
}

__global__ void __launch_bounds__(256) block2x2_merge_kernel(float* __restrict__ x, const float* __restrict__ y, Block2x2 M, int C,
                                                            int h, int w)
{
    const long long total = (long long)C * h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int jx = (int)(i % w);
        const long long r = i / w;
        const int iy = (int)(r % h), c = (int)(r / h);
        float v[4];
#pragma unroll
        for (int band = 0; band < 4; band++) v[band] = y[(size_t)band * total + i];
        float o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = M.m[4 * p + 0] * v[0] + M.m[4 * p + 1] * v[1] + M.m[4 * p + 2] * v[2] + M.m[4 * p + 3] * v[3];
        float* dst = x + ((size_t)c * 2 * h + 2 * iy) * (2 * w) + 2 * jx;
        *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        *reinterpret_cast<float2*>(dst + 2 * w) = make_float2(o[2], o[3]);
    }
}

// ToRGB's wavelet-domain skip (dual_styleunet.py:607-633: InverseHaarTransform -> Upsample -> HaarTransform, then added to the layer's output)
// as ONE linear map [4C, h, w] -> [4C, 2h, 2w].  Each of the three stages is local (a 2 x 2 block transform, a 4-tap FIR after zero
// stuffing, a 2 x 2 block transform) and separable, so the 4 x (2 x 2) outputs of site (i, j) -- sub-band s' = (u'y, u'x), parity (py, px) --
// depend on the 4 sub-bands s = (uy, ux) of the 3 x 3 sites around it through two 1-D maps:
//   out[(u'y, u'x)][2i+py][2j+px] (+)= sum_{uy,a} wy[u'y][py][uy][a] * sum_{ux,b} wx[u'x][px][ux][b] * skip[(uy, ux)][i+a-1][j+b-1]
// (zero outside: the FIR's zero padding; sub-band index = uy + 2 ux for the reference's order ll, lh, hl, hh).  The 2 x 24 coefficients
// are derived on the host from the Haar matrices and the FIR kernel (styleunet_ops.py) and passed by value (48 scalar registers).
// One pass over 4 + 32 floats per site instead of four kernels moving 108; 240 FMAs per site.
struct SkipTaps { float wy[2][2][2][3], wx[2][2][2][3]; };

__global__ void __launch_bounds__(256) skip_chain_forward_kernel(float* __restrict__ out, const float* __restrict__ skip,
                                                                 const SkipTaps t, int C, int h, int w, int accumulate)
{
    const long long plane = (long long)h * w, total = (long long)C * plane;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % w);
        const long long r = idx / w;
        const int i = (int)(r % h), c = (int)(r / h);
        float xs[2][3][2][2];                      // [uy][a][u'x][px]: the x stage
#pragma unroll
        for (int uy = 0; uy < 2; uy++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int ii = i + a - 1;
                float in[2][3];
#pragma unroll
                for (int ux = 0; ux < 2; ux++)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const int jj = j + b - 1;
                        in[ux][b] = (ii >= 0 && ii < h && jj >= 0 && jj < w) ? skip[((size_t)(uy + 2 * ux) * C + c) * plane + (size_t)ii * w + jj] : 0.f;
                    }
#pragma unroll
                for (int vx = 0; vx < 2; vx++)
#pragma unroll
                    for (int px = 0; px < 2; px++) {
                        float acc = 0.f;
#pragma unroll
                        for (int ux = 0; ux < 2; ux++)
#pragma unroll
                            for (int b = 0; b < 3; b++) acc = fmaf(t.wx[vx][px][ux][b], in[ux][b], acc);
                        xs[uy][a][vx][px] = acc;
                    }
            }
#pragma unroll
        for (int vy = 0; vy < 2; vy++)
#pragma unroll
            for (int vx = 0; vx < 2; vx++)
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    float acc[2] = { 0.f, 0.f };
#pragma unroll
                    for (int px = 0; px < 2; px++)
#pragma unroll
                        for (int uy = 0; uy < 2; uy++)
#pragma unroll
                            for (int a = 0; a < 3; a++) acc[px] = fmaf(t.wy[vy][py][uy][a], xs[uy][a][vx][px], acc[px]);
                    float2* dst = reinterpret_cast<float2*>(out + (((size_t)(vy + 2 * vx) * C + c) * 2 * h + 2 * i + py) * (2 * (size_t)w) + 2 * j);
                    float2 v = make_float2(acc[0], acc[1]);
                    if (accumulate) { const float2 o = *dst; v.x += o.x; v.y += o.y; }
                    *dst = v;
                }
    }
}

// adjoint: gskip[(uy, ux)][i][j] = sum over the sites (I, J) = (i - a + 1, j - b + 1) that read it of
//          wy[u'y][py][uy][a] * wx[u'x][px][ux][b] * g[(u'y, u'x)][2I+py][2J+px]
__global__ void __launch_bounds__(256) skip_chain_backward_kernel(float* __restrict__ gskip, const float* __restrict__ g,
                                                                  const SkipTaps t, int C, int h, int w)
{
    const long long plane = (long long)h * w, total = (long long)C * plane;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % w);
        const long long r = idx / w;
        const int i = (int)(r % h), c = (int)(r / h);
        float acc[2][2] = { { 0.f, 0.f }, { 0.f, 0.f } };      // [uy][ux]
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int I = i - a + 1;
            if (I < 0 || I >= h) continue;
#pragma unroll
            for (int vy = 0; vy < 2; vy++)
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    float rx[2] = { 0.f, 0.f };                  // [ux]: the x stage of row (I, u'y, py)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const int J = j - b + 1;
                        if (J < 0 || J >= w) continue;
#pragma unroll
                        for (int vx = 0; vx < 2; vx++) {
                            const float2 gv = *reinterpret_cast<const float2*>(g + (((size_t)(vy + 2 * vx) * C + c) * 2 * h + 2 * I + py) * (2 * (size_t)w) + 2 * J);
#pragma unroll
                            for (int ux = 0; ux < 2; ux++)
                                rx[ux] = fmaf(t.wx[vx][0][ux][b], gv.x, fmaf(t.wx[vx][1][ux][b], gv.y, rx[ux]));
                        }
                    }
#pragma unroll
                    for (int uy = 0; uy < 2; uy++)
#pragma unroll
                        for (int ux = 0; ux < 2; ux++) acc[uy][ux] = fmaf(t.wy[vy][py][uy][a], rx[ux], acc[uy][ux]);
                }
        }
#pragma unroll
        for (int uy = 0; uy < 2; uy++)
#pragma unroll
            for (int ux = 0; ux < 2; ux++) gskip[((size_t)(uy + 2 * ux) * C + c) * plane + (size_t)i * w + j] = acc[uy][ux];
    }
}

__device__ __forceinline__ int floor_div(int a, int b)
{
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

struct UpfirdnParams {
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int major, in_h, in_w, kernel_h, kernel_w, out_h, out_w;
};

constexpr int kMaxTaps = 1024;

__global__ void __launch_bounds__(256) upfirdn2d_kernel(float* __restrict__ out, const float* __restrict__ input,
                                                       const float* __restrict__ kernel, UpfirdnParams p)
{
    __shared__ float taps[kMaxTaps];
    const int ntaps = p.kernel_h * p.kernel_w;
    for (int i = threadIdx.x; i < ntaps; i += 256) taps[i] = kernel[i];
    __syncthreads();
    const long long total = (long long)p.major * p.out_h * p.out_w;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int out_x = (int)(o % p.out_w);
        const long long t = o / p.out_w;
        const int out_y = (int)(t % p.out_h);
        const long long mj = t / p.out_h;
        // first contributing input row/col and the kernel tap that meets it (upfirdn2d_kernel.cu:62-81)
        const int mid_y = out_y * p.down_y + p.up_y - 1 - p.pad_y0;
        const int in_y = min(max(floor_div(mid_y, p.up_y), 0), p.in_h);
        const int h = min(max(floor_div(mid_y + p.kernel_h, p.up_y), 0), p.in_h) - in_y;
        const int kernel_y = mid_y + p.kernel_h - (in_y + 1) * p.up_y;
        const int mid_x = out_x * p.down_x + p.up_x - 1 - p.pad_x0;
        const int in_x = min(max(floor_div(mid_x, p.up_x), 0), p.in_w);
        const int w = min(max(floor_div(mid_x + p.kernel_w, p.up_x), 0), p.in_w) - in_x;
        const int kernel_x = mid_x + p.kernel_w - (in_x + 1) * p.up_x;
        const float* xp = input + (mj * p.in_h + in_y) * p.in_w + in_x;
        float v = 0.0f;
        for (int y = 0; y < h; y++) {
            const float* kp = taps + (kernel_y - y * p.up_y) * p.kernel_w + kernel_x;
            for (int x = 0; x < w; x++) v += xp[(long long)y * p.in_w + x] * kp[-x * p.up_x];
        }
        out[o] = v;
    }
}

// The network's heavy FIR calls are plain 4 x 4 filters (up = down = 1: the blur behind every transposed convolution, the blur in front
// of every stride-2 convolution, and their adjoints).  The general kernel above spends ~40 integer instructions (two 64-bit divisions)
// and 16 dependent scalar loads per output: 203 us for the 64-channel 512^2 blur where the bytes take 27.  Here a thread produces a
// 4 x 2 block of outputs from a 7 x 5 window read as ten 16-byte loads (alignment is free on this part: profiles/ub/load_rate.hip), grid
// (x quads, row pairs, image) so there is no division at all; same accumulation order per output as the general kernel (rows, then
// columns, ascending; out-of-image taps contribute exact zeros), hence the same bits.
// ACT (round 3): NoiseInjection + FusedLeakyReLU applied to the filtered value before it is stored -- the Blur behind an up-sampling
// ModulatedConv2d and the StyledConv tail as ONE pass (dual_styleunet.py:188-193, 301-311, 596): channel = blockIdx.z, the value goes
// through exactly the expression of noise_bias_act_forward_kernel.
struct FirAct {
    const float* noise;      // [out_h * out_w] or null
    const float* nw;         // [1] (with noise)
    const float* bias;       // [major] or null
    float slope, scale;
};

template <bool ACT>
__global__ void __launch_bounds__(256) fir4x4_kernel(float* __restrict__ out, const float* __restrict__ input,
                                                     const float* __restrict__ kernel, UpfirdnParams p, FirAct act)
{
    __shared__ float taps[16];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < 16) taps[threadIdx.x] = kernel[threadIdx.x];
    __syncthreads();
    const int ox0 = (blockIdx.x * 64 + tx) * 4, oy0 = (blockIdx.y * 4 + ty) * 2;
    if (ox0 >= p.out_w || oy0 >= p.out_h) return;
    const float* img = input + (size_t)blockIdx.z * p.in_h * p.in_w;
    const int cx = ox0 - p.pad_x0, cy = oy0 - p.pad_y0;                 // window origin: columns cx .. cx + 6, rows cy .. cy + 4
    float win[5][8];
    const bool x_inside = cx >= 0 && cx + 8 <= p.in_w;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int iy = cy + r;
        const bool row_ok = iy >= 0 && iy < p.in_h;
        const float* row = img + (size_t)(row_ok ? iy : 0) * p.in_w;
        if (row_ok && x_inside) {
            const float4 a = *reinterpret_cast<const float4*>(row + cx), b = *reinterpret_cast<const float4*>(row + cx + 4);
            win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
            win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z; win[r][7] = b.w;
        } else {
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const int ix = cx + c;
                win[r][c] = (row_ok && ix >= 0 && ix < p.in_w) ? row[ix] : 0.0f;
            }
            win[r][7] = 0.0f;
        }
    }
    float* dst = out + ((size_t)blockIdx.z * p.out_h + oy0) * p.out_w + ox0;
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
        if (oy0 + dy >= p.out_h) break;
        float v[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float k = taps[(3 - y) * 4 + (3 - x)];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = fmaf(win[dy + y][q + x], k, v[q]);
            }
        if constexpr (ACT) {
            const float b = act.bias ? act.bias[blockIdx.z] : 0.f, w = act.noise ? act.nw[0] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float n = (act.noise && ox0 + q < p.out_w) ? act.noise[(size_t)(oy0 + dy) * p.out_w + ox0 + q] : 0.f;
                const float t = v[q] + w * n + b;
                v[q] = (t > 0.f ? t : t * act.slope) * act.scale;
            }
        }
        if (ox0 + 4 <= p.out_w && (((size_t)(dst + (size_t)dy * p.out_w)) & 15) == 0) {
            *reinterpret_cast<float4*>(dst + (size_t)dy * p.out_w) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (ox0 + q < p.out_w) dst[(size_t)dy * p.out_w + q] = v[q];
        }
    }
}

// Backward of fir4x4_kernel<true> with pads (1, 1) (the up-sampling StyledConv: [C, OH + 1, OW + 1] -> [C, OH, OW]): the gradient of the
// activation input g_pre = g_out * (out > 0 ? 1 : slope) * scale is formed on the fly from g_out and the saved output, its bias /
// noise-strength sums are reduced (every g_pre element is owned by exactly one thread), and the FIR's adjoint -- the flipped taps with
// pads (2, 2) -- is applied to it: one pass instead of noise_bias_act_backward + upfirdn2d, and g_pre never goes to memory.
// Thread = 4 x 2 outputs of the [OH + 1, OW + 1] grid from a 7 x 5 window; a workgroup walks kFirBwdRows rows so that the per-channel
// sums cost few same-address atomics.
constexpr int kFirBwdRows = 32;

__global__ void __launch_bounds__(256) fir4x4_nba_backward_kernel(float* __restrict__ g_in, const float* __restrict__ g_out,
                                                                  const float* __restrict__ y, const float* __restrict__ kflip,
                                                                  const float* __restrict__ noise, float* __restrict__ gbias,
                                                                  float* __restrict__ gnw, int OH, int OW, float slope, float scale)
{
    __shared__ float taps[16];
    __shared__ float s_red[2][4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < 16) taps[threadIdx.x] = kflip[threadIdx.x];
    __syncthreads();
    const int IH = OH + 1, IW = OW + 1;                                  // the convolution output this gradient belongs to
    const size_t c = blockIdx.z;
    const float* go = g_out + c * OH * OW;
    const float* yo = y + c * OH * OW;
    float* gi = g_in + c * IH * IW;
    const int ox0 = (blockIdx.x * 64 + tx) * 4;
    float sb = 0.f, sn = 0.f;
    for (int rr = 0; rr < kFirBwdRows; rr += 8) {
        const int oy0 = blockIdx.y * kFirBwdRows + rr + ty * 2;
        if (ox0 >= IW || oy0 >= IH) continue;
        const int cx = ox0 - 2, cy = oy0 - 2;                            // window origin in the [OH, OW] grid of g_pre
        float win[5][8];
        const bool x_inside = cx >= 0 && cx + 8 <= OW;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int iy = cy + r;
            const bool row_ok = iy >= 0 && iy < OH;
            const size_t ro = (size_t)(row_ok ? iy : 0) * OW;
            if (row_ok && x_inside) {       // two 16-byte loads per array (alignment is free on this part)
                const float4 ga = *reinterpret_cast<const float4*>(go + ro + cx), gb4 = *reinterpret_cast<const float4*>(go + ro + cx + 4);
                const float4 ya = *reinterpret_cast<const float4*>(yo + ro + cx), yb = *reinterpret_cast<const float4*>(yo + ro + cx + 4);
                win[r][0] = ga.x * (ya.x > 0.f ? 1.f : slope) * scale; win[r][1] = ga.y * (ya.y > 0.f ? 1.f : slope) * scale;
                win[r][2] = ga.z * (ya.z > 0.f ? 1.f : slope) * scale; win[r][3] = ga.w * (ya.w > 0.f ? 1.f : slope) * scale;
                win[r][4] = gb4.x * (yb.x > 0.f ? 1.f : slope) * scale; win[r][5] = gb4.y * (yb.y > 0.f ? 1.f : slope) * scale;
                win[r][6] = gb4.z * (yb.z > 0.f ? 1.f : slope) * scale; win[r][7] = 0.f;
            } else {
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    const int ix = cx + q;
                    float v = 0.f;
                    if (row_ok && ix >= 0 && ix < OW) v = go[ro + ix] * (yo[ro + ix] > 0.f ? 1.f : slope) * scale;
                    win[r][q] = v;
                }
                win[r][7] = 0.f;
            }
        }
        // sums over the g_pre elements this thread owns: window rows 2..3, columns 2..5 (= positions oy0.., ox0.. of the [OH, OW] grid)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int iy = oy0 + dy, ix = ox0 + q;
                if (iy < OH && ix < OW) {
                    const float r = win[2 + dy][2 + q];
                    sb += r;
                    if (noise) sn += r * noise[(size_t)iy * OW + ix];
                }
            }
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            if (oy0 + dy >= IH) break;
            float v[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const float k = taps[(3 - yy) * 4 + (3 - x)];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = fmaf(win[dy + yy][q + x], k, v[q]);
                }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (ox0 + q < IW) gi[(size_t)(oy0 + dy) * IW + ox0 + q] = v[q];
        }
    }
    if (gbias || gnw) {
        sb = wave_sum(sb);
        sn = wave_sum(sn);
        if ((threadIdx.x & 63) == 0) { s_red[0][ty] = sb; s_red[1][ty] = sn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (gbias) atomicAdd(gbias + c, (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
            if (gnw) atomicAdd(gnw, (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
        }
    }
}

}  // namespace ag

using namespace ag;

extern "C" {

int ag_fused_bias_act(float* out, const float* x, const float* bias, const float* ref, int32_t act, int32_t grad,
                      float alpha, float scale, int64_t size_x, int64_t step_b, int32_t size_b, void* stream)
{
    if (size_x < 0 || (size_x > 0 && (!out || !x))) { set_error("bad fused_bias_act arguments"); return AG_ERR_INVALID_ARGUMENT; }
    if (size_x == 0) return AG_OK;
    if (bias && (size_b <= 0 || step_b <= 0)) { set_error("bias given but size_b/step_b invalid"); return AG_ERR_INVALID_ARGUMENT; }
    if (!bias || size_b <= 0) { bias = nullptr; size_b = 1; step_b = 1; }
    const int mode = act * 10 + grad;
    const auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int vec_ok = (step_b % 4 == 0 || !bias) && al(out) && al(x) && (!ref || al(ref));
    long long work = vec_ok ? (size_x + 3) / 4 : size_x;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 4096) blocks = 4096;   // grid-stride: ~16 workgroups per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fused_bias_act_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, x, bias,
                       ref, mode, alpha, scale, (long long)size_x, (long long)step_b, (int)size_b, vec_ok);
    return check_hip(hipGetLastError(), "fused_bias_act_kernel");
}

int ag_noise_bias_act_forward(float* y, const float* x, const float* noise, const float* noise_weight, const float* bias,
                              int32_t C, int32_t HW, float slope, float scale, void* stream)
{
    if (C < 0 || HW < 0 || ((C > 0 && HW > 0) && (!y || !x)) || (noise && !noise_weight)) {
        set_error("bad noise_bias_act arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (C == 0 || HW == 0) return AG_OK;
    const int chunks = (HW + kNbaChunk - 1) / kNbaChunk;
    hipLaunchKernelGGL(noise_bias_act_forward_kernel, dim3(C * chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, x,
                       noise, noise_weight, bias, C, HW, chunks, slope, scale);
    return check_hip(hipGetLastError(), "noise_bias_act_forward_kernel");
}

int ag_noise_bias_act_backward(float* gx, const float* gy, const float* y, const float* noise, float* gbias, float* gnoise_weight,
                               int32_t C, int32_t HW, float slope, float scale, void* stream)
{
    if (C < 0 || HW < 0 || ((C > 0 && HW > 0) && (!gx || !gy || !y)) || (gnoise_weight && !noise)) {
        set_error("bad noise_bias_act_backward arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (gbias && gnoise_weight == gbias + C) {       // the caller put the scalar right behind the C bias sums: one fill for both
        if (check_hip(hipMemsetAsync(gbias, 0, (size_t)(C + 1) * sizeof(float), s), "memset gbias+gnw")) return AG_ERR_HIP;
    } else {
        if (gbias && check_hip(hipMemsetAsync(gbias, 0, (size_t)C * sizeof(float), s), "memset gbias")) return AG_ERR_HIP;
        if (gnoise_weight && check_hip(hipMemsetAsync(gnoise_weight, 0, sizeof(float), s), "memset gnw")) return AG_ERR_HIP;
    }
    if (C == 0 || HW == 0) return AG_OK;
    const int chunks = (HW + kNbaChunk - 1) / kNbaChunk;
    hipLaunchKernelGGL(noise_bias_act_backward_kernel, dim3(C * chunks), dim3(256), 0, s, gx, gy, y, noise, gbias, gnoise_weight, C,
                       HW, chunks, slope, scale);
    return check_hip(hipGetLastError(), "noise_bias_act_backward_kernel");
}

int ag_modulate_weight_forward(float* out, float* dcoef, const float* W, const float* style, float scale, int32_t demodulate,
                               int32_t Co, int32_t Ci, int32_t K2, int32_t transposed, void* stream)
{
    if (Co <= 0 || Ci <= 0 || K2 <= 0 || !out || !W || !style) { set_error("bad modulate_weight arguments"); return AG_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(modulate_weight_forward_kernel, dim3(Co), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, dcoef, W,
                       style, scale, demodulate, Co, Ci, K2, transposed);
    return check_hip(hipGetLastError(), "modulate_weight_forward_kernel");
}

int ag_modulate_weight_backward(float* dW, float* dstyle, const float* g, const float* W, const float* style, const float* dcoef,
                                float scale, int32_t demodulate, int32_t Co, int32_t Ci, int32_t K2, int32_t transposed, void* stream)
{
    if (Co <= 0 || Ci <= 0 || K2 <= 0 || !dW || !dstyle || !g || !W || !style || (demodulate && !dcoef)) {
        set_error("bad modulate_weight_backward arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (check_hip(hipMemsetAsync(dstyle, 0, (size_t)Ci * sizeof(float), s), "memset dstyle")) return AG_ERR_HIP;
    hipLaunchKernelGGL(modulate_weight_backward_kernel, dim3(Co), dim3(256), 0, s, dW, dstyle, g, W, style, dcoef, scale, demodulate,
                       Co, Ci, K2, transposed);
    return check_hip(hipGetLastError(), "modulate_weight_backward_kernel");
}

int ag_block2x2_transform(float* out, const float* in, const float* matrix16, int32_t merge, int32_t C, int32_t h, int32_t w,
                          void* stream)
{
    if (C < 0 || h < 0 || w < 0 || !matrix16) { set_error("bad block2x2 arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)C * h * w;
    if (total == 0) return AG_OK;
    if (!out || !in) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    Block2x2 M;
    for (int i = 0; i < 16; i++) M.m[i] = matrix16[i];       // host pointer: 16 coefficients passed by value to the kernel
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (merge) hipLaunchKernelGGL(block2x2_merge_kernel, dim3((int)blocks), dim3(256), 0, s, out, in, M, C, h, w);
    else       hipLaunchKernelGGL(block2x2_split_kernel, dim3((int)blocks), dim3(256), 0, s, out, in, M, C, h, w);
    return check_hip(hipGetLastError(), "block2x2 kernel");
}

int ag_skip_chain_forward(float* out, const float* skip, const float* taps, int32_t C, int32_t h, int32_t w, int32_t accumulate, void* stream)
{
    if (C < 0 || h < 0 || w < 0) { set_error("bad skip chain sizes"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)C * h * w;
    if (total == 0) return AG_OK;
    if (!out || !skip || !taps) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SkipTaps t;
    memcpy(&t, taps, sizeof(t));                                // host pointer: wy[24] then wx[24], passed to the kernel by value
    hipLaunchKernelGGL(skip_chain_forward_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, skip, t, C, h, w,
                       accumulate);
    return check_hip(hipGetLastError(), "skip_chain_forward_kernel");
}

int ag_skip_chain_backward(float* gskip, const float* gout, const float* taps, int32_t C, int32_t h, int32_t w, void* stream)
{
    if (C < 0 || h < 0 || w < 0) { set_error("bad skip chain sizes"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)C * h * w;
    if (total == 0) return AG_OK;
    if (!gskip || !gout || !taps) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SkipTaps t;
    memcpy(&t, taps, sizeof(t));
    hipLaunchKernelGGL(skip_chain_backward_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gskip, gout, t, C, h, w);
    return check_hip(hipGetLastError(), "skip_chain_backward_kernel");
}

int ag_upfirdn2d(float* out, const float* input, const float* kernel, int32_t major, int32_t in_h, int32_t in_w,
                 int32_t kernel_h, int32_t kernel_w, int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                 int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1, void* stream)
{
    if (major < 0 || in_h <= 0 || in_w <= 0 || kernel_h <= 0 || kernel_w <= 0 || up_x <= 0 || up_y <= 0 || down_x <= 0 ||
        down_y <= 0 || kernel_h * kernel_w > kMaxTaps) {
        set_error("bad upfirdn2d sizes");
        return AG_ERR_INVALID_ARGUMENT;
    }
    UpfirdnParams p;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.kernel_h = kernel_h; p.kernel_w = kernel_w;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) / down_x;
    if (p.out_h <= 0 || p.out_w <= 0) { set_error("upfirdn2d output would be empty (%d x %d)", p.out_h, p.out_w); return AG_ERR_INVALID_ARGUMENT; }
    if (major == 0) return AG_OK;
    if (!out || !input || !kernel) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kernel_h == 4 && kernel_w == 4 && major <= 65535) {
        dim3 grid((p.out_w + 255) / 256, (p.out_h + 7) / 8, major);
        hipLaunchKernelGGL(fir4x4_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, input, kernel, p, FirAct{});
        return check_hip(hipGetLastError(), "fir4x4_kernel");
    }
    const long long total = (long long)major * p.out_h * p.out_w;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, input, kernel, p);
    return check_hip(hipGetLastError(), "upfirdn2d_kernel");
}

/* Blur (4 x 4 taps, pads (pad0, pad1)) + NoiseInjection + FusedLeakyReLU in one pass: out [major, OH, OW] from input [major, in_h, in_w]. */
int ag_fir4x4_noise_bias_act_forward(float* out, const float* input, const float* kernel, int32_t major, int32_t in_h, int32_t in_w,
                                     int32_t pad0, int32_t pad1, const float* noise, const float* noise_weight, const float* bias,
                                     float slope, float scale, void* stream)
{
    if (major <= 0 || major > 65535 || in_h <= 0 || in_w <= 0 || !out || !input || !kernel || (noise && !noise_weight)) {
        set_error("bad fir4x4_noise_bias_act_forward arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    UpfirdnParams p;
    p.up_x = p.up_y = p.down_x = p.down_y = 1; p.pad_x0 = p.pad_y0 = pad0;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.kernel_h = p.kernel_w = 4;
    p.out_h = in_h + pad0 + pad1 - 3; p.out_w = in_w + pad0 + pad1 - 3;
    if (p.out_h <= 0 || p.out_w <= 0) { set_error("fir4x4: empty output"); return AG_ERR_INVALID_ARGUMENT; }
    FirAct act{ noise, noise_weight, bias, slope, scale };
    dim3 grid((p.out_w + 255) / 256, (p.out_h + 7) / 8, major);
    hipLaunchKernelGGL(fir4x4_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, input, kernel, p, act);
    return check_hip(hipGetLastError(), "fir4x4_kernel<act>");
}

/* Its backward for pads (1, 1): g_in [major, OH + 1, OW + 1] from g_out / the saved output y [major, OH, OW]; kernel_flipped = the taps
 * flipped in both axes; gbias [major] / gnoise_weight [1] are zeroed here and accumulated (either may be NULL). */
int ag_fir4x4_noise_bias_act_backward(float* g_in, const float* g_out, const float* y, const float* kernel_flipped, int32_t major,
                                      int32_t OH, int32_t OW, const float* noise, float* gbias, float* gnoise_weight, float slope,
                                      float scale, void* stream)
{
    if (major <= 0 || major > 65535 || OH <= 0 || OW <= 0 || !g_in || !g_out || !y || !kernel_flipped || (gnoise_weight && !noise)) {
        set_error("bad fir4x4_noise_bias_act_backward arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (gbias && gnoise_weight == gbias + major) {
        if (check_hip(hipMemsetAsync(gbias, 0, (size_t)(major + 1) * sizeof(float), s), "memset gbias+gnw")) return AG_ERR_HIP;
    } else {
        if (gbias && check_hip(hipMemsetAsync(gbias, 0, (size_t)major * sizeof(float), s), "memset gbias")) return AG_ERR_HIP;
        if (gnoise_weight && check_hip(hipMemsetAsync(gnoise_weight, 0, sizeof(float), s), "memset gnw")) return AG_ERR_HIP;
    }
    dim3 grid((OW + 1 + 255) / 256, (OH + 1 + kFirBwdRows - 1) / kFirBwdRows, major);
    hipLaunchKernelGGL(fir4x4_nba_backward_kernel, grid, dim3(256), 0, s, g_in, g_out, y, kernel_flipped, gnoise_weight ? noise : nullptr,
                       gbias, gnoise_weight, OH, OW, slope, scale);
    return check_hip(hipGetLastError(), "fir4x4_nba_backward_kernel");
}

}  // extern "C"
