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
#include "ag_groups.h"
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
// gbias[c] = sum_pix gx, gnw = sum_{c,pix} gx * noise[pix].  One workgroup owns a run of pixels of ONE channel, so the bias is a scalar
// and both sums are a workgroup reduction.  The reductions are DETERMINISTIC (round 4): every workgroup stores its two partial sums,
// nba_finish_kernel adds them in a fixed order -- the noise-strength gradient is ONE number summed over a whole feature map with mixed
// signs, and float atomics in arrival order moved it by up to 3e-2 of its value from run to run.
// Grouped (ag_groups.h): x, y are [G][C][HW]; noise / noise weight / bias come from per-instance pointer tables.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kNbaChunk = 4096;   // pixels per workgroup (16 per thread)

// ADDEND: a second pre-activation term [C][HW] per instance (the encoder-level half of a decoder's comb convolution, computed once per
// network and shared by its members: ag_layers.hip ag_grouped_comb_*), in the place of the noise term
template <bool ADDEND>
__global__ void __launch_bounds__(256) noise_bias_act_forward_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                                    const PtrTable noise_t, const PtrTable nw_t, const PtrTable bias_t,
                                                                    int C, int HW, int chunks, float slope, float scale)
{
    const int cg = blockIdx.x / chunks, p0 = (blockIdx.x - cg * chunks) * kNbaChunk;      // cg: channel index over all instances
    const int grp = cg / C, c = cg - grp * C;
    const float* __restrict__ noise = ADDEND ? (noise_t.p[grp] ? noise_t.p[grp] + (size_t)c * HW : nullptr) : noise_t.p[grp];
    const float* __restrict__ bias = bias_t.p[grp];
    const float b = bias ? bias[c] : 0.f, w = ADDEND ? 1.0f : (noise ? nw_t.p[grp][0] : 0.f);
    const float* xr = x + (size_t)cg * HW;
    float* yr = y + (size_t)cg * HW;
    const int pend = min(HW, p0 + kNbaChunk);
    if ((HW & 3) == 0) {
        for (int p = p0 + threadIdx.x * 4; p < pend; p += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xr + p);
            float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise) n = *reinterpret_cast<const float4*>(noise + p);
            float4 o;
            float t;
            // fmaf spelled out: the convolution epilogue evaluates the same expression when the activation is fused into it (ag_conv.hip)
            t = fmaf(w, n.x, v.x) + b; o.x = (t > 0.f ? t : t * slope) * scale;
            t = fmaf(w, n.y, v.y) + b; o.y = (t > 0.f ? t : t * slope) * scale;
            t = fmaf(w, n.z, v.z) + b; o.z = (t > 0.f ? t : t * slope) * scale;
            t = fmaf(w, n.w, v.w) + b; o.w = (t > 0.f ? t : t * slope) * scale;
            *reinterpret_cast<float4*>(yr + p) = o;
        }
    } else {
        for (int p = p0 + threadIdx.x; p < pend; p += 256) {
            const float t = fmaf(w, noise ? noise[p] : 0.f, xr[p]) + b;
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

// partials: [G * C * chunks][2] = (sum gx, sum gx * noise) of every workgroup, followed by [G * C * chunks] = max |gx| (the operand maximum
// the fp16 split form of the convolutions that consume gx needs, ag_groups.h: taken here, the tensor is not read a second time for it)
__global__ void __launch_bounds__(256) noise_bias_act_backward_kernel(float* __restrict__ gx, const float* __restrict__ gy,
                                                                     const float* __restrict__ y, const PtrTable noise_t,
                                                                     float* __restrict__ partials, int C, int HW, int chunks, float slope,
                                                                     float scale)
{
    __shared__ float s_red[3][4];
    float mx = 0.f;
    const int cg = blockIdx.x / chunks, p0 = (blockIdx.x - cg * chunks) * kNbaChunk;
    const float* __restrict__ noise = noise_t.p[cg / C];
    const float* gyr = gy + (size_t)cg * HW;
    const float* yr = y + (size_t)cg * HW;
    float* gxr = gx + (size_t)cg * HW;
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
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            sb += (r.x + r.y) + (r.z + r.w);
            sn += (r.x * n.x + r.y * n.y) + (r.z * n.z + r.w * n.w);
        }
    } else {
        for (int p = p0 + threadIdx.x; p < pend; p += 256) {
            const float r = gyr[p] * (yr[p] > 0.f ? 1.f : slope) * scale;
            gxr[p] = r;
            mx = fmaxf(mx, fabsf(r));
            sb += r;
            if (noise) sn += r * noise[p];
        }
    }
    if (!partials) return;
    sb = wave_sum(sb);
    sn = wave_sum(sn);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[0][wave] = sb; s_red[1][wave] = sn; s_red[2][wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * (size_t)blockIdx.x] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * (size_t)blockIdx.x + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        partials[2 * (size_t)gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_red[2][0], s_red[2][1]), fmaxf(s_red[2][2], s_red[2][3]));
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return r;
}

// One workgroup per instance: gbias[c] = the channel's chunk sums in chunk order; gnw = all C * chunks sums of the instance, each thread a
// strided subsequence in order, then the fixed butterfly of block_sum_256.  Same bits on every run.
// amax: [G][256] = the instance's max |gx|, every entry the same value (the layout conv_absmax's consumers finish)
__global__ void __launch_bounds__(256) nba_finish_kernel(const float* __restrict__ partials, float* __restrict__ gbias, long long gb_stride,
                                                         float* __restrict__ gnw, long long gnw_stride, float* __restrict__ amax, int C, int per_channel)
{
    __shared__ float s_red[4];
    const int grp = blockIdx.x;
    if (amax) {
        const float* pm = partials + 2 * (size_t)gridDim.x * C * per_channel + (size_t)grp * C * per_channel;
        float m = 0.f;
        for (int i = threadIdx.x; i < C * per_channel; i += 256) m = fmaxf(m, pm[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        amax[(size_t)grp * 256 + threadIdx.x] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        __syncthreads();
    }
    const float* part = partials + 2 * (size_t)grp * C * per_channel;
    if (gbias)
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (int k = 0; k < per_channel; k++) s += part[2 * ((size_t)c * per_channel + k)];
            gbias[(size_t)grp * gb_stride + c] = s;
        }
    if (gnw) {
        float t = 0.f;
        for (int i = threadIdx.x; i < C * per_channel; i += 256) t += part[2 * (size_t)i + 1];
        t = block_sum_256(t, s_red);
        if (threadIdx.x == 0) gnw[(size_t)grp * gnw_stride] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight modulation + demodulation of ModulatedConv2d's fused branch (dual_styleunet.py:254-259), one workgroup per
// output channel:  w'[co][ci][k] = (scale * W[co][ci][k]) * style[ci];  d[co] = rsqrt(sum w'^2 + 1e-8);  out = w' * d.
// `transposed` writes out[ci][co][k] (the layout conv_transpose2d takes, :268-272).  Grouped: blockIdx.x = instance * Co + co.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) modulate_weight_forward_kernel(float* __restrict__ out, float* __restrict__ dcoef,
                                                                     const PtrTable W_t, const PtrTable style_t,
                                                                     float scale, int demod, int Co, int Ci, int K2, int transposed,
                                                                     float* __restrict__ rowmax)
{
    __shared__ float s_red[4];
    float vmax = 0.f;
    const int grp = blockIdx.x / Co, co = blockIdx.x - grp * Co, n = Ci * K2;
    const float* __restrict__ style = style_t.p[grp];
    const float* w = W_t.p[grp] + (size_t)co * n;
    out += (size_t)grp * Co * n;
    float q = 0.f;
    if (demod) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float v = (scale * w[i]) * style[i / K2];
            q += v * v;
        }
        q = block_sum_256(q, s_red);
    }
    const float d = demod ? rsqrtf(q + 1e-8f) : 1.0f;
    if (threadIdx.x == 0 && dcoef) dcoef[blockIdx.x] = d;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / K2, k = i - ci * K2;
        const float v = (scale * w[i]) * style[ci];
        const size_t o = transposed ? ((size_t)ci * Co + co) * K2 + k : (size_t)co * n + i;
        const float r = demod ? v * d : v;
        out[o] = r;
        vmax = fmaxf(vmax, fabsf(r));
    }
    if (rowmax) {        // round 5: the row's largest magnitude as it is stored -- the fp16-split convolutions need the tensor's maximum, and
                         // Cout row maxima replace a sweep of the whole modulated weight (ag_layers.hip keep_maxima)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) rowmax[blockIdx.x] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    }
}

constexpr int kModCi = 256, kModMaxTaps = 9;

// g = dL/dout.  dL/dw' = g * d - d^3 * (sum g w') * w'   (g when there is no demodulation);
// dW = dL/dw' * scale * style[ci];  dstyle[ci] = sum_{co, k} dL/dw' * scale * W: every (co, ci) pair's sum over the taps goes to
// partials[grp][co][ci], modulate_finish_kernel adds them over co in a fixed order (deterministic; float atomics over co before round 4)
__global__ void __launch_bounds__(256) modulate_weight_backward_kernel(float* __restrict__ dW, float* __restrict__ partials,
                                                                      const float* __restrict__ g, const PtrTable W_t,
                                                                      const PtrTable style_t, const float* __restrict__ dcoef,
                                                                      float scale, int demod, int Co, int Ci, int K2, int transposed)
{
    __shared__ float s_red[4];
    const int grp = blockIdx.x / Co, co = blockIdx.x - grp * Co, n = Ci * K2;
    const float* __restrict__ style = style_t.p[grp];
    const float* w = W_t.p[grp] + (size_t)co * n;
    g += (size_t)grp * Co * n;
    dW += (size_t)grp * Co * n;
    float gw = 0.f;
    const float d = demod ? dcoef[blockIdx.x] : 1.0f;
    if (demod) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int ci = i / K2, k = i - ci * K2;
            const size_t o = transposed ? ((size_t)ci * Co + co) * K2 + k : (size_t)co * n + i;
            gw += g[o] * ((scale * w[i]) * style[ci]);
        }
        gw = block_sum_256(gw, s_red);
    }
    const float c3 = d * d * d * gw;
    if (!transposed && K2 <= kModMaxTaps) {
        // [Co][Ci][K2] layout (the layer calls): 256 input channels at a time go through LDS so that every global access of the row is a
        // contiguous run (a thread owning the K2 taps of one channel reads and writes at a 4 K2-byte stride: 9x the time of its bytes
        // on the 512 x 512 x 9 layers, round 4); in LDS the same stride is conflict-free for odd K2
        __shared__ float s_w[kModCi * kModMaxTaps], s_g[kModCi * kModMaxTaps];
        const float* grow = g + (size_t)co * n;
        float* dwrow = dW + (size_t)co * n;
        for (int c0 = 0; c0 < Ci; c0 += kModCi) {
            const int nci = min(kModCi, Ci - c0), cnt = nci * K2, base = c0 * K2;
            for (int i = threadIdx.x; i < cnt; i += 256) { s_w[i] = w[base + i]; s_g[i] = grow[base + i]; }
            __syncthreads();
            if ((int)threadIdx.x < nci) {
                const int ci = c0 + threadIdx.x;
                const float st = style[ci], sc = scale * st;
                float ds = 0.f;
                for (int k = 0; k < K2; k++) {
                    const int i = threadIdx.x * K2 + k;
                    const float sw = scale * s_w[i];
                    const float gp = demod ? s_g[i] * d - c3 * (sw * st) : s_g[i];
                    s_g[i] = gp * sc;
                    ds += gp * sw;
                }
                partials[(size_t)blockIdx.x * Ci + ci] = ds;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += 256) dwrow[base + i] = s_g[i];
            __syncthreads();
        }
        return;
    }
    // general form (conv_transpose2d's [Ci][Co][K2] layout of g, any K2): one thread per input channel keeps the K2 taps of that channel together
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
        partials[(size_t)blockIdx.x * Ci + ci] = ds;
    }
}

// dstyle[grp][ci] = sum_co partials[grp][co][ci]: a workgroup owns 64 channels, its sixteen waves a sixteenth of the rows each (four
// independent partial sums per thread: the loop is a chain of dependent loads otherwise), combined in a fixed order
constexpr int kModFinishWaves = 16;
__global__ void __launch_bounds__(64 * kModFinishWaves) modulate_finish_kernel(float* __restrict__ dstyle, const float* __restrict__ partials, int Co, int Ci)
{
    __shared__ float s_q[kModFinishWaves][64];
    const int grp = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int ci = blockIdx.x * 64 + lane;
    const float* part = partials + (size_t)grp * Co * Ci;
    const int per = (Co + kModFinishWaves - 1) / kModFinishWaves, r0 = min(Co, q * per), r1 = min(Co, r0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ci < Ci) {
        int co = r0;
        for (; co + 3 < r1; co += 4) {
            s0 += part[(size_t)co * Ci + ci];
            s1 += part[(size_t)(co + 1) * Ci + ci];
            s2 += part[(size_t)(co + 2) * Ci + ci];
            s3 += part[(size_t)(co + 3) * Ci + ci];
        }
        for (; co < r1; co++) s0 += part[(size_t)co * Ci + ci];
    }
    s_q[q][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && ci < Ci) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < kModFinishWaves; i++) t += s_q[i][lane];
        dstyle[(size_t)grp * Ci + ci] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 2x2 block transforms: the Haar analysis / synthesis of the wavelet skip path (dual_styleunet.py:374-425) as ONE pass
// instead of 4 upfirdn2d calls + 3 additions / a concatenation.
//   split: y[b][c][i][j] = sum_p M[b][p] * x[c][2i + p/2][2j + p%2]          x [C][2h][2w] -> y [4][C][h][w]
//   merge: x[c][2i + p/2][2j + p%2] = sum_b M[p][b] * y[b][c][i][j]          y [4][C][h][w] -> x [C][2h][2w]
// ------------------------------------------------------------------------------------------------------------------
struct Block2x2 { float m[16]; };

__global__ void __launch_bounds__(256) block2x2_split_kernel(float* __restrict__ y, const float* __restrict__ x, Block2x2 M, int G, int C,
                                                            int h, int w)
{
    const long long per = (long long)C * h * w, total = per * G;          // grouped: x [G][C][2h][2w] -> y [G][4][C][h][w]
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int jx = (int)(i % w);
        const long long r = i / w;
        const int iy = (int)(r % h);
        const long long cg = r / h;                                       // channel over all instances
        const long long grp = cg / C, il = i - grp * per;
        const float* src = x + ((size_t)cg * 2 * h + 2 * iy) * (2 * w) + 2 * jx;
        const float2 t = *reinterpret_cast<const float2*>(src), b = *reinterpret_cast<const float2*>(src + 2 * w);
        const float v[4] = { t.x, t.y, b.x, b.y };
        float* dst = y + (size_t)grp * 4 * per + il;
#pragma unroll
        for (int band = 0; band < 4; band++)
            dst[(size_t)band * per] = M.m[4 * band + 0] * v[0] + M.m[4 * band + 1] * v[1] + M.m[4 * band + 2] * v[2] + M.m[4 * band + 3] * v[3];
    }
}

__global__ void __launch_bounds__(256) block2x2_merge_kernel(float* __restrict__ x, const float* __restrict__ y, Block2x2 M, int G, int C,
                                                            int h, int w)
{
    const long long per = (long long)C * h * w, total = per * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int jx = (int)(i % w);
        const long long r = i / w;
        const int iy = (int)(r % h);
        const long long cg = r / h;
        const long long grp = cg / C, il = i - grp * per;
        const float* src = y + (size_t)grp * 4 * per + il;
        float v[4];
#pragma unroll
        for (int band = 0; band < 4; band++) v[band] = src[(size_t)band * per];
        float o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = M.m[4 * p + 0] * v[0] + M.m[4 * p + 1] * v[1] + M.m[4 * p + 2] * v[2] + M.m[4 * p + 3] * v[3];
        float* dst = x + ((size_t)cg * 2 * h + 2 * iy) * (2 * w) + 2 * jx;
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

__global__ void __launch_bounds__(256) skip_chain_forward_kernel(float* __restrict__ out_all, const float* __restrict__ skip_all,
                                                                 const SkipTaps t, int G, int C, int h, int w, int accumulate)
{
    const long long plane = (long long)h * w, total = (long long)G * C * plane;       // grouped: skip [G][4C][h][w] -> out [G][4C][2h][2w]
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % w);
        const long long r = idx / w;
        const int i = (int)(r % h);
        const int cg = (int)(r / h), grp = cg / C, c = cg - grp * C;
        const float* __restrict__ skip = skip_all + (size_t)grp * 4 * C * plane;
        float* __restrict__ out = out_all + (size_t)grp * 16 * C * plane;
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
__global__ void __launch_bounds__(256) skip_chain_backward_kernel(float* __restrict__ gskip_all, const float* __restrict__ g_all,
                                                                  const SkipTaps t, int G, int C, int h, int w)
{
    const long long plane = (long long)h * w, total = (long long)G * C * plane;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % w);
        const long long r = idx / w;
        const int i = (int)(r % h);
        const int cg = (int)(r / h), grp = cg / C, c = cg - grp * C;
        float* __restrict__ gskip = gskip_all + (size_t)grp * 4 * C * plane;
        const float* __restrict__ g = g_all + (size_t)grp * 16 * C * plane;
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
// 4-wide, kFirRows-tall strip of outputs while sliding a 4-row window of 8 columns down the image (two 16-byte loads per input row;
// alignment is free on this part: profiles/ub/load_rate.hip), threads numbered (column quad, strip) linearly over a plane so narrow
// images still fill their wavefronts; same accumulation order per output as the general kernel (rows, then columns, ascending;
// out-of-image taps contribute exact zeros), hence the same bits.
// Edges without divergence (round 4): the 16-byte loads are issued wherever they stay inside the TENSOR (a window hanging over a row's
// end reads the neighbouring row or plane) and the out-of-image columns are zeroed by selects; only the few threads whose loads would
// leave the tensor take the element-wise path.  The round-3 form branched per row on "window inside the row", and with 64 lanes on half
// a 512-pixel row EVERY wavefront had an edge lane and ran both paths (2.65 TB/s, profiles/r04_kernel_stats_fullstep.txt).
// ACT: NoiseInjection + FusedLeakyReLU applied to the filtered value before it is stored (the activation pass of an up-sampling
// StyledConv, ag_layers.hip), the same expression as noise_bias_act_forward_kernel, hence the same bits as the two passes.
constexpr int kFirRows = 8;
// amax: [G][256] zero-initialised slots that receive the largest magnitude the kernel stores, per group of amax_C planes (unsigned atomic
// maxima of the float bits: order-independent, hence deterministic) -- the operand maximum the fp16-split convolutions consuming the output
// need (ag_groups.h), taken while the data is in registers instead of by another sweep
struct FirAct { PtrTable noise, nw, bias; int C; float slope, scale; float* amax; int amax_C; };

template <bool ACT>
__global__ void __launch_bounds__(256) fir4x4_kernel(float* __restrict__ out, const float* __restrict__ input,
                                                     const float* __restrict__ kernel, UpfirdnParams p, int quads, int strips, int nblocks,
                                                     long long total, const FirAct act)
{
    float mx = 0.f;
    float taps[16];
#pragma unroll
    for (int i = 0; i < 16; i++) taps[i] = kernel[i];                    // uniform address: scalar loads
    const int plane = blockIdx.y;
    // a workgroup walks the plane's thread blocks gridDim.x apart (the launcher caps gridDim.x when `amax` is wanted: one atomic per
    // workgroup at the end, and few workgroups per cache line of slots)
    for (int bx = blockIdx.x; bx < nblocks; bx += gridDim.x) {
    const int t = bx * 256 + threadIdx.x;
    const int strip_t = t / quads, quad = t - strip_t * quads;
    const bool live = strip_t < strips;                                  // (no early exit: the wave reduction of `amax` wants every lane)
    const int strip = live ? strip_t : 0;
    const int ox0 = quad * 4, oy0 = strip * kFirRows;
    const long long plane0 = (long long)plane * p.in_h * p.in_w;
    const int cx = ox0 - p.pad_x0, cy = oy0 - p.pad_y0;                 // window origin: columns cx .. cx + 6, rows cy .. cy + kFirRows + 2
    bool col_ok[8];
#pragma unroll
    for (int c = 0; c < 8; c++) col_ok[c] = c < 7 && cx + c >= 0 && cx + c < p.in_w;
    float win[4][8];
    auto load_row = [&](int r, float (&dst)[8]) {
        const int iy = cy + r;
        if (iy < 0 || iy >= p.in_h) {
#pragma unroll
            for (int c = 0; c < 8; c++) dst[c] = 0.0f;
            return;
        }
        const long long off = plane0 + (long long)iy * p.in_w + cx;
        if (off >= 0 && off + 8 <= total) {
            const float4 a = *reinterpret_cast<const float4*>(input + off), b = *reinterpret_cast<const float4*>(input + off + 4);
            dst[0] = col_ok[0] ? a.x : 0.f; dst[1] = col_ok[1] ? a.y : 0.f; dst[2] = col_ok[2] ? a.z : 0.f; dst[3] = col_ok[3] ? a.w : 0.f;
            dst[4] = col_ok[4] ? b.x : 0.f; dst[5] = col_ok[5] ? b.y : 0.f; dst[6] = col_ok[6] ? b.z : 0.f; dst[7] = 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) dst[c] = col_ok[c] ? input[off + c] : 0.0f;
        }
    };
    load_row(0, win[0]);
    load_row(1, win[1]);
    load_row(2, win[2]);
    float* dst = out + ((size_t)plane * p.out_h + oy0) * p.out_w + ox0;
    float nw = 0.f, bs = 0.f;
    const float* noise = nullptr;
    if (ACT) {
        const int grp = plane / act.C, c = plane - grp * act.C;
        noise = act.noise.p[grp];
        nw = noise ? act.nw.p[grp][0] : 0.f;
        bs = act.bias.p[grp] ? act.bias.p[grp][c] : 0.f;
    }
#pragma unroll
    for (int dy = 0; dy < kFirRows; dy++) {
        if (oy0 + dy >= p.out_h) break;
        load_row(dy + 3, win[(dy + 3) & 3]);
        float v[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float k = taps[(3 - y) * 4 + (3 - x)];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = fmaf(win[(dy + y) & 3][q + x], k, v[q]);
            }
        float* o = dst + (size_t)dy * p.out_w;
        if (ACT) {
            const size_t pix = (size_t)(oy0 + dy) * p.out_w + ox0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float n = (noise && ox0 + q < p.out_w) ? noise[pix + q] : 0.f;
                const float tv = fmaf(nw, n, v[q]) + bs;
                v[q] = (tv > 0.f ? tv : tv * act.slope) * act.scale;
            }
        }
        if (!live) continue;
        if (ox0 + 4 <= p.out_w && (((size_t)o) & 15) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (ox0 + q < p.out_w) { o[q] = v[q]; mx = fmaxf(mx, fabsf(v[q])); }
        }
    }
    }   // bx
    if (act.amax) {      // one non-returning atomic per workgroup
        __shared__ float s_mx[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned int*>(act.amax) + (size_t)(plane / act.amax_C) * 256 + ((blockIdx.x + 37u * (unsigned)plane) & 255u),
                      __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
    }
}

static int launch_fir4x4(float* out, const float* input, const float* kernel, const UpfirdnParams& p, int major, const FirAct* act, hipStream_t s,
                         float* amax = nullptr, int amax_C = 1)
{
    const int quads = (p.out_w + 3) / 4, strips = (p.out_h + kFirRows - 1) / kFirRows;
    const long long threads = (long long)quads * strips, total = (long long)major * p.in_h * p.in_w;
    const int nblocks = (int)((threads + 255) / 256);
    // with `amax`: at most ~4096 workgroups in all (each ends with one atomic into its group's 16 cache lines of slots)
    int gx = nblocks;
    if (amax) gx = std::max(1, std::min(nblocks, 4096 / std::max(1, major)));
    dim3 grid((unsigned)gx, major);
    FirAct a = act ? *act : FirAct{};
    a.amax = amax; a.amax_C = amax_C > 0 ? amax_C : 1;
    if (act) hipLaunchKernelGGL(fir4x4_kernel<true>, grid, dim3(256), 0, s, out, input, kernel, p, quads, strips, nblocks, total, a);
    else     hipLaunchKernelGGL(fir4x4_kernel<false>, grid, dim3(256), 0, s, out, input, kernel, p, quads, strips, nblocks, total, a);
    return check_hip(hipGetLastError(), "fir4x4_kernel");
}

// ------------------------------------------------------------------------------------------------------------------
// grouped launchers (ag_groups.h); G = 1 is the per-kernel C ABI below
// ------------------------------------------------------------------------------------------------------------------
static bool bad_groups(int G) { return G < 1 || G > kMaxGroups; }

// Blur pad (1, 1) of a [G][C][H][W] stack followed by the activation pass, one kernel (the tail of an up-sampling StyledConv)
int blur_act_forward_g(float* y, const float* x, const float* taps, int G, const PtrTable& noise, const PtrTable& nw, const PtrTable& bias,
                       int C, int H, int W, float slope, float scale, hipStream_t s, float* out_amax)
{
    if (bad_groups(G) || C < 1 || H < 3 || W < 3 || !y || !x || !taps || (long long)G * C > 65535) { set_error("bad blur_act arguments"); return AG_ERR_INVALID_ARGUMENT; }
    for (int g = 0; g < G; g++)
        if (noise.p[g] && !nw.p[g]) { set_error("blur_act: noise without a noise weight"); return AG_ERR_INVALID_ARGUMENT; }
    UpfirdnParams p;
    p.up_x = p.up_y = p.down_x = p.down_y = 1; p.pad_x0 = p.pad_y0 = 1;
    p.major = G * C; p.in_h = H; p.in_w = W; p.kernel_h = p.kernel_w = 4;
    p.out_h = H - 1; p.out_w = W - 1;
    FirAct act{ noise, nw, bias, C, slope, scale };
    return launch_fir4x4(y, x, taps, p, G * C, &act, s, out_amax, C);
}

// 4 x 4 FIR with pads (pad, pad) of `planes` planes, leaving the largest magnitude of the output of every group of planes_per_group planes
// in amax[group][256] (zeroed by the caller): the Blur adjoint in front of an up-sampling StyledConv's backward convolutions
int fir4x4_amax_g(float* out, const float* in, const float* taps, int planes, int in_h, int in_w, int pad, float* amax, int planes_per_group,
                  hipStream_t s)
{
    if (planes < 1 || planes > 65535 || in_h < 1 || in_w < 1 || pad < 0 || !out || !in || !taps || in_h + 2 * pad < 4 || in_w + 2 * pad < 4) {
        set_error("bad fir4x4 arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    UpfirdnParams p;
    p.up_x = p.up_y = p.down_x = p.down_y = 1; p.pad_x0 = p.pad_y0 = pad;
    p.major = planes; p.in_h = in_h; p.in_w = in_w; p.kernel_h = p.kernel_w = 4;
    p.out_h = in_h + 2 * pad - 3; p.out_w = in_w + 2 * pad - 3;
    return launch_fir4x4(out, in, taps, p, planes, nullptr, s, amax, planes_per_group);
}

int noise_bias_act_forward_g(float* y, const float* x, int G, const PtrTable& noise, const PtrTable& nw, const PtrTable& bias, int C, int HW,
                             float slope, float scale, hipStream_t s)
{
    if (bad_groups(G) || C < 0 || HW < 0 || ((C > 0 && HW > 0) && (!y || !x))) { set_error("bad noise_bias_act arguments"); return AG_ERR_INVALID_ARGUMENT; }
    for (int g = 0; g < G; g++)
        if (noise.p[g] && !nw.p[g]) { set_error("noise_bias_act: noise without a noise weight"); return AG_ERR_INVALID_ARGUMENT; }
    if (C == 0 || HW == 0) return AG_OK;
    const int chunks = (HW + kNbaChunk - 1) / kNbaChunk;
    hipLaunchKernelGGL(noise_bias_act_forward_kernel<false>, dim3(G * C * chunks), dim3(256), 0, s, y, x, noise, nw, bias, C, HW, chunks, slope, scale);
    return check_hip(hipGetLastError(), "noise_bias_act_forward_kernel");
}

int bias_act_forward_addend_g(float* y, const float* x, int G, const PtrTable& addend, const PtrTable& bias, int C, int HW, float slope, float scale,
                              hipStream_t s)
{
    if (bad_groups(G) || C <= 0 || HW <= 0 || !y || !x) { set_error("bad bias_act_forward_addend arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int chunks = (HW + kNbaChunk - 1) / kNbaChunk;
    // x + 1.0f * addend + bias: the multiplication by 1 is exact, so this is the sum of the two convolution halves in the order (x + addend) + bias
    hipLaunchKernelGGL(noise_bias_act_forward_kernel<true>, dim3(G * C * chunks), dim3(256), 0, s, y, x, addend, PtrTable{}, bias, C, HW, chunks, slope, scale);
    return check_hip(hipGetLastError(), "noise_bias_act_forward_kernel<addend>");
}

struct MemberRanges { int begin[kMaxGroups + 1]; };

__global__ void __launch_bounds__(256) sum_member_ranges_kernel(float* __restrict__ out, const float* __restrict__ in, MemberRanges r, long long n)
{
    const int rr = blockIdx.y;
    const int m0 = r.begin[rr], m1 = r.begin[rr + 1];
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4*>(in + (size_t)m0 * n)[i];
        for (int m = m0 + 1; m < m1; m++) {
            const float4 v = reinterpret_cast<const float4*>(in + (size_t)m * n)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(out + (size_t)rr * n)[i] = acc;
    }
}

int sum_member_ranges(float* out, const float* in, const int* begin, int R, long long n, hipStream_t s)
{
    if (R < 1 || R > kMaxGroups || n <= 0 || (n & 3) || !out || !in || !begin) { set_error("bad sum_member_ranges arguments"); return AG_ERR_INVALID_ARGUMENT; }
    MemberRanges r{};
    for (int i = 0; i <= R; i++) r.begin[i] = begin[i];
    for (int i = 0; i < R; i++)
        if (r.begin[i + 1] <= r.begin[i]) { set_error("sum_member_ranges: empty range"); return AG_ERR_INVALID_ARGUMENT; }
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_member_ranges_kernel, dim3((int)blocks, R), dim3(256), 0, s, out, in, r, n);
    return check_hip(hipGetLastError(), "sum_member_ranges_kernel");
}

size_t noise_bias_act_partial_floats(int G, int C, int HW)
{
    if (G <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)3 * G * C * ((HW + kNbaChunk - 1) / kNbaChunk);      // (sum, sum * noise) pairs, then the maxima
}

int noise_bias_act_backward_g(float* gx, const float* gy, const float* y, int G, const PtrTable& noise, float* gbias, long long gb_stride,
                              float* gnw, long long gnw_stride, float* partials, int C, int HW, float slope, float scale, hipStream_t s, float* amax)
{
    if (bad_groups(G) || C < 0 || HW < 0 || ((C > 0 && HW > 0) && (!gx || !gy || !y)) || ((gbias || gnw || amax) && !partials)) {
        set_error("bad noise_bias_act_backward arguments (the parameter sums need the `partials` scratch)");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (gnw)
        for (int g = 0; g < G; g++)
            if (!noise.p[g]) { set_error("noise_bias_act_backward: noise-strength gradient without noise"); return AG_ERR_INVALID_ARGUMENT; }
    if (C == 0 || HW == 0) {
        for (int g = 0; g < G; g++) {
            if (gbias && C > 0 && check_hip(hipMemsetAsync(gbias + g * gb_stride, 0, (size_t)C * sizeof(float), s), "memset gbias")) return AG_ERR_HIP;
            if (gnw && check_hip(hipMemsetAsync(gnw + g * gnw_stride, 0, sizeof(float), s), "memset gnw")) return AG_ERR_HIP;
        }
        return AG_OK;
    }
    const int chunks = (HW + kNbaChunk - 1) / kNbaChunk;
    const bool sums = gbias || gnw || amax;
    hipLaunchKernelGGL(noise_bias_act_backward_kernel, dim3(G * C * chunks), dim3(256), 0, s, gx, gy, y, gnw ? noise : PtrTable{},
                       sums ? partials : nullptr, C, HW, chunks, slope, scale);
    if (check_hip(hipGetLastError(), "noise_bias_act_backward_kernel")) return AG_ERR_HIP;
    if (!sums) return AG_OK;
    hipLaunchKernelGGL(nba_finish_kernel, dim3(G), dim3(256), 0, s, partials, gbias, gb_stride, gnw, gnw_stride, amax, C, chunks);
    return check_hip(hipGetLastError(), "nba_finish_kernel");
}

int modulate_weight_forward_g(float* out, float* dcoef, int G, const PtrTable& W, const PtrTable& style, float scale, int demod, int Co, int Ci,
                              int K2, int transposed, hipStream_t s, float* rowmax)
{
    if (bad_groups(G) || Co <= 0 || Ci <= 0 || K2 <= 0 || !out || !table_complete(W, G) || !table_complete(style, G)) {
        set_error("bad modulate_weight arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(modulate_weight_forward_kernel, dim3(G * Co), dim3(256), 0, s, out, dcoef, W, style, scale, demod, Co, Ci, K2, transposed, rowmax);
    return check_hip(hipGetLastError(), "modulate_weight_forward_kernel");
}

size_t modulate_weight_partial_floats(int G, int Co, int Ci)
{
    if (G <= 0 || Co <= 0 || Ci <= 0) return 0;
    return (size_t)G * Co * Ci;
}

int modulate_weight_backward_g(float* dW, float* dstyle, float* partials, const float* g, int G, const PtrTable& W, const PtrTable& style,
                               const float* dcoef, float scale, int demod, int Co, int Ci, int K2, int transposed, hipStream_t s)
{
    if (bad_groups(G) || Co <= 0 || Ci <= 0 || K2 <= 0 || !dW || !dstyle || !partials || !g || !table_complete(W, G) || !table_complete(style, G) ||
        (demod && !dcoef)) {
        set_error("bad modulate_weight_backward arguments");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(modulate_weight_backward_kernel, dim3(G * Co), dim3(256), 0, s, dW, partials, g, W, style, dcoef, scale, demod, Co, Ci, K2,
                       transposed);
    if (check_hip(hipGetLastError(), "modulate_weight_backward_kernel")) return AG_ERR_HIP;
    hipLaunchKernelGGL(modulate_finish_kernel, dim3((Ci + 63) / 64, G), dim3(64 * kModFinishWaves), 0, s, dstyle, partials, Co, Ci);
    return check_hip(hipGetLastError(), "modulate_finish_kernel");
}

int block2x2_transform_g(float* out, const float* in, const float* matrix16, int merge, int G, int C, int h, int w, hipStream_t s)
{
    if (bad_groups(G) || C < 0 || h < 0 || w < 0 || !matrix16) { set_error("bad block2x2 arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)G * C * h * w;
    if (total == 0) return AG_OK;
    if (!out || !in) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    Block2x2 M;
    for (int i = 0; i < 16; i++) M.m[i] = matrix16[i];       // host pointer: 16 coefficients passed by value to the kernel
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (merge) hipLaunchKernelGGL(block2x2_merge_kernel, dim3((int)blocks), dim3(256), 0, s, out, in, M, G, C, h, w);
    else       hipLaunchKernelGGL(block2x2_split_kernel, dim3((int)blocks), dim3(256), 0, s, out, in, M, G, C, h, w);
    return check_hip(hipGetLastError(), "block2x2 kernel");
}

int skip_chain_forward_g(float* out, const float* skip, const float* taps, int G, int C, int h, int w, int accumulate, hipStream_t s)
{
    if (bad_groups(G) || C < 0 || h < 0 || w < 0) { set_error("bad skip chain sizes"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)G * C * h * w;
    if (total == 0) return AG_OK;
    if (!out || !skip || !taps) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SkipTaps t;
    memcpy(&t, taps, sizeof(t));                                // host pointer: wy[24] then wx[24], passed to the kernel by value
    hipLaunchKernelGGL(skip_chain_forward_kernel, dim3((int)blocks), dim3(256), 0, s, out, skip, t, G, C, h, w, accumulate);
    return check_hip(hipGetLastError(), "skip_chain_forward_kernel");
}

int skip_chain_backward_g(float* gskip, const float* gout, const float* taps, int G, int C, int h, int w, hipStream_t s)
{
    if (bad_groups(G) || C < 0 || h < 0 || w < 0) { set_error("bad skip chain sizes"); return AG_ERR_INVALID_ARGUMENT; }
    const long long total = (long long)G * C * h * w;
    if (total == 0) return AG_OK;
    if (!gskip || !gout || !taps) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SkipTaps t;
    memcpy(&t, taps, sizeof(t));
    hipLaunchKernelGGL(skip_chain_backward_kernel, dim3((int)blocks), dim3(256), 0, s, gskip, gout, t, G, C, h, w);
    return check_hip(hipGetLastError(), "skip_chain_backward_kernel");
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
    if (noise && !noise_weight) { set_error("bad noise_bias_act arguments"); return AG_ERR_INVALID_ARGUMENT; }
    return noise_bias_act_forward_g(y, x, 1, table_of(noise), table_of(noise_weight), table_of(bias), C, HW, slope, scale,
                                    reinterpret_cast<hipStream_t>(stream));
}

size_t ag_noise_bias_act_partial_floats(int32_t C, int32_t HW) { return noise_bias_act_partial_floats(1, C, HW); }

int ag_noise_bias_act_backward(float* gx, const float* gy, const float* y, const float* noise, float* gbias, float* gnoise_weight,
                               float* partials, int32_t C, int32_t HW, float slope, float scale, void* stream)
{
    if (gnoise_weight && !noise) { set_error("bad noise_bias_act_backward arguments"); return AG_ERR_INVALID_ARGUMENT; }
    return noise_bias_act_backward_g(gx, gy, y, 1, table_of(gnoise_weight ? noise : nullptr), gbias, 0, gnoise_weight, 0, partials, C, HW, slope,
                                     scale, reinterpret_cast<hipStream_t>(stream));
}

int ag_modulate_weight_forward(float* out, float* dcoef, const float* W, const float* style, float scale, int32_t demodulate,
                               int32_t Co, int32_t Ci, int32_t K2, int32_t transposed, void* stream)
{
    return modulate_weight_forward_g(out, dcoef, 1, table_of(W), table_of(style), scale, demodulate, Co, Ci, K2, transposed,
                                     reinterpret_cast<hipStream_t>(stream));
}

size_t ag_modulate_weight_partial_floats(int32_t Co, int32_t Ci) { return modulate_weight_partial_floats(1, Co, Ci); }

int ag_modulate_weight_backward(float* dW, float* dstyle, float* partials, const float* g, const float* W, const float* style,
                                const float* dcoef, float scale, int32_t demodulate, int32_t Co, int32_t Ci, int32_t K2, int32_t transposed,
                                void* stream)
{
    return modulate_weight_backward_g(dW, dstyle, partials, g, 1, table_of(W), table_of(style), dcoef, scale, demodulate, Co, Ci, K2, transposed,
                                      reinterpret_cast<hipStream_t>(stream));
}

int ag_block2x2_transform(float* out, const float* in, const float* matrix16, int32_t merge, int32_t C, int32_t h, int32_t w,
                          void* stream)
{
    return block2x2_transform_g(out, in, matrix16, merge, 1, C, h, w, reinterpret_cast<hipStream_t>(stream));
}

int ag_skip_chain_forward(float* out, const float* skip, const float* taps, int32_t C, int32_t h, int32_t w, int32_t accumulate, void* stream)
{
    return skip_chain_forward_g(out, skip, taps, 1, C, h, w, accumulate, reinterpret_cast<hipStream_t>(stream));
}

int ag_skip_chain_backward(float* gskip, const float* gout, const float* taps, int32_t C, int32_t h, int32_t w, void* stream)
{
    return skip_chain_backward_g(gskip, gout, taps, 1, C, h, w, reinterpret_cast<hipStream_t>(stream));
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
        return launch_fir4x4(out, input, kernel, p, major, nullptr, reinterpret_cast<hipStream_t>(stream));
    }
    const long long total = (long long)major * p.out_h * p.out_w;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, input, kernel, p);
    return check_hip(hipGetLastError(), "upfirdn2d_kernel");
}

}  // extern "C"
