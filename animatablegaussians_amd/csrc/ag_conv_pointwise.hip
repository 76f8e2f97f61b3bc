// 1x1 convolutions with a tiny channel count on one side (gfx950): the StyleUNet's ToRGB heads (C = 64..512 -> M = 12 or 32
// wavelet channels, dual_styleunet.py:607-633) and FromRGB convolutions (3 -> 128..512, :442-470), forward and gradients.
//
// As implicit GEMMs on the 128 x 128 MFMA tiles (ag_conv.hip) these problems pad 12 rows to 64 or 3 channels to 16 and are pure
// launch latency: 28-140 us each for <= 0.4 GFLOP, 36 launches per network pass = 8 % of its time (profiles/r02b_conv_layers.csv).
// They are streaming problems -- every input element is used 12 (or 3) times -- so they run here as HBM-bound VALU kernels with the
// small weight matrix in LDS:
//   small-M dgrad     dx[c][n] = sum_m w[m][c] dy[m][n]                                 thread = 4 (2) pixels, dy rows in registers
//   small-M wgrad     dw[m][c] = sum_n dy[m][n] x[c][n]                                 wave = 4 (2) channels x a pixel slice, lanes along
//                                                                                       pixels, dy shared through LDS, DPP wave reduction
//   small-C forward   (C <= 4)                                                          thread = 4 pixels, loops over the output rows
//   small-C wgrad     dw[m][c] = sum_n dy[m][n] x[c][n]                                 workgroup = one output row, no atomics
// fp32 FMA chains in channel / pixel order: same operation count as the reference's cuDNN call, different summation order.
#include "ag_common.h"
#include "ag_groups.h"
#include "../../include/ag_conv.h"

namespace ag {

typedef float pf4 __attribute__((ext_vector_type(4)));
typedef float pf2 __attribute__((ext_vector_type(2)));

template <int PX> struct PixVec;
template <> struct PixVec<4> { typedef pf4 type; };
template <> struct PixVec<2> { typedef pf2 type; };

// Grouped (ag_groups.h): blockIdx.z = instance; tensors at the *_gs float strides, weights / biases from the tables.
struct PwProblem {
    const float* x;          // [C][N]
    const float* dy;         // [M][N]
    const float* out_scale;  // [M] or null (single instance only)
    float* y;                // forward: [M][N]; dgrad: dx [C][N]; wgrad: dw [M][C]
    int C, M, N;
    float wscale;
    int slice;               // wgrad: pixels per workgroup slice (multiple of 256)
    float* partial;          // wgrad: [G][slices][M][C]
    long long x_gs, dy_gs, y_gs, partial_gs;
    PtrTable w_t;            // forward / dgrad: [M][C] per instance
    PtrTable bias_t;         // [M] or null per instance
};

// the instance's view of the problem (uniform per workgroup)
__device__ __forceinline__ void pw_select(PwProblem& p, const float*& w, const float*& bias)
{
    const int g = blockIdx.z;
    if (p.x) p.x += (size_t)g * p.x_gs;
    if (p.dy) p.dy += (size_t)g * p.dy_gs;
    p.y += (size_t)g * p.y_gs;
    if (p.partial) p.partial += (size_t)g * p.partial_gs;
    w = p.w_t.p[g];
    bias = p.bias_t.p[g];
}

constexpr int kPwMaxLds = 64 * 1024;

// ---- small M ---------------------------------------------------------------------------------------------------------------------
// Weights in LDS as [c][MT] (one broadcast read of MT consecutive floats per channel).
template <int MT>
__device__ __forceinline__ void stage_weights_cm(const PwProblem& p, const float* __restrict__ w, float* sw)
{
    // source order (m-major rows of C consecutive floats: coalesced), transposed on the way into LDS
    for (int i = threadIdx.x; i < p.C * MT; i += blockDim.x) {
        const int m = i / p.C, c = i - m * p.C;
        sw[c * MT + m] = (m < p.M) ? w[i] * p.wscale : 0.f;
    }
    __syncthreads();
}

template <int MT, int PX>
__global__ void __launch_bounds__(256) pw_small_m_dgrad_kernel(PwProblem p)
{
    typedef typename PixVec<PX>::type V;
    extern __shared__ float sw[];
    const float *w, *bias_unused;
    pw_select(p, w, bias_unused);
    stage_weights_cm<MT>(p, w, sw);
    const int n = (blockIdx.x * 256 + threadIdx.x) * PX;
    if (n >= p.N) return;
    V g[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) g[m] = (m < p.M) ? *reinterpret_cast<const V*>(p.dy + (size_t)m * p.N + n) : V(0.f);
    float* op = p.y + n;
#pragma unroll 2
    for (int c = 0; c < p.C; c++) {
        const float* wr = sw + c * MT;
        V v = V(0.f);
#pragma unroll
        for (int m = 0; m < MT; m++) v += wr[m] * g[m];
        *reinterpret_cast<V*>(op + (size_t)c * p.N) = v;
    }
}

// sum over the 64 lanes, result valid in lane 63: four row shifts, then the two cross-row broadcasts of the wave64 DPP reduction
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));   // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, true));   // row_bcast:31 -> rows 2, 3
    return v;
}

// workgroup = (16 or 8 channels, one pixel slice): its 4 waves take CB channels each; 256 pixels of dy at a time go through LDS and
// are shared by the waves; lanes run along pixels (coalesced rows of x and dy).  Every workgroup writes its [M][channels] partial sums
// to partial[slice] (a float atomic per (m, c) and workgroup piles 256 same-address atomics on each of the M * C words: 149 us at
// 64 -> 12 @512^2, profiles/r02_pointwise_kernels.log); pw_reduce_slices_kernel adds the slices in a fixed order.
template <int MT, int CB>
__global__ void __launch_bounds__(256) pw_small_m_wgrad_kernel(PwProblem p)
{
    __shared__ float sdy[MT][256];
    const float *w_unused, *bias_unused;
    pw_select(p, w_unused, bias_unused);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = (blockIdx.y * 4 + wave) * CB;
    const int nbeg = blockIdx.x * p.slice, nend = min(p.N, nbeg + p.slice);
    float acc[MT][CB];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < CB; j++) acc[m][j] = 0.f;
    const float* xrow[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) xrow[j] = p.x + (size_t)min(c0 + j, p.C - 1) * p.N;
    for (int n0 = nbeg; n0 < nend; n0 += 256) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; m++) sdy[m][tid] = (m < p.M && n0 + tid < nend) ? p.dy[(size_t)m * p.N + n0 + tid] : 0.f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int nl = q * 64 + lane, n = n0 + nl;
            float xv[CB];
#pragma unroll
            for (int j = 0; j < CB; j++) xv[j] = (n < nend) ? xrow[j][n] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float g = sdy[m][nl];
#pragma unroll
                for (int j = 0; j < CB; j++) acc[m][j] += g * xv[j];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < CB; j++) {
            const float s = wave_sum_to_lane63(acc[m][j]);
            if (lane == 63 && m < p.M && c0 + j < p.C) p.partial[((size_t)blockIdx.x * p.M + m) * p.C + c0 + j] = s;
        }
}

// one wave per output element: lanes stride over the slices, fixed-order DPP sum
__global__ void __launch_bounds__(256) pw_reduce_slices_kernel(const float* __restrict__ partial, float* __restrict__ dw, int MC, int slices, float wscale,
                                                               long long partial_gs, long long dw_gs)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= MC) return;
    partial += (size_t)blockIdx.y * partial_gs;
    dw += (size_t)blockIdx.y * dw_gs;
    float v = 0.f;
    for (int s = lane; s < slices; s += 64) v += partial[(size_t)s * MC + i];
    v = wave_sum_to_lane63(v);
    if (lane == 63) dw[i] = v * wscale;
}

// ---- small C (<= 4) --------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pw_small_c_forward_kernel(PwProblem p, int m_per_block)
{
    extern __shared__ float sw[];     // [m_per_block][4] weights, then out_scale and bias
    const float *w, *bias;
    pw_select(p, w, bias);
    const int mbeg = blockIdx.y * m_per_block, mcnt = min(m_per_block, p.M - mbeg);
    for (int i = threadIdx.x; i < mcnt * 4; i += 256) {
        const int m = i >> 2, c = i & 3;
        sw[i] = (c < p.C) ? w[(size_t)(mbeg + m) * p.C + c] * p.wscale : 0.f;
    }
    __syncthreads();
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= p.N) return;
    pf4 xv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) xv[c] = (c < p.C) ? *reinterpret_cast<const pf4*>(p.x + (size_t)c * p.N + n) : pf4(0.f);
    for (int m = 0; m < mcnt; m++) {
        const pf4 wv = *reinterpret_cast<const pf4*>(sw + 4 * m);
        pf4 v = wv[0] * xv[0];
        v += wv[1] * xv[1];
        v += wv[2] * xv[2];
        v += wv[3] * xv[3];
        if (p.out_scale) v *= p.out_scale[mbeg + m];
        if (bias) v += bias[mbeg + m];
        *reinterpret_cast<pf4*>(p.y + (size_t)(mbeg + m) * p.N + n) = v;
    }
}

// workgroup = one output row m: dw[m][c] = wscale * sum_n dy[m][n] x[c][n]; every element written exactly once (no memset, no atomics)
__global__ void __launch_bounds__(256) pw_small_c_wgrad_kernel(PwProblem p)
{
    __shared__ float red[4][4];
    const float *w_unused, *bias_unused;
    pw_select(p, w_unused, bias_unused);
    const int m = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* g = p.dy + (size_t)m * p.N;
    pf4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = pf4(0.f);
    for (int n = tid * 4; n < p.N; n += 1024) {
        const pf4 gv = *reinterpret_cast<const pf4*>(g + n);
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (c < p.C) acc[c] += gv * *reinterpret_cast<const pf4*>(p.x + (size_t)c * p.N + n);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float s = wave_sum_to_lane63(acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3]);
        if (lane == 63) red[wave][c] = s;
    }
    __syncthreads();
    if (tid < p.C) p.y[(size_t)m * p.C + tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * p.wscale;
}

// ---- dispatch ---------------------------------------------------------------------------------------------------------------------
// Which problems run here was decided on kernel durations (rocprofv3, profiles/r02_pointwise_kernels.log), against the MFMA path's
// pack + implicit-GEMM + split-K finish for the same problem:
//   small C (FromRGB) forward: always (5-15 us against 17-38);   weight gradient: up to 128^2 (one workgroup per output row re-reads x)
//   small M <= 16 (ToRGB of the colour / position networks) input / weight gradient: from 128^2 up (17-33 us against 39-105); below
//   that, for 32 rows (the other network: 47 / 159 us here -- 64 accumulators per lane leave no room to hide the loads) and for the
//   forward (17 us on the MFMA tiles at 512^2: the 12 output rows ride along in a 64-row tile), the MFMA path stays.
static bool pw_applicable(const AgConvDesc* d)
{
    return d->kind == AG_CONV && d->k == 1 && d->stride == 1 && d->padding == 0 && ((d->H * d->W) & 3) == 0;
}

static float pw_wscale(const AgConvDesc* d) { return d->weight_scale == 0.f ? 1.f : d->weight_scale; }
constexpr int kPwLargeN = 128 * 128;

// 0: not handled here (the caller runs the MFMA path); 1: launched; < 0: error
int pointwise_forward(const AgConvDesc* d, int G, const float* x, long long x_gs, const PtrTable& w, const float* out_scale, const PtrTable& bias,
                      float* y, long long y_gs, hipStream_t s)
{
    if (!pw_applicable(d) || d->Cin > 4) return 0;
    if (G > 1 && out_scale) return 0;
    PwProblem p{};
    p.x = x; p.w_t = w; p.out_scale = out_scale; p.bias_t = bias; p.y = y; p.x_gs = x_gs; p.y_gs = y_gs;
    p.C = d->Cin; p.M = d->Cout; p.N = d->H * d->W; p.wscale = pw_wscale(d);
    const int tiles_n = (p.N / 4 + 255) / 256;
    // enough workgroups to fill the chip: split the output rows when the image is small
    int groups = 1;
    while (tiles_n * groups * G < 512 && p.M / (groups * 2) >= 16) groups *= 2;
    const int mpb = (p.M + groups - 1) / groups;
    hipLaunchKernelGGL(pw_small_c_forward_kernel, dim3(tiles_n, (p.M + mpb - 1) / mpb, G), dim3(256), (size_t)mpb * 4 * sizeof(float), s, p, mpb);
    return check_hip(hipGetLastError(), "pw_small_c_forward_kernel") ? AG_ERR_HIP : 1;
}

int pointwise_backward_input(const AgConvDesc* d, int G, const float* dy, long long dy_gs, const PtrTable& w, float* dx, long long dx_gs, hipStream_t s)
{
    if (!pw_applicable(d) || d->Cout > 16 || d->Cin <= 4 || d->H * d->W < kPwLargeN) return 0;
    PwProblem p{};
    p.dy = dy; p.w_t = w; p.y = dx; p.dy_gs = dy_gs; p.y_gs = dx_gs; p.C = d->Cin; p.M = d->Cout; p.N = d->H * d->W; p.wscale = pw_wscale(d);
    const int MT = p.M <= 12 ? 12 : p.M <= 16 ? 16 : 32;
    const size_t lds = (size_t)p.C * MT * sizeof(float);
    if (lds > (size_t)kPwMaxLds) return 0;
    if (MT == 12)      hipLaunchKernelGGL((pw_small_m_dgrad_kernel<12, 4>), dim3((p.N / 4 + 255) / 256, 1, G), dim3(256), lds, s, p);
    else if (MT == 16) hipLaunchKernelGGL((pw_small_m_dgrad_kernel<16, 4>), dim3((p.N / 4 + 255) / 256, 1, G), dim3(256), lds, s, p);
    else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_small_m_dgrad_kernel<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kPwMaxLds);
        hipLaunchKernelGGL((pw_small_m_dgrad_kernel<32, 2>), dim3((p.N / 2 + 255) / 256, 1, G), dim3(256), lds, s, p);
    }
    return check_hip(hipGetLastError(), "pw_small_m_dgrad_kernel") ? AG_ERR_HIP : 1;
}

int pointwise_backward_weight(const AgConvDesc* d, int G, const float* x, long long x_gs, const float* dy, long long dy_gs, float* dw, long long dw_gs,
                              void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (!pw_applicable(d)) return 0;
    PwProblem p{};
    p.x = x; p.dy = dy; p.y = dw; p.x_gs = x_gs; p.dy_gs = dy_gs; p.y_gs = dw_gs;
    p.C = d->Cin; p.M = d->Cout; p.N = d->H * d->W; p.wscale = pw_wscale(d);
    if (G > 1 && dw_gs == 0) dw_gs = p.y_gs = (long long)p.M * p.C;
    if (p.M <= 16 && p.C > 4 && p.N >= kPwLargeN) {
        const int MT = p.M <= 12 ? 12 : p.M <= 16 ? 16 : 32;
        const int CB = MT == 32 ? 2 : 4;
        const int cgroups = (p.C + 4 * CB - 1) / (4 * CB);
        // pixel slices: ~1024 workgroups in total, at least 1024 pixels each
        int slices = (1024 + cgroups * G - 1) / (cgroups * G);
        const int max_slices = (p.N + 1023) / 1024;
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        p.slice = (((p.N + slices - 1) / slices) + 255) / 256 * 256;
        slices = (p.N + p.slice - 1) / p.slice;
        p.partial_gs = (long long)slices * p.M * p.C;
        const size_t need = (size_t)G * p.partial_gs * sizeof(float) + 256;
        if (!workspace || workspace_bytes < need) return 0;
        p.partial = reinterpret_cast<float*>(aligned_base(workspace));
        float* const partial0 = p.partial;
        // the slice kernel writes partial sums, not dw: its output stride is the partial image's
        p.y = nullptr; p.y_gs = 0;
        if (MT == 12)      hipLaunchKernelGGL((pw_small_m_wgrad_kernel<12, 4>), dim3(slices, cgroups, G), dim3(256), 0, s, p);
        else if (MT == 16) hipLaunchKernelGGL((pw_small_m_wgrad_kernel<16, 4>), dim3(slices, cgroups, G), dim3(256), 0, s, p);
        else               hipLaunchKernelGGL((pw_small_m_wgrad_kernel<32, 2>), dim3(slices, cgroups, G), dim3(256), 0, s, p);
        if (check_hip(hipGetLastError(), "pw_small_m_wgrad_kernel")) return AG_ERR_HIP;
        const int MC = p.M * p.C;
        hipLaunchKernelGGL(pw_reduce_slices_kernel, dim3((MC + 3) / 4, G), dim3(256), 0, s, partial0, dw, MC, slices, p.wscale, p.partial_gs, dw_gs);
        return check_hip(hipGetLastError(), "pw_reduce_slices_kernel") ? AG_ERR_HIP : 1;
    }
    if (p.C <= 4 && p.N <= kPwLargeN) {
        hipLaunchKernelGGL(pw_small_c_wgrad_kernel, dim3(p.M, 1, G), dim3(256), 0, s, p);
        return check_hip(hipGetLastError(), "pw_small_c_wgrad_kernel") ? AG_ERR_HIP : 1;
    }
    return 0;
}

}  // namespace ag
