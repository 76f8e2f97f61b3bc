// EqualLinear layers on one-row inputs (the style path of the StyleUNets) and the bilinear resize of the view-direction feature
// (include/ag_linear.h).  gfx950, wave64.  All launch-latency- or HBM-bound: the point of these kernels is that a whole group of layers is ONE
// launch that reads the parameter tensors where they lie and writes the per-parameter gradients where autograd wants them.
#include "ag_common.h"
#include "../../include/ag_linear.h"
#include "../../include/ag_raster.h"

namespace ag {

typedef float lf4 __attribute__((ext_vector_type(4)));
constexpr int kLinRowsPerChunk = 16;       // backward: rows of a job whose g_x contributions one workgroup adds up (one partial row per chunk)
constexpr float kSqrt2 = 1.41421356237309504880f;

struct LinearLaunch {
    AgEqualLinearArgs a;
    int32_t row_begin[AG_LINEAR_MAX_JOBS + 1];       // forward: first output row (= column of y) of job j
    int32_t chunk_begin[AG_LINEAR_MAX_JOBS + 1];     // backward: first 16-row chunk of job j
    int32_t total_cols;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 1 / sqrt(mean(x^2) + 1e-8) of one input row (PixelNorm, dual_styleunet.py:13-18), by the whole wave
__device__ __forceinline__ float pixel_norm_factor(const float* __restrict__ x, int in, int lane)
{
    float s = 0.f;
    for (int c = lane * 4; c < in; c += 256) {
        const lf4 v = *reinterpret_cast<const lf4*>(x + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    return rsqrtf(s / (float)in + 1e-8f);
}

// ---- forward: one wave per output row -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) equal_linear_forward_kernel(LinearLaunch L)
{
    const AgEqualLinearArgs& a = L.a;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L.total_cols) return;
    int j = 0;
    for (int i = 1; i < a.n_jobs; i++) j = (row >= L.row_begin[i]) ? i : j;
    j = __builtin_amdgcn_readfirstlane(j);
    const int o = row - L.row_begin[j], in = a.in_features;
    const float* __restrict__ w = a.weight[j] + (size_t)o * in;
    const float bias = a.bias[j] ? a.bias[j][o] * a.bias_mul[j] : 0.f;
    for (int b = 0; b < a.B; b++) {
        const float* __restrict__ x = a.x[j] + (size_t)b * in;
        float s = 0.f;
        for (int c = lane * 4; c < in; c += 256) {
            const lf4 wv = *reinterpret_cast<const lf4*>(w + c), xv = *reinterpret_cast<const lf4*>(x + c);
            s = fmaf(wv[0], xv[0], fmaf(wv[1], xv[1], fmaf(wv[2], xv[2], fmaf(wv[3], xv[3], s))));
        }
        s = wave_sum(s);
        if (a.normalize_input) s *= pixel_norm_factor(x, in, lane);
        float y = fmaf(a.alpha[j], s, bias);
        if (a.act) y = (y > 0.f ? y : 0.2f * y) * kSqrt2;
        if (lane == 0) a.y[j][(size_t)b * a.out_features[j] + o] = y;
    }
}

// ---- backward, launch 1: one WORKGROUP per 16 rows of a job, threads along the columns -------------------------------------------------
// g_W rows and g_bias are final; the workgroup's share of g_x (sum over its rows of g' W) goes to partial[chunk][b][in].  (First form: one WAVE
// per 16 rows -- 384 waves for the 6144 rows of a decoder branch's modulation layers, 1.5 per CU: 22 us per launch for 25 MB.  Threads along
// columns, two each, give every chunk four waves.)
typedef float lf2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) equal_linear_backward_kernel(LinearLaunch L, int total_chunks)
{
    const AgEqualLinearArgs& a = L.a;
    __shared__ float s_g[kLinRowsPerChunk];
    const int tid = threadIdx.x, lane = tid & 63;
    const int chunk = blockIdx.x;
    int j = 0;
    for (int i = 1; i < a.n_jobs; i++) j = (chunk >= L.chunk_begin[i]) ? i : j;
    const int in = a.in_features, out = a.out_features[j];
    const int o0 = (chunk - L.chunk_begin[j]) * kLinRowsPerChunk, nr = min(out, o0 + kLinRowsPerChunk) - o0;
    const float alpha = a.alpha[j];
    const float* __restrict__ W = a.weight[j];
    float* __restrict__ gW = a.g_weight[j];
    float* __restrict__ gb = a.g_bias[j];
    const bool want_gx = a.g_x[j] != nullptr;
    for (int b = 0; b < a.B; b++) {
        const float* __restrict__ x = a.x[j] + (size_t)b * in;
        const float nf = a.normalize_input ? pixel_norm_factor(x, in, lane) : 1.f;     // (every wave forms it: 2 KB from L2)
        __syncthreads();                                  // s_g of the previous batch row is no longer read
        if (tid < nr) {
            const size_t at = (size_t)b * out + o0 + tid;
            float gp = a.g_y[j][at];
            if (a.act) gp *= (a.y[j][at] > 0.f) ? kSqrt2 : 0.2f * kSqrt2;
            s_g[tid] = gp;
            if (gb) {                                     // bias gradient: summed over the batch rows in order (b ascending)
                const float t = a.bias_mul[j] * gp;
                gb[o0 + tid] = (b == 0) ? t : gb[o0 + tid] + t;
            }
        }
        __syncthreads();
        for (int c = tid * 2; c < in; c += 512) {
            lf2 xv = *reinterpret_cast<const lf2*>(x + c);
            xv *= nf * alpha;
            lf2 acc = { 0.f, 0.f };
            for (int r = 0; r < nr; r++) {
                const float g = s_g[r];
                if (want_gx) acc += g * *reinterpret_cast<const lf2*>(W + (size_t)(o0 + r) * in + c);
                if (gW) {
                    lf2* dst = reinterpret_cast<lf2*>(gW + (size_t)(o0 + r) * in + c);
                    const lf2 t = g * xv;
                    *dst = (b == 0) ? t : *dst + t;
                }
            }
            if (want_gx) *reinterpret_cast<lf2*>(a.scratch + ((size_t)chunk * a.B + b) * in + c) = alpha * acc;
        }
    }
}

// ---- backward, launch 2: g_x = the partial rows of the jobs that share the input, added in a fixed order -----------------------------
struct LinearReduce {
    float* dst[AG_LINEAR_MAX_JOBS];
    int32_t c0[AG_LINEAR_MAX_JOBS], c1[AG_LINEAR_MAX_JOBS];      // chunk range of input group g
    const float* partial;
    int32_t B, in;
};

constexpr int kLinParts = 16;
__global__ void __launch_bounds__(64 * kLinParts) equal_linear_reduce_kernel(LinearReduce R)
{
    // grid (ceil(in / 64), B, groups); 64 columns x 16 interleaved partitions of the chunk list (24 dependent loads per thread for the 384
    // chunks of a decoder branch instead of 96 with four), combined by a fixed pairwise tree
    __shared__ float s_part[kLinParts][64];
    const int g = blockIdx.z, b = blockIdx.y;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < R.in)
        for (int k = R.c0[g] + q; k < R.c1[g]; k += kLinParts) s += R.partial[((size_t)k * R.B + b) * R.in + c];
    s_part[q][cl] = s;
    __syncthreads();
#pragma unroll
    for (int w = kLinParts / 2; w >= 1; w >>= 1) {
        if (q < w) s_part[q][cl] += s_part[q + w][cl];
        __syncthreads();
    }
    if (q == 0 && c < R.in) R.dst[g][(size_t)b * R.in + c] = s_part[0][cl];
}

// ---- bilinear resize (align_corners = False) -------------------------------------------------------------------------------------------
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int o, float scale, int n_in)
{
    // torch's area_pixel_compute_source_index (align_corners false, not cubic): max(0, scale * (o + 0.5) - 0.5)
    float src = scale * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + ((r.i0 < n_in - 1) ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

// forward: a thread writes 4 consecutive outputs of a row (one 16-byte store when the row length allows); the two source rows and the up to 8
// source columns come through L1
__global__ void __launch_bounds__(256) bilinear_forward_kernel(float* __restrict__ out, const float* __restrict__ in, int N, int H, int W, int OH, int OW,
                                                               float sy, float sx)
{
    // block (64, 4): threadIdx.y + 4 blockIdx.x = the output row (plane * OH + oy), threadIdx.x strides over the row's 4-output groups -- no
    // 64-bit divisions per element (the first form spent most of its time in them: 25 us for 33 MB)
    const int owq = (OW + 3) >> 2;
    const bool vec = (OW & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const long long rows = (long long)N * OH;
    for (long long row = (long long)blockIdx.x * 4 + threadIdx.y; row < rows; row += (long long)gridDim.x * 4) {
    const size_t n = (size_t)((unsigned)row / (unsigned)OH);            // (the launcher refuses more than 2^31 rows); wave-uniform
    const int oy = (int)((unsigned)row - (unsigned)n * (unsigned)OH);
    for (int q = threadIdx.x; q < owq; q += 64) {
        const Lerp ly = lerp_of(oy, sy, H);
        const float* r0 = in + (n * H + ly.i0) * (size_t)W;
        const float* r1 = in + (n * H + ly.i1) * (size_t)W;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ox = min(4 * q + k, OW - 1);
            const Lerp lx = lerp_of(ox, sx, W);
            v[k] = ly.l0 * (lx.l0 * r0[lx.i0] + lx.l1 * r0[lx.i1]) + ly.l1 * (lx.l0 * r1[lx.i0] + lx.l1 * r1[lx.i1]);
        }
        float* dst = out + (n * OH + oy) * (size_t)OW + 4 * q;
        if (vec) *reinterpret_cast<lf4*>(dst) = lf4{ v[0], v[1], v[2], v[3] };
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (4 * q + k < OW) dst[k] = v[k];
        }
    }
    }
}

// weight with which output o reads input i along one axis (0 when it does not): lerp_of's two weights are the tent max(0, 1 - |src - i|) around
// the clamped source coordinate, except at the upper border, where the clamped upper neighbour folds both weights onto the last input
// (8 instructions per candidate against 14 for a full lerp_of and two compares; the backward is bound by these)
__device__ __forceinline__ float axis_weight(int o, int i, float scale, int n_in)
{
    float src = scale * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const float fi = (float)i;
    const float tent = fmaxf(0.f, 1.0f - fabsf(src - fi));
    return (i == n_in - 1 && src >= fi) ? 1.0f : tent;
}

// outputs that can read input i: source coordinates in (i - 1, i + 1), i.e. o in ((i - 0.5) / scale - 0.5, (i + 1.5) / scale - 0.5); floor / ceil of the
// bounds keep a candidate of margin on either side of the open interval (DESIGN: a candidate outside reads nothing, its weight is 0)
__device__ __forceinline__ void reader_range(int i, float scale, int n_out, int& lo, int& hi)
{
    const float inv = 1.0f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f);
    hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f);
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

// backward (the adjoint as a GATHER: fixed order, no atomics): a thread owns one input element and sums w_y(oy) w_x(ox) g[oy][ox] over the outputs
// that read it.  The candidate weights of both axes are formed once (<= kBilinearTaps per axis: up-sampling factors up to 2 and every
// down-sampling one; wider kernels take the loop form below).
constexpr int kBilinearTaps = 6;
__global__ void __launch_bounds__(256) bilinear_backward_kernel(float* __restrict__ g_in, const float* __restrict__ g_out, int N, int H, int W, int OH,
                                                                int OW, float sy, float sx)
{
    const long long rows = (long long)N * H;
    for (long long row = (long long)blockIdx.x * 4 + threadIdx.y; row < rows; row += (long long)gridDim.x * 4) {
    const size_t n = (size_t)((unsigned)row / (unsigned)H);
    const int iy = (int)((unsigned)row - (unsigned)n * (unsigned)H);
    for (int ix = threadIdx.x; ix < W; ix += 64) {
        const size_t e = (size_t)row * W + ix;
        int y0, y1, x0, x1;
        reader_range(iy, sy, OH, y0, y1);
        reader_range(ix, sx, OW, x0, x1);
        const float* g = g_out + n * (size_t)OH * OW;
        float s = 0.f;
        if (y1 - y0 < kBilinearTaps && x1 - x0 < kBilinearTaps) {
            float wx[kBilinearTaps];
#pragma unroll
            for (int k = 0; k < kBilinearTaps; k++) wx[k] = (x0 + k <= x1) ? axis_weight(x0 + k, ix, sx, W) : 0.f;
#pragma unroll
            for (int j = 0; j < kBilinearTaps; j++) {
                const int oy = y0 + j;
                const float wy = (oy <= y1) ? axis_weight(oy, iy, sy, H) : 0.f;
                if (wy == 0.f) continue;
                const float* row = g + (size_t)oy * OW + x0;
                float r = 0.f;
#pragma unroll
                for (int k = 0; k < kBilinearTaps; k++)
                    if (wx[k] != 0.f) r = fmaf(wx[k], row[k], r);
                s = fmaf(wy, r, s);
            }
        } else {
            for (int oy = y0; oy <= y1; oy++) {
                const float wy = axis_weight(oy, iy, sy, H);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = x0; ox <= x1; ox++) {
                    const float wx = axis_weight(ox, ix, sx, W);
                    if (wx != 0.f) row = fmaf(wx, g[(size_t)oy * OW + ox], row);
                }
                s = fmaf(wy, row, s);
            }
        }
        g_in[e] = s;
    }
    }
}

// ---- x[m] = out[src[m]] (+ bilinear resize of the member's view feature) ---------------------------------------------------------------
struct SelectAdd {
    float* x; const float* out; const float* vf;
    int src[16];
    int M, C, H, W, r0, r1, vh, vw;
    float sy, sx;
};

__global__ void __launch_bounds__(256) select_add_rows_kernel(SelectAdd p)
{
    // block (64, 4): threadIdx.y + 4 blockIdx.x = row of a plane (plane = m * C + c), threadIdx.x strides over the row's 4-pixel groups
    const int wq = (p.W + 3) >> 2;
    const bool vec = (p.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.out)) & 15) == 0;
    const long long rows = (long long)p.M * p.C * p.H;
    for (long long row = (long long)blockIdx.x * 4 + threadIdx.y; row < rows; row += (long long)gridDim.x * 4) {
        const unsigned plane = (unsigned)row / (unsigned)p.H, y = (unsigned)row - plane * (unsigned)p.H;
        const unsigned m = plane / (unsigned)p.C, c = plane - m * (unsigned)p.C;
        const float* __restrict__ srow = p.out + (((size_t)p.src[m] * p.C + c) * p.H + y) * p.W;
        float* __restrict__ drow = p.x + (size_t)row * p.W;
        const bool add = p.vf && (int)m >= p.r0 && (int)m < p.r1;
        const float* __restrict__ fplane = add ? p.vf + ((size_t)(m - p.r0) * p.C + c) * p.vh * p.vw : nullptr;
        const bool same = p.vh == p.H && p.vw == p.W;
        Lerp ly{};
        if (add && !same) ly = lerp_of((int)y, p.sy, p.vh);
        for (int q = threadIdx.x; q < wq; q += 64) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = min(4 * q + k, p.W - 1);
                float a = vec ? 0.f : srow[xx];
                if (add) {
                    if (same) a += fplane[(size_t)y * p.vw + xx];
                    else {
                        const Lerp lx = lerp_of(xx, p.sx, p.vw);
                        const float* r0p = fplane + (size_t)ly.i0 * p.vw;
                        const float* r1p = fplane + (size_t)ly.i1 * p.vw;
                        a += ly.l0 * (lx.l0 * r0p[lx.i0] + lx.l1 * r0p[lx.i1]) + ly.l1 * (lx.l0 * r1p[lx.i0] + lx.l1 * r1p[lx.i1]);
                    }
                }
                v[k] = a;
            }
            if (vec) {
                const lf4 s4 = *reinterpret_cast<const lf4*>(srow + 4 * q);
                *reinterpret_cast<lf4*>(drow + 4 * q) = lf4{ s4[0] + v[0], s4[1] + v[1], s4[2] + v[2], s4[3] + v[3] };
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) if (4 * q + k < p.W) drow[4 * q + k] = v[k];
            }
        }
    }
}

// ---- plane sums -------------------------------------------------------------------------------------------------------------------
constexpr int kPlaneSliceMin = 4096;      // floats per slice at least (a workgroup's 256 threads x 4 float4)
__host__ __device__ inline int plane_slices(int planes, long long len)
{
    long long s = (4096 + planes - 1) / planes;                   // ~4096 workgroups in all
    const long long most = (len + kPlaneSliceMin - 1) / kPlaneSliceMin;
    s = s > most ? most : s;
    return (int)(s < 1 ? 1 : s);
}

__global__ void __launch_bounds__(256) plane_partial_sums_kernel(float* __restrict__ partial, const float* __restrict__ in, long long len, int S)
{
    __shared__ float s_w[4];
    const int sl = blockIdx.x, p = blockIdx.y;
    const long long per = ((len + S - 1) / S + 3) & ~3LL;          // slice length, a multiple of 4
    const long long e0 = (long long)sl * per, e1 = e0 + per < len ? e0 + per : len;
    const float* __restrict__ x = in + (size_t)p * len;
    float s = 0.f;
    if ((((uintptr_t)x) & 15) == 0) {
        long long i = e0 + (long long)threadIdx.x * 4;
        for (; i + 3 < e1; i += 1024) {
            const lf4 v = *reinterpret_cast<const lf4*>(x + i);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; i < e1; i++) s += x[i];                              // the plane's ragged end (one thread)
    } else {
        for (long long i = e0 + threadIdx.x; i < e1; i += 256) s += x[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)p * S + sl] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(256) plane_final_sums_kernel(float* __restrict__ out, const float* __restrict__ partial, int planes, int S)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= planes) return;
    float s = 0.f;
    for (int k = 0; k < S; k++) s += partial[(size_t)p * S + k];
    out[p] = s;
}

}  // namespace ag

using namespace ag;

namespace {

bool prepare(const AgEqualLinearArgs* a, LinearLaunch& L, int& total_chunks, const char* who)
{
    if (!a || a->n_jobs < 1 || a->n_jobs > AG_LINEAR_MAX_JOBS || a->B < 1 || a->B > 8 || a->in_features < 4 || (a->in_features & 3)) {
        set_error("%s: bad job count / batch / in_features (a multiple of 4)", who);
        return false;
    }
    L.a = *a;
    long long rows = 0, chunks = 0;
    for (int j = 0; j < a->n_jobs; j++) {
        if (!a->x[j] || !a->weight[j] || !a->y[j] || a->out_features[j] < 1) { set_error("%s: job %d without input / weight / output / rows", who, j); return false; }
        if ((reinterpret_cast<uintptr_t>(a->x[j]) | reinterpret_cast<uintptr_t>(a->weight[j])) & 15) { set_error("%s: job %d: input / weight not 16-byte aligned", who, j); return false; }
        L.row_begin[j] = (int32_t)rows;
        L.chunk_begin[j] = (int32_t)chunks;
        rows += a->out_features[j];
        chunks += (a->out_features[j] + kLinRowsPerChunk - 1) / kLinRowsPerChunk;
    }
    if (rows > 0x3fffffffLL) { set_error("%s: too many rows", who); return false; }
    for (int j = a->n_jobs; j <= AG_LINEAR_MAX_JOBS; j++) { L.row_begin[j] = (int32_t)rows; L.chunk_begin[j] = (int32_t)chunks; }
    L.total_cols = (int32_t)rows;
    total_chunks = (int)chunks;
    return true;
}

}  // namespace

extern "C" {

size_t ag_equal_linear_args_bytes(void) { return sizeof(AgEqualLinearArgs); }

size_t ag_equal_linear_scratch_floats(const AgEqualLinearArgs* a)
{
    LinearLaunch L;
    int chunks = 0;
    if (!prepare(a, L, chunks, "ag_equal_linear_scratch_floats")) return 0;
    return (size_t)chunks * a->B * a->in_features + 64;
}

int ag_equal_linear_forward(const AgEqualLinearArgs* a, void* stream)
{
    LinearLaunch L;
    int chunks = 0;
    if (!prepare(a, L, chunks, "ag_equal_linear_forward")) return AG_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(equal_linear_forward_kernel, dim3((L.total_cols + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), L);
    return check_hip(hipGetLastError(), "equal_linear_forward_kernel");
}

int ag_equal_linear_backward(const AgEqualLinearArgs* a, void* stream)
{
    LinearLaunch L;
    int chunks = 0;
    if (!prepare(a, L, chunks, "ag_equal_linear_backward")) return AG_ERR_INVALID_ARGUMENT;
    for (int j = 0; j < a->n_jobs; j++)
        if (!a->g_y[j]) { set_error("ag_equal_linear_backward: job %d without g_y", j); return AG_ERR_INVALID_ARGUMENT; }
    // input groups: consecutive jobs with the same x share one g_x
    LinearReduce R{};
    int groups = 0;
    bool any_gx = false;
    for (int j = 0; j < a->n_jobs; j++) {
        if (a->g_weight[j] && (reinterpret_cast<uintptr_t>(a->g_weight[j]) & 15)) { set_error("ag_equal_linear_backward: g_weight not 16-byte aligned"); return AG_ERR_INVALID_ARGUMENT; }
        const bool cont = j > 0 && a->x[j] == a->x[j - 1];
        if (cont && a->g_x[j] != a->g_x[j - 1]) { set_error("ag_equal_linear_backward: jobs of one input must pass one g_x"); return AG_ERR_INVALID_ARGUMENT; }
        if (!a->g_x[j]) continue;
        any_gx = true;
        if (cont) { R.c1[groups - 1] = L.chunk_begin[j + 1]; continue; }
        R.dst[groups] = a->g_x[j]; R.c0[groups] = L.chunk_begin[j]; R.c1[groups] = L.chunk_begin[j + 1];
        groups++;
    }
    if (any_gx && a->normalize_input) { set_error("ag_equal_linear_backward: no input gradient through the PixelNorm option"); return AG_ERR_INVALID_ARGUMENT; }
    if (any_gx && !a->scratch) { set_error("ag_equal_linear_backward: g_x needs scratch"); return AG_ERR_INVALID_ARGUMENT; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(equal_linear_backward_kernel, dim3(chunks), dim3(256), 0, s, L, chunks);
    if (check_hip(hipGetLastError(), "equal_linear_backward_kernel")) return AG_ERR_HIP;
    if (groups) {
        R.partial = a->scratch; R.B = a->B; R.in = a->in_features;
        hipLaunchKernelGGL(equal_linear_reduce_kernel, dim3((a->in_features + 63) / 64, a->B, groups), dim3(64 * kLinParts), 0, s, R);
        if (check_hip(hipGetLastError(), "equal_linear_reduce_kernel")) return AG_ERR_HIP;
    }
    return AG_OK;
}

int ag_bilinear_resize_forward(float* out, const float* in, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream)
{
    if (!out || !in || N < 1 || H < 1 || W < 1 || OH < 1 || OW < 1 || (long long)N * OH > 0x7fffffffLL || (long long)N * H > 0x7fffffffLL) { set_error("ag_bilinear_resize_forward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const long long rows = (long long)N * OH;
    const unsigned grid = (unsigned)((rows + 3) / 4 < 65536 * 16 ? (rows + 3) / 4 : 65536 * 16);
    hipLaunchKernelGGL(bilinear_forward_kernel, dim3(grid), dim3(64, 4), 0, reinterpret_cast<hipStream_t>(stream), out, in, N, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    return check_hip(hipGetLastError(), "bilinear_forward_kernel");
}

int ag_bilinear_resize_backward(float* g_in, const float* g_out, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream)
{
    if (!g_in || !g_out || N < 1 || H < 1 || W < 1 || OH < 1 || OW < 1 || (long long)N * OH > 0x7fffffffLL || (long long)N * H > 0x7fffffffLL) { set_error("ag_bilinear_resize_backward: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const long long rows = (long long)N * H;
    const unsigned grid = (unsigned)((rows + 3) / 4 < 65536 * 16 ? (rows + 3) / 4 : 65536 * 16);
    hipLaunchKernelGGL(bilinear_backward_kernel, dim3(grid), dim3(64, 4), 0, reinterpret_cast<hipStream_t>(stream), g_in, g_out, N, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    return check_hip(hipGetLastError(), "bilinear_backward_kernel");
}

size_t ag_plane_sums_scratch_floats(int32_t planes, int64_t len)
{
    if (planes < 1 || len < 1) return 0;
    return (size_t)planes * plane_slices(planes, len) + 64;
}

int ag_plane_sums(float* out, const float* in, int32_t planes, int64_t len, float* scratch, void* stream)
{
    if (!out || !in || !scratch || planes < 1 || planes > 65535 || len < 1) { set_error("ag_plane_sums: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    const int S = plane_slices(planes, len);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(plane_partial_sums_kernel, dim3(S, planes), dim3(256), 0, s, scratch, in, (long long)len, S);
    if (check_hip(hipGetLastError(), "plane_partial_sums_kernel")) return AG_ERR_HIP;
    hipLaunchKernelGGL(plane_final_sums_kernel, dim3((planes + 255) / 256), dim3(256), 0, s, out, scratch, planes, S);
    return check_hip(hipGetLastError(), "plane_final_sums_kernel");
}

int ag_select_add_rows(float* x, const float* out, const int32_t* src, int32_t M, int32_t C, int32_t H, int32_t W, const float* vf, int32_t r0, int32_t r1,
                       int32_t vh, int32_t vw, void* stream)
{
    if (!x || !out || !src || M < 1 || M > 16 || C < 1 || H < 1 || W < 1 || (long long)M * C * H > 0x7fffffffLL ||
        (vf && (r0 < 0 || r1 > M || r0 >= r1 || vh < 1 || vw < 1))) { set_error("ag_select_add_rows: bad arguments"); return AG_ERR_INVALID_ARGUMENT; }
    SelectAdd p{};
    p.x = x; p.out = out; p.vf = vf; p.M = M; p.C = C; p.H = H; p.W = W; p.r0 = r0; p.r1 = r1; p.vh = vf ? vh : H; p.vw = vf ? vw : W;
    for (int m = 0; m < M; m++) {
        if (src[m] < 0) { set_error("ag_select_add_rows: negative source row"); return AG_ERR_INVALID_ARGUMENT; }
        p.src[m] = src[m];
    }
    p.sy = (float)p.vh / (float)H; p.sx = (float)p.vw / (float)W;
    const long long rows = (long long)M * C * H;
    const unsigned grid = (unsigned)((rows + 3) / 4 < 65536 * 16 ? (rows + 3) / 4 : 65536 * 16);
    hipLaunchKernelGGL(select_add_rows_kernel, dim3(grid), dim3(64, 4), 0, reinterpret_cast<hipStream_t>(stream), p);
    return check_hip(hipGetLastError(), "select_add_rows_kernel");
}

}  // extern "C"
