// Fused multi-tensor Adam (include/ag_optim.h).  HBM-bound streaming kernel: 16 B read per parameter (param, grad, exp_avg, exp_avg_sq) + 12 B written.
// One workgroup per 4096-element chunk of one tensor; the tensor of a workgroup comes from a 49-entry prefix table in the kernel arguments (scalar
// loads, uniform search); lanes move 16 bytes per load when the tensor's four base pointers are 16-byte aligned, scalars otherwise and in the tail.
#include "ag_common.h"
#include "../../include/ag_optim.h"
#include "../../include/ag_raster.h"

namespace ag {

constexpr int kAdamChunk = 4096;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdamLaunch {
    AgAdamArgs a;
    int32_t chunk_begin[AG_ADAM_MAX_TENSORS + 1];
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AgAdamArgs& a, float step_size, float bc2_sqrt)
{
    if (a.maximize) g = -g;
    if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
    m = fmaf(a.one_minus_beta1, g - m, m);
    v = fmaf(a.beta2, v, a.one_minus_beta2 * g * g);
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p -= step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(AdamLaunch L)
{
    const AgAdamArgs& a = L.a;
    const int b = blockIdx.x;
    int t = 0;
    for (int i = 1; i < a.n; i++) t = (b >= L.chunk_begin[i]) ? i : t;        // uniform: scalar compares on kernel arguments
    const int64_t n = a.numel[t];
    const int64_t e0 = (int64_t)(b - L.chunk_begin[t]) * kAdamChunk;
    const int64_t e1 = e0 + kAdamChunk < n ? e0 + kAdamChunk : n;
    float* __restrict__ P = a.param[t];
    const float* __restrict__ Gr = a.grad[t];
    float* __restrict__ M = a.exp_avg[t];
    float* __restrict__ V = a.exp_avg_sq[t];
    const float step_size = a.lr / a.bias_correction1[t], bc2 = a.bias_correction2_sqrt[t];
    const bool vec = ((((size_t)P) | ((size_t)Gr) | ((size_t)M) | ((size_t)V)) & 15) == 0;
    int64_t i = e0 + (int64_t)threadIdx.x * 4;
    if (vec) {
        for (; i + 3 < e1; i += 256 * 4) {
            f32x4 p = *reinterpret_cast<const f32x4*>(P + i), m = *reinterpret_cast<const f32x4*>(M + i), v = *reinterpret_cast<const f32x4*>(V + i);
            const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Gr + i));
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_one(pe, g[e], me, ve, a, step_size, bc2);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            *reinterpret_cast<f32x4*>(P + i) = p;
            *reinterpret_cast<f32x4*>(M + i) = m;
            *reinterpret_cast<f32x4*>(V + i) = v;
        }
        // the chunk's ragged end (only the last chunk of a tensor whose length is not a multiple of 4): lane-strided scalars
        const int64_t done = e0 + ((e1 - e0) & ~(int64_t)3);
        for (int64_t j = done + threadIdx.x; j < e1; j += 256) {
            float p = P[j], m = M[j], v = V[j];
            adam_one(p, Gr[j], m, v, a, step_size, bc2);
            P[j] = p; M[j] = m; V[j] = v;
        }
        return;
    }
    for (int64_t j = e0 + threadIdx.x; j < e1; j += 256) {
        float p = P[j], m = M[j], v = V[j];
        adam_one(p, Gr[j], m, v, a, step_size, bc2);
        P[j] = p; M[j] = m; V[j] = v;
    }
}

}  // namespace ag

using namespace ag;

extern "C" {

size_t ag_adam_args_bytes(void) { return sizeof(AgAdamArgs); }

int ag_adam_step(const AgAdamArgs* a, void* stream)
{
    if (!a || a->n < 1 || a->n > AG_ADAM_MAX_TENSORS) { set_error("ag_adam_step: bad tensor count"); return AG_ERR_INVALID_ARGUMENT; }
    AdamLaunch L;
    L.a = *a;
    long long chunks = 0;
    for (int i = 0; i < a->n; i++) {
        if (!a->param[i] || !a->grad[i] || !a->exp_avg[i] || !a->exp_avg_sq[i] || a->numel[i] < 0) { set_error("ag_adam_step: null tensor / negative length"); return AG_ERR_INVALID_ARGUMENT; }
        if (!(a->bias_correction1[i] > 0.f) || !(a->bias_correction2_sqrt[i] > 0.f)) { set_error("ag_adam_step: bias corrections must be positive"); return AG_ERR_INVALID_ARGUMENT; }
        L.chunk_begin[i] = (int32_t)chunks;
        chunks += (a->numel[i] + kAdamChunk - 1) / kAdamChunk;
        if (chunks > 0x7fffffffLL) { set_error("ag_adam_step: too many elements in one call"); return AG_ERR_INVALID_ARGUMENT; }
    }
    L.chunk_begin[a->n] = (int32_t)chunks;
    if (chunks == 0) return AG_OK;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), L);
    return check_hip(hipGetLastError(), "adam_kernel");
}

}  // extern "C"
