// How many other instructions does a v_mfma_f32_32x32x16_bf16 stream hide?  Every wave loops over 12 MFMAs (two accumulators
// alternating, as the split convolution kernels issue them) with F filler instructions spread evenly between them; 1, 2 or 4 waves per
// SIMD.  Output: SIMD cycles per MFMA (32 = the matrix pipe is never idle).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: VALU (v_add_u32), 1: SALU (s_add_u32), 2: ds_read_b128 (conflict-free), 3: mixed VALU / SALU alternating, 4: ds_write_b64, 5: s_waitcnt
__device__ int g_random_operands = 0;      // 1: operands with random mantissas / signs (switching activity of real data)

template <int F, int KIND>
__global__ void __launch_bounds__(256) fill_kernel(int iters, float* out)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + (threadIdx.x & 7) * 0.125f); b[i] = (__bf16)0.5f; }
    if (g_random_operands) {
        uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            h = h * 1664525u + 1013904223u;
            a[i] = (__bf16)(((int)(h >> 8 & 0xffff) - 32768) * (1.0f / 32768.f));
            h = h * 1664525u + 1013904223u;
            b[i] = (__bf16)(((int)(h >> 8 & 0xffff) - 32768) * (1.0f / 32768.f));
        }
    }
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    uint32_t v[8] = { threadIdx.x, 1, 2, 3, 4, 5, 6, 7 };
    uint32_t s0 = blockIdx.x, s1 = 3;
    const float4* lp = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
    float4 lacc = { 0, 0, 0, 0 };
    double lacc2 = threadIdx.x;
    constexpr int per = F / 12;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 12; m++) {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else       acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < per; f++) {
                const int k = (m * per + f) & 7;
                if (KIND == 0 || (KIND == 3 && (f & 1) == 0)) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v[k]));
                else if (KIND == 1 || KIND == 3) { if (f & 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc"); else asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc"); }
                else if (KIND == 4) { asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(size_t)lp), "v"(lacc2) : "memory"); }
                else if (KIND == 5) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                else { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((uint32_t)(size_t)lp)); lacc.x += t.x; }
            }
        }
    }
    float r = lacc.x;
#pragma unroll
    for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
    uint32_t u = s0 + s1;
#pragma unroll
    for (int i = 0; i < 8; i++) u += v[i];
    if (r == 12345.678f || u == 0x7fffffffu) out[0] = r;
}

template <int F, int KIND>
static void run(float* out, const char* kind)
{
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;                  // 256-thread blocks = one wave per SIMD each; wps blocks per CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((fill_kernel<F, KIND>), dim3(blocks), dim3(256), 0, 0, 100, out);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<F, KIND>), dim3(blocks), dim3(256), 0, 0, iters, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (double)iters * 12 * wps;
        printf("%-6s fillers/MFMA %5.2f  waves/SIMD %d : %6.1f ns per MFMA per SIMD = %5.1f cycles @2.1 GHz\n", kind, F / 12.0, wps,
               ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.1);
    }
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 64);
    run<0, 0>(out, "none");
    { int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random_operands), &one, sizeof(int)); }
    printf("random operands:\n");
    run<0, 0>(out, "none");
    run<48, 0>(out, "valu");
    { int zero = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random_operands), &zero, sizeof(int)); }
    printf("constant operands:\n");
    run<72, 0>(out, "valu");
    run<48, 1>(out, "salu"); run<96, 1>(out, "salu"); run<144, 1>(out, "salu");
    run<96, 3>(out, "mixed"); run<144, 3>(out, "mixed"); run<192, 3>(out, "mixed");
    run<12, 4>(out, "ldswr");
    return 0;
}
